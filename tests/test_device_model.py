"""Product-side reset sampler (host, vectorised) vs the oracle's literal scalar
sampler and Maxwell / ground-truth restatement: identical draws -> identical
devices, parameter blocks and initial states."""
import numpy as np
import pytest

import qd_oracle as O
from qadapt_hip import device_model as DM
from qadapt_hip.layout import layout


def _cfgs():
    return DM.load_yaml(None, "qarray_config.yaml"), DM.load_yaml(None, "env_config.yaml")


@pytest.mark.parametrize("N", [2, 3, 4, 6, 8])
def test_sampler_matches_oracle_draw_for_draw(N):
    q, e = _cfgs()
    samp = DM.DeviceSampler(N, q, e)
    seeds = [1234 + i for i in range(3)]
    u = np.stack([np.random.default_rng(s).random(samp.n_draws) for s in seeds])
    eb = samp.build(u)
    L = layout(N); G = N + 1; nb = N - 1
    for k, s in enumerate(seeds):
        rng = np.random.default_rng(s)
        so = O.sample_episode(rng, N)
        # the oracle consumed exactly n_draws uniforms
        assert rng.random() == np.random.default_rng(s).random(samp.n_draws + 1)[-1]
        for name in ("Cdd", "Cgd", "Cds", "Cgs", "Cbd", "Cbg", "Cbs", "Cbb"):
            assert np.array_equal(eb.extras[name][k], so[name]), name
        assert eb.extras["tc_base"][k] == so["tc_base"] and np.array_equal(eb.extras["alpha"][k], so["alpha"])
        assert eb.extras["coulomb_peak_width"][k] == so["coulomb_peak_width"]
        assert np.array_equal(eb.extras["p_inter"][k], so["latching"]["p_inter"])
        dev = O.device_from_sample(so)
        P = eb.params[k]
        assert np.allclose(P[L.cdd_inv:L.cdd_inv + G * G].reshape(G, G), dev.cdd_inv_full, rtol=1e-13, atol=1e-15)
        assert np.array_equal(P[L.cgd:L.cgd + G * 2 * N].reshape(G, 2 * N), dev.cgd_full)
        assert np.array_equal(P[L.cbg:L.cbg + nb * G].reshape(nb, G), dev.Cbg)
        U = P[L.ufac:L.ufac + N * N].reshape(N, N)
        assert np.allclose(np.triu(U), U) and np.allclose(U @ U.T, dev.cdd_inv_full[:N, :N], rtol=1e-12)
        # whole reset against the oracle env
        env = O.OracleEnv(N, 4)
        env.reset(so, np.zeros((nb, 3)), np.full((nb, 3), 5.0))      # log-var 5: every update rejected
        S = eb.state[k]
        assert np.allclose(P[L.pmin:L.pmin + N], env.plunger_min, rtol=1e-12)
        assert np.allclose(P[L.bmax:L.bmax + nb], env.barrier_max, rtol=1e-12)
        assert np.allclose(S[L.s_gate_v:L.s_gate_v + N], env.gate_v, rtol=1e-12)
        assert np.allclose(S[L.s_barrier_v:L.s_barrier_v + nb], env.barrier_v, rtol=1e-12)
        assert np.allclose(S[L.s_gate_gt:L.s_gate_gt + N], env.gate_gt, rtol=1e-6)
        assert np.allclose(S[L.s_barrier_gt:L.s_barrier_gt + nb], env.barrier_gt, rtol=1e-6)
        assert np.isclose(S[L.s_sensor_gt], env.sensor_gt, rtol=1e-9)
        assert P[L.scal + 2] == env.window


def test_reference_yaml_files_parse_to_same_plan():
    # our bundled YAMLs carry the reference's keys; a user's copy must work unchanged
    q, e = _cfgs()
    p = DM.make_draw_plan(4, q, e)
    assert len(p.lo) == DM.DeviceSampler(4, q, e).n_draws
    assert p.slices["window_delta"] == slice(0, 1)


def test_missing_config_raises_filenotfound():
    with pytest.raises(FileNotFoundError):
        DM.load_yaml("/nonexistent/env.yaml", "env_config.yaml")


def test_solver_options_that_would_change_the_numbers_are_refused():
    """`latched_model.use_sparse` selects a 50-step float32 Lanczos approximation in the reference; accepting the
    flag and returning the exact answer would silently differ, so it is refused (as is a kept-state count other than 32)."""
    import pytest
    from qadapt_hip import device_model as DM
    q = DM.load_yaml(None, "qarray_config.yaml")
    DM.check_solver_options(q)                                         # the shipped defaults pass
    q["simulator"]["latched_model"]["charge_state_batch_size"] = 250   # chunking only: fine
    DM.check_solver_options(q)
    q["simulator"]["latched_model"]["use_sparse"] = True
    with pytest.raises(NotImplementedError, match="use_sparse"):
        DM.check_solver_options(q)
    q["simulator"]["latched_model"]["use_sparse"] = False
    q["simulator"]["latched_model"]["num_charge_states"] = 64
    with pytest.raises(NotImplementedError, match="32"):
        DM.check_solver_options(q)
