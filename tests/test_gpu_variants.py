"""GPU parity of the env_config.yaml variants of rows a2/a3/a19 (reward curves, sparse reward, use_deltas,
update_method direct / perfect / null, legacy nearest_neighbour mode) against the oracle env, through the C ABI.
The updaters themselves are pinned to the reference classes by tests/golden/updater_variants.npz."""
import numpy as np
import pytest
import yaml

import qd_oracle as O
from qadapt_hip import device_model as DM

pytestmark = pytest.mark.gpu

VARIANTS = {
    "polynomial": dict(reward=dict(gate_curve_type="polynomial", gate_curve_exponent=2.5)),
    "exponential": dict(reward=dict(gate_curve_type="exponential", gate_curve_exponent=1.7)),
    "linear": dict(reward=dict(gate_curve_type="linear")),
    "sparse": dict(reward=dict(sparse_reward=True, plunger_radius=3, outer_plunger_radius=15,
                               outer_plunger_reward_max=0.4, barrier_radius=2.5)),
    "deltas": dict(simulator=dict(use_deltas=True, delta_max=4.0)),
    "direct": dict(capacitance_model=dict(update_method="direct")),
    "direct_nn": dict(capacitance_model=dict(update_method="direct", nearest_neighbour=True)),
    "kalman_nn": dict(capacitance_model=dict(nearest_neighbour=True)),
    "perfect": dict(capacitance_model=dict(update_method="perfect")),
    "none": dict(capacitance_model=dict(update_method=None)),
}


def _write_cfg(tmp_path, over):
    cfg = DM.load_yaml(None, "env_config.yaml")
    for sec, kv in over.items():
        cfg[sec].update(kv)
    p = tmp_path / "env.yaml"
    p.write_text(yaml.safe_dump(cfg))
    return str(p), cfg


@pytest.mark.parametrize("name", sorted(VARIANTS))
def test_config_variant_matches_oracle_env(tmp_path, name):
    import torch
    from qadapt_hip.vec_env import VecQuantumDeviceEnv, SyntheticCapacitanceModel
    N, R, B, seed = 4, 8, 3, 555
    C = N - 1
    path, cfg = _write_cfg(tmp_path, VARIANTS[name])
    cm = cfg["capacitance_model"]; rw = cfg["reward"]; sim = cfg["simulator"]
    K = 2 if cm["nearest_neighbour"] else 3
    env = VecQuantumDeviceEnv(B, num_dots=N, resolution=R, config_path=path, seed=seed, validate=True,
                              capacitance_model=SyntheticCapacitanceModel(3, outputs=K))
    rng = np.random.default_rng(9)

    def cnn():
        v = rng.normal(0, 0.1, (B, C, K)).astype(np.float32); lv = rng.uniform(-6, -2, (B, C, K)).astype(np.float32)
        t = None if cm["update_method"] in (None, "perfect") else (torch.as_tensor(v).cuda(), torch.as_tensor(lv).cuda())
        return v, lv, t

    v0, l0, t0 = cnn()
    env.reset(cnn_outputs=t0)
    rcfg = dict(gate_ramp_start=rw["gate_ramp_start"], gate_quadratic_start=rw["gate_quadratic_start"],
                barrier_ramp_start=rw["barrier_ramp_start"], gate_curve_type=rw["gate_curve_type"],
                gate_curve_exponent=rw["gate_curve_exponent"], sparse_reward=rw["sparse_reward"],
                plunger_radius=rw["plunger_radius"], outer_plunger_radius=rw["outer_plunger_radius"],
                outer_plunger_reward_max=rw["outer_plunger_reward_max"], barrier_radius=rw["barrier_radius"])
    oenvs = []
    for e in range(B):
        oe = O.OracleEnv(N, R, update_method=cm["update_method"], nearest_neighbour=cm["nearest_neighbour"],
                         use_deltas=sim["use_deltas"], delta_max=sim["delta_max"], reward_cfg=rcfg)
        oe.reset(O.sample_episode(np.random.default_rng(seed + e), N), v0[e], l0[e])
        oenvs.append(oe)
    ds = env.device_state()
    for e, oe in enumerate(oenvs):
        assert np.allclose(ds["virtual_gate_matrix"][e], oe.vgm, rtol=1e-8, atol=1e-10), name
        assert np.allclose(ds["gate_ground_truth"][e], oe.gate_gt, rtol=2e-6, atol=1e-6)
    L = env.L
    hit = np.zeros(3, bool)                                  # reward regions seen (zero / ramp / inner)
    for step in range(5):
        # aim at assorted distances from the ground truth so that every reward region is exercised
        P = env._params_host
        gt = np.concatenate([ds["gate_ground_truth"], ds["barrier_ground_truth"]], axis=1).astype(np.float64)
        lo = np.concatenate([P[:, L.pmin:L.pmin + N], P[:, L.bmin:L.bmin + C]], axis=1)
        hi = np.concatenate([P[:, L.pmax:L.pmax + N], P[:, L.bmax:L.bmax + C]], axis=1)
        span = rng.choice([0.3, 2.0, 12.0, 45.0], size=gt.shape)
        want = gt + rng.uniform(-1, 1, gt.shape) * span
        act = np.clip(2 * (want - lo) / (hi - lo) - 1, -1.2, 1.2).astype(np.float32)
        if sim["use_deltas"]:
            act[:, :N] = rng.uniform(-1.1, 1.1, (B, N)).astype(np.float32)
        v, l, t = cnn()
        obs, rew, term, trunc = env.step(torch.as_tensor(act).cuda(), cnn_outputs=t)
        ds = env.device_state()
        for e, oe in enumerate(oenvs):
            _, (gr, br), _, otrunc = oe.step(act[e, :N], act[e, N:], v[e], l[e])
            assert np.allclose(ds["current_gate_voltages"][e], oe.gate_v, rtol=1e-13, atol=0), (name, step)
            assert np.allclose(ds["current_barrier_voltages"][e], oe.barrier_v, rtol=1e-13)
            r = rew[e].cpu().numpy()
            assert np.allclose(r[:N], gr, rtol=1e-12, atol=1e-14), (name, step, r[:N], gr)
            assert np.allclose(r[N:], br, rtol=1e-12, atol=1e-14), (name, step)
            hit |= np.array([(gr == 0).any(), ((gr > 0) & (gr < 0.5)).any(), (gr >= 0.5).any()])
            assert np.allclose(ds["kalman_means"][e], oe.kalman.means, rtol=1e-12, atol=1e-15), (name, step)
            assert np.allclose(ds["kalman_variances"][e], oe.kalman.vars, rtol=1e-12, atol=1e-15)
            assert np.allclose(ds["virtual_gate_matrix"][e], oe.vgm, rtol=1e-8, atol=1e-10)
            assert np.allclose(ds["gate_ground_truth"][e], oe.gate_gt, rtol=2e-6, atol=1e-6)
            assert bool(trunc[e]) == otrunc
    if not sim["use_deltas"]:
        assert hit[1:].all() and (hit[0] or not rw["sparse_reward"]), hit
    env.close()


def test_rejected_variants(tmp_path):
    from qadapt_hip.vec_env import VecQuantumDeviceEnv
    for over, exc in ((dict(capacitance_model=dict(update_method="fake")), ValueError),
                      (dict(capacitance_model=dict(update_method="ema")), NotImplementedError),
                      (dict(capacitance_model=dict(update_method="nonsense")), ValueError),
                      (dict(reward=dict(gate_curve_type="cubic")), ValueError),
                      (dict(simulator=dict(use_barriers=False)), NotImplementedError)):
        path, _ = _write_cfg(tmp_path, over)
        with pytest.raises(exc):
            VecQuantumDeviceEnv(2, num_dots=2, resolution=8, config_path=path, capacitance_model=lambda x: None)
