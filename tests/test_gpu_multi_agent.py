"""H6 on the GPU: B logical multi-agent envs over one libqdsim handle -- one launch set per env-step whatever B is,
per-agent observations identical to the tensors the kernels wrote; and checkpoint / restore of the whole env state
(device blocks, host generators, Philox observation counter) with the stochastic stages on."""
import numpy as np
import pytest
import yaml

from qadapt_hip import device_model as DM

pytestmark = pytest.mark.gpu


class _CountingLib:
    """Proxy over the ctypes library that counts the entry points a step may use."""

    def __init__(self, lib):
        object.__setattr__(self, "_lib", lib)
        object.__setattr__(self, "counts", {})

    def __getattr__(self, name):
        fn = getattr(self._lib, name)
        if not name.startswith("qd_"):
            return fn

        def wrapped(*a):
            self.counts[name] = self.counts.get(name, 0) + 1
            return fn(*a)
        return wrapped


def test_64_views_one_launch_set_per_step():
    import torch
    from qadapt_hip.multi_agent import BatchedMultiAgentEnv
    from qadapt_hip.vec_env import SyntheticCapacitanceModel
    B, N, R = 64, 4, 16
    env = BatchedMultiAgentEnv(B, num_dots=N, resolution=R, seed=11, capacitance_model=SyntheticCapacitanceModel(5),
                               return_voltage=True, return_global_state=True)
    vec = env.vec
    views = env.views
    obs0, infos0 = views[0].reset()                      # first reset of any view resets the batch once
    obs7, _ = views[7].reset()
    assert obs0["plunger_0"]["image"].shape == (R, R, 2) and "current_device_state" in infos0["plunger_0"]
    vec._lib = _CountingLib(vec._lib)
    ids = env.roster.ids
    rng = np.random.default_rng(0)
    for step in range(3):
        acts = [{a: rng.uniform(-1, 1, 1).astype(np.float32) for a in ids} for _ in range(B)]
        for b in range(B):
            views[b].stage(acts[b])
        c = dict(vec._lib.counts)
        # one launch set: actions + observe + capacitance update, once each (not B times)
        assert c.get("qd_apply_actions", 0) == step + 1 and c.get("qd_observe", 0) == step + 1, c
        assert c.get("qd_update_capacitance", 0) == step + 1 and env.launches == step + 1
        pim = vec.plunger_images.cpu().numpy(); bim = vec.barrier_images.cpu().numpy(); vol = vec.voltages.cpu().numpy()
        rew = vec.rewards.cpu().numpy(); gim = vec.global_image.cpu().numpy()
        for b in (0, 13, 63):
            o, r, term, trunc, info = views[b].collect()
            for i in range(N):
                assert np.array_equal(o[f"plunger_{i}"]["image"], pim[b, i]) and o[f"plunger_{i}"]["voltage"][0] == vol[b, i]
            for j in range(N - 1):
                assert np.array_equal(o[f"barrier_{j}"]["image"], bim[b, j]) and r[f"barrier_{j}"] == rew[b, N + j]
            assert np.array_equal(o["barrier_0"]["global_image"], gim[b]) and np.array_equal(o["barrier_0"]["global_voltages"], vol[b])
            assert set(info["plunger_0"]) == {"ground_truth", "current_voltage"} and not term["__all__"]
        packed = np.stack([np.concatenate([acts[b][a] for a in ids]) for b in range(B)])
        assert np.array_equal(vec._keep[0].cpu().numpy(), packed)                 # all B action sets went into ONE tensor
        for b in range(B):
            if b not in (0, 13, 63):
                views[b].collect()
    env.close()


def test_checkpoint_restores_bit_identical_trajectory_with_noise(tmp_path):
    import torch
    from qadapt_hip.vec_env import VecQuantumDeviceEnv, SyntheticCapacitanceModel
    cfg = DM.load_yaml(None, "env_config.yaml")
    cfg["simulator"]["max_steps"] = 4                      # a reset (host generators!) falls inside the replayed window
    p = tmp_path / "env.yaml"; p.write_text(yaml.safe_dump(cfg))
    N, R, B = 4, 16, 3

    def make():
        return VecQuantumDeviceEnv(B, num_dots=N, resolution=R, config_path=str(p), seed=321, noise=["sensor", "radial", "latch"],
                                   capacitance_model=SyntheticCapacitanceModel(8))
    rng = np.random.default_rng(3)
    acts = [torch.as_tensor(rng.uniform(-0.3, 0.3, (B, 2 * N - 1)).astype(np.float32)).cuda() for _ in range(6)]
    a = make(); a.reset()
    st, steps = a.get_state(); steps[:] = [0, 1, 2]; a.set_state(st, steps)      # staggered: resets at different steps
    for k in range(2):
        a.step(acts[k], auto_reset=True)
    ck = a.get_checkpoint()
    want = []
    for k in range(2, 6):
        obs, rew, _, trunc = a.step(acts[k], auto_reset=True)
        want.append((obs["image"].cpu().numpy().copy(), rew.cpu().numpy().copy(), trunc.cpu().numpy().copy(), a.get_state()[0].copy()))
    assert any(w[2].any() for w in want), "the window was meant to contain a truncation + reset"
    a.close()
    b = make(); b.reset()                                   # a fresh process would do exactly this, then restore
    b.set_checkpoint(ck)
    for k, w in zip(range(2, 6), want):
        obs, rew, _, trunc = b.step(acts[k], auto_reset=True)
        assert np.array_equal(obs["image"].cpu().numpy(), w[0]), k
        assert np.array_equal(rew.cpu().numpy(), w[1]) and np.array_equal(trunc.cpu().numpy(), w[2])
        assert np.array_equal(b.get_state()[0], w[3])
    with pytest.raises(ValueError):
        c = VecQuantumDeviceEnv(B, num_dots=N, resolution=R, config_path=str(p), seed=322, capacitance_model=SyntheticCapacitanceModel(8))
        c.set_checkpoint(ck)
    b.close()


def test_default_seed_gives_different_devices_and_offsets_are_global():
    """unseeded envs must not all sample the same device; reset(seed=s) on shards with offsets must
    reproduce one big env."""
    from qadapt_hip.vec_env import VecQuantumDeviceEnv, SyntheticCapacitanceModel
    kw = dict(num_dots=2, resolution=8, capacitance_model=SyntheticCapacitanceModel(1))
    a = VecQuantumDeviceEnv(1, **kw); b = VecQuantumDeviceEnv(1, **kw)
    a.reset(); b.reset()
    assert a.seed != b.seed and not np.array_equal(a._params_host, b._params_host)
    a.close(); b.close()
    one = VecQuantumDeviceEnv(4, seed=5, **kw)
    lo = VecQuantumDeviceEnv(2, seed=5, env_id_offset=0, **kw); hi = VecQuantumDeviceEnv(2, seed=5, env_id_offset=2, **kw)
    for e in (one, lo, hi):
        e.reset(seed=77)
    assert np.array_equal(one._params_host, np.concatenate([lo._params_host, hi._params_host]))
    assert np.array_equal(one.global_image.cpu().numpy(), np.concatenate([lo.global_image.cpu().numpy(), hi.global_image.cpu().numpy()]))
    for e in (one, lo, hi):
        e.close()
