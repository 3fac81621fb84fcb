"""Row f3: batched device-side frame stacking vs the literal list-based restatement of the
reference's RLlib connector (custom_frame_stacking.py).  Pure torch: runs in the CPU tier; the
GPU test drives it from the real env.  RLlib itself is absent, so the connector class cannot be
executed here -- parity is against the restatement in oracle/qd_oracle.py (unpinned beyond it)."""
import numpy as np
import pytest
import torch

import qd_oracle as O
from qadapt_hip import spaces
from qadapt_hip.frame_stacking import BatchedFrameStacking, stack_episode, stacked_observation_space


def _fake_obs(rng, B, N, R):
    return {"plunger_images": torch.as_tensor(rng.random((B, N, R, R, 2)).astype(np.float32)),
            "obs_gate_voltages": torch.as_tensor(rng.uniform(-1, 1, (B, N)).astype(np.float32)),
            "barrier_images": torch.as_tensor(rng.random((B, N - 1, R, R, 1)).astype(np.float32)),
            "obs_barrier_voltages": torch.as_tensor(rng.uniform(-1, 1, (B, N - 1)).astype(np.float32))}


@pytest.mark.parametrize("F", [1, 3, 5])
def test_env_to_module_matches_literal_lists(F):
    rng = np.random.default_rng(F)
    B, N, R = 3, 4, 6
    fs = BatchedFrameStacking(B, N, R, F, device="cpu")
    hist = [[[] for _ in range(N)] for _ in range(B)]            # per env, per plunger: list of obs dicts
    resets = {0: [0], 4: [1], 5: [1, 2], 9: [0, 1, 2]}           # step -> envs starting a new episode
    for step in range(12):
        obs = _fake_obs(rng, B, N, R)
        mask = np.zeros(B, bool)
        for e in resets.get(step, []):
            mask[e] = True
            for i in range(N):
                hist[e][i] = []
        out = fs.push(obs, reset_mask=torch.as_tensor(mask) if mask.any() else None)
        for e in range(B):
            for i in range(N):
                hist[e][i].append({"image": obs["plunger_images"][e, i].numpy(),
                                   "voltage": obs["obs_gate_voltages"][e, i:i + 1].numpy()})
                want = O.frame_stack_env_to_module(hist[e][i], F)
                got = fs.agent_view(out, e, f"plunger_{i}")
                assert got["image"].shape == (F, R, R, 2)
                assert np.array_equal(got["image"], want["image"])
                assert np.array_equal(got["voltage"], want["voltage"]) and got["voltage"].dtype == np.float32
                assert np.array_equal(got["attention_mask"], want["attention_mask"]) and got["attention_mask"].dtype == np.int8
            b = fs.agent_view(out, e, "barrier_1")                # barrier agents: unchanged
            assert np.array_equal(b["image"], obs["barrier_images"][e, 1].numpy()) and "attention_mask" not in b
            assert np.array_equal(b["voltage"], obs["obs_barrier_voltages"][e, 1:2].numpy())


@pytest.mark.parametrize("T,F", [(1, 4), (3, 4), (7, 3), (10, 1)])
def test_learner_windows_match_literal(T, F):
    rng = np.random.default_rng(10 * T + F)
    images = rng.random((T, 5, 5, 2)).astype(np.float32); voltages = rng.uniform(-1, 1, (T, 1)).astype(np.float32)
    want = O.frame_stack_learner(images, voltages, F)
    got = stack_episode(images, voltages, F)
    assert np.array_equal(got["image"].numpy(), want["image"])
    assert np.array_equal(got["voltage"].numpy(), want["voltage"])
    assert np.array_equal(got["attention_mask"].numpy(), want["attention_mask"])
    # the last window of the learner pipeline is the env-to-module stack of the last step
    lst = [{"image": images[t], "voltage": voltages[t]} for t in range(T)]
    e2m = O.frame_stack_env_to_module(lst, F)
    assert np.array_equal(got["image"][-1].numpy(), e2m["image"]) and np.array_equal(got["attention_mask"][-1].numpy(), e2m["attention_mask"])


def test_learner_windows_with_lookback():
    rng = np.random.default_rng(3)
    F, T = 4, 5
    full_i = rng.random((T + 2, 3, 3, 2)).astype(np.float32); full_v = rng.uniform(-1, 1, (T + 2, 1)).astype(np.float32)
    got = stack_episode(full_i[2:], full_v[2:], F, lookback_images=full_i[:2], lookback_voltages=full_v[:2])
    # two real look-back frames: only one padded frame remains, in the first window
    assert got["attention_mask"].numpy().tolist()[0] == [1, 0, 0, 0] and got["attention_mask"][1:].sum() == 0
    assert np.array_equal(got["image"][0, 1:3].numpy(), full_i[:2]) and np.array_equal(got["image"][0, 3].numpy(), full_i[2])


def test_stacked_spaces():
    pl = spaces.Dict({"image": spaces.Box(0.0, 1.0, (8, 8, 2), np.float32), "voltage": spaces.Box(-1.0, 1.0, (1,), np.float32)})
    ba = spaces.Dict({"image": spaces.Box(0.0, 1.0, (8, 8, 1), np.float32), "voltage": spaces.Box(-1.0, 1.0, (1,), np.float32)})
    s = stacked_observation_space(pl, 4)
    assert s["image"].shape == (4, 8, 8, 2) and s["voltage"].shape == (4,) and s["attention_mask"].shape == (4,)
    assert s["attention_mask"].dtype == np.int8
    assert stacked_observation_space(ba, 4) is ba


@pytest.mark.gpu
def test_frame_stacking_on_the_real_env():
    from qadapt_hip.vec_env import VecQuantumDeviceEnv, SyntheticCapacitanceModel
    B, N, R, F = 4, 4, 16, 3
    env = VecQuantumDeviceEnv(B, num_dots=N, resolution=R, seed=1234, capacitance_model=SyntheticCapacitanceModel(2))
    fs = BatchedFrameStacking(B, N, R, F, device=env.device)
    frames = []
    out = fs.push(env.reset(), reset_mask=torch.ones(B, dtype=torch.bool))
    frames.append(env.plunger_images.clone())
    assert out["attention_mask"][0, 0].tolist() == [1, 1, 0]
    rng = np.random.default_rng(0)
    for t in range(3):
        obs, *_ = env.step(torch.as_tensor(rng.uniform(-0.1, 0.1, (B, 2 * N - 1)).astype(np.float32)).cuda())
        out = fs.push(obs)
        frames.append(env.plunger_images.clone())
    assert int(out["attention_mask"].sum()) == 0
    for f in range(F):
        assert torch.equal(out["image"][:, :, f], frames[len(frames) - F + f])
    assert out["image"].is_cuda and out["image"].shape == (B, N, F, R, R, 2)
    env.close()


def test_connector_hands_over_device_stacks_per_policy_module():
    """The ConnectorV2-shaped env-to-module wrapper: one tensor per policy module, identical to the reference's
    per-agent host stacks (oracle restatement), without consulting episode objects."""
    from qadapt_hip.frame_stacking import DeviceFrameStackingConnector
    rng = np.random.default_rng(4)
    B, N, R, F, T = 2, 3, 5, 3, 4
    fs = BatchedFrameStacking(B, N, R, F, device="cpu")
    feed = {}
    pl = spaces.Dict({"image": spaces.Box(0.0, 1.0, (R, R, 2), np.float32), "voltage": spaces.Box(-1.0, 1.0, (1,), np.float32)})
    ba = spaces.Dict({"image": spaces.Box(0.0, 1.0, (R, R, 1), np.float32), "voltage": spaces.Box(-1.0, 1.0, (1,), np.float32)})
    conn = DeviceFrameStackingConnector(spaces.Dict({"plunger_0": pl, "barrier_0": ba}), None, stacker=fs,
                                        source=lambda: (feed["obs"], feed["reset"]))
    assert conn.observation_space["plunger_0"]["image"].shape == (F, R, R, 2) and conn.observation_space["barrier_0"] is ba
    hist = []
    for t in range(T):
        obs = {"plunger_images": torch.as_tensor(rng.random((B, N, R, R, 2)).astype(np.float32)),
               "barrier_images": torch.as_tensor(rng.random((B, N - 1, R, R, 1)).astype(np.float32)),
               "obs_gate_voltages": torch.as_tensor(rng.uniform(-1, 1, (B, N)).astype(np.float32)),
               "obs_barrier_voltages": torch.as_tensor(rng.uniform(-1, 1, (B, N - 1)).astype(np.float32))}
        feed["obs"] = obs; feed["reset"] = torch.tensor([t == 0, t == 0 or t == 2])      # env 1 starts a new episode at t = 2
        hist.append(obs)
        batch = conn(rl_module=None, batch={}, episodes=[], explore=True, shared_data={})
    p = batch["obs"]["plunger_policy"]; b = batch["obs"]["barrier_policy"]
    assert p["image"].shape == (B * N, F, R, R, 2) and b["image"].shape == (B * (N - 1), R, R, 1) and b["voltage"].shape == (B * (N - 1), 1)
    for env, start in ((0, 0), (1, 2)):
        for i in range(N):
            lst = [{"image": hist[t]["plunger_images"][env, i].numpy(), "voltage": hist[t]["obs_gate_voltages"][env, i:i + 1].numpy()}
                   for t in range(start, T)]
            ref = O.frame_stack_env_to_module(lst, F)
            k = env * N + i
            assert np.array_equal(p["image"][k].numpy(), ref["image"]) and np.array_equal(p["voltage"][k].numpy(), ref["voltage"])
            assert np.array_equal(p["attention_mask"][k].numpy(), ref["attention_mask"])
    assert torch.equal(b["image"][1 * (N - 1) + 1], hist[-1]["barrier_images"][1, 1])
