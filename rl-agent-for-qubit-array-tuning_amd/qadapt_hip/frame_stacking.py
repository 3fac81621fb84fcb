"""
Frame stacking for the batched env -- the device-side counterpart of the
reference's RLlib connector `CustomFrameStacking`
(src/qadapt/training/utils/custom_frame_stacking.py:19-297).

The reference stacks, per plunger agent, the last `num_frames` observations of the running
episode: images `(H,W,2) -> (num_frames,H,W,2)`, voltages `(1,) -> (num_frames,)`, most recent
frame LAST, zero frames prepended while the episode is younger than `num_frames`, and an int8
`attention_mask` that is 1 on the padded frames (:203-241).  Barrier agents are passed through
unchanged (:187-198).  RLlib does this on the host from per-env Python lists; here the history of
all B envs lives in one tensor on the GPU, next to the observation tensors libqdsim writes, and a
step costs one shifted copy -- no host round trip, no per-agent Python objects.

  * `BatchedFrameStacking`    env-to-module pipeline (:184-249) for VecQuantumDeviceEnv observations
  * `stack_episode`           learner pipeline (:70-182): sliding windows over one recorded episode
  * `stacked_observation_space`  `_convert_individual_space` (:251-294)

Pure torch (plumbing, no custom kernel: at B = 4096, N = 8, F = 4, 64x64 the shifted copy moves
~4 GB, a few milliseconds next to a 0.7 s simulation step).  Runs on any torch device, so the
parity tests against the literal list-based restatement run in the CPU tier.
"""
from __future__ import annotations

from typing import Dict, Optional

import numpy as np
import torch

from . import spaces


class BatchedFrameStacking:
    """History of the last `num_frames` plunger observations of every env.

    push(obs, reset_mask=None) -> {"image": (B,N,F,R,R,2) f32, "voltage": (B,N,F) f32,
                                   "attention_mask": (B,N,F) int8, "barrier_images": passthrough,
                                   "barrier_voltage": (B,N-1) f32}
    `obs` is the dict VecQuantumDeviceEnv returns.  `reset_mask` (B,) bool marks envs whose
    `obs` is the FIRST observation of a new episode: their history is cleared first, exactly as a
    fresh RLlib episode has no earlier observations.  The returned tensors are views of the
    internal buffers, valid until the next push."""

    def __init__(self, num_envs: int, num_dots: int, resolution: int, num_frames: int, device="cuda"):
        if num_frames < 1:
            raise ValueError("num_frames must be >= 1")
        self.B, self.N, self.R, self.F = int(num_envs), int(num_dots), int(resolution), int(num_frames)
        self.device = torch.device(device)
        B, N, R, F = self.B, self.N, self.R, self.F
        self.images = torch.zeros((B, N, F, R, R, 2), dtype=torch.float32, device=self.device)
        self.voltages = torch.zeros((B, N, F), dtype=torch.float32, device=self.device)
        self.count = torch.zeros((B,), dtype=torch.int32, device=self.device)      # real frames held, capped at F
        self._slot = torch.arange(F, device=self.device, dtype=torch.int32)

    def reset(self, env_ids=None):
        """Forget the history of the listed envs (all if None)."""
        if env_ids is None:
            self.images.zero_(); self.voltages.zero_(); self.count.zero_()
        else:
            ids = torch.as_tensor(env_ids, device=self.device, dtype=torch.long)
            self.images[ids] = 0; self.voltages[ids] = 0; self.count[ids] = 0

    @torch.no_grad()
    def push(self, obs: Dict[str, torch.Tensor], reset_mask: Optional[torch.Tensor] = None):
        B, N, F = self.B, self.N, self.F
        img = obs["plunger_images"].to(self.device)                    # (B,N,R,R,2)
        vol = obs["obs_gate_voltages"].to(self.device).reshape(B, N)   # the plunger agent's own voltage (:135)
        if reset_mask is not None:
            m = torch.as_tensor(reset_mask, device=self.device, dtype=torch.bool)
            if bool(m.any()):
                self.reset(torch.nonzero(m).reshape(-1))
        if F > 1:                                                      # oldest frame drops out, most recent LAST
            self.images[:, :, :-1] = self.images[:, :, 1:].clone()
            self.voltages[:, :, :-1] = self.voltages[:, :, 1:].clone()
        self.images[:, :, -1] = img
        self.voltages[:, :, -1] = vol
        self.count = torch.clamp(self.count + 1, max=F)
        # padding sits at the FRONT: slots [0, F - count) are padding -> mask 1   (:232-236)
        mask = (self._slot[None, :] < (F - self.count)[:, None]).to(torch.int8)          # (B,F)
        out = {"image": self.images, "voltage": self.voltages,
               "attention_mask": mask[:, None, :].expand(B, N, F)}
        if "barrier_images" in obs:
            out["barrier_images"] = obs["barrier_images"]
        if "obs_barrier_voltages" in obs:
            out["barrier_voltage"] = obs["obs_barrier_voltages"]
        return out

    def agent_view(self, stacked, env: int, agent_id: str):
        """One agent's observation of one env in the reference's per-agent format (host NumPy)."""
        idx = int(agent_id.split("_")[1])
        if "barrier" in agent_id.lower():                              # unchanged (:187-198)
            return {"image": stacked["barrier_images"][env, idx].cpu().numpy(),
                    "voltage": stacked["barrier_voltage"][env, idx:idx + 1].cpu().numpy()}
        return {"image": stacked["image"][env, idx].cpu().numpy(),
                "voltage": stacked["voltage"][env, idx].cpu().numpy(),
                "attention_mask": stacked["attention_mask"][env, idx].cpu().numpy()}


@torch.no_grad()
def stack_episode(images, voltages, num_frames: int, lookback_images=None, lookback_voltages=None):
    """Learner pipeline (:90-182): all sliding windows of one plunger agent's episode.

    images (T,H,W,C), voltages (T,1) or (T,): the episode's observations (without the final one, as
    the connector receives them).  `lookback_*`: up to num_frames-1 earlier observations RLlib may hold
    in the episode's look-back buffer (:93-97); missing history is zero-padded at the front.
    Returns image (T,F,H,W,C), voltage (T,F), attention_mask (T,F) int8 (1 = padding)."""
    images = torch.as_tensor(images); voltages = torch.as_tensor(voltages).reshape(images.shape[0])
    T = images.shape[0]; F = int(num_frames)
    if lookback_images is not None and len(lookback_images):
        lb_i = torch.as_tensor(lookback_images)[-(F - 1):] if F > 1 else torch.as_tensor(lookback_images)[:0]
        lb_v = torch.as_tensor(lookback_voltages).reshape(-1)[-(F - 1):] if F > 1 else torch.as_tensor(lookback_voltages).reshape(-1)[:0]
        images = torch.cat([lb_i.to(images.dtype), images]); voltages = torch.cat([lb_v.to(voltages.dtype), voltages])
    actual = images.shape[0]
    required = T + F - 1
    pad = max(required - actual, 0)
    if pad:
        images = torch.cat([torch.zeros((pad,) + tuple(images.shape[1:]), dtype=images.dtype, device=images.device), images])
        voltages = torch.cat([torch.zeros((pad,), dtype=voltages.dtype, device=voltages.device), voltages])
    # windows [t, t+F) of the padded sequence
    win_i = images.unfold(0, F, 1)                                   # (T, H, W, C, F)
    win_i = win_i.permute(0, win_i.dim() - 1, *range(1, win_i.dim() - 1)).contiguous()
    win_v = voltages.unfold(0, F, 1).contiguous()                    # (T, F)
    t = torch.arange(T, device=images.device)[:, None]; f = torch.arange(F, device=images.device)[None, :]
    mask = ((t + f) < pad).to(torch.int8)
    return {"image": win_i, "voltage": win_v, "attention_mask": mask}


def stacked_observation_space(obs_space, num_frames: int):
    """`_convert_individual_space` (:251-294): plunger agents (2-channel images) get stacked spaces
    plus an attention mask; barrier agents (1 channel) keep their space."""
    assert isinstance(obs_space, spaces.Dict) and "image" in obs_space.spaces and "voltage" in obs_space.spaces, obs_space
    image_space = obs_space["image"]; voltage_space = obs_space["voltage"]
    num_channels = image_space.shape[-1]
    assert num_channels in [1, 2]
    if num_channels == 1:
        return obs_space
    return spaces.Dict({
        "image": spaces.Box(low=float(np.asarray(image_space.low).flat[0]), high=float(np.asarray(image_space.high).flat[0]),
                            shape=(num_frames,) + tuple(image_space.shape), dtype=image_space.dtype),
        "voltage": spaces.Box(low=float(np.asarray(voltage_space.low).flat[0]), high=float(np.asarray(voltage_space.high).flat[0]),
                              shape=(num_frames,), dtype=voltage_space.dtype),
        "attention_mask": spaces.Box(low=0, high=1, shape=(num_frames,), dtype=np.int8),
    })


# ---------------------------------------------------------------------------------------------------------
# RLlib ConnectorV2 wrapper (env-to-module).  The reference registers `CustomFrameStackingEnvToModule`
# (custom_frame_stacking.py:297, train.py:497-539), which rebuilds the stack on the host from each agent's episode
# object.  This connector hands the policy forward the stack that already sits on the GPU: one tensor per policy
# module, all B envs and all agents of that kind stacked on the batch axis, no host copy, no per-agent Python objects.
# ---------------------------------------------------------------------------------------------------------
try:                                            # pragma: no cover - ray is absent in the build container
    from ray.rllib.connectors.connector_v2 import ConnectorV2 as _ConnectorBase
    HAVE_RLLIB = True
except Exception:
    HAVE_RLLIB = False

    class _ConnectorBase:                       # same constructor / call shape as ConnectorV2
        def __init__(self, input_observation_space=None, input_action_space=None, **kwargs):
            self.input_observation_space = input_observation_space
            self.input_action_space = input_action_space
            self.observation_space = self.recompute_output_observation_space(input_observation_space, input_action_space) \
                if input_observation_space is not None else None

        def recompute_output_observation_space(self, input_observation_space, input_action_space):
            return input_observation_space


class DeviceFrameStackingConnector(_ConnectorBase):
    """env-to-module connector over a BatchedFrameStacking.  `source()` returns the observation dict of the batched
    env for the step being processed (VecQuantumDeviceEnv.step's first return value) and optionally the mask of envs
    that have just been reset; policy modules are addressed by the reference's mapping rule -- agent ids containing
    "plunger" / "barrier" (training/utils/policy_mapping.py:14-17)."""

    def __init__(self, input_observation_space=None, input_action_space=None, *, stacker: BatchedFrameStacking = None,
                 source=None, plunger_module="plunger_policy", barrier_module="barrier_policy", **kwargs):
        self.stacker, self.source = stacker, source
        self.num_frames = stacker.F if stacker is not None else kwargs.pop("num_frames", 1)
        self.plunger_module, self.barrier_module = plunger_module, barrier_module
        super().__init__(input_observation_space, input_action_space, **kwargs)

    def recompute_output_observation_space(self, input_observation_space, input_action_space):
        """Per-agent spaces -> stacked spaces for plunger agents, unchanged for barrier agents (:251-294)."""
        if input_observation_space is None:
            return None
        return spaces.Dict({a: stacked_observation_space(s, self.num_frames) for a, s in input_observation_space.items()})

    def __call__(self, *, rl_module=None, batch=None, episodes=None, explore=None, shared_data=None, **kwargs):
        """Fills batch["obs"][module] with DEVICE tensors, agents of one kind flattened onto the batch axis in
        (env, agent) order:
            plunger module: image (B*N, F, R, R, 2), voltage (B*N, F), attention_mask (B*N, F) int8
            barrier module: image (B*(N-1), R, R, 1), voltage (B*(N-1), 1)
        `episodes` is not consulted: the history lives in the stacker."""
        got = self.source()
        obs, reset_mask = got if isinstance(got, tuple) else (got, None)
        st = self.stacker.push(obs, reset_mask=reset_mask)
        B, N, F = self.stacker.B, self.stacker.N, self.stacker.F
        batch = {} if batch is None else batch
        out = batch.setdefault("obs", {})
        out[self.plunger_module] = {"image": st["image"].reshape((B * N, F) + tuple(st["image"].shape[3:])),
                                    "voltage": st["voltage"].reshape(B * N, F),
                                    "attention_mask": st["attention_mask"].reshape(B * N, F)}
        if "barrier_images" in st:
            bi = st["barrier_images"]
            out[self.barrier_module] = {"image": bi.reshape((B * (N - 1),) + tuple(bi.shape[2:])),
                                        "voltage": st["barrier_voltage"].reshape(B * (N - 1), 1)}
        return batch
