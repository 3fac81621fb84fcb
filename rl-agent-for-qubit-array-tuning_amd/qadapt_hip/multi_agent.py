"""
MultiAgentEnvWrapper -- mirror of the reference's RLlib wrapper
(src/qadapt/environment/multi_agent_wrapper.py:27-584) over the HIP-backed env:
same constructor arguments, agent ids, channel assignment, per-agent
observation/action spaces, reset()/step() return structure and error behaviour,
so `register_env("qarray_multiagent_env", ...)` in the reference's train.py
(:355-362) can return this class instead.

Subclasses ray.rllib's MultiAgentEnv when ray is importable; otherwise a plain
object with the same attributes (ray is absent in the build container).
The GIF / distance-history side logging of the reference (:587-895) is tooling,
not part of the simulation path, and is not built; those arguments are accepted
and ignored.
"""
from __future__ import annotations

from typing import Dict

import numpy as np

from . import spaces

try:                                            # pragma: no cover
    from ray.rllib.env.multi_agent_env import MultiAgentEnv as _Base
except Exception:
    class _Base:                                # ray absent: same surface, no RLlib registration
        def __init__(self):
            pass


class MultiAgentEnvWrapper(_Base):
    def __init__(self, training: bool = True, return_voltage: bool = False, return_global_state: bool = False,
                 gif_config: dict = None, distance_data_dir: str = None, env_config_path: str = None,
                 capacitance_model_checkpoint: str = None, is_collecting_data: bool = False,
                 base_env_class=None, **base_env_kwargs):
        super().__init__()
        if return_global_state and not return_voltage:
            raise ValueError("return_global_state=True requires return_voltage=True (the global "
                             "state extends the per-agent dict observation).")
        self.return_voltage = return_voltage
        self.return_global_state = return_global_state
        self.distance_data_dir = distance_data_dir
        self.is_collecting_data = is_collecting_data
        self.gif_config = gif_config
        if base_env_class is None:
            from .env import QuantumDeviceEnv
            base_env_class = QuantumDeviceEnv
        kw = dict(training=training, capacitance_model_checkpoint=capacitance_model_checkpoint, **base_env_kwargs)
        if env_config_path:
            kw["config_path"] = env_config_path
        self.base_env = base_env_class(**kw)

        self.num_gates = self.base_env.num_dots
        self.use_barriers = self.base_env.use_barriers
        self.num_barriers = self.base_env.num_dots - 1
        self.num_image_channels = self.base_env.num_dots - 1
        self.gate_agent_ids = [f"plunger_{i}" for i in range(self.num_gates)]
        self.barrier_agent_ids = [f"barrier_{i}" for i in range(self.num_barriers)]
        self.all_agent_ids = self.gate_agent_ids + self.barrier_agent_ids
        self._setup_channel_assignments()
        self.base_observation_space = self.base_env.observation_space
        self.base_action_space = self.base_env.action_space
        self._create_agent_spaces(self.base_observation_space, self.base_action_space)

    # multi_agent_wrapper.py:147-178
    def _setup_channel_assignments(self):
        self.agent_channel_map = {}
        for agent_id in self.gate_agent_ids:
            i = int(agent_id.split("_")[1])
            if i == 0:
                self.agent_channel_map[agent_id] = [0, 0]
            elif i == self.num_gates - 1:
                last = self.num_gates - 2
                self.agent_channel_map[agent_id] = [last, last]
            else:
                self.agent_channel_map[agent_id] = [i - 1, i]
        for agent_id in self.barrier_agent_ids:
            self.agent_channel_map[agent_id] = [int(agent_id.split("_")[1])]

    # multi_agent_wrapper.py:180-309
    def _create_agent_spaces(self, base_obs, base_action):
        image_shape = base_obs["image"].shape
        gate_low = float(np.min(base_action["action_gate_voltages"].low))
        gate_high = float(np.max(base_action["action_gate_voltages"].high))
        barrier_low = float(np.min(base_action["action_barrier_voltages"].low))
        barrier_high = float(np.max(base_action["action_barrier_voltages"].high))
        obs_spaces, act_spaces = {}, {}
        H, W = image_shape[0], image_shape[1]
        if self.return_voltage:
            n_glob = self.num_gates + self.num_barriers
            gimg = spaces.Box(low=0.0, high=1.0, shape=(H, W, self.num_image_channels), dtype=np.float32)
            gvol = spaces.Box(low=min(gate_low, barrier_low), high=max(gate_high, barrier_high),
                              shape=(n_glob,), dtype=np.float32)
        for ids, nch, lo, hi in ((self.gate_agent_ids, 2, gate_low, gate_high),
                                 (self.barrier_agent_ids, 1, barrier_low, barrier_high)):
            for agent_id in ids:
                img = spaces.Box(low=0.0, high=1.0, shape=(H, W, nch), dtype=np.float32)
                if self.return_voltage:
                    d = {"image": img, "voltage": spaces.Box(low=lo, high=hi, shape=(1,), dtype=np.float32)}
                    if self.return_global_state:
                        d["global_image"] = gimg
                        d["global_voltages"] = gvol
                    obs_spaces[agent_id] = spaces.Dict(d)
                else:
                    obs_spaces[agent_id] = img
                act_spaces[agent_id] = spaces.Box(low=lo, high=hi, shape=(1,), dtype=np.float32)
        self.observation_spaces = spaces.Dict(obs_spaces)
        self.action_spaces = spaces.Dict(act_spaces)
        self._agent_ids = set(self.all_agent_ids)
        self.observation_space = self.observation_spaces
        self.action_space = self.action_spaces
        self.agents = self._agent_ids.copy()
        self.possible_agents = self._agent_ids.copy()

    # multi_agent_wrapper.py:311-383
    def _extract_agent_observation(self, global_obs: Dict[str, np.ndarray], agent_id: str, device_state_info=None):
        channels = self.agent_channel_map[agent_id]
        global_image = global_obs["image"]
        if len(channels) == 2:
            agent_idx = int(agent_id.split("_")[1])
            img1 = global_image[:, :, channels[0]]
            img2 = global_image[:, :, channels[1]]
            if agent_idx == 0:
                agent_image = np.stack([img1, img2], axis=2)
            elif agent_idx == self.num_gates - 1:
                agent_image = np.stack([img1.T, img2.T], axis=2)
            else:
                agent_image = np.stack([img1, img2.T], axis=2)
        else:
            agent_image = global_image[:, :, channels[0]:channels[0] + 1]
        if not self.return_voltage:
            return agent_image.astype(np.float32)
        agent_idx = int(agent_id.split("_")[1])
        if "plunger" in agent_id:
            voltage = global_obs["obs_gate_voltages"][agent_idx:agent_idx + 1]
        else:
            voltage = global_obs["obs_barrier_voltages"][agent_idx:agent_idx + 1]
        agent_obs = {"image": agent_image.astype(np.float32), "voltage": voltage.astype(np.float32)}
        if self.return_global_state:
            agent_obs["global_image"] = global_obs["image"].astype(np.float32)
            agent_obs["global_voltages"] = np.concatenate(
                [global_obs["obs_gate_voltages"], global_obs["obs_barrier_voltages"]]).astype(np.float32)
        return agent_obs

    # multi_agent_wrapper.py:386-425
    def _combine_agent_actions(self, agent_actions):
        gate_actions = np.zeros(self.num_gates, dtype=np.float32)
        barrier_actions = np.zeros(self.num_barriers, dtype=np.float32)
        for ids, out in ((self.gate_agent_ids, gate_actions), (self.barrier_agent_ids, barrier_actions)):
            for agent_id in ids:
                if agent_id in agent_actions:
                    i = int(agent_id.split("_")[1])
                    v = agent_actions[agent_id]
                    out[i] = float(v[0]) if hasattr(v, "__len__") else float(v)
        return {"action_gate_voltages": gate_actions, "action_barrier_voltages": barrier_actions}

    # multi_agent_wrapper.py:427-457
    def _distribute_rewards(self, global_rewards):
        agent_rewards = {}
        if "gates" not in global_rewards:
            raise ValueError("Missing gate rewards in global_rewards")
        for agent_id in self.gate_agent_ids:
            agent_rewards[agent_id] = float(global_rewards["gates"][int(agent_id.split("_")[1])])
        if "barriers" not in global_rewards:
            raise ValueError("Missing barrier rewards in global_rewards")
        for agent_id in self.barrier_agent_ids:
            agent_rewards[agent_id] = float(global_rewards["barriers"][int(agent_id.split("_")[1])])
        return agent_rewards

    # multi_agent_wrapper.py:459-483
    def reset(self, *, seed=None, options=None):
        global_obs, global_info = self.base_env.reset(seed=seed, options=options)
        obs = {a: self._extract_agent_observation(global_obs, a, None) for a in self.all_agent_ids}
        infos = {a: global_info for a in self.all_agent_ids}
        return obs, infos

    # multi_agent_wrapper.py:485-584
    def step(self, agent_actions):
        assert len(agent_actions) == len(self.all_agent_ids), "Agent actions must match the number of agents"
        assert all(a in self.all_agent_ids for a in agent_actions.keys()), "Unknown agent IDs in actions"
        global_action = self._combine_agent_actions(agent_actions)
        global_obs, global_rewards, terminated, truncated, info = self.base_env.step(global_action)
        device_state_info = info.get("current_device_state", None)
        obs = {a: self._extract_agent_observation(global_obs, a, device_state_info) for a in self.all_agent_ids}
        rewards = self._distribute_rewards(global_rewards)
        term = dict.fromkeys(self.all_agent_ids, terminated); term["__all__"] = terminated
        trunc = dict.fromkeys(self.all_agent_ids, truncated); trunc["__all__"] = truncated
        if not device_state_info:
            infos = {a: {} for a in self.all_agent_ids}
        else:
            try:
                infos = {}
                for idx, a in enumerate(self.gate_agent_ids):
                    infos[a] = {"ground_truth": device_state_info["gate_ground_truth"][idx],
                                "current_voltage": device_state_info["current_gate_voltages"][idx]}
                for idx, a in enumerate(self.barrier_agent_ids):
                    infos[a] = {"ground_truth": device_state_info["barrier_ground_truth"][idx],
                                "current_voltage": device_state_info["current_barrier_voltages"][idx]}
            except Exception as e:
                raise RuntimeError(f"Error creating multi-agent info: {e}")
        return obs, rewards, term, trunc, infos

    def close(self):
        if hasattr(self.base_env, "close"):
            self.base_env.close()

    def get_agent_ids(self):
        return self._agent_ids
