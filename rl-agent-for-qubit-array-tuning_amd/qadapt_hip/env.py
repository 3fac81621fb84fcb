"""
QuantumDeviceEnv -- single-environment, NumPy-in / NumPy-out mirror of the
reference class of the same name (src/qadapt/environment/env.py:29-896), backed
by the batched HIP library with a batch of one.  It has the constructor signature
the reference's MultiAgentEnvWrapper expects from `base_env_class`
(multi_agent_wrapper.py:93-106): (training, config_path,
capacitance_model_checkpoint), plus .reset / .step / .close,
.observation_space / .action_space, .num_dots, .use_barriers and
.array.model.cgd_full, so it can be dropped into the reference wrapper or into
qadapt_hip.multi_agent.MultiAgentEnvWrapper unchanged.
"""
from __future__ import annotations

import types

import numpy as np

from . import spaces
from .device_model import load_yaml


class QuantumDeviceEnv:
    metadata = {"render_modes": []}

    def __init__(self, training=True, config_path="env_config.yaml", num_dots=None, use_barriers=None,
                 capacitance_model_checkpoint=None, capacitance_model=None, backend=None, seed=None,
                 qarray_config_path=None):
        self.config = load_yaml(config_path if config_path != "env_config.yaml" else None, "env_config.yaml")
        sim = self.config["simulator"]
        self.training = training
        self.num_dots = num_dots if num_dots is not None else sim["num_dots"]
        self.use_barriers = use_barriers if use_barriers is not None else sim["use_barriers"]
        self.capacitance_model_checkpoint = capacitance_model_checkpoint
        self.use_deltas = sim["use_deltas"]
        self.max_steps = sim["max_steps"]
        self.num_plunger_voltages = self.num_dots
        self.num_barrier_voltages = self.num_dots - 1
        self.resolution = sim["resolution"]
        if not self.use_barriers:
            raise NotImplementedError("env.py only supports barrier mode for now")      # env.py:61-62
        N, C, R = self.num_dots, self.num_dots - 1, self.resolution
        self.action_space = spaces.Dict({
            "action_gate_voltages": spaces.Box(low=-1.0, high=1.0, shape=(N,), dtype=np.float32),
            "action_barrier_voltages": spaces.Box(low=-1.0, high=1.0, shape=(C,), dtype=np.float32)})
        self.obs_channels = C
        self.observation_space = spaces.Dict({
            "image": spaces.Box(low=0.0, high=1.0, shape=(R, R, C), dtype=np.float32),
            "obs_gate_voltages": spaces.Box(low=-1.0, high=1.0, shape=(N,), dtype=np.float32),
            "obs_barrier_voltages": spaces.Box(low=-1.0, high=1.0, shape=(C,), dtype=np.float32)})
        update_method = self.config["capacitance_model"]["update_method"]
        try:                                                   # env.py:680-802: same exception type
            if update_method in (None, "perfect", "fake"):                 # env.py:683-689: no CNN
                capacitance_model = None
            elif capacitance_model is None:
                if not capacitance_model_checkpoint:
                    raise ValueError("Capacitance model weights must be provided via capacitance_model_checkpoint "
                                     f"when using update_method '{update_method}'.")
                # env.py:716-749: CapacitancePredictionModel(output_size) + checkpoint, on the GPU
                from .capacitance_cnn import build_device_model
                nearest = self.config["capacitance_model"].get("nearest_neighbour", False)
                capacitance_model = build_device_model(checkpoint=capacitance_model_checkpoint,
                                                       output_size=2 if nearest else 3)
        except Exception as e:
            raise RuntimeError(f"Error initialising capacitance model: {e}")
        if backend is None:
            from .vec_env import VecQuantumDeviceEnv
            backend = VecQuantumDeviceEnv(1, num_dots=N, config_path=config_path if config_path != "env_config.yaml" else None,
                                          qarray_config_path=qarray_config_path, resolution=R,
                                          capacitance_model=capacitance_model,
                                          seed=seed)            # None: fresh entropy per env, as the reference's unseeded RNGs
        self._b = backend
        self.current_step = 0
        self.array = types.SimpleNamespace(model=types.SimpleNamespace(cgd_full=None), barrier_alpha=None,
                                           gate_ground_truth=None)
        self.reset()

    # -- helpers ---------------------------------------------------------------
    @staticmethod
    def _np(x):
        return x.detach().cpu().numpy() if hasattr(x, "detach") else np.asarray(x)

    def _observation(self, obs):
        return {"image": self._np(obs["image"])[0].copy(),
                "obs_gate_voltages": self._np(obs["obs_gate_voltages"])[0].copy(),
                "obs_barrier_voltages": self._np(obs["obs_barrier_voltages"])[0].copy()}

    def _refresh_device_state(self):
        ds = self._b.device_state()
        self.device_state = {
            "gate_ground_truth": ds["gate_ground_truth"][0], "barrier_ground_truth": ds["barrier_ground_truth"][0],
            "sensor_ground_truth": float(ds["sensor_ground_truth"][0]),
            "current_gate_voltages": ds["current_gate_voltages"][0],
            "current_barrier_voltages": ds["current_barrier_voltages"][0],
            "virtual_gate_matrix": ds["virtual_gate_matrix"][0], "virtual_gate_origin": ds["virtual_gate_origin"][0]}
        self.array.gate_ground_truth = self.device_state["gate_ground_truth"]
        ep = getattr(self._b, "last_episode", None)
        if ep is not None:
            self.array.model.cgd_full = ep.extras["cgd"][0]
            self.array.barrier_alpha = ep.extras["alpha"][0]

    def _get_info(self):
        return {"current_device_state": self.device_state}

    # -- gym API ----------------------------------------------------------------
    def reset(self, seed=None, options=None):
        self.current_step = 0
        obs = self._b.reset(seed=seed)
        self._refresh_device_state()
        return self._observation(obs), self._get_info()

    def step(self, action, skip_obs=False):
        if skip_obs:
            raise NotImplementedError("skip_obs=True has no user in the reference and is not built")
        self.current_step += 1
        g = np.array(action["action_gate_voltages"]).flatten().astype(np.float32)
        b = np.array(action["action_barrier_voltages"]).flatten().astype(np.float32)
        act = np.concatenate([g, b])[None, :]
        obs, rewards, terminated, truncated = self._b.step(act)
        r = self._np(rewards)[0]
        self._refresh_device_state()
        reward = {"gates": r[:self.num_dots].copy(), "barriers": r[self.num_dots:].copy()}
        return (self._observation(obs), reward, bool(self._np(terminated)[0]), bool(self._np(truncated)[0]),
                self._get_info())

    def close(self):
        if hasattr(self._b, "close"):
            self._b.close()

    def _cleanup(self):
        pass
