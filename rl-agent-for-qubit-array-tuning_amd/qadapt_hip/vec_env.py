"""
VecQuantumDeviceEnv -- B quantum-dot tuning environments stepped as one batch on
one MI355X.  Host side of the C-ABI in include/qdsim.h; observation / action
buffers are PyTorch-ROCm tensors, the computation is the HIP library.

Semantics per env are those of the reference's QuantumDeviceEnv
(src/qadapt/environment/env.py:135-315): reset() builds a new random device,
identity VGM, ground truth, voltage ranges, random start, first observation and
a Kalman/VGM update; step() rescales the action, pays the reward against the
PREVIOUS ground truth, renders the N-1 CSD channels, normalises them, updates
Kalman/VGM and then the ground truth.  The capacitance CNN (env.py:568-581) is
an input provider here: pass `capacitance_model=callable`, or
`update_method: null` in the env config, or per-step `cnn_outputs`.
"""
from __future__ import annotations

import ctypes

import numpy as np
import torch

from . import _lib
from .device_model import DeviceSampler, check_solver_options, load_yaml
from .layout import layout


class SyntheticCapacitanceModel:
    """Stand-in for the capacitance CNN used by benchmarks and tests
    (values ~ N(0, 0.1^2), log_vars ~ U(-6, -2)); deterministic
    per (seed, call index)."""

    def __init__(self, seed=99, outputs=3):
        self.seed = seed
        self.calls = 0
        self.outputs = outputs

    def __call__(self, images):
        n = images.shape[0]
        g = torch.Generator(device="cpu").manual_seed(self.seed + self.calls)
        self.calls += 1
        values = torch.randn((n, self.outputs), generator=g, dtype=torch.float32) * 0.1
        log_vars = torch.rand((n, self.outputs), generator=g, dtype=torch.float32) * 4.0 - 6.0
        return values.to(images.device), log_vars.to(images.device)


class VecQuantumDeviceEnv:
    def __init__(self, num_envs, num_dots=None, config_path=None, qarray_config_path=None,
                 resolution=None, device=None, seed=None, env_id_offset=0, capacitance_model=None,
                 validate=False, env_chunk=0, reset_kalman_on_reset=False, noise=None,
                 vary_peak_width=False, peak_width_alpha=0.01, voltage_capacitance_model=None, pixel_search=False):
        """pixel_search: a9 by the per-pixel search only (A/B switch; the default runs one search per 8x8 tile).
        seed: base seed of the per-env device streams (PCG64(seed + global env id)) and the Philox key of
        the stochastic stages; None draws fresh OS entropy, as the reference's unseeded generators do
        (qarray_base_class.py:773-774, env.py:161).
        vary_peak_width / peak_width_alpha: QarrayBaseClass ctor arguments (qarray_base_class.py:42-43).
        voltage_capacitance_model: overrides `simulator.voltage_capacitance_model.type` of the qarray
        config (None keeps the file's value; "linear" or "none")."""
        if seed is None:
            seed = int(np.random.SeedSequence().entropy) & 0x7FFFFFFFFFFF      # 47 bits: seed + env id stays exact
        self.seed = int(seed)
        self.env_id_offset = int(env_id_offset)
        self.config = load_yaml(config_path, "env_config.yaml")
        self.qconfig = load_yaml(qarray_config_path, "qarray_config.yaml")
        check_solver_options(self.qconfig)
        if voltage_capacitance_model is not None:
            self.qconfig["simulator"]["voltage_capacitance_model"]["type"] = \
                None if voltage_capacitance_model in ("none", "null") else voltage_capacitance_model
        sim = self.config["simulator"]
        self.num_envs = int(num_envs)
        self.num_dots = int(num_dots if num_dots is not None else sim["num_dots"])
        self.use_barriers = bool(sim["use_barriers"])
        if not self.use_barriers:
            raise NotImplementedError("env.py only supports barrier mode for now")      # env.py:61-62
        rew = self.config["reward"]
        if rew.get("gate_curve_type", "constant") not in _lib.QD_CURVES:
            raise ValueError(f"Unknown curve type: {rew.get('gate_curve_type')}")            # env.py:441
        self.resolution = int(resolution if resolution is not None else sim["resolution"])
        self.max_steps = int(sim["max_steps"])
        self.update_method = self.config["capacitance_model"]["update_method"]
        if self.update_method in ("bayesian", "kriging", "ema"):                             # env.py:766-771
            raise NotImplementedError(f"update_method={self.update_method!r} requires an updater module the "
                                      "reference removed; use 'kalman' or 'direct'")
        if self.update_method == "fake":
            # env.py:555-559 hands fake_capacitance_model the (N, N+1) dot-only matrix, and
            # qarray_base_class.py:917-923 then stacks rows of N+nb+2 and N+nb+1 columns: the reference's own
            # barrier-mode path raises there, so there is no behaviour to reproduce.
            raise ValueError("update_method 'fake' fails in the reference's barrier mode (shape mismatch in "
                             "QarrayBaseClass._update_virtual_gate_matrix); not built")
        if self.update_method not in (None, "kalman", "direct", "perfect"):
            raise ValueError(f"Unknown update method: {self.update_method}")                 # env.py:788
        self.nearest_neighbour = bool(self.config["capacitance_model"].get("nearest_neighbour"))
        self.cnn_outputs = 2 if self.nearest_neighbour else 3
        if self.update_method in (None, "perfect"):
            capacitance_model = None                           # env.py:683-689: no CNN, no Kalman
        self.capacitance_model = capacitance_model
        if self.update_method in ("kalman", "direct") and capacitance_model is None:
            # same exception type as env.py:801-802
            raise RuntimeError("Error initialising capacitance model: update_method 'kalman' needs a "
                               "capacitance_model callable (images -> values, log_vars)")
        if not torch.cuda.is_available():
            raise RuntimeError("VecQuantumDeviceEnv needs a ROCm GPU (no CPU fallback)")
        self.device = torch.device(device if device is not None else "cuda")
        if self.device.index is None:                                  # "cuda" means the CURRENT device
            self.device = torch.device("cuda", torch.cuda.current_device())
        self.reset_kalman_on_reset = bool(reset_kalman_on_reset)
        N = self.num_dots; R = self.resolution; B = self.num_envs
        self.N, self.R, self.B, self.C = N, R, B, N - 1
        self.L = layout(N)
        self.sampler = DeviceSampler(N, self.qconfig, self.config, vary_peak_width=vary_peak_width,
                                     peak_width_alpha=peak_width_alpha,
                                     perfect_vgm=self.update_method == "perfect")
        self._rngs = [np.random.Generator(np.random.PCG64(self.seed + self.env_id_offset + e)) for e in range(B)]
        # ---- library handle ---------------------------------------------------
        self._lib = _lib.lib()
        cm = self.config["capacitance_model"]
        cfg = _lib.QdConfig(struct_size=ctypes.sizeof(_lib.QdConfig), n_dot=N, resolution=R, batch=B,
                            max_steps=self.max_steps, env_chunk=int(env_chunk),
                            flags=(_lib.QD_FLAG_VALIDATE if validate else 0) | (_lib.QD_FLAG_PIXEL_SEARCH if pixel_search else 0),
                            noise_flags=self._noise_flags(noise),
                            gate_ramp_start=float(rew["gate_ramp_start"]),
                            gate_quadratic_start=float(rew["gate_quadratic_start"]),
                            barrier_ramp_start=float(rew["barrier_ramp_start"]),
                            kalman_prior_mean=0.3, kalman_prior_variance=0.5, kalman_prior_mean_nnn=0.15,
                            kalman_variance_threshold=float(cm.get("variance_threshold", 0.05)),
                            kalman_process_noise=float(cm.get("process_noise", 0.0)),
                            rng_seed=self.seed & 0xFFFFFFFFFFFFFFFF, env_id_offset=self.env_id_offset,
                            use_deltas=1 if sim.get("use_deltas") else 0,
                            sparse_reward=1 if rew.get("sparse_reward") else 0,
                            gate_curve_type=_lib.QD_CURVES[rew.get("gate_curve_type", "constant")],
                            update_method=_lib.QD_UPDATE_DIRECT if self.update_method == "direct" else _lib.QD_UPDATE_KALMAN,
                            cnn_outputs=self.cnn_outputs, delta_max=float(sim.get("delta_max", 0.0)),
                            gate_curve_exponent=float(rew.get("gate_curve_exponent", 2.0)),
                            plunger_radius=float(rew.get("plunger_radius", 0.0)),
                            outer_plunger_radius=float(rew.get("outer_plunger_radius", 0.0)),
                            outer_plunger_reward_max=float(rew.get("outer_plunger_reward_max", 0.0)),
                            barrier_radius=float(rew.get("barrier_radius", 0.0)))
        self._h = ctypes.c_void_p()
        rc = self._lib.qd_create(ctypes.byref(cfg), self.device.index, ctypes.byref(self._h))
        if rc != 0:                          # a partially built handle carries the error text and must be released
            msg = self._lib.qd_last_error(self._h).decode() if self._h else "no handle"
            if self._h:
                self._lib.qd_destroy(self._h)
                self._h = None
            raise _lib.QdError(f"qd_create failed (code {rc}): {msg}")
        self.validate = bool(validate)
        # ---- caller-owned output tensors ---------------------------------------
        dev = self.device
        self.global_image = torch.zeros((B, R, R, self.C), dtype=torch.float32, device=dev)
        self.plunger_images = torch.zeros((B, N, R, R, 2), dtype=torch.float32, device=dev)
        self.barrier_images = torch.zeros((B, self.C, R, R, 1), dtype=torch.float32, device=dev)
        self.voltages = torch.zeros((B, 2 * N - 1), dtype=torch.float32, device=dev)
        self.rewards = torch.zeros((B, 2 * N - 1), dtype=torch.float64, device=dev)
        self.truncated = torch.zeros((B,), dtype=torch.uint8, device=dev)
        _lib.check(self._h, self._lib.qd_bind_outputs(self._h, self.global_image.data_ptr(),
                                                      self.plunger_images.data_ptr(),
                                                      self.barrier_images.data_ptr(),
                                                      self.voltages.data_ptr()), "qd_bind_outputs")
        self._params_host = np.zeros((B, self.L.size))
        self._steps_host = np.zeros(B, np.int64)      # host mirror of the device step counters (truncation is
        self._needs_reset = True                      # deterministic, so auto-reset needs no device read-back)
        self.obs_count = 0

    # ------------------------------------------------------------------ helpers
    def _noise_flags(self, noise):
        """noise=None/False: deterministic parity mode.  noise=True: what the configs enable
        (sensor white + telegraph noise always, radial noise if simulator.radial_noise.enabled).
        Or an iterable of {"sensor", "radial", "latch"}."""
        if not noise:
            return 0
        if noise is True:
            rn = self.config["simulator"].get("radial_noise") or {}
            lt = self.qconfig["simulator"]["model"].get("latching_model_parameters") or {}
            return (_lib.QD_NOISE_SENSOR | (_lib.QD_NOISE_RADIAL if rn.get("enabled") else 0)
                    | (_lib.QD_NOISE_LATCH if lt.get("Exists") else 0))
        f = 0
        for k in noise:
            f |= {"sensor": _lib.QD_NOISE_SENSOR, "radial": _lib.QD_NOISE_RADIAL, "latch": _lib.QD_NOISE_LATCH}[k]
        return f

    def _stream(self):
        return ctypes.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)

    def close(self):
        if getattr(self, "_h", None) is not None and self._h:
            self._lib.qd_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _obs(self):
        N = self.N
        return {"image": self.global_image, "obs_gate_voltages": self.voltages[:, :N],
                "obs_barrier_voltages": self.voltages[:, N:], "plunger_images": self.plunger_images,
                "barrier_images": self.barrier_images}

    def _cnn(self, env_ids=None):
        """Run the capacitance model on the current images of `env_ids` (all if None);
        returns full-batch (B,C,3) tensors as qd_update_capacitance expects."""
        if self.capacitance_model is None:
            return None, None
        img = self.barrier_images if env_ids is None else self.barrier_images[env_ids]
        n = img.shape[0]
        batch = img.reshape(n * self.C, 1, self.R, self.R)          # (C,1,R,R) per env, env.py:568-574
        values, log_vars = self.capacitance_model(batch)
        K = self.cnn_outputs
        values = values.to(torch.float32).reshape(n, self.C, K)
        log_vars = log_vars.to(torch.float32).reshape(n, self.C, K)
        if env_ids is None:
            return values.contiguous(), log_vars.contiguous()
        fv = torch.zeros((self.B, self.C, K), dtype=torch.float32, device=self.device)
        fl = torch.zeros_like(fv)
        fv[env_ids] = values; fl[env_ids] = log_vars
        return fv, fl

    # ------------------------------------------------------------------ reset
    def load_new_devices(self, env_ids=None, seed=None):
        """Sample new random devices for the listed envs and upload their parameter / initial
        state blocks (the device-construction half of reset(); no observation is rendered)."""
        ids = np.arange(self.B, dtype=np.int32) if env_ids is None else np.asarray(env_ids, dtype=np.int32).reshape(-1)
        if seed is not None:                 # keyed by GLOBAL env id, like the constructor's streams
            for e in ids:
                self._rngs[e] = np.random.Generator(np.random.PCG64(int(seed) + self.env_id_offset + int(e)))
        u = np.stack([self._rngs[e].random(self.sampler.n_draws) for e in ids])
        eb = self.sampler.build(u)
        self._params_host[ids] = eb.params
        self.last_episode = eb
        ip = ids.ctypes.data_as(ctypes.POINTER(ctypes.c_int32))
        rc = self._lib.qd_load_episodes(self._h, ip, int(ids.size), eb.params.ctypes.data, eb.state.ctypes.data,
                                        1 if self.reset_kalman_on_reset else 0, self._stream())
        _lib.check(self._h, rc, "qd_load_episodes")
        self._steps_host[ids] = 0
        self._needs_reset = False
        return eb

    def observe(self, env_ids_dev=None, n=0):
        """Render the observation of the current state (qd_observe) without stepping."""
        idp = None if env_ids_dev is None else ctypes.c_void_p(env_ids_dev.data_ptr())
        _lib.check(self._h, self._lib.qd_observe(self._h, idp, int(n), self._stream()), "qd_observe")
        return self._obs()

    def reset(self, env_ids=None, seed=None, options=None, cnn_outputs=None):
        """Reset the listed envs (all if None).  Returns the observation dict of
        the whole batch (device tensors, valid until the next call)."""
        ids = np.arange(self.B, dtype=np.int32) if env_ids is None else np.asarray(env_ids, dtype=np.int32).reshape(-1)
        if ids.size == 0:
            return self._obs()
        self.load_new_devices(ids, seed=seed)
        all_envs = ids.size == self.B and np.array_equal(ids, np.arange(self.B))
        # (pinned + non_blocking: a pageable upload would make the host wait for everything queued on the stream before it)
        ids_dev = None if all_envs else torch.from_numpy(np.ascontiguousarray(ids, dtype=np.int32)).pin_memory().to(self.device, non_blocking=True)
        idp = None if all_envs else ctypes.c_void_p(ids_dev.data_ptr())
        _lib.check(self._h, self._lib.qd_observe(self._h, idp, int(ids.size), self._stream()), "qd_observe")
        if cnn_outputs is not None:
            values, log_vars = cnn_outputs
        else:
            values, log_vars = self._cnn(None if all_envs else ids_dev.long())
        if values is not None:
            values = values.contiguous(); log_vars = log_vars.contiguous()
            rc = self._lib.qd_update_capacitance(self._h, idp, int(ids.size), values.data_ptr(), log_vars.data_ptr(),
                                                 0, self._stream())             # reset does not refresh the ground truth (env.py:233)
            _lib.check(self._h, rc, "qd_update_capacitance")
            self._keep = (values, log_vars, ids_dev)
        self._needs_reset = False
        return self._obs()

    # ------------------------------------------------------------------ step
    def step(self, actions, cnn_outputs=None, auto_reset=False):
        """actions: (B, 2N-1) float32 tensor (gates then barriers) or the reference's
        dict {"action_gate_voltages": (B,N), "action_barrier_voltages": (B,N-1)}.
        Returns (obs, rewards (B,2N-1) float64, terminated (B) bool, truncated (B) bool)."""
        if self._needs_reset:
            raise RuntimeError("step() called before reset()")
        if isinstance(actions, dict):
            actions = torch.cat([torch.as_tensor(actions["action_gate_voltages"]),
                                 torch.as_tensor(actions["action_barrier_voltages"])], dim=-1)
        actions = torch.as_tensor(actions, dtype=torch.float32, device=self.device).reshape(self.B, 2 * self.N - 1).contiguous()
        st = self._stream()
        if cnn_outputs is not None:
            values, log_vars = (t.to(torch.float32).contiguous() for t in cnn_outputs)
            rc = self._lib.qd_step(self._h, actions.data_ptr(), values.data_ptr(), log_vars.data_ptr(),
                                   self.rewards.data_ptr(), self.truncated.data_ptr(), st)
            _lib.check(self._h, rc, "qd_step")
            self._keep = (actions, values, log_vars)
        else:
            _lib.check(self._h, self._lib.qd_apply_actions(self._h, actions.data_ptr(), self.rewards.data_ptr(),
                                                           self.truncated.data_ptr(), st), "qd_apply_actions")
            _lib.check(self._h, self._lib.qd_observe(self._h, None, 0, st), "qd_observe")
            values, log_vars = self._cnn(None)
            vp = values.data_ptr() if values is not None else None
            lp = log_vars.data_ptr() if log_vars is not None else None
            _lib.check(self._h, self._lib.qd_update_capacitance(self._h, None, 0, vp, lp, 1, st),
                       "qd_update_capacitance")
            self._keep = (actions, values, log_vars)
        self._steps_host += 1
        truncated = self.truncated.bool()
        terminated = torch.zeros_like(truncated)
        obs = self._obs()
        if auto_reset:
            # the step counter alone decides truncation (env.py:281-285), so the host knows which envs are
            # done without reading the device: sampling the new devices below overlaps the kernels launched above
            done = np.nonzero(self._steps_host >= self.max_steps)[0]
            if done.size:
                truncated = truncated.clone()             # self.truncated is rewritten by the next step only, but be explicit
                obs = self.reset(env_ids=done.astype(np.int32))
        return obs, self.rewards, terminated, truncated

    # ------------------------------------------------------------------ state access
    def get_state(self):
        """Host copy of the per-env state blocks and step counters (checkpointing,
        infos, validation)."""
        st = np.zeros((self.B, self.L.s_size)); steps = np.zeros(self.B, np.int32)
        _lib.check(self._h, self._lib.qd_get_state(self._h, st.ctypes.data, steps.ctypes.data), "qd_get_state")
        return st, steps

    def set_state(self, state, steps):
        state = np.ascontiguousarray(state, dtype=np.float64); steps = np.ascontiguousarray(steps, dtype=np.int32)
        _lib.check(self._h, self._lib.qd_set_state(self._h, state.ctypes.data, steps.ctypes.data), "qd_set_state")
        self._steps_host[:] = steps

    def stagger_episodes(self):
        """Spread the episode phases (env e is put at step e mod max_steps) so that a steady 1/max_steps of
        the batch truncates and resets at every step -- what a long-running sampler looks like."""
        st, _ = self.get_state()
        self.set_state(st, (np.arange(self.B) + self.env_id_offset) % self.max_steps)

    # ------------------------------------------------------------------ checkpoint
    def get_checkpoint(self):
        """Everything needed to continue bit-identically (SURVEY 5 "expose RNG seeds/counters"): device state and
        step counters, the current devices' parameter blocks, every env's host generator state, the Philox
        observation counter of the stochastic stages and the synthetic CNN's call counter if one is used."""
        st, steps = self.get_state()
        ser = ctypes.c_uint64(0)
        _lib.check(self._h, self._lib.qd_get_rng_state(self._h, ctypes.byref(ser)), "qd_get_rng_state")
        return {"state": st, "steps": steps, "params": self._params_host.copy(),
                "rng": [g.bit_generator.state for g in self._rngs], "obs_serial": int(ser.value),
                "seed": self.seed, "env_id_offset": self.env_id_offset,
                "capacitance_model_calls": getattr(self.capacitance_model, "calls", None)}

    def set_checkpoint(self, ck):
        if ck["params"].shape != self._params_host.shape:
            raise ValueError("checkpoint belongs to an env of a different shape")
        if (ck["seed"], ck["env_id_offset"]) != (self.seed, self.env_id_offset):
            raise ValueError("checkpoint was taken with another seed / env_id_offset (the Philox key differs)")
        ids = np.arange(self.B, dtype=np.int32)
        params = np.ascontiguousarray(ck["params"], dtype=np.float64)
        state = np.ascontiguousarray(ck["state"], dtype=np.float64)
        rc = self._lib.qd_load_episodes(self._h, ids.ctypes.data_as(ctypes.POINTER(ctypes.c_int32)), self.B,
                                        params.ctypes.data, state.ctypes.data, 0, self._stream())
        _lib.check(self._h, rc, "qd_load_episodes")
        self._params_host[:] = params
        self.set_state(state, ck["steps"])
        for g, s_ in zip(self._rngs, ck["rng"]):
            g.bit_generator.state = s_
        _lib.check(self._h, self._lib.qd_set_rng_state(self._h, ctypes.c_uint64(ck["obs_serial"])), "qd_set_rng_state")
        if ck.get("capacitance_model_calls") is not None and hasattr(self.capacitance_model, "calls"):
            self.capacitance_model.calls = ck["capacitance_model_calls"]
        self._needs_reset = False

    def device_state(self):
        """The reference's info["current_device_state"] for every env (env.py:214-222)."""
        st, steps = self.get_state()
        L, N, G = self.L, self.N, self.N + 1
        return {"gate_ground_truth": st[:, L.s_gate_gt:L.s_gate_gt + N].astype(np.float32),
                "barrier_ground_truth": st[:, L.s_barrier_gt:L.s_barrier_gt + N - 1].astype(np.float32),
                "sensor_ground_truth": st[:, L.s_sensor_gt].copy(),
                "current_gate_voltages": st[:, L.s_gate_v:L.s_gate_v + N].copy(),
                "current_barrier_voltages": st[:, L.s_barrier_v:L.s_barrier_v + N - 1].copy(),
                "virtual_gate_matrix": st[:, L.s_vgm:L.s_vgm + G * G].reshape(-1, G, G).copy(),
                "virtual_gate_origin": self._params_host[:, L.origin:L.origin + G].copy(),
                "kalman_means": st[:, L.s_kmean:L.s_kmean + N * N].reshape(-1, N, N).copy(),
                "kalman_variances": st[:, L.s_kvar:L.s_kvar + N * N].reshape(-1, N, N).copy(),
                "steps": steps}

    def raw(self):
        raw = np.zeros((self.B, self.C, self.R * self.R)); pl = np.zeros((self.B, 2))
        _lib.check(self._h, self._lib.qd_get_raw(self._h, raw.ctypes.data, pl.ctypes.data), "qd_get_raw")
        return raw, pl

    def occupations(self):
        occ = np.zeros((self.B, self.C, self.R * self.R, self.N))
        _lib.check(self._h, self._lib.qd_get_occupations(self._h, occ.ctypes.data), "qd_get_occupations")
        return occ

    def eigen(self):
        """(B,C,P,2): ground energy of each pixel's 32-state Hamiltonian and the relative residual of the
        eigenpair the occupations came from (validate mode)."""
        eg = np.zeros((self.B, self.C, self.R * self.R, 2))
        _lib.check(self._h, self._lib.qd_get_eigen(self._h, eg.ctypes.data), "qd_get_eigen")
        return eg

    def search_stats(self):
        """Tile-search counters (validate mode): tiles, tiles redone whole, pixels redone, mean superset size."""
        out = (ctypes.c_uint64 * 16)()
        _lib.check(self._h, self._lib.qd_get_search_stats(self._h, out), "qd_get_search_stats")
        t = max(int(out[0]), 1)
        return {"tiles": int(out[0]), "tiles_redone": int(out[1]), "pixels_redone": int(out[2]),
                "pixels_redone_few_states": int(out[4]), "mean_superset": int(out[3]) / t,
                "tiles_redone_by_reason": {k: int(out[8 + i]) for i, k in
                                           enumerate(("", "ranges", "seeds", "frontier", "leaves", "superset")) if k}}

    def solver_stats(self):
        """Eigen-solver counters of the ground-state kernel (validate mode): tasks = hop components of >= 2 states
        solved, Laguerre iterations per task and per 64-task wave tile (a tile waits for its slowest lane), tasks by size."""
        out = (ctypes.c_uint64 * 16)()
        _lib.check(self._h, self._lib.qd_get_solver_stats(self._h, out), "qd_get_solver_stats")
        tasks, tiles = max(int(out[0]), 1), max(int(out[2]), 1)
        return {"tasks": int(out[0]), "laguerre_per_task": int(out[1]) / tasks, "tiles": int(out[2]),
                "laguerre_per_tile_max": int(out[3]) / tiles, "lane_fill": int(out[0]) / (64.0 * tiles),
                "tasks_by_size": {("9+" if k == 7 else str(k + 2)): int(out[4 + k]) for k in range(8)}}

    def candidates(self):
        st = np.zeros((self.B, self.C, self.R * self.R, 32, self.N), np.int32)
        _lib.check(self._h, self._lib.qd_get_candidates(self._h, st.ctypes.data), "qd_get_candidates")
        return st

    def time_ground_kernel(self, iters=3):
        ms = ctypes.c_float(0)
        _lib.check(self._h, self._lib.qd_time_ground_kernel(self._h, iters, ctypes.byref(ms), self._stream()),
                   "qd_time_ground_kernel")
        return float(ms.value)

    def time_candidates_kernel(self, iters=3):
        ms = ctypes.c_float(0)
        _lib.check(self._h, self._lib.qd_time_candidates_kernel(self._h, iters, ctypes.byref(ms), self._stream()),
                   "qd_time_candidates_kernel")
        return float(ms.value)

    def time_kernels(self, iters=3):
        """HIP-event duration (ms) of one launch over a launch chunk of each hot kernel: dict name -> ms, in pipeline order
        (qd_k_tile, qd_k_candidates = redo pass, qd_k_gs_structure, qd_k_gs_solve = all size classes, qd_k_gs_select)."""
        out = (ctypes.c_float * 5)()
        _lib.check(self._h, self._lib.qd_time_kernels(self._h, int(iters), out, self._stream()), "qd_time_kernels")
        return {self._lib.qd_timed_kernel_name(k).decode(): float(out[k]) for k in range(5)}

    def chunk_envs(self):
        """env-steps covered by one launch of the hot kernels."""
        return int(self._lib.qd_chunk_envs(self._h))
