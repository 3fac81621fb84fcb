"""
RLlib-facing multi-agent boundary of the batched simulator.

The reference steps ONE env per RLlib runner through `MultiAgentEnvWrapper`
(src/qadapt/environment/multi_agent_wrapper.py:27-584).  Here B environments share one
`VecQuantumDeviceEnv`, i.e. one libqdsim handle and one set of kernel launches per step:

  AgentRoster             agent ids, the per-agent channel assignment and observation / action spaces
                          (multi_agent_wrapper.py:114-116, 147-178, 194-309) -- pure bookkeeping, no arrays
  BatchedMultiAgentEnv    B logical envs over one backend.  step(list of B action dicts) packs all actions into
                          one (B, 2N-1) tensor, launches ONCE, copies the kernel-written per-agent tensors
                          (`plunger_images`, `barrier_images`, `voltages`, rewards, flags) to a pinned host mirror
                          and hands out FRESH numpy arrays per step (one bulk copy of the mirror per tensor, then
                          views per agent), as the reference wrapper does; `zero_copy=True` hands out views of the
                          2-deep pinned ring itself (valid for ONE further step only).  Nothing is re-derived on
                          the host.  This is the vector form RLlib's vectorised multi-agent runners call.
  MultiAgentEnvView       env i of a batch with the exact single-env surface of the reference wrapper (reset / step /
                          observation_spaces / action_spaces / get_agent_ids / close).  step() stages the view's
                          actions; the launch happens when the last view of the batch has staged (lazy batched step).
  MultiAgentEnvWrapper    the reference constructor signature; a batch of one, or -- through the reference's own
                          `base_env_class` hook -- any single gym-style base env.

Subclasses ray.rllib's MultiAgentEnv when ray is importable (it is absent in the build container).
The GIF / distance-history side logging of the reference (:587-895) is tooling, not simulation, and is not built;
those constructor arguments are accepted and ignored.
"""
from __future__ import annotations

import numpy as np

from . import spaces

try:                                            # pragma: no cover
    from ray.rllib.env.multi_agent_env import MultiAgentEnv as _RllibBase
except Exception:
    class _RllibBase:                           # ray absent: same surface, no RLlib registration
        def __init__(self):
            pass


class StepPending(RuntimeError):
    """A view of a batch was asked for results before every view had staged its actions."""


def per_agent_images(global_image, num_dots):
    """The per-agent image layout derived from a global (H,W,C) image: plunger (N,H,W,2), barrier (C,H,W,1).  Only for
    foreign base envs that do not provide the per-agent tensors themselves (the reference's `base_env_class` hook);
    the HIP backend's kernels write these tensors directly."""
    g = np.asarray(global_image, dtype=np.float32)
    gt = g.transpose(1, 0, 2)
    N, C = num_dots, num_dots - 1
    plunger = np.empty((N,) + g.shape[:2] + (2,), np.float32)
    plunger[0, ..., 0] = g[..., 0]; plunger[0, ..., 1] = g[..., 0]
    plunger[N - 1, ..., 0] = gt[..., C - 1]; plunger[N - 1, ..., 1] = gt[..., C - 1]
    for i in range(1, N - 1):
        plunger[i, ..., 0] = g[..., i - 1]; plunger[i, ..., 1] = gt[..., i]
    barrier = np.ascontiguousarray(g.transpose(2, 0, 1))[..., None]
    return plunger, barrier


# ----------------------------------------------------------------------------------------------------------
class AgentRoster:
    """Who the agents are and what each one sees.  Plunger i looks at the CSD channels left and right of its dot
    (the end dots see their single channel twice), barrier j at channel j; images of the right-hand channel are
    transposed so that the agent's own gate is always on the x axis (multi_agent_wrapper.py:147-178, 311-350)."""

    def __init__(self, num_dots, image_hw, return_voltage, return_global_state, gate_range=(-1.0, 1.0),
                 barrier_range=(-1.0, 1.0)):
        if return_global_state and not return_voltage:
            raise ValueError("return_global_state=True requires return_voltage=True (the global "
                             "state extends the per-agent dict observation).")
        self.N = int(num_dots); self.C = self.N - 1
        self.return_voltage = bool(return_voltage); self.return_global_state = bool(return_global_state)
        self.plungers = [f"plunger_{i}" for i in range(self.N)]
        self.barriers = [f"barrier_{j}" for j in range(self.C)]
        self.ids = self.plungers + self.barriers
        self.slot = {a: k for k, a in enumerate(self.ids)}            # position in the (2N-1) action / reward vectors
        last = self.N - 2
        self.channels = {f"plunger_{i}": ([0, 0] if i == 0 else [last, last] if i == self.N - 1 else [i - 1, i])
                         for i in range(self.N)}
        self.channels.update({f"barrier_{j}": [j] for j in range(self.C)})
        H, W = image_hw
        box = spaces.Box
        obs, act = {}, {}
        glob = {}
        if self.return_global_state:
            glob = {"global_image": box(low=0.0, high=1.0, shape=(H, W, self.C), dtype=np.float32),
                    "global_voltages": box(low=min(gate_range[0], barrier_range[0]), high=max(gate_range[1], barrier_range[1]),
                                           shape=(2 * self.N - 1,), dtype=np.float32)}
        for names, depth, (lo, hi) in ((self.plungers, 2, gate_range), (self.barriers, 1, barrier_range)):
            for a in names:
                img = box(low=0.0, high=1.0, shape=(H, W, depth), dtype=np.float32)
                act[a] = box(low=lo, high=hi, shape=(1,), dtype=np.float32)
                obs[a] = (spaces.Dict({"image": img, "voltage": box(low=lo, high=hi, shape=(1,), dtype=np.float32), **glob})
                          if self.return_voltage else img)
        self.observation_spaces = spaces.Dict(obs)
        self.action_spaces = spaces.Dict(act)

    def check_actions(self, agent_actions):
        assert len(agent_actions) == len(self.ids), "Agent actions must match the number of agents"
        assert all(a in self.slot for a in agent_actions.keys()), "Unknown agent IDs in actions"

    def pack_actions(self, agent_actions, out):
        """dict agent -> scalar or (1,) array  ->  out[2N-1] float32 (gates then barriers)."""
        for a, v in agent_actions.items():
            out[self.slot[a]] = v[0] if hasattr(v, "__len__") else v

    def observations(self, plunger, barrier, voltages, global_image=None):
        """plunger (N,H,W,2), barrier (C,H,W,1), voltages (2N-1,) of ONE env -> dict agent -> observation.  The
        arrays are handed out as they are (views), which is what makes the host side zero-copy."""
        out = {}
        for k, a in enumerate(self.ids):
            img = plunger[k] if k < self.N else barrier[k - self.N]
            if not self.return_voltage:
                out[a] = img
                continue
            o = {"image": img, "voltage": voltages[k:k + 1]}
            if self.return_global_state:
                o["global_image"] = global_image
                o["global_voltages"] = voltages
            out[a] = o
        return out

    def rewards(self, vec):
        return {a: float(vec[k]) for k, a in enumerate(self.ids)}

    def flags(self, value):
        d = dict.fromkeys(self.ids, value)
        d["__all__"] = value
        return d


# ----------------------------------------------------------------------------------------------------------
class _HostMirror:
    """Pinned host copies of the tensors a step produces.  By default every pull() hands out FRESH numpy arrays (one
    bulk copy out of the pinned staging buffers per tensor), because consumers keep observations: RLlib episode buffers
    and host-side frame stacking hold the arrays of step t long after step t+2, and the reference wrapper returns new
    arrays every call (multi_agent_wrapper.py:311-383).  zero_copy=True hands out views of the 2-deep pinned ring
    instead: the arrays of step t are overwritten by step t+2 -- only for consumers that copy or finish with an
    observation before stepping twice."""

    def __init__(self, vec, with_global, zero_copy=False):
        import torch
        self.torch = torch
        self.vec = vec
        self.zero_copy = bool(zero_copy)
        names = ["plunger_images", "barrier_images", "voltages", "rewards", "truncated"] + (["global_image"] if with_global else [])
        self.on_gpu = getattr(vec, "plunger_images").is_cuda         # (host tensors only in the CPU-tier tests)
        self.sets = []
        for _ in range(2):
            s = {}
            for n in names:
                t = getattr(vec, n)
                s[n] = torch.empty(t.shape, dtype=t.dtype, pin_memory=self.on_gpu)
            self.sets.append(s)
        self.turn = 0

    def pull(self):
        """device -> pinned host, one stream synchronisation; returns numpy views of the fresh set."""
        torch = self.torch
        s = self.sets[self.turn]; self.turn ^= 1
        for n, host in s.items():
            host.copy_(getattr(self.vec, n), non_blocking=True)
        if self.on_gpu:
            torch.cuda.current_stream(self.vec.device).synchronize()
        if self.zero_copy:
            return {n: h.numpy() for n, h in s.items()}
        return {n: h.numpy().copy() for n, h in s.items()}


class BatchedMultiAgentEnv:
    """B logical multi-agent envs over ONE batched backend (one launch set per step).

    backend: a VecQuantumDeviceEnv (built here from the keyword arguments when None), or any object with its
    surface (num_envs, N, R, reset(), step(actions, auto_reset=...), device_state(), the output tensors) -- the
    CPU-tier tests pass a fake."""

    def __init__(self, num_envs=None, return_voltage=True, return_global_state=False, env_config_path=None,
                 capacitance_model=None, backend=None, auto_reset=True, zero_copy=False, **vec_kwargs):
        """auto_reset (default True): an env that truncates gets a new random device and a fresh first observation inside
        the same batched step, the way a vectorised runner expects; with False a truncated env stays truncated until
        its view's reset() (or reset()) is called.  zero_copy: see _HostMirror."""
        if backend is None:
            from .vec_env import VecQuantumDeviceEnv
            backend = VecQuantumDeviceEnv(num_envs, config_path=env_config_path, capacitance_model=capacitance_model,
                                          **vec_kwargs)
        self.vec = backend
        self.B = int(backend.num_envs)
        self.num_dots = int(backend.N)
        self.auto_reset = bool(auto_reset)
        R = int(backend.R)
        self.roster = AgentRoster(self.num_dots, (R, R), return_voltage, return_global_state)
        self._mirror = backend.make_mirror(return_global_state) if hasattr(backend, "make_mirror") \
            else _HostMirror(backend, return_global_state, zero_copy=zero_copy)
        self._actions = np.zeros((self.B, 2 * self.num_dots - 1), np.float32)
        self._staged = np.zeros(self.B, bool)
        self._results = [None] * self.B
        self.views = [MultiAgentEnvView(self, i) for i in range(self.B)]
        self.launches = 0                       # batched steps issued (tests: one per step, whatever B is)
        self._latest = None                     # (host arrays, device state) of the most recent reset / step

    # -- vector API ----------------------------------------------------------------------------------------
    def reset(self, *, seed=None, options=None):
        self.vec.reset(seed=seed)
        host = self._mirror.pull()
        ds = self.vec.device_state()
        self._latest = (host, ds)
        self._staged[:] = False
        return [self._obs_of(host, b) for b in range(self.B)], [self._reset_info(ds, b) for b in range(self.B)]

    def _reset_info(self, ds, b):
        info = {"current_device_state": self._device_state_of(ds, b)}       # env.py:214-222, one dict shared by the agents
        return {a: info for a in self.roster.ids}

    def current(self, b):
        """First observation of env b's running episode as of the latest launch (after an automatic reset this is
        the new episode's first observation): what a view's reset() returns without touching the device."""
        if self._latest is None:
            self.reset()
        host, ds = self._latest
        return self._obs_of(host, b), self._reset_info(ds, b)

    def reset_env(self, b, seed=None):
        """Reset env b ALONE (new random device, step counter 0, fresh first observation), as the reference wrapper's
        reset() does for its one env (multi_agent_wrapper.py:459-483 -> env.py:135-237)."""
        if self._latest is None:
            self.reset(seed=seed)
            return self.current(b)
        self.vec.reset(env_ids=[b], seed=seed)
        host = self._mirror.pull()
        ds = self.vec.device_state()
        self._latest = (host, ds)
        return self._obs_of(host, b), self._reset_info(ds, b)

    def needs_reset(self, b):
        """True when env b's episode is over and nothing has replaced it yet (only possible without auto_reset)."""
        steps = getattr(self.vec, "_steps_host", getattr(self.vec, "steps", None))
        max_steps = getattr(self.vec, "max_steps", None)
        return steps is not None and max_steps is not None and int(steps[b]) >= int(max_steps)

    def step(self, actions_per_env):
        """actions_per_env: sequence of B dicts agent -> action.  ONE backend step.  Returns five lists of length B
        (obs, rewards, terminateds, truncateds, infos) in the reference wrapper's per-env format."""
        if len(actions_per_env) != self.B:
            raise ValueError(f"expected actions for {self.B} envs, got {len(actions_per_env)}")
        for b, acts in enumerate(actions_per_env):
            self.roster.check_actions(acts)
            self.roster.pack_actions(acts, self._actions[b])
        return self._launch()

    def _launch(self):
        self.vec.step(self._actions, auto_reset=self.auto_reset)
        self.launches += 1
        host = self._mirror.pull()
        ds = self.vec.device_state()
        self._latest = (host, ds)
        ro = self.roster
        obs, rews, terms, truncs, infos = [], [], [], [], []
        for b in range(self.B):
            obs.append(self._obs_of(host, b))
            rews.append(ro.rewards(host["rewards"][b]))
            tr = bool(host["truncated"][b])
            terms.append(ro.flags(False)); truncs.append(ro.flags(tr))
            try:
                info = {}
                for i, a in enumerate(ro.plungers):
                    info[a] = {"ground_truth": ds["gate_ground_truth"][b][i], "current_voltage": ds["current_gate_voltages"][b][i]}
                for j, a in enumerate(ro.barriers):
                    info[a] = {"ground_truth": ds["barrier_ground_truth"][b][j], "current_voltage": ds["current_barrier_voltages"][b][j]}
            except Exception as e:                                    # multi_agent_wrapper.py:572-573
                raise RuntimeError(f"Error creating multi-agent info: {e}")
            infos.append(info)
        self._staged[:] = False
        return obs, rews, terms, truncs, infos

    def _obs_of(self, host, b):
        return self.roster.observations(host["plunger_images"][b], host["barrier_images"][b], host["voltages"][b],
                                        host["global_image"][b] if "global_image" in host else None)

    @staticmethod
    def _device_state_of(ds, b):
        return {k: (v[b] if hasattr(v, "__len__") else v) for k, v in ds.items()}

    # -- lazy batched step for views ------------------------------------------------------------------------
    def _stage(self, b, agent_actions):
        self.roster.check_actions(agent_actions)
        if self._staged[b]:
            raise StepPending(f"env {b} stepped twice before the batch was launched")
        self.roster.pack_actions(agent_actions, self._actions[b])
        self._staged[b] = True
        if self._staged.all():
            out = self._launch()
            for k in range(self.B):
                self._results[k] = tuple(part[k] for part in out)

    def _collect(self, b):
        r = self._results[b]
        if r is None:
            raise StepPending(f"env {b}: the batched step is launched when all {self.B} views have staged their actions; "
                              f"{int(self._staged.sum())} have (use BatchedMultiAgentEnv.step for the vector form)")
        self._results[b] = None
        return r

    def close(self):
        if hasattr(self.vec, "close"):
            self.vec.close()


class MultiAgentEnvView(_RllibBase):
    """Env `index` of a BatchedMultiAgentEnv, with the reference wrapper's single-env surface."""

    def __init__(self, batch, index):
        super().__init__()
        self.batch, self.index = batch, index
        ro = batch.roster
        self.num_gates = ro.N; self.num_barriers = ro.C; self.num_image_channels = ro.C
        self.use_barriers = True
        self.gate_agent_ids, self.barrier_agent_ids, self.all_agent_ids = ro.plungers, ro.barriers, ro.ids
        self.agent_channel_map = ro.channels
        self.observation_spaces = self.observation_space = ro.observation_spaces
        self.action_spaces = self.action_space = ro.action_spaces
        self._agent_ids = set(ro.ids)
        self.agents = set(ro.ids); self.possible_agents = set(ro.ids)
        self._first_obs = None

    def get_agent_ids(self):
        return self._agent_ids

    def reset(self, *, seed=None, options=None):
        """The envs of a batch are reset together (first call) and, while running, one by one by the backend when
        they truncate (auto_reset, the default): then the env's first observation already exists and is returned
        without touching the device.  An env that truncated WITHOUT being replaced (auto_reset=False), or an explicit
        seed, gets a real reset of this env alone: new device, step counter 0, fresh observation -- the reference
        wrapper's reset semantics (multi_agent_wrapper.py:459-483)."""
        if seed is not None or self.batch.needs_reset(self.index):
            return self.batch.reset_env(self.index, seed=seed)
        return self.batch.current(self.index)

    def stage(self, agent_actions):
        self.batch._stage(self.index, agent_actions)

    def collect(self):
        return self.batch._collect(self.index)

    def step(self, agent_actions):
        self.stage(agent_actions)
        return self.collect()

    def close(self):
        pass



# ----------------------------------------------------------------------------------------------------------
class _SingleEnvBackend:
    """Adapter that lets ONE gym-style base env (the reference's `base_env_class` hook: ctor (training, config_path,
    capacitance_model_checkpoint), reset/step with the global Dict observation) stand behind BatchedMultiAgentEnv."""

    def __init__(self, base_env, roster_args):
        self.env = base_env
        self.num_envs = 1
        self.N = int(base_env.num_dots)
        img = base_env.observation_space["image"].shape
        self.R = int(img[0])
        self._hw = (int(img[0]), int(img[1]))
        self._last = None
        self._info = None
        self._want_global = roster_args["return_global_state"]

    def make_mirror(self, with_global):
        return self

    def pull(self):                                             # the "mirror" of a host env is its last observation
        return self._last

    def _remember(self, obs, rewards=None, truncated=False):
        N = self.N
        plunger, barrier = per_agent_images(obs["image"], N)
        v = np.concatenate([obs["obs_gate_voltages"], obs["obs_barrier_voltages"]]).astype(np.float32)
        rw = np.zeros(2 * N - 1) if rewards is None else np.concatenate([rewards["gates"], rewards["barriers"]])
        self._last = {"plunger_images": plunger[None], "barrier_images": barrier[None], "voltages": v[None],
                      "rewards": rw[None], "truncated": np.array([truncated]),
                      "global_image": np.asarray(obs["image"], np.float32)[None]}

    def reset(self, seed=None, **kw):
        obs, info = self.env.reset(seed=seed)
        self._info = info
        self._remember(obs)

    def step(self, actions, auto_reset=False):
        N = self.N
        a = np.asarray(actions, np.float32).reshape(-1)
        obs, rewards, terminated, truncated, info = self.env.step(
            {"action_gate_voltages": a[:N].copy(), "action_barrier_voltages": a[N:].copy()})
        for key in ("gates", "barriers"):
            if key not in rewards:
                raise ValueError(f"Missing {'gate' if key == 'gates' else 'barrier'} rewards in global_rewards")
        self._info = info
        self._remember(obs, rewards, truncated)

    def device_state(self):
        ds = (self._info or {}).get("current_device_state") or {}
        return {k: np.asarray(v)[None] if np.ndim(v) else np.asarray([v]) for k, v in ds.items()}

    def close(self):
        if hasattr(self.env, "close"):
            self.env.close()


class MultiAgentEnvWrapper(MultiAgentEnvView):
    """The reference's constructor (multi_agent_wrapper.py:43-54).  With `base_env_class` (the reference's own
    plug-in hook, :93-106) the given single env is wrapped; otherwise a batch of one on the HIP backend."""

    def __init__(self, training: bool = True, return_voltage: bool = False, return_global_state: bool = False,
                 gif_config: dict = None, distance_data_dir: str = None, env_config_path: str = None,
                 capacitance_model_checkpoint: str = None, is_collecting_data: bool = False,
                 base_env_class=None, **base_env_kwargs):
        if return_global_state and not return_voltage:
            raise ValueError("return_global_state=True requires return_voltage=True (the global "
                             "state extends the per-agent dict observation).")
        self.return_voltage, self.return_global_state = return_voltage, return_global_state
        self.distance_data_dir, self.is_collecting_data, self.gif_config = distance_data_dir, is_collecting_data, gif_config
        if base_env_class is None:
            from .env import QuantumDeviceEnv
            base_env_class = QuantumDeviceEnv
        kw = dict(training=training, capacitance_model_checkpoint=capacitance_model_checkpoint, **base_env_kwargs)
        if env_config_path:
            kw["config_path"] = env_config_path
        self.base_env = base_env_class(**kw)
        vec = getattr(self.base_env, "_b", None)
        if vec is not None and hasattr(vec, "plunger_images"):
            backend = vec                                         # our own env: use the kernel-written per-agent tensors
        else:
            backend = _SingleEnvBackend(self.base_env, dict(return_global_state=return_global_state))
        batch = BatchedMultiAgentEnv(return_voltage=return_voltage, return_global_state=return_global_state,
                                     backend=backend)
        super().__init__(batch, 0)
        self.base_observation_space = self.base_env.observation_space
        self.base_action_space = self.base_env.action_space

    def reset(self, *, seed=None, options=None):
        if isinstance(self.batch.vec, _SingleEnvBackend):
            obs, _ = self.batch.reset(seed=seed, options=options)
            info = self.batch.vec._info
            return obs[0], {a: info for a in self.all_agent_ids}
        obs, infos = self.batch.reset(seed=seed, options=options)
        if hasattr(self.base_env, "_refresh_device_state"):
            self.base_env.current_step = 0
            self.base_env._refresh_device_state()
        return obs[0], infos[0]

    def step(self, agent_actions):
        obs, rew, term, trunc, infos = super().step(agent_actions)
        if isinstance(self.batch.vec, _SingleEnvBackend) and not (self.batch.vec._info or {}).get("current_device_state"):
            infos = {a: {} for a in self.all_agent_ids}
        return obs, rew, term, trunc, infos

    def close(self):
        if hasattr(self.base_env, "close"):
            self.base_env.close()
