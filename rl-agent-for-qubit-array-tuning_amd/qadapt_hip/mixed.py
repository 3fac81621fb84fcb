"""Ragged batches (BASELINE config 5): environments with different dot counts.

The kernels are specialised per N (register-resident per-dot state, unrolled
N x N forms), so a mixed batch is bucketed by N: one VecQuantumDeviceEnv (one
library handle, one set of launches) per bucket, all on the same GPU and
stream.  Env ids are global across buckets so seeding matches a homogeneous run.
"""
from __future__ import annotations

from .vec_env import VecQuantumDeviceEnv


class MixedVecQuantumDeviceEnv:
    def __init__(self, counts, resolution=None, seed=1234, env_id_offset=0, capacitance_model_factory=None, **kw):
        """counts: {n_dots: n_envs}.  capacitance_model_factory(n_dots) -> callable or None."""
        self.buckets = {}
        off = env_id_offset
        for N in sorted(counts):
            cm = capacitance_model_factory(N) if capacitance_model_factory else kw.get("capacitance_model")
            k = dict(kw); k.pop("capacitance_model", None)
            self.buckets[N] = VecQuantumDeviceEnv(counts[N], num_dots=N, resolution=resolution, seed=seed,
                                                  env_id_offset=off, capacitance_model=cm, **k)
            off += counts[N]
        self.num_envs = off - env_id_offset

    def reset(self, **kw):
        return {N: e.reset(**kw) for N, e in self.buckets.items()}

    def step(self, actions, auto_reset=False):
        """actions: {n_dots: (B_n, 2n-1) tensor}.  Returns {n_dots: (obs, rewards, terminated, truncated)}."""
        return {N: e.step(actions[N], auto_reset=auto_reset) for N, e in self.buckets.items()}

    def close(self):
        for e in self.buckets.values():
            e.close()
