"""Ragged batches (BASELINE config 5): environments with different dot counts on one GPU.

The kernels are specialised per dot count N (register-resident per-dot state, unrolled N x N forms), so a mixed
batch is bucketed by N: one VecQuantumDeviceEnv -- one library handle, one set of launches -- per bucket.  The
buckets are independent, so each one runs on its own HIP stream: the small-N buckets (whose grids do not fill
256 CUs for long) overlap with the tail of the large-N ones instead of queueing behind them.  Global env ids
come from `shard.shard_mixed`, so devices and noise streams are the ones a single-process run of the whole
ragged batch would use, whatever the number of ranks.
"""
from __future__ import annotations

import torch

from . import shard
from .vec_env import VecQuantumDeviceEnv


class MixedVecQuantumDeviceEnv:
    def __init__(self, counts, resolution=None, seed=1234, rank=0, world=1, capacitance_model_factory=None,
                 device=None, streams=True, **kw):
        """counts: {n_dots: n_envs} of the WHOLE job; this object owns rank `rank`'s share of every bucket.
        capacitance_model_factory(n_dots) -> callable or None.  Other keywords go to VecQuantumDeviceEnv
        (e.g. noise=["latch"] for the latched model of config 5)."""
        R = resolution if resolution is not None else 64
        self.assignment = shard.shard_mixed(counts, rank, world, R)
        self.buckets, self.streams = {}, {}
        kw.pop("capacitance_model", None)
        for N, (first, n) in self.assignment.items():
            if n == 0:
                continue
            cm = capacitance_model_factory(N) if capacitance_model_factory else None
            self.buckets[N] = VecQuantumDeviceEnv(n, num_dots=N, resolution=resolution, seed=seed, env_id_offset=first,
                                                  capacitance_model=cm, device=device, **kw)
            self.streams[N] = torch.cuda.Stream(device=self.buckets[N].device) if streams else None
        self.num_envs = sum(e.num_envs for e in self.buckets.values())

    def _each(self, fn):
        """Run fn(N, env) for every bucket, each on its own stream (forked from / joined to the current one)."""
        out = {}
        cur = {}
        for N, e in self.buckets.items():
            s = self.streams[N]
            if s is None:
                out[N] = fn(N, e)
                continue
            cur[N] = torch.cuda.current_stream(e.device)
            s.wait_stream(cur[N])
            with torch.cuda.stream(s):
                out[N] = fn(N, e)
        for N, c in cur.items():
            c.wait_stream(self.streams[N])
        return out

    def reset(self, **kw):
        return self._each(lambda N, e: e.reset(**kw))

    def stagger_episodes(self):
        for e in self.buckets.values():
            e.stagger_episodes()

    def step(self, actions, auto_reset=False):
        """actions: {n_dots: (B_n, 2n-1) tensor}.  Returns {n_dots: (obs, rewards, terminated, truncated)}."""
        return self._each(lambda N, e: e.step(actions[N], auto_reset=auto_reset))

    def close(self):
        for e in self.buckets.values():
            e.close()
