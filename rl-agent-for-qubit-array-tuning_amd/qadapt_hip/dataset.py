"""
Bulk CSD dataset generation on the batched kernels.

Mirrors src/qadapt/qarray_dataset/symmetric_capacitance_generator_ray.py:183-300
(one sample = one random device whose VIRTUAL gate matrix is set so that the
effective gate-dot couplings equal a sampled symmetric target; the unnormalised
N-1 channel CSD stack is the input, the target coupling matrix the label) and its
on-disk format (:320-360, :363, :571-608):

    <out>/images/batch_XXX.npy        float32 (B, R, R, N-1)   unnormalised CSD stacks
    <out>/cgd_matrices/batch_XXX.npy  float32 (B, N, N+1)      eye + target couplings
    <out>/ground_truth/batch_XXX.json per-sample ground-truth / applied gate voltages
    <out>/metadata/dataset_info.json

which is what src/qadapt/capacitance_model/dataloader.py:54-55 globs.  A whole
batch of samples is rendered by one VecQuantumDeviceEnv observe call instead of
one Ray actor per GPU rendering sample by sample.
"""
from __future__ import annotations

import json
import os
from dataclasses import dataclass, field

import numpy as np

from .vec_env import VecQuantumDeviceEnv


@dataclass
class GenerationConfig:
    total_samples: int
    num_dots: int
    output_dir: str
    batch_size: int = 1000
    seed_base: int = 42
    coupling_min: float = -0.7            # nearest-neighbour target couplings
    coupling_max: float = 0.7
    nnn_coupling_min: float = -0.3        # next-nearest-neighbour
    nnn_coupling_max: float = 0.3
    gate_offset: float = 35.0             # +-V around the ground truth (generator :247-249)
    resolution: int | None = None
    env_config_path: str | None = None
    qarray_config_path: str | None = None
    noise: bool = True                    # sensor + radial noise as the reference's _get_obs applies them
    extra: dict = field(default_factory=dict)


def sample_targets(cfg: GenerationConfig, sample_ids):
    """Per-sample target coupling matrices and voltage offsets (generator :190-262), one numpy
    Generator per sample seeded seed_base + sample_id, draws in the reference order."""
    N = cfg.num_dots
    T = np.zeros((len(sample_ids), N, N)); label = np.zeros((len(sample_ids), N, N + 1), np.float32)
    rngs = []
    for k, sid in enumerate(sample_ids):
        rng = np.random.default_rng(cfg.seed_base + int(sid))
        t = np.eye(N); lab = np.eye(N, N + 1, dtype=np.float32)
        for c in range(N - 1):
            v = rng.uniform(cfg.coupling_min, cfg.coupling_max)
            t[c, c + 1] = t[c + 1, c] = -v
            lab[c, c + 1] = lab[c + 1, c] = v
        for i in range(N - 2):
            v = rng.uniform(cfg.nnn_coupling_min, cfg.nnn_coupling_max)
            t[i, i + 2] = t[i + 2, i] = -v
            lab[i, i + 2] = lab[i + 2, i] = v
        T[k] = t; label[k] = lab; rngs.append(rng)
    return T, label, rngs


class SymmetricCapacitanceGenerator:
    def __init__(self, cfg: GenerationConfig, device=None):
        self.cfg = cfg
        B = min(cfg.batch_size, cfg.total_samples)
        # update_method is irrelevant here (no Kalman step is taken); a dummy provider satisfies the ctor
        self.env = VecQuantumDeviceEnv(B, num_dots=cfg.num_dots, config_path=cfg.env_config_path,
                                       qarray_config_path=cfg.qarray_config_path, resolution=cfg.resolution,
                                       device=device, seed=cfg.seed_base, noise=True if cfg.noise else None,
                                       capacitance_model=lambda img: (None, None))
        self.barrier_offset = float(self.env.config["simulator"]["full_barrier_range_width"]["max"]) / 2
        for d in ("images", "cgd_matrices", "ground_truth", "metadata"):
            os.makedirs(os.path.join(cfg.output_dir, d), exist_ok=True)

    def render_batch(self, sample_ids):
        """-> (images (n,R,R,C) f32, labels (n,N,N+1) f32, gt (n,N) f64, applied gate voltages (n,N))."""
        env, cfg = self.env, self.cfg
        n = len(sample_ids); N = cfg.num_dots; G = N + 1; L = env.L
        ids = np.arange(n, dtype=np.int32)
        eb = env.load_new_devices(ids, seed=None)
        T, label, rngs = sample_targets(cfg, sample_ids)
        ex = eb.extras
        # qarray_base_class.py:948-989: VGM = -pinv(A) @ T_full, negated for electrons => pinv(A) @ T_full
        A = ex["cdd_inv"] @ ex["cgd"][:, :, :G]
        Tf = np.tile(np.eye(G), (n, 1, 1)); Tf[:, :N, :N] = T
        vgm = np.linalg.pinv(A) @ Tf
        origin = eb.params[:, L.origin:L.origin + G]
        virt = np.linalg.solve(vgm, (ex["vopt"] - origin)[:, :, None])[:, :, 0]
        gt = virt[:, :N]
        gate_v = np.zeros((n, N)); barrier_v = np.zeros((n, N - 1))
        for k, rng in enumerate(rngs):
            gate_v[k] = gt[k] + rng.uniform(-cfg.gate_offset, cfg.gate_offset, size=N)
            barrier_v[k] = ex["vb_opt"][k] + rng.uniform(-self.barrier_offset, self.barrier_offset, size=N - 1)
        st, steps = env.get_state()
        st[:n, L.s_vgm:L.s_vgm + G * G] = vgm.reshape(n, -1)
        st[:n, L.s_gate_v:L.s_gate_v + N] = gate_v
        st[:n, L.s_barrier_v:L.s_barrier_v + N - 1] = barrier_v
        st[:n, L.s_gate_gt:L.s_gate_gt + N] = gt                   # radial noise centre (generator :265)
        st[:n, L.s_sensor_gt] = 0.0                                # _get_obs(..., sensor_voltage=None) -> 0.0
        env.set_state(st, steps)
        env.observe()
        raw, _ = env.raw()                                         # (B, C, P) float64, unnormalised
        R = env.R
        images = raw[:n].reshape(n, N - 1, R, R).transpose(0, 2, 3, 1).astype(np.float32)
        return images, label, gt, gate_v

    def run(self):
        cfg = self.cfg
        nb = (cfg.total_samples + cfg.batch_size - 1) // cfg.batch_size
        done = 0
        for b in range(nb):
            ids = list(range(b * cfg.batch_size, min(cfg.total_samples, (b + 1) * cfg.batch_size)))
            images, labels, gt, gate_v = self.render_batch(ids)
            np.save(os.path.join(cfg.output_dir, "images", f"batch_{b:03d}.npy"), images)
            np.save(os.path.join(cfg.output_dir, "cgd_matrices", f"batch_{b:03d}.npy"), labels)
            with open(os.path.join(cfg.output_dir, "ground_truth", f"batch_{b:03d}.json"), "w") as fh:
                json.dump([{"ground_truth_voltages": gt[k].astype(np.float32).tolist(),
                            "gate_voltages": gate_v[k].astype(np.float32).tolist(), "sample_id": int(s)}
                           for k, s in enumerate(ids)], fh, indent=2)
            done += len(ids)
        R = self.env.R; N = cfg.num_dots
        meta = {"generation_config": {"total_samples": cfg.total_samples, "batch_size": cfg.batch_size,
                                      "num_dots": N, "use_barriers": True, "seed_base": cfg.seed_base,
                                      "coupling_range": [cfg.coupling_min, cfg.coupling_max],
                                      "nnn_coupling_range": [cfg.nnn_coupling_min, cfg.nnn_coupling_max]},
                "data_structure": {"images": f"Batched charge sensor images, shape per batch: (batch_size, {R}, {R}, {N - 1})",
                                   "cgd_matrices": f"Target effective coupling matrices, shape per batch: (batch_size, {N}, {N + 1})"},
                "num_batches": nb, "samples_generated": done, "generator": "qadapt_hip.dataset (HIP batched)"}
        with open(os.path.join(cfg.output_dir, "metadata", "dataset_info.json"), "w") as fh:
            json.dump(meta, fh, indent=2)
        return done

    def close(self):
        self.env.close()
