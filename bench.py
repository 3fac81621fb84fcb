#!/usr/bin/env python
"""bench.py -- env-steps/s of the batched quantum-dot tuning env on N MI355X.

    python bench.py --gpus N --steps K --warmup W

One "step" = one VecQuantumDeviceEnv.step over the whole batch: action rescale +
reward, N-1 CSD channels (exact k-best candidates -> tunnel-coupled ground state
-> sensor), percentile normalisation, global + per-agent images, Kalman / VGM /
ground-truth update.  The capacitance CNN is an input provider (synthetic
values ~ N(0, 0.1^2), log_vars ~ U(-6,-2)); actions ~ U(-1,1);
episodes of 50 steps with automatic reset (resets are inside the timed region).
Deterministic physics: latching and noise off (parity mode).

Multi-GPU: one process per GPU (torch.distributed, backend nccl = RCCL, used
ONLY for the start/stop barrier and the max-over-ranks reduction of the elapsed
time); envs shard independently, no data-path collective ("weak" scaling).
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
for p in (os.path.join(ROOT, "rl-agent-for-qubit-array-tuning_amd"),):
    if p not in sys.path:
        sys.path.insert(0, p)

import numpy as np
import torch

HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: HBM3E 8 TB/s spec
FP64_VECTOR_PEAK_TFLOPS = 78.6


def b_alg(N, R):
    """algorithmic bytes per env-step."""
    G, V, nb = N + 1, 2 * N, N - 1
    n_par = G * G + G * V + nb * G + G * G + G + nb + (2 * N - 1) + 10
    return 8 * n_par + 4 * (N - 1) * R * R + 4 * (3 * N - 1) * R * R


def f_alg(N, R):
    """literal-reference flops per env-step."""
    K, G = 32, N + 1
    per_pixel = 4 ** N * (2 * N * N + 3 * N) + K * (2 * N * N + 3 * N) + 9 * K ** 3 + 22 * G * G
    return per_pixel * (N - 1) * R * R


def valu_busy(N, R):
    f = os.path.join(ROOT, "profiles", "valu_busy.json")
    if not os.path.exists(f):
        return None
    d = json.load(open(f))
    out = {k.split("_")[0]: round(v["valu_busy_frac"], 3) for k, v in d.items()
           if isinstance(v, dict) and k.endswith(f"_{N}dot_{R}")}
    return out or None


def cpu_baseline(N, R, seconds_budget=20.0):
    """The plain-C oracle (literal reference algorithm, OpenMP) on the host cores,
    on a bounded sample: whole channels of one env until the budget is used."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    cores = min(os.cpu_count() or 1, 16)            # a one-GPU box's CPU share
    os.environ["OMP_NUM_THREADS"] = str(cores)
    import qd_oracle_c as OC
    import helpers as H
    eb = H.sample_blocks(N, [1234])
    par, st = eb.params[0], eb.state[0]
    dev = H.dev_view(N, par); sv = H.state_view(N, st)
    OC.csd_channel(dev, sv.vgm, dev.origin, sv.gate_v, sv.sensor_gt, sv.barrier_v, dev.window, 0, 8)  # warm
    t0 = time.perf_counter(); done = 0; rows = 0
    P = R * R
    # time row blocks of 8 rows so small budgets still give a number
    blk = P
    ch = 0; p = 0
    while time.perf_counter() - t0 < seconds_budget:
        OC.csd_channel(dev, sv.vgm, dev.origin, sv.gate_v, sv.sensor_gt, sv.barrier_v, dev.window, ch, R,
                       pix=(p, min(P, p + blk)))
        done += min(P, p + blk) - p
        p += blk
        if p >= P:
            p = 0; ch = (ch + 1) % (N - 1)
    dt = time.perf_counter() - t0
    px_per_s = done / dt
    return {"value": px_per_s / ((N - 1) * P), "unit": "env-steps/s", "cores": cores, "kind": "port",
            "sample": f"{done} pixels ({done / ((N - 1) * P):.3f} env-steps) of one {N}-dot {R}x{R} env in {dt:.1f}s, "
                      f"plain-C OpenMP restatement of the reference algorithm (4^N scan + dense eigensolve)"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=4)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--dots", type=int, default=8)
    ap.add_argument("--envs", type=int, default=4096, help="envs per GPU")
    ap.add_argument("--resolution", type=int, default=64)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-seconds", type=float, default=15.0)
    ap.add_argument("--backend", default="nccl", choices=["nccl", "gloo"],
                    help="torch.distributed backend for the barrier / max-time reduction (gloo: rehearsal of N>1 on one GPU)")
    ap.add_argument("--share-gpu", action="store_true", help="rehearsal only: all ranks use cuda:0")
    ap.add_argument("--noise", action="store_true", help="all stochastic stages on (throughput only)")
    ap.add_argument("--cnn", default="synthetic", choices=["synthetic", "mobilenet", "impala"],
                    help="capacitance model inside the step: synthetic outputs (the metric's definition, SURVEY 8d) or the "
                         "reference's CNN architecture with random weights running on the device (row f1)")
    ap.add_argument("--cnn-dtype", default="float32", choices=["float32", "bfloat16"])
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0")); world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if args.share_gpu:
        local = 0
    if world > 1:
        from qadapt_hip import shard as _sh
        dist = _sh.init(args.backend, local)
    else:
        dist = None
    torch.cuda.set_device(local)
    from qadapt_hip.vec_env import VecQuantumDeviceEnv, SyntheticCapacitanceModel

    N, R, B = args.dots, args.resolution, args.envs
    dev = torch.device(f"cuda:{local}")
    from qadapt_hip import shard as _shard
    first_env, _ = _shard.shard_env_ids(rank, world, B)
    if args.cnn == "synthetic":
        cap_model = SyntheticCapacitanceModel(99 + rank)
    else:
        from qadapt_hip.capacitance_cnn import build_device_model
        cap_model = build_device_model(backbone=args.cnn, device=dev, seed=7, dtype=getattr(torch, args.cnn_dtype))
    env = VecQuantumDeviceEnv(B, num_dots=N, resolution=R, device=dev, seed=1234, env_id_offset=first_env,
                              capacitance_model=cap_model, noise=True if args.noise else None)
    gen = torch.Generator(device="cpu").manual_seed(99 + rank)
    env.reset()

    def one_step():
        act = (torch.rand((B, 2 * N - 1), generator=gen) * 2 - 1).to(dev)
        env.step(act, auto_reset=True)

    for _ in range(args.warmup):
        one_step()
    torch.cuda.synchronize(dev)
    if dist:
        dist.barrier()
    torch.cuda.synchronize(dev)
    t0 = time.perf_counter()
    for _ in range(args.steps):
        one_step()
    torch.cuda.synchronize(dev)
    if dist:
        dist.barrier()
    torch.cuda.synchronize(dev)
    dt = time.perf_counter() - t0
    from qadapt_hip import shard
    dt = shard.max_over_ranks(dt, device=dev if args.backend == "nccl" else None)   # identity when not distributed

    if rank == 0:
        total_env_steps = B * world * args.steps
        value = total_env_steps / dt
        # dominant kernel: ground state (one launch covers `chunk` envs); HIP events inside the library,
        # on the stream the kernels are launched on
        gs_ms = env.time_ground_kernel(iters=2)
        cand_ms = env.time_candidates_kernel(iters=2)
        chunk = env.chunk_envs()
        alg_bytes = b_alg(N, R) * chunk
        achieved = alg_bytes / (gs_ms * 1e-3) / 1e9
        traffic = None
        tfile = os.path.join(ROOT, "profiles", "hbm_traffic.json")     # rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes
        if os.path.exists(tfile):
            t = json.load(open(tfile)).get(f"ground_{N}dot_{R}", None)
            if t:
                traffic = t["bytes_per_env_step"] * chunk
        out = {
            "metric": "env steps/sec (batched 8-dot 64x64 CSD solves)" if (N, R) == (8, 64) else f"env steps/sec ({N}-dot {R}x{R} CSD solves)",
            "value": value, "unit": "env-steps/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": dt / args.steps * 1e3, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "config": {"workload": f"{N}-dot array, {B} parallel envs per GPU, {R}x{R} CSD, "
                                   + ("all stochastic stages on" if args.noise else "deterministic physics (latching/noise off)")
                                   + ", 50-step episodes with auto-reset, "
                                   + ("synthetic CNN outputs" if args.cnn == "synthetic" else
                                      f"{args.cnn} capacitance CNN ({args.cnn_dtype}, random weights) on the device inside the step"),
                       "n_dots": N, "envs_per_gpu": B, "resolution": R, "pixels_per_s": value * (N - 1) * R * R},
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "traffic": traffic,
                         "kernel": f"qd_k_ground<{N}>", "kernel_ms": gs_ms, "envs_per_launch": chunk,
                         "algorithmic_bytes_per_env_step": b_alg(N, R),
                         "second_kernel": f"qd_k_candidates<{N}>", "second_kernel_ms": cand_ms,
                         "note": "the faithful path is float64 VALU/LDS bound, not HBM bound (SURVEY 7-H1); "
                                 "the HBM fraction is reported as the contract asks, see DESIGN.md"},
            # SURVEY 8(d) asks for both rooflines.  F_alg is the LITERAL reference flop count per env-step
            # (4^N-candidate scan + dense 32x32 eigh per pixel); the kernels do far less work than that
            # (exact k-best search, block-wise Lanczos), so the "literal-equivalent" rate may exceed the
            # float64 vector peak -- it measures algorithmic savings, not pipeline utilisation (measured
            # VALU issue utilisation: see measured_issue_utilisation).
            "valu": {"literal_flops_per_env_step": f_alg(N, R),
                     "literal_equivalent_tflops": f_alg(N, R) * value / world / 1e12,
                     "fp64_vector_peak_tflops": FP64_VECTOR_PEAK_TFLOPS,
                     # measured vector-ALU issue utilisation of the two kernels (rocprofv3 SQ counters,
                     # profiles/valu_busy.json): the roofline that actually bounds this path
                     "measured_issue_utilisation": valu_busy(N, R)},
        }
        if not args.no_cpu_baseline and world == 1:          # reported at N = 1 only
            out["cpu_baseline"] = cpu_baseline(N, R, args.cpu_seconds)
        print(json.dumps(out))
    env.close()
    if dist:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
