#!/usr/bin/env python
"""bench.py -- env-steps/s of the batched quantum-dot tuning env on N MI355X.

    python bench.py --gpus N --steps K --warmup W [--config headline|config2|config1|mixed]

One "step" = one VecQuantumDeviceEnv.step over the whole batch: action rescale + reward, N-1 CSD channels
(exact k-best candidates -> tunnel-coupled ground state -> sensor), percentile normalisation, global +
per-agent images, Kalman / VGM / ground-truth update.  The capacitance CNN is an input provider (synthetic
values ~ N(0, 0.1^2), log_vars ~ U(-6,-2)); actions ~ U(-1,1).  Episodes last 50 steps and the
episode phases are staggered at start (env e begins at step e mod 50), so 1/50 of the batch truncates, gets a
new random device (host sampler, overlapped with the kernels) and a fresh first observation INSIDE every timed
step -- the steady state of a long-running sampler.  Deterministic physics unless --noise / --config mixed.

Multi-GPU: one process per GPU.  Launched by torchrun (RANK / LOCAL_RANK / WORLD_SIZE in the environment), or
by this script itself: with --gpus N and no WORLD_SIZE the parent starts N rank processes BEFORE it touches
the GPU and only waits for them.  torch.distributed (nccl = RCCL) is used ONLY for the start/stop barrier and
the max-over-ranks of the elapsed time; env shards are independent, no data-path collective ("weak" scaling).
"""
import argparse
import hashlib
import json
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
PKG = os.path.join(ROOT, "rl-agent-for-qubit-array-tuning_amd")
if PKG not in sys.path:
    sys.path.insert(0, PKG)

HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: HBM3E 8 TB/s spec
FP64_VECTOR_PEAK_TFLOPS = 78.6

CONFIGS = {
    # BASELINE.json configs[2]: the configuration the metric is quoted on
    "headline": dict(dots=8, envs=4096, resolution=64),
    "config2": dict(dots=4, envs=256, resolution=64),
    "config1": dict(dots=2, envs=1, resolution=32),
    # configs[4]: mixed N in {2,4,6,8}, equal shares, latched-state model; per GPU 4 x 1024 envs
    "mixed": dict(dots=None, envs=4096, resolution=64),
}


def b_alg(N, R):
    """algorithmic bytes per env-step."""
    G, V, nb = N + 1, 2 * N, N - 1
    n_par = G * G + G * V + nb * G + G * G + G + nb + (2 * N - 1) + 10
    return 8 * n_par + 4 * (N - 1) * R * R + 4 * (3 * N - 1) * R * R


def kernel_source_hash():
    """Identity of the HIP sources: counter-derived figures under profiles/ are only quoted for the kernels
    they were measured on."""
    h = hashlib.sha256()
    csrc = os.path.join(PKG, "csrc")
    for f in sorted(os.listdir(csrc)):
        if f.endswith((".h", ".hip")):
            h.update(open(os.path.join(csrc, f), "rb").read())
    return h.hexdigest()[:16]


def git_rev():
    try:
        return subprocess.check_output(["git", "-C", ROOT, "rev-parse", "--short", "HEAD"], stderr=subprocess.DEVNULL).decode().strip()
    except Exception:
        return None


def profile_figures(N, R):
    """HBM traffic and VALU issue utilisation from the rocprofv3 PMC passes committed under profiles/
    (profiles/counters.json, written by scripts/pmc_summary.py) -- only if they were taken on THIS kernel source."""
    f = os.path.join(ROOT, "profiles", "counters.json")
    if not os.path.exists(f):
        return None
    d = json.load(open(f))
    ent = d.get(f"{N}dot_{R}")
    if not ent or ent.get("kernel_src_sha") != kernel_source_hash():
        return None
    return ent


def cpu_baseline_and_parity(N, R, seed, seconds_budget=15.0):
    """The oracle leg.  (1) cpu_baseline: the plain-C oracle (literal reference algorithm: 4^N scan + dense 32x32
    eigensolve, OpenMP) timed on the host cores over whole CSD channels of env 0 of the workload (same device,
    same seed) until the budget is used.  (2) parity: the same channels rendered by the HIP path in validate mode,
    compared with what the oracle just computed (kept charge states, occupations, raw sensor signal)."""
    for p in (os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests")):
        if p not in sys.path:
            sys.path.insert(0, p)
    import numpy as np
    cores = min(os.cpu_count() or 1, 16)            # a one-GPU box's CPU share
    os.environ["OMP_NUM_THREADS"] = str(cores)
    import qd_oracle_c as OC
    import helpers as H
    from qadapt_hip.vec_env import VecQuantumDeviceEnv, SyntheticCapacitanceModel
    env = VecQuantumDeviceEnv(1, num_dots=N, resolution=R, seed=seed, validate=True,
                              capacitance_model=SyntheticCapacitanceModel(99))
    env.reset()
    st, steps = env.get_state()
    # voltages within ~10 V / ~6 V of the ground truth: the regime where the oracle's dense float64 eigh resolves the
    # spectrum, so occupations are comparable pixel by pixel (random-action scenes are mostly beyond it, DESIGN 6;
    # the literal CPU algorithm costs the same anywhere)
    st[0] = H.place(N, st[0], "mid", np.random.default_rng(seed), vgm_noise=0.0)
    env.set_state(st, steps)
    env.observe()
    cand = env.candidates(); occ = env.occupations(); raw, _ = env.raw(); eig = env.eigen()
    dev = H.dev_view(N, env._params_host[0]); sv = H.state_view(N, st[0])
    P = R * R
    OC.csd_channel(dev, sv.vgm, dev.origin, sv.gate_v, sv.sensor_gt, sv.barrier_v, dev.window, 0, R, pix=(0, 64))  # warm
    t0 = time.perf_counter(); done = 0
    mism = 0; worst_occ = 0.0; worst_sig = 0.0; compared = 0; unres = 0; worst_res = 0.0
    ch = 0; first_pass = True
    while time.perf_counter() - t0 < seconds_budget:
        ref = OC.csd_channel(dev, sv.vgm, dev.origin, sv.gate_v, sv.sensor_gt, sv.barrier_v, dev.window, ch, R)
        done += P
        if first_pass:
            tp = time.perf_counter()
            mism += int((cand[0, ch] != ref["states"]).any(axis=(1, 2)).sum())
            ok = ref["tc"].max(axis=1) < 1e6            # where float64 resolves the dense eigh
            unres += int((~ok).sum()); compared += int(ok.sum())
            if ok.any():
                worst_occ = max(worst_occ, float(np.abs(occ[0, ch][ok] - ref["occ"][ok]).max()))
                worst_sig = max(worst_sig, float((np.abs(raw[0, ch][ok] - ref["z"][ok]) / np.maximum(np.abs(ref["z"][ok]), 1e-3)).max()))
            worst_res = max(worst_res, float(eig[0, ch, :, 1].max()))
            t0 += time.perf_counter() - tp           # the comparison is not part of the timed CPU work
        ch += 1
        if ch == N - 1:
            ch = 0; first_pass = False
    dt = time.perf_counter() - t0
    env.close()
    base = {"value": done / dt / ((N - 1) * P), "unit": "env-steps/s", "cores": cores, "kind": "port",
            "sample": f"{done} pixels ({done / ((N - 1) * P):.2f} env-steps) of env 0 of the workload ({N}-dot {R}x{R}) in {dt:.1f}s, "
                      f"plain-C OpenMP restatement of the reference algorithm (4^N scan + dense eigensolve)"}
    parity = {"scene": f"env 0 (seed {seed}) moved to within 10 V / 6 V of its ground truth, {min(done // P, N - 1)} of {N - 1} channels, {R}x{R}",
              "charge_state_list_mismatches": mism, "pixels_compared": compared, "pixels_unresolvable_in_f64": unres,
              "max_abs_occupation_error": worst_occ, "max_rel_signal_error": worst_sig,
              "max_eigen_residual": worst_res, "oracle": "parity unpinned for a6-a15, a17, a20-a23 (DESIGN 2)"}
    return base, parity


# ----------------------------------------------------------------------------------------------
def launch_ranks(args, argv):
    """--gpus N without torchrun: start N rank processes from a parent that never touches the GPU.  Every rank prints its
    own rate to stderr (imbalance is visible), host threads are capped per rank so that N device samplers do not
    oversubscribe the cores, and the children are POLLED: the first one that fails takes its siblings down with it
    (they would otherwise sit in a barrier until the backend times out) and its exit code is returned."""
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    threads = max(1, (os.cpu_count() or args.gpus) // args.gpus)
    procs = []
    for r in range(args.gpus):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(0 if args.share_gpu else r), WORLD_SIZE=str(args.gpus),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
        for k in ("OMP_NUM_THREADS", "OPENBLAS_NUM_THREADS", "MKL_NUM_THREADS"):
            env.setdefault(k, str(threads))
        # (QD_BENCH_RANK_SCRIPT: the CPU-tier test of this launcher substitutes a recording stub)
        script = os.environ.get("QD_BENCH_RANK_SCRIPT", os.path.abspath(__file__))
        procs.append(subprocess.Popen([sys.executable, script] + argv, env=env,
                                      stdout=None if r == 0 else subprocess.DEVNULL))
    rc = 0
    live = list(procs)
    while live and rc == 0:
        for p in list(live):
            code = p.poll()
            if code is None:
                continue
            live.remove(p)
            if code != 0:
                rc = abs(code) or 1
                print(f"bench.py: rank {procs.index(p)} exited with code {code}; stopping the other ranks", file=sys.stderr)
                break
        else:
            time.sleep(0.05)
    for p in live:                                   # only after a failure: stop exactly the processes started here
        p.terminate()
    for p in live:
        try:
            p.wait(timeout=10)
        except subprocess.TimeoutExpired:
            p.kill()
    return rc


def build_parser():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--config", default="headline", choices=sorted(CONFIGS))
    ap.add_argument("--dots", type=int, default=None)
    ap.add_argument("--envs", type=int, default=None, help="envs per GPU")
    ap.add_argument("--resolution", type=int, default=None)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-seconds", type=float, default=15.0)
    ap.add_argument("--backend", default="nccl", choices=["nccl", "gloo"],
                    help="torch.distributed backend for the barrier / max-time reduction (gloo: rehearsal of N>1 on one GPU)")
    ap.add_argument("--share-gpu", action="store_true", help="rehearsal only: all ranks use cuda:0")
    ap.add_argument("--noise", action="store_true", help="all stochastic stages on (throughput only)")
    ap.add_argument("--no-stagger", action="store_true", help="start all episodes in phase (no resets before step 50)")
    ap.add_argument("--cnn", default="synthetic", choices=["synthetic", "mobilenet", "impala"],
                    help="capacitance model inside the step: synthetic outputs (the metric's definition, SURVEY 8d) or the "
                         "reference's CNN architecture with random weights running on the device (row f1)")
    ap.add_argument("--cnn-dtype", default="float32", choices=["float32", "bfloat16"])
    return ap


def main(argv=None):
    argv = list(sys.argv[1:] if argv is None else argv)
    args = build_parser().parse_args(argv)
    if "WORLD_SIZE" not in os.environ:
        if args.gpus > 1:
            sys.exit(launch_ranks(args, argv))
        world, rank, local = 1, 0, 0
    else:
        rank = int(os.environ.get("RANK", "0")); world = int(os.environ["WORLD_SIZE"])
        local = int(os.environ.get("LOCAL_RANK", "0"))
        if world != args.gpus:
            print(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={world}; pass --gpus {world}", file=sys.stderr)
            sys.exit(2)
    if args.share_gpu:
        local = 0
    run_rank(args, rank, world, local)


def run_rank(args, rank, world, local):
    import torch
    from qadapt_hip import shard
    from qadapt_hip.vec_env import VecQuantumDeviceEnv, SyntheticCapacitanceModel
    from qadapt_hip.mixed import MixedVecQuantumDeviceEnv

    local = shard.local_device_index(local, torch.cuda.device_count())     # (HIP_VISIBLE_DEVICES may already select one GPU per rank)
    dist = shard.init(args.backend, local) if world > 1 else None
    torch.cuda.set_device(local)
    dev = torch.device(f"cuda:{local}")
    cfg = dict(CONFIGS[args.config])
    for k in ("dots", "envs", "resolution"):
        if getattr(args, k) is not None:
            cfg[k] = getattr(args, k)
    N, R, B = cfg["dots"], cfg["resolution"], cfg["envs"]
    seed = 1234
    mixed = args.config == "mixed"
    noise = True if args.noise else None
    if mixed:
        counts = {n: (B // 4) * world for n in (2, 4, 6, 8)}            # equal shares, whole job
        env = MixedVecQuantumDeviceEnv(counts, resolution=R, seed=seed, rank=rank, world=world, device=dev,
                                       capacitance_model_factory=lambda n: SyntheticCapacitanceModel(99 + rank + 10 * n),
                                       noise=["latch"])
        parts = env.buckets
        B = env.num_envs
    else:
        first_env, _ = shard.shard_env_ids(rank, world, B)
        if args.cnn == "synthetic":
            cap_model = SyntheticCapacitanceModel(99 + rank)
        else:
            from qadapt_hip.capacitance_cnn import build_device_model
            cap_model = build_device_model(backbone=args.cnn, device=dev, seed=7, dtype=getattr(torch, args.cnn_dtype))
        env = VecQuantumDeviceEnv(B, num_dots=N, resolution=R, device=dev, seed=seed, env_id_offset=first_env,
                                  capacitance_model=cap_model, noise=noise)
        parts = {N: env}
    gen = torch.Generator(device="cpu").manual_seed(99 + rank)
    env.reset()
    if not args.no_stagger:
        env.stagger_episodes()

    def one_step():
        if mixed:
            env.step({n: (torch.rand((e.B, 2 * n - 1), generator=gen) * 2 - 1).to(dev) for n, e in parts.items()},
                     auto_reset=True)
        else:
            env.step((torch.rand((B, 2 * N - 1), generator=gen) * 2 - 1).to(dev), auto_reset=True)

    for _ in range(args.warmup):
        one_step()
    torch.cuda.synchronize(dev)
    if dist:
        dist.barrier()
    torch.cuda.synchronize(dev)
    t0 = time.perf_counter()
    n_resets = 0
    for _ in range(args.steps):
        one_step()
        n_resets += sum(int((e._steps_host == 0).sum()) for e in parts.values())
    torch.cuda.synchronize(dev)
    if dist:
        dist.barrier()
    torch.cuda.synchronize(dev)
    dt = time.perf_counter() - t0
    # every rank reports its own rate (stderr): a slow rank / an unbalanced shard shows up here, the JSON line carries the max
    print(f"[bench rank {rank}/{world}] {B} envs on cuda:{local}: {B * args.steps / dt:.1f} env-steps/s ({dt / args.steps * 1e3:.1f} ms per step)",
          file=sys.stderr, flush=True)
    dt = shard.max_over_ranks(dt, device=dev if args.backend == "nccl" else None)   # identity when not distributed

    if rank == 0:
        total_env_steps = B * world * args.steps
        value = total_env_steps / dt
        # every hot kernel of the heaviest bucket by itself: HIP events inside the library, on the launch stream, one launch
        # chunk (`chunk` envs) per launch, on the state the timed steps left behind (the random-action regime)
        Nk = max(parts); ek = parts[Nk]
        kms = ek.time_kernels(iters=2)
        chunk = ek.chunk_envs()
        dom = max(kms, key=kms.get); dom_ms = kms[dom]
        achieved = b_alg(Nk, R) * chunk / (dom_ms * 1e-3) / 1e9
        prof = profile_figures(Nk, R)
        pk = ((prof or {}).get("kernels") or {}).get(dom) or {}
        traffic = pk.get("hbm_bytes_per_env_step")
        if traffic is not None:
            traffic = traffic * chunk
        if mixed:
            workload = (f"mixed N in {{2,4,6,8}} ragged batch, {B} envs per GPU ({B // 4} per dot count), {R}x{R} CSD, "
                        "latched-state model on, buckets on separate HIP streams")
            metric = f"env steps/sec (mixed 2/4/6/8-dot {R}x{R} CSD solves, latching on)"
        else:
            workload = (f"{N}-dot array, {B} parallel envs per GPU, {R}x{R} CSD, "
                        + ("all stochastic stages on" if args.noise else "deterministic physics (latching/noise off)"))
            metric = ("env steps/sec (batched 8-dot 64x64 CSD solves)" if (N, R) == (8, 64)
                      else f"env steps/sec ({N}-dot {R}x{R} CSD solves)")
        workload += (", 50-step episodes, staggered phases: " if not args.no_stagger else ", 50-step episodes in phase: ")
        workload += f"{n_resets} episode resets (new device + first observation) inside the timed region, "
        workload += ("synthetic CNN outputs" if args.cnn == "synthetic" else
                     f"{args.cnn} capacitance CNN ({args.cnn_dtype}, random weights) on the device inside the step")
        out = {
            "metric": metric, "value": value, "unit": "env-steps/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": dt / args.steps * 1e3, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "config": {"workload": workload, "n_dots": N if not mixed else [2, 4, 6, 8], "envs_per_gpu": B,
                       "resolution": R, "resets_in_timed_region": n_resets,
                       "target": "BASELINE.json asks >= 1e6 env-steps/s; the faithful float64 algorithm is far below it "
                                 "(see DESIGN.md 9), this line reports what is measured"},
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "traffic": traffic,
                         "kernel": f"{dom}<{Nk}>", "kernel_ms": dom_ms, "envs_per_launch": chunk,
                         "algorithmic_bytes_per_env_step": b_alg(Nk, R),
                         "kernels_ms": {f"{k}<{Nk}>" if k != "qd_k_gs_solve" else "qd_k_gs_solve<all size classes>": v for k, v in kms.items()},
                         "traffic_source": (None if traffic is None else
                                            {"file": "profiles/counters.json", "profile": prof.get("profile"),
                                             "git_rev": prof.get("git_rev"), "kernel_src_sha": prof.get("kernel_src_sha")}),
                         "note": "the faithful path is float64 VALU/LDS bound, not HBM bound (SURVEY 7-H1); "
                                 "the HBM fraction is reported as the contract asks, the VALU figures below say how busy "
                                 "the pipeline that binds is; see DESIGN.md"},
            # The roofline that binds: float64 vector issue.  From the rocprofv3 PMC passes under profiles/ (quoted only when
            # they were taken on THIS kernel source): per hot kernel the vector-ALU issue utilisation (SQ_ACTIVE_INST_VALU x 4
            # cycles / (duration of the same profiled dispatch x 2.4 GHz x 1024 SIMDs)), the enabled-lane fraction
            # (SQ_THREAD_CYCLES_VALU / (64 x SQ_ACTIVE_INST_VALU)) and the achieved lane-instruction rate against the peak of
            # 256 CUs x 4 SIMDs x 16 lanes/clk x 2.4 GHz = 39.3e12 lane-instructions/s (= 78.6 TFLOP/s if every one were an FMA).
            "valu": ({"peak_lane_instr_per_s": FP64_VECTOR_PEAK_TFLOPS / 2 * 1e12, "fp64_vector_peak_tflops": FP64_VECTOR_PEAK_TFLOPS,
                      "kernels": {k: {f: v.get(f) for f in ("valu_issue_utilisation", "enabled_lane_fraction", "lane_instr_per_s",
                                                             "frac_of_peak_lane_instr", "insts_valu_per_pixel", "profiled_ms")}
                                  for k, v in prof["kernels"].items()},
                      "source": {"file": "profiles/counters.json", "profile": prof.get("profile"), "workload": prof.get("workload")}}
                     if prof else {"peak_lane_instr_per_s": FP64_VECTOR_PEAK_TFLOPS / 2 * 1e12, "kernels": None,
                                   "note": "no PMC profile of this kernel source under profiles/ (scripts/collect_profiles.sh)"}),
            "kernel_src_sha": kernel_source_hash(), "git_rev": git_rev(),
        }
        if not args.no_cpu_baseline and world == 1:          # reported at N = 1 only
            base, parity = cpu_baseline_and_parity(Nk, R, seed, args.cpu_seconds)
            out["cpu_baseline"] = base
            out["parity"] = parity
        print(json.dumps(out), flush=True)
    env.close()
    if dist:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
