"""profiles/counters.json from a PMC summary (scripts/pmc_summary.py output): per hot kernel the instruction counts, the
vector-ALU issue utilisation, the enabled-lane fraction, the achieved lane-instruction rate and the HBM bytes per env-step,
tagged with the profile file, the git revision and the hash of the kernel sources they were measured on (bench.py only
quotes them when that hash equals the current sources').  Counters and durations come from the SAME profiled dispatches
(the DURATION_NS rows of the summary); the clock is taken as 2.4 GHz.

    python scripts/make_counters_json.py gpurun_out/<tag> profiles/<prefix> <N> <R> <envs_per_dispatch> "<workload text>"
"""
import csv, json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench

src, prefix, N, R, envs = sys.argv[1], sys.argv[2], int(sys.argv[3]), int(sys.argv[4]), int(sys.argv[5])
workload = sys.argv[6] if len(sys.argv) > 6 else ""
rows = list(csv.DictReader(open(os.path.join(src, "pmc_summary.csv"))))
def vals(kernel_sub, counter):
    return [(r["pass"], float(r["mean_per_dispatch"])) for r in rows if kernel_sub in r["kernel"] and r["counter"] == counter]
def val(kernel_sub, counter):
    v = vals(kernel_sub, counter)
    return v[0][1] if v else None
def dur_of_pass(kernel_sub, counter):
    """duration of the dispatches in the pass that collected `counter`"""
    p = vals(kernel_sub, counter)
    if not p:
        return None
    d = [x for pas, x in vals(kernel_sub, "DURATION_NS") if pas == p[0][0]]
    return d[0] if d else None
CLOCK, SIMDS = 2.4e9, 1024
pixels = envs * (N - 1) * R * R
kern = {"qd_k_tile": f"qd_k_tile<{N}>", "qd_k_candidates": f"qd_k_candidates<{N}, true>", "qd_k_gs_structure": f"qd_k_gs_structure<{N}, false, 4>",
        "qd_k_gs_select": f"qd_k_gs_select<{N}, false>"}
for b in range(10):
    kern[f"qd_k_gs_solve<{b}>"] = f"qd_k_gs_solve<{b}, false>"
out = {"profile": os.path.basename(prefix) + "_pmc_summary.csv", "git_rev": bench.git_rev(), "kernel_src_sha": bench.kernel_source_hash(),
       "envs_per_dispatch": envs, "workload": workload,
       "method": "rocprofv3 --pmc, one counter set per run, the LAST dispatch of every kernel (the micro-benchmark's timed launch); "
                 "HBM bytes = (2*FETCH_SIZE + WRITE_SIZE)*1024 per dispatch (gfx950 counts 128-B read requests at 64 B, "
                 "MI355X_MICROARCH.md); VALU issue utilisation = SQ_ACTIVE_INST_VALU (quad-cycles)*4 / (duration of the same "
                 "profiled dispatch * 2.4 GHz * 1024 SIMDs); enabled-lane fraction = SQ_THREAD_CYCLES_VALU / (64 * SQ_ACTIVE_INST_VALU); "
                 "lane-instruction rate = SQ_INSTS_VALU * 64 * enabled-lane fraction / duration",
       "kernels": {}}
tot = 0.0
for name, k in kern.items():
    f, w_ = val(k, "FETCH_SIZE"), val(k, "WRITE_SIZE")
    act, thr, ins = val(k, "SQ_ACTIVE_INST_VALU"), val(k, "SQ_THREAD_CYCLES_VALU"), val(k, "SQ_INSTS_VALU")
    if ins is None:
        continue
    ent = {"kernel": k, "insts_valu": ins, "insts_salu": val(k, "SQ_INSTS_SALU"), "insts_lds": val(k, "SQ_INSTS_LDS"),
           "insts_vmem_rd": val(k, "SQ_INSTS_VMEM_RD"), "insts_vmem_wr": val(k, "SQ_INSTS_VMEM_WR"),
           "wave_quad_cycles": val(k, "SQ_WAVE_CYCLES"), "active_inst_valu_quad_cycles": act, "thread_cycles_valu": thr,
           "wait_any_quad_cycles": val(k, "SQ_WAIT_ANY"), "lds_bank_conflict_cycles": val(k, "SQ_LDS_BANK_CONFLICT"),
           "waves": val(k, "SQ_WAVES"), "insts_valu_per_pixel": ins / pixels}
    if f is not None and w_ is not None:
        ent["hbm_bytes_per_env_step"] = (2 * f + w_) * 1024 / envs
        tot += ent["hbm_bytes_per_env_step"]
    d_act = dur_of_pass(k, "SQ_ACTIVE_INST_VALU")
    if act and d_act:
        ent["profiled_ms"] = d_act * 1e-6
        ent["valu_issue_utilisation"] = act * 4 / (d_act * 1e-9 * CLOCK * SIMDS)
    if act and thr:
        ent["enabled_lane_fraction"] = thr / (64.0 * act)
    d_ins = dur_of_pass(k, "SQ_INSTS_VALU")
    if d_ins and ent.get("enabled_lane_fraction"):
        ent["lane_instr_per_s"] = ins * 64 * ent["enabled_lane_fraction"] / (d_ins * 1e-9)
        ent["frac_of_peak_lane_instr"] = ent["lane_instr_per_s"] / (bench.FP64_VECTOR_PEAK_TFLOPS / 2 * 1e12)
    out["kernels"][name] = ent
out["hbm_bytes_per_env_step_all_hot_kernels"] = tot
path = os.path.join(ROOT, "profiles", "counters.json")
d = json.load(open(path)) if os.path.exists(path) else {}
d[f"{N}dot_{R}"] = out
json.dump(d, open(path, "w"), indent=1)
for k, v in out["kernels"].items():
    print(k, {f: (round(v[f], 4) if isinstance(v.get(f), float) else v.get(f)) for f in ("profiled_ms", "valu_issue_utilisation", "enabled_lane_fraction", "frac_of_peak_lane_instr", "insts_valu_per_pixel", "hbm_bytes_per_env_step")})
print("HBM bytes per env-step, all hot kernels:", tot)
