"""profiles/counters.json from a PMC summary (scripts/pmc_summary.py output) and the kernel micro-benchmark log of the
same run: per-config HBM traffic per env-step and vector-ALU issue utilisation of the hot kernels, tagged with the
profile files, the git revision and the hash of the kernel sources they were measured on (bench.py only quotes them
when that hash equals the current sources').

    python scripts/make_counters_json.py gpurun_out/<tag> profiles/<prefix> <N> <R> <envs_per_dispatch>
"""
import csv, json, os, re, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench

src, prefix, N, R, envs = sys.argv[1], sys.argv[2], int(sys.argv[3]), int(sys.argv[4]), int(sys.argv[5])
rows = list(csv.DictReader(open(os.path.join(src, "pmc_summary.csv"))))
def val(kernel_sub, counter):
    v = [float(r["mean_per_dispatch"]) for r in rows if kernel_sub in r["kernel"] and r["counter"] == counter]
    return v[0] if v else None
log = open(os.path.join(src, "kbench_64env_start.log")).read()
m = re.search(r"candidates\s+([0-9.]+) ms.*ground\s+([0-9.]+) ms", log)
cand_ms, ground_ms = float(m.group(1)), float(m.group(2))
kern = {"ground": (f"qd_k_ground<{N}, false>", ground_ms), "tile_search": (f"qd_k_tile<{N}, 0, false>", None),
        "pixel_search_redo": (f"qd_k_candidates<{N}>", None)}
out = {"profile": os.path.basename(prefix) + "_pmc_summary.csv", "git_rev": bench.git_rev(), "kernel_src_sha": bench.kernel_source_hash(),
       "envs_per_dispatch": envs, "workload": f"scripts/kbench.py --envs {envs} --modes start ({N}-dot, {R}x{R}, random start voltages)",
       "method": "rocprofv3 --pmc, one counter set per run; HBM bytes = (2*FETCH_SIZE + WRITE_SIZE)*1024 per dispatch (gfx950 counts "
                 "128-B read requests at 64 B, MI355X_MICROARCH.md); VALU issue utilisation = SQ_ACTIVE_INST_VALU (quad-cycles)*4 / "
                 "(kernel duration * 2.4 GHz * 1024 SIMDs), duration = HIP-event time of the same launch without the profiler",
       "kernels": {}}
tot = 0.0
for name, (k, ms) in kern.items():
    f, w_ = val(k, "FETCH_SIZE"), val(k, "WRITE_SIZE")
    ent = {"kernel": k, "insts_valu": val(k, "SQ_INSTS_VALU"), "insts_salu": val(k, "SQ_INSTS_SALU"), "insts_lds": val(k, "SQ_INSTS_LDS"),
           "wave_quad_cycles": val(k, "SQ_WAVE_CYCLES"), "active_inst_valu_quad_cycles": val(k, "SQ_ACTIVE_INST_VALU"),
           "wait_any_quad_cycles": val(k, "SQ_WAIT_ANY"), "lds_bank_conflict_cycles": val(k, "SQ_LDS_BANK_CONFLICT"),
           "lds_idx_active_cycles": val(k, "SQ_LDS_IDX_ACTIVE"), "waves": val(k, "SQ_WAVES")}
    if f is not None and w_ is not None:
        ent["hbm_bytes_per_env_step"] = (2 * f + w_) * 1024 / envs
        tot += ent["hbm_bytes_per_env_step"]
    if ms and ent["active_inst_valu_quad_cycles"]:
        ent["launch_ms"] = ms
        ent["valu_issue_utilisation"] = ent["active_inst_valu_quad_cycles"] * 4 / (ms * 1e-3 * 2.4e9 * 1024)
    out["kernels"][name] = ent
out["hbm_bytes_per_env_step"] = out["kernels"]["ground"].get("hbm_bytes_per_env_step")
out["hbm_bytes_per_env_step_all_hot_kernels"] = tot
out["valu_issue_utilisation"] = {k: round(v["valu_issue_utilisation"], 3) for k, v in out["kernels"].items() if "valu_issue_utilisation" in v}
path = os.path.join(ROOT, "profiles", "counters.json")
d = json.load(open(path)) if os.path.exists(path) else {}
d[f"{N}dot_{R}"] = out
json.dump(d, open(path, "w"), indent=1)
print(json.dumps(out["valu_issue_utilisation"]), out["hbm_bytes_per_env_step"], tot)
