"""Summarise rocprofv3 --pmc passes: one line per (pass, kernel, counter) with the mean value per dispatch, plus the mean
DURATION_NS of the same dispatches (End - Start timestamp of the profiled dispatch: counters and duration from one run).

    python scripts/pmc_summary.py gpurun_out/<tag>/pmc [--last K] > gpurun_out/<tag>/pmc_summary.csv

Each pass is a sub-directory of the argument holding rocprofv3's *_counter_collection.csv
(`rocprofv3 --pmc A B C -d <dir>/<pass> --output-format csv -- python3 scripts/kbench.py ...`, one counter set per run,
no trace options combined).  --last K keeps only the last K dispatches of every kernel (the micro-benchmark's timed
launches come last, on the state it was asked for; earlier dispatches belong to its set-up).  FETCH_SIZE / WRITE_SIZE are
KiB per dispatch; on gfx950 FETCH_SIZE under-reports wide reads by 2x (MI355X_MICROARCH.md, HBM section).  Kernel names
contain commas: the kernel column is quoted."""
import csv, glob, os, re, sys
from collections import defaultdict

args = [a for a in sys.argv[1:] if not a.startswith("--")]
root = args[0]
last = int(sys.argv[sys.argv.index("--last") + 1]) if "--last" in sys.argv else 0
rows_by = defaultdict(list)          # (pass, kernel, counter) -> [(dispatch id, value, duration)]
for path in sorted(glob.glob(os.path.join(root, "**", "*counter_collection.csv"), recursive=True)):
    pas = os.path.relpath(path, root).split(os.sep)[0]
    with open(path, newline="") as f:
        for row in csv.DictReader(f):
            k = re.sub(r"\(.*", "", row["Kernel_Name"]).strip()
            dur = float(row["End_Timestamp"]) - float(row["Start_Timestamp"]) if row.get("End_Timestamp") else 0.0
            rows_by[(pas, k, row["Counter_Name"])].append((int(row["Dispatch_Id"]), float(row["Counter_Value"]), dur))
w = csv.writer(sys.stdout)
w.writerow(["pass", "kernel", "counter", "dispatches", "mean_per_dispatch"])
seen_dur = set()
for (pas, k, c), lst in sorted(rows_by.items()):
    if not (k.startswith("void qd_k") or k.startswith("qd_k")):
        continue
    lst.sort()
    if last:
        lst = lst[-last:]
    w.writerow([pas, k, c, len(lst), sum(v for _, v, _ in lst) / len(lst)])
    if (pas, k) not in seen_dur:
        seen_dur.add((pas, k))
        w.writerow([pas, k, "DURATION_NS", len(lst), sum(d for _, _, d in lst) / len(lst)])
