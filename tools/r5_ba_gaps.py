"""After `rocprofv3 --kernel-trace --output-format csv -d gpurun_out/prof_ba -- python tools/r5_arrow_perf.py`: GPU busy time of the
last 12 ms of BA kernels on the main stream against the wall clock between them (how much of a resident C4 iteration the GPU idles)."""
import csv, glob, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
f = sorted(glob.glob(os.path.join(ROOT, "gpurun_out", "prof_ba", "**", "*kernel_trace.csv"), recursive=True), key=os.path.getmtime)[-1]
rows = [r for r in csv.DictReader(open(f))]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
# windows of activity: split where the gap exceeds 200 us (between solves)
wins, cur = [], [rows[0]]
for r in rows[1:]:
    if int(r["Start_Timestamp"]) - max(int(x["End_Timestamp"]) for x in cur[-8:]) > 200000:
        wins.append(cur)
        cur = []
    cur.append(r)
wins.append(cur)
for w in wins:
    if len(w) < 300:
        continue
    t0, t1 = int(w[0]["Start_Timestamp"]), max(int(x["End_Timestamp"]) for x in w)
    # union of busy intervals (any stream)
    iv = sorted((int(x["Start_Timestamp"]), int(x["End_Timestamp"])) for x in w)
    busy, s, e = 0, iv[0][0], iv[0][1]
    for a, b in iv[1:]:
        if a > e:
            busy += e - s
            s, e = a, b
        else:
            e = max(e, b)
    busy += e - s
    nlin = sum(1 for x in w if "lin_kernel" in x["Kernel_Name"] and "reduce" not in x["Kernel_Name"])
    print("window: %d kernels, %.3f ms wall, GPU busy %.3f ms (%.1f %%), %d linearisations -> %.1f us idle per iteration" %
          (len(w), (t1 - t0) / 1e6, busy / 1e6, 100.0 * busy / (t1 - t0), nlin, (t1 - t0 - busy) / 1e3 / max(1, nlin)))
