#!/bin/bash
# Kernel-level timeline of LM iterations of the C4 graph (rocprofv3 --kernel-trace): where the GPU idles between iterations.
#   bash tools/ba_iter_trace.sh   (through gpurun, from the repo root)
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
cat > /tmp/ba_iter.py <<PY
import sys
sys.path.insert(0, "$R")
from gslam_amd import hip, ba
from gslam_amd.ba_synth import make_graph
ctx = hip.Context(0)
g = make_graph(500, 50000, n_obs_per_point=6, seed=1)
G = ba.Graph(ctx, g, ba.default_options(max_iterations=12))
G.solve(ba.default_options(max_iterations=12))
G.update(cam_pose=g["cam_pose"], point_xyz=g["point_xyz"])
s, _ = G.solve(ba.default_options(max_iterations=12))
print("iterations", s.iterations, "total_ms", s.total_ms)
PY
rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/ba_iter_trace -- python /tmp/ba_iter.py > $R/gpurun_out/ba_iter_trace.log 2>&1
cd $R
f=$(ls -t gpurun_out/ba_iter_trace/*/*kernel_trace.csv | head -1)
python - "$f" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
# the last solve: from the last-but-... find last 12 lin_kernel launches
idx = [i for i, r in enumerate(rows) if "lin_kernel" in r["Kernel_Name"] and "reduce" not in r["Kernel_Name"]]
start = idx[-12]
rows = rows[start:]
t0 = int(rows[0]["Start_Timestamp"])
prev_end = None
busy = 0
iters = []
for r in rows:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    name = r["Kernel_Name"].split("(")[0].replace("void ", "").replace("(anonymous namespace)::", "")[:28]
    gap = (s - prev_end) / 1000 if prev_end is not None else 0.0
    if "lin_kernel" in r["Kernel_Name"] and "reduce" not in r["Kernel_Name"]:
        iters.append((s - t0) / 1000)
        print("---- iteration starts at %.1f us (gap before it %.1f us)" % ((s - t0) / 1000, gap))
    if gap > 3.0 and not ("lin_kernel" in r["Kernel_Name"] and "reduce" not in r["Kernel_Name"]):
        print("   gap %.1f us before %s" % (gap, name))
    prev_end = max(prev_end or 0, e)
print("iteration period: %.1f us average" % ((iters[-1] - iters[0]) / (len(iters) - 1)))
PY
