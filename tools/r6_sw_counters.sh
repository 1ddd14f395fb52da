#!/bin/bash
# SQ counters of the tile kernel (plane variant) and of the sliding-window kernel on the same frames (1080p x 300): instructions and
# where the wave-cycles go.  Separate --pmc passes, --kernel-trace only.
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out
cd /tmp && export TMPDIR=/tmp
i=0
for set in "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM GRBM_GUI_ACTIVE" \
           "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS"; do
  i=$((i+1))
  timeout 300 rocprofv3 --pmc $set --kernel-trace --output-format csv -d $O/prof_sw_$i -- python $R/tools/r6_sw_exp.py 0,1 > $O/prof_sw_$i.log 2>&1; echo "set $i rc=$?"
done
python - <<PY
import csv, glob, collections
acc = collections.defaultdict(lambda: collections.defaultdict(lambda: [0.0, 0]))
for path in glob.glob("$O/prof_sw_*/**/*_counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(path)):
        k = row["Kernel_Name"]
        key = "sliding window" if "fast_plane_sw" in k else ("tile (plane variant)" if "fast_cells_kernel" in k else None)
        if key is None or int(row["Grid_Size"]) < 3000000:  # (the 600-frame level-0 launches only)
            continue
        a = acc[key][row["Counter_Name"]]
        a[0] += float(row["Counter_Value"]); a[1] += 1
for key, d in sorted(acc.items()):
    print(key, "-- level-0 launches of 600 x 1080p")
    for c, (v, n) in sorted(d.items()):
        print("   %-24s %18.1f per launch (%d launches)" % (c, v / n, n))
PY
find $O/prof_sw_* -type f ! -name "*counter_collection.csv" ! -name "*.log" -delete 2>/dev/null
