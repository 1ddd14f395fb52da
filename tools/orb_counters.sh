#!/bin/bash
# SQ counter picture of orb_fast_cells for the pass-1 variants (GSLAM_HIP_ORB_PASS1): where do the wave-cycles go --
# issue (ACTIVE_INST_*), issue stalls (WAIT_INST_*), parked waves (WAIT_ANY: waitcnt / barrier) -- and the effective clock
# (GRBM_GUI_ACTIVE / duration).  Separate --pmc passes, --kernel-trace only.
#   bash tools/orb_counters.sh "0 1"      (through gpurun, from the repo root; argument = variants)
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out
V=${1:-"0 1"}
cd /tmp && export TMPDIR=/tmp
for v in $V; do
  i=0
  for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS GRBM_GUI_ACTIVE" \
             "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVES SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INST_CYCLES_SALU" \
             "SQ_INST_CYCLES_VALU SQ_THREAD_CYCLES_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL SQ_INSTS_BRANCH SQ_BUSY_CU_CYCLES"; do
    i=$((i+1))
    GSLAM_HIP_ORB_PASS1=$v timeout 120 rocprofv3 --pmc $set --kernel-trace --output-format csv -d $O/prof_orbc_${v}_$i -- python $R/tools/orb_perf.py 400 > $O/prof_orbc_${v}_$i.log 2>&1
    echo "== variant $v set $i: rc $?"
  done
done
python - <<PY
import csv, glob, collections, re
acc = collections.defaultdict(lambda: collections.defaultdict(lambda: [0.0, 0]))
dur = collections.defaultdict(lambda: [0.0, 0])
for path in glob.glob("$O/prof_orbc_*/**/*_counter_collection.csv", recursive=True):
    v = re.search(r"prof_orbc_(\d+)_", path).group(1)
    for row in csv.DictReader(open(path)):
        k = row["Kernel_Name"]
        for key in ("fast_cells_kernel", "describe_kernel"):
            if key in k:
                a = acc[(v, key)][row["Counter_Name"]]
                a[0] += float(row["Counter_Value"]); a[1] += 1
for path in glob.glob("$O/prof_orbc_*/**/*_kernel_trace.csv", recursive=True):
    v = re.search(r"prof_orbc_(\d+)_", path).group(1)
    for row in csv.DictReader(open(path)):
        for key in ("fast_cells_kernel", "describe_kernel"):
            if key in row["Kernel_Name"]:
                d = dur[(v, key)]
                d[0] += float(row["End_Timestamp"]) - float(row["Start_Timestamp"]); d[1] += 1
for key, d in sorted(acc.items()):
    print("variant %s %s   avg duration %.1f us (%d launches)" % (key[0], key[1], dur[key][0] / max(1, dur[key][1]) / 1e3, dur[key][1]))
    for c, (v, n) in sorted(d.items()):
        print("   %-28s %16.1f per launch (%d launches)" % (c, v / n, n))
PY
find $O/prof_orbc_* -type f ! -name "*.csv" -delete 2>/dev/null
find $O/prof_orbc_* -name "*kernel_trace.csv" -delete 2>/dev/null
