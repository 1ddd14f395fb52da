#!/bin/bash
# SQ counters of the MFMA matcher (and the popcount kernel beside it): where do a wave's cycles go?
#   bash tools/bf_mfma_counters.sh     (through gpurun, from the repo root)
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out
cd /tmp && export TMPDIR=/tmp
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" \
           "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_WAVES" \
           "SQ_INST_CYCLES_VMEM SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM" \
           "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_SCA"; do
  tag=$(echo $set | cut -d' ' -f1)
  timeout 200 rocprofv3 --pmc $set --kernel-trace --output-format csv -d $O/prof_bfm_$tag -- python $R/tools/bf_mfma_perf.py 32 > $O/prof_bfm_$tag.log 2>&1
  echo "== $set: rc $?"
done
python - <<PY
import csv, glob, collections
acc = collections.defaultdict(lambda: collections.defaultdict(lambda: [0.0, 0]))
for path in glob.glob("$O/prof_bfm_*/**/*_counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(path)):
        k = row["Kernel_Name"]
        for key in ("bf_match_pairs_mfma_kernel", "bf_match_pairs_kernel"):
            if key in k:
                a = acc[key][row["Counter_Name"]]
                a[0] += float(row["Counter_Value"]); a[1] += 1
for key, d in acc.items():
    print(key)
    for c, (v, n) in sorted(d.items()):
        print("   %-28s %16.1f per launch (%d launches)" % (c, v / n, n))
PY
find $O/prof_bfm_* -type f ! -name "*.csv" -delete 2>/dev/null
find $O/prof_bfm_* -name "*kernel_trace.csv" -delete 2>/dev/null
