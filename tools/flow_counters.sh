#!/bin/bash
# Counter evidence for ba_potrf_flow: is the launch waiting on memory (L2 misses / EA traffic from
# the hand-off polling and operand fetches) or on issue?  Separate --pmc passes, --kernel-trace only.
#   bash tools/flow_counters.sh     (through gpurun, from the repo root)
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out
cd /tmp && export TMPDIR=/tmp
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" \
           "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVES" \
           "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_READ_sum TCC_WRITE_sum" \
           "TCC_EA_RDREQ_sum TCC_EA_WRREQ_sum TCC_EA_RDREQ_32B_sum TCC_ATOMIC_sum" \
           "SQ_INST_CYCLES_VMEM SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM"; do
  tag=$(echo $set | cut -d' ' -f1)
  timeout 300 rocprofv3 --pmc $set --kernel-trace --output-format csv -d $O/prof_flow_$tag -- python $R/tools/ba_c4_probe.py > $O/prof_flow_$tag.log 2>&1
  echo "== $set: rc $?"
done
python - <<PY
import csv, glob, collections
acc = collections.defaultdict(lambda: collections.defaultdict(lambda: [0.0, 0]))
for path in glob.glob("$O/prof_flow_*/**/*_counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(path)):
        k = row["Kernel_Name"]
        for key in ("potrf_flow_kernel", "bwd_chain_kernel", "schur_blocks", "lin_kernel"):
            if key in k:
                a = acc[key][row["Counter_Name"]]
                a[0] += float(row["Counter_Value"]); a[1] += 1
for key, d in acc.items():
    print(key)
    for c, (v, n) in sorted(d.items()):
        print("   %-28s %16.1f per launch (%d launches)" % (c, v / n, n))
PY
find $O/prof_flow_* -type f ! -name "*.csv" -delete 2>/dev/null
find $O/prof_flow_* -name "*kernel_trace.csv" -delete 2>/dev/null
