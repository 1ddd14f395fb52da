#!/bin/bash
# Round 6, first GPU session: the new BA tests, the ordering cost on the box's host, the bench line, orb_describe's counters.
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out
mkdir -p $O
cd $R
t0=$(date +%s)
timeout 1500 python -m pytest tests/test_ba_order_gpu.py tests/test_cr_solver.py "tests/test_full_configs_gpu.py" -x -q -m gpu -k "not c2_all_pairs and not pnp" > $O/r6_tests1.log 2>&1; echo "tests1 rc=$? $(( $(date +%s) - t0 )) s"
tail -5 $O/r6_tests1.log
for t in 1 4 8; do
  GSLAM_HIP_BA_ORDER_THREADS=$t GSLAM_HIP_BA_TIMING=1 timeout 300 python -m pytest tests/test_ba_order.py -q -s -k "c5" 2>&1 | grep "order\]\|C5" | sed "s/^/threads $t: /"
done > $O/r6_order_cost.log 2>&1
cat $O/r6_order_cost.log
t0=$(date +%s)
timeout 900 python bench.py > $O/bench_r6_a.json 2> $O/bench_r6_a.err; echo "bench rc=$? $(( $(date +%s) - t0 )) s, line $(wc -c < $O/bench_r6_a.json) bytes"
tail -3 $O/bench_r6_a.err
cd /tmp && export TMPDIR=/tmp
rocprofv3 --list-avail 2>/dev/null | grep -o "TCC_[A-Z0-9_]*" | sort -u | tr '\n' ' ' > $O/r6_tcc_counters.txt; echo; head -c 3000 $O/r6_tcc_counters.txt; echo
i=0
for set in "FETCH_SIZE" "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum" "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum" "TCC_EA0_RDREQ_DRAM_sum TCC_EA0_RD_UNCACHED_32B_sum"; do
  i=$((i+1))
  timeout 200 rocprofv3 --pmc $set --kernel-trace --output-format csv -d $O/prof_calib_$i -- $R/build/fetch_calib > $O/prof_calib_$i.log 2>&1; echo "calib set $i ($set) rc=$?"
  timeout 200 rocprofv3 --pmc $set --kernel-trace --output-format csv -d $O/prof_desc_$i -- python $R/tools/orb_perf.py 1000 > $O/prof_desc_$i.log 2>&1; echo "describe set $i rc=$?"
done
grep "patch_rows\|stream_kernel" $O/prof_calib_1.log
python - <<PY
import csv, glob, collections
for tag in ("calib", "desc"):
    acc = collections.defaultdict(lambda: collections.defaultdict(lambda: [0.0, 0]))
    for path in glob.glob("$O/prof_%s_*/**/*_counter_collection.csv" % tag, recursive=True):
        for row in csv.DictReader(open(path)):
            k = row["Kernel_Name"].split("(")[0][:60]
            if not any(s in k for s in ("patch_rows", "stream_kernel", "describe", "fast_cells", "select_kernel")):
                continue
            a = acc[k][row["Counter_Name"]]
            a[0] += float(row["Counter_Value"]); a[1] += 1
    for k, d in sorted(acc.items()):
        print(tag, k)
        for c, (v, n) in sorted(d.items()):
            print("   %-32s %18.1f per launch (%d launches)" % (c, v / n, n))
PY
find $O/prof_calib_* $O/prof_desc_* -type f ! -name "*counter_collection.csv" ! -name "*.log" -delete 2>/dev/null
