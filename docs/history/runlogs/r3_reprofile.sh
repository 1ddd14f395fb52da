#!/bin/bash
# Round-3 closing re-collection after the packed arc score changed orb_fast_cells: SQ counters (one --pmc pass) and the
# rocprofv3 --stats summary, same commands as tools/collect_profiles.sh (FETCH_SIZE / WRITE_SIZE are not re-collected: the
# kernel's loads and stores did not change).
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out
mkdir -p $O
rm -rf $O/prof_stats $O/prof_sq
cd /tmp && export TMPDIR=/tmp
LIGHT="--steps 2 --warmup 1 --no-cpu-baseline --no-ba --no-bow --no-c3 --no-c5 --no-host-fed --no-all-pairs-full --no-range"
t0=$(date +%s)
timeout 100 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_WAVES --kernel-trace --output-format csv -d $O/prof_sq -- python $R/bench.py $LIGHT > $O/prof_sq.log 2>&1
echo "sq rc=$? $(( $(date +%s) - t0 )) s"
timeout 110 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_stats -- python $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-c3 --no-c5 --no-host-fed --no-all-pairs-full --no-range > $O/prof_stats.log 2>&1
echo "stats rc=$? $(( $(date +%s) - t0 )) s"
find $O/prof_stats $O/prof_sq -type f ! -name "*.csv" -delete 2>/dev/null
find $O/prof_stats -name "*kernel_trace.csv" -delete 2>/dev/null
tail -c 600 $O/prof_sq.log | tail -2 | cut -c1-400
tail -c 600 $O/prof_stats.log | tail -2 | cut -c1-400
du -sh $O
