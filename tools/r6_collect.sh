#!/bin/bash
# Round-6 evidence collection on the GPU box (through gpurun, from the repo root): the default bench line (compact line + full record),
# the rocprofv3 --stats summary of the same command (light side legs off) and the counter passes (FETCH_SIZE / WRITE_SIZE / SQ, each
# its own run with --kernel-trace only).  Back in the authoring container: python tools/parse_rocprof.py r06 1000 "<stats command>".
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out
mkdir -p $O
rm -rf $O/prof_stats $O/prof_fetch $O/prof_write $O/prof_sq
cd $R
t0=$(date +%s)
timeout 1200 python $R/bench.py > $O/BENCH_r06_n1.json 2> $O/BENCH_r06_n1.err; echo "bench rc=$? $(( $(date +%s) - t0 )) s, line $(wc -c < $O/BENCH_r06_n1.json) bytes"
cp $O/bench_full.json $O/BENCH_r06_n1_full.json
cd /tmp && export TMPDIR=/tmp
LIGHT="--steps 2 --warmup 1 --no-cpu-baseline --no-ba --no-bow --no-c3 --no-c5 --no-host-fed --no-all-pairs-full --no-range"
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_stats -- python $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-c3 --no-c5 --no-host-fed --no-all-pairs-full --no-range > $O/prof_stats.log 2>&1; echo "stats rc=$? $(( $(date +%s) - t0 )) s"
timeout 150 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/prof_fetch -- python $R/bench.py $LIGHT > $O/prof_fetch.log 2>&1; echo "fetch rc=$? $(( $(date +%s) - t0 )) s"
timeout 150 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/prof_write -- python $R/bench.py $LIGHT > $O/prof_write.log 2>&1; echo "write rc=$? $(( $(date +%s) - t0 )) s"
timeout 150 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_WAVES --kernel-trace --output-format csv -d $O/prof_sq -- python $R/bench.py $LIGHT > $O/prof_sq.log 2>&1; echo "sq rc=$? $(( $(date +%s) - t0 )) s"
find $O/prof_stats $O/prof_fetch $O/prof_write $O/prof_sq -type f ! -name "*.csv" -delete 2>/dev/null
find $O/prof_stats -name "*kernel_trace.csv" -delete 2>/dev/null
tail -c 300 $O/BENCH_r06_n1.err
du -sh $O
