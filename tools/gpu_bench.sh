#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out
mkdir -p $O
cd $R
t0=$(date +%s)
timeout 1200 python bench.py "$@" > $O/bench_r6.json 2> $O/bench_r6.err; echo "bench rc=$? $(( $(date +%s) - t0 )) s, line $(wc -c < $O/bench_r6.json) bytes"
tail -4 $O/bench_r6.err
cat $O/bench_r6.json
