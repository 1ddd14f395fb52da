#!/bin/bash
# Collects the rocprofv3 evidence behind profiles/ on the GPU box (run through gpurun from the repo root):
#   bash tools/collect_profiles.sh
# then, back in the authoring container:  python tools/parse_rocprof.py r03 1000 "<stats command>"
# Counter passes are separate runs with --kernel-trace only (FETCH_SIZE and WRITE_SIZE do not share a pass).
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out
cd /tmp && export TMPDIR=/tmp
LIGHT="--steps 2 --warmup 1 --no-cpu-baseline --no-ba --no-bow --no-c3 --no-c5 --no-host-fed --no-all-pairs-full --no-range"
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_stats -- python $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-c3 --no-c5 --no-host-fed --no-all-pairs-full > $O/prof_stats.log 2>&1
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/prof_fetch -- python $R/bench.py $LIGHT > $O/prof_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/prof_write -- python $R/bench.py $LIGHT > $O/prof_write.log 2>&1
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_WAVES --kernel-trace --output-format csv -d $O/prof_sq -- python $R/bench.py $LIGHT > $O/prof_sq.log 2>&1
# keep only the CSV summaries (the merge back is capped at 64 MiB)
find $O/prof_stats $O/prof_fetch $O/prof_write $O/prof_sq -type f ! -name "*.csv" -delete 2>/dev/null
find $O/prof_stats -name "*kernel_trace.csv" -delete 2>/dev/null
ls -la $O/prof_*/*/* 2>/dev/null | head -20
du -sh $O
