R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out
mkdir -p $O; rm -rf $O/prof_fetch $O/prof_write
cd /tmp && export TMPDIR=/tmp
LIGHT="--steps 2 --warmup 1 --no-cpu-baseline --no-ba --no-bow --no-c3 --no-c5 --no-host-fed --no-all-pairs-full --no-range"
timeout 60 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/prof_fetch -- python $R/bench.py $LIGHT > $O/prof_fetch.log 2>&1; echo "fetch rc=$?"
timeout 60 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/prof_write -- python $R/bench.py $LIGHT > $O/prof_write.log 2>&1; echo "write rc=$?"
find $O/prof_fetch $O/prof_write -type f ! -name "*.csv" -delete 2>/dev/null
du -sh $O
