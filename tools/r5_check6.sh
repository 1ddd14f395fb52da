# round 5, check 6: where the quadtree mode's time goes (wall clock, experiment switches)
mkdir -p gpurun_out
for e in "" nostore; do GSLAM_HIP_QT_EXP=$e timeout 300 python tools/r5_qt_exp.py 2>&1 | grep -v amdgpu.ids; done > gpurun_out/r5c6_perf.log 2>&1
cat gpurun_out/r5c6_perf.log
R=$(pwd); cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_qt -- python $R/tools/r5_qt_exp.py > $R/gpurun_out/prof_qt.log 2>&1
cd $R; find gpurun_out/prof_qt -name "*kernel_trace.csv" -delete; find gpurun_out/prof_qt -name "*kernel_stats.csv" | head -1 | xargs -I{} sh -c 'cut -c1-90 {} | head -14'
