# round 5, check 7: score plane with 64-byte aligned tile rows -- ORB parity tests, then wall clock + rocprofv3 stats of the quadtree mode
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_orb_gpu.py tests/test_orb_adversarial_gpu.py -m gpu -q -x --tb=short > gpurun_out/r5c7_t.log 2>&1; echo "orb tests rc=$?" > gpurun_out/r5c7_rc.log
timeout 300 python tools/r5_qt_exp.py 2>&1 | grep -v amdgpu.ids > gpurun_out/r5c7_perf.log
R=$(pwd); cd /tmp && export TMPDIR=/tmp
rm -rf $R/gpurun_out/prof_qt
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_qt -- python $R/tools/r5_qt_exp.py > $R/gpurun_out/prof_qt.log 2>&1
cd $R; find gpurun_out/prof_qt -name "*kernel_trace.csv" -delete
cat gpurun_out/r5c7_rc.log; tail -3 gpurun_out/r5c7_t.log; cat gpurun_out/r5c7_perf.log
