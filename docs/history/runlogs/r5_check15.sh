# round 5, check 15: structure of the border in the arrowhead solver -- parity tests, then LM it/s with / without (GSLAM_HIP_BA_ARROW_DENSE_BORDER=1)
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_cr_solver.py tests/test_ba_gpu.py -m gpu -q -x --tb=short > gpurun_out/r5c15_t.log 2>&1; echo "tests rc=$?" > gpurun_out/r5c15_rc.log
{ echo "== structure"; timeout 600 python tools/r5_arrow_perf.py --c5 2>&1 | grep -v amdgpu.ids; echo "== GSLAM_HIP_BA_ARROW_DENSE_BORDER=1"; GSLAM_HIP_BA_ARROW_DENSE_BORDER=1 timeout 600 python tools/r5_arrow_perf.py --c5 2>&1 | grep -v amdgpu.ids; } > gpurun_out/r5c15_perf.log 2>&1
cat gpurun_out/r5c15_rc.log; tail -3 gpurun_out/r5c15_t.log; grep "==\|arrow\|resident" gpurun_out/r5c15_perf.log
