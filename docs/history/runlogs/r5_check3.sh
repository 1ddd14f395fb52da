# round 5, check 3: arrowhead solver inside gh_ba_solve (loop closures) -- parity with the dense solver, then timings
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_cr_solver.py tests/test_ba_gpu.py -m gpu -q -x --tb=short > gpurun_out/r5c3_t.log 2>&1; echo "ba tests rc=$?" > gpurun_out/r5c3_rc.log
timeout 600 python tools/r5_arrow_perf.py > gpurun_out/r5c3_perf.log 2>&1; echo "perf rc=$?" >> gpurun_out/r5c3_rc.log
cat gpurun_out/r5c3_rc.log; tail -15 gpurun_out/r5c3_t.log; cat gpurun_out/r5c3_perf.log | grep -v amdgpu.ids | tail -30
