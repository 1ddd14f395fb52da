# round 5, check 8: the band / arrowhead solver's dense top -- parity tests at every GSLAM_HIP_CR_TOP, then LM it/s by setting
mkdir -p gpurun_out
for t in 4 1 8; do GSLAM_HIP_CR_TOP=$t timeout 900 python -m pytest tests/test_cr_solver.py -m gpu -q -x --tb=short > gpurun_out/r5c8_t$t.log 2>&1; echo "top $t: cr tests rc=$?"; tail -2 gpurun_out/r5c8_t$t.log; done > gpurun_out/r5c8_rc.log 2>&1
for t in 1 2 4 8 4 1; do echo "== GSLAM_HIP_CR_TOP=$t"; GSLAM_HIP_CR_TOP=$t timeout 300 python tools/r5_arrow_perf.py 2>&1 | grep -v amdgpu.ids; done > gpurun_out/r5c8_perf.log 2>&1
cat gpurun_out/r5c8_rc.log; grep "==\|resident\|auto" gpurun_out/r5c8_perf.log
