# round 5, check 1: orb_describe with the h-pass of the blur on MFMA -- parity (all ORB tests) and A/B timing
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_orb_gpu.py tests/test_orb_adversarial_gpu.py tests/test_orb_stream_gpu.py tests/test_stereo_gpu.py -m gpu -q -x --tb=short > gpurun_out/r5c1_t.log 2>&1; echo "orb tests rc=$?" > gpurun_out/r5c1_rc.log
for m in 1 0 1 0; do echo "== GSLAM_HIP_ORB_DESC_MFMA=$m"; GSLAM_HIP_ORB_DESC_MFMA=$m timeout 300 python tools/orb_perf.py 400 2>&1 | grep -v amdgpu.ids; done > gpurun_out/r5c1_perf.log 2>&1
cat gpurun_out/r5c1_rc.log; tail -5 gpurun_out/r5c1_t.log; grep "==\|extract 400\|orb_describe\|orb_fast" gpurun_out/r5c1_perf.log
