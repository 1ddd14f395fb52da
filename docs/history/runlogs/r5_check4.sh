# round 5, check 4: fused resize of interior tiles on MFMA (GSLAM_HIP_ORB_RESIZE_MFMA) -- parity and A/B timing
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_orb_gpu.py tests/test_orb_adversarial_gpu.py tests/test_orb_stream_gpu.py tests/test_stereo_gpu.py -m gpu -q -x --tb=short > gpurun_out/r5c4_t.log 2>&1; echo "orb tests rc=$?" > gpurun_out/r5c4_rc.log
for m in 1 0 1 0; do echo "== GSLAM_HIP_ORB_RESIZE_MFMA=$m"; GSLAM_HIP_ORB_RESIZE_MFMA=$m timeout 300 python tools/orb_perf.py 400 2>&1 | grep -v amdgpu.ids; done > gpurun_out/r5c4_perf.log 2>&1
cat gpurun_out/r5c4_rc.log; tail -5 gpurun_out/r5c4_t.log; grep "==\|extract 400\|orb_describe\|orb_fast" gpurun_out/r5c4_perf.log
