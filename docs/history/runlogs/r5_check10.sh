# round 5, check 10: continuous-steering orb_describe with the blur on MFMA -- parity, then A/B by GSLAM_HIP_ORB_DESC_MFMA
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_orb_gpu.py tests/test_orb_adversarial_gpu.py tests/test_orb_stream_gpu.py tests/test_stereo_gpu.py -m gpu -q -x --tb=short > gpurun_out/r5c10_t.log 2>&1; echo "orb tests rc=$?" > gpurun_out/r5c10_rc.log
{ for m in 2 0 2; do echo "== GSLAM_HIP_ORB_DESC_MFMA=$m"; GSLAM_HIP_ORB_DESC_MFMA=$m timeout 400 python tools/r4_quadtree_prof.py 2>&1 | grep -v amdgpu.ids; GSLAM_HIP_ORB_DESC_MFMA=$m timeout 300 python tools/r5_qt_exp.py 2>&1 | grep -v amdgpu.ids; done; } > gpurun_out/r5c10_perf.log 2>&1
cat gpurun_out/r5c10_rc.log; tail -5 gpurun_out/r5c10_t.log; grep "==\|mode quad\|orb_describe\|steer" gpurun_out/r5c10_perf.log
