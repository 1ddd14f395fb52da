# round 5, check 5: quadtree (ORB-SLAM) mode from the score plane -- parity and timing against the image-based cells (GSLAM_HIP_QT_PLANE=0)
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_orb_gpu.py tests/test_orb_adversarial_gpu.py -m gpu -q -x --tb=short > gpurun_out/r5c5_t.log 2>&1; echo "orb tests rc=$?" > gpurun_out/r5c5_rc.log
for m in 1 0; do echo "== GSLAM_HIP_QT_PLANE=$m"; GSLAM_HIP_QT_PLANE=$m timeout 400 python tools/r4_quadtree_prof.py 2>&1 | grep -v amdgpu.ids; done > gpurun_out/r5c5_perf.log 2>&1
cat gpurun_out/r5c5_rc.log; tail -5 gpurun_out/r5c5_t.log; cat gpurun_out/r5c5_perf.log | grep "==\|mode quadtree\|orb_slam\|orb_fast_plane\|orb_resize\|orb_describe" 
