# round 5, check 12: quadtree kernel table capacity 512 / 1024 / 2048 (GSLAM_HIP_QT_NODES) -- parity, then timing
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_orb_gpu.py -m gpu -q -x --tb=short > gpurun_out/r5c12_t.log 2>&1; echo "orb tests rc=$?" > gpurun_out/r5c12_rc.log
GSLAM_HIP_QT_NODES=1024 timeout 900 python -m pytest tests/test_orb_gpu.py -m gpu -q -x --tb=short -k quadtree > gpurun_out/r5c12_t2.log 2>&1; echo "orb tests (1024) rc=$?" >> gpurun_out/r5c12_rc.log
{ for m in 512 1024 2048 512; do echo "== GSLAM_HIP_QT_NODES=$m"; GSLAM_HIP_QT_NODES=$m timeout 400 python tools/r4_quadtree_prof.py 2>&1 | grep -v amdgpu.ids; GSLAM_HIP_QT_NODES=$m timeout 300 python tools/r5_qt_exp.py 2>&1 | grep -v amdgpu.ids; done; } > gpurun_out/r5c12_perf.log 2>&1
cat gpurun_out/r5c12_rc.log; tail -3 gpurun_out/r5c12_t.log; tail -2 gpurun_out/r5c12_t2.log; grep "==\|mode quad\|orb_slam_quadtree\|steer" gpurun_out/r5c12_perf.log
