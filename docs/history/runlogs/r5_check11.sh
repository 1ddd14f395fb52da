# round 5, check 11: steer describe at 8 waves per SIMD -- parity (quadtree / steering tests), then timing
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_orb_gpu.py -m gpu -q -x --tb=short > gpurun_out/r5c11_t.log 2>&1; echo "orb tests rc=$?" > gpurun_out/r5c11_rc.log
{ for i in 1 2; do timeout 400 python tools/r4_quadtree_prof.py 2>&1 | grep -v amdgpu.ids; timeout 300 python tools/r5_qt_exp.py 2>&1 | grep -v amdgpu.ids; done; } > gpurun_out/r5c11_perf.log 2>&1
cat gpurun_out/r5c11_rc.log; tail -3 gpurun_out/r5c11_t.log; grep "mode quad\|orb_describe\|steer" gpurun_out/r5c11_perf.log
