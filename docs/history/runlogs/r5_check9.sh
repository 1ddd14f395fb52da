# round 5, check 9: rank by one key comparison in orb_fast_cells; plane cells up to 40 x 40 -- parity, then timing
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_orb_gpu.py tests/test_orb_adversarial_gpu.py tests/test_orb_stream_gpu.py tests/test_stereo_gpu.py -m gpu -q -x --tb=short > gpurun_out/r5c9_t.log 2>&1; echo "orb tests rc=$?" > gpurun_out/r5c9_rc.log
{ for i in 1 2; do timeout 300 python tools/orb_perf.py 400 2>&1 | grep -v amdgpu.ids; done; timeout 400 python tools/r4_quadtree_prof.py 2>&1 | grep -v amdgpu.ids; timeout 300 python tools/r5_qt_exp.py 2>&1 | grep -v amdgpu.ids; } > gpurun_out/r5c9_perf.log 2>&1
cat gpurun_out/r5c9_rc.log; tail -5 gpurun_out/r5c9_t.log; grep "extract 400\|orb_describe\|orb_fast\|mode\|orb_slam\|steer" gpurun_out/r5c9_perf.log
