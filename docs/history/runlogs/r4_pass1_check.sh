# round-4 A/B of the SWAR pass 1 of orb_fast_cells (GSLAM_HIP_ORB_PASS1 = 0 packed 16-bit, 1 SWAR in registers, 2 SWAR on LDS planes):
# ORB parity tests for the two new variants, extraction throughput for all three (twice each)
mkdir -p gpurun_out
T="tests/test_orb_gpu.py tests/test_orb_adversarial_gpu.py tests/test_orb_stream_gpu.py"
GSLAM_HIP_ORB_PASS1=2 timeout 200 python -m pytest $T -m gpu -q --tb=short -x > gpurun_out/p1_t2.log 2>&1; echo "pass1=2 rc=$?" > gpurun_out/p1_rc.log
GSLAM_HIP_ORB_PASS1=1 timeout 200 python -m pytest $T -m gpu -q --tb=short -x > gpurun_out/p1_t1.log 2>&1; echo "pass1=1 rc=$?" >> gpurun_out/p1_rc.log
for v in 0 1 2 0 1 2; do GSLAM_HIP_ORB_PASS1=$v timeout 60 python tools/orb_perf.py 400 > gpurun_out/p1_perf_$v.txt 2>&1; echo "perf $v rc=$?" >> gpurun_out/p1_rc.log; grep -h "extract\|orb_fast_cells" gpurun_out/p1_perf_$v.txt | tail -2 >> gpurun_out/p1_rc.log; done
cat gpurun_out/p1_rc.log; tail -5 gpurun_out/p1_t2.log; tail -5 gpurun_out/p1_t1.log
