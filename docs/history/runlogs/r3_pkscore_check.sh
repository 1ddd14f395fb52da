# round-3 A/B of the packed-fp16 arc score (GSLAM_HIP_ORB_PKSCORE): whole GPU suite with the default (on), the ORB parity
# tests with it off, extraction throughput both ways
mkdir -p gpurun_out
timeout 330 python -m pytest tests -m gpu -q --tb=short > gpurun_out/pk_t_all.log 2>&1; echo "all rc=$?" > gpurun_out/pk_rc.log
GSLAM_HIP_ORB_PKSCORE=0 timeout 120 python -m pytest tests/test_orb_gpu.py tests/test_orb_adversarial_gpu.py -m gpu -q --tb=short > gpurun_out/pk_t_off.log 2>&1; echo "off rc=$?" >> gpurun_out/pk_rc.log
for v in 1 0 1 0; do GSLAM_HIP_ORB_PKSCORE=$v timeout 60 python tools/orb_perf.py 400 > gpurun_out/pk_perf_$v.txt 2>&1; echo "perf $v rc=$?" >> gpurun_out/pk_rc.log; grep -h "extract\|orb_fast_cells" gpurun_out/pk_perf_$v.txt | tail -2 >> gpurun_out/pk_rc.log; done
cat gpurun_out/pk_rc.log; tail -3 gpurun_out/pk_t_all.log; tail -3 gpurun_out/pk_t_off.log
