# round-4 second GPU check: SWAR pass 1 (perm split, lists in the tile, 8 waves / SIMD) at 8 / 7 / 6 workgroups per CU, matcher
# dispatch, BA parity on the co-visibility graphs with the 1e-8 state bar, bench with the overlapped matcher + in-run parity
mkdir -p gpurun_out
T="tests/test_orb_gpu.py tests/test_orb_adversarial_gpu.py tests/test_orb_stream_gpu.py tests/test_bf_gpu.py"
timeout 300 python -m pytest $T -m gpu -q --tb=short -x > gpurun_out/c2_t_orb.log 2>&1; echo "orb+bf rc=$?" > gpurun_out/c2_rc.log
for cfg in "0 0" "1 0" "1 2900" "1 6800" "0 0" "1 0"; do set -- $cfg
  GSLAM_HIP_ORB_PASS1=$1 GSLAM_HIP_ORB_LDSPAD=$2 timeout 60 python tools/orb_perf.py 400 > gpurun_out/c2_perf.txt 2>&1
  echo "pass1=$1 pad=$2: $(grep -h 'orb_fast_cells' gpurun_out/c2_perf.txt | tail -1) | $(grep -h 'extract' gpurun_out/c2_perf.txt | tail -1)" >> gpurun_out/c2_rc.log
done
timeout 600 python -m pytest tests/test_full_configs_gpu.py tests/test_ba_gpu.py -m gpu -q --tb=short > gpurun_out/c2_t_ba.log 2>&1; echo "ba rc=$?" >> gpurun_out/c2_rc.log
timeout 600 python bench.py --steps 20 --warmup 3 --no-c5 --no-host-fed --no-range --no-bow > gpurun_out/c2_bench.json 2> gpurun_out/c2_bench.err; echo "bench rc=$?" >> gpurun_out/c2_rc.log
cat gpurun_out/c2_rc.log; tail -5 gpurun_out/c2_t_orb.log; tail -15 gpurun_out/c2_t_ba.log; tail -5 gpurun_out/c2_bench.err
python tools/show_bench.py gpurun_out/c2_bench.json 2>/dev/null | head -60
