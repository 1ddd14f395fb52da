# round-4 third GPU check: orb_describe with the blurred patch in the raw patch's LDS (8 workgroups per CU) vs padded back to 6,
# ORB parity, SQ counters of the SWAR orb_fast_cells, bench (light) with in-run parity
mkdir -p gpurun_out
T="tests/test_orb_gpu.py tests/test_orb_adversarial_gpu.py tests/test_orb_stream_gpu.py"
timeout 300 python -m pytest $T -m gpu -q --tb=short -x > gpurun_out/c3_t_orb.log 2>&1; echo "orb rc=$?" > gpurun_out/c3_rc.log
for pad in 0 3000 0 3000; do
  GSLAM_HIP_ORB_DESC_LDSPAD=$pad timeout 60 python tools/orb_perf.py 400 > gpurun_out/c3_perf.txt 2>&1
  echo "desc pad=$pad: $(grep -h 'orb_describe' gpurun_out/c3_perf.txt | tail -1) | $(grep -h 'extract' gpurun_out/c3_perf.txt | tail -1)" >> gpurun_out/c3_rc.log
done
bash tools/orb_counters.sh "1" > gpurun_out/orb_counters_v1.txt 2>&1
timeout 600 python bench.py --steps 20 --warmup 3 --no-c5 --no-host-fed --no-bow --no-ba > gpurun_out/c3_bench.json 2> gpurun_out/c3_bench.err; echo "bench rc=$?" >> gpurun_out/c3_rc.log
cat gpurun_out/c3_rc.log; tail -3 gpurun_out/c3_t_orb.log; tail -5 gpurun_out/c3_bench.err
grep -A30 "variant 1 fast_cells" gpurun_out/orb_counters_v1.txt
python - <<'PY'
import json
for line in open("gpurun_out/c3_bench.json"):
    if line.startswith("{"):
        j = json.loads(line)
        print("value", j["value"], "ms/step", j["ms_per_step"], "parity", j.get("parity_in_run"))
        print("range parity", j["extra"].get("parity_in_run_range"), "c3", (j["extra"].get("c3_stereo") or {}).get("parity_in_run"))
        print("bf", {k: v for k, v in j["extra"]["bf_match"].items() if k in ("Gpairs_per_s", "kernel", "ms_per_step", "overlap")})
        print("kernels", j["extra"]["kernels"]); print("errors", j["extra"].get("errors")); print("cpu", j["cpu_baseline"])
PY
