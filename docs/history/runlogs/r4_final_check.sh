# round-4 closing GPU check: the whole GPU suite, the default bench line, and the matcher-overlap A/B (short)
mkdir -p gpurun_out
timeout 900 python -m pytest tests -m gpu -q --tb=short > gpurun_out/final_t.log 2>&1; echo "suite rc=$?" > gpurun_out/final_rc.log
timeout 900 python bench.py > gpurun_out/BENCH_r04_n1.json 2> gpurun_out/BENCH_r04_n1.err; echo "bench rc=$?" >> gpurun_out/final_rc.log
timeout 300 python bench.py --steps 20 --warmup 3 --match-overlap --no-cpu-baseline --no-ba --no-bow --no-c3 --no-c5 --no-host-fed --no-all-pairs-full --no-range > gpurun_out/BENCH_r04_overlap.json 2>/dev/null; echo "overlap rc=$?" >> gpurun_out/final_rc.log
cat gpurun_out/final_rc.log; tail -5 gpurun_out/final_t.log
python - <<'PY'
import json
for f in ("gpurun_out/BENCH_r04_n1.json", "gpurun_out/BENCH_r04_overlap.json"):
    for line in open(f):
        if line.startswith("{"):
            j = json.loads(line)
            print(f, "value", j["value"], "ms/step", j["ms_per_step"], "frac", j["roofline"]["frac"], "host_fed", (j["extra"].get("host_fed") or {}).get("Mkeypoints_per_s"),
                  "overlap", j["extra"]["bf_match"].get("overlap"), "errors", j["extra"].get("errors"))
PY
