# round 5: the whole GPU suite + the default bench line
mkdir -p gpurun_out
timeout 1500 python -m pytest tests -m gpu -q --tb=short > gpurun_out/r5_full_t.log 2>&1; echo "suite rc=$?" > gpurun_out/r5_full_rc.log
timeout 1200 python bench.py > gpurun_out/BENCH_r05_n1.json 2> gpurun_out/BENCH_r05_n1.err; echo "bench rc=$?" >> gpurun_out/r5_full_rc.log
cat gpurun_out/r5_full_rc.log; tail -8 gpurun_out/r5_full_t.log
python - <<'PY'
import json
for line in open("gpurun_out/BENCH_r05_n1.json"):
    if line.startswith("{"):
        j = json.loads(line)
        e = j["extra"]
        print("value", j["value"], "ms/step", j["ms_per_step"], "frac", j["roofline"]["frac"], "attainable", j["roofline"].get("attainable"))
        print("kernels", e["kernels"]); print("slam", e.get("orb_slam_mode"))
        print("pipeline", e["roofline_pipeline"])
        print("ba", {k: e["ba"].get(k) for k in ("iters_per_s", "resolve_iters_per_s")}, "lc", {k: (e["ba"].get("loop_closure") or {}).get(k) for k in ("iters_per_s", "resolve_iters_per_s", "border_cams")})
        c5 = e.get("ba_c5") or {}
        print("c5", c5.get("iters_per_s"), (c5.get("band_solver") or {}).get("iters_per_s"), "lc", {k: (c5.get("loop_closure") or {}).get(k) for k in ("iters_per_s", "border_cams")}, ((c5.get("loop_closure") or {}).get("to_convergence") or {}).get("iters_per_s"))
        print("graph", {k: (v.get("iters_per_s"), v.get("atomics_mode", {}).get("iters_per_s")) for k, v in (e.get("graph_solvers") or {}).items()})
        print("errors", e.get("errors"))
PY
