"""bench.py's final stdout line must stay small enough for the driver to parse: BENCH_r05.json came back `parsed: null` when
the line reached 28 KB (20 KB in round 4 parsed).  `compact_line` is built here from canned full records -- the round-5
line as committed (profiles/BENCH_r05_n1.json) and a synthetic worst case -- and must round-trip under 8 KB with the
contract keys, `roofline` and `cpu_baseline` intact."""
import json
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

CONTRACT = ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
            "dtype", "data", "config", "roofline", "cpu_baseline")


def _bench():
    import importlib.util
    spec = importlib.util.spec_from_file_location("bench_module", os.path.join(ROOT, "bench.py"))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    return m


def _check(bench, full):
    line = bench.compact_line(full)
    s = json.dumps(line)
    assert len(s) < bench.COMPACT_LINE_MAX == 8192, len(s)
    back = json.loads(s)
    assert back == line
    for k in CONTRACT:
        assert k in back, k
        if k not in ("roofline", "cpu_baseline", "config"):
            assert back[k] == full[k]
    assert "extra" not in back
    for k in ("bound", "achieved", "peak", "unit", "frac", "traffic"):
        assert back["roofline"][k] == full["roofline"][k]
    for k in ("value", "unit", "cores", "kind", "sample"):
        assert k in back["cpu_baseline"]
    assert back["config"]["workload"].startswith(full["config"]["workload"][:40])
    return back


def test_round5_record_compacts_under_8k():
    bench = _bench()
    full = json.load(open(os.path.join(ROOT, "profiles", "BENCH_r05_n1.json")))
    assert len(json.dumps(full)) > 20000  # the record that did not parse
    back = _check(bench, full)
    assert back["parity_in_run"] == full["parity_in_run"]
    assert back["ba_c4_lm_iters_per_s"] == full["ba_c4_lm_iters_per_s"]


def test_worst_case_record_still_fits():
    bench = _bench()
    full = json.load(open(os.path.join(ROOT, "profiles", "BENCH_r05_n1.json")))
    full["extra"]["errors"] = {"leg%d" % i: "x" * 5000 for i in range(40)}
    full["roofline"]["note"] = "n" * 50000
    full["cpu_baseline"]["sample"] = "s" * 50000
    full["config"]["parallelism"] = "p" * 20000
    _check(bench, full)
