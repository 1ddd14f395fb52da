"""Per-kernel times of the C5 graph (10 k cams / 1 M points / 6 M obs) through the band solver."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gslam_amd import hip, ba
from gslam_amd.ba_synth import make_graph
ctx = hip.Context(0)
g = make_graph(10000, 1000000, n_obs_per_point=6, seed=1)
ctx.set_ba_solver("band")
ba.solve(ctx, g, ba.default_options(max_iterations=1))
r = ba.solve(ctx, g, ba.default_options(max_iterations=5))
print("C5 band:", r[2].iterations, "iterations", round(r[2].total_ms, 2), "ms total (incl. set-up)", ctx.last_ba_solver())
ctx.prof_enable(True)
r = ba.solve(ctx, g, ba.default_options(max_iterations=5))
prof = ctx.prof_collect()
ctx.prof_enable(False)
its = r[2].iterations
tot = 0
for k, v in sorted(prof.items(), key=lambda kv: -kv[1]["total_ms"]):
    print(f"   {k:18s} {v['launches'] / its:6.1f} launches / it {v['total_ms'] / its * 1e3:9.1f} us / it")
    tot += v["total_ms"]
print("   kernels total", round(tot / its * 1e3, 1), "us / it; solve total_ms", round(r[2].total_ms, 2))
