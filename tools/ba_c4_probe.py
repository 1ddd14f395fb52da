import sys, time
import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gslam_amd import hip, ba
from gslam_amd.ba_synth import make_graph
ctx = hip.Context(0)
g = make_graph(500, 50000, n_obs_per_point=6, seed=1)
ba.solve(ctx, g, ba.default_options(max_iterations=2))
for it in (1, 12):
    t = time.perf_counter()
    r = ba.solve(ctx, g, ba.default_options(max_iterations=it, verbose=1 if it == 1 else 0))
    print(it, "iters", r[2].total_ms, "ms total;", (time.perf_counter() - t) * 1e3, "ms wall; solve", r[2].solve_ms_total)
