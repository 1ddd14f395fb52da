"""C5 (10 000 cameras, 1 M points, 6 M observations): the solve's own timing lines (GSLAM_HIP_BA_TIMING=1)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gslam_amd import hip, ba
from gslam_amd.ba_synth import make_graph
ctx = hip.Context(0)
g = make_graph(10000, 1000000, n_obs_per_point=6, seed=1)
it = int(sys.argv[1]) if len(sys.argv) > 1 else 2
for rep in range(2):
    t = time.perf_counter()
    r = ba.solve(ctx, g, ba.default_options(max_iterations=it))
    print(it, "iters", r[2].total_ms, "ms total;", (time.perf_counter() - t) * 1e3, "ms wall; solve", r[2].solve_ms_total)
