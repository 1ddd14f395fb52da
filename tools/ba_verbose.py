import os, sys
sys.path.insert(0, '/root/repo')
from gslam_amd import ba, hip
from gslam_amd.ba_synth import make_graph
ctx = hip.Context(0)
g = make_graph(500, 50000, n_obs_per_point=6, seed=1)
ba.solve(ctx, g, ba.default_options(max_iterations=2))
for r in range(3):
    _, _, s, _ = ba.solve(ctx, g, ba.default_options(max_iterations=12, verbose=1))
    print("total_ms", s.total_ms, "solve_ms_total", s.solve_ms_total, file=sys.stderr)
