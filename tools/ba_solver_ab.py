import sys, os, time
sys.path.insert(0, os.getcwd())
from gslam_amd import hip, ba
from gslam_amd.ba_synth import make_graph
ctx = hip.Context(0)
for cams, pts in ((1000, 100000), (10000, 1000000)):
    g = make_graph(cams, pts, n_obs_per_point=6, seed=1)
    for solver in ("band", "dense"):
        ctx.set_ba_solver(solver)
        ba.solve(ctx, g, ba.default_options(max_iterations=1))
        t = time.perf_counter()
        r = ba.solve(ctx, g, ba.default_options(max_iterations=5 if cams >= 10000 else 30))
        s = r[2]
        print(cams, solver, ctx.last_ba_solver(), s.iterations, "it", round(s.total_ms, 2), "ms", round(s.iterations / s.total_ms * 1e3, 2), "it/s cost", s.final_cost, flush=True)
