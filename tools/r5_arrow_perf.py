#!/usr/bin/env python3
"""GPU box: LM iterations per second on C4 / C5-sized trajectories with and without loop-closure points, and which solver ran."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from gslam_amd import ba, hip
from gslam_amd.ba_synth import make_graph

ctx = hip.Context(0, stream=torch.cuda.current_stream().cuda_stream)
cases = [("C4", 500, 50000, 0, None), ("C4+20", 500, 50000, 20, None)]
if "--c5" in sys.argv:
    cases += [("C5", 10000, 1000000, 0, None), ("C5+50", 10000, 1000000, 50, 5000)]
for name, nc, npts, lc, span in cases:
    g = make_graph(nc, npts, n_obs_per_point=6, seed=2, loop_closures=lc, closure_span=span)
    for solver in (["auto", "dense"] if nc <= 500 else ["auto"]):
        ctx.set_ba_solver(solver)
        iters = 30 if nc <= 500 else 8
        ba.solve(ctx, g, ba.default_options(max_iterations=2))  # warm the arenas
        ctx.prof_enable(False)
        t0 = time.perf_counter()
        p, x, s, st = ba.solve(ctx, g, ba.default_options(max_iterations=iters))
        dt = time.perf_counter() - t0
        used = ctx.last_ba_solver()
        print("%-6s solver %-5s -> %-5s (T %d, span %d): %2d iterations, cost %.9e, whole solve %.1f ms, %.1f LM it/s incl. set-up" %
              (name, solver, used[0], used[1], used[2], s.iterations, s.final_cost, dt * 1e3, s.iterations / dt))
        # resident graph: iterations only
        gr = ba.Graph(ctx, g, ba.default_options(max_iterations=iters))
        gr.solve(ba.default_options(max_iterations=iters))
        gr.update(cam_pose=g["cam_pose"], point_xyz=g["point_xyz"])
        t0 = time.perf_counter()
        s2, _ = gr.solve(ba.default_options(max_iterations=iters))
        dt = time.perf_counter() - t0
        print("         resident graph: %2d iterations in %.2f ms = %.1f LM it/s (%.3f ms each)" % (s2.iterations, dt * 1e3, s2.iterations / dt, dt * 1e3 / max(1, s2.iterations)))
        if solver == "auto" and "--prof" in sys.argv:
            ctx.prof_enable(True)
            gr.update(cam_pose=g["cam_pose"], point_xyz=g["point_xyz"])
            gr.solve(ba.default_options(max_iterations=iters))
            for k, v in sorted(ctx.prof_collect().items(), key=lambda kv: -kv[1]["total_ms"])[:16]:
                print("           %-22s launches %4d  total %8.3f ms" % (k, v["launches"], v["total_ms"]))
            ctx.prof_enable(False)
        gr.close()
ctx.set_ba_solver("auto")
