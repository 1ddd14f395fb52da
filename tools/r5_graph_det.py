#!/usr/bin/env python3
"""GPU box: gh_graph_solve on the bench graph (120 keyframes / 12 000 landmarks / 60 000 observations), reproducible mode vs atomics."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from gslam_amd import hip, posegraph
from gslam_amd.ba import default_options
from gslam_amd.pg_synth import make_landmark_graph
ctx = hip.Context(0, stream=torch.cuda.current_stream().cuda_stream)
truth, start, dof, prob = make_landmark_graph(n_frames=120, n_xyz=6000, n_idp=6000, kind="sim3", seed=5, noise=1e-3, pose_edges=True, obs_per_point=5, outliers=0.02)
for det in (1, 0, 1, 0):
    o = default_options(); o.huber_delta = 0.01; o.max_iterations = 15; o.deterministic = det
    posegraph.solve_graph(ctx, start, dof, prob, o)
    ts = []
    for _ in range(5):
        t = time.perf_counter(); r = posegraph.solve_graph(ctx, start, dof, prob, o); ts.append(time.perf_counter() - t)
    sm = r[3]
    print("deterministic=%d: %d iterations, final cost %.15e, median %.2f ms = %.1f LM it/s" % (det, sm.iterations, sm.final_cost, np.median(ts) * 1e3, sm.iterations / np.median(ts)))
    if det:
        ctx.prof_enable(True); posegraph.solve_graph(ctx, start, dof, prob, o)
        for k, v in sorted(ctx.prof_collect().items(), key=lambda kv: -kv[1]["total_ms"])[:10]:
            print("     %-18s launches %3d total %.3f ms" % (k, v["launches"], v["total_ms"]))
        ctx.prof_enable(False)
