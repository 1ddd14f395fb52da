#!/usr/bin/env python3
"""Wall time + per-kernel HIP-event time of gh_graph_solve on the bench's general graph (120 SIM3 keyframes, 12 000 landmarks,
60 000 observations) and on the self-calibration window; prints LM iterations per second."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gslam_amd import hip, posegraph  # noqa: E402
from gslam_amd.ba import default_options  # noqa: E402
from gslam_amd.pg_synth import make_landmark_graph, with_camera  # noqa: E402

ctx = hip.Context()


def run(name, start, dof, prob, huber):
    o = default_options()
    o.huber_delta, o.max_iterations = huber, 15
    posegraph.solve_graph(ctx, start, dof, prob, o)
    ts = []
    for _ in range(5):
        t = time.perf_counter()
        r = posegraph.solve_graph(ctx, start, dof, prob, o)
        ts.append(time.perf_counter() - t)
    sm = r[-2]
    dt = sorted(ts)[2]
    ctx.prof_enable(True)
    posegraph.solve_graph(ctx, start, dof, prob, o)
    pk = ctx.prof_collect()
    ctx.prof_enable(False)
    print("%s: %d iterations, %.2f ms, %.1f it/s, cost %.6e -> %.6e" % (name, sm.iterations, dt * 1e3, sm.iterations / dt, sm.initial_cost, sm.final_cost))
    for k, v in sorted(pk.items(), key=lambda kv: -kv[1]["total_ms"])[:6]:
        print("    %-18s %8.3f ms" % (k, v["total_ms"]))


truth, start, dof, prob = make_landmark_graph(n_frames=120, n_xyz=6000, n_idp=6000, kind="sim3", seed=5, noise=1e-3, pose_edges=True,
                                              obs_per_point=5, outliers=0.02)
run("general graph", start, dof, prob, 0.01)
truth, start, dof, base = make_landmark_graph(n_frames=120, n_xyz=10000, n_idp=2000, kind="se3", seed=6, noise=0.0, obs_per_point=5)
cam_true = np.array([520.0, 515.0, 318.0, 242.0, -0.28, 0.09, 1.2e-3, -8e-4, -0.01])
run("self-calibration", start, dof, with_camera(base, cam_true, cam_true * np.array([1.03, 0.97, 1.02, 0.98, 1.03, 0.97, 1, 1, 1]), 0b111111,
                                                pixel_noise=0.3, seed=7), 2.0)
