"""Latency of the per-call host entry points a SLAM thread makes once per frame / keyframe / loop closure, pageable host
memory in and out:  gh_bow_transform_host, gh_ransac_estimate(_conf), gh_triangulate, gh_align_sim3, gh_pg_solve,
gh_graph_solve.  Perf probe, not part of the product.

  python tools/host_call_probe.py                          # this build
  GSLAM_HIP_LIB=build/ab/libgslam_hip_old.so python tools/host_call_probe.py    # another build of the library, same box
  GSLAM_HIP_PG_ARENA=0 python tools/host_call_probe.py     # graph solvers with one hipMalloc per array (graph_arena.h)

Prints one JSON object (median wall time per call)."""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from gslam_amd import bow_synth, estimator, hip, posegraph  # noqa: E402
from gslam_amd.ba import default_options  # noqa: E402
from gslam_amd.bow import Vocabulary  # noqa: E402
from gslam_amd.pg_synth import make_landmark_graph, make_pose_graph  # noqa: E402


def med(fn, n=30, warm=3):
    for _ in range(warm):
        fn()
    t = []
    for _ in range(n):
        t0 = time.perf_counter()
        fn()
        t.append(time.perf_counter() - t0)
    t.sort()
    return t[len(t) // 2]


def main():
    ctx = hip.Context(0)
    rng = np.random.default_rng(5)
    out = {"lib": hip.LIB_PATH, "pg_arena": os.environ.get("GSLAM_HIP_PG_ARENA", "1")}

    voc = bow_synth.make_vocabulary(k=10, L=4, seed=1)
    v = Vocabulary(ctx, voc)
    desc = rng.integers(0, 256, (2000, 32), dtype=np.uint8)
    out["bow_transform_host_2000_us"] = round(med(lambda: v.transform_host(desc, 2)) * 1e6, 1)
    v.close()

    n = 1000
    src = rng.uniform(-1, 1, (n, 2))
    Hm = np.array([[1.0, 0.02, 0.1], [-0.03, 0.98, -0.05], [0.01, -0.02, 1.0]])
    ph = np.c_[src, np.ones(n)] @ Hm.T
    dst = ph[:, :2] / ph[:, 2:] + rng.normal(0, 1e-3, (n, 2))
    dst[::7] += rng.uniform(-0.3, 0.3, (len(dst[::7]), 2))
    out["ransac_homography_1000_us"] = round(med(lambda: estimator.estimate(ctx, estimator.HOMOGRAPHY, src, dst, 5e-3)) * 1e6, 1)
    out["ransac_homography_conf099_1000_us"] = round(
        med(lambda: estimator.estimate_conf(ctx, estimator.HOMOGRAPHY, src, dst, 5e-3, 0.99)) * 1e6, 1)
    out["ransac_fundamental_1000_us"] = round(med(lambda: estimator.estimate(ctx, estimator.FUNDAMENTAL, src, dst, 5e-3)) * 1e6, 1)

    T = np.array([0, 0, 0, 1, 0.3, 0.0, 0.02])
    X = rng.uniform(-1, 1, (n, 3)) + np.array([0, 0, 4.0])
    d1 = X / np.linalg.norm(X, axis=1, keepdims=True)
    Xc = X + T[4:]
    d2 = Xc / np.linalg.norm(Xc, axis=1, keepdims=True)
    out["triangulate_1000_us"] = round(med(lambda: estimator.triangulate(ctx, T, d1, d2)) * 1e6, 1)

    a = rng.uniform(-2, 2, (n, 3))
    b = 1.3 * a[:, [1, 2, 0]] + np.array([0.5, -0.2, 1.0]) + rng.normal(0, 1e-3, (n, 3))
    out["align_sim3_1000_us"] = round(med(lambda: posegraph.align_sim3(ctx, a, b)) * 1e6, 1)

    def pg(nf, loops, iters):
        truth, start, dof, prob = make_pose_graph(nf, loops, kind="sim3", seed=3, noise=0.01, perturb=0.05, scale_drift=0.1)
        o = default_options()
        o.max_iterations = iters
        its = [0]

        def run():
            S, sm, st = posegraph.solve(ctx, start, dof, prob, o)
            its[0] = sm.iterations
        t = med(run, n=10, warm=2)
        return {"ms_per_solve": round(t * 1e3, 3), "iterations": its[0], "iters_per_s": round(its[0] / t, 1)}

    out["pose_graph_200_dense"] = pg(200, 30, 30)
    out["pose_graph_400_sparse"] = pg(400, 60, 30)
    out["pose_graph_5000_sparse"] = pg(5000, 600, 15)

    truth, start, dof, prob = make_landmark_graph(n_frames=120, n_xyz=6000, n_idp=6000, kind="sim3", seed=5, noise=1e-3,
                                                  pose_edges=True, obs_per_point=5, outliers=0.02)
    o = default_options()
    o.huber_delta = 0.01
    o.max_iterations = 15
    its = [0]

    def run_g():
        r = posegraph.solve_graph(ctx, start, dof, prob, o)
        its[0] = r[3].iterations
    t = med(run_g, n=10, warm=2)
    out["general_graph_120"] = {"ms_per_solve": round(t * 1e3, 3), "iterations": its[0], "iters_per_s": round(its[0] / t, 1)}
    print(json.dumps(out))


if __name__ == "__main__":
    main()
