"""Cost of the solver's own camera ordering on shuffled C4 / C5 (GSLAM_HIP_BA_TIMING=1 prints the phases)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
from gslam_amd import hip, ba
from gslam_amd.ba_synth import make_graph
ctx = hip.Context(0)
def shuffle_cameras(g, seed):
    rng = np.random.default_rng(seed)
    nc = len(g["cam_dof"])
    new_of_old = rng.permutation(nc).astype(np.int32)
    old_of_new = np.argsort(new_of_old)
    h = dict(g)
    h["cam_pose"] = np.ascontiguousarray(g["cam_pose"][old_of_new]); h["cam_dof"] = np.ascontiguousarray(g["cam_dof"][old_of_new])
    h["obs_cam"] = new_of_old[g["obs_cam"]].astype(np.int32)
    return h
variants = [("", {})]
if len(sys.argv) > 1:
    variants = [("", {}), (" EARLY_UPLOAD=0", {"GSLAM_HIP_BA_EARLY_UPLOAD": "0"}), (" ORDER_FLIP=1", {"GSLAM_HIP_BA_ORDER_FLIP": "1"})]
for cams, pts, iters in ((500, 50000, 12), (10000, 1000000, 40)):
    g = make_graph(cams, pts, n_obs_per_point=6, seed=1)
    h = shuffle_cameras(g, 1)
    for name, gr, env in [("in order", g, {})] + [("shuffled" + vn, h, ve) for vn, ve in variants]:
        for k in ("GSLAM_HIP_BA_EARLY_UPLOAD", "GSLAM_HIP_BA_ORDER_FLIP"):
            os.environ.pop(k, None)
        os.environ.update(env)
        ba.solve(ctx, gr, ba.default_options(max_iterations=2))
        os.environ["GSLAM_HIP_BA_TIMING"] = "1"
        sys.stderr.write("== %d cams %s\n" % (cams, name)); sys.stderr.flush()
        _, _, s, _ = ba.solve(ctx, gr, ba.default_options(max_iterations=iters))
        del os.environ["GSLAM_HIP_BA_TIMING"]
        for rep in range(2):
            ctx.prof_enable(True)
            t0 = time.perf_counter()
            ba.solve(ctx, gr, ba.default_options(max_iterations=2))
            dt = time.perf_counter() - t0
            prof = ctx.prof_collect()
            ctx.prof_enable(False)
            print("   profiled solve %d: %.1f ms wall, kernels sum %.1f ms" % (rep, dt * 1e3, sum(v["total_ms"] for v in prof.values())), flush=True)
        print("   set-up kernels (2 iterations): " + ", ".join("%s %.2f" % (k, v["total_ms"]) for k, v in [kv for kv in sorted(prof.items(), key=lambda kv: -kv[1]["total_ms"]) if kv[1]["total_ms"] > 0.25]), flush=True)
        best = min(ba.solve(ctx, gr, ba.default_options(max_iterations=iters))[2].total_ms for _ in range(3))
        print("%d cams %s: %d iterations, %.2f ms (best of 3), %.1f it/s, camera order %s" % (cams, name, s.iterations, best, s.iterations / best * 1e3, ctx.last_ba_order()), flush=True)
