"""Band solver (chol_cr.hip) against the dense single-launch factorisation on the C4-shaped system (n = 3000,
half-bandwidth 149) and inside the LM loop of the C4 graph; per-kernel times from the context's event profiler.
   python tools/cr_probe.py [out.txt]"""
import os
import sys
import time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import torch
from gslam_amd import hip, ba
from gslam_amd.ba_synth import make_graph
from test_cr_solver import make_band

out = open(sys.argv[1], "w") if len(sys.argv) > 1 else sys.stdout
def say(*a):
    print(*a, file=out, flush=True)
    if out is not sys.stdout:
        print(*a, flush=True)

ctx = hip.Context(0)
for n, hb in ((3000, 149), (6000, 149), (3000, 60)):
    S = make_band(n, hb, seed=3)
    b = np.random.default_rng(0).standard_normal(n)
    xr = np.linalg.solve(S, b)
    x, info = ba.band_solve(ctx, S, b, hb)
    say(f"n={n} hb={hb}: band info {info} err {np.abs(x - xr).max() / np.abs(xr).max():.2e}")
    _, xd, infod = ba.potrf_solve(ctx, S, b)
    say(f"   dense info {infod} err {np.abs(xd - xr).max() / np.abs(xr).max():.2e}")
    for name, fn in (("band", lambda: ba.band_solve(ctx, S, b, hb)), ("dense", lambda: ba.potrf_solve(ctx, S, b))):
        fn()
        ctx.prof_enable(True)
        for _ in range(5):
            fn()
        prof = ctx.prof_collect()
        ctx.prof_enable(False)
        tot = sum(v["total_ms"] for v in prof.values()) / 5
        say(f"   {name}: kernels {tot:.3f} ms per solve")
        for k, v in sorted(prof.items()):
            say(f"      {k:18s} {v['launches'] / 5:5.1f} launches  {v['total_ms'] / 5 * 1e3:8.1f} us per solve  "
                f"{v['total_ms'] / v['launches'] * 1e3:7.1f} us each")

g = make_graph(500, 50000, n_obs_per_point=6, seed=1)
for solver in ("dense", "band"):
    ctx.set_ba_solver(solver)
    ba.solve(ctx, g, ba.default_options(max_iterations=2))
    best = None
    for rep in range(3):
        r = ba.solve(ctx, g, ba.default_options(max_iterations=50))
        s = r[2]
        if best is None or s.total_ms < best.total_ms:
            best = s
    say(f"C4 {solver}: {best.iterations} iterations, {best.total_ms:.2f} ms total, {best.iterations / best.total_ms * 1e3:.1f} it/s, "
        f"final cost {best.final_cost:.12e}, accepted {best.accepted}")
    ctx.prof_enable(True)
    r = ba.solve(ctx, g, ba.default_options(max_iterations=50))
    prof = ctx.prof_collect()
    ctx.prof_enable(False)
    its = r[2].iterations
    for k, v in sorted(prof.items()):
        say(f"      {k:18s} {v['launches'] / its:5.1f} launches / it  {v['total_ms'] / its * 1e3:8.1f} us / it")
ctx.set_ba_solver("auto")
