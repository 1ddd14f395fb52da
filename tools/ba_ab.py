"""A/B of the C4 bundle-adjustment solve under environment switches, on the GPU box:
    python tools/ba_ab.py GSLAM_HIP_BWD_CHAIN=1 GSLAM_HIP_BWD_CHAIN=0
Each variant is solved `reps` times, interleaved; prints the median / minimum whole-solve time and the per-kernel
table (HIP events) of one profiled solve per variant."""
import os
import sys
import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gslam_amd import ba, hip  # noqa: E402
from gslam_amd.ba_synth import make_graph  # noqa: E402

variants = [v for v in sys.argv[1:] if "=" in v] or ["X=0"]
reps = int(os.environ.get("REPS", "9"))
iters = int(os.environ.get("ITERS", "12"))
ctx = hip.Context(0)
g = make_graph(int(os.environ.get("CAMS", "500")), int(os.environ.get("POINTS", "50000")), n_obs_per_point=6, seed=1)


def setenv(v):
    for kv in v.split(","):
        k, x = kv.split("=")
        os.environ[k] = x


for v in variants:
    setenv(v)
    ba.solve(ctx, g, ba.default_options(max_iterations=2))
times = {v: [] for v in variants}
solve = {v: [] for v in variants}
final = {}
for r in range(reps):
    for v in variants:
        setenv(v)
        _, _, s, _ = ba.solve(ctx, g, ba.default_options(max_iterations=iters))
        times[v].append(s.total_ms)
        solve[v].append(s.solve_ms_total)
        final[v] = (s.iterations, s.final_cost)
for v in variants:
    t = np.array(times[v])
    it = final[v][0]
    print(f"{v}: iterations {it} final {final[v][1]!r}  total_ms median {np.median(t):.3f} min {t.min():.3f} "
          f"-> {it / np.median(t) * 1e3:.1f} it/s (best {it / t.min() * 1e3:.1f}); solve-part median {np.median(solve[v]):.3f} ms")
    setenv(v)
    ctx.prof_enable(True)
    _, _, sp, _ = ba.solve(ctx, g, ba.default_options(max_iterations=iters))
    p = ctx.prof_collect()
    ctx.prof_enable(False)
    tot = sum(x["total_ms"] for x in p.values())
    print("   kernels ms/iteration:", {k: (x["launches"], round(x["total_ms"] / sp.iterations, 4)) for k, x in
                                        sorted(p.items(), key=lambda kv: -kv[1]["total_ms"]) if x["total_ms"] > 0.01},
          "sum", round(tot / sp.iterations, 3), "launches/it", round(sum(x["launches"] for x in p.values()) / sp.iterations, 1))
