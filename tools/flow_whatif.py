"""Timing experiments on the dataflow factorisation inside the C4 bundle-adjustment solve: GSLAM_HIP_FLOW_WHATIF is a
bit mask of phases the kernel SKIPS (results are garbage; only the launch duration is looked at).  Needs a library built
with -DGH_FLOW_WHATIF (make lib WHATIF=1)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gslam_amd import ba, hip  # noqa: E402
from gslam_amd.ba_synth import make_graph  # noqa: E402

ctx = hip.Context(0)
g = make_graph(500, 50000, n_obs_per_point=6, seed=1)
masks = [int(x, 0) for x in sys.argv[1:]] or [0]
for m in masks:
    os.environ["GSLAM_HIP_FLOW_WHATIF"] = str(m)
    try:
        ba.solve(ctx, g, ba.default_options(max_iterations=3))
        ctx.prof_enable(True)
        ba.solve(ctx, g, ba.default_options(max_iterations=6))
        p = ctx.prof_collect()
        ctx.prof_enable(False)
        k = p["ba_potrf_flow"]
        print(f"whatif {m:#06x}: {k['total_ms'] / k['launches']:.4f} ms per launch ({k['launches']} launches)", flush=True)
    except Exception as e:  # garbage factors may make the solve give up
        ctx.prof_enable(False)
        print(f"whatif {m:#06x}: {e}", flush=True)
