"""Large pose graphs through gh_pg_solve: block-sparse (bsparse.hip) against the dense keyframe system, same box.
usage: python tools/pg_large_probe.py [n_frames ...]      -> one JSON line per size (profiles/pg_block_sparse_r03.txt)"""
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gslam_amd import ba, hip, posegraph  # noqa: E402
from gslam_amd.pg_synth import make_pose_graph  # noqa: E402

ctx = hip.Context()
sizes = [int(v) for v in sys.argv[1:]] or [700, 1500, 5000, 20000]
for nf in sizes:
    truth, start, dof, prob = make_pose_graph(nf, nf // 8, kind="sim3", seed=4, noise=0.01, perturb=0.03, scale_drift=0.1)
    rec = {"keyframes": nf, "edges": int(len(prob["sim3"][0])), "unknowns": 7 * nf}
    for mode, env, iters in (("block_sparse", "0", 10), ("dense", "100000000", 10 if nf <= 2000 else 3)):
        if mode == "dense" and nf > 8000:
            continue
        os.environ["GSLAM_HIP_PG_SPARSE_MIN"] = env
        o = ba.default_options(max_iterations=iters)
        posegraph.solve(ctx, start, dof, prob, o)
        ctx.prof_enable(True)
        t0 = time.perf_counter()
        S, sm, st = posegraph.solve(ctx, start, dof, prob, o)
        dt = time.perf_counter() - t0
        pk = ctx.prof_collect()
        ctx.prof_enable(False)
        rec[mode] = {"iterations": sm.iterations, "ms_per_iteration": round(dt * 1e3 / max(sm.iterations, 1), 3),
                     "linear_solve_ms_per_iteration": round(sm.solve_ms_total / max(sm.iterations, 1), 3),
                     "final_cost": sm.final_cost, "status": int(st),
                     "kernels_ms": {k: round(v["total_ms"], 3) for k, v in sorted(pk.items(), key=lambda kv: -kv[1]["total_ms"])[:6]}}
    pr, pc = np.maximum(prob["sim3"][0], prob["sim3"][1]), np.minimum(prob["sim3"][0], prob["sim3"][1])
    t0 = time.perf_counter()
    sym = posegraph.bs_symbolic(nf, pr, pc)
    rec["elimination"] = {"host_ms": round((time.perf_counter() - t0) * 1e3, 2), "sparse_columns": sym["ns"], "root_keyframes": sym["nr"],
                          "rounds": len(sym["round_ptr"]) - 1, "blocks": len(sym["rows"]), "block_products": sym["pair_products"]}
    if "dense" in rec:
        a, b = rec["dense"], rec["block_sparse"]
        rec["same_cost"] = abs(a["final_cost"] - b["final_cost"]) <= 1e-6 * abs(a["final_cost"]) if a["iterations"] == b["iterations"] else None
    print(json.dumps(rec), flush=True)
