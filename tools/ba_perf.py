#!/usr/bin/env python3
"""BA timing breakdown on the GPU box: fixed (setup) vs per-iteration cost at C4."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from gslam_amd import hip, ba
from gslam_amd.ba_synth import make_graph
ctx = hip.Context(0, stream=torch.cuda.current_stream().cuda_stream)
nc, npt = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (500, 50000)
g = make_graph(nc, npt, 6, seed=1)
ba.solve(ctx, g, ba.default_options(max_iterations=2))
res = {}
for it in (1, 4, 12, 24):
    best = 1e9
    for rep in range(3):
        t = time.perf_counter(); _, _, s, _ = ba.solve(ctx, g, ba.default_options(max_iterations=it)); best = min(best, (time.perf_counter() - t) * 1e3)
    res[it] = (best, s.iterations, s.total_ms, s.solve_ms_total)
    print("max_it %2d: wall %.2f ms  iterations %d  total_ms %.2f  solve_ms %.2f" % (it, best, s.iterations, s.total_ms, s.solve_ms_total))
per = (res[24][0] - res[4][0]) / (res[24][1] - res[4][1])
print("per-iteration %.3f ms, fixed %.2f ms" % (per, res[4][0] - per * res[4][1]))
