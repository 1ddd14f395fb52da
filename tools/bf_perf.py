#!/usr/bin/env python3
"""Quick BF matcher throughput check on the GPU box (C2 shape: 1000 frames x 2000 desc, consecutive pairs)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from gslam_amd import hip
from gslam_amd.matcher import BFMatcher

F, cap = 1000, 2000
ctx = hip.Context(0, stream=torch.cuda.current_stream().cuda_stream)
print(ctx.device_info())
m = BFMatcher(ctx)
print("valu probe: %.3f Tpairs/s-equivalent (16 VALU ops each)" % (m.valu_probe() / 1e12))
desc = torch.randint(0, 256, (F, cap, 32), dtype=torch.uint8, device="cuda")
counts = torch.full((F,), cap, dtype=torch.int32, device="cuda")
pq = torch.arange(0, F - 1, dtype=torch.int32, device="cuda")
pt = pq + 1
out = m.match_pairs(desc, counts, pq, pt)
torch.cuda.synchronize()
for rep in range(3):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    m.match_pairs(desc, counts, pq, pt, out=out)
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1)
    pairs = (F - 1) * cap * cap
    print("consecutive pairs: %.3f ms  %.3f Tpairs/s" % (ms, pairs / ms / 1e9))
