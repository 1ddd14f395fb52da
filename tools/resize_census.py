#!/usr/bin/env python3
"""How many wave-level passes through the fused resize body does a frame cost (debug counters), against the tiles that run it."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from gslam_amd import hip
from gslam_amd.orb import OrbExtractor, synth_frames
W, H, K, B = 1920, 1080, 2000, 8
ctx = hip.Context(0, stream=torch.cuda.current_stream().cuda_stream)
ex = OrbExtractor(ctx, W, H, max_batch=B, n_features=K)
fr = synth_frames(ctx, B, W, H)
ex.debug_counters(enable=True, read=False)
ex.extract(fr)
torch.cuda.synchronize()
c = ex.debug_counters(enable=True, read=True)
print(c)
tiles = 0
for l in range(8):
    w, h, q = ex.level(l)
    nbx, nby = ((w - 38 + 31) // 32 + 1) // 2, ((h - 38 + 31) // 32 + 1) // 2
    print("level", l, w, h, "tiles", nbx * nby)
    if l < 7:
        tiles += nbx * nby
print("resize passes per tile with a next level: %.2f" % (c["resize_passes"] / B / tiles), " cells per frame", c["cells"] / B)
