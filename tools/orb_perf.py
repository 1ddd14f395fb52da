#!/usr/bin/env python3
"""Quick ORB extraction throughput + per-kernel timing on the GPU box."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from gslam_amd import hip
from gslam_amd.orb import OrbExtractor, synth_frames

W, H, K = 1920, 1080, 2000
B = int(sys.argv[1]) if len(sys.argv) > 1 else 200
ctx = hip.Context(0, stream=torch.cuda.current_stream().cuda_stream)
ex = OrbExtractor(ctx, W, H, max_batch=B, n_features=K)
print("plan bytes: %.1f MB" % (ex.device_bytes() / 1e6))
fr = synth_frames(ctx, B, W, H)
out = ex.alloc_outputs(B)
ex.extract(fr, out)
torch.cuda.synchronize()
print("counts", out[2][:4].tolist())
for rep in range(3):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); ex.extract(fr, out); e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1)
    print("extract %d frames: %.3f ms  %.1f us/frame  %.1f Mkpts/s" % (B, ms, ms * 1e3 / B, int(out[2].sum()) / ms / 1e3))
ctx.prof_enable(True)
ex.extract(fr, out)
for k, v in ctx.prof_collect().items():
    print("  %-18s launches %3d  total %.3f ms" % (k, v["launches"], v["total_ms"]))
