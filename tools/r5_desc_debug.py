#!/usr/bin/env python3
"""GPU box: descriptors of orb_describe with the blur on MFMA vs on the VALU (GSLAM_HIP_ORB_DESC_MFMA), same frame."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from gslam_amd import hip
from gslam_amd.orb import OrbExtractor, synth_frames, kps_to_numpy

res = {}
for m in ("0", "1"):
    os.environ["GSLAM_HIP_ORB_DESC_MFMA"] = m
    ctx = hip.Context(0, stream=torch.cuda.current_stream().cuda_stream)
    ex = OrbExtractor(ctx, 640, 480, max_batch=1, n_features=1000)
    fr = synth_frames(ctx, 1, 640, 480, base_seed=0x5EED0000)
    kps, desc, counts = ex.extract(fr)
    torch.cuda.synchronize()
    res[m] = (kps_to_numpy(kps)[0].copy(), desc.cpu().numpy()[0].copy(), int(counts[0]))
    ex.close(); ctx.close()
k0, d0, n0 = res["0"]; k1, d1, n1 = res["1"]
print("counts", n0, n1, "kps equal", k0.tobytes() == k1.tobytes())
eq = (d0[:n0] == d1[:n0]).all(axis=1)
print("descriptors equal: %d of %d" % (eq.sum(), n0))
bits = np.unpackbits(d0[:n0] ^ d1[:n0], axis=1).sum(axis=1)
print("differing bits per keypoint: mean %.1f min %d max %d" % (bits.mean(), bits.min(), bits.max()))
x = (k0["x"][:n0] / np.array([1.2 ** o for o in k0["octave"][:n0]])).round().astype(int)
for a in range(4):
    sel = ((x - 16) & 3) == a
    print("  (x-16)&3 == %d: %d keypoints, %d equal, mean differing bits %.1f" % (a, sel.sum(), eq[sel].sum(), bits[sel].mean() if sel.any() else 0))
# which of the 256 tests differ most often
tb = np.unpackbits(d0[:n0] ^ d1[:n0], axis=1, bitorder="little").mean(axis=0)
print("per-test mismatch rate: min %.3f max %.3f; tests that never differ: %d" % (tb.min(), tb.max(), (tb == 0).sum()))
