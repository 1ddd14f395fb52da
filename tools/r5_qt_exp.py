"""GPU box: wall time of the quadtree (ORB-SLAM) extraction mode at 1080p, K = 2000, 100 frames, steering on / off, under
   whatever GSLAM_HIP_* switches the caller sets.  Best of three groups of 5 back-to-back calls, no per-kernel events."""
import sys, os, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from gslam_amd import hip
from gslam_amd.orb import OrbExtractor, synth_frames

ctx = hip.Context(0)
w, h, k, nfr = 1920, 1080, 2000, 100
fr = synth_frames(ctx, nfr, w, h, base_seed=0x5EED0000)
for steer in (1, 0):
    ex = OrbExtractor(ctx, w, h, max_batch=nfr, n_features=k)
    ex.set_distribution(1)
    ex.set_steering(steer)
    o = ex.alloc_outputs(nfr)
    for _ in range(2):
        ex.extract(fr, o)
    torch.cuda.synchronize()
    best = 1e9
    for rep in range(3):
        t = time.perf_counter()
        for _ in range(5):
            ex.extract(fr, o)
        torch.cuda.synchronize()
        best = min(best, (time.perf_counter() - t) / 5)
    kp = int(o[2].sum().item())
    print("steer=%d: %.3f ms per call, %.2f Mkeypoints/s" % (steer, best * 1e3, kp / best / 1e6))
    ex.close()
