"""Round-6 sliding-window experiment: which ingredient of version 2 races?  One process, outputs of every variant compared with the
tile kernel's (GSLAM_HIP_ORB_PLANE_SW is read per call).  bits: 1 on, 2 conditional stores, 4 flush at the end of the group, 8 C loads."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
from gslam_amd import hip
from gslam_amd.orb import OrbExtractor, synth_frames, kps_to_numpy

ctx = hip.Context(0)
for (w, h, k, nfr) in ((640, 480, 1000, 500), (1920, 1080, 2000, 600)):
    fr = synth_frames(ctx, nfr, w, h, base_seed=0x5EED0000)
    ex = OrbExtractor(ctx, w, h, max_batch=nfr, n_features=k)
    ex.set_distribution(1)
    def run(reps=1):
        o = ex.alloc_outputs(nfr)
        for _ in range(reps):
            ex.extract(fr, o)
        torch.cuda.synchronize()
        return kps_to_numpy(o[0]).copy(), o[1].cpu().numpy().copy(), o[2].cpu().numpy().copy()
    os.environ["GSLAM_HIP_ORB_PLANE_SW"] = "0"
    ref = run()
    for var in (1, 15, 7, 9, 3, 5, 1):
        os.environ["GSLAM_HIP_ORB_PLANE_SW"] = str(var)
        bad = []
        for rep in range(4):
            out = run(6)
            badf = [f for f in range(nfr) if out[2][f] != ref[2][f] or out[0][f].tobytes() != ref[0][f].tobytes() or not np.array_equal(out[1][f], ref[1][f])]
            bad.append(len(badf))
        print("%dx%d x%d variant %2d: frames that differ from the tile kernel's output in 4 runs: %s" % (w, h, nfr, var, bad), flush=True)
    ex.close()
