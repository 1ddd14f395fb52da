"""Round-6 experiment: the quadtree mode's score plane by the barrier-free sliding-window kernel (GSLAM_HIP_ORB_PLANE_SW, read per
call) against today's tile kernel (plane variant).  Per-kernel HIP-event times and a direct comparison of the outputs (the two
producers must agree bit for bit), one process.  Variant bits: 1 on, 2 conditional stores, 4 flush at the end of the group, 8 C loads."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
from gslam_amd import hip
from gslam_amd.orb import OrbExtractor, synth_frames, kps_to_numpy

ctx = hip.Context(0)
variants = [int(v) for v in (sys.argv[1] if len(sys.argv) > 1 else "0,1,15,0,1").split(",")]
for (w, h, k, nfr) in ((1920, 1080, 2000, 100), (640, 480, 1000, 500), (1920, 1080, 2000, 600)):
    fr = synth_frames(ctx, nfr, w, h, base_seed=0x5EED0000)
    ex = OrbExtractor(ctx, w, h, max_batch=nfr, n_features=k)
    ex.set_distribution(1)
    o = ex.alloc_outputs(nfr)
    ref = None
    for var in variants:
        os.environ["GSLAM_HIP_ORB_PLANE_SW"] = str(var)
        ex.extract(fr, o)
        torch.cuda.synchronize()
        t = time.perf_counter()
        for _ in range(5):
            ex.extract(fr, o)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t) / 5
        ctx.prof_enable(True)
        for _ in range(3):
            ex.extract(fr, o)
        torch.cuda.synchronize()
        prof = ctx.prof_collect()
        ctx.prof_enable(False)
        out = (kps_to_numpy(o[0]).copy(), o[1].cpu().numpy().copy(), o[2].cpu().numpy().copy())
        if var == 0 and ref is None:
            ref = out
        same = all(np.array_equal(a, b) if a.dtype != ref[0].dtype else a.tobytes() == b.tobytes() for a, b in zip(out, ref))
        print("%dx%d K=%d %d frames, PLANE_SW=%d: %.3f ms per call, outputs %s the tile kernel's" % (w, h, k, nfr, var, dt * 1e3, "EQUAL" if same else "DIFFER FROM"), flush=True)
        for name, e in sorted(prof.items(), key=lambda kv: -kv[1]["total_ms"]):
            if name.startswith("orb_fast_plane") or name == "orb_resize":
                print("    %-22s launches %3d  per call %9.3f ms" % (name, e["launches"] // 3, e["total_ms"] / 3), flush=True)
    ex.close()
