"""Round-6 experiment: the quadtree mode's score plane by the barrier-free sliding-window kernel (GSLAM_HIP_ORB_PLANE_SW=1) against
today's tile kernel (plane variant).  Per-kernel HIP-event times, 1080p x 100 and VGA x 500, plus a hash of the outputs (the two
producers must agree bit for bit).  Run once per setting (the switch is read once per process): tools/r6_sw_exp.sh."""
import hashlib, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from gslam_amd import hip
from gslam_amd.orb import OrbExtractor, synth_frames

ctx = hip.Context(0)
for (w, h, k, nfr) in ((1920, 1080, 2000, 100), (640, 480, 1000, 500)):
    fr = synth_frames(ctx, nfr, w, h, base_seed=0x5EED0000)
    ex = OrbExtractor(ctx, w, h, max_batch=nfr, n_features=k)
    ex.set_distribution(1)
    o = ex.alloc_outputs(nfr)
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
    hsh = hashlib.sha256(b"".join(bytes(x.cpu().numpy().tobytes()) for x in o)).hexdigest()[:16]
    print("%dx%d K=%d %d frames, PLANE_SW=%s: %.3f ms per call, outputs sha256 %s" % (w, h, k, nfr, os.environ.get("GSLAM_HIP_ORB_PLANE_SW", "0"), dt * 1e3, hsh))
    for name, e in sorted(prof.items(), key=lambda kv: -kv[1]["total_ms"]):
        print("    %-22s launches %3d  per call %9.3f ms" % (name, e["launches"] // 3, e["total_ms"] / 3))
    ex.close()
