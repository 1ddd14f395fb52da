"""Per-kernel HIP-event times of the ORB-SLAM compatible extraction mode (gh_orb_plan_set_distribution(1) +
gh_orb_plan_set_steering(1)) beside the default mode, same frames.  Output: profiles/orb_slam_mode_r04.txt"""
import sys, os, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from gslam_amd import hip
from gslam_amd.orb import OrbExtractor, synth_frames

ctx = hip.Context(0)
lines = []
for (w, h, k, nfr) in ((640, 480, 1000, 500), (1920, 1080, 2000, 100)):
    fr = synth_frames(ctx, nfr, w, h, base_seed=0x5EED0000)
    for mode in (0, 1):
        ex = OrbExtractor(ctx, w, h, max_batch=nfr, n_features=k)
        if mode:
            ex.set_distribution(1)
            ex.set_steering(1)
        o = ex.alloc_outputs(nfr)
        ex.extract(fr, o)
        torch.cuda.synchronize()
        t = time.perf_counter()
        for _ in range(3):
            ex.extract(fr, o)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t) / 3
        ctx.prof_enable(True)
        ex.extract(fr, o)
        torch.cuda.synchronize()
        prof = ctx.prof_collect()
        ctx.prof_enable(False)
        kp = int(o[2].sum().item())
        lines.append("%dx%d K=%d %d frames, mode %s: %.3f ms per call = %.1f us/frame, %.2f Mkeypoints/s (%d keypoints/frame), plan %.0f MB"
                     % (w, h, k, nfr, "quadtree+steer" if mode else "default", dt * 1e3, dt / nfr * 1e6, kp / dt / 1e6, kp // nfr,
                        ex.device_bytes() / 1e6))
        for name, e in sorted(prof.items(), key=lambda kv: -kv[1]["total_ms"]):
            lines.append("    %-22s launches %3d  total %9.3f ms" % (name, e["launches"], e["total_ms"]))
        ex.close()
print("\n".join(lines))
os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
open(os.path.join(ROOT, "gpurun_out", "orb_slam_mode_r04.txt"), "w").write("\n".join(lines) + "\n")
