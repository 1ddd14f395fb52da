"""What would a two-phase threshold be worth?  orb_fast_cells with min_th = 7 (the contract) against min_th = 20 (only the
strong pass: an upper bound of the saving), and the share of cells that hold no strong corner (they would need phase B)."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gslam_amd import hip
from gslam_amd.orb import OrbExtractor, synth_frames
torch.cuda.set_device(0)
ctx = hip.Context(0, stream=torch.cuda.current_stream().cuda_stream)
F, W, H, K = 400, 1920, 1080, 2000
frames = synth_frames(ctx, F, W, H, base_seed=0xC2000000)
for min_th in (7, 20):
    ex = OrbExtractor(ctx, W, H, max_batch=F, n_features=K, min_th=min_th)
    out = ex.extract(frames)
    torch.cuda.synchronize()
    ctx.prof_enable(True)
    for _ in range(3):
        ex.extract(frames, out=out) if "out" in ex.extract.__code__.co_varnames else ex.extract(frames)
    p = ctx.prof_collect()
    ctx.prof_enable(False)
    print("min_th", min_th, {k: round(v["total_ms"] / v["launches"], 4) for k, v in p.items() if k.startswith("orb_")},
          "kpts", int(out[2].sum()))
    ex.close()

# census on the bench texture: how many cells hold candidates but no corner above the initial threshold?
ex = OrbExtractor(ctx, W, H, max_batch=8, n_features=K)
ex.debug_counters(enable=True)
ex.extract(frames[:8])
torch.cuda.synchronize()
c = ex.debug_counters(enable=False)
print("bench texture, 8 frames:", {k: c[k] for k in ("cells", "weak_cells", "strong_silenced", "dense_cells")},
      "weak share %.3f" % (c["weak_cells"] / max(1, c["cells"])))
ex.close()
