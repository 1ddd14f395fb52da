"""GPU box experiment: does a SECOND extraction pipeline on its own stream fill the stalls of the first?  Two contexts (one HIP
stream each), each with its own plan and its own 1000 x 1080p frames; steps alternate between them.  Against the same number of
steps on one context.  (orb_describe is latency bound, orb_fast_cells issue bound: DESIGN.md 6a / 6c.)"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from gslam_amd import hip
from gslam_amd.orb import OrbExtractor, synth_frames

F, W, H, K, STEPS = int(sys.argv[1]) if len(sys.argv) > 1 else 1000, 1920, 1080, 2000, 20
streams = [torch.cuda.Stream(), torch.cuda.Stream()]
ctxs = [hip.Context(0, stream=s.cuda_stream) for s in streams]
exs, frs, outs = [], [], []
for c, s in zip(ctxs, streams):
    with torch.cuda.stream(s):
        exs.append(OrbExtractor(c, W, H, max_batch=F, n_features=K))
        frs.append(synth_frames(c, F, W, H, base_seed=0x5EED0000))
        outs.append(exs[-1].alloc_outputs(F))
torch.cuda.synchronize()


def run(order):
    for i in order:  # warm-up
        exs[i].extract(frs[i], outs[i])
    torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(STEPS // len(order)):
        for i in order:
            exs[i].extract(frs[i], outs[i])
    torch.cuda.synchronize()
    return (time.perf_counter() - t) / STEPS


for rep in range(2):
    one = run([0])
    two = run([0, 1])
    print("%d frames per step: one stream %.3f ms per step, two alternating streams %.3f ms per step (%.1f %%)" %
          (F, one * 1e3, two * 1e3, 100 * (two / one - 1)))
