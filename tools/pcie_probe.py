"""PCIe-inclusive extraction rate: frames start in pinned HOST memory, results end in pinned
host memory.  Two variants: serial (copy in, extract, copy out on one stream) and double-buffered chunks on two
streams (copies of chunk i+1 overlap the kernels of chunk i).  Perf probe only; bench.py's `value` is HBM-resident."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gslam_amd import hip  # noqa: E402
from gslam_amd.orb import OrbExtractor, synth_frames  # noqa: E402


def main():
    F, W, H, K, CH = 400, 1920, 1080, 2000, 50
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(0)
    ctx = hip.Context(0, stream=torch.cuda.current_stream().cuda_stream)
    frames_dev = synth_frames(ctx, F, W, H, device=dev)
    host = torch.empty((F, H, W), dtype=torch.uint8).pin_memory()
    host.copy_(frames_dev)
    torch.cuda.synchronize()
    ex = OrbExtractor(ctx, W, H, max_batch=F, n_features=K)
    out_dev = ex.alloc_outputs(F, dev)
    out_host = [torch.empty(t.shape, dtype=t.dtype).pin_memory() for t in out_dev]

    def serial():
        frames_dev.copy_(host, non_blocking=True)
        ex.extract(frames_dev, out_dev)
        for h, d in zip(out_host, out_dev):
            h.copy_(d, non_blocking=True)
        torch.cuda.synchronize()

    for _ in range(2):
        serial()
    t = time.perf_counter()
    n = 3
    for _ in range(n):
        serial()
    ts = (time.perf_counter() - t) / n
    kp = int(out_host[2].sum())
    print(f"serial   : {ts * 1e3:.1f} ms for {F} frames -> {kp / ts / 1e6:.1f} Mkeypoints/s "
          f"({F * W * H / ts / 1e9:.1f} GB/s of frames over PCIe)")

    # double-buffered: two streams, one context + plan per stream, chunks of CH frames
    streams = [torch.cuda.Stream(), torch.cuda.Stream()]
    ctxs = [hip.Context(0, stream=s.cuda_stream) for s in streams]
    exs = [OrbExtractor(c, W, H, max_batch=CH, n_features=K) for c in ctxs]
    bufs = [torch.empty((CH, H, W), dtype=torch.uint8, device=dev) for _ in streams]
    outs = [e.alloc_outputs(CH, dev) for e in exs]

    def pipelined():
        for i, c0 in enumerate(range(0, F, CH)):
            s = i & 1
            with torch.cuda.stream(streams[s]):
                bufs[s].copy_(host[c0:c0 + CH], non_blocking=True)
                exs[s].extract(bufs[s], outs[s])
                for h, d in zip(out_host, outs[s]):
                    h[c0:c0 + CH].copy_(d, non_blocking=True)
        torch.cuda.synchronize()

    for _ in range(2):
        pipelined()
    t = time.perf_counter()
    for _ in range(n):
        pipelined()
    tp = (time.perf_counter() - t) / n
    kp = int(out_host[2].sum())
    print(f"pipelined: {tp * 1e3:.1f} ms for {F} frames -> {kp / tp / 1e6:.1f} Mkeypoints/s "
          f"({F * W * H / tp / 1e9:.1f} GB/s of frames over PCIe)")
    # resident reference
    torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(n):
        ex.extract(frames_dev, out_dev)
    torch.cuda.synchronize()
    tr = (time.perf_counter() - t) / n
    print(f"resident : {tr * 1e3:.1f} ms for {F} frames -> {kp / tr / 1e6:.1f} Mkeypoints/s")


if __name__ == "__main__":
    main()
