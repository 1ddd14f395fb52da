#!/usr/bin/env python3
"""Where a tile of orb_fast_cells spends its time (measuring build of the library: -DGH_ORB_PHASES, GSLAM_HIP_LIB=build/ab/libgslam_hip_ph.so).
Thread 0 of every workgroup stamps the 100 MHz wall clock at the barriers of a tile and stores the differences in the tile's own
record (no contended atomics); averages over one extraction of B x 1080p frames.  (The per-keypoint stamps of orb_describe that the
second table reads -- records marked 2 -- are part of docs/history/orb_describe_asm_waits_r06.patch, not of the tree.)"""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from gslam_amd import hip
from gslam_amd.orb import OrbExtractor, synth_frames

B = int(sys.argv[1]) if len(sys.argv) > 1 else 400
ctx = hip.Context(0, stream=torch.cuda.current_stream().cuda_stream)
ex = OrbExtractor(ctx, 1920, 1080, max_batch=B, n_features=2000)
fr = synth_frames(ctx, B, 1920, 1080)
out = ex.alloc_outputs(B)
slots = B * 8192
buf = torch.zeros((slots, 8), dtype=torch.int32, device="cuda")
f = hip.lib.gh_orb_debug_phases
f.argtypes = [C.c_void_p]
assert f(C.c_void_p(buf.data_ptr())) == 0
ex.extract(fr, out); torch.cuda.synchronize()
buf.zero_()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record(); ex.extract(fr, out); e1.record(); torch.cuda.synchronize()
a_all = buf.cpu().numpy().astype(np.int64)
d = a_all[a_all[:, 6] == 2]  # describe_pipe_kernel: one record per wave (up to 8 keypoints), ticks summed over its keypoints
if len(d):
    dn = ["patch rows -> LDS", "moments + bin", "pattern words / next patch requested / blur (MFMA)", "256 tests (LDS reads + ballots)", "stores / next slot"]
    kp = float(out[2].sum())
    print("orb_describe: %d waves, %d keypoints" % (len(d), int(kp)))
    tot = 0.0
    for k, n in enumerate(dn):
        us = d[:, k].sum() * 0.01 / kp
        tot += us
        print("  %-52s %7.3f us per keypoint" % (n, us))
    print("  %-52s %7.3f us per keypoint per wave" % ("sum", tot))
a = a_all[a_all[:, 6] == 1]
names = ["start -> tile in LDS", "-> resize + pass 1 done", "-> pass 2 done", "-> wave 0's cell done", "-> all waves done (tile loop)", "(after the cell: own stores acknowledged)"]
print("persist=%s: %d tiles, extract %.3f ms" % (os.environ.get("GSLAM_HIP_ORB_PERSIST", "default"), len(a), e0.elapsed_time(e1)))
tot = 0.0
for k, n in enumerate(names):
    us = a[:, k].mean() * 0.01
    tot += us
    print("  %-32s mean %7.3f us  median %7.3f  p90 %7.3f" % (n, us, np.median(a[:, k]) * 0.01, np.percentile(a[:, k], 90) * 0.01))
print("  %-32s %7.3f us per tile  (x tiles / 2048 resident workgroups = %.3f ms)" % ("sum", tot, tot * len(a) / 2048 * 1e-3))
