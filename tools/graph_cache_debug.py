import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from gslam_amd import hip
from gslam_amd.orb import OrbExtractor, synth_frames
ctx = hip.Context(0, stream=torch.cuda.current_stream().cuda_stream)
ex = OrbExtractor(ctx, 320, 240, max_batch=1, n_features=300)
fr = synth_frames(ctx, 1, 320, 240, base_seed=0x5EED0000)
ref = [t.clone() for t in ex.extract(fr)]
torch.cuda.synchronize()
print("ref count", int(ref[2][0]))
for it in range(24):
    out = ex.extract(fr)
    torch.cuda.synchronize()
    eq = [bool(torch.equal(a, b)) for a, b in zip(out, ref)]
    print(it, eq, [hex(t.data_ptr()) for t in out], int(out[2][0]), int((out[1] != ref[1]).sum()))
