"""All-pairs matching rate: popcount kernel vs the MFMA formulation (same results)."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gslam_amd import hip
from gslam_amd.matcher import BFMatcher
from gslam_amd.sharding import all_pairs_block
torch.cuda.set_device(0)
ctx = hip.Context(0, stream=torch.cuda.current_stream().cuda_stream)
F, K = int(sys.argv[1]) if len(sys.argv) > 1 else 128, 2000
g = torch.Generator(device="cuda").manual_seed(1)
desc = torch.randint(0, 256, (F, K, 32), dtype=torch.uint8, device="cuda", generator=g)
counts = torch.full((F,), K, dtype=torch.int32, device="cuda")
ai, aj = all_pairs_block(0, 1, F, "cuda")
m = BFMatcher(ctx)
out = m.match_pairs(desc, counts, ai, aj)
for mf in (False, True):
    o = m.match_pairs(desc, counts, ai, aj, mfma=mf)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(3):
        m.match_pairs(desc, counts, ai, aj, out=o, mfma=mf)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / 3
    same = all(torch.equal(x, y) for x, y in zip(out, o))
    print(f"{'mfma' if mf else 'popcount'}: {ai.shape[0]} frame pairs, {dt * 1e3:.2f} ms, {ai.shape[0] * K * K / dt / 1e9:.0f} Gpairs/s, identical={same}")
