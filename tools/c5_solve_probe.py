"""One dense SPD solve at n = 60 000 (the C5 reduced camera system's size) for counter collection."""
import ctypes as C, os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gslam_amd import hip
n = int(sys.argv[1]) if len(sys.argv) > 1 else 60000
torch.cuda.set_device(0)
ctx = hip.Context(0, stream=torch.cuda.current_stream().cuda_stream)
g = torch.Generator(device="cuda").manual_seed(1)
M = torch.randn((n, 64), dtype=torch.float64, device="cuda", generator=g)
lda = n + 16
A = torch.zeros((n, lda), dtype=torch.float64, device="cuda")
if len(sys.argv) > 3 and sys.argv[3] == "blocks":  # block-diagonal (3000-wide blocks): the same launches multiply mostly zeros
    for k0 in range(0, n, 3000):
        k1 = min(k0 + 3000, n)
        A[k0:k1, k0:k1] = M[k0:k1] @ M[k0:k1].T / 64.0
else:
    A[:, :n] = M @ M.T / 64.0
A[:, :n].diagonal().add_(2.0)
b = torch.randn(n, dtype=torch.float64, device="cuda", generator=g)
del M
for rep in range(int(sys.argv[2]) if len(sys.argv) > 2 else 1):
    a, x = A.clone(), b.clone()
    info = C.c_int()
    torch.cuda.synchronize()
    t = time.perf_counter()
    ctx.check(hip.lib.gh_potrf_solve_dev(ctx.h, C.c_void_p(a.data_ptr()), n, lda, C.c_void_p(x.data_ptr()), C.byref(info)))
    ctx.sync()
    dt = time.perf_counter() - t
    print(f"n = {n}: {dt * 1e3:.1f} ms, {n ** 3 / 3 / dt / 1e12:.1f} TFLOP/s, info {info.value}")
    del a
