import torch, time
x = torch.empty(2_000_000_000, dtype=torch.uint8, device='cuda')
y = torch.empty_like(x)
for _ in range(3): y.copy_(x)
torch.cuda.synchronize()
t=time.perf_counter()
for _ in range(10): y.copy_(x)
torch.cuda.synchronize()
dt=(time.perf_counter()-t)/10
print("copy 2GB: %.3f ms -> %.2f TB/s (read+write)" % (dt*1e3, 4e9/dt/1e12))
x.zero_(); torch.cuda.synchronize()
t=time.perf_counter()
for _ in range(10): x.zero_()
torch.cuda.synchronize()
dt=(time.perf_counter()-t)/10
print("fill 2GB: %.3f ms -> %.2f TB/s" % (dt*1e3, 2e9/dt/1e12))
xf = x.view(torch.float32)
t=time.perf_counter()
for _ in range(10): s = xf.sum()
torch.cuda.synchronize()
dt=(time.perf_counter()-t)/10
print("sum 2GB: %.3f ms -> %.2f TB/s" % (dt*1e3, 2e9/dt/1e12))
