import torch, sys
x = torch.randint(0, 1000, (256, 100000, 8), dtype=torch.int32, device="cuda")
y = torch.empty_like(x)
def t(fn, n=20):
    for _ in range(3): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n
b = x.numel() * 4
ms = t(lambda: y.copy_(x)); print(f"copy  {ms*1e3:7.1f} us  {2*b/ms/1e6:7.1f} GB/s (read+write)")
ms = t(lambda: x.sum());   print(f"sum   {ms*1e3:7.1f} us  {b/ms/1e6:7.1f} GB/s (read)")
xf = x.view(torch.float32)
ms = t(lambda: xf.sum());  print(f"fsum  {ms*1e3:7.1f} us  {b/ms/1e6:7.1f} GB/s (read)")
ms = t(lambda: torch.max(x)); print(f"max   {ms*1e3:7.1f} us  {b/ms/1e6:7.1f} GB/s (read)")
