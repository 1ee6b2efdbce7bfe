"""HBM fill / copy / read-reduction rates of torch's own kernels on this box: the practical ceilings quoted in DESIGN.md."""
import torch, time
dev = torch.device("cuda:0")
def t(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize(); a = torch.cuda.Event(enable_timing=True); b = torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e-3
for mb in (128, 1024):
    x = torch.empty(mb * 1024 * 1024 // 4, device=dev); y = torch.empty_like(x)
    s = t(lambda: x.fill_(1.0)); print(f"{mb} MB fill  {mb/1024/s/1e3*1.0737:.2f} TB/s")
    s = t(lambda: y.copy_(x)); print(f"{mb} MB copy  {2*mb/1024/s/1e3*1.0737:.2f} TB/s (r+w)")
    s = t(lambda: x.sum()); print(f"{mb} MB sum   {mb/1024/s/1e3*1.0737:.2f} TB/s")
