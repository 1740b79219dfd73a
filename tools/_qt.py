import sys, os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from cp2_amd import ops
dev = 'cuda'
g = torch.Generator(device=dev).manual_seed(0)
B, K, P = 32, 65536, 196
def t(jobs, n=50, **kw):
    for _ in range(5): ops.masked_quantiles_multi(jobs, **kw)
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); a.record()
    for _ in range(n): ops.masked_quantiles_multi(jobs, **kw)
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e3
ma = (torch.rand(B, P, device=dev, generator=g) > 0.4).float(); mb = (torch.rand(B, P, device=dev, generator=g) > 0.5).float()
for name, centre, spread in (("wide", 0.0, 0.44), ("narrow", 0.87, 0.02), ("very narrow", 1.0, 1e-6)):
    lneg = centre + torch.randn(B, K, device=dev, generator=g) * spread
    logits = centre + torch.randn(B, P, P, device=dev, generator=g) * spread
    dense = dict(x=logits, stride_row=P * P, stride_elem=1, R=B, N=P * P, mask_a=ma, mask_b=mb)
    ln = dict(x=lneg, stride_row=K, stride_elem=1, R=B, N=K)
    print(f"{name:12s} lneg only {t([ln]):6.1f} us   dense pos only {t([dict(dense, want=1)]):6.1f}   dense neg only {t([dict(dense, want=0)]):6.1f}   all three {t([dict(dense, want=1), dict(dense, want=0), ln]):6.1f}", flush=True)
