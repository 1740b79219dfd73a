"""Logging-quantile kernels at the bench shapes and at BASELINE config 4's dense shape (for rocprofv3 --kernel-trace)."""
import sys; sys.path.insert(0, __import__('os').path.dirname(__import__('os').path.dirname(__import__('os').path.abspath(__file__))))
import torch
from cp2_amd import ops
dev = 'cuda'
g = torch.Generator(device=dev).manual_seed(0)


def step_shape(B, K, P, iters, centre=0.0, spread=0.088):
    lneg = centre + torch.randn(B, K, device=dev, generator=g) * spread
    logits = centre + torch.randn(B, P, P, device=dev, generator=g) * spread
    ma = (torch.rand(B, P, device=dev, generator=g) > 0.4).float(); mb = (torch.rand(B, P, device=dev, generator=g) > 0.5).float()
    dense = dict(x=logits, stride_row=P * P, stride_elem=1, R=B, N=P * P, mask_a=ma, mask_b=mb)
    for _ in range(iters):                                        # the step's three statistics in one call
        pos, neg, a = ops.masked_quantiles_multi([dict(dense, want=1), dict(dense, want=0),
                                                  dict(x=lneg, stride_row=K, stride_elem=1, R=B, N=K)])
    torch.cuda.synchronize()
    if K <= 65536:
        assert torch.equal(a.cpu(), torch.quantile(lneg.cpu(), torch.tensor([0.25, 0.5, 0.75]), dim=1))


step_shape(32, 65536, 196, 20)        # BASELINE config 2 (the bench step): quantiles_row_kernel, one workgroup per row
ops.QUANTILES_FORM = 2
step_shape(32, 65536, 196, 20)        # the same through quantiles_coop_kernel (one workgroup per chunk, row-local barriers)
ops.QUANTILES_FORM = 0
step_shape(8, 131072, 4096, 5)        # BASELINE config 4: 8 x 4096^2 dense logits (537 MB), K = 131072
step_shape(8, 131072, 4096, 5, centre=0.87, spread=0.02)   # the same with the narrow band of a freshly initialised encoder
print("ok")
