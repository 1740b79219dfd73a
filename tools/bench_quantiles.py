"""Logging-quantile kernel at the bench shapes (for rocprofv3 --kernel-trace)."""
import sys; sys.path.insert(0, __import__('os').path.dirname(__import__('os').path.dirname(__import__('os').path.abspath(__file__))))
import torch
from cp2_amd import ops
dev = 'cuda'
g = torch.Generator(device=dev).manual_seed(0)
B, K, P = 32, 65536, 196
lneg = torch.randn(B, K, device=dev, generator=g) * 0.088
logits = torch.randn(B, P, P, device=dev, generator=g) * 0.088
ma = (torch.rand(B, P, device=dev, generator=g) > 0.4).float(); mb = (torch.rand(B, P, device=dev, generator=g) > 0.5).float()
for _ in range(20):
    a = ops.masked_quantiles(lneg, K, 1, B, K)
    b = ops.masked_quantiles(logits, P * P, 1, B, P * P, mask_a=ma, mask_b=mb, want=1)
    c = ops.masked_quantiles(logits, P * P, 1, B, P * P, mask_a=ma, mask_b=mb, want=0)
torch.cuda.synchronize()
assert torch.equal(a.cpu(), torch.quantile(lneg.cpu(), torch.tensor([0.25, 0.5, 0.75]), dim=1))
print("ok")
