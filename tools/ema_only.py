import sys; sys.path.insert(0, __import__('os').path.dirname(__import__('os').path.dirname(__import__('os').path.abspath(__file__))))
import torch
from cp2_amd import ops
n = 47433472
k = torch.randn(n, device='cuda'); q = torch.randn(n, device='cuda')
flush = torch.empty(192 << 20, device='cuda')
kb = torch.empty(n, dtype=torch.bfloat16, device='cuda')
for i in range(6):
    flush.fill_(float(i))          # push k/q out of the 256 MiB infinity cache, as the rest of a step does
    ops.ema_flat_shadow(k, q, kb, 0.999)   # the variant the training step launches (fp32 k + bf16 shadow)
torch.cuda.synchronize()
print("done", n)
