"""sgd_flat_kernel alone (one 47.4 M-float tensor slot, momentum + weight decay + bf16 image) for the
`rocprofv3 --pmc FETCH_SIZE` / `--pmc WRITE_SIZE` passes -> profiles/sgd_traffic.json."""
import sys; sys.path.insert(0, __import__('os').path.dirname(__import__('os').path.dirname(__import__('os').path.abspath(__file__))))
import torch
from cp2_amd import ops
n = 47433472
p = torch.randn(n, device='cuda'); g = torch.randn(n, device='cuda') * 0.01; buf = torch.zeros(n, device='cuda')
img = torch.empty(n, dtype=torch.bfloat16, device='cuda')
flush = torch.empty(192 << 20, device='cuda')
plan = ops.SgdFlatPlan([0], [n], p.device)
plan.grads[0] = g.data_ptr()
for i in range(6):
    flush.fill_(float(i))          # push the buffers out of the 256 MiB infinity cache, as the rest of a step does
    ops.sgd_flat(plan, p, buf, img, 0.03, 0.9, 1e-4)
torch.cuda.synchronize()
print("done", n)
