import sys; sys.path.insert(0, __import__('os').path.dirname(__import__('os').path.dirname(__import__('os').path.abspath(__file__))))
import torch
from cp2_amd import ops
dev = 'cuda'
g = torch.Generator(device=dev).manual_seed(0)
b, C, S2, K = 32, 128, 196, 65536
rows = torch.nn.functional.normalize(torch.randn(b, C, S2, device=dev, generator=g), dim=1)
queue = torch.nn.functional.normalize(torch.randn(C, K, device=dev, generator=g), dim=0)
pos = torch.rand(b * S2, 1, device=dev, generator=g) * 2 - 1
R = b * S2
for prec in ("f32", "bf16x3"):
    for _ in range(4):
        ops.rowkey_infonce(rows, (S2, C * S2, 1, S2), R, queue, pos, 0.2, 1.0 / R, precision=prec)
B, P = 8, 4096
qd = torch.nn.functional.normalize(torch.randn(B, C, P, device=dev, generator=g), dim=1)
kd = torch.nn.functional.normalize(torch.randn(B, C, P, device=dev, generator=g), dim=1)
ma = (torch.rand(B, P, device=dev, generator=g) > 0.4).float(); mb = (torch.rand(B, P, device=dev, generator=g) > 0.5).float()
for _ in range(3):
    fw = ops.dense_infonce_fwd(qd, kd, ma, mb, 1.0)
    ops.dense_infonce_bwd(qd, kd, ma, mb, 1.0, fw, 0.2 / B)
l = torch.randn(32, K, device=dev, generator=g)
for _ in range(3):
    ops.masked_quantiles(l, K, 1, 32, K)
torch.cuda.synchronize()
