import sys; sys.path.insert(0, __import__('os').path.dirname(__import__('os').path.dirname(__import__('os').path.abspath(__file__))))
import torch
from cp2_amd import ops
dev = 'cuda'
def timeit(fn, n=10, warm=3):
    for _ in range(warm): fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / n
g = torch.Generator(device=dev).manual_seed(0)
# T19: config 5
b, C, S2, K = 32, 128, 196, 65536
rows = torch.nn.functional.normalize(torch.randn(b, C, S2, device=dev, generator=g), dim=1)
queue = torch.nn.functional.normalize(torch.randn(C, K, device=dev, generator=g), dim=0)
pos = torch.rand(b * S2, 1, device=dev, generator=g) * 2 - 1
R = b * S2
lay = (S2, C * S2, 1, S2)
fl1 = 2.0 * R * C * K
for prec, pre in (("f32", False), ("bf16x3", False), ("bf16x3", True)):
    t_fwd = timeit(lambda: ops.rowkey_infonce(rows, lay, R, queue, pos, 0.2, None, precision=prec, presplit=pre))
    t_all = timeit(lambda: ops.rowkey_infonce(rows, lay, R, queue, pos, 0.2, 1.0 / R, precision=prec, presplit=pre))
    print(f"T19 [{prec}{' presplit' if pre else ''}] R={R} K={K}: fwd-only {t_fwd:.3f} ms = {fl1 / t_fwd / 1e9:.1f} TFLOP/s ; fwd+grad {t_all:.3f} ms = {2 * fl1 / t_all / 1e9:.1f} TFLOP/s (algorithmic)")
# instance: config 2
q_pos = torch.nn.functional.normalize(torch.randn(32, C, device=dev, generator=g), dim=1)
ext = torch.rand(32, 1, device=dev, generator=g)
t_ins = timeit(lambda: ops.rowkey_infonce(q_pos, (1, C, 0, 1), 32, queue, ext, 0.2, 1.0 / 32), n=50)
print(f"instance R=32 K={K}: fwd+grad {t_ins * 1e3:.1f} us (queue read once = {C * K * 4 / t_ins / 1e6:.0f} GB/s)")
queue4 = torch.nn.functional.normalize(torch.randn(C, 131072, device=dev, generator=g), dim=0)
q8, e8 = q_pos[:8].contiguous(), ext[:8].contiguous()
t_ins4 = timeit(lambda: ops.rowkey_infonce(q8, (1, C, 0, 1), 8, queue4, e8, 0.2, 1.0 / 8), n=50)
print(f"instance R=8 K=131072: fwd+grad {t_ins4 * 1e3:.1f} us (queue read once = {C * 131072 * 4 / t_ins4 / 1e6:.0f} GB/s)")
# dense: config 4 (P=4096, B=8) and config 2 (P=196, B=32)
for B, P in ((8, 4096), (8, 1024), (32, 196)):
    qd = torch.nn.functional.normalize(torch.randn(B, C, P, device=dev, generator=g), dim=1)
    kd = torch.nn.functional.normalize(torch.randn(B, C, P, device=dev, generator=g), dim=1)
    ma = (torch.rand(B, P, device=dev, generator=g) > 0.4).float(); mb = (torch.rand(B, P, device=dev, generator=g) > 0.5).float()
    tf = timeit(lambda: ops.dense_infonce_fwd(qd, kd, ma, mb, 1.0))
    fw = ops.dense_infonce_fwd(qd, kd, ma, mb, 1.0)
    tb = timeit(lambda: ops.dense_infonce_bwd(qd, kd, ma, mb, 1.0, fw, 0.2 / B))
    fl = 2.0 * B * P * P * C
    print(f"dense B={B} P={P}: fwd {tf * 1e3:.1f} us = {fl / tf / 1e9:.1f} TFLOP/s ; bwd {tb * 1e3:.1f} us = {2 * fl / tb / 1e9:.1f} TFLOP/s")
