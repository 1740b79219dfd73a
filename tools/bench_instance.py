"""Instance InfoNCE (a10) kernel pair alone: R rows against a K-key queue, for rocprofv3 --kernel-trace --stats."""
import sys; sys.path.insert(0, __import__('os').path.dirname(__import__('os').path.dirname(__import__('os').path.abspath(__file__))))
import torch
from cp2_amd import ops
dev = 'cuda'
g = torch.Generator(device=dev).manual_seed(0)
C = 128
for R, K in ((32, 65536), (8, 131072)):
    q = torch.nn.functional.normalize(torch.randn(R, C, device=dev, generator=g), dim=1)
    queue = torch.nn.functional.normalize(torch.randn(C, K, device=dev, generator=g), dim=0)
    ext = torch.rand(R, 1, device=dev, generator=g)
    for _ in range(5):
        ops.rowkey_infonce(q, (1, C, 0, 1), R, queue, ext, 0.2, 1.0 / R)
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    n = 100
    for _ in range(n):
        ops.rowkey_infonce(q, (1, C, 0, 1), R, queue, ext, 0.2, 1.0 / R)
    b.record(); torch.cuda.synchronize()
    t = a.elapsed_time(b) / n
    print(f"instance R={R} K={K}: fwd+finalize {t * 1e3:.1f} us per call (host-paced), queue bytes {C * K * 4 / 1e6:.1f} MB")
