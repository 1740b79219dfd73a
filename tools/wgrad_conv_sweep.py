"""Sweep of the k x k weight-gradient kernel's workgroup target over the ResNet-50 + FCN head 3x3 shapes (32 images, 224^2)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from cp2_amd import ops
dev = "cuda"
# (count in the step, N, ci, co, H, stride, pad, dil)
SHAPES = [(3, 32, 64, 64, 56, 1, 1, 1), (1, 32, 128, 128, 56, 2, 1, 1), (3, 32, 128, 128, 28, 1, 1, 1), (1, 32, 256, 256, 28, 2, 1, 1),
          (5, 32, 256, 256, 14, 1, 1, 1), (1, 32, 512, 512, 14, 1, 1, 1), (2, 32, 512, 512, 14, 1, 2, 2),
          (1, 32, 2048, 512, 14, 1, 1, 1), (1, 32, 2560, 512, 14, 1, 1, 1)]
tot = 0.0
for cnt, N, ci, co, H, st, pad, dil in SHAPES:
    OH = (H + 2 * pad - dil * 2 - 1) // st + 1
    x = torch.randn(N, ci, H, H, device=dev).to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
    dy = torch.randn(N, co, OH, OH, device=dev).to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
    for _ in range(3):
        ops.wgrad_conv(dy, x, 3, st, pad, dil)
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(20):
        ops.wgrad_conv(dy, x, 3, st, pad, dil)
    b.record(); torch.cuda.synchronize()
    us = a.elapsed_time(b) / 20 * 1e3
    tot += cnt * us
    print(f"  {ci:4d}->{co:4d} {H:3d}x{H:<3d} s{st} d{dil}: {us:7.1f} us x{cnt}")
print(f"target {os.environ.get('CP2_WGRAD_TARGET_CONV', '512')}: {tot:.0f} us per step")
