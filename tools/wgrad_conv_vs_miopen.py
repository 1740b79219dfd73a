"""k x k weight gradient: cp2_wgrad_conv against MIOpen (weight gradient + its zero-fill / cast launches + the bf16->fp32
cast of the result), per distinct 3x3 shape of BASELINE configs 2 and 4, timed as graph replays (no host gaps)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("MIOPEN_FIND_MODE", "FAST")
import torch
from cp2_amd import ops
torch.backends.cudnn.benchmark = True
dev = torch.device("cuda", 0)
aten = torch.ops.aten


def gtime(f, n=10):
    for _ in range(3): f()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(n): f()
    g.replay(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); g.replay(); g.replay(); e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / (2 * n) * 1e3


# (config, count, N, ci, co, H, stride, pad, dil)
SHAPES = [("cfg2", 3, 32, 64, 64, 56, 1, 1, 1), ("cfg2", 1, 32, 128, 128, 56, 2, 1, 1), ("cfg2", 3, 32, 128, 128, 28, 1, 1, 1),
          ("cfg2", 1, 32, 256, 256, 28, 2, 1, 1), ("cfg2", 5, 32, 256, 256, 14, 1, 1, 1), ("cfg2", 1, 32, 512, 512, 14, 1, 1, 1),
          ("cfg2", 2, 32, 512, 512, 14, 1, 2, 2), ("cfg2", 1, 32, 2048, 512, 14, 1, 1, 1), ("cfg2", 1, 32, 2560, 512, 14, 1, 1, 1),
          ("cfg4", 3, 8, 64, 64, 128, 1, 1, 1), ("cfg4", 1, 8, 128, 128, 128, 2, 1, 1), ("cfg4", 3, 8, 128, 128, 64, 1, 1, 1),
          ("cfg4", 1, 8, 256, 256, 64, 1, 1, 1), ("cfg4", 22, 8, 256, 256, 64, 1, 2, 2), ("cfg4", 1, 8, 512, 512, 64, 1, 2, 2),
          ("cfg4", 2, 8, 512, 512, 64, 1, 4, 4), ("cfg4", 1, 8, 2048, 512, 64, 1, 1, 1), ("cfg4", 1, 8, 2560, 512, 64, 1, 1, 1)]
tot = {}
with torch.no_grad():
    for cfg, cnt, N, ci, co, H, st, pad, dil in SHAPES:
        OH = (H + 2 * pad - dil * 2 - 1) // st + 1
        x = torch.randn(N, ci, H, H, device=dev).to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
        w = (torch.randn(co, ci, 3, 3, device=dev) * 0.05).to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
        dy = torch.randn(N, co, OH, OH, device=dev).to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
        args = ([st, st], [pad, pad], [dil, dil], False, [0, 0], 1)
        t0 = gtime(lambda: aten.convolution_backward(dy, x, w, None, *args, [False, True, False])[1].float())
        t1 = gtime(lambda: ops.wgrad_conv(dy, x, 3, st, pad, dil))
        a = tot.setdefault(cfg, [0.0, 0.0]); a[0] += cnt * t0; a[1] += cnt * t1
        print(f"{cfg} M={N * OH * OH:6d} {ci:4d}->{co:4d} {H:3d}^2 s{st} d{dil} x{cnt:2d}: MIOpen (+fill, cast, fp32 cast) {t0:7.1f} us   cp2_wgrad_conv {t1:7.1f} us", flush=True)
for k, v in tot.items():
    print(k, "per step: MIOpen", round(v[0]), "us, cp2_wgrad_conv", round(v[1]), "us")
