"""1x1 weight gradient at BASELINE config 4's shapes (8 x 512^2, ResNet-101 at output stride 8): cp2_wgrad1x1 against
MIOpen (+ its fill / cast launches + the fp32 cast), graph-replay timing."""
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


# (count, N, HW, ci, co)
SHAPES = [(1, 8, 128, 64, 64), (3, 8, 128, 64, 256), (2, 8, 128, 256, 64), (1, 8, 128, 256, 128), (4, 8, 64, 128, 512), (3, 8, 64, 512, 128),
          (1, 8, 64, 512, 256), (23, 8, 64, 256, 1024), (22, 8, 64, 1024, 256), (1, 8, 64, 1024, 512), (3, 8, 64, 512, 2048),
          (2, 8, 64, 2048, 512), (1, 8, 64, 512, 512)]
tot = [0.0, 0.0]
with torch.no_grad():
    for cnt, N, HW, ci, co in SHAPES:
        x = torch.randn(N, ci, HW, HW, device=dev).to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
        w = (torch.randn(co, ci, 1, 1, device=dev) * 0.05).to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
        dy = torch.randn(N, co, HW, HW, device=dev).to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
        args = ([1, 1], [0, 0], [1, 1], False, [0, 0], 1)
        t0 = gtime(lambda: aten.convolution_backward(dy, x, w, None, *args, [False, True, False])[1].float())
        t1 = gtime(lambda: ops.wgrad1x1(dy, x))
        tot[0] += cnt * t0; tot[1] += cnt * t1
        print(f"M={N * HW * HW:6d} {ci:4d}->{co:4d} x{cnt:2d}: MIOpen {t0:7.1f} us   cp2_wgrad1x1 {t1:7.1f} us   ({2.0 * N * HW * HW * ci * co / 1e9:6.1f} GFLOP)", flush=True)
print("per step: MIOpen", round(tot[0]), "us, cp2_wgrad1x1", round(tot[1]), "us")
