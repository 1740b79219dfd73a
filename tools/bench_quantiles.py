"""Logging-quantile kernels on the distributions training actually produces, at the bench step's shapes (BASELINE config 2:
32 x 65536 queue logits + 2 x 32 x 196^2 dense pairs, one quantiles_row_kernel launch) and at BASELINE config 4's (8 x 4096^2
dense logits + 8 x 131072, the chunked six-launch form).  Distributions:
    spread    N(0, 0.088)       unit vectors against a random queue / an untrained projector
    band      N(0.87, 0.02)     positive dense scores of a trained encoder: two or three hot first-level bins
    tight     N(1, 1e-6)        a freshly initialised encoder: every logit within a few ulps of 1
Run under rocprofv3 --kernel-trace and summarise with tools/kstats.py; stand-alone it prints hipEvent times per call.
    python tools/bench_quantiles.py [iters]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from cp2_amd import ops  # noqa: E402

dev = "cuda"
g = torch.Generator(device=dev).manual_seed(0)
DISTS = (("spread", 0.0, 0.088), ("band", 0.87, 0.02), ("tight", 1.0, 1e-6))


def step_shape(B, K, P, iters, centre, spread, check=True):
    lneg = centre + torch.randn(B, K, device=dev, generator=g) * spread
    logits = centre + torch.randn(B, P, P, device=dev, generator=g) * spread
    ma = (torch.rand(B, P, device=dev, generator=g) > 0.4).float()
    mb = (torch.rand(B, P, device=dev, generator=g) > 0.5).float()
    dense = dict(x=logits, stride_row=P * P, stride_elem=1, R=B, N=P * P, mask_a=ma, mask_b=mb)
    jobs = [dict(dense, want=1), dict(dense, want=0), dict(x=lneg, stride_row=K, stride_elem=1, R=B, N=K)]
    for _ in range(3):
        pos, neg, a = ops.masked_quantiles_multi(jobs)
    t0, t1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    t0.record()
    for _ in range(iters):                                        # the step's three statistics in one call
        pos, neg, a = ops.masked_quantiles_multi(jobs)
    t1.record()
    torch.cuda.synchronize()
    if check:
        assert torch.equal(a.cpu(), torch.quantile(lneg.cpu(), torch.tensor([0.25, 0.5, 0.75]), dim=1))
    return t0.elapsed_time(t1) / iters * 1e3


iters = int(sys.argv[1]) if len(sys.argv) > 1 else 20
if len(sys.argv) > 2:                 # one distribution only (for a profiler run): python tools/bench_quantiles.py 8 tight
    DISTS = tuple(d for d in DISTS if d[0] == sys.argv[2])
print(f"{'shape':34s} " + " ".join(f"{n:>12s}" for n, _, _ in DISTS) + "   (us per call, hipEvents around back-to-back calls)")
for label, B, K, P, it in (("cfg2 step: 32x65536 + 2x32x196^2", 32, 65536, 196, iters), ("cfg4 step: 8x131072 + 2x8x4096^2", 8, 131072, 4096, max(3, iters // 4))):
    row = [step_shape(B, K, P, it, c, s_, check=K <= 65536) for _, c, s_ in DISTS]
    print(f"{label:34s} " + " ".join(f"{v:12.1f}" for v in row))
print("ok")
