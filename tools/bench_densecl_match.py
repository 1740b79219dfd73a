"""cp2_densecl_match (DenseCL positive selection, reference builder.py:818-864) at BASELINE config 5's shape -- 32 x 196
pixels, 2048 backbone channels as bf16 channels-last, 128 projection channels -- in the forms the step launches: plain,
with the coordinate mix, and with the logged matching-positives rate (rank 0).  hipEvent time per launch and the algorithmic
rates; run under rocprofv3 --kernel-trace for the kernel's own duration.    python tools/bench_densecl_match.py [iters]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from cp2_amd import ops  # noqa: E402

dev = "cuda"
iters = int(sys.argv[1]) if len(sys.argv) > 1 else 30
B, CE, P = 32, 2048, 196
g = torch.Generator(device=dev).manual_seed(0)
qe = torch.randn(B, CE, 14, 14, device=dev, generator=g).relu().bfloat16().contiguous(memory_format=torch.channels_last)
ke = torch.randn(B, CE, 14, 14, device=dev, generator=g).relu().bfloat16().contiguous(memory_format=torch.channels_last)
ql = torch.nn.functional.normalize(torch.randn(B, 128, P, device=dev, generator=g), dim=1)
kl = torch.nn.functional.normalize(torch.randn(B, 128, P, device=dev, generator=g), dim=1)
ids_q = torch.arange(1, P + 1, device=dev).repeat(B, 1) + 1000 * torch.arange(B, device=dev)[:, None]
ids_k = ids_q.roll(17, 1).clone()
ids_k[:, ::3] += 500                                       # two thirds of the pixels have a coordinate match
k_row = torch.randperm(B, device=dev)


def run(label, **kw):
    for _ in range(3):
        ops.densecl_match(qe, ke, ql, kl, k_row=k_row, normalize_k=True, **kw)
    t0, t1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    t0.record()
    for _ in range(iters):
        ops.densecl_match(qe, ke, ql, kl, k_row=k_row, normalize_k=True, **kw)
    t1.record()
    torch.cuda.synchronize()
    us = t0.elapsed_time(t1) / iters * 1e3
    flops, byts = 2.0 * B * P * P * CE, 2.0 * 2 * B * P * CE
    print(f"{label:44s} {us:8.1f} us   {flops / us / 1e6:7.1f} TFLOP/s (bf16 MFMA)   {byts / us / 1e3:7.1f} GB/s of feature maps read once")


run("plain (no ids)")
run("ids, coordinate mix 0.3", ids_q=ids_q, ids_k=ids_k, lmbd_coordinate=0.3)
run("ids, mix 0.3, matching rate (rank 0 logging)", ids_q=ids_q, ids_k=ids_k, lmbd_coordinate=0.3, want_metrics=True)
run("ids, no mix, matching rate (DENSECL rank 0)", ids_q=ids_q, ids_k=ids_k, want_metrics=True)
qf, kf = qe.float(), ke.float()
qe, ke = qf, kf
run("fp32 features (no autocast): f32 MFMA")
