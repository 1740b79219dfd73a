"""Which MIOpen solver serves the weight gradient of the ResNet-50 1x1 convolutions (bf16, channels-last, batch 32)?
Run with MIOPEN_LOG_LEVEL=6 and grep the log; part of the hipGraph replay investigation."""
import os
os.environ.setdefault("MIOPEN_FIND_MODE", "FAST")
import torch
torch.backends.cudnn.benchmark = True
dev = torch.device("cuda", 0)
aten = torch.ops.aten
shapes = [(32, 14, 512, 2048, 1), (32, 14, 2048, 512, 1), (32, 28, 256, 128, 1), (32, 56, 256, 64, 1), (32, 28, 1024, 2048, 2), (32, 14, 1024, 256, 1)]
with torch.no_grad():
    for (N, HW, ci, co, st) in shapes:
        x = torch.randn(N, ci, HW, HW, device=dev).to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
        w = (torch.randn(co, ci, 1, 1, device=dev) * 0.05).to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
        ho = (HW - 1) // st + 1
        dy = torch.randn(N, co, ho, ho, device=dev).to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
        print(f"### shape M={N*ho*ho} {ci}->{co} stride {st}", flush=True)
        for _ in range(2):
            aten.convolution_backward(dy, x, w, None, [st, st], [0, 0], [1, 1], False, [0, 0], 1, [False, True, False])
        torch.cuda.synchronize()
