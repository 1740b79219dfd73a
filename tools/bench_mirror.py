"""CutPaste composition and mirror-loss kernels at the reference's default shape (mirror_pretrain.py:44-46,62: 512 x 512,
batch 10, 2 classes) and at 3 classes, for rocprofv3 --kernel-trace.  Prints the algorithmic bytes per launch."""
import sys; sys.path.insert(0, __import__('os').path.dirname(__import__('os').path.dirname(__import__('os').path.abspath(__file__))))
import numpy as np
import torch
from cp2_amd import mirror as M
dev = 'cuda'
B, H, W = 10, 512, 512
imgs = torch.randint(0, 256, (64, H, W, 3), dtype=torch.uint8, device=dev)
np.random.seed(0)
for ncls, rot in ((2, (0, 0)), (3, (-45, 45))):
    s = M.CutPasteSampler(64, 0.02, 0.15, 1 / 3, 4 / 3, rot[0], rot[1], M.MirrorVariant.OUTPUT, ncls, 1)
    for _ in range(20):
        img, mir, mask = M.cutpaste_batch(imgs, s, list(range(B)))
    g = torch.Generator(device=dev).manual_seed(0)
    sl = torch.randn(B, ncls, H, W, device=dev, generator=g, requires_grad=True)
    tl = torch.randn(B, ncls, H, W, device=dev, generator=g, requires_grad=True)
    for _ in range(20):
        loss, st = M.mirror_loss(sl, tl, mask, 2.0, 0.01)
    torch.cuda.synchronize()
    px = B * H * W
    print(f"classes {ncls}: cutpaste bytes/launch = {px * (2 * 3 + 2 * 12 + 8)} (2 u8 images read, 2 fp32 images + int64 mask written); "
          f"mirror_loss bytes/launch = {px * (2 * ncls * 4 * 2 + 8 + 2 * 8)} (2 logit maps read, 2 gradient maps + 2 argmax maps written, masks read)")
print("ok")
