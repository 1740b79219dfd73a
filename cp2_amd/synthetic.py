"""Synthetic batches with the reference loader's output contract (reference
loader.py:66-118, main.py:206-245): images in [0,1), backgrounds with one exactly-zero rectangle
(RandomErasing(value=0) with area in [foreground_min, foreground_max]), and pixel-id maps of two
shifted / flipped crops of one id grid; region ids = pixel ids (MappingType.CP2, loader.py:84-85).
Generated directly on the target device so the benchmark starts with data resident in HBM."""
from __future__ import annotations

import math
from typing import Dict

import torch


def make_batch(b: int, h: int, w: int, device, seed: int = 0, foreground_min: float = 0.5,
               foreground_max: float = 0.8) -> Dict[str, torch.Tensor]:
    gen = torch.Generator(device="cpu").manual_seed(seed)
    dgen = torch.Generator(device=device).manual_seed(seed)
    img_a = torch.rand(b, 3, h, w, device=device, generator=dgen)
    img_b = torch.rand(b, 3, h, w, device=device, generator=dgen)
    ys = torch.arange(h, device=device)[None, :, None]
    xs = torch.arange(w, device=device)[None, None, :]
    bgs = []
    for _ in range(2):
        bg = torch.rand(b, 3, h, w, device=device, generator=dgen)
        area = torch.empty(b).uniform_(foreground_min, foreground_max, generator=gen) * h * w
        logr = torch.empty(b).uniform_(math.log(0.8), math.log(1.25), generator=gen)
        rh = torch.sqrt(area * torch.exp(logr)).round().clamp(1, h).long()
        rw = torch.sqrt(area / torch.exp(logr)).round().clamp(1, w).long()
        y0 = (torch.rand(b, generator=gen) * (h - rh + 1)).long()
        x0 = (torch.rand(b, generator=gen) * (w - rw + 1)).long()
        y0d, x0d, rhd, rwd = (t.to(device)[:, None, None] for t in (y0, x0, rh, rw))
        hole = (ys >= y0d) & (ys < y0d + rhd) & (xs >= x0d) & (xs < x0d + rwd)          # b x h x w
        bgs.append(bg.masked_fill(hole[:, None], 0.0))
    # view a: ids 1..h*w of sample n's own grid; view b: the same grid shifted by (dy,dx), odd samples flipped;
    # cells that fall outside view a's grid get fresh ids (unique, never matching)
    per = 4 * h * w
    base = torch.arange(b, device=device)[:, None, None] * per
    grid = (ys * (2 * w) + xs + 1)                                                     # 1 x h x w within a 2h x 2w canvas
    pixel_a = base + grid
    dy = (torch.rand(b, generator=gen) * (h // 2)).long().to(device)[:, None, None]
    dx = (torch.rand(b, generator=gen) * (w // 2)).long().to(device)[:, None, None]
    pixel_b = base + (ys + dy) * (2 * w) + (xs + dx) + 1
    flip = (torch.arange(b, device=device) % 2 == 1)[:, None, None]
    pixel_b = torch.where(flip, pixel_b.flip(2), pixel_b)
    return dict(img_a=img_a, img_b=img_b, bg0=bgs[0], bg1=bgs[1], pixel_ids_a=pixel_a.contiguous(),
                pixel_ids_b=pixel_b.contiguous(), region_ids_a=pixel_a.clone(), region_ids_b=pixel_b.clone())
