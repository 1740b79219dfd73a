"""On-device input pipeline: the two-crop foreground views with their pixel / region id maps and the
two erased background views of a CP2 step, made by HIP kernels (csrc/augment.hip) from a dataset that is resident in
HBM, with the random parameters drawn on the host the way the reference's transforms draw them:

    reference                                               here
    loader.py:50-118  A.RandomResizedCrop + A.HorizontalFlip  rrc_params() + flips -> ops.crop_resize_flip (image + ids)
    loader.py:39-43,66-73  pixel ids at pixel_ids_stride       id_stride argument of the kernel
    main.py:204-225   RandomResizedCrop, flip, RandomErasing   rrc_params(), erase_params() -> crop_resize_flip + erase_rect
    main.py:263-289   three DistributedSamplers (seeds 0/1024/2048)   EpochSampler
    main.py:212-216, loader.py:121-152  ColorJitter(0.4,0.4,0.4,0.1) p=0.8, grayscale p=0.2, GaussianBlur([0.1,2]) p=0.5
                                                            jitter_table() / blur_table() -> ops.color_ops, ops.blur_to_tensor

Pixel values.  Background views follow torchvision's PIL code path in Pillow's own integer / float arithmetic
(csrc/photometric.hip; pinned against Pillow by tests/golden/make_augment_goldens.py): crop + antialiased bilinear resize,
the four colour adjustments in their drawn order, grayscale, Gaussian blur, flip, ToTensor, RandomErasing.  Foreground
views are cv2 / albumentations in the reference (loader.py:93-109), neither installed nor pinned: the crop of a uint8
dataset is resampled in cv2.resize's INTER_LINEAR integer arithmetic as OpenCV's source publishes it (11-bit weights, int32
horizontal pass, the vertical pass's shifts, the 2 x 2 area special case; oracle/augment_oracle.py cv2_resize_linear_u8) --
restated from the dependency's source, PARITY-UNPINNED -- then goes through A.ColorJitter / A.ToGray in albumentations' own
arithmetic on cv2 (float64 look-up tables truncated to uint8, cv2's 8-bit RGB2GRAY / RGB2HSV / HSV2RGB, addWeighted: the
`arithmetic = 1` rows of cp2_color_ops, restated in oracle/augment_oracle.py albu_* / cv2_*, unpinned as well) and
Pillow's Gaussian blur, which IS what the reference calls for them (loader.py:136-152).
"""
from __future__ import annotations

import math
from typing import Dict, Optional

import numpy as np
import torch

from . import ops


def rrc_params(rng: np.random.Generator, n: int, hs: int, ws: int, scale=(0.2, 1.0), ratio=(3.0 / 4.0, 4.0 / 3.0)) -> np.ndarray:
    """RandomResizedCrop.get_params for n samples (the algorithm torchvision and albumentations share): up to ten
    draws of (area fraction ~ U(scale), log aspect ~ U(log ratio)); the first box that fits is placed uniformly; if none
    fits, the central crop with the aspect clamped into `ratio`.  Returns int32 [n, 4] = top, left, h, w."""
    area = float(hs * ws)
    ta = area * rng.uniform(scale[0], scale[1], size=(n, 10))
    ar = np.exp(rng.uniform(math.log(ratio[0]), math.log(ratio[1]), size=(n, 10)))
    w = np.rint(np.sqrt(ta * ar)).astype(np.int64)
    h = np.rint(np.sqrt(ta / ar)).astype(np.int64)
    ok = (w > 0) & (w <= ws) & (h > 0) & (h <= hs)
    first = np.where(ok.any(1), ok.argmax(1), -1)
    in_ratio = ws / hs
    if in_ratio < min(ratio):
        fw, fh = ws, int(round(ws / min(ratio)))
    elif in_ratio > max(ratio):
        fh, fw = hs, int(round(hs * max(ratio)))
    else:
        fw, fh = ws, hs
    rows = np.arange(n)
    hh = np.where(first >= 0, h[rows, np.maximum(first, 0)], fh)
    ww = np.where(first >= 0, w[rows, np.maximum(first, 0)], fw)
    top = np.where(first >= 0, np.floor(rng.random(n) * (hs - hh + 1)).astype(np.int64), (hs - hh) // 2)
    left = np.where(first >= 0, np.floor(rng.random(n) * (ws - ww + 1)).astype(np.int64), (ws - ww) // 2)
    return np.stack([top, left, hh, ww], 1).astype(np.int32)


def erase_params(rng: np.random.Generator, n: int, h: int, w: int, scale=(0.5, 0.8), ratio=(0.8, 1.25)) -> np.ndarray:
    """RandomErasing.get_params (torchvision): ten draws of (area fraction, log aspect); the first box strictly smaller
    than the image in both directions is placed uniformly; none -> nothing is erased (h = w = 0).  int32 [n, 4]."""
    area = float(h * w)
    ea = area * rng.uniform(scale[0], scale[1], size=(n, 10))
    ar = np.exp(rng.uniform(math.log(ratio[0]), math.log(ratio[1]), size=(n, 10)))
    eh = np.rint(np.sqrt(ea * ar)).astype(np.int64)
    ew = np.rint(np.sqrt(ea / ar)).astype(np.int64)
    ok = (eh < h) & (ew < w)
    first = np.where(ok.any(1), ok.argmax(1), -1)
    rows = np.arange(n)
    hh = np.where(first >= 0, eh[rows, np.maximum(first, 0)], 0)
    ww = np.where(first >= 0, ew[rows, np.maximum(first, 0)], 0)
    top = np.floor(rng.random(n) * (h - hh + 1)).astype(np.int64)
    left = np.floor(rng.random(n) * (w - ww + 1)).astype(np.int64)
    return np.stack([top, left, hh, ww], 1).astype(np.int32)


def jitter_table(rng: np.random.Generator, n: int, brightness=0.4, contrast=0.4, saturation=0.4, hue=0.1, p: float = 0.8,
                 p_gray: float = 0.2, arithmetic: int = 0) -> np.ndarray:
    """RandomApply([ColorJitter(b, c, s, h)], p) + RandomGrayscale(p_gray) parameters for n samples (torchvision
    ColorJitter.get_params / albumentations ColorJitter.get_params: a random order of the four adjustments, factors ~
    U(1-x, 1+x), hue ~ U(-h, h)), as the int32 [n, 12] rows of cp2_color_ops: order[4] (-1 = not applied), float bits of the
    three factors, uint8(hue * 255), gray, the arithmetic the kernel applies them in (0: torchvision on PIL images -- the
    background views, main.py:212-216; 1: albumentations on cv2 -- the foreground views, main.py:236-237), float bits of
    the hue factor."""
    t = np.zeros((n, 12), dtype=np.int32)
    apply = rng.random(n) < p
    order = rng.permuted(np.tile(np.arange(4, dtype=np.int32), (n, 1)), axis=1)
    t[:, 0:4] = np.where(apply[:, None], order, -1)
    f = np.stack([rng.uniform(1 - x, 1 + x, n) for x in (brightness, contrast, saturation)], 1).astype(np.float32)
    t[:, 4:7] = f.view(np.int32)
    hf = rng.uniform(-hue, hue, n)
    t[:, 7] = hue_shift_u8(hf)
    t[:, 8] = rng.random(n) < p_gray
    t[:, 9] = arithmetic
    t[:, 10] = hf.astype(np.float32).view(np.int32)
    return t


def hue_shift_u8(hue_factor) -> np.ndarray:
    """np.uint8(hue_factor * 255) as torchvision's adjust_hue computes it: truncation toward zero, then wrap-around."""
    return (np.trunc(np.asarray(hue_factor, dtype=np.float64) * 255).astype(np.int64) % 256).astype(np.int32)


def gaussian_box(sigma: float):
    """Pillow's box approximation of GaussianBlur(sigma) (BoxBlur.c _gaussian_blur_radius with 3 passes, then
    ImagingHorizontalBoxBlur's weights), in the float32 arithmetic of the C source: (integer radius, ww, fw)."""
    f32 = np.float32
    sigma2 = f32(f32(f32(sigma) * f32(sigma)) / f32(3))
    box_len = f32(math.sqrt(12.0 * float(sigma2) + 1.0))
    l = f32(math.floor((float(box_len) - 1.0) / 2.0))
    a = f32(f32(f32(2) * l + f32(1)) * f32(f32(l * f32(l + f32(1))) - f32(f32(3) * sigma2)))
    a = f32(a / f32(f32(6) * f32(sigma2 - f32(f32(l + f32(1)) * f32(l + f32(1))))))
    radius = f32(l + a)
    r = int(radius)
    ww = int(f32(1 << 24) / f32(f32(radius * f32(2)) + f32(1)))
    fw = ((1 << 24) - (r * 2 + 1) * ww) // 2
    return r, ww, fw


def blur_table(rng: np.random.Generator, n: int, sigma=(0.1, 2.0), p: float = 0.5):
    """RandomApply([GaussianBlur(sigma)], p) for n samples: int32 [n, 4] rows of cp2_blur_to_tensor (on, r, ww, fw) and
    the largest integer radius among them."""
    t = np.zeros((n, 4), dtype=np.int32)
    on = rng.random(n) < p
    sig = rng.uniform(sigma[0], sigma[1], n)
    rmax = 0
    for i in np.nonzero(on)[0]:
        r, ww, fw = gaussian_box(float(sig[i]))
        t[i] = (1, r, ww, fw)
        rmax = max(rmax, r)
    return t, rmax


def crop_table(src_index: np.ndarray, boxes: np.ndarray, flips: np.ndarray) -> np.ndarray:
    """int32 [n, 8] parameter rows of cp2_crop_resize_flip."""
    t = np.zeros((len(src_index), 8), dtype=np.int32)
    t[:, 0], t[:, 1:5], t[:, 5] = src_index, boxes, flips
    return t


class EpochSampler:
    """DistributedSampler(shuffle=True, drop_last=True, seed) of reference main.py:263-272: one permutation of the
    dataset per epoch from torch.Generator(seed + epoch), truncated to a multiple of the world size, strided by rank."""

    def __init__(self, n: int, world: int, rank: int, seed: int):
        self.n, self.world, self.rank, self.seed = n, world, rank, seed
        self.per_rank = n // world

    def indices(self, epoch: int) -> np.ndarray:
        g = torch.Generator().manual_seed(self.seed + epoch)
        perm = torch.randperm(self.n, generator=g)[: self.per_rank * self.world]
        return perm[self.rank:: self.world].numpy()


class DeviceDataset:
    """Images (uint8 or fp32 [N,3,Hs,Ws]) and optional region-id maps ([N,Hs,Ws] int64) resident in device memory."""

    def __init__(self, images: torch.Tensor, region_ids: Optional[torch.Tensor] = None, device="cuda"):
        if images.dim() != 4 or images.shape[1] != 3 or images.dtype not in (torch.uint8, torch.float32):
            raise ValueError("DeviceDataset: images must be uint8 or float32 [N,3,H,W]")
        self.images = images.to(device).contiguous()
        self.region_ids = None if region_ids is None else region_ids.to(device=device, dtype=torch.int64).contiguous()
        if self.region_ids is not None and tuple(self.region_ids.shape) != (images.shape[0],) + tuple(images.shape[2:]):
            raise ValueError("DeviceDataset: region_ids must be [N,H,W]")

    def __len__(self):
        return self.images.shape[0]

    @classmethod
    def from_file(cls, path: str, device="cuda") -> "DeviceDataset":
        """A torch.save()d dict {'images': uint8/float32 [N,3,H,W], optional 'region_ids': [N,H,W]} or a bare tensor."""
        obj = torch.load(path, map_location="cpu")
        if isinstance(obj, torch.Tensor):
            return cls(obj, None, device)
        return cls(obj["images"], obj.get("region_ids"), device)


def make_step_batch(ds: DeviceDataset, fg_idx: np.ndarray, bg0_idx: np.ndarray, bg1_idx: np.ndarray, h: int, w: int,
                    rng: np.random.Generator, foreground_min: float = 0.5, foreground_max: float = 0.8,
                    id_stride: int = 1, use_regions: bool = True, photometric: Optional[bool] = None) -> Dict[str, torch.Tensor]:
    """One training batch with the keyword set of MODEL.forward (main.py:616-628), every tensor made on the device.
    photometric (default: on for a uint8 dataset): ColorJitter / grayscale / GaussianBlur as the reference's loaders
    apply them to every view; False = geometry only (crop, flip, erase) in fp32."""
    dev = ds.images.device
    n, hs, ws = len(fg_idx), ds.images.shape[2], ds.images.shape[3]
    if photometric is None:
        photometric = ds.images.dtype == torch.uint8
    if photometric and ds.images.dtype != torch.uint8:
        raise ValueError("photometric augmentation works on uint8 images (the reference's PIL / cv2 images are uint8)")
    out = {}
    tabs = []
    for idx in (fg_idx, fg_idx, bg0_idx, bg1_idx):            # query view, key view (same images), two backgrounds
        tabs.append(crop_table(idx, rrc_params(rng, n, hs, ws), rng.random(n) < 0.5))
    rects = np.concatenate([erase_params(rng, n, h, w, (foreground_min, foreground_max)) for _ in range(2)])
    reg = ds.region_ids if use_regions else None
    if not photometric:
        table = torch.from_numpy(np.concatenate(tabs)).to(dev, non_blocking=True)        # one H2D copy per step
        rects_d = torch.from_numpy(rects).to(dev, non_blocking=True)
        fg = ops.crop_resize_flip(ds.images, reg, table[: 2 * n], h, w, id_stride, want_ids=True)
        bg = ops.crop_resize_flip(ds.images, None, table[2 * n:], h, w, 1, want_ids=False)
        ops.erase_rect(bg[0], rects_d)
        imgs = torch.cat([fg[0], bg[0]])
    else:
        colour = jitter_table(rng, 4 * n)
        colour[: 2 * n, 9] = 1                              # foreground views: albumentations-on-cv2 arithmetic (loader.py:93-109)
        blur, rmax = blur_table(rng, 4 * n)
        rects4 = np.concatenate([np.zeros((2 * n, 4), dtype=np.int32), rects])           # foreground views: nothing erased
        parts = [np.concatenate(tabs), colour, blur, rects4]
        flat = torch.from_numpy(np.concatenate([p.ravel() for p in parts])).to(dev, non_blocking=True)   # one H2D copy per step
        views, o = [], 0
        for p in parts:
            views.append(flat[o:o + p.size].view(p.shape))
            o += p.size
        table, colour_d, blur_d, rects_d = views
        rgbx = torch.empty((4 * n, h, w), dtype=torch.int32, device=dev)
        fg = ops.crop_resize_flip(ds.images, reg, table[: 2 * n], h, w, id_stride, want_ids=True, want_f32=False,
                                  out_rgbx=rgbx[: 2 * n])
        ops.pil_resize_crop(ds.images, table[2 * n:], h, w, out_rgbx=rgbx[2 * n:])
        ops.color_ops(rgbx, colour_d)
        imgs = ops.blur_to_tensor(rgbx, blur_d, rects_d, rmax)
    out["img_a"], out["img_b"] = imgs[:n], imgs[n:2 * n]
    out["pixel_ids_a"], out["pixel_ids_b"] = fg[1][:n], fg[1][n:]
    out["region_ids_a"], out["region_ids_b"] = fg[2][:n], fg[2][n:]
    out["bg0"], out["bg1"] = imgs[2 * n:3 * n], imgs[3 * n:]
    return out
