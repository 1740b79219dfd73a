"""TEST INFRASTRUCTURE ONLY -- CPU restatement (numpy / torch-CPU) of the supervised CutPaste "mirror" pre-training
path, SURVEY 8f rank 4.  Imported by tests/ only; cp2_amd/ never imports it.

Parity pin: tests/golden/mirror_cutpaste.npz and mirror_loss.npz are outputs of the reference's own
`CutPasteDataset.__getitem__` / `MirrorModule.shared_step` (tests/golden/make_mirror_goldens.py);
tests/test_mirror.py checks this file against them bit for bit (images, masks) / to 1e-6 (losses, gradients), and the
rotation restatement against Pillow itself (third-party dependency of the reference, `Pillow` -- requirements.txt
pins no version; 12.2.0 is installed in this image) over random sizes and angles.

Reference lines restated:
  datasets/pretrain_dataset.py:273-352   CutPasteDataset.cutpaste (patch draw, rotate, paste, mask)
  datasets/pretrain_dataset.py:357-412   __getitem__ (class choice, additional patches, logical_or of masks, ToTensor)
  networks/mirror_network.py:40-63       shared_step: class cross entropy + compare cross entropy of tempered softmaxes
  networks/segment_network.py:220-231    forward: bilinear resize (align_corners=False) of the logits to the image size
Pillow algorithm restated (published source: src/PIL/Image.py `Image.rotate`, src/libImaging/Geometry.c
`affine_fixed`): reverse affine matrix from the angle (cos / sin rounded to 15 decimals), expanded canvas from the
transformed corners, nearest-neighbour sampling in 16.16 fixed point, FIX(v) = floor(v * 65536 + 0.5).
"""
import math

import numpy as np
import torch
import torch.nn.functional as F


# ------------------------------------------------------------------ Pillow's rotate(angle, NEAREST, expand=True)
def rotate_geometry(w, h, angle):
    """(nw, nh, [a0..a5]) with source pixel of output (u, v) = ((a2 + u*a0 + v*a1) >> 16, (a5 + u*a3 + v*a4) >> 16)."""
    angle = angle % 360.0
    one, half = 65536, 32768
    if angle == 0:
        return w, h, [one, 0, half, 0, one, half]
    if angle == 180:                                   # Transpose.ROTATE_180
        return w, h, [-one, 0, (w - 1) * one + half, 0, -one, (h - 1) * one + half]
    if angle == 90:                                    # Transpose.ROTATE_90: out[Y][X] = in[X][w-1-Y]
        return h, w, [0, -one, (w - 1) * one + half, one, 0, half]
    if angle == 270:                                   # Transpose.ROTATE_270: out[Y][X] = in[h-1-X][Y]
        return h, w, [0, one, half, -one, 0, (h - 1) * one + half]
    cx, cy = w / 2, h / 2
    r = -math.radians(angle)
    m = [round(math.cos(r), 15), round(math.sin(r), 15), 0.0, round(-math.sin(r), 15), round(math.cos(r), 15), 0.0]

    def tf(x, y):
        return m[0] * x + m[1] * y + m[2], m[3] * x + m[4] * y + m[5]
    m[2], m[5] = tf(-cx, -cy)
    m[2] += cx
    m[5] += cy
    xs, ys = zip(*[tf(x, y) for x, y in ((0, 0), (w, 0), (w, h), (0, h))])
    nw = math.ceil(max(xs)) - math.floor(min(xs))
    nh = math.ceil(max(ys)) - math.floor(min(ys))
    m[2], m[5] = tf(-(nw - w) / 2.0, -(nh - h) / 2.0)

    def fix(v):
        return int(math.floor(v * 65536.0 + 0.5))
    return nw, nh, [fix(m[0]), fix(m[1]), fix(m[2] + m[0] * 0.5 + m[1] * 0.5),
                    fix(m[3]), fix(m[4]), fix(m[5] + m[3] * 0.5 + m[4] * 0.5)]


def rotate_nearest_expand(patch, angle):
    """patch: uint8 [h, w, ch] -> (rotated [nh, nw, ch] with zero fill, valid [nh, nw] bool)."""
    h, w = patch.shape[:2]
    nw, nh, a = rotate_geometry(w, h, angle)
    v, u = np.meshgrid(np.arange(nh, dtype=np.int64), np.arange(nw, dtype=np.int64), indexing="ij")
    xin = (a[2] + u * a[0] + v * a[1]) >> 16
    yin = (a[5] + u * a[3] + v * a[4]) >> 16
    valid = (xin >= 0) & (xin < w) & (yin >= 0) & (yin < h)
    out = np.zeros((nh, nw) + patch.shape[2:], dtype=patch.dtype)
    out[valid] = patch[yin[valid], xin[valid]]
    return out, valid


# ------------------------------------------------------------------ datasets/pretrain_dataset.py:273-352
def draw_patch(img_h, img_w, patch_type, min_area_scale, max_area_scale, min_aspect_ratio, max_aspect_ratio,
               min_rotation, max_rotation, rng=np.random):
    """The random draws of one cutpaste() call in the reference's order -> dict of integers + the angle."""
    if patch_type == 1:                                # REGULAR
        area_scale = rng.uniform(high=max_area_scale, low=min_area_scale)
        aspect = rng.uniform(high=max_aspect_ratio, low=min_aspect_ratio)
        rotation = 0
    else:                                              # SCAR
        area_scale = rng.uniform(high=max_area_scale * 0.5, low=min_area_scale)
        aspect = rng.uniform(3, 6)
        rotation = rng.uniform(low=min_rotation, high=max_rotation)
    area = int(img_h * img_w * area_scale)
    ph = int(np.sqrt(area / aspect))
    pw = int(ph * aspect)
    px = rng.randint(0, img_w - pw)
    py = rng.randint(0, img_h - ph)
    rw, rh, _ = rotate_geometry(pw, ph, rotation)
    x_pos = rng.randint(0, img_w - rw)
    y_pos = rng.randint(0, img_h - rh)
    return dict(px=px, py=py, pw=pw, ph=ph, rotation=rotation, x_pos=x_pos, y_pos=y_pos, cls=patch_type)


def cutpaste(image, mirror_image, p):
    """image / mirror_image: uint8 [H, W, 3] (mirror may be None); p: draw_patch() result.
    -> (image', mirror', mask int64 [H, W] with the patch class inside the pasted shape)."""
    H, W = image.shape[:2]
    patch = image[p["py"]:p["py"] + p["ph"], p["px"]:p["px"] + p["pw"]]
    rot, valid = rotate_nearest_expand(patch, p["rotation"])
    rh, rw = valid.shape
    ys, xs = slice(p["y_pos"], p["y_pos"] + rh), slice(p["x_pos"], p["x_pos"] + rw)
    out = image.copy()
    out[ys, xs][valid] = rot[valid]
    mir = None
    if mirror_image is not None:
        mir = mirror_image.copy()
        mir[ys, xs][valid] = rot[valid]
    mask = np.zeros((H, W), dtype=np.int64)
    mask[ys, xs] = valid.astype(np.int64) * p["cls"]
    return out, mir, mask


def cutpaste_item(image, mirror_image, cls, max_num_patches, cfg, rng=np.random):
    """__getitem__ after the images are loaded (datasets/pretrain_dataset.py:381-412): class 0 = untouched, else one
    patch + randint(max_num_patches) more (masks OR-ed to 0/1).  Returns float CHW images (ToTensor) + int64 mask."""
    H, W = image.shape[:2]
    if cls == 0:
        img, mir, mask = image, mirror_image, np.zeros((H, W), dtype=np.int64)
    else:
        img, mir, mask = cutpaste(image, mirror_image, draw_patch(H, W, cls, rng=rng, **cfg))
        for _ in range(rng.randint(max_num_patches)):
            old = mask
            img, mir, mask = cutpaste(img, mir, draw_patch(H, W, cls, rng=rng, **cfg))
            mask = np.logical_or(mask, old).astype(np.int64)

    def to_tensor(a):
        return None if a is None else (a.astype(np.float32).transpose(2, 0, 1) / np.float32(255))
    return to_tensor(img), to_tensor(mir), mask


# ------------------------------------------------------------------ networks/mirror_network.py:40-63
def resize_logits(logits, size):
    """segment_network.py:220-231 (mmseg resize = F.interpolate bilinear, align_corners=False)."""
    return F.interpolate(logits, size=size, mode="bilinear", align_corners=False)


def mirror_losses(s_logits, t_logits, masks, softmax_temp, lmbd_compare_loss):
    """s_logits / t_logits: [N, C, H, W] at image size (t_logits None = MirrorVariant.NONE); masks int64 [N, H, W].
    -> dict(loss, class_loss, compare_loss, argmax)."""
    if t_logits is not None:
        all_logits = torch.cat([s_logits, t_logits])
        all_masks = torch.cat([masks, masks])
        compare = F.cross_entropy(torch.softmax(s_logits / softmax_temp, 1), torch.softmax(t_logits / softmax_temp, 1))
    else:
        all_logits, all_masks = s_logits, masks
        compare = torch.zeros((), dtype=s_logits.dtype)
    cls = F.cross_entropy(all_logits, all_masks)
    return dict(loss=cls + lmbd_compare_loss * compare, class_loss=cls, compare_loss=compare,
                argmax=all_logits.argmax(dim=1))


def confusion(argmax, masks, num_classes):
    """[C, C] int64 counts, row = ground truth, column = prediction (what the torchmetrics collection of
    segment_network.py:176-214 is computed from; torchmetrics itself is not installed: metrics unpinned)."""
    idx = masks.reshape(-1) * num_classes + argmax.reshape(-1)
    return torch.bincount(idx, minlength=num_classes * num_classes).reshape(num_classes, num_classes)
