#!/usr/bin/env python3
"""Golden vectors for the photometric half of the input pipeline, made by **Pillow itself**.

The reference's background views are torchvision transforms on PIL images (main.py:204-225) and both kinds of view are
blurred with PIL's ImageFilter.GaussianBlur (loader.py:121-152).  torchvision is not installed in the build image, but its
PIL code path is a thin wrapper whose calls are restated here one for one (torchvision/transforms/_functional_pil.py):

    F.resized_crop(img, i, j, h, w, size, BILINEAR)  ->  img.crop((j, i, j + w, i + h)).resize(size[::-1], Image.BILINEAR)
    F.adjust_brightness / contrast / saturation      ->  ImageEnhance.Brightness / Contrast / Color(img).enhance(f)
    F.adjust_hue(img, f)                             ->  h, s, v = img.convert("HSV").split(); h += uint8(f * 255); merge; convert("RGB")
    F.rgb_to_grayscale(img, 3)                       ->  np.dstack([img.convert("L")] * 3)
    loader.GaussianBlur                              ->  img.filter(ImageFilter.GaussianBlur(radius=sigma))
    F.hflip                                          ->  img.transpose(Image.FLIP_LEFT_RIGHT)
    ToTensor                                         ->  uint8 HWC -> float32 CHW / 255
    RandomErasing(value=0)                           ->  img[:, i:i+h, j:j+w] = 0

Run in the build container (Pillow 12.2.0):  python tests/golden/make_augment_goldens.py
Writes tests/golden/augment_pillow.npz: inputs, explicit parameters, and Pillow's outputs -- data only.
"""
import os
import sys

import numpy as np
from PIL import Image, ImageEnhance, ImageFilter

HERE = os.path.dirname(os.path.abspath(__file__))


def pil_hue(img, f):
    h, s, v = img.convert("HSV").split()
    nh = np.array(h, dtype=np.uint8)
    with np.errstate(over="ignore"):
        nh += np.array(int(f * 255) % 256).astype(np.uint8)          # np.uint8(f * 255): truncation, wrap-around
    return Image.merge("HSV", (Image.fromarray(nh, "L"), s, v)).convert("RGB")


JITTER = [lambda im, f: ImageEnhance.Brightness(im).enhance(f), lambda im, f: ImageEnhance.Contrast(im).enhance(f),
          lambda im, f: ImageEnhance.Color(im).enhance(f), pil_hue]


def background_view(img, box, h, w, order, factors, gray, sigma, flip, rect):
    top, left, ch, cw = [int(v) for v in box]
    im = Image.fromarray(img).crop((left, top, left + cw, top + ch)).resize((w, h), Image.BILINEAR)
    for k in order:
        if k >= 0:
            im = JITTER[int(k)](im, float(factors[int(k)]))
    if gray:
        im = Image.fromarray(np.dstack([np.asarray(im.convert("L"))] * 3))
    if sigma > 0:
        im = im.filter(ImageFilter.GaussianBlur(radius=float(sigma)))
    if flip:
        im = im.transpose(Image.FLIP_LEFT_RIGHT)
    out = (np.asarray(im).astype(np.float32) / np.float32(255.0)).transpose(2, 0, 1).copy()
    t, l, eh, ew = [int(v) for v in rect]
    out[:, t:t + eh, l:l + ew] = 0.0
    return out


def main():
    rng = np.random.default_rng(20261004)
    rec = {}
    n, hs, ws, h, w = 10, 56, 72, 40, 48
    noise = rng.integers(0, 256, (n, hs, ws, 3), dtype=np.uint8)
    smooth = np.stack([np.asarray(Image.fromarray(x).filter(ImageFilter.GaussianBlur(2.5))) for x in noise[: n // 2]])
    src = np.concatenate([noise[n // 2:], smooth])                 # half white noise, half smooth images
    rec["src"] = src
    # ---- single operations
    boxes = np.array([[0, 0, hs, ws], [3, 5, 50, 60], [10, 20, 13, 17], [0, 0, 40, 72], [2, 0, 54, 48], [5, 5, 1, 1],
                      [0, 0, 56, 20], [20, 30, 30, 40], [1, 1, 55, 71], [8, 8, 40, 48]], dtype=np.int32)
    rec["boxes"] = boxes
    rec["resized"] = np.stack([np.asarray(Image.fromarray(src[i]).crop((b[1], b[0], b[1] + b[3], b[0] + b[2])).resize((w, h), Image.BILINEAR))
                               for i, b in enumerate(boxes)])
    f = rng.uniform(0.6, 1.4, (n, 3))
    f[0] = (1.0, 1.0, 1.0)
    f[1] = (0.6, 1.4, 0.6)
    hue = rng.uniform(-0.1, 0.1, n)
    hue[0], hue[1] = 0.0, -0.1
    rec["factors"], rec["hue"] = f, hue
    rec["brightness"] = np.stack([np.asarray(JITTER[0](Image.fromarray(src[i]), f[i, 0])) for i in range(n)])
    rec["contrast"] = np.stack([np.asarray(JITTER[1](Image.fromarray(src[i]), f[i, 1])) for i in range(n)])
    rec["saturation"] = np.stack([np.asarray(JITTER[2](Image.fromarray(src[i]), f[i, 2])) for i in range(n)])
    rec["hue_out"] = np.stack([np.asarray(pil_hue(Image.fromarray(src[i]), hue[i])) for i in range(n)])
    rec["gray"] = np.stack([np.asarray(Image.fromarray(src[i]).convert("L")) for i in range(n)])
    sig = rng.uniform(0.1, 2.0, n)
    sig[0], sig[1], sig[2] = 0.1, 2.0, 3.3                         # the transform's range ends, and an integer radius of 2
    rec["sigma"] = sig
    rec["blurred"] = np.stack([np.asarray(Image.fromarray(src[i]).filter(ImageFilter.GaussianBlur(radius=float(sig[i])))) for i in range(n)])
    # ---- whole background views with explicit parameters
    m = 16
    order = np.stack([rng.permutation(4) for _ in range(m)]).astype(np.int32)
    order[rng.random(m) < 0.25] = -1                               # RandomApply(p = 0.8) did not fire
    order[0] = (1, 0, 2, 3)                                        # contrast first
    order[1] = (3, 2, 0, 1)                                        # contrast last
    vf = np.concatenate([rng.uniform(0.6, 1.4, (m, 3)), rng.uniform(-0.1, 0.1, (m, 1))], 1)
    gray = rng.random(m) < 0.3
    vs = np.where(rng.random(m) < 0.6, rng.uniform(0.1, 2.0, m), 0.0)
    flip = rng.random(m) < 0.5
    idx = rng.integers(0, n, m)
    vbox = np.stack([boxes[rng.integers(0, len(boxes))] for _ in range(m)])
    rect = np.stack([(rng.integers(0, 10), rng.integers(0, 10), rng.integers(0, 30), rng.integers(0, 38)) for _ in range(m)]).astype(np.int32)
    rect[3] = 0
    rec.update(view_idx=idx, view_box=vbox, view_order=order, view_factors=vf, view_gray=gray, view_sigma=vs, view_flip=flip,
               view_rect=rect)
    rec["views"] = np.stack([background_view(src[idx[i]], vbox[i], h, w, order[i], vf[i], gray[i], vs[i], flip[i], rect[i]) for i in range(m)])
    out = os.path.join(HERE, "augment_pillow.npz")
    np.savez_compressed(out, **rec)
    print("wrote", out, os.path.getsize(out), "bytes; Pillow", Image.__version__ if hasattr(Image, "__version__") else "")


if __name__ == "__main__":
    sys.exit(main())
