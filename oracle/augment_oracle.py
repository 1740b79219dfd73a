"""CPU restatement of the photometric / resampling half of the reference's input pipeline -- TEST INFRASTRUCTURE ONLY.

Only tests/ and tests/golden/make_augment_goldens.py import this module; the product path (cp2_amd/augment.py,
csrc/photometric.hip) never does.

What the reference runs (main.py:204-245, loader.py:121-152) is torchvision / albumentations code on top of two
third-party image libraries that are not part of /root/reference:

  * background views: torchvision transforms on PIL images -- RandomResizedCrop = `img.crop(box).resize(size, BILINEAR)`,
    ColorJitter = `ImageEnhance.{Brightness,Contrast,Color}(img).enhance(f)` and the HSV round trip of `adjust_hue`,
    RandomGrayscale = `img.convert("L")` replicated to three bands, GaussianBlur = `img.filter(ImageFilter.GaussianBlur(s))`
    (loader.py:121-131), ToTensor = uint8 / 255.  The arithmetic lives in **Pillow** (un-pinned in requirements.txt;
    12.2.0 is installed in the build image): Resample.c (fixed-point separable convolution, 22-bit coefficients, uint8
    after each pass), Blend.c (float interpolation, truncation), Convert.c (rgb2l, rgb2hsv / hsv2rgb after colorsys),
    BoxBlur.c (Gaussian = three box passes per axis, 24-bit weights).  This file restates those published algorithms in
    numpy, and tests/test_augment_photometric.py / make_augment_goldens.py pin the restatement against Pillow itself.
  * foreground views: albumentations on cv2 (RandomResizedCrop = cv2.resize INTER_LINEAR, ColorJitter / ToGray by cv2
    colour conversions, LUTs and addWeighted) -- cv2 and albumentations are absent here, so that arithmetic is restated
    from the dependencies' published sources (cv2_* / albu_* below) and stays **parity-unpinned**; only the foreground's
    Gaussian blur is Pillow's (loader.py:136-152, the same ImageFilter.GaussianBlur).
"""
from __future__ import annotations

import math

import numpy as np

PRECISION_BITS = 32 - 8 - 2          # Pillow Resample.c: coefficients as 22-bit fixed point


# --------------------------------------------------------------------------- Resample.c: bilinear, uint8
def resample_coeffs(in_size: int, in0: float, in1: float, out_size: int):
    """precompute_coeffs + normalize_coeffs_8bpc for the BILINEAR filter (support 1): per output index the first input
    index, the tap count, and the integer taps.  Down-scaling widens the triangle by the scale (antialiasing)."""
    scale = (in1 - in0) / out_size
    filterscale = max(scale, 1.0)
    support = 1.0 * filterscale
    ksize = int(math.ceil(support)) * 2 + 1
    ss = 1.0 / filterscale
    bounds = np.zeros((out_size, 2), dtype=np.int32)
    kk = np.zeros((out_size, ksize), dtype=np.int32)
    for xx in range(out_size):
        center = in0 + (xx + 0.5) * scale
        xmin = int(center - support + 0.5)
        xmin = max(xmin, 0)
        xmax = int(center + support + 0.5)
        xmax = min(xmax, in_size) - xmin
        k = np.zeros(ksize, dtype=np.float64)
        ww = 0.0
        for x in range(xmax):
            v = (x + xmin - center + 0.5) * ss
            v = -v if v < 0 else v
            w = 1.0 - v if v < 1.0 else 0.0
            k[x] = w
            ww += w
        if ww != 0.0:
            k[:xmax] = k[:xmax] / ww
        bounds[xx] = (xmin, xmax)
        for x in range(ksize):                       # (int)(k * (1 << 22) +- 0.5): round half away from zero
            v = k[x] * (1 << PRECISION_BITS)
            kk[xx, x] = int(v - 0.5) if k[x] < 0 else int(v + 0.5)
    return bounds, kk


def _resample_axis0(img: np.ndarray, out_size: int) -> np.ndarray:
    """One pass along axis 0 of a uint8 array [n, ...] (the whole extent is the box)."""
    bounds, kk = resample_coeffs(img.shape[0], 0.0, float(img.shape[0]), out_size)
    out = np.empty((out_size,) + img.shape[1:], dtype=np.uint8)
    src = img.astype(np.int64)
    for i in range(out_size):
        xmin, n = bounds[i]
        acc = np.full(img.shape[1:], 1 << (PRECISION_BITS - 1), dtype=np.int64)
        for t in range(n):
            acc += src[xmin + t] * int(kk[i, t])
        out[i] = np.clip(acc >> PRECISION_BITS, 0, 255).astype(np.uint8)
    return out


def pil_crop_resize(img: np.ndarray, box, h: int, w: int) -> np.ndarray:
    """torchvision F.resized_crop on a PIL image = img.crop((left, top, left+cw, top+ch)).resize((w, h), BILINEAR).
    img: uint8 [Hs, Ws, 3]; box = (top, left, ch, cw).  Pillow runs the horizontal pass first and rounds to uint8 between
    the passes; a pass whose size does not change is skipped."""
    top, left, ch, cw = [int(v) for v in box]
    cur = img[top:top + ch, left:left + cw]
    if cw != w:
        cur = np.swapaxes(_resample_axis0(np.swapaxes(cur, 0, 1), w), 0, 1)
    if ch != h:
        cur = _resample_axis0(cur, h)
    return np.ascontiguousarray(cur)


# --------------------------------------------------------------------------- Convert.c
# ---- foreground views: cv2.resize(crop, (w, h), interpolation=cv2.INTER_LINEAR) on uint8   (reference loader.py:93-109:
# albumentations' RandomResizedCrop calls it).  cv2 is a third-party dependency, absent from /root/reference, un-pinned in
# its requirements.txt and NOT installed here: the arithmetic below is restated from OpenCV's published source
# (modules/imgproc/src/resize.cpp, 4.x: resizeGeneric_ / HResizeLinear / VResizeLinear<uchar, int, short, ...>) and is
# PARITY-UNPINNED -- no fixture and no live library can confirm it in this container:
#   * destination pixel centre -> source coordinate (d + 0.5) * (src / dst) - 0.5 (evaluated in double, stored as float),
#     left tap floor(f), fraction f - floor(f); a tap left of the image becomes tap 0 with fraction 0, a tap at or beyond
#     the last pixel becomes the last pixel with fraction 0;
#   * weights as 11-bit fixed point: saturate_cast<short>(w * 2048) = round-half-to-even of the float product;
#   * horizontal pass in int32: S[x0] * a0 + S[x1] * a1;
#   * vertical pass: (((b0 * (r0 >> 4)) >> 16) + ((b1 * (r1 >> 4)) >> 16) + 2) >> 2;
#   * an exact 2 x 2 shrink is served by INTER_AREA's fast path instead: (a + b + c + d + 2) >> 2.
def _cv2_linear_taps(src_size: int, dst_size: int):
    scale = float(src_size) / float(dst_size)
    d = np.arange(dst_size, dtype=np.float64)
    f = ((d + 0.5) * scale - 0.5).astype(np.float32)
    s = np.floor(f).astype(np.int64)
    f = (f - s.astype(np.float32)).astype(np.float32)
    low = s < 0
    s[low], f[low] = 0, 0.0
    high = s >= src_size - 1
    s[high], f[high] = src_size - 1, 0.0
    a0 = np.rint((np.float32(1.0) - f) * np.float32(2048.0)).astype(np.int64)      # cvRound: half to even
    a1 = np.rint(f * np.float32(2048.0)).astype(np.int64)
    return s, np.minimum(s + 1, src_size - 1), a0, a1


def cv2_resize_linear_u8(img: np.ndarray, h: int, w: int) -> np.ndarray:
    """img: (H, W, C) uint8 -> (h, w, C) uint8 as cv2.resize(img, (w, h), interpolation=cv2.INTER_LINEAR) (see above)."""
    H, W = img.shape[:2]
    src = img.astype(np.int64)
    if H == 2 * h and W == 2 * w:
        return ((src[0::2, 0::2] + src[0::2, 1::2] + src[1::2, 0::2] + src[1::2, 1::2] + 2) >> 2).astype(np.uint8)
    x0, x1, a0, a1 = _cv2_linear_taps(W, w)
    y0, y1, b0, b1 = _cv2_linear_taps(H, h)
    rows = src[:, x0] * a0[None, :, None] + src[:, x1] * a1[None, :, None]            # (H, w, C) int
    r0, r1 = rows[y0], rows[y1]
    out = (((b0[:, None, None] * (r0 >> 4)) >> 16) + ((b1[:, None, None] * (r1 >> 4)) >> 16) + 2) >> 2
    return out.astype(np.uint8)


def foreground_crop_u8(src: np.ndarray, box, flip: bool, h: int, w: int) -> np.ndarray:
    """src: (3, Hs, Ws) uint8; A.RandomResizedCrop (crop, cv2.resize INTER_LINEAR) then A.HorizontalFlip -> (h, w, 3) uint8."""
    top, left, ch, cw = [int(v) for v in box]
    out = cv2_resize_linear_u8(np.ascontiguousarray(src[:, top:top + ch, left:left + cw].transpose(1, 2, 0)), h, w)
    return out[:, ::-1] if flip else out


def rgb_to_l(img: np.ndarray) -> np.ndarray:
    """L24 / rgb2l: (R*19595 + G*38470 + B*7471 + 0x8000) >> 16."""
    r, g, b = (img[..., c].astype(np.int64) for c in range(3))
    return ((r * 19595 + g * 38470 + b * 7471 + 0x8000) >> 16).astype(np.uint8)


def rgb_to_hsv(img: np.ndarray) -> np.ndarray:
    """rgb2hsv_row (after colorsys.rgb_to_hsv): float arithmetic, hue and saturation truncated to uint8."""
    f32 = np.float32
    r, g, b = (img[..., c].astype(np.int32) for c in range(3))
    maxc, minc = np.maximum(r, np.maximum(g, b)), np.minimum(r, np.minimum(g, b))
    grey = maxc == minc
    cr = np.where(grey, 1, maxc - minc).astype(f32)
    mx = np.where(maxc == 0, 1, maxc).astype(f32)
    s = cr / mx
    rc, gc, bc = (maxc - r).astype(f32) / cr, (maxc - g).astype(f32) / cr, (maxc - b).astype(f32) / cr
    # the constants 2.0 / 4.0 / 6.0 / 1.0 are doubles in the C source: these sums run in double and land in a float
    d = np.float64
    h = np.where(r == maxc, (bc - gc).astype(d), np.where(g == maxc, 2.0 + rc.astype(d) - bc.astype(d), 4.0 + gc.astype(d) - rc.astype(d)))
    h = h.astype(f32)
    h = np.fmod(h.astype(d) / 6.0 + 1.0, 1.0).astype(f32)
    uh = np.clip((h.astype(d) * 255.0).astype(np.int64), 0, 255)
    us = np.clip((s.astype(d) * 255.0).astype(np.int64), 0, 255)
    out = np.stack([np.where(grey, 0, uh), np.where(grey, 0, us), maxc], -1)
    return out.astype(np.uint8)


def _round_half_away(x: np.ndarray) -> np.ndarray:
    return np.where(x >= 0, np.floor(x + 0.5), np.ceil(x - 0.5)).astype(np.int64)


def hsv_to_rgb(img: np.ndarray) -> np.ndarray:
    """hsv2rgb (after colorsys.hsv_to_rgb): sector from floor(h*6/255), p / q / t by C round()."""
    d, f32 = np.float64, np.float32
    h, s, v = (img[..., c].astype(np.int32) for c in range(3))
    hf = h.astype(f32).astype(d) * 6.0 / 255.0
    i = np.floor(hf).astype(np.int64)
    f = (hf - i.astype(f32).astype(d)).astype(f32)
    fs = (s.astype(f32).astype(d) / 255.0).astype(f32)
    vf = v.astype(f32).astype(d)
    p = np.clip(_round_half_away(vf * (1.0 - fs.astype(d))), 0, 255)
    q = np.clip(_round_half_away(vf * (1.0 - fs.astype(d) * f.astype(d))), 0, 255)
    t = np.clip(_round_half_away(vf * (1.0 - fs.astype(d) * (1.0 - f.astype(d)))), 0, 255)
    sect = i % 6
    table = [(v, t, p), (q, v, p), (p, v, t), (p, q, v), (t, p, v), (v, p, q)]
    out = np.zeros(img.shape, dtype=np.int64)
    for k, (rr, gg, bb) in enumerate(table):
        m = sect == k
        out[..., 0] = np.where(m, rr, out[..., 0])
        out[..., 1] = np.where(m, gg, out[..., 1])
        out[..., 2] = np.where(m, bb, out[..., 2])
    grey = s == 0
    for c in range(3):
        out[..., c] = np.where(grey, v, out[..., c])
    return out.astype(np.uint8)


# --------------------------------------------------------------------------- Blend.c + ImageEnhance
def blend(deg: np.ndarray, img: np.ndarray, factor: float) -> np.ndarray:
    """Image.blend(degenerate, image, factor): in1 + alpha * (in2 - in1) in float, truncated; clipped when alpha is
    outside [0, 1]."""
    f32 = np.float32
    a = f32(factor)
    v = (deg.astype(np.int32).astype(f32) + (a * (img.astype(np.int32) - deg.astype(np.int32)).astype(f32)).astype(f32)).astype(f32)
    if 0.0 <= float(a) <= 1.0:
        return v.astype(np.int64).astype(np.uint8)
    return np.where(v <= 0.0, 0, np.where(v >= 255.0, 255, v.astype(np.int64))).astype(np.uint8)


def adjust_brightness(img: np.ndarray, factor: float) -> np.ndarray:
    return blend(np.zeros_like(img), img, factor)


def contrast_mean(img: np.ndarray) -> int:
    """int(ImageStat.Stat(img.convert("L")).mean[0] + 0.5)"""
    l = rgb_to_l(img)
    return int(float(l.astype(np.int64).sum()) / float(l.size) + 0.5)


def adjust_contrast(img: np.ndarray, factor: float) -> np.ndarray:
    return blend(np.full_like(img, contrast_mean(img)), img, factor)


def adjust_saturation(img: np.ndarray, factor: float) -> np.ndarray:
    return blend(np.repeat(rgb_to_l(img)[..., None], 3, -1), img, factor)


def adjust_hue_shift(img: np.ndarray, shift: int) -> np.ndarray:
    """HSV round trip with the hue band advanced by `shift` (uint8 arithmetic, wrap-around)."""
    hsv = rgb_to_hsv(img)
    hsv[..., 0] = ((hsv[..., 0].astype(np.int64) + int(shift)) % 256).astype(np.uint8)
    return hsv_to_rgb(hsv)


def adjust_hue(img: np.ndarray, factor: float) -> np.ndarray:
    """torchvision F_pil.adjust_hue: HSV, h += np.uint8(factor * 255) (truncation toward zero, wrap-around), back to RGB."""
    return adjust_hue_shift(img, int(factor * 255) % 256)


def to_grayscale3(img: np.ndarray) -> np.ndarray:
    """RandomGrayscale: img.convert("L") replicated to three bands."""
    return np.repeat(rgb_to_l(img)[..., None], 3, -1)


JITTER_OPS = (adjust_brightness, adjust_contrast, adjust_saturation, adjust_hue)     # torchvision's fn_id order


def color_jitter(img: np.ndarray, order, factors) -> np.ndarray:
    """ColorJitter.forward: the four adjustments in the drawn order; factors[k] belongs to JITTER_OPS[k]."""
    for k in order:
        img = JITTER_OPS[int(k)](img, float(factors[int(k)]))
    return img


# --------------------------------------------------------------------------- foreground views: albumentations on cv2
# A.ColorJitter(0.4, 0.4, 0.4, 0.1, p=0.8) and A.ToGray(p=0.2) of main.py:236-237 on uint8 RGB images.  albumentations
# (functional.py: adjust_{brightness,contrast,saturation,hue}_torchvision, their *_uint8 helpers, to_gray) and OpenCV 4.x
# (imgproc color_rgb.simd.hpp RGB2Gray<uchar>, color_hsv.simd.hpp RGB2HSV_b / HSV2RGB_b / HSV2RGB_native, core arithm
# addWeighted, LUT) are absent from /root/reference and from this image: restated from the dependencies' published
# sources, PARITY-UNPINNED.  What is restated:
#   * brightness:  lut[i] = uint8(clip(i * f, 0, 255))  (astype truncates), cv2.LUT;
#   * contrast:    mean = float64 mean of cv2 RGB2GRAY(img); lut[i] = uint8(clip(i * f + mean * (1 - f), 0, 255)), cv2.LUT;
#   * saturation:  gray = RGB2GRAY -> GRAY2RGB; cv2.addWeighted(img, f, gray, 1 - f, 0): float32 a * alpha + b * beta + gamma,
#                  saturate_cast<uchar> = round half to even, clamped;
#   * hue:         RGB2HSV (8 bit: H in [0, 180)), H through lut[i] = uint8(mod(i + 180 f, 180)), HSV2RGB;
#   * RGB2GRAY 8 bit: (R * 9798 + G * 19235 + B * 3735 + 2^14) >> 15;
#   * RGB2HSV 8 bit: 12-bit fixed point with the two division tables (255 << 12) / v and (180 << 12) / (6 diff), rounded
#     half to even; HSV2RGB 8 bit: float32 sector arithmetic of HSV2RGB_native on (H, S / 255, V / 255), * 255, rounded.
def cv2_rgb2gray_u8(img: np.ndarray) -> np.ndarray:
    """(..., 3) uint8 RGB -> (...) uint8 as cv2.cvtColor(img, cv2.COLOR_RGB2GRAY)."""
    v = img.astype(np.int64)
    return ((v[..., 0] * 9798 + v[..., 1] * 19235 + v[..., 2] * 3735 + (1 << 14)) >> 15).astype(np.uint8)


_HSV_SHIFT = 12
_SDIV = np.zeros(256, dtype=np.int64)
_HDIV180 = np.zeros(256, dtype=np.int64)
_SDIV[1:] = np.rint((255 << _HSV_SHIFT) / (1.0 * np.arange(1, 256))).astype(np.int64)       # saturate_cast<int>(double): half to even
_HDIV180[1:] = np.rint((180 << _HSV_SHIFT) / (6.0 * np.arange(1, 256))).astype(np.int64)


def cv2_rgb2hsv_u8(img: np.ndarray) -> np.ndarray:
    """(..., 3) uint8 RGB -> (..., 3) uint8 (H in [0, 180), S, V) as cv2.cvtColor(img, cv2.COLOR_RGB2HSV) (RGB2HSV_b)."""
    v3 = img.astype(np.int64)
    r, g, b = v3[..., 0], v3[..., 1], v3[..., 2]
    v = np.maximum(np.maximum(r, g), b)
    diff = v - np.minimum(np.minimum(r, g), b)
    vr, vg = v == r, v == g
    s = (diff * _SDIV[v] + (1 << (_HSV_SHIFT - 1))) >> _HSV_SHIFT
    h = np.where(vr, g - b, np.where(vg, b - r + 2 * diff, r - g + 4 * diff))
    h = (h * _HDIV180[diff] + (1 << (_HSV_SHIFT - 1))) >> _HSV_SHIFT                        # arithmetic shift (floor), as in C
    h = h + np.where(h < 0, 180, 0)
    return np.stack([np.clip(h, 0, 255), s, v], -1).astype(np.uint8)


_SECTOR = np.array([[1, 3, 0], [1, 0, 2], [3, 0, 1], [0, 2, 1], [0, 1, 3], [2, 1, 0]])      # tab index of (b, g, r) per sector


def cv2_hsv2rgb_u8(img: np.ndarray) -> np.ndarray:
    """(..., 3) uint8 (H, S, V) -> (..., 3) uint8 RGB as cv2.cvtColor(img, cv2.COLOR_HSV2RGB) (HSV2RGB_b, hrange 180)."""
    f32 = np.float32
    h = img[..., 0].astype(f32)
    s = img[..., 1].astype(f32) * f32(1.0 / 255.0)
    v = img[..., 2].astype(f32) * f32(1.0 / 255.0)
    h = np.fmod(h * f32(6.0 / 180.0), f32(6.0)).astype(f32)
    sector = np.floor(h).astype(np.int64)
    h = (h - sector.astype(f32)).astype(f32)
    bad = (sector < 0) | (sector >= 6)
    sector = np.where(bad, 0, sector)
    h = np.where(bad, f32(0), h).astype(f32)
    one = f32(1.0)
    tab = np.stack([v, v * (one - s), v * (one - s * h), v * (one - s * (one - h))], -1).astype(f32)
    idx = _SECTOR[sector]                                                                   # (..., 3): b, g, r
    bgr = np.take_along_axis(tab, idx, -1)
    bgr = np.where((s == 0)[..., None], v[..., None], bgr).astype(f32)
    out = np.clip(np.rint(bgr * f32(255.0)), 0, 255).astype(np.uint8)                       # saturate_cast<uchar>(float): half to even
    return out[..., ::-1]                                                                   # -> r, g, b


def cv2_add_weighted_u8(a: np.ndarray, alpha: float, b: np.ndarray, beta: float, gamma: float = 0.0) -> np.ndarray:
    f32 = np.float32
    t = a.astype(f32) * f32(alpha) + b.astype(f32) * f32(beta) + f32(gamma)
    return np.clip(np.rint(t), 0, 255).astype(np.uint8)


def _lut_u8(values: np.ndarray) -> np.ndarray:
    return np.clip(values, 0, 255).astype(np.uint8)                                         # albumentations clip(): np.clip + astype (truncation)


def albu_adjust_brightness(img: np.ndarray, factor: float) -> np.ndarray:
    if factor == 0:
        return np.zeros_like(img)
    if factor == 1:
        return img
    return _lut_u8(np.arange(0, 256) * factor)[img]


def albu_contrast_mean(img: np.ndarray) -> float:
    return float(cv2_rgb2gray_u8(img).mean())


def albu_adjust_contrast(img: np.ndarray, factor: float) -> np.ndarray:
    if factor == 1:
        return img
    mean = albu_contrast_mean(img)
    if factor == 0:
        return np.full_like(img, int(mean + 0.5))
    return _lut_u8(np.arange(0, 256) * factor + mean * (1 - factor))[img]


def albu_adjust_saturation(img: np.ndarray, factor: float) -> np.ndarray:
    if factor == 1:
        return img
    gray = np.repeat(cv2_rgb2gray_u8(img)[..., None], 3, -1)
    if factor == 0:
        return gray
    return cv2_add_weighted_u8(img, factor, gray, 1 - factor, 0.0)


def albu_hue_lut(factor: float) -> np.ndarray:
    return np.mod(np.arange(0, 256, dtype=np.int16) + 180 * factor, 180).astype(np.uint8)


def albu_adjust_hue(img: np.ndarray, factor: float) -> np.ndarray:
    if factor == 0:
        return img
    hsv = cv2_rgb2hsv_u8(img)
    hsv[..., 0] = albu_hue_lut(factor)[hsv[..., 0]]
    return cv2_hsv2rgb_u8(hsv)


ALBU_JITTER_OPS = (albu_adjust_brightness, albu_adjust_contrast, albu_adjust_saturation, albu_adjust_hue)


def albu_color_jitter(img: np.ndarray, order, factors) -> np.ndarray:
    """A.ColorJitter.apply: the four adjustments in the drawn order; factors = (brightness, contrast, saturation, hue)."""
    for k in order:
        img = ALBU_JITTER_OPS[int(k)](img, float(factors[int(k)]))
    return img


def albu_to_gray(img: np.ndarray) -> np.ndarray:
    """A.ToGray: cv2 RGB2GRAY then GRAY2RGB."""
    return np.repeat(cv2_rgb2gray_u8(img)[..., None], 3, -1)


# --------------------------------------------------------------------------- BoxBlur.c
def gaussian_box_radius(sigma: float, passes: int = 3) -> float:
    """_gaussian_blur_radius: the (fractional) box radius whose `passes`-fold repetition has variance sigma^2."""
    f32 = np.float32
    sigma2 = f32(f32(f32(sigma) * f32(sigma)) / f32(passes))
    big_l = f32(math.sqrt(12.0 * float(sigma2) + 1.0))
    l = f32(math.floor((float(big_l) - 1.0) / 2.0))
    a = f32(f32(f32(2) * l + f32(1)) * f32(f32(l * f32(l + f32(1))) - f32(f32(3) * sigma2)))
    a = f32(a / f32(f32(6) * f32(sigma2 - f32(f32(l + f32(1)) * f32(l + f32(1))))))
    return float(f32(l + a))


def box_weights(radius: float):
    """(integer radius, ww, fw) of ImagingHorizontalBoxBlur: 24-bit weights of the inner window and the two far pixels."""
    f32 = np.float32
    r = int(radius)
    ww = int(f32(1 << 24) / f32(f32(f32(radius) * f32(2)) + f32(1)))
    fw = ((1 << 24) - (r * 2 + 1) * ww) // 2
    return r, ww, fw


def _box_pass_axis1(img: np.ndarray, r: int, ww: int, fw: int) -> np.ndarray:
    n = img.shape[1]
    src = img.astype(np.int64)
    idx = np.arange(n)
    acc = np.zeros(img.shape, dtype=np.int64)
    for d in range(-r, r + 1):
        acc += src[:, np.clip(idx + d, 0, n - 1)]
    far = src[:, np.clip(idx - r - 1, 0, n - 1)] + src[:, np.clip(idx + r + 1, 0, n - 1)]
    return (((acc * ww + far * fw) & 0xFFFFFFFF) + (1 << 23) >> 24).astype(np.uint8)


def gaussian_blur(img: np.ndarray, sigma: float) -> np.ndarray:
    """img.filter(ImageFilter.GaussianBlur(sigma)) = ImagingGaussianBlur(passes=3): three box passes along x, then three
    along y, uint8 after every pass, edges replicated."""
    return box_blur3(img, *box_weights(gaussian_box_radius(sigma)))


def box_blur3(img: np.ndarray, r: int, ww: int, fw: int) -> np.ndarray:
    """ImagingBoxBlur(n = 3) with explicit weights."""
    cur = img
    for _ in range(3):
        cur = _box_pass_axis1(cur, r, ww, fw)
    cur = np.swapaxes(cur, 0, 1)
    for _ in range(3):
        cur = _box_pass_axis1(cur, r, ww, fw)
    return np.ascontiguousarray(np.swapaxes(cur, 0, 1))


def to_tensor(img: np.ndarray) -> np.ndarray:
    """ToTensor: HWC uint8 -> CHW float32 / 255."""
    return (img.astype(np.float32) / np.float32(255.0)).transpose(2, 0, 1)


def background_view(img, box, h, w, jitter, gray, sigma, flip, rect):
    """One background view of main.py:204-225 with explicit parameters: jitter = None or (order, factors); gray / flip
    bool; sigma = None or the blur's sigma; rect = erase box (top, left, h, w)."""
    cur = pil_crop_resize(img, box, h, w)
    if jitter is not None:
        cur = color_jitter(cur, *jitter)
    if gray:
        cur = to_grayscale3(cur)
    if sigma is not None:
        cur = gaussian_blur(cur, sigma)
    if flip:
        cur = cur[:, ::-1]
    out = to_tensor(cur).copy()
    t, l, eh, ew = [int(v) for v in rect]
    out[:, t:t + eh, l:l + ew] = 0.0
    return out
