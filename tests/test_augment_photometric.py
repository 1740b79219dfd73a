"""f1, photometric half (reference main.py:204-225, loader.py:121-152): ColorJitter, grayscale, GaussianBlur and the
background's crop + resize in Pillow's arithmetic.

CPU: oracle/augment_oracle.py against tests/golden/augment_pillow.npz (outputs of Pillow itself, written by
make_augment_goldens.py) bit for bit, and -- where Pillow is importable, as it is in the build image and on the GPU box --
against live Pillow calls over random sizes / parameters and over all 2^24 colours for the HSV round trip.
GPU: the HIP kernels (csrc/photometric.hip) through the C ABI against the same goldens and against the oracle on larger
random cases, bit for bit; the foreground's uint8 view against the oracle's rule (that resampling is parity-unpinned:
cv2 is not installed)."""
import os

import numpy as np
import pytest
import torch

from cp2_amd import augment as A
from oracle import augment_oracle as P
from oracle import cp2_oracle as O

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "augment_pillow.npz")


@pytest.fixture(scope="module")
def gold():
    return dict(np.load(GOLD))


def _order_and_factors(g, i):
    order = [int(k) for k in g["view_order"][i] if k >= 0]
    vf = g["view_factors"][i]
    return order, (vf[0], vf[1], vf[2], vf[3])


# ------------------------------------------------------------------ CPU: oracle == Pillow's outputs
def test_oracle_single_operations_equal_pillow_goldens(gold):
    g = gold
    src, h, w = g["src"], g["resized"].shape[1], g["resized"].shape[2]
    for i in range(len(src)):
        assert np.array_equal(P.pil_crop_resize(src[i], g["boxes"][i], h, w), g["resized"][i]), i
        assert np.array_equal(P.adjust_brightness(src[i], g["factors"][i, 0]), g["brightness"][i]), i
        assert np.array_equal(P.adjust_contrast(src[i], g["factors"][i, 1]), g["contrast"][i]), i
        assert np.array_equal(P.adjust_saturation(src[i], g["factors"][i, 2]), g["saturation"][i]), i
        assert np.array_equal(P.adjust_hue(src[i], g["hue"][i]), g["hue_out"][i]), i
        assert np.array_equal(P.rgb_to_l(src[i]), g["gray"][i]), i
        assert np.array_equal(P.gaussian_blur(src[i], g["sigma"][i]), g["blurred"][i]), i


def test_oracle_background_views_equal_pillow_goldens(gold):
    g = gold
    h, w = g["views"].shape[2:]
    for i in range(len(g["views"])):
        order, factors = _order_and_factors(g, i)
        got = P.background_view(g["src"][g["view_idx"][i]], g["view_box"][i], h, w, (order, factors) if order else None,
                                bool(g["view_gray"][i]), float(g["view_sigma"][i]) if g["view_sigma"][i] > 0 else None,
                                bool(g["view_flip"][i]), g["view_rect"][i])
        assert np.array_equal(got, g["views"][i]), i


def test_oracle_against_live_pillow():
    """Random sizes, boxes, factors; and the HSV conversions over every colour."""
    Image = pytest.importorskip("PIL.Image")
    from PIL import ImageEnhance, ImageFilter
    rng = np.random.default_rng(7)
    v = np.arange(0, 256, dtype=np.uint8)
    allc = np.stack(np.meshgrid(v, v, v, indexing="ij"), -1).reshape(4096, 4096, 3)
    assert np.array_equal(np.asarray(Image.fromarray(allc).convert("HSV")), P.rgb_to_hsv(allc))
    assert np.array_equal(np.asarray(Image.fromarray(allc, "HSV").convert("RGB")), P.hsv_to_rgb(allc))
    assert np.array_equal(np.asarray(Image.fromarray(allc).convert("L")), P.rgb_to_l(allc))
    for it in range(40):
        hs, ws = int(rng.integers(8, 200)), int(rng.integers(8, 200))
        src = rng.integers(0, 256, (hs, ws, 3), dtype=np.uint8)
        im = Image.fromarray(src)
        f = float(rng.uniform(0.6, 1.4))
        assert np.array_equal(np.asarray(ImageEnhance.Brightness(im).enhance(f)), P.adjust_brightness(src, f))
        assert np.array_equal(np.asarray(ImageEnhance.Contrast(im).enhance(f)), P.adjust_contrast(src, f))
        assert np.array_equal(np.asarray(ImageEnhance.Color(im).enhance(f)), P.adjust_saturation(src, f))
        sg = float(rng.uniform(0.1, 2.0 if it % 4 else 5.0))
        assert np.array_equal(np.asarray(im.filter(ImageFilter.GaussianBlur(radius=sg))), P.gaussian_blur(src, sg)), sg
        ch, cw = int(rng.integers(1, hs + 1)), int(rng.integers(1, ws + 1))
        top, left = int(rng.integers(0, hs - ch + 1)), int(rng.integers(0, ws - cw + 1))
        H, W = int(rng.integers(1, 120)), int(rng.integers(1, 120))
        want = np.asarray(im.crop((left, top, left + cw, top + ch)).resize((W, H), Image.BILINEAR))
        assert np.array_equal(want, P.pil_crop_resize(src, (top, left, ch, cw), H, W)), (ch, cw, H, W)


def test_host_tables_follow_the_transforms():
    rng = np.random.default_rng(3)
    t = A.jitter_table(rng, 4000)
    applied = t[:, 0] >= 0
    assert 0.77 < applied.mean() < 0.83 and 0.17 < t[:, 8].mean() < 0.23                     # p = 0.8, grayscale p = 0.2
    assert all(sorted(r) == [0, 1, 2, 3] for r in t[applied][:200, :4].tolist()) and (t[~applied, :4] == -1).all()
    f = t[:, 4:7].copy().view(np.float32)
    assert f.min() >= 0.6 and f.max() <= 1.4 and abs(f.mean() - 1.0) < 0.02
    assert set(np.unique(t[:, 7])) <= set(range(0, 26)) | set(range(231, 256))               # uint8(U(-0.1, 0.1) * 255) wraps
    assert A.hue_shift_u8([-0.1, -0.05, -0.0001, 0.03, 0.1]).tolist() == [231, 244, 0, 7, 25]
    b, rmax = A.blur_table(rng, 4000)
    assert 0.46 < b[:, 0].mean() < 0.54 and rmax == 1 and (b[b[:, 0] == 0] == 0).all()      # sigma <= 2: box radius 0 or 1
    for s in rng.uniform(0.05, 3.4, 300):                                                    # product-side copy == oracle's
        assert A.gaussian_box(float(s)) == P.box_weights(P.gaussian_box_radius(float(s)))
    r, ww, fw = A.gaussian_box(2.0)
    assert (2 * r + 1) * ww + 2 * fw in ((1 << 24) - 1, 1 << 24)                             # the weights sum to one (24-bit)


# ------------------------------------------------------------------ GPU: kernels == Pillow's outputs
def _pack(img_hwc):
    x = img_hwc.astype(np.int32)
    return x[..., 0] | (x[..., 1] << 8) | (x[..., 2] << 16)


def _unpack(t):
    x = t.cpu().numpy().astype(np.int64)
    return np.stack([x & 255, (x >> 8) & 255, (x >> 16) & 255], -1).astype(np.uint8)


def _colour_row(order=(), factors=(1.0, 1.0, 1.0), hue=0.0, gray=False, arithmetic=0):
    row = np.zeros(12, dtype=np.int32)
    row[:4] = -1
    row[:len(order)] = order
    row[4:7] = np.asarray(factors, dtype=np.float32).view(np.int32)
    row[7] = A.hue_shift_u8(hue)
    row[8] = int(gray)
    row[9] = arithmetic
    row[10] = np.asarray([hue], dtype=np.float32).view(np.int32)[0]
    return row


def test_cv2_colour_restatement_known_answers():
    """The restatement of OpenCV's 8-bit colour conversions and of albumentations' ColorJitter helpers (the reference's
    foreground views, main.py:236-237; neither library is in the image: PARITY-UNPINNED) on values that follow from the
    published definitions: the primaries' hue sextants (H = 0, 30, ..., 150 on cv2's 0..179 scale, S = V = 255), gray
    levels with S = 0, Y = 0.299 R + 0.587 G + 0.114 B in 15-bit fixed point, exact HSV round trips of saturated colours,
    albumentations' look-up tables, the identity cases of every adjustment."""
    px = lambda *c: np.array([[c]], dtype=np.uint8)                                   # noqa: E731
    assert [int(P.cv2_rgb2gray_u8(px(*c))[0, 0]) for c in ((255, 0, 0), (0, 255, 0), (0, 0, 255), (255, 255, 255), (0, 0, 0))] == [76, 150, 29, 255, 0]
    sext = {(255, 0, 0): 0, (255, 255, 0): 30, (0, 255, 0): 60, (0, 255, 255): 90, (0, 0, 255): 120, (255, 0, 255): 150}
    for rgb, h in sext.items():
        hsv = P.cv2_rgb2hsv_u8(px(*rgb))
        assert hsv[0, 0].tolist() == [h, 255, 255]
        assert P.cv2_hsv2rgb_u8(hsv)[0, 0].tolist() == list(rgb)
    for g in (0, 1, 77, 128, 255):
        assert P.cv2_rgb2hsv_u8(px(g, g, g))[0, 0].tolist() == [0, 0, g]
        assert P.cv2_hsv2rgb_u8(np.array([[[37, 0, g]]], dtype=np.uint8))[0, 0].tolist() == [g, g, g]
    # S = 127.5 exactly; the table entry round(255 * 4096 / 200) = 5222 is a little low: (100 * 5222 + 2048) >> 12 = 127
    assert P.cv2_rgb2hsv_u8(px(200, 100, 100))[0, 0].tolist() == [0, 127, 200]
    assert P.cv2_rgb2hsv_u8(px(100, 100, 200))[0, 0, 0] == 120 and P.cv2_rgb2hsv_u8(px(100, 200, 150))[0, 0, 0] == 75
    # albumentations' tables: truncation, python-style modulo
    assert P.albu_hue_lut(0.1)[[0, 161, 162, 179]].tolist() == [18, 179, 0, 17]
    assert P.albu_hue_lut(-0.05)[[0, 8, 9, 179]].tolist() == [171, 179, 0, 170]
    rng = np.random.default_rng(5)
    img = rng.integers(0, 256, (16, 16, 3), dtype=np.uint8)
    assert P.albu_adjust_brightness(img, 1.3).max() == 255 and np.array_equal(P.albu_adjust_brightness(img, 0.5), (img * 0.5).astype(np.uint8))
    for op in P.ALBU_JITTER_OPS[:3]:
        assert np.array_equal(op(img, 1), img)
    assert np.array_equal(P.albu_adjust_hue(img, 0), img)
    assert np.array_equal(P.albu_adjust_saturation(img, 0), P.albu_to_gray(img))
    m = P.albu_contrast_mean(img)
    assert np.array_equal(P.albu_adjust_contrast(img, 0), np.full_like(img, int(m + 0.5)))
    assert np.array_equal(P.albu_adjust_contrast(img, 0.5), np.clip(img * 0.5 + m * 0.5, 0, 255).astype(np.uint8))
    # the round trip through cv2's 8-bit HSV is lossy by a few levels, never more (180 hue steps, 8-bit S)
    rt = P.cv2_hsv2rgb_u8(P.cv2_rgb2hsv_u8(img))
    assert np.abs(rt.astype(int) - img.astype(int)).max() <= 6


def _blur_row(sigma):
    if sigma is None or sigma <= 0:
        return np.zeros(4, dtype=np.int32)
    r, ww, fw = A.gaussian_box(float(sigma))
    return np.array([1, r, ww, fw], dtype=np.int32)


@pytest.mark.gpu
def test_kernels_single_operations_equal_pillow_goldens(gold):
    from cp2_amd import ops
    g = gold
    src = g["src"]
    n, hs, ws = src.shape[:3]
    h, w = g["resized"].shape[1:3]
    planar = torch.from_numpy(np.ascontiguousarray(src.transpose(0, 3, 1, 2))).cuda()
    tab = torch.from_numpy(A.crop_table(np.arange(n), g["boxes"], np.zeros(n, dtype=bool))).cuda()
    assert np.array_equal(_unpack(ops.pil_resize_crop(planar, tab, h, w)), g["resized"])
    packed = torch.from_numpy(_pack(src)).cuda()
    for name, k in (("brightness", 0), ("contrast", 1), ("saturation", 2)):
        x = packed.clone()
        rows = np.stack([_colour_row((k,), g["factors"][i]) for i in range(n)])
        ops.color_ops(x, torch.from_numpy(rows).cuda())
        assert np.array_equal(_unpack(x), g[name]), name
    x = packed.clone()
    ops.color_ops(x, torch.from_numpy(np.stack([_colour_row((3,), hue=g["hue"][i]) for i in range(n)])).cuda())
    assert np.array_equal(_unpack(x), g["hue_out"])
    x = packed.clone()
    ops.color_ops(x, torch.from_numpy(np.stack([_colour_row(gray=True)] * n)).cuda())
    assert np.array_equal(_unpack(x)[..., 0], g["gray"]) and np.array_equal(_unpack(x)[..., 1], g["gray"])
    rows = np.stack([_blur_row(s) for s in g["sigma"]])
    out = ops.blur_to_tensor(packed, torch.from_numpy(rows).cuda(), None, int(rows[:, 1].max()))
    want = g["blurred"].astype(np.float32).transpose(0, 3, 1, 2) / np.float32(255.0)
    assert np.array_equal(out.cpu().numpy(), want)


@pytest.mark.gpu
def test_kernels_background_views_equal_pillow_goldens(gold):
    """The whole background chain -- crop + resize + flip, colour adjustments in the drawn order, grayscale, blur,
    ToTensor, erase -- in four launches for the batch, against Pillow's own output of the same chain."""
    from cp2_amd import ops
    g = gold
    m = len(g["views"])
    h, w = g["views"].shape[2:]
    planar = torch.from_numpy(np.ascontiguousarray(g["src"].transpose(0, 3, 1, 2))).cuda()
    tab = torch.from_numpy(A.crop_table(g["view_idx"], g["view_box"], g["view_flip"])).cuda()
    rgbx = ops.pil_resize_crop(planar, tab, h, w)
    rows = []
    for i in range(m):
        order, f = _order_and_factors(g, i)
        rows.append(_colour_row(order, f[:3], f[3], bool(g["view_gray"][i])))
    ops.color_ops(rgbx, torch.from_numpy(np.stack(rows)).cuda())
    brow = np.stack([_blur_row(s) for s in g["view_sigma"]])
    out = ops.blur_to_tensor(rgbx, torch.from_numpy(brow).cuda(), torch.from_numpy(g["view_rect"]).cuda(), int(brow[:, 1].max()))
    assert np.array_equal(out.cpu().numpy(), g["views"])


@pytest.mark.gpu
def test_hue_kernel_over_all_colours_and_shifts():
    """rgb -> hsv -> +shift -> rgb for every one of the 2^24 colours at several shifts, against the oracle (which equals
    Pillow's Convert.c over all colours, see the CPU test)."""
    from cp2_amd import ops
    v = np.arange(256, dtype=np.uint8)
    allc = np.stack(np.meshgrid(v, v, v, indexing="ij"), -1).reshape(16, 1024, 1024, 3)
    packed = torch.from_numpy(_pack(allc)).cuda()
    for hue in (0.0, 0.1, -0.1, 0.037):
        x = packed.clone()
        ops.color_ops(x, torch.from_numpy(np.stack([_colour_row((3,), hue=hue)] * 16)).cuda())
        got = _unpack(x)
        for b in range(0, 16, 5):
            assert np.array_equal(got[b], P.adjust_hue(allc[b], hue)), (hue, b)


@pytest.mark.gpu
def test_cv2_arithmetic_kernel_over_all_colours():
    """The `arithmetic = 1` rows of cp2_color_ops (albumentations' ColorJitter / ToGray on cv2's 8-bit conversions: the
    foreground views) against the oracle's restatement for every one of the 2^24 colours: hue at several factors, saturation,
    brightness, grayscale; contrast (needs the image mean) on whole images below."""
    from cp2_amd import ops
    v = np.arange(256, dtype=np.uint8)
    allc = np.stack(np.meshgrid(v, v, v, indexing="ij"), -1).reshape(16, 1024, 1024, 3)
    packed = torch.from_numpy(_pack(allc)).cuda()
    cases = [((3,), dict(hue=h), lambda im, h=h: P.albu_adjust_hue(im, float(np.float32(h)))) for h in (0.1, -0.1, 0.037, -0.0625)]
    cases += [((2,), dict(factors=(1.0, 1.0, f)), lambda im, f=f: P.albu_adjust_saturation(im, float(np.float32(f)))) for f in (0.6, 1.37)]
    cases += [((0,), dict(factors=(f, 1.0, 1.0)), lambda im, f=f: P.albu_adjust_brightness(im, float(np.float32(f)))) for f in (0.61, 1.4)]
    cases += [((), dict(gray=True), P.albu_to_gray)]
    for order, kw, fn in cases:
        x = packed.clone()
        ops.color_ops(x, torch.from_numpy(np.stack([_colour_row(order, arithmetic=1, **kw)] * 16)).cuda())
        got = _unpack(x)
        for b in range(0, 16, 5):
            assert np.array_equal(got[b], fn(allc[b])), (order, kw, b)


@pytest.mark.gpu
def test_cv2_arithmetic_kernel_on_images_in_every_order():
    """Whole images through the albumentations-on-cv2 rows: all four adjustments in drawn orders (contrast first / in the
    middle / last: its mean is the float64 mean of cv2's gray image AT THAT POINT), with and without ToGray, beside
    Pillow-arithmetic rows in the same launch."""
    from cp2_amd import ops
    rng = np.random.default_rng(21)
    B, H, W = 10, 96, 80
    imgs = rng.integers(0, 256, (B, H, W, 3), dtype=np.uint8)
    imgs[:4] = np.stack([P.gaussian_blur(s, 2.0) for s in imgs[:4]])
    colour = A.jitter_table(rng, B, p=1.0, p_gray=0.3, arithmetic=1)
    colour[0, :4] = (1, 3, 0, 2)
    colour[1, :4] = (0, 2, 3, 1)
    colour[2, :4] = (3, 1, 2, 0)
    colour[3, :4] = -1
    colour[8:, 9] = 0                                                                   # two Pillow rows in the same call
    x = torch.from_numpy(_pack(imgs)).cuda()
    ops.color_ops(x, torch.from_numpy(colour).cuda())
    got = _unpack(x)
    for b in range(B):
        order = [int(k) for k in colour[b, :4] if k >= 0]
        f = colour[b, 4:7].copy().view(np.float32)
        if colour[b, 9] == 1:
            hue = float(colour[b, 10:11].copy().view(np.float32)[0])
            want = P.albu_color_jitter(imgs[b], order, (float(f[0]), float(f[1]), float(f[2]), hue))
            if colour[b, 8]:
                want = P.albu_to_gray(want)
        else:
            want = imgs[b]
            for k in order:
                want = P.adjust_hue_shift(want, int(colour[b, 7])) if k == 3 else P.JITTER_OPS[k](want, float(f[k]))
            if colour[b, 8]:
                want = P.to_grayscale3(want)
        assert np.array_equal(got[b], want), (b, order, int(colour[b, 9]))


@pytest.mark.gpu
def test_kernels_vs_oracle_at_training_size():
    """224 x 224 views from 300 x 400 sources (antialiased down-scaling: 5-7 taps), every adjustment order that puts the
    contrast step first / in the middle / last, blur radii 0 and 1, erase rectangles."""
    from cp2_amd import ops
    rng = np.random.default_rng(11)
    N, Hs, Ws, H, W, B = 6, 300, 400, 224, 224, 12
    src = rng.integers(0, 256, (N, Hs, Ws, 3), dtype=np.uint8)
    src[:3] = np.stack([P.gaussian_blur(s, 3.0) for s in src[:3]])                      # some smooth images
    planar = torch.from_numpy(np.ascontiguousarray(src.transpose(0, 3, 1, 2))).cuda()
    idx = rng.integers(0, N, B)
    boxes = A.rrc_params(rng, B, Hs, Ws)
    boxes[0] = (0, 0, Hs, Ws)
    boxes[1] = (10, 20, 224, 224)                                                       # no resampling at all
    boxes[2] = (10, 20, 100, 224)                                                       # vertical pass only
    flips = rng.random(B) < 0.5
    colour = A.jitter_table(rng, B, p=1.0, p_gray=0.3)
    colour[3, :4] = (1, 3, 0, 2)
    colour[4, :4] = (0, 2, 3, 1)
    colour[5, :4] = -1
    blur, rmax = A.blur_table(rng, B, p=0.7)
    sig = {}
    rects = A.erase_params(rng, B, H, W)
    tab = torch.from_numpy(A.crop_table(idx, boxes, flips)).cuda()
    rgbx = ops.pil_resize_crop(planar, tab, H, W)
    resized = _unpack(rgbx)
    ops.color_ops(rgbx, torch.from_numpy(colour).cuda())
    coloured = _unpack(rgbx)
    out = ops.blur_to_tensor(rgbx, torch.from_numpy(blur).cuda(), torch.from_numpy(rects).cuda(), rmax).cpu().numpy()
    for b in range(B):
        want = P.pil_crop_resize(src[idx[b]], boxes[b], H, W)
        if flips[b]:
            want = want[:, ::-1]
        assert np.array_equal(resized[b], want), ("resize", b)
        order = [int(k) for k in colour[b, :4] if k >= 0]
        f = colour[b, 4:7].copy().view(np.float32)
        for k in order:
            want = P.adjust_hue_shift(want, int(colour[b, 7])) if k == 3 else P.JITTER_OPS[k](want, float(f[k]))
        if colour[b, 8]:
            want = P.to_grayscale3(want)
        assert np.array_equal(coloured[b], want), ("colour", b, order)
        if blur[b, 0]:
            want = P.box_blur3(want, int(blur[b, 1]), int(blur[b, 2]), int(blur[b, 3]))
        t = P.to_tensor(want).copy()
        et, el, eh, ew = rects[b]
        t[:, et:et + eh, el:el + ew] = 0.0
        assert np.array_equal(out[b], t), ("blur/tensor", b)


@pytest.mark.gpu
def test_foreground_uint8_view_and_blur_radius_guard():
    """The foreground crop's uint8 view = cv2.resize(crop, INTER_LINEAR) in its integer arithmetic as the oracle restates it
    from OpenCV's source (11-bit weights, int32 horizontal pass, shifted vertical pass; an exact 2 x 2 shrink through the
    area fast path) -- kernel == restatement bit for bit; cv2 itself is absent, so the restatement is parity-unpinned -- and
    the argument checks of the blur entry."""
    from cp2_amd import _lib, ops
    rng = np.random.default_rng(2)
    N, Hs, Ws, H, W, B = 3, 100, 120, 40, 44, 7
    src = rng.integers(0, 256, (N, 3, Hs, Ws), dtype=np.uint8)
    idx, boxes, flips = rng.integers(0, N, B), A.rrc_params(rng, B, Hs, Ws), rng.random(B) < 0.5
    boxes[5] = (7, 9, 2 * H, 2 * W)                              # exact 2 x 2 shrink
    boxes[6] = (0, 0, 13, 17)                                    # enlargement
    tab = torch.from_numpy(A.crop_table(idx, boxes, flips)).cuda()
    rgbx = torch.empty((B, H, W), dtype=torch.int32, device="cuda")
    img, pix, reg = ops.crop_resize_flip(torch.from_numpy(src).cuda(), None, tab, H, W, 1, want_f32=False, out_rgbx=rgbx)
    assert img is None
    for b in range(B):
        _, want_pix, _ = O.crop_resize_flip(src[idx[b]].astype(np.float32) / np.float32(255.0), None, boxes[b], bool(flips[b]), H, W)
        assert np.array_equal(_unpack(rgbx)[b], P.foreground_crop_u8(src[idx[b]], boxes[b], bool(flips[b]), H, W)), b
        assert np.array_equal(pix[b].cpu().numpy(), want_pix)
    with pytest.raises(_lib.Cp2LibraryError):
        ops.blur_to_tensor(rgbx, torch.zeros((B, 4), dtype=torch.int32, device="cuda"), None, 99)
