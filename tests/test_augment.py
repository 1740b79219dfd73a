"""f1: the on-device input pipeline.  CPU part: the oracle restatement against the reference's own
known-answer test for the id-map rule (tests/test_correlation_mapping.py:188-206) and the host-side parameter samplers.
GPU part: the HIP kernels against the oracle with the same explicit parameters -- bit-exact id maps, erase rectangles
and (fp32, same operation order, no FMA) images -- and the batch contract MODEL.forward expects."""
import numpy as np
import pytest
import torch

from cp2_amd import augment as A
from oracle import cp2_oracle as O


# ------------------------------------------------------------------ CPU
def test_kat_pixel_id_resize_rule():
    """reference tests/test_correlation_mapping.py:188-206: stride 1 reproduces arange(1..h*w); stride 2 halves the
    grid (shape[0] * stride == h) and the resize back has the original shape."""
    h, w = 10, 15
    ids = np.arange(1, h * w + 1).reshape(h, w)
    assert np.array_equal(O.pixel_id_map(h, w, 1), ids)
    small = ids[1::2, 1::2]
    assert small.shape[0] * 2 == h
    up = O.pixel_id_map(h, w, 2)
    assert up.shape == ids.shape
    # INTER_NEAREST_EXACT: source = floor((dst + 0.5) * small / big): rows 10 -> 5 pair up exactly, columns 15 <- 7 do not
    assert np.array_equal(up[0::2], up[1::2]) and np.array_equal(up[0::2, ((2 * np.arange(7) + 1) * 15) // 14], small)
    cols = ((2 * np.arange(w) + 1) * small.shape[1]) // (2 * w)
    assert np.array_equal(up, small[np.arange(h) // 2][:, cols]) and set(np.unique(up)) == set(np.unique(small))


def test_oracle_crop_identity_flip_and_nearest_ids():
    rng = np.random.default_rng(0)
    src = rng.random((3, 12, 20), dtype=np.float32)
    img, pix, reg = O.crop_resize_flip(src, None, (0, 0, 12, 20), False, 12, 20)
    assert np.array_equal(img, src) and np.array_equal(pix, np.arange(1, 241).reshape(12, 20)) and np.array_equal(reg, pix)
    imgf, pixf, _ = O.crop_resize_flip(src, None, (0, 0, 12, 20), True, 12, 20)
    assert np.array_equal(imgf, src[:, :, ::-1]) and np.array_equal(pixf, pix[:, ::-1])
    # 2x up-sampling of a crop: every source id appears as a 2x2 block
    _, pix2, _ = O.crop_resize_flip(src, None, (2, 4, 5, 8), False, 10, 16)
    want = (np.arange(2, 7)[:, None] * 20 + np.arange(4, 12)[None, :] + 1)
    assert np.array_equal(pix2[::2, ::2], want) and np.array_equal(pix2[1::2, 1::2], want)
    region = rng.integers(0, 9, (12, 20))
    _, _, reg2 = O.crop_resize_flip(src, region, (2, 4, 5, 8), True, 10, 16)
    assert np.array_equal(reg2[::2, ::2], region[2:7, 4:12][:, ::-1])


def test_parameter_samplers_follow_the_transforms_rules():
    rng = np.random.default_rng(1)
    hs, ws = 300, 400
    box = A.rrc_params(rng, 4000, hs, ws)
    top, left, h, w = box.T
    assert (top >= 0).all() and (left >= 0).all() and (top + h <= hs).all() and (left + w <= ws).all() and (h > 0).all()
    frac, ar = h * w / (hs * ws), w / h
    assert frac.min() > 0.19 and frac.max() <= 1.0 and 0.70 < ar.min() and ar.max() < 1.40          # scale (0.2, 1), ratio (3/4, 4/3) up to rounding
    assert 0.4 < frac.mean() < 0.62                                                                    # U(0.2, 1), large boxes rejected more often
    er = A.erase_params(rng, 4000, 224, 224, (0.5, 0.8))
    t, l, eh, ew = er.T
    assert (eh < 224).all() and (ew < 224).all() and (t + eh <= 224).all() and (l + ew <= 224).all()
    ef = eh * ew / 224.0 ** 2
    assert (eh > 0).mean() > 0.99 and 0.49 < ef[eh > 0].min() and ef.max() < 0.81
    # a box that can never fit falls back to the clamped central crop (RandomResizedCrop) / to "no erase" (RandomErasing)
    fb = A.rrc_params(rng, 8, 10, 100, scale=(0.9, 1.0))
    assert (fb == np.array([0, 43, 10, 13], dtype=np.int32)).all()                                      # h = H, w = round(H * 4/3)
    assert (A.erase_params(rng, 8, 8, 8, (1.0, 1.0), (1.0, 1.0))[:, 2:] == 0).all()
    tab = A.crop_table(np.arange(3), box[:3], np.array([True, False, True]))
    assert tab.shape == (3, 8) and tab.dtype == np.int32 and tab[:, 5].tolist() == [1, 0, 1] and (tab[:, 6:] == 0).all()


def test_epoch_sampler_partitions_like_distributed_sampler():
    n, world = 103, 4
    parts = [A.EpochSampler(n, world, r, 1024).indices(3) for r in range(world)]
    assert all(len(p) == n // world for p in parts)
    allidx = np.concatenate(parts)
    assert len(set(allidx.tolist())) == len(allidx)                                  # disjoint across ranks
    g = torch.Generator().manual_seed(1024 + 3)
    perm = torch.randperm(n, generator=g)[: (n // world) * world].numpy()
    assert np.array_equal(np.stack(parts, 1).reshape(-1), perm)                      # rank r takes perm[r::world]
    assert not np.array_equal(parts[0], A.EpochSampler(n, world, 0, 1024).indices(4))


def test_cv2_linear_restatement_properties():
    """oracle.augment_oracle.cv2_resize_linear_u8 (OpenCV's INTER_LINEAR on uint8, restated from its source: cv2 is absent,
    so these are the properties the arithmetic must have, not a pin): same-size resize is the identity, constants survive,
    an exact 2 x 2 shrink is the rounded block mean (INTER_AREA's fast path), a ramp stays monotone, borders replicate."""
    from oracle import augment_oracle as AO
    rng = np.random.default_rng(0)
    img = rng.integers(0, 256, (37, 53, 3), dtype=np.uint8)
    assert np.array_equal(AO.cv2_resize_linear_u8(img, 37, 53), img)
    assert (AO.cv2_resize_linear_u8(np.full((30, 41, 3), 173, np.uint8), 64, 80) == 173).all()
    big = rng.integers(0, 256, (48, 80, 3), dtype=np.uint8).astype(np.int64)
    want = (big[0::2, 0::2] + big[0::2, 1::2] + big[1::2, 0::2] + big[1::2, 1::2] + 2) >> 2
    assert np.array_equal(AO.cv2_resize_linear_u8(big.astype(np.uint8), 24, 40), want.astype(np.uint8))
    ramp = np.tile(np.linspace(0, 255, 20).astype(np.uint8)[None, :, None], (5, 1, 3))
    up = AO.cv2_resize_linear_u8(ramp, 5, 77).astype(int)
    assert (np.diff(up[0, :, 0]) >= 0).all() and up[0, 0, 0] == 0 and up[0, -1, 0] == 255
    f = AO.foreground_crop_u8(img.transpose(2, 0, 1), (2, 3, 20, 30), True, 16, 24)
    assert np.array_equal(f, AO.foreground_crop_u8(img.transpose(2, 0, 1), (2, 3, 20, 30), False, 16, 24)[:, ::-1])


# ------------------------------------------------------------------ GPU
@pytest.mark.gpu
@pytest.mark.parametrize("u8,stride", [(False, 1), (True, 1), (False, 2)])
def test_crop_resize_flip_and_erase_match_oracle_bit_for_bit(u8, stride):
    from cp2_amd import ops
    rng = np.random.default_rng(5)
    N, Hs, Ws, H, W, B = 5, 70, 93, 64, 80, 12
    src = rng.integers(0, 256, (N, 3, Hs, Ws), dtype=np.uint8) if u8 else rng.random((N, 3, Hs, Ws), dtype=np.float32)
    region = rng.integers(0, 30, (N, Hs, Ws)).astype(np.int64)
    idx = rng.integers(0, N, B)
    boxes = A.rrc_params(rng, B, Hs, Ws)
    boxes[0] = (0, 0, Hs, Ws)                                    # whole image
    boxes[1] = (Hs - 1, Ws - 1, 1, 1)                            # a single source pixel
    flips = rng.random(B) < 0.5
    tab = torch.from_numpy(A.crop_table(idx, boxes, flips)).cuda()
    img, pix, reg = ops.crop_resize_flip(torch.from_numpy(src).cuda(), torch.from_numpy(region).cuda(), tab, H, W, stride)
    rects = A.erase_params(rng, B, H, W, (0.5, 0.8))
    rects[2] = (0, 0, 0, 0)                                      # "no erase"
    erased = img.clone()
    ops.erase_rect(erased, torch.from_numpy(rects).cuda())
    srcf = (src.astype(np.float32) / np.float32(255.0)) if u8 else src
    for b in range(B):
        want_img, want_pix, want_reg = O.crop_resize_flip(srcf[idx[b]], region[idx[b]], boxes[b], bool(flips[b]), H, W, stride)
        if u8:      # a uint8 dataset is resampled as cv2.resize(INTER_LINEAR) does on uint8, then divided by 255 (ToTensor)
            from oracle import augment_oracle as AO
            want_img = (AO.foreground_crop_u8(src[idx[b]], boxes[b], bool(flips[b]), H, W).transpose(2, 0, 1).astype(np.float32)
                        / np.float32(255.0))
        assert np.array_equal(pix[b].cpu().numpy(), want_pix), b
        assert np.array_equal(reg[b].cpu().numpy(), want_reg), b
        assert np.array_equal(img[b].cpu().numpy(), want_img), b                                  # same fp32 operations, same order
        assert np.array_equal(erased[b].cpu().numpy(), O.erase_rect(want_img, rects[b])), b
    # region ids default to the pixel ids (MappingType.CP2, loader.py:84-85)
    _, pix2, reg2 = ops.crop_resize_flip(torch.from_numpy(src).cuda(), None, tab, H, W, stride)
    assert torch.equal(pix2, reg2) and torch.equal(pix2, pix)


@pytest.mark.gpu
@pytest.mark.parametrize("photometric", [True, False])
def test_step_batch_contract_and_model_step(photometric):
    """make_step_batch gives MODEL.forward its keyword set (main.py:616-628): dtypes, shapes, an exactly-zero rectangle
    in every background, id maps of two crops of the same image that share ids -- and a model step runs on it.  With the
    photometric transforms on (the default for a uint8 dataset) every view is a multiple of 1/255 (ToTensor of a uint8
    image); the source pixels are >= 128 so that no colour adjustment can reach zero outside the erased rectangle."""
    import os
    from cp2_amd import builder, ops
    from cp2_amd.config import Config
    from cp2_amd.pretrain_types import PretrainType
    g = torch.Generator().manual_seed(0)
    ds = A.DeviceDataset(torch.randint(128, 256, (40, 3, 96, 120), dtype=torch.uint8, generator=g))
    rng = np.random.default_rng(0)
    b, H = 8, 64
    s = [A.EpochSampler(len(ds), 1, 0, seed).indices(0) for seed in (0, 1024, 2048)]
    batch = A.make_step_batch(ds, s[0][:b], s[1][:b], s[2][:b], H, H, rng, photometric=photometric)
    if photometric:
        for k in ("img_a", "img_b", "bg0", "bg1"):
            q = batch[k] * 255.0
            assert (q - q.round()).abs().max() < 1e-4
    else:
        assert A.make_step_batch(ds, s[0][:b], s[1][:b], s[2][:b], H, H, np.random.default_rng(0), photometric=None)["img_a"].shape == (b, 3, H, H)
    assert set(batch) == {"img_a", "img_b", "bg0", "bg1", "pixel_ids_a", "pixel_ids_b", "region_ids_a", "region_ids_b"}
    for k in ("img_a", "img_b", "bg0", "bg1"):
        assert batch[k].shape == (b, 3, H, H) and batch[k].dtype == torch.float32 and 0 <= float(batch[k].min()) and float(batch[k].max()) <= 1
    for k in ("pixel_ids_a", "pixel_ids_b", "region_ids_a", "region_ids_b"):
        assert batch[k].shape == (b, H, H) and batch[k].dtype == torch.int64 and int(batch[k].min()) >= 1
    for k in ("bg0", "bg1"):
        zero = (batch[k] == 0).all(1)                            # [b, H, H]: erased in all three channels
        frac = zero.float().mean((1, 2))
        assert (frac > 0.45).all() and (frac < 0.85).all()       # RandomErasing scale (0.5, 0.8); source pixels are >= 1/255
        assert torch.equal(zero, batch[k][:, 0] == 0)            # the mask rule of builder.py:1146 sees the same rectangle
    iou, _ = ops.corr_iou(batch["pixel_ids_a"], batch["pixel_ids_b"])      # full-resolution maps (P = 4096)
    assert float(iou.max()) > 0                                  # two crops of one image share source pixels somewhere
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    cfg = Config.fromfile(os.path.join(root, "configs", "config_pretrain_r18.py"))
    model = builder.MODEL(cfg, rank=0, K=256, pretrain_from_scratch=True, pretrain_type=PretrainType.CP2, device="cuda").cuda().train()
    loss = model(visualize=False, step=0, new_epoch=False, **batch)
    loss.backward()
    assert torch.isfinite(loss)
