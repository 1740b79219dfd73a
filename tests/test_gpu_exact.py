"""GPU: bit-exact parity of the integer / mask / copy kernels with the oracle and
with the golden vectors recorded from the reference (through the C ABI)."""
import os

import numpy as np
import pytest
import torch

from cp2_amd import ops
from oracle import cp2_oracle as O

pytestmark = pytest.mark.gpu
DEV = "cuda"


def load(golden_dir, name):
    return dict(np.load(os.path.join(golden_dir, name + ".npz")))


def G(x):
    return torch.from_numpy(np.ascontiguousarray(x)).to(DEV)


def synth_bg(b, h, w, gen):
    bg = torch.rand(b, 3, h, w, generator=gen)
    for n in range(b):
        rh, rw = int(h * 0.6), int(w * 0.7)
        y0 = int(torch.randint(0, h - rh + 1, (1,), generator=gen))
        x0 = int(torch.randint(0, w - rw + 1, (1,), generator=gen))
        bg[n, :, y0:y0 + rh, x0:x0 + rw] = 0.0
    return bg


@pytest.mark.parametrize("name", ["cp2_b4_64_k64", "cp2_b4_96_k64_wrap_bg", "cp2_b3_80x112_k1024"])
def test_compose_and_gathers_golden(golden_dir, name):
    g = load(golden_dir, name)
    stride = int(g["cfg"][4])
    out, mfull, mds = ops.compose_mask(G(g["in_img_a"]), G(g["in_bg0"]), stride, want_full_mask=True)
    assert np.array_equal(out.cpu().numpy(), g["img_a"])
    b = out.shape[0]
    assert np.array_equal(mds.reshape(b, -1).cpu().numpy(), g["mask_a"])
    assert np.array_equal(mfull.cpu().numpy(), (g["in_bg0"][:, 0] == 0).astype(np.float32))
    for k in ("pixel_ids_a", "pixel_ids_b", "region_ids_a", "region_ids_b"):
        assert np.array_equal(ops.strided_gather(G(g["in_" + k]), stride).cpu().numpy(), g[k]), k
    _, _, mds_b = ops.compose_mask(G(g["in_img_b"]), G(g["in_bg1"]), stride)
    assert np.array_equal(mds_b.reshape(b, -1).cpu().numpy(), g["mask_b"])
    # a3-a5 on the down-sampled maps
    iou, ioum = ops.corr_iou(G(g["region_ids_a"]), G(g["region_ids_b"]), G(g["mask_a"]), G(g["mask_b"]))
    assert np.array_equal(iou.cpu().numpy(), g["iou"]) and np.array_equal(ioum.cpu().numpy(), g["iou_masked"])
    iou, ioum = ops.corr_iou(G(g["pixel_ids_a"]), G(g["pixel_ids_b"]), G(g["mask_a"]), G(g["mask_b"]))
    assert np.array_equal(iou.cpu().numpy(), g["pixel_iou"]) and np.array_equal(ioum.cpu().numpy(), g["pixel_iou_masked"])


@pytest.mark.parametrize("name", ["cp2_b4_64_k64", "cp2_b4_96_k64_wrap_bg", "cp2_b3_80x112_k1024"])
def test_compose_pair_and_strided_iou_golden(golden_dir, name):
    """The training step's forms against the reference's goldens: both views composed in one launch (bit-exact images
    and down-sampled masks), and the IoUs read from the FULL-resolution id maps (the strided slices folded in)."""
    g = load(golden_dir, name)
    stride = int(g["cfg"][4])
    # the reference's local img_b is the key batch AFTER shuffle-BN (builder.py:1274): the golden holds it in that order,
    # which is what compose_pair writes when it is given the permutation
    perm = torch.argsort(torch.from_numpy(g["idx_unshuffle"])).to(DEV)
    out_a, out_b, md_a, md_b = ops.compose_pair(G(g["in_img_a"]), G(g["in_bg0"]), G(g["in_img_b"]), G(g["in_bg1"]), stride, perm)
    b = out_a.shape[0]
    assert np.array_equal(out_a.cpu().numpy(), g["img_a"]) and np.array_equal(out_b.cpu().numpy(), g["img_b"])
    assert np.array_equal(md_a.reshape(b, -1).cpu().numpy(), g["mask_a"]) and np.array_equal(md_b.reshape(b, -1).cpu().numpy(), g["mask_b"])
    for kind, iou_key, ioum_key in (("region_ids", "iou", "iou_masked"), ("pixel_ids", "pixel_iou", "pixel_iou_masked")):
        iou, ioum = ops.corr_iou_strided(G(g[f"in_{kind}_a"]), G(g[f"in_{kind}_b"]), stride, md_a.reshape(b, -1), md_b.reshape(b, -1))
        assert np.array_equal(iou.cpu().numpy(), g[iou_key]) and np.array_equal(ioum.cpu().numpy(), g[ioum_key]), kind


@pytest.mark.parametrize("channels_last", [False, True])
@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_compose_pair_layouts_shuffle_and_bf16(channels_last, dtype):
    """compose_pair at the bench shape: the key view written in shuffle order (out_b[j] = composed_b[perm[j]], reference
    builder.py:630), masks in the original order, channels-last memory, and bf16 = torch's round-to-nearest-even cast
    of the exact fp32 composition (what autocast hands the stem convolution)."""
    b, h, w, stride = 32, 224, 224, 16
    gen = torch.Generator().manual_seed(9)
    img_a, img_b = torch.rand(b, 3, h, w, generator=gen), torch.rand(b, 3, h, w, generator=gen)
    bg0, bg1 = synth_bg(b, h, w, gen), synth_bg(b, h, w, gen)
    img_a[0, 0, 0, 0] = float("inf")
    perm = torch.randperm(b, generator=gen)
    want_a, m_a = O.compose_mask(img_a, bg0)
    want_b, m_b = O.compose_mask(img_b, bg1)
    out_a, out_b, md_a, md_b = ops.compose_pair(img_a.to(DEV), bg0.to(DEV), img_b.to(DEV), bg1.to(DEV), stride, perm.to(DEV),
                                                channels_last, dtype)
    assert out_a.dtype == dtype and out_a.shape == (b, 3, h, w)
    assert out_a.is_contiguous(memory_format=torch.channels_last if channels_last else torch.contiguous_format)
    assert np.array_equal(out_a.float().cpu().numpy(), want_a.to(dtype).float().numpy(), equal_nan=True)
    assert np.array_equal(out_b.float().cpu().numpy(), O.shuffle_take(want_b, perm, 0, 1).to(dtype).float().numpy())
    assert torch.equal(md_a.cpu(), O.strided_gather(m_a, stride)) and torch.equal(md_b.cpu(), O.strided_gather(m_b, stride))
    with pytest.raises(Exception):
        ops.compose_pair(img_a[..., :222].contiguous().to(DEV), bg0[..., :222].contiguous().to(DEV),
                         img_b[..., :222].contiguous().to(DEV), bg1[..., :222].contiguous().to(DEV), stride)   # W % 4 != 0


def test_corr_iou_strided_equals_sliced_maps():
    gen = torch.Generator().manual_seed(3)
    for (b, h, w, s) in ((32, 224, 224, 16), (5, 50, 37, 8), (3, 64, 64, 32), (2, 20, 24, 1)):
        a = torch.randint(0, 40, (b, h, w), generator=gen)
        c = torch.randint(0, 40, (b, h, w), generator=gen)
        da, dc = O.strided_gather(a, s), O.strided_gather(c, s)
        p = da[0].numel()
        ma, mb = (torch.rand(b, p, generator=gen) > 0.4).float(), (torch.rand(b, p, generator=gen) > 0.5).float()
        iou, ioum = ops.corr_iou_strided(a.to(DEV), c.to(DEV), s, ma.to(DEV), mb.to(DEV))
        ref_iou, ref_ioum = ops.corr_iou(da.contiguous().to(DEV), dc.contiguous().to(DEV), ma.to(DEV), mb.to(DEV))
        assert torch.equal(iou, ref_iou) and torch.equal(torch.isnan(ioum), torch.isnan(ref_ioum))
        assert torch.equal(torch.nan_to_num(ioum, nan=-1.0), torch.nan_to_num(ref_ioum, nan=-1.0))
        assert torch.equal(iou.cpu(), O.masked_iou(da.reshape(b, -1), dc.reshape(b, -1), torch.ones(b, p), torch.ones(b, p)))


@pytest.mark.parametrize("shape,stride", [((32, 224, 224), 16), ((3, 50, 37), 8), ((2, 17, 21), 1), ((2, 64, 66), 32)])
def test_compose_full_size_vs_oracle(shape, stride):
    b, h, w = shape
    gen = torch.Generator().manual_seed(5)
    img, bg = torch.rand(b, 3, h, w, generator=gen), synth_bg(b, h, w, gen)
    img[0, 0, 0, 0] = float("inf")           # inf * 0 must stay NaN exactly as in torch
    want, mask = O.compose_mask(img, bg)
    out, mfull, mds = ops.compose_mask(img.to(DEV), bg.to(DEV), stride, want_full_mask=True)
    assert np.array_equal(out.cpu().numpy(), want.numpy(), equal_nan=True)
    assert torch.equal(mfull.cpu(), mask)
    assert torch.equal(mds.cpu(), O.strided_gather(mask, stride))
    ids = torch.randint(0, 1 << 40, (b, h, w), generator=gen)
    assert torch.equal(ops.strided_gather(ids.to(DEV), stride).cpu(), O.strided_gather(ids, stride))


@pytest.mark.parametrize("tag", ["unique", "shared", "random"])
def test_corr_iou_kats(golden_dir, tag):
    g = load(golden_dir, "corrmap_kats")
    a, b = torch.from_numpy(g[f"{tag}_map_a"]).long(), torch.from_numpy(g[f"{tag}_map_b"]).long()
    iou, ioum = ops.corr_iou(a.to(DEV), b.to(DEV), G(g[f"{tag}_mask_a"]), G(g[f"{tag}_mask_b"]))
    assert np.array_equal(iou.cpu().numpy(), g[f"{tag}_iou"])
    assert np.array_equal(ioum.cpu().numpy(), g[f"{tag}_iou_masked"])
    if tag == "unique":
        assert torch.equal(iou.cpu(), torch.ones(4) * (12 / 38)) and torch.equal(ioum.cpu(), torch.ones(4) / 3)
    if tag == "shared":
        assert torch.equal(iou.cpu(), torch.tensor([4 / 7])) and torch.equal(ioum.cpu(), torch.tensor([2 / 3]))


# P <= 4096: keys counted in an LDS hash table (round 3: up to 2047); above: bitonic sort (both forms, the boundary between
# them, and the old boundary); the last case has ids below -1 (keys of either sign)
@pytest.mark.parametrize("B,P,hi,lo", [(32, 196, 60000, 0), (8, 4096, 5000, 0), (5, 1, 3, 0), (4, 1024, 1 << 30, 0), (2, 16383, 100, 0),
                                       (3, 2047, 900, 0), (3, 2048, 900, 0), (6, 196, 4, 0), (3, 4097, 3000, 0), (3, 8192, 3000, 0),
                                       (8, 4096, 1 << 40, 0), (4, 300, 40, -40)])
def test_corr_iou_random_vs_oracle(B, P, hi, lo):
    gen = torch.Generator().manual_seed(P)
    a = torch.randint(lo, hi, (B, P), generator=gen)
    b = torch.randint(lo, hi, (B, P), generator=gen)
    ma = (torch.rand(B, P, generator=gen) > 0.3).float()
    mb = (torch.rand(B, P, generator=gen) > 0.5).float()
    ma[0] = 0
    mb[0] = 0                                  # empty masks -> 0/0 -> NaN (reference raises there)
    iou, ioum = ops.corr_iou(a.to(DEV), b.to(DEV), ma.to(DEV), mb.to(DEV))
    assert np.array_equal(iou.cpu().numpy(), O.masked_iou(a, b, torch.ones_like(ma), torch.ones_like(mb)).numpy())
    assert np.array_equal(ioum.cpu().numpy(), O.masked_iou(a, b, ma, mb).numpy(), equal_nan=True)
    assert torch.isnan(ioum[0])


def test_ema_golden_and_full_size(golden_dir):
    g = load(golden_dir, "queue_ema_shuffle")
    n = len([k for k in g if k.startswith("ema_q_")])
    pk = [G(g[f"ema_k0_{i}"]) for i in range(n)]
    pq = [G(g[f"ema_q_{i}"]) for i in range(n)]
    plan = ops.EmaMultiPlan(pk, pq)
    for rnd in (1, 2):
        plan.run(float(g["ema_m"]))
        for i in range(n):
            assert np.array_equal(pk[i].cpu().numpy(), g[f"ema_k{rnd}_{i}"]), (rnd, i)
    # flat form at encoder scale (66 M floats + a ragged tail), against the oracle
    gen = torch.Generator().manual_seed(0)
    N = 66_000_003
    k, q = torch.randn(N, generator=gen), torch.randn(N, generator=gen)
    want = O.momentum_update([k], [q], 0.999)[0]
    kd, qd = k.to(DEV), q.to(DEV)
    ops.ema_flat(kd, qd, 0.999)
    assert torch.equal(kd.cpu(), want)
    # multi-tensor form with ragged, unaligned views
    base_k, base_q = torch.randn(300_001, generator=gen).to(DEV), torch.randn(300_001, generator=gen).to(DEV)
    cuts = [0, 1, 66, 70_001, 200_000, 300_001]
    vk = [base_k[a:b] for a, b in zip(cuts[:-1], cuts[1:])]
    vq = [base_q[a:b] for a, b in zip(cuts[:-1], cuts[1:])]
    want = O.momentum_update([base_k.cpu()], [base_q.cpu()], 0.99)[0]
    ops.EmaMultiPlan(vk, vq).run(0.99)
    assert torch.equal(base_k.cpu(), want)


def test_enqueue_golden_and_wrap(golden_dir):
    g = load(golden_dir, "queue_ema_shuffle")
    for tag in ("plain", "wrap", "exact", "big"):
        queue, keys = G(g[f"enq_{tag}_queue_before"]), G(g[f"enq_{tag}_keys"])
        ptr = torch.tensor([int(g[f"enq_{tag}_ptr_before"])], dtype=torch.long, device=DEV)
        ops.enqueue(queue, keys, ptr)
        assert np.array_equal(queue.cpu().numpy(), g[f"enq_{tag}_queue_after"]), tag
        assert int(ptr) == int(g[f"enq_{tag}_ptr_after"]), tag
    # BASELINE size: K=65536, 256 keys per step, many steps across the wrap point
    gen = torch.Generator().manual_seed(1)
    K, C, n = 65536, 128, 256
    q_cpu = torch.randn(C, K, generator=gen)
    queue, ptr, p = q_cpu.to(DEV), torch.tensor([K - 3 * n - 7], dtype=torch.long, device=DEV), K - 3 * n - 7
    for _ in range(5):
        keys = torch.randn(n, C, generator=gen)
        q_cpu, p = O.dequeue_and_enqueue(q_cpu, p, keys)
        ops.enqueue(queue, keys.to(DEV), ptr)
    assert torch.equal(queue.cpu(), q_cpu) and int(ptr) == p


def test_gather_rows_shuffle(golden_dir):
    g = load(golden_dir, "queue_ema_shuffle")
    x, perm = G(g["shuf_x"]), torch.from_numpy(g["shuf_perm"]).to(DEV)
    assert np.array_equal(ops.gather_rows(x, perm).cpu().numpy(), g["shuf_out"])
    back = ops.gather_rows(G(g["shuf_out"]), torch.from_numpy(g["shuf_idx_unshuffle"]).to(DEV))
    assert np.array_equal(back.cpu().numpy(), g["shuf_x"])
    gen = torch.Generator().manual_seed(2)
    big = torch.rand(64, 3, 224, 224, generator=gen)
    idx = torch.randperm(64, generator=gen)[:32]
    assert torch.equal(ops.gather_rows(big.to(DEV), idx.to(DEV)).cpu(), big[idx])
    odd = torch.rand(9, 7, 5, generator=gen)
    assert torch.equal(ops.gather_rows(odd.to(DEV), torch.tensor([8, 0, 3], device=DEV)).cpu(), odd[[8, 0, 3]])
    flag = torch.zeros(1, dtype=torch.int32, device=DEV)
    ops.gather_rows(odd.to(DEV), torch.tensor([9], device=DEV), flag)
    assert int(flag) == 1
