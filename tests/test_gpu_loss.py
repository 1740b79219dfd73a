"""GPU: the fp32 MFMA loss kernels through the C ABI against the CPU oracle and the
golden vectors recorded from the reference.

Tolerances (stated per north_star: fp32 logits within 1e-4 of the reference):
  raw logits / unit vectors      |err| <= 2e-6   (f32 MFMA = exact fp32 fma chain)
  losses (scalars, O(1..10))     |err| <= 2e-5
  gradient d loss / d q_feat     max|err| <= 2e-5 * max|grad| + 1e-9
"""
import os

import numpy as np
import pytest
import torch

from cp2_amd import functional as CF
from cp2_amd import ops
from oracle import cp2_oracle as O

pytestmark = pytest.mark.gpu
DEV = "cuda"
CP2_CASES = ["cp2_b4_64_k64", "cp2_b4_96_k64_wrap_bg", "cp2_b3_80x112_k1024", "cp2_proposed_weights"]


def load(golden_dir, name):
    return dict(np.load(os.path.join(golden_dir, name + ".npz")))


def G(x):
    return torch.from_numpy(np.ascontiguousarray(x)).to(DEV)


def assert_close(got, want, atol, rtol=0.0, what=""):
    got = torch.as_tensor(got).detach().double().cpu()
    want = torch.as_tensor(want).detach().double().cpu()
    assert got.shape == want.shape, (what, got.shape, want.shape)
    assert torch.equal(torch.isnan(got), torch.isnan(want)), f"{what}: NaN pattern differs"
    got, want = torch.nan_to_num(got, nan=0.0), torch.nan_to_num(want, nan=0.0)
    err = (got - want).abs().max().item() if got.numel() else 0.0
    lim = atol + rtol * want.abs().max().item()
    assert err <= lim, f"{what}: max err {err:.3e} > {lim:.3e}"


def grad_close(got, want, what="grad"):
    assert_close(got, want, 1e-9, 2e-5, what)


@pytest.mark.parametrize("name", CP2_CASES)
def test_loss_section_golden(golden_dir, name):
    g = load(golden_dir, name)
    b, h, w, K, stride, inc_bg = [int(v) for v in g["cfg"]]
    tg, tl, lmbd, wp, wr, wn, _ = [float(v) for v in g["cfg_f"]]
    q = G(g["q_feat"]).requires_grad_(True)
    ids = None
    if name == "cp2_proposed_weights":
        ids = tuple(G(g[k]).reshape(b, -1) for k in ("pixel_ids_a", "pixel_ids_b", "region_ids_a", "region_ids_b"))
    out = CF.cp2_loss_section(q, G(g["k_feat"]), G(g["mask_a"]), G(g["mask_b"]), G(g["queue_before"]),
                              temp_global=tg, temp_local=tl, lmbd_dense=lmbd, include_background=bool(inc_bg),
                              ids=ids, weights=(wp, wr, wn), want_lneg=True, want_quartiles=True)
    out.loss.backward()
    assert_close(out.q_pos, g["q_pos"], 2e-6, what="q_pos")
    assert_close(out.k_pos, g["k_pos"], 2e-6, what="k_pos")
    assert_close(out.instance_pos, g["l_pos"][:, 0], 2e-6, what="l_pos")
    assert_close(out.lneg, g["l_neg"], 2e-6, what="l_neg")
    assert_close(out.loss_instance, g["loss_instance"], 2e-5, what="loss_instance")
    assert_close(out.loss_dense, g["loss_dense"], 2e-5, what="loss_dense")
    assert_close(out.loss, g["loss"], 2e-5, what="loss")
    grad_close(q.grad, g["dq_feat"], "dq_feat")
    assert_close(out.acc_dense, g["acc_dense"], 1e-4, what="acc_dense")
    assert_close(out.acc1, g["acc1"], 1e-4, what="acc1")
    assert_close(out.acc5, g["acc5"], 1e-4, what="acc5")
    assert_close(out.dense_sample[:, 3], g["dense_positive_average"], 2e-6, what="dense +mean")
    assert_close(out.dense_sample[:, 4], g["dense_negative_average"], 2e-6, what="dense -mean")
    # a15 quartiles (reference: torch.nanquantile / torch.quantile on sorted data)
    assert_close(out.dense_pos_quartiles, g["dense_positive_quartiles"], 3e-6, what="dense + quartiles")
    assert_close(out.dense_neg_quartiles, g["dense_negative_quartiles"], 3e-6, what="dense - quartiles")
    assert_close(out.instance_neg_quartiles, g["instance_negative_quartiles"], 3e-6, what="instance quartiles")
    assert_close(out.instance_neg_mean, g["instance_average_negative_scores"], 3e-6, what="instance - mean")


@pytest.mark.parametrize("nt", ["none", "fixed", "average", "median", "hard"])
def test_negative_type_golden(golden_dir, nt):
    """NegativeType reshaping of the negative dense logits (reference builder.py:1332-1386) on the reference's recorded
    encoder outputs: loss, dense loss, gradient and the logging statistics (which use the raw scores)."""
    g = load(golden_dir, "cp2_neg_" + nt)
    b = int(g["cfg"][0])
    tg, tl, lmbd, wp, wr, wn, _ = [float(v) for v in g["cfg_f"]]
    ids = tuple(G(g[k]).reshape(b, -1) for k in ("pixel_ids_a", "pixel_ids_b", "region_ids_a", "region_ids_b"))
    q = G(g["q_feat"]).requires_grad_(True)
    out = CF.cp2_loss_section(q, G(g["k_feat"]), G(g["mask_a"]), G(g["mask_b"]), G(g["queue_before"]), temp_global=tg,
                              temp_local=tl, lmbd_dense=lmbd, ids=ids, weights=(wp, wr, wn), want_quartiles=True,
                              negative_type=int(g["negative"][0]), negative_scale=float(g["negative"][1]))
    out.loss.backward()
    assert_close(out.loss_dense, g["loss_dense"], 2e-5, what="loss_dense")
    assert_close(out.loss_instance, g["loss_instance"], 2e-5, what="loss_instance")
    assert_close(out.loss, g["loss"], 2e-5, what="loss")
    grad_close(q.grad, g["dq_feat"], "dq_feat")
    assert_close(out.acc_dense, g["acc_dense"], 1e-4, what="acc_dense")
    assert_close(out.dense_sample[:, 4], g["dense_negative_average"], 2e-6, what="dense -mean")
    assert_close(out.dense_neg_quartiles, g["dense_negative_quartiles"], 3e-6, what="dense - quartiles")


def test_feat_kernels_vs_oracle_and_channels_last():
    gen = torch.Generator().manual_seed(0)
    B, C, h, w = 5, 128, 9, 13
    feat = torch.randn(B, C, h, w, generator=gen)
    feat[1, :, 2, 3] = 0.0                              # a zero vector: clamp path of F.normalize
    mask = (torch.rand(B, h * w, generator=gen) > 0.4).float()
    mask[2] = 0.0                                       # empty foreground -> pooled vector is zero
    d_ref, pos_ref, neg_ref = O.normalize_and_pool(feat, mask)
    for fmt in (torch.contiguous_format, torch.channels_last):
        fd = feat.to(DEV).contiguous(memory_format=fmt)
        dense, inv, part = ops.feat_normalize_pool(fd, mask.to(DEV))
        assert_close(dense, d_ref, 1e-6, what="dense")
        q_pos, q_neg, q_norms, k_pos, k_neg, extras = ops.pool_finalize(part, part, h * w)
        assert_close(q_pos, pos_ref, 2e-6, what="pos")
        assert_close(q_neg, neg_ref, 2e-6, what="neg")
        assert_close(k_pos, pos_ref, 2e-6, what="kpos")
        assert_close(extras[:, 1], (pos_ref * neg_ref).sum(1), 3e-6, what="extras")


@pytest.mark.parametrize("fmt", [torch.contiguous_format, torch.channels_last])
def test_feat_pair_with_row_index_equals_two_single_launches(fmt):
    """cp2_feat_normalize_pool_pair (query + key in one launch, key rows through the un-shuffle index, channels-last maps
    staged through LDS) == the two single-map launches on a gathered key map, bit for bit; P = 196 (a ragged last tile)."""
    gen = torch.Generator().manual_seed(1)
    B, C, h, w = 7, 128, 14, 14
    qf = torch.randn(B, C, h, w, generator=gen).to(DEV).contiguous(memory_format=fmt)
    kf = torch.randn(B, C, h, w, generator=gen).to(DEV).contiguous(memory_format=fmt)
    ma = (torch.rand(B, h * w, generator=gen) > 0.4).float().to(DEV)
    mb = (torch.rand(B, h * w, generator=gen) > 0.6).float().to(DEV)
    perm = torch.randperm(B, generator=gen).to(DEV)
    qd, kd, inv, qp, kp = ops.feat_normalize_pool_pair(qf, kf, ma, mb, perm)
    rq = ops.feat_normalize_pool(qf, ma)
    rk = ops.feat_normalize_pool(kf[perm].contiguous(memory_format=fmt), mb)
    assert torch.equal(qd, rq[0]) and torch.equal(inv, rq[1]) and torch.equal(qp, rq[2])
    assert torch.equal(kd, rk[0]) and torch.equal(kp, rk[2])
    qd2, kd2, _, _, kp2 = ops.feat_normalize_pool_pair(qf, kf, ma, mb, None)
    rk2 = ops.feat_normalize_pool(kf, mb)
    assert torch.equal(kd2, rk2[0]) and torch.equal(kp2, rk2[2]) and torch.equal(qd2, qd)


@pytest.mark.parametrize("fmt,inc_bg,S", [(torch.contiguous_format, False, 1), (torch.channels_last, True, 4), (torch.channels_last, False, 3)])
def test_feat_bwd_fused_equals_separate_launches(fmt, inc_bg, S):
    """cp2_feat_bwd_fused == dense_grad_sum (split order) + cp2_pool_bwd + cp2_feat_bwd, bit for bit."""
    gen = torch.Generator().manual_seed(2)
    B, C, h, w = 6, 128, 14, 14
    P = h * w
    feat = torch.randn(B, C, h, w, generator=gen).to(DEV).contiguous(memory_format=fmt)
    mask = (torch.rand(B, P, generator=gen) > 0.4).float().to(DEV)
    dense, inv, part = ops.feat_normalize_pool(feat, mask)
    q_pos, q_neg, q_norms, k_pos, k_neg, _ = ops.pool_finalize(part, part.flip(0).contiguous(), P)
    g_part = torch.randn(S, B, C, P, generator=gen).to(DEV)
    drows = torch.randn(B, C, generator=gen).to(DEV)
    NE = 3 if inc_bg else 1
    dE = torch.randn(B, NE, generator=gen).to(DEV)
    got = ops.feat_bwd_fused(dense, inv, mask, g_part if S > 1 else g_part[0].contiguous(), S, drows, dE, q_pos, q_neg, k_pos, k_neg,
                             q_norms, inc_bg, feat)
    g = g_part[0].clone()
    for sidx in range(1, S):
        g = g + g_part[sidx]
    dE3 = torch.zeros(B, 3, device=DEV)
    dE3[:, :NE] = dE
    ds_pos, ds_neg = ops.pool_bwd(drows, dE3, q_pos, q_neg, k_pos, k_neg, q_norms, inc_bg)
    want = ops.feat_bwd(dense, inv, mask, g, ds_pos, ds_neg, feat)
    assert got.stride() == feat.stride() and torch.equal(got, want)


def test_step_scalars_vs_torch():
    """Every returned / logged scalar of the step from one launch, against the torch expressions round 2 ran
    (reference builder.py:1431-1448, 1265, 1282, 1553-1604)."""
    gen = torch.Generator().manual_seed(4)
    B, C = 32, 128
    ins_loss = torch.rand((), generator=gen) * 8
    cnt = torch.randint(0, 9, (B,), generator=gen, dtype=torch.int32)
    extras = torch.randn(B, 3, generator=gen)
    sample = torch.rand(B, 8, generator=gen)
    sample[:, 5] = (sample[:, 5] > 0.5).float()
    q_pos = torch.nn.functional.normalize(torch.randn(B, C, generator=gen), dim=1)
    k_pos = torch.nn.functional.normalize(torch.randn(B, C, generator=gen), dim=1)
    quart = [torch.randn(3, B, generator=gen) for _ in range(3)]
    lmean = torch.randn(B, generator=gen)
    lam = 0.2
    d = lambda t: t.to(DEV)  # noqa: E731
    out = ops.step_scalars(d(ins_loss), d(cnt), d(extras), d(sample), d(q_pos), d(k_pos), lam, d(quart[0]), d(quart[1]), d(quart[2]),
                           d(lmean)).cpu()
    l_den = sample[:, 2].mean()
    want = {0: ins_loss + l_den * lam, 1: ins_loss, 2: l_den, 3: (cnt < 1).float().mean() * 100, 4: (cnt < 5).float().mean() * 100,
            5: sample[:, 5].mean() * 100, 6: sample[:, 3].mean(), 7: sample[:, 4].mean(), 8: extras[:, 0].mean(),
            9: q_pos.std(0).mean(), 10: k_pos.std(0).mean(), 20: lmean.mean()}
    for j in range(3):
        for k in range(3):
            want[11 + 3 * j + k] = quart[j][k].mean()
    for i, v in want.items():
        assert abs(float(out[i]) - float(v)) <= 2e-6 * max(1.0, abs(float(v))), (i, float(out[i]), float(v))
    # NaN in a sample's loss propagates as torch's mean does; absent quartiles read as zero
    sample[3, 2] = float("nan")
    out2 = ops.step_scalars(d(ins_loss), d(cnt), d(extras[:, :1].contiguous()), d(sample), d(q_pos), d(k_pos), lam).cpu()
    assert torch.isnan(out2[0]) and torch.isnan(out2[2]) and float(out2[11]) == 0.0 and float(out2[20]) == 0.0
    assert abs(float(out2[8]) - float(extras[:, 0].mean())) <= 2e-6


def _rand_case(B, hw, K, seed, stride_fmt=torch.contiguous_format):
    gen = torch.Generator().manual_seed(seed)
    h, w = hw
    q = torch.randn(B, 128, h, w, generator=gen)
    k = torch.randn(B, 128, h, w, generator=gen)
    ma = (torch.rand(B, h * w, generator=gen) > 0.45).float()
    mb = (torch.rand(B, h * w, generator=gen) > 0.55).float()
    queue = torch.nn.functional.normalize(torch.randn(128, K, generator=gen), dim=0)
    return q, k, ma, mb, queue


def _oracle_loss(q, k, ma, mb, queue, tg, tl, lmbd, inc_bg):
    q = q.clone().requires_grad_(True)
    qd, qp, qn = O.normalize_and_pool(q, ma)
    with torch.no_grad():
        kd, kp, kn = O.normalize_and_pool(k, mb)
    li, logits, _, _ = O.instance_infonce(qp, kp, queue, tg, qn, kn, inc_bg)
    ld, per_sample, _ = O.dense_infonce(O.dense_logits(qd, kd), ma, mb, tl)
    loss = li + lmbd * ld
    loss.backward()
    return loss.detach(), li.detach(), ld.detach(), q.grad, per_sample.detach()


@pytest.mark.parametrize("B,hw,K,inc_bg,fmt", [
    (32, (14, 14), 65536, False, torch.channels_last),     # BASELINE config 2 shapes
    (3, (5, 7), 300, True, torch.contiguous_format),       # ragged everything, K not a tile multiple
    (40, (2, 2), 1000, False, torch.contiguous_format),    # 2 row tiles in the instance kernel
    (70, (3, 3), 4096, True, torch.contiguous_format),     # > 64 rows
    (2, (32, 32), 2048, False, torch.channels_last),       # P = 1024 (config 4 at OS16)
    (1, (12, 11), 64, False, torch.contiguous_format),     # P = 132: partial 32-tile and partial 64-tile
    (20, (2, 2), 1001, True, torch.contiguous_format),     # K % 4 != 0: the register-staged one-row-tile kernel
    (32, (3, 3), 131072, False, torch.contiguous_format),  # config-4 queue: two sub-tiles per wave in the LDS-DMA kernel
    (7, (4, 4), 2052, True, torch.contiguous_format),      # LDS-DMA kernel with a ragged last tile (2052 = 8 * 256 + 4)
])
def test_loss_section_random_vs_oracle(B, hw, K, inc_bg, fmt):
    q, k, ma, mb, queue = _rand_case(B, hw, K, seed=B * 1000 + K)
    tg, tl, lmbd = 0.2, 0.7, 0.35
    loss, li, ld, dq, per_sample = _oracle_loss(q, k, ma, mb, queue, tg, tl, lmbd, inc_bg)
    qg = q.to(DEV).contiguous(memory_format=fmt).requires_grad_(True)
    out = CF.cp2_loss_section(qg, k.to(DEV).contiguous(memory_format=fmt), ma.to(DEV), mb.to(DEV), queue.to(DEV),
                              temp_global=tg, temp_local=tl, lmbd_dense=lmbd, include_background=inc_bg)
    (out.loss * 1.5).backward()                              # non-unit upstream gradient
    assert_close(out.loss_instance, li, 2e-5, what="loss_instance")
    assert_close(out.loss_dense, ld, 2e-5, what="loss_dense")
    assert_close(out.dense_sample[:, 2], per_sample, 2e-5, what="loss_dense per sample")
    assert_close(out.loss, loss, 2e-5, what="loss")
    grad_close(qg.grad, dq * 1.5, "dq_feat")
    assert qg.grad.stride() == qg.stride()


def test_empty_mask_gives_nan_like_reference():
    q, k, ma, mb, queue = _rand_case(2, (4, 4), 128, seed=9)
    mb[1] = 0.0
    _, _, ld, _, _ = _oracle_loss(q, k, ma, mb, queue, 0.2, 1.0, 0.2, False)
    out = CF.cp2_loss_section(q.to(DEV), k.to(DEV), ma.to(DEV), mb.to(DEV), queue.to(DEV))
    assert torch.isnan(ld) and torch.isnan(out.loss_dense)
    assert not torch.isnan(out.dense_sample[0, 2])


@pytest.mark.parametrize("name", ["densecl_b2_128_k64", "densecl_b2_96_k64_coord"])
def test_rowkey_densecl_golden(golden_dir, name):
    """T19: per-pixel rows against queue2 (reference builder.py:866-873,906-908)."""
    g = load(golden_dir, name)
    tl = float(g["cfg_f"][1])
    ql = torch.from_numpy(g["q_local"]).clone().requires_grad_(True)          # [b, C, S2]
    pos = torch.from_numpy(g["pos_local"]).reshape(-1).clone().requires_grad_(True)
    b, C, S2 = ql.shape
    loss_ref = O.queue_infonce(ql.permute(0, 2, 1).reshape(-1, C), pos, torch.from_numpy(g["queue2_before"]), tl)
    loss_ref.backward()
    assert_close(loss_ref, g["loss_local"], 2e-6)
    qd = ql.detach().to(DEV)
    R = b * S2
    res = ops.rowkey_infonce(qd, (S2, C * S2, 1, S2), R, G(g["queue2_before"]), pos.detach().reshape(-1, 1).to(DEV),
                             tl, grad_scale=1.0 / R, want_lneg=True)
    assert_close(res.lnegT.t(), g["neg_local"], 2e-6, what="neg_local")
    assert_close(res.loss, g["loss_local"], 2e-5, what="loss_local")
    grad_close(res.drows, ql.grad, "d q_local")
    grad_close(res.dE[:, 0], pos.grad, "d pos")


def test_rowkey_large_split_property():
    """BASELINE config-5 shape (6272 rows x 65536 keys) is too big for the CPU oracle in a
    unit test: check it through size-independent properties instead --
    (1) a 640-row x 8192-key slice against the oracle, (2) logsumexp over the whole queue
    equals logaddexp of the two half queues, (3) permuting keys leaves loss and grads unchanged."""
    gen = torch.Generator().manual_seed(3)
    b, C, S2, K = 32, 128, 196, 65536
    rows = torch.nn.functional.normalize(torch.randn(b, C, S2, generator=gen), dim=1).to(DEV)
    queue = torch.nn.functional.normalize(torch.randn(C, K, generator=gen), dim=0).to(DEV)
    pos = (torch.rand(b * S2, 1, generator=gen) * 2 - 1).to(DEV)
    R, lay, T = b * S2, (S2, C * S2, 1, S2), 0.2
    full = ops.rowkey_infonce(rows, lay, R, queue, pos, T, grad_scale=1.0 / R)
    none = torch.full_like(pos, -1e30)                          # an extra logit that contributes exp(-inf) = 0
    h1 = ops.rowkey_infonce(rows, lay, R, queue[:, :K // 2].contiguous(), none, T, None)
    h2 = ops.rowkey_infonce(rows, lay, R, queue[:, K // 2:].contiguous(), pos, T, None)
    assert_close(torch.logaddexp(h1.lse, h2.lse), full.lse, 2e-5, what="lse split")
    perm = torch.randperm(K, generator=gen).to(DEV)
    pq = ops.rowkey_infonce(rows, lay, R, queue[:, perm].contiguous(), pos, T, grad_scale=1.0 / R)
    assert_close(pq.loss, full.loss, 2e-5, what="loss under key permutation")
    grad_close(pq.drows, full.drows, "grad under key permutation")
    # slice against the oracle
    rs, ks = 640, 8192
    sub_rows = rows.reshape(b, C, S2)[:4].contiguous()           # 4 samples = 784 rows; use the first 640
    r_cpu = sub_rows.cpu().permute(0, 2, 1).reshape(-1, C)[:rs].clone().requires_grad_(True)
    p_cpu = pos[:rs, 0].cpu().clone().requires_grad_(True)
    want = O.queue_infonce(r_cpu, p_cpu, queue[:, :ks].cpu(), T)
    want.backward()
    got = ops.rowkey_infonce(sub_rows, lay, rs, queue[:, :ks].contiguous(), pos[:rs].contiguous(), T, grad_scale=1.0 / rs)
    assert_close(got.loss, want.detach(), 2e-5, what="slice loss")
    got_rows = got.drows.permute(0, 2, 1).reshape(-1, C)[:rs]
    grad_close(got_rows, r_cpu.grad, "slice d rows")
    grad_close(got.dE[:, 0], p_cpu.grad, "slice d pos")


def test_masked_quantiles_bit_exact_vs_torch():
    """Same input -> the radix-select quantiles equal torch.quantile / torch.nanquantile bit for bit
    (the reference's convention, tests/test_contrastive_metrics.py:50-57): the one-launch row kernel for rows up to 131072
    elements, the chunked six-launch form above (with the positive / negative classes of one logit map served by one read
    per level), plus the row means torch's x.mean(1) gives."""
    _masked_quantiles_cases()
    gen = torch.Generator().manual_seed(8)
    x = torch.randn(6, 70_001, generator=gen)
    x[2, 17] = float("nan")
    means = torch.empty(6, device=DEV)
    q = ops.masked_quantiles_multi([dict(x=x.to(DEV), stride_row=70_001, stride_elem=1, R=6, N=70_001, mean_out=means)])[0]
    assert np.array_equal(q.cpu().numpy(), torch.nanquantile(x, torch.tensor([0.25, 0.5, 0.75]), dim=1).numpy(), equal_nan=True)
    want = x.double().mean(1).float()
    assert torch.isnan(means[2]) and (means.cpu()[[0, 1, 3, 4, 5]] - want[[0, 1, 3, 4, 5]]).abs().max() <= 1e-6


def _masked_quantiles_cases():
    kat = torch.tensor([[1., 2, 3, 4, 5, 6], [1, 2, 3, 7, 8, 9]])
    got = ops.masked_quantiles(kat.to(DEV), 6, 1, 2, 6)
    assert torch.equal(got.cpu(), torch.tensor([[2.25, 2.25], [3.5, 5.0], [4.75, 7.75]]))
    gen = torch.Generator().manual_seed(0)
    for R, N in ((32, 65536), (5, 1000), (3, 7), (4, 1)):
        x = torch.randn(R, N, generator=gen)
        x[0, : N // 2] = 0.5                                    # heavy duplicates
        if N > 4:
            x[1, ::3] = float("nan")
            x[1, 1] = -0.0
        for qs in (torch.tensor([0.25, 0.5, 0.75]), torch.tensor([0.0, 0.1, 0.999, 1.0])):   # up to 4 per call
            want = torch.nanquantile(x, qs, dim=1)
            got = ops.masked_quantiles(x.to(DEV), N, 1, R, N, q=qs.to(DEV))
            assert torch.equal(got.cpu(), want), (R, N)
            xt = x.t().contiguous()                             # key-major storage, as lnegT
            got_t = ops.masked_quantiles(xt.to(DEV), 1, R, R, N, q=qs.to(DEV))
            assert torch.equal(got_t.cpu(), want), (R, N, "strided")
    # chunked rows: many chunks per row, all-equal rows, a narrow band (every element in one first-level bin: the
    # candidate lists overflow and the select re-reads the row), +-inf, and a workspace re-used by consecutive calls
    # (rows above 131072 elements take the chunked three-launch path, the others the one-launch row kernel)
    for R, N, kind in ((2, 3_000_000, "normal"), (3, 270_001, "const"), (3, 70_001, "const"), (2, 200_003, "band"),
                       (2, 100_000, "band"), (2, 150_000, "inf"), (2, 50_000, "inf"),
                       # rows the row kernel holds in registers (<= 65536 elements): one- and two-bin bands (every lane of a
                       # wavefront adds to the same histogram bin), a band straddling 1.0, constants, a ragged tail
                       (2, 60_000, "band"), (2, 40_000, "band"), (2, 65_536, "straddle"), (3, 30_001, "const"), (2, 40_961, "normal")):
        x = torch.randn(R, N, generator=gen) * 0.09
        if kind == "const":
            x[:] = -0.25
            x[2, 5] = 1.0
        elif kind == "band":
            x = 0.5 + torch.rand(R, N, generator=gen) * 1e-4
        elif kind == "straddle":
            x = 1.0 + torch.randn(R, N, generator=gen) * 1e-6
        elif kind == "inf":
            x[0, ::7] = float("inf")
            x[1, ::5] = float("-inf")
        qs = torch.tensor([0.25, 0.5, 0.75])
        want = torch.nanquantile(x, qs, dim=1)
        xd = x.to(DEV)
        for _ in range(2):
            got = ops.masked_quantiles(xd, N, 1, R, N, q=qs.to(DEV))
            assert np.array_equal(got.cpu().numpy(), want.numpy(), equal_nan=True), (R, N, kind)
    # padded rows (row stride > N, 16-byte aligned): the register-resident path with a ragged last 16-byte load, 10 and 16 loads
    for N, stride in ((40_961, 40_964), (65_533, 65_536), (5, 8)):
        xp = torch.randn(3, stride, generator=gen)
        xp[1, 3] = float("nan")
        qs = torch.tensor([0.25, 0.5, 0.75])
        got = ops.masked_quantiles(xp.to(DEV), stride, 1, 3, N, q=qs.to(DEV))
        assert np.array_equal(got.cpu().numpy(), torch.nanquantile(xp[:, :N], qs, dim=1).numpy(), equal_nan=True), (N, stride)
    # masked form against the oracle's dense statistics
    for B, P in ((2, 420), (2, 390), (2, 260), (2, 240), (2, 196), (2, 130)):   # P % 4 == 0 and != 0; chunked (P*P > 131072), row kernel re-reading (260), registers (240: 16 loads, 196 / 130: 10)
        logits = torch.randn(B, P, P, generator=gen) * 0.1
        ma = (torch.rand(B, P, generator=gen) > 0.4).float()
        mb = (torch.rand(B, P, generator=gen) > 0.5).float()
        st = O.dense_loss_stats(logits, ma[:, :, None] * mb[:, None, :])
        jobs = [dict(x=logits.to(DEV), stride_row=P * P, stride_elem=1, R=B, N=P * P, mask_a=ma.to(DEV), mask_b=mb.to(DEV), want=w)
                for w in (1, 0)]
        pos, neg = ops.masked_quantiles_multi(jobs)
        assert np.array_equal(pos.cpu().numpy(), st["positive"]["quartiles"].numpy(), equal_nan=True)
        assert np.array_equal(neg.cpu().numpy(), st["negative"]["quartiles"].numpy(), equal_nan=True)
    # chunked form, pairing rules: (want=1, want=0) over the same map = one read per level; the reverse order, a lone class
    # and a pair next to an unmasked long-row job take the unpaired path / mix both -- all must agree with the oracle
    B, P = 2, 400
    logits = torch.randn(B, P, P, generator=gen) * 0.05 + 0.8
    ma = (torch.rand(B, P, generator=gen) > 0.3).float()
    mb = (torch.rand(B, P, generator=gen) > 0.6).float()
    st = O.dense_loss_stats(logits, ma[:, :, None] * mb[:, None, :])
    lneg = torch.randn(3, 140_000, generator=gen)
    wq = torch.quantile(lneg, torch.tensor([0.25, 0.5, 0.75]), dim=1)
    job = lambda w: dict(x=logits.to(DEV), stride_row=P * P, stride_elem=1, R=B, N=P * P, mask_a=ma.to(DEV), mask_b=mb.to(DEV), want=w)  # noqa: E731
    long_job = dict(x=lneg.to(DEV), stride_row=140_000, stride_elem=1, R=3, N=140_000)
    for order in ((1, 0), (0, 1), (1,), (0,), (1, 0, "long"), ("long", 1, 0)):
        outs = ops.masked_quantiles_multi([long_job if w == "long" else job(w) for w in order])
        for w, o in zip(order, outs):
            want = wq if w == "long" else st["positive" if w == 1 else "negative"]["quartiles"]
            assert np.array_equal(o.cpu().numpy(), want.numpy(), equal_nan=True), (order, w)
    B, P = 3, 37
    logits = torch.randn(B, P, P, generator=gen)
    ma = (torch.rand(B, P, generator=gen) > 0.4).float()
    mb = (torch.rand(B, P, generator=gen) > 0.5).float()
    mb[2] = 0.0                                                 # no positives in sample 2 -> NaN, negatives = everything
    st = O.dense_loss_stats(logits, ma[:, :, None] * mb[:, None, :])
    pos = ops.masked_quantiles(logits.to(DEV), P * P, 1, B, P * P, mask_a=ma.to(DEV), mask_b=mb.to(DEV), want=1)
    neg = ops.masked_quantiles(logits.to(DEV), P * P, 1, B, P * P, mask_a=ma.to(DEV), mask_b=mb.to(DEV), want=0)
    assert np.array_equal(pos.cpu().numpy(), st["positive"]["quartiles"].numpy(), equal_nan=True)
    assert np.array_equal(neg.cpu().numpy(), st["negative"]["quartiles"].numpy(), equal_nan=True)


@pytest.mark.parametrize("R_rows,K", [(640, 8192), (200, 1000), (4096, 4096)])
def test_rowkey_bf16x3_vs_oracle(R_rows, K):
    """Split-bf16 mode of the rows-vs-queue kernel (hi*hi + hi*lo + lo*hi on bf16 MFMA): north_star's bound is fp32
    logits within 1e-4; measured bounds asserted here: raw logits 3e-5, loss 5e-5, gradients 2e-4 * max|grad|."""
    gen = torch.Generator().manual_seed(R_rows)
    C, T = 128, 0.2
    rows = torch.nn.functional.normalize(torch.randn(R_rows, C, generator=gen), dim=1)
    queue = torch.nn.functional.normalize(torch.randn(C, K, generator=gen), dim=0)
    pos = torch.rand(R_rows, generator=gen) * 2 - 1
    r_cpu, p_cpu = rows.clone().requires_grad_(True), pos.clone().requires_grad_(True)
    want = O.queue_infonce(r_cpu, p_cpu, queue, T)
    want.backward()
    got = ops.rowkey_infonce(rows.to(DEV), (1, C, 0, 1), R_rows, queue.to(DEV), pos.reshape(-1, 1).to(DEV), T,
                             grad_scale=1.0 / R_rows, want_lneg=True, precision="bf16x3")
    assert_close(got.lnegT.t(), rows @ queue, 3e-5, what="raw logits")
    assert_close(got.loss, want.detach(), 5e-5, what="loss")
    assert_close(got.drows, r_cpu.grad, 1e-9, 2e-4, "d rows")
    assert_close(got.dE[:, 0], p_cpu.grad, 1e-9, 2e-4, "d pos")
    exact = ops.rowkey_infonce(rows.to(DEV), (1, C, 0, 1), R_rows, queue.to(DEV), pos.reshape(-1, 1).to(DEV), T,
                               grad_scale=1.0 / R_rows, precision="f32")
    assert_close(exact.loss, want.detach(), 2e-5, what="f32 loss")
    assert torch.equal(got.cnt_gt, exact.cnt_gt) or (got.cnt_gt - exact.cnt_gt).abs().max() <= 2


@pytest.mark.parametrize("B,P,K,with_ids", [(32, 196, 65536, False), (5, 37, 4096, True), (8, 1024, 8192, False)])
def test_loss_post_equals_the_two_separate_tail_launches(B, P, K, with_ids):
    """cp2_loss_post = cp2_rowkey_infonce_finalize (instance loss) + the dense loss's post-pass in ONE launch (round 4).
    Everything the two calls return is compared with the separately finalized calls: the instance side bit for bit; the
    dense side bit for bit up to P = 256 (one key pixel per thread either way), beyond that the per-sample sums are taken by
    1024 instead of 256 threads (another fixed order): 1e-6 relative."""
    gen = torch.Generator().manual_seed(B * P)
    C = 128
    unit = lambda *shape, dim: torch.nn.functional.normalize(torch.randn(*shape, generator=gen), dim=dim).to(DEV)  # noqa: E731
    qd, kd = unit(B, C, P, dim=1), unit(B, C, P, dim=1)
    ma = (torch.rand(B, P, generator=gen) > 0.4).float().to(DEV)
    mb = (torch.rand(B, P, generator=gen) > 0.5).float().to(DEV)
    ids = None
    if with_ids:
        ids = tuple(torch.randint(0, 50, (B, P), generator=gen).to(DEV) for _ in range(4))
    rows, queue = unit(B, C, dim=1), unit(C, K, dim=0)
    ext = (torch.rand(B, 3, generator=gen) * 2 - 1).to(DEV)

    def jobs_of(ins, den, mean_out):
        dense = dict(x=den.logits, stride_row=P * P, stride_elem=1, R=B, N=P * P, mask_a=ma, mask_b=mb)
        return [dict(dense, want=1), dict(dense, want=0), dict(x=ins.lneg, stride_row=K, stride_elem=1, R=B, N=K, mean_out=mean_out)]

    def run(merged, with_quartiles=False):
        ins = ops.rowkey_infonce(rows, (1, C, 0, 1), B, queue, ext, 0.2, grad_scale=1.0 / B, want_lneg=True, lneg_row_major=True,
                                 finalize=not merged)
        den = ops.dense_infonce_fwd(qd, kd, ma, mb, 0.7, ids, (1.0, 0.7, 0.2), want_logits=True,
                                    defer_post=merged and ins.pending is not None)
        mean = torch.empty(B, device=DEV)
        if merged:
            assert ins.pending is not None and den.pending is not None
            quart = ops.loss_post(ins, den, jobs_of(ins, den, mean) if with_quartiles else None)
            assert ins.pending is None and den.pending is None
        else:
            quart = ops.masked_quantiles_multi(jobs_of(ins, den, mean))
        return ins, den, quart, mean
    (i0, d0, q0, m0), (i1, d1, _, _) = run(False), run(True)
    # ... and with the step's three quartile sets riding in the same launch (cp2_step_post; row form only: P * P <= 131072)
    if P * P <= ops.QUANTILES_ROW_MAX:
        i2, d2, q2, m2 = run(True, with_quartiles=True)
        for a_, b_ in zip(q0, q2):
            assert torch.equal(a_, b_) or (torch.equal(torch.isnan(a_), torch.isnan(b_)) and torch.equal(a_.nan_to_num(), b_.nan_to_num()))
        assert torch.equal(m0, m2)
        assert torch.equal(i0.loss_rows, i2.loss_rows) and torch.equal(i0.drows, i2.drows) and torch.equal(d1.sample_scal, d2.sample_scal)
        assert torch.equal(i0.loss, i2.loss) and torch.equal(i0.cnt_gt, i2.cnt_gt) and torch.equal(i0.dE, i2.dE)
    for name in ("loss", "lse", "loss_rows", "cnt_gt", "drows", "dE", "lneg"):
        assert torch.equal(getattr(i0, name), getattr(i1, name)), name
    for name in ("lse", "colmax", "argx", "logits"):
        assert torch.equal(getattr(d0, name), getattr(d1, name)), name
    if P <= 256:
        assert torch.equal(d0.sample_scal, d1.sample_scal)
    else:
        assert torch.equal(d0.sample_scal[:, [0, 1, 5]], d1.sample_scal[:, [0, 1, 5]])       # mask sums and the arg-max label: exact
        for j in (2, 3, 4):                                                                     # loss_n, mean positive / negative score
            assert_close(d1.sample_scal[:, j], d0.sample_scal[:, j], 1e-9, 2e-6, f"sample_scal[:, {j}]")
    g0 = ops.dense_infonce_bwd(qd, kd, ma, mb, 0.7, d0, 0.1, ids, (1.0, 0.7, 0.2))
    g1 = ops.dense_infonce_bwd(qd, kd, ma, mb, 0.7, d1, 0.1, ids, (1.0, 0.7, 0.2))
    assert torch.equal(g0, g1)                                                                  # reads lse, Sa, Sb only


@pytest.mark.parametrize("R_rows,K,grad,shift", [(640, 8192, True, 0), (200, 1008, True, 0), (75, 2064, False, 0), (333, 1040, False, 1)])
def test_rowkey_bf16x3_row_major_logits_equal_the_key_major_ones(R_rows, K, grad, shift):
    """The row-major logits of the split-bf16 kernel (what the DenseCL score statistics read, builder.py:875-886) leave the
    launch through an in-LDS turn of every 32 x 32 tile (round 4; whole 128-byte row segments per store instruction instead
    of 64 four-byte requests).  They are the same accumulator registers as the key-major output: bit-equal, for ragged row
    counts, a last tile of 16 keys, with and without the gradient product, and through the scalar path when the buffer is
    not 16-byte aligned (`shift`); everything else the call returns is unchanged by the layout."""
    gen = torch.Generator().manual_seed(R_rows + K)
    C, T = 128, 0.2
    rows = torch.nn.functional.normalize(torch.randn(R_rows, C, generator=gen), dim=1).to(DEV)
    queue = torch.nn.functional.normalize(torch.randn(C, K, generator=gen), dim=0).to(DEV)
    pos = (torch.rand(R_rows, 1, generator=gen) * 2 - 1).to(DEV)
    gs = 1.0 / R_rows if grad else None
    a = ops.rowkey_infonce(rows, (1, C, 0, 1), R_rows, queue, pos, T, grad_scale=gs, want_lneg=True, precision="bf16x3")
    buf = torch.full((R_rows * K + 8,), float("nan"), device=DEV)
    b = ops.rowkey_infonce(rows, (1, C, 0, 1), R_rows, queue, pos, T, grad_scale=gs, want_lneg=True, precision="bf16x3",
                           lneg_row_major=True, lneg_out=buf[shift:shift + R_rows * K])
    assert a.lnegT.shape == (K, R_rows) and b.lneg.shape == (R_rows, K)
    assert torch.equal(a.lnegT.t(), b.lneg)
    assert torch.isnan(buf[:shift]).all() and torch.isnan(buf[shift + R_rows * K:]).all()      # nothing outside the R x K block
    assert torch.equal(a.loss_rows, b.loss_rows) and torch.equal(a.cnt_gt, b.cnt_gt)
    if grad:
        assert torch.equal(a.drows, b.drows) and torch.equal(a.dE, b.dE)
    assert_close(b.lneg, rows @ queue, 3e-5, what="raw logits")


@pytest.mark.parametrize("name", ["densecl_b2_128_k64", "densecl_b2_96_k64_coord"])
def test_densecl_local_positives_and_losses_golden(golden_dir, name):
    """T18 + T19 of forward_densecl on the reference's recorded tensors (builder.py:808-910, :760-772)."""
    from cp2_amd import builder
    g = load(golden_dir, name)
    tg, tl, lmbd, lc = [float(v) for v in g["cfg_f"]]
    b = g["q_local"].shape[0]
    pos, best = builder.densecl_local_positives(G(g["q_embed"]), G(g["k_embed"]), G(g["q_local"]), G(g["k_local"]),
                                                G(g["q_pixel_ids"]).reshape(b, -1), G(g["k_pixel_ids"]).reshape(b, -1), lc)
    assert np.array_equal(best.cpu().numpy(), g["pos_global_k_idx"])
    assert_close(pos.reshape(-1, 1), g["pos_local"], 3e-6, what="pos_local")
    ql = G(g["q_local"]).requires_grad_(True)
    loss_local = builder.queue_infonce(ql, pos.reshape(-1).detach(), G(g["queue2_before"]), tl)
    loss_local.backward()
    assert_close(loss_local, g["loss_local"], 2e-5, what="loss_local")
    qg = G(g["q_global"])
    loss_global = builder.queue_infonce(qg, (qg * G(g["k_global"])).sum(1), G(g["queue_before"]), tg)
    assert_close(loss_global, g["loss_global"], 2e-5, what="loss_global")
    assert_close((1 - lmbd) * loss_global + lmbd * loss_local, g["loss"], 2e-5, what="loss")
    assert torch.isfinite(ql.grad).all() and float(ql.grad.abs().max()) > 0


def test_densecl_coordinate_mix_golden(golden_dir):
    """cp2_densecl_match on the reference's recorded tensors WITH id overlap (the coordinate mix, builder.py:838-855;
    fixture taken at the point where the reference's :861 raises): indices exact, mixed positives to 3e-6."""
    from cp2_amd import builder
    g = load(golden_dir, "densecl_coord_overlap")
    lc, b = float(g["cfg_f"][3]), g["q_local"].shape[0]
    st = {}
    pos, best = builder.densecl_local_positives(G(g["q_embed"]), G(g["k_embed"]), G(g["q_local"]), G(g["k_local"]),
                                                G(g["q_pixel_ids"]).reshape(b, -1), G(g["k_pixel_ids"]).reshape(b, -1), lc,
                                                metrics=st)
    assert np.array_equal(best.cpu().numpy(), g["pos_global_k_idx"])
    assert_close(pos, g["pos_local"], 3e-6, what="pos_local (mixed)")
    want = O.densecl_matching_rate(torch.from_numpy(g["q_local"]), torch.from_numpy(g["k_local"]),
                                   torch.from_numpy(g["q_pixel_ids"]), torch.from_numpy(g["k_pixel_ids"]))
    assert abs(float(st["matching_positives_rate"]) - want) < 1e-6


def _match_case(B, P, CE, seed, dup=True):
    """Random DenseCL features whose backbone similarities are well separated + id maps with overlap and repeated ids."""
    g = torch.Generator().manual_seed(seed)
    qe, ke = torch.randn(B, CE, P, generator=g), torch.randn(B, CE, P, generator=g)
    ql = torch.nn.functional.normalize(torch.randn(B, 128, P, generator=g), dim=1)
    kl = torch.nn.functional.normalize(torch.randn(B, 128, P, generator=g), dim=1)
    ids_q = torch.stack([torch.randperm(4 * P, generator=g)[:P] + 1 + n * 10 * P for n in range(B)])
    ids_k = torch.stack([torch.randperm(4 * P, generator=g)[:P] + 1 + n * 10 * P for n in range(B)])
    ids_k[:, : P // 3] = ids_q[:, torch.randperm(P, generator=g)[: P // 3]]          # a third of the key pixels match
    if dup:
        ids_k[:, P // 3: P // 3 + 5] = ids_k[:, :5]                                   # five ids occur twice among the keys
    return qe, ke, ql, kl, ids_q, ids_k


def _match_oracle(qe, ke, ql, kl, ids_q, ids_k, lmbd):
    """oracle positives + d sum(pos * w) / d q_local, with the top-2 gap of the backbone similarity per query pixel."""
    qn, kn = torch.nn.functional.normalize(qe.double(), dim=1), torch.nn.functional.normalize(ke.double(), dim=1)
    sim = torch.einsum("ncx,ncy->nxy", qn, kn)
    top2 = sim.topk(2, dim=2).values
    q = ql.clone().requires_grad_(True)
    queue2 = torch.nn.functional.normalize(torch.randn(128, 8, generator=torch.Generator().manual_seed(1)), dim=0)
    _, pos, _, best = O.densecl_local_loss(qn.float(), kn.float(), q, kl, ids_q, ids_k, queue2, lmbd_coordinate=lmbd)
    w = torch.linspace(0.5, 1.5, pos.numel()).reshape(pos.shape)
    (pos * w).sum().backward()
    return pos.detach(), best, q.grad, w, (top2[..., 0] - top2[..., 1]).float(), sim.argmax(2)


@pytest.mark.parametrize("B,P,CE,form,lmbd", [(3, 36, 2048, "f32_nchw", 0.3), (2, 196, 2048, "bf16_cl", 0.3), (2, 196, 512, "f32_cl", 0.0),
                                              (2, 300, 256, "bf16_cl", 0.25), (1, 300, 128, "f32_nchw", 0.5), (4, 49, 2048, "bf16_cl4", 0.0)])
def test_densecl_match_vs_oracle(B, P, CE, form, lmbd):
    """cp2_densecl_match (T18, builder.py:818-864) on raw backbone features in every layout it reads in place -- fp32
    NCHW / channels-last, bf16 channels-last 3-D and 4-D -- with a key-row permutation, overlapping and repeated ids:
    arg-max index equal to the fp64 arg-max wherever the top-2 gap exceeds 1e-5 (elsewhere the chosen similarity is within
    1e-5 of the maximum), positives to 3e-6, d pos / d q_local to 3e-6, the matching-positives counts exact."""
    from cp2_amd import builder
    qe, ke, ql, kl, ids_q, ids_k = _match_case(B, P, CE, seed=B * 1000 + P + CE)
    if form.startswith("bf16"):
        qe, ke = qe.bfloat16().float(), ke.bfloat16().float()           # the values a bf16 backbone hands over
    pos_w, best_w, grad_w, w, gap, best64 = _match_oracle(qe, ke, ql, kl, ids_q, ids_k, lmbd)
    perm = torch.randperm(B, generator=torch.Generator().manual_seed(5))
    k_row = torch.argsort(perm)                                         # sample n's key side sits at row k_row[n]
    ke_s, kl_s = ke[perm], kl[perm]

    def dev_embed(x):
        x = x.to(DEV)
        if form == "f32_nchw":
            return x.contiguous()
        if form == "f32_cl":
            return x.transpose(1, 2).contiguous().transpose(1, 2)
        if form == "bf16_cl":
            return x.bfloat16().transpose(1, 2).contiguous().transpose(1, 2)
        s = int(round(P ** 0.5))
        return x.bfloat16().reshape(B, CE, s, s).contiguous(memory_format=torch.channels_last)
    q = ql.to(DEV).requires_grad_(True)
    st = {}
    pos, best = builder.densecl_local_positives(dev_embed(qe), dev_embed(ke_s), q, kl_s.to(DEV), ids_q.to(DEV), ids_k.to(DEV), lmbd,
                                                metrics=st, k_row=k_row.to(DEV), normalize_k=True)
    (pos * w.to(DEV)).sum().backward()
    best = best.cpu()
    clear = gap > 1e-5
    assert torch.equal(best[clear], best64[clear]), "arg-max differs where the top-2 gap is clear"
    # rows whose arg-max is decided inside rounding noise: whichever was chosen must be a maximum to 1e-5
    sim = torch.einsum("ncx,ncy->nxy", torch.nn.functional.normalize(qe, dim=1), torch.nn.functional.normalize(ke, dim=1))
    chosen = sim.gather(2, best.unsqueeze(2)).squeeze(2)
    assert float((sim.max(2).values - chosen).max()) <= 1e-5
    same = best == best_w
    assert same.float().mean() > 0.99
    assert_close(pos.cpu()[same], pos_w[same], 3e-6, what="pos")
    samec = same.unsqueeze(1).expand_as(grad_w)
    assert_close(q.grad.cpu()[samec], grad_w[samec], 3e-6, what="d pos / d q_local")
    want = O.densecl_matching_rate(ql, kl, ids_q, ids_k)
    assert abs(float(st["matching_positives_rate"]) - want) <= 1.0 / max(1, int((ids_q[:, :, None] == ids_k[:, None, :]).any(-1).sum())) + 1e-6


def test_densecl_match_without_ids_and_normalised_inputs():
    """No id maps (plain DenseCL) and inputs that are unit vectors already (normalize_k=False): the golden path's form."""
    from cp2_amd import ops
    qe, ke, ql, kl, _, _ = _match_case(2, 64, 256, seed=3)
    qn, kn = torch.nn.functional.normalize(qe, dim=1), torch.nn.functional.normalize(ke, dim=1)
    res = ops.densecl_match(qn.to(DEV), kn.to(DEV), ql.to(DEV), kl.to(DEV), normalize_k=False, want_kvec=True)
    sim = torch.einsum("ncx,ncy->nxy", qn.double(), kn.double())
    top2 = sim.topk(2, dim=2).values
    clear = (top2[..., 0] - top2[..., 1]) > 1e-5
    best = res.best.cpu().long()
    assert torch.equal(best[clear], sim.argmax(2)[clear])
    want_k = torch.gather(kl, 2, best.unsqueeze(1).expand(-1, 128, -1))
    assert torch.equal(res.kvec.cpu(), want_k)                       # the gathered key vectors, bit for bit
    assert_close(res.pos, (ql * want_k).sum(1), 2e-6, what="pos")


def test_queue_infonce_chunked_statistics_equal_one_shot(monkeypatch):
    """The rank-0 DenseCL score statistics (reference builder.py:875-886) walked in groups of samples through one re-used
    logit buffer (_queue_infonce_chunked) against the one-shot form that materialises every row's logits: per-row
    quartiles and means bit for bit (same kernels, same per-row arithmetic), loss and gradients to 1e-6."""
    from cp2_amd import builder
    g = torch.Generator(DEV).manual_seed(11)
    b, C, S2, K = 6, 128, 196, 4096                                   # 1176 rows: the bf16x3 kernel, as at 32 x 196
    rows = torch.nn.functional.normalize(torch.randn(b, C, S2, device=DEV, generator=g), dim=1)
    queue = torch.nn.functional.normalize(torch.randn(C, K, device=DEV, generator=g), dim=0)
    pos = torch.rand(b * S2, device=DEV, generator=g)
    out = {}
    for tag, limit in (("one", 1 << 40), ("chunked", 2 * S2 * K * 4)):
        monkeypatch.setattr(builder, "STATS_CHUNK_BYTES", limit)
        r = rows.clone().requires_grad_(True)
        p = pos.clone().requires_grad_(True)
        st = {}
        loss = builder.queue_infonce(r, p, queue, 0.2, stats=st)
        loss.backward()
        out[tag] = (loss.detach(), r.grad, p.grad, st["neg_mean"], st["neg_quartiles"])
    a, c = out["one"], out["chunked"]
    assert torch.equal(a[4], c[4]) and torch.equal(a[3], c[3])
    assert_close(c[0], a[0], 1e-6, what="loss")
    assert_close(c[1], a[1], 1e-9, 1e-6, what="d rows")
    assert_close(c[2], a[2], 1e-9, 1e-6, what="d pos")
    lneg = torch.einsum("ncx,ck->nxk", rows, queue).reshape(b * S2, K)
    want = torch.quantile(lneg, torch.tensor([0.25, 0.5, 0.75], device=DEV), dim=1)
    assert_close(c[4], want, 1e-4, what="quartiles vs torch.quantile of an fp32 product (bf16x3 logits: 3e-5)")


def test_densecl_symmetric_golden(golden_dir):
    """PROPOSED_V2 symmetric pass (reference builder.py:944-972): both passes' global and local losses on the recorded
    tensors, summed as the reference does."""
    from cp2_amd import builder
    g = load(golden_dir, "densecl_v2_symmetric")
    tg, tl, lmbd, lc = [float(v) for v in g["cfg_f"]]
    b = g["q_local"].shape[0]
    tot_l = tot_g = 0.0
    for sfx in ("", "_2"):
        pos, best = builder.densecl_local_positives(G(g["q_embed" + sfx]), G(g["k_embed" + sfx]), G(g["q_local" + sfx]),
                                                    G(g["k_local" + sfx]), G(g["q_pixel_ids" + sfx]).reshape(b, -1),
                                                    G(g["k_pixel_ids" + sfx]).reshape(b, -1), lc)
        assert np.array_equal(best.cpu().numpy(), g["pos_global_k_idx" + sfx])
        assert_close(pos.reshape(-1, 1), g["pos_local" + sfx], 3e-6, what="pos_local" + sfx)
        loss_local = builder.queue_infonce(G(g["q_local" + sfx]), pos.reshape(-1), G(g["queue2_before"]), tl)
        assert_close(loss_local, g["loss_local" + sfx], 2e-5, what="loss_local" + sfx)
        qg = G(g["q_global" + sfx])
        loss_global = builder.queue_infonce(qg, (qg * G(g["k_global" + sfx])).sum(1), G(g["queue_before"]), tg)
        assert_close(loss_global, g["loss_global" + sfx], 2e-5, what="loss_global" + sfx)
        tot_l, tot_g = tot_l + loss_local, tot_g + loss_global
    assert_close((1 - lmbd) * tot_g + lmbd * tot_l, g["loss"], 4e-5, what="loss")


@pytest.mark.parametrize("B,P", [(2, 196), (1, 132), (3, 1024), (5, 70)])
def test_dense_range_split_equals_single_walk_with_weights(B, P):
    """The dense kernels with the query / key range shared by several workgroups (cp2_dense_num_splits > 1, merge and
    partial-sum kernels) against the single-walk form (pinned by the goldens), with correspondence weights on:
    statistics and gradients to 2e-6 relative, arg-max indices exact."""
    from cp2_amd import _lib, ops
    assert _lib.load().cp2_dense_num_splits(B, P) > 1
    g = torch.Generator(DEV).manual_seed(B * 7 + P)
    qd = torch.nn.functional.normalize(torch.randn(B, 128, P, device=DEV, generator=g), dim=1)
    kd = torch.nn.functional.normalize(torch.randn(B, 128, P, device=DEV, generator=g), dim=1)
    ma = (torch.rand(B, P, device=DEV, generator=g) > 0.4).float()
    mb = (torch.rand(B, P, device=DEV, generator=g) > 0.5).float()
    pix_a = torch.randint(1, P // 2, (B, P), device=DEV, generator=g)
    pix_b = torch.randint(1, P // 2, (B, P), device=DEV, generator=g)
    reg_a, reg_b = pix_a // 4, pix_b // 4                       # region 0 = unknown for some pixels
    ids, w = (pix_a, pix_b, reg_a, reg_b), (2.0, 0.5, 0.25)
    res = {}
    for split in (True, False):
        fw = ops.dense_infonce_fwd(qd, kd, ma, mb, 0.7, ids=ids, weights=w, want_logits=True, split=split)
        gq = ops.dense_infonce_bwd(qd, kd, ma, mb, 0.7, fw, 0.3, ids=ids, weights=w, split=split)
        res[split] = (fw, gq)
    a, b = res[True], res[False]
    assert torch.equal(a[0].argx, b[0].argx) and torch.equal(a[0].colmax, b[0].colmax)
    assert torch.equal(a[0].logits, b[0].logits)
    for name in ("lse", "sample_scal"):
        x, y = getattr(a[0], name), getattr(b[0], name)
        assert torch.allclose(x, y, rtol=2e-6, atol=2e-6, equal_nan=True), name
    assert torch.allclose(a[0].loss, b[0].loss, rtol=2e-6, atol=2e-6) and torch.equal(a[0].acc, b[0].acc)
    assert (a[1] - b[1]).abs().max().item() <= 2e-6 * b[1].abs().max().item() + 1e-9
