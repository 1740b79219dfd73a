"""GPU: BASELINE configs[3] -- ResNet-101 + dilated FCN head, 512x512 crops (P = 4096 at output stride 8),
queue = 131072 -- through the C ABI against the CPU oracle.

  * dense InfoNCE forward + backward at P = 4096 (reference builder.py:1289-1292, 1431-1437), with and without the
    range split;
  * rows-vs-queue InfoNCE at K = 131072 (reference builder.py:1395-1397, 1420-1428): the instance kernel on all 32 rows,
    a DenseCL-style row slice against the oracle, and the size-independent properties used at 65536;
  * one model-level forward + backward of configs/config_pretrain_r101_d8.py at 512x512 against the oracle fed with
    the very same encoder outputs.

Tolerances as in test_gpu_loss.py: losses 2e-5, gradients 2e-5 * max|grad|, raw logits 2e-6.
"""
import os

import numpy as np
import pytest
import torch

from cp2_amd import builder, ops, synthetic
from cp2_amd.config import Config
from cp2_amd.pretrain_types import PretrainType
from oracle import cp2_oracle as O
from tests.test_gpu_loss import assert_close, grad_close

pytestmark = pytest.mark.gpu
DEV = "cuda"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
K4 = 131072


@pytest.mark.parametrize("split", [True, False])
def test_dense_infonce_p4096_vs_oracle(split):
    gen = torch.Generator().manual_seed(4096)
    B, C, P, T, gs = 2, 128, 4096, 1.0, 0.2 / 2
    qd = torch.nn.functional.normalize(torch.randn(B, C, P, generator=gen), dim=1)
    kd = torch.nn.functional.normalize(torch.randn(B, C, P, generator=gen), dim=1)
    ma = (torch.rand(B, P, generator=gen) > 0.35).float()
    mb = (torch.rand(B, P, generator=gen) > 0.55).float()
    q_cpu = qd.clone().requires_grad_(True)
    loss, per_sample, _ = O.dense_infonce(O.dense_logits(q_cpu, kd), ma, mb, T)
    (loss * (gs * B)).backward()                         # kernel convention: grad_scale * d(sum_n loss_n)/dq
    fw = ops.dense_infonce_fwd(qd.to(DEV), kd.to(DEV), ma.to(DEV), mb.to(DEV), T, split=split)
    g = ops.dense_infonce_bwd(qd.to(DEV), kd.to(DEV), ma.to(DEV), mb.to(DEV), T, fw, gs, split=split)
    assert_close(fw.loss, loss.detach(), 2e-5, what="loss_dense")
    assert_close(fw.sample_scal[:, 2], per_sample.detach(), 2e-5, what="loss_dense per sample")
    grad_close(g, q_cpu.grad, "d q_dense")
    # logging sums that ride along: mean positive / negative raw score per sample
    raw = O.dense_logits(qd, kd)
    lab = ma[:, :, None] * mb[:, None, :]
    assert_close(fw.sample_scal[:, 3], (raw * lab).sum((1, 2)) / lab.sum((1, 2)), 3e-6, what="+mean")
    assert_close(fw.sample_scal[:, 4], (raw * (1 - lab)).sum((1, 2)) / (1 - lab).sum((1, 2)), 3e-6, what="-mean")


def test_instance_infonce_k131072_vs_oracle():
    """The a10 kernel at the config-4 queue length: all 32 rows against the full queue on the CPU."""
    gen = torch.Generator().manual_seed(131)
    B, C, T = 8, 128, 0.2
    for R in (B, 32):
        q_pos = torch.nn.functional.normalize(torch.randn(R, C, generator=gen), dim=1)
        k_pos = torch.nn.functional.normalize(torch.randn(R, C, generator=gen), dim=1)
        queue = torch.nn.functional.normalize(torch.randn(C, K4, generator=gen), dim=0)
        q_cpu = q_pos.clone().requires_grad_(True)
        loss, logits, l_pos, l_neg = O.instance_infonce(q_cpu, k_pos, queue, T)
        loss.backward()
        res = ops.rowkey_infonce(q_pos.to(DEV), (1, C, 0, 1), R, queue.to(DEV), l_pos.detach().to(DEV), T,
                                 grad_scale=1.0 / R, want_lneg=True, lneg_row_major=(R == 32))
        assert_close(res.lneg if R == 32 else res.lnegT.t(), l_neg.detach(), 2e-6, what="l_neg")
        assert_close(res.loss, loss.detach(), 2e-5, what="loss_instance")
        # d loss / d q_pos through the queue logits only (the positive's share is dE * k_pos, added by pool_bwd)
        want = q_cpu.grad - (torch.softmax(logits.detach(), 1)[:, :1] - 1) / T / R * k_pos
        grad_close(res.drows, want, "d q_pos (queue part)")
        top = (logits.detach()[:, 1:] > logits.detach()[:, :1]).sum(1).int()
        assert torch.equal(res.cnt_gt.cpu(), top)


def test_rowkey_k131072_slice_and_properties():
    """DenseCL-style rows (pixel-major layout) against a 131072-key queue: a 640-row slice against the oracle, then
    logsumexp over the whole queue = logaddexp of the two half queues, and invariance under a key permutation."""
    gen = torch.Generator().manual_seed(7)
    b, C, S2, T = 8, 128, 256, 0.2
    rows = torch.nn.functional.normalize(torch.randn(b, C, S2, generator=gen), dim=1).to(DEV)
    queue = torch.nn.functional.normalize(torch.randn(C, K4, generator=gen), dim=0).to(DEV)
    pos = (torch.rand(b * S2, 1, generator=gen) * 2 - 1).to(DEV)
    R, lay = b * S2, (S2, C * S2, 1, S2)
    for prec in ("f32", "bf16x3"):
        tol_l, tol_g = (2e-5, 2e-5) if prec == "f32" else (5e-5, 2e-4)
        full = ops.rowkey_infonce(rows, lay, R, queue, pos, T, grad_scale=1.0 / R, precision=prec)
        none = torch.full_like(pos, -1e30)
        h1 = ops.rowkey_infonce(rows, lay, R, queue[:, :K4 // 2].contiguous(), none, T, None, precision=prec)
        h2 = ops.rowkey_infonce(rows, lay, R, queue[:, K4 // 2:].contiguous(), pos, T, None, precision=prec)
        assert_close(torch.logaddexp(h1.lse, h2.lse), full.lse, tol_l, what=f"lse split {prec}")
        perm = torch.randperm(K4, generator=gen).to(DEV)
        pq = ops.rowkey_infonce(rows, lay, R, queue[:, perm].contiguous(), pos, T, grad_scale=1.0 / R, precision=prec)
        assert_close(pq.loss, full.loss, tol_l, what=f"loss under key permutation {prec}")
        assert_close(pq.drows, full.drows, 1e-9, tol_g, f"grad under key permutation {prec}")
        rs = 640
        sub = rows[:3].contiguous()                                  # 768 rows, use the first 640
        r_cpu = sub.cpu().permute(0, 2, 1).reshape(-1, C)[:rs].clone().requires_grad_(True)
        p_cpu = pos[:rs, 0].cpu().clone().requires_grad_(True)
        want = O.queue_infonce(r_cpu, p_cpu, queue.cpu(), T)
        want.backward()
        got = ops.rowkey_infonce(sub, lay, rs, queue, pos[:rs].contiguous(), T, grad_scale=1.0 / rs, precision=prec)
        assert_close(got.loss, want.detach(), tol_l, what=f"slice loss {prec}")
        assert_close(got.drows.permute(0, 2, 1).reshape(-1, C)[:rs], r_cpu.grad, 1e-9, tol_g, f"slice d rows {prec}")
        assert_close(got.dE[:, 0], p_cpu.grad, 1e-9, tol_g, f"slice d pos {prec}")


def test_enqueue_k131072_wraps():
    gen = torch.Generator().manual_seed(5)
    C, n = 128, 64
    queue = torch.randn(C, K4, generator=gen)
    keys = torch.randn(n, C, generator=gen)
    qd, ptr = queue.to(DEV), torch.tensor([K4 - 40], dtype=torch.long, device=DEV)
    ops.enqueue(qd, keys.to(DEV), ptr)
    want, p2 = O.dequeue_and_enqueue(queue, K4 - 40, keys)
    assert int(ptr) == p2 == 24 and torch.equal(qd.cpu(), want)


def test_model_forward_r101_d8_512_matches_oracle_on_same_features():
    """configs/config_pretrain_r101_d8.py at 512x512, 2 images per GPU, queue 131072: the whole forward_cp2 (composition,
    strided masks, EMA, shuffle, both encoders, fused loss section, enqueue) against the oracle on the very same encoder
    outputs, and a backward pass that reaches every trainable parameter of the query encoder."""
    torch.manual_seed(0)
    cfg = Config.fromfile(os.path.join(ROOT, "configs", "config_pretrain_r101_d8.py"))
    model = builder.MODEL(cfg, rank=0, K=K4, pretrain_from_scratch=True, pretrain_type=PretrainType.CP2, device=DEV,
                          amp_dtype=torch.bfloat16, channels_last=True).to(DEV).train()
    assert model.output_stride == 8
    b, hw = 2, 512
    batch = synthetic.make_batch(b, hw, hw, DEV, seed=11)
    feats = {}
    hq = model.encoder_q.register_forward_hook(lambda m, i, o: feats.__setitem__("q", o))
    hk = model.encoder_k.register_forward_hook(lambda m, i, o: feats.__setitem__("k", o))
    queue0, ptr0 = model.queue.clone(), int(model.queue_ptr)
    perm = torch.randperm(b)
    model.key_forward_graph = False
    loss = model(visualize=False, step=0, new_epoch=False, idx_shuffle=perm.to(DEV), **batch)
    loss.backward()
    hq.remove(), hk.remove()
    assert feats["q"].shape == (b, 128, 64, 64)
    q_feat = feats["q"].detach().float().cpu().requires_grad_(True)
    k_feat = O.unshuffle_take(feats["k"].detach().float().cpu(), perm, 0, 1)
    r = O.cp2_loss_section(q_feat, k_feat, batch["bg0"].cpu(), batch["bg1"].cpu(), batch["pixel_ids_a"].cpu(),
                           batch["pixel_ids_b"].cpu(), batch["region_ids_a"].cpu(), batch["region_ids_b"].cpu(),
                           queue0.cpu(), output_stride=8, with_stats=True)
    assert abs(float(loss) - float(r["loss"])) <= 2e-5, (float(loss), float(r["loss"]))
    logs = model.flush_logs()[0][1]
    assert abs(logs["train/loss_ins_step"] - float(r["loss_instance"])) <= 2e-5
    assert abs(logs["train/loss_dense_step"] - float(r["loss_dense"])) <= 2e-5
    assert abs(logs["train/acc_seg_step"] - float(r["acc_dense"])) <= 1e-4
    st = r["dense_stats"]
    assert abs(logs["step/dense_per_sample_median_positive_scores"] - float(st["positive"]["quartiles"][1].mean())) <= 3e-6
    assert abs(logs["step/dense_per_sample_upper_negative_scores"] - float(st["negative"]["quartiles"][2].mean())) <= 3e-6
    assert abs(logs["step/instance_median_negative_scores"] - float(r["instance_neg_quartiles"][1].mean())) <= 3e-6
    qa, ptr = O.dequeue_and_enqueue(queue0.cpu(), ptr0, r["k_pos"].detach())
    assert int(model.queue_ptr) == ptr == b
    assert (model.queue.cpu() - qa).abs().max() <= 2e-6
    ious, ious_m = model.epoch_ious()
    assert np.array_equal(np.float32(ious), r["iou"].numpy()) and np.array_equal(np.float32(ious_m), r["iou_masked"].numpy())
    grads = [p.grad for n, p in model.encoder_q.named_parameters() if "conv_seg" not in n]
    assert all(g is not None and torch.isfinite(g).all() for g in grads)
