"""CPU: the oracle (oracle/cp2_oracle.py) against golden vectors recorded from the
reference's own code (tests/golden/make_goldens.py) and against the reference's
known-answer tests.  Integer / index / mask results must be bit-exact; fp32
results are compared at 1e-6 (same torch CPU kernels, possibly different
association order)."""
import os

import numpy as np
import pytest
import torch

from oracle import cp2_oracle as O

CP2_CASES = ["cp2_b4_64_k64", "cp2_b4_96_k64_wrap_bg", "cp2_b3_80x112_k1024", "cp2_proposed_weights"]


def load(golden_dir, name):
    return dict(np.load(os.path.join(golden_dir, name + ".npz")))


def T(x):
    return torch.from_numpy(np.asarray(x))


def close(a, b, tol=1e-6):
    a, b = np.asarray(a, dtype=np.float64), np.asarray(b, dtype=np.float64)
    assert a.shape == b.shape, (a.shape, b.shape)
    np.testing.assert_allclose(a, b, rtol=tol, atol=tol)


# ---- reference known-answer tests (reference tests/test_correlation_mapping.py:65-77,118-130)
def test_kat_unique_ids(golden_dir):
    g = load(golden_dir, "corrmap_kats")
    r = O.masked_correlation_map(T(g["unique_map_a"]), T(g["unique_map_b"]), T(g["unique_mask_a"]), T(g["unique_mask_b"]))
    assert torch.equal(r["iou"], torch.ones(4) * (12 / (12 + 25 - 12 + 25 - 12)))
    assert torch.equal(r["iou_masked"], torch.ones(4) * (1 / 3))


def test_kat_shared_ids(golden_dir):
    g = load(golden_dir, "corrmap_kats")
    r = O.masked_correlation_map(T(g["shared_map_a"]), T(g["shared_map_b"]), T(g["shared_mask_a"]), T(g["shared_mask_b"]))
    assert torch.equal(r["iou"], torch.tensor([4 / 7]))
    assert torch.equal(r["iou_masked"], torch.tensor([2 / 3]))


# ---- reference tests/test_correlation_mapping.py:188-206 (stride gather rule)
def test_kat_rescale_ids():
    ids = torch.arange(1, 151).reshape(10, 15)
    assert torch.equal(O.strided_gather(ids, 1), ids)
    s2 = O.strided_gather(ids, 2)
    assert s2.shape == (5, 7) and int(s2[0, 0]) == 17 and s2.shape[0] * 2 == 10


# ---- reference tests/test_contrastive_metrics.py:17-57 (quantile convention)
def test_kat_quantile_convention():
    scores = torch.tensor([[[1., 2, 3], [4, 5, 6]], [[1, 2, 3], [7, 8, 9]]])
    st = O.dense_loss_stats(scores, torch.ones_like(scores))
    assert torch.equal(st["positive"]["quartiles"], torch.tensor([[2.25, 2.25], [3.5, 5.0], [4.75, 7.75]]))
    assert torch.equal(st["positive"]["average"], torch.tensor([3.5, 5.0]))


@pytest.mark.parametrize("tag", ["unique", "shared", "random"])
def test_corrmap_golden(golden_dir, tag):
    g = load(golden_dir, "corrmap_kats")
    r = O.masked_correlation_map(T(g[f"{tag}_map_a"]), T(g[f"{tag}_map_b"]), T(g[f"{tag}_mask_a"]), T(g[f"{tag}_mask_b"]))
    for k in ("corr_map", "corr_map_a", "corr_map_b", "iou", "iou_masked"):
        assert np.array_equal(r[k].numpy(), g[f"{tag}_{k}"]), k
    for k in ("corr_mask", "corr_map_a_masked", "corr_map_b_masked"):
        assert np.array_equal(r[k].numpy().astype(np.float32), g[f"{tag}_{k}"].astype(np.float32)), k


@pytest.mark.parametrize("name", CP2_CASES)
def test_cp2_loss_section_golden(golden_dir, name):
    g = load(golden_dir, name)
    b, h, w, K, stride, inc_bg = [int(v) for v in g["cfg"]]
    tg, tl, lmbd, wp, wr, wn, m = [float(v) for v in g["cfg_f"]]
    if name != "cp2_proposed_weights":
        wp, wr, wn = int(wp), int(wr), int(wn)       # argparse defaults are python ints (main.py:75-77)
    # a1: composition, bit exact
    out_a, _ = O.compose_mask(T(g["in_img_a"]), T(g["in_bg0"]))
    assert np.array_equal(out_a.numpy(), g["img_a"])
    # the reference's local img_b is the key batch AFTER shuffle-BN (builder.py:1274)
    out_b, _ = O.compose_mask(T(g["in_img_b"]), T(g["in_bg1"]))
    perm = torch.argsort(T(g["idx_unshuffle"]))
    assert np.array_equal(O.shuffle_take(out_b, perm, 0, 1).numpy(), g["img_b"])
    q = T(g["q_feat"]).clone().requires_grad_(True)
    r = O.cp2_loss_section(q, T(g["k_feat"]), T(g["in_bg0"]), T(g["in_bg1"]), T(g["in_pixel_ids_a"]),
                           T(g["in_pixel_ids_b"]), T(g["in_region_ids_a"]), T(g["in_region_ids_b"]),
                           T(g["queue_before"]), output_stride=stride, temp_global=tg, temp_local=tl,
                           lmbd_dense=lmbd, include_background=bool(inc_bg), w_pixel=wp, w_region=wr,
                           w_not=wn, with_stats=True)
    r["loss"].backward()
    # bit exact: masks, ids, ious, weights
    for k in ("mask_a", "mask_b", "pixel_ids_a", "pixel_ids_b", "region_ids_a", "region_ids_b",
              "iou", "iou_masked", "pixel_iou", "pixel_iou_masked"):
        assert np.array_equal(r[k].numpy(), g[k]), k
    assert np.array_equal(r["corr_weights"].numpy().astype(np.float32), g["corr_weights"].astype(np.float32))
    # fp32
    for k in ("q_dense", "k_dense", "q_pos", "k_pos", "q_neg", "k_neg", "l_pos", "l_neg", "logits_moco"):
        close(r[k].detach(), g[k])
    close(r["logits_dense_raw"].detach(), g["_logits_dense"])
    for k in ("loss_instance", "loss_dense", "loss"):
        close(r[k].detach(), g[k], 2e-6)
    close(q.grad, g["dq_feat"], 1e-6)
    # logging statistics
    close(r["dense_stats"]["positive"]["average"], g["dense_positive_average"])
    close(r["dense_stats"]["negative"]["average"], g["dense_negative_average"])
    close(r["dense_stats"]["positive"]["quartiles"], g["dense_positive_quartiles"])
    close(r["dense_stats"]["negative"]["quartiles"], g["dense_negative_quartiles"])
    close(r["instance_neg_mean"], g["instance_average_negative_scores"])
    close(r["instance_neg_quartiles"], g["instance_negative_quartiles"])
    close(r["acc1"], g["acc1"]); close(r["acc5"], g["acc5"]); close(r["acc_dense"], g["acc_dense"])
    # a13: the enqueue that forward_cp2 performs with k_pos (world size 1)
    qa, ptr = O.dequeue_and_enqueue(T(g["queue_before"]), int(g["ptr_before"]), T(g["k_pos"]))
    assert np.array_equal(qa.numpy(), g["queue_after"]) and ptr == int(g["ptr_after"])


@pytest.mark.parametrize("nt", ["none", "fixed", "average", "median", "hard"])
def test_negative_type_golden(golden_dir, nt):
    """the NegativeType reshaping of the negative dense logits (reference builder.py:1332-1386), recorded
    from the reference for the same inputs under every type; HARD must equal NONE (it edits a copy)."""
    g = load(golden_dir, "cp2_neg_" + nt)
    b, h, w, K, stride, inc_bg = [int(v) for v in g["cfg"]]
    tg, tl, lmbd, wp, wr, wn, m = [float(v) for v in g["cfg_f"]]
    ntype, nscale = int(g["negative"][0]), float(g["negative"][1])
    assert ntype == {"none": O.NEG_NONE, "fixed": O.NEG_FIXED, "average": O.NEG_AVERAGE, "median": O.NEG_MEDIAN, "hard": O.NEG_HARD}[nt]
    q = T(g["q_feat"]).clone().requires_grad_(True)
    mi = (T(g["mask_a"]), T(g["mask_b"]), T(g["pixel_ids_a"]), T(g["pixel_ids_b"]), T(g["region_ids_a"]), T(g["region_ids_b"]))
    r = O.cp2_loss_section(q, T(g["k_feat"]), None, None, None, None, None, None, T(g["queue_before"]), output_stride=stride,
                           temp_global=tg, temp_local=tl, lmbd_dense=lmbd, include_background=bool(inc_bg), w_pixel=wp,
                           w_region=wr, w_not=wn, with_stats=True, negative_type=ntype, negative_scale=nscale, masks_and_ids=mi)
    r["loss"].backward()
    assert np.array_equal(r["corr_weights"].numpy().astype(np.float32), g["corr_weights"].astype(np.float32))
    close(r["logits_dense_reshaped"].detach(), g["_logits_dense"])      # the reference edits _logits_dense in place
    close(r["logits_dense_scaled"].detach(), g["logits_dense"], 2e-6)
    for k in ("loss_instance", "loss_dense", "loss"):
        close(r[k].detach(), g[k], 2e-6)
    close(q.grad, g["dq_feat"], 1e-6)
    close(r["dense_stats"]["negative"]["average"], g["dense_negative_average"])
    close(r["dense_stats"]["negative"]["quartiles"], g["dense_negative_quartiles"])
    close(r["acc_dense"], g["acc_dense"])
    if nt == "hard":
        none = load(golden_dir, "cp2_neg_none")
        for k in ("loss", "loss_dense", "dq_feat", "logits_dense"):
            assert np.array_equal(g[k], none[k]), k


def test_densecl_symmetric_golden(golden_dir):
    """PROPOSED_V2 with use_symmetrical_loss (reference builder.py:944-972): both passes' losses are summed, and at an
    even step the SECOND pass's keys go into the queues."""
    g = load(golden_dir, "densecl_v2_symmetric")
    tg, tl, lmbd, lc = [float(v) for v in g["cfg_f"]]
    tot_l = tot_g = 0.0
    for sfx in ("", "_2"):
        loss_l, pos, _, best = O.densecl_local_loss(T(g["q_embed" + sfx]), T(g["k_embed" + sfx]), T(g["q_local" + sfx]),
                                                    T(g["k_local" + sfx]), T(g["q_pixel_ids" + sfx]), T(g["k_pixel_ids" + sfx]),
                                                    T(g["queue2_before"]), temp_local=tl, lmbd_coordinate=lc)
        assert np.array_equal(best.numpy(), g["pos_global_k_idx" + sfx])
        close(pos.reshape(-1, 1), g["pos_local" + sfx])
        close(loss_l, g["loss_local" + sfx], 2e-6)
        loss_g = O.densecl_global_loss(T(g["q_global" + sfx]), T(g["k_global" + sfx]), T(g["queue_before"]), tg)
        close(loss_g, g["loss_global" + sfx], 2e-6)
        tot_l, tot_g = tot_l + loss_l, tot_g + loss_g
    close((1 - lmbd) * tot_g + lmbd * tot_l, g["loss"], 2e-6)
    assert int(g["step"]) % 2 == 0
    qa, ptr = O.dequeue_and_enqueue(T(g["queue_before"]), 0, T(g["k_global_2"]))
    assert np.array_equal(qa.numpy(), g["queue_after"]) and ptr == int(g["ptr_after"])
    qa2, ptr2 = O.dequeue_and_enqueue(T(g["queue2_before"]), 0, T(g["k_local_pooled_2"]))
    assert np.array_equal(qa2.numpy(), g["queue2_after"]) and ptr2 == int(g["ptr2_after"])


def test_ema_inside_forward_golden(golden_dir):
    g = load(golden_dir, "cp2_b4_64_k64")
    m = float(g["cfg_f"][6])
    new = O.momentum_update([T(g["ema_k_before_w"]), T(g["ema_k_before_b"])], [T(g["ema_q_w"]), T(g["ema_q_b"])], m)
    assert np.array_equal(new[0].numpy(), g["ema_k_after_w"]) and np.array_equal(new[1].numpy(), g["ema_k_after_b"])


def test_queue_ema_shuffle_golden(golden_dir):
    g = load(golden_dir, "queue_ema_shuffle")
    for tag in ("plain", "wrap", "exact", "big"):
        qa, ptr = O.dequeue_and_enqueue(T(g[f"enq_{tag}_queue_before"]), int(g[f"enq_{tag}_ptr_before"]), T(g[f"enq_{tag}_keys"]))
        assert np.array_equal(qa.numpy(), g[f"enq_{tag}_queue_after"]), tag
        assert ptr == int(g[f"enq_{tag}_ptr_after"]), tag
    n = len([k for k in g if k.startswith("ema_q_")])
    pk = [T(g[f"ema_k0_{i}"]) for i in range(n)]
    pq = [T(g[f"ema_q_{i}"]) for i in range(n)]
    for rnd in (1, 2):
        pk = O.momentum_update(pk, pq, float(g["ema_m"]))
        for i in range(n):
            assert np.array_equal(pk[i].numpy(), g[f"ema_k{rnd}_{i}"]), (rnd, i)
    # the fp32 scalars the HIP kernel receives reproduce the same bits
    m32, om32 = O.ema_scalars(float(g["ema_m"]))
    k0, q0 = g["ema_k0_0"], g["ema_q_0"]
    assert np.array_equal((k0 * m32 + q0 * om32).astype(np.float32), g["ema_k1_0"])
    perm = T(g["shuf_perm"])
    x = T(g["shuf_x"])
    assert np.array_equal(O.shuffle_take(x, perm, 0, 1).numpy(), g["shuf_out"])
    assert np.array_equal(torch.argsort(perm).numpy(), g["shuf_idx_unshuffle"])
    assert np.array_equal(O.unshuffle_take(T(g["shuf_out"]), perm, 0, 1).numpy(), g["unshuf_out"])
    assert np.array_equal(g["unshuf_out"], g["shuf_x"])


@pytest.mark.parametrize("name", ["densecl_b2_128_k64", "densecl_b2_96_k64_coord"])
def test_densecl_golden(golden_dir, name):
    g = load(golden_dir, name)
    tg, tl, lmbd, lc = [float(v) for v in g["cfg_f"]]
    loss_l, pos, neg, best = O.densecl_local_loss(T(g["q_embed"]), T(g["k_embed"]), T(g["q_local"]), T(g["k_local"]),
                                                  T(g["q_pixel_ids"]), T(g["k_pixel_ids"]), T(g["queue2_before"]),
                                                  temp_local=tl, lmbd_coordinate=lc)
    assert np.array_equal(best.numpy(), g["pos_global_k_idx"])
    close(pos.reshape(-1, 1), g["pos_local"]); close(neg, g["neg_local"])
    close(loss_l, g["loss_local"], 2e-6)
    loss_g = O.densecl_global_loss(T(g["q_global"]), T(g["k_global"]), T(g["queue_before"]), tg)
    close(loss_g, g["loss_global"], 2e-6)
    close((1 - lmbd) * loss_g + lmbd * loss_l, g["loss"], 2e-6)
    close(O.queue_infonce(T(g["q_local"]).permute(0, 2, 1).reshape(-1, 128), pos.reshape(-1), T(g["queue2_before"]), tl),
          g["loss_local"], 2e-6)
    K = int(g["cfg"][3])
    qa, ptr = O.dequeue_and_enqueue(T(g["queue_before"]), 0, T(g["k_global"]))
    assert np.array_equal(qa.numpy(), g["queue_after"]) and ptr == int(g["ptr_after"])
    qa2, ptr2 = O.dequeue_and_enqueue(T(g["queue2_before"]), 0, T(g["k_local_pooled"]))
    assert np.array_equal(qa2.numpy(), g["queue2_after"]) and ptr2 == int(g["ptr2_after"])


def test_densecl_coordinate_mix_golden(golden_dir):
    """The coordinate mix of builder.py:838-855 WITH id overlap: the reference's own locals at the point where its
    :861 raises (make_goldens.run_densecl_overlap_case) -- arg-max indices equal, mixed positives to 1e-6."""
    g = load(golden_dir, "densecl_coord_overlap")
    lc = float(g["cfg_f"][3])
    queue2 = torch.nn.functional.normalize(torch.randn(128, 64, generator=torch.Generator().manual_seed(0)), dim=0)
    _, pos, _, best = O.densecl_local_loss(T(g["q_embed"]), T(g["k_embed"]), T(g["q_local"]), T(g["k_local"]),
                                           T(g["q_pixel_ids"]), T(g["k_pixel_ids"]), queue2, lmbd_coordinate=lc)
    assert int(g["overlap_pixels"].sum()) == 32
    assert np.array_equal(best.numpy(), g["pos_global_k_idx"])
    close(pos, g["pos_local"])
    # without the mix the overlapping pixels would differ: the fixture does exercise it
    _, plain, _, _ = O.densecl_local_loss(T(g["q_embed"]), T(g["k_embed"]), T(g["q_local"]), T(g["k_local"]),
                                          T(g["q_pixel_ids"]), T(g["k_pixel_ids"]), queue2, lmbd_coordinate=0.0)
    assert float((plain - pos).abs().max()) > 1e-3
    rate = O.densecl_matching_rate(T(g["q_local"]), T(g["k_local"]), T(g["q_pixel_ids"]), T(g["k_pixel_ids"]))
    assert 0.0 <= rate <= 1.0


def test_sgd_restatement_matches_torch_sgd_on_cpu():
    """oracle.sgd_momentum_step against torch.optim.SGD itself (the optimizer the reference constructs, main.py:467-477)
    on CPU: three steps with momentum and weight decay, one parameter without gradient; <= 1 ulp."""
    import torch
    from oracle import cp2_oracle as O
    torch.manual_seed(0)
    ps = [torch.randn(7, 5), torch.randn(64), torch.randn(3, 3, 2, 2)]
    ref = [torch.nn.Parameter(p.clone()) for p in ps]
    opt = torch.optim.SGD(ref, lr=0.03, momentum=0.9, weight_decay=1e-4)
    cur, bufs = [p.clone() for p in ps], [None, None, None]
    for step in range(3):
        grads = [torch.randn_like(p) for p in ps]
        grads[1] = None if step == 1 else grads[1]
        for r, g in zip(ref, grads):
            r.grad = None if g is None else g.clone()
        opt.step()
        cur, bufs = O.sgd_momentum_step(cur, grads, bufs, 0.03, 0.9, 1e-4)
        for a, r in zip(cur, ref):
            assert torch.allclose(a, r.detach(), rtol=2e-7, atol=1e-9), step
