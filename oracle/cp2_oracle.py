"""CPU oracle for the CP2 hot path -- TEST INFRASTRUCTURE ONLY.

This file is a CPU (torch fp32 / numpy) restatement of the arithmetic on the
reference's copy-paste-contrastive hot path (kimathikaai/CP2 builder.py
`MODEL.forward_cp2` / `forward_densecl` and tools/correlation_mapping.py).
It is the *checker* for the HIP kernels in cp2_amd/csrc: only `tests/`,
`__graft_entry__.smoke()` and the `cpu_baseline` leg of `bench.py` may import
it.  Nothing under `cp2_amd/` imports it and nothing here is a fallback for
the product path.

Parity pin: every function here is checked against golden vectors produced by
running the reference's own code in the build container
(tests/golden/make_goldens.py -> tests/golden/*.npz, tests/test_oracle_golden.py)
and against the reference's known-answer tests
(tests/test_correlation_mapping.py:65-77,118-130, tests/test_contrastive_metrics.py:17-57).

All citations `file:line` are relative to the reference repository root.
The functions are encoder-free: they take feature maps (b, C, h, w) as input
so kernels can be checked without the ResNet in the loop.
"""
from __future__ import annotations

import math
from typing import Dict, Optional, Sequence, Tuple

import numpy as np
import torch
import torch.nn.functional as F


# --------------------------------------------------------------------------
# a1: copy-paste composition                          builder.py:1146-1152
# --------------------------------------------------------------------------
def compose_mask(img: torch.Tensor, bg: torch.Tensor) -> Tuple[torch.Tensor, torch.Tensor]:
    """Foreground mask = pixels where background channel 0 is exactly zero;
    composed image = img * mask + bg (mask broadcast over the 3 channels)."""
    fg = torch.eq(bg[:, 0], 0).to(torch.float32)
    out = torch.add(torch.mul(img, fg[:, None]), bg)
    return out, fg


# --------------------------------------------------------------------------
# a2: centre-tap strided down-sample    builder.py:1155-1186, loader.py:39-43
# --------------------------------------------------------------------------
def strided_gather(x: torch.Tensor, stride: int) -> torch.Tensor:
    """x[..., s//2::s, s//2::s] on the last two axes."""
    o = stride // 2
    return x[..., o::stride, o::stride]


# --------------------------------------------------------------------------
# a4: masked IoU of two id maps          tools/correlation_mapping.py:103-138
# --------------------------------------------------------------------------
def masked_iou(map_a, map_b, mask_a, mask_b) -> torch.Tensor:
    """Per sample: keys = float32(id + 1) * mask over the concatenation of both
    maps (plus one leading zero); union = #distinct keys - 1 (the zero key);
    intersection = #distinct non-zero keys seen at least twice.

    The reference carries ids through float32 (torch.cat with a float zero
    column promotes int64 ids), so ids >= 2**24 may collide; that behaviour is
    kept.  A sample whose keys are all zero divides 0/0: the reference raises
    ZeroDivisionError there, the oracle returns NaN for that sample.
    """
    a = np.asarray(torch.as_tensor(map_a).reshape(len(map_a), -1).cpu())
    b = np.asarray(torch.as_tensor(map_b).reshape(len(map_b), -1).cpu())
    ma = np.asarray(torch.as_tensor(mask_a).reshape(len(mask_a), -1).cpu(), dtype=np.float32)
    mb = np.asarray(torch.as_tensor(mask_b).reshape(len(mask_b), -1).cpu(), dtype=np.float32)
    out = np.zeros(a.shape[0], dtype=np.float32)
    for n in range(a.shape[0]):
        ka = (a[n] + 1).astype(np.float32) * ma[n]
        kb = (b[n] + 1).astype(np.float32) * mb[n]
        keys = np.concatenate([np.zeros(1, np.float32), ka, kb])
        vals, counts = np.unique(keys, return_counts=True)
        union = len(vals) - 1
        inter = int(np.count_nonzero(counts[1:] > 1))
        out[n] = np.float32(inter / union) if union > 0 else np.float32("nan")
    return torch.from_numpy(out)


# --------------------------------------------------------------------------
# a3: id-equality correlation map        tools/correlation_mapping.py:141-189
# --------------------------------------------------------------------------
def correlation_map(map_a: torch.Tensor, map_b: torch.Tensor) -> Dict[str, torch.Tensor]:
    n = map_a.shape[0]
    fa = map_a.reshape(n, -1)
    fb = map_b.reshape(n, -1)
    same = fa[:, :, None] == fb[:, None, :]           # (n, P, P) bool
    ones_a = torch.ones(fa.shape, dtype=torch.float32)
    ones_b = torch.ones(fb.shape, dtype=torch.float32)
    return {
        "corr_map": same,
        "corr_map_a": same.sum(2),
        "corr_map_b": same.sum(1),
        "iou": masked_iou(fa, fb, ones_a, ones_b),
    }


# --------------------------------------------------------------------------
# a5: masked correlation map             tools/correlation_mapping.py:192-247
# --------------------------------------------------------------------------
def masked_correlation_map(map_a, map_b, mask_a, mask_b) -> Dict[str, torch.Tensor]:
    res = correlation_map(map_a, map_b)
    n = mask_a.shape[0]
    fma = mask_a.reshape(n, -1)
    fmb = mask_b.reshape(n, -1)
    both = fma[:, :, None] * fmb[:, None, :]
    corr_mask = res["corr_map"] * both
    res.update(
        corr_mask=corr_mask,
        corr_map_a_masked=corr_mask.sum(2),
        corr_map_b_masked=corr_mask.sum(1),
        iou_masked=masked_iou(map_a.reshape(n, -1), map_b.reshape(n, -1), fma, fmb),
    )
    return res


# --------------------------------------------------------------------------
# a6: correspondence weights                          builder.py:1225-1243
# --------------------------------------------------------------------------
def corr_weights(pixel_corr_map, region_corr_map, region_ids_a, region_ids_b,
                 w_pixel, w_region, w_not) -> torch.Tensor:
    """weight = w_pixel where pixel ids match, else w_region where region ids
    match and both region ids are non-zero, else 0; entries that are still 0
    then receive w_not.  (All ones for MappingType.CP2.)"""
    n = region_ids_a.shape[0]
    known = (region_ids_a.reshape(n, -1)[:, :, None] * region_ids_b.reshape(n, -1)[:, None, :]).bool()
    region = region_corr_map & known
    w = w_region * region
    w = torch.where(pixel_corr_map, torch.as_tensor(w_pixel, dtype=w.dtype), w)
    w = w + (w == 0) * w_not
    return w


# --------------------------------------------------------------------------
# a7: per-pixel normalise + masked pooling   builder.py:1261-1268,1279-1285
# --------------------------------------------------------------------------
def normalize_and_pool(feat: torch.Tensor, mask: torch.Tensor):
    """feat (b, C, h, w) or (b, C, P); mask (b, P) of 0/1 floats.
    Returns dense (b, C, P) unit vectors per pixel, pos = unit(sum over masked
    pixels), neg = unit(sum over un-masked pixels)."""
    b, c = feat.shape[:2]
    dense = F.normalize(feat.reshape(b, c, -1), dim=1)
    pos = F.normalize(torch.einsum("ncx,nx->nc", dense, mask), dim=1)
    inv = (~mask.bool()).to(torch.float32)
    neg = F.normalize(torch.einsum("ncx,nx->nc", dense, inv), dim=1)
    return dense, pos, neg


# --------------------------------------------------------------------------
# a8/a9: dense logits + dense InfoNCE        builder.py:1289-1292,1392,1431-1437
# --------------------------------------------------------------------------
def dense_logits(q_dense, k_dense) -> torch.Tensor:
    return torch.einsum("ncx,ncy->nxy", q_dense, k_dense)


def dense_infonce(logits_dense, mask_a, mask_b, temp_local=1.0, weights=None):
    """-log_softmax over the QUERY-pixel axis (dim=1 of (n, x, y)), averaged
    over the positive pairs mask_a[x]*mask_b[y] of each sample, then over the
    batch.  A sample with no positive pair gives 0/0 = NaN (kept)."""
    lg = logits_dense if weights is None else logits_dense * weights
    lg = lg / temp_local
    nll = -torch.log_softmax(lg, dim=1)
    labels = mask_a[:, :, None] * mask_b[:, None, :]
    n = lg.shape[0]
    per_sample = (nll * labels).reshape(n, -1).sum(1) / labels.reshape(n, -1).sum(1)
    return per_sample.mean(), per_sample, lg


# --------------------------------------------------------------------------
# f4: NegativeType reshaping of the negative dense logits   builder.py:1332-1386
# --------------------------------------------------------------------------
NEG_NONE, NEG_FIXED, NEG_AVERAGE, NEG_MEDIAN, NEG_HARD = 0, 1, 2, 3, 4


def reshape_negatives(logits_dense, labels_dense, negative_type=NEG_NONE, negative_scale=2.0, stats=None):
    """Raw dense logits (n, x, y) -> logits with the NEGATIVE pairs (label 0) squashed through
    2 / (1 + exp(-scale * (L - centre))) - 1, centre = 0 (FIXED), the sample's mean negative score (AVERAGE) or its
    median negative score (MEDIAN); both centres are detached statistics of the raw logits (builder.py:1298-1300).
    HARD multiplies a COPY of the selected negatives (chained advanced indexing, builder.py:1377-1381), so the
    logits the loss sees are unchanged: it is the identity, like NONE."""
    if negative_type in (NEG_NONE, NEG_HARD):
        return logits_dense
    n = logits_dense.shape[0]
    if negative_type == NEG_FIXED:
        centre = torch.zeros(n)
    else:
        st = stats if stats is not None else dense_loss_stats(logits_dense.detach(), labels_dense)
        centre = st["negative"]["average"] if negative_type == NEG_AVERAGE else st["negative"]["quartiles"][1]
    squashed = 2.0 / (1.0 + torch.exp((logits_dense - centre.detach().reshape(n, 1, 1)) * (-negative_scale))) - 1.0
    return torch.where(labels_dense.bool(), logits_dense, squashed)


# --------------------------------------------------------------------------
# a10: instance InfoNCE against the queue    builder.py:1395-1397,1414-1428
# --------------------------------------------------------------------------
def instance_infonce(q_pos, k_pos, queue, temp_global=0.2, q_neg=None, k_neg=None,
                     include_background=False):
    l_pos = (q_pos * k_pos).sum(1, keepdim=True)
    l_neg = q_pos @ queue
    parts = [l_pos, l_neg]
    if include_background:
        parts.append((q_pos * q_neg).sum(1, keepdim=True))
        parts.append((q_pos * k_neg).sum(1, keepdim=True))
    logits = torch.cat(parts, dim=1) / temp_global
    target = torch.zeros(logits.shape[0], dtype=torch.long)
    return F.cross_entropy(logits, target), logits, l_pos, l_neg


# --------------------------------------------------------------------------
# a15: logging statistics
# --------------------------------------------------------------------------
def dense_loss_stats(logits_dense, labels_dense):
    """Per-sample mean and quartiles of positive / negative pair scores.
    tools/correlation_mapping.py:11-53 (nanmean / nanquantile, linear interp)."""
    qs = torch.tensor([0.25, 0.5, 0.75])

    def stats(x):
        return {"average": x.nanmean((1, 2)), "quartiles": torch.nanquantile(x.flatten(1), qs, dim=1)}

    pos = torch.where(labels_dense.bool(), logits_dense, torch.tensor(float("nan")))
    neg = torch.where(labels_dense.bool(), torch.tensor(float("nan")), logits_dense)
    return {"positive": stats(pos), "negative": stats(neg)}


def instance_stats(l_neg):
    """builder.py:1399-1406: row mean and row quartiles of the queue logits."""
    return l_neg.mean(1), torch.quantile(l_neg, torch.tensor([0.25, 0.5, 0.75]), dim=1)


def topk_accuracy(logits, target, topk=(1, 5)):
    """builder.py:1690-1706."""
    kmax = max(topk)
    pred = logits.topk(kmax, dim=1).indices          # (n, kmax)
    hit = pred == target[:, None]
    return [hit[:, :k].any(1).float().sum() * (100.0 / logits.shape[0]) for k in topk]


def dense_argmax_accuracy(logits_dense_scaled, mask_a, mask_b):
    """builder.py:1442-1448: label at the flat arg-max pair of each sample."""
    n = logits_dense_scaled.shape[0]
    labels = (mask_a[:, :, None] * mask_b[:, None, :]).reshape(n, -1)
    idx = logits_dense_scaled.reshape(n, -1).argmax(1)
    return labels[torch.arange(n), idx].float().mean() * 100.0


# --------------------------------------------------------------------------
# whole CP2 loss section (everything after the encoders)  builder.py:1145-1448
# --------------------------------------------------------------------------
def cp2_loss_section(q_feat, k_feat, bg0, bg1, pixel_ids_a, pixel_ids_b, region_ids_a,
                     region_ids_b, queue, *, output_stride, temp_global=0.2, temp_local=1.0,
                     lmbd_dense=0.2, include_background=False,
                     w_pixel=1, w_region=1, w_not=1, with_stats=False,
                     negative_type=NEG_NONE, negative_scale=2.0, masks_and_ids=None):
    """Everything forward_cp2 computes from encoder outputs to the loss.
    q_feat/k_feat: (b, C, h', w') encoder outputs (k already un-shuffled).
    masks_and_ids = (mask_a, mask_b, pa, pb, ra, rb) already down-sampled replaces bg*/ids (encoder-free fixtures)."""
    if masks_and_ids is not None:
        mask_a, mask_b, pa, pb, ra, rb = masks_and_ids
    else:
        mask_a = strided_gather(torch.eq(bg0[:, 0], 0).float(), output_stride)
        mask_b = strided_gather(torch.eq(bg1[:, 0], 0).float(), output_stride)
        pa, pb = strided_gather(pixel_ids_a, output_stride), strided_gather(pixel_ids_b, output_stride)
        ra, rb = strided_gather(region_ids_a, output_stride), strided_gather(region_ids_b, output_stride)
    pix = masked_correlation_map(pa, pb, mask_a, mask_b)
    reg = masked_correlation_map(ra, rb, mask_a, mask_b)
    w = corr_weights(pix["corr_map"], reg["corr_map"], ra, rb, w_pixel, w_region, w_not)
    b = q_feat.shape[0]
    fma, fmb = mask_a.reshape(b, -1), mask_b.reshape(b, -1)
    q_dense, q_pos, q_neg = normalize_and_pool(q_feat, fma)
    with torch.no_grad():
        k_dense, k_pos, k_neg = normalize_and_pool(k_feat, fmb)
    raw = dense_logits(q_dense, k_dense)
    loss_ins, logits_moco, l_pos, l_neg = instance_infonce(
        q_pos, k_pos, queue, temp_global, q_neg, k_neg, include_background)
    labels = fma[:, :, None] * fmb[:, None, :]
    reshaped = reshape_negatives(raw, labels, negative_type, negative_scale)      # identity unless a NegativeType is set
    loss_den, loss_den_per_sample, lg_scaled = dense_infonce(reshaped, fma, fmb, temp_local, w)
    loss = loss_ins + loss_den * lmbd_dense
    out = dict(mask_a=fma, mask_b=fmb, pixel_ids_a=pa, pixel_ids_b=pb, region_ids_a=ra, region_ids_b=rb,
               iou=reg["iou"], iou_masked=reg["iou_masked"], pixel_iou=pix["iou"],
               pixel_iou_masked=pix["iou_masked"], corr_weights=w,
               q_dense=q_dense, k_dense=k_dense, q_pos=q_pos, k_pos=k_pos, q_neg=q_neg, k_neg=k_neg,
               logits_dense_raw=raw, logits_dense_reshaped=reshaped, logits_dense_scaled=lg_scaled, logits_moco=logits_moco, l_pos=l_pos, l_neg=l_neg,
               loss_instance=loss_ins, loss_dense=loss_den, loss_dense_per_sample=loss_den_per_sample,
               loss=loss)
    if with_stats:
        out["dense_stats"] = dense_loss_stats(raw.detach(), labels)
        out["instance_neg_mean"], out["instance_neg_quartiles"] = instance_stats(l_neg.detach())
        tgt = torch.zeros(b, dtype=torch.long)
        out["acc1"], out["acc5"] = topk_accuracy(logits_moco.detach(), tgt)
        out["acc_dense"] = dense_argmax_accuracy(lg_scaled.detach(), fma, fmb)
    return out


# --------------------------------------------------------------------------
# a11: momentum (EMA) update of the key encoder          builder.py:557-567
# --------------------------------------------------------------------------
def momentum_update(params_k: Sequence[torch.Tensor], params_q: Sequence[torch.Tensor], m: float):
    """theta_k <- theta_k * m + theta_q * (1 - m); two products then one sum,
    each rounded to fp32 (no fused multiply-add); the scalar (1.0 - m) is
    formed in double precision first, exactly as the reference's Python does."""
    one_minus = 1.0 - m
    return [pk * m + pq * one_minus for pk, pq in zip(params_k, params_q)]


def ema_scalars(m: float) -> Tuple[np.float32, np.float32]:
    """The two fp32 multipliers the EMA uses: fp32(m), fp32(1.0 - m)."""
    return np.float32(m), np.float32(1.0 - m)


# --------------------------------------------------------------------------
# a17: optimizer step of the training loop        main.py:467-477, :640-642
# --------------------------------------------------------------------------
def sgd_momentum_step(params: Sequence[torch.Tensor], grads: Sequence[Optional[torch.Tensor]],
                      bufs: Sequence[Optional[torch.Tensor]], lr: float, momentum: float, weight_decay: float):
    """torch.optim.SGD(params, lr, momentum, weight_decay) as the reference builds it (dampening 0, no Nesterov),
    one step, restated on the public algorithm of torch/optim/sgd.py:
        g' = g + wd * p ;  buf = g' on the first step, else buf * momentum + g' ;  p = p - lr * buf
    A parameter without gradient is left alone.  Returns (new params, new bufs).  Evaluated in float64 and rounded
    to fp32 once per statement, which is what a fused multiply-add does up to double rounding (so GPU / CPU torch
    kernels agree with it to 1 ulp, not necessarily bit for bit)."""
    out_p, out_b = [], []
    for p, g, b in zip(params, grads, bufs):
        if g is None:
            out_p.append(p.clone()); out_b.append(None if b is None else b.clone())
            continue
        p64, g64 = p.double(), g.double()
        if weight_decay != 0:
            g64 = (g64 + float(np.float32(weight_decay)) * p64).float().double()
        if momentum != 0:
            b64 = g64 if b is None else ((b.double() * float(np.float32(momentum))).float().double() + g64).float().double()
            out_b.append(b64.float())
            g64 = b64
        else:
            out_b.append(None)
        out_p.append((p64 - float(np.float32(lr)) * g64).float())
    return out_p, out_b


# --------------------------------------------------------------------------
# a13: queue enqueue with wrap-around                    builder.py:569-587
# --------------------------------------------------------------------------
def dequeue_and_enqueue(queue: torch.Tensor, ptr: int, keys: torch.Tensor) -> Tuple[torch.Tensor, int]:
    """queue (C, K): key i goes to column (ptr + i) mod K; returns the new queue
    and the advanced pointer.  keys (n, C) are already gathered over ranks."""
    c, k = queue.shape
    n = keys.shape[0]
    assert n <= k
    cols = (ptr + torch.arange(n)) % k
    out = queue.clone()
    out[:, cols] = keys.t()
    return out, (ptr + n) % k


# --------------------------------------------------------------------------
# a12: shuffle-BN index plan                             builder.py:609-649
# --------------------------------------------------------------------------
def shuffle_take(x_gather: torch.Tensor, idx_shuffle: torch.Tensor, rank: int, world: int):
    """Rows of the all-gathered batch this rank feeds to its key encoder."""
    return x_gather[idx_shuffle.view(world, -1)[rank]]


def unshuffle_take(k_gather: torch.Tensor, idx_shuffle: torch.Tensor, rank: int, world: int):
    """Rows of the all-gathered keys that restore this rank's original order."""
    idx_unshuffle = torch.argsort(idx_shuffle)
    return k_gather[idx_unshuffle.view(world, -1)[rank]]


# --------------------------------------------------------------------------
# a16: DenseCL losses (config 5)                          builder.py:760-910
# --------------------------------------------------------------------------
def contrastive_head(pos, neg, temperature):
    """builder.py:150-176: CE over [pos | neg] / T with target 0."""
    logits = torch.cat([pos, neg], dim=1) / temperature
    return F.cross_entropy(logits, torch.zeros(pos.shape[0], dtype=torch.long))


def densecl_global_loss(q_global, k_global, queue, temp_global=0.2):
    """builder.py:760-772."""
    pos = (q_global * k_global).sum(1, keepdim=True)
    return contrastive_head(pos, q_global @ queue, temp_global)


def densecl_local_loss(q_embed, k_embed, q_local, k_local, q_pixel_ids, k_pixel_ids, queue2,
                       temp_local=0.2, lmbd_coordinate=0.0):
    """builder.py:808-910.  q_embed/k_embed (b, Cb, S2) and q_local/k_local
    (b, C, S2) are unit-normalised over the channel axis; pixel ids (b, s, s).
    Positive of query pixel x = local similarity with the key pixel that
    maximises the BACKBONE similarity; where the id maps overlap that score
    is mixed with the summed local similarity over id-matching key pixels."""
    backbone_sim = torch.einsum("ncx,ncy->nxy", q_embed, k_embed)
    best = backbone_sim.argmax(dim=2)                         # (b, S2)
    local_sim = torch.einsum("ncx,ncy->nxy", q_local, k_local)
    pos = torch.gather(local_sim, 2, best[:, :, None])[:, :, 0]
    corr = correlation_map(q_pixel_ids, k_pixel_ids)["corr_map"]
    overlap = corr.sum(-1) > 0
    coord = (local_sim * corr).sum(-1)
    pos = torch.where(overlap, pos * (1 - lmbd_coordinate) + coord * lmbd_coordinate, pos)
    b, c, s2 = q_local.shape
    rows = q_local.permute(0, 2, 1).reshape(b * s2, c)
    neg = rows @ queue2
    loss = contrastive_head(pos.reshape(-1, 1), neg, temp_local)
    return loss, pos, neg, best


def densecl_matching_rate(q_local, k_local, q_pixel_ids, k_pixel_ids):
    """builder.py:856-864, as evidently intended: over the query pixels whose id occurs in the key map, how often the
    arg-max of the local similarity row is the arg-max (= first match) of the id-equality row; -1 without overlap.
    The reference's own expression cannot be pinned: `corr_map[overlap_pixels, :].max(dim=2)` (:861) indexes dim 2 of a
    2-D tensor and raises IndexError whenever there is overlap (see tests/golden/make_goldens.py
    run_densecl_overlap_case)."""
    corr = correlation_map(q_pixel_ids, k_pixel_ids)["corr_map"]
    overlap = corr.sum(-1) > 0
    if int(overlap.sum()) == 0:
        return -1.0
    local_sim = torch.einsum("ncx,ncy->nxy", q_local, k_local)
    corr_max = corr[overlap].float().argmax(dim=1)
    sim_max = local_sim[overlap].argmax(dim=1)
    return float((corr_max == sim_max).float().mean())


def queue_infonce(rows, pos, queue, temperature):
    """The rows-vs-queue InfoNCE on its own (T19): mean_r [lse_r - pos_r/T]."""
    return contrastive_head(pos.reshape(-1, 1), rows @ queue, temperature)


# --------------------------------------------------------------------------
# f1: two-crop augmentation with explicit parameters   loader.py:39-43,50-118; main.py:204-225
# --------------------------------------------------------------------------
def pixel_id_map(hs: int, ws: int, stride: int = 1) -> np.ndarray:
    """loader.py:66-73: arange(1 .. H*W) -> rescale_ids(stride) (centre taps) -> resized back to (H, W) with
    INTER_NEAREST_EXACT, whose published rule is  source = floor((dst + 0.5) * small / big)."""
    ids = np.arange(1, hs * ws + 1, dtype=np.int64).reshape(hs, ws)
    if stride <= 1:
        return ids
    small = ids[stride // 2:: stride, stride // 2:: stride]
    yi = ((2 * np.arange(hs) + 1) * small.shape[0]) // (2 * hs)
    xi = ((2 * np.arange(ws) + 1) * small.shape[1]) // (2 * ws)
    return small[yi][:, xi]


def crop_resize_flip(src: np.ndarray, region: Optional[np.ndarray], box, flip: bool, h: int, w: int, id_stride: int = 1):
    """One sample.  src: (3, Hs, Ws) float32 in [0,1]; box = (top, left, ch, cw).  Image: bilinear with half-pixel
    centres and replicated edges, every product / sum rounded to fp32 in the kernel's order; ids: nearest neighbour,
    source cell floor(dst * crop / out); then the horizontal flip of both."""
    top, left, ch, cw = [int(v) for v in box]
    hs, ws = src.shape[1:]
    f32 = np.float32
    ys, xs = np.arange(h), np.arange(w)
    xr = (w - 1 - xs) if flip else xs
    sy = top + (ys * ch) // h
    sx = left + (xr * cw) // w
    pid_map = pixel_id_map(hs, ws, id_stride)
    pix = pid_map[sy][:, sx]
    # loader.py:75-83: the region map goes through the same rescale_ids + INTER_NEAREST_EXACT round trip as the pixel
    # ids, i.e. it is read at the very cell whose pixel id was kept (pixel id - 1 = row * ws + column of that cell)
    reg = region.reshape(-1)[pix - 1] if region is not None else pix
    fy = (ys.astype(f32) + f32(0.5)) * (f32(ch) / f32(h)) - f32(0.5)
    fx = (xr.astype(f32) + f32(0.5)) * (f32(cw) / f32(w)) - f32(0.5)
    cy = np.minimum(np.maximum(fy, f32(0)), f32(ch - 1)).astype(f32)
    cx = np.minimum(np.maximum(fx, f32(0)), f32(cw - 1)).astype(f32)
    y0, x0 = np.floor(cy).astype(np.int64), np.floor(cx).astype(np.int64)
    y1, x1 = np.minimum(y0 + 1, ch - 1), np.minimum(x0 + 1, cw - 1)
    wy, wx = (cy - y0.astype(f32)).astype(f32)[:, None], (cx - x0.astype(f32)).astype(f32)[None, :]
    s = src.astype(f32)
    g = lambda yy, xx: s[:, top + yy][:, :, left + xx]          # noqa: E731
    one = f32(1.0)
    r0 = (g(y0, x0) * (one - wx)).astype(f32) + (g(y0, x1) * wx).astype(f32)
    r1 = (g(y1, x0) * (one - wx)).astype(f32) + (g(y1, x1) * wx).astype(f32)
    img = (r0.astype(f32) * (one - wy)).astype(f32) + (r1.astype(f32) * wy).astype(f32)
    return img.astype(f32), pix, reg


def quantize_u8(img: np.ndarray) -> np.ndarray:
    """The fp32 view as the uint8 image handed to the photometric stages: value * 255 rounded half up, (H, W, 3).
    (cv2.resize would produce the foreground's uint8 image in the reference; its fixed-point arithmetic is not restated:
    parity-unpinned)"""
    q = np.floor(img.astype(np.float32) * np.float32(255.0) + np.float32(0.5))
    return np.clip(q, 0, 255).astype(np.uint8).transpose(1, 2, 0)


def erase_rect(img: np.ndarray, rect) -> np.ndarray:
    """RandomErasing(value=0) of one sample (3, H, W): exact zeros in the rectangle (main.py:218-224)."""
    top, left, hh, ww = [int(v) for v in rect]
    out = img.copy()
    out[:, top:top + hh, left:left + ww] = 0.0
    return out
