"""Autograd front end of the fused CP2 loss section.

`cp2_loss_section` is everything reference `MODEL.forward_cp2` computes between the
two encoder calls and the returned loss (builder.py:1261-1268, 1279-1292, 1392-1448)
as ten kernel launches with no host synchronisation (normalise + pool both maps, pool
finalize, rows-vs-queue + finalize, dense forward + per-sample fold, dense backward, feature
backward, the three quartile sets, every returned / logged scalar).  The gradient with respect
to the query feature map is produced inside the forward pass (the queue and the P x P logits are
each visited once more at most), so `backward` is a single scale.
"""
from __future__ import annotations

from dataclasses import dataclass
from typing import Optional, Tuple

import torch

from . import ops


NEG_NONE, NEG_FIXED, NEG_AVERAGE, NEG_MEDIAN, NEG_HARD = 0, 1, 2, 3, 4     # builder.NegativeType values


@dataclass
class CP2LossOutputs:
    loss: torch.Tensor               # scalar, differentiable w.r.t. q_feat
    loss_instance: torch.Tensor      # scalars below are detached
    loss_dense: torch.Tensor
    acc_dense: torch.Tensor          # builder.py:1442-1448
    acc1: torch.Tensor               # builder.py:1441 (top-1 / top-5 of the instance logits)
    acc5: torch.Tensor
    k_pos: torch.Tensor              # [B,C] keys to enqueue (builder.py:1426)
    q_pos: torch.Tensor
    dense_sample: torch.Tensor       # [B,8] Sa, Sb, loss_n, mean +score, mean -score, arg-max label
    instance_pos: torch.Tensor       # [B] raw positive logit q_pos.k_pos
    lneg: Optional[torch.Tensor] = None                   # [B,K] raw queue logits (want_lneg / want_quartiles)
    # logging quartiles (want_quartiles): [3,B] each = torch.(nan)quantile(..., [.25,.5,.75]) of the reference
    dense_pos_quartiles: Optional[torch.Tensor] = None    # tools/correlation_mapping.py:16-53, positive pairs
    dense_neg_quartiles: Optional[torch.Tensor] = None    # ... negative pairs
    instance_neg_quartiles: Optional[torch.Tensor] = None # builder.py:1401-1406
    instance_neg_mean: Optional[torch.Tensor] = None      # [B] builder.py:1400
    scalars: Optional[torch.Tensor] = None                # [24] every returned / logged scalar (cp2_step_scalars layout, S_* indices)
    iou: Optional[torch.Tensor] = None                    # [B] IoU / masked IoU of the down-sampled region-id maps (tail["iou"])
    iou_masked: Optional[torch.Tensor] = None


# positions in the cp2_step_scalars vector (include/cp2hip.h)
S_LOSS, S_LOSS_INS, S_LOSS_DENSE, S_ACC1, S_ACC5, S_ACC_DENSE, S_POS_SCORE, S_NEG_SCORE, S_INS_POS = range(9)
S_VAR_SRC, S_VAR_TGT, S_DPOS_Q, S_DNEG_Q, S_INS_Q, S_INS_NEG_MEAN = 9, 10, 11, 14, 17, 20


class _CP2LossFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, q_feat, k_feat, mask_a, mask_b, queue, cfg):
        (temp_global, temp_local, lmbd_dense, include_background, ids, weights, want_lneg, want_quart, negative, k_row, tail) = cfg
        want_lneg = want_lneg or want_quart
        B = q_feat.shape[0]
        need_grad = q_feat.requires_grad
        # query and key map in one launch; the key side through the un-shuffle index when the caller passes one
        q_dense, k_dense, q_inv, q_part, k_part = ops.feat_normalize_pool_pair(q_feat, k_feat, mask_a, mask_b, k_row)
        P = q_dense.shape[2]
        q_pos, q_neg, q_norms, k_pos, k_neg, extras = ops.pool_finalize(q_part, k_part, P)
        ext = extras if include_background else extras[:, :1].contiguous()
        C = q_pos.shape[1]
        # finalize=False: the merge of its splits rides in the dense loss's post-pass launch (ops.loss_post below)
        ins = ops.rowkey_infonce(q_pos, (1, C, 0, 1), B, queue, ext, temp_global,
                                 grad_scale=(1.0 / B) if need_grad else None, want_lneg=want_lneg, lneg_row_major=True,
                                 finalize=False)
        # NegativeType (reference builder.py:1332-1386): the negative pairs' logits are squashed around 0 (FIXED) or
        # around the sample's mean / median negative score, which needs one un-reshaped pass first; HARD edits a copy
        # in the reference, i.e. it is the identity.
        neg = None
        ntype, nscale = negative
        if ntype == NEG_FIXED:
            neg = (nscale, None)
        elif ntype in (NEG_AVERAGE, NEG_MEDIAN):
            pre = ops.dense_infonce_fwd(q_dense, k_dense, mask_a, mask_b, temp_local, ids, weights,
                                        want_logits=(ntype == NEG_MEDIAN))
            if ntype == NEG_AVERAGE:
                centre = pre.sample_scal[:, 4].contiguous()
            else:
                centre = ops.masked_quantiles(pre.logits, P * P, 1, B, P * P, mask_a=mask_a, mask_b=mask_b, want=0)[1].contiguous()
            neg = (nscale, centre)
        den = ops.dense_infonce_fwd(q_dense, k_dense, mask_a, mask_b, temp_local, ids, weights, want_logits=want_quart,
                                    negative=neg, defer_post=ins.pending is not None)
        # the two tails (and, with quartile logging, the step's three quartile sets) in one launch: cp2_step_post / cp2_loss_post
        qs = (None, None, None)
        lneg_mean = None
        jobs = None
        if want_quart:
            K = queue.shape[1]
            dense = dict(x=den.logits, stride_row=P * P, stride_elem=1, R=B, N=P * P, mask_a=mask_a, mask_b=mask_b)
            row_form = K <= ops.QUANTILES_ROW_MAX and P * P <= ops.QUANTILES_ROW_MAX
            lneg_mean = torch.empty(B, dtype=torch.float32, device=q_feat.device) if row_form else ins.lneg.mean(1)
            jobs = [dict(dense, want=1), dict(dense, want=0),
                    dict(x=ins.lneg, stride_row=K, stride_elem=1, R=B, N=K, mean_out=lneg_mean if row_form else None)]
        outs = ops.loss_post(ins, den, jobs)
        if outs is not None:
            qs = outs
        if need_grad:
            # the dense kernel's split gradients stay un-summed: feat_bwd_fused adds them, and computes the pooled-vector
            # backward per workgroup (round 2: dense_grad_sum + pool_bwd + two fill kernels for dE)
            g_part, S = ops.dense_infonce_bwd(q_dense, k_dense, mask_a, mask_b, temp_local, den, lmbd_dense / B, ids, weights,
                                              negative=neg, keep_partials=True)
            dq = ops.feat_bwd_fused(q_dense, q_inv, mask_a, g_part, S, ins.drows, ins.dE, q_pos, q_neg, k_pos, k_neg, q_norms,
                                    include_background, q_feat)
            ctx.save_for_backward(dq)
        # every scalar the step returns or logs: one launch (the loss combination included) -- and, when the caller hands over
        # the step's tail (`tail`), the keys' enqueue and the logged IoUs ride in the same launch (cp2_step_tail)
        iou = iou_masked = None
        if tail is None:
            scal = ops.step_scalars(ins.loss, ins.cnt_gt, ext, den.sample_scal, q_pos, k_pos, lmbd_dense, qs[0], qs[1], qs[2], lneg_mean)
        else:
            enq = None
            if tail.get("enqueue") is not None:
                queue_ptr, gather = tail["enqueue"]
                enq = (queue, (gather(k_pos) if gather is not None else k_pos).contiguous(), queue_ptr)
            io = tail.get("iou")
            scal, iou, iou_masked = ops.step_tail(ins.loss, ins.cnt_gt, ext, den.sample_scal, q_pos, k_pos, lmbd_dense, qs[0], qs[1],
                                                  qs[2], lneg_mean, enqueue=enq,
                                                  iou=None if io is None else (io[0], io[1], io[2], mask_a, mask_b))
        outs = (scal[S_LOSS], scal, k_pos, q_pos, den.sample_scal, extras[:, 0])
        if want_lneg:
            outs = outs + (ins.lneg,)
        if want_quart:
            outs = outs + (qs[0], qs[1], qs[2], lneg_mean)
        if iou is not None:
            outs = outs + (iou, iou_masked)
        ctx.mark_non_differentiable(*outs[1:])
        return outs

    @staticmethod
    def backward(ctx, g_loss, *unused):
        (dq,) = ctx.saved_tensors
        return dq * g_loss, None, None, None, None, None


def cp2_loss_section(q_feat: torch.Tensor, k_feat: torch.Tensor, mask_a: torch.Tensor, mask_b: torch.Tensor,
                     queue: torch.Tensor, *, temp_global: float = 0.2, temp_local: float = 1.0,
                     lmbd_dense: float = 0.2, include_background: bool = False, ids=None,
                     weights: Tuple[float, float, float] = (1.0, 1.0, 1.0), want_lneg: bool = False,
                     want_quartiles: bool = False, negative_type: int = 0, negative_scale: float = 2.0,
                     k_row: Optional[torch.Tensor] = None, tail: Optional[dict] = None) -> CP2LossOutputs:
    """q_feat / k_feat: encoder outputs [B,128,h,w] (NCHW or channels-last, fp32; no grad into k); mask_a / mask_b: [B,P]
    down-sampled foreground masks; queue [128,K].  k_row (int64 [B], optional): sample n's key features are row k_row[n]
    of k_feat -- the un-shuffle of builder.py:649 without a gather launch; None: k_feat is already in sample order.
    ids = (pixel_ids_a, pixel_ids_b, region_ids_a, region_ids_b) int64 [B,P] when the
    correspondence weights are not all one (reference builder.py:1225-1243).
    tail (the training step): {"enqueue": (queue_ptr, gather_fn or None), "iou": (region_ids_a, region_ids_b [B,H,W], stride)}
    -- either entry optional -- puts the enqueue of the keys (all ranks' keys through gather_fn; reference builder.py:1426,
    :569-587) and the IoUs of the down-sampled id maps (:1204-1219) into the launch that forms the step's scalars; the
    queue is then updated when this function returns, and the outputs carry `iou` / `iou_masked`."""
    if ids is not None and tuple(float(w) for w in weights) == (1.0, 1.0, 1.0):
        ids = None                                   # all weights one: the predicate is never needed
    if int(negative_type) not in (NEG_NONE, NEG_FIXED, NEG_AVERAGE, NEG_MEDIAN, NEG_HARD):
        raise ValueError(f"negative_type {negative_type!r}")
    cfg = (float(temp_global), float(temp_local), float(lmbd_dense), bool(include_background), ids,
           tuple(float(w) for w in weights), bool(want_lneg), bool(want_quartiles), (int(negative_type), float(negative_scale)),
           k_row, tail)
    outs = _CP2LossFn.apply(q_feat, k_feat.detach(), mask_a, mask_b, queue, cfg)
    loss, scal, k_pos, q_pos, sample, ins_pos = outs[:6]
    res = CP2LossOutputs(loss=loss, loss_instance=scal[S_LOSS_INS], loss_dense=scal[S_LOSS_DENSE], acc_dense=scal[S_ACC_DENSE],
                         acc1=scal[S_ACC1], acc5=scal[S_ACC5], k_pos=k_pos, q_pos=q_pos, dense_sample=sample, instance_pos=ins_pos,
                         scalars=scal)
    i = 6
    if want_lneg or want_quartiles:
        res.lneg = outs[i]
        i += 1
    if want_quartiles:
        res.dense_pos_quartiles, res.dense_neg_quartiles, res.instance_neg_quartiles, res.instance_neg_mean = outs[i:i + 4]
        i += 4
    if tail is not None and tail.get("iou") is not None:
        res.iou, res.iou_masked = outs[i:i + 2]
    return res
