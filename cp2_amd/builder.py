"""`builder.MODEL` for MI355X: the operator surface reference `main.py` drives
(kimathikaai/CP2 builder.py:277-304 constructor, :651-665 forward, :1608 on_train_epoch_end,
:1710 concat_all_gather, enums :30,40,140), with every non-encoder op of the hot path running
as a hand-written gfx950 kernel from libcp2hip.so:

    reference (builder.py)                         here
    :1146-1186 mask, compose, strided slices       ops.compose_mask / ops.strided_gather
    :1204-1257 correlation maps, IoUs (+4b syncs)  ops.corr_iou (device, no sync)
    :557-567   EMA python loop (~600 launches)     ops.ema_flat over flat parameter buffers (1 launch)
    :609-649   shuffle-BN gather / scatter         dist.concat_all_gather + ops.gather_rows
    :1261-1292, 1392-1448  normalise, pool, dense + instance InfoNCE, backward
                                                   functional.cp2_loss_section (f32 MFMA, 10 launches)
    :569-587   enqueue (host sync on the pointer)  ops.enqueue (pointer stays on the device)

The encoders (ResNet + ASPP/FCN head) stay in PyTorch-ROCm.  There is no CPU path: forward()
raises unless the model and its inputs are on the GPU and libcp2hip.so is built.
"""
from __future__ import annotations

import copy
from enum import Enum
from typing import Optional

import torch
import torch.nn as nn
import torch.nn.functional as F

from . import dist as cdist
from . import functional as CF
from . import ops
from .dist import concat_all_gather  # noqa: F401  (module-level name main.py / callers import from builder)
from .encoder import build_segmentor
from .pretrain_types import PretrainType


DDP_BUCKET_MB = 25        # bucket_cap_mb main.py / bench.py hand DistributedDataParallel (torch's default)


class BackboneType(Enum):
    DEEPLABV3 = 0
    UNET_ENCODER_ONLY = 1
    UNET_TRUNCATED = 2


class MappingType(Enum):
    CP2 = 0
    PIXEL_ID = 1
    REGION_ID = 2
    PIXEL_REGION_ID = 3


class NegativeType(Enum):
    NONE = 0
    FIXED = 1
    AVERAGE = 2
    MEDIAN = 3
    HARD = 4


class AverageMeter(object):
    """Running value / average (reference builder.py:51-73)."""

    def __init__(self, name, fmt=":f"):
        self.name, self.fmt = name, fmt
        self.reset()

    def reset(self):
        self.val = self.avg = self.sum = self.count = 0

    def update(self, val, n=1):
        self.val = val
        self.sum += val * n
        self.count += n
        self.avg = self.sum / self.count

    def __str__(self):
        return ("{name} {val" + self.fmt + "} ({avg" + self.fmt + "})").format(**self.__dict__)


class DenseCLNeck(nn.Module):
    """fc-relu-fc on the pooled map and conv-relu-conv on the grid, plus predictors
    (reference builder.py:179-274; same sub-module names so checkpoints interchange)."""

    def __init__(self, in_channels, hid_channels, out_channels, num_grid=None):
        super().__init__()
        self.avgpool_global = nn.AdaptiveAvgPool2d((1, 1))
        mlp = lambda a, b, c: nn.Sequential(nn.Linear(a, b), nn.ReLU(inplace=True), nn.Linear(b, c))  # noqa: E731
        # encoder.Conv2d = nn.Conv2d (same parameter names) that reads the bf16 weight image under bf16 autocast and sends wide
        # 1x1 layers through the GEMM / cp2_wgrad1x1 nodes, like the backbone's layers
        from .encoder import Conv2d
        # (ReLU out of place: the GEMM-routed 1x1 node returns a view of its matrix product, which autograd refuses to modify)
        cnv = lambda a, b, c: nn.Sequential(Conv2d(a, b, 1), nn.ReLU(inplace=False), Conv2d(b, c, 1))  # noqa: E731
        self.global_projector = mlp(in_channels, hid_channels, out_channels)
        self.global_predictor = mlp(out_channels, hid_channels, out_channels)
        self.with_pool = num_grid is not None
        if self.with_pool:
            self.pool = nn.AdaptiveAvgPool2d((num_grid, num_grid))
        self.local_projector = cnv(in_channels, hid_channels, out_channels)
        self.local_predictor = cnv(out_channels, hid_channels, out_channels)
        self.avgpool_local = nn.AdaptiveAvgPool2d((1, 1))
        for m in self.modules():
            if isinstance(m, (nn.Conv2d, nn.Linear)):
                nn.init.xavier_normal_(m.weight)
                if m.bias is not None:
                    nn.init.constant_(m.bias, 0)

    def forward(self, x):
        g = self.avgpool_global(x).flatten(1)
        g_proj = self.global_projector(g)
        g_pred = self.global_predictor(g_proj)
        if self.with_pool:
            x = self.pool(x)
        l_proj = self.local_projector(x)
        l_pred = self.local_predictor(l_proj)
        return {"x_global_proj": g_proj, "x_local_proj": l_proj, "x_global_pred": g_pred, "x_local_pred": l_pred,
                "x_avgpool_local_pred": self.avgpool_local(l_pred).flatten(1),
                "x_avgpool_local_proj": self.avgpool_local(l_proj).flatten(1)}


# Raw-logit buffer of the rows-vs-queue score statistics (logging): see _queue_infonce_chunked.  Eight samples of 196 pixels
# against 65536 keys (411 MB) per group at the DenseCL shapes: groups of two samples (103 MB, cache-resident) were measured
# first and lost -- the loss kernel on 392 rows runs at a third of its 6272-row rate and a 392-workgroup radix select leaves
# half the chip idle: 16 x (105 + 120) us against 0.6 + 0.5 ms for the whole batch (bench.py --workload cfg5, round 4).
STATS_CHUNK_BYTES = 448 << 20


def _queue_infonce_chunked(rows, ext, queue, temperature, layout, R, need, total_rows=None):
    """rows-vs-queue InfoNCE WITH the per-row score statistics for row counts whose raw logits should not all exist at once
    (DenseCL local loss: 6272 x 65536 fp32 = 1.64 GB, which the reference materialises and sorts, builder.py:871-886).  The
    rows are walked in groups of whole samples: each group's loss / gradient launch writes its raw logits into ONE re-used
    buffer of at most STATS_CHUNK_BYTES, the radix-select launch behind it reads them back -- same kernels, same per-row
    results, bit for bit, as the one-shot form; the queue's bf16 split is made once.  `need` None: statistics only (no loss
    outputs wanted: forward-only launches).  Returns (loss, drows, dE, neg_mean [R], neg_quartiles [3,R])."""
    C, K = queue.shape
    per = max(1, STATS_CHUNK_BYTES // (4 * K))
    if rows.dim() == 3:
        b, _, S2 = rows.shape
        m = max(1, per // S2)
        spans = [(n0, min(b, n0 + m), S2) for n0 in range(0, b, m)]
    else:
        spans = [(r0, min(R, r0 + per), 1) for r0 in range(0, R, per)]
    dev = rows.device
    lbuf = torch.empty(max((a1 - a0) * rp for a0, a1, rp in spans) * K, dtype=torch.float32, device=dev)
    drows = torch.empty_like(rows) if need else None
    neg_mean = torch.empty(R, dtype=torch.float32, device=dev)
    row_form = K <= ops.QUANTILES_ROW_MAX
    loss_rows, dE, quart = [], [], []
    ksplit = None
    precision = "bf16x3" if (total_rows or R) >= 1024 else "f32"     # what "auto" picks for the whole batch (a group alone may fall below)
    for a0, a1, rp in spans:
        r0, Rc = a0 * rp, (a1 - a0) * rp
        res = ops.rowkey_infonce(rows[a0:a1], layout, Rc, queue, ext[r0:r0 + Rc], temperature,
                                 grad_scale=(1.0 / R) if need else None, want_lneg=True, lneg_row_major=True,
                                 precision=precision, ksplit=ksplit, ksplit_ready=ksplit is not None, lneg_out=lbuf,
                                 drows_out=drows[a0:a1] if need else None)
        ksplit = res.ksplit
        loss_rows.append(res.loss_rows)
        if need:
            dE.append(res.dE)
        mean_out = neg_mean[r0:r0 + Rc]
        quart.append(ops.masked_quantiles_multi([dict(x=res.lneg, stride_row=K, stride_elem=1, R=Rc, N=K,
                                                      mean_out=mean_out if row_form else None)])[0])
        if not row_form:
            mean_out.copy_(res.lneg.mean(1))
    return torch.cat(loss_rows).mean(), drows, (torch.cat(dE) if need else None), neg_mean, torch.cat(quart, dim=1)


@torch.no_grad()
def row_score_stats_over_ranks(rows: torch.Tensor, queue: torch.Tensor, src: int = 0) -> torch.Tensor:
    """[mean, lower, median, upper] over rank `src`'s rows of each row's mean / quartiles of its raw queue logits (the DenseCL
    score statistics the reference computes on rank 0 alone, builder.py:875-886: a torch.quantile over all 411 M logits,
    every step, while the other ranks wait for rank 0 at the first gradient bucket).  Here rank `src` broadcasts its rows
    (b x 128 x S^2 floats: 3.2 MB), every rank takes the statistics of an equal share of them against its own replica of the
    queue (identical on every rank), and one all-reduce of four sums brings them together: 1/W of the work per rank, nobody
    a straggler.  Per-row quartiles are the same kernels' results as on one rank (bit for bit); the mean over the rows is
    summed in another order.  rows: [b,C,S2] (every rank passes its own; only rank `src`'s are used)."""
    import torch.distributed as tdist
    b, C, S2 = rows.shape
    K = queue.shape[1]
    w, r = cdist.world_size(), cdist.rank()
    src_rows = rows.detach().float().contiguous()
    if w > 1:
        if r != src:
            src_rows = torch.empty_like(src_rows)
        cdist.COLLECTIVES.note("score statistics: broadcast of rank 0's rows", tdist.broadcast(src_rows, src, async_op=True)).wait()
    n0, n1 = (b * r) // w, (b * (r + 1)) // w
    sums = torch.zeros(4, dtype=torch.float32, device=rows.device)
    if n1 > n0:
        part = src_rows[n0:n1]
        R = (n1 - n0) * S2
        ext = torch.zeros((R, 1), dtype=torch.float32, device=rows.device)
        layout = (S2, C * S2, 1, S2)
        if R * K * 4 > STATS_CHUNK_BYTES:
            _, _, _, neg_mean, quart = _queue_infonce_chunked(part, ext, queue, 1.0, layout, R, False, total_rows=b * S2)
        else:
            res = ops.rowkey_infonce(part, layout, R, queue, ext, 1.0, grad_scale=None, want_lneg=True, lneg_row_major=True,
                                     precision="bf16x3" if b * S2 >= 1024 else "f32")
            row_form = K <= ops.QUANTILES_ROW_MAX
            neg_mean = torch.empty(R, dtype=torch.float32, device=rows.device) if row_form else res.lneg.mean(1)
            quart = ops.masked_quantiles_multi([dict(x=res.lneg, stride_row=K, stride_elem=1, R=R, N=K,
                                                     mean_out=neg_mean if row_form else None)])[0]
        sums = torch.cat([neg_mean.sum().reshape(1), quart.sum(1)])
    if w > 1:
        cdist.COLLECTIVES.note("score statistics: all-reduce of the four sums", tdist.all_reduce(sums, async_op=True)).wait()
    return sums / float(b * S2)


class _QueueInfoNCEFn(torch.autograd.Function):
    """InfoNCE of row vectors against a queue (reference ContrastiveHead, builder.py:150-176, fed by
    :762-772 or :866-873): one fused kernel pass, gradients for rows and positives produced in forward."""

    @staticmethod
    def forward(ctx, rows, pos, queue, temperature, layout, R, stats):
        need = rows.requires_grad or pos.requires_grad
        K = queue.shape[1]
        ext = pos.reshape(R, 1).contiguous()
        if stats is not None and R * K * 4 > STATS_CHUNK_BYTES:
            loss, drows, dE, stats["neg_mean"], stats["neg_quartiles"] = _queue_infonce_chunked(rows, ext, queue, temperature,
                                                                                                 layout, R, need)
        else:
            res = ops.rowkey_infonce(rows, layout, R, queue, ext, temperature, grad_scale=(1.0 / R) if need else None,
                                     want_lneg=stats is not None, lneg_row_major=stats is not None)
            loss, drows, dE = res.loss, res.drows, res.dE
            if stats is not None:   # reference builder.py:776-786 / :876-886: mean and quartiles of every row's negatives
                row_form = K <= ops.QUANTILES_ROW_MAX
                stats["neg_mean"] = torch.empty(R, dtype=torch.float32, device=rows.device) if row_form else res.lneg.mean(1)
                stats["neg_quartiles"] = ops.masked_quantiles_multi([dict(x=res.lneg, stride_row=K, stride_elem=1, R=R, N=K,
                                                                          mean_out=stats["neg_mean"] if row_form else None)])[0]
        if need:
            ctx.save_for_backward(drows, dE.reshape(pos.shape))
        return loss

    @staticmethod
    def backward(ctx, g):
        drows, dpos = ctx.saved_tensors
        return drows * g, dpos * g, None, None, None, None, None


def queue_infonce(rows: torch.Tensor, pos: torch.Tensor, queue: torch.Tensor, temperature: float, stats=None) -> torch.Tensor:
    """rows: [R,C] (one vector per row) or [b,C,S2] (one vector per pixel, reference layout); pos: R positives.
    stats: None, or a dict that receives 'neg_mean' [R] and 'neg_quartiles' [3,R] of the raw queue logits (logging)."""
    rows = rows.float().contiguous()
    if rows.dim() == 2:
        R, C = rows.shape
        layout = (1, C, 0, 1)
    else:
        b, C, S2 = rows.shape
        R, layout = b * S2, (S2, C * S2, 1, S2)
    return _QueueInfoNCEFn.apply(rows, pos.float(), queue, float(temperature), layout, R, stats)


class _LocalPositivesFn(torch.autograd.Function):
    """pos, best = cp2_densecl_match(...) with d pos / d q_local = kvec (reference builder.py:818-855; the arg-max and the
    key side carry no gradient)."""

    @staticmethod
    def forward(ctx, q_local, q_embed, k_embed, k_local, ids_q, ids_k, lmbd, k_row, normalize_k, metrics):
        res = ops.densecl_match(q_embed, k_embed, q_local, k_local, ids_q, ids_k, lmbd, k_row=k_row, normalize_k=normalize_k,
                                want_kvec=q_local.requires_grad, want_metrics=metrics is not None)
        if metrics is not None:
            metrics["counts"] = res.counts
        if q_local.requires_grad:
            ctx.save_for_backward(res.kvec)
        ctx.mark_non_differentiable(res.best)
        return res.pos, res.best

    @staticmethod
    def backward(ctx, g_pos, _g_best):
        (kvec,) = ctx.saved_tensors
        return (kvec * g_pos.unsqueeze(1),) + (None,) * 9


def _matching_rate(counts: torch.Tensor) -> torch.Tensor:
    """builder.py:856-864: how often the best local match is the coordinate match, over the overlapping pixels (-1: none)."""
    tot = counts.sum(0)
    return torch.where(tot[0] > 0, tot[1].float() / tot[0].clamp(min=1).float(), torch.full((), -1.0, device=counts.device))


def densecl_local_positives(q_embed, k_embed, q_local, k_local, ids_q, ids_k, lmbd_coordinate: float = 0.0, metrics=None,
                            k_row=None, normalize_k: bool = False):
    """Positive score of every query pixel for the DenseCL local loss (reference builder.py:818-855):
    local similarity with the key pixel that maximises the BACKBONE similarity; where the two id maps overlap it is
    mixed with the summed local similarity over id-matching key pixels.  One cp2_densecl_match launch; no b x S2 x S2
    map exists.  q_embed / k_embed: (b, Cb, S2) or (b, Cb, h, w), channel-normalised (normalize_k=False) or the raw
    backbone features (normalize_k=True; bf16 channels-last maps are read in place); q_local / k_local: (b, C, S2) unit
    vectors; ids: (b, S2).  k_row: sample n's key side is row k_row[n].  Returns (pos (b, S2), arg-max index (b, S2));
    differentiable with respect to q_local."""
    q_local, k_local = q_local.float().contiguous(), k_local.detach().float().contiguous()
    b, _, S2 = q_local.shape
    ids_q = ids_q.reshape(b, S2).contiguous() if ids_q is not None else None
    ids_k = ids_k.reshape(b, S2).contiguous() if ids_k is not None else None
    pos, best = _LocalPositivesFn.apply(q_local, q_embed.detach(), k_embed.detach(), k_local, ids_q, ids_k,
                                        float(lmbd_coordinate), k_row, bool(normalize_k), metrics)
    if metrics is not None:
        metrics["matching_positives_rate"] = _matching_rate(metrics.pop("counts"))
    return pos, best.long()


class _CommTimer:
    """with-block that brackets an exchange step with timing events on the current stream (no-op when `sink` is None)."""

    def __init__(self, sink, name):
        self.sink, self.name = sink, name

    def __enter__(self):
        if self.sink is not None:
            self.t0 = torch.cuda.Event(enable_timing=True)
            self.t0.record()
        return self

    def __exit__(self, *exc):
        if self.sink is not None:
            t1 = torch.cuda.Event(enable_timing=True)
            t1.record()
            self.sink.setdefault(self.name, []).append((self.t0, t1))
        return False


def _rows_as_f32(x: torch.Tensor):
    """(rows, restore): a dense [B, ...] tensor (NCHW or channels-last, fp32 or bf16) as a [B, n] float32 matrix that
    shares its memory -- what the row gathers and the row exchange move -- and the function that gives a matrix of that
    shape back the tensor's shape, strides and dtype."""
    B = x.shape[0]
    cl = x.dim() == 4 and not x.is_contiguous() and x.is_contiguous(memory_format=torch.channels_last)
    flat = (x.permute(0, 2, 3, 1) if cl else x).reshape(B, -1)
    shape, dtype = x.shape, x.dtype
    if dtype != torch.float32:
        if (flat.shape[1] * flat.element_size()) % 4:
            raise ValueError("row size is not a multiple of 4 bytes")
        flat = flat.view(torch.float32)

    def restore(rows: torch.Tensor) -> torch.Tensor:
        r = rows if dtype == torch.float32 else rows.view(dtype)
        if cl:
            return r.reshape(rows.shape[0], shape[2], shape[3], shape[1]).permute(0, 3, 1, 2)
        return r.reshape((rows.shape[0],) + tuple(shape[1:]))
    return flat, restore


# (log name, index into the cp2_step_scalars vector); the reference's wandb scalar names (builder.py:1553-1604)
_CP2_LOG_NAMES = (("train/loss_step", CF.S_LOSS), ("train/loss_ins_step", CF.S_LOSS_INS), ("train/loss_dense_step", CF.S_LOSS_DENSE),
                  ("train/acc_ins_step", CF.S_ACC1), ("train/acc_seg_step", CF.S_ACC_DENSE), ("train/+ive_scores_step", CF.S_POS_SCORE),
                  ("train/-ive_scores_step", CF.S_NEG_SCORE), ("step/instance_average_positive_scores", CF.S_INS_POS),
                  ("train/cross_image_variance_source_step", CF.S_VAR_SRC), ("train/cross_image_variance_target_step", CF.S_VAR_TGT))
_CP2_LOG_NAMES_Q = _CP2_LOG_NAMES + (
    ("step/dense_per_sample_average_positive_scores", CF.S_POS_SCORE),
    ("step/dense_per_sample_lower_positive_scores", CF.S_DPOS_Q), ("step/dense_per_sample_median_positive_scores", CF.S_DPOS_Q + 1),
    ("step/dense_per_sample_upper_positive_scores", CF.S_DPOS_Q + 2),
    ("step/dense_per_sample_average_negative_scores", CF.S_NEG_SCORE),
    ("step/dense_per_sample_lower_negative_scores", CF.S_DNEG_Q), ("step/dense_per_sample_median_negative_scores", CF.S_DNEG_Q + 1),
    ("step/dense_per_sample_upper_negative_scores", CF.S_DNEG_Q + 2),
    ("step/instance_average_negative_scores", CF.S_INS_NEG_MEAN),
    ("step/instance_lower_negative_scores", CF.S_INS_Q), ("step/instance_median_negative_scores", CF.S_INS_Q + 1),
    ("step/instance_upper_negative_scores", CF.S_INS_Q + 2))


class MODEL(nn.Module):
    def __init__(self, cfg, rank, dim=128, K=65536, m=0.999, instance_logits_temp=0.2, pretrain_from_scratch=False,
                 include_background=False, lmbd_cp2_dense_loss=0.2, lmbd_pixel_corr_weight=1,
                 lmbd_region_corr_weight=1, lmbd_not_corr_weight=1, negative_type=NegativeType.NONE,
                 negative_scale=2, pretrain_type=PretrainType.CP2, backbone_type=BackboneType.DEEPLABV3,
                 mapping_type=MappingType.CP2, dense_logits_temp=1, unet_truncated_dec_blocks=2, use_predictor=False,
                 use_avgpool_global=False, use_symmetrical_loss=False, lmbd_coordinate=0, device=None,
                 # additive knobs (not in the reference)
                 amp_dtype: Optional[torch.dtype] = None, channels_last: bool = False, log_fn=None):
        super().__init__()
        self.queue_len, self.momentum, self.dim = K, m, dim
        self.temp_global, self.temp_local = instance_logits_temp, dense_logits_temp
        self.include_background, self.lmbd_dense_loss = include_background, lmbd_cp2_dense_loss
        self.device, self.rank, self.epoch = device, rank, 0
        self.use_predictor, self.use_avgpool_global = use_predictor, use_avgpool_global
        self.use_symmetrical_loss = use_symmetrical_loss
        assert 0 <= lmbd_coordinate <= 1, f"{lmbd_coordinate = }"
        self.lmbd_coordinate = lmbd_coordinate
        assert mapping_type in MappingType
        self.mapping_type = mapping_type
        if mapping_type == MappingType.CP2:          # same validation as reference builder.py:329-344
            assert lmbd_pixel_corr_weight == 1 and lmbd_region_corr_weight == 1 and lmbd_not_corr_weight == 1
        elif mapping_type == MappingType.PIXEL_ID:
            assert lmbd_region_corr_weight == 1 and lmbd_pixel_corr_weight > 1
        elif mapping_type == MappingType.REGION_ID:
            assert lmbd_pixel_corr_weight == 1 and lmbd_region_corr_weight > 1
        self.lmbd_pixel_corr_weight = lmbd_pixel_corr_weight
        self.lmbd_region_corr_weight = lmbd_region_corr_weight
        self.lmbd_not_corr_weight = lmbd_not_corr_weight
        assert pretrain_type in PretrainType and negative_type in NegativeType and backbone_type in BackboneType
        self.pretrain_type, self.negative_type, self.negative_scale = pretrain_type, negative_type, negative_scale
        self.backbone_type = backbone_type
        if backbone_type != BackboneType.DEEPLABV3:
            raise NotImplementedError(f"{backbone_type = }: the U-Net backbones need segmentation_models_pytorch "
                                      "(out of scope of the MI355X hot path; SURVEY.md section 2.1)")
        self.amp_dtype, self.channels_last, self.log_fn = amp_dtype, channels_last, log_fn

        self.encoder_q = build_segmentor(cfg.model, train_cfg=cfg.get("train_cfg"), test_cfg=cfg.get("test_cfg"))
        self.encoder_k = build_segmentor(cfg.model, train_cfg=cfg.get("train_cfg"), test_cfg=cfg.get("test_cfg"))
        print(f"[INFO] Initializing with imagenet weights: {not pretrain_from_scratch}")
        if not pretrain_from_scratch:
            self.encoder_q.backbone.init_weights()
            self.encoder_k.backbone.init_weights()

        # output strides from a CPU dry run, as the reference does (builder.py:393-402)
        probe = torch.rand(2, 3, 224, 224)
        self.output_stride = int(probe.shape[2] / self.encoder_q(probe).shape[2])
        print(f"{self.output_stride = }")
        self.backbone_output_stride = int(probe.shape[2] / self.encoder_q.backbone(torch.rand(2, 3, 224, 224))[3].shape[2])
        print(f"{self.backbone_output_stride = }")

        if pretrain_type in (PretrainType.BYOL, PretrainType.MOCO):
            raise NotImplementedError(f"{pretrain_type = }: image-level baselines are outside the CP2 hot path")
        elif pretrain_type == PretrainType.CP2:
            assert self.negative_type == NegativeType.NONE and self.mapping_type == MappingType.CP2
        elif pretrain_type in (PretrainType.DENSECL, PretrainType.PROPOSED_V2):
            feat = self.encoder_q.backbone.feat_dim
            self.encoder_q.neck = DenseCLNeck(in_channels=feat, hid_channels=2048, out_channels=self.dim)
            self.encoder_k.neck = DenseCLNeck(in_channels=feat, hid_channels=2048, out_channels=self.dim)
            assert self.momentum == 0.999 and self.lmbd_dense_loss == 0.5, (self.momentum, self.lmbd_dense_loss)
            assert self.temp_global == 0.2 and self.temp_local == 0.2, (self.temp_global, self.temp_local)
            if pretrain_type == PretrainType.DENSECL:
                assert not (use_predictor or use_avgpool_global or use_symmetrical_loss) and lmbd_coordinate == 0
            # forward_densecl never runs the segmentation head, nor the neck heads its flags do not select.  The
            # reference lets DDP search for them every step (find_unused_parameters=True, main.py:456-460); here they
            # are frozen up front -- the same training dynamics (a parameter without gradient is not touched by SGD,
            # weight decay included) with no per-step graph walk, and DDP never waits for their buckets.
            neck = self.encoder_q.neck
            unused = [self.encoder_q.decode_head]
            if not use_predictor:
                unused += [neck.global_predictor, neck.local_predictor]
            if use_avgpool_global:
                unused += [neck.global_projector, neck.global_predictor]
            for mod in unused:
                mod.requires_grad_(False)
        for pq, pk in zip(self.encoder_q.parameters(), self.encoder_k.parameters()):
            pk.data.copy_(pq.data)
            pk.requires_grad = False

        self.register_buffer("queue", F.normalize(torch.randn(self.dim, self.queue_len), dim=0))
        self.register_buffer("queue_ptr", torch.zeros(1, dtype=torch.long))
        self.register_buffer("queue2", F.normalize(torch.randn(self.dim, self.queue_len), dim=0))
        self.register_buffer("queue2_ptr", torch.zeros(1, dtype=torch.long))

        self.loss_o = AverageMeter("Loss_overall", ":.4f")
        self.loss_i = AverageMeter("Loss_ins", ":.4f")
        self.loss_d = AverageMeter("Loss_den", ":.4f")
        self.acc_ins = AverageMeter("Acc_ins", ":6.2f")
        self.acc_seg = AverageMeter("Acc_seg", ":6.2f")
        self.cross_image_variance_source = AverageMeter("Cross_Image_Variance_Source", ":6.2f")
        self.cross_image_variance_target = AverageMeter("Cross_Image_Variance_Target", ":6.2f")
        self.correlation_ious, self.masked_correlation_ious = [], []
        self._flat_q = self._flat_k = None
        self.log_quartiles = True        # per-step quartile statistics of the reference (builder.py:1298,1399-1406), sort-free
        self.key_weight_shadow = amp_dtype == torch.bfloat16   # EMA also emits bf16 key weights for the key encoder's convs
        self.neck_autocast = True        # DenseCL neck in the encoders' autocast precision (False: fp32), see _neck
        self._flat_k_bf16 = None
        self._flat_q_bf16 = None         # bf16 image of the query weights, written by optim.FlatSGD (enable_query_shadow)
        self._q_shadow_version = None
        self.ema_in_forward = True       # False: the caller runs _momentum_update_key_encoder() itself before forward
        self.overlap_key_branch = None   # None / False: one stream (default); "gather": the exchange steps on a side stream, see forward_cp2
        self._side_stream = None
        self.key_forward_graph = True    # replay the (gradient-free) key encoder forward from a hipGraph after warm-up
        self._key_graphs = {}
        self._pending_logs = []          # device scalars waiting for one batched device->host copy
        self.shuffle_exchange = "all_to_all"   # shuffle-BN rows by all-to-all; "all_gather" = the reference's form
        self.comm_events = None          # bench.py: {} -> every exchange step records a (start, stop) event pair
        self.sync_logs_every = 0         # 0: only when flush_logs() / on_train_epoch_end() is called

    # ------------------------------------------------------------------ parameters as flat buffers
    def flatten_parameters(self):
        """Re-home every encoder parameter into one contiguous fp32 buffer per encoder (same offsets
        in both), so the EMA is a single streaming kernel over 12 bytes per parameter.  Idempotent;
        called lazily by the first forward (after .to(device) / DDP wrapping)."""
        first = next(self.encoder_q.parameters())
        if self._flat_q is not None and self._flat_q.device == first.device and first.data_ptr() == self._flat_q.data_ptr():
            return                                           # already flat (O(1) check, runs every step)
        pq, pk = list(self.encoder_q.parameters()), list(self.encoder_k.parameters())
        dev = pq[0].device
        offs, total = [], 0
        for p in pq:
            offs.append(total)
            total += (p.numel() + 63) // 64 * 64            # 256-byte aligned slots
        flat_q = torch.zeros(total, dtype=torch.float32, device=dev)
        flat_k = torch.zeros(total, dtype=torch.float32, device=dev)
        for p, k, o in zip(pq, pk, offs):
            assert p.dtype == torch.float32 and k.shape == p.shape
            if k.stride() != p.stride():
                k.data = k.data.clone(memory_format=torch.preserve_format).as_strided(p.shape, p.stride()).copy_(k.data)
            # keep each parameter's own dense layout (e.g. channels-last conv weights) inside its slot: the slot
            # holds the elements in PHYSICAL order, the same order in both encoders, so the EMA stays elementwise
            vq = torch.as_strided(flat_q, p.shape, p.stride(), o)
            vk = torch.as_strided(flat_k, p.shape, p.stride(), o)
            vq.copy_(p.data)
            vk.copy_(k.data)
            p.data, k.data = vq, vk
        # bf16 shadow of the key weights (same element order), refreshed by every EMA launch: the key encoder's
        # convolutions read it directly, so autocast launches no cast kernel per weight tensor
        self._flat_k_bf16 = flat_k.to(torch.bfloat16) if self.key_weight_shadow else None
        if self._flat_k_bf16 is not None:
            from .encoder import Conv2d
            by_param = {id(k): o for k, o in zip(pk, offs)}
            for mod in self.encoder_k.modules():
                if isinstance(mod, Conv2d) and id(mod.weight) in by_param:
                    mod.shadow_weight = torch.as_strided(self._flat_k_bf16, mod.weight.shape, mod.weight.stride(),
                                                         by_param[id(mod.weight)])
        self._flat_q, self._flat_k, self._flat_offsets = flat_q, flat_k, offs
        if self._flat_q_bf16 is not None:                   # re-homed: the old image belongs to the old buffer
            self._flat_q_bf16 = None
            self.enable_query_shadow()

    # ------------------------------------------------------------------ bf16 image of the query weights
    def enable_query_shadow(self):
        """Give the query encoder's convolutions a bf16 image of their weights (one flat buffer, same offsets as the
        fp32 parameters) so bf16 autocast launches no cast kernel per weight tensor.  optim.FlatSGD writes the image
        in its update kernel; any other in-place change of a parameter is noticed through the tensors' version
        counters at the next forward and the image is rebuilt (_refresh_query_shadow)."""
        if self.amp_dtype != torch.bfloat16 or self._flat_q is None or self._flat_q_bf16 is not None:
            return
        from .encoder import Conv2d
        self._flat_q_bf16 = torch.empty(self._flat_q.numel(), dtype=torch.bfloat16, device=self._flat_q.device)
        pq = list(self.encoder_q.parameters())
        by_param = {id(p): o for p, o in zip(pq, self._flat_offsets)}
        for mod in self.encoder_q.modules():
            if isinstance(mod, Conv2d) and id(mod.weight) in by_param:
                mod.shadow_weight = torch.as_strided(self._flat_q_bf16, mod.weight.shape, mod.weight.stride(),
                                                     by_param[id(mod.weight)])
        self._q_shadow_params = pq
        self._q_shadow_version = None
        self._refresh_query_shadow()

    def _query_version(self):
        return sum(p._version for p in self._q_shadow_params)

    def _query_shadow_written(self):
        """Called by the optimizer kernel's host side: image and parameters were written together."""
        if self._flat_q_bf16 is not None:
            self._q_shadow_version = self._query_version()

    def _refresh_query_shadow(self):
        if self._flat_q_bf16 is None:
            return
        v = self._query_version()
        if v != self._q_shadow_version:
            ops.bf16_image(self._flat_q, self._flat_q_bf16)
            self._q_shadow_version = v

    @torch.no_grad()
    def _momentum_update_key_encoder(self):
        """theta_k = theta_k*m + theta_q*(1-m) for every encoder parameter (reference builder.py:557-567)."""
        self.flatten_parameters()
        if self._flat_k_bf16 is not None:
            ops.ema_flat_shadow(self._flat_k, self._flat_q, self._flat_k_bf16, self.momentum)
        else:
            ops.ema_flat(self._flat_k, self._flat_q, self.momentum)

    def _encode_key(self, img):
        """Key encoder forward (reference builder.py:1276).  No autograd, fixed shapes, static weights addresses
        (the flat buffers): replayed from a hipGraph (engine.ForwardGraph) so its ~280 launches cost no host time."""
        return self._key_forward("cp2", lambda x: self._encode(self.encoder_k, x), img)

    def _key_forward(self, name, fn, img):
        """fn(img) -- a gradient-free forward through (parts of) the key encoder returning a tensor or a tuple of tensors --
        replayed from a hipGraph after three eager calls.  One graph per `name` (the DenseCL step has two key passes whose
        outputs must both stay alive: "densecl0", "densecl1")."""
        if not self.key_forward_graph or not img.is_cuda:
            return fn(img)
        graphs = self._key_graphs
        if name not in graphs:
            from .encoder import FusedBatchNorm2d
            from .engine import ForwardGraph
            bns = [m for m in self.encoder_k.modules() if isinstance(m, FusedBatchNorm2d)]

            def fwd(x):
                # a captured call does not execute: take back the lazy num_batches_tracked ticks it made
                capturing = torch.cuda.is_current_stream_capturing()
                before = [m._pending_batches for m in bns] if capturing else None
                out = fn(x)
                if capturing:
                    fwd.fused = [m for m, b in zip(bns, before) if m._pending_batches != b]
                    for m, b in zip(bns, before):
                        m._pending_batches = b
                return out

            fwd.fused = []

            def tick():
                for m in fwd.fused:
                    m._pending_batches += 1

            graphs[name] = ForwardGraph(fwd, warmup=3, on_replay=tick)
        # the graph reads the key weights where flatten_parameters() put them: a new home invalidates it
        return graphs[name](img, tag=(self._flat_k.data_ptr() if self._flat_k is not None else None,
                                      self.encoder_k.training, torch.is_grad_enabled()))

    @property
    def _key_graph(self):
        """The CP2 key-forward graph (None before its first use): what the tests and tools inspect."""
        return self._key_graphs.get("cp2")

    def _key_stream(self):
        if self._side_stream is None:
            self._side_stream = torch.cuda.Stream()
        return self._side_stream

    # ------------------------------------------------------------------ queue
    @torch.no_grad()
    def _enqueue(self, queue, ptr, keys):
        """All-gather the keys over the ranks (C4) and write them into the queue (reference builder.py:569-607).  With
        overlap_key_branch set, both run on the side HIP stream: nothing on the main stream needs the updated queue before
        the NEXT step's loss section, which waits for `_enqueue_done` (so does anything that reads the queue from
        outside: wait_enqueue())."""
        keys = keys.contiguous()
        cur = torch.cuda.current_stream()
        if not cdist.multi() or torch.cuda.is_current_stream_capturing():
            ops.enqueue(queue, keys, ptr)
            return
        if not self.overlap_key_branch:                   # default: in order on the step's stream (see forward_cp2)
            with self._comm("c4_key_gather_enqueue"):
                ops.enqueue(queue, concat_all_gather(keys), ptr)
            return
        side = self._key_stream()
        side.wait_stream(cur)
        with torch.cuda.stream(side):
            keys.record_stream(side)
            with self._comm("c4_key_gather_enqueue"):
                ops.enqueue(queue, concat_all_gather(keys), ptr)
            self._enqueue_done = torch.cuda.Event()
            self._enqueue_done.record(side)

    def state_dict(self, *args, **kwargs):
        self.wait_enqueue()
        return super().state_dict(*args, **kwargs)

    def wait_enqueue(self):
        """Make the current stream wait for an enqueue still running on the side stream."""
        ev = getattr(self, "_enqueue_done", None)
        if ev is not None:
            torch.cuda.current_stream().wait_event(ev)
            self._enqueue_done = None

    def _gather_keys(self, keys):
        """All ranks' keys in rank order (C4, reference builder.py:572), timed as the step's key gather."""
        with self._comm("c4_key_gather_enqueue"):
            return concat_all_gather(keys.contiguous())

    def _keep_ious(self, iou, iou_masked):
        self.correlation_ious.append(iou)
        self.masked_correlation_ious.append(iou_masked)
        if len(self.correlation_ious) >= 1024:              # keep the python lists short (device-side concatenation, no sync)
            self.correlation_ious[:] = [torch.cat(self.correlation_ious)]
            self.masked_correlation_ious[:] = [torch.cat(self.masked_correlation_ious)]

    def _dequeue_and_enqueue(self, keys):
        self._enqueue(self.queue, self.queue_ptr, keys)

    def _dequeue_and_enqueue2(self, keys):
        self._enqueue(self.queue2, self.queue2_ptr, keys)

    # ------------------------------------------------------------------ shuffle-BN
    def _comm(self, name):
        """Event pair around one exchange step on the current stream when bench.py asks for it (comm_events = {})."""
        return _CommTimer(self.comm_events, name)

    @torch.no_grad()
    def _batch_shuffle_ddp(self, x, idx_shuffle=None):
        """Rows of the permuted global batch this rank's key encoder takes (reference builder.py:609-630) and what
        _batch_unshuffle_ddp needs to undo it.  shuffle_exchange = "all_to_all" (default): only the rows this rank keeps
        travel (dist.ShufflePlan), the permutation is drawn on every host from a generator seeded once by rank 0;
        "all_gather": the reference's form -- gather everything, broadcast the permutation, keep a slice."""
        if not (x.is_contiguous() or (x.dim() == 4 and x.is_contiguous(memory_format=torch.channels_last))):
            x = x.contiguous()
        if x.is_cuda:
            x.record_stream(torch.cuda.current_stream())      # may be read here on the side stream after its owner drops it
        w = cdist.world_size()
        if cdist.multi() and self.shuffle_exchange == "all_to_all":
            n_all = x.shape[0] * w
            host = idx_shuffle.cpu() if idx_shuffle is not None else cdist.shared_permutation(n_all, x.device)
            plan = cdist.ShufflePlan(host, cdist.rank(), w)
            with self._comm("c1_image_exchange"):
                rows, restore = _rows_as_f32(x)
                taken = restore(cdist.exchange_rows(rows, plan, take=ops.gather_rows))
            return taken, plan
        with self._comm("c1_image_exchange"):
            rows, restore = _rows_as_f32(x)                  # any dense layout / fp32 or bf16: rows of 4-byte words
            x_gather = concat_all_gather(rows)
            if idx_shuffle is None:
                idx_shuffle = cdist.make_shuffle_index(x_gather.shape[0], x.device)
            idx_unshuffle = torch.argsort(idx_shuffle)
            taken = restore(ops.gather_rows(x_gather, cdist.shuffle_rows_for_rank(idx_shuffle, cdist.rank(), w).contiguous()))
        return taken, idx_unshuffle

    @torch.no_grad()
    def _batch_unshuffle_ddp(self, x, idx_unshuffle):
        """Undo _batch_shuffle_ddp on the key encoder's outputs (reference builder.py:632-649)."""
        x = x.float().contiguous()
        with self._comm("c3_key_unshuffle"):
            if isinstance(idx_unshuffle, cdist.ShufflePlan):
                return cdist.exchange_rows(x, idx_unshuffle, backward=True, take=ops.gather_rows)
            w = cdist.world_size()
            return ops.gather_rows(concat_all_gather(x), idx_unshuffle.view(w, -1)[cdist.rank()].contiguous())

    @torch.no_grad()
    def _key_rows(self, feats, ctx):
        """Key-encoder outputs of the shuffled batch -> (tensors, k_row) with sample n's features at row k_row[n] of each
        tensor (k_row None: they are in sample order).  One rank: nothing moves, k_row is the un-shuffle index itself;
        all-to-all exchange: the rows travel back and stay in arrival order (keep_order); the reference's all-gather form
        (builder.py:632-649) returns them ordered."""
        if isinstance(ctx, cdist.ShufflePlan):
            outs = []
            with self._comm("c3_key_unshuffle"):
                for f in feats:
                    if not (f.is_contiguous() or (f.dim() == 4 and f.is_contiguous(memory_format=torch.channels_last))):
                        f = f.contiguous()
                    rows, restore = _rows_as_f32(f)
                    outs.append(restore(cdist.exchange_rows(rows, ctx, backward=True, take=ops.gather_rows, keep_order=True)))
            return outs, ctx.device_tables(outs[0].device)[3]
        if cdist.multi():
            return [self._batch_unshuffle_ddp(f, ctx) for f in feats], None
        return list(feats), ctx

    # ------------------------------------------------------------------ dispatch
    def forward(self, **kwargs):
        self.wait_enqueue()                     # the previous step's queue update (when it ran on the side stream)
        if self._flat_q_bf16 is not None:       # bf16 weight image in use: rebuild it if a parameter changed elsewhere
            self.flatten_parameters()
            self._refresh_query_shadow()
        if self.pretrain_type in (PretrainType.CP2, PretrainType.PROPOSED):
            return self.forward_cp2(**kwargs)
        if self.pretrain_type in (PretrainType.DENSECL, PretrainType.PROPOSED_V2):
            return self.forward_densecl(**kwargs)
        raise NotImplementedError(f"{self.pretrain_type = }")

    def _encode(self, enc, img):
        if self.channels_last:
            img = img.contiguous(memory_format=torch.channels_last)
        if self.amp_dtype is not None:
            with torch.autocast("cuda", dtype=self.amp_dtype):
                return enc(img)
        return enc(img)

    def _neck(self, enc, feat):
        """DenseCL neck (reference builder.py:179-274) on the last backbone map.  Under autocast it runs in the encoders'
        precision like the decode head of the CP2 path (round 4; it ran in fp32 before: two 2048 -> 2048 1x1 convolutions
        of 6272 pixels = 1.6 ms of fp32 MIOpen kernels per step at config 5); `neck_autocast = False` keeps it in fp32.  The
        normalisation of its outputs and everything after is fp32 either way."""
        if self.amp_dtype is not None and self.neck_autocast:
            with torch.autocast("cuda", dtype=self.amp_dtype):
                return enc.neck(feat)
        return enc.neck(feat.float())

    # ------------------------------------------------------------------ CP2
    def forward_cp2(self, img_a, img_b, bg0, bg1, visualize, step, new_epoch, pixel_ids_a, pixel_ids_b,
                    region_ids_a, region_ids_b, idx_shuffle=None):
        s = self.output_stride
        b, _, H, W = img_a.shape
        # ---- composition of both views in ONE launch, already in the layout / precision the stem convolution reads, the
        # key view's rows already in the order the shuffle-BN exchange sends them (reference builder.py:1146-1159, :609-630)
        multi = cdist.multi()
        plan = idx_unshuffle = None
        a2a = multi and self.shuffle_exchange == "all_to_all"
        if a2a:
            host = idx_shuffle.cpu() if idx_shuffle is not None else cdist.shared_permutation(b * cdist.world_size(), img_a.device)
            plan = cdist.ShufflePlan(host, cdist.rank(), cdist.world_size())
            row_b = plan.device_tables(img_a.device)[0]                    # send order
        elif not multi:
            if idx_shuffle is None:
                # one rank: permutation and its inverse on the host (the global torch RNG, as builder.py:618), one copy --
                # a device argsort is a sort kernel plus four helper launches per step
                host = torch.randperm(b)
                both = torch.stack([host, torch.argsort(host)])
                both = both.pin_memory().to(img_a.device, non_blocking=True) if img_a.is_cuda else both
                idx_shuffle, idx_unshuffle = both[0], both[1]
            else:
                idx_unshuffle = torch.argsort(idx_shuffle)
            row_b = idx_shuffle.contiguous()
        else:
            row_b = None                                                   # the reference's all-gather form shuffles after the gather
        fused = W % 4 == 0 and all(t.is_contiguous() and t.data_ptr() % 16 == 0 for t in (img_a, img_b, bg0, bg1))
        if fused:
            out_dtype = torch.bfloat16 if self.amp_dtype == torch.bfloat16 else torch.float32
            img_a, img_b, mask_a, mask_b = ops.compose_pair(img_a, bg0, img_b, bg1, s, row_b, self.channels_last, out_dtype)
        else:
            img_a, _, mask_a = ops.compose_mask(img_a.contiguous(), bg0.contiguous(), s)
            img_b, _, mask_b = ops.compose_mask(img_b.contiguous(), bg1.contiguous(), s)
            if row_b is not None and not a2a:                # one rank: the shuffle is this gather; all-to-all: exchange_rows gathers
                img_b = ops.gather_rows(img_b, row_b)
        mask_a, mask_b = mask_a.reshape(b, -1), mask_b.reshape(b, -1)
        ids = None
        weights = (float(self.lmbd_pixel_corr_weight), float(self.lmbd_region_corr_weight), float(self.lmbd_not_corr_weight))
        if weights != (1.0, 1.0, 1.0):
            ids = tuple(ops.strided_gather(t.contiguous(), s).reshape(b, -1) for t in (pixel_ids_a, pixel_ids_b, region_ids_a, region_ids_b))
        # IoUs of the down-sampled region-id maps, read straight from the full-resolution maps (logged per epoch, on device):
        # part of the step's tail launch when their hash table fits LDS (cp2_step_tail), else a launch of their own
        iou_in_tail = ops.tail_iou_supported(H, W, s)
        if not iou_in_tail:
            self._keep_ious(*ops.corr_iou_strided(region_ids_a.contiguous(), region_ids_b.contiguous(), s, mask_a, mask_b))

        # The key branch (EMA -> shuffle-BN exchange -> key encoder -> un-shuffle) does not depend on the query encoder
        # (reference order builder.py:1260-1277 is serial).  overlap_key_branch:
        #   False / None (default): everything in order on ONE stream.  Measured with the step's collectives running over
        #            RCCL with one rank (bench.py --rehearse-collectives): 13.73 ms per step in order against 15.1-15.6 ms with
        #            "gather" -- a fork / join between HIP streams costs this stack far more than the ~0.14 ms of EMA +
        #            exchange it can hide when no byte moves; with real peers bench.py --overlap auto
        #            times both forms and takes the faster;
        #   "gather": EMA + the image exchange (and the key all-gather + enqueue) on a side HIP stream, overlapping the query
        #            forward (north_star's form); the key encoder itself follows on the main stream.
        # (The whole key branch on the side stream -- two compute-heavy branches interleaved on one GPU -- measured 4 % slower
        # in round 1 and 1.3 ms slower in round 3; that form was removed in round 4.)
        self.flatten_parameters()        # on the main stream, before the fork: it re-homes the query parameters too
        self._refresh_query_shadow()
        cur = torch.cuda.current_stream()
        mode = self.overlap_key_branch or False
        if mode not in (False, "gather"):
            raise ValueError(f"overlap_key_branch = {self.overlap_key_branch!r}: None / False (one stream) or 'gather'")
        side = self._key_stream() if mode else cur
        if side is not cur:
            side.wait_stream(cur)

        def exchange_in(x):
            """This rank's composed key images -> (the rows its key encoder takes, what exchange_out needs to undo it)."""
            if a2a:
                with self._comm("c1_image_exchange"):
                    rows, restore = _rows_as_f32(x)
                    return restore(cdist.exchange_rows(rows, plan, take=ops.gather_rows, presorted=fused)), None
            if multi:
                return self._batch_shuffle_ddp(x, idx_shuffle)
            return x, None                                   # one rank: compose_pair wrote the rows in shuffled order

        def exchange_out(k_enc, ctx):
            """Key encoder outputs -> (tensor, row index) such that sample n's features are tensor[row[n]]."""
            if a2a:
                with self._comm("c3_key_unshuffle"):
                    k32 = k_enc.float().contiguous()
                    return cdist.exchange_rows(k32, plan, backward=True, take=ops.gather_rows, keep_order=True), \
                        plan.device_tables(k32.device)[3]
            if multi:
                return self._batch_unshuffle_ddp(k_enc, ctx), None
            return k_enc.float(), idx_unshuffle

        k = k_row = None
        with torch.cuda.stream(side), torch.no_grad():
            if self.ema_in_forward:
                self._momentum_update_key_encoder()
            if side is not cur and img_b.is_cuda:
                img_b.record_stream(side)
            img_k, ctx = exchange_in(img_b)
            if not mode:
                k, k_row = exchange_out(self._encode_key(img_k), ctx)
        q = self._encode(self.encoder_q, img_a).float()                          # queries: b x C x h x w
        if side is not cur:
            with self._comm("key_branch_wait_exposed"):       # main stream idle until EMA + image exchange are done
                cur.wait_stream(side)
        if k is None:
            with torch.no_grad():
                if side is not cur:
                    img_k.record_stream(cur)          # allocated on the side stream, read on this one
                k, k_row = exchange_out(self._encode_key(img_k), ctx)

        # the step's tail -- returned / logged scalars, the keys' enqueue (after the all-gather over the ranks, C4) and the
        # logged IoUs -- is one launch; with the exchange steps on a side stream the enqueue stays with them (_enqueue)
        tail = {}
        if not mode and not torch.cuda.is_current_stream_capturing():
            tail["enqueue"] = (self.queue_ptr, self._gather_keys if multi else None)
        if iou_in_tail:
            tail["iou"] = (region_ids_a.contiguous(), region_ids_b.contiguous(), s)
        out = CF.cp2_loss_section(q, k, mask_a, mask_b, self.queue, temp_global=self.temp_global,
                                  temp_local=self.temp_local, lmbd_dense=self.lmbd_dense_loss,
                                  include_background=self.include_background, ids=ids, weights=weights,
                                  want_quartiles=self.log_quartiles, negative_type=self.negative_type.value,
                                  negative_scale=self.negative_scale, k_row=k_row, tail=tail)
        if "enqueue" not in tail:
            self._dequeue_and_enqueue(out.k_pos)
        if iou_in_tail:
            self._keep_ious(out.iou, out.iou_masked)
        # the logged scalars come out of the loss section as one device vector (cp2_step_scalars); names as the
        # reference's wandb.log (builder.py:1553-1604)
        names = _CP2_LOG_NAMES_Q if self.log_quartiles else _CP2_LOG_NAMES
        self._log_vector(step, b, names, out.scalars)
        if new_epoch:
            self.epoch += 1
        return out.loss

    # ------------------------------------------------------------------ DenseCL (BASELINE config 5)
    def forward_densecl(self, img_a, img_b, bg0, bg1, visualize, step, new_epoch, pixel_ids_a, pixel_ids_b,
                        region_ids_a, region_ids_b, idx_shuffle=None):
        bs = self.backbone_output_stride
        b = img_a.shape[0]
        pix_a = ops.strided_gather(pixel_ids_a.contiguous(), bs).reshape(b, -1)
        pix_b = ops.strided_gather(pixel_ids_b.contiguous(), bs).reshape(b, -1)

        def query_features(img):
            feat = self._encode(self.encoder_q.backbone, img)[3]         # bf16 channels-last under autocast, else fp32
            out = self._neck(self.encoder_q, feat)
            local = out["x_local_pred"] if self.use_predictor else out["x_local_proj"]
            glob = out["x_global_pred"] if self.use_predictor else out["x_global_proj"]
            if self.use_avgpool_global:
                glob = out["x_avgpool_local_pred"] if self.use_predictor else out["x_avgpool_local_proj"]
            # the backbone map goes to cp2_densecl_match as it is: only its arg-max is used (no gradient, reference
            # builder.py:818-821), and a positive factor per query pixel cannot change an arg-max -- no normalised copy
            return feat.detach(), F.normalize(local.flatten(2).float(), dim=1), F.normalize(glob.float(), dim=1)

        key_pass = [0]

        @torch.no_grad()
        def key_features(img):
            """-> (backbone map, local, global, pooled local, k_row): the two maps stay in the key encoder's (shuffled) row
            order and sample n is row k_row[n] (None: already in sample order); the two vectors are in sample order."""
            if self.ema_in_forward:
                self._momentum_update_key_encoder()
            img, ctx = self._batch_shuffle_ddp(img, idx_shuffle)

            def run(x):
                feat = self._encode(self.encoder_k.backbone, x)[3]
                out = self._neck(self.encoder_k, feat)
                glob = out["x_avgpool_local_proj"] if self.use_avgpool_global else out["x_global_proj"]
                return (feat, F.normalize(out["x_local_proj"].flatten(2).float(), dim=1), F.normalize(glob.float(), dim=1),
                        F.normalize(out["x_avgpool_local_proj"].float(), dim=1))
            # backbone + neck of the key side as one hipGraph replay per pass (the eager form costs ~2.5 ms of host time per
            # step: cfg5 measured host-bound at 17.1 ms); the two passes of the symmetric loss keep separate output buffers
            feat, local, glob, pooled = self._key_forward(f"densecl{key_pass[0]}", run, img)
            key_pass[0] += 1
            (feat, local, glob, pooled), k_row = self._key_rows((feat, local, glob, pooled), ctx)
            if k_row is not None:
                glob, pooled = glob.index_select(0, k_row), pooled.index_select(0, k_row)
            return feat, local, glob, pooled, k_row

        # rank 0 logs the score statistics of the FIRST pass (reference builder.py:774-804, 875-904: log_metrics=True there only)
        extra = {} if (self.rank == 0 and self.log_quartiles) else None
        # with peers, the per-pixel score statistics (6272 x 65536 logits) are shared out over the ranks instead of making
        # rank 0 the straggler every step: row_score_stats_over_ranks (a collective: every rank takes part)
        share_stats = self.log_quartiles and cdist.multi() and cdist.world_size() > 1

        def global_loss(qg, kg, log=False):
            st = {} if (log and extra is not None) else None
            pos = (qg * kg).sum(1)
            loss = queue_infonce(qg, pos, self.queue, self.temp_global, stats=st)
            if st is not None:
                nq = st["neg_quartiles"].mean(1)
                extra.update({"step/instance_average_positive_scores": pos.detach().mean(),
                              "step/instance_average_negative_scores": st["neg_mean"].mean(),
                              "step/instance_lower_negative_scores": nq[0], "step/instance_median_negative_scores": nq[1],
                              "step/instance_upper_negative_scores": nq[2],
                              "step/cross_image_variance_source_step": qg.detach().std(0).mean(),
                              "step/cross_image_variance_target_step": kg.std(0).mean()})
            return loss

        def local_loss(q_embed, k_embed, q_local, k_local, ids_q, ids_k, k_row, log=False):
            st = {} if (log and extra is not None) else None
            pos, _ = densecl_local_positives(q_embed, k_embed, q_local, k_local, ids_q, ids_k, self.lmbd_coordinate, metrics=st,
                                             k_row=k_row, normalize_k=True)
            loss = queue_infonce(q_local, pos.reshape(-1), self.queue2, self.temp_local,
                                 stats=st if not share_stats else None)
            shared = row_score_stats_over_ranks(q_local, self.queue2) if (log and share_stats) else None
            if st is not None:
                neg_mean, nq = (shared[0], shared[1:]) if shared is not None else (st["neg_mean"].mean(), st["neg_quartiles"].mean(1))
                iou, _ = ops.corr_iou(ids_q, ids_k)
                extra.update({"step/dense_average_positive_scores": pos.detach().mean(),
                              "step/dense_average_negative_scores": neg_mean,
                              "step/dense_lower_negative_scores": nq[0], "step/dense_median_negative_scores": nq[1],
                              "step/dense_upper_negative_scores": nq[2], "step/average_iou": iou.mean(),
                              "step/non_zero_iou_ratio": (iou != 0).float().mean(),
                              "step/matching_positives_rate": st["matching_positives_rate"]})
            return loss

        eq, lq, gq = query_features(img_a)
        ek, lk, gk, pooled_k, k_row = key_features(img_b)
        loss_global, loss_local = global_loss(gq, gk, log=True), local_loss(eq, ek, lq, lk, pix_a, pix_b, k_row, log=True)
        update = (gk, pooled_k)
        if self.use_symmetrical_loss:
            eq2, lq2, gq2 = query_features(img_b)
            ek2, lk2, gk2, pooled_k2, k_row2 = key_features(img_a)
            loss_global = loss_global + global_loss(gq2, gk2)
            loss_local = loss_local + local_loss(eq2, ek2, lq2, lk2, pix_b, pix_a, k_row2)
            if step % 2 == 0:
                update = (gk2, pooled_k2)
        loss = (1 - self.lmbd_dense_loss) * loss_global + self.lmbd_dense_loss * loss_local
        self._dequeue_and_enqueue(update[0])
        self._dequeue_and_enqueue2(update[1])
        logs = {"train/loss_step": loss.detach(), "train/loss_ins_step": loss_global.detach(),
                "train/loss_dense_step": loss_local.detach()}
        if extra:
            logs.update(extra)
        self._log_step(step, b, logs)
        return loss

    # ------------------------------------------------------------------ logging without per-step host syncs
    def _log_step(self, step, n, scalars):
        """The reference calls .item() several times per step (builder.py:1553-1604).  Here the scalars
        stay on the device; flush_logs() moves everything queued so far in one copy."""
        names = list(scalars)
        self._pending_logs.append((step, n, names, torch.stack([scalars[k].detach().float().reshape(()) for k in names])))
        # bounded even if the caller never flushes: one batched copy every `sync_logs_every` (default: 4096) records
        limit = self.sync_logs_every or 4096
        if len(self._pending_logs) >= limit and not torch.cuda.is_current_stream_capturing():
            self.flush_logs()

    def _log_vector(self, step, n, names, vec):
        """A record whose values already sit in one device vector: names = ((log name, index into vec), ...)."""
        self._pending_logs.append((step, n, names, vec))
        limit = self.sync_logs_every or 4096
        if len(self._pending_logs) >= limit and not torch.cuda.is_current_stream_capturing():
            self.flush_logs()

    def flush_logs(self):
        if not self._pending_logs:
            return []
        # one device -> host copy for everything queued: records of equal length are stacked first
        by_len = {}
        for i, p in enumerate(self._pending_logs):
            by_len.setdefault(p[3].numel(), []).append(i)
        rows = [None] * len(self._pending_logs)
        for idxs in by_len.values():
            block = torch.stack([self._pending_logs[i][3].detach().float() for i in idxs]).cpu().tolist()
            for i, r in zip(idxs, block):
                rows[i] = r
        meters = {"train/loss_step": self.loss_o, "train/loss_ins_step": self.loss_i, "train/loss_dense_step": self.loss_d,
                  "train/acc_ins_step": self.acc_ins, "train/acc_seg_step": self.acc_seg,
                  "train/cross_image_variance_source_step": self.cross_image_variance_source,
                  "train/cross_image_variance_target_step": self.cross_image_variance_target}
        out = []
        for (step, n, names, _), row in zip(self._pending_logs, rows):
            if names and isinstance(names[0], tuple):          # (name, index) pairs into a scalar vector
                rec = {k: row[i] for k, i in names}
            else:
                rec = dict(zip(names, row))
            for k, meter in meters.items():
                if k in rec:
                    meter.update(rec[k], n)
            if self.rank == 0 and self.log_fn is not None:
                self.log_fn(dict(rec, **{"update-step": step}))
            out.append((step, rec))
        self._pending_logs = []
        return out

    def epoch_ious(self):
        """(iou, iou_masked) lists accumulated since the last call (reference keeps python lists, :1254-1257)."""
        cat = lambda xs: torch.cat(xs).cpu().tolist() if xs else []  # noqa: E731
        r = cat(self.correlation_ious), cat(self.masked_correlation_ious)
        self.correlation_ious, self.masked_correlation_ious = [], []
        return r

    def on_train_epoch_end(self, step):
        self.wait_enqueue()
        self.flush_logs()
        if self.rank == 0 and self.log_fn is not None:
            rec = {"train/loss": self.loss_o.avg, "train/loss_ins": self.loss_i.avg, "train/loss_dense": self.loss_d.avg}
            if self.pretrain_type in (PretrainType.CP2, PretrainType.PROPOSED):
                rec.update({"train/acc_ins": self.acc_ins.avg, "train/acc_seg": self.acc_seg.avg})
            self.log_fn(rec)
        self.reset_metrics()

    def reset_metrics(self):
        for m in (self.loss_o, self.loss_i, self.loss_d, self.acc_ins, self.acc_seg,
                  self.cross_image_variance_source, self.cross_image_variance_target):
            m.reset()
