"""Supervised CutPaste / "mirror" pre-training behind the reference's own surface:

    reference                                               here
    datasets/pretrain_dataset.py:176-180  CutPastePatchType      CutPastePatchType (same members / values)
    datasets/pretrain_dataset.py:182-185  MirrorVariant          MirrorVariant
    datasets/pretrain_dataset.py:187-412  CutPasteDataset        CutPasteSampler (the random draws, on the host, in the
                                          (numpy + Pillow, CPU)  reference's order) + cutpaste_batch (HIP, csrc/mirror.hip)
    networks/mirror_network.py:9-86       MirrorModule           MirrorModule (same constructor arguments, forward,
                                          (LightningModule)      shared_step / training_step / validation_step,
                                                                 configure_optimizers; lightning is not installed, so it
                                                                 is a plain nn.Module driven by mirror_pretrain.py)
    networks/segment_network.py:71-93     checkpoint loading     load_pretrained()

The loss section (class cross entropy + compare cross entropy + argmax + confusion counts + both gradients) is one HIP
launch (ops.mirror_loss).  There is no CPU path.
"""
from __future__ import annotations

import math
from enum import Enum
from typing import Dict, List, Optional, Sequence

import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F

from . import ops
from .pretrain_types import PretrainType

BACKGROUND_CLASS = 0


class CutPastePatchType(Enum):
    NONE = 0
    REGULAR = 1
    SCAR = 2


class MirrorVariant(Enum):
    NONE = 0
    OUTPUT = 1


class Stage(Enum):
    TRAIN = 0
    VAL = 1
    TEST = 2
    PSEUDOTEST = 3


_ONE, _HALF = 1 << 16, 1 << 15


def rotate_matrix(w: int, h: int, angle: float):
    """Geometry of `Image.rotate(angle, expand=True)` with the default nearest-neighbour filter, as Pillow computes it:
    the expanded size and the reverse (destination -> source) affine matrix in 16.16 fixed point.  Multiples of 90
    degrees are Pillow's transpose fast paths, written here as exact integer matrices."""
    angle = angle % 360.0
    if angle in (0.0, 180.0):
        s = _ONE if angle == 0.0 else -_ONE
        return w, h, [s, 0, _HALF if s > 0 else (w - 1) * _ONE + _HALF, 0, s, _HALF if s > 0 else (h - 1) * _ONE + _HALF]
    if angle == 90.0:
        return h, w, [0, -_ONE, (w - 1) * _ONE + _HALF, _ONE, 0, _HALF]
    if angle == 270.0:
        return h, w, [0, _ONE, _HALF, -_ONE, 0, (h - 1) * _ONE + _HALF]
    rad = -math.radians(angle)
    c, s = round(math.cos(rad), 15), round(math.sin(rad), 15)
    cx, cy = w / 2, h / 2
    # rotation about the patch centre: x' = c*x + s*y + tx, y' = -s*x + c*y + ty
    tx = c * -cx + s * -cy + 0.0 + cx
    ty = -s * -cx + c * -cy + 0.0 + cy
    corners = [(c * x + s * y + tx, -s * x + c * y + ty) for x, y in ((0, 0), (w, 0), (w, h), (0, h))]
    nw = math.ceil(max(p[0] for p in corners)) - math.floor(min(p[0] for p in corners))
    nh = math.ceil(max(p[1] for p in corners)) - math.floor(min(p[1] for p in corners))
    ex, ey = -(nw - w) / 2.0, -(nh - h) / 2.0
    tx, ty = c * ex + s * ey + tx, -s * ex + c * ey + ty
    fix = lambda v: int(math.floor(v * 65536.0 + 0.5))  # noqa: E731  (Geometry.c FIX)
    return nw, nh, [fix(c), fix(s), fix(tx + c * 0.5 + s * 0.5), fix(-s), fix(c), fix(ty + -s * 0.5 + c * 0.5)]


class CutPasteSampler:
    """The random choices of CutPasteDataset, drawn on the host from `rng` (default: numpy's global state, which is
    what the reference uses) in exactly the reference's order, so the same seed gives the same patches:
    constructor -> per-image class targets (np.random.choice, p = 0.1/0.9 or 0.1/0.45/0.45); per item -> [mirror index],
    patch (area, aspect, [angle], source x, y, paste x, y), number of additional patches, their draws."""

    def __init__(self, num_images: int, min_area_scale: float, max_area_scale: float, min_aspect_ratio: float,
                 max_aspect_ratio: float, min_rotation: float, max_rotation: float, mirror_variant: MirrorVariant,
                 num_classes: int, max_num_patches: int, rng=np.random):
        assert mirror_variant in MirrorVariant
        assert max_num_patches >= 1
        assert max_num_patches == 1 or num_classes <= 2      # pretrain_dataset.py:236
        self.n = num_images
        self.min_area_scale, self.max_area_scale = min_area_scale, max_area_scale
        self.min_aspect_ratio, self.max_aspect_ratio = min_aspect_ratio, max_aspect_ratio
        self.min_rotation, self.max_rotation = min_rotation, max_rotation
        self.mirror_variant, self.max_num_patches, self.rng = mirror_variant, max_num_patches, rng
        self.classes = list(range(num_classes))
        self.targets = rng.choice(self.classes, size=num_images, replace=True,
                                  p=[0.1, 0.45, 0.45] if num_classes == 3 else [0.1, 0.9])

    def draw_patch(self, img_h: int, img_w: int, patch_type: CutPastePatchType) -> List[int]:
        """One cutpaste() call -> [class, patch x, y, w, h, paste x, y, rotated w, h, a0..a5]."""
        rng = self.rng
        if patch_type == CutPastePatchType.REGULAR:
            area_scale = rng.uniform(high=self.max_area_scale, low=self.min_area_scale)
            aspect = rng.uniform(high=self.max_aspect_ratio, low=self.min_aspect_ratio)
            rotation = 0
        elif patch_type == CutPastePatchType.SCAR:
            area_scale = rng.uniform(high=self.max_area_scale * 0.5, low=self.min_area_scale)
            aspect = rng.uniform(3, 6)
            rotation = rng.uniform(low=self.min_rotation, high=self.max_rotation)
        else:
            raise Exception(f"No handling for patch type {patch_type}")
        area = int(img_h * img_w * area_scale)
        ph = int(np.sqrt(area / aspect))
        pw = int(ph * aspect)
        px = rng.randint(0, img_w - pw)
        py = rng.randint(0, img_h - ph)
        rw, rh, mat = rotate_matrix(pw, ph, rotation)
        x_pos = rng.randint(0, img_w - rw)
        y_pos = rng.randint(0, img_h - rh)
        return [patch_type.value, px, py, pw, ph, x_pos, y_pos, rw, rh] + mat

    def draw_item(self, idx: int, img_h: int, img_w: int):
        """__getitem__(idx): (mirror index or -1, [patch rows])."""
        cls = int(self.targets[idx])
        mirror_idx = int(self.rng.randint(self.n)) if self.mirror_variant == MirrorVariant.OUTPUT else -1
        patches = []
        if cls != 0:
            ptype = CutPastePatchType(cls)
            patches.append(self.draw_patch(img_h, img_w, ptype))
            for _ in range(self.rng.randint(self.max_num_patches)):
                patches.append(self.draw_patch(img_h, img_w, ptype))
        return mirror_idx, patches

    def batch_tables(self, indices: Sequence[int], img_h: int, img_w: int) -> List[np.ndarray]:
        """Parameter tables of cp2_cutpaste, one int32 [B, 20] per patch round (round 0 reads the dataset, later rounds
        the previous round's output: source index = position in the batch)."""
        items = [self.draw_item(int(i), img_h, img_w) for i in indices]
        rounds = max(1, max(len(p) for _, p in items))
        tabs = [np.zeros((len(items), ops.CUTPASTE_PARAMS), dtype=np.int32) for _ in range(rounds)]
        for b, (idx, (mirror_idx, patches)) in enumerate(zip(indices, items)):
            for r in range(rounds):
                row = tabs[r][b]
                row[0], row[1] = (int(idx), max(mirror_idx, 0)) if r == 0 else (b, b)
                if r < len(patches):
                    row[2:17] = patches[r]
        return tabs


def cutpaste_batch(images_u8: torch.Tensor, sampler: CutPasteSampler, indices: Sequence[int]):
    """A training batch of the mirror pre-trainer made on the device: (img, mirror_img or None, mask) = what the
    reference's DataLoader collates from CutPasteDataset.__getitem__ (float [B,3,H,W] in [0,1], int64 [B,H,W]).
    images_u8: the resized dataset, uint8 [N,H,W,3], resident in device memory."""
    _, H, W, _ = images_u8.shape
    tabs = sampler.batch_tables(indices, H, W)
    two = sampler.mirror_variant == MirrorVariant.OUTPUT
    table = torch.from_numpy(np.stack(tabs)).to(images_u8.device, non_blocking=True)      # one H2D copy per batch
    src, src_m, mask, out = images_u8, images_u8 if two else None, None, None
    for r in range(len(tabs)):
        last = r == len(tabs) - 1
        out = ops.cutpaste(src, src_m, table[r], mask=mask, want_u8=not last, want_f32=last)
        src, src_m, mask = out["u8"], out["mirror_u8"], out["mask"]
    return out["f32"], out["mirror_f32"], out["mask"]


# ---------------------------------------------------------------------------------------------- loss section
class _MirrorLossFn(torch.autograd.Function):
    """loss = class_loss + lmbd * compare_loss of networks/mirror_network.py:40-63; one launch computes the three
    scalars, d loss / d logits of both views, the argmax maps and the confusion counts."""

    @staticmethod
    def forward(ctx, s_logits, t_logits, masks, softmax_temp, lmbd, confusion, stats):
        need = ctx.needs_input_grad[0] or ctx.needs_input_grad[1]
        out3, gs, gt, am = ops.mirror_loss(s_logits.contiguous(), None if t_logits is None else t_logits.contiguous(), masks.contiguous(),
                                           softmax_temp, lmbd, want_grad=need, want_argmax=True, confusion=confusion)
        ctx.two = t_logits is not None
        ctx.save_for_backward(gs, gt)
        stats["class_loss"], stats["compare_loss"], stats["argmax"] = out3[1], out3[2], am
        return out3[0]

    @staticmethod
    def backward(ctx, g):
        gs, gt = ctx.saved_tensors
        return (None if gs is None else gs * g), (gt * g if ctx.two and gt is not None else None), None, None, None, None, None


def mirror_loss(s_logits, t_logits, masks, softmax_temp, lmbd_compare_loss, confusion=None):
    """-> (loss with autograd graph, dict(class_loss, compare_loss, argmax))."""
    stats: Dict[str, torch.Tensor] = {}
    loss = _MirrorLossFn.apply(s_logits, t_logits, masks, float(softmax_temp), float(lmbd_compare_loss), confusion, stats)
    return loss, stats


def load_pretrained(module: "MirrorModule", checkpoint_path: str, pretrain_type: PretrainType, use_backbone_only: bool = False):
    """The checkpoint hand-off of networks/segment_network.py:71-101 for the types this repository produces."""
    checkpoint = torch.load(checkpoint_path, map_location="cpu")
    if pretrain_type in (PretrainType.CP2, PretrainType.MOCO, PretrainType.BYOL, PretrainType.PROPOSED, PretrainType.DENSECL,
                         PretrainType.PROPOSED_V2):
        assert checkpoint["pretrain_type"] == pretrain_type.name, f"{checkpoint['pretrain_type']} != {pretrain_type}"
        flt = "encoder_q.backbone" if use_backbone_only else "encoder_q."
        sd = {k.replace("module.encoder_q.", ""): v for k, v in checkpoint["state_dict"].items() if flt in k}
        sd = {k: v for k, v in sd.items() if "conv_seg" not in k}       # num_classes differs
        return module.model.load_state_dict(sd, strict=False)
    if pretrain_type == PretrainType.MIRROR:
        sd = {k: v for k, v in checkpoint["state_dict"].items() if "conv_seg" not in k}
        return module.load_state_dict(sd, strict=False)
    raise NotImplementedError(f"{pretrain_type = }")


class MirrorModule(nn.Module):
    """networks/mirror_network.py MirrorModule on top of networks/segment_network.py SegmentationModule: a segmentor
    whose logits are resized to the image size; loss = cross entropy against the CutPaste masks (both views) +
    lmbd_compare_loss * cross entropy between the tempered softmaxes of the two views."""

    def __init__(self, model_config, pretrain_type: PretrainType, learning_rate, weight_decay, num_classes, image_shape,
                 lmbd_compare_loss, softmax_temp, mirror_variant, use_backbone_only: bool = False, amp_dtype=None):
        super().__init__()
        from .encoder import build_segmentor
        assert pretrain_type in PretrainType
        assert mirror_variant in MirrorVariant
        if not 2 <= num_classes <= ops.MIRROR_MAX_CLASSES:
            raise ValueError(f"num_classes must be in [2, {ops.MIRROR_MAX_CLASSES}]")
        self.model = build_segmentor(model_config.model)
        if pretrain_type == PretrainType.NONE:
            # the reference initialises from 'torchvision://resnet50' (a download); here only a local file works
            ckpt = (getattr(model_config.model.backbone, "init_cfg", None) or {}).get("checkpoint", "")
            if not ckpt or str(ckpt).startswith("torchvision://"):
                raise RuntimeError("PretrainType.NONE needs a local ImageNet checkpoint in backbone.init_cfg.checkpoint "
                                   "(no network here); use PretrainType.RANDOM to train from scratch")
            self.model.backbone.init_weights(pretrained=ckpt)
        elif pretrain_type != PretrainType.RANDOM:
            ckpt = model_config.model.backbone.init_cfg["checkpoint"]
            print(load_pretrained(self, ckpt, pretrain_type, use_backbone_only))
        self.learning_rate, self.weight_decay = learning_rate, weight_decay
        self.num_classes, self.image_shape = num_classes, image_shape
        self.lmbd_compare_loss, self.softmax_temp = lmbd_compare_loss, softmax_temp
        self.mirror_variant = mirror_variant
        self.amp_dtype = amp_dtype
        # confusion counts (row = ground truth, column = prediction) per stage: what the reference's torchmetrics
        # collection (segment_network.py:176-214) is computed from
        for st in (Stage.TRAIN, Stage.VAL):
            self.register_buffer(f"confusion_{st.name.lower()}", torch.zeros(num_classes, num_classes, dtype=torch.int64),
                                 persistent=False)
        self.logged: Dict[str, torch.Tensor] = {}

    def log(self, name, value, **kw):
        self.logged[name] = value.detach() if isinstance(value, torch.Tensor) else value

    def forward(self, images):
        if self.amp_dtype is not None:
            with torch.autocast("cuda", dtype=self.amp_dtype):
                logits = self.model(images)
            logits = logits.float()
        else:
            logits = self.model(images)
        logits = F.interpolate(logits, size=tuple(self.image_shape[1:]), mode="bilinear", align_corners=False)  # 32 -> 512
        return logits, logits.argmax(dim=1)

    def _logits(self, images):
        return self.forward(images)[0]

    def shared_step(self, batch, stage: Stage):
        conf = getattr(self, f"confusion_{stage.name.lower()}", None)
        if self.mirror_variant == MirrorVariant.OUTPUT:
            s_img, t_img, masks = batch
            # two separate passes as the reference makes them (BatchNorm statistics are per view)
            loss, stats = mirror_loss(self._logits(s_img), self._logits(t_img), masks, self.softmax_temp,
                                      self.lmbd_compare_loss, conf)
        elif self.mirror_variant == MirrorVariant.NONE:
            img, masks = batch
            loss, stats = mirror_loss(self._logits(img), None, masks, self.softmax_temp, self.lmbd_compare_loss, conf)
        else:
            raise NotImplementedError(f"{self.mirror_variant = }")
        s = stage.name.lower()
        self.log(f"{s}_loss", loss)
        self.log(f"{s}_compare_loss", stats["compare_loss"])
        self.log(f"{s}_class_loss", stats["class_loss"])
        self.last_argmax = stats["argmax"]
        return loss

    def training_step(self, batch, batch_idx=0):
        return self.shared_step(batch, Stage.TRAIN)

    def validation_step(self, batch, batch_idx=0, dataloader_idx=0):
        return self.shared_step(batch, Stage.VAL)

    def metrics(self, stage: Stage, reset: bool = True) -> Dict[str, float]:
        """Micro-averaged Jaccard / Dice / precision / recall / F1 of the foreground classes from the accumulated
        confusion counts (binary task: class 1 is the positive; multi-class: class 0 ignored, as the reference sets
        ignore_index).  torchmetrics is not installed, so these follow the textbook definitions (unpinned)."""
        buf = getattr(self, f"confusion_{stage.name.lower()}")
        counts = buf.clone()
        import torch.distributed as dist
        if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
            dist.all_reduce(counts)                      # every rank counted its own shard (torchmetrics' dist_sync_on_step=False + sync at compute)
        c = counts.double().cpu()
        if reset:
            buf.zero_()
        fg = slice(1, None)
        tp = c.diag()[fg].sum()
        fp = c[:, fg].sum() - tp - (0 if self.num_classes == 2 else c[0, fg].sum())   # ignored ground truth is dropped
        fn = c[fg, :].sum() - tp
        div = lambda a, b: float(a / b) if float(b) > 0 else 0.0  # noqa: E731
        pre, rec = div(tp, tp + fp), div(tp, tp + fn)
        p = f"{stage.name.lower()}_"
        return {p + "jaccard": div(tp, tp + fp + fn), p + "dice": div(2 * tp, 2 * tp + fp + fn), p + "precision": pre,
                p + "recall": rec, p + "f1": div(2 * pre * rec, pre + rec)}

    def configure_optimizers(self):
        optimizer = torch.optim.Adam(self.parameters(), lr=self.learning_rate, weight_decay=self.weight_decay)
        return {"optimizer": optimizer}
