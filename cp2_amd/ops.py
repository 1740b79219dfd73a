"""Torch-tensor front end of the C ABI (include/cp2hip.h).

Each function checks device / dtype / layout on the host, passes raw device
pointers and the CURRENT torch stream to libcp2hip.so, and raises on any
non-zero return code.  Tensors must live on the GPU: there is no CPU path.
"""
from __future__ import annotations

from typing import List, Optional, Sequence, Tuple

import numpy as np
import torch

from . import _lib


_raw_stream = getattr(torch._C, "_cuda_getCurrentRawStream", None)


def _stream() -> int:
    """Handle of the current HIP stream.  torch.cuda.current_stream().cuda_stream builds a Python Stream object per
    call (~10 us, 1.3 ms per training step over all ops); the raw accessor returns the same handle in ~0.3 us."""
    if _raw_stream is not None:
        return _raw_stream(torch.cuda.current_device())
    return torch.cuda.current_stream().cuda_stream


def _dev(t: torch.Tensor, name: str, dtype=None) -> int:
    if not isinstance(t, torch.Tensor):
        raise TypeError(f"{name}: expected a torch.Tensor")
    if not t.is_cuda:
        raise _lib.Cp2LibraryError(f"{name}: tensor is on {t.device}; cp2_amd ops run on the GPU only")
    if dtype is not None and t.dtype != dtype:
        raise TypeError(f"{name}: expected {dtype}, got {t.dtype}")
    if not t.is_contiguous():
        raise ValueError(f"{name}: tensor must be contiguous")
    return t.data_ptr()


def _opt(t: Optional[torch.Tensor], name: str, dtype=None) -> Optional[int]:
    return None if t is None else _dev(t, name, dtype)


# bench.py sets PROFILE to a dict {kernel family: [hipevents.EventPair, ...]}: each profiled launch then carries its
# own start / stop events (cp2_profile_next_launch), i.e. kernel-exact durations measured inside the run.
PROFILE = None


def _profile(name: str) -> None:
    if PROFILE is None or torch.cuda.is_current_stream_capturing():
        return
    from .hipevents import EventPair
    ev = EventPair()
    _lib.check(_lib.load().cp2_profile_next_launch(ev.start, ev.stop), "cp2_profile_next_launch")
    PROFILE.setdefault(name, []).append(ev)


def ds_size(n: int, stride: int) -> int:
    """len(range(stride // 2, n, stride))"""
    return (n - stride // 2 + stride - 1) // stride


# ---------------------------------------------------------------- a1 / a2
def compose_mask(img: torch.Tensor, bg: torch.Tensor, stride: int = 0, want_full_mask: bool = False):
    """Copy-paste composition (reference builder.py:1146-1159).
    Returns (composed image, full-resolution mask or None, down-sampled mask or None)."""
    lib = _lib.load()
    B, ch, H, W = img.shape
    if ch != 3 or bg.shape != img.shape:
        raise ValueError(f"compose_mask: img/bg must both be [B,3,H,W], got {tuple(img.shape)} {tuple(bg.shape)}")
    out = torch.empty_like(img)
    mfull = torch.empty((B, H, W), dtype=torch.float32, device=img.device) if want_full_mask else None
    mds = None
    if stride > 0:
        mds = torch.empty((B, ds_size(H, stride), ds_size(W, stride)), dtype=torch.float32, device=img.device)
    rc = lib.cp2_compose_mask(_dev(img, "img", torch.float32), _dev(bg, "bg", torch.float32), out.data_ptr(),
                              _opt(mfull, "mask_full"), _opt(mds, "mask_ds"), B, H, W, max(stride, 1), _stream())
    _lib.check(rc, "cp2_compose_mask")
    return out, mfull, mds


def compose_pair(img_a, bg0, img_b, bg1, stride: int, row_b: Optional[torch.Tensor] = None, channels_last: bool = False,
                 out_dtype: torch.dtype = torch.float32):
    """Both views of the step in one launch (reference builder.py:1146-1159): returns (out_a, out_b, mask_ds_a, mask_ds_b).
    row_b (int64 [B] on the GPU): out_b[j] = compose(img_b[row_b[j]], bg1[row_b[j]]) -- the shuffle-BN gather of
    builder.py:630 folded in; mask_ds_b stays in the original order.  channels_last / out_dtype=torch.bfloat16: the layout
    and the precision the stem convolution reads (bf16 = round-to-nearest-even of the fp32 value, as autocast's cast)."""
    lib = _lib.load()
    B, ch, H, W = img_a.shape
    if ch != 3 or any(tuple(t.shape) != (B, 3, H, W) for t in (bg0, img_b, bg1)):
        raise ValueError("compose_pair: four [B,3,H,W] tensors expected")
    if out_dtype not in (torch.float32, torch.bfloat16):
        raise TypeError("compose_pair: out_dtype float32 or bfloat16")
    fmt = torch.channels_last if channels_last else torch.contiguous_format
    out_a = torch.empty((B, 3, H, W), dtype=out_dtype, device=img_a.device, memory_format=fmt)
    out_b = torch.empty((B, 3, H, W), dtype=out_dtype, device=img_a.device, memory_format=fmt)
    md_a = torch.empty((B, ds_size(H, stride), ds_size(W, stride)), dtype=torch.float32, device=img_a.device)
    md_b = torch.empty_like(md_a)
    if row_b is not None and (row_b.numel() != B or row_b.dtype != torch.int64):
        raise ValueError("compose_pair: row_b must hold B int64 indices")
    _profile("compose_pair")
    rc = lib.cp2_compose_pair(_dev(img_a, "img_a", torch.float32), _dev(bg0, "bg0", torch.float32), _dev(img_b, "img_b", torch.float32),
                              _dev(bg1, "bg1", torch.float32), out_a.data_ptr(), out_b.data_ptr(), md_a.data_ptr(), md_b.data_ptr(),
                              _opt(row_b, "row_b", torch.int64), B, H, W, stride, int(channels_last),
                              int(out_dtype == torch.bfloat16), _stream())
    _lib.check(rc, "cp2_compose_pair")
    return out_a, out_b, md_a, md_b


def strided_gather(x: torch.Tensor, stride: int) -> torch.Tensor:
    """x[:, s//2::s, s//2::s] for [B,H,W] float32 / int64 (reference builder.py:1155-1186)."""
    lib = _lib.load()
    B, H, W = x.shape
    y = torch.empty((B, ds_size(H, stride), ds_size(W, stride)), dtype=x.dtype, device=x.device)
    if x.dtype == torch.float32:
        rc = lib.cp2_strided_gather_f32(_dev(x, "x"), y.data_ptr(), B, H, W, stride, _stream())
    elif x.dtype == torch.int64:
        rc = lib.cp2_strided_gather_i64(_dev(x, "x"), y.data_ptr(), B, H, W, stride, _stream())
    else:
        raise TypeError(f"strided_gather: float32 or int64 expected, got {x.dtype}")
    _lib.check(rc, "cp2_strided_gather")
    return y


def gather_rows(src: torch.Tensor, idx: torch.Tensor, err_flag: Optional[torch.Tensor] = None) -> torch.Tensor:
    """dst[r] = src[idx[r]] along dim 0 (shuffle-BN take, reference builder.py:630,649)."""
    lib = _lib.load()
    n_src = src.shape[0]
    row_elems = src[0].numel()
    rows = idx.numel()
    dst = torch.empty((rows,) + tuple(src.shape[1:]), dtype=src.dtype, device=src.device)
    rc = lib.cp2_gather_rows_f32(_dev(src, "src", torch.float32), _dev(idx, "idx", torch.int64), dst.data_ptr(),
                                 rows, n_src, row_elems, _opt(err_flag, "err_flag", torch.int32), _stream())
    _lib.check(rc, "cp2_gather_rows_f32")
    return dst


# ---------------------------------------------------------------- f1: on-device augmentation
def crop_resize_flip(src: torch.Tensor, src_region: Optional[torch.Tensor], params: torch.Tensor, H: int, W: int,
                     id_stride: int = 1, want_ids: bool = True, want_f32: bool = True, out_rgbx: Optional[torch.Tensor] = None):
    """RandomResizedCrop + HorizontalFlip with explicit parameters (reference loader.py:50-118, main.py:206-216).
    src: [N,3,Hs,Ws] uint8 / float32 on the GPU; params: int32 [B,8] device table (augment.crop_table).
    Returns (img [B,3,H,W] f32 or None, pixel_ids [B,H,W] int64 or None, region_ids or None); out_rgbx: optional int32
    [B,H,W] tensor that receives the view as packed uint8 RGB for the photometric stages (color_ops, blur_to_tensor)."""
    lib = _lib.load()
    if not src.is_cuda or src.dtype not in (torch.uint8, torch.float32) or src.dim() != 4 or src.shape[1] != 3:
        raise _lib.Cp2LibraryError("crop_resize_flip: src must be a uint8 / float32 [N,3,Hs,Ws] GPU tensor")
    N, _, Hs, Ws = src.shape
    B = params.shape[0]
    if not want_f32 and out_rgbx is None:
        raise ValueError("crop_resize_flip: nothing to write (want_f32=False and no out_rgbx)")
    if out_rgbx is not None and (tuple(out_rgbx.shape) != (B, H, W) or out_rgbx.dtype != torch.int32):
        raise ValueError("crop_resize_flip: out_rgbx must be int32 [B,H,W]")
    img = torch.empty((B, 3, H, W), dtype=torch.float32, device=src.device) if want_f32 else None
    pix = torch.empty((B, H, W), dtype=torch.int64, device=src.device) if want_ids else None
    reg = torch.empty((B, H, W), dtype=torch.int64, device=src.device) if want_ids else None
    rc = lib.cp2_crop_resize_flip(_dev(src, "src"), int(src.dtype == torch.uint8), _opt(src_region, "src_region", torch.int64),
                                  N, Hs, Ws, _dev(params, "params", torch.int32), _opt(img, "img"), _opt(pix, "pix"),
                                  _opt(reg, "reg"), B, H, W, int(id_stride), _opt(out_rgbx, "out_rgbx"), _stream())
    _lib.check(rc, "cp2_crop_resize_flip")
    return img, pix, reg


def pil_resize_crop(src: torch.Tensor, params: torch.Tensor, H: int, W: int, out_rgbx: Optional[torch.Tensor] = None) -> torch.Tensor:
    """Background RandomResizedCrop + flip in Pillow's arithmetic: img.crop(box).resize((W, H), BILINEAR)
    (reference main.py:209-211,217).  src: uint8 [N,3,Hs,Ws]; params as crop_resize_flip.  Returns int32 [B,H,W] packed
    uint8 RGB (R | G<<8 | B<<16)."""
    lib = _lib.load()
    if not src.is_cuda or src.dtype != torch.uint8 or src.dim() != 4 or src.shape[1] != 3:
        raise _lib.Cp2LibraryError("pil_resize_crop: src must be a uint8 [N,3,Hs,Ws] GPU tensor (PIL images are uint8)")
    N, _, Hs, Ws = src.shape
    B = params.shape[0]
    if out_rgbx is None:
        out_rgbx = torch.empty((B, H, W), dtype=torch.int32, device=src.device)
    elif tuple(out_rgbx.shape) != (B, H, W) or out_rgbx.dtype != torch.int32:
        raise ValueError("pil_resize_crop: out_rgbx must be int32 [B,H,W]")
    nbytes = lib.cp2_pil_resize_workspace_bytes(B, Hs, Ws, H, W)
    if nbytes <= 0:
        raise ValueError("pil_resize_crop: bad shapes")
    ws = torch.empty(nbytes // 4, dtype=torch.int32, device=src.device)
    rc = lib.cp2_pil_resize_crop(_dev(src, "src"), N, Hs, Ws, _dev(params, "params", torch.int32), _dev(out_rgbx, "out_rgbx"),
                                 B, H, W, ws.data_ptr(), nbytes, _stream())
    _lib.check(rc, "cp2_pil_resize_crop")
    return out_rgbx


COLOR_PARAMS = 12          # CP2_COLOR_PARAMS
BLUR_MAX_RADIUS = 4        # CP2_BLUR_MAX_RADIUS


def color_ops(rgbx: torch.Tensor, params: torch.Tensor) -> None:
    """In place ColorJitter + RandomGrayscale on packed uint8 RGB [B,H,W] (reference main.py:212-215; Pillow's
    ImageEnhance / adjust_hue / convert("L") arithmetic).  params: int32 [B,12] (augment.jitter_table)."""
    lib = _lib.load()
    B, H, W = rgbx.shape
    if rgbx.dtype != torch.int32 or tuple(params.shape) != (B, COLOR_PARAMS):
        raise ValueError(f"color_ops: rgbx int32 [B,H,W] and params int32 [B,{COLOR_PARAMS}] expected")
    lsum = torch.empty(B, dtype=torch.int64, device=rgbx.device)
    rc = lib.cp2_color_ops(_dev(rgbx, "rgbx"), _dev(params, "params", torch.int32), lsum.data_ptr(), B, H, W, _stream())
    _lib.check(rc, "cp2_color_ops")


def blur_to_tensor(rgbx: torch.Tensor, params: torch.Tensor, rects: Optional[torch.Tensor], rmax: int) -> torch.Tensor:
    """GaussianBlur (Pillow's three box passes per axis) + ToTensor + RandomErasing(value=0) in one pass (reference
    main.py:216-224, loader.py:121-152).  params: int32 [B,4] (augment.blur_table), rects: int32 [B,4] or None.
    Returns float32 [B,3,H,W]."""
    lib = _lib.load()
    B, H, W = rgbx.shape
    if rgbx.dtype != torch.int32 or tuple(params.shape) != (B, 4) or (rects is not None and tuple(rects.shape) != (B, 4)):
        raise ValueError("blur_to_tensor: rgbx int32 [B,H,W], params int32 [B,4], rects int32 [B,4] expected")
    out = torch.empty((B, 3, H, W), dtype=torch.float32, device=rgbx.device)
    rc = lib.cp2_blur_to_tensor(_dev(rgbx, "rgbx"), _dev(params, "params", torch.int32), _opt(rects, "rects", torch.int32),
                                out.data_ptr(), B, H, W, int(rmax), _stream())
    _lib.check(rc, "cp2_blur_to_tensor")
    return out


def erase_rect(img: torch.Tensor, rects: torch.Tensor) -> None:
    """In place RandomErasing(value=0): img[b,:,top:top+h,left:left+w] = 0; rects int32 [B,4] on the GPU (main.py:218-224)."""
    lib = _lib.load()
    B, ch, H, W = img.shape
    if ch != 3 or rects.shape != (B, 4):
        raise ValueError("erase_rect: img [B,3,H,W] and rects [B,4] expected")
    rc = lib.cp2_erase_rect(_dev(img, "img", torch.float32), _dev(rects, "rects", torch.int32), B, H, W, _stream())
    _lib.check(rc, "cp2_erase_rect")


# ---------------------------------------------------------------- a3-a5
def corr_iou(ids_a: torch.Tensor, ids_b: torch.Tensor, mask_a: Optional[torch.Tensor] = None,
             mask_b: Optional[torch.Tensor] = None) -> Tuple[torch.Tensor, Optional[torch.Tensor]]:
    """(iou, iou_masked) of two id maps (reference tools/correlation_mapping.py:103-138,173,231).
    ids: int64 [B,...]; masks: float32 of the same element count, or None to skip iou_masked."""
    lib = _lib.load()
    B = ids_a.shape[0]
    a, b = ids_a.reshape(B, -1), ids_b.reshape(B, -1)
    P = a.shape[1]
    iou = torch.empty(B, dtype=torch.float32, device=a.device)
    ioum = torch.empty(B, dtype=torch.float32, device=a.device) if mask_a is not None else None
    ma = mask_a.reshape(B, -1) if mask_a is not None else None
    mb = mask_b.reshape(B, -1) if mask_b is not None else None
    rc = lib.cp2_corr_iou(_dev(a, "ids_a", torch.int64), _dev(b, "ids_b", torch.int64), _opt(ma, "mask_a", torch.float32),
                          _opt(mb, "mask_b", torch.float32), iou.data_ptr(), _opt(ioum, "iou_masked"), B, P, _stream())
    _lib.check(rc, "cp2_corr_iou")
    return iou, ioum


def corr_iou_strided(ids_a: torch.Tensor, ids_b: torch.Tensor, stride: int, mask_a: Optional[torch.Tensor] = None,
                     mask_b: Optional[torch.Tensor] = None) -> Tuple[torch.Tensor, Optional[torch.Tensor]]:
    """corr_iou of the centre-tap down-sampled id maps (builder.py:1155-1186 + tools/correlation_mapping.py:103-138)
    without materialising them: ids are the full-resolution int64 [B,H,W] maps, masks the down-sampled [B,P] ones."""
    lib = _lib.load()
    B, H, W = ids_a.shape
    P = ds_size(H, stride) * ds_size(W, stride)
    iou = torch.empty(B, dtype=torch.float32, device=ids_a.device)
    ioum = torch.empty(B, dtype=torch.float32, device=ids_a.device) if mask_a is not None else None
    ma = mask_a.reshape(B, -1) if mask_a is not None else None
    mb = mask_b.reshape(B, -1) if mask_b is not None else None
    if ma is not None and (ma.shape[1] != P or mb.shape[1] != P):
        raise ValueError(f"corr_iou_strided: masks must have {P} elements per sample")
    rc = lib.cp2_corr_iou_strided(_dev(ids_a, "ids_a", torch.int64), _dev(ids_b, "ids_b", torch.int64), _opt(ma, "mask_a", torch.float32),
                                  _opt(mb, "mask_b", torch.float32), iou.data_ptr(), _opt(ioum, "iou_masked"), B, H, W, stride, _stream())
    _lib.check(rc, "cp2_corr_iou_strided")
    return iou, ioum


# ---------------------------------------------------------------- a11
def ema_scalars(m: float) -> Tuple[float, float]:
    """fp32(m) and fp32(1.0 - m) -- the subtraction is done in double first, as the
    reference's Python expression `1.0 - self.momentum` does (builder.py:565-567)."""
    return float(np.float32(m)), float(np.float32(1.0 - m))


def ema_flat(k: torch.Tensor, q: torch.Tensor, m: float) -> None:
    """In place k = k*m + q*(1-m) over one flat fp32 span."""
    lib = _lib.load()
    if k.numel() != q.numel():
        raise ValueError("ema_flat: k and q differ in size")
    m32, om32 = ema_scalars(m)
    rc = lib.cp2_ema_flat(_dev(k, "k", torch.float32), _dev(q, "q", torch.float32), k.numel(), m32, om32, _stream())
    _lib.check(rc, "cp2_ema_flat")


def ema_flat_shadow(k: torch.Tensor, q: torch.Tensor, k_bf16: torch.Tensor, m: float, events=None) -> None:
    """ema_flat that also stores bf16(k_new) into `k_bf16` (same element order); optional hipevents.EventPair."""
    lib = _lib.load()
    if k_bf16.numel() != k.numel():
        raise ValueError("ema_flat_shadow: shadow buffer differs in size")
    m32, om32 = ema_scalars(m)
    rc = lib.cp2_ema_flat_shadow(_dev(k, "k", torch.float32), _dev(q, "q", torch.float32), _dev(k_bf16, "k_bf16", torch.bfloat16),
                                 k.numel(), m32, om32, events.start if events else None, events.stop if events else None,
                                 _stream())
    _lib.check(rc, "cp2_ema_flat_shadow")


def ema_flat_timed(k: torch.Tensor, q: torch.Tensor, m: float, events) -> None:
    """ema_flat with an `hipevents.EventPair` bracketing exactly this kernel."""
    lib = _lib.load()
    m32, om32 = ema_scalars(m)
    rc = lib.cp2_ema_flat_timed(_dev(k, "k", torch.float32), _dev(q, "q", torch.float32), k.numel(), m32, om32,
                                events.start, events.stop, _stream())
    _lib.check(rc, "cp2_ema_flat_timed")


def bf16_image(src: torch.Tensor, dst: torch.Tensor) -> None:
    """dst = bf16(src) for flat buffers of equal length (a multiple of 4)."""
    lib = _lib.load()
    if src.numel() != dst.numel():
        raise ValueError("bf16_image: buffers differ in size")
    rc = lib.cp2_bf16_image(_dev(src, "src", torch.float32), _dev(dst, "dst", torch.bfloat16), src.numel(), _stream())
    _lib.check(rc, "cp2_bf16_image")


SGD_BLOCK_FLOATS = 512      # floats per workgroup of sgd_flat_kernel (csrc/sgd.hip)


class SgdFlatPlan:
    """Static tables of cp2_sgd_flat for a flat parameter buffer: `offsets[i]`, `numels[i]` = slot start and element
    count of tensor i (slots are 256-byte aligned, see MODEL.flatten_parameters)."""

    def __init__(self, offsets, numels, device):
        import ctypes

        import numpy as np
        rows, first = [], [0]
        for t, (off, n) in enumerate(zip(offsets, numels)):
            if off % 4:
                raise ValueError("SgdFlatPlan: slots must start on 16-byte boundaries")
            for g in range(0, n, SGD_BLOCK_FLOATS):
                rows.append((t, off + g, g, min(SGD_BLOCK_FLOATS, n - g)))
            first.append(len(rows))
        self.ntensors = len(numels)
        self.blk_tab = torch.from_numpy(np.asarray(rows, dtype=np.int32).reshape(-1, 4)).to(device)
        self.first = (ctypes.c_int32 * len(first))(*first)
        self.grads = (ctypes.c_void_p * self.ntensors)()


def sgd_flat(plan: SgdFlatPlan, p: torch.Tensor, buf: torch.Tensor, p_bf16: Optional[torch.Tensor], lr, momentum: float,
             weight_decay: float) -> None:
    """One SGD(momentum, weight decay) step on the flat buffer; plan.grads[i] must hold the device pointer of tensor
    i's fp32 gradient (element order of its slot) or None.  `lr`: float, or a one-element fp32 device tensor."""
    import numpy as np
    lib = _lib.load()
    lr_dev = None
    if isinstance(lr, torch.Tensor):
        lr_dev, lr = _dev(lr, "lr", torch.float32), 0.0
    _profile("sgd_flat")
    rc = lib.cp2_sgd_flat(_dev(p, "p", torch.float32), _dev(buf, "momentum_buf", torch.float32),
                          _opt(p_bf16, "p_bf16", torch.bfloat16), plan.grads, plan.ntensors, plan.blk_tab.data_ptr(),
                          plan.first, float(np.float32(lr)), lr_dev, float(np.float32(momentum)),
                          float(np.float32(weight_decay)), _stream())
    _lib.check(rc, "cp2_sgd_flat")


def pack_grads(plan: SgdFlatPlan, flat_grad: torch.Tensor, grad_ptrs, t_begin: int, t_end: int, scale: float) -> None:
    """flat_grad[slot of tensor t] = scale * gradient t for t in [t_begin, t_end), one launch (cp2_pack_grads).
    grad_ptrs: ctypes array of ntensors device pointers (None: the slot is zeroed)."""
    import numpy as np
    if not 0 <= t_begin < t_end <= plan.ntensors:
        raise ValueError(f"pack_grads: tensor range [{t_begin}, {t_end}) outside 0..{plan.ntensors}")
    rc = _lib.load().cp2_pack_grads(_dev(flat_grad, "flat_grad", torch.float32), grad_ptrs, t_begin, t_end,
                                    plan.blk_tab.data_ptr(), plan.first, float(np.float32(scale)), _stream())
    _lib.check(rc, "cp2_pack_grads")


class EmaMultiPlan:
    """Device tables for cp2_ema_multi over two parameter lists (built once, reused every step)."""

    CHUNK = 1 << 16

    def __init__(self, params_k: Sequence[torch.Tensor], params_q: Sequence[torch.Tensor]):
        assert len(params_k) == len(params_q) and len(params_k) > 0
        dev = params_k[0].device
        for pk, pq in zip(params_k, params_q):
            _dev(pk, "param_k", torch.float32), _dev(pq, "param_q", torch.float32)
            if pk.numel() != pq.numel():
                raise ValueError("EmaMultiPlan: parameter lists differ in shape")
        self.keys = [(pk.data_ptr(), pq.data_ptr(), pk.numel()) for pk, pq in zip(params_k, params_q)]
        tens, offs, lens = [], [], []
        for t, pk in enumerate(params_k):
            n = pk.numel()
            for off in range(0, n, self.CHUNK):
                tens.append(t), offs.append(off), lens.append(min(self.CHUNK, n - off))
        self.n_chunks = len(tens)
        self.k_ptrs = torch.tensor([p.data_ptr() for p in params_k], dtype=torch.int64, device=dev)
        self.q_ptrs = torch.tensor([p.data_ptr() for p in params_q], dtype=torch.int64, device=dev)
        self.chunk_tensor = torch.tensor(tens, dtype=torch.int32, device=dev)
        self.chunk_off = torch.tensor(offs, dtype=torch.int64, device=dev)
        self.chunk_len = torch.tensor(lens, dtype=torch.int32, device=dev)

    def matches(self, params_k, params_q) -> bool:
        return self.keys == [(pk.data_ptr(), pq.data_ptr(), pk.numel()) for pk, pq in zip(params_k, params_q)]

    def run(self, m: float) -> None:
        lib = _lib.load()
        m32, om32 = ema_scalars(m)
        rc = lib.cp2_ema_multi(self.k_ptrs.data_ptr(), self.q_ptrs.data_ptr(), self.chunk_tensor.data_ptr(),
                               self.chunk_off.data_ptr(), self.chunk_len.data_ptr(), self.n_chunks, m32, om32, _stream())
        _lib.check(rc, "cp2_ema_multi")


# ---------------------------------------------------------------- a13
_ENQUEUE_TICKETS = {}


def _enqueue_ticket(queue: torch.Tensor) -> torch.Tensor:
    """One zeroed int32 per queue buffer: the ticket counter of cp2_enqueue (the kernel leaves it zero).  Keyed by the
    queue's storage, so `queue` and `queue2`, or enqueues issued on different streams for different queues, never share
    a counter; a launch that failed re-zeroes its counter (enqueue())."""
    key = (queue.device.index, queue.data_ptr())
    t = _ENQUEUE_TICKETS.get(key)
    if t is None:
        if len(_ENQUEUE_TICKETS) >= 64:                   # queues come and go in tests: drop the oldest entry
            _ENQUEUE_TICKETS.pop(next(iter(_ENQUEUE_TICKETS)))
        t = _ENQUEUE_TICKETS[key] = torch.zeros(1, dtype=torch.int32, device=queue.device)
    return t


def enqueue(queue: torch.Tensor, keys: torch.Tensor, ptr: torch.Tensor) -> None:
    """In place: queue[:, (ptr+i) % K] = keys[i]; ptr = (ptr + n) % K, all on the device, one launch
    (reference builder.py:569-587; keys are already gathered over ranks)."""
    lib = _lib.load()
    C, K = queue.shape
    n, c2 = keys.shape
    if c2 != C:
        raise ValueError(f"enqueue: keys have {c2} channels, queue has {C}")
    if ptr.numel() != 1:
        raise ValueError("enqueue: ptr must hold one int64")
    ticket = _enqueue_ticket(queue)
    rc = lib.cp2_enqueue(_dev(queue, "queue", torch.float32), _dev(keys, "keys", torch.float32),
                         _dev(ptr, "queue_ptr", torch.int64), ticket.data_ptr(), n, C, K, _stream())
    if rc:
        ticket.zero_()                                    # re-establish the "kernels leave it zero" invariant
    _lib.check(rc, "cp2_enqueue")


# ---------------------------------------------------------------- a7
def _feat_strides(feat: torch.Tensor) -> Tuple[int, int, int, int, int, int]:
    """(B, C, P, stride_n, stride_c, stride_p) of an encoder output [B,C,h,w] or [B,C,P]
    whose pixel axes collapse to one stride (true for NCHW-contiguous and channels-last)."""
    if feat.dim() == 4:
        B, C, h, w = feat.shape
        sn, sc, sh, sw = feat.stride()
        if h > 1 and sh != w * sw:
            raise ValueError("feature map pixels are not uniformly strided; call .contiguous() first")
        return B, C, h * w, sn, sc, sw
    B, C, P = feat.shape
    sn, sc, sp = feat.stride()
    return B, C, P, sn, sc, sp


def feat_normalize_pool(feat: torch.Tensor, mask: torch.Tensor):
    """dense [B,C,P], inv_norm [B,P], pool_partial [B,NT,2,C] (reference builder.py:1261-1268)."""
    lib = _lib.load()
    if not feat.is_cuda:
        raise _lib.Cp2LibraryError(f"feat is on {feat.device}; cp2_amd ops run on the GPU only")
    if feat.dtype != torch.float32:
        raise TypeError(f"feat: expected float32, got {feat.dtype}")
    B, C, P, sn, sc, sp = _feat_strides(feat)
    dense = torch.empty((B, C, P), dtype=torch.float32, device=feat.device)
    inv_norm = torch.empty((B, P), dtype=torch.float32, device=feat.device)
    partial = torch.empty((B, (P + 63) // 64, 2, C), dtype=torch.float32, device=feat.device)
    rc = lib.cp2_feat_normalize_pool(feat.data_ptr(), sn, sc, sp, _dev(mask, "mask", torch.float32), dense.data_ptr(),
                                     inv_norm.data_ptr(), partial.data_ptr(), B, C, P, _stream())
    _lib.check(rc, "cp2_feat_normalize_pool")
    return dense, inv_norm, partial


def pool_finalize(q_partial: torch.Tensor, k_partial: torch.Tensor, P: int):
    lib = _lib.load()
    B, _, _, C = q_partial.shape
    dev = q_partial.device
    vec = lambda: torch.empty((B, C), dtype=torch.float32, device=dev)  # noqa: E731
    q_pos, q_neg, k_pos, k_neg = vec(), vec(), vec(), vec()
    q_norms = torch.empty((B, 2), dtype=torch.float32, device=dev)
    extras = torch.empty((B, 3), dtype=torch.float32, device=dev)
    rc = lib.cp2_pool_finalize(_dev(q_partial, "q_partial"), _dev(k_partial, "k_partial"), q_pos.data_ptr(),
                               q_neg.data_ptr(), q_norms.data_ptr(), k_pos.data_ptr(), k_neg.data_ptr(),
                               extras.data_ptr(), B, C, P, _stream())
    _lib.check(rc, "cp2_pool_finalize")
    return q_pos, q_neg, q_norms, k_pos, k_neg, extras


def pool_bwd(drow_pos, dE, q_pos, q_neg, k_pos, k_neg, q_norms, include_background: bool):
    lib = _lib.load()
    B, C = q_pos.shape
    ds_pos, ds_neg = torch.empty_like(q_pos), torch.empty_like(q_pos)
    rc = lib.cp2_pool_bwd(_dev(drow_pos, "drow_pos"), _dev(dE, "dE"), _dev(q_pos, "q_pos"), _dev(q_neg, "q_neg"),
                          _dev(k_pos, "k_pos"), _dev(k_neg, "k_neg"), _dev(q_norms, "q_norms"), int(include_background),
                          ds_pos.data_ptr(), ds_neg.data_ptr(), B, C, _stream())
    _lib.check(rc, "cp2_pool_bwd")
    return ds_pos, ds_neg


def feat_bwd(dense, inv_norm, mask, g_dense, ds_pos, ds_neg, like: torch.Tensor) -> torch.Tensor:
    """d loss / d feat, laid out like `like` (the encoder output)."""
    lib = _lib.load()
    dfeat = torch.empty_like(like)           # preserves NCHW / channels-last strides
    B, C, P, sn, sc, sp = _feat_strides(dfeat)
    rc = lib.cp2_feat_bwd(_dev(dense, "dense"), _dev(inv_norm, "inv_norm"), _dev(mask, "mask"), _dev(g_dense, "g_dense"),
                          _dev(ds_pos, "ds_pos"), _dev(ds_neg, "ds_neg"), dfeat.data_ptr(), sn, sc, sp, B, C, P, _stream())
    _lib.check(rc, "cp2_feat_bwd")
    return dfeat


def feat_normalize_pool_pair(q_feat: torch.Tensor, k_feat: torch.Tensor, mask_a: torch.Tensor, mask_b: torch.Tensor,
                             k_row: Optional[torch.Tensor] = None):
    """feat_normalize_pool of the query and the key map in one launch; sample n of the key side is read from row
    k_row[n] of k_feat (the un-shuffle index, builder.py:649).  Returns (q_dense, k_dense, q_inv_norm, q_partial, k_partial)."""
    lib = _lib.load()
    for t, name in ((q_feat, "q_feat"), (k_feat, "k_feat")):
        if not t.is_cuda:
            raise _lib.Cp2LibraryError(f"{name} is on {t.device}; cp2_amd ops run on the GPU only")
        if t.dtype != torch.float32:
            raise TypeError(f"{name}: expected float32, got {t.dtype}")
    B, C, P, qsn, qsc, qsp = _feat_strides(q_feat)
    B2, C2, P2, ksn, ksc, ksp = _feat_strides(k_feat)
    if (B2, C2, P2) != (B, C, P):
        raise ValueError("feat_normalize_pool_pair: query and key maps differ in shape")
    dev = q_feat.device
    q_dense = torch.empty((B, C, P), dtype=torch.float32, device=dev)
    k_dense = torch.empty((B, C, P), dtype=torch.float32, device=dev)
    inv = torch.empty((B, P), dtype=torch.float32, device=dev)
    NT = (P + 63) // 64
    q_part = torch.empty((B, NT, 2, C), dtype=torch.float32, device=dev)
    k_part = torch.empty((B, NT, 2, C), dtype=torch.float32, device=dev)
    rc = lib.cp2_feat_normalize_pool_pair(q_feat.data_ptr(), qsn, qsc, qsp, k_feat.data_ptr(), ksn, ksc, ksp,
                                          _opt(k_row, "k_row", torch.int64), _dev(mask_a, "mask_a", torch.float32),
                                          _dev(mask_b, "mask_b", torch.float32), q_dense.data_ptr(), k_dense.data_ptr(),
                                          inv.data_ptr(), q_part.data_ptr(), k_part.data_ptr(), B, C, P, _stream())
    _lib.check(rc, "cp2_feat_normalize_pool_pair")
    return q_dense, k_dense, inv, q_part, k_part


def feat_bwd_fused(dense, inv_norm, mask, g_part: torch.Tensor, S: int, drow_pos, dE, q_pos, q_neg, k_pos, k_neg, q_norms,
                   include_background: bool, like: torch.Tensor) -> torch.Tensor:
    """feat_bwd with the sum of the dense kernel's S split gradients (g_part: [S,B,C,P], or [B,C,P] with S = 1) and the
    pooled-vector backward (pool_bwd) folded in; dE: [B,NE]."""
    lib = _lib.load()
    dfeat = torch.empty_like(like)
    B, C, P, sn, sc, sp = _feat_strides(dfeat)
    rc = lib.cp2_feat_bwd_fused(_dev(dense, "dense"), _dev(inv_norm, "inv_norm"), _dev(mask, "mask"), _dev(g_part, "g_part", torch.float32),
                                int(S), B * C * P, _dev(drow_pos, "drow_pos"), _dev(dE, "dE"), dE.shape[1], _dev(q_pos, "q_pos"),
                                _dev(q_neg, "q_neg"), _dev(k_pos, "k_pos"), _dev(k_neg, "k_neg"), _dev(q_norms, "q_norms"),
                                int(include_background), dfeat.data_ptr(), sn, sc, sp, B, C, P, _stream())
    _lib.check(rc, "cp2_feat_bwd_fused")
    return dfeat


STEP_SCALARS = 24          # CP2_STEP_SCALARS


def step_scalars(ins_loss, cnt_gt, extras, sample_scal, q_pos, k_pos, lmbd_dense: float, dense_pos_q=None, dense_neg_q=None,
                 ins_neg_q=None, lneg_mean=None) -> torch.Tensor:
    """Every scalar the step returns or logs in one launch -> float32 [24] (layout: include/cp2hip.h cp2_step_scalars)."""
    lib = _lib.load()
    B, C = q_pos.shape
    out = torch.empty(STEP_SCALARS, dtype=torch.float32, device=q_pos.device)
    rc = lib.cp2_step_scalars(_dev(ins_loss, "ins_loss", torch.float32), _dev(cnt_gt, "cnt_gt", torch.int32),
                              _dev(extras, "extras", torch.float32), extras.shape[1], _dev(sample_scal, "sample_scal", torch.float32),
                              _dev(q_pos, "q_pos", torch.float32), _dev(k_pos, "k_pos", torch.float32),
                              _opt(dense_pos_q, "dense_pos_q", torch.float32), _opt(dense_neg_q, "dense_neg_q", torch.float32),
                              _opt(ins_neg_q, "ins_neg_q", torch.float32), _opt(lneg_mean, "lneg_mean", torch.float32),
                              float(lmbd_dense), out.data_ptr(), B, C, _stream())
    _lib.check(rc, "cp2_step_scalars")
    return out


def step_tail(ins_loss, cnt_gt, extras, sample_scal, q_pos, k_pos, lmbd_dense: float, dense_pos_q=None, dense_neg_q=None,
              ins_neg_q=None, lneg_mean=None, enqueue=None, iou=None):
    """step_scalars + enqueue + corr_iou_strided as ONE launch (cp2_step_tail).
    enqueue: None or (queue [C,K], keys [n,C], ptr int64[1]); iou: None or (ids_a, ids_b [B,H,W] int64, stride, mask_a,
    mask_b [B,P]).  Returns (scalars [24], iou [B] or None, iou_masked [B] or None)."""
    lib = _lib.load()
    B, C = q_pos.shape
    dev = q_pos.device
    out = torch.empty(STEP_SCALARS, dtype=torch.float32, device=dev)
    queue = keys = ptr = ticket = None
    n_keys = K = 0
    if enqueue is not None:
        queue, keys, ptr = enqueue
        K, (n_keys, c2) = queue.shape[1], keys.shape
        if c2 != C or queue.shape[0] != C or ptr.numel() != 1:
            raise ValueError("step_tail: queue [C,K], keys [n,C] and a one-element pointer expected")
        ticket = _enqueue_ticket(queue)
    ids_a = ids_b = ma = mb = iou_o = ioum_o = None
    H = W = stride = 0
    if iou is not None:
        ids_a, ids_b, stride, ma, mb = iou
        _, H, W = ids_a.shape
        P = ds_size(H, stride) * ds_size(W, stride)
        if ma.shape[1] != P or mb.shape[1] != P:
            raise ValueError(f"step_tail: masks must have {P} elements per sample")
        iou_o = torch.empty(B, dtype=torch.float32, device=dev)
        ioum_o = torch.empty(B, dtype=torch.float32, device=dev)
    rc = lib.cp2_step_tail(_dev(ins_loss, "ins_loss", torch.float32), _dev(cnt_gt, "cnt_gt", torch.int32),
                           _dev(extras, "extras", torch.float32), extras.shape[1], _dev(sample_scal, "sample_scal", torch.float32),
                           _dev(q_pos, "q_pos", torch.float32), _dev(k_pos, "k_pos", torch.float32),
                           _opt(dense_pos_q, "dense_pos_q", torch.float32), _opt(dense_neg_q, "dense_neg_q", torch.float32),
                           _opt(ins_neg_q, "ins_neg_q", torch.float32), _opt(lneg_mean, "lneg_mean", torch.float32),
                           float(lmbd_dense), out.data_ptr(), B, C,
                           _opt(queue, "queue", torch.float32), _opt(keys, "keys", torch.float32), _opt(ptr, "queue_ptr", torch.int64),
                           None if ticket is None else ticket.data_ptr(), n_keys, K,
                           _opt(ids_a, "ids_a", torch.int64), _opt(ids_b, "ids_b", torch.int64), _opt(ma, "mask_a", torch.float32),
                           _opt(mb, "mask_b", torch.float32), _opt(iou_o, "iou"), _opt(ioum_o, "iou_masked"), H, W, int(stride), _stream())
    if rc and ticket is not None:
        ticket.zero_()
    _lib.check(rc, "cp2_step_tail")
    return out, iou_o, ioum_o


def tail_iou_supported(H: int, W: int, stride: int) -> bool:
    """The IoU part of step_tail counts keys in an LDS hash table: at most 4096 down-sampled cells per map."""
    return ds_size(H, stride) * ds_size(W, stride) <= 4096


# ---------------------------------------------------------------- a10 / a16
class RowKeyResult:
    __slots__ = ("loss", "lse", "loss_rows", "cnt_gt", "drows", "dE", "lnegT", "lneg", "ksplit", "pending")


def rowkey_infonce(rows: torch.Tensor, row_layout: Tuple[int, int, int, int], R: int, keys: torch.Tensor,
                   extras: torch.Tensor, temperature: float, grad_scale: Optional[float],
                   drows_like: Optional[torch.Tensor] = None, want_lneg: bool = False,
                   precision: str = "auto", presplit: bool = True, lneg_row_major: bool = False,
                   ksplit: Optional[torch.Tensor] = None, ksplit_ready: bool = False, lneg_out: Optional[torch.Tensor] = None,
                   drows_out: Optional[torch.Tensor] = None, finalize: bool = True) -> RowKeyResult:
    """InfoNCE of R row vectors against the queue `keys` [C,K] with `extras` [R,NE] prepended
    (column 0 = positive).  row_layout = (RP, stride_n, stride_x, stride_c): element (c, r) of
    `rows` lives at (r//RP)*stride_n + (r%RP)*stride_x + c*stride_c.
    grad_scale None -> forward only; else drows (same layout, allocated like `drows_like` or
    `rows`) and dE carry grad_scale * d(sum_r loss_r)/d(rows, extras).
    precision: "f32" (exact fp32 on the f32-input MFMA), "bf16x3" (split-bf16, logits within 3e-5, ~3x faster when
    MFMA-bound; K % 16 == 0, else the f32 kernel serves the call) or "auto" = bf16x3 from 1024 rows up (the DenseCL
    per-pixel case), f32 below.
    For callers that walk the rows in several calls (the chunked DenseCL statistics): `ksplit` = the bf16x3 workspace to
    use (4*C*K bf16; `ksplit_ready`: an earlier call on this stream already filled it for this queue), `lneg_out` = the
    buffer (>= R*K floats) that receives the raw logits, `drows_out` = where the row gradient goes (a view in the rows'
    layout)."""
    if precision not in ("auto", "f32", "bf16x3"):
        raise ValueError(f"precision {precision!r}")
    if precision == "auto":
        prec = 1 if R >= 1024 else 0
    else:
        prec = {"f32": 0, "bf16x3": 1}[precision]
    lib = _lib.load()
    C, K = keys.shape
    RP, sn, sx, sc = row_layout
    NE = extras.shape[1]
    dev = keys.device
    ns = lib.cp2_rowkey_num_splits(R, K)
    _lib.check(min(ns, 0), "cp2_rowkey_num_splits")
    part_m = torch.empty((ns, R), dtype=torch.float32, device=dev)
    part_s = torch.empty((ns, R), dtype=torch.float32, device=dev)
    part_cnt = torch.empty((ns, R), dtype=torch.int32, device=dev)
    want_grad = grad_scale is not None
    part_U = torch.empty((ns, C, R), dtype=torch.float32, device=dev) if want_grad else None
    out = RowKeyResult()
    out.lnegT = out.lneg = None          # raw logits rows.keys: lnegT [K,R] (key-major) or lneg [R,K] (row-major)
    if want_lneg and lneg_out is not None:
        if lneg_out.numel() < R * K or lneg_out.dtype != torch.float32 or not lneg_out.is_contiguous():
            raise ValueError("rowkey_infonce: lneg_out must be a contiguous float32 buffer of at least R*K elements")
        flat = lneg_out.reshape(-1)[:R * K]
        out.lneg, out.lnegT = (flat.view(R, K), None) if lneg_row_major else (None, flat.view(K, R))
    elif want_lneg and lneg_row_major:
        out.lneg = torch.empty((R, K), dtype=torch.float32, device=dev)
    elif want_lneg:
        out.lnegT = torch.empty((K, R), dtype=torch.float32, device=dev)
    lbuf = out.lneg if out.lneg is not None else out.lnegT
    # bf16x3: the queue's hi/lo split in both layouts, written once per call by a prep kernel (4*C*K bf16)
    if not (prec == 1 and R > 64 and K % 16 == 0 and presplit):
        ksplit, ksplit_ready = None, False
    elif ksplit is None:
        ksplit, ksplit_ready = torch.empty(4 * C * K, dtype=torch.bfloat16, device=dev), False
    elif ksplit.numel() < 4 * C * K or ksplit.dtype != torch.bfloat16:
        raise ValueError("rowkey_infonce: ksplit must hold 4*C*K bfloat16")
    if ksplit_ready:
        prec = 3
    out.ksplit = ksplit
    if not rows.is_cuda or rows.dtype != torch.float32:
        raise _lib.Cp2LibraryError("rowkey_infonce: rows must be a float32 GPU tensor")
    _profile("rowkey_fwd" if R <= 32 else "rowkey_fwd_rows")
    rc = lib.cp2_rowkey_infonce_fwd(rows.data_ptr(), RP, sn, sx, sc, R, _dev(keys, "keys", torch.float32), K,
                                    _dev(extras, "extras", torch.float32), NE, float(temperature), ns,
                                    part_m.data_ptr(), part_s.data_ptr(), part_cnt.data_ptr(), _opt(part_U, "part_U"),
                                    _opt(lbuf, "lneg"), int(lneg_row_major), prec, _opt(ksplit, "keys_split"), C, _stream())
    _lib.check(rc, "cp2_rowkey_infonce_fwd")
    out.lse = torch.empty(R, dtype=torch.float32, device=dev)
    out.loss_rows = torch.empty(R, dtype=torch.float32, device=dev)
    out.cnt_gt = torch.empty(R, dtype=torch.int32, device=dev)
    out.loss = torch.empty((), dtype=torch.float32, device=dev)
    out.drows = out.dE = None
    if want_grad:
        out.drows = drows_out if drows_out is not None else torch.empty_like(rows if drows_like is None else drows_like)
        out.dE = torch.empty((R, NE), dtype=torch.float32, device=dev)
    fin = (part_m.data_ptr(), part_s.data_ptr(), part_cnt.data_ptr(), _opt(part_U, "part_U"),
           ns, extras.data_ptr(), NE, float(temperature),
           float(grad_scale if want_grad else 0.0), R, RP, sn, sx, sc, out.lse.data_ptr(),
           out.loss_rows.data_ptr(), out.cnt_gt.data_ptr(),
           None if out.drows is None else out.drows.data_ptr(),
           _opt(out.dE, "dE"), out.loss.data_ptr())
    out.pending = None
    if not finalize and R <= 32 and ns >= 16:
        # finalize=False: the merge of the splits is left to loss_post (one launch with the dense loss's post-pass); the
        # partial buffers stay alive in `pending` until then.  Other shapes have no merged form: finalized here as usual.
        out.pending = (fin, C, (part_m, part_s, part_cnt, part_U, extras))
        return out
    rc = lib.cp2_rowkey_infonce_finalize(*fin, C, _stream())
    _lib.check(rc, "cp2_rowkey_infonce_finalize")
    return out


# ---------------------------------------------------------------- a16: DenseCL positive selection (T18)
class DenseclMatch:
    __slots__ = ("best", "pos", "kvec", "counts")


def _embed_view(t: torch.Tensor, name: str) -> torch.Tensor:
    """A backbone feature map [B,C,h,w] / [B,C,P] in a form cp2_densecl_match reads in place: fp32 with any uniform pixel
    stride, or bf16 with the channel index fastest (channels-last) -- anything else is re-laid once."""
    if not t.is_cuda:
        raise _lib.Cp2LibraryError(f"{name} is on {t.device}; cp2_amd ops run on the GPU only")
    if t.dtype not in (torch.float32, torch.bfloat16):
        raise TypeError(f"{name}: float32 or bfloat16 expected, got {t.dtype}")
    if t.dim() == 4:
        _, _, h, w = t.shape
        uniform = h == 1 or t.stride(2) == w * t.stride(3)
        cl_ok = t.stride(1) == 1 and t.stride(3) % 8 == 0 and t.stride(0) % 8 == 0 and t.data_ptr() % 16 == 0
        if not uniform or (t.dtype == torch.bfloat16 and not cl_ok):
            t = t.contiguous(memory_format=torch.channels_last)
        return t
    if t.dim() != 3:
        raise ValueError(f"{name}: [B,C,h,w] or [B,C,P] expected")
    if t.dtype == torch.bfloat16 and not (t.stride(1) == 1 and t.stride(2) % 8 == 0 and t.stride(0) % 8 == 0 and t.data_ptr() % 16 == 0):
        t = t.transpose(1, 2).contiguous().transpose(1, 2)
    return t


def densecl_match(q_embed: torch.Tensor, k_embed: torch.Tensor, q_local: torch.Tensor, k_local: torch.Tensor,
                  ids_q: Optional[torch.Tensor] = None, ids_k: Optional[torch.Tensor] = None, lmbd_coordinate: float = 0.0,
                  k_row: Optional[torch.Tensor] = None, normalize_k: bool = True, want_kvec: bool = True,
                  want_metrics: bool = False) -> DenseclMatch:
    """Positive of every query pixel of the DenseCL local loss (reference builder.py:818-864) in one launch.
    q_embed / k_embed: backbone features [B,CE,h,w] or [B,CE,P], bf16 (channels-last) or fp32, raw (normalize_k=True) or
    already channel-normalised (normalize_k=False); q_local / k_local: fp32 [B,128,P] unit vectors; ids: int64 [B,P].
    k_row (int64 [B]): sample n's key side is row k_row[n] of k_embed / k_local.  Returns best (int32 [B,P]), pos [B,P],
    kvec [B,128,P] = d pos / d q_local (want_kvec), counts int32 [groups,2] (want_metrics: overlapping pixels, hits)."""
    lib = _lib.load()
    qe, ke = _embed_view(q_embed, "q_embed"), _embed_view(k_embed, "k_embed")
    if qe.dtype != ke.dtype:
        raise TypeError("densecl_match: q_embed and k_embed must have the same dtype")
    B, CE, P, qsn, qsc, qsp = _feat_strides(qe)
    B2, CE2, P2, ksn, ksc, ksp = _feat_strides(ke)
    CL = q_local.shape[1]
    if (B2, CE2, P2) != (B, CE, P) or tuple(q_local.shape) != (B, CL, P) or tuple(k_local.shape) != (B, CL, P):
        raise ValueError("densecl_match: feature shapes do not agree")
    if (ids_q is None) != (ids_k is None):
        raise ValueError("densecl_match: ids_q and ids_k come together")
    if ids_q is not None and (ids_q.numel() != B * P or ids_k.numel() != B * P):
        raise ValueError("densecl_match: ids must hold B*P elements")
    if k_row is not None and (k_row.numel() != B or k_row.dtype != torch.int64):
        raise ValueError("densecl_match: k_row must hold B int64 indices")
    dev = q_local.device
    out = DenseclMatch()
    out.best = torch.empty((B, P), dtype=torch.int32, device=dev)
    out.pos = torch.empty((B, P), dtype=torch.float32, device=dev)
    out.kvec = torch.empty((B, CL, P), dtype=torch.float32, device=dev) if want_kvec else None
    out.counts = torch.empty((B * ((P + 31) // 32), 2), dtype=torch.int32, device=dev) if want_metrics else None
    lm = float(np.float32(lmbd_coordinate))
    one_minus = float(np.float32(1.0 - lmbd_coordinate))      # the reference's python `1 - self.lmbd_coordinate` (builder.py:853)
    _profile("densecl_match")
    rc = lib.cp2_densecl_match(qe.data_ptr(), ke.data_ptr(), int(qe.dtype == torch.bfloat16), qsn, qsc, qsp, ksn, ksc, ksp,
                               _opt(k_row, "k_row", torch.int64), _dev(q_local, "q_local", torch.float32),
                               _dev(k_local, "k_local", torch.float32), _opt(ids_q, "ids_q", torch.int64),
                               _opt(ids_k, "ids_k", torch.int64), lm, one_minus, int(normalize_k), int(want_metrics),
                               out.best.data_ptr(), out.pos.data_ptr(), _opt(out.kvec, "kvec"), _opt(out.counts, "counts"),
                               B, CE, CL, P, _stream())
    _lib.check(rc, "cp2_densecl_match")
    return out


# ---------------------------------------------------------------- a8 / a9
class DenseResult:
    __slots__ = ("lse", "sample_scal", "colmax", "argx", "logits", "pending")

    @property
    def loss(self):          # mean_n loss_n (builder.py:1431-1437)
        return self.sample_scal[:, 2].mean()

    @property
    def acc(self):           # 100 * mean_n arg-max label (builder.py:1442-1448)
        return 100.0 * self.sample_scal[:, 5].mean()


def _ids4(ids):
    if ids is None:
        return None, None, None, None
    pa, pb, ra, rb = ids
    return (_dev(pa, "pixel_ids_a", torch.int64), _dev(pb, "pixel_ids_b", torch.int64),
            _dev(ra, "region_ids_a", torch.int64), _dev(rb, "region_ids_b", torch.int64))


def _negative(negative):
    """negative = None | (scale, centre): centre None (FIXED, builder.py:1332-1338) or a device float[B] tensor
    (AVERAGE / MEDIAN, :1340-1373) -> (mode, scale, centre pointer)."""
    if negative is None:
        return 0, 0.0, None
    scale, centre = negative
    return 1, float(scale), _opt(centre, "negative_center", torch.float32)


def dense_infonce_fwd(q_dense, k_dense, mask_a, mask_b, temperature: float, ids=None,
                      weights=(1.0, 1.0, 1.0), want_logits: bool = False, split: bool = True, negative=None,
                      defer_post: bool = False) -> DenseResult:
    """Per-sample results in sample_scal [B,8] (Sa, Sb, loss_n, mean +score, mean -score, arg-max label): the batch means
    are formed by cp2_step_scalars in the step (`loss` / `acc` below are those means taken with tensor ops, for callers
    outside the step).  defer_post: the fold of the splits and sample_scal are left to loss_post."""
    lib = _lib.load()
    nmode, nscale, ncen = _negative(negative)
    B, C, P = q_dense.shape
    dev = q_dense.device
    f = lambda: torch.empty((B, P), dtype=torch.float32, device=dev)  # noqa: E731
    res = DenseResult()
    res.lse, colsum, possum, allsum, res.colmax = f(), f(), f(), f(), f()
    res.argx = torch.empty((B, P), dtype=torch.int32, device=dev)
    res.sample_scal = torch.empty((B, 8), dtype=torch.float32, device=dev)
    res.logits = torch.empty((B, P, P), dtype=torch.float32, device=dev) if want_logits else None
    pa, pb, ra, rb = _ids4(ids)
    S = lib.cp2_dense_num_splits(B, P) if split else 1
    split_ws = torch.empty(7 * S * B * P, dtype=torch.float32, device=dev) if S > 1 else None
    _profile("dense_fwd")
    rc = lib.cp2_dense_infonce_fwd(_dev(q_dense, "q_dense", torch.float32), _dev(k_dense, "k_dense", torch.float32),
                                   _dev(mask_a, "mask_a", torch.float32), _dev(mask_b, "mask_b", torch.float32),
                                   pa, pb, ra, rb, float(weights[0]), float(weights[1]), float(weights[2]),
                                   float(temperature), res.lse.data_ptr(), colsum.data_ptr(), possum.data_ptr(),
                                   allsum.data_ptr(), res.colmax.data_ptr(), res.argx.data_ptr(),
                                   None if defer_post else res.sample_scal.data_ptr(), _opt(res.logits, "logits"),
                                   split_ws.data_ptr() if split_ws is not None else None, nmode, nscale, ncen, B, C, P, _stream())
    _lib.check(rc, "cp2_dense_infonce_fwd")
    res.pending = None
    if defer_post:
        res.pending = ((_dev(mask_a, "mask_a", torch.float32), _dev(mask_b, "mask_b", torch.float32), res.lse.data_ptr(),
                        colsum.data_ptr(), possum.data_ptr(), allsum.data_ptr(), res.colmax.data_ptr(), res.argx.data_ptr(),
                        res.sample_scal.data_ptr(), split_ws.data_ptr() if split_ws is not None else None, B, C, P),
                       (colsum, possum, allsum, split_ws, mask_a, mask_b))
    return res


def loss_post(ins: RowKeyResult, den: DenseResult, quant_jobs=None, q: Optional[torch.Tensor] = None):
    """Finish a rowkey_infonce(..., finalize=False) and a dense_infonce_fwd(..., defer_post=True) call with ONE launch
    (cp2_loss_post); either argument whose work is not pending is left alone, a single pending one gets its own launch.
    quant_jobs (a masked_quantiles_multi job list): the step's quartile statistics too -- in the SAME launch when both tails
    are pending and every row fits the one-launch quantile form (cp2_step_post), by masked_quantiles_multi otherwise;
    returns their outputs (None without quant_jobs)."""
    lib = _lib.load()
    if (quant_jobs is not None and ins.pending is not None and den.pending is not None
            and all(j["N"] <= QUANTILES_ROW_MAX for j in quant_jobs)):
        if q is None:
            q = _quartile_tensor(quant_jobs[0]["x"].device)
        args, outs = _quant_job_args(quant_jobs, q)
        _profile("quantiles")
        rc = lib.cp2_step_post(*args, *ins.pending[0], *den.pending[0], _stream())
        _lib.check(rc, "cp2_step_post")
        ins.pending = den.pending = None
        return outs
    if ins.pending is not None and den.pending is not None:
        fin = ins.pending[0]
        post = den.pending[0]
        _profile("loss_post")
        rc = lib.cp2_loss_post(*fin, *post, _stream())
        _lib.check(rc, "cp2_loss_post")
    elif ins.pending is not None:
        rc = lib.cp2_rowkey_infonce_finalize(*ins.pending[0], ins.pending[1], _stream())
        _lib.check(rc, "cp2_rowkey_infonce_finalize")
    elif den.pending is not None:
        raise _lib.Cp2LibraryError("loss_post: a deferred dense post-pass needs the instance loss's pending finalize beside it")
    ins.pending = den.pending = None
    return masked_quantiles_multi(quant_jobs, q) if quant_jobs is not None else None


def dense_infonce_bwd(q_dense, k_dense, mask_a, mask_b, temperature: float, fwd: DenseResult, grad_scale: float,
                      ids=None, weights=(1.0, 1.0, 1.0), split: bool = True, negative=None, keep_partials: bool = False):
    """d loss / d q_dense.  keep_partials (the step): return (g_part, S) -- the S split gradients [S,B,C,P] un-summed
    (cp2_feat_bwd_fused adds them, in split order), or the gradient itself with S = 1 when the shape needs no split.
    Otherwise the partials are added here with one tensor op (same order; for callers outside the step)."""
    lib = _lib.load()
    nmode, nscale, ncen = _negative(negative)
    B, C, P = q_dense.shape
    pa, pb, ra, rb = _ids4(ids)
    S = lib.cp2_dense_num_splits(B, P) if split else 1
    split_ws = torch.empty((S, B, C, P), dtype=torch.float32, device=q_dense.device) if S > 1 else None
    g = None if S > 1 else torch.empty_like(q_dense)
    _profile("dense_bwd")
    rc = lib.cp2_dense_infonce_bwd(_dev(q_dense, "q_dense"), _dev(k_dense, "k_dense"), _dev(mask_a, "mask_a"),
                                   _dev(mask_b, "mask_b"), pa, pb, ra, rb, float(weights[0]), float(weights[1]),
                                   float(weights[2]), float(temperature), fwd.lse.data_ptr(), fwd.sample_scal.data_ptr(),
                                   float(grad_scale), _opt(g, "g_dense"), split_ws.data_ptr() if split_ws is not None else None,
                                   nmode, nscale, ncen, B, C, P, _stream())
    _lib.check(rc, "cp2_dense_infonce_bwd")
    if keep_partials:
        return (split_ws, S) if S > 1 else (g, 1)
    if S > 1:                                             # sum_s in split order, as feat_bwd_fused does
        g = split_ws[0].clone()
        for sp in range(1, S):
            g += split_ws[sp]
    return g


# ---------------------------------------------------------------- a15
_QUARTILES = {}


def _quartile_tensor(device) -> torch.Tensor:
    key = str(device)
    if key not in _QUARTILES:
        _QUARTILES[key] = torch.tensor([0.25, 0.5, 0.75], dtype=torch.float32, device=device)
    return _QUARTILES[key]


_QUANT_WS = {}
QUANTILES_ROW_MAX = 131072     # CP2_QUANTILES_ROW_MAX
QUANTILES_CHUNK = 8192         # QCHUNK in csrc/quantile.hip


def _quant_workspace(dev, R, N, NQ):
    """Zeroed workspace of cp2_masked_quantiles(_multi), one per (device, shape signature): the kernels leave it zero,
    so it is allocated and cleared once (a few MB at the training shapes) and re-used by every later step."""
    import ctypes
    key = (dev.index, _stream(), tuple(R), tuple(N), NQ)     # per stream: two streams never share counters
    ws = _QUANT_WS.get(key)
    if ws is None:
        n = len(R)
        I32 = ctypes.c_int * n
        nbytes = _lib.load().cp2_quantiles_workspace_bytes(n, I32(*R), I32(*N), NQ)
        if nbytes <= 0:
            raise ValueError("masked_quantiles: bad job shapes")
        if len(_QUANT_WS) >= 16:                      # tests sweep many shapes: drop the least recently used ONE (the
            _QUANT_WS.pop(next(iter(_QUANT_WS)))      # others may be in flight on a stream; the allocator keeps a freed block
        ws = torch.zeros(nbytes // 4, dtype=torch.int32, device=dev)   # alive until queued work on its stream has run)
    else:
        del _QUANT_WS[key]
    _QUANT_WS[key] = ws                               # re-insert: dict order = recency
    return ws


def masked_quantiles(x: torch.Tensor, stride_row: int, stride_elem: int, R: int, N: int, q: Optional[torch.Tensor] = None,
                     mask_a: Optional[torch.Tensor] = None, mask_b: Optional[torch.Tensor] = None, want: int = -1) -> torch.Tensor:
    """out[j, r] = nanquantile of the kept elements of row r at q[j] (linear interpolation), no sort.
    want = -1 keeps everything; 1 / 0 keep the positive / negative pairs of a [P,P] logit map per row
    (reference tools/correlation_mapping.py:16-53, builder.py:1399-1406)."""
    return masked_quantiles_multi([dict(x=x, stride_row=stride_row, stride_elem=stride_elem, R=R, N=N, mask_a=mask_a,
                                        mask_b=mask_b, want=want)], q)[0]


def masked_quantiles_multi(jobs, q: Optional[torch.Tensor] = None):
    """Several masked_quantiles problems in one call.  jobs: list of dicts with the keyword arguments of
    masked_quantiles (x, stride_row, stride_elem, R, N, mask_a, mask_b, want); returns the list of outputs."""
    lib = _lib.load()
    dev = jobs[0]["x"].device
    if q is None:
        q = _quartile_tensor(dev)
    args, outs = _quant_job_args(jobs, q)
    # rows of at most QUANTILES_ROW_MAX elements: one launch, one workgroup per row, no workspace; longer rows: the chunked
    # six-launch path with its zeroed workspace
    small = all(j["N"] <= QUANTILES_ROW_MAX for j in jobs)
    ws = None if small else _quant_workspace(dev, [j["R"] for j in jobs], [j["N"] for j in jobs], q.numel())
    _profile("quantiles")
    rc = lib.cp2_masked_quantiles_multi(*args, None if ws is None else ws.data_ptr(), 0 if ws is None else ws.numel() * 4, _stream())
    if rc and ws is not None:
        ws.zero_()                                        # a failed call may have left counts behind
    _lib.check(rc, "cp2_masked_quantiles_multi")
    return outs


def _quant_job_args(jobs, q: torch.Tensor):
    """The leading arguments (njobs .. mean_out) of cp2_masked_quantiles_multi / cp2_step_post for a job list, and the output
    tensors they point at."""
    import ctypes
    n = len(jobs)
    dev = jobs[0]["x"].device
    outs = [torch.empty((q.numel(), j["R"]), dtype=torch.float32, device=dev) for j in jobs]
    # job key "mean_out": a float32 [R] tensor that receives the row means (one-launch form, unmasked jobs only)
    means = [_opt(j.get("mean_out"), "mean_out", torch.float32) for j in jobs]
    P_ = ctypes.c_void_p * n
    I64, I32 = ctypes.c_int64 * n, ctypes.c_int * n
    for j in jobs:
        if not j["x"].is_cuda or j["x"].dtype != torch.float32:
            raise _lib.Cp2LibraryError("masked_quantiles: x must be a float32 GPU tensor")
    ma = [_opt(j.get("mask_a"), "mask_a", torch.float32) for j in jobs]
    mb = [_opt(j.get("mask_b"), "mask_b", torch.float32) for j in jobs]
    args = (n, P_(*[j["x"].data_ptr() for j in jobs]), I64(*[j["stride_row"] for j in jobs]), I64(*[j["stride_elem"] for j in jobs]),
            I32(*[j["R"] for j in jobs]), I32(*[j["N"] for j in jobs]), P_(*ma), P_(*mb),
            I32(*[(j["mask_a"].shape[1] if j.get("mask_a") is not None else 0) for j in jobs]),
            I32(*[j.get("want", -1) for j in jobs]), _dev(q, "q", torch.float32), q.numel(), P_(*[o.data_ptr() for o in outs]),
            P_(*means) if any(m is not None for m in means) else None)
    return args, outs


# ---------------------------------------------------------------- encoder fast path: fused BatchNorm
def bn_supported(x: torch.Tensor) -> bool:
    """channels-last bf16 GPU activations with C a multiple of 64."""
    if not (x.is_cuda and x.dtype == torch.bfloat16 and x.dim() == 4):
        return False
    C = x.shape[1]
    return C % 64 == 0 and C <= 8192 and x.is_contiguous(memory_format=torch.channels_last)


def _bn_partials(M: int, C: int) -> int:
    """Same geometry as bn_geom_stats() in csrc/bn.hip (checked against cp2_bn_num_partials in tests/test_cabi.py)."""
    cg = C // 8
    cgs = 16 if cg % 16 == 0 else 8
    ns, rp = cg // cgs, 256 // cgs
    gr = max(1, min(1024 // ns, M // (4 * rp), 256))
    rpb = -(-(-(-M // gr)) // rp) * rp
    return -(-M // rpb)


def bn_fwd(x, residual, weight, bias, running_mean, running_var, momentum: float, eps: float, relu: bool):
    lib = _lib.load()
    N, C, H, W = x.shape
    M = N * H * W
    G = _bn_partials(M, C)
    stream = _stream()
    y = torch.empty_like(x)                               # channels-last like x
    ws = torch.empty((2 * G + 4, C), dtype=torch.float32, device=x.device)   # partials | scale, shift | mean, invstd
    base, row = ws.data_ptr(), 4 * C
    rc = lib.cp2_bn_fwd(x.data_ptr(), residual.data_ptr() if residual is not None else None,
                        weight.data_ptr(), bias.data_ptr(), running_mean.data_ptr(), running_var.data_ptr(),
                        momentum, eps, int(relu), y.data_ptr(), base + (2 * G + 2) * row, base + (2 * G + 3) * row,
                        base, base + 2 * G * row, M, C, stream)
    if rc:
        _lib.check(rc, "cp2_bn_fwd")
    return y, ws[2 * G + 2:]


def bn_bwd(x, dy, y, weight, stats, relu: bool, want_dres: bool):
    lib = _lib.load()
    N, C, H, W = x.shape
    M = N * H * W
    G = _bn_partials(M, C)
    stream = _stream()
    dx = torch.empty_like(x)
    dres = torch.empty_like(x) if (want_dres and relu) else None
    ws = torch.empty((2 * G + 5, C), dtype=torch.float32, device=x.device)   # partials | coef[3] | dgamma, dbeta
    base, row = ws.data_ptr(), 4 * C
    sp = stats.data_ptr()
    rc = lib.cp2_bn_bwd(x.data_ptr(), dy.data_ptr(), y.data_ptr() if relu else None, weight.data_ptr(), sp, sp + row,
                        int(relu), dx.data_ptr(), dres.data_ptr() if dres is not None else None,
                        base + (2 * G + 3) * row, base + (2 * G + 4) * row, base, base + 2 * G * row, M, C, stream)
    if rc:
        _lib.check(rc, "cp2_bn_bwd")
    if want_dres and not relu:
        dres = dy
    return dx, dres, ws[2 * G + 3], ws[2 * G + 4]


# ---------------------------------------------------------------- encoder fast path: 1x1 convolution weight gradient
def wgrad1x1_supported(co: int, ci: int) -> bool:
    return co % 64 == 0 and ci % 64 == 0


def wgrad1x1(dy: torch.Tensor, x: torch.Tensor) -> torch.Tensor:
    """dW [CO, CI, 1, 1] fp32 = sum over (n, h, w) of dy[n, :, h, w] (x) x[n, :, h, w] for channels-last bf16 NCHW
    tensors (csrc/wgrad.hip)."""
    lib = _lib.load()
    N, CO, H, W = dy.shape
    CI = x.shape[1]
    M = N * H * W
    if not (dy.is_contiguous(memory_format=torch.channels_last) and x.is_contiguous(memory_format=torch.channels_last)):
        raise _lib.Cp2LibraryError("wgrad1x1: channels-last tensors expected")
    S = lib.cp2_wgrad1x1_num_splits(M, CO, CI)
    if S < 0:
        _lib.check(S, "cp2_wgrad1x1_num_splits")
    dw = torch.empty((CO, CI, 1, 1), dtype=torch.float32, device=dy.device)
    part = torch.empty(S * CO * CI, dtype=torch.float32, device=dy.device) if S > 1 else dw
    if not (dy.is_cuda and x.is_cuda and dy.dtype == torch.bfloat16 and x.dtype == torch.bfloat16 and x.shape[0] == N
            and x.shape[2:] == dy.shape[2:]):
        raise _lib.Cp2LibraryError("wgrad1x1: bf16 GPU tensors of matching batch / spatial size expected")
    rc = lib.cp2_wgrad1x1(dy.data_ptr(), x.data_ptr(), dw.data_ptr(), part.data_ptr(), M, CO, CI, _stream())
    _lib.check(rc, "cp2_wgrad1x1")
    return dw


def wgrad_conv(dy: torch.Tensor, x: torch.Tensor, kernel: int, stride: int, pad: int, dil: int) -> torch.Tensor:
    """dW [CO, CI, k, k] fp32 in channels-last strides (memory order [CO][kh][kw][CI]) of a k x k convolution on
    channels-last bf16 tensors: dy [N, CO, OH, OW], x [N, CI, H, W] (csrc/wgrad.hip, deterministic split sums)."""
    lib = _lib.load()
    N, CO, OH, OW = dy.shape
    _, CI, H, W = x.shape
    if not (dy.is_cuda and x.is_cuda and dy.dtype == torch.bfloat16 and x.dtype == torch.bfloat16 and x.shape[0] == N
            and dy.is_contiguous(memory_format=torch.channels_last) and x.is_contiguous(memory_format=torch.channels_last)):
        raise _lib.Cp2LibraryError("wgrad_conv: channels-last bf16 GPU tensors of matching batch size expected")
    S = lib.cp2_wgrad_conv_num_splits(N, OH, OW, CO, CI, kernel, kernel)
    if S < 0:
        _lib.check(S, "cp2_wgrad_conv_num_splits")
    dw = torch.empty((CO, CI, kernel, kernel), dtype=torch.float32, device=dy.device, memory_format=torch.channels_last)
    part = torch.empty(S * dw.numel(), dtype=torch.float32, device=dy.device) if S > 1 else dw
    rc = lib.cp2_wgrad_conv(dy.data_ptr(), x.data_ptr(), dw.data_ptr(), part.data_ptr(), N, H, W, OH, OW, CO, CI, kernel, kernel,
                            stride, pad, dil, _stream())
    _lib.check(rc, "cp2_wgrad_conv")
    return dw


# ---------------------------------------------------------------- encoder fast path: stem max-pool
def maxpool3s2_supported(x: torch.Tensor) -> bool:
    return (x.is_cuda and x.dtype == torch.bfloat16 and x.dim() == 4 and x.shape[1] % 8 == 0
            and x.is_contiguous(memory_format=torch.channels_last))


def maxpool3s2_fwd(x: torch.Tensor):
    """MaxPool2d(3, 2, 1) of a channels-last bf16 activation -> (y channels-last, idx uint8 position codes)."""
    N, C, H, W = x.shape
    OH, OW = (H - 1) // 2 + 1, (W - 1) // 2 + 1
    y = torch.empty((N, C, OH, OW), dtype=x.dtype, device=x.device, memory_format=torch.channels_last)
    idx = torch.empty(N * OH * OW * C, dtype=torch.uint8, device=x.device)
    rc = _lib.load().cp2_maxpool3s2_fwd(x.data_ptr(), y.data_ptr(), idx.data_ptr(), N, H, W, C, _stream())
    if rc:
        _lib.check(rc, "cp2_maxpool3s2_fwd")
    return y, idx


def maxpool3s2_bwd(dy: torch.Tensor, idx: torch.Tensor, shape) -> torch.Tensor:
    N, C, H, W = shape
    dx = torch.empty((N, C, H, W), dtype=dy.dtype, device=dy.device, memory_format=torch.channels_last)
    rc = _lib.load().cp2_maxpool3s2_bwd(dy.data_ptr(), idx.data_ptr(), dx.data_ptr(), N, H, W, C, _stream())
    if rc:
        _lib.check(rc, "cp2_maxpool3s2_bwd")
    return dx


# ---------------------------------------------------------------- f4: supervised CutPaste / mirror pre-training
CUTPASTE_PARAMS = 20      # CP2_CUTPASTE_PARAMS
MIRROR_MAX_CLASSES = 8    # CP2_MIRROR_MAX_CLASSES


def cutpaste(src: torch.Tensor, src_mirror: Optional[torch.Tensor], params: torch.Tensor, mask: Optional[torch.Tensor] = None,
             want_u8: bool = False, want_f32: bool = True):
    """One patch round of CutPasteDataset.cutpaste for a batch (reference datasets/pretrain_dataset.py:273-352).
    src / src_mirror: uint8 [*,H,W,3] on the GPU (src_mirror None = MirrorVariant.NONE); params: int32 [B,20] device
    table (mirror.cutpaste_table); mask: None (first round: written) or the int64 [B,H,W] mask of the previous round
    (updated in place with logical_or).  Returns dict(u8, mirror_u8, f32, mirror_f32, mask)."""
    lib = _lib.load()
    if not src.is_cuda or src.dtype != torch.uint8 or src.dim() != 4 or src.shape[3] != 3:
        raise _lib.Cp2LibraryError("cutpaste: src must be a uint8 [*,H,W,3] GPU tensor")
    _, H, W, _ = src.shape
    if src_mirror is not None and (src_mirror.dtype != torch.uint8 or tuple(src_mirror.shape[1:]) != (H, W, 3)):
        raise ValueError("cutpaste: src_mirror must be uint8 [*,H,W,3] of the same image size")
    if params.dim() != 2 or params.shape[1] != CUTPASTE_PARAMS:
        raise ValueError(f"cutpaste: params must be int32 [B,{CUTPASTE_PARAMS}]")
    B = params.shape[0]
    dev = src.device
    mask_or = mask is not None
    if mask is None:
        mask = torch.empty((B, H, W), dtype=torch.int64, device=dev)
    elif tuple(mask.shape) != (B, H, W):
        raise ValueError("cutpaste: mask must be int64 [B,H,W]")
    mir = src_mirror is not None
    u8 = torch.empty((B, H, W, 3), dtype=torch.uint8, device=dev) if want_u8 else None
    mu8 = torch.empty((B, H, W, 3), dtype=torch.uint8, device=dev) if want_u8 and mir else None
    f32 = torch.empty((B, 3, H, W), dtype=torch.float32, device=dev) if want_f32 else None
    mf32 = torch.empty((B, 3, H, W), dtype=torch.float32, device=dev) if want_f32 and mir else None
    rc = lib.cp2_cutpaste(_dev(src, "src"), _opt(src_mirror, "src_mirror"), _dev(params, "params", torch.int32), _opt(u8, "u8"),
                          _opt(mu8, "mu8"), _opt(f32, "f32"), _opt(mf32, "mf32"), _dev(mask, "mask", torch.int64),
                          int(mask_or), B, H, W, _stream())
    _lib.check(rc, "cp2_cutpaste")
    return dict(u8=u8, mirror_u8=mu8, f32=f32, mirror_f32=mf32, mask=mask)


def mirror_loss(s_logits: torch.Tensor, t_logits: Optional[torch.Tensor], masks: torch.Tensor, softmax_temp: float,
                lmbd_compare_loss: float, want_grad: bool = True, want_argmax: bool = True,
                confusion: Optional[torch.Tensor] = None):
    """MirrorModule.shared_step's loss section in one pass (reference networks/mirror_network.py:40-63).
    s_logits / t_logits: float32 [N,C,H,W] at image size (t_logits None = MirrorVariant.NONE); masks int64 [N,H,W].
    Returns (out3 = [loss, class_loss, compare_loss], grad_s, grad_t, argmax [(2)N,H,W]); confusion (int64 [C,C],
    row = ground truth) is added to in place when given."""
    lib = _lib.load()
    N, C, H, W = s_logits.shape
    HW = H * W
    dev = s_logits.device
    if t_logits is not None and t_logits.shape != s_logits.shape:
        raise ValueError("mirror_loss: s_logits and t_logits must have the same shape")
    if tuple(masks.shape) != (N, H, W):
        raise ValueError(f"mirror_loss: masks must be [N,H,W] = {(N, H, W)}, got {tuple(masks.shape)}")
    if confusion is not None and tuple(confusion.shape) != (C, C):
        raise ValueError("mirror_loss: confusion must be int64 [C,C]")
    two = t_logits is not None
    gs = torch.empty_like(s_logits) if want_grad else None
    gt = torch.empty_like(s_logits) if want_grad and two else None
    am = torch.empty(((2 if two else 1) * N, H, W), dtype=torch.int64, device=dev) if want_argmax else None
    nparts = lib.cp2_mirror_loss_num_partials(N, HW)
    part = torch.empty(2 * nparts, dtype=torch.float64, device=dev)
    out3 = torch.empty(3, dtype=torch.float32, device=dev)
    _profile("mirror_loss")
    rc = lib.cp2_mirror_loss(_dev(s_logits, "s_logits", torch.float32), _opt(t_logits, "t_logits", torch.float32),
                             _dev(masks, "masks", torch.int64), float(softmax_temp), float(lmbd_compare_loss), _opt(gs, "gs"),
                             _opt(gt, "gt"), _opt(am, "argmax"), _opt(confusion, "confusion", torch.int64), part.data_ptr(),
                             out3.data_ptr(), N, C, HW, _stream())
    _lib.check(rc, "cp2_mirror_loss")
    return out3, gs, gt, am
