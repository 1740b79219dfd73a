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


def _stream() -> int:
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
def enqueue(queue: torch.Tensor, keys: torch.Tensor, ptr: torch.Tensor) -> None:
    """In place: queue[:, (ptr+i) % K] = keys[i]; ptr = (ptr + n) % K, all on the device
    (reference builder.py:569-587; keys are already gathered over ranks)."""
    lib = _lib.load()
    C, K = queue.shape
    n, c2 = keys.shape
    if c2 != C:
        raise ValueError(f"enqueue: keys have {c2} channels, queue has {C}")
    if ptr.numel() != 1:
        raise ValueError("enqueue: ptr must hold one int64")
    rc = lib.cp2_enqueue(_dev(queue, "queue", torch.float32), _dev(keys, "keys", torch.float32),
                         _dev(ptr, "queue_ptr", torch.int64), n, C, K, _stream())
    _lib.check(rc, "cp2_enqueue")
