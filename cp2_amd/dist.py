"""Collectives of the hot path (reference builder.py:609-649,1710-1722) on torch.distributed;
backend "nccl" is RCCL over xGMI on ROCm, "gloo" on CPU for tests.  Pure plumbing: no kernels here.

Shuffle-BN exchange.  The reference all-gathers every rank's whole image batch and keeps one slice of the permuted
concatenation (C1: W x b x 3 x H x W floats arrive at every rank, 154 MB at W = 8, b = 32, 224^2), then all-gathers the
keys and keeps one slice again (C3).  `ShufflePlan` + `exchange_rows` move only the rows a rank keeps: with the same
permutation known on every rank, rank s sends rank r exactly the rows of idx_shuffle[r*b:(r+1)*b] that it owns
(one all_to_all_single with per-peer row counts: (W-1)/W x b rows arrive instead of (W-1) x b -- 16.9 MB instead of
134.9 MB per rank and step for C1 at W = 8).  The rows a rank ends up with are the reference's, bit for bit, in the
reference's order (tests/test_dist_gloo.py asserts equality with the all-gather form).
"""
from __future__ import annotations

from typing import Callable, List, Optional, Tuple

import torch
import torch.distributed as dist


def is_dist() -> bool:
    return dist.is_available() and dist.is_initialized()


def world_size() -> int:
    return dist.get_world_size() if is_dist() else 1


# Rehearsal switch (tests/test_gpu_nccl.py): run every collective and side-stream branch even with a single rank, so
# the RCCL calls of the multi-GPU run are exercised on a one-GPU box.  Never set in production.
FORCE_COLLECTIVES = False


def multi() -> bool:
    """True when the collectives of the hot path have to run (more than one rank, or the rehearsal switch)."""
    return is_dist() and (world_size() > 1 or FORCE_COLLECTIVES)


def rank() -> int:
    return dist.get_rank() if is_dist() else 0


def _gloo_on_gpu(t: torch.Tensor) -> bool:
    """Rehearsal only (two ranks sharing one GPU over gloo): gloo has no fused GPU collectives."""
    return dist.get_backend() == "gloo" and t.is_cuda


@torch.no_grad()
def concat_all_gather(tensor: torch.Tensor) -> torch.Tensor:
    """All ranks' tensors concatenated on dim 0, in rank order; no gradient
    (reference builder.py:1710-1722).  One all_gather_into_tensor into a single
    pre-sized buffer instead of W temporaries plus a cat."""
    if not multi():
        return tensor
    tensor = tensor.contiguous()
    out = torch.empty((world_size() * tensor.shape[0],) + tuple(tensor.shape[1:]), dtype=tensor.dtype,
                      device=tensor.device)
    if _gloo_on_gpu(tensor):
        dist.all_gather(list(out.chunk(world_size(), dim=0)), tensor)
    else:
        dist.all_gather_into_tensor(out, tensor)
    return out


@torch.no_grad()
def make_shuffle_index(batch_all: int, device, generator: Optional[torch.Generator] = None) -> torch.Tensor:
    """Shuffle-BN permutation: drawn on the host from the global torch RNG (or `generator`)
    and overwritten with rank 0's by a broadcast, exactly as reference builder.py:618-621."""
    idx = torch.randperm(batch_all, generator=generator).to(device)
    if multi():
        dist.broadcast(idx, src=0)
    return idx


def shuffle_rows_for_rank(idx_shuffle: torch.Tensor, r: int, w: int) -> torch.Tensor:
    """Global row indices rank r feeds to its key encoder (builder.py:627-630)."""
    return idx_shuffle.view(w, -1)[r]


def unshuffle_rows_for_rank(idx_shuffle: torch.Tensor, r: int, w: int) -> Tuple[torch.Tensor, torch.Tensor]:
    """(idx_unshuffle, the rows of the gathered key batch that restore rank r's order) (builder.py:624,647-649)."""
    idx_unshuffle = torch.argsort(idx_shuffle)
    return idx_unshuffle, idx_unshuffle.view(w, -1)[r]


# ---------------------------------------------------------------- the permutation, known on every host
_SHARED_GEN: Optional[torch.Generator] = None


def shared_generator(device=None) -> torch.Generator:
    """A host generator with the same state on every rank: rank 0 draws its seed from the global torch RNG (the
    stream the reference's randperm consumes, builder.py:618) and broadcasts it ONCE.  Afterwards every rank draws
    each step's permutation itself -- the per-step index broadcast (C2) disappears and the permutation is known on the
    host, which is what lets the exchange be planned without a device-to-host copy."""
    global _SHARED_GEN
    if _SHARED_GEN is None:
        seed = torch.randint(0, 2 ** 62, (1,), dtype=torch.int64)
        if multi():
            on_dev = dist.get_backend() != "gloo" and device is not None
            t = seed.to(device) if on_dev else seed
            dist.broadcast(t, src=0)
            seed = t.cpu()
        _SHARED_GEN = torch.Generator().manual_seed(int(seed))
    return _SHARED_GEN


def reset_shared_generator() -> None:
    global _SHARED_GEN
    _SHARED_GEN = None


def shared_permutation(batch_all: int, device=None) -> torch.Tensor:
    """This step's shuffle-BN permutation as a HOST int64 tensor, identical on every rank."""
    return torch.randperm(batch_all, generator=shared_generator(device))


class ShufflePlan:
    """Who sends which rows to whom for one step's shuffle-BN exchange, derived on the host from the permutation.

    Global row g lives on rank g // b as local row g % b.  Rank r's key encoder takes the global rows
    idx_shuffle[r*b : (r+1)*b], in that order (reference builder.py:627-630).

    forward (images, C1):   send  x[send_rows]  split by destination as send_counts;
                            the received rows arrive grouped by source rank; out[j] = recv[place[j]].
    backward (keys, C3):    the same pairs in the other direction: send k[back_rows] split as recv_counts,
                            receive split as send_counts, out[i] = recv[back_place[i]].
    """

    def __init__(self, idx_shuffle_host: torch.Tensor, r: int, w: int):
        idx = idx_shuffle_host.to("cpu", torch.int64).reshape(-1)
        n = idx.numel()
        if n % w:
            raise ValueError(f"ShufflePlan: {n} rows do not split over {w} ranks")
        b = n // w
        self.b, self.rank, self.world = b, r, w
        self.idx_shuffle = idx
        owner = idx // b                                          # owner[p]: the rank holding the row wanted at slot p
        # what this rank sends: slots p (ascending = destination-major, then position) whose row it owns
        slots = (owner == r).nonzero().reshape(-1)
        self.send_rows = idx[slots] % b
        self.send_counts: List[int] = torch.bincount(slots // b, minlength=w).tolist()
        # what this rank receives: its own slots, grouped by source rank (stable: position order inside a group)
        src = owner[r * b:(r + 1) * b]
        self.back_rows = torch.argsort(src, stable=True)          # receive-buffer order -> slot j
        self.recv_counts: List[int] = torch.bincount(src, minlength=w).tolist()
        self.place = torch.argsort(self.back_rows)                # slot j -> position in the receive buffer
        self.back_place = torch.argsort(self.send_rows)           # local row i -> position in the returned buffer
        self._dev = None

    def device_tables(self, device):
        """(send_rows, place, back_rows, back_place) as int64 device tensors: one small host-to-device copy."""
        if self._dev is None or self._dev[0].device != torch.device(device):
            packed = torch.stack([self.send_rows, self.place, self.back_rows, self.back_place])
            if torch.device(device).type == "cuda":
                # page-locked source + non_blocking: a blocking copy would make the host wait for everything queued on the
                # stream (the previous step's backward pass and update) before it can enqueue this step
                packed = packed.pin_memory().to(device, non_blocking=True)
            else:
                packed = packed.to(device)
            self._dev = tuple(packed[i] for i in range(4))
        return self._dev

    def bytes_received(self, row_bytes: int) -> int:
        """Bytes arriving from OTHER ranks per exchange (forward or backward: the pair set is the same)."""
        return (self.b - self.recv_counts[self.rank]) * row_bytes


def _index_rows(x: torch.Tensor, idx: torch.Tensor) -> torch.Tensor:
    return x[idx]


@torch.no_grad()
def _all_to_all_rows(x: torch.Tensor, n_out: int, out_counts: List[int], in_counts: List[int]) -> torch.Tensor:
    out = torch.empty((n_out,) + tuple(x.shape[1:]), dtype=x.dtype, device=x.device)
    if _gloo_on_gpu(x):
        host = torch.empty(out.shape, dtype=x.dtype)
        dist.all_to_all_single(host, x.cpu(), out_counts, in_counts)
        out.copy_(host)
    else:
        dist.all_to_all_single(out, x.contiguous(), out_counts, in_counts)
    return out


@torch.no_grad()
def exchange_rows(x: torch.Tensor, plan: ShufflePlan, backward: bool = False,
                  take: Callable[[torch.Tensor, torch.Tensor], torch.Tensor] = _index_rows, presorted: bool = False,
                  keep_order: bool = False) -> torch.Tensor:
    """Shuffle (backward=False: this rank's b local rows -> the b rows its key encoder takes) or un-shuffle
    (backward=True: the encoder's b outputs -> this rank's own rows in their original order).  `take(x, idx)` is the
    local row gather (ops.gather_rows on the GPU; plain indexing in the CPU tests).
    presorted (forward): x is already x[send_rows] (the composition kernel wrote its rows in send order).
    keep_order (backward): return the received rows as they arrive; the caller reads row back_place[i] for sample i
    (the loss section's row index) instead of paying a gather."""
    send_rows, place, back_rows, back_place = plan.device_tables(x.device)
    if not backward:
        recv = _all_to_all_rows(x if presorted else take(x, send_rows), plan.b, plan.recv_counts, plan.send_counts)
        return take(recv, place)
    recv = _all_to_all_rows(take(x, back_rows), plan.b, plan.send_counts, plan.recv_counts)
    return recv if keep_order else take(recv, back_place)
