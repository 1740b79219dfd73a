"""Collectives of the hot path (reference builder.py:609-649,1710-1722) on torch.distributed;
backend "nccl" is RCCL over xGMI on ROCm, "gloo" on CPU for tests.  Pure plumbing: no kernels here."""
from __future__ import annotations

from typing import Optional, Tuple

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
    if dist.get_backend() == "gloo" and tensor.is_cuda:      # rehearsal only: gloo has no fused all-gather for GPU tensors
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
