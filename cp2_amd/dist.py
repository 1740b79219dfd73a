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

import collections
import datetime
import os
import sys
import threading
import time
from typing import Callable, List, Optional, Tuple

import torch
import torch.distributed as dist

DEFAULT_TIMEOUT_S = 120.0      # process-group timeout (torch's default for nccl is 10 min = the whole budget of a bench run)


def is_dist() -> bool:
    return dist.is_available() and dist.is_initialized()


def world_size() -> int:
    return dist.get_world_size() if is_dist() else 1


def multi() -> bool:
    """True when the collectives of the hot path run: whenever a process group exists.  One-rank groups included -- the
    exchange steps then move nothing but are issued all the same, which is how bench.py --rehearse-collectives and
    tests/test_gpu_nccl.py exercise the multi-GPU code path on a one-GPU box; cp2_amd.main creates a group only for
    world_size > 1, so single-GPU training never pays for them."""
    return is_dist()


def init_process_group(backend: str, rank: int, world: int, init_method: Optional[str] = None,
                       timeout_s: float = DEFAULT_TIMEOUT_S, watchdog: bool = True,
                       startup_timeout_s: Optional[float] = None) -> None:
    """The ONE place the package creates its process group (cp2_amd.main, bench.py, the multi-process tests):
      * HSA_ENABLE_IPC_MODE_LEGACY=0 -- the dmabuf IPC path; with the legacy mode RCCL's peer-memory set-up fails on this
        driver stack (hipIpcGetMemHandle: invalid argument), so it must be in the environment before the first HIP call of
        a multi-process run, not only in bench.py;
      * a bounded timeout (torch's nccl default is 10 minutes): a collective that never completes ends the process;
      * the hang watchdog below, which says WHICH exchange step did not complete before that happens.
    Two limits: until the caller declares the run steady (`steady()`, after the warm-up steps) a collective may stay pending
    for `startup_timeout_s` (default max(480 s, 4 x timeout_s)): the first steps run MIOpen's solver search, which takes
    15 s on an idle one-GPU box and several times that when W ranks share a host and its solver database, and it does not
    take the same time on every rank -- a peer that is still searching is not a hang (found by a 4-ranks-on-one-GPU rehearsal
    of bench.py that the 120 s limit killed during its first step).  From `steady()` on the limit is `timeout_s`."""
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    startup = max(480.0, 4.0 * timeout_s) if startup_timeout_s is None else float(startup_timeout_s)
    startup = max(startup, timeout_s)
    kw = dict(backend=backend, rank=rank, world_size=world, timeout=datetime.timedelta(seconds=startup))
    if init_method:
        kw["init_method"] = init_method
    dist.init_process_group(**kw)
    COLLECTIVES.clear()
    _LIMITS[:] = [0.8 * startup, 0.8 * timeout_s, float(timeout_s), False]
    if watchdog and world > 1:
        start_watchdog(0.8 * timeout_s, startup_s=0.8 * startup)


def steady() -> None:
    """The warm-up is over (every kernel of the step has run once on every rank): from now on a collective that stays
    pending longer than the steady-state limit is a hang.  Also lowers the process group's own timeout to that limit."""
    if _LIMITS[3] or not is_dist():
        return
    _LIMITS[3] = True
    try:
        from torch.distributed.distributed_c10d import _set_pg_timeout
        _set_pg_timeout(datetime.timedelta(seconds=_LIMITS[2]))
    except Exception:                               # noqa: BLE001 -- an older torch / a backend without it: the start-up limit stays
        pass


# ---------------------------------------------------------------- which collective hangs?
class _CollectiveLog:
    """The last exchange steps this rank enqueued: (step, name, work handle or None).  With RCCL a hung collective shows
    up minutes later as an abort from torch's watchdog thread that names an opcode and a sequence number; this log turns it
    into "C3 key un-shuffle of step 17" (reference call sites: builder.py:612, :640, :572; main.py:456-461)."""

    def __init__(self, keep: int = 64):
        self.items = collections.deque(maxlen=keep)
        self.step = 0

    def clear(self):
        self.items.clear()
        self.step = 0

    def note(self, name: str, work=None):
        self.items.append((self.step, name, work, time.monotonic()))
        return work

    def first_incomplete(self):
        """(step, name, seconds since it was enqueued) of the oldest tracked collective that has not completed, or None."""
        for step, name, work, t0 in list(self.items):
            try:
                if work is not None and not work.is_completed():
                    return step, name, time.monotonic() - t0
            except Exception:                       # noqa: BLE001 -- a backend without is_completed()
                continue
        return None

    def describe(self) -> str:
        inc = self.first_incomplete()
        last = self.items[-1] if self.items else None
        if inc is not None:
            return f"first collective not completed: {inc[1]} (step {inc[0]}), pending for {inc[2]:.0f} s"
        if last is not None:
            return f"every tracked collective completed; last one enqueued: {last[1]} (step {last[0]})"
        return "no collective was enqueued yet"


COLLECTIVES = _CollectiveLog()
_PROGRESS = [time.monotonic(), 0]
_LIMITS = [0.8 * 480.0, 0.8 * DEFAULT_TIMEOUT_S, DEFAULT_TIMEOUT_S, False]   # watchdog: start-up, steady; pg steady timeout; steady?
_WATCHDOG: Optional[threading.Thread] = None
HANG_EXIT_CODE = 3


def progress(step: Optional[int] = None) -> None:
    """Called by the training / bench loop once per step (host side): feeds the hang watchdog."""
    _PROGRESS[0] = time.monotonic()
    if step is not None:
        _PROGRESS[1] = COLLECTIVES.step = step


def start_watchdog(timeout_s: float, poll_s: float = 2.0, on_hang: Optional[Callable[[str], None]] = None,
                   startup_s: Optional[float] = None) -> None:
    """A daemon thread: a tracked collective that stays PENDING longer than the limit -> print which exchange step it is and
    end the process with HANG_EXIT_CODE (the host thread may be blocked inside a HIP call by then; a blocked call releases
    the GIL).  The limit is `startup_s` (default: `timeout_s`) until steady() is called, `timeout_s` afterwards; both run out
    before the process group's own timeout so that the named message comes first.  A process that is merely busy or idle on
    its host (solver search, a checkpoint, data loading, an epoch's validation) has no pending collective and is left alone."""
    global _WATCHDOG
    if _WATCHDOG is not None:
        return
    progress()
    _LIMITS[0], _LIMITS[1] = (timeout_s if startup_s is None else startup_s), timeout_s

    def run():
        while True:
            time.sleep(poll_s)
            inc = COLLECTIVES.first_incomplete()
            limit = _LIMITS[1] if _LIMITS[3] else _LIMITS[0]
            if inc is not None and inc[2] > limit:
                idle = time.monotonic() - _PROGRESS[0]
                msg = (f"cp2_amd: rank {rank()} of {world_size()} made no progress for {idle:.0f} s at step {_PROGRESS[1]}; "
                       + COLLECTIVES.describe())
                if on_hang is not None:
                    on_hang(msg)
                    return
                print(msg, file=sys.stderr, flush=True)
                os._exit(HANG_EXIT_CODE)

    _WATCHDOG = threading.Thread(target=run, name="cp2-hang-watchdog", daemon=True)
    _WATCHDOG.start()


def tracked(name: str, work):
    """Register an asynchronous collective with the hang log and wait for it: `tracked("...", dist.broadcast(t, 0, async_op=True))`
    is a blocking call whose name the watchdog can print."""
    COLLECTIVES.note(name, work)
    if work is not None:
        work.wait()
    return work


def barrier(name: str = "barrier") -> None:
    if is_dist():
        tracked(name, dist.barrier(async_op=True))


def assert_same_on_all_ranks(what: str, value: int, device=None) -> None:
    """Every rank must hold the same `value` (a hash of something all ranks derive independently, e.g. FlatDDP's bucket
    table): one small all-gather, RuntimeError naming the ranks that differ."""
    if not multi():
        return
    on_dev = dist.get_backend() != "gloo" and device is not None
    mine = torch.tensor([value & 0x7FFFFFFFFFFFFFFF], dtype=torch.int64, device=device if on_dev else "cpu")
    out = [torch.empty_like(mine) for _ in range(world_size())]
    tracked(f"agreement check: {what}", dist.all_gather(out, mine, async_op=True))
    vals = [int(t) for t in out]
    if len(set(vals)) != 1:
        odd = [r for r, v in enumerate(vals) if v != vals[0]]
        raise RuntimeError(f"{what} differs between ranks: ranks {odd} disagree with rank 0 ({vals})")


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
        w = dist.all_gather_into_tensor(out, tensor, async_op=True)
        COLLECTIVES.note(f"all_gather {tuple(tensor.shape)}", w)
        w.wait()                    # orders the current stream behind the collective; the host does not block (RCCL)
    return out


@torch.no_grad()
def make_shuffle_index(batch_all: int, device, generator: Optional[torch.Generator] = None) -> torch.Tensor:
    """Shuffle-BN permutation: drawn on the host from the global torch RNG (or `generator`)
    and overwritten with rank 0's by a broadcast, exactly as reference builder.py:618-621."""
    idx = torch.randperm(batch_all, generator=generator)
    if torch.device(device).type == "cuda":
        # page-locked source + non_blocking, as ShufflePlan's tables: the blocking copy of a pageable tensor cost the DenseCL
        # step 0.6 ms of host time per call (tools/host_profile.py ... cfg5), on a step that is host-bound
        idx = idx.pin_memory().to(device, non_blocking=True)
    else:
        idx = idx.to(device)
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
        if not torch.equal(torch.sort(idx).values, torch.arange(n)):
            # the kernels clamp an out-of-range row to the identity for memory safety (cp2_compose_pair,
            # cp2_feat_normalize_pool_pair): a bad table must be caught here, on the host, where it is built
            raise ValueError("ShufflePlan: idx_shuffle is not a permutation of 0..n-1")
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
        w = dist.all_to_all_single(out, x.contiguous(), out_counts, in_counts, async_op=True)
        COLLECTIVES.note(f"all_to_all rows {tuple(x.shape)}", w)
        w.wait()
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
