"""Gradient averaging over the ranks for a model whose parameters live in one flat buffer.

Reference: main.py:456-460 wraps the model in torch's DistributedDataParallel.  That reducer registers one autograd
hook per parameter and copies each gradient into its bucket with one small kernel -- 161 launches per ResNet-50 step,
which the single-rank RCCL rehearsal on an MI355X shows as ~1.0 ms of the 13.7 ms step, the
collectives themselves being almost free.  `FlatDDP` keeps DDP's contract (same constructor role, `.module`, `no_sync()`,
"module."-prefixed state_dict, rank 0's parameters and buffers broadcast at construction, averaged gradients in
`p.grad` when `backward()` returns, all-reduce overlapped with the rest of the backward pass) and does the local half
with one HIP launch per bucket: the parameters are split into a few contiguous ranges of the flat buffer (`bucket_mb`
each, formed from the END of the parameter list, the order the backward pass produces gradients in); when the last
gradient of a range has arrived, `cp2_pack_grads` writes the whole range, scaled by 1/W, into the flat gradient buffer
and one asynchronous all_reduce(SUM) over that range starts.  At the end of the backward pass every `p.grad` becomes a
view of the flat gradient buffer (so `optim.FlatSGD` reads constant addresses).

Every trainable parameter must receive a gradient in every synchronised backward pass, as with DDP's default
find_unused_parameters=False (builder.MODEL freezes what the chosen path never uses); which parameters are trainable is
read once, at construction (freeze or unfreeze afterwards: build a new FlatDDP, as with DDP).
"""
from __future__ import annotations

import contextlib
import ctypes
from typing import Callable, List, Optional, Sequence

import torch
import torch.distributed as dist

from . import dist as cdist


TAIL_BUCKET_MB = 4.0     # at most this much is all-reduced after the backward pass has ended (plan_buckets)


def _same_element_order(g: torch.Tensor, p: torch.Tensor) -> bool:
    return all(sg == sp for sg, sp, n in zip(g.stride(), p.stride(), p.shape) if n > 1)


def plan_buckets(numels: Sequence[int], bucket_elems: int, tail_elems: int = 0) -> List[tuple]:
    """[(t_begin, t_end)] tensor ranges in the order they complete during a backward pass: formed from the last
    tensor towards the first, a range closes once it holds >= bucket_elems elements.  The range that completes LAST (the
    front of the list: stem and first stage) is the one whose all-reduce nothing overlaps any more; with tail_elems > 0 it
    is cut so that its final piece holds about tail_elems elements only."""
    out, hi, acc = [], len(numels), 0
    for t in range(len(numels) - 1, -1, -1):
        acc += numels[t]
        if acc >= bucket_elems or t == 0:
            out.append((t, hi))
            hi, acc = t, 0
    lo, hi = out[-1]
    if tail_elems > 0 and hi - lo > 1 and sum(numels[lo:hi]) > 2 * tail_elems:
        acc, cut = 0, lo + 1
        for t in range(lo, hi - 1):
            acc += numels[t]
            cut = t + 1
            if acc >= tail_elems:
                break
        out[-1:] = [(cut, hi), (lo, cut)]
    return out


class GradReducer:
    """The bucket bookkeeping, independent of the model class: `params` in flat-buffer order, `offsets[i]` the first
    element of parameter i's slot, `total` the buffer length.  `pack(t_begin, t_end, grads, reducer)` performs the local
    copy (grads[i]: the gradient tensor of parameter i in its slot's element order, or None); the default -- and the only one
    the package ships -- is the HIP launch; the gloo tests of the bookkeeping pass their own for CPU tensors."""

    def __init__(self, params: Sequence[torch.nn.Parameter], offsets: Sequence[int], total: int, bucket_mb: float = 25.0,
                 pack: Optional[Callable] = None):
        self.params = list(params)
        self.offsets = list(offsets)
        n = len(self.params)
        if n == 0 or len(self.offsets) != n:
            raise ValueError("GradReducer: one slot offset per parameter")
        dev = self.params[0].device
        self.world = cdist.world_size()
        self.scale = 1.0 / self.world
        self.flat = torch.zeros(total, dtype=torch.float32, device=dev)
        self.views = [torch.as_strided(self.flat, p.shape, p.stride(), o) for p, o in zip(self.params, self.offsets)]
        numels = [p.numel() for p in self.params]
        per_mb = (1 << 20) // 4
        self.buckets = plan_buckets(numels, max(1, int(bucket_mb * per_mb)), int(min(bucket_mb / 4, TAIL_BUCKET_MB) * per_mb))
        self.ranges = [(self.offsets[lo], self.offsets[hi] if hi < n else total) for lo, hi in self.buckets]
        self.bucket_of = [0] * n
        for b, (lo, hi) in enumerate(self.buckets):
            for t in range(lo, hi):
                self.bucket_of[t] = b
        self.expect = [sum(1 for t in range(lo, hi) if self.params[t].requires_grad) for lo, hi in self.buckets]
        self.enabled = True
        self.layout_copies = 0                       # gradients that arrived in another element order (copied first)
        self._armed = False
        self._next = 0
        self._left: List[int] = []
        self._works: list = []
        self._keep: list = []
        self._pack = pack
        if pack is None:
            from . import ops
            self._plan = ops.SgdFlatPlan(self.offsets, numels, dev)
            self._ptrs = (ctypes.c_void_p * n)()
        self._hooks = [p.register_post_accumulate_grad_hook(self._make_hook(i)) for i, p in enumerate(self.params)
                       if p.requires_grad]

    def remove_hooks(self):
        for h in self._hooks:
            h.remove()
        self._hooks = []

    def _make_hook(self, i: int):
        def hook(param):
            self._ready(i)
        return hook

    def reset(self) -> None:
        """Forget a backward pass that did not reach its end (an exception inside backward() skips the engine's final
        callbacks, so _finish never ran): without this the next pass would neither re-arm nor all-reduce, and the replicas
        would drift apart silently.  FlatDDP.forward calls it, as DDP prepares its reducer at every forward."""
        if self._armed:
            for w in self._works:
                try:
                    w.wait()
                except Exception:                    # noqa: BLE001 -- the pass is being abandoned anyway
                    pass
        self._armed = False
        self._works, self._keep, self._next, self._left = [], [], 0, []

    def table_hash(self) -> int:
        """A hash of everything the ranks must agree on to issue the same collectives in the same order."""
        import zlib
        text = repr((self.buckets, self.ranges, self.expect, self.world)).encode()
        return zlib.crc32(text) | (len(self.buckets) << 32)

    def _ready(self, i: int) -> None:
        if not self.enabled:
            return
        if not self._armed:
            self._armed = True
            self._left = list(self.expect)
            self._works, self._keep, self._next = [], [], 0
            torch.autograd.Variable._execution_engine.queue_callback(self._finish)
        self._left[self.bucket_of[i]] -= 1
        # strictly in bucket order, whatever order the gradients arrive in: every rank issues the same collectives
        while self._next < len(self.buckets) and self._left[self._next] == 0:
            self._launch(self._next)
            self._next += 1

    def _launch(self, b: int) -> None:
        lo, hi = self.buckets[b]
        grads: List[Optional[torch.Tensor]] = []
        for t in range(lo, hi):
            p = self.params[t]
            g = p.grad if p.requires_grad else None
            if g is not None:
                if g.dtype != torch.float32 or g.is_sparse:
                    raise TypeError("GradReducer: fp32 dense gradients only")
                if g.stride() != p.stride() and not _same_element_order(g, p):
                    g = torch.empty_strided(p.shape, p.stride(), dtype=g.dtype, device=g.device).copy_(g)
                    self.layout_copies += 1
            grads.append(g)
        if self._pack is not None:
            self._pack(lo, hi, grads, self)
        else:
            from . import ops
            for t, g in zip(range(lo, hi), grads):
                self._ptrs[t] = None if g is None else g.data_ptr()
            ops.pack_grads(self._plan, self.flat, self._ptrs, lo, hi, self.scale)
        self._keep.append(grads)                          # alive until the pack launch is ordered before their reuse
        f_lo, f_hi = self.ranges[b]
        work = dist.all_reduce(self.flat[f_lo:f_hi], op=dist.ReduceOp.SUM, async_op=True)
        self._works.append(cdist.COLLECTIVES.note(f"c5 gradient all-reduce, bucket {b} of {len(self.buckets)} "
                                                  f"({(f_hi - f_lo) * 4 >> 20} MB)", work))

    def _finish(self) -> None:
        """End of the backward pass (autograd engine callback): wait for the all-reduces, hand out the averaged gradients."""
        self._armed = False
        missing = [b for b, left in enumerate(self._left) if left != 0]
        if missing:
            lo, hi = self.buckets[missing[0]]
            names = [t for t in range(lo, hi) if self.params[t].requires_grad and self.params[t].grad is None]
            raise RuntimeError(f"GradReducer: parameters {names[:4]} (flat order) of bucket {missing[0]} received no gradient in "
                               "this backward pass; freeze them (requires_grad=False) or run the step under no_sync()")
        for w in self._works:
            w.wait()
        self._works, self._keep = [], []
        for p, v in zip(self.params, self.views):
            if p.requires_grad:
                p.grad = v


class FlatDDP(torch.nn.Module):
    """Drop-in for DistributedDataParallel(model, ...) around builder.MODEL (reference main.py:456-460)."""

    def __init__(self, module: torch.nn.Module, bucket_mb: Optional[float] = None, broadcast_at_init: bool = True):
        super().__init__()
        if not cdist.is_dist():
            raise RuntimeError("FlatDDP: torch.distributed is not initialised")
        from . import builder
        self.module = module
        module.flatten_parameters()
        params = list(module.encoder_q.parameters())
        known = {id(p) for p in params}
        stray = [n for n, p in module.named_parameters() if p.requires_grad and id(p) not in known]
        if stray:
            raise ValueError(f"FlatDDP: trainable parameters outside encoder_q: {stray[:3]}")
        if broadcast_at_init and cdist.multi():
            self._broadcast_state()
        if not params[0].is_cuda:
            raise RuntimeError("FlatDDP: the model is on " + str(params[0].device) + "; the gradient pack kernel (cp2_pack_grads) "
                               "runs on the GPU only")
        self.reducer = GradReducer(params, module._flat_offsets, module._flat_q.numel(),
                                   builder.DDP_BUCKET_MB if bucket_mb is None else bucket_mb)
        # the bucket table is derived on every rank from its own parameter list: a rank that froze another set of
        # parameters, or was built with another bucket size, would issue other all-reduces and the job would hang in the
        # first backward pass -- fail here instead, with the ranks named
        cdist.assert_same_on_all_ranks("FlatDDP bucket table", self.reducer.table_hash(), params[0].device)

    @torch.no_grad()
    def _broadcast_state(self) -> None:
        """Rank 0's parameters and buffers everywhere, as DDP's constructor does."""
        m = self.module
        cdist.tracked("FlatDDP: broadcast of rank 0's query parameters", dist.broadcast(m._flat_q, src=0, async_op=True))
        cdist.tracked("FlatDDP: broadcast of rank 0's key parameters", dist.broadcast(m._flat_k, src=0, async_op=True))
        for n, b in m.named_buffers():
            if b.numel():
                cdist.tracked(f"FlatDDP: broadcast of buffer {n}", dist.broadcast(b, src=0, async_op=True))
        if getattr(m, "_flat_k_bf16", None) is not None:
            m._flat_k_bf16.copy_(m._flat_k)
        if getattr(m, "_flat_q_bf16", None) is not None:
            m._q_shadow_version = None
            m._refresh_query_shadow()

    def forward(self, *args, **kwargs):
        self.reducer.reset()
        return self.module(*args, **kwargs)

    @contextlib.contextmanager
    def no_sync(self):
        """Backward passes inside accumulate gradients locally without the all-reduce (DDP.no_sync)."""
        old = self.reducer.enabled
        self.reducer.enabled = False
        try:
            yield
        finally:
            self.reducer.enabled = old
