"""One optimisation step of CP2 pre-training (reference main.py:572-647) as a replayable unit.

Eager mode runs forward / backward / optimizer like the reference loop.  Graph mode captures
the WHOLE step -- composition, both encoders, the fused loss kernels, backward, the SGD update,
the EMA and the enqueue -- into one hipGraph and replays it: this is possible because nothing in
cp2_amd's hot path synchronises with the host (queue pointer, IoUs, meters all stay on device).
"""
from __future__ import annotations

from typing import Dict, Optional

import torch

INPUT_KEYS = ("img_a", "img_b", "bg0", "bg1", "pixel_ids_a", "pixel_ids_b", "region_ids_a", "region_ids_b")


class ForwardGraph:
    """A gradient-free forward (the key encoder: reference builder.py:1276) replayed as one hipGraph.

    The eager step is bound by host launch time (about 1100 launches, 15.7 ms of Python / dispatcher work against
    15.1 ms of GPU work at 32 img/GPU); the key encoder is a quarter of those launches, needs no autograd graph and
    contains no collective and no weight-gradient convolution (the MIOpen solvers that misbehave under replay, see
    DESIGN.md section 5), so it can be replayed from a graph in any mode, DDP included.  The first `warmup` calls per
    input signature run eagerly (MIOpen's find pass must not be captured); then the call is captured once and every
    later call is: copy the input into the static buffer, replay, hand out the static output.
    `on_replay` is called after each replay (host-side bookkeeping the captured code would have done)."""

    def __init__(self, fn, warmup: int = 3, on_replay=None):
        self.fn, self.warmup, self.on_replay = fn, warmup, on_replay
        self.entries: Dict[tuple, dict] = {}

    def reset(self):
        self.entries.clear()

    def __call__(self, x: torch.Tensor, tag=None) -> torch.Tensor:
        if torch.cuda.is_current_stream_capturing():       # already inside a whole-step capture
            return self.fn(x)
        key = (tuple(x.shape), tuple(x.stride()), x.dtype, x.device.index, tag)
        e = self.entries.get(key)
        if e is None:
            e = self.entries[key] = {"calls": 0, "graph": None}
        if e["graph"] is None:
            e["calls"] += 1
            if e["calls"] <= self.warmup:
                return self.fn(x)
            if e["calls"] < 0:                              # capture failed earlier: stay eager
                return self.fn(x)
            e["in"] = x.clone(memory_format=torch.preserve_format)
            graph = torch.cuda.CUDAGraph()
            try:
                # thread_local: another thread's HIP calls (RCCL watchdog, autograd workers) do not invalidate the capture
                with torch.cuda.graph(graph, capture_error_mode="thread_local"):
                    e["out"] = self.fn(e["in"])
            except Exception as err:                        # noqa: BLE001 -- any capture failure: same kernels, eagerly
                import warnings
                warnings.warn(f"ForwardGraph: capture failed ({type(err).__name__}: {err}); running this forward eagerly")
                e["calls"] = -(1 << 30)
                e.pop("in", None)
                return self.fn(x)
            e["graph"] = graph
        e["in"].copy_(x)
        e["graph"].replay()
        if self.on_replay is not None:
            self.on_replay()
        return e["out"]


class TrainStep:
    def __init__(self, model, optimizer, use_graph: bool = False, warmup_steps: int = 3):
        self.model, self.optimizer = model, optimizer
        self.use_graph, self.warmup_steps = use_graph, warmup_steps
        if use_graph:
            # libraries initialise per shape on first use (hipBLASLt refuses to do that while a stream is capturing): the
            # eager warm-up steps must run exactly the kernels the capture will record
            from .encoder import Conv2d
            Conv2d.graph_step = True
        self.graph: Optional[torch.cuda.CUDAGraph] = None
        self.static: Dict[str, torch.Tensor] = {}
        self.static_loss = None
        self._eager_calls = 0
        self.step_idx = 0

    def _inner(self):
        return self.model.module if hasattr(self.model, "module") else self.model

    def _eager(self, batch, idx_shuffle=None):
        loss = self.model(visualize=False, step=self.step_idx, new_epoch=False, idx_shuffle=idx_shuffle, **batch)
        self.optimizer.zero_grad(set_to_none=True)
        loss.backward()
        self.optimizer.step()
        return loss

    def _shuffle_index(self, n_all, device):
        from . import dist as cdist
        return cdist.make_shuffle_index(n_all, device)

    def __call__(self, batch: Dict[str, torch.Tensor]) -> torch.Tensor:
        from . import dist as cdist
        dev = batch["img_a"].device
        n_all = batch["img_a"].shape[0] * cdist.world_size()
        if not self.use_graph:
            loss = self._eager(batch)
            self.step_idx += 1
            return loss.detach()
        if self.graph is None:
            if not self.static:
                self.static = {k: batch[k].clone() for k in INPUT_KEYS}
                self.static_idx = self._shuffle_index(n_all, dev)
            if self._eager_calls < self.warmup_steps:      # warm-up on a side stream, as torch's capture rules ask
                s = torch.cuda.Stream()
                s.wait_stream(torch.cuda.current_stream())
                with torch.cuda.stream(s):
                    for k in INPUT_KEYS:
                        self.static[k].copy_(batch[k])
                    self.static_idx.copy_(self._shuffle_index(n_all, dev))
                    loss = self._eager(self.static, self.static_idx)
                torch.cuda.current_stream().wait_stream(s)
                self._eager_calls += 1
                self.step_idx += 1
                return loss.detach()
            self._inner().flush_logs()
            self.graph = torch.cuda.CUDAGraph()
            self.optimizer.zero_grad(set_to_none=True)
            with torch.cuda.graph(self.graph):
                self.static_loss = self._eager(self.static, self.static_idx).detach()
        for k in INPUT_KEYS:
            self.static[k].copy_(batch[k])
        self.static_idx.copy_(self._shuffle_index(n_all, dev))
        self.graph.replay()
        self.step_idx += 1
        return self.static_loss
