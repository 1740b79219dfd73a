"""One optimisation step of CP2 pre-training (reference main.py:572-647), and the key encoder's forward as a hipGraph.

`TrainStep` runs forward / backward / optimizer like the reference loop.  Nothing in cp2_amd's hot path synchronises
with the host (queue pointer, IoUs, meters all stay on the device), so the host runs ahead of the GPU and the step is
GPU-bound at 32 images per GPU.  A whole-step hipGraph mode existed in rounds 1-2; it measured 1.3 % slower than this
eager step and was removed in round 3 (DESIGN.md section 5 keeps what it found out about replay faults).
"""
from __future__ import annotations

from typing import Dict

import torch

INPUT_KEYS = ("img_a", "img_b", "bg0", "bg1", "pixel_ids_a", "pixel_ids_b", "region_ids_a", "region_ids_b")


class ForwardGraph:
    """A gradient-free forward (the key encoder: reference builder.py:1276) replayed as one hipGraph.

    The eager step is bound by host launch time (about 1100 launches, 15.7 ms of Python / dispatcher work against
    15.1 ms of GPU work at 32 img/GPU); the key encoder is a quarter of those launches, needs no autograd graph and
    contains no collective and no weight-gradient convolution (no "zero-fill, then accumulate through atomics" node,
    the kind that misbehaved under replay), so it can be replayed from a graph in any mode, DDP
    included.  The first `warmup` calls per
    input signature run eagerly (MIOpen's find pass must not be captured); then the call is captured once and every
    later call is: copy the input into the static buffer, replay, hand out the static output.
    `on_replay` is called after each replay (host-side bookkeeping the captured code would have done)."""

    def __init__(self, fn, warmup: int = 3, on_replay=None):
        self.fn, self.warmup, self.on_replay = fn, warmup, on_replay
        self.entries: Dict[tuple, dict] = {}

    def reset(self):
        self.entries.clear()

    def __call__(self, x: torch.Tensor, tag=None) -> torch.Tensor:
        if torch.cuda.is_current_stream_capturing():       # the caller is capturing a graph of its own
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
    """model(**batch) -> zero_grad -> backward -> optimizer.step (reference main.py:616-644), counting steps."""

    def __init__(self, model, optimizer):
        self.model, self.optimizer = model, optimizer
        self.step_idx = 0

    def __call__(self, batch: Dict[str, torch.Tensor], idx_shuffle=None) -> torch.Tensor:
        loss = self.model(visualize=False, step=self.step_idx, new_epoch=False, idx_shuffle=idx_shuffle, **batch)
        self.optimizer.zero_grad(set_to_none=True)
        loss.backward()
        self.optimizer.step()
        self.step_idx += 1
        return loss.detach()
