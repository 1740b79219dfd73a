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
    """verify (graph mode): before the first replay is trusted, one eager forward/backward on the same batch from the
    same state provides reference gradients; the first replay's gradients must agree tensor by tensor (MIOpen
    weight-gradient solvers have been seen to return garbage under replay).  On disagreement, or
    if the capture itself fails, the state is rolled back and the step continues eagerly (`self.fallback_reason`)."""

    def __init__(self, model, optimizer, use_graph: bool = False, warmup_steps: int = 3, verify: bool = True,
                 reverify_every: int = 200, verify_first: int = 3):
        self.model, self.optimizer = model, optimizer
        self.use_graph, self.warmup_steps, self.verify = use_graph, warmup_steps, verify
        # the replay fault of DESIGN.md section 5 depends on what the graph's memory pool holds, so a replay that was
        # right once can go wrong later: with verify on, every `reverify_every`-th replay is checked again
        self.reverify_every = reverify_every
        # ... and both faults found so far (MIOpen's atomic weight gradients, ATen's semaphore reduction) showed up on the
        # SECOND replay, not the first: the first `verify_first` replays are all checked
        self.verify_first = verify_first
        self._replays = 0
        self.fallback_reason = None
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
            return self._capture_and_first_replay(batch, n_all, dev)
        for k in INPUT_KEYS:
            self.static[k].copy_(batch[k])
        self.static_idx.copy_(self._shuffle_index(n_all, dev))
        self._replays += 1
        if self.verify and ((self.reverify_every and self._replays % self.reverify_every == 0) or self._replays < self.verify_first):
            return self._checked_replay(batch)
        self.graph.replay()
        self._after_replay()
        self.step_idx += 1
        return self.static_loss

    def _after_replay(self):
        """Host-side bookkeeping the captured code did once, at capture time: per-step log record, IoU lists, BatchNorm
        batch counters (reference builder.py:1254-1257,1553-1604; torch.nn.BatchNorm2d.num_batches_tracked)."""
        inner = self._inner()
        if getattr(self, "_log_template", None) is not None:
            step0, n, names, vals = self._log_template
            inner._pending_logs.append((self.step_idx, n, names, vals.clone()))
        for src, dst in getattr(self, "_iou_templates", ()):
            dst.append(src.clone())
            if len(dst) >= 1024:
                dst[:] = [torch.cat(dst)]
        if len(inner._pending_logs) >= (inner.sync_logs_every or 4096):
            inner.flush_logs()
        for m in getattr(self, "_bn_modules", ()):
            m._pending_batches += 1

    # ------------------------------------------------------------------ capture, checked against an eager step
    def _snapshot(self):
        import copy
        inner = self._inner()
        return {"model": {k: v.detach().clone() for k, v in inner.state_dict().items()},
                "optim": copy.deepcopy(self.optimizer.state_dict()),
                "lists": (len(inner._pending_logs), len(inner.correlation_ious), len(inner.masked_correlation_ious))}

    def _restore(self, snap, optimizer_too: bool):
        inner = self._inner()
        inner.load_state_dict(snap["model"])
        if optimizer_too:
            self.optimizer.load_state_dict(snap["optim"])
        n_logs, n_a, n_b = snap["lists"]
        del inner._pending_logs[n_logs:], inner.correlation_ious[n_a:], inner.masked_correlation_ious[n_b:]
        for m in inner.modules():                  # state_dict() folded the lazy BN batch counters into the snapshot
            if hasattr(m, "_pending_batches"):
                m._pending_batches = 0

    def _give_up_graph(self, reason, batch):
        import warnings
        from .encoder import Conv2d
        warnings.warn(f"TrainStep: hipGraph step disabled ({reason}); continuing eagerly")
        self.fallback_reason, self.use_graph, self.graph = reason, False, None
        Conv2d.graph_step = False
        loss = self._eager(batch)
        self.step_idx += 1
        return loss.detach()

    def _capture_and_first_replay(self, batch, n_all, dev):
        inner = self._inner()
        inner.flush_logs()
        for k in INPUT_KEYS:
            self.static[k].copy_(batch[k])
        self.static_idx.copy_(self._shuffle_index(n_all, dev))
        params = [p for p in inner.parameters() if p.requires_grad]
        ref_grads = ref_loss = None
        snap = None
        if self.verify:
            snap, ref_loss, ref_grads = self._eager_probe(params)
        n_logs, n_iou = len(inner._pending_logs), len(inner.correlation_ious)
        bn_before = {m: m._pending_batches for m in inner.modules() if hasattr(m, "_pending_batches")}
        graph = torch.cuda.CUDAGraph()
        self.optimizer.zero_grad(set_to_none=True)
        try:
            with torch.cuda.graph(graph):
                self.static_loss = self._eager(self.static, self.static_idx).detach()
        except Exception as err:                                # noqa: BLE001
            if snap is not None:
                self._restore(snap, optimizer_too=False)        # a capture executes nothing, but host-side lists grew
            return self._give_up_graph(f"capture failed: {type(err).__name__}: {err}", batch)
        self.graph = graph
        # what the captured python code appended once is what every replay has to append again (meters and
        # IoU lists saw one record per flush in graph mode, BN batch counters stalled)
        self._log_template = inner._pending_logs[n_logs] if len(inner._pending_logs) > n_logs else None
        self._iou_templates = []
        if len(inner.correlation_ious) > n_iou:
            self._iou_templates = [(inner.correlation_ious[n_iou], inner.correlation_ious),
                                   (inner.masked_correlation_ious[n_iou], inner.masked_correlation_ious)]
        self._bn_modules = [m for m, b in bn_before.items() if m._pending_batches != b]
        del inner._pending_logs[n_logs:], inner.correlation_ious[n_iou:], inner.masked_correlation_ious[n_iou:]
        for m, b in bn_before.items():
            m._pending_batches = b                              # a captured call executes nothing
        self.graph.replay()
        if self.verify:
            bad = self._compare(params, ref_loss, ref_grads)
            if bad:
                self._restore(snap, optimizer_too=True)
                return self._give_up_graph("replayed gradients differ from eager: " + bad, batch)
        self._after_replay()
        self.step_idx += 1
        return self.static_loss

    def _eager_probe(self, params):
        """One eager forward / backward on the static batch from the current state (on a side stream: a backward on the
        stream a capture later starts from made hipStreamEndCapture fail), then the state is put back."""
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            snap = self._snapshot()
            loss = self.model(visualize=False, step=self.step_idx, new_epoch=False, idx_shuffle=self.static_idx, **self.static)
            self.optimizer.zero_grad(set_to_none=True)
            loss.backward()
            ref_loss = loss.detach().clone()
            ref_grads = [None if p.grad is None else p.grad.detach().clone() for p in params]
            del loss
            self._restore(snap, optimizer_too=False)          # the eager probe advanced EMA / BN statistics / queue
            self.optimizer.zero_grad(set_to_none=True)
            inner = self._inner()
            if hasattr(inner, "_refresh_query_shadow"):        # load_state_dict bumped every parameter version: rebuild the
                inner._refresh_query_shadow()                  # bf16 image here, not inside the captured forward
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        return snap, ref_loss, ref_grads

    def _compare(self, params, ref_loss, ref_grads):
        """None if the replay agrees with the eager probe.  Per tensor: L2 error <= 1e-1 |g| and no element further
        than 0.25 max|g| from the eager value, both with a floor of 5e-4 of the whole gradient's norm; non-finite
        anywhere fails.  Two eager runs of the bf16 encoders differ by ~1e-2 per tensor (MIOpen data-gradient kernels
        that accumulate in bf16 atomics), more at the end of the chain: the stem's weight gradient of the 8-image test
        model reached 5.2e-2 once in four runs, so 5e-2 was too tight a bound for a check that now runs on the first
        three replays; both faults this check exists for produced non-finite values.  (Round 1 accepted 0.2 |g|.)"""
        if not bool(torch.isfinite(self.static_loss)) or abs(float(self.static_loss) - float(ref_loss)) > 2e-2 * max(1.0, abs(float(ref_loss))):
            return f"loss {float(self.static_loss):.5f} vs eager {float(ref_loss):.5f}"
        # a few gradients (the first BatchNorm's bias) are sums that cancel to ~1e-3 of the others: their run-to-run
        # noise is set by the size of the terms, not of the result, hence the floor relative to the whole gradient
        total = float(torch.sqrt(sum((g.float() ** 2).sum() for g in ref_grads if g is not None)))
        for p, g in zip(params, ref_grads):
            if g is None or p.grad is None:
                continue
            a, b = p.grad.float(), g.float()
            den, err, amax = float(b.norm()), float((a - b).norm()), float((a - b).abs().max())
            if not (err == err) or err > max(1e-1 * den, 5e-4 * total) + 1e-6 or amax > max(0.25 * float(b.abs().max()), 5e-4 * total) + 1e-6:
                return f"gradient of a {tuple(p.shape)} parameter: |replay - eager| = {err:.3e}, |eager| = {den:.3e}, max |diff| = {amax:.3e}"
        return None

    def _checked_replay(self, batch):
        inner = self._inner()
        params = [p for p in inner.parameters() if p.requires_grad]
        snap, ref_loss, ref_grads = self._eager_probe(params)
        self.graph.replay()
        bad = self._compare(params, ref_loss, ref_grads)
        if bad:
            self._restore(snap, optimizer_too=True)
            return self._give_up_graph(f"replay {self._replays} differs from eager: " + bad, batch)
        self._after_replay()
        self.step_idx += 1
        return self.static_loss
