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


class TrainStep:
    def __init__(self, model, optimizer, use_graph: bool = False, warmup_steps: int = 3):
        self.model, self.optimizer = model, optimizer
        self.use_graph, self.warmup_steps = use_graph, warmup_steps
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
