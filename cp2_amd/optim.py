"""Optimizer step of the query encoder as one HIP launch on the flat parameter buffer.

Reference: main.py:467-477 builds `torch.optim.SGD(params, lr, momentum, weight_decay)` and main.py:640-642 calls
`optimizer.step()`; torch runs that as ~17 multi-tensor launches.  `FlatSGD` is a `torch.optim.Optimizer` with the same
constructor arguments, `param_groups` (so the LR schedule of main.py:693-698 works unchanged) and `state_dict()` layout
(`momentum_buffer` per parameter), whose `step()` is `cp2_sgd_flat` (csrc/sgd.hip): bit-identical updates, one launch,
and the bf16 image of the new weights that the query encoder's convolutions read under autocast.
"""
from __future__ import annotations

import torch

from . import ops


def _same_element_order(g: torch.Tensor, p: torch.Tensor) -> bool:
    """Strides agree on every dimension that has more than one element (a [co, ci, 1, 1] weight is the same memory in
    'contiguous' and 'channels-last' strides)."""
    return all(sg == sp for sg, sp, n in zip(g.stride(), p.stride(), p.shape) if n > 1)


class FlatSGD(torch.optim.Optimizer):
    def __init__(self, model, lr, momentum: float = 0.0, weight_decay: float = 0.0, dampening: float = 0.0,
                 nesterov: bool = False):
        if dampening != 0.0 or nesterov:
            raise NotImplementedError("FlatSGD: dampening / Nesterov are not part of the reference recipe")
        inner = model.module if hasattr(model, "module") else model
        inner.flatten_parameters()
        self._model = inner
        self._all = list(inner.encoder_q.parameters())            # flat-buffer order
        trainable = [p for p in self._all if p.requires_grad]
        known = {id(p) for p in self._all}
        stray = [n for n, p in inner.named_parameters() if p.requires_grad and id(p) not in known]
        if stray:
            raise ValueError(f"FlatSGD: trainable parameters outside encoder_q: {stray[:3]}")
        super().__init__(trainable, dict(lr=lr, momentum=momentum, weight_decay=weight_decay, dampening=0.0,
                                         nesterov=False, maximize=False, foreach=None, differentiable=False, fused=None))
        self._flat_id = None
        self._bind()

    def _bind(self):
        """(Re)build the tables when the model re-homed its flat buffers (e.g. after .to(device))."""
        m = self._model
        m.flatten_parameters()
        if self._flat_id == m._flat_q.data_ptr():
            return
        self._flat_id = m._flat_q.data_ptr()
        self._plan = ops.SgdFlatPlan(m._flat_offsets, [p.numel() for p in self._all], m._flat_q.device)
        old = getattr(self, "_buf", None)
        self._buf = torch.zeros_like(m._flat_q)
        if old is not None and old.numel() == self._buf.numel():
            self._buf.copy_(old)
        for p, off in zip(self._all, m._flat_offsets):
            if p.requires_grad and "momentum_buffer" in self.state.get(p, {}):
                self.state[p]["momentum_buffer"] = torch.as_strided(self._buf, p.shape, p.stride(), off)
        m.enable_query_shadow()

    def _momentum_views(self):
        for p, off in zip(self._all, self._model._flat_offsets):
            if p.requires_grad and p.grad is not None and "momentum_buffer" not in self.state[p]:
                self.state[p]["momentum_buffer"] = torch.as_strided(self._buf, p.shape, p.stride(), off)

    @torch.no_grad()
    def step(self, closure=None):
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        if len(self.param_groups) != 1:
            raise NotImplementedError("FlatSGD: one parameter group (the reference uses one)")
        self._bind()
        g0 = self.param_groups[0]
        grads, keep = self._plan.grads, []
        for i, p in enumerate(self._all):
            g = p.grad if p.requires_grad else None
            if g is None:
                grads[i] = None
                continue
            if g.dtype != torch.float32 or g.is_sparse:
                raise TypeError("FlatSGD: fp32 dense gradients only")
            if g.stride() != p.stride() and not _same_element_order(g, p):   # slot order = the parameter's own dense layout
                g = torch.empty_strided(p.shape, p.stride(), dtype=g.dtype, device=g.device).copy_(g)
                keep.append(g)
            grads[i] = g.data_ptr()
        if g0["momentum"] != 0 and len(self.state) < len(self.param_groups[0]["params"]):
            self._momentum_views()
        m = self._model
        ops.sgd_flat(self._plan, m._flat_q, self._buf, m._flat_q_bf16, g0["lr"], g0["momentum"], g0["weight_decay"])
        m._query_shadow_written()
        return loss

    def load_state_dict(self, state_dict):
        super().load_state_dict(state_dict)
        # torch replaced the momentum buffers by the loaded tensors: copy them back into the flat buffer
        for p, off in zip(self._all, self._model._flat_offsets):
            st = self.state.get(p)
            if st and "momentum_buffer" in st and st["momentum_buffer"] is not None:
                view = torch.as_strided(self._buf, p.shape, p.stride(), off)
                view.copy_(st["momentum_buffer"])
                st["momentum_buffer"] = view
