"""CPU baseline of one CP2 pre-training step -- TEST / BASELINE INFRASTRUCTURE ONLY.

The reference cannot run on CPU unmodified (torch.cuda.set_device, .cuda(), nvidia-smi:
main.py:316,355; builder.py:175,618,1420), so the CPU baseline is this port: the same step
(reference main.py:616-647 + builder.py:1124-1448) assembled from the oracle functions of
oracle/cp2_oracle.py (each pinned against the reference's own code by the golden vectors) around
the shared plain-torch encoder definition.  Only bench.py's `cpu_baseline` leg and tests import it.
"""
from __future__ import annotations

import copy
import time

import torch

from cp2_amd.encoder import build_segmentor   # plain torch.nn module definition, no kernels
from oracle import cp2_oracle as O


class CpuCP2Step:
    def __init__(self, cfg, K=65536, dim=128, m=0.999, temp_global=0.2, temp_local=1.0, lmbd_dense=0.2,
                 output_stride=16, lr=0.03, momentum=0.9, weight_decay=1e-4, seed=0):
        torch.manual_seed(seed)
        self.enc_q = build_segmentor(cfg.model)
        self.enc_k = copy.deepcopy(self.enc_q)
        for p in self.enc_k.parameters():
            p.requires_grad = False
        self.queue = torch.nn.functional.normalize(torch.randn(dim, K), dim=0)
        self.ptr = 0
        self.m, self.tg, self.tl, self.lmbd, self.os = m, temp_global, temp_local, lmbd_dense, output_stride
        self.opt = torch.optim.SGD([p for p in self.enc_q.parameters() if p.requires_grad], lr, momentum=momentum,
                                   weight_decay=weight_decay)
        self.split = {}

    def step(self, batch):
        """One step; wall time per phase is accumulated in self.split (seconds): composition, query forward, EMA,
        key forward (+ shuffle), loss section incl. statistics, enqueue, backward + SGD."""
        tick = time.perf_counter
        t = [tick()]
        img_a, _ = O.compose_mask(batch["img_a"], batch["bg0"])
        img_b, _ = O.compose_mask(batch["img_b"], batch["bg1"])
        t.append(tick())
        q = self.enc_q(img_a)
        t.append(tick())
        with torch.no_grad():
            new = O.momentum_update([p.data for p in self.enc_k.parameters()], [p.data for p in self.enc_q.parameters()], self.m)
            for p, v in zip(self.enc_k.parameters(), new):
                p.data = v
            t.append(tick())
            perm = torch.randperm(img_b.shape[0])
            k = O.unshuffle_take(self.enc_k(O.shuffle_take(img_b, perm, 0, 1)), perm, 0, 1)
        t.append(tick())
        out = O.cp2_loss_section(q, k, batch["bg0"], batch["bg1"], batch["pixel_ids_a"], batch["pixel_ids_b"],
                                 batch["region_ids_a"], batch["region_ids_b"], self.queue, output_stride=self.os,
                                 temp_global=self.tg, temp_local=self.tl, lmbd_dense=self.lmbd, with_stats=True)
        t.append(tick())
        self.queue, self.ptr = O.dequeue_and_enqueue(self.queue, self.ptr, out["k_pos"].detach())
        t.append(tick())
        self.opt.zero_grad()
        out["loss"].backward()
        self.opt.step()
        t.append(tick())
        for name, a, b in zip(("compose", "query_fwd", "ema", "key_fwd", "loss_section", "enqueue", "backward_sgd"), t, t[1:]):
            self.split[name] = self.split.get(name, 0.0) + (b - a)
        return float(out["loss"].detach())


def time_cpu_baseline(cfg, make_batch, b, h, w, K, steps=2, warmup=1, threads=None):
    """images/sec of the CPU port on `threads` host threads (default: all)."""
    if threads:
        torch.set_num_threads(threads)
    runner = CpuCP2Step(cfg, K=K)
    batches = [{k: v.cpu() for k, v in make_batch(b, h, w, "cpu", seed=100 + i).items()} for i in range(warmup + steps)]
    for i in range(warmup):
        runner.step(batches[i])
    runner.split = {}
    t0 = time.perf_counter()
    for i in range(warmup, warmup + steps):
        runner.step(batches[i])
    dt = time.perf_counter() - t0
    time_cpu_baseline.last_split_ms = {k: round(1e3 * v / steps, 1) for k, v in runner.split.items()}   # per step
    return b * steps / dt, torch.get_num_threads(), dt


time_cpu_baseline.last_split_ms = {}
