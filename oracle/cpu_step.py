"""CPU baseline of one CP2 pre-training step -- TEST / BASELINE INFRASTRUCTURE ONLY.

The reference cannot run on CPU unmodified (torch.cuda.set_device, .cuda(), nvidia-smi:
main.py:316,355; builder.py:175,618,1420), so the CPU baseline is this port: the same step
(reference main.py:616-647 + builder.py:1124-1448) assembled from the oracle functions of
oracle/cp2_oracle.py (each pinned against the reference's own code by the golden vectors) around
the shared plain-torch encoder definition.  Only bench.py's `cpu_baseline` leg and tests import it.
"""
from __future__ import annotations

import copy
import sys
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


class CpuDenseCLStep:
    """BASELINE config 5: one DenseCL step (reference builder.py:667-999 with the flags of main.py:148-153 / scripts/
    10-11-densecl.sh:33-52) on the CPU, assembled from the oracle's DenseCL functions: backbone + DenseCLNeck of both
    encoders, EMA, shuffle, global loss vs `queue`, local loss vs `queue2` with the logits materialised as the reference
    does, rank 0's score statistics (torch.quantile over every row's negatives, :776-786 / :875-886), both enqueues."""

    def __init__(self, cfg, K=65536, dim=128, m=0.999, temperature=0.2, lmbd=0.5, lr=0.03, momentum=0.9, weight_decay=1e-4,
                 seed=0, with_stats=True):
        from cp2_amd.builder import DenseCLNeck              # plain torch.nn definition (reference builder.py:179-274)
        torch.manual_seed(seed)
        self.bb_q = build_segmentor(cfg.model).backbone
        self.neck_q = DenseCLNeck(self.bb_q.feat_dim, 2048, dim)
        self.bb_k, self.neck_k = copy.deepcopy(self.bb_q), copy.deepcopy(self.neck_q)
        for p in list(self.bb_k.parameters()) + list(self.neck_k.parameters()):
            p.requires_grad = False
        self.neck_q.global_predictor.requires_grad_(False)      # unused without use_predictor (find_unused_parameters in the reference)
        self.neck_q.local_predictor.requires_grad_(False)
        self.queue = torch.nn.functional.normalize(torch.randn(dim, K), dim=0)
        self.queue2 = torch.nn.functional.normalize(torch.randn(dim, K), dim=0)
        self.ptr = self.ptr2 = 0
        self.m, self.t, self.lmbd, self.with_stats = m, temperature, lmbd, with_stats
        with torch.no_grad():
            self.stride = 224 // self.bb_q(torch.rand(1, 3, 224, 224))[3].shape[2]
        self.q_params = list(self.bb_q.parameters()) + list(self.neck_q.parameters())
        self.k_params = list(self.bb_k.parameters()) + list(self.neck_k.parameters())
        self.opt = torch.optim.SGD([p for p in self.q_params if p.requires_grad], lr, momentum=momentum, weight_decay=weight_decay)
        self.split = {}

    def step(self, batch):
        tick = time.perf_counter
        F = torch.nn.functional
        t = [tick()]
        ids_a, ids_b = O.strided_gather(batch["pixel_ids_a"], self.stride), O.strided_gather(batch["pixel_ids_b"], self.stride)
        eq = self.bb_q(batch["img_a"])[3]
        nq = self.neck_q(eq)
        t.append(tick())
        with torch.no_grad():
            new = O.momentum_update([p.data for p in self.k_params], [p.data for p in self.q_params], self.m)
            for p, v in zip(self.k_params, new):
                p.data = v
            t.append(tick())
            perm = torch.randperm(eq.shape[0])
            ek = self.bb_k(O.shuffle_take(batch["img_b"], perm, 0, 1))[3]
            nk = self.neck_k(ek)
            un = lambda x: O.unshuffle_take(x, perm, 0, 1)          # noqa: E731
            ek, kl, kg, kp = un(ek), un(nk["x_local_proj"]), un(nk["x_global_proj"]), un(nk["x_avgpool_local_proj"])
        t.append(tick())
        qg, kg = F.normalize(nq["x_global_proj"], dim=1), F.normalize(kg, dim=1)
        loss_g = O.densecl_global_loss(qg, kg, self.queue, self.t)
        loss_l, _, neg, _ = O.densecl_local_loss(F.normalize(eq.flatten(2), dim=1), F.normalize(ek.flatten(2), dim=1),
                                                 F.normalize(nq["x_local_proj"].flatten(2), dim=1), F.normalize(kl.flatten(2), dim=1),
                                                 ids_a, ids_b, self.queue2, self.t, 0.0)
        if self.with_stats:
            with torch.no_grad():
                qs = torch.tensor([0.25, 0.5, 0.75])
                torch.quantile(neg, qs, dim=1), neg.mean(1), torch.quantile(qg @ self.queue, qs, dim=1)
        loss = (1 - self.lmbd) * loss_g + self.lmbd * loss_l
        t.append(tick())
        self.queue, self.ptr = O.dequeue_and_enqueue(self.queue, self.ptr, kg.detach())
        self.queue2, self.ptr2 = O.dequeue_and_enqueue(self.queue2, self.ptr2, F.normalize(kp, dim=1).detach())
        t.append(tick())
        self.opt.zero_grad()
        loss.backward()
        self.opt.step()
        t.append(tick())
        for name, a, b in zip(("query_fwd", "ema", "key_fwd", "loss_section", "enqueue", "backward_sgd"), t, t[1:]):
            self.split[name] = self.split.get(name, 0.0) + (b - a)
        return float(loss.detach())


def time_cpu_baseline(cfg, make_batch, b, h, w, K, steps=2, warmup=1, threads=None, densecl=False, output_stride=16,
                      temp_local=1.0, lmbd_dense=0.2, progress=True):
    """images/sec of the CPU port on `threads` host threads (default: all)."""
    if threads:
        torch.set_num_threads(threads)
    runner = CpuDenseCLStep(cfg, K=K) if densecl else CpuCP2Step(cfg, K=K, output_stride=output_stride, temp_local=temp_local,
                                                                 lmbd_dense=lmbd_dense)
    batches = [{k: v.cpu() for k, v in make_batch(b, h, w, "cpu", seed=100 + i).items()} for i in range(warmup + steps)]
    def note(what):                                      # one line per CPU step: a config-4 step takes a minute
        if progress:
            print(f"[cpu baseline] {what}", file=sys.stderr, flush=True)
    note(f"{warmup} warm-up + {steps} timed steps of {b} images on {torch.get_num_threads()} threads")
    for i in range(warmup):
        runner.step(batches[i])
        note(f"warm-up step {i + 1} done")
    runner.split = {}
    t0 = time.perf_counter()
    for i in range(warmup, warmup + steps):
        runner.step(batches[i])
        note(f"timed step {i - warmup + 1} of {steps} done, {time.perf_counter() - t0:.1f} s")
    dt = time.perf_counter() - t0
    time_cpu_baseline.last_split_ms = {k: round(1e3 * v / steps, 1) for k, v in runner.split.items()}   # per step
    return b * steps / dt, torch.get_num_threads(), dt


time_cpu_baseline.last_split_ms = {}
