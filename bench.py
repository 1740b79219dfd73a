#!/usr/bin/env python3
"""Headline benchmark: CP2 pre-training images/sec (BASELINE.json metric).

    python bench.py --gpus 1 --steps 30 --warmup 10
    python bench.py --gpus N --steps K --warmup W              # starts its own N ranks (one process per GPU)
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W  # or under a launcher (RANK / WORLD_SIZE in the env)

A step = one full optimisation step of the CP2 hot path on one batch of synthetic copy-paste
pairs: composition, both encoders (PyTorch-ROCm, bf16 autocast), EMA, shuffle-BN, fused dense +
instance InfoNCE (hand-written gfx950 kernels, fp32), backward, SGD update, enqueue.  Workload at
N=1 = BASELINE.json configs[1]: ResNet-50 + FCN(contrast) head at output stride 16, 224x224,
queue 65536, 32 images per GPU (weak scaling: 32 per GPU at every N).  Prints ONE JSON line.

With N > 1 the line also carries `comm`: backend, rccl_ranks, and per-step times of every exchange step measured with
events on the stream it runs on -- C1 shuffle-BN image exchange, C3 key un-shuffle, C4 key all-gather + enqueue, the
main stream's wait for the side stream when --overlap selects one, and the exposed (non-overlapped) part of the gradient
averaging (step time minus the step time of a few extra steps under no_sync()) -- so that the first multi-GPU run explains
its own scaling.  `host_issue_ms_per_step` is the time the CPU needs to enqueue a step: when it approaches `ms_per_step`
the run was host-bound, not GPU-bound.  `--rehearse-collectives` runs the N > 1 code path with one rank over RCCL.

`--workload cfg2|cfg4|cfg5` selects the BASELINE.json configuration (default cfg2 = configs[1], the one the metric is
quoted on): cfg4 = configs[3] (ResNet-101 + dilated FCN head at output stride 8, 512x512, queue 131072, 8 img/GPU), cfg5 =
configs[4] (DenseCL, scripts/10-11-densecl.sh: ResNet-50 backbone of configs/config_pretrain.py + DenseCL neck, per-pixel
rows against queue2, no copy-paste mask).  `metric`, `config.workload`, `roofline` (that workload's dominant hand-written
kernel) and `cpu_baseline` (the CPU port of the same step) are built from what actually ran.
"""
import argparse
import json
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
# MIOpen: let PyTorch ask for the measured-fastest convolution solver (cudnn.benchmark) and keep the search
# short (FAST find mode, ~15 s of warm-up on a fresh box).  Immediate mode picks asm implicit-GEMM solvers that
# are ~40 % slower for these ResNet-50 shapes in bf16 NHWC (measured: 22.1 vs 15.5 ms for the encoder work).
os.environ.setdefault("MIOPEN_FIND_MODE", "FAST")
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")     # dmabuf IPC: what RCCL needs on this driver
# ... starting from the solver rankings MIOpen wrote on an MI355X for these convolutions (cp2_amd/miopen_db) instead of searching
# them again: the same step time (same-box A/B), 57 s less start-up per process, no W concurrent searches at N > 1
# (cp2_amd/miopen_cache.py; CP2_MIOPEN_DB=0 searches)
from cp2_amd.miopen_cache import use_shipped_find_db  # noqa: E402
MIOPEN_DB = use_shipped_find_db()

HBM_PEAK_GBS = 8000.0          # MI355X HBM3E peak, /opt/skills/guides/MI355X_MICROARCH.md "Chip-level parameters"
BF16_DENSE_PEAK_TFLOPS = 2500.0
F32_MFMA_PEAK_TFLOPS = 157.3   # v_mfma_f32_32x32x2_f32: the f32 vector rate (same guide, "Matrix cores")


# BASELINE.json configs the bench can run end to end on one GPU (reference: configs/*.py, scripts/10-11-densecl.sh:33-52)
WORKLOADS = {
    "cfg2": dict(label="BASELINE configs[1]", config="config_pretrain_r50_fcn.py", img=224, queue=65536, batch=32, densecl=False,
                 cpu=(32, 5, 2)),
    "cfg4": dict(label="BASELINE configs[3]", config="config_pretrain_r101_d8.py", img=512, queue=131072, batch=8, densecl=False,
                 cpu=(8, 2, 1)),
    "cfg5": dict(label="BASELINE configs[4]", config="config_pretrain.py", img=224, queue=65536, batch=32, densecl=True,
                 cpu=(8, 2, 1)),
}


def parse():
    p = argparse.ArgumentParser()
    p.add_argument("--gpus", type=int, default=1)
    p.add_argument("--steps", type=int, default=50)
    p.add_argument("--warmup", type=int, default=10)
    p.add_argument("--workload", default="cfg2", choices=sorted(WORKLOADS),
                   help="BASELINE.json configuration: cfg2 = configs[1] (the metric's), cfg4 = configs[3], cfg5 = configs[4] (DenseCL)")
    p.add_argument("--batch-per-gpu", type=int, default=None, help="default: the workload's (32; cfg4: 8)")
    p.add_argument("--img", type=int, default=None, help="default: the workload's (224; cfg4: 512)")
    p.add_argument("--queue", type=int, default=None, help="default: the workload's (65536; cfg4: 131072)")
    p.add_argument("--config", default=None, help="model config file; default: the workload's")
    p.add_argument("--amp", default="bf16", choices=["bf16", "none"])
    p.add_argument("--no-cpu-baseline", action="store_true")
    p.add_argument("--gemm-1x1", default="on", choices=["on", "off"],
                   help="1x1 stride-1 convolutions: hipBLASLt GEMM for forward / data gradient where faster (A/B)")
    p.add_argument("--cpp-nodes", default="on", choices=["on", "off"],
                   help="C++ autograd nodes for the weight images / 1x1 convolutions (off: the Python nodes) (A/B)")
    p.add_argument("--flat-sgd", default="on", choices=["on", "off"],
                   help="optimizer step as one HIP launch on the flat parameter buffer (off: torch.optim.SGD) (A/B)")
    p.add_argument("--fused-bn", default="on", choices=["on", "off"], help="encoder fast path: fused BN(+add)(+ReLU) kernels (A/B)")
    p.add_argument("--overlap", default="auto", choices=["auto", "gather", "off"],
                   help="side HIP stream for the key branch: off = everything in order on one stream, gather = EMA + shuffle "
                        "exchange + key gather on a side stream (north_star's form); auto = one stream at N = 1, and at N > 1 "
                        "MEASURED: a few steps of each of off / gather after the warm-up, the faster one (max over ranks) runs the "
                        "timed region and both numbers go into `comm`")
    p.add_argument("--calib-steps", type=int, default=6, help="--overlap auto at N > 1: steps per candidate (>= 5)")
    p.add_argument("--settle-seconds", type=float, default=6.0,
                   help="after the warm-up steps: untimed chunks of 10 steps until the step time stops moving (0.5 %%), at most "
                        "this long (0 = off); a cold box's first process measures 2 %% slow without it")
    p.add_argument("--timeout", type=float, default=120.0, help="process-group timeout in seconds (N > 1); the hang watchdog "
                   "names the exchange step that did not complete at 0.8 x this and exits non-zero")
    p.add_argument("--shuffle-exchange", default="all_to_all", choices=["all_to_all", "all_gather"],
                   help="shuffle-BN rows by all-to-all (only the rows a rank keeps travel) or the reference's all-gather form (A/B)")
    p.add_argument("--backend", default="nccl", help="torch.distributed backend (nccl = RCCL; gloo only to rehearse N>1 on one GPU)")
    p.add_argument("--one-device", action="store_true", help="rehearsal: every rank uses cuda:0")
    p.add_argument("--rehearse-collectives", action="store_true",
                   help="N = 1 only: run every collective of the N > 1 step (shuffle exchange, key gathers, DDP all-reduce) over the "
                        "chosen backend with ONE rank -- no data moves, but every stream hand-over of the multi-GPU step is paid: "
                        "the plumbing cost of the collectives on this box")
    p.add_argument("--bucket-mb", type=int, default=0, help="gradient bucket size in MB (0 = the model's default, builder.DDP_BUCKET_MB)")
    p.add_argument("--grad-sync", default="flat", choices=["flat", "ddp"],
                   help="N > 1: gradient averaging by cp2_amd.ddp.FlatDDP (one pack launch + one all-reduce per bucket of the flat "
                        "gradient buffer; what cp2_amd.main uses) or by torch's DistributedDataParallel (one copy launch per parameter)")
    p.add_argument("--nosync-steps", type=int, default=10,
                   help="N > 1: extra steps under DDP.no_sync() after the timed region (exposed all-reduce time); 0 = skip")
    p.add_argument("--cpu-batch", type=int, default=None, help="CPU baseline: images per step (default: cfg2 32 = the same b; cfg4 8; cfg5 8)")
    p.add_argument("--cpu-steps", type=int, default=None, help="CPU baseline: timed steps after --cpu-warmup (default: cfg2 5; cfg4 / cfg5 2)")
    p.add_argument("--cpu-warmup", type=int, default=None, help="default: cfg2 2; cfg4 / cfg5 1")
    args = p.parse_args()
    wl = WORKLOADS[args.workload]
    # a line is labelled with a BASELINE configuration only when every size-defining argument is that configuration's
    args.as_baseline = all(v is None for v in (args.batch_per_gpu, args.img, args.queue, args.config))
    args.batch_per_gpu = wl["batch"] if args.batch_per_gpu is None else args.batch_per_gpu
    args.img = wl["img"] if args.img is None else args.img
    args.queue = wl["queue"] if args.queue is None else args.queue
    args.config = os.path.join(ROOT, "configs", wl["config"]) if args.config is None else args.config
    cb, cs, cw = wl["cpu"]
    args.cpu_batch = cb if args.cpu_batch is None else args.cpu_batch
    args.cpu_steps = cs if args.cpu_steps is None else args.cpu_steps
    args.cpu_warmup = cw if args.cpu_warmup is None else args.cpu_warmup
    args.densecl = wl["densecl"]
    return args


def _free_port() -> int:
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def launch_ranks(n: int) -> int:
    """Start one child process per GPU (this process touches no GPU: reference main.py:732 spawns its ranks the same way)
    with the launcher environment torch.distributed.run would give them; rank 0's stdout is this process's stdout."""
    port = _free_port()
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), CP2_BENCH_LAUNCHER="self")
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env,
                                      stdout=None if r == 0 else subprocess.DEVNULL))
    rc = 0
    try:
        pending = set(range(n))
        while pending:
            for r in sorted(pending):
                code = procs[r].poll()
                if code is None:
                    continue
                pending.discard(r)
                if code != 0 and rc == 0:
                    rc = code
                    print(f"bench.py: rank {r} exited with code {code}; stopping the other ranks", file=sys.stderr)
                    for o in pending:                          # a dead peer leaves the others in a collective forever
                        procs[o].terminate()
            time.sleep(0.2)
    finally:
        for p in procs:
            if p.poll() is None:
                p.kill()
    return rc


def physical_cores() -> int:
    """Physical cores this process may run on (distinct (package, core) pairs among its allowed CPUs)."""
    try:
        allowed = os.sched_getaffinity(0)
        seen = set()
        for cpu in allowed:
            base = f"/sys/devices/system/cpu/cpu{cpu}/topology/"
            with open(base + "physical_package_id") as f:
                pkg = f.read().strip()
            with open(base + "core_id") as f:
                seen.add((pkg, f.read().strip()))
        return len(seen) or len(allowed)
    except OSError:
        return os.cpu_count() or 1


def count_flops_per_image(torch, model, batch, densecl=False):
    """FLOPs of one step (q forward+backward, k forward, loss GEMMs excluded) via torch's flop counter.  The counter only
    sees ATen operators, so the pass runs with every convolution on the ATen path (the product path sends the 1x1 and
    k x k weight gradients through cp2_wgrad1x1 / cp2_wgrad_conv, which it would not count)."""
    from torch.utils.flop_counter import FlopCounterMode
    from cp2_amd.encoder import Conv2d
    saved = (Conv2d.cpp_nodes, Conv2d.gemm_1x1, Conv2d.hip_wgrad_kxk)
    Conv2d.cpp_nodes, Conv2d.gemm_1x1, Conv2d.hip_wgrad_kxk = False, False, False
    def fwd(enc, img):
        if not densecl:
            return enc(img).float().mean()
        with torch.autocast("cuda", dtype=torch.bfloat16):    # backbone and neck under autocast, as in the step
            out = enc.neck(enc.backbone(img)[3])
        return out["x_global_proj"].float().mean() + out["x_local_proj"].float().mean()
    try:
        with FlopCounterMode(display=False) as fc:
            with torch.autocast("cuda", dtype=torch.bfloat16, enabled=not densecl):
                y = fwd(model.encoder_q, batch["img_a"])
                with torch.no_grad():
                    fwd(model.encoder_k, batch["img_b"])
            y.backward()
    finally:
        Conv2d.cpp_nodes, Conv2d.gemm_1x1, Conv2d.hip_wgrad_kxk = saved
    model.encoder_q.zero_grad(set_to_none=True)
    return fc.get_total_flops() / batch["img_a"].shape[0]


def _traffic(name, algorithmic_or_fused_bytes):
    """PMC result of the same kernel on the same buffer size (tools/ema_only.py / tools/sgd_only.py under
    rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, separate passes); None when the committed file is for another size."""
    path = os.path.join(ROOT, "profiles", name)
    try:
        tj = json.load(open(path))
        if int(tj.get("algorithmic_bytes_per_launch", -1)) == int(algorithmic_or_fused_bytes):
            return tj.get("hbm_bytes_per_launch"), f"profiles/{name} (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes)"
    except Exception:
        pass
    return None, None


def main():
    args = parse()
    if args.gpus > 1 and "RANK" not in os.environ:
        # no launcher: start the ranks ourselves, BEFORE anything touches the GPU (never fork / exec after HIP is up)
        raise SystemExit(launch_ranks(args.gpus))

    # stdout carries ONE line, the JSON: everything else that writes to file descriptor 1 (the constructor's prints, but also
    # C++ code such as gloo's "[Gloo] Rank 0 is connected ..." banner) is sent to stderr for the duration of the run
    sys.stdout.flush()
    json_fd = os.dup(1)
    os.dup2(2, 1)

    import torch
    import torch.distributed as dist

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"WORLD_SIZE={world} but --gpus {args.gpus}")
    assert torch.cuda.is_available(), "bench.py needs a GPU"
    if args.one_device:
        local = 0
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    rehearse = args.rehearse_collectives and world == 1
    from cp2_amd import dist as cdist
    if world > 1:
        # bounded timeout (torch's nccl default, 10 min, is the driver's whole bench limit) + the hang watchdog that names the
        # exchange step (C1 / C3 / C4 / gradient bucket i) a stuck rank was waiting for, exit code cdist.HANG_EXIT_CODE
        cdist.init_process_group(args.backend, rank, world, timeout_s=args.timeout)
    elif rehearse:              # a one-rank group: every exchange step of the N > 1 path is issued (dist.multi()), nothing moves
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", str(_free_port()))
        cdist.init_process_group(args.backend, 0, 1, timeout_s=args.timeout)

    from cp2_amd import builder, ops, synthetic
    from cp2_amd.encoder import FusedBatchNorm2d
    FusedBatchNorm2d.fused = args.fused_bn == "on"
    from cp2_amd.encoder import StemMaxPool
    StemMaxPool.fused = args.fused_bn == "on" and os.environ.get("CP2_STEM_POOL", "1") == "1"   # same encoder fast path (env: A/B)
    from cp2_amd.encoder import Conv2d
    Conv2d.gemm_1x1 = args.gemm_1x1 == "on"
    Conv2d.cpp_nodes = args.cpp_nodes == "on"
    Conv2d.hip_wgrad_kxk = os.environ.get("CP2_WGRAD_KXK", "1") == "1"     # k x k weight gradients by cp2_wgrad_conv (env: A/B vs MIOpen)
    from cp2_amd.config import Config
    from cp2_amd.engine import TrainStep
    from cp2_amd.main import make_optimizer
    from cp2_amd.pretrain_types import PretrainType

    torch.manual_seed(0)
    torch.backends.cudnn.benchmark = True
    cfg = Config.fromfile(args.config)
    amp = torch.bfloat16 if args.amp == "bf16" else None
    import contextlib
    ptype = PretrainType.DENSECL if args.densecl else PretrainType.CP2
    # DENSECL: the flag overrides of reference main.py:148-153 / scripts/10-11-densecl.sh:48-50 (T = 0.2 twice, lambda = 0.5)
    extra = dict(instance_logits_temp=0.2, dense_logits_temp=0.2, lmbd_cp2_dense_loss=0.5) if args.densecl else {}
    with contextlib.redirect_stdout(sys.stderr):       # the constructor prints what the reference prints; stdout = the JSON line only
        model = builder.MODEL(cfg, rank=rank, K=args.queue, pretrain_from_scratch=True, pretrain_type=ptype,
                              device=dev, amp_dtype=amp, channels_last=True, **extra).to(dev)
    model.encoder_q.to(memory_format=torch.channels_last)
    model.encoder_k.to(memory_format=torch.channels_last)
    model.train()
    model.shuffle_exchange = args.shuffle_exchange
    wrapped = model
    if (world > 1 or rehearse) and args.grad_sync == "flat":
        from cp2_amd.ddp import FlatDDP
        wrapped = FlatDDP(model, bucket_mb=args.bucket_mb or builder.DDP_BUCKET_MB)
    elif world > 1 or rehearse:
        wrapped = torch.nn.parallel.DistributedDataParallel(model, device_ids=[local], output_device=local,
                                                            broadcast_buffers=False, gradient_as_bucket_view=True,
                                                            bucket_cap_mb=args.bucket_mb or builder.DDP_BUCKET_MB)

    class A:  # optimizer settings of reference main.py defaults
        lr, momentum, weight_decay, optim = 0.03, 0.9, 1e-4, "sgd"
    opt = make_optimizer(list(model.parameters()), A, dev, capturable=False,
                         model=model if args.flat_sgd == "on" else None)
    b, hw = args.batch_per_gpu, args.img
    batches = [synthetic.make_batch(b, hw, hw, dev, seed=rank * 9973 + i) for i in range(4)]
    flops_img = count_flops_per_image(torch, model, batches[0], args.densecl)
    runner = TrainStep(wrapped, opt)
    if os.environ.get("CP2_BENCH_QUART", "1") != "1":
        model.log_quartiles = False

    # the EMA is hoisted in front of the rest of the step so each of its launches can be bracketed by HIP events on the
    # launch stream; it reads theta_q after the previous optimizer step and runs before the key encoder, exactly where
    # the reference's call does (builder.py:1272).
    model.overlap_key_branch = {"auto": None, "gather": "gather", "off": False}[args.overlap]
    model.ema_in_forward = False
    ema_events = []

    from cp2_amd.hipevents import EventPair
    model.flatten_parameters()

    def one_step(i, timed):
        if timed:
            ev = EventPair()                                  # hipEvent_t pair attached to the kernel launch itself
            if model._flat_k_bf16 is not None:
                ops.ema_flat_shadow(model._flat_k, model._flat_q, model._flat_k_bf16, model.momentum, ev)
            else:
                ops.ema_flat_timed(model._flat_k, model._flat_q, model.momentum, ev)
            ema_events.append(ev)
        else:
            model._momentum_update_key_encoder()
        return runner(batches[i % len(batches)])

    host_issue, local_dt = [None], [None]

    def timed_region(n_steps, timed_kernels):
        if world > 1:
            cdist.barrier("bench: barrier in front of the timed region")
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        last = None
        for i in range(n_steps):
            cdist.progress(i)
            last = one_step(i, timed_kernels)
        host_issue[0] = (time.perf_counter() - t0) / n_steps * 1e3      # host time to ENQUEUE a step (no device wait inside)
        torch.cuda.synchronize()
        if world > 1:
            cdist.barrier("bench: barrier behind the timed region")
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        local_dt[0] = dt
        if world > 1:
            t = torch.tensor([dt], device=dev, dtype=torch.float64)
            cdist.tracked("bench: max over the ranks of the region's time", dist.all_reduce(t, op=dist.ReduceOp.MAX, async_op=True))
            dt = float(t)
        return dt, last

    for i in range(args.warmup):
        cdist.progress(i)
        one_step(i, False)
    torch.cuda.synchronize()
    if args.warmup >= 2:
        cdist.steady()       # every kernel has run on this rank: from here on a collective pending for --timeout seconds is a hang
    # Settle (untimed, after the W warm-up steps): with the solver rankings shipped a process reaches this point 20 s after it
    # started, and as the FIRST process on a cold box its next steps measured 2 % slow (2498 / 2504 img/s against 2550-2560 for
    # every later process; a process that spends its first minute in MIOpen's search does not show it).  So: chunks of 10
    # steps until two consecutive chunks agree within 0.5 %, at most --settle-seconds of them.  The decision uses the
    # max-over-ranks time, so every rank runs the same number of steps.
    settle = {"chunks": 0, "ms_per_step": []}
    if args.settle_seconds > 0:
        t_settle, prev = time.perf_counter(), None
        while time.perf_counter() - t_settle < args.settle_seconds:
            dt_s, _ = timed_region(10, False)
            cur = dt_s / 10 * 1e3
            settle["chunks"] += 1
            settle["ms_per_step"].append(round(cur, 3))
            if prev is not None and abs(cur - prev) <= 0.005 * cur:
                break
            prev = cur
        settle["ms_per_step"] = settle["ms_per_step"][-4:]
    # --overlap auto with peers: north_star asks for the key branch's exchange steps on a side HIP stream; with ONE rank the
    # fork / join measured slower than the ~0.14 ms it hides -- but a one-rank rehearsal moves no bytes.
    # So with real peers both forms are timed here, after the warm-up, and the faster one (MAX over the ranks) is used.
    calib = None
    if args.overlap == "auto" and world > 1 and not args.densecl:
        k = max(5, args.calib_steps)
        calib = {}
        for name, mode in (("off", None), ("gather", "gather")):
            model.overlap_key_branch = mode
            for i in range(2):                               # the side stream's first use allocates; not timed
                one_step(i, False)
            dt_c, _ = timed_region(k, False)
            calib[name] = round(dt_c / k * 1e3, 4)
        choice = min(calib, key=calib.get)                   # dt is already the MAX over the ranks: every rank picks the same
        model.overlap_key_branch = {"off": None, "gather": "gather"}[choice]
        calib = {"steps_each": k, "ms_per_step": calib, "chosen": choice}
    ops.PROFILE = {}               # every profiled launch of the timed steps carries its own start/stop hipEvents
    model.comm_events = {} if (world > 1 or rehearse) else None
    dt, loss = timed_region(args.steps, True)
    host_issue_ms = host_issue[0]
    per_rank = None
    if world > 1:              # every rank's own step time and host issue time: a straggling or host-bound rank is visible
        mine = torch.tensor([local_dt[0] / args.steps * 1e3, host_issue_ms], device=dev, dtype=torch.float64)
        allr = [torch.empty_like(mine) for _ in range(world)]
        dist.all_gather(allr, mine)
        ms_r, host_r = [float(t[0]) for t in allr], [float(t[1]) for t in allr]
        per_rank = {"ms_per_step": {"min": round(min(ms_r), 3), "max": round(max(ms_r), 3), "by_rank": [round(v, 3) for v in ms_r]},
                    "host_issue_ms_per_step": {"min": round(min(host_r), 3), "max": round(max(host_r), 3),
                                               "by_rank": [round(v, 3) for v in host_r]}}
    loss_val = float(loss)
    assert loss_val == loss_val, "loss is NaN"
    prof, ops.PROFILE = ops.PROFILE or {}, None
    comm_events, model.comm_events = model.comm_events, None

    comm = None
    if world > 1 or rehearse:
        def ev_ms(name):
            evs = comm_events.get(name) or []
            return round(sum(a.elapsed_time(z) for a, z in evs) / args.steps, 4) if evs else None
        nosync_ms = exposed = None
        if args.nosync_steps > 0:
            with wrapped.no_sync():                            # same step without the gradient all-reduce (replicas diverge:
                dt_ns, _ = timed_region(args.nosync_steps, False)   # last thing this process does with the model)
            nosync_ms = dt_ns / args.nosync_steps * 1e3
            exposed = round(dt / args.steps * 1e3 - nosync_ms, 4)
        n_grad = sum(p.numel() for p in model.parameters() if p.requires_grad)
        # composed images travel as cp2_compose_pair wrote them: bf16 under bf16 autocast (W % 4 == 0), else fp32
        img_row, key_row = 3 * hw * hw * (2 if (amp is not None and hw % 4 == 0) else 4), 128 * (hw // model.output_stride) ** 2 * 4
        frac = (world - 1) / world
        comm = {
            "backend": dist.get_backend(), "rccl_ranks": dist.get_world_size(), "one_device_rehearsal": bool(args.one_device),
            "single_rank_rehearsal": bool(rehearse),
            "launcher": os.environ.get("CP2_BENCH_LAUNCHER", "external"), "shuffle_exchange": args.shuffle_exchange,
            "overlap_key_branch": args.overlap, "overlap_used": "gather" if model.overlap_key_branch == "gather" else "off",
            "overlap_calibration": calib, "per_rank": per_rank, "timeout_s": args.timeout, "grad_sync": args.grad_sync,
            "ddp_bucket_mb": args.bucket_mb or builder.DDP_BUCKET_MB,
            "grad_buckets": len(wrapped.reducer.buckets) if args.grad_sync == "flat" else None,
            "ms_per_step": {   # stream time between the events around each exchange step (it includes waiting for peers)
                "c1_image_exchange": ev_ms("c1_image_exchange"), "c3_key_unshuffle": ev_ms("c3_key_unshuffle"),
                "c4_key_gather_enqueue": ev_ms("c4_key_gather_enqueue"),
                "key_branch_wait_exposed": ev_ms("key_branch_wait_exposed"),
                "step_without_grad_allreduce": None if nosync_ms is None else round(nosync_ms, 4),
                "ddp_allreduce_exposed": exposed},
            "bytes_received_per_rank_per_step": {
                "c1_image_exchange": int((frac if args.shuffle_exchange == "all_to_all" else world - 1) * b * img_row),
                "c3_key_unshuffle": int((frac if args.shuffle_exchange == "all_to_all" else world - 1) * b * key_row),
                "c4_key_gather": (world - 1) * b * 128 * 4,
                "c5_grad_allreduce_ring": int(2 * frac * 4 * n_grad)},
        }

    ema_ms = sum(ev.elapsed_ms() for ev in ema_events) / len(ema_events)
    n_slots = model._flat_q.numel()

    def avg_ms(name):
        evs = prof.get(name) or []
        return sum(e.elapsed_ms() for e in evs) / len(evs) if evs else None

    # the optimizer kernel skips the slots of frozen parameters (no gradient pointer: the decode head on the DenseCL path,
    # conv_seg on the CP2 path); the EMA walks the whole flat buffer
    sgd_slots = sum((p.numel() + 63) // 64 * 64 for p in model.encoder_q.parameters() if p.requires_grad)

    def hbm_entry(kernel, ms, alg_per_slot, fused_per_slot, traffic_file, what, n_slots=n_slots):
        """achieved = ALGORITHMIC bytes / kernel time; the fused figure counts the bf16 weight image the same
        pass also writes (2 B per slot), i.e. what the launch really moves."""
        alg, fused = alg_per_slot * n_slots, fused_per_slot * n_slots
        traffic, source = _traffic(traffic_file, fused)
        ach = alg / (ms * 1e-3) / 1e9
        return {"kernel": kernel, "bound": "hbm", "achieved": round(ach, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": round(ach / HBM_PEAK_GBS, 4), "traffic": traffic, "traffic_source": source,
                "bytes_algorithmic": alg, "bytes_per_slot_algorithmic": alg_per_slot,
                "bytes_fused": fused, "bytes_per_slot_fused": fused_per_slot,
                "achieved_fused": round(fused / (ms * 1e-3) / 1e9, 1), "frac_fused": round(fused / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
                "avg_launch_ms": round(ms, 4), "parameter_slots": n_slots, "bytes_note": what}

    shadow = model._flat_k_bf16 is not None
    ema_entry = hbm_entry("ema_flat_kernel (momentum update of the key encoder, builder.py:557-567)", ema_ms, 12,
                          14 if shadow else 12, "ema_traffic.json",
                          "algorithmic: read k, read q, write k in fp32 (SURVEY 8d: 3*4*N); fused: + the bf16 image of the new key weights")
    sgd_ms = avg_ms("sgd_flat")
    sgd_entry = None
    if sgd_ms:
        qshadow = model._flat_q_bf16 is not None
        sgd_entry = hbm_entry("sgd_flat_kernel (SGD momentum + weight decay, main.py:467-477,640-642)", sgd_ms, 20,
                              22 if qshadow else 20, "sgd_traffic.json",
                              "algorithmic: read p, g, momentum; write p, momentum in fp32; fused: + the bf16 image of the new query weights"
                              + ("" if sgd_slots == n_slots else f"; {n_slots - sgd_slots} slots of frozen parameters are skipped"),
                              n_slots=sgd_slots)

    # per-kernel figures of the other hand-written kernels of the step, each from its own launches' events
    C, K = 128, args.queue
    P = (hw // (model.backbone_output_stride if args.densecl else model.output_stride)) ** 2
    kernels = []

    def entry(name, kernel, bound, work, unit, peak, note, **more):
        """`work` = algorithmic bytes / flops of ONE launch; several launches per step (the chunked statistics walk) are
        averaged, so achieved = work / mean launch duration either way."""
        ms = avg_ms(name)
        if not ms:
            return None
        ach = work / (ms * 1e-3) / (1e9 if unit == "GB/s" else 1e12)
        e = {"kernel": kernel, "bound": bound, "achieved": round(ach, 1), "peak": peak, "unit": unit, "frac": round(ach / peak, 4),
             "traffic": None, "avg_launch_us": round(ms * 1e3, 2), "launches_per_step": round(len(prof.get(name) or []) / args.steps, 2),
             "work_per_launch": work, "note": note}
        e.update(more)
        return e

    def add(*a, **kw):
        e = entry(*a, **kw)
        if e is not None:
            kernels.append(e)
        return e
    px_bytes = 4 * 3 * 4 + 2 * 3 * (2 if amp is not None else 4)      # per pixel: img_a, bg0, img_b, bg1 read in fp32; two views written
    add("compose_pair", "compose_pair_kernel (copy-paste composition of both views, builder.py:1146-1159, key rows in shuffle order)", "hbm",
        b * hw * hw * px_bytes, "GB/s", HBM_PEAK_GBS, "algorithmic bytes = 4 fp32 image reads + 2 composed views written (bf16 under autocast); "
        "PMC traffic 1.002 x (profiles/compose_traffic.json)")
    add("rowkey_fwd", "rowkey_small_kernel (instance InfoNCE: q_pos x queue, builder.py:1395-1428; DenseCL: the global loss, :762-772)", "hbm",
        4 * C * K, "GB/s", HBM_PEAK_GBS,
        "algorithmic bytes = the fp32 queue read once (4*C*K); with quartile logging on the launch also writes the b x K logits")
    dense_fwd = add("dense_fwd", "dense_fwd_kernel (P x P logits + column soft-max statistics, builder.py:1289-1292,1431-1437)", "mfma",
                    2.0 * b * P * P * C, "TFLOP/s", F32_MFMA_PEAK_TFLOPS,
                    "f32-input MFMA; at P=196 the launch is latency-bound (64 (sample, tile) items)")
    dense_bwd = add("dense_bwd", "dense_bwd_kernel (recomputed P x P logits + gradient product, builder.py:1289-1292,1431-1437 backward)", "mfma",
                    4.0 * b * P * P * C, "TFLOP/s", F32_MFMA_PEAK_TFLOPS, "f32-input MFMA (exact fp32 fma chains), two products per pixel pair")
    # DenseCL: the rows-vs-queue kernel (T19).  One launch covers R_l rows: all b*P of them, or -- on rank 0 with the score
    # statistics on -- one group of whole samples of the chunked walk (builder._queue_infonce_chunked)
    rows_evs = prof.get("rowkey_fwd_rows") or []
    rowkey_rows = None
    if rows_evs:
        R_l = b * P / (len(rows_evs) / args.steps)
        useful = 4.0 * R_l * C * K                       # logits (2 R C K) + the gradient product sum_j p_j k_j (2 R C K)
        if b * P >= 1024:           # what precision "auto" selects for the DenseCL row counts (ops.rowkey_infonce)
            rowkey_rows = add("rowkey_fwd_rows", "rowkey_bf16x3_dma_kernel (DenseCL per-pixel rows x queue2, forward + fused gradient product, "
                              "builder.py:866-873,906-908,150-176)", "mfma", useful, "TFLOP/s", BF16_DENSE_PEAK_TFLOPS,
                              "achieved = USEFUL flop rate (2 products of 2*R*C*K: logits and gradient); split-bf16 (hi*hi + hi*lo + lo*hi) issues "
                              "three bf16 MFMAs per product for logits within 3e-5 of fp32: the matrix pipe runs at achieved_issued",
                              rows_per_launch=int(R_l))
            if rowkey_rows is not None:
                rowkey_rows["achieved_issued"] = round(3 * rowkey_rows["achieved"], 1)
                rowkey_rows["frac_issued"] = round(3 * rowkey_rows["achieved"] / BF16_DENSE_PEAK_TFLOPS, 4)
        else:
            rowkey_rows = add("rowkey_fwd_rows", "rowkey_fwd_kernel (rows x queue in exact fp32, forward + fused gradient product, "
                              "builder.py:866-873,906-908,150-176)", "mfma", useful, "TFLOP/s", F32_MFMA_PEAK_TFLOPS,
                              "f32-input MFMA (fewer than 1024 rows: the exact-fp32 kernel)", rows_per_launch=int(R_l))
    CE = getattr(model.encoder_q.backbone, "feat_dim", 2048)
    add("densecl_match", "densecl_match_kernel (DenseCL positive selection: backbone-similarity arg-max + local positives, "
        "builder.py:818-864)", "mfma", 2.0 * b * P * P * CE, "TFLOP/s", BF16_DENSE_PEAK_TFLOPS,
        "bf16 MFMA on the backbone features as the encoder returns them (exact products, fp32 accumulation); at P=196 the launch is "
        "bound by streaming the two 25.7 MB feature maps and by its 224 work items, not by the matrix pipe")
    add("quantiles", "quantiles kernels (the logging quartiles of the step)" if args.densecl or K > 131072 or P * P > 131072 else
        "step_post_kernel (the step's quartile rows; the instance-loss finalize and the dense post-pass ride in the same launch)", "hbm",
        4 * ((b * K + 2 * b * P * P) if not args.densecl else (b * P * K + b * K) / max(1.0, len(prof.get("quantiles") or [1]) / args.steps)),
        "GB/s", HBM_PEAK_GBS, "radix select over rows kept in registers: latency / LDS-atomic bound, not a streaming kernel")
    if sgd_entry is not None:
        kernels.insert(0, ema_entry)
    # the line's `roofline` = the dominant hand-written kernel of THIS workload: cfg2 the optimizer update (HBM), cfg4 the
    # dense backward (f32 MFMA), cfg5 the rows-vs-queue kernel (bf16 MFMA)
    roofline = sgd_entry if sgd_entry is not None else ema_entry
    if args.workload == "cfg4" and dense_bwd is not None:
        roofline = dense_bwd
        kernels = [k for k in kernels if k is not dense_bwd] + ([sgd_entry] if sgd_entry is not None else [])
    elif args.workload == "cfg5" and rowkey_rows is not None:
        roofline = rowkey_rows
        kernels = [k for k in kernels if k is not rowkey_rows] + ([sgd_entry] if sgd_entry is not None else [])
    imgs = b * world * args.steps
    value = imgs / dt
    depth = cfg.model["backbone"]["depth"]
    head = cfg.model["decode_head"]
    head_name = {"FCNHead": "FCN", "ASPPHead": "ASPP"}.get(head["type"], head["type"]) + ("(contrast)" if head.get("contrast") else "")
    label = WORKLOADS[args.workload]["label"] if args.as_baseline else "custom sizes (not a BASELINE configuration)"
    common = (f"{b} img/GPU, encoders bf16 autocast channels-last, " if amp is not None else f"{b} img/GPU, encoders fp32 channels-last, ")
    if args.densecl:
        method = "DenseCL"
        workload = (f"{label}: DenseCL (scripts/10-11-densecl.sh) ResNet-{depth} backbone OS{model.backbone_output_stride} + DenseCL neck, "
                    f"{hw}x{hw} crops, {P} pixels per image against queue2, queues 2 x {args.queue}, {common}"
                    f"{'neck fp32, ' if amp is None or not model.neck_autocast else ''}loss kernels split-bf16 / fp32 (logits within 3e-5), SGD(0.9, wd 1e-4), random-init weights")
    else:
        method = "CP2"
        workload = (f"{label}: ResNet-{depth} + {head_name} head OS{model.output_stride}, {hw}x{hw} copy-paste pairs, "
                    f"queue={args.queue}, {common}loss kernels fp32 (f32 MFMA), "
                    f"SGD(0.9, wd 1e-4), random-init weights")
    out = {
        "metric": f"pretrain images/sec (whole node), ResNet-{depth} {method} {hw}^2, queue={args.queue}",
        "value": round(value, 2), "unit": "images/sec", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": round(dt / args.steps * 1e3, 3), "host_issue_ms_per_step": round(host_issue_ms, 3),
        "higher_is_better": True, "scaling": "weak",
        "vs_baseline": None, "dtype": "bf16" if amp is not None else "f32", "data": "synthetic",
        "config": {"workload": workload,
                   "global_batch": b * world, "parallelism": f"dp{world}", "hipgraph": "key encoder forward only",
                   "miopen_find_db": "shipped rankings (cp2_amd/miopen_db) + search of anything new" if MIOPEN_DB else "searched at start-up",
                   "settle": settle,
                   "final_loss": round(loss_val, 4)},
        # the dominant hand-written kernel of the step by time (cfg2: the optimizer update, then the EMA, first entry of roofline_kernels)
        "roofline": roofline,
        "roofline_kernels": kernels,
        "step_compute": {"flops_per_img": round(flops_img / 1e9, 2), "unit": "GFLOP (encoders fwd+bwd, flop counter)",
                         "achieved_tflops_per_gpu": round(flops_img * b / (dt / args.steps) / 1e12, 1),
                         "frac_of_bf16_dense_peak": round(flops_img * b / (dt / args.steps) / 1e12 / BF16_DENSE_PEAK_TFLOPS, 4)},
    }
    if comm is not None:
        out["comm"] = comm
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        from oracle.cpu_step import time_cpu_baseline
        ips, threads, secs = time_cpu_baseline(cfg, synthetic.make_batch, args.cpu_batch, hw, hw, args.queue,
                                               steps=args.cpu_steps, warmup=args.cpu_warmup, densecl=args.densecl,
                                               output_stride=model.output_stride)
        out["cpu_baseline"] = {"value": round(ips, 3), "unit": "images/sec", "cores": physical_cores(), "threads": threads,
                               "kind": "port",
                               "sample": f"{args.cpu_steps} steps of {args.cpu_batch} images ({hw}x{hw}, queue {args.queue}, "
                                         f"same model{', DenseCL step with the rank-0 score statistics' if args.densecl else ''}) "
                                         f"after {args.cpu_warmup} warm-up steps, fp32, {secs:.1f} s",
                               "split_ms_per_step": time_cpu_baseline.last_split_ms}
    if rank == 0:
        sys.stdout.flush()
        os.write(json_fd, (json.dumps(out) + "\n").encode())
    if world > 1 or rehearse:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
