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

HBM_PEAK_GBS = 8000.0          # MI355X HBM3E peak, /opt/skills/guides/MI355X_MICROARCH.md "Chip-level parameters"
BF16_DENSE_PEAK_TFLOPS = 2500.0
F32_MFMA_PEAK_TFLOPS = 157.3   # v_mfma_f32_32x32x2_f32: the f32 vector rate (same guide, "Matrix cores")


def parse():
    p = argparse.ArgumentParser()
    p.add_argument("--gpus", type=int, default=1)
    p.add_argument("--steps", type=int, default=50)
    p.add_argument("--warmup", type=int, default=10)
    p.add_argument("--batch-per-gpu", type=int, default=32)
    p.add_argument("--img", type=int, default=224)
    p.add_argument("--queue", type=int, default=65536)
    p.add_argument("--config", default=os.path.join(ROOT, "configs", "config_pretrain_r50_fcn.py"))
    p.add_argument("--amp", default="bf16", choices=["bf16", "none"])
    p.add_argument("--no-cpu-baseline", action="store_true")
    p.add_argument("--gemm-1x1", default="on", choices=["on", "off"],
                   help="1x1 stride-1 convolutions: hipBLASLt GEMM for forward / data gradient where faster (A/B)")
    p.add_argument("--cpp-nodes", default="on", choices=["on", "off"],
                   help="C++ autograd nodes for the weight images / 1x1 convolutions (off: the Python nodes) (A/B)")
    p.add_argument("--flat-sgd", default="on", choices=["on", "off"],
                   help="optimizer step as one HIP launch on the flat parameter buffer (off: torch.optim.SGD) (A/B)")
    p.add_argument("--fused-bn", default="on", choices=["on", "off"], help="encoder fast path: fused BN(+add)(+ReLU) kernels (A/B)")
    p.add_argument("--overlap", default="auto", choices=["auto", "gather", "on", "off"],
                   help="side HIP stream for the key branch: off = everything in order on one stream (auto: measured fastest, also "
                        "with the collectives of N > 1), gather = EMA + shuffle exchange + key gather on a side stream, on = the key "
                        "encoder too")
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
    p.add_argument("--cpu-batch", type=int, default=32, help="CPU baseline: images per step (BASELINE.md section 3: the same b)")
    p.add_argument("--cpu-steps", type=int, default=5, help="CPU baseline: timed steps (after --cpu-warmup)")
    p.add_argument("--cpu-warmup", type=int, default=2)
    return p.parse_args()


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


def count_flops_per_image(torch, model, batch):
    """FLOPs of one step (q forward+backward, k forward, loss GEMMs excluded) via torch's flop counter.  The counter only
    sees ATen operators, so the pass runs with every convolution on the ATen path (the product path sends the 1x1 and
    k x k weight gradients through cp2_wgrad1x1 / cp2_wgrad_conv, which it would not count)."""
    from torch.utils.flop_counter import FlopCounterMode
    from cp2_amd.encoder import Conv2d
    saved = (Conv2d.cpp_nodes, Conv2d.gemm_1x1, Conv2d.hip_wgrad_kxk)
    Conv2d.cpp_nodes, Conv2d.gemm_1x1, Conv2d.hip_wgrad_kxk = False, False, False
    try:
        with FlopCounterMode(display=False) as fc:
            with torch.autocast("cuda", dtype=torch.bfloat16):
                y = model.encoder_q(batch["img_a"])
                with torch.no_grad():
                    model.encoder_k(batch["img_b"])
            y.float().mean().backward()
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
    if world > 1:
        dist.init_process_group(args.backend, rank=rank, world_size=world)
    elif rehearse:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", str(_free_port()))
        dist.init_process_group(args.backend, rank=0, world_size=1)
        from cp2_amd import dist as cdist
        cdist.FORCE_COLLECTIVES = True

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
    with contextlib.redirect_stdout(sys.stderr):       # the constructor prints what the reference prints; stdout = the JSON line only
        model = builder.MODEL(cfg, rank=rank, K=args.queue, pretrain_from_scratch=True, pretrain_type=PretrainType.CP2,
                              device=dev, amp_dtype=amp, channels_last=True).to(dev)
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
    flops_img = count_flops_per_image(torch, model, batches[0])
    runner = TrainStep(wrapped, opt)
    if os.environ.get("CP2_BENCH_QUART", "1") != "1":
        model.log_quartiles = False

    # the EMA is hoisted in front of the rest of the step so each of its launches can be bracketed by HIP events on the
    # launch stream; it reads theta_q after the previous optimizer step and runs before the key encoder, exactly where
    # the reference's call does (builder.py:1272).
    model.overlap_key_branch = {"auto": None, "gather": "gather", "on": True, "off": False}[args.overlap]
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

    host_issue = [None]

    def timed_region(n_steps, timed_kernels):
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        last = None
        for i in range(n_steps):
            last = one_step(i, timed_kernels)
        host_issue[0] = (time.perf_counter() - t0) / n_steps * 1e3      # host time to ENQUEUE a step (no device wait inside)
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        if world > 1:
            t = torch.tensor([dt], device=dev, dtype=torch.float64)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            dt = float(t)
        return dt, last

    for i in range(args.warmup):
        one_step(i, False)
    torch.cuda.synchronize()
    ops.PROFILE = {}               # every profiled launch of the timed steps carries its own start/stop hipEvents
    model.comm_events = {} if (world > 1 or rehearse) else None
    dt, loss = timed_region(args.steps, True)
    host_issue_ms = host_issue[0]
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
            "overlap_key_branch": args.overlap, "grad_sync": args.grad_sync,
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

    def hbm_entry(kernel, ms, alg_per_slot, fused_per_slot, traffic_file, what):
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
                              "algorithmic: read p, g, momentum; write p, momentum in fp32; fused: + the bf16 image of the new query weights")

    # per-kernel figures of the other hand-written kernels of the step, each from its own launches' events
    C, K, P = 128, args.queue, (hw // model.output_stride) ** 2
    kernels = []

    def add(name, kernel, bound, work, unit, peak, note):
        ms = avg_ms(name)
        if ms:
            ach = work / (ms * 1e-3) / (1e9 if unit == "GB/s" else 1e12)
            kernels.append({"kernel": kernel, "bound": bound, "achieved": round(ach, 1), "peak": peak, "unit": unit,
                            "frac": round(ach / peak, 4), "avg_launch_us": round(ms * 1e3, 2), "work_per_launch": work, "note": note})
    px_bytes = 4 * 3 * 4 + 2 * 3 * (2 if amp is not None else 4)      # per pixel: img_a, bg0, img_b, bg1 read in fp32; two views written
    add("compose_pair", "compose_pair_kernel (copy-paste composition of both views, builder.py:1146-1159, key rows in shuffle order)", "hbm",
        b * hw * hw * px_bytes, "GB/s", HBM_PEAK_GBS, "algorithmic bytes = 4 fp32 image reads + 2 composed views written (bf16 under autocast); "
        "PMC traffic 1.002 x (profiles/compose_traffic.json)")
    add("rowkey_fwd", "rowkey_small_kernel (instance InfoNCE: q_pos x queue, builder.py:1395-1428)", "hbm", 4 * C * K, "GB/s",
        HBM_PEAK_GBS, "algorithmic bytes = the fp32 queue read once (4*C*K); with quartile logging on the launch also writes the b x K logits")
    add("dense_fwd", "dense_fwd_kernel (P x P logits + column soft-max statistics, builder.py:1289-1292,1431-1437)", "mfma", 2.0 * b * P * P * C,
        "TFLOP/s", F32_MFMA_PEAK_TFLOPS, "f32-input MFMA; at P=196 the launch is latency-bound (64 (sample, tile) items)")
    add("dense_bwd", "dense_bwd_kernel (recomputed logits + gradient product)", "mfma", 4.0 * b * P * P * C, "TFLOP/s", F32_MFMA_PEAK_TFLOPS,
        "f32-input MFMA, two products per pair")
    add("quantiles", "quantiles kernels (3 x 3 logging quartiles of the step)", "hbm", 4 * (b * K + 2 * b * P * P), "GB/s",
        HBM_PEAK_GBS, "radix select, three passes over the rows: latency / LDS-atomic bound, not a streaming kernel")
    if sgd_entry is not None:
        kernels.insert(0, ema_entry)
    imgs = b * world * args.steps
    value = imgs / dt
    out = {
        "metric": "pretrain images/sec (whole node), ResNet-50 CP2 224^2, queue=65536",
        "value": round(value, 2), "unit": "images/sec", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": round(dt / args.steps * 1e3, 3), "host_issue_ms_per_step": round(host_issue_ms, 3),
        "higher_is_better": True, "scaling": "weak",
        "vs_baseline": None, "dtype": "bf16" if amp is not None else "f32", "data": "synthetic",
        "config": {"workload": f"BASELINE configs[1]: ResNet-50 + FCN(contrast) head OS16, {hw}x{hw} copy-paste pairs, "
                               f"queue={args.queue}, {b} img/GPU, encoders bf16 autocast channels-last, loss kernels fp32 (f32 MFMA), "
                               f"SGD(0.9, wd 1e-4), random-init weights",
                   "global_batch": b * world, "parallelism": f"dp{world}", "hipgraph": "key encoder forward only",
                   "final_loss": round(loss_val, 4)},
        # the dominant hand-written kernel of the step by time: the optimizer update (then the EMA, first entry of roofline_kernels)
        "roofline": sgd_entry if sgd_entry is not None else ema_entry,
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
                                               steps=args.cpu_steps, warmup=args.cpu_warmup)
        out["cpu_baseline"] = {"value": round(ips, 3), "unit": "images/sec", "cores": physical_cores(), "threads": threads,
                               "kind": "port",
                               "sample": f"{args.cpu_steps} steps of {args.cpu_batch} images ({hw}x{hw}, queue {args.queue}, "
                                         f"same model) after {args.cpu_warmup} warm-up steps, fp32, {secs:.1f} s",
                               "split_ms_per_step": time_cpu_baseline.last_split_ms}
    if rank == 0:
        sys.stdout.flush()
        os.write(json_fd, (json.dumps(out) + "\n").encode())
    if world > 1 or rehearse:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
