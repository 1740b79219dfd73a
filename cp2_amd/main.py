"""`python -m cp2_amd.main ...` -- the pre-training CLI of reference main.py on MI355X.

Every flag of reference main.py:37-132 is accepted with the same name, type and default, so the
authors' launch scripts keep working; additive flags are grouped at the end of get_args().
One process per GPU: either started by torchrun (RANK / LOCAL_RANK / WORLD_SIZE in the env) or,
as the reference does (main.py:732), spawned here with --world-size N.  Collectives run on RCCL
(`--dist-backend nccl` is RCCL on ROCm).  Input: `--tensor_dataset FILE` keeps the whole image set in HBM and builds
every batch on the device (cp2_amd/augment.py: two-crop views with id maps + erased backgrounds, the geometry of
reference loader.py:50-118 / main.py:204-245); `--synthetic` feeds generated batches with the same contract.
Decoding image folders (PIL / cv2) and the photometric transforms stay outside this hot path.
"""
from __future__ import annotations

import os

os.environ.setdefault("MIOPEN_FIND_MODE", "FAST")   # short MIOpen solver search; see bench.py
from .miopen_cache import use_shipped_find_db  # noqa: E402
use_shipped_find_db()                               # start from the shipped solver rankings (miopen_cache.py; CP2_MIOPEN_DB=0: search)

import argparse
import math
import shutil
import time
from pathlib import Path

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp
from torch.nn.parallel import DistributedDataParallel

from . import builder, synthetic
from . import dist as cdist
from .ddp import FlatDDP
from .config import Config
from .engine import TrainStep
from .pretrain_types import PretrainType

DEFAULT_QUEUE_SIZE = 65536


def get_args(argv=None):
    # fmt: off
    p = argparse.ArgumentParser(description="Copy-Paste Contrastive Pretraining (MI355X-native)")
    p.add_argument("--config", help="path to configuration file")
    p.add_argument("--run_id", required=True, type=str)
    p.add_argument("--tags", nargs="+", default=[])
    p.add_argument("--offline_wandb", action="store_true")
    p.add_argument("--debug", action="store_true")
    p.add_argument("--pretrain_from_scratch", action="store_true")
    p.add_argument("--use_predictor", action="store_true")
    p.add_argument("--use_avgpool_global", action="store_true")
    p.add_argument("--use_symmetrical_loss", action="store_true")
    p.add_argument("--lmbd_coordinate", default=0, type=float)
    p.add_argument("--log_dir", type=str, required=True)
    p.add_argument("--wandb_project", type=str, default="ssl-pretraining")
    p.add_argument("--wandb_team", type=str, default="critical-ml-dg")
    p.add_argument("--data_dirs", metavar="DIR", nargs="+", default=[])
    p.add_argument("--directory_type", type=str, default="FILENAME")
    p.add_argument("--backbone_type", type=str, choices=[x.name for x in builder.BackboneType], default=builder.BackboneType.DEEPLABV3.name)
    p.add_argument("--pretrain_type", type=str, choices=[x.name for x in PretrainType], default=PretrainType.CP2.name)
    p.add_argument("--mapping_type", type=str, choices=[x.name for x in builder.MappingType], default=builder.MappingType.CP2.name)
    p.add_argument("--negative_type", type=str, choices=[x.name for x in builder.NegativeType], default=builder.NegativeType.NONE.name)
    p.add_argument("--negative_scale", type=float, default=2)
    p.add_argument("--num-workers", default=32, type=int)
    p.add_argument("--lmbd_cp2_dense_loss", default=0.2, type=float)
    p.add_argument("--lmbd_region_corr_weight", default=1, type=float)
    p.add_argument("--lmbd_pixel_corr_weight", default=1, type=float)
    p.add_argument("--lmbd_not_corr_weight", default=1, type=float)
    p.add_argument("--pixel_ids_stride", default=1, type=int)
    p.add_argument("--unet_truncated_dec_blocks", default=2, type=int)
    p.add_argument("--same_foreground", action="store_true")
    p.add_argument("--cap_queue", action="store_true")
    p.add_argument("--include_background", action="store_true")
    p.add_argument("--dense_logits_temp", default=1, type=float)
    p.add_argument("--instance_logits_temp", default=0.2, type=float)
    p.add_argument("--lemon_data", action="store_true")
    p.add_argument("--img_height", default=224, type=int)
    p.add_argument("--img_width", default=224, type=int)
    p.add_argument("--foreground_min", default=0.5, type=float)
    p.add_argument("--foreground_max", default=0.8, type=float)
    p.add_argument("--dist-url", default="tcp://127.0.0.1:10001", type=str)
    p.add_argument("--dist-backend", default="nccl", type=str)
    p.add_argument("--dist_timeout", default=cdist.DEFAULT_TIMEOUT_S, type=float,
                   help="seconds a collective may take before the process group aborts the run (additive flag; torch's nccl "
                        "default is 600); the hang watchdog reports the exchange step that did not complete at 0.8 x this")
    p.add_argument("--world-size", default=1, type=int)
    p.add_argument("--epochs", default=200, type=int)
    p.add_argument("--max_steps", default=np.inf, type=float)
    p.add_argument("--num-images", default=1281167, type=int)
    p.add_argument("--start-epoch", default=0, type=int)
    p.add_argument("-b", "--batch-size", default=256, type=int, help="total batch size over all GPUs")
    p.add_argument("--lr", "--learning-rate", default=0.03, type=float, dest="lr")
    p.add_argument("--remove_lr_scheduler", action="store_true")
    p.add_argument("--momentum", default=0.9, type=float)
    p.add_argument("--optim", default="sgd")
    p.add_argument("--wd", "--weight-decay", default=1e-4, type=float, dest="weight_decay")
    p.add_argument("-p", "--print-freq", default=10, type=int)
    p.add_argument("--scalar-freq", default=100, type=int)
    p.add_argument("--ckpt-freq", default=100, type=int)
    p.add_argument("--resume", default="", type=str)
    p.add_argument("--seed", default=0, type=int)
    # ---- additive (not in the reference)
    p.add_argument("--queue_size", default=DEFAULT_QUEUE_SIZE, type=int, help="MoCo queue length K (reference: fixed 65536)")
    p.add_argument("--synthetic", action="store_true", help="on-device synthetic batches (loader contract of SURVEY 8d)")
    p.add_argument("--tensor_dataset", default="", type=str,
                   help="torch file {'images': uint8/float32 [N,3,H,W], optional 'region_ids': [N,H,W]}: resident in HBM, "
                        "augmented on the device (cp2_amd/augment.py): two-crop views with id maps, backgrounds with the "
                        "erased rectangle, and -- for uint8 images -- ColorJitter / grayscale / GaussianBlur as reference "
                        "main.py:204-245 applies them (a float32 dataset gets the geometric transforms only)")
    p.add_argument("--no_photometric", action="store_true",
                   help="with --tensor_dataset: geometric transforms only (no ColorJitter / grayscale / GaussianBlur)")
    p.add_argument("--steps_per_epoch", default=100, type=int, help="with --synthetic: steps per epoch")
    p.add_argument("--amp", default="bf16", choices=["none", "bf16"], help="encoder autocast dtype")
    p.add_argument("--no_channels_last", action="store_true")
    p.add_argument("--one_device", action="store_true",
                   help="rehearsal of the multi-rank path on a one-GPU box: every rank uses cuda:0 (RCCL refuses two ranks on "
                        "one GPU: combine with --dist-backend gloo)")
    p.add_argument("--grad_sync", default="flat", choices=["flat", "ddp"],
                   help="world size > 1: gradient averaging by cp2_amd.ddp.FlatDDP (default) or torch's DistributedDataParallel")
    # fmt: on
    args = p.parse_args(argv)
    args.pretrain_type = PretrainType[args.pretrain_type]
    args.backbone_type = builder.BackboneType[args.backbone_type]
    args.mapping_type = builder.MappingType[args.mapping_type]
    args.negative_type = builder.NegativeType[args.negative_type]
    if args.lemon_data:
        args.img_height = args.img_width = 512
    if args.pretrain_type == PretrainType.DENSECL:          # reference main.py:148-153
        args.dense_logits_temp = args.instance_logits_temp = 0.2
        args.use_predictor = False
        args.lmbd_cp2_dense_loss = 0.5
        assert args.pixel_ids_stride == 1
    if args.pretrain_type == PretrainType.PROPOSED_V2:
        assert args.pixel_ids_stride == 1
    return args


def make_optimizer(params, args, device, capturable: bool, model=None):
    """SGD(momentum, wd) / AdamW as reference main.py:467-477.  capturable: keep the learning rate in a device tensor
    (a caller that captures the update into a hipGraph of its own then needs no re-capture per LR change).  With `model` (a builder.MODEL on the GPU) the SGD step is
    cp2_amd.optim.FlatSGD: the same update as torch.optim.SGD, bit for bit, in one HIP launch on the flat buffer."""
    lr = torch.tensor(float(args.lr), device=device) if capturable else args.lr
    if args.optim == "adamw":
        return torch.optim.AdamW(params, lr, weight_decay=0.01, fused=capturable or None, capturable=capturable)
    if args.optim == "sgd" and model is not None and torch.device(device).type == "cuda":
        from .optim import FlatSGD
        return FlatSGD(model, lr, momentum=args.momentum, weight_decay=args.weight_decay)
    if args.optim == "sgd":
        return torch.optim.SGD(params, lr, momentum=args.momentum, weight_decay=args.weight_decay,
                               fused=True if capturable else None)
    raise NotImplementedError("Only sgd and adamw optimizers are supported.")


def adjust_learning_rate(optimizer, epoch, args):
    lr = args.lr * 0.5 * (1.0 + math.cos(math.pi * epoch / args.epochs))     # reference main.py:693-698
    for g in optimizer.param_groups:
        if isinstance(g["lr"], torch.Tensor):
            g["lr"].fill_(lr)
        else:
            g["lr"] = lr
    return lr


def save_checkpoint(state, is_best, epoch, filename="checkpoint.ckpt"):
    torch.save(state, filename)                                               # reference main.py:661-670
    if is_best:
        shutil.copyfile(filename, os.path.join(Path(filename).parent, "checkpoint.ckpt"))


def build_model(args, cfg, rank, device):
    m_val = 0.999 if args.pretrain_type in (PretrainType.CP2, PretrainType.PROPOSED, PretrainType.DENSECL,
                                            PretrainType.PROPOSED_V2) else 0.996
    K = min(args.num_images, args.queue_size) if args.cap_queue else args.queue_size
    return builder.MODEL(
        cfg, m=m_val, K=K, dim=128, pretrain_from_scratch=args.pretrain_from_scratch,
        include_background=args.include_background, lmbd_cp2_dense_loss=args.lmbd_cp2_dense_loss,
        pretrain_type=args.pretrain_type, backbone_type=args.backbone_type, mapping_type=args.mapping_type,
        negative_type=args.negative_type, negative_scale=args.negative_scale,
        lmbd_pixel_corr_weight=args.lmbd_pixel_corr_weight, lmbd_region_corr_weight=args.lmbd_region_corr_weight,
        lmbd_not_corr_weight=args.lmbd_not_corr_weight, dense_logits_temp=args.dense_logits_temp,
        instance_logits_temp=args.instance_logits_temp, unet_truncated_dec_blocks=args.unet_truncated_dec_blocks,
        use_predictor=args.use_predictor, use_avgpool_global=args.use_avgpool_global,
        use_symmetrical_loss=args.use_symmetrical_loss, lmbd_coordinate=args.lmbd_coordinate, device=device, rank=rank,
        amp_dtype=torch.bfloat16 if args.amp == "bf16" else None, channels_last=not args.no_channels_last)


def main_worker(rank, args):
    world = args.world_size
    if not torch.cuda.is_available():
        raise RuntimeError("cp2_amd.main needs a GPU: the CP2 hot path has no CPU implementation "
                           "(the CPU baseline lives in oracle/ and is driven by bench.py)")
    local = 0 if args.one_device else int(os.environ.get("LOCAL_RANK", rank))
    torch.cuda.set_device(local)
    device = torch.device("cuda", local)
    torch.manual_seed(args.seed)
    np.random.seed(args.seed)
    torch.backends.cudnn.benchmark = True            # measured-fastest MIOpen solvers (1.4x on the encoder work)
    cfg = Config.fromfile(args.config)
    if world > 1 and not dist.is_initialized():
        # bounded timeout + hang watchdog + the RCCL environment, in one place (dist.init_process_group); a single-GPU run
        # creates no process group at all, so none of the exchange steps is issued (dist.multi())
        cdist.init_process_group(args.dist_backend, rank, world, init_method=args.dist_url, timeout_s=args.dist_timeout)
    model = build_model(args, cfg, rank, device).to(device)
    if not args.no_channels_last:
        model.encoder_q.to(memory_format=torch.channels_last)
        model.encoder_k.to(memory_format=torch.channels_last)
    wrapped = model
    if world > 1:
        # queue / BN buffers are updated identically on every rank, so the per-forward buffer
        # broadcast of the reference's default DDP is dropped; every parameter the chosen
        # path never uses is frozen in builder.MODEL (conv_seg on the contrast path, the segmentation head
        # and the unselected neck heads for DENSECL / PROPOSED_V2), so no unused-parameter search either.
        if args.grad_sync == "flat":     # one pack launch + one all-reduce per bucket of the flat gradient buffer (ddp.py)
            wrapped = FlatDDP(model)
        else:
            wrapped = DistributedDataParallel(model, device_ids=[local], output_device=local, broadcast_buffers=False,
                                              gradient_as_bucket_view=True, bucket_cap_mb=builder.DDP_BUCKET_MB)
    optimizer = make_optimizer([p for p in model.parameters()], args, device, capturable=False, model=model)
    if args.resume and os.path.isfile(args.resume):
        ck = torch.load(args.resume, map_location=device)
        args.start_epoch = ck["epoch"]
        wrapped.load_state_dict(ck["state_dict"]) if world > 1 else model.load_state_dict(
            {k.replace("module.", "", 1): v for k, v in ck["state_dict"].items()})
        optimizer.load_state_dict(ck["optimizer"])
    per_gpu = args.batch_size // world
    dataset = None
    if args.tensor_dataset:
        from . import augment
        dataset = augment.DeviceDataset.from_file(args.tensor_dataset, device)
        # three independently shuffled passes over the same images: foreground pairs, background 0, background 1
        samplers = [augment.EpochSampler(len(dataset), world, rank, seed) for seed in (0, 1024, 2048)]   # main.py:286-288
        steps_per_epoch = (len(dataset) // world) // per_gpu
        if steps_per_epoch < 1:
            raise ValueError(f"--tensor_dataset holds {len(dataset)} images: fewer than one batch of {args.batch_size}")
        use_regions = args.mapping_type in (builder.MappingType.REGION_ID, builder.MappingType.PIXEL_REGION_ID)
    elif args.synthetic:
        steps_per_epoch = args.steps_per_epoch
    else:
        raise NotImplementedError("give --tensor_dataset FILE (images resident in HBM, augmented on the device) or "
                                  "--synthetic; decoding image folders (reference datasets/, PIL / cv2) is outside the hot path")
    runner = TrainStep(wrapped, optimizer)
    step = 0
    for epoch in range(args.start_epoch, args.epochs):
        lr = args.lr if args.remove_lr_scheduler else adjust_learning_rate(optimizer, epoch, args)
        model.train()
        t0, seen = time.time(), 0
        if dataset is not None:
            order = [s.indices(epoch) for s in samplers]
            rng = np.random.default_rng([args.seed, rank, epoch])      # a resumed run draws epoch e's parameters again
        for i in range(steps_per_epoch):
            if step > args.max_steps:
                break
            if dataset is not None:
                sl = slice(i * per_gpu, (i + 1) * per_gpu)
                batch = augment.make_step_batch(dataset, order[0][sl], order[1][sl], order[2][sl], args.img_height,
                                                args.img_width, rng, args.foreground_min, args.foreground_max,
                                                id_stride=args.pixel_ids_stride, use_regions=use_regions,
                                                photometric=False if args.no_photometric else None)
            else:
                batch = synthetic.make_batch(per_gpu, args.img_height, args.img_width, device,
                                             seed=args.seed + rank + 1000 * step, foreground_min=args.foreground_min,
                                             foreground_max=args.foreground_max)
            if args.same_foreground:
                batch["img_b"], batch["pixel_ids_b"], batch["region_ids_b"] = batch["img_a"], batch["pixel_ids_a"], batch["region_ids_a"]
            cdist.progress(step)
            if step == 2:
                cdist.steady()                  # the solver searches of the first steps are over: the steady-state hang limit applies
            loss = runner(batch)
            seen += per_gpu * world
            if i % args.print_freq == 0 and rank == 0:
                print(f"Epoch: [{epoch}][{i}/{steps_per_epoch}] loss {float(loss):.4f} lr {lr:.5f} "
                      f"{seen / (time.time() - t0):.0f} img/s", flush=True)
            step += 1
        model.on_train_epoch_end(step)
        last = epoch % args.ckpt_freq == args.ckpt_freq - 1 or step > args.max_steps or epoch >= args.epochs - 1
        if last and rank == 0:
            sd = {("module." + k): v for k, v in model.state_dict().items()}    # DDP-prefixed keys (segment_network.py:84-92)
            save_checkpoint({"epoch": epoch + 1, "state_dict": sd, "optimizer": optimizer.state_dict(),
                             "pretrain_type": args.pretrain_type.name, "backbone_type": args.backbone_type.name},
                            epoch=epoch, is_best=True,
                            filename=os.path.join(args.log_dir, args.run_id, f"{step}_{epoch}_checkpoint.ckpt"))
        if step > args.max_steps:
            break
    if dist.is_initialized():
        dist.destroy_process_group()


def main(argv=None):
    args = get_args(argv)
    os.makedirs(os.path.join(args.log_dir, args.run_id), exist_ok=True)
    if args.debug:
        args.batch_size, args.world_size = 8, 1
    if "RANK" in os.environ:                                   # torchrun
        args.world_size = int(os.environ["WORLD_SIZE"])
        args.dist_url = "env://"
        main_worker(int(os.environ["RANK"]), args)
    elif args.world_size == 1:
        main_worker(0, args)
    else:
        mp.spawn(main_worker, nprocs=args.world_size, args=(args,))


if __name__ == "__main__":
    main()
