#!/usr/bin/env python3
"""Supervised CutPaste / mirror pre-training driver: the flag set of the reference's mirror_pretrain.py:21-85 and its
main() (:148-249) without Lightning / wandb (not installed): a plain loop (Adam, one process per GPU under
torch.distributed.run, DDP over RCCL), validation loss each epoch, `checkpoint.ckpt` of the best validation loss in
<log_dir>/<run_id>/ with the key layout networks/segment_network.py:94-100 loads for PretrainType.MIRROR.

Data: --tensor_dataset FILE (torch.save()d uint8 [N,H,W,3] or [N,3,H,W] tensor, or a dict with 'train' / 'val'
tensors), already resized to --img_x_size x --img_y_size (the reference's base_transform), kept resident in HBM; the
CutPaste composition runs on the device (csrc/mirror.hip).  --synthetic N makes N random images instead.

    python -m cp2_amd.mirror_pretrain --run_id r0 --log_dir /tmp/runs --synthetic 64 --config configs/config_pretrain_r18.py \
        -x 64 -y 64 --epochs 2 --batch-size 8
"""
from __future__ import annotations

import argparse
import os
import sys

import numpy as np
import torch

from . import mirror as M
from .config import Config
from .pretrain_types import PretrainType


def get_args(argv=None):
    parser = argparse.ArgumentParser()
    # fmt:off
    parser.add_argument('--config', default='configs/config_finetune.py', help='path to configuration file')
    parser.add_argument("--seed", type=int, default=0, help='Set global seed')
    parser.add_argument("--run_id", type=str, required=True, help='Unique identifier for a run')
    parser.add_argument("--tags", nargs='+', default=[], help='Tags to include for logging')
    parser.add_argument("--data_dirs", nargs='+', help='Folder(s) containing image data (image decoding is out of scope: use --tensor_dataset)')
    parser.add_argument("--log_dir", type=str, required=True, help='For storing artifacts')
    parser.add_argument("--wandb_project", type=str, default='ssl-pretraining', help='(accepted, unused: wandb is not installed)')
    parser.add_argument("--wandb_team", type=str, default='critical-ml-dg', help='(accepted, unused)')
    parser.add_argument("--num_gpus", type=int, default=2, help='number of gpus (the launcher decides: torch.distributed.run)')
    parser.add_argument("--num-workers", type=int, default=0, help='(accepted, unused: the data is resident on the device)')
    parser.add_argument("--fast_dev_run", action='store_true', help="For debugging: one batch of train / val")
    parser.add_argument("--use_profiler", action='store_true', help="(accepted, unused)")
    parser.add_argument("-x", "--img_x_size", type=int, default=512, help='height of image')
    parser.add_argument("-y", "--img_y_size", type=int, default=512, help='width of image')
    parser.add_argument("--num_classes", type=int, default=2)
    parser.add_argument('--lemon_data', action='store_true', help='Running with lemon data')
    # cutpaste
    parser.add_argument('--softmax_temp', type=float, default=2)
    parser.add_argument("--lmbd_compare_loss", type=float, default=0.01, help='Loss coefficient')
    parser.add_argument('--variant', choices=[x.name for x in M.MirrorVariant], default=M.MirrorVariant.OUTPUT.name)
    parser.add_argument("--max_num_patches", type=int, default=1, help='Maximum number of cutpastes')
    parser.add_argument("--min_area_scale", type=float, default=0.02, help='minimum area of patch')
    parser.add_argument("--max_area_scale", type=float, default=0.15, help='maximum area of patch')
    parser.add_argument("--min_aspect_ratio", type=float, default=1/3, help='minimum aspect ratio of patch')
    parser.add_argument("--max_aspect_ratio", type=float, default=4/3, help='maximum aspect ratio of patch')
    parser.add_argument("--min_rotation", type=int, default=0, help='minimum rotation angle of patch')
    parser.add_argument("--max_rotation", type=int, default=0, help='max rotation angle of patch')
    parser.add_argument("--batch-size", type=int, default=10, help='Batch size to train with')
    parser.add_argument("--lr", type=float, default=0.001, help='Max learning rate used during training')
    parser.add_argument("--epochs", type=int, default=200, help='Number of training epochs')
    parser.add_argument("--weight_decay", type=float, default=0.0001, help='weight decay of optimizer')
    # additions of this build
    parser.add_argument("--tensor_dataset", type=str, default=None, help='torch file with the resized uint8 images')
    parser.add_argument("--synthetic", type=int, default=0, help='use this many random images instead of a dataset')
    parser.add_argument("--pretrain_type", choices=[x.name for x in PretrainType], default=PretrainType.RANDOM.name,
                        help='initialisation (the reference hard-codes NONE = ImageNet download; no network here)')
    parser.add_argument("--amp", choices=["bf16", "none"], default="none", help='the reference trains with precision=32')
    parser.add_argument("--no_sync_batchnorm", action="store_true",
                        help='more than one rank: keep BatchNorm statistics per rank (the reference syncs them: Trainer(sync_batchnorm=True))')
    parser.add_argument("--dist_backend", type=str, default="nccl", help='torch.distributed backend (nccl = RCCL)')
    # fmt:on
    args = parser.parse_args(argv)
    args.log_dir = os.path.abspath(os.path.expanduser(args.log_dir))
    args.variant = M.MirrorVariant[args.variant]
    args.pretrain_type = PretrainType[args.pretrain_type]
    if args.lemon_data:
        args.img_x_size = 544
        args.img_y_size = 1024
        args.epochs = 200
        args.max_area_scale = 0.007
        args.min_area_scale = 0.0003
        args.max_num_patches = 1
    return args


def _load_images(args, device):
    """-> (train uint8 [N,H,W,3], val uint8 [M,H,W,3]) on the device."""
    H, W = args.img_x_size, args.img_y_size
    if args.synthetic:
        g = torch.Generator().manual_seed(args.seed)
        x = torch.randint(0, 256, (args.synthetic, H, W, 3), generator=g, dtype=torch.uint8)
        n_val = max(1, args.synthetic // 8)
        return x[n_val:].to(device), x[:n_val].to(device)
    if not args.tensor_dataset:
        raise SystemExit("mirror_pretrain: give --tensor_dataset FILE or --synthetic N (image folders are not decoded here)")
    obj = torch.load(args.tensor_dataset, map_location="cpu")
    tr, va = (obj["train"], obj.get("val")) if isinstance(obj, dict) else (obj, None)

    def hwc(t):
        if t.dtype != torch.uint8 or t.dim() != 4:
            raise SystemExit("mirror_pretrain: images must be uint8 [N,H,W,3] or [N,3,H,W]")
        t = t.permute(0, 2, 3, 1) if t.shape[1] == 3 and t.shape[3] != 3 else t
        if tuple(t.shape[1:]) != (H, W, 3):
            raise SystemExit(f"mirror_pretrain: images are {tuple(t.shape[1:])}, expected {(H, W, 3)} (resize them first)")
        return t.contiguous().to(device)
    tr = hwc(tr)
    if va is None:
        n_val = max(1, len(tr) // 10)
        tr, va = tr[n_val:], tr[:n_val]
    else:
        va = hwc(va)
    return tr, va


def shard_len(n: int, world: int, batch: int) -> int:
    """Samples per rank and epoch: the same on every rank and, when the data allow it, a whole number of batches."""
    per_rank = n // world
    return (per_rank // batch) * batch if per_rank >= batch else per_rank


def main(args):
    import torch.distributed as dist
    world, rank, local = (int(os.environ.get(k, d)) for k, d in (("WORLD_SIZE", "1"), ("RANK", "0"), ("LOCAL_RANK", "0")))
    torch.cuda.set_device(local)
    device = torch.device("cuda", local)
    if world > 1:
        from . import dist as cdist
        cdist.init_process_group(args.dist_backend, rank, world)
    np.random.seed(args.seed + rank)
    torch.manual_seed(args.seed)
    train_u8, val_u8 = _load_images(args, device)
    mk = lambda n: M.CutPasteSampler(n, args.min_area_scale, args.max_area_scale, args.min_aspect_ratio,  # noqa: E731
                                     args.max_aspect_ratio, args.min_rotation, args.max_rotation, args.variant,
                                     args.num_classes, args.max_num_patches)
    train_s, val_s = mk(len(train_u8)), mk(len(val_u8))
    args.run_dir = os.path.join(args.log_dir, args.run_id)
    os.makedirs(args.run_dir, exist_ok=True)

    cfg = Config.fromfile(args.config)
    cfg.model.decode_head.num_classes = args.num_classes
    cfg.model.decode_head.contrast = False
    model = M.MirrorModule(model_config=cfg, pretrain_type=args.pretrain_type, learning_rate=args.lr,
                           weight_decay=args.weight_decay, num_classes=args.num_classes,
                           image_shape=(3, args.img_x_size, args.img_y_size), lmbd_compare_loss=args.lmbd_compare_loss,
                           softmax_temp=args.softmax_temp, mirror_variant=args.variant,
                           amp_dtype=torch.bfloat16 if args.amp == "bf16" else None).to(device)
    step_mod = model
    if world > 1 and not args.no_sync_batchnorm:
        # the reference's Trainer(sync_batchnorm=True) (mirror_pretrain.py:229-231): batch statistics over all ranks
        from .encoder import convert_sync_batchnorm
        model = convert_sync_batchnorm(model)
        step_mod = model
    if world > 1:
        # broadcast_buffers=False: the confusion counts are per-rank tallies (summed over ranks in MirrorModule.metrics);
        # DDP's default would overwrite them with rank 0's on every forward
        step_mod = torch.nn.parallel.DistributedDataParallel(_StepWrapper(model), device_ids=[local], broadcast_buffers=False)
    optimizer = model.configure_optimizers()["optimizer"]
    best, b = float("inf"), args.batch_size
    for epoch in range(args.epochs):
        model.train()
        perm = torch.randperm(len(train_u8), generator=torch.Generator().manual_seed(args.seed + epoch)).tolist()
        # equal shards on every rank (Lightning's DistributedSampler pads; here the tail that does not fill one batch on every
        # rank is dropped): a rank with one more step than its peers would wait for a gradient all-reduce that never comes
        per_rank = shard_len(len(perm), world, b)
        perm = perm[rank:per_rank * world:world]
        nb = max(1, len(perm) // b) if not args.fast_dev_run else 1
        for i in range(nb):
            batch = M.cutpaste_batch(train_u8, train_s, perm[i * b:(i + 1) * b])
            batch = tuple(t for t in batch if t is not None)
            loss = step_mod(batch) if world > 1 else model.training_step(batch, i)
            optimizer.zero_grad(set_to_none=True)
            loss.backward()
            optimizer.step()
        model.eval()
        tot, cnt = torch.zeros((), device=device), 0
        with torch.no_grad():
            idx = list(range(len(val_u8)))[rank::world]
            for i in range(0, len(idx) if not args.fast_dev_run else min(b, len(idx)), b):
                batch = tuple(t for t in M.cutpaste_batch(val_u8, val_s, idx[i:i + b]) if t is not None)
                tot += model.validation_step(batch, i) * len(idx[i:i + b])
                cnt += len(idx[i:i + b])
        stat = torch.stack([tot, torch.tensor(float(cnt), device=device)])
        if world > 1:
            dist.all_reduce(stat)
        val_loss = float(stat[0] / stat[1].clamp_min(1))
        tm, vm = model.metrics(M.Stage.TRAIN), model.metrics(M.Stage.VAL)
        if rank == 0:
            print(f"epoch {epoch}: train_loss {float(model.logged['train_loss']):.4f} val_loss_epoch {val_loss:.4f} "
                  f"train_jaccard {tm['train_jaccard']:.4f} val_jaccard {vm['val_jaccard']:.4f}", flush=True)
            if val_loss < best:                                       # ModelCheckpoint(monitor="val_loss_epoch", mode="min")
                best = val_loss
                torch.save({"epoch": epoch + 1, "state_dict": model.state_dict(), "pretrain_type": PretrainType.MIRROR.name,
                            "hyper_parameters": {k: str(v) for k, v in vars(args).items()}},
                           os.path.join(args.run_dir, "checkpoint.ckpt"))
        if args.fast_dev_run:
            break
    if world > 1:
        dist.destroy_process_group()
    return best


class _StepWrapper(torch.nn.Module):
    """DDP wraps forward(); the training step is the forward of this wrapper."""

    def __init__(self, module):
        super().__init__()
        self.module = module

    def forward(self, batch):
        return self.module.training_step(batch)


if __name__ == "__main__":
    a = get_args()
    print("Module Command Line Arguments: ", vars(a))
    main(a)
    sys.exit(0)
