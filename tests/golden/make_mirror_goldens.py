#!/usr/bin/env python3
"""Golden vectors of the supervised CutPaste / mirror pre-training path, made by running the
REFERENCE's own code on CPU in the build container (the reference tree does not travel):

    cd /tmp && PYTHONDONTWRITEBYTECODE=1 python /root/repo/tests/golden/make_mirror_goldens.py

What runs, unmodified: /root/reference/datasets/pretrain_dataset.py `CutPasteDataset.cutpaste` / `__getitem__`
(numpy + Pillow, both installed here) and /root/reference/networks/mirror_network.py `MirrorModule.shared_step`
(plain torch).  Packages that are not installed (lightning, mmseg, torchmetrics, cv2, albumentations, torchvision) are
replaced by inert stub modules so the imports succeed; the dataset / module objects are created without their
constructors (which need those packages) and given the attributes the constructors would set.  Harness-only shims:
`cv2.imread` hands back the in-memory test image, `base_transform` is the identity, `T.ToTensor` is its documented
arithmetic (uint8 HWC -> float CHW / 255), mmseg's `resize` is `torch.nn.functional.interpolate` (which it wraps).

Output: tests/golden/mirror_cutpaste.npz, tests/golden/mirror_loss.npz (inputs + expected outputs only).
"""
import importlib.abc
import importlib.machinery
import os
import sys
from unittest.mock import MagicMock

import numpy as np

REF = "/root/reference"
OUT = os.path.dirname(os.path.abspath(__file__))
ABSENT = ("cv2", "segmentation_models_pytorch", "torchvision", "wandb", "mmseg", "torchmetrics",
          "lightning", "albumentations", "mmengine", "dotenv", "parameterized")


class _StubFinder(importlib.abc.MetaPathFinder, importlib.abc.Loader):
    def find_spec(self, name, path, target=None):
        if name.split(".")[0] in ABSENT:
            return importlib.machinery.ModuleSpec(name, self, is_package=True)

    def create_module(self, spec):
        m = MagicMock(name=spec.name)
        m.__path__ = []
        m.__spec__ = spec
        m.__name__ = spec.name
        return m

    def exec_module(self, m):
        pass


sys.meta_path.insert(0, _StubFinder())
sys.path.insert(0, REF)
sys.dont_write_bytecode = True

import torch  # noqa: E402
import torch.nn as nn  # noqa: E402
import torch.nn.functional as F  # noqa: E402
import lightning  # noqa: E402

lightning.LightningModule = nn.Module
lightning.LightningDataModule = object
from datasets import pretrain_dataset as ref_ds  # noqa: E402
from networks import mirror_network as ref_mn  # noqa: E402
from networks import segment_network as ref_sn  # noqa: E402

ref_sn.resize = lambda input, size, mode, align_corners: F.interpolate(input, size=size, mode=mode, align_corners=align_corners)


def to_tensor(img):
    """torchvision.transforms.ToTensor on a uint8 HWC array / PIL image."""
    a = np.asarray(img)
    return torch.from_numpy(a.copy()).permute(2, 0, 1).contiguous().to(torch.float32).div(255)


def blank_dataset(images, variant, num_classes, max_num_patches, rot=(0, 0), area=(0.02, 0.15), aspect=(1 / 3, 4 / 3)):
    ds = ref_ds.CutPasteDataset.__new__(ref_ds.CutPasteDataset)
    ds.images_list = list(range(len(images)))
    ds.base_transform = lambda image: {"image": image}
    ds.to_tensor = to_tensor
    ds.debug = True
    ds.min_rotation, ds.max_rotation = rot
    ds.min_area_scale, ds.max_area_scale = area
    ds.min_aspect_ratio, ds.max_aspect_ratio = aspect
    ds.mirror_variant = variant
    ds.max_num_patches = max_num_patches
    P = ref_ds.CutPastePatchType
    ds.transforms_list = [
        lambda im: (im[0], im[1], np.zeros(shape=im[0].shape[:2], dtype=np.uint8)),
        lambda im: ds.cutpaste(image=im[0], mirror_image=im[1], patch_type=P.REGULAR),
        lambda im: ds.cutpaste(image=im[0], mirror_image=im[1], patch_type=P.SCAR),
    ]
    ds.classes = list(range(num_classes))
    return ds


def cutpaste_cases():
    rng = np.random.RandomState(1234)
    H, W, N = 72, 104, 6
    # RGB test images; cv2.imread returns BGR and __getitem__ converts BGR -> RGB, so the shim hands back BGR
    images = rng.randint(0, 256, size=(N, H, W, 3), dtype=np.uint8)
    ref_ds.cv2.imread = lambda idx: images[idx][:, :, ::-1]
    ref_ds.cv2.cvtColor = lambda im, code: im[:, :, ::-1]
    out = {"images": images}
    cases = [
        # name, variant, num_classes, max_num_patches, rotation range, targets, seed
        ("regular", "OUTPUT", 2, 1, (0, 0), [1, 1, 0, 1], 11),
        ("scar", "OUTPUT", 3, 1, (-45, 45), [2, 1, 2, 0, 2, 2], 12),
        ("multi", "OUTPUT", 2, 3, (0, 0), [1, 1, 1, 0, 1, 1], 13),
        ("none_variant", "NONE", 3, 1, (10, 170), [2, 2, 1], 14),
    ]
    for name, variant, ncls, maxp, rot, targets, seed in cases:
        ds = blank_dataset(images, ref_ds.MirrorVariant[variant], ncls, maxp, rot=rot)
        ds.targets = np.asarray(targets)
        np.random.seed(seed)
        imgs, mirrors, masks = [], [], []
        for i in range(len(targets)):
            item = ds[i % N]
            if variant == "OUTPUT":
                img, mir, mask, cls = item
                mirrors.append(mir.numpy())
            else:
                img, mask, cls = item
            assert cls == targets[i]
            imgs.append(img.numpy())
            masks.append(mask.numpy())
        out[f"{name}.seed"] = np.int64(seed)
        out[f"{name}.targets"] = np.asarray(targets, dtype=np.int64)
        out[f"{name}.index"] = np.asarray([i % N for i in range(len(targets))], dtype=np.int64)
        out[f"{name}.num_classes"] = np.int64(ncls)
        out[f"{name}.max_num_patches"] = np.int64(maxp)
        out[f"{name}.rotation"] = np.asarray(rot, dtype=np.float64)
        out[f"{name}.img"] = np.stack(imgs)
        out[f"{name}.mask"] = np.stack(masks)
        if mirrors:
            out[f"{name}.mirror"] = np.stack(mirrors)
    np.savez_compressed(os.path.join(OUT, "mirror_cutpaste.npz"), **out)
    print("mirror_cutpaste.npz:", {k: v.shape for k, v in out.items() if hasattr(v, "shape") and v.ndim > 1})


class _Net(nn.Module):
    """Stand-in for the segmentor: identity on a parameter-free path; the logits are the leaf inputs themselves."""

    def forward(self, x):
        return x


def blank_module(num_classes, image_shape, T, lmbd, variant):
    m = ref_mn.MirrorModule.__new__(ref_mn.MirrorModule)
    nn.Module.__init__(m)
    m.model = _Net()
    m.num_classes, m.image_shape = num_classes, image_shape
    m.class_loss = nn.CrossEntropyLoss()
    m.compare_loss = nn.CrossEntropyLoss()
    m.lmbd_compare_loss, m.softmax_temp = lmbd, T
    m.softmax = nn.Softmax(dim=1)
    m.mirror_variant = variant
    m.logged = {}
    m.log = lambda name, value, **kw: m.logged.__setitem__(name, value)
    m.log_dict = lambda *a, **k: None
    m.train_metrics = MagicMock()
    m.val_metrics = MagicMock()
    return m


def loss_cases():
    out = {}
    g = torch.Generator().manual_seed(7)
    cases = [("c2", 2, 3, 6, 10, 24, 40, 2.0, 0.01, "OUTPUT"),
             ("c3", 3, 2, 5, 7, 20, 28, 1.5, 0.5, "OUTPUT"),
             ("c3_none", 3, 4, 6, 6, 24, 24, 2.0, 0.01, "NONE")]
    for name, C, n, h, w, H, W, T, lmbd, variant in cases:
        m = blank_module(C, (3, H, W), T, lmbd, ref_ds.MirrorVariant[variant])
        # the segmentor's low-resolution logits are the leaves; forward() resizes them to the image size
        s = (torch.randn(n, C, h, w, generator=g) * 2).requires_grad_(True)
        t = (torch.randn(n, C, h, w, generator=g) * 2).requires_grad_(True)
        masks = torch.randint(0, C, (n, H, W), generator=g)
        if variant == "OUTPUT":
            loss = m.shared_step((s, t, masks), ref_sn.Stage.TRAIN)
        else:
            loss = m.shared_step((s, masks), ref_sn.Stage.TRAIN)
        loss.backward()
        with torch.no_grad():
            s_up, s_arg = m.forward(s)
        out[f"{name}.C"], out[f"{name}.T"], out[f"{name}.lmbd"] = np.int64(C), np.float64(T), np.float64(lmbd)
        out[f"{name}.image_hw"] = np.asarray([H, W], dtype=np.int64)
        out[f"{name}.s_logits"], out[f"{name}.masks"] = s.detach().numpy(), masks.numpy()
        out[f"{name}.s_up"], out[f"{name}.s_argmax"] = s_up.numpy(), s_arg.numpy()
        out[f"{name}.loss"] = loss.detach().numpy()
        out[f"{name}.class_loss"] = np.asarray(float(m.logged["train_class_loss"]), dtype=np.float32)
        out[f"{name}.compare_loss"] = np.asarray(float(m.logged["train_compare_loss"]), dtype=np.float32)
        out[f"{name}.grad_s"] = s.grad.numpy()
        if variant == "OUTPUT":
            out[f"{name}.t_logits"], out[f"{name}.grad_t"] = t.detach().numpy(), t.grad.numpy()
        print(name, float(loss), float(m.logged["train_class_loss"]), float(m.logged["train_compare_loss"]))
    np.savez_compressed(os.path.join(OUT, "mirror_loss.npz"), **out)


if __name__ == "__main__":
    cutpaste_cases()
    loss_cases()
