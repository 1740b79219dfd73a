#!/usr/bin/env python3
"""Generate golden vectors by running the REFERENCE's own code on CPU.

Run in the build container only (the reference tree does not travel):

    cd /tmp && PYTHONDONTWRITEBYTECODE=1 MPLBACKEND=Agg \
        python /root/repo/tests/golden/make_goldens.py

What runs: /root/reference/builder.py (`MODEL.forward_cp2`, `forward_densecl`,
`_momentum_update_key_encoder`, `_dequeue_and_enqueue`, `_batch_shuffle_ddp`,
`_batch_unshuffle_ddp`, `concat_all_gather`) and
/root/reference/tools/correlation_mapping.py, unmodified.  Third-party
packages that are not installed here (mmseg, wandb, cv2, ...) are replaced by
inert stub modules so `import builder` succeeds; none of them is touched by
the arithmetic being recorded.  The model object is created without running
its constructor (which needs mmseg) and given a tiny stand-in encoder, so the
fixtures are encoder-free: they record the encoder OUTPUTS as inputs of the
hot path.  Locals of the reference functions are captured with a profile hook
at function return -- the reference source is never edited or copied.

Outputs: tests/golden/*.npz (inputs + expected outputs only).
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
import lightning  # noqa: E402

lightning.LightningModule = nn.Module
lightning.LightningDataModule = object
import builder as ref  # noqa: E402  (the reference's builder.py)
from networks.segment_network import PretrainType  # noqa: E402
from tools import correlation_mapping as ref_cm  # noqa: E402

torch.Tensor.cuda = lambda self, *a, **k: self  # builder.py:175,618,1420 call .cuda()


def np_(x):
    if isinstance(x, torch.Tensor):
        return x.detach().cpu().numpy()
    return np.asarray(x)


class Capture:
    """Grab the locals of a reference function when it returns."""

    def __init__(self, code):
        self.code, self.locals = code, None

    def __enter__(self):
        def prof(frame, event, arg):
            if event == "return" and frame.f_code is self.code:
                self.locals = dict(frame.f_locals)
        sys.setprofile(prof)
        return self

    def __exit__(self, *a):
        sys.setprofile(None)


class StandIn(nn.Module):
    """Tiny encoder: b x 3 x H x W -> b x 128 x H/s x W/s; records its output."""

    def __init__(self, stride, dim=128):
        super().__init__()
        self.conv = nn.Conv2d(3, dim, stride, stride)
        self.last = None

    def forward(self, x):
        y = self.conv(x)
        if y.requires_grad:
            y.retain_grad()
        self.last = y
        return y


def blank_model(K, dim=128, stride=16, **kw):
    m = ref.MODEL.__new__(ref.MODEL)
    nn.Module.__init__(m)
    m.queue_len, m.momentum, m.dim = K, 0.999, dim
    m.temp_global = kw.get("temp_global", 0.2)
    m.temp_local = kw.get("temp_local", 1)
    m.include_background = kw.get("include_background", False)
    m.lmbd_dense_loss = kw.get("lmbd_dense", 0.2)
    m.device, m.rank, m.epoch = "cpu", 1, 0
    m.contrastive_head = ref.ContrastiveHead()
    m.use_predictor, m.use_avgpool_global = kw.get("use_predictor", False), kw.get("use_avgpool_global", False)
    m.use_symmetrical_loss = kw.get("use_symmetrical_loss", False)
    m.lmbd_coordinate = kw.get("lmbd_coordinate", 0)
    m.mapping_type = kw.get("mapping_type", ref.MappingType.CP2)
    m.lmbd_pixel_corr_weight = kw.get("w_pixel", 1)
    m.lmbd_region_corr_weight = kw.get("w_region", 1)
    m.lmbd_not_corr_weight = kw.get("w_not", 1)
    m.pretrain_type = kw.get("pretrain_type", PretrainType.CP2)
    m.negative_type, m.negative_scale = kw.get("negative_type", ref.NegativeType.NONE), kw.get("negative_scale", 2)
    m.backbone_type = ref.BackboneType.DEEPLABV3
    m.output_stride = stride
    m.backbone_output_stride = kw.get("backbone_stride", 32)
    m.register_buffer("queue", nn.functional.normalize(torch.randn(dim, K), dim=0))
    m.register_buffer("queue_ptr", torch.zeros(1, dtype=torch.long))
    m.register_buffer("queue2", nn.functional.normalize(torch.randn(dim, K), dim=0))
    m.register_buffer("queue2_ptr", torch.zeros(1, dtype=torch.long))
    for n in ("loss_o", "loss_i", "loss_d", "acc_ins", "acc_seg",
              "cross_image_variance_source", "cross_image_variance_target", "idx_unshuffle"):
        setattr(m, n, ref.AverageMeter(n))
    m.correlation_ious, m.masked_correlation_ious = [], []
    return m


def synth_inputs(b, h, w, gen, shared_regions=False):
    """Synthetic batch with the reference loader's output contract
    (loader.py:66-118, main.py:206-245): images in [0,1), backgrounds with an
    exactly-zero rectangle, pixel-id maps of two shifted crops of one id grid."""
    img_a = torch.rand(b, 3, h, w, generator=gen)
    img_b = torch.rand(b, 3, h, w, generator=gen)
    bgs = []
    for _ in range(2):
        bg = torch.rand(b, 3, h, w, generator=gen)
        for n in range(b):
            rh = int(torch.randint(h // 2, (4 * h) // 5, (1,), generator=gen))
            rw = int(torch.randint(w // 2, (4 * w) // 5, (1,), generator=gen))
            y0 = int(torch.randint(0, h - rh + 1, (1,), generator=gen))
            x0 = int(torch.randint(0, w - rw + 1, (1,), generator=gen))
            bg[n, :, y0:y0 + rh, x0:x0 + rw] = 0.0
        bgs.append(bg)
    big = torch.arange(1, 4 * h * w + 1).reshape(2 * h, 2 * w)
    pa, pb = [], []
    for n in range(b):
        dy = int(torch.randint(0, h // 2, (1,), generator=gen))
        dx = int(torch.randint(0, w // 2, (1,), generator=gen))
        pa.append(big[0:h, 0:w] + n * 4 * h * w)
        crop = big[dy:dy + h, dx:dx + w] + n * 4 * h * w
        pb.append(torch.flip(crop, dims=[1]) if n % 2 else crop)
    pixel_a, pixel_b = torch.stack(pa), torch.stack(pb)
    if shared_regions:   # coarse region ids (blocks of 24x24 pixels), id 0 = unknown region
        region_a = (pixel_a - 1) % (4 * h * w)
        region_a = ((region_a // (2 * w)) // 24) * 8 + ((region_a % (2 * w)) // 24)
        region_b = (pixel_b - 1) % (4 * h * w)
        region_b = ((region_b // (2 * w)) // 24) * 8 + ((region_b % (2 * w)) // 24)
    else:                # MappingType.CP2: region ids are the pixel ids (loader.py:84-85)
        region_a, region_b = pixel_a.clone(), pixel_b.clone()
    return dict(img_a=img_a, img_b=img_b, bg0=bgs[0], bg1=bgs[1], pixel_ids_a=pixel_a,
                pixel_ids_b=pixel_b, region_ids_a=region_a, region_ids_b=region_b)


CP2_KEEP = ("mask_a", "mask_b", "pixel_ids_a", "pixel_ids_b", "region_ids_a", "region_ids_b",
            "img_a", "img_b", "corr_weights", "q_dense", "q_pos", "q_neg", "k_dense", "k_pos", "k_neg",
            "_logits_dense", "logits_dense", "labels_dense", "l_pos", "l_neg", "logits_moco",
            "loss_instance", "loss_dense", "loss", "acc_dense", "positive_scores_average",
            "negative_scores_average", "instance_average_negative_scores", "instance_negative_quartiles",
            "cross_image_variance_source", "cross_image_variance_target", "idx_unshuffle")


def run_cp2_case(name, b, h, w, K, stride, seed, ptr0=0, slim=False, **kw):
    torch.manual_seed(seed)
    gen = torch.Generator().manual_seed(seed)
    m = blank_model(K, stride=stride, **kw)
    m.encoder_q, m.encoder_k = StandIn(stride), StandIn(stride)
    for pq, pk in zip(m.encoder_q.parameters(), m.encoder_k.parameters()):
        pk.data.copy_(pq.data + 0.01 * torch.randn_like(pq))   # k != q so the EMA is visible
        pk.requires_grad = False
    m.queue_ptr[0] = ptr0
    inp = synth_inputs(b, h, w, gen, shared_regions=kw.get("shared_regions", False))
    rec = {"in_" + k: np_(v) for k, v in inp.items()}
    rec["queue_before"], rec["ptr_before"] = np_(m.queue.clone()), np.int64(ptr0)
    keep_ema = kw.get("keep_ema", False)   # the EMA inside forward_cp2 (runs before the key encoder)
    if keep_ema:
        rec["ema_k_before_w"], rec["ema_q_w"] = np_(m.encoder_k.conv.weight), np_(m.encoder_q.conv.weight)
        rec["ema_k_before_b"], rec["ema_q_b"] = np_(m.encoder_k.conv.bias), np_(m.encoder_q.conv.bias)
    with Capture(ref.MODEL.forward_cp2.__code__) as cap:
        loss = m.forward_cp2(visualize=False, step=0, new_epoch=False, **inp)
    loss.backward()
    loc = cap.locals
    if slim:     # experimental-variant fixtures: encoder outputs in, losses / gradient / reshaped logits out
        rec = {k: v for k, v in rec.items() if k in ("queue_before", "ptr_before")}
    for k in (("mask_a", "mask_b", "pixel_ids_a", "pixel_ids_b", "region_ids_a", "region_ids_b", "corr_weights",
               "_logits_dense", "logits_dense", "l_pos", "loss_instance", "loss_dense", "loss",
               "acc_dense", "idx_unshuffle", "negative_scores_average") if slim else CP2_KEEP):
        v = loc[k]
        rec[k] = np_(torch.stack(list(v)) if isinstance(v, (tuple, list)) else v)
    stats = loc["contrast_stats"]
    for side in ("positive", "negative"):
        rec[f"dense_{side}_average"] = np_(stats[side]["average"])
        rec[f"dense_{side}_quartiles"] = np_(torch.stack(list(stats[side]["quartiles"])))
    rec["acc1"], rec["acc5"] = np_(loc["acc1"][0]), np_(loc["acc5"][0])
    rec["iou"], rec["iou_masked"] = np_(loc["region_corr_results"]["iou"]), np_(loc["region_corr_results"]["iou_masked"])
    rec["pixel_iou"], rec["pixel_iou_masked"] = np_(loc["pixel_corr_results"]["iou"]), np_(loc["pixel_corr_results"]["iou_masked"])
    rec["pixel_corr_map"] = np_(loc["pixel_corr_results"]["corr_map"])
    # the key encoder ran on the shuffled batch; k_feat is stored in the ORIGINAL sample order
    rec["q_feat"] = np_(m.encoder_q.last)
    rec["k_feat_shuffled"] = np_(m.encoder_k.last)
    rec["k_feat"] = rec["k_feat_shuffled"][np_(loc["idx_unshuffle"])]
    rec["dq_feat"] = np_(m.encoder_q.last.grad)
    rec["queue_after"], rec["ptr_after"] = np_(m.queue), np_(m.queue_ptr)[0]
    if keep_ema:
        rec["ema_k_after_w"], rec["ema_k_after_b"] = np_(m.encoder_k.conv.weight), np_(m.encoder_k.conv.bias)
    rec["cfg"] = np.array([b, h, w, K, stride, int(m.include_background)], dtype=np.int64)
    rec["cfg_f"] = np.array([m.temp_global, m.temp_local, m.lmbd_dense_loss, m.lmbd_pixel_corr_weight,
                             m.lmbd_region_corr_weight, m.lmbd_not_corr_weight, m.momentum], dtype=np.float64)
    rec["negative"] = np.array([m.negative_type.value, m.negative_scale], dtype=np.float64)
    np.savez_compressed(os.path.join(OUT, name + ".npz"), **rec)
    print(f"{name}: loss={float(loss):.6f} ins={float(loc['loss_instance']):.6f} "
          f"dense={float(loc['loss_dense']):.6f} ptr={int(m.queue_ptr)}")


class DenseBackbone(nn.Module):
    def __init__(self, stride):
        super().__init__()
        self.conv = nn.Conv2d(3, 2048, stride, stride)

    def forward(self, x):
        y = self.conv(x)
        return (y, y, y, y)


class DenseEnc(nn.Module):
    def __init__(self, stride):
        super().__init__()
        self.backbone = DenseBackbone(stride)
        self.neck = ref.DenseCLNeck(in_channels=2048, hid_channels=64, out_channels=128)


def run_densecl_case(name, b, h, w, K, stride, seed, lmbd_coordinate=0.0, symmetric=False, step=0, **kw):
    torch.manual_seed(seed)
    gen = torch.Generator().manual_seed(seed)
    m = blank_model(K, backbone_stride=stride, temp_local=0.2, lmbd_dense=0.5,
                    pretrain_type=PretrainType.PROPOSED_V2 if symmetric else PretrainType.DENSECL,
                    lmbd_coordinate=lmbd_coordinate, use_symmetrical_loss=symmetric, **kw)
    m.encoder_q, m.encoder_k = DenseEnc(stride), DenseEnc(stride)
    for pq, pk in zip(m.encoder_q.parameters(), m.encoder_k.parameters()):
        pk.data.copy_(pq.data + 0.01 * torch.randn_like(pq))
        pk.requires_grad = False
    inp = synth_inputs(b, h, w, gen)
    rec = {"in_pixel_ids_a": np_(inp["pixel_ids_a"]), "in_pixel_ids_b": np_(inp["pixel_ids_b"])}
    rec["queue_before"], rec["queue2_before"] = np_(m.queue.clone()), np_(m.queue2.clone())
    caps = {}
    codes = {}
    for const in ref.MODEL.forward_densecl.__code__.co_consts:
        if hasattr(const, "co_name") and const.co_name in ("compute_local_loss", "compute_global_loss",
                                                            "get_query_features", "get_key_features"):
            codes[const.co_name] = const

    calls = {nm: [] for nm in codes}

    def prof(frame, event, arg):
        if event == "return":
            for nm, code in codes.items():
                if frame.f_code is code:
                    d = dict(frame.f_locals)
                    d["__ret"] = arg
                    calls[nm].append(d)
    sys.setprofile(prof)
    loss = m.forward_densecl(visualize=False, step=step, new_epoch=False, **inp)
    sys.setprofile(None)
    caps = {nm: c[0] for nm, c in calls.items()}
    loc, glob = caps["compute_local_loss"], caps["compute_global_loss"]
    rec["q_embed"], rec["k_embed"] = np_(loc["q_embed"]), np_(loc["k_embed"])
    rec["k_local"] = np_(loc["k_local"])
    rec["q_local"] = np_(caps["get_query_features"]["q_local"])      # (b, C, S2), before the row reshape
    rec["q_pixel_ids"], rec["k_pixel_ids"] = np_(loc["q_pixel_ids"]), np_(loc["k_pixel_ids"])
    rec["pos_global_k_idx"] = np_(loc["pos_global_k_idx"])
    rec["pos_local"], rec["neg_local"] = np_(loc["pos_local"]), np_(loc["neg_local"])
    rec["loss_local"] = np_(loc["__ret"])
    rec["q_global"], rec["k_global"] = np_(glob["q"]), np_(glob["k"])
    rec["loss_global"] = np_(glob["__ret"])
    rec["k_local_pooled"] = np_(caps["get_key_features"]["k_local_proj_pooled"])
    rec["loss"] = np_(loss)
    if symmetric:        # second pass (views swapped), reference builder.py:944-972
        loc2, glob2 = calls["compute_local_loss"][1], calls["compute_global_loss"][1]
        for k in ("q_embed", "k_embed", "k_local", "q_pixel_ids", "k_pixel_ids", "pos_global_k_idx", "pos_local"):
            rec[k + "_2"] = np_(loc2[k])
        rec["q_local_2"] = np_(calls["get_query_features"][1]["q_local"])
        rec["loss_local_2"], rec["loss_global_2"] = np_(loc2["__ret"]), np_(glob2["__ret"])
        rec["q_global_2"], rec["k_global_2"] = np_(glob2["q"]), np_(glob2["k"])
        rec["k_local_pooled_2"] = np_(calls["get_key_features"][1]["k_local_proj_pooled"])
        rec["step"] = np.int64(step)
    rec["queue_after"], rec["queue2_after"] = np_(m.queue), np_(m.queue2)
    rec["ptr_after"], rec["ptr2_after"] = np_(m.queue_ptr)[0], np_(m.queue2_ptr)[0]
    rec["cfg"] = np.array([b, h, w, K, stride], dtype=np.int64)
    rec["cfg_f"] = np.array([m.temp_global, m.temp_local, m.lmbd_dense_loss, lmbd_coordinate], dtype=np.float64)
    np.savez_compressed(os.path.join(OUT, name + ".npz"), **rec)
    print(f"{name}: loss={float(loss):.6f} global={float(glob['__ret']):.6f} local={float(loc['__ret']):.6f}")


def run_densecl_overlap_case(name, b, h, w, K, stride, seed, lmbd_coordinate):
    """DenseCL local positives WITH id overlap between the two views (the coordinate mix of builder.py:838-855).
    The reference cannot finish such a step: builder.py:861 calls `.max(dim=2)` on the 2-D tensor
    `corr_map[overlap_pixels, :]` and raises IndexError as soon as one pixel id occurs in both down-sampled maps
    (which is why the other DenseCL fixtures, made from a whole forward pass, contain no overlap).  The mix itself is
    computed BEFORE that line, so this fixture records compute_local_loss's locals at the moment the reference raises:
    the inputs, pos_global_k_idx, overlap_pixels and pos_local after the mix."""
    torch.manual_seed(seed)
    gen = torch.Generator().manual_seed(seed)
    m = blank_model(K, backbone_stride=stride, temp_local=0.2, lmbd_dense=0.5, pretrain_type=PretrainType.PROPOSED_V2,
                    lmbd_coordinate=lmbd_coordinate)
    m.encoder_q, m.encoder_k = DenseEnc(stride), DenseEnc(stride)
    for pq, pk in zip(m.encoder_q.parameters(), m.encoder_k.parameters()):
        pk.data.copy_(pq.data + 0.01 * torch.randn_like(pq))
        pk.requires_grad = False
    inp = synth_inputs(b, h, w, gen)
    # second view = the first view's id grid shifted by whole strides (centre taps coincide where the crops overlap);
    # sample 1 additionally repeats ids (two key pixels carry the id of one query pixel)
    big = torch.arange(1, 4 * h * w + 1).reshape(2 * h, 2 * w)
    pb = []
    for n in range(b):
        dy, dx = stride * (1 + n % 2), stride * (2 - n % 2)
        crop = (big[dy:dy + h, dx:dx + w] + n * 4 * h * w).clone()
        if n % 2:
            crop[:, w // 2:] = crop[:, : w - w // 2]
        pb.append(crop)
    inp["pixel_ids_b"] = torch.stack(pb)
    inp["region_ids_b"] = inp["pixel_ids_b"].clone()
    code = [c for c in ref.MODEL.forward_densecl.__code__.co_consts if getattr(c, "co_name", "") == "compute_local_loss"][0]
    qcode = [c for c in ref.MODEL.forward_densecl.__code__.co_consts if getattr(c, "co_name", "") == "get_query_features"][0]
    grabbed = {}

    def prof(frame, event, arg):
        if event == "return" and frame.f_code is code and "loc" not in grabbed:
            grabbed["loc"], grabbed["returned"] = dict(frame.f_locals), arg is not None
        if event == "return" and frame.f_code is qcode and "q" not in grabbed:
            grabbed["q"] = dict(frame.f_locals)
    sys.setprofile(prof)
    raised = None
    try:
        m.forward_densecl(visualize=False, step=0, new_epoch=False, **inp)
    except IndexError as e:
        raised = str(e)
    finally:
        sys.setprofile(None)
    loc = grabbed["loc"]
    assert raised is not None and not grabbed["returned"], "the reference was expected to raise at builder.py:861"
    assert int(loc["overlap_pixels"].sum()) > 0
    rec = {"q_embed": np_(loc["q_embed"]), "k_embed": np_(loc["k_embed"]), "q_local": np_(grabbed["q"]["q_local"]),
           "k_local": np_(loc["k_local"]), "q_pixel_ids": np_(loc["q_pixel_ids"]), "k_pixel_ids": np_(loc["k_pixel_ids"]),
           "pos_global_k_idx": np_(loc["pos_global_k_idx"]), "overlap_pixels": np_(loc["overlap_pixels"]),
           "pos_local": np_(loc["pos_local"]), "cfg": np.array([b, h, w, K, stride], dtype=np.int64),
           "cfg_f": np.array([m.temp_global, m.temp_local, m.lmbd_dense_loss, lmbd_coordinate], dtype=np.float64)}
    np.savez_compressed(os.path.join(OUT, name + ".npz"), **rec)
    print(f"{name}: reference raised {raised!r}; {int(loc['overlap_pixels'].sum())} overlapping query pixels recorded")


def run_queue_ema_cases():
    rec = {}
    torch.manual_seed(7)
    # enqueue: plain, crossing the wrap boundary, landing exactly on K  (builder.py:569-587)
    for tag, K, n, ptr0 in (("plain", 64, 8, 16), ("wrap", 64, 8, 60), ("exact", 64, 8, 56), ("big", 1024, 48, 1000)):
        m = blank_model(K)
        m.queue_ptr[0] = ptr0
        keys = nn.functional.normalize(torch.randn(n, 128), dim=1)
        rec[f"enq_{tag}_queue_before"], rec[f"enq_{tag}_keys"] = np_(m.queue.clone()), np_(keys)
        rec[f"enq_{tag}_ptr_before"] = np.int64(ptr0)
        m._dequeue_and_enqueue(keys)
        rec[f"enq_{tag}_queue_after"], rec[f"enq_{tag}_ptr_after"] = np_(m.queue), np_(m.queue_ptr)[0]
    # EMA over a 3-tensor toy encoder, two consecutive updates            (builder.py:557-567)
    m = blank_model(64)
    m.encoder_q = nn.Sequential(nn.Conv2d(3, 8, 3), nn.BatchNorm2d(8), nn.Conv2d(8, 5, 1, bias=False))
    m.encoder_k = nn.Sequential(nn.Conv2d(3, 8, 3), nn.BatchNorm2d(8), nn.Conv2d(8, 5, 1, bias=False))
    for i, p in enumerate(m.encoder_q.parameters()):
        rec[f"ema_q_{i}"] = np_(p)
    for i, p in enumerate(m.encoder_k.parameters()):
        rec[f"ema_k0_{i}"] = np_(p)
    for rnd in (1, 2):
        m._momentum_update_key_encoder()
        for i, p in enumerate(m.encoder_k.parameters()):
            rec[f"ema_k{rnd}_{i}"] = np_(p)
    rec["ema_m"] = np.float64(m.momentum)
    # shuffle / unshuffle on one rank                                      (builder.py:609-649)
    x = torch.randn(6, 3, 4, 4)
    torch.manual_seed(11)
    xs, idx_un = m._batch_shuffle_ddp(x)
    torch.manual_seed(11)
    rec["shuf_perm"] = np_(torch.randperm(6))
    rec["shuf_x"], rec["shuf_out"], rec["shuf_idx_unshuffle"] = np_(x), np_(xs), np_(idx_un)
    rec["unshuf_out"] = np_(m._batch_unshuffle_ddp(xs, idx_un))
    np.savez_compressed(os.path.join(OUT, "queue_ema_shuffle.npz"), **rec)
    print("queue_ema_shuffle: ok")


def run_corrmap_kats():
    """The reference's own known-answer cases (tests/test_correlation_mapping.py:15-132),
    evaluated with the reference's get_masked_correlation_map."""
    rec = {}
    rng = np.random.RandomState(0)
    base = torch.arange(1, 4 * 10 * 10 + 1)[torch.from_numpy(rng.permutation(400))].reshape(4, 10, 10)
    a, b = base[:, :5, :5], base[:, 1:6, 2:7]
    ma = torch.zeros(4, 5, 5); ma[:, 2:4, 1:3] = 1
    mb = torch.zeros(4, 5, 5); mb[:, 1:3, 0:2] = 1
    shared = torch.tensor([[[1, 2, 2, 3, 4, 5], [6, 2, 2, 3, 3, 3], [7, 8, 9, 10, 11, 12],
                            [13, 8, 8, 8, 14, 15]]], dtype=torch.float32)
    cases = {"unique": (a, b, ma, mb),
             "shared": (shared[:, 0:3, 1:4], shared[:, 0:3, 2:5],
                        torch.tensor([[[1., 1, 1], [1, 1, 1], [0, 0, 0]]]),
                        torch.tensor([[[1., 0, 0], [1, 0, 0], [1, 0, 0]]]))}
    # a larger random case with repeated ids and float masks
    g = torch.Generator().manual_seed(3)
    big_a = torch.randint(0, 40, (3, 7, 9), generator=g)
    big_b = torch.randint(0, 40, (3, 7, 9), generator=g)
    cases["random"] = (big_a, big_b, (torch.rand(3, 7, 9, generator=g) > 0.4).float(),
                       (torch.rand(3, 7, 9, generator=g) > 0.6).float())
    for tag, (xa, xb, mka, mkb) in cases.items():
        r = ref_cm.get_masked_correlation_map(xa, xb, mka, mkb)
        rec[f"{tag}_map_a"], rec[f"{tag}_map_b"] = np_(xa), np_(xb)
        rec[f"{tag}_mask_a"], rec[f"{tag}_mask_b"] = np_(mka), np_(mkb)
        for k in ("corr_map", "corr_mask", "corr_map_a", "corr_map_b", "corr_map_a_masked",
                  "corr_map_b_masked", "iou", "iou_masked"):
            rec[f"{tag}_{k}"] = np_(r[k])
    np.savez_compressed(os.path.join(OUT, "corrmap_kats.npz"), **rec)
    print("corrmap_kats:", {k: rec[k] for k in rec if k.endswith("iou") or k.endswith("iou_masked")})


def main():
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT="29557")
    torch.distributed.init_process_group("gloo", rank=0, world_size=1)
    run_corrmap_kats()
    run_queue_ema_cases()
    run_cp2_case("cp2_b4_64_k64", b=4, h=64, w=64, K=64, stride=16, seed=0, ptr0=0, keep_ema=True)
    run_cp2_case("cp2_b4_96_k64_wrap_bg", b=4, h=96, w=96, K=64, stride=16, seed=1, ptr0=62,
                 include_background=True)
    run_cp2_case("cp2_b3_80x112_k1024", b=3, h=80, w=112, K=1024, stride=8, seed=2, ptr0=1000,
                 temp_local=0.5, lmbd_dense=0.7)
    run_cp2_case("cp2_proposed_weights", b=2, h=96, w=96, K=64, stride=16, seed=3,
                 pretrain_type=PretrainType.PROPOSED, mapping_type=ref.MappingType.PIXEL_REGION_ID,
                 w_pixel=10.0, w_region=2.0, w_not=0.5, shared_regions=True)
    run_densecl_case("densecl_b2_128_k64", b=2, h=128, w=128, K=64, stride=32, seed=4)
    run_densecl_case("densecl_b2_96_k64_coord", b=2, h=96, w=96, K=64, stride=16, seed=5, lmbd_coordinate=0.3)
    # experimental variants: NegativeType reshaping of the negative dense logits (builder.py:1332-1386) ...
    for nt in ("NONE", "FIXED", "AVERAGE", "MEDIAN", "HARD"):
        run_cp2_case(f"cp2_neg_{nt.lower()}", b=3, h=64, w=80, K=64, stride=16, seed=6, slim=True,
                     pretrain_type=PretrainType.PROPOSED, mapping_type=ref.MappingType.PIXEL_REGION_ID,
                     w_pixel=3.0, w_region=2.0, w_not=1.0, shared_regions=True, temp_local=0.5,
                     negative_type=getattr(ref.NegativeType, nt), negative_scale=2)
    # ... and the symmetric PROPOSED_V2 pass with coordinate mixing (builder.py:944-972)
    run_densecl_case("densecl_v2_symmetric", b=2, h=96, w=96, K=64, stride=16, seed=8, lmbd_coordinate=0.3,
                     symmetric=True, step=0)
    run_densecl_overlap_case("densecl_coord_overlap", b=2, h=96, w=96, K=64, stride=16, seed=9, lmbd_coordinate=0.3)
    torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
