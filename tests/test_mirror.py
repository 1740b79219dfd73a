"""SURVEY 8f rank 4, second half: the supervised CutPaste / mirror pre-training path.
CPU part: the oracle restatement (oracle/mirror_oracle.py) against the goldens recorded from the reference's own
CutPasteDataset / MirrorModule (tests/golden/make_mirror_goldens.py) and against Pillow itself; the host-side sampler
of cp2_amd/mirror.py against the same goldens' random streams.  GPU part: cp2_cutpaste bit-exact against the goldens
and the oracle, cp2_mirror_loss against the goldens (loss 1e-6, gradients 1e-6 * max) and the oracle at larger
shapes, the MirrorModule training step."""
import os

import numpy as np
import pytest
import torch

from cp2_amd import mirror as M
from oracle import mirror_oracle as MO

CFG = dict(min_area_scale=0.02, max_area_scale=0.15, min_aspect_ratio=1 / 3, max_aspect_ratio=4 / 3)
CASES = [("regular", "OUTPUT"), ("scar", "OUTPUT"), ("multi", "OUTPUT"), ("none_variant", "NONE")]


@pytest.fixture(scope="module")
def gold(golden_dir):
    return np.load(os.path.join(golden_dir, "mirror_cutpaste.npz")), np.load(os.path.join(golden_dir, "mirror_loss.npz"))


def _cfg(g, name):
    rot = g[name + ".rotation"]
    return dict(CFG, min_rotation=float(rot[0]), max_rotation=float(rot[1]))


# ------------------------------------------------------------------------------------------------------ CPU
def test_oracle_rotation_equals_pillow():
    Image = pytest.importorskip("PIL.Image")
    rng = np.random.RandomState(0)
    for it in range(1500):
        w, h = rng.randint(1, 60), rng.randint(1, 60)
        ang = float(rng.choice([0, 90, 180, 270, 360, -90, 450])) if it % 10 == 0 else rng.uniform(-400, 400)
        a = rng.randint(0, 256, (h, w, 3), dtype=np.uint8)
        ref = np.asarray(Image.fromarray(a).rotate(ang, expand=True))
        refm = np.asarray(Image.new("L", (w, h), 255).rotate(ang, expand=True))
        out, valid = MO.rotate_nearest_expand(a, ang)
        assert out.shape == ref.shape and np.array_equal(out, ref) and np.array_equal(valid, refm > 0), (w, h, ang)


def test_product_rotation_matrix_equals_oracle():
    rng = np.random.RandomState(1)
    for it in range(4000):
        w, h = rng.randint(1, 700), rng.randint(1, 700)
        ang = float(rng.choice([0, 90, 180, 270, 360, -90, -180])) if it % 10 == 0 else rng.uniform(-720, 720)
        assert M.rotate_matrix(w, h, ang) == MO.rotate_geometry(w, h, ang), (w, h, ang)


@pytest.mark.parametrize("name,variant", CASES)
def test_oracle_cutpaste_equals_reference_goldens(gold, name, variant):
    g = gold[0]
    imgs = g["images"]
    np.random.seed(int(g[name + ".seed"]))
    for i, (idx, cls) in enumerate(zip(g[name + ".index"], g[name + ".targets"])):
        mir = imgs[np.random.randint(len(imgs))] if variant == "OUTPUT" else None
        img, m, mask = MO.cutpaste_item(imgs[idx], mir, int(cls), int(g[name + ".max_num_patches"]), _cfg(g, name))
        assert np.array_equal(img, g[name + ".img"][i]) and np.array_equal(mask, g[name + ".mask"][i])
        if variant == "OUTPUT":
            assert np.array_equal(m, g[name + ".mirror"][i])
    assert g["scar.mask"].max() == 2 and g["multi.mask"].max() == 1


@pytest.mark.parametrize("name", ["c2", "c3", "c3_none"])
def test_oracle_loss_equals_reference_goldens(gold, name):
    L = gold[1]
    s = torch.tensor(L[name + ".s_logits"], requires_grad=True)
    H, W = (int(v) for v in L[name + ".image_hw"])
    t = torch.tensor(L[name + ".t_logits"], requires_grad=True) if name + ".t_logits" in L else None
    su = MO.resize_logits(s, (H, W))
    assert np.array_equal(su.detach().numpy(), L[name + ".s_up"])
    r = MO.mirror_losses(su, None if t is None else MO.resize_logits(t, (H, W)), torch.tensor(L[name + ".masks"]),
                         float(L[name + ".T"]), float(L[name + ".lmbd"]))
    r["loss"].backward()
    assert abs(float(r["loss"]) - float(L[name + ".loss"])) < 1e-6
    assert abs(float(r["class_loss"]) - float(L[name + ".class_loss"])) < 1e-6
    assert abs(float(r["compare_loss"]) - float(L[name + ".compare_loss"])) < 1e-6
    assert np.allclose(s.grad.numpy(), L[name + ".grad_s"], atol=1e-9)
    n = s.shape[0]
    assert np.array_equal(r["argmax"][:n].numpy(), L[name + ".s_argmax"])
    if t is not None:
        assert np.allclose(t.grad.numpy(), L[name + ".grad_t"], atol=1e-9)


def _sampler(g, name, variant, n_images):
    s = M.CutPasteSampler.__new__(M.CutPasteSampler)      # the goldens fix the class targets; skip the constructor's draw
    cfg = _cfg(g, name)
    s.n, s.rng = n_images, np.random
    s.min_area_scale, s.max_area_scale = cfg["min_area_scale"], cfg["max_area_scale"]
    s.min_aspect_ratio, s.max_aspect_ratio = cfg["min_aspect_ratio"], cfg["max_aspect_ratio"]
    s.min_rotation, s.max_rotation = cfg["min_rotation"], cfg["max_rotation"]
    s.mirror_variant, s.max_num_patches = M.MirrorVariant[variant], int(g[name + ".max_num_patches"])
    return s


@pytest.mark.parametrize("name,variant", CASES)
def test_host_sampler_draws_in_the_reference_order(gold, name, variant):
    """Same seed -> the sampler's tables describe exactly the patches the oracle (pinned above) draws."""
    g = gold[0]
    H, W = g["images"].shape[1:3]
    n = len(g["images"])
    targets, index = g[name + ".targets"], g[name + ".index"]
    s = _sampler(g, name, variant, n)
    s.targets = np.zeros(n, dtype=np.int64)
    np.random.seed(int(g[name + ".seed"]))
    got = []
    for i, idx in enumerate(index):                        # item by item: the golden's targets are per ITEM, not per image
        s.targets[idx] = targets[i]
        got.append(s.draw_item(int(idx), H, W))
    np.random.seed(int(g[name + ".seed"]))
    for i, idx in enumerate(index):
        mir = np.random.randint(n) if variant == "OUTPUT" else -1
        want = []
        if targets[i] != 0:
            want.append(MO.draw_patch(H, W, int(targets[i]), **_cfg(g, name)))
            for _ in range(np.random.randint(s.max_num_patches)):
                want.append(MO.draw_patch(H, W, int(targets[i]), **_cfg(g, name)))
        assert got[i][0] == mir and len(got[i][1]) == len(want)
        for row, p in zip(got[i][1], want):
            assert row[:7] == [p["cls"], p["px"], p["py"], p["pw"], p["ph"], p["x_pos"], p["y_pos"]]
            assert tuple(row[7:9]) == MO.rotate_geometry(p["pw"], p["ph"], p["rotation"])[:2]


def test_sampler_constructor_draws_targets_like_the_reference():
    np.random.seed(5)
    s = M.CutPasteSampler(1000, 0.02, 0.15, 1 / 3, 4 / 3, 0, 0, M.MirrorVariant.OUTPUT, 3, 1)
    np.random.seed(5)
    want = np.random.choice([0, 1, 2], size=1000, replace=True, p=[0.1, 0.45, 0.45])   # pretrain_dataset.py:264-269
    assert np.array_equal(s.targets, want)
    with pytest.raises(AssertionError):
        M.CutPasteSampler(10, 0.02, 0.15, 1 / 3, 4 / 3, 0, 0, M.MirrorVariant.OUTPUT, 3, 2)


def test_mirror_cli_accepts_reference_flags():
    from cp2_amd import mirror_pretrain as MP
    a = MP.get_args(["--run_id", "r", "--log_dir", "/tmp/x", "--variant", "NONE", "--max_num_patches", "2", "--lemon_data",
                     "--softmax_temp", "3", "--lmbd_compare_loss", "0.1", "--num_classes", "2", "--batch-size", "4"])
    assert a.variant == M.MirrorVariant.NONE and a.img_x_size == 544 and a.img_y_size == 1024 and a.max_num_patches == 1
    assert a.max_area_scale == 0.007 and a.softmax_temp == 3 and a.batch_size == 4


# ------------------------------------------------------------------------------------------------------ GPU
gpu = pytest.mark.gpu


def _run_cutpaste(g, name, variant, dev):
    from cp2_amd import ops
    imgs = torch.from_numpy(g["images"]).to(dev)
    H, W = imgs.shape[1:3]
    s = _sampler(g, name, variant, len(imgs))
    targets, index = g[name + ".targets"], g[name + ".index"]
    np.random.seed(int(g[name + ".seed"]))
    outs = []
    for i, idx in enumerate(index):                        # one item per call so that the per-item targets apply
        s.targets = np.zeros(len(imgs), dtype=np.int64)
        s.targets[idx] = targets[i]
        outs.append(M.cutpaste_batch(imgs, s, [int(idx)]))
    return outs


@gpu
@pytest.mark.parametrize("name,variant", CASES)
def test_gpu_cutpaste_bit_exact_vs_reference_goldens(gold, name, variant):
    g = gold[0]
    outs = _run_cutpaste(g, name, variant, "cuda")
    for i, (img, mir, mask) in enumerate(outs):
        assert torch.equal(img[0].cpu(), torch.from_numpy(g[name + ".img"][i]))
        assert torch.equal(mask[0].cpu(), torch.from_numpy(g[name + ".mask"][i]))
        if variant == "OUTPUT":
            assert torch.equal(mir[0].cpu(), torch.from_numpy(g[name + ".mirror"][i]))
        else:
            assert mir is None


@gpu
@pytest.mark.parametrize("hw", [(96, 128), (75, 101), (512, 512)])
def test_gpu_cutpaste_batch_vs_oracle(hw):
    """Whole batches (several rounds, rotated scars, odd widths = the scalar kernel) against the oracle."""
    H, W = hw
    rng = np.random.RandomState(3)
    imgs = rng.randint(0, 256, (9, H, W, 3), dtype=np.uint8)
    dev_imgs = torch.from_numpy(imgs).cuda()
    for variant, ncls, maxp, rot in [("OUTPUT", 3, 1, (-60, 60)), ("OUTPUT", 2, 4, (0, 0)), ("NONE", 3, 1, (0, 360))]:
        np.random.seed(21)
        s = M.CutPasteSampler(len(imgs), 0.02, 0.15, 1 / 3, 4 / 3, rot[0], rot[1], M.MirrorVariant[variant], ncls, maxp)
        idx = [4, 0, 8, 3, 3, 7, 1]
        st = np.random.get_state()
        img, mir, mask = M.cutpaste_batch(dev_imgs, s, idx)
        np.random.set_state(st)
        cfg = dict(CFG, min_rotation=rot[0], max_rotation=rot[1])
        for b, i in enumerate(idx):
            m_src = imgs[np.random.randint(len(imgs))] if variant == "OUTPUT" else None
            wi, wm, wmask = MO.cutpaste_item(imgs[i], m_src, int(s.targets[i]), maxp, cfg)
            assert np.array_equal(img[b].cpu().numpy(), wi) and np.array_equal(mask[b].cpu().numpy(), wmask)
            if variant == "OUTPUT":
                assert np.array_equal(mir[b].cpu().numpy(), wm)


@gpu
@pytest.mark.parametrize("name", ["c2", "c3", "c3_none"])
def test_gpu_mirror_loss_vs_reference_goldens(gold, name):
    L = gold[1]
    H, W = (int(v) for v in L[name + ".image_hw"])
    s = torch.tensor(L[name + ".s_logits"], device="cuda", requires_grad=True)
    two = name + ".t_logits" in L
    t = torch.tensor(L[name + ".t_logits"], device="cuda", requires_grad=True) if two else None
    up = lambda x: torch.nn.functional.interpolate(x, size=(H, W), mode="bilinear", align_corners=False)  # noqa: E731
    C = s.shape[1]
    conf = torch.zeros(C, C, dtype=torch.int64, device="cuda")
    masks = torch.tensor(L[name + ".masks"], device="cuda")
    loss, stats = M.mirror_loss(up(s), up(t) if two else None, masks, float(L[name + ".T"]), float(L[name + ".lmbd"]), conf)
    loss.backward()
    assert abs(float(loss) - float(L[name + ".loss"])) < 2e-6
    assert abs(float(stats["class_loss"]) - float(L[name + ".class_loss"])) < 2e-6
    assert abs(float(stats["compare_loss"]) - float(L[name + ".compare_loss"])) < 2e-6
    gs = L[name + ".grad_s"]
    assert np.abs(s.grad.cpu().numpy() - gs).max() <= 1e-6 * np.abs(gs).max() + 1e-10
    n = s.shape[0]
    assert np.array_equal(stats["argmax"][:n].cpu().numpy(), L[name + ".s_argmax"])
    if two:
        gt = L[name + ".grad_t"]
        assert np.abs(t.grad.cpu().numpy() - gt).max() <= 1e-6 * np.abs(gt).max() + 1e-10
    want = MO.confusion(stats["argmax"].cpu(), torch.cat([masks.cpu()] * (2 if two else 1)), C)
    assert torch.equal(conf.cpu(), want)


@gpu
@pytest.mark.parametrize("N,C,H,W,two", [(4, 2, 512, 512, True), (3, 3, 250, 333, True), (2, 8, 64, 96, True), (5, 3, 128, 128, False)])
def test_gpu_mirror_loss_random_vs_oracle(N, C, H, W, two):
    g = torch.Generator().manual_seed(N * 100 + C)
    s = (torch.randn(N, C, H, W, generator=g) * 3).requires_grad_(True)
    t = (torch.randn(N, C, H, W, generator=g) * 3).requires_grad_(True) if two else None
    masks = torch.randint(0, C, (N, H, W), generator=g)
    T, lmbd = 2.0, 0.3
    r = MO.mirror_losses(s, t, masks, T, lmbd)
    r["loss"].backward()
    sd = s.detach().cuda().requires_grad_(True)
    td = t.detach().cuda().requires_grad_(True) if two else None
    loss, stats = M.mirror_loss(sd, td, masks.cuda(), T, lmbd)
    (loss * 2.0).backward()                                     # the upstream gradient is applied
    assert abs(float(loss) - float(r["loss"])) < 2e-6 * max(1.0, abs(float(r["loss"])))
    assert abs(float(stats["compare_loss"]) - float(r["compare_loss"])) < 2e-6
    assert torch.equal(stats["argmax"].cpu(), r["argmax"])
    gs = s.grad.numpy() * 2.0
    assert np.abs(sd.grad.cpu().numpy() - gs).max() <= 2e-6 * np.abs(gs).max()
    if two:
        gt = t.grad.numpy() * 2.0
        assert np.abs(td.grad.cpu().numpy() - gt).max() <= 2e-6 * np.abs(gt).max()


@gpu
def test_gpu_mirror_module_training_steps_and_checkpoint(tmp_path):
    """MirrorModule with the reference's constructor arguments: a few Adam steps on device-made CutPaste batches reduce
    the loss; the checkpoint reloads through load_pretrained (PretrainType.MIRROR, segment_network.py:94-100)."""
    from cp2_amd.config import Config
    from cp2_amd.pretrain_types import PretrainType
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    cfg = Config.fromfile(os.path.join(root, "configs", "config_pretrain_r18.py"))
    cfg.model.decode_head.num_classes = 2
    cfg.model.decode_head.contrast = False
    torch.manual_seed(0)
    H = W = 64
    mod = M.MirrorModule(model_config=cfg, pretrain_type=PretrainType.RANDOM, learning_rate=1e-3, weight_decay=1e-4, num_classes=2,
                         image_shape=(3, H, W), lmbd_compare_loss=0.01, softmax_temp=2, mirror_variant=M.MirrorVariant.OUTPUT).cuda().train()
    opt = mod.configure_optimizers()["optimizer"]
    rng = np.random.RandomState(0)
    images = torch.from_numpy(rng.randint(0, 256, (16, H, W, 3), dtype=np.uint8)).cuda()
    np.random.seed(0)
    sampler = M.CutPasteSampler(16, 0.02, 0.15, 1 / 3, 4 / 3, 0, 0, M.MirrorVariant.OUTPUT, 2, 1)
    losses = []
    for step in range(12):
        batch = M.cutpaste_batch(images, sampler, list(range(8)) if step % 2 == 0 else list(range(8, 16)))
        loss = mod.training_step(batch, step)
        opt.zero_grad(set_to_none=True)
        loss.backward()
        opt.step()
        losses.append(float(loss))
    assert all(l == l for l in losses) and min(losses[-4:]) < losses[0]
    assert int(mod.confusion_train.sum()) == 12 * 2 * 8 * H * W
    m = mod.metrics(M.Stage.TRAIN)
    assert set(m) == {"train_jaccard", "train_dice", "train_precision", "train_recall", "train_f1"} and int(mod.confusion_train.sum()) == 0
    logits, am = mod(batch[0])
    assert logits.shape == (8, 2, H, W) and am.shape == (8, H, W)
    path = tmp_path / "checkpoint.ckpt"
    torch.save({"state_dict": mod.state_dict(), "pretrain_type": "MIRROR"}, path)
    cfg.model.backbone.init_cfg = dict(type="Pretrained", checkpoint=str(path))
    mod2 = M.MirrorModule(model_config=cfg, pretrain_type=PretrainType.MIRROR, learning_rate=1e-3, weight_decay=1e-4, num_classes=2,
                          image_shape=(3, H, W), lmbd_compare_loss=0.01, softmax_temp=2, mirror_variant=M.MirrorVariant.NONE).cuda()
    k = "model.backbone.layer1.0.conv1.weight"
    assert torch.equal(mod2.state_dict()[k], mod.state_dict()[k])


@gpu
def test_gpu_mirror_pretrain_cli_runs_and_writes_checkpoint(tmp_path):
    from cp2_amd import mirror_pretrain as MP
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    a = MP.get_args(["--run_id", "r0", "--log_dir", str(tmp_path), "--synthetic", "40", "--config",
                     os.path.join(root, "configs", "config_pretrain_r18.py"), "-x", "64", "-y", "96", "--epochs", "2",
                     "--batch-size", "8", "--num_classes", "3", "--min_rotation", "-30", "--max_rotation", "30"])
    best = MP.main(a)
    assert best == best and best < 10
    ck = torch.load(tmp_path / "r0" / "checkpoint.ckpt", map_location="cpu")
    assert ck["pretrain_type"] == "MIRROR" and any(k.startswith("model.backbone.") for k in ck["state_dict"])


@gpu
def test_gpu_mirror_ops_reject_cpu_tensors():
    from cp2_amd import _lib, ops
    with pytest.raises(_lib.Cp2LibraryError):
        ops.cutpaste(torch.zeros(2, 8, 8, 3, dtype=torch.uint8), None, torch.zeros(2, 20, dtype=torch.int32))
    with pytest.raises(_lib.Cp2LibraryError):
        ops.mirror_loss(torch.zeros(1, 2, 4, 4), None, torch.zeros(1, 4, 4, dtype=torch.int64), 2.0, 0.1)


def test_ddp_shards_have_the_same_number_of_steps_on_every_rank():
    """perm[rank::world] gave ranks different step counts when len(train) % world != 0 (one rank then waits for
    an all-reduce its peers never start).  Every rank now takes the same number of samples, whole batches when possible."""
    from cp2_amd.mirror_pretrain import shard_len
    for n, world, b in ((103, 4, 5), (17, 2, 8), (64, 8, 8), (9, 4, 8), (1000, 3, 16)):
        per = shard_len(n, world, b)
        shards = [list(range(n))[r:per * world:world] for r in range(world)]
        assert len({len(s_) for s_ in shards}) == 1 and len(shards[0]) == per
        assert per <= n // world and (per % b == 0 or per < b)
        flat = [i for s_ in shards for i in s_]
        assert len(set(flat)) == len(flat)                          # disjoint
