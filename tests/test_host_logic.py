"""CPU: host-side logic that needs no GPU -- CLI surface, config loader, encoder restatement, synthetic loader
contract, MODEL construction and its refusal to run on CPU."""
import os

import pytest
import torch

from cp2_amd import _lib, builder, synthetic
from cp2_amd.config import Config
from cp2_amd.encoder import build_segmentor
from cp2_amd.main import adjust_learning_rate, get_args
from cp2_amd.pretrain_types import PretrainType
from oracle import cp2_oracle as O

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_cli_accepts_reference_flags_and_densecl_overrides():
    a = get_args(["--config", "c.py", "--run_id", "r", "--log_dir", "/tmp", "--data_dirs", "x", "--pretrain_type", "DENSECL",
                  "--lmbd_pixel_corr_weight", "10", "--world-size", "2", "-b", "64", "--cap_queue", "--include_background"])
    assert a.pretrain_type == PretrainType.DENSECL and a.dense_logits_temp == 0.2 and a.lmbd_cp2_dense_loss == 0.5
    assert a.mapping_type == builder.MappingType.CP2 and a.negative_type == builder.NegativeType.NONE
    assert a.lmbd_pixel_corr_weight == 10.0 and a.lmbd_region_corr_weight == 1 and a.queue_size == 65536
    d = get_args(["--config", "c.py", "--run_id", "r", "--log_dir", "/tmp"])
    assert (d.lr, d.momentum, d.weight_decay, d.batch_size, d.epochs) == (0.03, 0.9, 1e-4, 256, 200)


def test_cosine_lr_schedule():
    opt = torch.optim.SGD([torch.nn.Parameter(torch.zeros(1))], lr=0.03)

    class A:
        lr, epochs = 0.03, 200
    assert adjust_learning_rate(opt, 0, A) == pytest.approx(0.03)
    assert adjust_learning_rate(opt, 100, A) == pytest.approx(0.015)
    assert opt.param_groups[0]["lr"] == pytest.approx(0.015)


@pytest.mark.parametrize("name,stride,params_m", [("config_pretrain.py", 16, 66.05), ("config_pretrain_r50_fcn.py", 16, 47.43),
                                                  ("config_pretrain_r18.py", 16, 11.80)])
def test_encoder_restatement_shapes_and_names(name, stride, params_m):
    cfg = Config.fromfile(os.path.join(ROOT, "configs", name))
    enc = build_segmentor(cfg.model, train_cfg=cfg.get("train_cfg"), test_cfg=cfg.get("test_cfg"))
    assert sum(p.numel() for p in enc.parameters()) / 1e6 == pytest.approx(params_m, abs=0.01)
    with torch.no_grad():
        y = enc(torch.rand(2, 3, 64, 64))
        maps = enc.backbone(torch.rand(2, 3, 64, 64))
    assert y.shape == (2, 128, 64 // stride, 64 // stride) and len(maps) == 4
    keys = enc.state_dict().keys()
    for k in ("backbone.conv1.weight", "backbone.bn1.running_mean", "backbone.layer2.0.downsample.0.weight",
              "backbone.layer2.0.downsample.1.weight", "decode_head.conv_seg.weight", "decode_head.contrast_conv.2.weight"):
        assert k in keys, k
    assert not enc.decode_head.conv_seg.weight.requires_grad


def test_synthetic_batch_contract():
    b = synthetic.make_batch(6, 64, 48, "cpu", seed=3)
    assert b["img_a"].shape == (6, 3, 64, 48) and b["pixel_ids_a"].dtype == torch.int64
    assert float(b["img_a"].min()) >= 0 and float(b["img_a"].max()) < 1
    for bg in (b["bg0"], b["bg1"]):
        hole = (bg[:, 0] == 0)
        frac = hole.float().mean((1, 2))
        assert (frac > 0.4).all() and (frac < 0.9).all()
        assert torch.equal(hole, (bg[:, 1] == 0)) and torch.equal(hole, (bg[:, 2] == 0))   # all three channels erased
    assert int(b["pixel_ids_a"].min()) >= 1
    # the two views overlap: the id maps share ids, so the unmasked IoU is strictly between 0 and 1
    iou = O.masked_iou(b["pixel_ids_a"], b["pixel_ids_b"], torch.ones(6, 64 * 48), torch.ones(6, 64 * 48))
    assert ((iou > 0) & (iou < 1)).all()
    assert torch.equal(b["region_ids_a"], b["pixel_ids_a"])
    again = synthetic.make_batch(6, 64, 48, "cpu", seed=3)
    assert all(torch.equal(b[k], again[k]) for k in b)


def test_model_constructs_on_cpu_but_refuses_to_run_there():
    cfg = Config.fromfile(os.path.join(ROOT, "configs", "config_pretrain_r18.py"))
    m = builder.MODEL(cfg, rank=0, K=256, pretrain_from_scratch=True)
    assert m.output_stride == 16 and m.queue.shape == (128, 256) and int(m.queue_ptr) == 0
    assert torch.allclose(m.queue.norm(dim=0), torch.ones(256), atol=1e-5)
    for pq, pk in zip(m.encoder_q.parameters(), m.encoder_k.parameters()):
        assert torch.equal(pq, pk) and not pk.requires_grad
    batch = synthetic.make_batch(2, 64, 64, "cpu", seed=0)
    with pytest.raises(_lib.Cp2LibraryError):
        m(visualize=False, step=0, new_epoch=False, **batch)
    with pytest.raises(AssertionError):
        builder.MODEL(cfg, rank=0, K=256, pretrain_from_scratch=True, lmbd_pixel_corr_weight=10)   # CP2 mapping asserts weights == 1


def test_shipped_miopen_find_db_is_copied_once_and_respects_the_callers_choice(tmp_path, monkeypatch):
    """cp2_amd.miopen_cache: the shipped MIOpen user databases go to a per-user cache directory (never written in the repo),
    files already there are kept (MIOpen appends to them), and a caller who set MIOPEN_USER_DB_PATH or CP2_MIOPEN_DB=0 is left alone."""
    import os
    from cp2_amd import miopen_cache as mc
    shipped = [f for f in os.listdir(mc.SHIPPED) if f.endswith((".udb.txt", ".ufdb.txt"))]
    assert len(shipped) == 2 and all(f.startswith("gfx950") for f in shipped)
    monkeypatch.delenv("MIOPEN_USER_DB_PATH", raising=False)
    monkeypatch.delenv("CP2_MIOPEN_DB", raising=False)
    monkeypatch.setenv("XDG_CACHE_HOME", str(tmp_path))
    dst = mc.use_shipped_find_db()
    assert dst == str(tmp_path / "cp2_amd" / "miopen_db") and os.environ["MIOPEN_USER_DB_PATH"] == dst
    assert sorted(os.listdir(dst)) == sorted(shipped)
    grown = os.path.join(dst, shipped[0])
    with open(grown, "a") as f:
        f.write("appended-by-miopen\n")
    monkeypatch.delenv("MIOPEN_USER_DB_PATH")
    assert mc.use_shipped_find_db() == dst and open(grown).read().endswith("appended-by-miopen\n")
    monkeypatch.setenv("MIOPEN_USER_DB_PATH", "/somewhere/else")
    assert mc.use_shipped_find_db() is None and os.environ["MIOPEN_USER_DB_PATH"] == "/somewhere/else"
    monkeypatch.delenv("MIOPEN_USER_DB_PATH")
    monkeypatch.setenv("CP2_MIOPEN_DB", "0")
    assert mc.use_shipped_find_db() is None and "MIOPEN_USER_DB_PATH" not in os.environ
