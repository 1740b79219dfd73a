"""GPU: `python -m cp2_amd.main` end to end on BASELINE config 1 shapes (ResNet-18, 64x64 crops, queue 1024):
runs two short epochs with the synthetic loader and checks the checkpoint contract the reference's fine-tuning
loader relies on (main.py:528-550,661-670; networks/segment_network.py:79-92)."""
import os
import subprocess
import sys

import pytest
import torch

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.timeout(600)
def test_main_cli_synthetic_and_checkpoint(tmp_path):
    cmd = [sys.executable, "-m", "cp2_amd.main", "--config", os.path.join(ROOT, "configs", "config_pretrain_r18.py"),
           "--run_id", "t", "--log_dir", str(tmp_path), "--synthetic", "--pretrain_from_scratch", "--queue_size", "1024",
           "--img_height", "64", "--img_width", "64", "-b", "8", "--epochs", "2", "--steps_per_epoch", "3", "--lr", "0.01",
           "--dist-url", "tcp://127.0.0.1:29533", "--print-freq", "1", "--include_background"]
    env = dict(os.environ, PYTHONPATH=ROOT)
    res = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=550)
    assert res.returncode == 0, res.stdout[-2000:] + res.stderr[-4000:]
    assert "Epoch: [1][2/3]" in res.stdout
    ck_path = tmp_path / "t" / "checkpoint.ckpt"
    assert ck_path.exists() and (tmp_path / "t" / "6_1_checkpoint.ckpt").exists()
    ck = torch.load(ck_path, map_location="cpu", weights_only=False)
    assert ck["epoch"] == 2 and ck["pretrain_type"] == "CP2" and ck["backbone_type"] == "DEEPLABV3"
    sd = ck["state_dict"]
    for k in ("module.queue", "module.queue_ptr", "module.queue2", "module.encoder_q.backbone.conv1.weight",
              "module.encoder_q.backbone.bn1.num_batches_tracked", "module.encoder_k.decode_head.contrast_conv.2.weight"):
        assert k in sd, k
    assert sd["module.queue"].shape == (128, 1024) and int(sd["module.queue_ptr"]) == (6 * 8) % 1024
    assert int(sd["module.encoder_q.backbone.bn1.num_batches_tracked"]) >= 6          # lazily counted fused-BN steps
    # what reference segment_network.py:84-92 does with it: keep encoder_q.*, strip the prefix, drop conv_seg
    enc = {k.replace("module.encoder_q.", ""): v for k, v in sd.items() if "encoder_q." in k and "conv_seg" not in k}
    from cp2_amd.config import Config
    from cp2_amd.encoder import build_segmentor
    m = build_segmentor(Config.fromfile(os.path.join(ROOT, "configs", "config_pretrain_r18.py")).model)
    missing, unexpected = m.load_state_dict(enc, strict=False)
    assert not unexpected and all("conv_seg" in k for k in missing)
    assert all(torch.isfinite(v).all() for v in enc.values() if v.dtype.is_floating_point)
    assert len(ck["optimizer"]["state"]) > 20


@pytest.mark.timeout(600)
def test_main_cli_tensor_dataset_on_device_augmentation(tmp_path):
    """`cp2_amd.main` without --synthetic: a uint8 image set with region maps stays in HBM, every batch is
    cropped / flipped / erased on the device; PROPOSED weights so the region ids are consumed, pixel ids at stride 2."""
    g = torch.Generator().manual_seed(0)
    n = 48
    torch.save({"images": torch.randint(1, 256, (n, 3, 80, 96), dtype=torch.uint8, generator=g),
                "region_ids": torch.randint(0, 12, (n, 80, 96), generator=g)}, tmp_path / "ds.pt")
    cmd = [sys.executable, "-m", "cp2_amd.main", "--config", os.path.join(ROOT, "configs", "config_pretrain_r18.py"),
           "--run_id", "d", "--log_dir", str(tmp_path), "--tensor_dataset", str(tmp_path / "ds.pt"), "--pretrain_from_scratch",
           "--queue_size", "256", "--img_height", "64", "--img_width", "64", "-b", "8", "--epochs", "2", "--lr", "0.01",
           "--dist-url", "tcp://127.0.0.1:29534", "--print-freq", "1", "--pretrain_type", "PROPOSED", "--mapping_type",
           "PIXEL_REGION_ID", "--lmbd_pixel_corr_weight", "3", "--lmbd_region_corr_weight", "2", "--pixel_ids_stride", "2",
           "--negative_type", "AVERAGE"]
    res = subprocess.run(cmd, cwd=ROOT, env=dict(os.environ, PYTHONPATH=ROOT), capture_output=True, text=True, timeout=550)
    assert res.returncode == 0, res.stdout[-2000:] + res.stderr[-4000:]
    assert "Epoch: [1][5/6]" in res.stdout                       # 48 images / batch 8 = 6 steps per epoch
    ck = torch.load(tmp_path / "d" / "checkpoint.ckpt", map_location="cpu", weights_only=False)
    assert ck["epoch"] == 2 and ck["pretrain_type"] == "PROPOSED" and int(ck["state_dict"]["module.queue_ptr"]) == (12 * 8) % 256


@pytest.mark.timeout(900)
@pytest.mark.parametrize("grad_sync", ["flat", "ddp"])
def test_main_cli_two_ranks_one_device_and_resume(tmp_path, grad_sync):
    """`cp2_amd.main --world-size 2` (reference main.py:732 spawns its ranks the same way) as a rehearsal on one GPU over
    gloo: gradient averaging by ddp.FlatDDP (default) or torch DDP, checkpoint with the reference's "module." keys written by
    rank 0, then a second run resumes from it for one more epoch."""
    base = [sys.executable, "-m", "cp2_amd.main", "--config", os.path.join(ROOT, "configs", "config_pretrain_r18.py"),
            "--log_dir", str(tmp_path), "--synthetic", "--pretrain_from_scratch", "--queue_size", "256",
            "--img_height", "64", "--img_width", "64", "-b", "8", "--steps_per_epoch", "3", "--lr", "0.01",
            "--print-freq", "1", "--world-size", "2", "--dist-backend", "gloo", "--one_device", "--grad_sync", grad_sync]
    env = dict(os.environ, PYTHONPATH=ROOT)
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK"):
        env.pop(k, None)
    port = 29540 + (1 if grad_sync == "ddp" else 0)
    res = subprocess.run(base + ["--run_id", "a", "--epochs", "2", "--dist-url", f"tcp://127.0.0.1:{port}"], cwd=ROOT, env=env,
                         capture_output=True, text=True, timeout=420)
    assert res.returncode == 0, res.stdout[-2000:] + res.stderr[-4000:]
    assert "Epoch: [1][2/3]" in res.stdout
    ck_path = tmp_path / "a" / "checkpoint.ckpt"
    ck = torch.load(ck_path, map_location="cpu", weights_only=False)
    sd = ck["state_dict"]
    assert ck["epoch"] == 2 and "module.encoder_q.backbone.conv1.weight" in sd and "module.queue" in sd
    assert int(sd["module.queue_ptr"]) == (6 * 8) % 256                                  # both ranks' keys, every step
    assert all(torch.isfinite(v).all() for v in sd.values() if v.dtype.is_floating_point)
    res = subprocess.run(base + ["--run_id", "b", "--epochs", "3", "--resume", str(ck_path), "--dist-url", f"tcp://127.0.0.1:{port + 2}"],
                         cwd=ROOT, env=env, capture_output=True, text=True, timeout=420)
    assert res.returncode == 0, res.stdout[-2000:] + res.stderr[-4000:]
    assert "Epoch: [2][2/3]" in res.stdout and "Epoch: [1]" not in res.stdout
    ck2 = torch.load(tmp_path / "b" / "checkpoint.ckpt", map_location="cpu", weights_only=False)
    assert ck2["epoch"] == 3 and int(ck2["state_dict"]["module.queue_ptr"]) == (9 * 8) % 256
    w0, w1 = sd["module.encoder_q.backbone.conv1.weight"], ck2["state_dict"]["module.encoder_q.backbone.conv1.weight"]
    assert not torch.equal(w0, w1)                                                       # the resumed run kept training
