"""BASELINE config 5 (DenseCL: per-pixel q.k against queue2, no copy-paste mask): the training step end to end on one
GPU -- ResNet-50 backbone + DenseCL neck, 224^2, queue 65536, 32 img/GPU, bf16 encoders -- images/s and, under
rocprofv3 --kernel-trace, the kernel breakdown.    python tools/bench_densecl.py [steps]"""
import os
import sys
import time
os.environ.setdefault("MIOPEN_FIND_MODE", "FAST")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from cp2_amd import builder, synthetic
from cp2_amd.config import Config
from cp2_amd.engine import TrainStep
from cp2_amd.main import make_optimizer
from cp2_amd.pretrain_types import PretrainType

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 30
torch.manual_seed(0)
torch.backends.cudnn.benchmark = True
dev = torch.device("cuda", 0)
cfg = Config.fromfile(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "configs", "config_moco.py"))
model = builder.MODEL(cfg, rank=0, K=65536, pretrain_from_scratch=True, pretrain_type=PretrainType.DENSECL, device=dev,
                      instance_logits_temp=0.2, dense_logits_temp=0.2, lmbd_cp2_dense_loss=0.5,
                      amp_dtype=torch.bfloat16, channels_last=True).to(dev)
model.encoder_q.to(memory_format=torch.channels_last); model.encoder_k.to(memory_format=torch.channels_last)
model.train()


class A:
    lr, momentum, weight_decay, optim = 0.03, 0.9, 1e-4, "sgd"


opt = make_optimizer(list(model.parameters()), A, dev, capturable=False, model=model)
runner = TrainStep(model, opt)
batches = [synthetic.make_batch(32, 224, 224, dev, seed=i) for i in range(4)]
for i in range(8):
    runner(batches[i % 4])
torch.cuda.synchronize()
t0 = time.perf_counter()
for i in range(steps):
    loss = runner(batches[i % 4])
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / steps
print(f"DenseCL step: {dt * 1e3:.2f} ms, {32 / dt:.1f} img/s, loss {float(loss):.4f}")
