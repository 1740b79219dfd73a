"""Host-side cost of one training step: cProfile over the bench configuration's steps (no device synchronisation inside the
profiled region, so what is counted is the time the CPU needs to ENQUEUE a step -- bench.py's host_issue_ms_per_step).
usage: python tools/host_profile.py [steps=30] [top=45] [cfg2|cfg5]"""
import cProfile
import contextlib
import os
import pstats
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from cp2_amd import builder, synthetic  # noqa: E402
from cp2_amd.config import Config  # noqa: E402
from cp2_amd.engine import TrainStep  # noqa: E402
from cp2_amd.optim import FlatSGD  # noqa: E402
from cp2_amd.pretrain_types import PretrainType  # noqa: E402

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 30
top = int(sys.argv[2]) if len(sys.argv) > 2 else 45
dev = torch.device("cuda", 0)
torch.manual_seed(0)
torch.backends.cudnn.benchmark = True
workload = sys.argv[3] if len(sys.argv) > 3 else "cfg2"
densecl = workload == "cfg5"                       # BASELINE configs[4]: DenseCL on the backbone of configs/config_pretrain.py
cfg = Config.fromfile(os.path.join(ROOT, "configs", "config_pretrain.py" if densecl else "config_pretrain_r50_fcn.py"))
extra = dict(instance_logits_temp=0.2, dense_logits_temp=0.2, lmbd_cp2_dense_loss=0.5) if densecl else {}
with contextlib.redirect_stdout(sys.stderr):
    model = builder.MODEL(cfg, rank=0, K=65536, pretrain_from_scratch=True,
                          pretrain_type=PretrainType.DENSECL if densecl else PretrainType.CP2, device=dev,
                          amp_dtype=torch.bfloat16, channels_last=True, **extra).to(dev).train()
model.encoder_q.to(memory_format=torch.channels_last)
model.encoder_k.to(memory_format=torch.channels_last)
opt = FlatSGD(model, 0.03, momentum=0.9, weight_decay=1e-4)
runner = TrainStep(model, opt)
batches = [synthetic.make_batch(32, 224, 224, dev, seed=i) for i in range(4)]
for i in range(12):
    runner(batches[i % 4])
torch.cuda.synchronize()
# sections, host time only
marks = {}
orig_fwd = model.forward


def timed(name, fn):
    def wrap(*a, **k):
        t = time.perf_counter()
        r = fn(*a, **k)
        marks[name] = marks.get(name, 0.0) + time.perf_counter() - t
        return r
    return wrap


model._encode_key = timed("key encoder (hipGraph replay)", model._encode_key)
model._key_forward = timed("key pass (hipGraph replay)", model._key_forward) if densecl else model._key_forward
if densecl:
    model.encoder_q.backbone.forward = timed("query backbone forward", model.encoder_q.backbone.forward)
    model.encoder_q.neck.forward = timed("query neck forward", model.encoder_q.neck.forward)
    builder.densecl_local_positives = timed("positive selection (cp2_densecl_match)", builder.densecl_local_positives)
    builder.queue_infonce = timed("queue_infonce (global + local, statistics)", builder.queue_infonce)
    model._log_step = timed("_log_step", model._log_step)
model._momentum_update_key_encoder = timed("EMA", model._momentum_update_key_encoder)
model.encoder_q.forward = timed("query encoder forward", model.encoder_q.forward)
opt.step = timed("optimizer step", opt.step)
bw = torch.Tensor.backward
torch.Tensor.backward = timed("backward()", bw)
t0 = time.perf_counter()
pr = cProfile.Profile()
pr.enable()
for i in range(steps):
    runner(batches[i % 4])
pr.disable()
host = (time.perf_counter() - t0) / steps * 1e3
torch.cuda.synchronize()
wall = (time.perf_counter() - t0) / steps * 1e3
torch.Tensor.backward = bw
print(f"host issue {host:.3f} ms/step (under cProfile), wall {wall:.3f} ms/step")
for k, v in sorted(marks.items(), key=lambda kv: -kv[1]):
    print(f"  {v / steps * 1e3:8.3f} ms/step  {k}")
st = pstats.Stats(pr)
st.sort_stats("tottime").print_stats(top)
