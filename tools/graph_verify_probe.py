"""Whole-step hipGraph with / without the eager verification probe in front of the capture (the
open item of round 2): the REAL step (EMA, enqueue, FlatSGD at the training learning rate) for STEPS steps, printing per
step the loss and the norms of the state the step advances, so the two trajectories can be laid side by side.
    python tools/graph_verify_probe.py [0|1]        (verify off / on)
"""
import os
import sys
os.environ.setdefault("MIOPEN_FIND_MODE", "FAST")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from cp2_amd import builder, synthetic
from cp2_amd.config import Config
from cp2_amd.engine import TrainStep
from cp2_amd.main import make_optimizer
from cp2_amd.pretrain_types import PretrainType

verify = (sys.argv[1] if len(sys.argv) > 1 else "1") == "1"
graph = os.environ.get("GRAPH", "1") == "1"
torch.manual_seed(0)
torch.backends.cudnn.benchmark = True
dev = torch.device("cuda", 0)
cfg = Config.fromfile("configs/config_pretrain_r50_fcn.py")
model = builder.MODEL(cfg, rank=0, K=65536, pretrain_from_scratch=True, pretrain_type=PretrainType.CP2, device=dev,
                      amp_dtype=torch.bfloat16, channels_last=True).to(dev)
model.encoder_q.to(memory_format=torch.channels_last); model.encoder_k.to(memory_format=torch.channels_last)
model.train()
if os.environ.get("QUART", "1") != "1":
    model.log_quartiles = False


class A:
    lr, momentum, weight_decay, optim = 0.03, 0.9, 1e-4, "sgd"


opt = make_optimizer(list(model.parameters()), A, dev, capturable=graph, model=model)
runner = TrainStep(model, opt, use_graph=graph, warmup_steps=3, verify=verify, reverify_every=0)
batches = [synthetic.make_batch(32, 224, 224, dev, seed=i) for i in range(4)]
for i in range(int(os.environ.get("STEPS", "24"))):
    torch.manual_seed(1000 + i)
    loss = runner(batches[i % 4])
    torch.cuda.synchronize()
    kind = "graph" if runner.graph is not None else "eager"
    bn = model.encoder_q.backbone.bn1
    print(f"step {i:2d} {kind} loss {float(loss):.4f} |q| {float(model._flat_q.norm()):.4f} |k| {float(model._flat_k.norm()):.4f} "
          f"|q16| {float(model._flat_q_bf16.float().norm()):.4f} |k16| {float(model._flat_k_bf16.float().norm()):.4f} "
          f"|mom| {float(opt._buf.norm()):.5f} ptr {int(model.queue_ptr)} |queue| {float(model.queue.norm()):.3f} "
          f"bn1.rm {float(bn.running_mean.norm()):.4f} lr {float(opt.param_groups[0]['lr']):.4f} fallback {runner.fallback_reason}", flush=True)
    badg = [(n, tuple(p.shape), int((~torch.isfinite(p.grad)).sum())) for n, p in model.named_parameters()
            if p.grad is not None and not bool(torch.isfinite(p.grad).all())]
    if badg and i < 8:
        print(f"   non-finite gradients after step {i}: {len(badg)}", badg[:12], flush=True)
