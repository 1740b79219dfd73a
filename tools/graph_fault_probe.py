"""Probe for the whole-step hipGraph replay fault of DESIGN.md section 5 (1x1 weight gradients coming back as garbage
when MIOpen produces them inside the captured graph).  Runs the R50 bench step captured as one graph with the 1x1 weight
gradients routed to MIOpen (the configuration that failed in round 1), reports which gradients are non-finite / far from
an eager reference after the first replays, for the solver set selected by the MIOPEN_DEBUG_* environment.
    python tools/graph_fault_probe.py [miopen|hip]      (weight gradient of the 1x1 layers by MIOpen or by cp2_wgrad1x1)
"""
import os
import sys
os.environ.setdefault("MIOPEN_FIND_MODE", "FAST")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from cp2_amd import builder, synthetic
from cp2_amd import encoder
from cp2_amd.config import Config
from cp2_amd.engine import TrainStep
from cp2_amd.main import make_optimizer
from cp2_amd.pretrain_types import PretrainType

mode = sys.argv[1] if len(sys.argv) > 1 else "miopen"
encoder._Conv1x1Fn.hip_wgrad = mode == "hip"
encoder.Conv2d.cpp_nodes = os.environ.get("CPP_NODES", "1" if mode == "hip" else "0") == "1"
torch.manual_seed(0)
torch.backends.cudnn.benchmark = True
dev = torch.device("cuda", 0)
cfg = Config.fromfile("configs/config_pretrain_r50_fcn.py")
FROZEN = os.environ.get("FROZEN", "0") == "1"
# FROZEN: m = 1 (EMA is the identity), no enqueue, fixed shuffle: every step computes the SAME gradients, so any replay
# can be compared tightly with the eager warm-up steps.  Otherwise the real step (EMA, enqueue), with the optimizer step
# skipped so that a garbage gradient is reported instead of poisoning the parameters: gradients drift by ~10 % per
# step, garbage = non-finite or more than 10x the eager norm.
model = builder.MODEL(cfg, rank=0, K=65536, m=1.0 if FROZEN else 0.999, pretrain_from_scratch=True, pretrain_type=PretrainType.CP2,
                      device=dev, amp_dtype=torch.bfloat16, channels_last=True).to(dev)
if FROZEN:
    model._dequeue_and_enqueue = lambda keys: None
model.encoder_q.to(memory_format=torch.channels_last); model.encoder_k.to(memory_format=torch.channels_last)
model.train()
model.log_quartiles = False


class A:
    lr, momentum, weight_decay, optim = 0.0, 0.9, 0.0, "sgd"       # lr 0: parameters stay put, every step has the same gradients


OPT = os.environ.get("OPT", "none")                                # none: gradients only | fused: torch fused SGD | flat: FlatSGD
LR0 = os.environ.get("LR0", "0") == "1"                            # optimizer kernel in the graph, but lr = wd = 0: parameters stay
if OPT != "none" and not LR0:                                      # put, so every replay must reproduce the eager gradients
    A.lr, A.weight_decay = 0.03, 1e-4
opt = make_optimizer(list(model.parameters()), A, dev, capturable=True, model=model if OPT == "flat" else None)
if OPT == "none":
    opt.step = lambda *a, **k: None                                # gradients only
# dirty the caching allocator's free blocks, as a long run would
junk = [torch.full((1 << 26,), float("nan"), device=dev) for _ in range(24)]
torch.cuda.synchronize(); del junk
runner = TrainStep(model, opt, use_graph=True, warmup_steps=3, verify=os.environ.get("VERIFY", "0") == "1", reverify_every=0)
batch = synthetic.make_batch(32, 224, 224, dev, seed=0)
names = [n for n, p in model.named_parameters() if p.requires_grad]
params = [p for n, p in model.named_parameters() if p.requires_grad]
ref = None
for i in range(int(os.environ.get("STEPS", "14"))):
    torch.manual_seed(123)                                          # same shuffle permutation every step
    loss = runner(batch)
    torch.cuda.synchronize()
    grads = [None if p.grad is None else p.grad.detach().float().clone() for p in params]
    kind = "graph" if runner.graph is not None else "eager"
    if kind == "eager":
        ref = grads
    bad = []
    for n, g, r in zip(names, grads, ref):
        if g is None or r is None:
            continue
        err = (g - r).norm().item()
        if not (err == err) or err > (0.05 if FROZEN else 10.0) * r.norm().item() + 1e-6:
            bad.append((n, tuple(g.shape), f"{err:.2e}/{r.norm().item():.2e}"))
    nonfinite = [n for n, p in model.named_parameters() if not torch.isfinite(p).all()]
    if OPT != "none" and not LR0:
        bad = [b for b in bad if b[2].startswith("nan") or b[2].startswith("inf")]
    worst = max(((g - r).norm().item() / (r.norm().item() + 1e-12), n) for n, g, r in zip(names, grads, ref) if g is not None and r is not None)
    print(f"step {i} {kind} loss {float(loss):.4f} bad-gradients {len(bad)}", bad[:6], "non-finite params", len(nonfinite), nonfinite[:4],
          f"worst rel err {worst[0]:.3e} at {worst[1]}", flush=True)
