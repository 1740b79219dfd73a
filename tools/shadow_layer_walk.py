"""Where do the key encoder's outputs with and without the bf16 weight image start to differ?  (round-3 verdict, weak #1)

Walks the eval-mode key encoder module by module (forward hooks on every leaf module), three ways:
  A  with the weight image (`Conv2d.shadow_weight`, what the training step runs),
  A' the same again (is a forward run-to-run reproducible at all?),
  B  without it (autocast casts the fp32 weights per call),
and the same with the GEMM routing of the wide 1x1 layers switched off (`Conv2d.gemm_1x1 = False`), then prints the
first leaf whose outputs differ, how far apart they are in bf16 ulps, and which GPU kernels served that layer either way
(torch.profiler).    python tools/shadow_layer_walk.py [config] [batch] [size]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("MIOPEN_FIND_MODE", "FAST")
import torch  # noqa: E402

from cp2_amd import builder  # noqa: E402
from cp2_amd.config import Config  # noqa: E402
from cp2_amd.encoder import Conv2d  # noqa: E402
from cp2_amd.pretrain_types import PretrainType  # noqa: E402

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
cfg_name = sys.argv[1] if len(sys.argv) > 1 else "config_pretrain_r18.py"
B = int(sys.argv[2]) if len(sys.argv) > 2 else 4
HW = int(sys.argv[3]) if len(sys.argv) > 3 else 64
DEV = "cuda"


def ulps_bf16(a: torch.Tensor, b: torch.Tensor) -> int:
    """Largest distance in units of the last place of bf16 between two bf16 tensors (sign-magnitude -> ordered integers)."""
    def order(t):
        i = t.contiguous().view(torch.int16).to(torch.int32) & 0xFFFF
        return torch.where(i >= 0x8000, 0x8000 - i, i)
    return int((order(a) - order(b)).abs().max())


def walk(model, x, use_shadow: bool):
    outs = []
    hooks = []
    for name, m in model.encoder_k.named_modules():
        if len(list(m.children())) == 0:
            hooks.append(m.register_forward_hook(
                lambda mod, inp, out, name=name: outs.append((name, type(mod).__name__, out.detach().clone(), inp[0].detach().clone()))
                if isinstance(out, torch.Tensor) else None))
    saved = {}
    if not use_shadow:
        for m in model.encoder_k.modules():
            if isinstance(m, Conv2d):
                saved[m] = m.shadow_weight
                m.shadow_weight = None
    with torch.no_grad(), torch.autocast("cuda", dtype=torch.bfloat16):
        y = model.encoder_k(x)
    for m, w in saved.items():
        m.shadow_weight = w
    for h in hooks:
        h.remove()
    return outs, y


def first_difference(a, b):
    for (na, ta, xa, ia), (nb, tb, xb, ib) in zip(a, b):
        assert na == nb
        if xa.dtype != xb.dtype or not torch.equal(xa, xb):
            return na, ta, xa, xb, ia, torch.equal(ia, ib)
    return None


def kernels_of(fn):
    from torch.profiler import ProfilerActivity, profile
    fn()
    torch.cuda.synchronize()
    with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA]) as prof:
        fn()
        torch.cuda.synchronize()
    return sorted({e.name for e in prof.events() if e.device_type == torch.autograd.DeviceType.CUDA})


def main():
    torch.manual_seed(0)
    torch.backends.cudnn.benchmark = os.environ.get("WALK_BENCHMARK", "0") == "1"
    cfg = Config.fromfile(os.path.join(ROOT, "configs", cfg_name))
    model = builder.MODEL(cfg, rank=0, K=1024, pretrain_from_scratch=True, pretrain_type=PretrainType.CP2, device=DEV,
                          amp_dtype=torch.bfloat16, channels_last=True).to(DEV).train()
    model.encoder_q.to(memory_format=torch.channels_last)
    model.encoder_k.to(memory_format=torch.channels_last)
    for p in model.encoder_q.parameters():
        p.data.add_(0.003 * torch.randn_like(p))
    model._momentum_update_key_encoder()
    model.encoder_k.eval()
    x = torch.rand(B, 3, HW, HW, device=DEV).contiguous(memory_format=torch.channels_last)
    print(f"config {cfg_name}, input {B}x3x{HW}x{HW}, cudnn.benchmark={torch.backends.cudnn.benchmark}")
    for gemm in (True, False):
        Conv2d.gemm_1x1 = gemm
        a, ya = walk(model, x, True)
        a2, ya2 = walk(model, x, True)
        b, yb = walk(model, x, False)
        print(f"--- Conv2d.gemm_1x1 = {gemm}: {len(a)} leaf outputs")
        d = first_difference(a, a2)
        print("  image vs image again:", "identical at every leaf" if d is None else f"first difference at {d[0]} ({d[1]})")
        d = first_difference(a, b)
        if d is None:
            print("  image vs per-call cast: identical at every leaf; final outputs equal:", torch.equal(ya, yb))
            continue
        name, typ, xa, xb, xin, same_in = d
        print(f"  image vs per-call cast: first difference at {name} ({typ}), output {tuple(xa.shape)} {xa.dtype}; inputs identical: {same_in}")
        if xa.dtype == torch.bfloat16:
            diff = (xa.float() - xb.float()).abs()
            print(f"    max bf16 ulp gap {ulps_bf16(xa, xb)}, max |diff| {float(diff.max()):.3e} on values up to {float(xb.float().abs().max()):.3e}, "
                  f"{float((diff > 0).float().mean()) * 100:.2f} % of the elements differ")
        print(f"    final outputs: max |diff| {float((ya.float() - yb.float()).abs().max()):.3e} on values up to {float(yb.float().abs().max()):.3e}, "
              f"max bf16 ulp gap {ulps_bf16(ya.bfloat16(), yb.bfloat16())}")
        mod = dict(model.encoder_k.named_modules())[name]
        if isinstance(mod, Conv2d):
            w = mod.shadow_weight
            print(f"    layer: {mod.in_channels}->{mod.out_channels} k{mod.kernel_size} s{mod.stride} p{mod.padding}; weight image strides "
                  f"{tuple(w.stride())}, contiguous(channels_last)={w.is_contiguous(memory_format=torch.channels_last)}, "
                  f"image == cast(weight): {torch.equal(w, mod.weight.to(torch.bfloat16))}")

            def run(shadow):
                keep = mod.shadow_weight
                if not shadow:
                    mod.shadow_weight = None
                with torch.no_grad(), torch.autocast("cuda", dtype=torch.bfloat16):
                    mod(xin)
                mod.shadow_weight = keep
            print("    kernels with the image:   ", kernels_of(lambda: run(True)))
            print("    kernels with per-call cast:", kernels_of(lambda: run(False)))
    Conv2d.gemm_1x1 = True


if __name__ == "__main__":
    main()
