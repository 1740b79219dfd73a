"""GPU: the encoder fast path -- fused training-mode BatchNorm (+ residual) (+ ReLU) on channels-last bf16
activations (cp2_amd/csrc/bn.hip) against torch.nn.BatchNorm2d evaluated in fp32 on the same bf16 inputs.
Tolerance: outputs / input gradients are bf16 (8 significant bits): |err| <= 1.6e-2 * max|ref|;
statistics, running buffers and parameter gradients are fp32: 2e-3 relative (inputs carry bf16 rounding)."""
import pytest
import torch
import torch.nn.functional as F

from cp2_amd.encoder import FusedBatchNorm2d

pytestmark = pytest.mark.gpu
DEV = "cuda"


def close(got, want, rel, what):
    got, want = got.float().cpu(), want.float().cpu()
    err = (got - want).abs().max().item()
    lim = rel * want.abs().max().item() + 1e-6
    assert err <= lim, f"{what}: {err:.3e} > {lim:.3e}"


@pytest.mark.parametrize("N,C,H,W", [(32, 64, 56, 56), (5, 256, 7, 9), (32, 2048, 14, 14), (3, 128, 1, 1), (2, 512, 13, 5),
                                     (4, 192, 9, 9), (32, 1024, 14, 14), (8, 256, 56, 56)])
@pytest.mark.parametrize("relu,use_res", [(True, False), (True, True), (False, False), (False, True)])
def test_fused_bn_matches_fp32_batchnorm(N, C, H, W, relu, use_res):
    torch.manual_seed(C + H)
    x = (torch.randn(N, C, H, W, device=DEV) * 1.7 + 0.3).to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
    res = torch.randn(N, C, H, W, device=DEV).to(torch.bfloat16).contiguous(memory_format=torch.channels_last) if use_res else None
    up = torch.randn(N, C, H, W, device=DEV).to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
    fused = FusedBatchNorm2d(C).to(DEV).train()
    ref = torch.nn.BatchNorm2d(C).to(DEV).train()
    with torch.no_grad():
        fused.weight.uniform_(0.5, 1.5); fused.bias.uniform_(-0.5, 0.5)
        ref.weight.copy_(fused.weight); ref.bias.copy_(fused.bias)
    xf = x.clone().requires_grad_(True)
    rf = res.clone().requires_grad_(True) if use_res else None
    y = fused(xf, residual=rf, relu=relu)
    assert y.dtype == torch.bfloat16 and y.is_contiguous(memory_format=torch.channels_last)
    y.backward(up)
    xr = x.float().requires_grad_(True)
    rr = res.float().requires_grad_(True) if use_res else None
    yr = ref(xr)
    if use_res:
        yr = yr + rr
    if relu:
        yr = F.relu(yr)
    yr.backward(up.float())
    close(y, yr, 1.6e-2, "y")
    close(fused.running_mean, ref.running_mean, 2e-3, "running_mean")
    close(fused.running_var, ref.running_var, 2e-3, "running_var")
    # the ReLU mask comes from the bf16 output, so a few units whose pre-activation is ~0 flip on/off relative to the
    # fp32 reference and their gradient differs by a whole dy: allow 0.2 % such elements, the rest to bf16 precision
    err = (xf.grad.float() - xr.grad).abs()
    lim = 2.5e-2 * xr.grad.abs().max().item()
    assert (err > lim).float().mean().item() < 2e-3, f"dx: {(err > lim).float().mean().item():.4f} of elements off"
    close(fused.weight.grad, ref.weight.grad, 1e-2, "dgamma")
    close(fused.bias.grad, ref.bias.grad, 1e-2, "dbeta")
    if use_res:
        e2 = (rf.grad.float() - rr.grad).abs()
        assert (e2 > 1.6e-2 * rr.grad.abs().max().item()).float().mean().item() < 2e-3, "dres"
    sd = fused.state_dict()
    assert set(sd) == set(ref.state_dict()) and int(sd["num_batches_tracked"]) == 1


def test_fused_bn_is_run_to_run_identical():
    """Partial sums are combined in a fixed order (no atomics): repeated launches are bit-identical."""
    torch.manual_seed(1)
    outs = []
    for C, H in ((64, 56), (2048, 14), (256, 28)):
        x = torch.randn(16, C, H, H, device=DEV).to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
        up = torch.randn_like(x)
        for rep in range(3):
            bn = FusedBatchNorm2d(C).to(DEV).train()
            xf = x.clone().requires_grad_(True)
            y = bn(xf, relu=True)
            y.backward(up)
            cur = (y.detach().clone(), xf.grad.clone(), bn.weight.grad.clone(), bn.bias.grad.clone(), bn.running_var.clone())
            if rep:
                for a, b in zip(outs, cur):
                    assert torch.equal(a, b)
            outs = cur


def test_fallback_paths_match_stock_batchnorm():
    torch.manual_seed(0)
    fused = FusedBatchNorm2d(64).to(DEV)
    ref = torch.nn.BatchNorm2d(64).to(DEV)
    x = torch.randn(4, 64, 8, 8, device=DEV)                      # fp32, NCHW -> stock path
    assert torch.equal(fused(x), ref(x))
    res = torch.randn_like(x)
    assert torch.equal(fused(x, residual=res, relu=True), F.relu(ref(x) + res))
    fused.eval(); ref.eval()
    xb = x.to(torch.bfloat16).contiguous(memory_format=torch.channels_last)   # eval mode -> stock path
    assert torch.equal(fused(xb), ref(xb))


def test_cpp_and_python_autograd_nodes_launch_the_same_kernels():
    """FusedBatchNorm2d.cpp_node selects the C++ autograd node (csrc_torch/autograd_ext.cpp) or its Python twin: same
    cp2_bn_fwd / cp2_bn_bwd launches, so outputs, gradients and running statistics are bit-identical."""
    from cp2_amd import _cext
    if _cext.load() is None:
        pytest.skip("C++ autograd nodes not built")
    torch.manual_seed(5)
    for (N, C, H, W), relu, use_res in (((8, 256, 14, 14), True, True), ((4, 64, 28, 28), True, False), ((3, 128, 7, 5), False, True)):
        x = torch.randn(N, C, H, W, device=DEV).to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
        res = torch.randn_like(x) if use_res else None
        up = torch.randn_like(x)
        got = []
        for cpp in (True, False):
            FusedBatchNorm2d.cpp_node = cpp
            try:
                bn = FusedBatchNorm2d(C).to(DEV).train()
                with torch.no_grad():
                    bn.weight.copy_(torch.linspace(0.5, 1.5, C)); bn.bias.copy_(torch.linspace(-0.3, 0.3, C))
                xf = x.clone().requires_grad_(True)
                rf = res.clone().requires_grad_(True) if use_res else None
                y = bn(xf, residual=rf, relu=relu)
                y.backward(up)
                got.append((y.detach(), xf.grad, bn.weight.grad, bn.bias.grad, bn.running_mean.clone(), bn.running_var.clone())
                           + ((rf.grad,) if use_res else ()))
            finally:
                FusedBatchNorm2d.cpp_node = True
        for a, b in zip(*got):
            assert torch.equal(a, b)
