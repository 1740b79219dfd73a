"""GPU: encoder plumbing around the hot path -- the 1x1 convolution routed through hipBLASLt (_Conv1x1Fn) against
F.conv2d on the same bf16 operands (tolerance: bf16 outputs, 2e-2 of the largest reference value; fp32 parameter
gradients 2e-2 as both sides accumulate bf16 products in different orders)."""
import pytest
import torch
import torch.nn.functional as F

from cp2_amd.encoder import Conv2d, _Conv1x1Fn

pytestmark = pytest.mark.gpu
DEV = "cuda"


def close(got, want, rel, what):
    err = (got.float() - want.float()).abs().max().item()
    lim = rel * want.float().abs().max().item() + 1e-6
    assert err <= lim, f"{what}: {err:.3e} > {lim:.3e}"


@pytest.mark.parametrize("N,HW,ci,co", [(8, 14, 512, 2048), (8, 14, 1024, 256), (4, 28, 512, 128), (2, 56, 64, 256), (3, 7, 256, 128)])
@pytest.mark.parametrize("bias", [False, True])
@pytest.mark.parametrize("mm_fwd,mm_dgrad", [(True, True), (False, True), (True, False), (False, False)])
def test_conv1x1_function_matches_conv2d(N, HW, ci, co, bias, mm_fwd, mm_dgrad):
    torch.manual_seed(ci + co)
    x = torch.randn(N, ci, HW, HW, device=DEV).to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
    w = (torch.randn(co, ci, 1, 1, device=DEV) * ci ** -0.5).contiguous(memory_format=torch.channels_last)
    b = torch.randn(co, device=DEV) if bias else None
    up = torch.randn(N, co, HW, HW, device=DEV).to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
    xa, wa = x.clone().requires_grad_(True), w.clone().requires_grad_(True)
    ba = b.clone().requires_grad_(True) if bias else None
    y = _Conv1x1Fn.apply(xa, wa, wa.detach().to(torch.bfloat16), ba, mm_fwd, mm_dgrad)
    assert y.dtype == torch.bfloat16 and y.is_contiguous(memory_format=torch.channels_last)
    y.backward(up)
    xr, wr = x.clone().requires_grad_(True), w.clone().requires_grad_(True)
    br = b.clone().requires_grad_(True) if bias else None
    with torch.autocast("cuda", dtype=torch.bfloat16):
        yr = F.conv2d(xr, wr, br)
    yr.backward(up)
    close(y, yr, 2e-2, "y")
    close(xa.grad, xr.grad, 2e-2, "dx")
    assert wa.grad.dtype == torch.float32 and wa.grad.shape == w.shape
    close(wa.grad, wr.grad, 2e-2, "dw")
    if bias:
        close(ba.grad, br.grad, 2e-2, "dbias")


def test_conv2d_module_routes_1x1_and_keeps_other_layers():
    torch.manual_seed(1)
    conv = Conv2d(512, 1024, 1, bias=False).to(DEV).to(memory_format=torch.channels_last)
    conv.shadow_weight = conv.weight.detach().to(torch.bfloat16)
    x = torch.randn(8, 512, 14, 14, device=DEV).to(torch.bfloat16).contiguous(memory_format=torch.channels_last).requires_grad_(True)
    y = conv(x)                                    # eager + autograd: the MIOpen path (the Function costs host time)
    assert type(y.grad_fn).__name__ != "_Conv1x1FnBackward"
    y.float().square().mean().backward()
    assert conv.weight.grad is not None and conv.weight.grad.dtype == torch.float32 and x.grad is not None
    with torch.no_grad():                          # gradient-free forward: GEMM, same values to bf16 precision
        yg = conv(x)
        Conv2d.gemm_1x1 = False
        try:
            yc = conv(x)
        finally:
            Conv2d.gemm_1x1 = True
    assert yg.is_contiguous(memory_format=torch.channels_last)
    close(yg, yc, 2e-2, "no-grad forward")
    c3 = Conv2d(64, 64, 3, padding=1, bias=False).to(DEV).to(memory_format=torch.channels_last)
    c3.shadow_weight = c3.weight.detach().to(torch.bfloat16)
    x3 = torch.randn(2, 64, 8, 8, device=DEV).to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
    assert type(c3(x3).grad_fn).__name__ != "_Conv1x1FnBackward"


@pytest.mark.parametrize("N,H,W,co,ci", [(2, 7, 7, 64, 64), (8, 14, 14, 512, 2048), (4, 28, 28, 128, 512), (2, 56, 56, 64, 256),
                                         (3, 5, 3, 192, 320), (32, 14, 14, 1024, 256), (1, 1, 1, 128, 128), (5, 9, 11, 256, 128)])
def test_wgrad1x1_matches_fp32_matmul(N, H, W, co, ci):
    """cp2_wgrad1x1 (transposing LDS reads + bf16 MFMA, fp32 accumulation, deterministic split-K) against the fp32
    matmul of the same bf16 operands: products are exact in fp32, only the summation order differs -> 2e-5 of max."""
    from cp2_amd import ops
    torch.manual_seed(co + ci + H)
    dy = torch.randn(N, co, H, W, device=DEV).to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
    x = torch.randn(N, ci, H, W, device=DEV).to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
    dw = ops.wgrad1x1(dy, x)
    assert dw.shape == (co, ci, 1, 1) and dw.dtype == torch.float32
    ref = dy.permute(0, 2, 3, 1).reshape(-1, co).double().t() @ x.permute(0, 2, 3, 1).reshape(-1, ci).double()
    err = (dw.view(co, ci).double() - ref).abs().max().item()
    assert err <= 2e-5 * ref.abs().max().item() + 1e-6, err
    assert torch.equal(dw, ops.wgrad1x1(dy, x))                 # run-to-run identical


@pytest.mark.parametrize("N,HW,ci,co,bias", [(8, 14, 512, 2048, False), (4, 28, 512, 128, True), (2, 56, 64, 256, False), (3, 7, 192, 320, True)])
@pytest.mark.parametrize("mm_fwd,mm_dgrad", [(True, True), (False, False)])
def test_cpp_conv1x1_node_equals_python_node(N, HW, ci, co, bias, mm_fwd, mm_dgrad):
    """The C++ autograd node launches the same kernels as the Python node: identical outputs and gradients."""
    from cp2_amd import _cext
    ext = _cext.load()
    assert ext is not None
    torch.manual_seed(ci * 3 + co)
    x = torch.randn(N, ci, HW, HW, device=DEV).to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
    w = (torch.randn(co, ci, 1, 1, device=DEV) * ci ** -0.5).contiguous(memory_format=torch.channels_last)
    b = torch.randn(co, device=DEV) if bias else None
    up = torch.randn(N, co, HW, HW, device=DEV).to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
    res = []
    for use_cpp in (True, False):
        xa, wa = x.clone().requires_grad_(True), w.clone().requires_grad_(True)
        ba = b.clone().requires_grad_(True) if bias else None
        sh = wa.detach().to(torch.bfloat16)
        y = ext.conv1x1(xa, wa, sh, ba, mm_fwd, mm_dgrad, True) if use_cpp else _Conv1x1Fn.apply(xa, wa, sh, ba, mm_fwd, mm_dgrad)
        y.backward(up)
        res.append((y.detach(), xa.grad, wa.grad, ba.grad if bias else None))
    for a, c, name in zip(res[0], res[1], ("y", "dx", "dw", "db")):
        if a is None:
            continue
        assert a.dtype == c.dtype and a.shape == c.shape, name
        if name == "dw":
            assert a.stride() == w.stride()                     # C++ node: the master weight's own strides
            assert torch.equal(a, c), name                      # both from cp2_wgrad1x1 (deterministic)
        else:
            close(a, c, 2e-2, name)                             # MIOpen / hipBLASLt may differ run to run in bf16


def test_cpp_shadow_weight_node():
    from cp2_amd import _cext
    ext = _cext.load()
    w = torch.randn(64, 32, 3, 3, device=DEV).contiguous(memory_format=torch.channels_last).requires_grad_(True)
    sh = w.detach().to(torch.bfloat16)
    out = ext.shadow_weight(w, sh)
    assert out.data_ptr() == sh.data_ptr() and out.dtype == torch.bfloat16 and out.requires_grad
    g = torch.randn(64, 32, 3, 3, device=DEV).to(torch.bfloat16)            # NCHW-dense gradient for a channels-last weight
    out.backward(g)
    assert w.grad.dtype == torch.float32 and w.grad.stride() == w.stride() and torch.equal(w.grad, g.float())


@pytest.mark.parametrize("N,C,H,W", [(4, 64, 112, 112), (2, 16, 37, 53), (1, 8, 5, 4), (3, 64, 56, 57)])
def test_stem_maxpool_equals_torch(N, C, H, W):
    """csrc/pool.hip against torch.nn.MaxPool2d(3, 2, 1) on the same channels-last bf16 input: outputs bit-equal
    (ties between equal bf16 values resolved as torch does: first in window order), input gradients bit-equal."""
    from cp2_amd.encoder import StemMaxPool
    torch.manual_seed(N * C + H)
    x = torch.randn(N, C, H, W, device=DEV).relu().to(torch.bfloat16).contiguous(memory_format=torch.channels_last)   # many ties at 0
    x[0, :, 0, 0] = float("nan") if H > 5 else 1.0
    up = torch.randn(N, C, (H - 1) // 2 + 1, (W - 1) // 2 + 1, device=DEV).to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
    pool = StemMaxPool(3, stride=2, padding=1)
    xa = x.clone().requires_grad_(True)
    y = pool(xa)
    y.backward(up)
    xr = x.clone().requires_grad_(True)
    yr = torch.nn.functional.max_pool2d(xr, 3, 2, 1)
    yr.backward(up)
    assert y.shape == yr.shape and y.is_contiguous(memory_format=torch.channels_last)
    assert torch.equal(torch.nan_to_num(y.float(), nan=-7.0), torch.nan_to_num(yr.float(), nan=-7.0))
    assert torch.equal(xa.grad, xr.grad)
    with torch.no_grad():
        assert torch.equal(torch.nan_to_num(pool(x).float(), nan=-7.0), torch.nan_to_num(yr.float(), nan=-7.0))
    # anything else falls through to PyTorch
    xf = torch.randn(2, 8, 9, 9, device=DEV)
    assert torch.equal(pool(xf), torch.nn.functional.max_pool2d(xf, 3, 2, 1))


@pytest.mark.parametrize("N,ci,co,HW,k,stride,pad,dil,bias", [
    (4, 64, 64, 56, 3, 1, 1, 1, False),       # layer1 conv2
    (3, 128, 128, 29, 3, 2, 1, 1, False),     # a strided block (odd size: borders on every side)
    (2, 512, 512, 14, 3, 1, 2, 2, False),     # layer4: dilation 2
    (2, 2048, 512, 14, 3, 1, 1, 1, False),    # FCN head
    (2, 192, 64, 9, 3, 1, 1, 1, True),        # 64-wide tiles, bias
])
@pytest.mark.parametrize("cpp", [True, False])
def test_conv_kxk_weight_gradient_matches_conv2d(N, ci, co, HW, k, stride, pad, dil, bias, cpp):
    """k x k convolution node (forward / data gradient MIOpen, weight gradient cp2_wgrad_conv in fp32) against autograd's
    F.conv2d on the same bf16 operands: fp32 weight gradient within 2e-2 of its largest value (both sides sum bf16
    products in different orders), same layout as the channels-last master weight."""
    from cp2_amd import _cext
    from cp2_amd.encoder import _ConvKxKFn
    torch.manual_seed(ci + co + HW)
    x = torch.randn(N, ci, HW, HW, device=DEV).to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
    w = (torch.randn(co, ci, k, k, device=DEV) * (ci * k * k) ** -0.5).contiguous(memory_format=torch.channels_last)
    b = torch.randn(co, device=DEV) if bias else None
    xa, wa = x.clone().requires_grad_(True), w.clone().requires_grad_(True)
    ba = b.clone().requires_grad_(True) if bias else None
    shadow = wa.detach().to(torch.bfloat16)
    ext = _cext.load() if cpp else None
    if cpp and ext is None:
        pytest.skip("C++ autograd nodes not built")
    y = ext.conv_kxk(xa, wa, shadow, ba, stride, pad, dil) if cpp else _ConvKxKFn.apply(xa, wa, shadow, ba, stride, pad, dil)
    up = torch.randn_like(y)
    y.backward(up)
    xr, wr = x.clone().requires_grad_(True), w.clone().requires_grad_(True)
    br = b.clone().requires_grad_(True) if bias else None
    with torch.autocast("cuda", dtype=torch.bfloat16):
        yr = F.conv2d(xr, wr, br, stride, pad, dil)
    yr.backward(up)
    close(y, yr, 2e-2, "y")
    close(xa.grad, xr.grad, 2e-2, "dx")
    assert wa.grad.dtype == torch.float32 and wa.grad.stride() == wa.stride()
    close(wa.grad, wr.grad, 2e-2, "dw")
    # exact reference of the weight gradient from the same bf16 operands in fp64
    ref = torch.nn.grad.conv2d_weight(x.double(), w.shape, up.double(), stride=stride, padding=pad, dilation=dil)
    close(wa.grad.double(), ref, 2e-3, "dw vs fp64")
    if bias:
        close(ba.grad, br.grad, 2e-2, "db")
