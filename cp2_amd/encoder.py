"""Encoder surface below the hot path: ResNet + ASPP / FCN head with `contrast=True`.

The reference obtains this from the external `mmseg` package
(`build_segmentor(cfg.model)`, builder.py:366-371); the only in-tree picture of it is
the vendored mmseg_/models/{backbones/resnet.py, decode_heads/*.py,
segmentors/encoder_decoder.py}.  This file restates that surface in plain torch.nn so
the step runs without mmseg/mmcv, keeping
  * the call contract:  enc(img) -> [b,128,H/os,W/os];  enc.backbone(img) -> 4 stage maps;
    enc.backbone.init_weights()            (encoder_decoder.py:137-140, resnet.py:632-647)
  * the parameter names of a mmseg checkpoint: backbone.conv1 / bn1 / layer{1..4}.{i}.
    conv{1,2,3} / bn{1,2,3} / downsample.{0,1};  decode_head.image_pool.1.{conv,bn},
    aspp_modules.{i}.{conv,bn}, bottleneck.{conv,bn}, convs.{i}.{conv,bn}, conv_cat,
    contrast_conv.{0,2}, conv_seg          (resnet.py:161-206,567-579; aspp_head.py:69-97;
                                            fcn_head.py:41-79; decode_head.py:84)
The encoder stays in PyTorch-ROCm (MIOpen / hipBLASLt) by design (north_star): it is
plumbing around the hand-written kernels, not part of them.
"""
from __future__ import annotations

import os
import warnings
from typing import Sequence

import torch
import torch.nn as nn
import torch.nn.functional as F


from . import _cext  # noqa: E402
from . import ops as _bn_ops  # noqa: E402  (ctypes front end; only touched on the GPU fast path)


class _FusedBNFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, weight, bias, residual, running_mean, running_var, momentum, eps, relu):
        ops = _bn_ops
        y, stats = ops.bn_fwd(x, residual, weight, bias, running_mean, running_var, momentum, eps, relu)
        ctx.save_for_backward(x, y if relu else None, weight, stats)
        ctx.relu, ctx.has_res = relu, residual is not None
        return y

    @staticmethod
    def backward(ctx, dy):
        ops = _bn_ops
        x, y, weight, stats = ctx.saved_tensors
        if dy.dtype != torch.bfloat16 or not dy.is_contiguous(memory_format=torch.channels_last):
            dy = dy.to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
        dx, dres, dgamma, dbeta = ops.bn_bwd(x, dy, y, weight, stats, ctx.relu, ctx.has_res)
        return dx, dgamma, dbeta, dres, None, None, None, None, None


class FusedBatchNorm2d(nn.BatchNorm2d):
    """nn.BatchNorm2d (same parameters, buffers and state-dict keys) whose training forward on channels-last bf16
    GPU activations is the fused BN (+ residual) (+ ReLU) kernel set of cp2_amd/csrc/bn.hip; anything else
    (fp32, eval mode, NCHW, CPU) takes the stock PyTorch path.  `num_batches_tracked` is advanced lazily (it only
    matters when momentum is None, which is never fused)."""

    fused = True            # class-wide switch (bench.py --fused-bn off for A/B)
    cpp_node = True         # autograd node from csrc_torch/autograd_ext.cpp when built (else the Python node below)

    def __init__(self, *a, **kw):
        super().__init__(*a, **kw)
        self._pending_batches = 0

    def forward(self, x, residual=None, relu=False):
        if (FusedBatchNorm2d.fused and self.training and x.dtype == torch.bfloat16 and x.is_cuda
                and self.track_running_stats and self.momentum is not None and self.affine
                and self.weight.dtype == torch.float32 and _bn_ops.bn_supported(x)
                and (residual is None or (residual.shape == x.shape and _bn_ops.bn_supported(residual)))):
            self._pending_batches += 1
            ext = _cext.load() if FusedBatchNorm2d.cpp_node else None
            if ext is not None:      # the same node in C++: ~20 us less interpreter time per call and direction
                return ext.fused_bn(x, self.weight, self.bias, residual, self.running_mean, self.running_var,
                                    self.momentum, self.eps, relu)
            return _FusedBNFn.apply(x, self.weight, self.bias, residual, self.running_mean, self.running_var,
                                    self.momentum, self.eps, relu)
        y = super().forward(x)
        if residual is not None:
            y = y + residual
        return F.relu(y) if relu else y

    def _save_to_state_dict(self, destination, prefix, keep_vars):
        if self._pending_batches and self.num_batches_tracked is not None:
            self.num_batches_tracked += self._pending_batches
            self._pending_batches = 0
        super()._save_to_state_dict(destination, prefix, keep_vars)


class SyncFusedBatchNorm2d(nn.SyncBatchNorm):
    """torch.nn.SyncBatchNorm (batch statistics over all ranks: what Lightning's Trainer(sync_batchnorm=True) turns every
    BatchNorm into, reference mirror_pretrain.py:229-231) with the (+ residual) (+ ReLU) call signature the ResNet blocks use."""

    def forward(self, x, residual=None, relu=False):
        y = super().forward(x)
        if residual is not None:
            y = y + residual
        return F.relu(y) if relu else y


def convert_sync_batchnorm(module: nn.Module, process_group=None) -> nn.Module:
    """Replace every FusedBatchNorm2d / BatchNorm2d below `module` by a SyncFusedBatchNorm2d holding the same parameters and
    buffers (state-dict keys unchanged) -- torch.nn.SyncBatchNorm.convert_sync_batchnorm for modules whose forward takes
    (x, residual, relu).  The fused single-rank BN kernels (csrc/bn.hip) see one rank's rows only; statistics over the
    ranks go through torch's SyncBatchNorm (all-gather of per-channel mean / inverse std / count)."""
    out = module
    if isinstance(module, nn.BatchNorm2d) and not isinstance(module, nn.SyncBatchNorm):
        out = SyncFusedBatchNorm2d(module.num_features, module.eps, module.momentum, module.affine, module.track_running_stats,
                                   process_group)
        if module.affine:
            with torch.no_grad():
                out.weight, out.bias = module.weight, module.bias
        out.running_mean, out.running_var = module.running_mean, module.running_var
        if isinstance(module, FusedBatchNorm2d) and module._pending_batches and module.num_batches_tracked is not None:
            module.num_batches_tracked += module._pending_batches
        out.num_batches_tracked = module.num_batches_tracked
        out.training = module.training
    for name, child in module.named_children():
        out.add_module(name, convert_sync_batchnorm(child, process_group))
    return out


class _MaxPool3s2Fn(torch.autograd.Function):
    """MaxPool2d(3, 2, 1) of the stem on channels-last bf16 activations by csrc/pool.hip (one byte of argmax per output
    element, gather backward): 13 + 20 us instead of ATen's 35 + 85 us at 32 x 64 x 112 x 112."""

    @staticmethod
    def forward(ctx, x):
        y, idx = _bn_ops.maxpool3s2_fwd(x)
        ctx.save_for_backward(idx)
        ctx.xshape = tuple(x.shape)
        return y

    @staticmethod
    def backward(ctx, dy):
        (idx,) = ctx.saved_tensors
        if dy.dtype != torch.bfloat16 or not dy.is_contiguous(memory_format=torch.channels_last):
            dy = dy.to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
        return _bn_ops.maxpool3s2_bwd(dy, idx, ctx.xshape)


class StemMaxPool(nn.MaxPool2d):
    """nn.MaxPool2d(3, stride=2, padding=1) (no parameters, same module name in the tree) with the HIP fast path."""

    fused = True

    def forward(self, x):
        if (StemMaxPool.fused and self.kernel_size == 3 and self.stride == 2 and self.padding == 1 and self.dilation == 1
                and not self.ceil_mode and not self.return_indices and _bn_ops.maxpool3s2_supported(x)):
            if torch.is_grad_enabled() and x.requires_grad:
                return _MaxPool3s2Fn.apply(x)
            return _bn_ops.maxpool3s2_fwd(x)[0]
        return super().forward(x)


class _ShadowWeightFn(torch.autograd.Function):
    """bf16 image of an fp32 weight as seen by autograd: forward hands out the image (no cast kernel), backward
    returns the gradient in fp32 to the master weight -- what autocast's own cast node does."""

    @staticmethod
    def forward(ctx, weight, shadow):
        ctx.wshape, ctx.wstride = weight.shape, weight.stride()
        return shadow.view_as(shadow)

    @staticmethod
    def backward(ctx, g):
        # MIOpen returns some weight gradients NCHW-dense although the weight is channels-last: cast and re-lay in ONE
        # kernel, so that neither the optimizer (slot order = the parameter's layout) nor DDP (bucket views follow the
        # parameter's strides) has to copy the gradient again
        if any(sg != sw for sg, sw, n in zip(g.stride(), ctx.wstride, ctx.wshape) if n > 1):
            return torch.empty_strided(ctx.wshape, ctx.wstride, dtype=torch.float32, device=g.device).copy_(g), None
        return g.to(torch.float32), None


class _Conv1x1Fn(torch.autograd.Function):
    """1x1 stride-1 convolution on channels-last bf16 activations = a plain GEMM over the [N*H*W, C] view.
    MIOpen's implicit-GEMM solvers lose to hipBLASLt on the data gradient of almost every such layer of the ResNet
    (MI355X, graph-replayed launches: 393 -> 263 us summed over the distinct ResNet-50 shapes) and on the forward of
    the wide low-resolution ones, while hipBLASLt's [Co, M] x [M, Ci] weight gradient is 4-15x slower at large M:
    forward and data gradient pick the GEMM per shape (`mm_fwd`, `mm_dgrad`); the weight gradient is the hand-written
    cp2_wgrad1x1 (csrc/wgrad.hip: deterministic, fp32 output, at or above MIOpen's speed on every ResNet-50 shape).
    `weight` is the fp32 master (receives the fp32 gradient, as autocast's cast node would deliver it), `shadow` its
    bf16 image."""

    hip_wgrad = True         # weight gradient by cp2_wgrad1x1 (False: MIOpen) -- A/B switch

    @staticmethod
    def forward(ctx, x, weight, shadow, bias, mm_fwd, mm_dgrad):
        N, C, H, W = x.shape
        co = shadow.shape[0]
        b16 = bias.to(torch.bfloat16) if bias is not None else None
        if mm_fwd:
            x2, w2 = x.permute(0, 2, 3, 1).reshape(-1, C), shadow.reshape(co, C)
            y2 = torch.mm(x2, w2.t()) if b16 is None else torch.addmm(b16, x2, w2.t())
            y = y2.view(N, H, W, co).permute(0, 3, 1, 2)
        else:
            y = F.conv2d(x, shadow, b16)
        ctx.save_for_backward(x, shadow)
        ctx.mm_dgrad, ctx.has_bias = mm_dgrad, bias is not None
        return y

    @staticmethod
    def backward(ctx, dy):
        x, w = ctx.saved_tensors
        N, C, H, W = x.shape
        co = w.shape[0]
        if dy.dtype != torch.bfloat16 or not dy.is_contiguous(memory_format=torch.channels_last):
            dy = dy.to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
        need_dx, need_dw, need_db = ctx.needs_input_grad[0], ctx.needs_input_grad[1], ctx.has_bias and ctx.needs_input_grad[3]
        dx = None
        if need_dx and ctx.mm_dgrad:
            dx = torch.mm(dy.permute(0, 2, 3, 1).reshape(-1, co), w.reshape(co, C)).view(N, H, W, C).permute(0, 3, 1, 2)
        dw = None
        if need_dw and _Conv1x1Fn.hip_wgrad and _bn_ops.wgrad1x1_supported(co, C):
            dw = _bn_ops.wgrad1x1(dy, x)          # deterministic split-K on the matrix cores, fp32 result (csrc/wgrad.hip)
        # bias gradient as two block-local fp32 reductions (no multi-block semaphore reduction: see autograd_ext.cpp)
        db = dy.permute(0, 2, 3, 1).reshape(N, H * W, co).sum(1, dtype=torch.float32).sum(0) if need_db else None
        want = [need_dx and dx is None, need_dw and dw is None, False]
        if any(want):
            rest = torch.ops.aten.convolution_backward(dy, x, w, None, [1, 1], [0, 0], [1, 1], False, [0, 0], 1, want)
            if want[0]:
                dx = rest[0]
            if want[1]:
                dw = rest[1].to(torch.float32)
        return dx, dw, None, db, None, None


class _ConvKxKFn(torch.autograd.Function):
    """Python twin of the C++ node conv_kxk (csrc_torch/autograd_ext.cpp): k x k convolution whose weight gradient comes
    from cp2_wgrad_conv in fp32 and in the master weight's channels-last layout; forward and data gradient by MIOpen."""

    @staticmethod
    def forward(ctx, x, weight, shadow, bias, stride, pad, dil):
        b16 = bias.to(torch.bfloat16) if bias is not None else None
        ctx.save_for_backward(x, shadow)
        ctx.geom, ctx.has_bias = (stride, pad, dil), bias is not None
        return F.conv2d(x, shadow, b16, stride, pad, dil)

    @staticmethod
    def backward(ctx, dy):
        x, w = ctx.saved_tensors
        stride, pad, dil = ctx.geom
        if dy.dtype != torch.bfloat16 or not dy.is_contiguous(memory_format=torch.channels_last):
            dy = dy.to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
        N, co, OH, OW = dy.shape
        dx = dw = db = None
        if ctx.needs_input_grad[1]:
            dw = _bn_ops.wgrad_conv(dy, x, w.shape[2], stride, pad, dil)
        if ctx.has_bias and ctx.needs_input_grad[3]:
            db = dy.permute(0, 2, 3, 1).reshape(N, OH * OW, co).sum(1, dtype=torch.float32).sum(0)
        if ctx.needs_input_grad[0]:
            dx = torch.ops.aten.convolution_backward(dy, x, w, None, [stride, stride], [pad, pad], [dil, dil], False, [0, 0], 1,
                                                     [True, False, False])[0]
        return dx, dw, None, db, None, None, None


class Conv2d(nn.Conv2d):
    """nn.Conv2d that uses `shadow_weight` -- a bf16 copy of `weight` kept current by someone else (the EMA kernel
    writes it for the key encoder, the optimizer kernel for the query encoder) -- when the input is bf16, so autocast
    launches no per-tensor cast kernel; 1x1 stride-1 layers go through _Conv1x1Fn (GEMM where it is faster)."""

    shadow_weight = None
    gemm_1x1 = True          # class-wide switch (A/B)
    cpp_nodes = True         # autograd nodes from the C++ extension when it is built (A/B: False = the Python nodes)
    hip_wgrad_kxk = True     # weight gradient of the k x k layers by cp2_wgrad_conv (A/B: False = MIOpen)
    kxk_wgrad_max_flops = 32e9   # ... for layers up to this much weight-gradient work in eager steps (see _kxk_wgrad_small)

    def _kxk_wgrad_small(self, x) -> bool:
        """cp2_wgrad_conv or MIOpen for this layer's weight gradient.
        Measured per shape (tools/wgrad_conv_vs_miopen.py, MIOpen incl. its fill / cast launches): cp2_wgrad_conv is
        ahead up to ~30 GFLOP of weight-gradient work (64->64 at 32 x 56^2: 37 vs 42 us; 128->128: 30-38 vs 35-44;
        512->512 at 32 x 14^2: 93 vs 95), behind above it (256->256 at 8 x 64^2, 39 GFLOP: 103 vs 96 us; 512->512 at
        8 x 64^2: 350 vs 264; the FCN head at 32 x 14^2: 250 / 273 vs 236 / 262, at 8 x 64^2: 1200 vs 859)."""
        k, s_, p_, d_ = self.kernel_size[0], self.stride[0], self.padding[0], self.dilation[0]
        oh = (x.shape[2] + 2 * p_ - d_ * (k - 1) - 1) // s_ + 1
        ow = (x.shape[3] + 2 * p_ - d_ * (k - 1) - 1) // s_ + 1
        return 2.0 * x.shape[0] * oh * ow * self.in_channels * self.out_channels * k * k <= Conv2d.kxk_wgrad_max_flops

    def forward(self, x):
        w = self.shadow_weight
        if w is not None and x.dtype == torch.bfloat16:
            ext = _cext.load() if Conv2d.cpp_nodes else None
            stride1 = self.stride == (1, 1)
            if (Conv2d.gemm_1x1 and self.kernel_size == (1, 1) and stride1 and self.padding == (0, 0)
                    and self.groups == 1 and x.dim() == 4 and x.is_contiguous(memory_format=torch.channels_last)):
                m, ci, co = x.shape[0] * x.shape[2] * x.shape[3], self.in_channels, self.out_channels
                mm_fwd = m <= 32768 and ci * co >= 131072
                if not (torch.is_grad_enabled() and (self.weight.requires_grad or x.requires_grad)):
                    if mm_fwd:                      # gradient-free forward (key encoder): no autograd node needed
                        return _Conv1x1Fn.forward(_NoCtx, x, None, w, self.bias, True, False)
                elif ext is not None:
                    # C++ node: GEMM forward / data gradient where faster, weight gradient by cp2_wgrad1x1 (fp32)
                    return ext.conv1x1(x, self.weight, w, self.bias, mm_fwd, ci >= 128, _Conv1x1Fn.hip_wgrad)
                # without the C++ extension the layer stays on MIOpen: the Python twin of the node (_Conv1x1Fn) costs
                # ~0.1 ms of host time per layer and step, which makes the eager step host-bound
            if (Conv2d.hip_wgrad_kxk and self.kernel_size[0] == self.kernel_size[1] and self.kernel_size[0] > 1 and self.groups == 1
                    and self.stride[0] == self.stride[1] and self.padding[0] == self.padding[1] and self.dilation[0] == self.dilation[1]
                    and isinstance(self.padding[0], int) and self.in_channels % 64 == 0 and self.out_channels % 64 == 0
                    and x.dim() == 4 and x.is_contiguous(memory_format=torch.channels_last)
                    and self.weight.requires_grad and torch.is_grad_enabled()
                    and self.weight.is_contiguous(memory_format=torch.channels_last)
                    and self._kxk_wgrad_small(x)):
                # k x k layers: weight gradient by cp2_wgrad_conv (fp32, deterministic) instead of MIOpen's zero-fill +
                # atomic split-K kernel + cast, forward / data gradient unchanged (MIOpen)
                if ext is not None:
                    return ext.conv_kxk(x, self.weight, w, self.bias, self.stride[0], self.padding[0], self.dilation[0])
                return _ConvKxKFn.apply(x, self.weight, w, self.bias, self.stride[0], self.padding[0], self.dilation[0])
            if self.weight.requires_grad and torch.is_grad_enabled():
                w = ext.shadow_weight(self.weight, w) if ext is not None else _ShadowWeightFn.apply(self.weight, w)
            b = self.bias.to(torch.bfloat16) if self.bias is not None else None
            return self._conv_forward(x, w, b)
        return super().forward(x)


class _NoCtx:
    """Stand-in for the autograd context when _Conv1x1Fn.forward is used as a plain function."""

    @staticmethod
    def save_for_backward(*a):
        pass


class ConvBNAct(nn.Module):
    """conv -> BN -> ReLU with mmcv ConvModule's sub-module names (`conv`, `bn`, `activate`)."""

    def __init__(self, cin, cout, k, padding=0, dilation=1, norm=True):
        super().__init__()
        self.conv = Conv2d(cin, cout, k, padding=padding, dilation=dilation, bias=not norm)
        if norm:
            self.bn = FusedBatchNorm2d(cout)
        self.activate = nn.ReLU(inplace=True)
        self.with_norm = norm

    def forward(self, x):
        x = self.conv(x)
        if self.with_norm:
            return self.bn(x, relu=True)
        return self.activate(x)


def _downsample(cin, cout, stride):
    return nn.Sequential(Conv2d(cin, cout, 1, stride=stride, bias=False), FusedBatchNorm2d(cout))


class BasicBlock(nn.Module):
    expansion = 1

    def __init__(self, inplanes, planes, stride=1, dilation=1, downsample=None):
        super().__init__()
        self.conv1 = Conv2d(inplanes, planes, 3, stride=stride, padding=dilation, dilation=dilation, bias=False)
        self.bn1 = FusedBatchNorm2d(planes)
        self.conv2 = Conv2d(planes, planes, 3, padding=1, bias=False)
        self.bn2 = FusedBatchNorm2d(planes)
        self.relu = nn.ReLU(inplace=True)
        self.downsample = downsample

    def forward(self, x):
        idt = x if self.downsample is None else self.downsample(x)
        y = self.bn1(self.conv1(x), relu=True)
        return self.bn2(self.conv2(y), residual=idt, relu=True)


class Bottleneck(nn.Module):
    expansion = 4

    def __init__(self, inplanes, planes, stride=1, dilation=1, downsample=None):
        super().__init__()   # style='pytorch': the stride sits on the 3x3 conv
        self.conv1 = Conv2d(inplanes, planes, 1, bias=False)
        self.bn1 = FusedBatchNorm2d(planes)
        self.conv2 = Conv2d(planes, planes, 3, stride=stride, padding=dilation, dilation=dilation, bias=False)
        self.bn2 = FusedBatchNorm2d(planes)
        self.conv3 = Conv2d(planes, planes * 4, 1, bias=False)
        self.bn3 = FusedBatchNorm2d(planes * 4)
        self.relu = nn.ReLU(inplace=True)
        self.downsample = downsample

    def forward(self, x):
        idt = x if self.downsample is None else self.downsample(x)
        y = self.bn1(self.conv1(x), relu=True)
        y = self.bn2(self.conv2(y), relu=True)
        return self.bn3(self.conv3(y), residual=idt, relu=True)


_ARCH = {18: (BasicBlock, (2, 2, 2, 2)), 34: (BasicBlock, (3, 4, 6, 3)), 50: (Bottleneck, (3, 4, 6, 3)),
         101: (Bottleneck, (3, 4, 23, 3)), 152: (Bottleneck, (3, 8, 36, 3))}


class ResNet(nn.Module):
    def __init__(self, depth=50, in_channels=3, stem_channels=64, base_channels=64, num_stages=4,
                 strides=(1, 2, 2, 2), dilations=(1, 1, 1, 1), out_indices=(0, 1, 2, 3), style="pytorch",
                 contract_dilation=False, zero_init_residual=True, norm_cfg=None, norm_eval=False,
                 init_cfg=None, **unused):
        super().__init__()
        if depth not in _ARCH:
            raise KeyError(f"invalid depth {depth} for resnet")
        if style != "pytorch":
            raise NotImplementedError("only style='pytorch' is restated")
        block, blocks = _ARCH[depth]
        self.depth, self.out_indices, self.norm_eval = depth, tuple(out_indices), norm_eval
        self.zero_init_residual, self.init_cfg = zero_init_residual, init_cfg
        self.conv1 = Conv2d(in_channels, stem_channels, 7, stride=2, padding=3, bias=False)
        self.bn1 = FusedBatchNorm2d(stem_channels)
        self.relu = nn.ReLU(inplace=True)
        self.maxpool = StemMaxPool(3, stride=2, padding=1)
        inplanes = stem_channels
        self.res_layers = []
        for i, nb in enumerate(blocks[:num_stages]):
            planes = base_channels * 2 ** i
            stride, dil = strides[i], dilations[i]
            first_dil = dil // 2 if (dil > 1 and contract_dilation) else dil
            down = _downsample(inplanes, planes * block.expansion, stride) \
                if (stride != 1 or inplanes != planes * block.expansion) else None
            layers = [block(inplanes, planes, stride, first_dil, down)]
            inplanes = planes * block.expansion
            layers += [block(inplanes, planes, 1, dil) for _ in range(1, nb)]
            name = f"layer{i + 1}"
            self.add_module(name, nn.Sequential(*layers))
            self.res_layers.append(name)
        self.feat_dim = inplanes

    def init_weights(self, pretrained=None):
        """ImageNet initialisation when a LOCAL checkpoint is configured and present
        (the reference's default 'torchvision://resnet50' needs the network: configs/config_pretrain.py:3,18);
        otherwise Kaiming / constant init as mmseg does without a checkpoint (resnet.py:600-630)."""
        ckpt = pretrained or (self.init_cfg or {}).get("checkpoint")
        if ckpt and os.path.isfile(str(ckpt)):
            state = torch.load(ckpt, map_location="cpu")
            state = state.get("state_dict", state)
            missing, unexpected = self.load_state_dict(state, strict=False)
            print(f"[cp2_amd] backbone weights from {ckpt}: {len(missing)} missing, {len(unexpected)} unexpected")
            return
        if ckpt:
            warnings.warn(f"backbone checkpoint {ckpt!r} is not a local file (no network here): random init")
        for m in self.modules():
            if isinstance(m, nn.Conv2d):
                nn.init.kaiming_normal_(m.weight, mode="fan_out", nonlinearity="relu")
            elif isinstance(m, nn.BatchNorm2d):
                nn.init.ones_(m.weight)
                nn.init.zeros_(m.bias)
        if self.zero_init_residual:
            for m in self.modules():
                if isinstance(m, Bottleneck):
                    nn.init.zeros_(m.bn3.weight)
                elif isinstance(m, BasicBlock):
                    nn.init.zeros_(m.bn2.weight)

    def forward(self, x):
        x = self.maxpool(self.bn1(self.conv1(x), relu=True))
        outs = []
        for i, name in enumerate(self.res_layers):
            x = getattr(self, name)(x)
            if i in self.out_indices:
                outs.append(x)
        return tuple(outs)

    def train(self, mode=True):
        super().train(mode)
        if mode and self.norm_eval:
            for m in self.modules():
                if isinstance(m, nn.BatchNorm2d):
                    m.eval()
        return self


class _DecodeHead(nn.Module):
    def __init__(self, in_channels, channels, num_classes, dropout_ratio=0.1, in_index=-1, align_corners=False,
                 contrast=False, **unused):
        super().__init__()
        self.in_channels, self.channels, self.num_classes = in_channels, channels, num_classes
        self.in_index, self.align_corners, self.contrast = in_index, align_corners, contrast
        self.conv_seg = nn.Conv2d(channels, num_classes, 1)
        self.dropout = nn.Dropout2d(dropout_ratio) if dropout_ratio > 0 else None
        nn.init.normal_(self.conv_seg.weight, mean=0.0, std=0.01)
        nn.init.zeros_(self.conv_seg.bias)

    def _add_contrast(self):
        if self.contrast:     # aspp_head.py:93-97 / fcn_head.py:75-79: the 128-d 1x1-conv projector
            self.contrast_conv = nn.Sequential(Conv2d(self.channels, self.channels, 1), nn.ReLU(),
                                               Conv2d(self.channels, 128, 1))
            # conv_seg never receives a gradient on the contrast path; freezing it keeps its
            # state-dict entry while letting DDP run without find_unused_parameters.
            self.conv_seg.requires_grad_(False)

    def cls_seg(self, feat):
        if self.dropout is not None:
            feat = self.dropout(feat)
        return self.conv_seg(feat)

    def _finish(self, feat):
        return self.contrast_conv(feat) if self.contrast else self.cls_seg(feat)


class ASPPHead(_DecodeHead):
    def __init__(self, dilations=(1, 6, 12, 18), **kw):
        super().__init__(**kw)
        cin, ch = self.in_channels, self.channels
        self.image_pool = nn.Sequential(nn.AdaptiveAvgPool2d(1), ConvBNAct(cin, ch, 1))
        self.aspp_modules = nn.ModuleList(
            [ConvBNAct(cin, ch, 1 if d == 1 else 3, padding=0 if d == 1 else d, dilation=d) for d in dilations])
        self.bottleneck = ConvBNAct((len(dilations) + 1) * ch, ch, 3, padding=1)
        self._add_contrast()

    def forward(self, inputs):
        x = inputs[self.in_index]
        pooled = F.interpolate(self.image_pool(x), size=x.shape[2:], mode="bilinear", align_corners=self.align_corners)
        feats = torch.cat([pooled] + [m(x) for m in self.aspp_modules], dim=1)
        return self._finish(self.bottleneck(feats))


class FCNHead(_DecodeHead):
    def __init__(self, num_convs=2, kernel_size=3, concat_input=True, dilation=1, **kw):
        super().__init__(**kw)
        cin, ch = self.in_channels, self.channels
        if num_convs == 0:
            assert cin == ch
            self.convs = nn.Identity()
        else:
            pad = (kernel_size // 2) * dilation
            self.convs = nn.Sequential(*[ConvBNAct(cin if i == 0 else ch, ch, kernel_size, padding=pad, dilation=dilation)
                                         for i in range(num_convs)])
        self.concat_input = concat_input
        if concat_input:
            self.conv_cat = ConvBNAct(cin + ch, ch, kernel_size, padding=kernel_size // 2)
        self._add_contrast()

    def forward(self, inputs):
        x = inputs[self.in_index]
        y = self.convs(x)
        if self.concat_input:
            y = self.conv_cat(torch.cat([x, y], dim=1))
        return self._finish(y)


class EncoderDecoder(nn.Module):
    """backbone + decode_head; `forward(img)` returns the head output on the feature grid
    (mmseg_/models/segmentors/encoder_decoder.py:137-140 with a `contrast` head)."""

    def __init__(self, backbone: nn.Module, decode_head: nn.Module):
        super().__init__()
        self.backbone, self.decode_head = backbone, decode_head

    def extract_feat(self, img):
        return self.backbone(img)

    def forward(self, img):
        return self.decode_head(self.backbone(img))


_BACKBONES = {"ResNet": ResNet}
_HEADS = {"ASPPHead": ASPPHead, "FCNHead": FCNHead}


def build_segmentor(model_cfg, train_cfg=None, test_cfg=None) -> EncoderDecoder:
    """Same call shape as mmseg.models.build_segmentor (mmseg_/models/builder.py:35-46) for the
    config schema of configs/config_pretrain.py / config_moco.py."""
    cfg = dict(model_cfg)
    if cfg.get("type", "EncoderDecoder") != "EncoderDecoder":
        raise NotImplementedError(f"segmentor type {cfg.get('type')!r}")
    bb = dict(cfg["backbone"])
    hd = dict(cfg["decode_head"])
    backbone = _BACKBONES[bb.pop("type")](**bb)
    hd.pop("norm_cfg", None), hd.pop("loss_decode", None)
    head = _HEADS[hd.pop("type")](**hd)
    return EncoderDecoder(backbone, head)
