// C++ autograd nodes for the encoder fast path (host-side only: no kernels here).
//
// The Python torch.autograd.Function versions of these two nodes (cp2_amd/encoder.py: _ShadowWeightFn, _Conv1x1Fn) cost
// 25-120 us of interpreter time per layer and step, which made the eager step host-bound as soon as the 1x1 layers were
// routed through them.  Here the same nodes are torch::autograd::Function subclasses: a few microseconds each.
//   shadow_weight(weight, shadow)  forward: the bf16 image `shadow` of the fp32 master `weight` (no cast kernel);
//                                  backward: the gradient cast to fp32 AND re-laid to the master's strides in one kernel.
//   conv1x1(x, weight, shadow, bias, mm_fwd, mm_dgrad, hip_wgrad)
//                                  1x1 stride-1 convolution on channels-last bf16 activations: forward / data gradient
//                                  by hipBLASLt GEMM where the caller says so (else MIOpen), weight gradient by
//                                  cp2_wgrad1x1 from libcp2hip.so (function pointers handed over by set_wgrad), fp32.
#include <torch/extension.h>
#include <c10/hip/HIPStream.h>

namespace {

typedef int (*wgrad_fn_t)(const void*, const void*, float*, float*, int, int, int, void*);
typedef int (*wgrad_splits_fn_t)(int, int, int);
wgrad_fn_t g_wgrad = nullptr;
wgrad_splits_fn_t g_wgrad_splits = nullptr;

typedef int (*wgrad_conv_fn_t)(const void*, const void*, float*, float*, int, int, int, int, int, int, int, int, int, int, int, int, void*);
typedef int (*wgrad_conv_splits_fn_t)(int, int, int, int, int, int, int);
wgrad_conv_fn_t g_wgrad_conv = nullptr;
wgrad_conv_splits_fn_t g_wgrad_conv_splits = nullptr;

void set_wgrad_conv(int64_t fn, int64_t splits_fn) {
    g_wgrad_conv = reinterpret_cast<wgrad_conv_fn_t>(fn);
    g_wgrad_conv_splits = reinterpret_cast<wgrad_conv_splits_fn_t>(splits_fn);
}

void set_wgrad(int64_t fn, int64_t splits_fn) {
    g_wgrad = reinterpret_cast<wgrad_fn_t>(fn);
    g_wgrad_splits = reinterpret_cast<wgrad_splits_fn_t>(splits_fn);
}

bool same_element_order(at::IntArrayRef gs, at::IntArrayRef ws, at::IntArrayRef shape) {
    for (size_t i = 0; i < shape.size(); ++i)
        if (shape[i] > 1 && gs[i] != ws[i]) return false;
    return true;
}

at::Tensor grad_to_master(const at::Tensor& g, at::IntArrayRef wshape, at::IntArrayRef wstride) {
    if (!same_element_order(g.strides(), wstride, wshape))
        return at::empty_strided(wshape, wstride, g.options().dtype(at::kFloat)).copy_(g);
    return g.to(at::kFloat);
}

struct ShadowWeightFn : public torch::autograd::Function<ShadowWeightFn> {
    static at::Tensor forward(torch::autograd::AutogradContext* ctx, const at::Tensor& weight, const at::Tensor& shadow) {
        ctx->saved_data["wshape"] = weight.sizes().vec();
        ctx->saved_data["wstride"] = weight.strides().vec();
        return shadow.alias();
    }
    static torch::autograd::variable_list backward(torch::autograd::AutogradContext* ctx, torch::autograd::variable_list grads) {
        const auto wshape = ctx->saved_data["wshape"].toIntVector(), wstride = ctx->saved_data["wstride"].toIntVector();
        return {grad_to_master(grads[0], wshape, wstride), at::Tensor()};
    }
};

struct Conv1x1Fn : public torch::autograd::Function<Conv1x1Fn> {
    static at::Tensor forward(torch::autograd::AutogradContext* ctx, const at::Tensor& x, const at::Tensor& weight,
                              const at::Tensor& shadow, const c10::optional<at::Tensor>& bias, bool mm_fwd, bool mm_dgrad,
                              bool hip_wgrad) {
        const int64_t N = x.size(0), C = x.size(1), H = x.size(2), W = x.size(3), co = shadow.size(0);
        c10::optional<at::Tensor> b16;
        if (bias.has_value() && bias->defined()) b16 = bias->to(at::kBFloat16);
        at::Tensor y;
        if (mm_fwd) {
            const at::Tensor x2 = x.permute({0, 2, 3, 1}).reshape({-1, C}), w2 = shadow.reshape({co, C});
            const at::Tensor y2 = b16.has_value() ? at::addmm(*b16, x2, w2.t()) : at::mm(x2, w2.t());
            y = y2.view({N, H, W, co}).permute({0, 3, 1, 2});
        } else {
            y = at::conv2d(x, shadow, b16);
        }
        ctx->save_for_backward({x, shadow});
        ctx->saved_data["mm_dgrad"] = mm_dgrad;
        ctx->saved_data["hip_wgrad"] = hip_wgrad;
        ctx->saved_data["has_bias"] = b16.has_value();
        ctx->saved_data["wshape"] = weight.sizes().vec();
        ctx->saved_data["wstride"] = weight.strides().vec();
        return y;
    }

    static torch::autograd::variable_list backward(torch::autograd::AutogradContext* ctx, torch::autograd::variable_list grads) {
        const auto saved = ctx->get_saved_variables();
        const at::Tensor& x = saved[0];
        const at::Tensor& w = saved[1];
        const int64_t N = x.size(0), C = x.size(1), H = x.size(2), W = x.size(3), co = w.size(0);
        at::Tensor dy = grads[0];
        if (dy.scalar_type() != at::kBFloat16 || !dy.is_contiguous(at::MemoryFormat::ChannelsLast))
            dy = dy.to(at::kBFloat16).contiguous(at::MemoryFormat::ChannelsLast);
        const bool need_dx = ctx->needs_input_grad(0), need_dw = ctx->needs_input_grad(1);
        const bool need_db = ctx->saved_data["has_bias"].toBool() && ctx->needs_input_grad(3);
        at::Tensor dx, dw, db;
        if (need_dx && ctx->saved_data["mm_dgrad"].toBool())
            dx = at::mm(dy.permute({0, 2, 3, 1}).reshape({-1, co}), w.reshape({co, C})).view({N, H, W, C}).permute({0, 3, 1, 2});
        if (need_dw && ctx->saved_data["hip_wgrad"].toBool() && g_wgrad != nullptr && co % 64 == 0 && C % 64 == 0 &&
            x.is_contiguous(at::MemoryFormat::ChannelsLast)) {
            const int M = (int)(N * H * W);
            const int S = g_wgrad_splits(M, (int)co, (int)C);
            TORCH_CHECK(S >= 1, "cp2_wgrad1x1_num_splits failed: ", S);
            // the master's own strides (for a [co, ci, 1, 1] weight every dense layout is the same memory): the gradient
            // then matches DDP's bucket views and the optimizer's slot order without another copy
            dw = at::empty_strided(ctx->saved_data["wshape"].toIntVector(), ctx->saved_data["wstride"].toIntVector(),
                                   x.options().dtype(at::kFloat));
            at::Tensor part = S > 1 ? at::empty({(int64_t)S * co * C}, dw.options()) : dw;
            const int rc = g_wgrad(dy.data_ptr(), x.data_ptr(), dw.data_ptr<float>(), part.data_ptr<float>(), M, (int)co, (int)C,
                                   c10::hip::getCurrentHIPStream().stream());
            TORCH_CHECK(rc == 0, "cp2_wgrad1x1 failed: ", rc);
        }
        // Bias gradient = column sums of dy, as two block-local reductions ([N, H*W, co] over H*W, then over N) in fp32.
        // ATen's own bias gradient for channels-last (one sum over N, H, W) is a multi-block reduction with a zero-filled
        // semaphore: under whole-step hipGraph replay it returned a non-finite element from the second replay on
        // (tools/graph_verify_probe.py) -- the same "zero-fill, then accumulate" shape as the
        // MIOpen weight-gradient solvers that failed there.
        if (need_db) db = dy.permute({0, 2, 3, 1}).reshape({N, H * W, co}).sum(1, false, at::kFloat).sum(0);
        const bool rest_dx = need_dx && !dx.defined(), rest_dw = need_dw && !dw.defined();
        if (rest_dx || rest_dw) {
            const auto r = at::convolution_backward(dy, x, w, c10::nullopt, {1, 1}, {0, 0}, {1, 1}, false, {0, 0}, 1,
                                                    {rest_dx, rest_dw, false});
            if (rest_dx) dx = std::get<0>(r);
            if (rest_dw) dw = grad_to_master(std::get<1>(r), ctx->saved_data["wshape"].toIntVector(), ctx->saved_data["wstride"].toIntVector());
        }
        return {dx, dw, at::Tensor(), db, at::Tensor(), at::Tensor(), at::Tensor()};
    }
};

// k x k convolution (the 3x3 layers): forward and data gradient by MIOpen, weight gradient by cp2_wgrad_conv -- fp32, in the
// master weight's channels-last strides, deterministic; MIOpen's split-K weight-gradient solvers for these layers run a
// zero-fill kernel, the atomic-accumulating kernel and a cast kernel, and hand back bf16 that is cast to fp32 once more.
struct ConvKxKFn : public torch::autograd::Function<ConvKxKFn> {
    static at::Tensor forward(torch::autograd::AutogradContext* ctx, const at::Tensor& x, const at::Tensor& weight,
                              const at::Tensor& shadow, const c10::optional<at::Tensor>& bias, int64_t stride, int64_t pad,
                              int64_t dil) {
        c10::optional<at::Tensor> b16;
        if (bias.has_value() && bias->defined()) b16 = bias->to(at::kBFloat16);
        const at::Tensor y = at::conv2d(x, shadow, b16, {stride, stride}, {pad, pad}, {dil, dil});
        ctx->save_for_backward({x, shadow});
        ctx->saved_data["geom"] = std::vector<int64_t>{stride, pad, dil};
        ctx->saved_data["has_bias"] = b16.has_value();
        ctx->saved_data["wshape"] = weight.sizes().vec();
        ctx->saved_data["wstride"] = weight.strides().vec();
        return y;
    }

    static torch::autograd::variable_list backward(torch::autograd::AutogradContext* ctx, torch::autograd::variable_list grads) {
        const auto saved = ctx->get_saved_variables();
        const at::Tensor& x = saved[0];
        const at::Tensor& w = saved[1];
        const auto geom = ctx->saved_data["geom"].toIntVector();
        const int64_t stride = geom[0], pad = geom[1], dil = geom[2];
        const int64_t N = x.size(0), C = x.size(1), H = x.size(2), W = x.size(3), co = w.size(0), k = w.size(2);
        at::Tensor dy = grads[0];
        if (dy.scalar_type() != at::kBFloat16 || !dy.is_contiguous(at::MemoryFormat::ChannelsLast))
            dy = dy.to(at::kBFloat16).contiguous(at::MemoryFormat::ChannelsLast);
        const int64_t OH = dy.size(2), OW = dy.size(3);
        const bool need_dx = ctx->needs_input_grad(0), need_dw = ctx->needs_input_grad(1);
        const bool need_db = ctx->saved_data["has_bias"].toBool() && ctx->needs_input_grad(3);
        at::Tensor dx, dw, db;
        if (need_dw) {
            const int S = g_wgrad_conv_splits((int)N, (int)OH, (int)OW, (int)co, (int)C, (int)k, (int)k);
            TORCH_CHECK(S >= 1, "cp2_wgrad_conv_num_splits failed: ", S);
            dw = at::empty_strided(ctx->saved_data["wshape"].toIntVector(), ctx->saved_data["wstride"].toIntVector(),
                                   x.options().dtype(at::kFloat));
            at::Tensor part = S > 1 ? at::empty({(int64_t)S * dw.numel()}, dw.options()) : dw;
            const int rc = g_wgrad_conv(dy.data_ptr(), x.data_ptr(), dw.data_ptr<float>(), part.data_ptr<float>(), (int)N, (int)H, (int)W,
                                        (int)OH, (int)OW, (int)co, (int)C, (int)k, (int)k, (int)stride, (int)pad, (int)dil,
                                        c10::hip::getCurrentHIPStream().stream());
            TORCH_CHECK(rc == 0, "cp2_wgrad_conv failed: ", rc);
        }
        if (need_db) db = dy.permute({0, 2, 3, 1}).reshape({N, OH * OW, co}).sum(1, false, at::kFloat).sum(0);
        if (need_dx) {
            const auto r = at::convolution_backward(dy, x, w, c10::nullopt, {stride, stride}, {pad, pad}, {dil, dil}, false, {0, 0}, 1,
                                                    {true, false, false});
            dx = std::get<0>(r);
        }
        return {dx, dw, at::Tensor(), db, at::Tensor(), at::Tensor(), at::Tensor()};
    }
};

at::Tensor shadow_weight(const at::Tensor& weight, const at::Tensor& shadow) { return ShadowWeightFn::apply(weight, shadow); }

at::Tensor conv_kxk(const at::Tensor& x, const at::Tensor& weight, const at::Tensor& shadow, const c10::optional<at::Tensor>& bias,
                    int64_t stride, int64_t pad, int64_t dil) {
    TORCH_CHECK(g_wgrad_conv != nullptr, "set_wgrad_conv has not been called");
    return ConvKxKFn::apply(x, weight, shadow, bias, stride, pad, dil);
}

at::Tensor conv1x1(const at::Tensor& x, const at::Tensor& weight, const at::Tensor& shadow, const c10::optional<at::Tensor>& bias,
                   bool mm_fwd, bool mm_dgrad, bool hip_wgrad) {
    return Conv1x1Fn::apply(x, weight, shadow, bias, mm_fwd, mm_dgrad, hip_wgrad);
}

}  // namespace

// Fused training-mode BatchNorm (+ residual) (+ ReLU) on channels-last bf16 activations: cp2_bn_fwd / cp2_bn_bwd of
// libcp2hip.so (csrc/bn.hip).  The Python node (encoder._FusedBNFn + ops.bn_fwd / bn_bwd) costs ~21 us of interpreter time
// per forward call and about as much per backward call; the ResNet-50 step makes 53 + 53 of them on the query encoder
// (tools/host_profile.py: 1.2 ms of the 4.6 ms the host needs to enqueue the query forward).  Workspace layout as ops.bn_fwd:
// one fp32 [2G + 4, C] block = partials | scale, shift | mean, invstd; the last two rows are what backward needs.
typedef int (*bn_partials_fn_t)(int, int);
typedef int (*bn_fwd_fn_t)(const void*, const void*, const float*, const float*, float*, float*, float, float, int, void*, float*,
                           float*, float*, float*, int, int, void*);
typedef int (*bn_bwd_fn_t)(const void*, const void*, const void*, const float*, const float*, const float*, int, void*, void*,
                           float*, float*, float*, float*, int, int, void*);
bn_partials_fn_t g_bn_partials = nullptr;
bn_fwd_fn_t g_bn_fwd = nullptr;
bn_bwd_fn_t g_bn_bwd = nullptr;

void set_bn(int64_t partials_fn, int64_t fwd_fn, int64_t bwd_fn) {
    g_bn_partials = reinterpret_cast<bn_partials_fn_t>(partials_fn);
    g_bn_fwd = reinterpret_cast<bn_fwd_fn_t>(fwd_fn);
    g_bn_bwd = reinterpret_cast<bn_bwd_fn_t>(bwd_fn);
}

struct FusedBNFn : public torch::autograd::Function<FusedBNFn> {
    static at::Tensor forward(torch::autograd::AutogradContext* ctx, const at::Tensor& x, const at::Tensor& weight,
                              const at::Tensor& bias, const c10::optional<at::Tensor>& residual, at::Tensor running_mean,
                              at::Tensor running_var, double momentum, double eps, bool relu) {
        TORCH_CHECK(g_bn_fwd != nullptr, "fused_bn: set_bn() has not been called");
        const int64_t N = x.size(0), C = x.size(1), H = x.size(2), W = x.size(3);
        const int M = (int)(N * H * W);
        const int G = g_bn_partials(M, (int)C);
        TORCH_CHECK(G >= 1, "cp2_bn_num_partials failed: ", G);
        const bool has_res = residual.has_value() && residual->defined();
        at::Tensor y = at::empty_like(x);
        at::Tensor ws = at::empty({2 * (int64_t)G + 4, C}, x.options().dtype(at::kFloat));
        float* base = ws.data_ptr<float>();
        const int rc = g_bn_fwd(x.data_ptr(), has_res ? residual->data_ptr() : nullptr, weight.data_ptr<float>(), bias.data_ptr<float>(),
                                running_mean.data_ptr<float>(), running_var.data_ptr<float>(), (float)momentum, (float)eps, relu ? 1 : 0,
                                y.data_ptr(), base + (2 * (int64_t)G + 2) * C, base + (2 * (int64_t)G + 3) * C, base,
                                base + 2 * (int64_t)G * C, M, (int)C, c10::hip::getCurrentHIPStream().stream());
        TORCH_CHECK(rc == 0, "cp2_bn_fwd failed: ", rc);
        ctx->save_for_backward({x, relu ? y : at::Tensor(), weight, ws.narrow(0, 2 * (int64_t)G + 2, 2)});
        ctx->saved_data["relu"] = relu;
        ctx->saved_data["has_res"] = has_res;
        ctx->saved_data["G"] = (int64_t)G;
        return y;
    }

    static torch::autograd::variable_list backward(torch::autograd::AutogradContext* ctx, torch::autograd::variable_list grads) {
        const auto saved = ctx->get_saved_variables();
        const at::Tensor& x = saved[0];
        const at::Tensor& y = saved[1];
        const at::Tensor& weight = saved[2];
        const at::Tensor& stats = saved[3];
        const bool relu = ctx->saved_data["relu"].toBool(), has_res = ctx->saved_data["has_res"].toBool();
        const int64_t G = ctx->saved_data["G"].toInt();
        const int64_t N = x.size(0), C = x.size(1), H = x.size(2), W = x.size(3);
        const int M = (int)(N * H * W);
        at::Tensor dy = grads[0];
        if (dy.scalar_type() != at::kBFloat16 || !dy.is_contiguous(at::MemoryFormat::ChannelsLast))
            dy = dy.to(at::kBFloat16).contiguous(at::MemoryFormat::ChannelsLast);
        at::Tensor dx = at::empty_like(x);
        at::Tensor dres;
        if (has_res && relu) dres = at::empty_like(x);
        at::Tensor ws = at::empty({2 * G + 5, C}, x.options().dtype(at::kFloat));       // partials | coef[3] | dgamma, dbeta
        float* base = ws.data_ptr<float>();
        const float* sp = stats.data_ptr<float>();
        const int rc = g_bn_bwd(x.data_ptr(), dy.data_ptr(), relu ? y.data_ptr() : nullptr, weight.data_ptr<float>(), sp, sp + C,
                                relu ? 1 : 0, dx.data_ptr(), dres.defined() ? dres.data_ptr() : nullptr, base + (2 * G + 3) * C,
                                base + (2 * G + 4) * C, base, base + 2 * G * C, M, (int)C, c10::hip::getCurrentHIPStream().stream());
        TORCH_CHECK(rc == 0, "cp2_bn_bwd failed: ", rc);
        if (has_res && !relu) dres = dy;
        return {dx, ws[2 * G + 3], ws[2 * G + 4], dres, at::Tensor(), at::Tensor(), at::Tensor(), at::Tensor(), at::Tensor()};
    }
};

at::Tensor fused_bn(const at::Tensor& x, const at::Tensor& weight, const at::Tensor& bias, const c10::optional<at::Tensor>& residual,
                    at::Tensor running_mean, at::Tensor running_var, double momentum, double eps, bool relu) {
    return FusedBNFn::apply(x, weight, bias, residual, running_mean, running_var, momentum, eps, relu);
}

PYBIND11_MODULE(TORCH_EXTENSION_NAME, m) {
    m.def("set_wgrad", &set_wgrad, "function pointers of cp2_wgrad1x1 / cp2_wgrad1x1_num_splits (libcp2hip.so)");
    m.def("set_wgrad_conv", &set_wgrad_conv, "function pointers of cp2_wgrad_conv / cp2_wgrad_conv_num_splits (libcp2hip.so)");
    m.def("set_bn", &set_bn, "function pointers of cp2_bn_num_partials / cp2_bn_fwd / cp2_bn_bwd (libcp2hip.so)");
    m.def("fused_bn", &fused_bn);
    m.def("conv_kxk", &conv_kxk);
    m.def("shadow_weight", &shadow_weight);
    m.def("conv1x1", &conv1x1);
}
