// Query-encoder optimizer step on the flat parameter buffer -- reference main.py:467-477 (torch.optim.SGD with
// momentum and weight decay, the default multi-tensor implementation) + main.py:640-642 (optimizer.step()).
//   g' = g + wd*p ;  buf = buf*momentum + g' ;  p = p - lr*buf        (dampening 0, no Nesterov)
// evaluated with the roundings of torch's _foreach kernels (fused multiply-add where ATen's `a + alpha*b` contracts),
// so the result is bit-identical to torch.optim.SGD on the same GPU (tests/test_gpu_optim.py).
// One launch instead of 17 multi_tensor_apply launches; HBM-bound: read p, g, buf, write p, buf (20 B per parameter)
// + 2 B for the bf16 copy of the new weights that the query encoder's convolutions read under autocast (so no
// per-tensor cast kernel runs in the forward pass).
// Gradients live wherever autograd / DDP put them: one pointer per parameter tensor in the kernel arguments (NULL:
// the parameter got no gradient and is skipped, as torch does).  Workgroup b handles 512 consecutive floats of one
// parameter's slot; blk_tab[b] = {tensor, offset in the flat buffers, offset in the gradient, valid floats}.
#include "common.hpp"

typedef float sgd_f4 __attribute__((ext_vector_type(4)));
typedef unsigned short sgd_u4 __attribute__((ext_vector_type(4)));
constexpr int kSgdThreads = 128;

struct SgdArgs {
    float* p; float* buf; unsigned short* p_bf16;
    const int4* blk_tab;
    int blk0, t0;
    float lr, momentum, wd;
    const float* lr_dev;                      // overrides lr when not NULL (graph capture: schedule without re-capture)
    const float* grads[CP2_SGD_MAX_TENSORS];
};

__device__ __forceinline__ unsigned short sgd_f2bf(float f) {
    unsigned u = __float_as_uint(f);
    if ((u & 0x7fffffffu) > 0x7f800000u) return (unsigned short)((u >> 16) | 0x40u);
    u += 0x7fffu + ((u >> 16) & 1u);
    return (unsigned short)(u >> 16);
}

__device__ __forceinline__ void sgd1(float& p, float g, float& buf, float lr, float mom, float wd) {
    if (wd != 0.f) g = __fmaf_rn(wd, p, g);            // _foreach_add(grads, params, alpha=wd)
    if (mom != 0.f) {
        buf = __fadd_rn(__fmul_rn(buf, mom), g);        // _foreach_mul_(bufs, momentum); _foreach_add_(bufs, grads, alpha=1)
        g = buf;
    }
    p = __fmaf_rn(-lr, g, p);                           // _foreach_add_(params, bufs, alpha=-lr)
}

__global__ __launch_bounds__(kSgdThreads) void sgd_flat_kernel(SgdArgs a) {
    const int4 e = a.blk_tab[a.blk0 + blockIdx.x];
    const float* __restrict__ g = a.grads[e.x - a.t0];
    if (g == nullptr) return;
    const float lr = a.lr_dev ? *a.lr_dev : a.lr;
    const int i = threadIdx.x * 4;
    if (i >= e.w) return;
    float* p = a.p + e.y + i;
    float* b = a.buf + e.y + i;
    g += e.z + i;
    if (i + 4 <= e.w && (reinterpret_cast<uintptr_t>(g) & 15u) == 0) {
        sgd_f4 pv = __builtin_nontemporal_load(reinterpret_cast<const sgd_f4*>(p));
        sgd_f4 bv = __builtin_nontemporal_load(reinterpret_cast<const sgd_f4*>(b));
        const sgd_f4 gv = __builtin_nontemporal_load(reinterpret_cast<const sgd_f4*>(g));
        float px = pv.x, py = pv.y, pz = pv.z, pw = pv.w, bx = bv.x, by = bv.y, bz = bv.z, bw = bv.w;
        sgd1(px, gv.x, bx, lr, a.momentum, a.wd); sgd1(py, gv.y, by, lr, a.momentum, a.wd);
        sgd1(pz, gv.z, bz, lr, a.momentum, a.wd); sgd1(pw, gv.w, bw, lr, a.momentum, a.wd);
        pv.x = px; pv.y = py; pv.z = pz; pv.w = pw; bv.x = bx; bv.y = by; bv.z = bz; bv.w = bw;
        __builtin_nontemporal_store(pv, reinterpret_cast<sgd_f4*>(p));
        if (a.momentum != 0.f) __builtin_nontemporal_store(bv, reinterpret_cast<sgd_f4*>(b));
        if (a.p_bf16) {
            sgd_u4 s;
            s.x = sgd_f2bf(px); s.y = sgd_f2bf(py); s.z = sgd_f2bf(pz); s.w = sgd_f2bf(pw);
            *reinterpret_cast<sgd_u4*>(a.p_bf16 + e.y + i) = s;
        }
    } else {
        const int n = min(4, e.w - i);
        for (int j = 0; j < n; ++j) {
            float pj = p[j], bj = b[j];
            sgd1(pj, g[j], bj, lr, a.momentum, a.wd);
            p[j] = pj;
            if (a.momentum != 0.f) b[j] = bj;
            if (a.p_bf16) a.p_bf16[e.y + i + j] = sgd_f2bf(pj);
        }
    }
}

// fp32 -> bf16 copy of a flat buffer (round to nearest even, as Tensor.to(bfloat16)): (re)build the bf16 image of the
// query weights after anything but cp2_sgd_flat changed them (initialisation, load_state_dict).
__global__ __launch_bounds__(256) void bf16_image_kernel(const float* __restrict__ src, unsigned short* __restrict__ dst,
                                                         int64_t n4) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n4) return;
    const sgd_f4 v = reinterpret_cast<const sgd_f4*>(src)[i];
    sgd_u4 s;
    s.x = sgd_f2bf(v.x); s.y = sgd_f2bf(v.y); s.z = sgd_f2bf(v.z); s.w = sgd_f2bf(v.w);
    reinterpret_cast<sgd_u4*>(dst)[i] = s;
}

CP2_API int cp2_bf16_image(const float* src, void* dst, int64_t n, void* stream) {
    if (!src || !dst) return CP2_ERR_NULL;
    if (n <= 0 || (n & 3)) return CP2_ERR_SHAPE;
    if (!cp2_aligned16(src) || (reinterpret_cast<uintptr_t>(dst) & 7u)) return CP2_ERR_ALIGN;
    const int64_t n4 = n / 4, blocks = (n4 + 255) / 256;
    if (blocks > 0x7fffffffLL) return CP2_ERR_UNSUPPORTED;
    hipLaunchKernelGGL(bf16_image_kernel, dim3((unsigned)blocks), dim3(256), 0, cp2_stream(stream), src,
                       static_cast<unsigned short*>(dst), n4);
    return cp2_launch_status();
}

CP2_API int cp2_sgd_flat(float* p, float* momentum_buf, void* p_bf16, const void* const* grads, int ntensors,
                         const int32_t* blk_tab, const int32_t* tensor_first_block, float lr, const float* lr_dev,
                         float momentum, float weight_decay, void* stream) {
    if (!p || !momentum_buf || !grads || !blk_tab || !tensor_first_block) return CP2_ERR_NULL;
    if (ntensors <= 0) return CP2_ERR_SHAPE;
    if (!cp2_aligned16(p) || !cp2_aligned16(momentum_buf) || !cp2_aligned16(blk_tab) ||
        (p_bf16 && (reinterpret_cast<uintptr_t>(p_bf16) & 7u)))
        return CP2_ERR_ALIGN;
    hipStream_t s = cp2_stream(stream);
    for (int t0 = 0; t0 < ntensors; t0 += CP2_SGD_MAX_TENSORS) {
        const int t1 = t0 + CP2_SGD_MAX_TENSORS < ntensors ? t0 + CP2_SGD_MAX_TENSORS : ntensors;
        SgdArgs a;
        a.p = p; a.buf = momentum_buf; a.p_bf16 = static_cast<unsigned short*>(p_bf16);
        a.blk_tab = reinterpret_cast<const int4*>(blk_tab);
        a.blk0 = tensor_first_block[t0]; a.t0 = t0;
        a.lr = lr; a.momentum = momentum; a.wd = weight_decay; a.lr_dev = lr_dev;
        bool any = false;
        for (int t = t0; t < t1; ++t) { a.grads[t - t0] = static_cast<const float*>(grads[t]); any |= grads[t] != nullptr; }
        for (int t = t1 - t0; t < CP2_SGD_MAX_TENSORS; ++t) a.grads[t] = nullptr;
        const int nblk = tensor_first_block[t1] - tensor_first_block[t0];
        if (nblk < 0) return CP2_ERR_SHAPE;
        if (!any || nblk == 0) continue;
        CP2_LAUNCH_PROFILED(sgd_flat_kernel, dim3((unsigned)nblk), dim3(kSgdThreads), 0, s, a);
    }
    return cp2_launch_status();
}

// Gradient packing for the averaging over ranks (reference main.py:456-460 wraps the model in DistributedDataParallel,
// whose reducer copies every parameter's gradient into a bucket with one small kernel each -- 161 launches per step on
// ResNet-50).  Here the gradients of a contiguous range of parameter tensors go into their slots of ONE flat gradient
// buffer (the layout of the flat parameter buffer) in a single launch, already scaled by 1 / world size as DDP's bucket
// copies are (g * float(1/W) before the sum; exact for the power-of-two world sizes of a node), and the flat range is then
// all-reduced in place.  Same block table as cp2_sgd_flat.  A tensor without gradient: slot zeroed.
// HBM-bound: 8 B per parameter (read g, write flat).
struct PackArgs {
    float* flat;
    const int4* blk_tab;
    int blk0, t0;
    float scale;
    const float* grads[CP2_SGD_MAX_TENSORS];
};

__global__ __launch_bounds__(kSgdThreads) void pack_grads_kernel(PackArgs a) {
    const int4 e = a.blk_tab[a.blk0 + blockIdx.x];
    const float* __restrict__ g = a.grads[e.x - a.t0];
    const int i = threadIdx.x * 4;
    if (i >= e.w) return;
    float* d = a.flat + e.y + i;
    if (g == nullptr) {
        for (int j = 0; j < min(4, e.w - i); ++j) d[j] = 0.f;
        return;
    }
    g += e.z + i;
    if (g == d && a.scale == 1.f) return;              // the gradient already lives in its slot
    if (i + 4 <= e.w && (reinterpret_cast<uintptr_t>(g) & 15u) == 0) {
        sgd_f4 v = __builtin_nontemporal_load(reinterpret_cast<const sgd_f4*>(g));
        if (a.scale != 1.f) { v.x = __fmul_rn(v.x, a.scale); v.y = __fmul_rn(v.y, a.scale); v.z = __fmul_rn(v.z, a.scale); v.w = __fmul_rn(v.w, a.scale); }
        *reinterpret_cast<sgd_f4*>(d) = v;
    } else {
        for (int j = 0; j < min(4, e.w - i); ++j) d[j] = a.scale != 1.f ? __fmul_rn(g[j], a.scale) : g[j];
    }
}

CP2_API int cp2_pack_grads(float* flat_grad, const void* const* grads, int t_begin, int t_end, const int32_t* blk_tab,
                           const int32_t* tensor_first_block, float scale, void* stream) {
    if (!flat_grad || !grads || !blk_tab || !tensor_first_block) return CP2_ERR_NULL;
    if (t_begin < 0 || t_end <= t_begin) return CP2_ERR_SHAPE;
    if (!cp2_aligned16(flat_grad) || !cp2_aligned16(blk_tab)) return CP2_ERR_ALIGN;
    hipStream_t s = cp2_stream(stream);
    for (int t0 = t_begin; t0 < t_end; t0 += CP2_SGD_MAX_TENSORS) {
        const int t1 = t0 + CP2_SGD_MAX_TENSORS < t_end ? t0 + CP2_SGD_MAX_TENSORS : t_end;
        PackArgs a;
        a.flat = flat_grad; a.blk_tab = reinterpret_cast<const int4*>(blk_tab);
        a.blk0 = tensor_first_block[t0]; a.t0 = t0; a.scale = scale;
        for (int t = t0; t < t1; ++t) a.grads[t - t0] = static_cast<const float*>(grads[t]);
        for (int t = t1 - t0; t < CP2_SGD_MAX_TENSORS; ++t) a.grads[t] = nullptr;
        const int nblk = tensor_first_block[t1] - tensor_first_block[t0];
        if (nblk < 0) return CP2_ERR_SHAPE;
        if (nblk == 0) continue;
        hipLaunchKernelGGL(pack_grads_kernel, dim3((unsigned)nblk), dim3(kSgdThreads), 0, s, a);
    }
    return cp2_launch_status();
}
