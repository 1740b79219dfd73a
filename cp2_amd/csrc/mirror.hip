// SURVEY 8f rank 4 (second half): the supervised CutPaste / "mirror" pre-training path on the device.
//   datasets/pretrain_dataset.py:273-352   CutPasteDataset.cutpaste: cut a patch, rotate it (Pillow nearest neighbour,
//                                          expanded canvas), paste it through its shape mask onto the image AND onto
//                                          the mirror image, label the pasted shape with the patch class
//   datasets/pretrain_dataset.py:357-412   additional patches (masks OR-ed), ToTensor (uint8 HWC -> float CHW / 255)
//   networks/mirror_network.py:40-63       class cross entropy over cat(s, t) + lambda * cross entropy between the
//                                          tempered softmaxes of the two views (probability targets), argmax
// The reference does the composition per sample on the CPU with numpy / Pillow and the loss as ~10 ATen passes over
// [2N, C, H, W].  Here: ONE pass over the image per patch round (byte gathers for the patch, served by L2) and ONE
// pass over the logits that produces both losses' partial sums, both gradients, the argmax map and the confusion
// counts.  HBM-bound integer / fp32 elementwise work, no MFMA.  The random patch parameters are drawn on the host in
// the reference's order (cp2_amd/mirror.py), the rotation arrives as Pillow's 16.16 fixed-point reverse matrix, so
// images and masks are bit-exact against the reference's output.
#include "common.hpp"
#include <math.h>

struct CutPasteArgs {
    const unsigned char* src;          // [*][H][W][3]
    const unsigned char* src_mirror;   // [*][H][W][3] or NULL
    const int32_t* params;             // [B][CP2_CUTPASTE_PARAMS]
    unsigned char* dst; unsigned char* dst_mirror;   // [B][H][W][3] (either may be NULL)
    float* dst_f32; float* dst_mirror_f32;           // [B][3][H][W] (either may be NULL)
    int64_t* mask;                     // [B][H][W]
    int mask_or, B, H, W;
};

// one thread = VEC consecutive pixels of one row
template <int VEC>
__global__ __launch_bounds__(256) void cutpaste_kernel(CutPasteArgs a) {
    const int x0 = (blockIdx.x * 256 + threadIdx.x) * VEC, y = blockIdx.y, b = blockIdx.z;
    if (x0 >= a.W) return;
    const int32_t* p = a.params + b * CP2_CUTPASTE_PARAMS;
    const int si = p[0], mi = p[1], cls = p[2], px = p[3], py = p[4], pw = p[5], ph = p[6], xp = p[7], yp = p[8], rw = p[9],
              rh = p[10];
    const int a0 = p[11], a1 = p[12], a2 = p[13], a3 = p[14], a4 = p[15], a5 = p[16];
    const int64_t HW = (int64_t)a.H * a.W;
    const unsigned char* img = a.src + (int64_t)si * HW * 3;
    const unsigned char* mir = a.src_mirror ? a.src_mirror + (int64_t)mi * HW * 3 : nullptr;
    const int64_t o = (int64_t)y * a.W + x0;
    unsigned char vi[VEC * 3], vm[VEC * 3];
    if constexpr (VEC == 4) {                        // 12 bytes = three aligned dwords (W % 4 == 0)
        const uint32_t* q = reinterpret_cast<const uint32_t*>(img + o * 3);
        uint32_t w3[3] = {q[0], q[1], q[2]};
        __builtin_memcpy(vi, w3, 12);
        if (mir) {
            const uint32_t* qm = reinterpret_cast<const uint32_t*>(mir + o * 3);
            uint32_t m3[3] = {qm[0], qm[1], qm[2]};
            __builtin_memcpy(vm, m3, 12);
        }
    } else {
#pragma unroll
        for (int e = 0; e < 3; ++e) { vi[e] = img[o * 3 + e]; if (mir) vm[e] = mir[o * 3 + e]; }
    }
    int64_t mk[VEC];
    const bool row_in = cls != 0 && y >= yp && y < yp + rh;
#pragma unroll
    for (int u = 0; u < VEC; ++u) {
        const int x = x0 + u;
        bool hit = false;
        if (row_in && x >= xp && x < xp + rw) {
            // Pillow affine_fixed: source = (a2 + u*a0 + v*a1) >> 16 in int arithmetic, floor for negatives
            const int uu = x - xp, vv = y - yp;
            const int xin = (a2 + uu * a0 + vv * a1) >> 16, yin = (a5 + uu * a3 + vv * a4) >> 16;
            if (xin >= 0 && xin < pw && yin >= 0 && yin < ph) {
                hit = true;
                const unsigned char* s = img + ((int64_t)(py + yin) * a.W + (px + xin)) * 3;
#pragma unroll
                for (int e = 0; e < 3; ++e) { vi[u * 3 + e] = s[e]; vm[u * 3 + e] = s[e]; }
            }
        }
        mk[u] = hit ? (int64_t)cls : 0;
    }
    const int64_t ob = (int64_t)b * HW + o;
    if (a.mask_or) {
#pragma unroll
        for (int u = 0; u < VEC; ++u) {
            const int64_t old = a.mask[ob + u];
            mk[u] = cls == 0 ? old : (int64_t)((mk[u] != 0) || (old != 0));   // np.logical_or(mask, old_mask).long()
        }
    }
#pragma unroll
    for (int u = 0; u < VEC; ++u) a.mask[ob + u] = mk[u];
    auto put_u8 = [&](unsigned char* dst, const unsigned char* v) {
        if (VEC == 4) {
            uint32_t w3[3];
            __builtin_memcpy(w3, v, 12);
            uint32_t* q = reinterpret_cast<uint32_t*>(dst + ob * 3);
            q[0] = w3[0]; q[1] = w3[1]; q[2] = w3[2];
        } else {
#pragma unroll
            for (int e = 0; e < 3; ++e) dst[ob * 3 + e] = v[e];
        }
    };
    auto put_f32 = [&](float* dst, const unsigned char* v) {       // ToTensor: float(u8) / 255 (IEEE division)
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            float* pl = dst + ((int64_t)b * 3 + c) * HW + o;
            if (VEC == 4) {
                float4 f;
                f.x = __fdiv_rn((float)v[0 + c], 255.0f); f.y = __fdiv_rn((float)v[3 + c], 255.0f);
                f.z = __fdiv_rn((float)v[6 + c], 255.0f); f.w = __fdiv_rn((float)v[9 + c], 255.0f);
                *reinterpret_cast<float4*>(pl) = f;
            } else {
                pl[0] = __fdiv_rn((float)v[c], 255.0f);
            }
        }
    };
    if (a.dst) put_u8(a.dst, vi);
    if (a.dst_f32) put_f32(a.dst_f32, vi);
    if (mir) {
        if (a.dst_mirror) put_u8(a.dst_mirror, vm);
        if (a.dst_mirror_f32) put_f32(a.dst_mirror_f32, vm);
    }
}

CP2_API int cp2_cutpaste(const unsigned char* src, const unsigned char* src_mirror, const int32_t* params,
                         unsigned char* dst, unsigned char* dst_mirror, float* dst_f32, float* dst_mirror_f32,
                         int64_t* mask, int mask_or, int B, int H, int W, void* stream) {
    if (!src || !params || !mask) return CP2_ERR_NULL;
    if (B <= 0 || H <= 0 || W <= 0) return CP2_ERR_SHAPE;
    if (B > 65535 || H > 65535 || H >= 32768 || W >= 32768) return CP2_ERR_UNSUPPORTED;   // 16.16 range of the matrix
    if (dst == src || (src_mirror && dst_mirror == src_mirror)) return CP2_ERR_UNSUPPORTED;  // patches are read from src
    CutPasteArgs a{src, src_mirror, params, dst, dst_mirror, dst_f32, dst_mirror_f32, mask, mask_or, B, H, W};
    const bool vec = (W % 4 == 0) && ((reinterpret_cast<uintptr_t>(src) & 3u) == 0) &&
                     (!src_mirror || (reinterpret_cast<uintptr_t>(src_mirror) & 3u) == 0) &&
                     (!dst || (reinterpret_cast<uintptr_t>(dst) & 3u) == 0) &&
                     (!dst_mirror || (reinterpret_cast<uintptr_t>(dst_mirror) & 3u) == 0) &&
                     (!dst_f32 || cp2_aligned16(dst_f32)) && (!dst_mirror_f32 || cp2_aligned16(dst_mirror_f32));
    if (vec)
        hipLaunchKernelGGL(cutpaste_kernel<4>, dim3(cp2_cdiv(W, 1024), H, B), dim3(256), 0, cp2_stream(stream), a);
    else
        hipLaunchKernelGGL(cutpaste_kernel<1>, dim3(cp2_cdiv(W, 256), H, B), dim3(256), 0, cp2_stream(stream), a);
    return cp2_launch_status();
}

// ------------------------------------------------------------------------------------------------ mirror loss
struct MirrorLossArgs {
    const float* s; const float* t;          // [N][C][HW] logits at image size (t NULL: MirrorVariant.NONE)
    const int64_t* masks;                    // [N][HW]
    float temp, lmbd;
    float* grad_s; float* grad_t;            // NULL: forward only
    int64_t* argmax;                         // [(t ? 2 : 1) * N][HW] or NULL
    unsigned long long* confusion;           // [C][C] (row = ground truth) or NULL; caller zeroes it
    double* partial;                         // [gridDim.x * gridDim.y][2] class sum, compare sum
    int N, C; int64_t HW;
};

constexpr int ML_T = 256;
constexpr int ML_CMAX = CP2_MIRROR_MAX_CLASSES;

// log-softmax pieces of one pixel's C logits: returns max and log(sum(exp(x - max)))
template <int C>
__device__ __forceinline__ void lse_of(const float (&x)[ML_CMAX], float& mx, float& ls) {
    mx = x[0];
#pragma unroll
    for (int c = 1; c < C; ++c) mx = fmaxf(mx, x[c]);
    float s = 0.f;
#pragma unroll
    for (int c = 0; c < C; ++c) s += expf(x[c] - mx);
    ls = logf(s);
}

// class cross entropy of one pixel: adds -log_softmax(x)[lab] to acc, g = (softmax(x) - onehot) / m, returns the argmax
// (first maximum, as torch.argmax)
template <int C>
__device__ __forceinline__ int class_ce(const float (&x)[ML_CMAX], int lab, float m, float (&g)[ML_CMAX], float& acc) {
    float mx, ls;
    lse_of<C>(x, mx, ls);
    int am = 0;
    float best = x[0];
#pragma unroll
    for (int c = 0; c < C; ++c) {
        const float lsm = (x[c] - mx) - ls;
        if (c == lab) acc += -lsm;
        g[c] = (expf(lsm) - (c == lab ? 1.f : 0.f)) / m;
        if (x[c] > best) { best = x[c]; am = c; }
    }
    return am;
}

// Everything one pixel contributes: both class cross entropies, the compare loss, both gradients, both arg-maxes.
template <int C>
__device__ __forceinline__ void mirror_pixel(const float (&xs)[ML_CMAX], const float (&xt)[ML_CMAX], int lab, bool two, bool want_grad,
                                             float temp, float lmbd, float m_cls, float m_cmp, float (&gs)[ML_CMAX],
                                             float (&gt)[ML_CMAX], int& am_s, int& am_t, float& acc_cls, float& acc_cmp) {
    // ---- class cross entropy (nn.CrossEntropyLoss over cat(s, t) with cat(masks, masks)), mean over all pixels
    am_s = class_ce<C>(xs, lab, m_cls, gs, acc_cls);
    am_t = 0;
    if (!two) return;
    am_t = class_ce<C>(xt, lab, m_cls, gt, acc_cls);
    // ---- compare loss: cross_entropy(softmax(s / T), softmax(t / T)) with probability targets, mean over N*HW
    float zs[ML_CMAX], zt[ML_CMAX], ps[ML_CMAX], pt[ML_CMAX], u[ML_CMAX];
#pragma unroll
    for (int c = 0; c < C; ++c) { zs[c] = xs[c] / temp; zt[c] = xt[c] / temp; }
    float mx, ls, sum;
    lse_of<C>(zs, mx, ls);
    sum = expf(ls);
#pragma unroll
    for (int c = 0; c < C; ++c) ps[c] = expf(zs[c] - mx) / sum;
    lse_of<C>(zt, mx, ls);
    sum = expf(ls);
#pragma unroll
    for (int c = 0; c < C; ++c) pt[c] = expf(zt[c] - mx) / sum;
    lse_of<C>(ps, mx, ls);
    float l = 0.f, spt = 0.f;
#pragma unroll
    for (int c = 0; c < C; ++c) { u[c] = (ps[c] - mx) - ls; l += pt[c] * u[c]; spt += pt[c]; }
    acc_cmp += -l;
    if (want_grad) {
        // d/du = -pt/M;  log_softmax backward: g_ps = g_u - exp(u) * sum(g_u);  softmax backward, then / T
        float gps[ML_CMAX], gpt[ML_CMAX], dot_s = 0.f, dot_t = 0.f;
#pragma unroll
        for (int c = 0; c < C; ++c) {
            gps[c] = (-pt[c] + expf(u[c]) * spt) / m_cmp;
            gpt[c] = -u[c] / m_cmp;
            dot_s += gps[c] * ps[c];
            dot_t += gpt[c] * pt[c];
        }
#pragma unroll
        for (int c = 0; c < C; ++c) {
            gs[c] += lmbd * (ps[c] * (gps[c] - dot_s) / temp);
            gt[c] += lmbd * (pt[c] * (gpt[c] - dot_t) / temp);
        }
    }
}

// VEC = 4: a thread owns four consecutive pixels (16-byte loads and stores per channel plane; HW % 4 == 0, aligned
// planes); VEC = 1: one pixel per thread.
template <int C, int VEC>
__global__ __launch_bounds__(ML_T) void mirror_loss_kernel(MirrorLossArgs a) {
    __shared__ unsigned int conf[ML_CMAX * ML_CMAX];
    __shared__ double red[2][ML_T / 64];
    const int n = blockIdx.y, tid = threadIdx.x;
    if (a.confusion) { for (int i = tid; i < C * C; i += ML_T) conf[i] = 0; __syncthreads(); }
    const bool two = a.t != nullptr, want_grad = a.grad_s != nullptr;
    const float m_cls = (float)((two ? 2.0 : 1.0) * (double)a.N * (double)a.HW), m_cmp = (float)((double)a.N * (double)a.HW);
    float acc_cls = 0.f, acc_cmp = 0.f;
    for (int64_t i = ((int64_t)blockIdx.x * ML_T + tid) * VEC; i < a.HW; i += (int64_t)gridDim.x * ML_T * VEC) {
        const int64_t base = (int64_t)n * C * a.HW + i;
        float xs[VEC][ML_CMAX], xt[VEC][ML_CMAX], gs[VEC][ML_CMAX], gt[VEC][ML_CMAX];
        int lab[VEC], am_s[VEC], am_t[VEC];
        if (VEC == 4) {
#pragma unroll
            for (int c = 0; c < C; ++c) {
                const float4 v = *reinterpret_cast<const float4*>(a.s + base + c * a.HW);
                xs[0][c] = v.x; xs[1 % VEC][c] = v.y; xs[2 % VEC][c] = v.z; xs[3 % VEC][c] = v.w;
                float4 w = make_float4(0.f, 0.f, 0.f, 0.f);
                if (two) w = *reinterpret_cast<const float4*>(a.t + base + c * a.HW);
                xt[0][c] = w.x; xt[1 % VEC][c] = w.y; xt[2 % VEC][c] = w.z; xt[3 % VEC][c] = w.w;
            }
            const longlong2* mp = reinterpret_cast<const longlong2*>(a.masks + (int64_t)n * a.HW + i);
            const longlong2 m0 = mp[0], m1 = mp[1];
            lab[0] = (int)m0.x; lab[1 % VEC] = (int)m0.y; lab[2 % VEC] = (int)m1.x; lab[3 % VEC] = (int)m1.y;
        } else {
#pragma unroll
            for (int c = 0; c < C; ++c) { xs[0][c] = a.s[base + c * a.HW]; xt[0][c] = two ? a.t[base + c * a.HW] : 0.f; }
            lab[0] = (int)a.masks[(int64_t)n * a.HW + i];
        }
#pragma unroll
        for (int v = 0; v < VEC; ++v) {
            mirror_pixel<C>(xs[v], xt[v], lab[v], two, want_grad, a.temp, a.lmbd, m_cls, m_cmp, gs[v], gt[v], am_s[v], am_t[v], acc_cls, acc_cmp);
            if (a.confusion && lab[v] >= 0 && lab[v] < C) {
                atomicAdd(&conf[lab[v] * C + am_s[v]], 1u);
                if (two) atomicAdd(&conf[lab[v] * C + am_t[v]], 1u);
            }
        }
        if (VEC == 4) {
            if (a.argmax) {
                longlong2* o = reinterpret_cast<longlong2*>(a.argmax + (int64_t)n * a.HW + i);
                o[0] = longlong2{am_s[0], am_s[1 % VEC]}; o[1] = longlong2{am_s[2 % VEC], am_s[3 % VEC]};
                if (two) {
                    longlong2* ot = reinterpret_cast<longlong2*>(a.argmax + ((int64_t)(a.N + n)) * a.HW + i);
                    ot[0] = longlong2{am_t[0], am_t[1 % VEC]}; ot[1] = longlong2{am_t[2 % VEC], am_t[3 % VEC]};
                }
            }
            if (want_grad) {
#pragma unroll
                for (int c = 0; c < C; ++c) {
                    *reinterpret_cast<float4*>(a.grad_s + base + c * a.HW) = make_float4(gs[0][c], gs[1 % VEC][c], gs[2 % VEC][c], gs[3 % VEC][c]);
                    if (two) *reinterpret_cast<float4*>(a.grad_t + base + c * a.HW) = make_float4(gt[0][c], gt[1 % VEC][c], gt[2 % VEC][c], gt[3 % VEC][c]);
                }
            }
        } else {
            if (a.argmax) {
                a.argmax[(int64_t)n * a.HW + i] = am_s[0];
                if (two) a.argmax[((int64_t)(a.N + n)) * a.HW + i] = am_t[0];
            }
            if (want_grad) {
#pragma unroll
                for (int c = 0; c < C; ++c) { a.grad_s[base + c * a.HW] = gs[0][c]; if (two) a.grad_t[base + c * a.HW] = gt[0][c]; }
            }
        }
    }
    // per-workgroup partial sums in a fixed order (deterministic); the finalize kernel adds them in a fixed order too
    double dc = (double)wave_sum(acc_cls), dm = (double)wave_sum(acc_cmp);
    if ((tid & 63) == 0) { red[0][tid >> 6] = dc; red[1][tid >> 6] = dm; }
    __syncthreads();
    if (tid == 0) {
        double c0 = 0, c1 = 0;
        for (int w = 0; w < ML_T / 64; ++w) { c0 += red[0][w]; c1 += red[1][w]; }
        double* o = a.partial + 2 * ((int64_t)blockIdx.y * gridDim.x + blockIdx.x);
        o[0] = c0; o[1] = c1;
    }
    if (a.confusion) {
        for (int i = tid; i < C * C; i += ML_T)
            if (conf[i]) atomicAdd(&a.confusion[i], (unsigned long long)conf[i]);
    }
}

// out[0] = loss, out[1] = class_loss, out[2] = compare_loss (fp32, as the reference logs them).  One workgroup: thread
// t adds partials t, t + 256, ... in that order, thread 0 adds the 256 lane sums in lane order -- a fixed tree.
__global__ __launch_bounds__(256) void mirror_loss_finalize_kernel(const double* __restrict__ partial, int nparts, double m_cls,
                                                                   double m_cmp, float lmbd, int two, float* __restrict__ out) {
    __shared__ double r0[256], r1[256];
    double c0 = 0, c1 = 0;
    for (int i = threadIdx.x; i < nparts; i += 256) { c0 += partial[2 * i]; c1 += partial[2 * i + 1]; }
    r0[threadIdx.x] = c0; r1[threadIdx.x] = c1;
    __syncthreads();
    if (threadIdx.x != 0) return;
    c0 = 0; c1 = 0;
    for (int i = 0; i < 256; ++i) { c0 += r0[i]; c1 += r1[i]; }
    const float cls = (float)(c0 / m_cls), cmp = two ? (float)(c1 / m_cmp) : 0.f;
    out[0] = cls + lmbd * cmp;
    out[1] = cls;
    out[2] = cmp;
}

CP2_API int cp2_mirror_loss_num_partials(int N, int64_t HW) {
    if (N <= 0 || HW <= 0) return 0;
    int gx = cp2_cdiv(HW, (int64_t)ML_T * 4);
    const int cap = (2048 + N - 1) / N;                 // ~2048 workgroups in all
    if (gx > cap) gx = cap;
    if (gx < 1) gx = 1;
    return gx * N;
}

CP2_API int cp2_mirror_loss(const float* s_logits, const float* t_logits, const int64_t* masks, float softmax_temp,
                            float lmbd_compare_loss, float* grad_s, float* grad_t, int64_t* argmax, int64_t* confusion,
                            double* partial, float* out3, int N, int C, int64_t HW, void* stream) {
    if (!s_logits || !masks || !partial || !out3) return CP2_ERR_NULL;
    if ((grad_s != nullptr) != (grad_t != nullptr) && t_logits) return CP2_ERR_NULL;
    if (N <= 0 || HW <= 0 || C < 2) return CP2_ERR_SHAPE;
    if (C > ML_CMAX || N > 65535) return CP2_ERR_UNSUPPORTED;
    if (!(softmax_temp > 0.f)) return CP2_ERR_SHAPE;
    const int nparts = cp2_mirror_loss_num_partials(N, HW), gx = nparts / N;
    MirrorLossArgs a{s_logits, t_logits, masks, softmax_temp, lmbd_compare_loss, grad_s, t_logits ? grad_t : nullptr, argmax,
                     reinterpret_cast<unsigned long long*>(confusion), partial, N, C, HW};
    hipStream_t st = cp2_stream(stream);
    const dim3 grid(gx, N), block(ML_T);
    // measured at 10 x 512 x 512 (rocprofv3): the 4-pixel form is 8-12 % SLOWER than one pixel per thread (58 vs 54 us at
    // C = 2, 80 vs 72 at C = 3) -- the kernel is bound by its ~20 expf / logf per pixel, not by the shape of its memory
    // accesses, and the wider form only adds register pressure; it stays compiled for reference and is not selected
    const bool vec = false && C <= 4 && HW % 4 == 0 && cp2_aligned16(s_logits) && (!t_logits || cp2_aligned16(t_logits)) && cp2_aligned16(masks) &&
                     (!grad_s || cp2_aligned16(grad_s)) && (!grad_t || cp2_aligned16(grad_t)) && (!argmax || cp2_aligned16(argmax));
    switch (C) {
#define CP2_ML_LAUNCH(CC, VV) CP2_LAUNCH_PROFILED((mirror_loss_kernel<CC, VV>), grid, block, 0, st, a)
        case 2: if (vec) CP2_ML_LAUNCH(2, 4); else CP2_ML_LAUNCH(2, 1); break;
        case 3: if (vec) CP2_ML_LAUNCH(3, 4); else CP2_ML_LAUNCH(3, 1); break;
        case 4: if (vec) CP2_ML_LAUNCH(4, 4); else CP2_ML_LAUNCH(4, 1); break;
        case 5: CP2_ML_LAUNCH(5, 1); break;
        case 6: CP2_ML_LAUNCH(6, 1); break;
        case 7: CP2_ML_LAUNCH(7, 1); break;
        case 8: CP2_ML_LAUNCH(8, 1); break;
#undef CP2_ML_LAUNCH
        default: return CP2_ERR_UNSUPPORTED;
    }
    int rc = cp2_launch_status();
    if (rc) return rc;
    const double two = t_logits ? 2.0 : 1.0;
    hipLaunchKernelGGL(mirror_loss_finalize_kernel, dim3(1), dim3(256), 0, st, partial, nparts, two * (double)N * (double)HW,
                       (double)N * (double)HW, lmbd_compare_loss, t_logits ? 1 : 0, out3);
    return cp2_launch_status();
}
