// a1/a2/a12: copy-paste composition, centre-tap strided gathers, row gather.
// All HBM-bound streaming kernels: 16-byte accesses per lane where the shape allows.
#include "common.hpp"

// ---------------------------------------------------------------------------
// compose: mask = (bg[:,0]==0); out = img*mask + bg  (reference builder.py:1146-1152)
// One thread = VEC consecutive pixels of one image row, all three channels.
// The product and the sum are rounded separately (__fmul_rn/__fadd_rn) so the
// result is bit-identical to torch's `img * mask + bg`.
// ---------------------------------------------------------------------------
template <int VEC>
__global__ __launch_bounds__(256) void compose_mask_kernel(const float* __restrict__ img,
                                                           const float* __restrict__ bg,
                                                           float* __restrict__ out,
                                                           float* __restrict__ mask_full,
                                                           float* __restrict__ mask_ds, int B, int H, int W,
                                                           int stride, int Hs, int Ws) {
    const int64_t plane = (int64_t)H * W;
    const int wv = W / VEC;
    const int64_t total = (int64_t)B * H * wv;
    const int off = stride >> 1;
    for (int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; t < total;
         t += (int64_t)gridDim.x * blockDim.x) {
        const int xg = (int)(t % wv);
        const int y = (int)((t / wv) % H);
        const int b = (int)(t / ((int64_t)wv * H));
        const int x0 = xg * VEC;
        const int64_t base = (int64_t)b * 3 * plane + (int64_t)y * W + x0;
        float bgv[3][VEC], iv[3][VEC], m[VEC];
        if constexpr (VEC == 4) {
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                const float4 a = *reinterpret_cast<const float4*>(bg + base + c * plane);
                const float4 i4 = *reinterpret_cast<const float4*>(img + base + c * plane);
                bgv[c][0] = a.x; bgv[c][1] = a.y; bgv[c][2] = a.z; bgv[c][3] = a.w;
                iv[c][0] = i4.x; iv[c][1] = i4.y; iv[c][2] = i4.z; iv[c][3] = i4.w;
            }
        } else {
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                bgv[c][0] = bg[base + c * plane];
                iv[c][0] = img[base + c * plane];
            }
        }
#pragma unroll
        for (int v = 0; v < VEC; ++v) m[v] = (bgv[0][v] == 0.0f) ? 1.0f : 0.0f;
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            float o[VEC];
#pragma unroll
            for (int v = 0; v < VEC; ++v) o[v] = __fadd_rn(__fmul_rn(iv[c][v], m[v]), bgv[c][v]);
            if constexpr (VEC == 4)
                *reinterpret_cast<float4*>(out + base + c * plane) = make_float4(o[0], o[1], o[2], o[3]);
            else
                out[base + c * plane] = o[0];
        }
        if (mask_full) {
            const int64_t mo = (int64_t)b * plane + (int64_t)y * W + x0;
            if constexpr (VEC == 4)
                *reinterpret_cast<float4*>(mask_full + mo) = make_float4(m[0], m[1], m[2], m[3]);
            else
                mask_full[mo] = m[0];
        }
        if (mask_ds && y >= off && (y - off) % stride == 0) {
            const int ys = (y - off) / stride;
#pragma unroll
            for (int v = 0; v < VEC; ++v) {
                const int x = x0 + v;
                if (x >= off && (x - off) % stride == 0)
                    mask_ds[((int64_t)b * Hs + ys) * Ws + (x - off) / stride] = m[v];
            }
        }
    }
}

static inline int ds_size(int n, int s) { return (n - s / 2 + s - 1) / s; }

CP2_API int cp2_compose_mask(const float* img, const float* bg, float* out_img, float* mask_full,
                             float* mask_ds, int B, int H, int W, int stride, void* stream) {
    if (!img || !bg || !out_img) return CP2_ERR_NULL;
    if (B <= 0 || H <= 0 || W <= 0 || (mask_ds && stride <= 0)) return CP2_ERR_SHAPE;
    if (stride <= 0) stride = 1;
    const int Hs = ds_size(H, stride), Ws = ds_size(W, stride);
    const bool vec = (W % 4 == 0) && cp2_aligned16(img) && cp2_aligned16(bg) && cp2_aligned16(out_img) &&
                     (!mask_full || cp2_aligned16(mask_full));
    const int64_t total = (int64_t)B * H * (vec ? W / 4 : W);
    int blocks = cp2_cdiv(total, 256);
    if (blocks > 8192) blocks = 8192;
    if (vec)
        hipLaunchKernelGGL(compose_mask_kernel<4>, dim3(blocks), dim3(256), 0, cp2_stream(stream), img, bg,
                           out_img, mask_full, mask_ds, B, H, W, stride, Hs, Ws);
    else
        hipLaunchKernelGGL(compose_mask_kernel<1>, dim3(blocks), dim3(256), 0, cp2_stream(stream), img, bg,
                           out_img, mask_full, mask_ds, B, H, W, stride, Hs, Ws);
    return cp2_launch_status();
}

// ---------------------------------------------------------------------------
// compose_pair: both views of the step in ONE launch (reference builder.py:1146-1159 runs the composition once per
// view), with what the training step did in separate launches afterwards folded in:
//   * the key view's rows can be written in shuffle-BN order (out_b[j] = compose(img_b[row_b[j]], bg1[row_b[j]]),
//     reference builder.py:630 applies the same permutation as a gather after the composition) -- the gather launch and
//     its 2 x 19 MB of traffic disappear; the down-sampled mask stays in the ORIGINAL row order (the loss needs it there);
//   * the output can be written channels-last and / or in bf16 -- the layout copy and the autocast cast kernel in front
//     of the stem convolution disappear; bf16 is the round-to-nearest-even of the very fp32 value the fp32 path stores.
// One thread = 4 consecutive pixels of one row (W % 4 == 0), all three channels.
// ---------------------------------------------------------------------------
struct ComposePairArgs {
    const float* img[2]; const float* bg[2];
    void* out[2];                    // [B,3,H,W] logical; memory NCHW or NHWC, fp32 or bf16
    float* mask_ds[2];               // [B,Hs,Ws] each (may be NULL)
    const int64_t* row_b;            // device int64 [B] or NULL: source row of output row j of view b
    int B, H, W, stride, Hs, Ws, nhwc, bf16;
};

__device__ __forceinline__ unsigned short f32_to_bf16_rne(float f) {
    uint32_t u = __float_as_uint(f);
    if ((u & 0x7fffffffu) > 0x7f800000u) return (unsigned short)((u >> 16) | 0x40);     // NaN stays NaN (quiet)
    u += 0x7fffu + ((u >> 16) & 1u);
    return (unsigned short)(u >> 16);
}

__global__ __launch_bounds__(256) void compose_pair_kernel(ComposePairArgs a) {
    const int64_t plane = (int64_t)a.H * a.W;
    const int wv = a.W / 4;
    const int64_t per_view = (int64_t)a.B * a.H * wv;
    const int off = a.stride >> 1;
    for (int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; t < 2 * per_view; t += (int64_t)gridDim.x * blockDim.x) {
        const int view = t >= per_view;
        const int64_t tt = view ? t - per_view : t;
        const int xg = (int)(tt % wv), y = (int)((tt / wv) % a.H), j = (int)(tt / ((int64_t)wv * a.H));
        const int x0 = xg * 4;
        int b = j;                                               // source row
        if (view && a.row_b) {
            const int64_t r = a.row_b[j];
            b = (r < 0 || r >= a.B) ? j : (int)r;                // a bad index cannot fault; tests check the permutation
        }
        const int64_t base = (int64_t)b * 3 * plane + (int64_t)y * a.W + x0;
        float bgv[3][4], o[3][4], m[4];
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            const float4 g = *reinterpret_cast<const float4*>(a.bg[view] + base + c * plane);
            bgv[c][0] = g.x, bgv[c][1] = g.y, bgv[c][2] = g.z, bgv[c][3] = g.w;
        }
#pragma unroll
        for (int v = 0; v < 4; ++v) m[v] = (bgv[0][v] == 0.0f) ? 1.0f : 0.0f;
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            const float4 i4 = *reinterpret_cast<const float4*>(a.img[view] + base + c * plane);
            o[c][0] = __fadd_rn(__fmul_rn(i4.x, m[0]), bgv[c][0]), o[c][1] = __fadd_rn(__fmul_rn(i4.y, m[1]), bgv[c][1]);
            o[c][2] = __fadd_rn(__fmul_rn(i4.z, m[2]), bgv[c][2]), o[c][3] = __fadd_rn(__fmul_rn(i4.w, m[3]), bgv[c][3]);
        }
        const int64_t obase = (int64_t)j * 3 * plane;            // output row j
        if (!a.nhwc) {
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                const int64_t e = obase + c * plane + (int64_t)y * a.W + x0;
                if (a.bf16) {
                    ushort4 q{f32_to_bf16_rne(o[c][0]), f32_to_bf16_rne(o[c][1]), f32_to_bf16_rne(o[c][2]), f32_to_bf16_rne(o[c][3])};
                    *reinterpret_cast<ushort4*>(static_cast<unsigned short*>(a.out[view]) + e) = q;
                } else {
                    *reinterpret_cast<float4*>(static_cast<float*>(a.out[view]) + e) = make_float4(o[c][0], o[c][1], o[c][2], o[c][3]);
                }
            }
        } else {
            const int64_t e = obase + ((int64_t)y * a.W + x0) * 3;   // 12 consecutive values: pixel-major, channel-minor
            float lin[12];
#pragma unroll
            for (int v = 0; v < 4; ++v)
#pragma unroll
                for (int c = 0; c < 3; ++c) lin[v * 3 + c] = o[c][v];
            if (a.bf16) {
                unsigned short* d = static_cast<unsigned short*>(a.out[view]) + e;       // 24 bytes, 8-byte aligned
#pragma unroll
                for (int k = 0; k < 3; ++k)
                    *reinterpret_cast<ushort4*>(d + 4 * k) = ushort4{f32_to_bf16_rne(lin[4 * k]), f32_to_bf16_rne(lin[4 * k + 1]),
                                                                      f32_to_bf16_rne(lin[4 * k + 2]), f32_to_bf16_rne(lin[4 * k + 3])};
            } else {
                float* d = static_cast<float*>(a.out[view]) + e;                         // 48 bytes, 16-byte aligned
#pragma unroll
                for (int k = 0; k < 3; ++k) *reinterpret_cast<float4*>(d + 4 * k) = make_float4(lin[4 * k], lin[4 * k + 1], lin[4 * k + 2], lin[4 * k + 3]);
            }
        }
        float* md = a.mask_ds[view];
        if (md && y >= off && (y - off) % a.stride == 0) {
            const int ys = (y - off) / a.stride;
#pragma unroll
            for (int v = 0; v < 4; ++v) {
                const int x = x0 + v;
                if (x >= off && (x - off) % a.stride == 0) md[((int64_t)b * a.Hs + ys) * a.Ws + (x - off) / a.stride] = m[v];
            }
        }
    }
}

CP2_API int cp2_compose_pair(const float* img_a, const float* bg0, const float* img_b, const float* bg1, void* out_a, void* out_b,
                             float* mask_ds_a, float* mask_ds_b, const int64_t* row_b, int B, int H, int W, int stride,
                             int channels_last, int out_bf16, void* stream) {
    if (!img_a || !bg0 || !img_b || !bg1 || !out_a || !out_b) return CP2_ERR_NULL;
    if (B <= 0 || H <= 0 || W <= 0 || ((mask_ds_a || mask_ds_b) && stride <= 0)) return CP2_ERR_SHAPE;
    if (W % 4) return CP2_ERR_UNSUPPORTED;                        // the per-view entry cp2_compose_mask takes any width
    if (!cp2_aligned16(img_a) || !cp2_aligned16(bg0) || !cp2_aligned16(img_b) || !cp2_aligned16(bg1) || !cp2_aligned16(out_a) ||
        !cp2_aligned16(out_b))
        return CP2_ERR_ALIGN;
    if (stride <= 0) stride = 1;
    ComposePairArgs a{{img_a, img_b}, {bg0, bg1}, {out_a, out_b}, {mask_ds_a, mask_ds_b}, row_b, B, H, W, stride,
                      ds_size(H, stride), ds_size(W, stride), channels_last != 0, out_bf16 != 0};
    int blocks = cp2_cdiv((int64_t)2 * B * H * (W / 4), 256);
    if (blocks > 16384) blocks = 16384;
    CP2_LAUNCH_PROFILED(compose_pair_kernel, dim3(blocks), dim3(256), 0, cp2_stream(stream), a);
    return cp2_launch_status();
}

// ---------------------------------------------------------------------------
// strided gather: y[b,i,j] = x[b, s/2+s*i, s/2+s*j]   (builder.py:1155-1186)
// ---------------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(256) void strided_gather_kernel(const T* __restrict__ x, T* __restrict__ y, int B,
                                                             int H, int W, int stride, int Hs, int Ws) {
    const int64_t total = (int64_t)B * Hs * Ws;
    const int off = stride >> 1;
    for (int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; t < total;
         t += (int64_t)gridDim.x * blockDim.x) {
        const int j = (int)(t % Ws);
        const int i = (int)((t / Ws) % Hs);
        const int b = (int)(t / ((int64_t)Ws * Hs));
        y[t] = x[((int64_t)b * H + off + (int64_t)stride * i) * W + off + (int64_t)stride * j];
    }
}

template <typename T>
static int strided_gather_launch(const T* x, T* y, int B, int H, int W, int stride, void* stream) {
    if (!x || !y) return CP2_ERR_NULL;
    if (B <= 0 || H <= 0 || W <= 0 || stride <= 0) return CP2_ERR_SHAPE;
    const int Hs = ds_size(H, stride), Ws = ds_size(W, stride);
    if (Hs <= 0 || Ws <= 0) return CP2_ERR_SHAPE;
    int blocks = cp2_cdiv((int64_t)B * Hs * Ws, 256);
    if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(strided_gather_kernel<T>, dim3(blocks), dim3(256), 0, cp2_stream(stream), x, y, B, H, W,
                       stride, Hs, Ws);
    return cp2_launch_status();
}

CP2_API int cp2_strided_gather_f32(const float* x, float* y, int B, int H, int W, int stride, void* stream) {
    return strided_gather_launch<float>(x, y, B, H, W, stride, stream);
}
CP2_API int cp2_strided_gather_i64(const int64_t* x, int64_t* y, int B, int H, int W, int stride,
                                   void* stream) {
    return strided_gather_launch<int64_t>(x, y, B, H, W, stride, stream);
}

// ---------------------------------------------------------------------------
// row gather (shuffle-BN take): dst[r,:] = src[idx[r],:]     (builder.py:630,649)
// grid = (column chunks, rows); each lane moves 16 bytes per step.
// ---------------------------------------------------------------------------
template <int VEC>
__global__ __launch_bounds__(256) void gather_rows_kernel(const float* __restrict__ src,
                                                          const int64_t* __restrict__ idx,
                                                          float* __restrict__ dst, int n_src,
                                                          int64_t row_elems, int32_t* err_flag) {
    const int r = blockIdx.y;
    const int64_t s = idx[r];
    if (s < 0 || s >= n_src) {
        if (err_flag && threadIdx.x == 0 && blockIdx.x == 0) atomicOr(err_flag, 1);
        return;
    }
    const float* sp = src + s * row_elems;
    float* dp = dst + (int64_t)r * row_elems;
    const int64_t nv = row_elems / VEC;
    for (int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; t < nv;
         t += (int64_t)gridDim.x * blockDim.x) {
        if constexpr (VEC == 4)
            reinterpret_cast<float4*>(dp)[t] = reinterpret_cast<const float4*>(sp)[t];
        else
            dp[t] = sp[t];
    }
}

CP2_API int cp2_gather_rows_f32(const float* src, const int64_t* idx, float* dst, int rows, int n_src,
                                int64_t row_elems, int32_t* err_flag, void* stream) {
    if (!src || !idx || !dst) return CP2_ERR_NULL;
    if (rows <= 0 || n_src <= 0 || row_elems <= 0) return CP2_ERR_SHAPE;
    const bool vec = (row_elems % 4 == 0) && cp2_aligned16(src) && cp2_aligned16(dst);
    const int64_t nv = vec ? row_elems / 4 : row_elems;
    int chunks = cp2_cdiv(nv, 256 * 4);
    if (chunks < 1) chunks = 1;
    if (chunks > 256) chunks = 256;
    if (vec)
        hipLaunchKernelGGL(gather_rows_kernel<4>, dim3(chunks, rows), dim3(256), 0, cp2_stream(stream), src, idx,
                           dst, n_src, row_elems, err_flag);
    else
        hipLaunchKernelGGL(gather_rows_kernel<1>, dim3(chunks, rows), dim3(256), 0, cp2_stream(stream), src, idx,
                           dst, n_src, row_elems, err_flag);
    return cp2_launch_status();
}
