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
