// a15: logging statistics that the reference obtains by sorting every step on every rank --
//   tools/correlation_mapping.py:16-53  per-sample nanquantile([.25,.5,.75]) and nanmean of the positive /
//                                       negative dense scores (pairs selected by mask_a[x]*mask_b[y])
//   builder.py:1399-1406                row quantiles of the b x K queue logits
// Exact order statistics without sorting: one workgroup per (row, quantile) runs a 4-pass radix select
// (8 bits per pass, LDS histogram) on the order-preserving integer image of the floats, then interpolates
// exactly as torch.quantile(..., interpolation='linear') does:  rank = q*(n-1) in fp32, lerp(v_lo, v_hi, frac).
#include "common.hpp"
#include <math.h>

struct QuantArgs {
    const float* x; int64_t s_row, s_elem; int N;         // element i of row r at x[r*s_row + i*s_elem]
    const float* mask_a; const float* mask_b; int P; int want;  // want < 0: all elements; else keep i iff (mask_a[r][i/P]*mask_b[r][i%P] != 0) == want
    const float* q; int NQ;
    float* out;                                            // [NQ][R] (torch.quantile layout)
    int R;
};

__device__ __forceinline__ unsigned f2key(float f) {
    const unsigned u = __float_as_uint(f);
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
__device__ __forceinline__ float key2f(unsigned k) {
    return __uint_as_float((k & 0x80000000u) ? (k & 0x7fffffffu) : ~k);
}

__global__ __launch_bounds__(1024) void masked_quantile_kernel(QuantArgs a) {
    __shared__ unsigned hist[256];
    __shared__ unsigned sh_prefix, sh_k, sh_cnt, sh_min;
    const int r = blockIdx.x, qi = blockIdx.y, tid = threadIdx.x;
    const float* row = a.x + (int64_t)r * a.s_row;
    auto keep = [&](int i, float v) -> bool {
        if (v != v) return false;                          // nanquantile ignores NaN
        if (a.want < 0) return true;
        const bool lab = (a.mask_a[(int64_t)r * a.P + i / a.P] * a.mask_b[(int64_t)r * a.P + i % a.P]) != 0.f;
        return lab == (a.want != 0);
    };
    // number of kept elements
    if (tid == 0) sh_cnt = 0;
    __syncthreads();
    unsigned c = 0;
    for (int i = tid; i < a.N; i += blockDim.x) c += keep(i, row[(int64_t)i * a.s_elem]) ? 1u : 0u;
    c = (unsigned)wave_sum_i((int)c);
    if ((tid & 63) == 0 && c) atomicAdd(&sh_cnt, c);
    __syncthreads();
    const unsigned n = sh_cnt;
    if (n == 0) {
        if (tid == 0) a.out[(int64_t)qi * a.R + r] = NAN;
        return;
    }
    const float rank = a.q[qi] * (float)(n - 1);
    const float lo_f = floorf(rank);
    const unsigned lo = (unsigned)lo_f;
    const float w = rank - lo_f;
    // radix select of the element with 0-based rank `lo`
    if (tid == 0) { sh_prefix = 0; sh_k = lo; }
    for (int pass = 3; pass >= 0; --pass) {
        for (int i = tid; i < 256; i += blockDim.x) hist[i] = 0;
        __syncthreads();
        const unsigned prefix = sh_prefix, hi_mask = pass == 3 ? 0u : (0xFFFFFFFFu << (8 * (pass + 1)));
        for (int i = tid; i < a.N; i += blockDim.x) {
            const float v = row[(int64_t)i * a.s_elem];
            if (!keep(i, v)) continue;
            const unsigned k = f2key(v);
            if ((k & hi_mask) == prefix) atomicAdd(&hist[(k >> (8 * pass)) & 255u], 1u);
        }
        __syncthreads();
        if (tid == 0) {
            unsigned kk = sh_k, b = 0;
            for (; b < 256; ++b) {
                if (kk < hist[b]) break;
                kk -= hist[b];
            }
            sh_k = kk;
            sh_prefix = prefix | (b << (8 * pass));
        }
        __syncthreads();
    }
    const unsigned key_lo = sh_prefix;
    const float v_lo = key2f(key_lo);
    float v_hi = v_lo;
    if (w != 0.f) {
        // the next order statistic: v_lo again if it has duplicates reaching rank lo+1, else the smallest larger value
        if (tid == 0) { sh_cnt = 0; sh_min = 0xFFFFFFFFu; }
        __syncthreads();
        unsigned le = 0, mn = 0xFFFFFFFFu;
        for (int i = tid; i < a.N; i += blockDim.x) {
            const float v = row[(int64_t)i * a.s_elem];
            if (!keep(i, v)) continue;
            const unsigned k = f2key(v);
            if (k <= key_lo) ++le; else mn = min(mn, k);
        }
        le = (unsigned)wave_sum_i((int)le);
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) mn = min(mn, (unsigned)__shfl_xor((int)mn, off, 64));
        if ((tid & 63) == 0) { if (le) atomicAdd(&sh_cnt, le); atomicMin(&sh_min, mn); }
        __syncthreads();
        if (sh_cnt <= lo + 1) v_hi = key2f(sh_min);
    }
    if (tid == 0) {
        // at::lerp: weight < 0.5 ? a + w*(b-a) : b - (b-a)*(1-w)
        const float d = v_hi - v_lo;
        a.out[(int64_t)qi * a.R + r] = (w < 0.5f) ? v_lo + w * d : v_hi - d * (1.f - w);
    }
}

CP2_API int cp2_masked_quantiles(const float* x, int64_t stride_row, int64_t stride_elem, int R, int N,
                                 const float* mask_a, const float* mask_b, int P, int want, const float* q, int NQ,
                                 float* out, void* stream) {
    if (!x || !q || !out) return CP2_ERR_NULL;
    if (R <= 0 || N <= 0 || NQ <= 0) return CP2_ERR_SHAPE;
    if (want >= 0 && (!mask_a || !mask_b || P <= 0 || (int64_t)P * P != N)) return CP2_ERR_SHAPE;
    QuantArgs a{x, stride_row, stride_elem, N, mask_a, mask_b, P, want, q, NQ, out, R};
    hipLaunchKernelGGL(masked_quantile_kernel, dim3(R, NQ), dim3(1024), 0, cp2_stream(stream), a);
    return cp2_launch_status();
}
