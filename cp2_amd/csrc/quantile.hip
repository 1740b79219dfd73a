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

constexpr int QT = 1024;      // threads per workgroup
constexpr int QU = 4;         // elements in flight per thread (the passes are load-latency bound otherwise)

// One workgroup per (row, quantile):
//   pass 0        count kept elements
//   passes 1..4   radix select, 8 bits each, LDS histogram
//   pass 5        for the interpolation partner: #elements <= v_lo and the smallest element above it
// Masks are staged in LDS and the (x, y) pixel pair of element i is advanced incrementally (no division per
// element); histogram updates are run-length compressed per thread (cosine logits share their top byte, so the
// first pass would otherwise serialise on two LDS bins); QU independent loads per thread per iteration.
__global__ __launch_bounds__(QT) void masked_quantile_kernel(QuantArgs a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char q_smem[];
    __shared__ unsigned hist[256];
    __shared__ unsigned sh_prefix, sh_k, sh_le, sh_min, sh_cnt;
    float* lma = reinterpret_cast<float*>(q_smem);
    float* lmb = lma + (a.want >= 0 ? a.P : 0);
    const int r = blockIdx.x, qi = blockIdx.y, tid = threadIdx.x;
    const float* row = a.x + (int64_t)r * a.s_row;
    const bool masked = a.want >= 0;
    if (masked) {
        for (int i = tid; i < a.P; i += QT) { lma[i] = a.mask_a[(int64_t)r * a.P + i]; lmb[i] = a.mask_b[(int64_t)r * a.P + i]; }
    }
    if (tid == 0) sh_cnt = 0;
    __syncthreads();
    const int dx = masked ? QT / a.P : 0, dy = masked ? QT % a.P : 0;
    const int x_start = masked ? tid / a.P : 0, y_start = masked ? tid % a.P : 0;
    auto keep_at = [&](int x, int y, float v) -> bool {
        if (v != v) return false;                          // nanquantile ignores NaN
        if (!masked) return true;
        return ((lma[x] * lmb[y]) != 0.f) == (a.want != 0);
    };
#define CP2_Q_FOREACH(BODY)                                                                  \
    {                                                                                        \
        int x_ = x_start, y_ = y_start;                                                      \
        for (int i0 = tid; i0 < a.N; i0 += QU * QT) {                                        \
            float vv[QU];                                                                    \
            _Pragma("unroll") for (int u = 0; u < QU; ++u) {                                 \
                const int i = i0 + u * QT;                                                   \
                vv[u] = i < a.N ? row[(int64_t)i * a.s_elem] : NAN;                          \
            }                                                                                \
            _Pragma("unroll") for (int u = 0; u < QU; ++u) {                                 \
                const float v = vv[u];                                                       \
                if (keep_at(x_, y_, v)) { BODY }                                             \
                if (masked) { x_ += dx; y_ += dy; if (y_ >= a.P) { y_ -= a.P; ++x_; } }      \
            }                                                                                \
        }                                                                                    \
    }
    unsigned c = 0;
    CP2_Q_FOREACH(++c;)
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
    if (tid == 0) { sh_prefix = 0; sh_k = lo; }
    for (int pass = 3; pass >= 0; --pass) {
        for (int i = tid; i < 256; i += QT) hist[i] = 0;
        __syncthreads();
        const unsigned prefix = sh_prefix, hi_mask = pass == 3 ? 0u : (0xFFFFFFFFu << (8 * (pass + 1)));
        unsigned run_bin = 0, run_cnt = 0;
        CP2_Q_FOREACH(
            const unsigned k = f2key(v);
            if ((k & hi_mask) == prefix) {
                const unsigned bin = (k >> (8 * pass)) & 255u;
                if (bin == run_bin) ++run_cnt;
                else { if (run_cnt) atomicAdd(&hist[run_bin], run_cnt); run_bin = bin; run_cnt = 1; }
            })
        if (run_cnt) atomicAdd(&hist[run_bin], run_cnt);
        __syncthreads();
        if (tid < 64) {   // wave 0 finds the bin holding rank sh_k: lane l owns bins 4l..4l+3, prefix sums by shuffles
            const unsigned h0 = hist[4 * tid], h1 = hist[4 * tid + 1], h2 = hist[4 * tid + 2], h3 = hist[4 * tid + 3];
            const unsigned tot = h0 + h1 + h2 + h3;
            unsigned incl = tot;
#pragma unroll
            for (int off = 1; off < 64; off <<= 1) {
                const unsigned t = (unsigned)__shfl_up((int)incl, off, 64);
                if (tid >= off) incl += t;
            }
            const unsigned excl = incl - tot, kk0 = sh_k;
            if (kk0 >= excl && kk0 < incl) {
                unsigned kk = kk0 - excl, bsel = 4 * tid;
                if (kk >= h0) { kk -= h0; ++bsel; if (kk >= h1) { kk -= h1; ++bsel; if (kk >= h2) { kk -= h2; ++bsel; } } }
                sh_k = kk;
                sh_prefix = prefix | (bsel << (8 * pass));
            }
        }
        __syncthreads();
    }
    const unsigned key_lo = sh_prefix;
    const float v_lo = key2f(key_lo);
    float v_hi = v_lo;
    if (w != 0.f) {
        if (tid == 0) { sh_le = 0; sh_min = 0xFFFFFFFFu; }
        __syncthreads();
        unsigned le = 0, mn = 0xFFFFFFFFu;
        CP2_Q_FOREACH(
            const unsigned k = f2key(v);
            if (k <= key_lo) ++le; else mn = min(mn, k);)
        le = (unsigned)wave_sum_i((int)le);
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) mn = min(mn, (unsigned)__shfl_xor((int)mn, off, 64));
        if ((tid & 63) == 0) { if (le) atomicAdd(&sh_le, le); atomicMin(&sh_min, mn); }
        __syncthreads();
        if (sh_le <= lo + 1) v_hi = key2f(sh_min);
    }
    if (tid == 0) {
        const float d = v_hi - v_lo;                         // at::lerp: w < 0.5 ? a + w*(b-a) : b - (b-a)*(1-w)
        a.out[(int64_t)qi * a.R + r] = (w < 0.5f) ? v_lo + w * d : v_hi - d * (1.f - w);
    }
#undef CP2_Q_FOREACH
}

CP2_API int cp2_masked_quantiles(const float* x, int64_t stride_row, int64_t stride_elem, int R, int N,
                                 const float* mask_a, const float* mask_b, int P, int want, const float* q, int NQ,
                                 float* out, void* stream) {
    if (!x || !q || !out) return CP2_ERR_NULL;
    if (R <= 0 || N <= 0 || NQ <= 0) return CP2_ERR_SHAPE;
    if (want >= 0 && (!mask_a || !mask_b || P <= 0 || (int64_t)P * P != N)) return CP2_ERR_SHAPE;
    if (want >= 0 && P > 16384) return CP2_ERR_UNSUPPORTED;
    QuantArgs a{x, stride_row, stride_elem, N, mask_a, mask_b, P, want, q, NQ, out, R};
    const size_t lds = want >= 0 ? 2 * (size_t)P * sizeof(float) : 0;
    hipLaunchKernelGGL(masked_quantile_kernel, dim3(R, NQ), dim3(QT), lds, cp2_stream(stream), a);
    return cp2_launch_status();
}
