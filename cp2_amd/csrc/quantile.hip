// a15: logging statistics that the reference obtains by sorting every step on every rank --
//   tools/correlation_mapping.py:16-53  per-sample nanquantile([.25,.5,.75]) of the positive / negative dense
//                                       scores (pairs selected by mask_a[x]*mask_b[y])
//   builder.py:1399-1406                row quantiles of the b x K queue logits
// Exact order statistics without sorting: ONE workgroup per row finds all requested quantiles together by a
// three-pass radix select (12 + 10 + 10 bits of the order-preserving integer image of the floats, LDS histograms),
// then interpolates exactly as torch.quantile(..., interpolation='linear') does:
//   rank = q*(n-1) in fp32, lerp(v_lo, v_hi, frac).
// The interpolation partner (the next larger element) comes out of the last pass for free: it is the same key again
// (multiplicity), the next non-empty bin of the last histogram, or the smallest key above the 22-bit prefix, which
// that pass tracks with a running minimum.  (Round 1 ran one workgroup per (row, quantile) and six passes each.)
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
constexpr int QMAX = 4;       // quantiles per call
constexpr int QB0 = 4096, QB1 = 1024;

// inclusive prefix sum of one unsigned per thread over the 1024 threads of the workgroup (wtot: 16 words of LDS)
__device__ __forceinline__ unsigned block_scan_incl(unsigned v, unsigned* wtot) {
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    unsigned incl = v;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        const unsigned t = (unsigned)__shfl_up((int)incl, off, 64);
        if (lane >= off) incl += t;
    }
    __syncthreads();                                       // wtot may still be read by the previous scan
    if (lane == 63) wtot[w] = incl;
    __syncthreads();
    unsigned base = 0;
    for (int j = 0; j < w; ++j) base += wtot[j];
    return incl + base;
}

// Up to QJOBS independent problems in ONE launch (the three logging statistics of a CP2 step: queue-logit rows,
// positive and negative dense pairs): a workgroup handles one row of one job, so 96 CUs work for one launch duration
// instead of 32 CUs three times in a row.
constexpr int QJOBS = 4;
struct QuantJobs {
    QuantArgs job[QJOBS];
    int first_row[QJOBS + 1];                              // workgroup b belongs to job j with first_row[j] <= b < first_row[j+1]
};

__global__ __launch_bounds__(QT) void quantiles_kernel(QuantJobs jobs) {
    int jsel = 0;
#pragma unroll
    for (int j = 1; j < QJOBS; ++j) jsel += ((int)blockIdx.x >= jobs.first_row[j]) ? 1 : 0;
    const QuantArgs& a = jobs.job[jsel];
    extern __shared__ __attribute__((aligned(16))) unsigned char q_smem[];
    __shared__ unsigned hist0[QB0];
    __shared__ unsigned hist[QMAX][QB1];
    __shared__ unsigned wtot[16];
    __shared__ unsigned sh_prefix[QMAX], sh_k[QMAX], sh_min[QMAX], sh_next[QMAX], sh_n;
    float* lma = reinterpret_cast<float*>(q_smem);
    float* lmb = lma + (a.want >= 0 ? a.P : 0);
    const int r = (int)blockIdx.x - jobs.first_row[jsel], tid = threadIdx.x, NQ = a.NQ;
    const float* row = a.x + (int64_t)r * a.s_row;
    const bool masked = a.want >= 0;
    if (masked) {
        for (int i = tid; i < a.P; i += QT) { lma[i] = a.mask_a[(int64_t)r * a.P + i]; lmb[i] = a.mask_b[(int64_t)r * a.P + i]; }
    }
    for (int i = tid; i < QB0; i += QT) hist0[i] = 0;
    __syncthreads();
    const bool vec = a.s_elem == 1 && ((reinterpret_cast<uintptr_t>(row) & 15u) == 0);
    auto keep_at = [&](int x, int y, float v) -> bool {
        if (v != v) return false;                          // nanquantile ignores NaN
        if (!masked) return true;
        return ((lma[x] * lmb[y]) != 0.f) == (a.want != 0);
    };
    // Visit every kept element of the row once: 16-byte loads when the row is contiguous, four loads in flight otherwise.
#define CP2_Q_FOREACH(BODY)                                                                              \
    if (vec) {                                                                                           \
        const int n4 = (a.N + 3) >> 2;                                                                   \
        for (int j4 = tid; j4 < n4; j4 += 4 * QT) {                                                      \
            float4 t4[4];                                                                                \
            _Pragma("unroll") for (int g = 0; g < 4; ++g) {      /* four 16-byte loads in flight */       \
                const int i0 = (j4 + g * QT) * 4;                                                        \
                if (i0 + 3 < a.N) {                                                                      \
                    t4[g] = *reinterpret_cast<const float4*>(row + i0);                                  \
                } else {                                                                                 \
                    t4[g].x = (i0 + 0 < a.N) ? row[i0 + 0] : NAN;                                        \
                    t4[g].y = (i0 + 1 < a.N) ? row[i0 + 1] : NAN;                                        \
                    t4[g].z = (i0 + 2 < a.N) ? row[i0 + 2] : NAN;                                        \
                    t4[g].w = NAN;                                                                       \
                }                                                                                        \
            }                                                                                            \
            _Pragma("unroll") for (int g = 0; g < 4; ++g) {                                              \
                const int i0 = (j4 + g * QT) * 4;                                                        \
                const float vv[4] = {t4[g].x, t4[g].y, t4[g].z, t4[g].w};                                \
                int x_ = masked ? i0 / a.P : 0, y_ = masked ? i0 % a.P : 0;                              \
                _Pragma("unroll") for (int u = 0; u < 4; ++u) {                                          \
                    const float v = vv[u];                                                               \
                    if (keep_at(x_, y_, v)) { BODY }                                                     \
                    if (masked && ++y_ >= a.P) { y_ = 0; ++x_; }                                         \
                }                                                                                        \
            }                                                                                            \
        }                                                                                                \
    } else {                                                                                             \
        for (int i0 = tid; i0 < a.N; i0 += 4 * QT) {                                                     \
            float vv[4];                                                                                 \
            _Pragma("unroll") for (int u = 0; u < 4; ++u) {                                              \
                const int i = i0 + u * QT;                                                               \
                vv[u] = i < a.N ? row[(int64_t)i * a.s_elem] : NAN;                                      \
            }                                                                                            \
            _Pragma("unroll") for (int u = 0; u < 4; ++u) {                                              \
                const int i = i0 + u * QT;                                                               \
                const float v = vv[u];                                                                   \
                if (keep_at(masked ? i / a.P : 0, masked ? i % a.P : 0, v)) { BODY }                     \
            }                                                                                            \
        }                                                                                                \
    }

    // ---- pass 0: top 12 bits, one histogram for all quantiles (its total is n)
    CP2_Q_FOREACH(atomicAdd(&hist0[f2key(v) >> 20], 1u);)
    __syncthreads();
    {
        const unsigned h0 = hist0[4 * tid], h1 = hist0[4 * tid + 1], h2 = hist0[4 * tid + 2], h3 = hist0[4 * tid + 3];
        const unsigned tot = h0 + h1 + h2 + h3;
        const unsigned incl = block_scan_incl(tot, wtot), excl = incl - tot;
        if (tid == QT - 1) sh_n = incl;
        __syncthreads();
        const unsigned n = sh_n;
        if (n == 0) {
            if (tid < NQ) a.out[(int64_t)tid * a.R + r] = NAN;
            return;
        }
        for (int j = 0; j < NQ; ++j) {
            const unsigned lo = (unsigned)floorf(a.q[j] * (float)(n - 1));
            if (lo >= excl && lo < incl) {
                unsigned kk = lo - excl, b = 4 * tid;
                if (kk >= h0) { kk -= h0; ++b; if (kk >= h1) { kk -= h1; ++b; if (kk >= h2) { kk -= h2; ++b; } } }
                sh_prefix[j] = b;
                sh_k[j] = kk;
            }
        }
    }
    // ---- passes 1 and 2: ten more bits each, one histogram per quantile
    for (int pass = 1; pass <= 2; ++pass) {
        for (int i = tid; i < QMAX * QB1; i += QT) (&hist[0][0])[i] = 0;
        if (tid < QMAX) { sh_min[tid] = 0xFFFFFFFFu; sh_next[tid] = 0xFFFFFFFFu; }
        __syncthreads();
        unsigned pre[QMAX], mn[QMAX];
#pragma unroll
        for (int j = 0; j < QMAX; ++j) { pre[j] = j < NQ ? sh_prefix[j] : 0xFFFFFFFFu; mn[j] = 0xFFFFFFFFu; }
        const int sh = pass == 1 ? 20 : 10;
        CP2_Q_FOREACH(
            const unsigned k = f2key(v);
            const unsigned top = k >> sh;
            const unsigned bin = (k >> (sh - 10)) & (QB1 - 1);
            _Pragma("unroll") for (int j = 0; j < QMAX; ++j) {
                if (top == pre[j]) atomicAdd(&hist[j][bin], 1u);
                else if (pass == 2 && j < NQ && top > pre[j]) mn[j] = min(mn[j], k);
            })
        if (pass == 2) {
#pragma unroll
            for (int j = 0; j < QMAX; ++j) {
                unsigned m = mn[j];
#pragma unroll
                for (int off = 32; off > 0; off >>= 1) m = min(m, (unsigned)__shfl_xor((int)m, off, 64));
                if ((tid & 63) == 0 && m != 0xFFFFFFFFu) atomicMin(&sh_min[j], m);
            }
        }
        __syncthreads();
        for (int j = 0; j < NQ; ++j) {
            const unsigned hv = hist[j][tid];
            const unsigned incl = block_scan_incl(hv, wtot), excl = incl - hv;
            const unsigned kk0 = sh_k[j];
            __syncthreads();                               // everyone has read sh_k[j] before it is rewritten
            if (kk0 >= excl && kk0 < incl) {
                sh_prefix[j] = (pre[j] << 10) | (unsigned)tid;
                sh_k[j] = kk0 - excl;
            }
            __syncthreads();
            if (pass == 2) {
                // the next non-empty bin above the selected one (the partner when the selected key is not repeated)
                const unsigned sel = sh_prefix[j] & (QB1 - 1);
                if (hv != 0 && (unsigned)tid > sel) atomicMin(&sh_next[j], (unsigned)tid);
            }
        }
        __syncthreads();
    }
    if (tid < NQ) {
        const int j = tid;
        const unsigned n = sh_n;
        const float rank = a.q[j] * (float)(n - 1);
        const float lo_f = floorf(rank), w = rank - lo_f;
        const unsigned key_lo = sh_prefix[j];
        const float v_lo = key2f(key_lo);
        float v_hi = v_lo;
        if (w != 0.f) {
            const unsigned mult = hist[j][key_lo & (QB1 - 1)];
            if (sh_k[j] + 1 >= mult) {                     // the element of rank lo + 1 is a larger key
                if (sh_next[j] != 0xFFFFFFFFu) v_hi = key2f((key_lo & ~(unsigned)(QB1 - 1)) | sh_next[j]);
                else if (sh_min[j] != 0xFFFFFFFFu) v_hi = key2f(sh_min[j]);
            }
        }
        const float d = v_hi - v_lo;                         // at::lerp: w < 0.5 ? a + w*(b-a) : b - (b-a)*(1-w)
        a.out[(int64_t)j * a.R + r] = (w < 0.5f) ? v_lo + w * d : v_hi - d * (1.f - w);
    }
#undef CP2_Q_FOREACH
}

static int quant_check(const QuantArgs& a) {
    if (!a.x || !a.q || !a.out) return CP2_ERR_NULL;
    if (a.R <= 0 || a.N <= 0 || a.NQ <= 0) return CP2_ERR_SHAPE;
    if (a.NQ > QMAX) return CP2_ERR_UNSUPPORTED;
    if (a.want >= 0 && (!a.mask_a || !a.mask_b || a.P <= 0 || (int64_t)a.P * a.P != a.N)) return CP2_ERR_SHAPE;
    if (a.want >= 0 && a.P > 8192) return CP2_ERR_UNSUPPORTED;
    return CP2_OK;
}

static int quant_launch(const QuantJobs& jobs, int njobs, hipStream_t stream) {
    size_t lds = 0;
    for (int j = 0; j < njobs; ++j) {
        int rc = quant_check(jobs.job[j]);
        if (rc) return rc;
        const size_t l = jobs.job[j].want >= 0 ? 2 * (size_t)jobs.job[j].P * sizeof(float) : 0;
        if (l > lds) lds = l;
    }
    if (lds > 16384) {
        hipError_t e_ = hipFuncSetAttribute(reinterpret_cast<const void*>(quantiles_kernel),
                                            hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e_ != hipSuccess) return (int)e_;
    }
    CP2_LAUNCH_PROFILED(quantiles_kernel, dim3(jobs.first_row[njobs]), dim3(QT), lds, stream, jobs);
    return cp2_launch_status();
}

CP2_API int cp2_masked_quantiles(const float* x, int64_t stride_row, int64_t stride_elem, int R, int N,
                                 const float* mask_a, const float* mask_b, int P, int want, const float* q, int NQ,
                                 float* out, void* stream) {
    QuantJobs jobs{};
    jobs.job[0] = QuantArgs{x, stride_row, stride_elem, N, mask_a, mask_b, P, want, q, NQ, out, R};
    for (int j = 1; j <= QJOBS; ++j) jobs.first_row[j] = R > 0 ? R : 0;
    return quant_launch(jobs, 1, cp2_stream(stream));
}

CP2_API int cp2_masked_quantiles_multi(int njobs, const float* const* x, const int64_t* stride_row, const int64_t* stride_elem,
                                       const int* R, const int* N, const float* const* mask_a, const float* const* mask_b,
                                       const int* P, const int* want, const float* q, int NQ, float* const* out, void* stream) {
    if (njobs <= 0 || njobs > QJOBS) return CP2_ERR_UNSUPPORTED;
    if (!x || !stride_row || !stride_elem || !R || !N || !mask_a || !mask_b || !P || !want || !out) return CP2_ERR_NULL;
    QuantJobs jobs{};
    int rows = 0;
    for (int j = 0; j < njobs; ++j) {
        jobs.job[j] = QuantArgs{x[j], stride_row[j], stride_elem[j], N[j], mask_a[j], mask_b[j], P[j], want[j], q, NQ, out[j], R[j]};
        jobs.first_row[j] = rows;
        rows += R[j] > 0 ? R[j] : 0;
    }
    for (int j = njobs; j <= QJOBS; ++j) jobs.first_row[j] = rows;
    return quant_launch(jobs, njobs, cp2_stream(stream));
}
