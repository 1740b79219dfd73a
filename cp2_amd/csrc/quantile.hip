// a15: logging statistics that the reference obtains by sorting every step on every rank --
//   tools/correlation_mapping.py:16-53  per-sample nanquantile([.25,.5,.75]) of the positive / negative dense
//                                       scores (pairs selected by mask_a[x]*mask_b[y])
//   builder.py:1399-1406                row quantiles of the b x K queue logits
// Exact order statistics without sorting: a radix select over the order-preserving integer image of the floats
// (12 + 10 + 10 bits), all requested quantiles of a row together, then the interpolation of
// torch.quantile(..., interpolation='linear'):  rank = q*(n-1) in fp32, lerp(v_lo, v_hi, frac).
//
// Two forms.  Rows of at most CP2_QUANTILES_ROW_MAX elements (the training step: 32 x 65536 queue logits + 2 x 32 x 196^2
// dense pairs): quantiles_row_kernel, ONE launch, one workgroup per row through the three levels (42 us per step; 66 before
// the row was kept in registers).
// Longer rows (BASELINE config 4: 16 rows of 16.7 M dense logits) are cut into 8192-element chunks and every level is
// a chunk-parallel histogram pass with a tiny per-row select between the passes:
//   K1  quantile_hist_kernel<0>   chunk: LDS histogram of the top 12 bits         -> global hist0 (atomic adds)
//   S1  quantile_select_kernel<0> row:   rank -> (bin, rank in bin) per quantile, number of kept elements
//   K2  quantile_hist_kernel<1>   chunk: next 10 bits of the elements in the selected first-level bins -> hist1
//   S2  quantile_select_kernel<1> row
//   K3  quantile_hist_kernel<2>   chunk: last 10 bits of the elements with the selected 22-bit prefix -> hist2, and the
//                                 smallest key above that prefix (the interpolation partner when a bin's maximum is hit)
//   S3  quantile_select_kernel<2> row:   final key, partner, interpolation; the row's workspace is left zeroed
// The positive and the negative dense scores are two classes of ONE logit map (an element belongs to exactly one): their
// two jobs are served by one read of the map per level (round 4: 6 x 1.07 GB -> 3 x 1.07 GB per call at config 4).
// Every level counts ALL matching elements, so the cost does not depend on the data: an earlier form of this path
// compacted the first-level candidates and finished them in one workgroup per row -- 1.2 ms on Gaussian test data, but
// the dense logits of a freshly initialised encoder are a handful of distinct fp32 values next to 1.0, every candidate
// list overflowed, and the per-row fallback cost 10.9-12.7 ms per step at config 4 (19 % of the step).
// Integer counting only: the result does not depend on the order in which atomics arrive (bit-exact vs torch.nanquantile).
// The logits are read from memory, i.e. they ARE materialised by the loss kernels when quartile logging is on: building
// the first histogram inside the loss kernels would need 32 rows x 4096 bins of LDS per tile (512 KB), and recomputing
// the P x P logits in every pass costs three more MFMA passes (3 x 0.44 ms at config 4) against 0.36 ms for writing
// them once and reading them three times at HBM speed.
#include "common.hpp"
#include "rowkey_small_fin.hpp"
#include "dense_post.hpp"
#include <math.h>

struct QuantArgs {
    const float* x; int64_t s_row, s_elem; int N;         // element i of row r at x[r*s_row + i*s_elem]
    const float* mask_a; const float* mask_b; int P; int want;  // want < 0: all elements; else keep i iff (mask_a[r][i/P]*mask_b[r][i%P] != 0) == want
    const float* q; int NQ;
    float* out;                                            // [NQ][R] (torch.quantile layout)
    int R;
    int chunks;                                            // chunk workgroups per row: ceil(ceil(N / QCHUNK) / QCPW)
    float* mean_out;                                       // NULL, or [R]: mean of the row as torch's x.mean(1) (NaN if the row holds one); want < 0, row kernel only
    int pair;                                              // chunked form: 1 = this job (want 1) and the next one (want 0) read the same logits:
                                                           // its chunk workgroups fill both jobs' histograms from ONE read; 2 = that next job
};

__device__ __forceinline__ unsigned f2key(float f) {
    const unsigned u = __float_as_uint(f);
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
__device__ __forceinline__ float key2f(unsigned k) {
    return __uint_as_float((k & 0x80000000u) ? (k & 0x7fffffffu) : ~k);
}

constexpr int QMAX = 4;                   // quantiles per call
constexpr int QB0 = 4096, QB1 = 1024;     // bins of the first / second and third level
constexpr int QCHUNK = 8192;              // elements per chunk of the chunk-parallel passes (one set of loads in flight)
constexpr int QCPW = 4;                   // chunks per workgroup of those passes
constexpr int QT1 = 256;                  // threads per workgroup of the chunk-parallel passes
constexpr int QT3 = 1024;                 // threads per workgroup of the per-row kernels
constexpr int QJOBS = 4;
constexpr unsigned QNONE = 0xFFFFFFFFu;

struct QuantJobs {
    QuantArgs job[QJOBS];
    int first_row[QJOBS + 1];              // rows of job j: [first_row[j], first_row[j+1])
    int first_chunk[QJOBS + 1];            // chunk workgroups of job j
    // workspace of the chunked form (device, zero between calls): per global row rt
    unsigned* hist0;                       // [rows][QB0]
    unsigned* hist1;                       // [rows][QMAX][QB1]
    unsigned* hist2;                       // [rows][QMAX][QB1]
    unsigned* above;                       // [rows][QMAX] max over ~key of the elements above the 22-bit prefix (0 = none)
    unsigned* sel;                         // [rows][QSEL]: n | per quantile: prefix so far, rank inside it
};
constexpr int QSEL = 1 + 2 * QMAX;
// per-row workspace words: hist0 | hist1 | hist2 | above | sel
constexpr int64_t QROW_WORDS = QB0 + 2 * QMAX * QB1 + QMAX + QSEL;

template <int NT>
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

// Histogram adds of FOUR consecutive elements per lane (one 16-byte load) for a converged 64-lane wavefront.  The logits of
// a young encoder lie in a band a few float bins wide, where 256 adds to one LDS address serialise: if every element of
// every lane is wanted and has the same bin, lane 0 adds 256 at once (one readfirstlane + compares + one ballot per
// four elements); otherwise the lanes add for themselves.
__device__ __forceinline__ void hist_add4_wave(unsigned* h, const unsigned (&bin)[4], const bool (&pred)[4]) {
    const unsigned b0 = (unsigned)__builtin_amdgcn_readfirstlane((int)bin[0]);
    const bool uni = pred[0] && pred[1] && pred[2] && pred[3] && bin[0] == b0 && bin[1] == b0 && bin[2] == b0 && bin[3] == b0;
    if (__ballot(!uni) == 0ull) {
        if ((threadIdx.x & 63) == 0) atomicAdd(&h[b0], 256u);
    } else {
#pragma unroll
        for (int u = 0; u < 4; ++u)
            if (pred[u]) atomicAdd(&h[bin[u]], 1u);
    }
}
// (Round 4 measured a more general form -- the bin of the first pending element counted over the wave by ballots, one lane
// adding the count, twice, stragglers for themselves: 114 -> 73-82 us on rows within a few ulps of 1, 70 -> 65 us on a two-bin
// band, but 40 -> 50 us on spread-out rows and 42 -> 60 us inside the bench step, whose workgroups already run at the
// 128-register limit: the extra ballots spill scalar registers.  Not kept.)

// One chunk (QCHUNK elements = 8 float4 per thread of a 256-thread workgroup), split into "issue every load" and
// "process": the passes put all global loads of a workgroup (data, mask values, histograms) in flight at once, so a
// workgroup pays ONE memory round trip instead of one per loop iteration.
template <int NT>
struct ChunkData {
    static constexpr int G = QCHUNK / (NT * 4);
    float4 v[G];
    float4 mb4[G]; float fa[G];                            // masked jobs with P % 4 == 0: mask_b of the four elements, mask_a of their x
    bool vec, mvec;
};

template <int NT>
__device__ __forceinline__ void chunk_load(const QuantArgs& a, int r, int begin, int end, ChunkData<NT>& d) {
    const float* row = a.x + (int64_t)r * a.s_row;
    d.vec = a.s_elem == 1 && ((reinterpret_cast<uintptr_t>(row) & 15u) == 0);
    d.mvec = false;
    if (!d.vec) return;
#pragma unroll
    for (int g = 0; g < ChunkData<NT>::G; ++g) {
        const int i = begin + (g * NT + (int)threadIdx.x) * 4;
        if (i + 3 < end) {
            d.v[g] = *reinterpret_cast<const float4*>(row + i);
        } else {
            d.v[g].x = (i + 0 < end) ? row[i + 0] : NAN;
            d.v[g].y = (i + 1 < end) ? row[i + 1] : NAN;
            d.v[g].z = (i + 2 < end) ? row[i + 2] : NAN;
            d.v[g].w = NAN;
        }
    }
    if (a.want >= 0) {
        const float* ma = a.mask_a + (int64_t)r * a.P;
        const float* mb = a.mask_b + (int64_t)r * a.P;
        d.mvec = (a.P & 3) == 0 && ((reinterpret_cast<uintptr_t>(mb) & 15u) == 0);
        if (d.mvec) {
            // (x, y) of a group's first element: begin + loc = x * P + y; the four elements of a group share x (P % 4 == 0)
            const int x0 = begin / a.P, rem0 = begin - x0 * a.P;
            const float invP = 1.0f / (float)a.P;
#pragma unroll
            for (int g = 0; g < ChunkData<NT>::G; ++g) {
                const int t = rem0 + (g * NT + (int)threadIdx.x) * 4;       // < P + QCHUNK: exact through a float quotient and one correction
                int dx = (int)((float)t * invP);
                if (dx * a.P > t) --dx; else if ((dx + 1) * a.P <= t) ++dx;
                const int x_ = x0 + dx, y_ = t - dx * a.P;
                const bool in = x_ < a.P;
                d.fa[g] = in ? ma[x_] : 0.f;
                d.mb4[g] = in ? *reinterpret_cast<const float4*>(mb + y_) : make_float4(0.f, 0.f, 0.f, 0.f);
            }
        }
    }
}

// key[u] / cls[u] of the four elements of the thread's group g (vec chunks): cls 1 = positive pair (mask_a[x] * mask_b[y] != 0),
// 0 = negative pair or unmasked job, -1 = not an element (NaN, beyond the row's end)
template <int NT>
__device__ __forceinline__ void chunk_classify(const QuantArgs& a, int r, int begin, int end, const ChunkData<NT>& d, int g,
                                               unsigned (&key)[4], int (&cls)[4]) {
    const bool masked = a.want >= 0;
    const int loc = (g * NT + (int)threadIdx.x) * 4, i = begin + loc;
    const float vv[4] = {d.v[g].x, d.v[g].y, d.v[g].z, d.v[g].w};
    if (!masked) {
#pragma unroll
        for (int u = 0; u < 4; ++u) { key[u] = f2key(vv[u]); cls[u] = (i + u < end && vv[u] == vv[u]) ? 0 : -1; }
    } else if (d.mvec) {
        const float mm[4] = {d.mb4[g].x, d.mb4[g].y, d.mb4[g].z, d.mb4[g].w};
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            key[u] = f2key(vv[u]);
            cls[u] = (i + u < end && vv[u] == vv[u]) ? (((d.fa[g] * mm[u]) != 0.f) ? 1 : 0) : -1;
        }
    } else {
        const float* ma = a.mask_a + (int64_t)r * a.P;
        const float* mb = a.mask_b + (int64_t)r * a.P;
        int x_ = i / a.P, y_ = i - x_ * a.P;
        float fa = x_ < a.P ? ma[x_] : 0.f;
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            key[u] = f2key(vv[u]);
            cls[u] = (i + u < end && vv[u] == vv[u]) ? (((fa * (x_ < a.P ? mb[y_] : 0.f)) != 0.f) ? 1 : 0) : -1;
            if (++y_ >= a.P) { y_ = 0; ++x_; fa = (x_ < a.P) ? ma[x_] : 0.f; }
        }
    }
}

// Strided or unaligned rows: element by element, f(key, cls) for every non-NaN element.
template <int NT, typename F>
__device__ __forceinline__ void chunk_visit_scalar(const QuantArgs& a, int r, int begin, int end, F&& f) {
    const bool masked = a.want >= 0;
    const float* ma = masked ? a.mask_a + (int64_t)r * a.P : nullptr;
    const float* mb = masked ? a.mask_b + (int64_t)r * a.P : nullptr;
    const float* row = a.x + (int64_t)r * a.s_row;
    for (int i = begin + (int)threadIdx.x; i < end; i += NT) {
        const float v = row[(int64_t)i * a.s_elem];
        if (v != v) continue;
        int c = 0;
        if (masked) { const int x_ = i / a.P; c = ((ma[x_] * mb[i - x_ * a.P]) != 0.f) ? 1 : 0; }
        f(f2key(v), c);
    }
}

__device__ __forceinline__ int job_of(const int* first, int b) {
    int jsel = 0;
#pragma unroll
    for (int j = 1; j < QJOBS; ++j) jsel += (b >= first[j]) ? 1 : 0;
    return jsel;
}

// The thread that holds rank `lo` among its BPT consecutive bins (exclusive / inclusive prefix of its bins' total:
// excl, incl) publishes the bin and the rank inside it.
template <int BPT>
__device__ __forceinline__ void locate(const unsigned (&hv)[BPT], unsigned excl, unsigned incl, unsigned lo, unsigned* bin,
                                       unsigned* kk) {
    if (lo >= excl && lo < incl) {
        unsigned k = lo - excl;
        int b = 0;
#pragma unroll
        for (int u = 0; u < BPT - 1; ++u)
            if (b == u && k >= hv[u]) { k -= hv[u]; ++b; }
        *bin = (unsigned)(BPT * (int)threadIdx.x + b);
        *kk = k;
    }
}

// ---- chunk-parallel histogram of level LEVEL (0: top 12 bits of every kept element; 1: next 10 bits of the elements in
// the selected first-level bin; 2: last 10 bits of the elements with the selected 22-bit prefix + smallest key above it).
// A chunk of a PAIRED job (QuantArgs::pair == 1: the positive and the negative class of one logit map) is read once and
// every element goes to the histograms of its own class (side 0 = this job's row, side 1 = the next job's row).
// Quantiles whose prefix so far is equal share one LDS histogram; the flush adds it to each of their global ones.
//
// What bounds these passes (round 4, rocprofv3 --pmc on the config-4 shape): not the LDS atomics and not HBM but VALU
// instructions -- the first version spent 48 (level 0) to 128 (levels 1, 2) vector instructions per element slot, 270 M
// wave instructions per launch = 0.5 ms on the chip's 1024 SIMDs, whatever the data.  Hence the shape of the loop below:
// an element's class is folded into its key once ("extended top" = prefix bits, class bit), every table the loop
// compares with is wave-uniform (scalar registers), a level-1 / 2 element costs one compare and one select per DISTINCT
// live prefix, and the LDS histogram is cleared and flushed with 16-byte accesses.
template <int LEVEL>
__global__ __launch_bounds__(QT1) void quantile_hist_kernel(QuantJobs jobs) {
    constexpr int NB = LEVEL == 0 ? QB0 : QMAX * QB1;
    constexpr unsigned NEVERX = 0xFFFFFFFFu;               // no extended top has this value (tops have at most 22 bits)
    __shared__ __attribute__((aligned(16))) unsigned h[2 * NB];
    __shared__ unsigned sh_min[2 * QMAX];
    const int jsel = job_of(jobs.first_chunk, (int)blockIdx.x);
    const QuantArgs& a = jobs.job[jsel];
    const int c = (int)blockIdx.x - jobs.first_chunk[jsel], r = c / a.chunks, s = c - r * a.chunks;
    const int tid = threadIdx.x, NQ = a.NQ;
    const bool pair = a.pair == 1;
    const int nside = pair ? 2 : 1;
    const int64_t rt[2] = {jobs.first_row[jsel] + r, pair ? jobs.first_row[jsel + 1] + r : 0};
    const unsigned* sel[2] = {jobs.sel + rt[0] * QSEL, jobs.sel + rt[1] * QSEL};
    bool live[2] = {true, pair};
    if (LEVEL > 0) {                                       // (readfirstlane: loaded values, uniform by construction -> scalar registers)
        live[0] = __builtin_amdgcn_readfirstlane((int)sel[0][0]) != 0;
        live[1] = pair && __builtin_amdgcn_readfirstlane((int)sel[1][0]) != 0;
        if (!live[0] && !live[1]) return;                  // nothing kept in this row (workgroup-uniform)
    }
    // class (0: negative pair / unmasked, 1: positive pair) -> side, or -1 when the class is not counted by this job
    int side_of[2];
    side_of[0] = pair ? 1 : ((a.want < 0 || a.want == 0) ? 0 : -1);
    side_of[1] = pair ? 0 : ((a.want < 0 || a.want == 1) ? 0 : -1);
    // per class: distinct live prefixes as extended tops, the LDS base of their histogram, and (level 2) the first key above
    unsigned effx[2][QMAX], hi1[2][QMAX], pre[2][QMAX];
    int hbase[2][QMAX], slot[2][QMAX];
#pragma unroll
    for (int cl = 0; cl < 2; ++cl) {
        const int sd = side_of[cl];
#pragma unroll
        for (int j = 0; j < QMAX; ++j) {
            const bool on = LEVEL > 0 && j < NQ && sd >= 0 && live[sd < 0 ? 0 : sd];
            pre[cl][j] = on ? (unsigned)__builtin_amdgcn_readfirstlane((int)sel[sd][1 + 2 * j]) : QNONE;
            slot[cl][j] = j;
#pragma unroll
            for (int i = j - 1; i >= 0; --i)
                if (pre[cl][i] == pre[cl][j]) slot[cl][j] = i;
            const bool first = on && slot[cl][j] == j;     // a duplicate prefix counts in the first one's histogram
            effx[cl][j] = first ? ((pre[cl][j] << 1) | (unsigned)cl) : NEVERX;
            hbase[cl][j] = (sd < 0 ? 0 : sd) * NB + j * QB1;
            // smallest key above the prefix as min over (k - hi1) mod 2^32 (see qrow_body_cached): hi1 = first key above its range
            hi1[cl][j] = (LEVEL == 2 && first) ? ((pre[cl][j] << 10) | (unsigned)(QB1 - 1)) + 1u : 0u;
        }
    }
    unsigned mn[2][QMAX];
#pragma unroll
    for (int cl = 0; cl < 2; ++cl)
#pragma unroll
        for (int j = 0; j < QMAX; ++j) mn[cl][j] = NEVERX;
    {
        const uint4 z = make_uint4(0u, 0u, 0u, 0u);
        for (int i = tid; i < nside * NB / 4; i += QT1) reinterpret_cast<uint4*>(h)[i] = z;
    }
    if (LEVEL == 2 && tid < 2 * QMAX) sh_min[tid] = QNONE;
    __syncthreads();
    // element -> histogram index (or -1: not counted); level 2 also tracks the smallest key above every live prefix
    auto index_of = [&](unsigned k, int cl) -> int {       // cl: 0 / 1, or -1 = not an element
        if (LEVEL == 0) {
            const int sd = cl < 0 ? -1 : (cl ? side_of[1] : side_of[0]);
            return sd < 0 ? -1 : sd * NB + (int)(k >> 20);
        }
        constexpr int SH = LEVEL == 1 ? 20 : 10;
        const unsigned topx = cl < 0 ? NEVERX : (((k >> SH) << 1) | (unsigned)cl);
        int idx = -1;
#pragma unroll
        for (int c2 = 0; c2 < 2; ++c2)
#pragma unroll
            for (int j = 0; j < QMAX; ++j)
                if (effx[c2][j] != NEVERX && topx == effx[c2][j]) idx = hbase[c2][j];      // first test: scalar, skips dead entries
        if (LEVEL == 2) {
#pragma unroll
            for (int c2 = 0; c2 < 2; ++c2) {
                const unsigned kc = cl == c2 ? k : NEVERX;     // elements of the other class (or none) are neutral for the minimum
#pragma unroll
                for (int j = 0; j < QMAX; ++j)
                    if (effx[c2][j] != NEVERX) mn[c2][j] = min(mn[c2][j], kc - hi1[c2][j]);
            }
        }
        return idx < 0 ? -1 : idx + (int)((k >> (SH - 10)) & (QB1 - 1));
    };
    // QCPW consecutive chunks per workgroup: the LDS histogram is cleared and flushed (global atomics, the dearest part of a
    // spread-out row's pass) once per 32768 elements
    for (int cc = 0; cc < QCPW; ++cc) {
        const int begin = (s * QCPW + cc) * QCHUNK, end = min(a.N, begin + QCHUNK);
        if (begin >= a.N) break;
        ChunkData<QT1> d;
        chunk_load<QT1>(a, r, begin, end, d);              // every load of the chunk in flight
        if (d.vec) {
#pragma unroll
            for (int g = 0; g < ChunkData<QT1>::G; ++g) {
                unsigned key[4];
                int cl[4];
                chunk_classify<QT1>(a, r, begin, end, d, g, key, cl);
                // one LDS atomic per element: measured against a one-lane add of a hot bin's count (hist_add4_wave, what the row
                // kernel uses) these passes are 20-35 % faster WITHOUT it on every distribution -- they are bound by vector
                // instructions and by the LDS atomic rate (about one lane per clock and CU), not by same-address conflicts
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const int ix = index_of(key[u], cl[u]);
                    if (ix >= 0) atomicAdd(&h[ix], 1u);
                }
            }
        } else {
            chunk_visit_scalar<QT1>(a, r, begin, end, [&](unsigned k, int c2) {
                const int ix = index_of(k, c2);
                if (ix >= 0) atomicAdd(&h[ix], 1u);
            });
        }
    }
    if (LEVEL == 2) {
#pragma unroll
        for (int cl = 0; cl < 2; ++cl) {
            const int sd = side_of[cl];
#pragma unroll
            for (int j = 0; j < QMAX; ++j) {
                if (effx[cl][j] == NEVERX) continue;       // scalar
                unsigned m = mn[cl][j];
#pragma unroll
                for (int off = 32; off > 0; off >>= 1) m = min(m, (unsigned)__shfl_xor((int)m, off, 64));
                const unsigned kmin = m + hi1[cl][j];      // back to a key; below hi1 = wrapped = nothing above the prefix
                if ((tid & 63) == 0 && kmin >= hi1[cl][j] && kmin != NEVERX) atomicMin(&sh_min[sd * QMAX + j], kmin);
            }
        }
    }
    __syncthreads();
    for (int sd = 0; sd < nside; ++sd) {
        if (!live[sd]) continue;
        const int cl = side_of[0] == sd ? 0 : 1;           // the class this side counts (an unmasked job: class 0)
        // lanes add CONSECUTIVE bins: a wave instruction's atomics fall into 256 contiguous bytes (one lane per 16 bytes
        // quadruples the memory-side requests: measured 274 -> 600 us for the first-level pass)
        if (LEVEL == 0) {
            unsigned* g = jobs.hist0 + rt[sd] * QB0;
            for (int i = tid; i < QB0; i += QT1) {
                const unsigned v = h[sd * NB + i];
                if (v) atomicAdd(&g[i], v);
            }
        } else {
            unsigned* g = (LEVEL == 1 ? jobs.hist1 : jobs.hist2) + rt[sd] * QMAX * QB1;
            for (int j = 0; j < NQ; ++j) {
                const int sj = cl ? slot[1][j] : slot[0][j];
                for (int i = tid; i < QB1; i += QT1) {
                    const unsigned v = h[sd * NB + sj * QB1 + i];
                    if (v) atomicAdd(&g[j * QB1 + i], v);
                }
            }
            if (LEVEL == 2 && tid < NQ) {
                const int sj = cl ? slot[1][tid] : slot[0][tid];      // a duplicate shares the first one's "smallest key above"
                const unsigned m = sh_min[sd * QMAX + sj];
                if (m != QNONE) atomicMax(&jobs.above[rt[sd] * QMAX + tid], ~m);
            }
        }
    }
}

// ---- per-row select after level LEVEL: one workgroup per row
template <int LEVEL>
__global__ __launch_bounds__(QT3) void quantile_select_kernel(QuantJobs jobs) {
    __shared__ unsigned wtot[QT3 / 64];
    __shared__ unsigned sh_bin[QMAX], sh_kk[QMAX], sh_next[QMAX], sh_n;
    const int jsel = job_of(jobs.first_row, (int)blockIdx.x);
    const QuantArgs& a = jobs.job[jsel];
    const int r = (int)blockIdx.x - jobs.first_row[jsel], tid = threadIdx.x, NQ = a.NQ;
    const int64_t rt = blockIdx.x;
    unsigned* sel = jobs.sel + rt * QSEL;
    if (LEVEL == 0) {
        constexpr int BPT = QB0 / QT3;
        unsigned* g0 = jobs.hist0 + rt * QB0;
        unsigned hv[BPT];
        {
            const uint4 t = reinterpret_cast<const uint4*>(g0)[tid];
            hv[0] = t.x; hv[1] = t.y; hv[2] = t.z; hv[3] = t.w;
        }
        const unsigned tot = hv[0] + hv[1] + hv[2] + hv[3];
        const unsigned incl = block_scan_incl<QT3>(tot, wtot), excl = incl - tot;
        if (tid == QT3 - 1) sh_n = incl;
        __syncthreads();
        const unsigned n = sh_n;
        if (n > 0) {
            for (int j = 0; j < NQ; ++j)
                locate<BPT>(hv, excl, incl, (unsigned)floorf(a.q[j] * (float)(n - 1)), &sh_bin[j], &sh_kk[j]);
        }
        __syncthreads();
        if (tid == 0) sel[0] = n;
        if (n > 0 && tid < NQ) { sel[1 + 2 * tid] = sh_bin[tid]; sel[2 + 2 * tid] = sh_kk[tid]; }
        if (n == 0 && tid < NQ) a.out[(int64_t)tid * a.R + r] = NAN;        // nothing kept: the later passes skip the row
        reinterpret_cast<uint4*>(g0)[tid] = make_uint4(0u, 0u, 0u, 0u);     // hist0 is done: zero for the next call
        if (n == 0 && tid == 0) sel[0] = 0;
        return;
    }
    const unsigned n = sel[0];
    if (n == 0) return;                                    // (the level-0 select wrote NaN; nothing was added anywhere)
    unsigned* g = (LEVEL == 1 ? jobs.hist1 : jobs.hist2) + rt * QMAX * QB1;
    unsigned hvj[QMAX], pre[QMAX], kk[QMAX];
#pragma unroll
    for (int j = 0; j < QMAX; ++j) {                       // one memory round trip for everything this kernel reads
        hvj[j] = j < NQ ? g[j * QB1 + tid] : 0u;
        pre[j] = j < NQ ? sel[1 + 2 * j] : 0u;
        kk[j] = j < NQ ? sel[2 + 2 * j] : 0u;
    }
    if (tid < QMAX) sh_next[tid] = QNONE;
#pragma unroll
    for (int j = 0; j < QMAX; ++j) {
        if (j < NQ) {
            const unsigned one[1] = {hvj[j]};
            const unsigned incl = block_scan_incl<QT3>(hvj[j], wtot);
            locate<1>(one, incl - hvj[j], incl, kk[j], &sh_bin[j], &sh_kk[j]);
        }
    }
    __syncthreads();
    if (LEVEL == 1) {
        if (tid < NQ) { sel[1 + 2 * tid] = (pre[tid] << 10) | sh_bin[tid]; sel[2 + 2 * tid] = sh_kk[tid]; }
    } else {
        // the next non-empty bin above the selected one (the partner when the selected key is not repeated)
#pragma unroll
        for (int j = 0; j < QMAX; ++j)
            if (j < NQ && hvj[j] != 0 && (unsigned)tid > sh_bin[j]) atomicMin(&sh_next[j], (unsigned)tid);
        __syncthreads();
        if (tid < NQ) {
            const int j = tid;
            const float rank = a.q[j] * (float)(n - 1);
            const float lo_f = floorf(rank), w = rank - lo_f;
            const unsigned key_lo = (pre[j] << 10) | sh_bin[j];
            const float v_lo = key2f(key_lo);
            float v_hi = v_lo;
            if (w != 0.f) {
                const unsigned mult = g[j * QB1 + sh_bin[j]];
                if (sh_kk[j] + 1 >= mult) {                // the element of rank lo + 1 is a larger key
                    const unsigned ab = jobs.above[rt * QMAX + j];
                    if (sh_next[j] != QNONE) v_hi = key2f((pre[j] << 10) | sh_next[j]);
                    else if (ab != 0u) v_hi = key2f(~ab);
                }
            }
            const float dlt = v_hi - v_lo;                   // at::lerp: w < 0.5 ? a + w*(b-a) : b - (b-a)*(1-w)
            a.out[(int64_t)j * a.R + r] = (w < 0.5f) ? v_lo + w * dlt : v_hi - dlt * (1.f - w);
        }
        __syncthreads();
        if (tid < QMAX) jobs.above[rt * QMAX + tid] = 0;
        if (tid < QSEL) sel[tid] = 0;
    }
    // this level's histograms are done: zero them for the next call (stream order makes it visible)
#pragma unroll
    for (int j = 0; j < QMAX; ++j) g[j * QB1 + tid] = 0u;
}

// ---- small rows (N <= QROW_MAX): ONE launch, one workgroup per row through all three levels (round 1 / early round 2).
// At the training step's shapes (32 x 65536 queue logits + 2 x 32 x 196^2 dense pairs) three chunked launches pay
// their per-workgroup set-up and two launch boundaries for 18 MB of data, and the logits of a freshly initialised
// encoder lie in a band so narrow that a 12-bit float prefix holds most of a row (the candidate lists overflow):
// measured inside the step 74 us for the three launches against 66 us for this kernel.  The chunked form above is for
// rows a single CU cannot stream (BASELINE config 4: 16.7 M elements per row).
// (Measured and rejected: 8 replicas per histogram bin, lane % 8, against same-address serialisation of the LDS atomics
// on narrow-band rows -- 70 us instead of 63-66: the passes are bound by their ~30 VALU instructions per element on the
// one CU a row has, not by atomic conflicts.)
constexpr int QROW_MAX = CP2_QUANTILES_ROW_MAX;

// Shared state of one row's workgroup (declared once in the kernel: the body below is instantiated per mask mode).
struct QRowShared {
    unsigned* hist0;           // [QB0 + 1]
    unsigned* hist;            // [QMAX][QB1]
    unsigned* wtot;            // [16]
    unsigned* prefix;          // [QMAX]
    unsigned* k;               // [QMAX]
    unsigned* mn;              // [QMAX]
    unsigned* next;            // [QMAX]
    unsigned* n;               // [1]
    double* sum_w;             // [QT3 / 64]
    float* lma; float* lmb;
};

// Visit every element of the row once: f(v, keep).  16-byte loads (four in flight) when the row is contiguous.  keep is
// false for NaN (nanquantile ignores it) and, for a masked job, for elements outside the wanted mask class; the position
// (x, y) = (i / P, i % P) of a masked row advances incrementally (one division per thread and pass, not per load).
template <bool MASKED, typename F>
__device__ __forceinline__ void qrow_foreach(const QuantArgs& a, const float* __restrict__ row, const float* lma, const float* lmb,
                                             bool vec, F&& f) {
    const int tid = threadIdx.x, P = MASKED ? a.P : 1;
    const bool wantpos = a.want != 0;
    if (vec) {
        const int n4 = (a.N + 3) >> 2;
        constexpr int STEP = 4 * QT3;                        // elements between two consecutive float4 of a thread
        int x_ = 0, y_ = 0;
        const int dq = MASKED ? STEP / P : 0, dr = MASKED ? STEP - dq * P : 0;
        if (MASKED) { x_ = (tid * 4) / P; y_ = tid * 4 - x_ * P; }
        for (int j4 = tid; j4 < n4; j4 += 4 * QT3) {
            float4 t4[4];
#pragma unroll
            for (int g = 0; g < 4; ++g) {                    // four 16-byte loads in flight
                const int i0 = (j4 + g * QT3) * 4;
                if (i0 + 3 < a.N) {
                    t4[g] = *reinterpret_cast<const float4*>(row + i0);
                } else {
                    t4[g].x = (i0 + 0 < a.N) ? row[i0 + 0] : NAN;
                    t4[g].y = (i0 + 1 < a.N) ? row[i0 + 1] : NAN;
                    t4[g].z = (i0 + 2 < a.N) ? row[i0 + 2] : NAN;
                    t4[g].w = NAN;
                }
            }
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const float vv[4] = {t4[g].x, t4[g].y, t4[g].z, t4[g].w};
                int xx = x_, yy = y_;
                float fa = 0.f;
                if (MASKED) fa = xx < P ? lma[xx] : 0.f;
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const float v = vv[u];
                    bool keep = v == v;
                    if (MASKED) {
                        keep = keep && (((fa * lmb[yy]) != 0.f) == wantpos);
                        if (++yy >= P) { yy = 0; ++xx; fa = xx < P ? lma[xx] : 0.f; }
                    }
                    f(v, keep);
                }
                if (MASKED) {                                // the thread's next float4 is STEP elements further
                    y_ += dr; x_ += dq;
                    if (y_ >= P) { y_ -= P; ++x_; }
                }
            }
        }
    } else {
        for (int i0 = tid; i0 < a.N; i0 += 4 * QT3) {
            float vv[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int i = i0 + u * QT3;
                vv[u] = i < a.N ? row[(int64_t)i * a.s_elem] : NAN;
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int i = i0 + u * QT3;
                const float v = vv[u];
                bool keep = v == v;
                if (MASKED && keep) { const int x = i / P; keep = ((lma[x] * lmb[i - x * P]) != 0.f) == wantpos; }
                f(v, keep);
            }
        }
    }
}

// The three levels of one row.  What keeps the instruction count per element down (the kernel is VALU-bound on the one CU
// a row has -- 7.6 us per 1000 instructions per thread):
//   * the mask arithmetic exists only in the MASKED instantiation (the 65536-element queue-logit rows carry none);
//   * quantiles whose prefix so far is the same share ONE histogram (on the narrow-band rows of a young encoder all three
//     quartiles sit in one 12-bit bin: one LDS add per element and level instead of three); a duplicate's prefix is
//     replaced by a value no key has, its selection reads the histogram of the first quantile with that prefix;
//   * the prefixes live in scalar registers and one test ("does this element match any of them") guards the rest;
//   * the smallest key above each prefix (the interpolation partner) is tracked with compare + select + min, no branch.
template <bool MASKED, int NQT>
__device__ __forceinline__ void qrow_body(const QuantArgs& a, int r, const QRowShared& S) {
    const int tid = threadIdx.x, NQ = a.NQ;
    const float* row = a.x + (int64_t)r * a.s_row;
    const bool vec = a.s_elem == 1 && ((reinterpret_cast<uintptr_t>(row) & 15u) == 0);
    // ---- level 0: top 12 bits, one histogram for all quantiles (its total is n); the row sum rides along
    float lsum = 0.f;
    qrow_foreach<MASKED>(a, row, S.lma, S.lmb, vec, [&](float v, bool keep) {
        if (keep) { atomicAdd(&S.hist0[f2key(v) >> 20], 1u); lsum += v; }
    });
    if (a.mean_out) {
        double ds = (double)lsum;
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) ds += __shfl_xor(ds, off, 64);
        if ((tid & 63) == 0) S.sum_w[tid >> 6] = ds;
    }
    __syncthreads();
    {
        const unsigned h0 = S.hist0[4 * tid], h1 = S.hist0[4 * tid + 1], h2 = S.hist0[4 * tid + 2], h3 = S.hist0[4 * tid + 3];
        const unsigned tot = h0 + h1 + h2 + h3;
        const unsigned incl = block_scan_incl<QT3>(tot, S.wtot), excl = incl - tot;
        if (tid == QT3 - 1) *S.n = incl;
        __syncthreads();
        const unsigned n = *S.n;
        if (a.mean_out && tid == 0) {
            double t = 0;
            for (int i = 0; i < QT3 / 64; ++i) t += S.sum_w[i];
            a.mean_out[r] = (n == (unsigned)a.N) ? (float)(t / (double)a.N) : NAN;   // a NaN element makes torch's mean NaN
        }
        if (n == 0) {
            if (tid < NQ) a.out[(int64_t)tid * a.R + r] = NAN;
            return;
        }
        for (int j = 0; j < NQ; ++j) {
            const unsigned lo = (unsigned)floorf(a.q[j] * (float)(n - 1));
            if (lo >= excl && lo < incl) {
                unsigned kk = lo - excl, b = 4 * tid;
                if (kk >= h0) { kk -= h0; ++b; if (kk >= h1) { kk -= h1; ++b; if (kk >= h2) { kk -= h2; ++b; } } }
                S.prefix[j] = b;
                S.k[j] = kk;
            }
        }
    }
    // ---- levels 1 and 2: ten more bits each, one histogram per DISTINCT prefix
    constexpr unsigned NEVER = 0xFFFFFFFFu;                  // no key has this 12- / 22-bit prefix
    int slot[NQT];
    for (int pass = 1; pass <= 2; ++pass) {
        for (int i = tid; i < QMAX * QB1; i += QT3) S.hist[i] = 0;
        if (tid < QMAX) { S.mn[tid] = NEVER; S.next[tid] = NEVER; }
        __syncthreads();
        unsigned pre[NQT], eff[NQT], hi[NQT], mn[NQT];
        const int sh = pass == 1 ? 20 : 10;
#pragma unroll
        for (int j = 0; j < NQT; ++j) {
            pre[j] = j < NQ ? (unsigned)__builtin_amdgcn_readfirstlane((int)S.prefix[j]) : NEVER;
            slot[j] = j;
#pragma unroll
            for (int i = j - 1; i >= 0; --i)
                if (pre[i] == pre[j]) slot[j] = i;
            eff[j] = slot[j] == j ? pre[j] : NEVER;
            hi[j] = (pass == 2 && j < NQ) ? ((pre[j] << 10) | (unsigned)(QB1 - 1)) : NEVER;   // keys above it have a larger prefix
            mn[j] = NEVER;
        }
        qrow_foreach<MASKED>(a, row, S.lma, S.lmb, vec, [&](float v, bool keep) {
            const unsigned k = f2key(v);
            const unsigned top = k >> sh;
            bool any = false;
#pragma unroll
            for (int j = 0; j < NQT; ++j) any = any || top == eff[j];
            if (keep && any) {
                const unsigned bin = (k >> (sh - 10)) & (QB1 - 1);
#pragma unroll
                for (int j = 0; j < NQT; ++j)
                    if (top == eff[j]) atomicAdd(&S.hist[j * QB1 + bin], 1u);
            }
            if (pass == 2) {
#pragma unroll
                for (int j = 0; j < NQT; ++j) mn[j] = min(mn[j], (keep && k > hi[j]) ? k : NEVER);
            }
        });
        if (pass == 2) {
#pragma unroll
            for (int j = 0; j < NQT; ++j) {
                unsigned m = mn[j];
#pragma unroll
                for (int off = 32; off > 0; off >>= 1) m = min(m, (unsigned)__shfl_xor((int)m, off, 64));
                if ((tid & 63) == 0 && m != NEVER) atomicMin(&S.mn[j], m);
            }
        }
        __syncthreads();
#pragma unroll
        for (int j = 0; j < NQT; ++j) {
            if (j >= NQ) break;
            const unsigned hv = S.hist[slot[j] * QB1 + tid];
            const unsigned incl = block_scan_incl<QT3>(hv, S.wtot), excl = incl - hv;
            const unsigned kk0 = S.k[j];
            __syncthreads();                               // everyone has read k[j] before it is rewritten
            if (kk0 >= excl && kk0 < incl) {
                S.prefix[j] = (pre[j] << 10) | (unsigned)tid;
                S.k[j] = kk0 - excl;
            }
            __syncthreads();
            if (pass == 2) {
                // the next non-empty bin above the selected one (the partner when the selected key is not repeated)
                const unsigned sel = S.prefix[j] & (QB1 - 1);
                if (hv != 0 && (unsigned)tid > sel) atomicMin(&S.next[j], (unsigned)tid);
            }
        }
        __syncthreads();
    }
    if (tid < NQ) {
        const int j = tid;
        const unsigned n = *S.n;
        const float rank = a.q[j] * (float)(n - 1);
        const float lo_f = floorf(rank), w = rank - lo_f;
        const unsigned key_lo = S.prefix[j];
        const float v_lo = key2f(key_lo);
        float v_hi = v_lo;
        if (w != 0.f) {
            int sj = j;                                      // the histogram this quantile shared on the last level
            for (int i = j - 1; i >= 0; --i)
                if ((S.prefix[i] >> 10) == (key_lo >> 10)) sj = i;
            const unsigned mult = S.hist[sj * QB1 + (key_lo & (QB1 - 1))];
            if (S.k[j] + 1 >= mult) {                      // the element of rank lo + 1 is a larger key
                if (S.next[j] != NEVER) v_hi = key2f((key_lo & ~(unsigned)(QB1 - 1)) | S.next[j]);
                else if (S.mn[j] != NEVER) v_hi = key2f(S.mn[j]);
            }
        }
        const float d = v_hi - v_lo;                         // at::lerp: w < 0.5 ? a + w*(b-a) : b - (b-a)*(1-w)
        a.out[(int64_t)j * a.R + r] = (w < 0.5f) ? v_lo + w * d : v_hi - d * (1.f - w);
    }
}

// Inclusive scans of NV values per thread at once over the QT3 threads (one pair of barriers for all of them).
template <int NV>
__device__ __forceinline__ void block_scan_incl_n(unsigned (&v)[NV], unsigned* wtot /* [16 * NV] */) {
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
#pragma unroll
    for (int q = 0; q < NV; ++q) {
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) {
            const unsigned t = (unsigned)__shfl_up((int)v[q], off, 64);
            if (lane >= off) v[q] += t;
        }
    }
    __syncthreads();                                       // wtot may still be read by the previous scan
    if (lane == 63) {
#pragma unroll
        for (int q = 0; q < NV; ++q) wtot[q * 16 + w] = v[q];
    }
    __syncthreads();
#pragma unroll
    for (int q = 0; q < NV; ++q) {
        unsigned base = 0;
        for (int j = 0; j < w; ++j) base += wtot[q * 16 + j];
        v[q] += base;
    }
}

// The same three levels with the row held in registers: F4 16-byte loads per thread, all in flight at once, converted to
// keys on arrival (an element that is not kept -- NaN, outside the mask class, beyond N -- becomes key 0xFFFFFFFF, the
// image of no kept float: it matches no prefix and is the neutral element of the "smallest key above" minimum, so levels
// 1 and 2 need neither the mask nor the NaN test nor memory).  N <= F4 * 4 * QT3, contiguous 16-byte aligned rows.
template <bool MASKED, int NQT, int F4>
__device__ __forceinline__ void qrow_body_cached(const QuantArgs& a, int r, const QRowShared& S) {
    constexpr unsigned NEVER = 0xFFFFFFFFu;
    const int tid = threadIdx.x, NQ = a.NQ, P = MASKED ? a.P : 1;
    const float* row = a.x + (int64_t)r * a.s_row;
    const bool wantpos = a.want != 0;
    unsigned key[F4 * 4];
    float lsum = 0.f;
    {
        constexpr int STEP = 4 * QT3;
        int x_ = 0, y_ = 0;
        const int dq = MASKED ? STEP / P : 0, dr = MASKED ? STEP - dq * P : 0;
        if (MASKED) { x_ = (tid * 4) / P; y_ = tid * 4 - x_ * P; }
        // ---- level 0: top 12 bits, one histogram for all quantiles (its total is n); the row sum rides along.
        // Loads in batches of at most eight (the keys of the whole row stay live: 64 registers at F4 = 16).
        constexpr int BATCH = F4 > 8 ? (F4 + 1) / 2 : F4;
#pragma unroll
        for (int g0 = 0; g0 < F4; g0 += BATCH) {
            float4 t4[BATCH];
#pragma unroll
            for (int gb = 0; gb < BATCH; ++gb) {
                const int g = g0 + gb;
                if (g >= F4) break;
                const int i0 = (tid + g * QT3) * 4;
                if (i0 + 3 < a.N) {
                    t4[gb] = *reinterpret_cast<const float4*>(row + i0);
                } else {
                    t4[gb].x = (i0 + 0 < a.N) ? row[i0 + 0] : NAN;
                    t4[gb].y = (i0 + 1 < a.N) ? row[i0 + 1] : NAN;
                    t4[gb].z = (i0 + 2 < a.N) ? row[i0 + 2] : NAN;
                    t4[gb].w = NAN;
                }
            }
#pragma unroll
            for (int gb = 0; gb < BATCH; ++gb) {
                const int g = g0 + gb;
                if (g >= F4) break;
                const float vv[4] = {t4[gb].x, t4[gb].y, t4[gb].z, t4[gb].w};
                int xx = x_, yy = y_;
                float fa = 0.f;
                if (MASKED) fa = xx < P ? S.lma[xx] : 0.f;
                unsigned bin[4];
                bool keep[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const float v = vv[u];
                    keep[u] = v == v;
                    if (MASKED) {
                        keep[u] = keep[u] && (((fa * S.lmb[yy]) != 0.f) == wantpos);
                        if (++yy >= P) { yy = 0; ++xx; fa = xx < P ? S.lma[xx] : 0.f; }
                    }
                    const unsigned k = keep[u] ? f2key(v) : NEVER;
                    key[g * 4 + u] = k;
                    bin[u] = k >> 20;
                    lsum += keep[u] ? v : 0.f;
                }
                hist_add4_wave(S.hist0, bin, keep);
                if (MASKED) {
                    y_ += dr; x_ += dq;
                    if (y_ >= P) { y_ -= P; ++x_; }
                }
            }
        }
    }
    if (a.mean_out) {
        double ds = (double)lsum;
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) ds += __shfl_xor(ds, off, 64);
        if ((tid & 63) == 0) S.sum_w[tid >> 6] = ds;
    }
    __syncthreads();
    {
        const unsigned h0 = S.hist0[4 * tid], h1 = S.hist0[4 * tid + 1], h2 = S.hist0[4 * tid + 2], h3 = S.hist0[4 * tid + 3];
        const unsigned tot = h0 + h1 + h2 + h3;
        const unsigned incl = block_scan_incl<QT3>(tot, S.wtot), excl = incl - tot;
        if (tid == QT3 - 1) *S.n = incl;
        __syncthreads();
        const unsigned n = *S.n;
        if (a.mean_out && tid == 0) {
            double t = 0;
            for (int i = 0; i < QT3 / 64; ++i) t += S.sum_w[i];
            a.mean_out[r] = (n == (unsigned)a.N) ? (float)(t / (double)a.N) : NAN;   // a NaN element makes torch's mean NaN
        }
        if (n == 0) {
            if (tid < NQ) a.out[(int64_t)tid * a.R + r] = NAN;
            return;
        }
        for (int j = 0; j < NQ; ++j) {
            const unsigned lo = (unsigned)floorf(a.q[j] * (float)(n - 1));
            if (lo >= excl && lo < incl) {
                unsigned kk = lo - excl, b = 4 * tid;
                if (kk >= h0) { kk -= h0; ++b; if (kk >= h1) { kk -= h1; ++b; if (kk >= h2) { kk -= h2; ++b; } } }
                S.prefix[j] = b;
                S.k[j] = kk;
            }
        }
    }
    // ---- levels 1 and 2 on the keys in registers: one histogram per DISTINCT prefix
    int slot[NQT];
    for (int pass = 1; pass <= 2; ++pass) {
        for (int i = tid; i < NQT * QB1; i += QT3) S.hist[i] = 0;
        if (tid < QMAX) { S.mn[tid] = NEVER; S.next[tid] = NEVER; }
        __syncthreads();
        unsigned pre[NQT], eff[NQT], hi[NQT], mn[NQT];
        const int sh = pass == 1 ? 20 : 10;
#pragma unroll
        for (int j = 0; j < NQT; ++j) {
            pre[j] = j < NQ ? (unsigned)__builtin_amdgcn_readfirstlane((int)S.prefix[j]) : NEVER;
            slot[j] = j;
#pragma unroll
            for (int i = j - 1; i >= 0; --i)
                if (pre[i] == pre[j]) slot[j] = i;
            eff[j] = slot[j] == j ? pre[j] : NEVER;
            // smallest key above the prefix: min over (k - hi1) mod 2^32 with hi1 = first key above the prefix's range.  Keys
            // above give small values, the not-kept image 0xFFFFFFFF the largest of those, keys at or below wrap to still
            // larger ones -- so the minimum is the wanted key whenever one exists (two instructions per element, no compare)
            hi[j] = (pass == 2 && j < NQ) ? ((pre[j] << 10) | (unsigned)(QB1 - 1)) + 1u : 0u;
            mn[j] = NEVER;
        }
#pragma unroll
        for (int g = 0; g < F4; ++g) {
            unsigned top[4];
            bool any = false;
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                top[u] = key[g * 4 + u] >> sh;
#pragma unroll
                for (int j = 0; j < NQT; ++j) any = any || top[u] == eff[j];
            }
            if (__ballot(any) != 0ull) {                     // wave-uniform: most wavefronts of a spread-out row skip this
                unsigned bin[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) bin[u] = (key[g * 4 + u] >> (sh - 10)) & (QB1 - 1);
#pragma unroll
                for (int j = 0; j < NQT; ++j) {
                    if (eff[j] == NEVER) continue;            // a duplicate prefix (scalar test)
                    const bool m[4] = {top[0] == eff[j], top[1] == eff[j], top[2] == eff[j], top[3] == eff[j]};
                    hist_add4_wave(S.hist + j * QB1, bin, m);
                }
            }
            if (pass == 2) {
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const unsigned k = key[g * 4 + u];
#pragma unroll
                    for (int j = 0; j < NQT; ++j) mn[j] = min(mn[j], k - hi[j]);
                }
            }
        }
        if (pass == 2) {
#pragma unroll
            for (int j = 0; j < NQT; ++j) {
                unsigned m = mn[j];
#pragma unroll
                for (int off = 32; off > 0; off >>= 1) m = min(m, (unsigned)__shfl_xor((int)m, off, 64));
                const unsigned kmin = m + hi[j];                 // back to a key; below hi1 = wrapped = nothing above the prefix
                if ((tid & 63) == 0 && j < NQ && kmin >= hi[j] && kmin != NEVER) atomicMin(&S.mn[j], kmin);
            }
        }
        __syncthreads();
        unsigned hv[NQT], sc[NQT];
#pragma unroll
        for (int j = 0; j < NQT; ++j) { hv[j] = S.hist[slot[j] * QB1 + tid]; sc[j] = hv[j]; }
        block_scan_incl_n<NQT>(sc, S.wtot);
        unsigned kk0[NQT];
#pragma unroll
        for (int j = 0; j < NQT; ++j) kk0[j] = j < NQ ? S.k[j] : 0u;
        __syncthreads();                                   // everyone has read k[] before it is rewritten
#pragma unroll
        for (int j = 0; j < NQT; ++j) {
            const unsigned excl = sc[j] - hv[j];
            if (j < NQ && kk0[j] >= excl && kk0[j] < sc[j]) {
                S.prefix[j] = (pre[j] << 10) | (unsigned)tid;
                S.k[j] = kk0[j] - excl;
            }
        }
        __syncthreads();
        if (pass == 2) {
#pragma unroll
            for (int j = 0; j < NQT; ++j) {
                if (j >= NQ) break;
                // the next non-empty bin above the selected one (the partner when the selected key is not repeated)
                const unsigned sel = S.prefix[j] & (QB1 - 1);
                if (hv[j] != 0 && (unsigned)tid > sel) atomicMin(&S.next[j], (unsigned)tid);
            }
        }
        __syncthreads();
    }
    if (tid < NQ) {
        const int j = tid;
        const unsigned n = *S.n;
        const float rank = a.q[j] * (float)(n - 1);
        const float lo_f = floorf(rank), w = rank - lo_f;
        const unsigned key_lo = S.prefix[j];
        const float v_lo = key2f(key_lo);
        float v_hi = v_lo;
        if (w != 0.f) {
            int sj = j;                                      // the histogram this quantile shared on the last level
            for (int i = j - 1; i >= 0; --i)
                if ((S.prefix[i] >> 10) == (key_lo >> 10)) sj = i;
            const unsigned mult = S.hist[sj * QB1 + (key_lo & (QB1 - 1))];
            if (S.k[j] + 1 >= mult) {                      // the element of rank lo + 1 is a larger key
                if (S.next[j] != NEVER) v_hi = key2f((key_lo & ~(unsigned)(QB1 - 1)) | S.next[j]);
                else if (S.mn[j] != NEVER) v_hi = key2f(S.mn[j]);
            }
        }
        const float d = v_hi - v_lo;                         // at::lerp: w < 0.5 ? a + w*(b-a) : b - (b-a)*(1-w)
        a.out[(int64_t)j * a.R + r] = (w < 0.5f) ? v_lo + w * d : v_hi - d * (1.f - w);
    }
}

// `block` = the row, counted over all jobs (the kernels below pass blockIdx.x or blockIdx.x minus their rider workgroups)
__device__ __forceinline__ void quantiles_row_block(const QuantJobs& jobs, int block) {
    int jsel = 0;
#pragma unroll
    for (int j = 1; j < QJOBS; ++j) jsel += (block >= jobs.first_row[j]) ? 1 : 0;
    const QuantArgs& a = jobs.job[jsel];
    extern __shared__ __attribute__((aligned(16))) unsigned char q_smem[];
    __shared__ unsigned hist0[QB0];
    __shared__ unsigned hist[QMAX * QB1];
    __shared__ unsigned wtot[16 * QMAX];
    __shared__ unsigned sh_prefix[QMAX], sh_k[QMAX], sh_min[QMAX], sh_next[QMAX], sh_n;
    __shared__ double sum_w[QT3 / 64];
    const bool masked = a.want >= 0;
    QRowShared S;
    S.hist0 = hist0; S.hist = hist; S.wtot = wtot; S.prefix = sh_prefix; S.k = sh_k; S.mn = sh_min; S.next = sh_next; S.n = &sh_n;
    S.sum_w = sum_w;
    S.lma = reinterpret_cast<float*>(q_smem);
    S.lmb = S.lma + (masked ? a.P : 0);
    const int r = block - jobs.first_row[jsel], tid = threadIdx.x;
    if (masked) {
        for (int i = tid; i < a.P; i += QT3) { S.lma[i] = a.mask_a[(int64_t)r * a.P + i]; S.lmb[i] = a.mask_b[(int64_t)r * a.P + i]; }
    }
    for (int i = tid; i < QB0; i += QT3) hist0[i] = 0;
    __syncthreads();
    // NQ <= 3 (the step's quartiles) gets loops of three; a fourth quantile the general body.  Rows of at most 65536
    // contiguous elements are held in registers through the three levels (the step: 38416 and 65536 elements).
    const bool vec = a.s_elem == 1 && (((reinterpret_cast<uintptr_t>(a.x) | (uintptr_t)((int64_t)a.s_row * 4)) & 15u) == 0);
    if (a.NQ <= 3 && vec && a.N <= 10 * 4 * QT3) {
        if (masked) qrow_body_cached<true, 3, 10>(a, r, S); else qrow_body_cached<false, 3, 10>(a, r, S);
    } else if (a.NQ <= 3 && vec && a.N <= 16 * 4 * QT3) {
        if (masked) qrow_body_cached<true, 3, 16>(a, r, S); else qrow_body_cached<false, 3, 16>(a, r, S);
    } else if (a.NQ <= 3) {
        if (masked) qrow_body<true, 3>(a, r, S); else qrow_body<false, 3>(a, r, S);
    } else {
        if (masked) qrow_body<true, QMAX>(a, r, S); else qrow_body<false, QMAX>(a, r, S);
    }
}

__global__ __launch_bounds__(QT3) void quantiles_row_kernel(QuantJobs jobs) { quantiles_row_block(jobs, (int)blockIdx.x); }

// The step's form (round 4): the quartile rows AND the two tails of the loss section that depend on nothing the quartiles
// write -- the instance loss's finalize (rowkey_small_finalize_body, nfin workgroups) and the dense loss's post-pass (one
// workgroup per sample) -- in ONE launch.  The riders come first in the grid (they are short; the 96 row workgroups of the
// bench step need 42 us); all of them have 1024 threads.  cp2_step_post, see include/cp2hip.h.
__global__ __launch_bounds__(QT3) void step_post_kernel(QuantJobs jobs, RowKeyFinArgs fa, float* __restrict__ loss_mean, DenseArgs da,
                                                        float* __restrict__ sample_scal, int64_t BP, int nfin, int npost) {
    const int b = (int)blockIdx.x;
    if (b < nfin) { rowkey_small_finalize_body(fa, loss_mean, b); return; }
    if (b < nfin + npost) { dense_post_body<QT3>(da, sample_scal, BP, b - nfin); return; }
    quantiles_row_block(jobs, b - nfin - npost);
}

static int quant_check(const QuantArgs& a) {
    if (!a.x || !a.q || !a.out) return CP2_ERR_NULL;
    if (a.R <= 0 || a.N <= 0 || a.NQ <= 0) return CP2_ERR_SHAPE;
    if (a.NQ > QMAX) return CP2_ERR_UNSUPPORTED;
    if (a.want >= 0 && (!a.mask_a || !a.mask_b || a.P <= 0 || (int64_t)a.P * a.P != a.N)) return CP2_ERR_SHAPE;
    if (a.mean_out && (a.want >= 0 || a.N > QROW_MAX)) return CP2_ERR_UNSUPPORTED;      // (the chunked three-level form has no sums)
    return CP2_OK;
}

// words (4 bytes) of workspace for the given jobs
static int64_t quant_ws_words(int njobs, const int* R, const int* N, int NQ) {
    int64_t rows = 0;
    for (int j = 0; j < njobs; ++j)
        if (R[j] > 0 && N[j] > 0) rows += R[j];
    return rows * QROW_WORDS;
}

CP2_API int64_t cp2_quantiles_workspace_bytes(int njobs, const int* R, const int* N, int NQ) {
    if (njobs <= 0 || njobs > QJOBS || !R || !N || NQ <= 0 || NQ > QMAX) return 0;
    return 4 * quant_ws_words(njobs, R, N, NQ);
}

struct QuantRider {                                        // the loss-section tails that ride in the row launch (cp2_step_post)
    RowKeyFinArgs fa; float* loss_mean; DenseArgs da; float* sample_scal; int B;
};

static int quant_launch(QuantJobs& jobs, int njobs, void* workspace, int64_t workspace_bytes, hipStream_t stream,
                        const QuantRider* rider = nullptr) {
    bool small = true;
    for (int j = 0; j < njobs; ++j) small = small && jobs.job[j].N <= QROW_MAX;
    if (rider && !small) return CP2_ERR_UNSUPPORTED;       // long rows: cp2_loss_post + cp2_masked_quantiles_multi
    if (small) {                                           // one launch, one workgroup per row, no workspace
        size_t lds = 0;
        int rows = 0;
        for (int j = 0; j < njobs; ++j) {
            int rc = quant_check(jobs.job[j]);
            if (rc) return rc;
            const size_t l = jobs.job[j].want >= 0 ? 2 * (size_t)jobs.job[j].P * sizeof(float) : 0;
            if (l > lds) lds = l;
            jobs.first_row[j] = rows;
            rows += jobs.job[j].R;
        }
        for (int j = njobs; j <= QJOBS; ++j) jobs.first_row[j] = rows;
        if (rider) {
            const int nfin = CH / FS2_CPB;
            CP2_LAUNCH_PROFILED(step_post_kernel, dim3(nfin + rider->B + rows), dim3(QT3), lds, stream, jobs, rider->fa, rider->loss_mean,
                                rider->da, rider->sample_scal, (int64_t)rider->B * rider->da.P, nfin, rider->B);
            return cp2_launch_status();
        }
        CP2_LAUNCH_PROFILED(quantiles_row_kernel, dim3(rows), dim3(QT3), lds, stream, jobs);
        return cp2_launch_status();
    }
    if (!workspace) return CP2_ERR_NULL;
    if (!cp2_aligned16(workspace)) return CP2_ERR_ALIGN;
    int rows = 0, chunks = 0;
    int Rs[QJOBS], Ns[QJOBS];
    for (int j = 0; j < njobs; ++j) jobs.job[j].pair = 0;
    for (int j = 0; j + 1 < njobs; ++j) {                  // (want 1, want 0) over the same logits and masks: one read per level
        QuantArgs& p = jobs.job[j];
        QuantArgs& n = jobs.job[j + 1];
        if (p.pair == 0 && p.want == 1 && n.want == 0 && p.x == n.x && p.s_row == n.s_row && p.s_elem == n.s_elem && p.N == n.N &&
            p.R == n.R && p.P == n.P && p.mask_a == n.mask_a && p.mask_b == n.mask_b && p.NQ == n.NQ) {
            p.pair = 1;
            n.pair = 2;
        }
    }
    for (int j = 0; j < njobs; ++j) {
        QuantArgs& a = jobs.job[j];
        int rc = quant_check(a);
        if (rc) return rc;
        a.chunks = cp2_cdiv(cp2_cdiv(a.N, QCHUNK), QCPW);  // chunk WORKGROUPS per row
        jobs.first_row[j] = rows;
        jobs.first_chunk[j] = chunks;
        rows += a.R;
        if (a.pair != 2) {                                 // the second job of a pair has no chunk workgroups of its own
            if ((int64_t)chunks + (int64_t)a.R * a.chunks > 0x7fffffff) return CP2_ERR_UNSUPPORTED;
            chunks += a.R * a.chunks;
        }
        Rs[j] = a.R; Ns[j] = a.N;
    }
    for (int j = njobs; j <= QJOBS; ++j) { jobs.first_row[j] = rows; jobs.first_chunk[j] = chunks; }
    if (4 * quant_ws_words(njobs, Rs, Ns, jobs.job[0].NQ) > workspace_bytes) return CP2_ERR_SHAPE;
    unsigned* w = static_cast<unsigned*>(workspace);
    jobs.hist0 = w;
    jobs.hist1 = jobs.hist0 + (int64_t)rows * QB0;
    jobs.hist2 = jobs.hist1 + (int64_t)rows * QMAX * QB1;
    jobs.above = jobs.hist2 + (int64_t)rows * QMAX * QB1;
    jobs.sel = jobs.above + (int64_t)rows * QMAX;
    // measurement aid: the armed start event rides on the first launch, the stop event on the last (elapsed = all six)
    const Cp2LaunchEvents ev = cp2_next_events;
    cp2_next_events = Cp2LaunchEvents{};
    if (ev.start) hipExtLaunchKernelGGL(quantile_hist_kernel<0>, dim3(chunks), dim3(QT1), 0, stream, ev.start, nullptr, 0, jobs);
    else hipLaunchKernelGGL(quantile_hist_kernel<0>, dim3(chunks), dim3(QT1), 0, stream, jobs);
    hipLaunchKernelGGL(quantile_select_kernel<0>, dim3(rows), dim3(QT3), 0, stream, jobs);
    hipLaunchKernelGGL(quantile_hist_kernel<1>, dim3(chunks), dim3(QT1), 0, stream, jobs);
    hipLaunchKernelGGL(quantile_select_kernel<1>, dim3(rows), dim3(QT3), 0, stream, jobs);
    hipLaunchKernelGGL(quantile_hist_kernel<2>, dim3(chunks), dim3(QT1), 0, stream, jobs);
    if (ev.stop) hipExtLaunchKernelGGL(quantile_select_kernel<2>, dim3(rows), dim3(QT3), 0, stream, nullptr, ev.stop, 0, jobs);
    else hipLaunchKernelGGL(quantile_select_kernel<2>, dim3(rows), dim3(QT3), 0, stream, jobs);
    return cp2_launch_status();
}

CP2_API int cp2_masked_quantiles(const float* x, int64_t stride_row, int64_t stride_elem, int R, int N,
                                 const float* mask_a, const float* mask_b, int P, int want, const float* q, int NQ,
                                 float* out, void* workspace, int64_t workspace_bytes, void* stream) {
    QuantJobs jobs{};
    jobs.job[0] = QuantArgs{x, stride_row, stride_elem, N, mask_a, mask_b, P, want, q, NQ, out, R, 0, nullptr, 0};
    return quant_launch(jobs, 1, workspace, workspace_bytes, cp2_stream(stream));
}

CP2_API int cp2_masked_quantiles_multi(int njobs, const float* const* x, const int64_t* stride_row, const int64_t* stride_elem,
                                       const int* R, const int* N, const float* const* mask_a, const float* const* mask_b,
                                       const int* P, const int* want, const float* q, int NQ, float* const* out,
                                       float* const* mean_out, void* workspace, int64_t workspace_bytes, void* stream) {
    if (njobs <= 0 || njobs > QJOBS) return CP2_ERR_UNSUPPORTED;
    if (!x || !stride_row || !stride_elem || !R || !N || !mask_a || !mask_b || !P || !want || !out) return CP2_ERR_NULL;
    QuantJobs jobs{};
    for (int j = 0; j < njobs; ++j)
        jobs.job[j] = QuantArgs{x[j], stride_row[j], stride_elem[j], N[j], mask_a[j], mask_b[j], P[j], want[j], q, NQ, out[j], R[j], 0,
                                mean_out ? mean_out[j] : nullptr, 0};
    return quant_launch(jobs, njobs, workspace, workspace_bytes, cp2_stream(stream));
}

// cp2_masked_quantiles_multi (its one-launch row form only: every N <= CP2_QUANTILES_ROW_MAX) + cp2_loss_post in ONE launch
CP2_API int cp2_step_post(int njobs, const float* const* x, const int64_t* stride_row, const int64_t* stride_elem,
                          const int* R, const int* N, const float* const* mask_a, const float* const* mask_b,
                          const int* P, const int* want, const float* q, int NQ, float* const* out, float* const* mean_out,
                          const float* part_m, const float* part_s, const int32_t* part_cnt, const float* part_U, int nsplit,
                          const float* extras, int NE, float temperature, float grad_scale, int fR, int RP, int64_t d_sn,
                          int64_t d_sx, int64_t d_sc, float* lse, float* loss_rows, int32_t* cnt_gt, float* drows, float* dE,
                          float* loss_mean, const float* d_mask_a, const float* d_mask_b, float* d_lse, float* colsum_a,
                          float* possum, float* allsum, float* colmax, int32_t* argx, float* sample_scal, float* split_ws,
                          int B, int C, int dP, void* stream) {
    if (njobs <= 0 || njobs > QJOBS) return CP2_ERR_UNSUPPORTED;
    if (!x || !stride_row || !stride_elem || !R || !N || !mask_a || !mask_b || !P || !want || !out) return CP2_ERR_NULL;
    QuantRider rider{};
    int rc = loss_post_fill(part_m, part_s, part_cnt, part_U, nsplit, extras, NE, temperature, grad_scale, fR, RP, d_sn, d_sx, d_sc, lse,
                            loss_rows, cnt_gt, drows, dE, d_mask_a, d_mask_b, d_lse, colsum_a, possum, allsum, colmax, argx, sample_scal,
                            split_ws, B, C, dP, &rider.fa, &rider.da);
    if (rc) return rc;
    rider.loss_mean = loss_mean; rider.sample_scal = sample_scal; rider.B = B;
    QuantJobs jobs{};
    for (int j = 0; j < njobs; ++j)
        jobs.job[j] = QuantArgs{x[j], stride_row[j], stride_elem[j], N[j], mask_a[j], mask_b[j], P[j], want[j], q, NQ, out[j], R[j], 0,
                                mean_out ? mean_out[j] : nullptr, 0};
    return quant_launch(jobs, njobs, nullptr, 0, cp2_stream(stream), &rider);
}
