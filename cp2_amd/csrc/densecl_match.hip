// a16 (T18): the positive of every query pixel of the DenseCL local loss -- reference builder.py:818-864.
//
//   best[n,x] = argmax_y < q_embed[n,:,x], k_embed[n,:,y] / |k_embed[n,:,y]| >         (backbone features, C = 2048)
//   pos[n,x]  = < q_local[n,:,x], k_local[n,:,best] >                                  (projection features, C = 128)
//             mixed, where the two id maps overlap, with sum_{y: id_k[y] == id_q[x]} < q_local[:,x], k_local[:,y] >
//   kvec[n,:,x] = d pos / d q_local[n,:,x]   (the key vector, or its mix with the summed id-matching key vectors)
//   counts      = overlapping query pixels / those whose best LOCAL match is the first id match (logged rate, :856-864)
//
// The reference materialises three b x S^2 x S^2 maps (backbone similarity, local similarity, id-equality map) and
// gathers from them; here nothing of that size exists.  One workgroup owns 32 query pixels of one sample and walks the
// sample's key pixels: the two (32 x 2048)(2048 x P) products run on the matrix cores with both operands streamed
// through LDS in channel chunks, the running arg-max stays on the lane that owns the query pixel (accumulator layout
// D[y][x]: lane = x), the query norm is never needed (a positive factor per x) and the key norm is accumulated while
// the chunks are staged.
//   * bf16 backbone features (what the bf16-autocast backbone returns): v_mfma_f32_32x32x16_bf16.  A bf16 x bf16
//     product is exact in fp32 and the sum is accumulated in fp32, so this is the precision of an fp32 GEMM on the
//     same values at 16x the f32-MFMA rate -- no split needed;
//   * fp32 features (no autocast; the reference goldens): v_mfma_f32_32x32x2_f32, an exact fp32 fma chain.
// The arg-max tie rule is torch's on the CPU: the first maximum.
#include "infonce_common.hpp"

typedef __bf16 dm_bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned int dm_u32x4 __attribute__((ext_vector_type(4)));

constexpr int DM_X = 32;      // query pixels per workgroup
constexpr int DM_Y = 256;     // key pixels per round: two 32-pixel tiles per wave
constexpr int DM_CKB = 64;    // channels per staged chunk, bf16 path (128 B per pixel row)
constexpr int DM_PB = DM_CKB + 8;   // LDS row pitch in bf16 (36 dwords: the 16 rows of a ds_read_b128 lane group hit 16 distinct 4-bank groups)
constexpr int DM_CKF = 32;    // channels per staged chunk, fp32 path
constexpr int DM_PQ = DM_X + 1, DM_PK = DM_Y + 1;   // fp32 tiles are channel-major [c][pixel], odd pitch
constexpr int DM_MAXP = 4096; // key pixels per sample (the id list lives in LDS)
constexpr int DM_ML = 8;      // id matches per query pixel kept in the LDS list (more: that pixel re-scans the key ids itself)

struct MatchArgs {
    const void* qe; const void* ke;                  // backbone features, element (n, c, p) at n*sn + c*sc + p*sp
    int64_t qe_sn, qe_sc, qe_sp, ke_sn, ke_sc, ke_sp;
    const int64_t* k_row;                            // NULL, or sample n's key side is row k_row[n] of ke / kl
    const float* ql; const float* kl;                // [B][CH][P] unit vectors
    const int64_t* ids_q; const int64_t* ids_k;      // [B][P] or NULL
    float lmbd, one_minus_lmbd;
    int normalize_k, want_metrics;
    int32_t* best; float* pos; float* kvec; int32_t* counts;
    int CE, P, XT;
};

struct DmLds {
    union {
        struct { __bf16 q[DM_X * DM_PB]; __bf16 k[DM_Y * DM_PB]; } b;
        struct { float q[DM_CKF * DM_PQ]; float k[DM_CKF * DM_PK]; } f;
    } t;
    float n2[DM_Y];
    float wv[4 * DM_X]; int wy[4 * DM_X];
    int best[DM_X]; int simbest[DM_X];
    float red[2][8 * DM_X];
    int first[DM_X];
    int cnt[2];
    int nm[DM_X]; int ml[DM_X * DM_ML];              // per query pixel: number of id-matching key pixels and the first DM_ML of them
    int64_t idk[DM_MAXP];
};

__device__ __forceinline__ float dm_bf2f(unsigned short b) { return __uint_as_float(((unsigned int)b) << 16); }

// first-maximum merge
__device__ __forceinline__ void dm_take(float& bv, int& by, float ov, int oy) {
    if (ov > bv || (ov == bv && oy < by)) { bv = ov; by = oy; }
}

// running arg-max of one round's two accumulators (rows = key pixels y_base + 64 w + 32 j + rho, lane column = query pixel)
__device__ __forceinline__ void dm_round_epilogue(const f32x16 (&acc)[2], const float* __restrict__ n2, bool normalize, int y_base,
                                                  int wid, int h, int P, float& bv, int& by) {
#pragma unroll
    for (int j = 0; j < 2; ++j) {
#pragma unroll
        for (int reg = 0; reg < 16; ++reg) {
            const int yl = 64 * wid + 32 * j + rho(reg, h), y = y_base + yl;
            if (y < P) {
                float v = acc[j][reg];
                if (normalize) v *= 1.0f / fmaxf(sqrtf(n2[yl]), 1e-12f);
                if (v > bv) { bv = v; by = y; }      // y grows with (round, j, reg) on a lane: strictly greater = first maximum
            }
        }
    }
}

// halves -> waves -> out[32] (LDS); every thread of the workgroup calls it
__device__ __forceinline__ void dm_merge(DmLds& L, float bv, int by, int tid, int* __restrict__ out) {
    const int lane = tid & 63, wid = tid >> 6, r = lane & 31, h = lane >> 5;
    dm_take(bv, by, __shfl_xor(bv, 32, 64), __shfl_xor(by, 32, 64));
    if (h == 0) { L.wv[wid * DM_X + r] = bv; L.wy[wid * DM_X + r] = by; }
    __syncthreads();
    if (tid < DM_X) {
        float v = L.wv[tid]; int y = L.wy[tid];
#pragma unroll
        for (int w = 1; w < 4; ++w) dm_take(v, y, L.wv[w * DM_X + tid], L.wy[w * DM_X + tid]);
        out[tid] = y;
    }
    __syncthreads();
}

// ---- fp32 operands, any strides: arg-max over y of sum_c q[c][x0+x] k[c][y] (optionally / |k[:,y]|) -> out[32]
// Staging: thread t handles elements e = t + 256 i of a [32 channels] x [pixels] chunk; with the channel index fastest
// in memory e = pixel * 32 + channel, else e = channel * pixels + pixel -- consecutive lanes read consecutive addresses
// either way, and both are a fixed start plus i times a fixed step.
__device__ __forceinline__ void dm_argmax_f32(DmLds& L, const float* __restrict__ q, int64_t q_sc, int64_t q_sp,
                                                        const float* __restrict__ k, int64_t k_sc, int64_t k_sp, int C, int P, int x0,
                                                        bool normalize, int tid, int* __restrict__ out) {
    const int lane = tid & 63, wid = tid >> 6, r = lane & 31, h = lane >> 5;
    const bool q_cfast = q_sc == 1, k_cfast = k_sc == 1;
    constexpr int NQ = DM_CKF * DM_X / 256, NK = DM_CKF * DM_Y / 256;
    // query chunk: channel fastest -> (c = t & 31, x = t / 32 + 8 i); else (c = t / 32 + 8 i, x = t & 31)
    const int qc0 = q_cfast ? (tid & 31) : (tid >> 5), qx0 = q_cfast ? (tid >> 5) : (tid & 31);
    const int qdc = q_cfast ? 0 : 8, qdx = q_cfast ? 8 : 0;
    // key chunk: channel fastest -> (c = t & 31, y = t / 32 + 8 i); else (c = i, y = t)
    const int kc0 = k_cfast ? (tid & 31) : 0, ky0 = k_cfast ? (tid >> 5) : tid;
    const int kdc = k_cfast ? 0 : 1, kdy = k_cfast ? 8 : 0;
    const int64_t qstep = (int64_t)qdc * q_sc + (int64_t)qdx * q_sp, kstep = (int64_t)kdc * k_sc + (int64_t)kdy * k_sp;
    float* const lq = L.t.f.q + qc0 * DM_PQ + qx0;
    float* const lk = L.t.f.k + kc0 * DM_PK + ky0;
    const int lqs = qdc * DM_PQ + qdx, lks = kdc * DM_PK + kdy;
    float bv = -INFINITY; int by = 0;
    float rq[NQ], rk[NK];
    for (int y_base = 0; y_base < P; y_base += DM_Y) {
        f32x16 acc[2] = {{0}, {0}};
        float nsq = 0.f;
        auto load = [&](int c0) __attribute__((always_inline)) {
            const float* pq = q + (int64_t)(c0 + qc0) * q_sc + (int64_t)(x0 + qx0) * q_sp;
#pragma unroll
            for (int i = 0; i < NQ; ++i) {
                rq[i] = (c0 + qc0 + i * qdc < C && x0 + qx0 + i * qdx < P) ? *pq : 0.f;
                pq += qstep;
            }
            const float* pk = k + (int64_t)(c0 + kc0) * k_sc + (int64_t)(y_base + ky0) * k_sp;
#pragma unroll
            for (int i = 0; i < NK; ++i) {
                rk[i] = (c0 + kc0 + i * kdc < C && y_base + ky0 + i * kdy < P) ? *pk : 0.f;
                pk += kstep;
            }
        };
        load(0);
        for (int c0 = 0; c0 < C; c0 += DM_CKF) {
            __syncthreads();                       // the previous chunk's MFMAs have read the tiles
#pragma unroll
            for (int i = 0; i < NQ; ++i) lq[i * lqs] = rq[i];
#pragma unroll
            for (int i = 0; i < NK; ++i) lk[i * lks] = rk[i];
            if (c0 + DM_CKF < C) load(c0 + DM_CKF);    // next chunk in flight during the MFMAs
            __syncthreads();
            if (normalize) {
#pragma unroll
                for (int c = 0; c < DM_CKF; ++c) { const float v = L.t.f.k[c * DM_PK + tid]; nsq = fmaf(v, v, nsq); }
            }
            if (y_base + 64 * wid < P) {
                const float* pq = L.t.f.q + h * DM_PQ + r;
                const float* pk = L.t.f.k + h * DM_PK + 64 * wid + r;
#pragma unroll
                for (int t = 0; t < DM_CKF / 2; ++t) {
                    const float b = pq[2 * t * DM_PQ];
                    acc[0] = mfma32(pk[2 * t * DM_PK], b, acc[0]);
                    acc[1] = mfma32(pk[2 * t * DM_PK + 32], b, acc[1]);
                }
            }
        }
        __syncthreads();
        L.n2[tid] = nsq;
        __syncthreads();
        dm_round_epilogue(acc, L.n2, normalize, y_base, wid, h, P, bv, by);
    }
    dm_merge(L, bv, by, tid, out);
}

// ---- bf16 operands, channels-last (sc == 1), C % 64 == 0, 16-byte aligned rows.
// The chunk loop is latency-bound unless the loads run well ahead (8 MFMAs = 256 cycles of work per chunk and wave against
// an L2 round trip of ~1 us under load: 47 us per launch with the next chunk's loads issued one chunk ahead): two register
// sets, the loads of chunk c + 2 are issued as soon as chunk c has been stored to LDS.
struct DmRegs { dm_u32x4 q; dm_u32x4 k[8]; };

__device__ __forceinline__ void dm_argmax_bf16(DmLds& L, const unsigned short* __restrict__ q, int64_t q_sp, const unsigned short* __restrict__ k,
                               int64_t k_sp, int C, int P, int x0, bool normalize, int tid, int* __restrict__ out) {
    const int lane = tid & 63, wid = tid >> 6, r = lane & 31, h = lane >> 5;
    const int row = tid >> 3, seg = tid & 7;      // staging: 8 lanes x 16 B = one pixel's 64-channel chunk
    float bv = -INFINITY; int by = 0;
    const dm_u32x4 zero = {0, 0, 0, 0};
    for (int y_base = 0; y_base < P; y_base += DM_Y) {
        f32x16 acc[2] = {{0}, {0}};
        float nsq[8] = {0, 0, 0, 0, 0, 0, 0, 0};
        auto load = [&](DmRegs& rg, int c0) __attribute__((always_inline)) {
            rg.q = (x0 + row < P) ? *reinterpret_cast<const dm_u32x4*>(q + (int64_t)(x0 + row) * q_sp + c0 + seg * 8) : zero;
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const int y = y_base + row + 32 * i;
                rg.k[i] = (y < P) ? *reinterpret_cast<const dm_u32x4*>(k + (int64_t)y * k_sp + c0 + seg * 8) : zero;
            }
        };
        auto stage = [&](DmRegs& rg, int c0) __attribute__((always_inline)) {       // registers -> LDS (+ key norms), refill the set two chunks ahead, multiply
            __syncthreads();                         // the previous chunk's MFMAs have read the tiles
            *reinterpret_cast<dm_u32x4*>(L.t.b.q + row * DM_PB + seg * 8) = rg.q;
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                *reinterpret_cast<dm_u32x4*>(L.t.b.k + (row + 32 * i) * DM_PB + seg * 8) = rg.k[i];
                if (normalize) {
#pragma unroll
                    for (int w = 0; w < 4; ++w) {
                        const float lo = __uint_as_float(rg.k[i][w] << 16), hi = __uint_as_float(rg.k[i][w] & 0xFFFF0000u);
                        nsq[i] = fmaf(lo, lo, nsq[i]);
                        nsq[i] = fmaf(hi, hi, nsq[i]);
                    }
                }
            }
            if (c0 + 2 * DM_CKB < C) load(rg, c0 + 2 * DM_CKB);
            __syncthreads();
            if (y_base + 64 * wid < P) {
                const __bf16* pq = L.t.b.q + r * DM_PB + 8 * h;
                const __bf16* pk = L.t.b.k + (64 * wid + r) * DM_PB + 8 * h;
#pragma unroll
                for (int ks = 0; ks < DM_CKB / 16; ++ks) {
                    const dm_bf16x8 b = *reinterpret_cast<const dm_bf16x8*>(pq + 16 * ks);
                    const dm_bf16x8 a0 = *reinterpret_cast<const dm_bf16x8*>(pk + 16 * ks);
                    const dm_bf16x8 a1 = *reinterpret_cast<const dm_bf16x8*>(pk + 32 * DM_PB + 16 * ks);
                    acc[0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a0, b, acc[0], 0, 0, 0);
                    acc[1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1, b, acc[1], 0, 0, 0);
                }
            }
        };
        DmRegs ra, rb;
        load(ra, 0);
        if (DM_CKB < C) load(rb, DM_CKB);
        for (int c0 = 0; c0 < C; c0 += 2 * DM_CKB) {
            stage(ra, c0);
            if (c0 + DM_CKB < C) stage(rb, c0 + DM_CKB);
        }
        __syncthreads();
        if (normalize) {
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                float v = nsq[i];
                v += __shfl_xor(v, 1, 64); v += __shfl_xor(v, 2, 64); v += __shfl_xor(v, 4, 64);
                if (seg == 0) L.n2[row + 32 * i] = v;
            }
        }
        __syncthreads();
        dm_round_epilogue(acc, L.n2, normalize, y_base, wid, h, P, bv, by);
    }
    dm_merge(L, bv, by, tid, out);
}

// METRICS instantiations carry the second (fp32) arg-max routine.  One workgroup per CU (B x ceil(P / 32) work items: 224 at
// 32 x 196 pixels, fewer than the chip has CUs): the register budget goes to the two load sets in flight, not to occupancy.
template <bool BF, bool METRICS>
__global__ __launch_bounds__(256) void densecl_match_kernel(MatchArgs a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char dm_smem[];
    DmLds& L = *reinterpret_cast<DmLds*>(dm_smem);
    const int tid = threadIdx.x;
    const int P = a.P;
    // the x tiles of one sample share that sample's key map: contiguous runs of (sample, tile) items per XCD
    const int nitems = gridDim.x, id = blockIdx.x;
    const int q8 = nitems / 8, rem = nitems % 8, xcd = id % 8, jj = id / 8;
    const int item = (xcd < rem ? xcd * (q8 + 1) : rem * (q8 + 1) + (xcd - rem) * q8) + jj;
    const int n = item / a.XT, x0 = (item % a.XT) * DM_X;
    const int64_t nk = a.k_row ? a.k_row[n] : n;
    const float* ql = a.ql + (int64_t)n * CH * P;
    const float* kl = a.kl + nk * CH * P;

    // ---- phase A: arg-max of the backbone similarity
    if (BF)
        dm_argmax_bf16(L, static_cast<const unsigned short*>(a.qe) + n * a.qe_sn, a.qe_sp,
                       static_cast<const unsigned short*>(a.ke) + nk * a.ke_sn, a.ke_sp, a.CE, P, x0, a.normalize_k != 0, tid, L.best);
    else
        dm_argmax_f32(L, static_cast<const float*>(a.qe) + n * a.qe_sn, a.qe_sc, a.qe_sp,
                      static_cast<const float*>(a.ke) + nk * a.ke_sn, a.ke_sc, a.ke_sp, a.CE, P, x0, a.normalize_k != 0, tid, L.best);
    // ---- phase M (logging only): arg-max of the LOCAL similarity, same routine on the unit projection vectors
    if (METRICS) dm_argmax_f32(L, ql, P, 1, kl, P, 1, CH, P, x0, false, tid, L.simbest);

    // ---- phase B: the positive score, the coordinate mix and d pos / d q_local
    const bool scan = a.ids_q != nullptr && (a.lmbd > 0.f || METRICS);
    if (scan)
        for (int y = tid; y < P; y += 256) L.idk[y] = a.ids_k[(int64_t)n * P + y];
    if (tid < 2) L.cnt[tid] = 0;
    __syncthreads();
    const int xl = tid & 31, g = tid >> 5, x = x0 + xl;
    const bool x_ok = x < P;
    const int yb = L.best[xl];
    float qv[16], kb[16], S[16];
    float pos_part = 0.f, coord_part = 0.f;
    int first = 0x7fffffff;
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        qv[i] = x_ok ? ql[(int64_t)(16 * g + i) * P + x] : 0.f;
        kb[i] = x_ok ? kl[(int64_t)(16 * g + i) * P + yb] : 0.f;
        pos_part = fmaf(qv[i], kb[i], pos_part);
        S[i] = 0.f;
    }
    if (scan) {
        // the eight threads of a query pixel share the search for key pixels with its id (an eighth of the key map each),
        // collect them in a short LDS list, and one of them puts the list in y order (the sums below must not depend on
        // which thread found a match first)
        if (tid < DM_X) L.nm[tid] = 0;
        __syncthreads();
        const int64_t idq = x_ok ? a.ids_q[(int64_t)n * P + x] : 0;
        const int per = (P + 7) / 8;
        if (x_ok)
            for (int y = g * per; y < min(P, (g + 1) * per); ++y)
                if (L.idk[y] == idq) {
                    const int at = atomicAdd(&L.nm[xl], 1);
                    if (at < DM_ML) L.ml[xl * DM_ML + at] = y;
                }
        __syncthreads();
        const int nm = L.nm[xl];
        if (g == 0 && nm > 1 && nm <= DM_ML) {
            int* lst = L.ml + xl * DM_ML;
            for (int i = 1; i < nm; ++i) {
                const int v = lst[i];
                int j = i - 1;
                for (; j >= 0 && lst[j] > v; --j) lst[j + 1] = lst[j];
                lst[j + 1] = v;
            }
        }
        __syncthreads();
        auto take = [&](int y) __attribute__((always_inline)) {
            first = min(first, y);
            float d = 0.f;
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                const float kv = kl[(int64_t)(16 * g + i) * P + y];
                d = fmaf(qv[i], kv, d);
                S[i] += kv;
            }
            coord_part += d;          // builder.py:846-848: sum over the matching key pixels, in y order
        };
        if (x_ok && nm <= DM_ML) {
            for (int m = 0; m < nm; ++m) take(L.ml[xl * DM_ML + m]);
        } else if (x_ok) {            // an id repeated more than DM_ML times among the keys: this pixel walks the whole list
            for (int y = 0; y < P; ++y)
                if (L.idk[y] == idq) take(y);
        }
    }
    L.red[0][g * DM_X + xl] = pos_part;
    L.red[1][g * DM_X + xl] = coord_part;
    if (g == 0) L.first[xl] = first;
    __syncthreads();
    const bool overlap = L.first[xl] != 0x7fffffff;
    const bool mix = overlap && a.lmbd > 0.f;
    if (tid < DM_X && x_ok) {
        float p = 0.f, c = 0.f;
#pragma unroll
        for (int gg = 0; gg < 8; ++gg) { p += L.red[0][gg * DM_X + tid]; c += L.red[1][gg * DM_X + tid]; }
        if (mix) p = p * a.one_minus_lmbd + c * a.lmbd;          // builder.py:852-855
        a.pos[(int64_t)n * P + x] = p;
        a.best[(int64_t)n * P + x] = yb;
        if (METRICS && overlap) {
            atomicAdd(&L.cnt[0], 1);
            if (L.simbest[tid] == L.first[tid]) atomicAdd(&L.cnt[1], 1);   // corr row's arg-max = its first match
        }
    }
    if (a.kvec && x_ok) {
#pragma unroll
        for (int i = 0; i < 16; ++i)
            a.kvec[((int64_t)n * CH + 16 * g + i) * P + x] = mix ? kb[i] * a.one_minus_lmbd + S[i] * a.lmbd : kb[i];
    }
    if (a.counts) {
        __syncthreads();
        if (tid < 2) a.counts[2 * item + tid] = L.cnt[tid];
    }
}

CP2_API int cp2_densecl_match(const void* q_embed, const void* k_embed, int embed_bf16, int64_t qe_sn, int64_t qe_sc, int64_t qe_sp,
                              int64_t ke_sn, int64_t ke_sc, int64_t ke_sp, const int64_t* k_row, const float* q_local,
                              const float* k_local, const int64_t* ids_q, const int64_t* ids_k, float lmbd_coordinate,
                              float one_minus_lmbd, int normalize_k, int want_metrics, int32_t* best_idx, float* pos, float* kvec,
                              int32_t* counts, int B, int CE, int CL, int P, void* stream) {
    if (!q_embed || !k_embed || !q_local || !k_local || !best_idx || !pos) return CP2_ERR_NULL;
    if ((ids_q == nullptr) != (ids_k == nullptr)) return CP2_ERR_NULL;
    if (want_metrics && (!ids_q || !counts)) return CP2_ERR_NULL;
    if (B <= 0 || CE <= 0 || P <= 0 || !(lmbd_coordinate >= 0.f && lmbd_coordinate <= 1.f)) return CP2_ERR_SHAPE;
    if (CL != CH || P > DM_MAXP) return CP2_ERR_UNSUPPORTED;
    if (embed_bf16) {
        if (qe_sc != 1 || ke_sc != 1 || CE % DM_CKB) return CP2_ERR_UNSUPPORTED;     // channels-last rows of whole chunks
        if (!cp2_aligned16(q_embed) || !cp2_aligned16(k_embed) || qe_sp % 8 || ke_sp % 8 || qe_sn % 8 || ke_sn % 8) return CP2_ERR_ALIGN;
    }
    const int XT = cp2_cdiv(P, DM_X);
    MatchArgs a{q_embed, k_embed, qe_sn, qe_sc, qe_sp, ke_sn, ke_sc, ke_sp, k_row, q_local, k_local, ids_q, ids_k,
                lmbd_coordinate, one_minus_lmbd, normalize_k, want_metrics, best_idx, pos, kvec, counts, CE, P, XT};
    const dim3 grid(B * XT), block(256);
    const size_t lds = sizeof(DmLds);
    auto kfn = embed_bf16 ? (want_metrics ? densecl_match_kernel<true, true> : densecl_match_kernel<true, false>)
                          : (want_metrics ? densecl_match_kernel<false, true> : densecl_match_kernel<false, false>);
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kfn), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return (int)e;
    CP2_LAUNCH_PROFILED(kfn, grid, block, lds, cp2_stream(stream), a);
    return cp2_launch_status();
}
