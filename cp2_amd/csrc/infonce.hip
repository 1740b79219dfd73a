// a8-a10, a16: the contrastive logit kernels, fp32 on the matrix cores.
//
// All three contractions of the reference are over the C=128 channel axis of
// unit-norm feature vectors stored channel-major ([C][n], the reference's own NCHW /
// [C,K] queue layout), so both MFMA operands are read without any transpose:
//   v_mfma_f32_32x32x2_f32:  D[i][j] += A[i][k] B[k][j],  lane l: A[i=l&31][k=l>>5], B[k=l>>5][j=l&31]
// f32-input MFMA is an exact fp32 fmaf chain (parity with the fp32 reference to ~1e-6)
// at 1/16 of the bf16 rate; see DESIGN.md for the split-bf16 plan.
//
// "Transposed flash" orientation: the 32 OWNER vectors of a wave (the side whose
// softmax statistics are kept) sit in registers as the B operand, one owner per lane
// column; the OTHER side streams through LDS as the A operand.  The accumulator then
// holds S^T[other][owner] with the owner on the lane, so max / sum-exp / rescale are
// lane-local, and the accumulator registers are directly the B operand of the second
// product  U^T[c][owner] += sum_other X[c][other] * P[other][owner]  (no LDS round trip).
//
//   rowkey_fwd   owners = rows (pooled q vectors, or DenseCL pixels), others = queue keys
//                builder.py:1395-1428 (instance InfoNCE), :866-873,906-908 (DenseCL local)
//   dense_fwd    owners = key pixels y, others = query pixels x of the same sample
//                builder.py:1289-1292,1392,1431-1437 (column-wise log-softmax, dim=1)
//   dense_bwd    owners = query pixels x, others = key pixels y: d loss / d q_dense
#include "infonce_common.hpp"
#include "rowkey_small_fin.hpp"
#include "dense_post.hpp"

// LDS tile T[c][j], j < KT, row pitch KT+1 floats (odd pitch: both the row read of
// product 1 and the column read of product 2 are bank-conflict free).  The fill is split
// in two so the global loads of the NEXT tile are in flight while the current one is used:
// tile_load (global -> registers, 16 bytes per lane when aligned) ... tile_store (-> LDS).
template <int KT, int NT = 256>
struct TileRegs {
    float4 v[CH * KT / 4 / NT];
};

template <int KT, int NT = 256>
__device__ __forceinline__ void tile_load(TileRegs<KT, NT>& rg, const float* __restrict__ src, int64_t ld, int j0, int jmax,
                                          int tid, bool vec_ok) {
    constexpr int Q = KT / 4;  // float4 per tile row
#pragma unroll
    for (int i = 0; i < CH * KT / 4 / NT; ++i) {
        const int e = tid + i * NT, c = e / Q, j = (e % Q) * 4;
        const float* p = src + (int64_t)c * ld + j0 + j;
        if (vec_ok && j0 + j + 3 < jmax) {
            rg.v[i] = *reinterpret_cast<const float4*>(p);
        } else {
            rg.v[i].x = (j0 + j + 0 < jmax) ? p[0] : 0.f;
            rg.v[i].y = (j0 + j + 1 < jmax) ? p[1] : 0.f;
            rg.v[i].z = (j0 + j + 2 < jmax) ? p[2] : 0.f;
            rg.v[i].w = (j0 + j + 3 < jmax) ? p[3] : 0.f;
        }
    }
}

template <int KT, int NT = 256>
__device__ __forceinline__ void tile_store(float* __restrict__ T, const TileRegs<KT, NT>& rg, int tid) {
    constexpr int Q = KT / 4, KP = KT + 1;
#pragma unroll
    for (int i = 0; i < CH * KT / 4 / NT; ++i) {
        const int e = tid + i * NT, c = e / Q, j = (e % Q) * 4;
        float* d = T + c * KP + j;
        d[0] = rg.v[i].x; d[1] = rg.v[i].y; d[2] = rg.v[i].z; d[3] = rg.v[i].w;
    }
}

template <int KT, int NT = 256>
__device__ __forceinline__ void fill_tile(float* __restrict__ T, const float* __restrict__ src, int64_t ld, int j0,
                                          int jmax, int tid) {
    TileRegs<KT, NT> rg;
    const bool vec_ok = (ld % 4 == 0) && ((reinterpret_cast<uintptr_t>(src) & 15u) == 0) && (j0 % 4 == 0);
    tile_load<KT, NT>(rg, src, ld, j0, jmax, tid, vec_ok);
    tile_store<KT, NT>(T, rg, tid);
}

// S^T[other = kk + rho(reg,h)][owner = lane&31] for one 32x32 sub-tile.
// The A operands are read from LDS one group of PF values ahead of the MFMAs that consume them, so the
// ~100-cycle ds_read latency hides behind the previous group's 64-cycle MFMAs instead of stalling each pair.
constexpr int PF = 8;
template <int KP>
__device__ __forceinline__ f32x16 product1(const float* __restrict__ T, int kk, const float (&bq)[CH / 2], int r, int h) {
    f32x16 acc = {0};
    const float* p = T + h * KP + kk + r;
    float cur[PF], nxt[PF];
#pragma unroll
    for (int j = 0; j < PF; ++j) cur[j] = p[(2 * j) * KP];
#pragma unroll
    for (int g = 0; g < CH / 2 / PF; ++g) {
        if (g + 1 < CH / 2 / PF) {
#pragma unroll
            for (int j = 0; j < PF; ++j) nxt[j] = p[(2 * ((g + 1) * PF + j)) * KP];
        }
#pragma unroll
        for (int j = 0; j < PF; ++j) acc = mfma32(cur[j], bq[g * PF + j], acc);
#pragma unroll
        for (int j = 0; j < PF; ++j) cur[j] = nxt[j];
    }
    return acc;
}
// U^T[c = cb*32 + rho(reg',h)][owner] += sum_other T[c][other] * p[other][owner]
template <int KP>
__device__ __forceinline__ void product2(const float* __restrict__ T, int kk, const float (&p)[16], f32x16 (&U)[4], int r,
                                         int h) {
    const float* base = T + r * KP + kk + 4 * h;
    float cur[16], nxt[16];
#pragma unroll
    for (int reg = 0; reg < 16; ++reg) cur[reg] = base[rho(reg, 0)];
#pragma unroll
    for (int cb = 0; cb < 4; ++cb) {
        if (cb + 1 < 4) {
#pragma unroll
            for (int reg = 0; reg < 16; ++reg) nxt[reg] = base[(cb + 1) * 32 * KP + rho(reg, 0)];
        }
#pragma unroll
        for (int reg = 0; reg < 16; ++reg) U[cb] = mfma32(cur[reg], p[reg], U[cb]);
#pragma unroll
        for (int reg = 0; reg < 16; ++reg) cur[reg] = nxt[reg];
    }
}

// ===========================================================================
// rows-vs-queue InfoNCE, forward with fused gradient accumulation
// ===========================================================================

template <int WR, int WK, bool WITH_U>
__global__ __launch_bounds__(256, WK == 1 ? 2 : 1) void rowkey_fwd_kernel(RowKeyArgs a) {
    // WK == 1 (many row tiles: DenseCL rows): two waves per SIMD, so one workgroup's softmax / barrier / tile
    // store overlaps the other's MFMA chain: 80.7 -> 108.8 TFLOP/s at 6272 x 65536 (9 spilled VGPRs).
    constexpr int NSUB = (WK == 1) ? 2 : 1;
    constexpr int KT = 32 * WK * NSUB, KP = KT + 1;
    extern __shared__ __attribute__((aligned(16))) float T[];
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int wr = wid / WK, wk = wid % WK, r = lane & 31, h = lane >> 5;
    const int row0 = (blockIdx.x * WR + wr) * 32, row = row0 + r;
    const bool row_ok = row < a.R;
    float bq[CH / 2];
    {
        const int rr = row_ok ? row : 0;
        const float* base = a.rows + (int64_t)(rr / a.RP) * a.r_sn + (int64_t)(rr % a.RP) * a.r_sx;
#pragma unroll
        for (int t = 0; t < CH / 2; ++t) bq[t] = row_ok ? base[(int64_t)(2 * t + h) * a.r_sc] : 0.f;
    }
    const float pos_s = (row_ok && a.NE > 0) ? a.extras[(int64_t)row * a.NE] * a.inv_t : INFINITY;
    float m_run = -INFINITY, s_run = 0.f;
    int cnt = 0;
    f32x16 U[4];
#pragma unroll
    for (int cb = 0; cb < 4; ++cb) U[cb] = (f32x16){0};
    const int k_begin = blockIdx.y * a.keys_per_split;
    const int k_end = min(a.K, k_begin + a.keys_per_split);
    const bool vec_ok = (a.K % 4 == 0) && ((reinterpret_cast<uintptr_t>(a.keys) & 15u) == 0);
    TileRegs<KT> rg;
    tile_load<KT>(rg, a.keys, a.K, k_begin, k_end, tid, vec_ok);
    for (int k0 = k_begin; k0 < k_end; k0 += KT) {
        __syncthreads();
        tile_store<KT>(T, rg, tid);
        __syncthreads();
        if (k0 + KT < k_end) tile_load<KT>(rg, a.keys, a.K, k0 + KT, k_end, tid, vec_ok);  // next tile, in flight during the MFMAs
#pragma unroll
        for (int s = 0; s < NSUB; ++s) {
            const int kk = (wk + WK * s) * 32;
            if (k0 + kk >= k_end) continue;  // wave-uniform
            const f32x16 acc = product1<KP>(T, kk, bq, r, h);
            float sv[16];
            float tmax = -INFINITY;
#pragma unroll
            for (int reg = 0; reg < 16; ++reg) {
                const int key = k0 + kk + rho(reg, h);
                const bool valid = key < k_end;
                if (a.lnegT && valid && row_ok) a.lnegT[(int64_t)key * a.ln_sk + (int64_t)row * a.ln_sr] = acc[reg];
                sv[reg] = valid ? acc[reg] * a.inv_t : -INFINITY;
                tmax = fmaxf(tmax, sv[reg]);
                cnt += (sv[reg] > pos_s) ? 1 : 0;
            }
            tmax = fmaxf(tmax, __shfl_xor(tmax, 32, 64));
            if (__any(tmax > m_run)) {
                const float m_new = fmaxf(m_run, tmax);
                const float sc = __expf(m_run - m_new);  // exp(-inf) = 0 on the first tile
                s_run *= sc;
                if (WITH_U) {
#pragma unroll
                    for (int cb = 0; cb < 4; ++cb) U[cb] *= sc;
                }
                m_run = m_new;
            }
            float p[16];
#pragma unroll
            for (int reg = 0; reg < 16; ++reg) {
                p[reg] = __expf(sv[reg] - m_run);
                s_run += p[reg];
            }
            if (WITH_U) product2<KP>(T, kk, p, U, r, h);
        }
    }
    const float s_tot = s_run + __shfl_xor(s_run, 32, 64);
    const int cnt_tot = cnt + __shfl_xor(cnt, 32, 64);
    const int slot = blockIdx.y;
    if constexpr (WK == 1) {
        if (row_ok) {
            if (h == 0) {
                a.part_m[(int64_t)slot * a.R + row] = m_run;
                a.part_s[(int64_t)slot * a.R + row] = s_tot;
                a.part_cnt[(int64_t)slot * a.R + row] = cnt_tot;
            }
            if (WITH_U) {
#pragma unroll
                for (int cb = 0; cb < 4; ++cb)
#pragma unroll
                    for (int reg = 0; reg < 16; ++reg)
                        a.part_U[((int64_t)slot * CH + cb * 32 + rho(reg, h)) * a.R + row] = U[cb][reg];
            }
        }
    } else {
        // merge the WK key-waves that share a row tile through LDS, then one coalesced write
        __syncthreads();
        float* Ubuf = T;                                  // [WR*WK][CH][32]
        float* mbuf = T + WR * WK * CH * 32;              // [WR*WK][32]
        float* sbuf = mbuf + WR * WK * 32;
        int* cbuf = reinterpret_cast<int*>(sbuf + WR * WK * 32);
        if (h == 0) mbuf[wid * 32 + r] = m_run;
        __syncthreads();
        float M = -INFINITY;
#pragma unroll
        for (int j = 0; j < WK; ++j) M = fmaxf(M, mbuf[(wr * WK + j) * 32 + r]);
        const float f = (m_run == -INFINITY) ? 0.f : __expf(m_run - M);
        if (h == 0) { sbuf[wid * 32 + r] = s_tot * f; cbuf[wid * 32 + r] = cnt_tot; }
        if (WITH_U) {
#pragma unroll
            for (int cb = 0; cb < 4; ++cb)
#pragma unroll
                for (int reg = 0; reg < 16; ++reg) Ubuf[(wid * CH + cb * 32 + rho(reg, h)) * 32 + r] = U[cb][reg] * f;
        }
        __syncthreads();
        for (int e = tid; e < WR * 32; e += 256) {
            const int w2 = e / 32, rr = e % 32, grow = (blockIdx.x * WR + w2) * 32 + rr;
            if (grow < a.R) {
                float mm = -INFINITY, ss = 0.f;
                int cc = 0;
                for (int j = 0; j < WK; ++j) {
                    mm = fmaxf(mm, mbuf[(w2 * WK + j) * 32 + rr]);
                    ss += sbuf[(w2 * WK + j) * 32 + rr];
                    cc += cbuf[(w2 * WK + j) * 32 + rr];
                }
                a.part_m[(int64_t)slot * a.R + grow] = mm;
                a.part_s[(int64_t)slot * a.R + grow] = ss;
                a.part_cnt[(int64_t)slot * a.R + grow] = cc;
            }
        }
        if (WITH_U) {
            for (int e = tid; e < WR * CH * 32; e += 256) {
                const int rr = e % 32, c = (e / 32) % CH, w2 = e / (32 * CH);
                const int grow = (blockIdx.x * WR + w2) * 32 + rr;
                if (grow < a.R) {
                    float u = 0.f;
                    for (int j = 0; j < WK; ++j) u += Ubuf[((w2 * WK + j) * CH + c) * 32 + rr];
                    a.part_U[((int64_t)slot * CH + c) * a.R + grow] = u;
                }
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------------------------
// Split-bf16 form for the MFMA-bound case (thousands of rows, config 5): rowkey_bf16x3.hip.  Every fp32 operand x is used
// as hi + lo with hi = bf16(x), lo = bf16(x - hi) and every product as hi*hi + hi*lo + lo*hi on v_mfma_f32_32x32x16_bf16
// (fp32 accumulate): 3 MFMAs of 32 cycles replace 8 f32 MFMAs of 64 cycles per 32x32x16 block at <= 3*2^-18 relative error
// per term, i.e. <= 1.2e-5 on logits of unit vectors (the reference bound is 1e-4).  Here: the prep kernel of its key operand.
// ---------------------------------------------------------------------------------------------------------------
__device__ __forceinline__ void split_bf(float v, __bf16& hi, __bf16& lo) {
    hi = (__bf16)v;
    lo = (__bf16)(v - (float)hi);
}

// Pre-split form of the key operand for the bf16x3 kernel: the fp32 queue [C][K] is converted ONCE per call into
// four bf16 arrays -- hi/lo, key-major [K][C] (product 1) and channel-major [C][K] (product 2) -- so the main kernel
// fills its LDS images with 16-byte copies.  (Doing the conversion inside the main kernel repeated it for every one
// of the 49 row blocks and built the key-major image with 2-byte LDS writes that were 8-way bank conflicted: PMC
// SQ_LDS_BANK_CONFLICT was 60 % of SQ_LDS_IDX_ACTIVE.)   split = [qT_hi | qT_lo | q_hi | q_lo], each C*K bf16.
__global__ __launch_bounds__(256) void keys_split_kernel(const float* __restrict__ keys, int K, __bf16* __restrict__ split) {
    __shared__ float tile[64][CH + 1];                    // [key][c]
    const int k0 = blockIdx.x * 64, tid = threadIdx.x;
    const int64_t CK = (int64_t)CH * K;
    __bf16 *qth = split, *qtl = split + CK, *qh = split + 2 * CK, *ql = split + 3 * CK;
    for (int e = tid; e < CH * 64; e += 256) {            // coalesced along keys
        const int c = e / 64, j = e % 64;
        const float v = (k0 + j < K) ? keys[(int64_t)c * K + k0 + j] : 0.f;
        tile[j][c] = v;
        if (k0 + j < K) {
            __bf16 hi, lo;
            split_bf(v, hi, lo);
            // channel-major image: every 16-key block in the order [0-3, 8-11, 4-7, 12-15] (bits 2 and 3 of the key index
            // swapped) = the row order of the 32x32 accumulator, so 8 consecutive stored keys are one MFMA fragment
            const int js = (j & ~12) | ((j & 4) << 1) | ((j & 8) >> 1);
            qh[(int64_t)c * K + k0 + js] = hi;
            ql[(int64_t)c * K + k0 + js] = lo;
        }
    }
    __syncthreads();
    for (int e = tid; e < CH * 64; e += 256) {            // coalesced along channels
        const int j = e / CH, c = e % CH;
        if (k0 + j < K) {
            __bf16 hi, lo;
            split_bf(tile[j][c], hi, lo);
            qth[(int64_t)(k0 + j) * CH + c] = hi;
            qtl[(int64_t)(k0 + j) * CH + c] = lo;
        }
    }
}

// Merge the per-split partials: lse, per-row loss, count of negatives above the
// positive, d loss / d row (in the rows' own layout) and d loss / d extra logit.

constexpr int FIN_CPW = 4;  // channels per wave in the finalize kernel: grid.y = CH / (4 * FIN_CPW) channel groups

__global__ __launch_bounds__(256) void rowkey_finalize_kernel(RowKeyFinArgs a) {
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int row = blockIdx.x * 64 + lane;
    if (row >= a.R) return;
    float M = -INFINITY;
    for (int s = 0; s < a.S; ++s) M = fmaxf(M, a.part_m[(int64_t)s * a.R + row]);
    float e[4] = {0.f, 0.f, 0.f, 0.f};
    for (int j = 0; j < a.NE; ++j) { e[j] = a.extras[(int64_t)row * a.NE + j] * a.inv_t; M = fmaxf(M, e[j]); }
    float Z = 0.f;
    for (int s = 0; s < a.S; ++s) {
        const float ms = a.part_m[(int64_t)s * a.R + row];
        if (ms != -INFINITY) Z += a.part_s[(int64_t)s * a.R + row] * expf(ms - M);
    }
    for (int j = 0; j < a.NE; ++j) Z += expf(e[j] - M);
    const float lse = M + logf(Z);
    if (w == 0 && blockIdx.y == 0) {
        int cnt = 0;
        for (int s = 0; s < a.S; ++s) cnt += a.part_cnt[(int64_t)s * a.R + row];
        for (int j = 1; j < a.NE; ++j) cnt += (e[j] > e[0]) ? 1 : 0;  // extra negatives also rank against the positive
        a.lse[row] = lse;
        a.loss_rows[row] = lse - e[0];
        a.cnt_gt[row] = cnt;
        if (a.dE)
            for (int j = 0; j < a.NE; ++j)
                a.dE[(int64_t)row * a.NE + j] = a.grad_scale * a.inv_t * (expf(e[j] - lse) - (j == 0 ? 1.f : 0.f));
    }
    if (!a.drows) return;
    const int c0 = (blockIdx.y * 4 + w) * FIN_CPW;
    float acc[FIN_CPW];
#pragma unroll
    for (int i = 0; i < FIN_CPW; ++i) acc[i] = 0.f;
#pragma unroll 4
    for (int s = 0; s < a.S; ++s) {
        const float ms = a.part_m[(int64_t)s * a.R + row];
        const float ws = (ms == -INFINITY) ? 0.f : expf(ms - lse);
        const float* u = a.part_U + ((int64_t)s * CH + c0) * a.R + row;
#pragma unroll
        for (int i = 0; i < FIN_CPW; ++i) acc[i] += u[(int64_t)i * a.R] * ws;
    }
    float* d = a.drows + (int64_t)(row / a.RP) * a.d_sn + (int64_t)(row % a.RP) * a.d_sx;
    const float gs = a.grad_scale * a.inv_t;
#pragma unroll
    for (int i = 0; i < FIN_CPW; ++i) d[(int64_t)(c0 + i) * a.d_sc] = acc[i] * gs;
}

// Many-splits form (instance loss: R = batch size, S = 256 splits): a serial loop over S per row is pure
// load latency (measured 204-236 us), so here FS_SL "split lanes" share each row: thread = (row r of 32, lane sl),
// each lane reduces the splits s = sl, sl+FS_SL, ... and the partial results meet in LDS.  grid = (R/32, CH/8).
// (32 split lanes: with 8, the three dependent passes over S = 256 partials were 32 serial load rounds each -- 33 us for
// the 4 MB of partials of the instance loss.)
constexpr int FS_SL = 32, FS_CPB = 8;

__global__ __launch_bounds__(32 * FS_SL) void rowkey_finalize_spar_kernel(RowKeyFinArgs a) {
    __shared__ float red[FS_SL][32];
    __shared__ int redi[FS_SL][32];
    __shared__ float red2[FS_CPB][FS_SL][32];
    const int r = threadIdx.x & 31, sl = threadIdx.x >> 5;
    const int row = blockIdx.x * 32 + r;
    const bool ok = row < a.R;
    const int rr = ok ? row : 0;
    float M = -INFINITY;
    for (int s = sl; s < a.S; s += FS_SL) M = fmaxf(M, a.part_m[(int64_t)s * a.R + rr]);
    red[sl][r] = M;
    __syncthreads();
#pragma unroll
    for (int j = 0; j < FS_SL; ++j) M = fmaxf(M, red[j][r]);
    float e[4] = {0.f, 0.f, 0.f, 0.f};
    for (int j = 0; j < a.NE; ++j) { e[j] = a.extras[(int64_t)rr * a.NE + j] * a.inv_t; M = fmaxf(M, e[j]); }
    __syncthreads();
    float z = 0.f;
    int cnt = 0;
    for (int s = sl; s < a.S; s += FS_SL) {
        const float ms = a.part_m[(int64_t)s * a.R + rr];
        if (ms != -INFINITY) z += a.part_s[(int64_t)s * a.R + rr] * expf(ms - M);
        cnt += a.part_cnt[(int64_t)s * a.R + rr];
    }
    red[sl][r] = z;
    redi[sl][r] = cnt;
    __syncthreads();
    float Z = 0.f;
    cnt = 0;
#pragma unroll
    for (int j = 0; j < FS_SL; ++j) { Z += red[j][r]; cnt += redi[j][r]; }
    for (int j = 0; j < a.NE; ++j) Z += expf(e[j] - M);
    for (int j = 1; j < a.NE; ++j) cnt += (e[j] > e[0]) ? 1 : 0;
    const float lse = M + logf(Z);
    if (sl == 0 && blockIdx.y == 0 && ok) {
        a.lse[row] = lse;
        a.loss_rows[row] = lse - e[0];
        a.cnt_gt[row] = cnt;
        if (a.dE)
            for (int j = 0; j < a.NE; ++j)
                a.dE[(int64_t)row * a.NE + j] = a.grad_scale * a.inv_t * (expf(e[j] - lse) - (j == 0 ? 1.f : 0.f));
    }
    if (!a.drows) return;
    const int c0 = blockIdx.y * FS_CPB;
    float acc[FS_CPB];
#pragma unroll
    for (int i = 0; i < FS_CPB; ++i) acc[i] = 0.f;
#pragma unroll 2
    for (int s = sl; s < a.S; s += FS_SL) {
        const float ms = a.part_m[(int64_t)s * a.R + rr];
        const float ws = (ms == -INFINITY) ? 0.f : expf(ms - lse);
        const float* u = a.part_U + ((int64_t)s * CH + c0) * a.R + rr;
#pragma unroll
        for (int i = 0; i < FS_CPB; ++i) acc[i] += u[(int64_t)i * a.R] * ws;
    }
#pragma unroll
    for (int i = 0; i < FS_CPB; ++i) red2[i][sl][r] = acc[i];
    __syncthreads();
    if (sl >= FS_CPB) return;
    float tot = 0.f;                       // split lane sl finishes channel c0 + sl
#pragma unroll
    for (int j = 0; j < FS_SL; ++j) tot += red2[sl][j][r];
    if (ok) {
        float* d = a.drows + (int64_t)(row / a.RP) * a.d_sn + (int64_t)(row % a.RP) * a.d_sx;
        d[(int64_t)(c0 + sl) * a.d_sc] = tot * a.grad_scale * a.inv_t;
    }
}

// out[0] = mean(x[0..n)) (single workgroup: deterministic order)
__global__ __launch_bounds__(256) void mean_kernel(const float* __restrict__ x, int n, float* __restrict__ out) {
    __shared__ float red[4];
    float s = 0.f;
    for (int i = threadIdx.x; i < n; i += 256) s += x[i];
    s = wave_sum(s);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) out[0] = (red[0] + red[1] + red[2] + red[3]) / (float)n;
}

// One 32-row tile and 16-byte addressable key rows: the barrier-free LDS-DMA kernel of rowkey_small.hip
static bool rowkey_use_small(int R, int K) { return R <= 32 && K % 4 == 0 && K <= (1 << 21); }
// ... which also needs channel-contiguous, 16-byte aligned rows (else the register-staged kernels run with the same split count)
static bool rowkey_small_rows_ok(const float* rows, int64_t r_sn, int64_t r_sx, int64_t r_sc, const float* keys) {
    return r_sc == 1 && r_sn % 4 == 0 && r_sx % 4 == 0 && cp2_aligned16(rows) && cp2_aligned16(keys);
}

static int rowkey_shape(int R, int* WR, int* WK) {
    if (R <= 32) { *WR = 1; *WK = 4; }
    else if (R <= 64) { *WR = 2; *WK = 2; }
    else { *WR = 4; *WK = 1; }
    return 0;
}

CP2_API int cp2_rowkey_num_splits(int R, int K) {
    // Workgroups = row blocks x key splits.  Two workgroups fit a CU (about 200 VGPRs, 33-67 KB LDS), so the chip
    // holds 512 at once; a grid slightly above a multiple of 512 runs a nearly empty last wave (539 workgroups
    // took as long as 1024 would).  Pick the split count whose last wave is fullest, preferring 2-3 waves so that
    // uneven workgroups still balance, with at least 8 LDS tiles of keys per split (2 when there is one row block).
    if (R <= 0 || K <= 0) return CP2_ERR_SHAPE;
    if (rowkey_use_small(R, K)) return rowkey_small_num_splits(K, nullptr);
    int WR, WK;
    rowkey_shape(R, &WR, &WK);
    const int KT = 32 * WK * (WK == 1 ? 2 : 1);
    const int row_blocks = cp2_cdiv(R, 32 * WR);
    const int slots = 512;
    const int max_ns = cp2_cdiv(K, (row_blocks == 1 ? 2 : 8) * KT) < 1 ? 1 : cp2_cdiv(K, (row_blocks == 1 ? 2 : 8) * KT);
    int best = 1;
    double best_score = -1.0;
    for (int ns = 1; ns <= max_ns; ++ns) {
        const int64_t blocks = (int64_t)row_blocks * ns;
        const int64_t waves = (blocks + slots - 1) / slots;
        if (waves > 3) break;
        double eff = (double)blocks / (double)(waves * slots);
        if (blocks < slots) eff = (double)blocks / slots;             // a partly filled chip
        const double score = eff + 0.01 * (double)waves;  // ties: more waves balance better
        if (score > best_score) { best_score = score; best = ns; }
    }
    // every split must own at least one key after rounding the split length up to whole tiles
    while (best > 1 && (int64_t)(cp2_cdiv(cp2_cdiv(K, best), KT) * KT) * (best - 1) >= K) --best;
    return best;
}

CP2_API int cp2_rowkey_infonce_fwd(const float* rows, int RP, int64_t r_sn, int64_t r_sx, int64_t r_sc, int R,
                                   const float* keys, int K, const float* extras, int NE, float temperature,
                                   int nsplit, float* part_m, float* part_s, int32_t* part_cnt, float* part_U,
                                   float* lnegT, int lneg_row_major, int precision, void* keys_split, int C, void* stream) {
    if (!rows || !keys || !part_m || !part_s || !part_cnt) return CP2_ERR_NULL;
    if (NE > 0 && !extras) return CP2_ERR_NULL;
    if (R <= 0 || K <= 0 || RP <= 0 || nsplit <= 0 || NE < 0 || NE > 4 || !(temperature > 0.f)) return CP2_ERR_SHAPE;
    if (C != CH) return CP2_ERR_UNSUPPORTED;
    if (precision != 0 && precision != 1 && precision != 3) return CP2_ERR_SHAPE;
    const bool split_ready = precision == 3;      // keys_split already holds this queue's split (an earlier call wrote it)
    if (split_ready) precision = 1;
    const int64_t ln_sk = lneg_row_major ? 1 : R, ln_sr = lneg_row_major ? K : 1;
    if (rowkey_use_small(R, K) && rowkey_small_rows_ok(rows, r_sn, r_sx, r_sc, keys)) {
        RowKeyArgs sa{rows, RP, r_sn, r_sx, r_sc, R, keys, K, extras, NE, 1.0f / temperature, 0,
                      part_m, part_s, part_cnt, part_U, lnegT, ln_sk, ln_sr};
        return rowkey_small_launch(sa, nsplit, part_U != nullptr, cp2_stream(stream));
    }
    int WR, WK;
    rowkey_shape(R, &WR, &WK);
    const int KT = 32 * WK * (WK == 1 ? 2 : 1);
    int kps = cp2_cdiv(K, nsplit);
    kps = cp2_cdiv(kps, KT) * KT;
    if ((int64_t)kps * (nsplit - 1) >= K && nsplit > 1) return CP2_ERR_SHAPE;  // an empty split: caller must use cp2_rowkey_num_splits
    RowKeyArgs a{rows, RP, r_sn, r_sx, r_sc, R, keys, K, extras, NE, 1.0f / temperature, kps,
                 part_m, part_s, part_cnt, part_U, lnegT, ln_sk, ln_sr};
    size_t lds = (size_t)CH * (KT + 1) * sizeof(float);
    if (WK > 1) {
        const size_t merge = ((size_t)WR * WK * CH * 32 + 3 * (size_t)WR * WK * 32) * sizeof(float);
        if (merge > lds) lds = merge;
    }
    const dim3 grid(cp2_cdiv(R, 32 * WR), nsplit), block(256);
    const bool wu = part_U != nullptr;
    if (precision == 1 && WR == 4) {   // split-bf16 on the matrix cores (many-row case only; otherwise the f32 kernel)
        __bf16* ks = static_cast<__bf16*>(keys_split);
        const bool pre = ks != nullptr && K % 16 == 0 && K <= (1 << 20) && cp2_aligned16(ks);
        if (pre) {      // the queue's hi / lo split once per call, then the LDS-DMA double-buffered kernel (rowkey_bf16x3.hip)
            if (!split_ready) {
                hipLaunchKernelGGL(keys_split_kernel, dim3(cp2_cdiv(K, 64)), dim3(256), 0, cp2_stream(stream), keys, K, ks);
                int rc0 = cp2_launch_status();
                if (rc0) return rc0;
            }
            return rowkey_bf16x3_dma_launch(a, ks, grid, wu, cp2_stream(stream));
        }
        if (split_ready) return CP2_ERR_UNSUPPORTED;   // the caller's split cannot be used for this shape
        // (K % 16 != 0 or no workspace: the exact-fp32 kernel below serves the call)
    }
#define CP2_LAUNCH_RK(wr_, wk_, wu_)                                                                            \
    do {                                                                                                        \
        auto kfn = rowkey_fwd_kernel<wr_, wk_, wu_>;                                                            \
        hipError_t e_ = hipFuncSetAttribute(reinterpret_cast<const void*>(kfn),                                 \
                                            hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);              \
        if (e_ != hipSuccess) return (int)e_;                                                                   \
        CP2_LAUNCH_PROFILED(kfn, grid, block, lds, cp2_stream(stream), a);                                      \
    } while (0)
    if (WR == 1) { if (wu) CP2_LAUNCH_RK(1, 4, true); else CP2_LAUNCH_RK(1, 4, false); }
    else if (WR == 2) { if (wu) CP2_LAUNCH_RK(2, 2, true); else CP2_LAUNCH_RK(2, 2, false); }
    else { if (wu) CP2_LAUNCH_RK(4, 1, true); else CP2_LAUNCH_RK(4, 1, false); }
#undef CP2_LAUNCH_RK
    return cp2_launch_status();
}

CP2_API int cp2_rowkey_infonce_finalize(const float* part_m, const float* part_s, const int32_t* part_cnt,
                                        const float* part_U, int nsplit, const float* extras, int NE,
                                        float temperature, float grad_scale, int R, int RP, int64_t d_sn,
                                        int64_t d_sx, int64_t d_sc, float* lse, float* loss_rows, int32_t* cnt_gt,
                                        float* drows, float* dE, float* loss_mean, int C, void* stream) {
    if (!part_m || !part_s || !part_cnt || !lse || !loss_rows || !cnt_gt) return CP2_ERR_NULL;
    if (drows && !part_U) return CP2_ERR_NULL;
    if (NE > 0 && !extras) return CP2_ERR_NULL;
    if (R <= 0 || RP <= 0 || nsplit <= 0 || NE < 0 || NE > 4 || !(temperature > 0.f)) return CP2_ERR_SHAPE;
    if (C != CH) return CP2_ERR_UNSUPPORTED;
    RowKeyFinArgs a{part_m, part_s, part_cnt, part_U, nsplit, extras, NE, 1.0f / temperature, grad_scale,
                    R, RP, d_sn, d_sx, d_sc, lse, loss_rows, cnt_gt, drows, dE};
    if (R <= 32 && nsplit >= 16)      // one launch: merge, gradient, per-row outputs and the mean (rowkey_small.hip)
        return rowkey_small_finalize_launch(a, loss_mean, cp2_stream(stream));
    if (nsplit >= 16)
        hipLaunchKernelGGL(rowkey_finalize_spar_kernel, dim3(cp2_cdiv(R, 32), drows ? CH / FS_CPB : 1), dim3(32 * FS_SL), 0,
                           cp2_stream(stream), a);
    else
        hipLaunchKernelGGL(rowkey_finalize_kernel, dim3(cp2_cdiv(R, 64), drows ? CH / (4 * FIN_CPW) : 1), dim3(256), 0,
                           cp2_stream(stream), a);
    int rc = cp2_launch_status();
    if (rc || !loss_mean) return rc;
    hipLaunchKernelGGL(mean_kernel, dim3(1), dim3(256), 0, cp2_stream(stream), loss_rows, R, loss_mean);
    return cp2_launch_status();
}

// ===========================================================================
// dense (pixel-to-pixel) InfoNCE of one sample pair
// ===========================================================================
// (struct DenseArgs: infonce_common.hpp -- dense_post.hpp and quantile.hip's step_post_kernel use it too)

// value and derivative of the squashing function of the negative pairs
__device__ __forceinline__ float neg_squash(float raw, float scale, float cen, float* dfd) {
    const float sig = 1.f / (1.f + expf(-scale * (raw - cen)));
    if (dfd) *dfd = 2.f * scale * sig * (1.f - sig);
    return 2.f * sig - 1.f;
}

__device__ __forceinline__ float corr_weight(int64_t pa, int64_t pb, int64_t ra, int64_t rb, float wp, float wr, float wn) {
    // builder.py:1225-1243: pixel match -> w_pixel, else known-region match -> w_region, else 0; zeros become w_not
    float w = (pa == pb) ? wp : ((ra == rb && ra != 0 && rb != 0) ? wr : 0.f);
    return w == 0.f ? wn : w;
}

constexpr int DKT = 64, DKP = DKT + 1;
#ifndef CP2_DNW
#define CP2_DNW 4
#endif
constexpr int DNW = CP2_DNW, DNT = DNW * 64;  // dense workgroup = DNW waves = 32*DNW owner pixels

// Workgroups are dealt round-robin over the 8 XCDs (blockIdx % 8 labels the XCD group).  Every workgroup of a
// sample streams that sample's whole other-side map (P x 128 floats), so the workgroups of one sample should
// share one XCD's 4 MiB L2: linear id -> work item such that each XCD group owns a contiguous run of
// (sample, tile) items (bijective for any grid size; MI355X guide T1).
__device__ __forceinline__ int xcd_work_item(int id, int n_items) {
    const int q = n_items / 8, rem = n_items % 8, xcd = id % 8, j = id / 8;
    return (xcd < rem ? xcd * (q + 1) : rem * (q + 1) + (xcd - rem) * q) + j;
}
struct DenseLds {
    float T[CH * DKP];
    float ma[DKT]; float aux[DKT]; float aux2[DKT];
    int64_t pid[DKT]; int64_t rid[DKT];
};

// owners = key pixels y (lane), others = query pixels x (LDS tile).  Column-wise softmax
// statistics over x for every y, the masked column sums, and the logging sums.
template <bool WEIGHTS, bool NEG = false>
__global__ __launch_bounds__(DNT, 2) void dense_fwd_kernel(DenseArgs a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    DenseLds& L = *reinterpret_cast<DenseLds*>(smem);
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6, r = lane & 31, h = lane >> 5;
    const int P = a.P, tiles = (P + 32 * DNW - 1) / (32 * DNW), S = a.splits;
    const int item = xcd_work_item(blockIdx.x, gridDim.x);
    const int sp = item % S, st = item / S;
    const int n = st / tiles;
    const int y = ((st % tiles) * DNW + wid) * 32 + r;
    const bool y_ok = y < P;
    const float* qd = a.qd + (int64_t)n * CH * P;
    const float* kd = a.kd + (int64_t)n * CH * P;
    float bq[CH / 2];
#pragma unroll
    for (int t = 0; t < CH / 2; ++t) bq[t] = y_ok ? kd[(int64_t)(2 * t + h) * P + y] : 0.f;
    int64_t pb = 0, rb = 0;
    if (WEIGHTS && y_ok) { pb = a.pix_b[(int64_t)n * P + y]; rb = a.reg_b[(int64_t)n * P + y]; }
    float m_run = -INFINITY, s_run = 0.f, a_run = 0.f, pos_run = 0.f, all_run = 0.f, best_v = -INFINITY;
    int best_x = 0;
    const float mb_y = (NEG && y_ok) ? a.mask_b[(int64_t)n * P + y] : 0.f;
    const float cen = (NEG && a.neg_center) ? a.neg_center[n] : 0.f;
    const bool vec_ok = (P % 4 == 0) && ((reinterpret_cast<uintptr_t>(qd) & 15u) == 0);
    const int xs = (((P + S - 1) / S + DKT - 1) / DKT) * DKT;     // this split's query pixels: [x_begin, x_end)
    const int x_begin = sp * xs, x_end = min(P, x_begin + xs);
    TileRegs<DKT, DNT> rg;
    if (x_begin < x_end) tile_load<DKT, DNT>(rg, qd, P, x_begin, P, tid, vec_ok);
    for (int x0 = x_begin; x0 < x_end; x0 += DKT) {
        __syncthreads();
        tile_store<DKT, DNT>(L.T, rg, tid);
        if (x0 + DKT < x_end) tile_load<DKT, DNT>(rg, qd, P, x0 + DKT, P, tid, vec_ok);  // next tile in flight during the MFMAs
        if (tid < DKT) {
            const int x = x0 + tid;
            L.ma[tid] = x < P ? a.mask_a[(int64_t)n * P + x] : 0.f;
            if (WEIGHTS) {
                L.pid[tid] = x < P ? a.pix_a[(int64_t)n * P + x] : -1;
                L.rid[tid] = x < P ? a.reg_a[(int64_t)n * P + x] : 0;
            }
        }
        __syncthreads();
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            const int kk = s * 32;
            if (x0 + kk >= P) continue;
            const f32x16 acc = product1<DKP>(L.T, kk, bq, r, h);
            float sv[16];
            float tmax = -INFINITY;
#pragma unroll
            for (int reg = 0; reg < 16; ++reg) {
                const int xi = kk + rho(reg, h), x = x0 + xi;
                const bool valid = x < P;
                const float raw = acc[reg];
                if (a.logits_out && valid && y_ok) a.logits_out[((int64_t)n * P + x) * P + y] = raw;
                float w = 1.f;
                if (WEIGHTS) w = corr_weight(L.pid[xi], pb, L.rid[xi], rb, a.w_pixel, a.w_region, a.w_not);
                float rs = raw;                            // the logging sums below keep the raw score (stats come first, :1298)
                if (NEG && (L.ma[xi] * mb_y) == 0.f) rs = neg_squash(raw, a.neg_scale, cen, nullptr);
                const float v = rs * w * a.inv_t;
                sv[reg] = valid ? v : -INFINITY;
                if (valid) {
                    const float ma = L.ma[xi];
                    a_run += ma * v;
                    pos_run += ma * raw;
                    all_run += raw;
                    if (v > best_v) { best_v = v; best_x = x; }
                }
                tmax = fmaxf(tmax, sv[reg]);
            }
            tmax = fmaxf(tmax, __shfl_xor(tmax, 32, 64));
            const float m_new = fmaxf(m_run, tmax);
            s_run *= __expf(m_run - m_new);
            m_run = m_new;
#pragma unroll
            for (int reg = 0; reg < 16; ++reg) s_run += __expf(sv[reg] - m_run);
        }
    }
    // combine the two lane halves (they saw disjoint x)
    s_run += __shfl_xor(s_run, 32, 64);
    a_run += __shfl_xor(a_run, 32, 64);
    pos_run += __shfl_xor(pos_run, 32, 64);
    all_run += __shfl_xor(all_run, 32, 64);
    const float ov = __shfl_xor(best_v, 32, 64);
    const int ox = __shfl_xor(best_x, 32, 64);
    if (ov > best_v || (ov == best_v && ox < best_x)) { best_v = ov; best_x = ox; }
    if (y_ok && h == 0) {
        const int64_t o = (int64_t)n * P + y;
        if (S == 1) {
            a.lse[o] = m_run + logf(s_run);
            a.colsum_a[o] = a_run;
            a.possum[o] = pos_run;
            a.allsum[o] = all_run;
            a.colmax[o] = best_v;
            a.argx[o] = best_x;
        } else {
            const int64_t BP = (int64_t)(gridDim.x / (tiles * S)) * P, arr = (int64_t)S * BP;
            float* q = a.part + (int64_t)sp * BP + o;
            q[0] = m_run; q[arr] = s_run; q[2 * arr] = a_run; q[3 * arr] = pos_run; q[4 * arr] = all_run;
            q[5 * arr] = best_v; q[6 * arr] = __int_as_float(best_x);
        }
    }
}

// One workgroup per sample folds the S partial column statistics of its own key pixels (splits cover increasing x ranges,
// so "first maximum" -- torch.argmax's tie rule -- = strictly-greater replacement in split order), stores the merged
// per-key values the backward and the callers read, and finishes the sample's scalars from them.
__global__ __launch_bounds__(256) void dense_post_kernel(DenseArgs a, float* __restrict__ sample_scal, int64_t BP) {
    dense_post_body<256>(a, sample_scal, BP, (int)blockIdx.x);
}

// The instance loss's finalize (rowkey_small_finalize_body: CH / FS2_CPB workgroups of 1024 threads) and the dense loss's
// post-pass (one workgroup per sample) in ONE launch: neither reads what the other writes, both follow their producer
// kernels in stream order (round 4: loss section 11 -> 10 launches).
__global__ __launch_bounds__(1024) void loss_post_kernel(RowKeyFinArgs fa, float* __restrict__ loss_mean, DenseArgs da,
                                                         float* __restrict__ sample_scal, int64_t BP, int nfin) {
    if ((int)blockIdx.x < nfin) { rowkey_small_finalize_body(fa, loss_mean, (int)blockIdx.x); return; }
    dense_post_body<1024>(da, sample_scal, BP, (int)blockIdx.x - nfin);
}

// owners = query pixels x (lane), others = key pixels y (LDS tile).
//   d loss_n / d Ls[x][y] = mb[y] (Sa softmax_x(Ls)[x][y] - ma[x]) / (Sa Sb),  Ls = L w / T
//   g_dense[c][x] = grad_scale * sum_y kd[c][y] * (w/T) * dLs[x][y]
template <bool WEIGHTS, bool NEG = false>
__global__ __launch_bounds__(DNT, 2) void dense_bwd_kernel(DenseArgs a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    DenseLds& L = *reinterpret_cast<DenseLds*>(smem);
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6, r = lane & 31, h = lane >> 5;
    const int P = a.P, tiles = (P + 32 * DNW - 1) / (32 * DNW), S = a.splits;
    const int item = xcd_work_item(blockIdx.x, gridDim.x);
    const int sp = item % S, st = item / S;
    const int n = st / tiles;
    const int x = ((st % tiles) * DNW + wid) * 32 + r;
    const bool x_ok = x < P;
    const float* qd = a.qd + (int64_t)n * CH * P;
    const float* kd = a.kd + (int64_t)n * CH * P;
    float bq[CH / 2];
#pragma unroll
    for (int t = 0; t < CH / 2; ++t) bq[t] = x_ok ? qd[(int64_t)(2 * t + h) * P + x] : 0.f;
    const float Sa = a.sample_scal[(int64_t)n * 8 + 0], Sb = a.sample_scal[(int64_t)n * 8 + 1];
    const float gs = a.grad_scale * a.inv_t / (Sa * Sb);
    const float ma = x_ok ? a.mask_a[(int64_t)n * P + x] : 0.f;
    const float cen = (NEG && a.neg_center) ? a.neg_center[n] : 0.f;
    int64_t pa = -1, ra = 0;
    if (WEIGHTS && x_ok) { pa = a.pix_a[(int64_t)n * P + x]; ra = a.reg_a[(int64_t)n * P + x]; }
    f32x16 U[4];
#pragma unroll
    for (int cb = 0; cb < 4; ++cb) U[cb] = (f32x16){0};
    const bool vec_ok = (P % 4 == 0) && ((reinterpret_cast<uintptr_t>(kd) & 15u) == 0);
    const int ys = (((P + S - 1) / S + DKT - 1) / DKT) * DKT;     // this split's key pixels: [y_begin, y_end)
    const int y_begin = sp * ys, y_end = min(P, y_begin + ys);
    TileRegs<DKT, DNT> rg;
    if (y_begin < y_end) tile_load<DKT, DNT>(rg, kd, P, y_begin, P, tid, vec_ok);
    for (int y0 = y_begin; y0 < y_end; y0 += DKT) {
        __syncthreads();
        tile_store<DKT, DNT>(L.T, rg, tid);
        if (y0 + DKT < y_end) tile_load<DKT, DNT>(rg, kd, P, y0 + DKT, P, tid, vec_ok);  // next tile in flight during the MFMAs
        if (tid < DKT) {
            const int y = y0 + tid;
            L.ma[tid] = y < P ? a.mask_b[(int64_t)n * P + y] : 0.f;   // mb
            L.aux[tid] = y < P ? a.lse[(int64_t)n * P + y] : 0.f;
            if (WEIGHTS) {
                L.pid[tid] = y < P ? a.pix_b[(int64_t)n * P + y] : -2;
                L.rid[tid] = y < P ? a.reg_b[(int64_t)n * P + y] : 0;
            }
        }
        __syncthreads();
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            const int kk = s * 32;
            if (y0 + kk >= P) continue;
            const f32x16 acc = product1<DKP>(L.T, kk, bq, r, h);
            float p[16];
#pragma unroll
            for (int reg = 0; reg < 16; ++reg) {
                const int yi = kk + rho(reg, h);
                const bool valid = (y0 + yi) < P;
                float w = 1.f;
                if (WEIGHTS) w = corr_weight(pa, L.pid[yi], ra, L.rid[yi], a.w_pixel, a.w_region, a.w_not);
                float rs = acc[reg], dfd = 1.f;
                if (NEG && (ma * L.ma[yi]) == 0.f) rs = neg_squash(acc[reg], a.neg_scale, cen, &dfd);
                const float ls = rs * w * a.inv_t;
                const float sm = __expf(ls - L.aux[yi]);
                p[reg] = (valid && x_ok) ? gs * w * dfd * L.ma[yi] * (Sa * sm - ma) : 0.f;
            }
            product2<DKP>(L.T, kk, p, U, r, h);
        }
    }
    if (x_ok) {
        // S > 1: partial gradient of this split, part[sp][B][CH][P]; dense_grad_sum_kernel adds the splits in order
        float* base = S == 1 ? a.g_dense : a.part + (int64_t)sp * (gridDim.x / (tiles * S)) * CH * P;
        float* g = base + (int64_t)n * CH * P + x;
#pragma unroll
        for (int cb = 0; cb < 4; ++cb)
#pragma unroll
            for (int reg = 0; reg < 16; ++reg) g[(int64_t)(cb * 32 + rho(reg, h)) * P] = U[cb][reg];
    }
}

static int dense_check(const float* qd, const float* kd, const float* ma, const float* mb, const int64_t* pa,
                       const int64_t* pb, const int64_t* ra, const int64_t* rb, int B, int C, int P, float t) {
    if (!qd || !kd || !ma || !mb) return CP2_ERR_NULL;
    const int nid = (pa != nullptr) + (pb != nullptr) + (ra != nullptr) + (rb != nullptr);
    if (nid != 0 && nid != 4) return CP2_ERR_NULL;
    if (B <= 0 || P <= 0 || !(t > 0.f)) return CP2_ERR_SHAPE;
    if (C != CH) return CP2_ERR_UNSUPPORTED;
    return CP2_OK;
}

// Splits of the query-pixel range per (sample, key tile): enough workgroups for two per CU (the soft-max section of
// one then overlaps the MFMA chain of the other), never less than one 64-pixel tile per split.
CP2_API int cp2_dense_num_splits(int B, int P) {
    if (B <= 0 || P <= 0) return CP2_ERR_SHAPE;
    const int64_t base = (int64_t)cp2_cdiv(P, 32 * DNW) * B;
    int s = (int)((512 + base - 1) / base);
    const int max_s = cp2_cdiv(P, DKT);
    if (s > max_s) s = max_s;
    if (s > 16) s = 16;
    return s < 1 ? 1 : s;
}

CP2_API int cp2_dense_infonce_fwd(const float* q_dense, const float* k_dense, const float* mask_a,
                                  const float* mask_b, const int64_t* pix_a, const int64_t* pix_b,
                                  const int64_t* reg_a, const int64_t* reg_b, float w_pixel, float w_region,
                                  float w_not, float temperature, float* lse, float* colsum_a, float* possum,
                                  float* allsum, float* colmax, int32_t* argx, float* sample_scal,
                                  float* logits_out, float* split_ws, int negative_mode, float negative_scale,
                                  const float* negative_center, int B, int C, int P, void* stream) {
    int rc = dense_check(q_dense, k_dense, mask_a, mask_b, pix_a, pix_b, reg_a, reg_b, B, C, P, temperature);
    if (rc) return rc;
    if (!lse || !colsum_a || !possum || !allsum || !colmax || !argx) return CP2_ERR_NULL;   // sample_scal NULL: post-pass deferred to cp2_loss_post
    const int S = split_ws ? cp2_dense_num_splits(B, P) : 1;
    DenseArgs a{q_dense, k_dense, mask_a, mask_b, pix_a, pix_b, reg_a, reg_b, w_pixel, w_region, w_not,
                1.0f / temperature, P, lse, colsum_a, possum, allsum, colmax, argx, logits_out, nullptr, 0.f, nullptr,
                S, split_ws, negative_scale, negative_center};
    const dim3 grid(cp2_cdiv(P, 32 * DNW) * B * S);
    const size_t lds = sizeof(DenseLds);
    if (negative_mode) {
        if (pix_a) CP2_LAUNCH_PROFILED((dense_fwd_kernel<true, true>), grid, dim3(DNT), lds, cp2_stream(stream), a);
        else CP2_LAUNCH_PROFILED((dense_fwd_kernel<false, true>), grid, dim3(DNT), lds, cp2_stream(stream), a);
    } else if (pix_a) CP2_LAUNCH_PROFILED((dense_fwd_kernel<true, false>), grid, dim3(DNT), lds, cp2_stream(stream), a);
    else CP2_LAUNCH_PROFILED((dense_fwd_kernel<false, false>), grid, dim3(DNT), lds, cp2_stream(stream), a);
    rc = cp2_launch_status();
    if (rc || !sample_scal) return rc;
    hipLaunchKernelGGL(dense_post_kernel, dim3(B), dim3(256), 0, cp2_stream(stream), a, sample_scal, (int64_t)B * P);
    return cp2_launch_status();                            // the batch means are formed by cp2_step_scalars
}

// Finalize of cp2_rowkey_infonce_fwd's partials (the small-R form: R <= 32, nsplit >= 16 -- the instance loss) and the
// post-pass of a cp2_dense_infonce_fwd call that was given sample_scal = NULL, in one launch.  Arguments as in
// cp2_rowkey_infonce_finalize, then the dense call's own output arrays, split workspace (NULL: no split) and masks.
CP2_API int cp2_loss_post(const float* part_m, const float* part_s, const int32_t* part_cnt, const float* part_U, int nsplit,
                          const float* extras, int NE, float temperature, float grad_scale, int R, int RP, int64_t d_sn,
                          int64_t d_sx, int64_t d_sc, float* lse, float* loss_rows, int32_t* cnt_gt, float* drows, float* dE,
                          float* loss_mean, const float* mask_a, const float* mask_b, float* d_lse, float* colsum_a,
                          float* possum, float* allsum, float* colmax, int32_t* argx, float* sample_scal, float* split_ws,
                          int B, int C, int P, void* stream) {
    RowKeyFinArgs fa;
    DenseArgs da;
    const int rc = loss_post_fill(part_m, part_s, part_cnt, part_U, nsplit, extras, NE, temperature, grad_scale, R, RP, d_sn, d_sx,
                                  d_sc, lse, loss_rows, cnt_gt, drows, dE, mask_a, mask_b, d_lse, colsum_a, possum, allsum, colmax,
                                  argx, sample_scal, split_ws, B, C, P, &fa, &da);
    if (rc) return rc;
    const int nfin = CH / FS2_CPB;
    hipLaunchKernelGGL(loss_post_kernel, dim3(nfin + B), dim3(1024), 0, cp2_stream(stream), fa, loss_mean, da, sample_scal,
                       (int64_t)B * P, nfin);
    return cp2_launch_status();
}

CP2_API int cp2_dense_infonce_bwd(const float* q_dense, const float* k_dense, const float* mask_a,
                                  const float* mask_b, const int64_t* pix_a, const int64_t* pix_b,
                                  const int64_t* reg_a, const int64_t* reg_b, float w_pixel, float w_region,
                                  float w_not, float temperature, const float* lse, const float* sample_scal,
                                  float grad_scale, float* g_dense, float* split_ws, int negative_mode,
                                  float negative_scale, const float* negative_center, int B, int C, int P,
                                  void* stream) {
    int rc = dense_check(q_dense, k_dense, mask_a, mask_b, pix_a, pix_b, reg_a, reg_b, B, C, P, temperature);
    if (rc) return rc;
    if (!lse || !sample_scal) return CP2_ERR_NULL;
    const int S = split_ws ? cp2_dense_num_splits(B, P) : 1;
    if (S == 1 ? !g_dense : g_dense != nullptr) return CP2_ERR_NULL;   // S > 1: the S partial gradients stay in split_ws (cp2_feat_bwd_fused adds them)
    DenseArgs a{q_dense, k_dense, mask_a, mask_b, pix_a, pix_b, reg_a, reg_b, w_pixel, w_region, w_not,
                1.0f / temperature, P, const_cast<float*>(lse), nullptr, nullptr, nullptr, nullptr, nullptr, nullptr,
                sample_scal, grad_scale, g_dense, S, split_ws, negative_scale, negative_center};
    const dim3 grid(cp2_cdiv(P, 32 * DNW) * B * S);
    const size_t lds = sizeof(DenseLds);
    if (negative_mode) {
        if (pix_a) CP2_LAUNCH_PROFILED((dense_bwd_kernel<true, true>), grid, dim3(DNT), lds, cp2_stream(stream), a);
        else CP2_LAUNCH_PROFILED((dense_bwd_kernel<false, true>), grid, dim3(DNT), lds, cp2_stream(stream), a);
    } else if (pix_a) CP2_LAUNCH_PROFILED((dense_bwd_kernel<true, false>), grid, dim3(DNT), lds, cp2_stream(stream), a);
    else CP2_LAUNCH_PROFILED((dense_bwd_kernel<false, false>), grid, dim3(DNT), lds, cp2_stream(stream), a);
    return cp2_launch_status();
}
