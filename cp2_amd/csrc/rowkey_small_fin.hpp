// rowkey_small_finalize_body: shared by rowkey_small.hip (its own launch) and infonce.hip (loss_post_kernel: the same work
// beside the dense loss's post-pass in one launch).
#pragma once
#include "infonce_common.hpp"

// ---------------------------------------------------------------------------------------------------------------
// Merge of the S per-workgroup partials for R <= 32 rows, ONE launch (was: merge kernel + mean kernel, and three
// dependent passes over the partials).  Workgroup = 1024 threads = (32 split lanes) x (32 rows); grid = CH / FS2_CPB
// channel groups.  Every load a thread needs -- its S/32 partial maxima / sums / counts and its S/32 x FS2_CPB gradient
// partials -- is issued up front (the addresses do not depend on each other), so the kernel is one memory round
// trip plus three LDS reductions instead of a chain of dependent loops.  Workgroup 0 also writes the per-row outputs
// and the batch mean of the loss.  Fixed reduction order: deterministic.
// ---------------------------------------------------------------------------------------------------------------
constexpr int FS2_CPB = 2, FS2_SL = 32, FS2_NS = 8;      // up to 32 * 8 = 256 splits per pass of the unrolled loads

// `block` = channel group 0 .. CH / FS2_CPB - 1; the calling workgroup has 32 * FS2_SL = 1024 threads
__device__ __forceinline__ void rowkey_small_finalize_body(const RowKeyFinArgs& a, float* __restrict__ loss_mean, int block) {
    __shared__ float red[FS2_SL][33];
    __shared__ int redi[FS2_SL][33];
    __shared__ float red2[FS2_CPB][FS2_SL][33];
    const int r = threadIdx.x & 31, sl = threadIdx.x >> 5;
    const bool ok = r < a.R;
    const int rr = ok ? r : 0;
    const int c0 = block * FS2_CPB;
    float M = -INFINITY, z = 0.f, acc[FS2_CPB];
    int cnt = 0;
#pragma unroll
    for (int i = 0; i < FS2_CPB; ++i) acc[i] = 0.f;
    float e[4] = {0.f, 0.f, 0.f, 0.f};
    for (int j = 0; j < a.NE; ++j) e[j] = a.extras[(int64_t)rr * a.NE + j] * a.inv_t;
    // pass structure: S <= 256 is one pass (the common case); larger S accumulates pass by pass with a running maximum
    for (int s0 = 0; s0 < a.S; s0 += FS2_SL * FS2_NS) {
        float mv[FS2_NS], sv[FS2_NS], uv[FS2_CPB][FS2_NS];
        int cv[FS2_NS];
#pragma unroll
        for (int j = 0; j < FS2_NS; ++j) {
            const int s = s0 + sl + FS2_SL * j;
            const bool v = s < a.S;
            const int64_t o = (int64_t)(v ? s : 0) * a.R + rr;
            mv[j] = a.part_m[o]; sv[j] = a.part_s[o]; cv[j] = a.part_cnt[o];
            if (a.drows) {
#pragma unroll
                for (int i = 0; i < FS2_CPB; ++i) uv[i][j] = a.part_U[((int64_t)(v ? s : 0) * CH + c0 + i) * a.R + rr];
            }
            if (!v) { mv[j] = -INFINITY; sv[j] = 0.f; cv[j] = 0; }
        }
        float m_new = M;
#pragma unroll
        for (int j = 0; j < FS2_NS; ++j) m_new = fmaxf(m_new, mv[j]);
        const float sc = (M == -INFINITY) ? 0.f : __expf(M - m_new);
        z *= sc;
#pragma unroll
        for (int i = 0; i < FS2_CPB; ++i) acc[i] *= sc;
#pragma unroll
        for (int j = 0; j < FS2_NS; ++j) {
            const float wj = (mv[j] == -INFINITY) ? 0.f : __expf(mv[j] - m_new);
            z += sv[j] * wj;
            cnt += cv[j];
            if (a.drows) {
#pragma unroll
                for (int i = 0; i < FS2_CPB; ++i) acc[i] += uv[i][j] * wj;
            }
        }
        M = m_new;
    }
    // this thread now holds (M, z, acc) of its own splits relative to its own maximum M; combine the 32 split lanes
    red[sl][r] = M;
    __syncthreads();
    float Mg = -INFINITY;
#pragma unroll
    for (int j = 0; j < FS2_SL; ++j) Mg = fmaxf(Mg, red[j][r]);
    for (int j = 0; j < a.NE; ++j) Mg = fmaxf(Mg, e[j]);
    const float f = (M == -INFINITY) ? 0.f : __expf(M - Mg);
    __syncthreads();
    red[sl][r] = z * f;
    redi[sl][r] = cnt;
#pragma unroll
    for (int i = 0; i < FS2_CPB; ++i) red2[i][sl][r] = acc[i] * f;
    __syncthreads();
    float Z = 0.f;
    int ctot = 0;
#pragma unroll
    for (int j = 0; j < FS2_SL; ++j) { Z += red[j][r]; ctot += redi[j][r]; }
    for (int j = 0; j < a.NE; ++j) Z += __expf(e[j] - Mg);
    for (int j = 1; j < a.NE; ++j) ctot += (e[j] > e[0]) ? 1 : 0;     // extra negatives also rank against the positive
    const float lse = Mg + logf(Z);
    if (a.drows && sl < FS2_CPB && ok) {                  // split lane sl finishes channel c0 + sl
        float tot = 0.f;
#pragma unroll
        for (int j = 0; j < FS2_SL; ++j) tot += red2[sl][j][r];
        float* d = a.drows + (int64_t)(r / a.RP) * a.d_sn + (int64_t)(r % a.RP) * a.d_sx;
        d[(int64_t)(c0 + sl) * a.d_sc] = tot * __expf(Mg - lse) * a.grad_scale * a.inv_t;
    }
    if (block == 0 && sl == 0) {                     // one wave half: the per-row outputs and the batch mean
        const float lrow = lse - e[0];
        if (ok) {
            a.lse[r] = lse;
            a.loss_rows[r] = lrow;
            a.cnt_gt[r] = ctot;
            if (a.dE)
                for (int j = 0; j < a.NE; ++j)
                    a.dE[(int64_t)r * a.NE + j] = a.grad_scale * a.inv_t * (__expf(e[j] - lse) - (j == 0 ? 1.f : 0.f));
        }
        if (loss_mean) {
            float t = ok ? lrow : 0.f;
#pragma unroll
            for (int off = 16; off > 0; off >>= 1) t += __shfl_xor(t, off, 64);
            if (r == 0) loss_mean[0] = t / (float)a.R;
        }
    }
}

