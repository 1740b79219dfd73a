// dense_post_body: the post-pass of the dense InfoNCE forward (fold of the split statistics + the per-sample scalars),
// shared by infonce.hip (dense_post_kernel, loss_post_kernel) and quantile.hip (step_post_kernel).
// One workgroup per sample folds the S partial column statistics of its own key pixels (splits cover increasing x ranges,
// so "first maximum" -- torch.argmax's tie rule -- = strictly-greater replacement in split order), stores the merged
// per-key values the backward and the callers read, and finishes the sample's scalars from them.
#pragma once
#include "infonce_common.hpp"

template <int NT>
__device__ __forceinline__ void dense_post_body(const DenseArgs& a, float* __restrict__ sample_scal, int64_t BP, int n) {
    constexpr int NW = NT / 64;
    __shared__ float red[6][NW];
    __shared__ float bv[NW];
    __shared__ int bi[NW];
    const int P = a.P, tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int S = a.splits;
    const int64_t arr = (int64_t)S * BP;
    float sa = 0.f, sb = 0.f, t_lse = 0.f, t_a = 0.f, t_pos = 0.f, t_all = 0.f, best = -INFINITY;
    int64_t best_flat = 0;
    for (int y = tid; y < P; y += NT) {
        const int64_t o = (int64_t)n * P + y;
        float lse, ca, pos, all, cmax;
        int ax;
        if (S > 1) {
            const float* q = a.part + o;
            float M = -INFINITY;
            for (int sp = 0; sp < S; ++sp) M = fmaxf(M, q[(int64_t)sp * BP]);
            float s = 0.f;
            ca = 0.f, pos = 0.f, all = 0.f, cmax = -INFINITY, ax = 0;
            for (int sp = 0; sp < S; ++sp) {
                const float* e = q + (int64_t)sp * BP;
                const float m = e[0];
                if (m > -INFINITY) s += e[arr] * __expf(m - M);
                ca += e[2 * arr]; pos += e[3 * arr]; all += e[4 * arr];
                const float v = e[5 * arr];
                if (v > cmax) { cmax = v; ax = __float_as_int(e[6 * arr]); }
            }
            lse = M + logf(s);
            a.lse[o] = lse, a.colsum_a[o] = ca, a.possum[o] = pos, a.allsum[o] = all, a.colmax[o] = cmax, a.argx[o] = ax;
        } else {
            lse = a.lse[o], ca = a.colsum_a[o], pos = a.possum[o], all = a.allsum[o], cmax = a.colmax[o], ax = a.argx[o];
        }
        const float mb = a.mask_b[o];
        sa += a.mask_a[o];
        sb += mb;
        t_lse += mb * lse;
        t_a += mb * ca;
        t_pos += mb * pos;
        t_all += all;
        const int64_t flat = (int64_t)ax * P + y;
        if (cmax > best || (cmax == best && flat < best_flat)) { best = cmax; best_flat = flat; }
    }
    float vals[6] = {sa, sb, t_lse, t_a, t_pos, t_all};
#pragma unroll
    for (int j = 0; j < 6; ++j) {
        const float s = wave_sum(vals[j]);
        if (lane == 0) red[j][w] = s;
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        const float ov = __shfl_xor(best, off, 64);
        const int64_t of = __shfl_xor(best_flat, off, 64);
        if (ov > best || (ov == best && of < best_flat)) { best = ov; best_flat = of; }
    }
    if (lane == 0) { bv[w] = best; bi[w] = (int)best_flat; }
    __syncthreads();
    if (tid == 0) {
        float t[6];
        for (int j = 0; j < 6; ++j) {
            // pairwise over the waves: with 4 waves this is ((w0 + w1) + w2) + w3 as before; a 16-wave workgroup sums its
            // own 16 partials in the same left-to-right order (the two workgroup sizes differ in the last bits: each is
            // deterministic, and the step uses one of them throughout)
            float acc = red[j][0];
            for (int i = 1; i < NW; ++i) acc += red[j][i];
            t[j] = acc;
        }
        float bb = bv[0];
        int bf = bi[0];
        for (int j = 1; j < NW; ++j)
            if (bv[j] > bb || (bv[j] == bb && bi[j] < bf)) { bb = bv[j]; bf = bi[j]; }
        const float Sa = t[0], Sb = t[1], npos = Sa * Sb;
        float* o = sample_scal + (int64_t)n * 8;
        o[0] = Sa;
        o[1] = Sb;
        o[2] = (Sa * t[2] - t[3]) / npos;                 // 0/0 = NaN when a mask is empty, as the reference
        o[3] = t[4] / npos;
        o[4] = (t[5] - t[4]) / ((float)P * (float)P - npos);
        o[5] = a.mask_a[(int64_t)n * P + bf / P] * a.mask_b[(int64_t)n * P + bf % P];
        o[6] = 0.f;
        o[7] = 0.f;
    }
}


// Argument check + structs of the two tails cp2_loss_post / cp2_step_post run (see include/cp2hip.h)
static inline int loss_post_fill(const float* part_m, const float* part_s, const int32_t* part_cnt, const float* part_U, int nsplit,
                                 const float* extras, int NE, float temperature, float grad_scale, int R, int RP, int64_t d_sn,
                                 int64_t d_sx, int64_t d_sc, float* lse, float* loss_rows, int32_t* cnt_gt, float* drows, float* dE,
                                 const float* mask_a, const float* mask_b, float* d_lse, float* colsum_a, float* possum,
                                 float* allsum, float* colmax, int32_t* argx, float* sample_scal, float* split_ws, int B, int C,
                                 int P, RowKeyFinArgs* fa, DenseArgs* da) {
    if (!part_m || !part_s || !part_cnt || !lse || !loss_rows || !cnt_gt) return CP2_ERR_NULL;
    if (drows && !part_U) return CP2_ERR_NULL;
    if (NE > 0 && !extras) return CP2_ERR_NULL;
    if (!mask_a || !mask_b || !d_lse || !colsum_a || !possum || !allsum || !colmax || !argx || !sample_scal) return CP2_ERR_NULL;
    if (R <= 0 || RP <= 0 || nsplit <= 0 || NE < 0 || NE > 4 || !(temperature > 0.f) || B <= 0 || P <= 0) return CP2_ERR_SHAPE;
    if (C != CH || R > 32 || nsplit < 16) return CP2_ERR_UNSUPPORTED;       // other row counts: the separate entry points
    *fa = RowKeyFinArgs{part_m, part_s, part_cnt, part_U, nsplit, extras, NE, 1.0f / temperature, grad_scale,
                        R, RP, d_sn, d_sx, d_sc, lse, loss_rows, cnt_gt, drows, dE};
    *da = DenseArgs{};
    da->mask_a = mask_a; da->mask_b = mask_b; da->P = P;
    da->lse = d_lse; da->colsum_a = colsum_a; da->possum = possum; da->allsum = allsum; da->colmax = colmax; da->argx = argx;
    da->splits = split_ws ? cp2_dense_num_splits(B, P) : 1;
    da->part = split_ws;
    return CP2_OK;
}
