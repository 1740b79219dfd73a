// a7: per-pixel L2 normalisation over channels + masked pooling, forward and backward.
// Reference: builder.py:1261-1268 (query side, needs grad) and :1279-1285 (key side).
//   dense[c][x] = feat[c][x] / max(||feat[:,x]||, 1e-12)
//   pos = unit(sum_x dense[:,x]*mask[x]),  neg = unit(sum_x dense[:,x]*(1-mask[x]))
// Layout: feature maps are addressed with explicit strides (NCHW or channels-last
// encoder output); dense is written [B][C][P] (pixel index contiguous), the layout the
// MFMA kernels in infonce.hip consume for both operands.
// A workgroup = 4 waves = 64 pixels (lane = pixel, coalesced) x 4 channel quarters.
#include "common.hpp"

constexpr float kNormEps = 1e-12f;  // F.normalize default eps

template <int C>
__global__ __launch_bounds__(256) void feat_normalize_pool_kernel(
    const float* __restrict__ feat, int64_t sn, int64_t sc, int64_t sp, const float* __restrict__ mask,
    float* __restrict__ dense, float* __restrict__ inv_norm, float* __restrict__ pool_partial, int P, int NT) {
    constexpr int CQ = C / 4;
    __shared__ float ssq[4][64];
    const int n = blockIdx.y, tile = blockIdx.x;
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int x = tile * 64 + lane;
    const bool ok = x < P;
    const float* f = feat + n * sn + (int64_t)(ok ? x : 0) * sp + (int64_t)(w * CQ) * sc;
    float v[CQ];
    float sq = 0.f;
#pragma unroll
    for (int i = 0; i < CQ; ++i) {
        v[i] = ok ? f[(int64_t)i * sc] : 0.f;
        sq += v[i] * v[i];
    }
    ssq[w][lane] = sq;
    __syncthreads();
    const float nrm = sqrtf(ssq[0][lane] + ssq[1][lane] + ssq[2][lane] + ssq[3][lane]);
    const float den = fmaxf(nrm, kNormEps);
    const float m = ok ? mask[(int64_t)n * P + x] : 0.f;
    const float mneg = ok ? ((m != 0.f) ? 0.f : 1.f) : 0.f;  // (~mask.bool()).float()
    if (w == 0 && ok) inv_norm[(int64_t)n * P + x] = 1.0f / den;
    float* d = dense + ((int64_t)n * C + w * CQ) * P + x;
    float* pp = pool_partial + (((int64_t)n * NT + tile) * 2) * C + w * CQ;
#pragma unroll
    for (int i = 0; i < CQ; ++i) {
        const float y = v[i] / den;
        if (ok) d[(int64_t)i * P] = y;
        const float sp_ = wave_sum(y * m);
        const float sn_ = wave_sum(y * mneg);
        if (lane == 0) { pp[i] = sp_; pp[C + i] = sn_; }
    }
}

// One workgroup per sample: reduce the per-tile partial sums of both encoders' maps,
// normalise, and form the three "extra" instance logits (raw dot products):
//   E[n][0] = q_pos.k_pos   E[n][1] = q_pos.q_neg   E[n][2] = q_pos.k_neg   (builder.py:1395,1416-1417)
template <int C>
__global__ __launch_bounds__(C) void pool_finalize_kernel(const float* __restrict__ q_partial,
                                                          const float* __restrict__ k_partial, int NT,
                                                          float* __restrict__ q_pos, float* __restrict__ q_neg,
                                                          float* __restrict__ q_norms, float* __restrict__ k_pos,
                                                          float* __restrict__ k_neg, float* __restrict__ extras) {
    __shared__ float red[8][C / 64];
    const int n = blockIdx.x, c = threadIdx.x, lane = c & 63, w = c >> 6;
    float s[4] = {0.f, 0.f, 0.f, 0.f};  // q_pos, q_neg, k_pos, k_neg sums for channel c
    for (int t = 0; t < NT; ++t) {
        const float* qp = q_partial + (((int64_t)n * NT + t) * 2) * C;
        const float* kp = k_partial + (((int64_t)n * NT + t) * 2) * C;
        s[0] += qp[c]; s[1] += qp[C + c]; s[2] += kp[c]; s[3] += kp[C + c];
    }
    float nr[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const float ws = wave_sum(s[j] * s[j]);
        if (lane == 0) red[j][w] = ws;
    }
    __syncthreads();
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        float t = 0.f;
        for (int i = 0; i < C / 64; ++i) t += red[j][i];
        nr[j] = sqrtf(t);
    }
    const float u0 = s[0] / fmaxf(nr[0], kNormEps), u1 = s[1] / fmaxf(nr[1], kNormEps);
    const float u2 = s[2] / fmaxf(nr[2], kNormEps), u3 = s[3] / fmaxf(nr[3], kNormEps);
    q_pos[(int64_t)n * C + c] = u0; q_neg[(int64_t)n * C + c] = u1;
    k_pos[(int64_t)n * C + c] = u2; k_neg[(int64_t)n * C + c] = u3;
    if (c == 0) { q_norms[n * 2 + 0] = nr[0]; q_norms[n * 2 + 1] = nr[1]; }
    __syncthreads();
    const float e[3] = {u0 * u2, u0 * u1, u0 * u3};
#pragma unroll
    for (int j = 0; j < 3; ++j) {
        const float ws = wave_sum(e[j]);
        if (lane == 0) red[4 + j][w] = ws;
    }
    __syncthreads();
    if (c < 3) {
        float t = 0.f;
        for (int i = 0; i < C / 64; ++i) t += red[4 + c][i];
        extras[n * 3 + c] = t;
    }
}

// Backward of the pooled vectors: from dL/dq_pos (queue part + extras) and dL/dq_neg
// to dL/d(sum) for both pools.  unit(s) = s/max(|s|,eps):  ds = (g - u (u.g)) / |s|.
template <int C>
__global__ __launch_bounds__(C) void pool_bwd_kernel(const float* __restrict__ drow_pos /*[B][C]*/,
                                                     const float* __restrict__ dE /*[B][3]*/,
                                                     const float* __restrict__ q_pos, const float* __restrict__ q_neg,
                                                     const float* __restrict__ k_pos, const float* __restrict__ k_neg,
                                                     const float* __restrict__ q_norms, int use_bg,
                                                     float* __restrict__ ds_pos, float* __restrict__ ds_neg) {
    __shared__ float red[2][C / 64];
    const int n = blockIdx.x, c = threadIdx.x, lane = c & 63, w = c >> 6;
    const int64_t o = (int64_t)n * C + c;
    const float qp = q_pos[o], qn = q_neg[o];
    float gp = drow_pos[o] + dE[n * 3 + 0] * k_pos[o];
    float gn = 0.f;
    if (use_bg) {
        gp += dE[n * 3 + 1] * qn + dE[n * 3 + 2] * k_neg[o];
        gn = dE[n * 3 + 1] * qp;
    }
    const float d0 = wave_sum(qp * gp), d1 = wave_sum(qn * gn);
    if (lane == 0) { red[0][w] = d0; red[1][w] = d1; }
    __syncthreads();
    float dp = 0.f, dn = 0.f;
    for (int i = 0; i < C / 64; ++i) { dp += red[0][i]; dn += red[1][i]; }
    const float np_ = q_norms[n * 2 + 0], nn_ = q_norms[n * 2 + 1];
    ds_pos[o] = np_ >= kNormEps ? (gp - qp * dp) / np_ : gp / kNormEps;
    ds_neg[o] = nn_ >= kNormEps ? (gn - qn * dn) / nn_ : gn / kNormEps;
}

// Backward of the per-pixel normalisation, with the pooled gradients folded in:
//   G[c][x] = g_dense[c][x] + mask[x]*ds_pos[c] + (1-mask[x])*ds_neg[c]
//   dfeat[c][x] = (G[c][x] - dense[c][x] * sum_c dense[c][x] G[c][x]) * inv_norm[x]
template <int C>
__global__ __launch_bounds__(256) void feat_bwd_kernel(const float* __restrict__ dense,
                                                       const float* __restrict__ inv_norm,
                                                       const float* __restrict__ mask,
                                                       const float* __restrict__ g_dense,
                                                       const float* __restrict__ ds_pos,
                                                       const float* __restrict__ ds_neg, float* __restrict__ dfeat,
                                                       int64_t sn, int64_t sc, int64_t sp, int P) {
    constexpr int CQ = C / 4;
    __shared__ float sdot[4][64];
    const int n = blockIdx.y, tile = blockIdx.x;
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int x = tile * 64 + lane;
    const bool ok = x < P;
    const int xs = ok ? x : 0;
    const float m = mask[(int64_t)n * P + xs];
    const float mneg = (m != 0.f) ? 0.f : 1.f;
    const float* d = dense + ((int64_t)n * C + w * CQ) * P + xs;
    const float* g = g_dense + ((int64_t)n * C + w * CQ) * P + xs;
    float y[CQ], G[CQ];
    float dot = 0.f;
#pragma unroll
    for (int i = 0; i < CQ; ++i) {
        const int c = w * CQ + i;
        y[i] = d[(int64_t)i * P];
        G[i] = g[(int64_t)i * P] + m * ds_pos[(int64_t)n * C + c] + mneg * ds_neg[(int64_t)n * C + c];
        dot += y[i] * G[i];
    }
    sdot[w][lane] = dot;
    __syncthreads();
    const float tot = sdot[0][lane] + sdot[1][lane] + sdot[2][lane] + sdot[3][lane];
    const float inv = inv_norm[(int64_t)n * P + xs];
    // clamped pixels (norm < eps, inv == 1/eps): y = x/eps, so dx = G/eps
    const bool clamped = inv >= 1.0f / kNormEps;
    if (ok) {
        float* o = dfeat + n * sn + (int64_t)x * sp + (int64_t)(w * CQ) * sc;
#pragma unroll
        for (int i = 0; i < CQ; ++i) o[(int64_t)i * sc] = (clamped ? G[i] : (G[i] - y[i] * tot)) * inv;
    }
}

// ---------------------------------------------------------------------------------------------------------------
// The training step's forms (round 3): query and key map in ONE launch, the key side reading its rows through the
// un-shuffle index (reference builder.py:649 applies it as a gather after the key encoder), channels-last encoder
// outputs staged through LDS so that global reads and writes are 16-byte coalesced; and the backward with the sum of the
// dense kernel's split gradients (dense_grad_sum_kernel) and the pooled-vector backward (pool_bwd_kernel) folded in.
// ---------------------------------------------------------------------------------------------------------------
struct FeatPairArgs {
    const float* feat[2]; int64_t sn[2], sc[2], sp[2];       // 0 = query map, 1 = key map
    const int64_t* k_row;                                    // NULL, or [B]: row of the key encoder's output that belongs to sample n
    const float* mask[2];
    float* dense[2]; float* inv_norm; float* partial[2];     // inv_norm: query side only (needed by the backward)
    int B, P, NT;
};

template <int C>
__global__ __launch_bounds__(256) void feat_normalize_pool_pair_kernel(FeatPairArgs a) {
    constexpr int CQ = C / 4, TP = C + 1;
    __shared__ float tile[64 * TP];
    __shared__ float ssq[4][64];
    const int which = blockIdx.z, n = blockIdx.y, t = blockIdx.x;
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int x = t * 64 + lane, P = a.P;
    const bool ok = x < P;
    int row = n;
    if (which == 1 && a.k_row) {
        const int64_t r = a.k_row[n];
        row = (r < 0 || r >= a.B) ? n : (int)r;
    }
    const int64_t sn = a.sn[which], sc = a.sc[which], sp = a.sp[which];
    const float* fb = a.feat[which] + row * sn;
    float v[CQ];
    const bool staged = sc == 1 && sp == C && ((reinterpret_cast<uintptr_t>(fb) & 15u) == 0);
    if (staged) {
        // the 64-pixel tile is one contiguous span of npx * C floats: coalesced 16-byte loads, transposed through LDS
        const int npx = min(64, P - t * 64);
        const float4* src = reinterpret_cast<const float4*>(fb + (int64_t)t * 64 * C);
        for (int e = threadIdx.x; e < npx * (C / 4); e += 256) {
            const float4 q = src[e];
            float* d = tile + (e / (C / 4)) * TP + (e % (C / 4)) * 4;
            d[0] = q.x, d[1] = q.y, d[2] = q.z, d[3] = q.w;
        }
        __syncthreads();
#pragma unroll
        for (int i = 0; i < CQ; ++i) v[i] = ok ? tile[lane * TP + w * CQ + i] : 0.f;
    } else {
        const float* f = fb + (int64_t)(ok ? x : 0) * sp + (int64_t)(w * CQ) * sc;
#pragma unroll
        for (int i = 0; i < CQ; ++i) v[i] = ok ? f[(int64_t)i * sc] : 0.f;
    }
    float sq = 0.f;
#pragma unroll
    for (int i = 0; i < CQ; ++i) sq += v[i] * v[i];
    ssq[w][lane] = sq;
    __syncthreads();
    const float nrm = sqrtf(ssq[0][lane] + ssq[1][lane] + ssq[2][lane] + ssq[3][lane]);
    const float den = fmaxf(nrm, kNormEps);
    const float m = ok ? a.mask[which][(int64_t)n * P + x] : 0.f;
    const float mneg = ok ? ((m != 0.f) ? 0.f : 1.f) : 0.f;
    if (which == 0 && w == 0 && ok) a.inv_norm[(int64_t)n * P + x] = 1.0f / den;
    float* d = a.dense[which] + ((int64_t)n * C + w * CQ) * P + x;
    float* pp = a.partial[which] + (((int64_t)n * a.NT + t) * 2) * C + w * CQ;
#pragma unroll
    for (int i = 0; i < CQ; ++i) {
        const float y = v[i] / den;
        if (ok) d[(int64_t)i * P] = y;
        const float sp_ = wave_sum(y * m);
        const float sn_ = wave_sum(y * mneg);
        if (lane == 0) { pp[i] = sp_; pp[C + i] = sn_; }
    }
}

struct FeatBwdArgs {
    const float* dense; const float* inv_norm; const float* mask;
    const float* g_part; int S; int64_t split_stride;         // dense gradient = sum over s < S of g_part[s * split_stride + ...], in order
    // pooled-vector backward (pool_bwd_kernel's inputs)
    const float* drow_pos; const float* dE; int NE; const float* q_pos; const float* q_neg; const float* k_pos; const float* k_neg;
    const float* q_norms; int use_bg;
    float* dfeat; int64_t sn, sc, sp;
    int P;
};

template <int C>
__global__ __launch_bounds__(256) void feat_bwd_fused_kernel(FeatBwdArgs a) {
    constexpr int CQ = C / 4, TP = C + 1;
    __shared__ float tile[64 * TP];
    __shared__ float sdot[4][64];
    __shared__ float sds[2][C];
    __shared__ float red[2][C / 64];
    const int n = blockIdx.y, t = blockIdx.x, P = a.P;
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    // ---- this thread's pixel: every global load is issued before the prologue below needs the LDS / barriers, and the
    // split gradients are fetched side by side (a first version added them in a run-time loop of dependent loads: 32.6 us
    // for the launch against 16.7 + 4.9 + 3.8 us of the three launches it replaces)
    const int x = t * 64 + lane;
    const bool ok = x < P;
    const int xs = ok ? x : 0;
    const float m = a.mask[(int64_t)n * P + xs];
    const float mneg = (m != 0.f) ? 0.f : 1.f;
    const float* d = a.dense + ((int64_t)n * C + w * CQ) * P + xs;
    const float* g = a.g_part + ((int64_t)n * C + w * CQ) * P + xs;
    const int S = a.S;
    const int64_t ss = a.split_stride;
    float y[CQ], G[CQ];
#pragma unroll
    for (int i = 0; i < CQ; ++i) y[i] = d[(int64_t)i * P];
#pragma unroll
    for (int i = 0; i < CQ; ++i) {
        const float* gi = g + (int64_t)i * P;
        const float t0 = gi[0];
        const float t1 = S > 1 ? gi[ss] : 0.f, t2 = S > 2 ? gi[2 * ss] : 0.f, t3 = S > 3 ? gi[3 * ss] : 0.f;
        float gs = t0;                                  // split order, as dense_grad_sum_kernel adds them
        if (S > 1) gs += t1;
        if (S > 2) gs += t2;
        if (S > 3) gs += t3;
        for (int sp = 4; sp < S; ++sp) gs += gi[(int64_t)sp * ss];
        G[i] = gs;
    }
    // ---- d loss / d (pooled sums) of this sample, as pool_bwd_kernel computes it (same operation order)
    float gp = 0.f, gn = 0.f, qp = 0.f, qn = 0.f;
    if (tid < C) {
        const int64_t o = (int64_t)n * C + tid;
        qp = a.q_pos[o], qn = a.q_neg[o];
        gp = a.drow_pos[o] + a.dE[n * a.NE + 0] * a.k_pos[o];
        if (a.use_bg) {
            gp += a.dE[n * a.NE + 1] * qn + a.dE[n * a.NE + 2] * a.k_neg[o];
            gn = a.dE[n * a.NE + 1] * qp;
        }
        const float d0 = wave_sum(qp * gp), d1 = wave_sum(qn * gn);
        if (lane == 0) { red[0][w] = d0; red[1][w] = d1; }
    }
    __syncthreads();
    if (tid < C) {
        float dp = 0.f, dn = 0.f;
        for (int i = 0; i < C / 64; ++i) { dp += red[0][i]; dn += red[1][i]; }
        const float np_ = a.q_norms[n * 2 + 0], nn_ = a.q_norms[n * 2 + 1];
        sds[0][tid] = np_ >= kNormEps ? (gp - qp * dp) / np_ : gp / kNormEps;
        sds[1][tid] = nn_ >= kNormEps ? (gn - qn * dn) / nn_ : gn / kNormEps;
    }
    __syncthreads();
    float dot = 0.f;
#pragma unroll
    for (int i = 0; i < CQ; ++i) {
        const int c = w * CQ + i;
        G[i] = G[i] + m * sds[0][c] + mneg * sds[1][c];
        dot += y[i] * G[i];
    }
    sdot[w][lane] = dot;
    __syncthreads();
    const float tot = sdot[0][lane] + sdot[1][lane] + sdot[2][lane] + sdot[3][lane];
    const float inv = a.inv_norm[(int64_t)n * P + xs];
    const bool clamped = inv >= 1.0f / kNormEps;
    float* ob = a.dfeat + n * a.sn;
    const bool staged = a.sc == 1 && a.sp == C && ((reinterpret_cast<uintptr_t>(ob) & 15u) == 0);
    if (staged) {
#pragma unroll
        for (int i = 0; i < CQ; ++i) tile[lane * TP + w * CQ + i] = (clamped ? G[i] : (G[i] - y[i] * tot)) * inv;
        __syncthreads();
        const int npx = min(64, P - t * 64);
        float4* dst = reinterpret_cast<float4*>(ob + (int64_t)t * 64 * C);
        for (int e = tid; e < npx * (C / 4); e += 256) {
            const float* q = tile + (e / (C / 4)) * TP + (e % (C / 4)) * 4;
            dst[e] = make_float4(q[0], q[1], q[2], q[3]);
        }
    } else if (ok) {
        float* o = ob + (int64_t)x * a.sp + (int64_t)(w * CQ) * a.sc;
#pragma unroll
        for (int i = 0; i < CQ; ++i) o[(int64_t)i * a.sc] = (clamped ? G[i] : (G[i] - y[i] * tot)) * inv;
    }
}

CP2_API int cp2_feat_normalize_pool_pair(const float* q_feat, int64_t q_sn, int64_t q_sc, int64_t q_sp, const float* k_feat,
                                         int64_t k_sn, int64_t k_sc, int64_t k_sp, const int64_t* k_row, const float* mask_a,
                                         const float* mask_b, float* q_dense, float* k_dense, float* q_inv_norm, float* q_partial,
                                         float* k_partial, int B, int C, int P, void* stream) {
    if (!q_feat || !k_feat || !mask_a || !mask_b || !q_dense || !k_dense || !q_inv_norm || !q_partial || !k_partial) return CP2_ERR_NULL;
    if (B <= 0 || P <= 0) return CP2_ERR_SHAPE;
    if (C != 128 || B > 65535) return CP2_ERR_UNSUPPORTED;
    const int NT = cp2_cdiv(P, 64);
    FeatPairArgs a{{q_feat, k_feat}, {q_sn, k_sn}, {q_sc, k_sc}, {q_sp, k_sp}, k_row, {mask_a, mask_b}, {q_dense, k_dense},
                   q_inv_norm, {q_partial, k_partial}, B, P, NT};
    hipLaunchKernelGGL(feat_normalize_pool_pair_kernel<128>, dim3(NT, B, 2), dim3(256), 0, cp2_stream(stream), a);
    return cp2_launch_status();
}

CP2_API int cp2_feat_bwd_fused(const float* dense, const float* inv_norm, const float* mask, const float* g_part, int S,
                               int64_t split_stride, const float* drow_pos, const float* dE, int NE, const float* q_pos, const float* q_neg,
                               const float* k_pos, const float* k_neg, const float* q_norms, int include_background, float* dfeat,
                               int64_t stride_n, int64_t stride_c, int64_t stride_p, int B, int C, int P, void* stream) {
    if (!dense || !inv_norm || !mask || !g_part || !drow_pos || !dE || !q_pos || !q_neg || !k_pos || !k_neg || !q_norms || !dfeat)
        return CP2_ERR_NULL;
    if (B <= 0 || P <= 0 || S <= 0 || NE < (include_background ? 3 : 1)) return CP2_ERR_SHAPE;
    if (C != 128 || B > 65535) return CP2_ERR_UNSUPPORTED;
    FeatBwdArgs a{dense, inv_norm, mask, g_part, S, split_stride, drow_pos, dE, NE, q_pos, q_neg, k_pos, k_neg, q_norms, include_background,
                  dfeat, stride_n, stride_c, stride_p, P};
    hipLaunchKernelGGL(feat_bwd_fused_kernel<128>, dim3(cp2_cdiv(P, 64), B), dim3(256), 0, cp2_stream(stream), a);
    return cp2_launch_status();
}

CP2_API int cp2_feat_normalize_pool(const float* feat, int64_t stride_n, int64_t stride_c, int64_t stride_p,
                                    const float* mask, float* dense, float* inv_norm, float* pool_partial, int B,
                                    int C, int P, void* stream) {
    if (!feat || !mask || !dense || !inv_norm || !pool_partial) return CP2_ERR_NULL;
    if (B <= 0 || P <= 0) return CP2_ERR_SHAPE;
    if (C != 128) return CP2_ERR_UNSUPPORTED;
    const int NT = cp2_cdiv(P, 64);
    hipLaunchKernelGGL(feat_normalize_pool_kernel<128>, dim3(NT, B), dim3(256), 0, cp2_stream(stream), feat,
                       stride_n, stride_c, stride_p, mask, dense, inv_norm, pool_partial, P, NT);
    return cp2_launch_status();
}

CP2_API int cp2_pool_finalize(const float* q_partial, const float* k_partial, float* q_pos, float* q_neg,
                              float* q_norms, float* k_pos, float* k_neg, float* extras, int B, int C, int P,
                              void* stream) {
    if (!q_partial || !k_partial || !q_pos || !q_neg || !q_norms || !k_pos || !k_neg || !extras) return CP2_ERR_NULL;
    if (B <= 0 || P <= 0) return CP2_ERR_SHAPE;
    if (C != 128) return CP2_ERR_UNSUPPORTED;
    hipLaunchKernelGGL(pool_finalize_kernel<128>, dim3(B), dim3(128), 0, cp2_stream(stream), q_partial, k_partial,
                       cp2_cdiv(P, 64), q_pos, q_neg, q_norms, k_pos, k_neg, extras);
    return cp2_launch_status();
}

CP2_API int cp2_pool_bwd(const float* drow_pos, const float* dE, const float* q_pos, const float* q_neg,
                         const float* k_pos, const float* k_neg, const float* q_norms, int include_background,
                         float* ds_pos, float* ds_neg, int B, int C, void* stream) {
    if (!drow_pos || !dE || !q_pos || !q_neg || !k_pos || !k_neg || !q_norms || !ds_pos || !ds_neg) return CP2_ERR_NULL;
    if (B <= 0) return CP2_ERR_SHAPE;
    if (C != 128) return CP2_ERR_UNSUPPORTED;
    hipLaunchKernelGGL(pool_bwd_kernel<128>, dim3(B), dim3(128), 0, cp2_stream(stream), drow_pos, dE, q_pos, q_neg,
                       k_pos, k_neg, q_norms, include_background, ds_pos, ds_neg);
    return cp2_launch_status();
}

CP2_API int cp2_feat_bwd(const float* dense, const float* inv_norm, const float* mask, const float* g_dense,
                         const float* ds_pos, const float* ds_neg, float* dfeat, int64_t stride_n, int64_t stride_c,
                         int64_t stride_p, int B, int C, int P, void* stream) {
    if (!dense || !inv_norm || !mask || !g_dense || !ds_pos || !ds_neg || !dfeat) return CP2_ERR_NULL;
    if (B <= 0 || P <= 0) return CP2_ERR_SHAPE;
    if (C != 128) return CP2_ERR_UNSUPPORTED;
    hipLaunchKernelGGL(feat_bwd_kernel<128>, dim3(cp2_cdiv(P, 64), B), dim3(256), 0, cp2_stream(stream), dense,
                       inv_norm, mask, g_dense, ds_pos, ds_neg, dfeat, stride_n, stride_c, stride_p, P);
    return cp2_launch_status();
}
