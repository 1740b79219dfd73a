// Fused training-mode BatchNorm (+ residual add) (+ ReLU) for channels-last bf16 activations.
//
// Not one of the reference's own call sites: the reference leaves BN / ReLU / residual adds to mmseg's
// ResNet (mmseg_/models/backbones/resnet.py:267-304).  After the convolutions were tuned (MIOpen find mode)
// these memory-bound ops are half of the step in MIOpen / ATen kernels (3 + 1 + 1 launches per BN block forward,
// 3 + 1 backward), so the encoder gets an optional fast path: 3 launches forward, 3 backward, every tensor read
// as 16-byte lanes of 8 bf16 channels, statistics in fp32 (final combine in fp64).
//   forward : stats  (sum, sum of squares per channel)      -> finalize (mean, invstd, running stats, scale/shift)
//             apply   y = relu(x*scale + shift + residual)
//   backward: stats  (sum g, sum g*xhat),  g = dy * (y > 0)  -> finalize (dgamma, dbeta, per-channel coefficients)
//             apply   dx = gamma*invstd*(g - mean(g) - xhat*mean(g*xhat));  dres = g
// Layout: x, y, dy, dx, residual are [M = N*H*W][C] bf16 (channels-last memory of an NCHW tensor), C % 8 == 0, C <= 2048.
#include "common.hpp"
#include <math.h>

typedef unsigned short u16x8 __attribute__((ext_vector_type(8)));

__device__ __forceinline__ float bf2f(unsigned short u) { return __uint_as_float(((unsigned)u) << 16); }
__device__ __forceinline__ unsigned short f2bf(float f) {   // round to nearest even; NaN stays NaN
    unsigned u = __float_as_uint(f);
    if ((u & 0x7fffffffu) > 0x7f800000u) return (unsigned short)((u >> 16) | 0x40u);
    u += 0x7fffu + ((u >> 16) & 1u);
    return (unsigned short)(u >> 16);
}

struct BnGeom {
    int M, C, CG, CGb, RP, rows_per_block;   // CG = C/8 channel groups; CGb = min(CG,256) groups per pass; RP = 256/CGb rows per pass
};

// ---------------------------------------------------------------------------------------------------------------
// statistics: two per-channel sums over the rows of this workgroup; MODE 0: (x, x^2)   MODE 1: (g, g*xhat)
// ---------------------------------------------------------------------------------------------------------------
template <int MODE, bool RELU>
__global__ __launch_bounds__(256) void bn_stats_kernel(const u16x8* __restrict__ x, const u16x8* __restrict__ dy,
                                                       const u16x8* __restrict__ y, const float* __restrict__ mean,
                                                       const float* __restrict__ invstd, float* __restrict__ part,
                                                       BnGeom g) {
    __shared__ float red[256][17];
    const int tid = threadIdx.x, cg = tid % g.CGb, rl = tid / g.CGb;
    float s0[8], s1[8], mu[8], is[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) { s0[j] = 0.f; s1[j] = 0.f; mu[j] = 0.f; is[j] = 1.f; }
    if (MODE == 1) {
#pragma unroll
        for (int j = 0; j < 8; ++j) { mu[j] = mean[cg * 8 + j]; is[j] = invstd[cg * 8 + j]; }
    }
    const int r0 = blockIdx.x * g.rows_per_block;
    const int r1 = min(g.M, r0 + g.rows_per_block);
    for (int r = r0 + rl; r < r1; r += g.RP) {
        const int64_t o = (int64_t)r * g.CG + cg;
        const u16x8 xv = x[o];
        if (MODE == 0) {
#pragma unroll
            for (int j = 0; j < 8; ++j) { const float f = bf2f(xv[j]); s0[j] += f; s1[j] += f * f; }
        } else {
            const u16x8 dv = dy[o];
            u16x8 yv = dv;
            if (RELU) yv = y[o];
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                float gg = bf2f(dv[j]);
                if (RELU && !(bf2f(yv[j]) > 0.f)) gg = 0.f;
                s0[j] += gg;
                s1[j] += gg * (bf2f(xv[j]) - mu[j]) * is[j];
            }
        }
    }
#pragma unroll
    for (int j = 0; j < 8; ++j) { red[tid][j] = s0[j]; red[tid][8 + j] = s1[j]; }
    __syncthreads();
    if (rl == 0) {
        for (int k = 1; k < g.RP; ++k) {
#pragma unroll
            for (int j = 0; j < 8; ++j) { s0[j] += red[k * g.CGb + cg][j]; s1[j] += red[k * g.CGb + cg][8 + j]; }
        }
        float* p0 = part + ((int64_t)blockIdx.x * 2 + 0) * g.C + cg * 8;
        float* p1 = part + ((int64_t)blockIdx.x * 2 + 1) * g.C + cg * 8;
#pragma unroll
        for (int j = 0; j < 8; ++j) { p0[j] = s0[j]; p1[j] = s1[j]; }
    }
}

// Both finalize kernels: a workgroup = 32 channels x 32 "partial lanes"; lane gl sums the partials k = gl, gl+32, ...
// (a serial loop over all partials with one thread per channel was pure load latency: hundreds of us per layer).
__device__ __forceinline__ void bn_reduce_partials(const float* __restrict__ part, int G, int C, int c, int gl,
                                                   double (&red)[2][32][33], double* s0, double* s1) {
    double a = 0.0, b = 0.0;
    if (c < C) {
        for (int k = gl; k < G; k += 32) {
            a += part[((int64_t)k * 2 + 0) * C + c];
            b += part[((int64_t)k * 2 + 1) * C + c];
        }
    }
    red[0][gl][threadIdx.x & 31] = a;
    red[1][gl][threadIdx.x & 31] = b;
    __syncthreads();
    a = 0.0; b = 0.0;
    for (int k = 0; k < 32; ++k) { a += red[0][k][threadIdx.x & 31]; b += red[1][k][threadIdx.x & 31]; }
    *s0 = a; *s1 = b;
}

// forward finalize: per channel mean / biased var -> invstd, scale, shift; running statistics as torch.nn.BatchNorm2d
__global__ __launch_bounds__(1024) void bn_fwd_finalize_kernel(const float* __restrict__ part, int G, int M, int C,
                                                               const float* __restrict__ weight,
                                                               const float* __restrict__ bias, float* running_mean,
                                                               float* running_var, float momentum, float eps,
                                                               float* __restrict__ scale, float* __restrict__ shift,
                                                               float* __restrict__ save_mean,
                                                               float* __restrict__ save_invstd) {
    __shared__ double red[2][32][33];
    const int c = blockIdx.x * 32 + (threadIdx.x & 31), gl = threadIdx.x >> 5;
    double s, ss;
    bn_reduce_partials(part, G, C, c, gl, red, &s, &ss);
    if (gl != 0 || c >= C) return;
    const double mean = s / M;
    double var = ss / M - mean * mean;
    if (var < 0.0) var = 0.0;
    const float invstd = (float)(1.0 / sqrt(var + (double)eps));
    const float gam = weight ? weight[c] : 1.f, bet = bias ? bias[c] : 0.f;
    const float sc = gam * invstd;
    scale[c] = sc;
    shift[c] = bet - (float)mean * sc;
    save_mean[c] = (float)mean;
    save_invstd[c] = invstd;
    if (running_mean) running_mean[c] = (1.f - momentum) * running_mean[c] + momentum * (float)mean;
    if (running_var) {
        const double unbiased = M > 1 ? var * (double)M / (double)(M - 1) : var;
        running_var[c] = (1.f - momentum) * running_var[c] + momentum * (float)unbiased;
    }
}

// backward finalize: dgamma, dbeta and the three per-channel coefficients of dx
__global__ __launch_bounds__(1024) void bn_bwd_finalize_kernel(const float* __restrict__ part, int G, int M, int C,
                                                               const float* __restrict__ weight,
                                                               const float* __restrict__ invstd,
                                                               float* __restrict__ dgamma, float* __restrict__ dbeta,
                                                               float* __restrict__ coef /*[3][C]*/) {
    __shared__ double red[2][32][33];
    const int c = blockIdx.x * 32 + (threadIdx.x & 31), gl = threadIdx.x >> 5;
    double s, sx;
    bn_reduce_partials(part, G, C, c, gl, red, &s, &sx);
    if (gl != 0 || c >= C) return;
    if (dbeta) dbeta[c] = (float)s;
    if (dgamma) dgamma[c] = (float)sx;
    coef[c] = (weight ? weight[c] : 1.f) * invstd[c];
    coef[C + c] = (float)(s / M);
    coef[2 * C + c] = (float)(sx / M);
}

// forward apply: y = relu(x*scale + shift + residual)
template <bool RELU, bool RES>
__global__ __launch_bounds__(256) void bn_fwd_apply_kernel(const u16x8* __restrict__ x, const u16x8* __restrict__ res,
                                                           const float* __restrict__ scale,
                                                           const float* __restrict__ shift, u16x8* __restrict__ y,
                                                           BnGeom g) {
    const int tid = threadIdx.x, cg = tid % g.CGb, rl = tid / g.CGb;
    float sc[8], sh[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) { sc[j] = scale[cg * 8 + j]; sh[j] = shift[cg * 8 + j]; }
    const int r0 = blockIdx.x * g.rows_per_block;
    const int r1 = min(g.M, r0 + g.rows_per_block);
    for (int r = r0 + rl; r < r1; r += g.RP) {
        const int64_t o = (int64_t)r * g.CG + cg;
        const u16x8 xv = x[o];
        u16x8 rv = xv;
        if (RES) rv = res[o];
        u16x8 out;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            float f = bf2f(xv[j]) * sc[j] + sh[j];
            if (RES) f += bf2f(rv[j]);
            if (RELU) f = fmaxf(f, 0.f);
            out[j] = f2bf(f);
        }
        y[o] = out;
    }
}

// backward apply: dx = c0*(g - c1 - xhat*c2), optional dres = g
template <bool RELU, bool DRES>
__global__ __launch_bounds__(256) void bn_bwd_apply_kernel(const u16x8* __restrict__ x, const u16x8* __restrict__ dy,
                                                           const u16x8* __restrict__ y, const float* __restrict__ mean,
                                                           const float* __restrict__ invstd,
                                                           const float* __restrict__ coef, u16x8* __restrict__ dx,
                                                           u16x8* __restrict__ dres, BnGeom g) {
    const int tid = threadIdx.x, cg = tid % g.CGb, rl = tid / g.CGb;
    float mu[8], is[8], c0[8], c1[8], c2[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const int c = cg * 8 + j;
        mu[j] = mean[c]; is[j] = invstd[c]; c0[j] = coef[c]; c1[j] = coef[g.C + c]; c2[j] = coef[2 * g.C + c];
    }
    const int r0 = blockIdx.x * g.rows_per_block;
    const int r1 = min(g.M, r0 + g.rows_per_block);
    for (int r = r0 + rl; r < r1; r += g.RP) {
        const int64_t o = (int64_t)r * g.CG + cg;
        const u16x8 xv = x[o], dv = dy[o];
        u16x8 yv = dv;
        if (RELU) yv = y[o];
        u16x8 ox, og;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            float gg = bf2f(dv[j]);
            if (RELU && !(bf2f(yv[j]) > 0.f)) gg = 0.f;
            const float xh = (bf2f(xv[j]) - mu[j]) * is[j];
            ox[j] = f2bf(c0[j] * (gg - c1[j] - xh * c2[j]));
            og[j] = f2bf(gg);
        }
        dx[o] = ox;
        if (DRES) dres[o] = og;
    }
}

static int bn_geom(int M, int C, BnGeom* g, int* G) {
    if (M <= 0 || C <= 0) return CP2_ERR_SHAPE;
    if (C % 8 != 0 || C > 2048) return CP2_ERR_UNSUPPORTED;
    const int CG = C / 8;
    int CGb = 1;
    while (CGb < CG) CGb <<= 1;           // threads of one pass cover a power-of-two number of groups
    if (CGb != CG) return CP2_ERR_UNSUPPORTED;   // C must be 8 * 2^k (64, 128, 256, 512, 1024, 2048)
    const int RP = 256 / CGb;
    // about 512 workgroups (2 per CU), each a whole number of passes
    int rpb = cp2_cdiv(M, 512);
    rpb = cp2_cdiv(rpb, RP) * RP;
    if (rpb < RP) rpb = RP;
    *g = BnGeom{M, C, CG, CGb, RP, rpb};
    *G = cp2_cdiv(M, rpb);
    return CP2_OK;
}

static bool bn_al(const void* p) { return cp2_aligned16(p); }

// Number of partial-sum slots the workspace `part` needs: part is float [G][2][C].
CP2_API int cp2_bn_num_partials(int M, int C) {
    BnGeom g; int G;
    const int rc = bn_geom(M, C, &g, &G);
    return rc ? rc : G;
}

CP2_API int cp2_bn_fwd(const void* x, const void* residual, const float* weight, const float* bias,
                       float* running_mean, float* running_var, float momentum, float eps, int relu, void* y,
                       float* save_mean, float* save_invstd, float* part, float* scale_shift, int M, int C,
                       void* stream) {
    if (!x || !y || !save_mean || !save_invstd || !part || !scale_shift) return CP2_ERR_NULL;
    BnGeom g; int G;
    int rc = bn_geom(M, C, &g, &G);
    if (rc) return rc;
    if (!bn_al(x) || !bn_al(y) || (residual && !bn_al(residual))) return CP2_ERR_ALIGN;
    const u16x8* xv = static_cast<const u16x8*>(x);
    const u16x8* rv = static_cast<const u16x8*>(residual);
    hipStream_t s = cp2_stream(stream);
    hipLaunchKernelGGL((bn_stats_kernel<0, false>), dim3(G), dim3(256), 0, s, xv, nullptr, nullptr, nullptr, nullptr, part, g);
    hipLaunchKernelGGL(bn_fwd_finalize_kernel, dim3(cp2_cdiv(C, 32)), dim3(1024), 0, s, part, G, M, C, weight, bias,
                       running_mean, running_var, momentum, eps, scale_shift, scale_shift + C, save_mean, save_invstd);
    u16x8* yv = static_cast<u16x8*>(y);
    if (relu && residual) hipLaunchKernelGGL((bn_fwd_apply_kernel<true, true>), dim3(G), dim3(256), 0, s, xv, rv, scale_shift, scale_shift + C, yv, g);
    else if (relu) hipLaunchKernelGGL((bn_fwd_apply_kernel<true, false>), dim3(G), dim3(256), 0, s, xv, rv, scale_shift, scale_shift + C, yv, g);
    else if (residual) hipLaunchKernelGGL((bn_fwd_apply_kernel<false, true>), dim3(G), dim3(256), 0, s, xv, rv, scale_shift, scale_shift + C, yv, g);
    else hipLaunchKernelGGL((bn_fwd_apply_kernel<false, false>), dim3(G), dim3(256), 0, s, xv, rv, scale_shift, scale_shift + C, yv, g);
    return cp2_launch_status();
}

CP2_API int cp2_bn_bwd(const void* x, const void* dy, const void* y, const float* weight, const float* save_mean,
                       const float* save_invstd, int relu, void* dx, void* dres, float* dgamma, float* dbeta,
                       float* part, float* coef, int M, int C, void* stream) {
    if (!x || !dy || !dx || !save_mean || !save_invstd || !part || !coef) return CP2_ERR_NULL;
    if (relu && !y) return CP2_ERR_NULL;
    BnGeom g; int G;
    int rc = bn_geom(M, C, &g, &G);
    if (rc) return rc;
    if (!bn_al(x) || !bn_al(dy) || !bn_al(dx) || (y && !bn_al(y)) || (dres && !bn_al(dres))) return CP2_ERR_ALIGN;
    const u16x8 *xv = static_cast<const u16x8*>(x), *dv = static_cast<const u16x8*>(dy), *yv = static_cast<const u16x8*>(y);
    hipStream_t s = cp2_stream(stream);
    if (relu) hipLaunchKernelGGL((bn_stats_kernel<1, true>), dim3(G), dim3(256), 0, s, xv, dv, yv, save_mean, save_invstd, part, g);
    else hipLaunchKernelGGL((bn_stats_kernel<1, false>), dim3(G), dim3(256), 0, s, xv, dv, yv, save_mean, save_invstd, part, g);
    hipLaunchKernelGGL(bn_bwd_finalize_kernel, dim3(cp2_cdiv(C, 32)), dim3(1024), 0, s, part, G, M, C, weight, save_invstd,
                       dgamma, dbeta, coef);
    u16x8 *ox = static_cast<u16x8*>(dx), *og = static_cast<u16x8*>(dres);
    if (relu && dres) hipLaunchKernelGGL((bn_bwd_apply_kernel<true, true>), dim3(G), dim3(256), 0, s, xv, dv, yv, save_mean, save_invstd, coef, ox, og, g);
    else if (relu) hipLaunchKernelGGL((bn_bwd_apply_kernel<true, false>), dim3(G), dim3(256), 0, s, xv, dv, yv, save_mean, save_invstd, coef, ox, og, g);
    else if (dres) hipLaunchKernelGGL((bn_bwd_apply_kernel<false, true>), dim3(G), dim3(256), 0, s, xv, dv, yv, save_mean, save_invstd, coef, ox, og, g);
    else hipLaunchKernelGGL((bn_bwd_apply_kernel<false, false>), dim3(G), dim3(256), 0, s, xv, dv, yv, save_mean, save_invstd, coef, ox, og, g);
    return cp2_launch_status();
}
