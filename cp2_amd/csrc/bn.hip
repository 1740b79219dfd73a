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
    int M, C, CG;     // rows, channels, CG = C/8 channel groups (one 16-byte lane each)
    int CGs, NS, RP;  // groups per channel slice (8 or 16), slices = CG/CGs, row lanes per workgroup = 256/CGs
    int rpb, Gr;      // rows per workgroup, row chunks; grid = Gr * NS, blockIdx = chunk * NS + slice
};

// Arguments of the per-channel epilogue kernels (bn_finalize_kernel).
struct BnFin {
    const float* weight; const float* bias;
    float* running_mean; float* running_var;
    float momentum, eps;
    float* o0; float* o1; float* o2; float* o3;   // fwd: scale, shift, save_mean, save_invstd   bwd: dgamma, dbeta, coef[3][C], -
    const float* invstd;                          // bwd only
};

// forward: mean / biased var -> invstd, scale, shift; running statistics as torch.nn.BatchNorm2d
__device__ __forceinline__ void bn_fwd_channel(const BnFin& f, int c, int M, double s, double ss) {
    const double mean = s / M;
    double var = ss / M - mean * mean;
    if (var < 0.0) var = 0.0;
    const float invstd = (float)(1.0 / sqrt(var + (double)f.eps));
    const float gam = f.weight ? f.weight[c] : 1.f, bet = f.bias ? f.bias[c] : 0.f;
    const float sc = gam * invstd;
    f.o0[c] = sc;
    f.o1[c] = bet - (float)mean * sc;
    f.o2[c] = (float)mean;
    f.o3[c] = invstd;
    if (f.running_mean) f.running_mean[c] = (1.f - f.momentum) * f.running_mean[c] + f.momentum * (float)mean;
    if (f.running_var) {
        const double unbiased = M > 1 ? var * (double)M / (double)(M - 1) : var;
        f.running_var[c] = (1.f - f.momentum) * f.running_var[c] + f.momentum * (float)unbiased;
    }
}

// backward: dgamma, dbeta and the three per-channel coefficients of dx
__device__ __forceinline__ void bn_bwd_channel(const BnFin& f, int c, int M, int C, double s, double sx) {
    if (f.o1) f.o1[c] = (float)s;
    if (f.o0) f.o0[c] = (float)sx;
    f.o2[c] = (f.weight ? f.weight[c] : 1.f) * f.invstd[c];
    f.o2[C + c] = (float)(s / M);
    f.o2[2 * C + c] = (float)(sx / M);
}

// ---------------------------------------------------------------------------------------------------------------
// statistics: workgroup (chunk, slice) -> two per-channel sums over rows [chunk*rpb, ...) of the slice's 64 / 128
// channels, part[chunk][2][C].   MODE 0: (x, x^2)   MODE 1: (g, g*xhat), g = dy * (y > 0)
// (Letting the last-arriving workgroup of a slice run the epilogue in the same launch was measured and rejected: the
// agent-scope release it needs is a buffer_wbl2 per workgroup, 46 us per launch instead of 7.  So was letting the
// apply workgroups of small layers add the partials themselves: the load -> LDS -> math -> LDS prologue costs what the
// finalize launch costs, 5.6 -> 9.4 us per forward apply, and the 32-chunk cap it needs slows the statistics kernels.
// Round 4 measured a third form for the 14 x 14 maps (M = 6272): ONE launch in which a 512-thread workgroup owns 16
// channels over all rows -- the slab held in registers, statistics, epilogue and apply without a partner workgroup.
// Bit-level tests green, and the training step went from 12.53 to 13.72 ms: C / 16 = 16 ... 128 workgroups pulling 32-byte
// pieces of 512 ... 4096-byte rows reach a fraction of a CU's bandwidth, so each layer took several times the 16-18 us of
// the three chip-wide launches it replaced.  Removed; what these layers need is the statistics out of the producing
// convolution's epilogue.)
// ---------------------------------------------------------------------------------------------------------------
template <int MODE, bool RELU>
__global__ __launch_bounds__(256) void bn_stats_kernel(const u16x8* __restrict__ x, const u16x8* __restrict__ dy,
                                                       const u16x8* __restrict__ y, const float* __restrict__ mean,
                                                       const float* __restrict__ invstd, float* __restrict__ part,
                                                       BnGeom g) {
    __shared__ float red[256][17];
    constexpr int U = MODE == 0 ? 4 : 2;          // rows in flight per thread
    const int tid = threadIdx.x, cgl = tid % g.CGs, rl = tid / g.CGs;
    const int slice = blockIdx.x % g.NS, chunk = blockIdx.x / g.NS;
    const int cg = slice * g.CGs + cgl;
    float s0[8], s1[8], mu[8], is[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) { s0[j] = 0.f; s1[j] = 0.f; mu[j] = 0.f; is[j] = 1.f; }
    if (MODE == 1) {
#pragma unroll
        for (int j = 0; j < 8; ++j) { mu[j] = mean[cg * 8 + j]; is[j] = invstd[cg * 8 + j]; }
    }
    const int r0 = chunk * g.rpb;
    const int r1 = min(g.M, r0 + g.rpb);
    for (int r = r0 + rl; r < r1; r += g.RP * U) {
        u16x8 xv[U], dv[U], yv[U];
        bool ok[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int rr = r + u * g.RP;
            ok[u] = rr < r1;
            const int64_t o = (int64_t)(ok[u] ? rr : r) * g.CG + cg;
            xv[u] = x[o];
            if (MODE == 1) { dv[u] = dy[o]; if (RELU) yv[u] = y[o]; }
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            if (MODE == 0) {
#pragma unroll
                for (int j = 0; j < 8; ++j) { const float v = ok[u] ? bf2f(xv[u][j]) : 0.f; s0[j] += v; s1[j] += v * v; }
            } else {
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    float gg = ok[u] ? bf2f(dv[u][j]) : 0.f;
                    if (RELU && !(bf2f(yv[u][j]) > 0.f)) gg = 0.f;
                    s0[j] += gg;
                    s1[j] += gg * (bf2f(xv[u][j]) - mu[j]) * is[j];
                }
            }
        }
    }
#pragma unroll
    for (int j = 0; j < 8; ++j) { red[tid][j] = s0[j]; red[tid][8 + j] = s1[j]; }
    __syncthreads();
    const int SC = g.CGs * 8;                      // channels of this slice
    if (tid < 2 * SC) {                            // one (sum, channel) pair per thread, row lanes added in order
        const int s = tid / SC, ch = tid % SC;
        float acc = 0.f;
        for (int k = 0; k < g.RP; ++k) acc += red[k * g.CGs + (ch >> 3)][s * 8 + (ch & 7)];
        part[((int64_t)chunk * 2 + s) * g.C + slice * SC + ch] = acc;
    }
}

// Per-channel epilogue: a workgroup = 32 channels (8 float4 columns) x 32 partial lanes; every thread has all of its
// <= 8 x 2 partial loads in flight at once (Gr <= 256), lanes are added in a fixed order in fp64.
template <int MODE>
__global__ __launch_bounds__(256) void bn_finalize_kernel(const float* __restrict__ part, BnGeom g, BnFin f) {
    __shared__ double fin[256 * 8];
    const int tid = threadIdx.x, c4 = tid & 7, kl = tid >> 3;
    const int64_t rowq = g.C / 4;
    const float4* p4 = reinterpret_cast<const float4*>(part) + blockIdx.x * 8 + c4;
    float4 va[8], vb[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) {
        const int k = kl + u * 32;
        const int64_t kc = k < g.Gr ? k : 0;
        va[u] = p4[(kc * 2 + 0) * rowq];
        vb[u] = p4[(kc * 2 + 1) * rowq];
    }
    double a[4] = {0.0, 0.0, 0.0, 0.0}, b[4] = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
    for (int u = 0; u < 8; ++u) {
        if (kl + u * 32 < g.Gr) {
            a[0] += va[u].x; a[1] += va[u].y; a[2] += va[u].z; a[3] += va[u].w;
            b[0] += vb[u].x; b[1] += vb[u].y; b[2] += vb[u].z; b[3] += vb[u].w;
        }
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) { fin[tid * 8 + j] = a[j]; fin[tid * 8 + 4 + j] = b[j]; }
    __syncthreads();
    if (tid < 32) {
        double s = 0.0, ss = 0.0;
        for (int l = 0; l < 32; ++l) {
            const int idx = (l * 8 + (tid >> 2)) * 8 + (tid & 3);
            s += fin[idx];
            ss += fin[idx + 4];
        }
        const int c = blockIdx.x * 32 + tid;
        if (MODE == 0) bn_fwd_channel(f, c, g.M, s, ss);
        else bn_bwd_channel(f, c, g.M, g.C, s, ss);
    }
}

// forward apply: y = relu(x*scale + shift + residual)
template <bool RELU, bool RES>
__global__ __launch_bounds__(256) void bn_fwd_apply_kernel(const u16x8* __restrict__ x, const u16x8* __restrict__ res,
                                                           const float* __restrict__ scale,
                                                           const float* __restrict__ shift, u16x8* __restrict__ y,
                                                           BnGeom g) {
    constexpr int U = RES ? 2 : 4;
    const int tid = threadIdx.x, cgl = tid % g.CGs, rl = tid / g.CGs;
    const int slice = blockIdx.x % g.NS, chunk = blockIdx.x / g.NS;
    const int cg = slice * g.CGs + cgl;
    float sc[8], sh[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) { sc[j] = scale[cg * 8 + j]; sh[j] = shift[cg * 8 + j]; }
    const int r0 = chunk * g.rpb;
    const int r1 = min(g.M, r0 + g.rpb);
    for (int r = r0 + rl; r < r1; r += g.RP * U) {
        u16x8 xv[U], rv[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int rr = r + u * g.RP;
            const int64_t o = (int64_t)(rr < r1 ? rr : r) * g.CG + cg;
            xv[u] = x[o];
            if (RES) rv[u] = res[o];
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int rr = r + u * g.RP;
            if (rr >= r1) break;
            u16x8 out;
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                float v = bf2f(xv[u][j]) * sc[j] + sh[j];
                if (RES) v += bf2f(rv[u][j]);
                if (RELU) v = fmaxf(v, 0.f);
                out[j] = f2bf(v);
            }
            y[(int64_t)rr * g.CG + cg] = out;
        }
    }
}

// backward apply: dx = c0*(g - c1 - xhat*c2), optional dres = g
template <bool RELU, bool DRES>
__global__ __launch_bounds__(256) void bn_bwd_apply_kernel(const u16x8* __restrict__ x, const u16x8* __restrict__ dy,
                                                           const u16x8* __restrict__ y, const float* __restrict__ mean,
                                                           const float* __restrict__ invstd,
                                                           const float* __restrict__ coef, u16x8* __restrict__ dx,
                                                           u16x8* __restrict__ dres, BnGeom g) {
    constexpr int U = 2;
    const int tid = threadIdx.x, cgl = tid % g.CGs, rl = tid / g.CGs;
    const int slice = blockIdx.x % g.NS, chunk = blockIdx.x / g.NS;
    const int cg = slice * g.CGs + cgl;
    float mu[8], is[8], c0[8], c1[8], c2[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const int c = cg * 8 + j;
        mu[j] = mean[c]; is[j] = invstd[c]; c0[j] = coef[c]; c1[j] = coef[g.C + c]; c2[j] = coef[2 * g.C + c];
    }
    const int r0 = chunk * g.rpb;
    const int r1 = min(g.M, r0 + g.rpb);
    for (int r = r0 + rl; r < r1; r += g.RP * U) {
        u16x8 xv[U], dv[U], yv[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int rr = r + u * g.RP;
            const int64_t o = (int64_t)(rr < r1 ? rr : r) * g.CG + cg;
            xv[u] = x[o];
            dv[u] = dy[o];
            if (RELU) yv[u] = y[o];
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int rr = r + u * g.RP;
            if (rr >= r1) break;
            u16x8 ox, og;
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                float gg = bf2f(dv[u][j]);
                if (RELU && !(bf2f(yv[u][j]) > 0.f)) gg = 0.f;
                const float xh = (bf2f(xv[u][j]) - mu[j]) * is[j];
                ox[j] = f2bf(c0[j] * (gg - c1[j] - xh * c2[j]));
                og[j] = f2bf(gg);
            }
            const int64_t o = (int64_t)rr * g.CG + cg;
            dx[o] = ox;
            if (DRES) dres[o] = og;
        }
    }
}

// Geometry: channel slices of 128 (64 when C is not a multiple of 128) channels, `target` workgroups in all, every
// thread at least 4 rows, at most max_chunks row chunks (bn_finalize_kernel adds at most 256 partials per channel).
static int bn_geom(int M, int C, int target, int max_chunks, BnGeom* g) {
    if (M <= 0 || C <= 0) return CP2_ERR_SHAPE;
    if (C % 64 != 0 || C > 8192) return CP2_ERR_UNSUPPORTED;
    const int CG = C / 8;
    const int CGs = (CG % 16 == 0) ? 16 : 8;
    const int NS = CG / CGs, RP = 256 / CGs;
    int gr = target / NS;
    if (gr > M / (4 * RP)) gr = M / (4 * RP);
    if (gr > max_chunks) gr = max_chunks;
    if (gr < 1) gr = 1;
    int rpb = cp2_cdiv(M, gr);
    rpb = cp2_cdiv(rpb, RP) * RP;
    *g = BnGeom{M, C, CG, CGs, NS, RP, rpb, cp2_cdiv(M, rpb)};
    return CP2_OK;
}
static int bn_geom_stats(int M, int C, BnGeom* g) { return bn_geom(M, C, 1024, 256, g); }
static int bn_geom_apply(int M, int C, BnGeom* g) { return bn_geom(M, C, 2048, 1 << 20, g); }

static bool bn_al(const void* p) { return cp2_aligned16(p); }

// Number of partial-sum slots the workspace `part` needs: part is float [G][2][C].
CP2_API int cp2_bn_num_partials(int M, int C) {
    BnGeom g;
    const int rc = bn_geom_stats(M, C, &g);
    return rc ? rc : g.Gr;
}

CP2_API int cp2_bn_fwd(const void* x, const void* residual, const float* weight, const float* bias,
                       float* running_mean, float* running_var, float momentum, float eps, int relu, void* y,
                       float* save_mean, float* save_invstd, float* part, float* scale_shift, int M, int C,
                       void* stream) {
    if (!x || !y || !save_mean || !save_invstd || !part || !scale_shift) return CP2_ERR_NULL;
    BnGeom gs, ga;
    int rc = bn_geom_stats(M, C, &gs);
    if (rc) return rc;
    bn_geom_apply(M, C, &ga);
    if (!bn_al(x) || !bn_al(y) || !bn_al(part) || (residual && !bn_al(residual))) return CP2_ERR_ALIGN;
    const u16x8* xv = static_cast<const u16x8*>(x);
    const u16x8* rv = static_cast<const u16x8*>(residual);
    hipStream_t s = cp2_stream(stream);
    const BnFin f{weight, bias, running_mean, running_var, momentum, eps, scale_shift, scale_shift + C, save_mean,
                  save_invstd, nullptr};
    hipLaunchKernelGGL((bn_stats_kernel<0, false>), dim3(gs.Gr * gs.NS), dim3(256), 0, s, xv, nullptr, nullptr, nullptr,
                       nullptr, part, gs);
    hipLaunchKernelGGL((bn_finalize_kernel<0>), dim3(C / 32), dim3(256), 0, s, part, gs, f);
    u16x8* yv = static_cast<u16x8*>(y);
    const dim3 grid(ga.Gr * ga.NS);
    if (relu && residual) hipLaunchKernelGGL((bn_fwd_apply_kernel<true, true>), grid, dim3(256), 0, s, xv, rv, scale_shift, scale_shift + C, yv, ga);
    else if (relu) hipLaunchKernelGGL((bn_fwd_apply_kernel<true, false>), grid, dim3(256), 0, s, xv, rv, scale_shift, scale_shift + C, yv, ga);
    else if (residual) hipLaunchKernelGGL((bn_fwd_apply_kernel<false, true>), grid, dim3(256), 0, s, xv, rv, scale_shift, scale_shift + C, yv, ga);
    else hipLaunchKernelGGL((bn_fwd_apply_kernel<false, false>), grid, dim3(256), 0, s, xv, rv, scale_shift, scale_shift + C, yv, ga);
    return cp2_launch_status();
}

CP2_API int cp2_bn_bwd(const void* x, const void* dy, const void* y, const float* weight, const float* save_mean,
                       const float* save_invstd, int relu, void* dx, void* dres, float* dgamma, float* dbeta,
                       float* part, float* coef, int M, int C, void* stream) {
    if (!x || !dy || !dx || !save_mean || !save_invstd || !part || !coef) return CP2_ERR_NULL;
    if (relu && !y) return CP2_ERR_NULL;
    BnGeom gs, ga;
    int rc = bn_geom_stats(M, C, &gs);
    if (rc) return rc;
    bn_geom_apply(M, C, &ga);
    if (!bn_al(x) || !bn_al(dy) || !bn_al(dx) || !bn_al(part) || (y && !bn_al(y)) || (dres && !bn_al(dres))) return CP2_ERR_ALIGN;
    const u16x8 *xv = static_cast<const u16x8*>(x), *dv = static_cast<const u16x8*>(dy), *yv = static_cast<const u16x8*>(y);
    hipStream_t s = cp2_stream(stream);
    const BnFin f{weight, nullptr, nullptr, nullptr, 0.f, 0.f, dgamma, dbeta, coef, nullptr, save_invstd};
    const dim3 sgrid(gs.Gr * gs.NS);
    if (relu) hipLaunchKernelGGL((bn_stats_kernel<1, true>), sgrid, dim3(256), 0, s, xv, dv, yv, save_mean, save_invstd, part, gs);
    else hipLaunchKernelGGL((bn_stats_kernel<1, false>), sgrid, dim3(256), 0, s, xv, dv, yv, save_mean, save_invstd, part, gs);
    hipLaunchKernelGGL((bn_finalize_kernel<1>), dim3(C / 32), dim3(256), 0, s, part, gs, f);
    u16x8 *ox = static_cast<u16x8*>(dx), *og = static_cast<u16x8*>(dres);
    const dim3 grid(ga.Gr * ga.NS);
    if (relu && dres) hipLaunchKernelGGL((bn_bwd_apply_kernel<true, true>), grid, dim3(256), 0, s, xv, dv, yv, save_mean, save_invstd, coef, ox, og, ga);
    else if (relu) hipLaunchKernelGGL((bn_bwd_apply_kernel<true, false>), grid, dim3(256), 0, s, xv, dv, yv, save_mean, save_invstd, coef, ox, og, ga);
    else if (dres) hipLaunchKernelGGL((bn_bwd_apply_kernel<false, true>), grid, dim3(256), 0, s, xv, dv, yv, save_mean, save_invstd, coef, ox, og, ga);
    else hipLaunchKernelGGL((bn_bwd_apply_kernel<false, false>), grid, dim3(256), 0, s, xv, dv, yv, save_mean, save_invstd, coef, ox, og, ga);
    return cp2_launch_status();
}
