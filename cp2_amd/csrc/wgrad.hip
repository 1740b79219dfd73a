// Weight gradient of a 1x1 stride-1 convolution on channels-last bf16 activations (encoder fast path, not a reference
// call site: mmseg's ResNet leaves it to the framework, mmseg_/models/backbones/resnet.py:267-304):
//     dW[co][ci] = sum_m dY[m][co] * X[m][ci],      m = (n, h, w),  dY: [M][CO] bf16,  X: [M][CI] bf16,  dW: [CO][CI] fp32
// Both operands have the reduction index m as their SLOW memory index, so neither can feed the matrix cores by rows.
// The tiles go into LDS exactly as they lie in memory ([m][channel], 16-byte copies) and are read back with
// ds_read_b64_tr_b16, gfx950's transposing LDS read: a lane receives 4 consecutive m of one channel, two reads make
// the 8-deep k slice v_mfma_f32_32x32x16_bf16 wants (cdna_hip_programming.md T10).
// Work decomposition: workgroup = (output tile of 64x64 or 128x128 channels, split of the m range); partial tiles are
// written as fp32 [S][CO][CI] and added in a fixed order by wgrad_sum_kernel -- deterministic (MIOpen's split-K solvers
// for this case zero a workspace, accumulate with atomics and cast: 3 launches and run-to-run different results) and
// the fp32 result is the gradient of the fp32 master weight directly (no bf16 round trip, no cast kernel).
// HBM / L2 bound: 2 bytes * M * (CO + CI) per output-tile row/column pass.
#include "common.hpp"
#include <stdlib.h>

typedef float wg_f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 wg_bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 wg_bf16x4 __attribute__((ext_vector_type(4)));
typedef unsigned int wg_u32x4 __attribute__((ext_vector_type(4)));

constexpr int WG_KM = 32;                     // m rows per LDS stage (two MFMA k-steps of 16)

__device__ __forceinline__ int wg_rho(int reg, int h) { return (reg & 3) + 8 * (reg >> 2) + 4 * h; }

// 32 channels x 16 m operand of v_mfma_f32_32x32x16_bf16 from a [m][channel] LDS tile: lane -> channel chbase + lane%32,
// m = ks + 8*(lane/32) + 0..7.  Per 16-lane group one transposing read per 4 m rows.
template <int PITCH>
__device__ __forceinline__ wg_bf16x8 wg_frag(const unsigned short* tile, int ks, int chbase, int lane) {
    const int g = lane >> 4, i = lane & 15, q = i >> 2, p = i & 3;
    const unsigned short* a0 = tile + (ks + 8 * (g >> 1) + q) * PITCH + chbase + 16 * (g & 1) + 4 * p;
    typedef wg_bf16x4 __attribute__((address_space(3))) * lds4_t;
    const wg_bf16x4 v0 = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds4_t)(a0));
    const wg_bf16x4 v1 = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds4_t)(a0 + 4 * PITCH));
    return __builtin_shufflevector(v0, v1, 0, 1, 2, 3, 4, 5, 6, 7);
}

// Workgroups are dealt round-robin over the 8 XCDs (blockIdx % 8).  All output tiles of one m-split read the same rows
// of dY and X, so a split's tiles should share one XCD's L2: linear id -> item such that each XCD owns a contiguous run
// of (split, tile) items, split-major (bijective for any grid size; MI355X guide T1).  Without it every XCD pulls every
// row range through its own L2 (8x the fill traffic on the wide layers).
__device__ __forceinline__ int wg_xcd_item(int id, int n_items) {
    const int q = n_items / 8, rem = n_items % 8, xcd = id % 8, j = id / 8;
    return (xcd < rem ? xcd * (q + 1) : rem * (q + 1) + (xcd - rem) * q) + j;
}

// k x k convolution (CONV): the same product per filter tap, dW[co][kh][kw][ci] = sum_m dY[m][co] * X[row(m, kh, kw)][ci] with
// m = (n, oh, ow) an OUTPUT pixel and row(m, kh, kw) the input pixel (oh*stride - pad + kh*dil, ow*stride - pad + kw*dil) of
// image n, or a zero row outside the image.  A workgroup owns (co tile, tap, ci tile); its threads track (n, oh, ow) of
// their rows incrementally from stage to stage (one division when the workgroup starts).
struct WgConv { int H, W, OH, OW, KW, taps, stride, pad, dil; };

template <int F, bool CONV>   // F x F MFMA tiles per wave; workgroup tile = 64F x 64F channels, 4 waves as 2 x 2
__global__ __launch_bounds__(256) void wgrad1x1_kernel(const unsigned short* __restrict__ dy,
                                                       const unsigned short* __restrict__ x, float* __restrict__ part,
                                                       int M, int CO, int CI, int rows_per_split, int tiles_ci, int tiles, WgConv cv) {
    constexpr int T = 64 * F, PITCH = T + 8, CPR = T / 8;            // channels per tile edge, LDS row pitch, 16-B chunks per row
    __shared__ __attribute__((aligned(16))) unsigned short lds[2][2][WG_KM * PITCH];
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6, wy = wid >> 1, wx = wid & 1;
    const int r = lane & 31, h = lane >> 5;
    const int item = wg_xcd_item(blockIdx.x, gridDim.x), split = item / tiles, tile = item % tiles;
    const int taps = CONV ? cv.taps : 1;
    const int tap = CONV ? (tile / tiles_ci) % taps : 0;
    const int co0 = (tile / (tiles_ci * taps)) * T, ci0 = (tile % tiles_ci) * T;
    const int m_begin = split * rows_per_split, m_end = min(M, m_begin + rows_per_split);
    // CONV: (image, output row, output column) of this thread's F rows of the current stage, and the tap's input offset
    int pn[F], poh[F], pow_[F];
    const int dh = CONV ? (tap / cv.KW) * cv.dil - cv.pad : 0, dw_ = CONV ? (tap % cv.KW) * cv.dil - cv.pad : 0;
    if (CONV) {
#pragma unroll
        for (int j = 0; j < F; ++j) {
            const int m = m_begin + (tid + j * 256) / CPR;
            const int per = cv.OH * cv.OW;
            pn[j] = m / per;
            const int rem = m - pn[j] * per;
            poh[j] = rem / cv.OW;
            pow_[j] = rem - poh[j] * cv.OW;
        }
    }
    wg_f32x16 acc[F][F];
#pragma unroll
    for (int a = 0; a < F; ++a)
#pragma unroll
        for (int b = 0; b < F; ++b) acc[a][b] = (wg_f32x16){0};
    wg_u32x4 ra[F], rb[F];
    auto load = [&](int m0) {
#pragma unroll
        for (int j = 0; j < F; ++j) {
            const int c = tid + j * 256, row = c / CPR, col = (c % CPR) * 8, m = m0 + row;
            const bool ok = m < m_end;
            ra[j] = ok ? *reinterpret_cast<const wg_u32x4*>(dy + (int64_t)m * CO + co0 + col) : (wg_u32x4){0, 0, 0, 0};
            if (CONV) {
                const int ih = poh[j] * cv.stride + dh, iw = pow_[j] * cv.stride + dw_;
                const bool in = ok && ih >= 0 && ih < cv.H && iw >= 0 && iw < cv.W;
                rb[j] = in ? *reinterpret_cast<const wg_u32x4*>(x + (((int64_t)pn[j] * cv.H + ih) * cv.W + iw) * CI + ci0 + col)
                           : (wg_u32x4){0, 0, 0, 0};
                pow_[j] += WG_KM;                              // the same thread's row of the next stage: m + WG_KM
                while (pow_[j] >= cv.OW) { pow_[j] -= cv.OW; ++poh[j]; }
                while (poh[j] >= cv.OH) { poh[j] -= cv.OH; ++pn[j]; }
            } else {
                rb[j] = ok ? *reinterpret_cast<const wg_u32x4*>(x + (int64_t)m * CI + ci0 + col) : (wg_u32x4){0, 0, 0, 0};
            }
        }
    };
    auto store = [&](int buf) {
#pragma unroll
        for (int j = 0; j < F; ++j) {
            const int c = tid + j * 256, row = c / CPR, col = (c % CPR) * 8;
            *reinterpret_cast<wg_u32x4*>(&lds[buf][0][row * PITCH + col]) = ra[j];
            *reinterpret_cast<wg_u32x4*>(&lds[buf][1][row * PITCH + col]) = rb[j];
        }
    };
    int buf = 0;
    if (m_begin < m_end) { load(m_begin); store(0); }
    __syncthreads();
    for (int m0 = m_begin; m0 < m_end; m0 += WG_KM) {
        const bool more = m0 + WG_KM < m_end;
        if (more) load(m0 + WG_KM);                       // next stage in flight during the MFMAs
#pragma unroll
        for (int ks = 0; ks < WG_KM; ks += 16) {
            wg_bf16x8 fa[F], fb[F];
#pragma unroll
            for (int a = 0; a < F; ++a) fa[a] = wg_frag<PITCH>(lds[buf][0], ks, wy * 32 * F + a * 32, lane);
#pragma unroll
            for (int b = 0; b < F; ++b) fb[b] = wg_frag<PITCH>(lds[buf][1], ks, wx * 32 * F + b * 32, lane);
#pragma unroll
            for (int a = 0; a < F; ++a)
#pragma unroll
                for (int b = 0; b < F; ++b) acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[a], fb[b], acc[a][b], 0, 0, 0);
        }
        if (more) store(buf ^ 1);
        __syncthreads();
        buf ^= 1;
    }
    float* out = part + (int64_t)split * CO * CI * taps;
#pragma unroll
    for (int a = 0; a < F; ++a)
#pragma unroll
        for (int b = 0; b < F; ++b)
#pragma unroll
            for (int reg = 0; reg < 16; ++reg)
                out[((int64_t)(co0 + wy * 32 * F + a * 32 + wg_rho(reg, h)) * taps + tap) * CI + ci0 + wx * 32 * F + b * 32 + r] = acc[a][b][reg];
}

// dw = sum over splits, in split order per element: 16 float4 columns x 16 split lanes per workgroup, 8 loads in flight
// per thread (a serial loop over S = 600 partials per element was 150 us of load latency).
__global__ __launch_bounds__(256) void wgrad_sum_kernel(const float* __restrict__ part, float* __restrict__ dw, int S, int64_t n4) {
    __shared__ float4 red[16][16];
    const int col = threadIdx.x & 15, sl = threadIdx.x >> 4;
    const int64_t i = (int64_t)blockIdx.x * 16 + col;
    const float4* p = reinterpret_cast<const float4*>(part);
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    if (i < n4) {
        for (int s0 = sl; s0 < S; s0 += 16 * 8) {
            float4 v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int s = s0 + 16 * u;
                v[u] = s < S ? p[(int64_t)s * n4 + i] : make_float4(0.f, 0.f, 0.f, 0.f);
            }
#pragma unroll
            for (int u = 0; u < 8; ++u) { acc.x += v[u].x; acc.y += v[u].y; acc.z += v[u].z; acc.w += v[u].w; }
        }
    }
    red[sl][col] = acc;
    __syncthreads();
    if (sl == 0 && i < n4) {
        float4 t = red[0][col];
#pragma unroll
        for (int j = 1; j < 16; ++j) { t.x += red[j][col].x; t.y += red[j][col].y; t.z += red[j][col].z; t.w += red[j][col].w; }
        reinterpret_cast<float4*>(dw)[i] = t;
    }
}

static int wgrad_geom(int M, int CO, int CI, int* F, int* S, int* rows, int taps = 1) {
    if (M <= 0 || CO <= 0 || CI <= 0 || taps <= 0) return CP2_ERR_SHAPE;
    if (CO % 64 != 0 || CI % 64 != 0) return CP2_ERR_UNSUPPORTED;
    *F = (CO % 128 == 0 && CI % 128 == 0) ? 2 : 1;
    const int T = 64 * *F;
    const int64_t tiles = (int64_t)(CO / T) * (CI / T) * taps;
    static int target = 0, target_conv = 0;                // tuning knobs (workgroups in all): CP2_WGRAD_TARGET / _CONV
    if (!target) {
        const char* e = getenv("CP2_WGRAD_TARGET");
        const char* c = getenv("CP2_WGRAD_TARGET_CONV");
        target = e ? atoi(e) : 512;
        target_conv = c ? atoi(c) : 512;
        if (target < 1) target = 512;
        if (target_conv < 1) target_conv = 512;
    }
    const int tg = taps > 1 ? target_conv : target;
    // measured and rejected (round 2): a two-deep register ring (loads two stages ahead) -- 128-142 VGPRs besides the 64
    // accumulators drop the F = 2 kernels to 1-2 waves per SIMD and the FCN head's 3x3 layers went 250 -> 562 us; the
    // workgroup count for the k x k form (tools/wgrad_conv_sweep.py): 256 / 512 / 768 / 1024 / 1536 -> 1459 / 1250 /
    // 1322 / 1319 / 1405 us per step
    int64_t s = (tg + tiles - 1) / tiles;                  // two workgroups per CU in all (measured: 512 < 768 < 1024 < 1536
                                                           // in total time over the ResNet-50 shapes, 398 / 423 / 433 / 448 us)
    const int64_t max_s = (M + 2 * WG_KM - 1) / (2 * WG_KM);
    if (s > max_s) s = max_s;
    // partial tiles are written and read once more: keep that within 3x the operand traffic (measured: parallelism
    // matters more than those bytes -- 512->2048 channels at M = 6272 takes 39 us with 12 splits, 115 us with one)
    const int64_t cap = ((int64_t)3 * 2 * M * (CO + CI) * taps) / ((int64_t)8 * CO * CI * taps);
    if (s > cap) s = cap;
    if (s < 1) s = 1;
    if (s < 1) s = 1;
    int rps = (int)((M + s - 1) / s);
    rps = (rps + WG_KM - 1) / WG_KM * WG_KM;
    *rows = rps;
    *S = (M + rps - 1) / rps;
    return CP2_OK;
}

// Splits of the m range = leading dimension of the workspace `part` (float [S][CO][CI]).
CP2_API int cp2_wgrad1x1_num_splits(int M, int CO, int CI) {
    int F, S, rows;
    const int rc = wgrad_geom(M, CO, CI, &F, &S, &rows);
    return rc ? rc : S;
}

CP2_API int cp2_wgrad1x1(const void* dy, const void* x, float* dw, float* part, int M, int CO, int CI, void* stream) {
    if (!dy || !x || !dw || !part) return CP2_ERR_NULL;
    int F, S, rows;
    const int rc = wgrad_geom(M, CO, CI, &F, &S, &rows);
    if (rc) return rc;
    if (!cp2_aligned16(dy) || !cp2_aligned16(x) || !cp2_aligned16(dw) || !cp2_aligned16(part)) return CP2_ERR_ALIGN;
    const unsigned short* d = static_cast<const unsigned short*>(dy);
    const unsigned short* xx = static_cast<const unsigned short*>(x);
    hipStream_t s = cp2_stream(stream);
    const int T = 64 * F, tiles_ci = CI / T;
    const int tiles = (CO / T) * tiles_ci;
    const dim3 grid(tiles * S);
    float* out = S == 1 ? dw : part;
    const WgConv none{};
    if (F == 2) hipLaunchKernelGGL((wgrad1x1_kernel<2, false>), grid, dim3(256), 0, s, d, xx, out, M, CO, CI, rows, tiles_ci, tiles, none);
    else hipLaunchKernelGGL((wgrad1x1_kernel<1, false>), grid, dim3(256), 0, s, d, xx, out, M, CO, CI, rows, tiles_ci, tiles, none);
    int rc2 = cp2_launch_status();
    if (rc2 || S == 1) return rc2;
    const int64_t n4 = (int64_t)CO * CI / 4;
    hipLaunchKernelGGL(wgrad_sum_kernel, dim3((unsigned)((n4 + 15) / 16)), dim3(256), 0, s, part, dw, S, n4);
    return cp2_launch_status();
}

// ---- k x k convolution (3x3 of the ResNet bottlenecks and of the FCN head): dy [N,OH,OW,CO], x [N,H,W,CI] bf16
// channels-last, dw fp32 [CO][KH][KW][CI] (the memory order of a channels-last weight)
static int wgrad_conv_check(int N, int H, int W, int OH, int OW, int KH, int KW, int stride, int pad, int dil) {
    if (N <= 0 || H <= 0 || W <= 0 || OH <= 0 || OW <= 0 || KH <= 0 || KW <= 0 || stride <= 0 || pad < 0 || dil <= 0) return CP2_ERR_SHAPE;
    if (KH * KW > 49) return CP2_ERR_UNSUPPORTED;
    if ((H + 2 * pad - dil * (KH - 1) - 1) / stride + 1 != OH || (W + 2 * pad - dil * (KW - 1) - 1) / stride + 1 != OW) return CP2_ERR_SHAPE;
    if ((int64_t)N * OH * OW > 0x7fffffff) return CP2_ERR_UNSUPPORTED;
    return CP2_OK;
}

CP2_API int cp2_wgrad_conv_num_splits(int N, int OH, int OW, int CO, int CI, int KH, int KW) {
    if (N <= 0 || OH <= 0 || OW <= 0 || KH <= 0 || KW <= 0) return CP2_ERR_SHAPE;
    int F, S, rows;
    const int rc = wgrad_geom(N * OH * OW, CO, CI, &F, &S, &rows, KH * KW);
    return rc ? rc : S;
}

CP2_API int cp2_wgrad_conv(const void* dy, const void* x, float* dw, float* part, int N, int H, int W, int OH, int OW, int CO,
                           int CI, int KH, int KW, int stride, int pad, int dil, void* stream) {
    if (!dy || !x || !dw || !part) return CP2_ERR_NULL;
    int rc = wgrad_conv_check(N, H, W, OH, OW, KH, KW, stride, pad, dil);
    if (rc) return rc;
    const int M = N * OH * OW, taps = KH * KW;
    int F, S, rows;
    rc = wgrad_geom(M, CO, CI, &F, &S, &rows, taps);
    if (rc) return rc;
    if (!cp2_aligned16(dy) || !cp2_aligned16(x) || !cp2_aligned16(dw) || !cp2_aligned16(part)) return CP2_ERR_ALIGN;
    const unsigned short* d = static_cast<const unsigned short*>(dy);
    const unsigned short* xx = static_cast<const unsigned short*>(x);
    hipStream_t s = cp2_stream(stream);
    const int T = 64 * F, tiles_ci = CI / T;
    const int tiles = (CO / T) * taps * tiles_ci;
    const dim3 grid(tiles * S);
    float* out = S == 1 ? dw : part;
    const WgConv cv{H, W, OH, OW, KW, taps, stride, pad, dil};
    if (F == 2) hipLaunchKernelGGL((wgrad1x1_kernel<2, true>), grid, dim3(256), 0, s, d, xx, out, M, CO, CI, rows, tiles_ci, tiles, cv);
    else hipLaunchKernelGGL((wgrad1x1_kernel<1, true>), grid, dim3(256), 0, s, d, xx, out, M, CO, CI, rows, tiles_ci, tiles, cv);
    int rc2 = cp2_launch_status();
    if (rc2 || S == 1) return rc2;
    const int64_t n4 = (int64_t)CO * CI * taps / 4;
    hipLaunchKernelGGL(wgrad_sum_kernel, dim3((unsigned)((n4 + 15) / 16)), dim3(256), 0, s, part, dw, S, n4);
    return cp2_launch_status();
}
