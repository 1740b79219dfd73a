// Encoder fast path (not a reference call site; off the SURVEY 8 table): the ResNet stem's MaxPool2d(3, stride 2, padding 1)
// (mmseg_/models/backbones/resnet.py:413, :632-636) on channels-last bf16 activations.  ATen's NHWC kernels take 35 us
// (forward) and 85 us (backward) for the 32 x 64 x 112 x 112 stem output, where the data moved is 64 MB / 77 MB.
// Forward: one thread = one output pixel x 8 channels (16-byte lanes), nine 16-byte loads, the position of the maximum
// (0..8, first maximum in row-major window order, NaN wins -- ATen's rule) kept as one byte per element.
// Backward: one thread = one INPUT pixel x 8 channels; a pixel lies in at most four windows and takes dy of every
// window whose recorded position is this pixel (gather form: no atomics, deterministic, every dx element written once).
#include "common.hpp"

typedef unsigned short u16x8 __attribute__((ext_vector_type(8)));

__device__ __forceinline__ float bf2f(unsigned short v) { return __uint_as_float((unsigned)v << 16); }

struct PoolGeom { int N, H, W, C8, OH, OW; };

__global__ __launch_bounds__(256) void maxpool3s2_fwd_kernel(const u16x8* __restrict__ x, u16x8* __restrict__ y,
                                                             unsigned long long* __restrict__ idx, PoolGeom g) {
    const int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const int64_t total = (int64_t)g.N * g.OH * g.OW * g.C8;
    if (t >= total) return;
    const int c8 = (int)(t % g.C8);
    int64_t p = t / g.C8;
    const int ow = (int)(p % g.OW); p /= g.OW;
    const int oh = (int)(p % g.OH);
    const int n = (int)(p / g.OH);
    float best[8];
    unsigned short bv[8];
    unsigned char bi[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) { best[e] = -INFINITY; bv[e] = 0xFF80; bi[e] = 0; }   // 0xFF80 = -inf in bf16
    bool first = true;
#pragma unroll
    for (int kh = 0; kh < 3; ++kh) {
        const int h = 2 * oh - 1 + kh;
#pragma unroll
        for (int kw = 0; kw < 3; ++kw) {
            const int w = 2 * ow - 1 + kw;
            if (h < 0 || h >= g.H || w < 0 || w >= g.W) continue;
            const u16x8 v = x[(((int64_t)n * g.H + h) * g.W + w) * g.C8 + c8];
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const float f = bf2f(v[e]);
                if (first || f > best[e] || f != f) { best[e] = f; bv[e] = v[e]; bi[e] = (unsigned char)(kh * 3 + kw); }
            }
            first = false;
        }
    }
    u16x8 o;
    unsigned long long pk = 0;
#pragma unroll
    for (int e = 0; e < 8; ++e) { o[e] = bv[e]; pk |= (unsigned long long)bi[e] << (8 * e); }
    y[t] = o;
    idx[t] = pk;
}

__global__ __launch_bounds__(256) void maxpool3s2_bwd_kernel(const u16x8* __restrict__ dy, const unsigned long long* __restrict__ idx,
                                                             u16x8* __restrict__ dx, PoolGeom g) {
    const int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const int64_t total = (int64_t)g.N * g.H * g.W * g.C8;
    if (t >= total) return;
    const int c8 = (int)(t % g.C8);
    int64_t p = t / g.C8;
    const int w = (int)(p % g.W); p /= g.W;
    const int h = (int)(p % g.H);
    const int n = (int)(p / g.H);
    float acc[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) acc[e] = 0.f;
    // windows (oh, ow) that contain (h, w): 2 oh - 1 <= h <= 2 oh + 1
    const int oh0 = h >> 1, ow0 = w >> 1;                     // kh = h - 2 oh + 1 in {1, 2} for oh0; oh0 + 1 gives kh = 0 when h is odd
#pragma unroll
    for (int a = 0; a < 2; ++a) {
        const int oh = oh0 + a, kh = h - 2 * oh + 1;
        if (kh < 0 || kh > 2 || oh >= g.OH) continue;
#pragma unroll
        for (int b = 0; b < 2; ++b) {
            const int ow = ow0 + b, kw = w - 2 * ow + 1;
            if (kw < 0 || kw > 2 || ow >= g.OW) continue;
            const int64_t o = (((int64_t)n * g.OH + oh) * g.OW + ow) * g.C8 + c8;
            const unsigned long long pk = idx[o];
            const u16x8 d = dy[o];
            const unsigned me = (unsigned)(kh * 3 + kw);
#pragma unroll
            for (int e = 0; e < 8; ++e)
                if (((unsigned)(pk >> (8 * e)) & 0xFFu) == me) acc[e] += bf2f(d[e]);
        }
    }
    u16x8 o;
#pragma unroll
    for (int e = 0; e < 8; ++e) {                             // fp32 -> bf16, round to nearest even (sums of at most 4 bf16 values)
        const unsigned u = __float_as_uint(acc[e]);
        o[e] = (unsigned short)((u + 0x7FFFu + ((u >> 16) & 1u)) >> 16);
    }
    dx[t] = o;
}

static int pool_geom(int N, int H, int W, int C, PoolGeom* g) {
    if (N <= 0 || H <= 0 || W <= 0 || C <= 0) return CP2_ERR_SHAPE;
    if (C % 8 != 0) return CP2_ERR_UNSUPPORTED;
    *g = PoolGeom{N, H, W, C / 8, (H + 2 - 3) / 2 + 1, (W + 2 - 3) / 2 + 1};
    return CP2_OK;
}

CP2_API int cp2_maxpool3s2_fwd(const void* x, void* y, void* idx, int N, int H, int W, int C, void* stream) {
    if (!x || !y || !idx) return CP2_ERR_NULL;
    PoolGeom g;
    int rc = pool_geom(N, H, W, C, &g);
    if (rc) return rc;
    if (!cp2_aligned16(x) || !cp2_aligned16(y) || (reinterpret_cast<uintptr_t>(idx) & 7u)) return CP2_ERR_ALIGN;
    const int64_t total = (int64_t)g.N * g.OH * g.OW * g.C8;
    if (total > (int64_t)0x7fffffff * 256) return CP2_ERR_UNSUPPORTED;
    hipLaunchKernelGGL(maxpool3s2_fwd_kernel, dim3(cp2_cdiv(total, 256)), dim3(256), 0, cp2_stream(stream),
                       static_cast<const u16x8*>(x), static_cast<u16x8*>(y), static_cast<unsigned long long*>(idx), g);
    return cp2_launch_status();
}

CP2_API int cp2_maxpool3s2_bwd(const void* dy, const void* idx, void* dx, int N, int H, int W, int C, void* stream) {
    if (!dy || !idx || !dx) return CP2_ERR_NULL;
    PoolGeom g;
    int rc = pool_geom(N, H, W, C, &g);
    if (rc) return rc;
    if (!cp2_aligned16(dy) || !cp2_aligned16(dx) || (reinterpret_cast<uintptr_t>(idx) & 7u)) return CP2_ERR_ALIGN;
    const int64_t total = (int64_t)g.N * g.H * g.W * g.C8;
    if (total > (int64_t)0x7fffffff * 256) return CP2_ERR_UNSUPPORTED;
    hipLaunchKernelGGL(maxpool3s2_bwd_kernel, dim3(cp2_cdiv(total, 256)), dim3(256), 0, cp2_stream(stream),
                       static_cast<const u16x8*>(dy), static_cast<const unsigned long long*>(idx), static_cast<u16x8*>(dx), g);
    return cp2_launch_status();
}
