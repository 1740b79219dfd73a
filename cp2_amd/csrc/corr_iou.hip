// a3-a5: IoU of two id maps -- reference tools/correlation_mapping.py:103-138.
// The reference runs torch.unique (a sort) per sample inside a Python loop, four
// times per step, each with a host sync.  Here one workgroup per (sample, masked?)
// sorts the 2P+1 float keys in LDS (bitonic network), then counts runs:
//   union = #runs - 1, intersection = #runs of a non-zero key with length >= 2.
// Keys are float32(id+1)*mask exactly as the reference forms them (ids pass
// through float32 there because torch.cat promotes int64 with a float zero).
#include "common.hpp"
#include <math.h>

__global__ __launch_bounds__(1024) void corr_iou_kernel(const int64_t* __restrict__ ids_a,
                                                        const int64_t* __restrict__ ids_b,
                                                        const float* __restrict__ mask_a,
                                                        const float* __restrict__ mask_b,
                                                        float* __restrict__ iou, float* __restrict__ iou_masked,
                                                        int P, int N2, int H, int W, int stride, int Ws) {
    extern __shared__ __attribute__((aligned(16))) float keys[];
    __shared__ int red[2][16];
    const int n = blockIdx.x;
    const bool masked = blockIdx.y == 1;
    float* out = masked ? iou_masked : iou;
    if (!out) return;
    const int nvalid = 2 * P + 1;
    // stride > 0: the id maps are the full-resolution [H, W] ones and element p of the P = Hs x Ws grid is their centre
    // tap (s/2 + s * (p / Ws), s/2 + s * (p % Ws)) -- reference builder.py:1155-1186 slices first, then compares
    auto at = [&](int p) -> int64_t {
        if (stride <= 0) return (int64_t)n * P + p;
        const int off = stride >> 1;
        return ((int64_t)n * H + off + (int64_t)stride * (p / Ws)) * W + off + (int64_t)stride * (p % Ws);
    };
    for (int i = threadIdx.x; i < N2; i += blockDim.x) {
        float v = INFINITY;  // padding sorts behind every real key
        if (i == 0) {
            v = 0.0f;
        } else if (i <= P) {
            const float idf = (float)(ids_a[at(i - 1)] + 1);
            v = masked ? __fmul_rn(idf, mask_a[(int64_t)n * P + (i - 1)]) : idf;
        } else if (i < nvalid) {
            const float idf = (float)(ids_b[at(i - 1 - P)] + 1);
            v = masked ? __fmul_rn(idf, mask_b[(int64_t)n * P + (i - 1 - P)]) : idf;
        }
        keys[i] = v;
    }
    __syncthreads();
    for (int k = 2; k <= N2; k <<= 1) {
        for (int j = k >> 1; j > 0; j >>= 1) {
            for (int i = threadIdx.x; i < N2; i += blockDim.x) {
                const int p = i ^ j;
                if (p > i) {
                    const float a = keys[i], b = keys[p];
                    const bool up = (i & k) == 0;
                    if ((a > b) == up) { keys[i] = b; keys[p] = a; }
                }
            }
            __syncthreads();
        }
    }
    int uniq = 0, inter = 0;
    for (int i = threadIdx.x; i < nvalid; i += blockDim.x) {
        const float v = keys[i];
        const bool start = (i == 0) || (v != keys[i - 1]);
        if (start) {
            ++uniq;
            if (v != 0.0f && i + 1 < nvalid && keys[i + 1] == v) ++inter;
        }
    }
    uniq = wave_sum_i(uniq);
    inter = wave_sum_i(inter);
    const int w = threadIdx.x >> 6;
    if ((threadIdx.x & 63) == 0) { red[0][w] = uniq; red[1][w] = inter; }
    __syncthreads();
    if (threadIdx.x == 0) {
        int u = 0, s = 0;
        for (int i = 0; i < (int)(blockDim.x >> 6); ++i) { u += red[0][i]; s += red[1][i]; }
        const int uni = u - 1;  // minus the zero key
        out[n] = uni > 0 ? (float)((double)s / (double)uni) : NAN;
    }
}

// Maps of up to 4096 cells (every training shape, config 4's included) are counted in an LDS hash table instead: tail.hip.
int cp2_tail_iou_table(int P);
int cp2_tail_iou_launch(const int64_t* ids_a, const int64_t* ids_b, const float* mask_a, const float* mask_b, float* iou,
                        float* iou_masked, int B, int P, int H, int W, int stride, int Ws, void* stream);

static int corr_iou_launch(const int64_t* ids_a, const int64_t* ids_b, const float* mask_a, const float* mask_b, float* iou,
                           float* iou_masked, int B, int P, int H, int W, int stride, int Ws, void* stream) {
    if (!ids_a || !ids_b) return CP2_ERR_NULL;
    if (!iou && !iou_masked) return CP2_ERR_NULL;
    if (iou_masked && (!mask_a || !mask_b)) return CP2_ERR_NULL;
    if (B <= 0 || P <= 0) return CP2_ERR_SHAPE;
    if (P > 16383) return CP2_ERR_UNSUPPORTED;
    if (cp2_tail_iou_table(P) > 0)                           // hash-count form (two barriers instead of 45 sort stages)
        return cp2_tail_iou_launch(ids_a, ids_b, mask_a, mask_b, iou, iou_masked, B, P, H, W, stride, Ws, stream);
    int N2 = 64;
    while (N2 < 2 * P + 1) N2 <<= 1;
    const size_t lds = (size_t)N2 * sizeof(float);
    if (lds > 48 * 1024) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(corr_iou_kernel),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return (int)e;
    }
    const int threads = N2 >= 2048 ? 1024 : (N2 / 2 < 64 ? 64 : N2 / 2);
    hipLaunchKernelGGL(corr_iou_kernel, dim3(B, 2), dim3(threads), lds, cp2_stream(stream), ids_a, ids_b, mask_a,
                       mask_b, iou, iou_masked, P, N2, H, W, stride, Ws);
    return cp2_launch_status();
}

CP2_API int cp2_corr_iou(const int64_t* ids_a, const int64_t* ids_b, const float* mask_a, const float* mask_b,
                         float* iou, float* iou_masked, int B, int P, void* stream) {
    return corr_iou_launch(ids_a, ids_b, mask_a, mask_b, iou, iou_masked, B, P, 0, 0, 0, 1, stream);
}

// The same with the centre-tap down-sampling of the id maps folded in: ids are [B,H,W], the masks already [B,Hs*Ws].
CP2_API int cp2_corr_iou_strided(const int64_t* ids_a, const int64_t* ids_b, const float* mask_a, const float* mask_b,
                                 float* iou, float* iou_masked, int B, int H, int W, int stride, void* stream) {
    if (H <= 0 || W <= 0 || stride <= 0) return CP2_ERR_SHAPE;
    const int off = stride / 2, Hs = (H - off + stride - 1) / stride, Ws = (W - off + stride - 1) / stride;
    if (Hs <= 0 || Ws <= 0) return CP2_ERR_SHAPE;
    return corr_iou_launch(ids_a, ids_b, mask_a, mask_b, iou, iou_masked, B, Hs * Ws, H, W, stride, Ws, stream);
}

