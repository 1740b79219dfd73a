// a11: momentum (EMA) update of the key encoder -- reference builder.py:557-567.
//   k = k*m + q*(1-m), as two rounded products and one rounded sum (torch evaluates
//   `pk * m + pq * (1.0 - m)` as three separate fp32 kernels), so the result is
//   bit-identical to the reference.  HBM-bound: 12 algorithmic bytes per element
//   (read k, read q, write k).  One launch for the whole encoder.
#include "common.hpp"

__device__ __forceinline__ float ema1(float k, float q, float m, float om) {
    return __fadd_rn(__fmul_rn(k, m), __fmul_rn(q, om));
}
__device__ __forceinline__ float4 ema4(float4 k, float4 q, float m, float om) {
    return make_float4(ema1(k.x, q.x, m, om), ema1(k.y, q.y, m, om), ema1(k.z, q.z, m, om),
                       ema1(k.w, q.w, m, om));
}

// Flat span: each block owns UNROLL*256 consecutive float4; all loads of an
// iteration are issued before the first store so 2*UNROLL 16-byte loads per lane
// are in flight.
template <int UNROLL>
__global__ __launch_bounds__(256) void ema_flat_kernel(float* __restrict__ k, const float* __restrict__ q,
                                                       int64_t n4, int64_t n, float m, float om) {
    float4* k4 = reinterpret_cast<float4*>(k);
    const float4* q4 = reinterpret_cast<const float4*>(q);
    const int64_t span = (int64_t)blockDim.x * UNROLL;
    for (int64_t base = (int64_t)blockIdx.x * span; base < n4; base += (int64_t)gridDim.x * span) {
        float4 kv[UNROLL], qv[UNROLL];
#pragma unroll
        for (int u = 0; u < UNROLL; ++u) {
            const int64_t i = base + threadIdx.x + (int64_t)u * blockDim.x;
            if (i < n4) {
                kv[u] = k4[i];
                qv[u] = q4[i];
            }
        }
#pragma unroll
        for (int u = 0; u < UNROLL; ++u) {
            const int64_t i = base + threadIdx.x + (int64_t)u * blockDim.x;
            if (i < n4) k4[i] = ema4(kv[u], qv[u], m, om);
        }
    }
    // tail (n not a multiple of 4)
    if (blockIdx.x == 0) {
        for (int64_t i = n4 * 4 + threadIdx.x; i < n; i += blockDim.x) k[i] = ema1(k[i], q[i], m, om);
    }
}

CP2_API int cp2_ema_flat(float* k, const float* q, int64_t n, float m, float one_minus_m, void* stream) {
    if (!k || !q) return CP2_ERR_NULL;
    if (n <= 0) return CP2_ERR_SHAPE;
    if (!cp2_aligned16(k) || !cp2_aligned16(q)) return CP2_ERR_ALIGN;
    constexpr int UNROLL = 4;
    const int64_t n4 = n / 4;
    int blocks = cp2_cdiv(n4 > 0 ? n4 : 1, 256 * UNROLL);
    if (blocks > 256 * 16) blocks = 256 * 16;  // 16 workgroups per CU, grid-stride beyond
    hipLaunchKernelGGL(ema_flat_kernel<UNROLL>, dim3(blocks), dim3(256), 0, cp2_stream(stream), k, q, n4, n, m,
                       one_minus_m);
    return cp2_launch_status();
}

// Multi-tensor form: block c handles chunk c = (tensor, offset, length).
__global__ __launch_bounds__(256) void ema_multi_kernel(float* const* __restrict__ k_ptrs,
                                                        const float* const* __restrict__ q_ptrs,
                                                        const int32_t* __restrict__ chunk_tensor,
                                                        const int64_t* __restrict__ chunk_off,
                                                        const int32_t* __restrict__ chunk_len, float m,
                                                        float om) {
    const int c = blockIdx.x;
    const int t = chunk_tensor[c];
    const int64_t off = chunk_off[c];
    const int len = chunk_len[c];
    float* k = k_ptrs[t] + off;
    const float* q = q_ptrs[t] + off;
    const bool vec = ((reinterpret_cast<uintptr_t>(k) | reinterpret_cast<uintptr_t>(q)) & 15u) == 0;
    if (vec) {
        const int n4 = len >> 2;
        float4* k4 = reinterpret_cast<float4*>(k);
        const float4* q4 = reinterpret_cast<const float4*>(q);
        for (int i = threadIdx.x; i < n4; i += blockDim.x) k4[i] = ema4(k4[i], q4[i], m, om);
        for (int i = (n4 << 2) + threadIdx.x; i < len; i += blockDim.x) k[i] = ema1(k[i], q[i], m, om);
    } else {
        for (int i = threadIdx.x; i < len; i += blockDim.x) k[i] = ema1(k[i], q[i], m, om);
    }
}

CP2_API int cp2_ema_multi(float* const* k_ptrs, const float* const* q_ptrs, const int32_t* chunk_tensor,
                          const int64_t* chunk_off, const int32_t* chunk_len, int n_chunks, float m,
                          float one_minus_m, void* stream) {
    if (!k_ptrs || !q_ptrs || !chunk_tensor || !chunk_off || !chunk_len) return CP2_ERR_NULL;
    if (n_chunks <= 0) return CP2_ERR_SHAPE;
    hipLaunchKernelGGL(ema_multi_kernel, dim3(n_chunks), dim3(256), 0, cp2_stream(stream), k_ptrs, q_ptrs,
                       chunk_tensor, chunk_off, chunk_len, m, one_minus_m);
    return cp2_launch_status();
}
