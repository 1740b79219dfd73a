// a13: queue enqueue with wrap-around -- reference builder.py:569-587.
// keys [n,C] (row = key) are transposed into columns (ptr+i) % K of queue [C,K].
// 32x32 tiles go through LDS so both the read (along C) and the write (along K)
// are coalesced.  The pointer lives on the device (the reference does int(queue_ptr), a host
// synchronisation): every workgroup reads it before it stores anything, and the workgroup that
// draws the last ticket of a caller-owned counter advances it, so the whole enqueue is ONE launch.
#include "common.hpp"

__global__ __launch_bounds__(256) void enqueue_scatter_kernel(float* __restrict__ queue,
                                                              const float* __restrict__ keys,
                                                              int64_t* __restrict__ ptr, int32_t* __restrict__ ticket,
                                                              int n, int C, int K) {
    __shared__ float tile[32][33];
    const int i0 = blockIdx.x * 32;  // key tile
    const int c0 = blockIdx.y * 32;  // channel tile
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;  // 32 x 8
    const int64_t p = *ptr;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int i = i0 + ty + 8 * r, c = c0 + tx;
        tile[ty + 8 * r][tx] = (i < n && c < C) ? keys[(int64_t)i * C + c] : 0.0f;
    }
    __syncthreads();
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int c = c0 + ty + 8 * r, i = i0 + tx;
        if (i < n && c < C) queue[(int64_t)c * K + (p + i) % K] = tile[tx][ty + 8 * r];
    }
    // every thread of this workgroup holds p in a register by now (its stores were addressed with it)
    __syncthreads();
    if (threadIdx.x == 0) {
        const int total = gridDim.x * gridDim.y;
        if (atomicAdd(ticket, 1) == total - 1) {     // all other workgroups have read the pointer: advance it
            *ptr = (p + n) % K;
            *ticket = 0;                             // ready for the next launch (stream order makes it visible)
        }
    }
}

CP2_API int cp2_enqueue(float* queue, const float* keys, int64_t* ptr, int32_t* ticket, int n, int C, int K, void* stream) {
    if (!queue || !keys || !ptr || !ticket) return CP2_ERR_NULL;
    if (n <= 0 || C <= 0 || K <= 0) return CP2_ERR_SHAPE;
    if (n > K) return CP2_ERR_UNSUPPORTED;
    hipLaunchKernelGGL(enqueue_scatter_kernel, dim3(cp2_cdiv(n, 32), cp2_cdiv(C, 32)), dim3(256), 0,
                       cp2_stream(stream), queue, keys, ptr, ticket, n, C, K);
    return cp2_launch_status();
}
