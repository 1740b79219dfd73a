// a11: momentum (EMA) update of the key encoder -- reference builder.py:557-567.
//   k = k*m + q*(1-m), as two rounded products and one rounded sum (torch evaluates
//   `pk * m + pq * (1.0 - m)` as three separate fp32 kernels), so the result is
//   bit-identical to the reference.  HBM-bound: 12 algorithmic bytes per element
//   (read k, read q, write k).  One launch for the whole encoder.
#include "common.hpp"
#include <hip/hip_ext.h>

__device__ __forceinline__ float ema1(float k, float q, float m, float om) {
    return __fadd_rn(__fmul_rn(k, m), __fmul_rn(q, om));
}
__device__ __forceinline__ float4 ema4(float4 k, float4 q, float m, float om) {
    return make_float4(ema1(k.x, q.x, m, om), ema1(k.y, q.y, m, om), ema1(k.z, q.z, m, om),
                       ema1(k.w, q.w, m, om));
}

// Flat span, tuned on MI355X (tools/ema_tune.hip, kernel-exact events, infinity cache flushed
// between launches): 128-thread workgroups, ONE 16-byte element per lane, exact grid, and
// non-temporal loads/stores (every byte is touched once per step, so nothing should linger in
// L2 / MALL) reached 6.4-6.5 TB/s = 81 % of the 8 TB/s HBM3E peak; 256 threads x 4 elements
// with plain loads (the first version) 4.5 TB/s.
typedef float f32x4_t __attribute__((ext_vector_type(4)));
constexpr int kEmaThreads = 128;

// SHADOW: also store bf16(k_new) (round-to-nearest-even, what autocast's cast would produce) into a second flat
// buffer with the same element order, so the key encoder's convolutions can take bf16 weights directly instead of
// launching one cast kernel per weight tensor every step (+2 bytes written per parameter: 14 B instead of 12 B).
typedef unsigned short u16x4_t __attribute__((ext_vector_type(4)));
__device__ __forceinline__ unsigned short ema_f2bf(float f) {
    unsigned u = __float_as_uint(f);
    if ((u & 0x7fffffffu) > 0x7f800000u) return (unsigned short)((u >> 16) | 0x40u);
    u += 0x7fffu + ((u >> 16) & 1u);
    return (unsigned short)(u >> 16);
}

template <bool SHADOW>
__global__ __launch_bounds__(kEmaThreads) void ema_flat_kernel(float* __restrict__ k, const float* __restrict__ q,
                                                               unsigned short* __restrict__ kb, int64_t n4, int64_t n,
                                                               float m, float om) {
    const int64_t i = (int64_t)blockIdx.x * kEmaThreads + threadIdx.x;
    if (i < n4) {
        f32x4_t* k4 = reinterpret_cast<f32x4_t*>(k);
        const f32x4_t* q4 = reinterpret_cast<const f32x4_t*>(q);
        const f32x4_t kv = __builtin_nontemporal_load(k4 + i);
        const f32x4_t qv = __builtin_nontemporal_load(q4 + i);
        f32x4_t r;
        r.x = ema1(kv.x, qv.x, m, om); r.y = ema1(kv.y, qv.y, m, om);
        r.z = ema1(kv.z, qv.z, m, om); r.w = ema1(kv.w, qv.w, m, om);
        __builtin_nontemporal_store(r, k4 + i);
        if (SHADOW) {
            u16x4_t b;
            b.x = ema_f2bf(r.x); b.y = ema_f2bf(r.y); b.z = ema_f2bf(r.z); b.w = ema_f2bf(r.w);
            reinterpret_cast<u16x4_t*>(kb)[i] = b;     // read again soon by the key encoder: default cache policy
        }
    }
    if (blockIdx.x == 0) {  // tail (n not a multiple of 4)
        for (int64_t j = n4 * 4 + threadIdx.x; j < n; j += kEmaThreads) {
            const float r = ema1(k[j], q[j], m, om);
            k[j] = r;
            if (SHADOW) kb[j] = ema_f2bf(r);
        }
    }
}

static int ema_flat_launch(float* k, const float* q, void* k_bf16, int64_t n, float m, float one_minus_m,
                           hipEvent_t start, hipEvent_t stop, void* stream) {
    if (!k || !q) return CP2_ERR_NULL;
    if (n <= 0) return CP2_ERR_SHAPE;
    if (!cp2_aligned16(k) || !cp2_aligned16(q) || (k_bf16 && (reinterpret_cast<uintptr_t>(k_bf16) & 7u))) return CP2_ERR_ALIGN;
    const int64_t n4 = n / 4;
    const int64_t blocks = n4 > 0 ? (n4 + kEmaThreads - 1) / kEmaThreads : 1;
    if (blocks > 0x7fffffffLL) return CP2_ERR_UNSUPPORTED;
    unsigned short* kb = static_cast<unsigned short*>(k_bf16);
    if (kb)
        hipExtLaunchKernelGGL(ema_flat_kernel<true>, dim3((unsigned)blocks), dim3(kEmaThreads), 0, cp2_stream(stream), start,
                              stop, 0, k, q, kb, n4, n, m, one_minus_m);
    else
        hipExtLaunchKernelGGL(ema_flat_kernel<false>, dim3((unsigned)blocks), dim3(kEmaThreads), 0, cp2_stream(stream), start,
                              stop, 0, k, q, kb, n4, n, m, one_minus_m);
    return cp2_launch_status();
}

CP2_API int cp2_ema_flat(float* k, const float* q, int64_t n, float m, float one_minus_m, void* stream) {
    return ema_flat_launch(k, q, nullptr, n, m, one_minus_m, nullptr, nullptr, stream);
}

// Same launch with a pair of caller-owned hipEvent_t that bracket exactly this kernel
// (hipExtLaunchKernelGGL start/stop events): used by bench.py for the roofline figure.
CP2_API int cp2_ema_flat_timed(float* k, const float* q, int64_t n, float m, float one_minus_m, void* start_event,
                               void* stop_event, void* stream) {
    if (!start_event || !stop_event) return CP2_ERR_NULL;
    return ema_flat_launch(k, q, nullptr, n, m, one_minus_m, reinterpret_cast<hipEvent_t>(start_event),
                           reinterpret_cast<hipEvent_t>(stop_event), stream);
}

// EMA + bf16 shadow copy of the updated key weights (k_bf16: n bf16 values, 8-byte aligned); events may be NULL.
CP2_API int cp2_ema_flat_shadow(float* k, const float* q, void* k_bf16, int64_t n, float m, float one_minus_m,
                                void* start_event, void* stop_event, void* stream) {
    if (!k_bf16) return CP2_ERR_NULL;
    return ema_flat_launch(k, q, k_bf16, n, m, one_minus_m, reinterpret_cast<hipEvent_t>(start_event),
                           reinterpret_cast<hipEvent_t>(stop_event), stream);
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
