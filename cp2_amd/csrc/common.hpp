// Shared helpers for the libcp2hip.so kernels (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <stdint.h>
#include "../../include/cp2hip.h"

#define CP2_API extern "C" __attribute__((visibility("default")))

static inline hipStream_t cp2_stream(void* s) { return reinterpret_cast<hipStream_t>(s); }

// Return code after a kernel launch: 0 or the (positive) hipError_t.
static inline int cp2_launch_status() {
    hipError_t e = hipGetLastError();
    return e == hipSuccess ? CP2_OK : static_cast<int>(e);
}

static inline bool cp2_aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15u) == 0; }
static inline int cp2_cdiv(int64_t a, int64_t b) { return static_cast<int>((a + b - 1) / b); }

// Measurement aid (bench.py's per-kernel roofline figures): cp2_profile_next_launch(start, stop) arms a pair of
// caller-owned hipEvent_t for the calling thread; the next launch made through CP2_LAUNCH_PROFILED attaches them to
// the kernel itself (hipExtLaunchKernelGGL), so hipEventElapsedTime is that kernel's own duration, as
// rocprofv3 --kernel-trace reports it.  Nothing is armed in normal operation and the plain launch runs.
struct Cp2LaunchEvents { hipEvent_t start = nullptr; hipEvent_t stop = nullptr; };
inline thread_local Cp2LaunchEvents cp2_next_events;
#define CP2_LAUNCH_PROFILED(kernel, grid, block, lds, stream, ...)                                                    \
    do {                                                                                                             \
        if (cp2_next_events.start) {                                                                                 \
            hipExtLaunchKernelGGL(kernel, grid, block, lds, stream, cp2_next_events.start, cp2_next_events.stop, 0,   \
                                  __VA_ARGS__);                                                                      \
            cp2_next_events = Cp2LaunchEvents{};                                                                     \
        } else {                                                                                                     \
            hipLaunchKernelGGL(kernel, grid, block, lds, stream, __VA_ARGS__);                                       \
        }                                                                                                            \
    } while (0)

constexpr int kWave = 64;  // gfx950 wavefront width

// Wave-level sum over all 64 lanes (result valid in every lane).
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
    return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v = fmaxf(v, __shfl_xor(v, off, 64));
    return v;
}
__device__ __forceinline__ int wave_sum_i(int v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
    return v;
}
