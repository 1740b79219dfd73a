// A/B of EMA kernel variants in one process (interleaved rounds, median) on 47.4M / 66M floats.
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <cstdio>
#include <vector>
#include <algorithm>
#include <string>
typedef float f4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ float e1(float k, float q, float m, float om) { return __fadd_rn(__fmul_rn(k, m), __fmul_rn(q, om)); }
__device__ __forceinline__ f4 e4(f4 k, f4 q, float m, float om) { f4 r; r.x=e1(k.x,q.x,m,om); r.y=e1(k.y,q.y,m,om); r.z=e1(k.z,q.z,m,om); r.w=e1(k.w,q.w,m,om); return r; }

template <int U, int NT /*0 plain, 1 nt q load, 2 nt q load + nt k store*/>
__global__ void ema(float* __restrict__ k, const float* __restrict__ q, long n4, float m, float om) {
  f4* k4 = (f4*)k; const f4* q4 = (const f4*)q;
  const long span = (long)blockDim.x * U;
  for (long base = (long)blockIdx.x * span; base < n4; base += (long)gridDim.x * span) {
    f4 kv[U], qv[U];
#pragma unroll
    for (int u = 0; u < U; ++u) { long i = base + threadIdx.x + (long)u * blockDim.x; if (i < n4) { kv[u] = (NT==3) ? __builtin_nontemporal_load(k4 + i) : k4[i]; qv[u] = NT ? __builtin_nontemporal_load(q4 + i) : q4[i]; } }
#pragma unroll
    for (int u = 0; u < U; ++u) { long i = base + threadIdx.x + (long)u * blockDim.x; if (i < n4) { f4 r = e4(kv[u], qv[u], m, om); if (NT >= 2) __builtin_nontemporal_store(r, k4 + i); else k4[i] = r; } }
  }
}
struct V { std::string name; void (*fn)(float*, const float*, long, float, float); int threads; int U; int mode; std::vector<float> ms; };
int main() {
  const long n = 47433472;  // FCN config slots (approx)
  float *k, *q; hipMalloc(&k, n * 4); hipMalloc(&q, n * 4); hipMemset(k, 0, n * 4); hipMemset(q, 0, n * 4);
  // a big scratch buffer to flush the 256 MiB infinity cache between launches
  float* flush; const long fn = 128L << 20; hipMalloc(&flush, fn * 4);
  std::vector<V> vs = {
    {"U1 t256 ntq", ema<1,1>, 256, 1, 0}, {"U1 t256 nt all", ema<1,3>, 256, 1, 0}, {"U2 t256 nt all", ema<2,3>, 256, 2, 0},
    {"U2 t512 nt all", ema<2,3>, 512, 2, 0}, {"U1 t512 nt all", ema<1,3>, 512, 1, 0}, {"U1 t128 nt all", ema<1,3>, 128, 1, 0},
    {"U2 t128 nt all", ema<2,3>, 128, 2, 0}, {"U1 t256 nt all cap16384", ema<1,3>, 256, 1, 16384}, {"U2 t256 nt all cap8192", ema<2,3>, 256, 2, 8192},
    {"U1 t1024 nt all", ema<1,3>, 1024, 1, 0},
  };
  hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
  const long n4 = n / 4;
  for (int round = 0; round < 12; ++round) {
    for (auto& v : vs) {
      hipMemsetAsync(flush, round, fn * 4, 0);   // evict k/q from the infinity cache (as a real step does)
      long blocks = (n4 + (long)v.threads * v.U - 1) / ((long)v.threads * v.U);
      if (v.mode && blocks > v.mode) blocks = v.mode;
      hipExtLaunchKernelGGL(v.fn, dim3(blocks), dim3(v.threads), 0, 0, a, b, 0, k, q, n4, 0.999f, 0.001f);
      hipEventSynchronize(b);
      float ms; hipEventElapsedTime(&ms, a, b); if (round >= 2) v.ms.push_back(ms);
    }
  }
  for (auto& v : vs) { std::sort(v.ms.begin(), v.ms.end()); float med = v.ms[v.ms.size() / 2];
    printf("%-26s median %.4f ms  min %.4f  -> %.0f GB/s (median)\n", v.name.c_str(), med, v.ms[0], 12.0 * n / med / 1e6); }
  return 0;
}
