// How fast are LDS atomic adds on gfx950, and what does it depend on?  (quartile kernels.)
// One 1024-thread workgroup per CU; every thread issues 64 x REPS ds_add_u32 to a 4096-bin LDS histogram with bins taken from
// a pattern: 0 = all lanes distinct addresses and banks, 1 = random over 4096 bins, 2 = random over 24 bins (a "spread" row's
// hot first-level bins), 3 = the same 24 bins spread over 4 bank-shifted copies (lane & 3), 4 = one bin, 5 = plain LDS stores
// to distinct addresses (the instruction-issue floor).
//   hipcc -O3 --offload-arch=gfx950 tools/lds_atomic_rate.hip -o /tmp/lds_atomic_rate && /tmp/lds_atomic_rate
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>

constexpr int NT = 1024, PER = 64, REPS = 16, NB = 4096, PAD = 16;

__global__ __launch_bounds__(NT) void k(const unsigned* __restrict__ bins, int mode, unsigned* __restrict__ out) {
    __shared__ unsigned h[4 * (NB + PAD)];
    for (int i = threadIdx.x; i < 4 * (NB + PAD); i += NT) h[i] = 0;
    unsigned b[PER];
#pragma unroll
    for (int i = 0; i < PER; ++i) b[i] = bins[(size_t)i * NT + threadIdx.x];
    __syncthreads();
    for (int r = 0; r < REPS; ++r) {
#pragma unroll
        for (int i = 0; i < PER; ++i) {
            if (mode == 5) h[b[i]] = r;
            else atomicAdd(&h[b[i]], 1u);
        }
    }
    __syncthreads();
    unsigned s = 0;
    for (int i = threadIdx.x; i < 4 * (NB + PAD); i += NT) s += h[i];
    if (s == 0xdeadbeef) out[blockIdx.x] = s;
}

int main() {
    const int modes = 6;
    unsigned* hb = (unsigned*)malloc(sizeof(unsigned) * NT * PER);
    unsigned *db, *dout;
    hipMalloc(&db, sizeof(unsigned) * NT * PER);
    hipMalloc(&dout, 4096);
    const char* names[modes] = {"distinct addresses / banks", "random over 4096 bins", "random over 24 bins", "24 bins x 4 shifted copies", "one bin", "plain stores (issue floor)"};
    for (int m = 0; m < modes; ++m) {
        srand(1);
        for (int i = 0; i < PER; ++i)
            for (int t = 0; t < NT; ++t) {
                unsigned v;
                const int lane = t & 63;
                switch (m) {
                    case 0: case 5: v = (unsigned)((t + 64 * i) % NB); break;
                    case 1: v = (unsigned)(rand() % NB); break;
                    case 2: v = (unsigned)(1000 + rand() % 24); break;
                    case 3: v = (unsigned)(1000 + rand() % 24) + (unsigned)(lane & 3) * (NB + PAD); break;
                    default: v = 1234u;
                }
                hb[(size_t)i * NT + t] = v;
            }
        hipMemcpy(db, hb, sizeof(unsigned) * NT * PER, hipMemcpyHostToDevice);
        hipEvent_t e0, e1;
        hipEventCreate(&e0); hipEventCreate(&e1);
        hipLaunchKernelGGL(k, dim3(256), dim3(NT), 0, 0, db, m, dout);
        hipDeviceSynchronize();
        hipEventRecord(e0);
        for (int it = 0; it < 10; ++it) hipLaunchKernelGGL(k, dim3(256), dim3(NT), 0, 0, db, m, dout);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms;
        hipEventElapsedTime(&ms, e0, e1);
        const double us = ms * 1e3 / 10.0, adds = (double)NT * PER * REPS;
        printf("%-30s %8.1f us per launch   %6.2f adds per ns and CU   (%.1f lanes per clock at 2.4 GHz)\n", names[m], us, adds / (us * 1e3), adds / (us * 1e3) / 2.4);
    }
    return 0;
}
