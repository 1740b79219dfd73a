// Standalone timing / stamp harness of the a10 instance kernel (not part of the product):
//   hipcc -O3 -std=c++17 --offload-arch=gfx950 -ffp-contract=off -DCP2_STAMPS -I cp2_amd/csrc tools/rowkey_small_bench.hip -o /tmp/rsb && /tmp/rsb
// Launches the kernel back to back (no host pacing, so the clock is what a busy training step sees), reports the
// average kernel time from HIP events and, with -DCP2_STAMPS, where a wave spends its cycles.
#include "../cp2_amd/csrc/rowkey_small.hip"
#include <stdio.h>
#include <vector>
#include <algorithm>
#include <random>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s failed: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

int main(int argc, char** argv) {
    const int R = argc > 1 ? atoi(argv[1]) : 32, K = argc > 2 ? atoi(argv[2]) : 65536, iters = argc > 3 ? atoi(argv[3]) : 200;
    const bool with_u = argc > 4 ? atoi(argv[4]) != 0 : true;
    std::mt19937 rng(1);
    std::normal_distribution<float> nd(0.f, 1.f);
    std::vector<float> hq((size_t)R * CH), hk((size_t)CH * K), he(R);
    for (auto& v : hq) v = nd(rng) * 0.088f;
    for (auto& v : hk) v = nd(rng) * 0.088f;
    for (auto& v : he) v = 0.3f;
    float *q, *keys, *ext, *pm, *ps, *pu; int* pc;
    int tpw;
    const int S = rowkey_small_num_splits(K, &tpw);
    CK(hipMalloc(&q, hq.size() * 4)); CK(hipMalloc(&keys, hk.size() * 4)); CK(hipMalloc(&ext, R * 4));
    CK(hipMalloc(&pm, (size_t)S * R * 4)); CK(hipMalloc(&ps, (size_t)S * R * 4)); CK(hipMalloc(&pc, (size_t)S * R * 4));
    CK(hipMalloc(&pu, (size_t)S * CH * R * 4));
    CK(hipMemcpy(q, hq.data(), hq.size() * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(keys, hk.data(), hk.size() * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(ext, he.data(), R * 4, hipMemcpyHostToDevice));
    RowKeyArgs a{q, 1, CH, 0, 1, R, keys, K, ext, 1, 5.0f, 0, pm, ps, pc, with_u ? pu : nullptr, nullptr};
    hipStream_t st; CK(hipStreamCreate(&st));
    for (int i = 0; i < 20; ++i) if (rowkey_small_launch(a, S, with_u, st)) return 2;
    CK(hipStreamSynchronize(st));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    CK(hipEventRecord(e0, st));
    for (int i = 0; i < iters; ++i) rowkey_small_launch(a, S, with_u, st);
    CK(hipEventRecord(e1, st));
    CK(hipStreamSynchronize(st));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    printf("R=%d K=%d S=%d tiles/wg=%d with_u=%d: %.2f us per launch back to back = %.0f GB/s of queue\n", R, K, S, tpw, (int)with_u,
           ms * 1e3 / iters, (double)CH * K * 4 / (ms * 1e-3 / iters) / 1e9);
#ifdef CP2_STAMPS
    std::vector<unsigned long long> hs(256 * 8 * 16);
    CK(hipMemcpyFromSymbol(hs.data(), HIP_SYMBOL(cp2_stamps), hs.size() * 8));
    // per wave slot: medians over workgroups of (stamp i - stamp 0) in shader cycles
    const char* names[11] = {"start", "row image + positives landed", "first key pieces landed", "product 1 done", "product 2 done (last tile)", "before merge barrier", "end", "(realtime)", "p split done, product 2 starts", "product 2 half done", "scaled logits, max, rescale done"};
    for (int w = 0; w < 8; ++w) {
        printf("wave %d:", w);
        for (int i : {1, 2, 3, 10, 8, 9, 4, 5, 6}) {
            std::vector<long long> d;
            for (int b = 0; b < S && b < 256; ++b) d.push_back((long long)(hs[(b * 8 + w) * 16 + i] - hs[(b * 8 + w) * 16 + 0]));
            std::sort(d.begin(), d.end());
            printf("  [%d]%lld", i, d[d.size() / 2]);
        }
        printf("\n");
    }
    for (int i : {1, 2, 3, 10, 8, 9, 4, 5, 6}) printf("  [%d] = %s\n", i, names[i]);
    // clock: span of stamp 0 (memtime) vs stamp 7 (memrealtime, 100 MHz) between two workgroups is not comparable; report cycles only
#endif
    return 0;
}
