// a10: instance InfoNCE of at most 32 pooled query vectors against the whole queue -- reference
// builder.py:1395-1397 (l_neg = q_pos @ queue), :1420-1428 (cross entropy over [l_pos | l_neg]).
//
// Bound: the fp32 queue [C=128][K] is streamed from HBM exactly once (4*C*K bytes: 33.5 MB at K = 65536); the two
// products (logits, and softmax-weighted key sum for the gradient) are 4*R*C*K flop of exact-fp32 MFMA, which at
// R = 32 is ~7 us of matrix-pipe time per CU -- the same order as the stream, so the two must overlap.
//
// Structure: one workgroup of 8 waves per CU, NO barrier in the main loop.  A wave owns 32-key sub-tiles: it pulls
// its own 16 KB [128 channels][32 keys] straight into its private LDS region with 16 LDS-DMA instructions
// (global_load_lds_dwordx4, no staging registers), waits on its own vmcnt only, and runs product 1 -> soft-max ->
// product 2 on it.  The two waves of a SIMD drift apart, so one wave's DMA wait sits under the other's MFMA chain,
// and 8 x 16 KB = 128 KB of loads are in flight per CU when the kernel starts.
//
// LDS image of a sub-tile (DMA writes lane-linear: destination = base + 16 * (64 i + lane), so the layout is chosen
// through each lane's SOURCE address): 16-byte chunk (channel c, key quad kq) lives in slot
//     slot(c, kq) = (c >> 1) * 16 + (c & 1) * 8 + ((kq + (c >> 1)) & 7)
//   product 1 (lane = key, fixed channel): the 8 key quads of a channel sit in 8 different 16-byte bank groups
//       -> ds_read_b32 conflict free;
//   product 2 (lane = channel, fixed key quad): the 16 lanes of a ds_read_b128 group hit 16 different slots mod 16
//       -> conflict free; one b128 read = the 4 consecutive keys rho(4g..4g+3, h) of the accumulator layout.
// MFMA k-index of product 1: lane half h carries channel 64 h + t at step t (any bijection works as long as both
// operands agree), so a lane's 64 row values are contiguous in memory.
#include "infonce_common.hpp"
#include <stdlib.h>

constexpr int SM_WAVES = 8;
constexpr int SM_KEYS = 32 * SM_WAVES;                 // keys per workgroup tile
constexpr int SM_TILE_F = CH * 32;                     // floats of one wave's sub-tile (16 KB)
constexpr int SM_Q_F = 32 * CH;                        // floats of the shared row image (16 KB)
constexpr int SM_LDS = (SM_WAVES * SM_TILE_F + SM_Q_F + 3 * SM_WAVES * 32 + SM_WAVES * 64 + SM_WAVES * 32) * 4;

typedef float f32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ void glds16(const float* g, float* lds) {
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)g,
                                     (__attribute__((address_space(3))) void*)lds, 16, 0, 0);
}

// LDS reads with a hand-placed schedule.  hipcc sinks every ds_read to just before its consumer and waits
// lgkmcnt(0) there, which puts the ~100-cycle LDS latency between the 64-cycle MFMAs of a dependent chain; these
// asm reads are issued one group ahead and retired with counted waits (guide section 5.7, form (ii): the wait
// statement names every destination register "+v").
__device__ __forceinline__ unsigned lds_addr(const float* p) {
    return (unsigned)(size_t)(const __attribute__((address_space(3))) float*)p;
}
template <int OFF>
__device__ __forceinline__ void lds_read_b32(float& dst, unsigned addr) {
    asm volatile("ds_read_b32 %0, %1 offset:%2" : "=v"(dst) : "v"(addr), "i"(OFF));
}
template <int OFF>
__device__ __forceinline__ void lds_read_b128(f32x4& dst, unsigned addr) {
    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(dst) : "v"(addr), "i"(OFF));
}
template <int N>
__device__ __forceinline__ void lds_wait8(float (&v)[8]) {      // all but the N youngest LDS reads are done
    asm volatile("s_waitcnt lgkmcnt(%8)"
                 : "+v"(v[0]), "+v"(v[1]), "+v"(v[2]), "+v"(v[3]), "+v"(v[4]), "+v"(v[5]), "+v"(v[6]), "+v"(v[7])
                 : "i"(N));
}
template <int N>
__device__ __forceinline__ void lds_wait4x4(f32x4 (&v)[4]) {
    asm volatile("s_waitcnt lgkmcnt(%4)" : "+v"(v[0]), "+v"(v[1]), "+v"(v[2]), "+v"(v[3]) : "i"(N));
}
// group G of product 1: A values of steps t = 8 G .. 8 G + 7 (channel 64 h + t, this lane's key)
// The sub-tile's 16 DMA pieces are issued in the order 0, 8, 1, 9, ... (piece i = channels 8 i .. 8 i + 7), so group G
// needs everything up to the (2 G + 2)-th piece: vmcnt(14 - 2 G) -- product 1 starts while the rest of the sub-tile lands.
template <int G>
__device__ __forceinline__ void p1_read_group(float (&v)[8], const unsigned (&a1)[8]) {
    asm volatile("s_waitcnt vmcnt(%0)" ::"i"(14 - 2 * G) : "memory");
    lds_read_b32<(((8 * G + 0) >> 1) * 64 + 0) * 4>(v[0], a1[((8 * G + 0) >> 1) & 7]);
    lds_read_b32<(((8 * G + 1) >> 1) * 64 + 32) * 4>(v[1], a1[((8 * G + 1) >> 1) & 7]);
    lds_read_b32<(((8 * G + 2) >> 1) * 64 + 0) * 4>(v[2], a1[((8 * G + 2) >> 1) & 7]);
    lds_read_b32<(((8 * G + 3) >> 1) * 64 + 32) * 4>(v[3], a1[((8 * G + 3) >> 1) & 7]);
    lds_read_b32<(((8 * G + 4) >> 1) * 64 + 0) * 4>(v[4], a1[((8 * G + 4) >> 1) & 7]);
    lds_read_b32<(((8 * G + 5) >> 1) * 64 + 32) * 4>(v[5], a1[((8 * G + 5) >> 1) & 7]);
    lds_read_b32<(((8 * G + 6) >> 1) * 64 + 0) * 4>(v[6], a1[((8 * G + 6) >> 1) & 7]);
    lds_read_b32<(((8 * G + 7) >> 1) * 64 + 32) * 4>(v[7], a1[((8 * G + 7) >> 1) & 7]);
}
template <int G>
__device__ __forceinline__ f32x16 p1_group(f32x16 acc, float (&cur)[8], float (&nxt)[8], const unsigned (&a1)[8],
                                           const f32x4 (&bq)[CH / 8]) {
    if constexpr (G + 1 < 8) {
        p1_read_group<G + 1>(nxt, a1);
        lds_wait8<8>(cur);
    } else {
        lds_wait8<0>(cur);
    }
#pragma unroll
    for (int j = 0; j < 8; ++j) acc = mfma32(cur[j], bq[2 * G + (j >> 2)][j & 3], acc);
    return acc;
}
// Product 2 (the gradient's key sum  U^T[c][row] += sum_key T[c][key] p[key][row]) runs as split-bf16 on
// v_mfma_f32_32x32x16_bf16: every fp32 operand x = hi + lo (hi = bf16(x), lo = bf16(x - hi)) and every product =
// hi*hi + hi*lo + lo*hi with fp32 accumulation -- 24 MFMAs of 32 cycles per sub-tile instead of 64 of 64 cycles.
// The logits (product 1) stay exact fp32; the split only touches the 65536-term sum of the gradient, where the
// independent 2^-17-relative rounding errors average out: measured error 5e-7 * max|grad| (plain fp32 summation
// order noise: 1e-7), against a test bound of 2e-5.  Without this the kernel is bound by the f32 matrix pipe
// (2 x 128 MFMAs x 64 cycles per SIMD = 8.6 us at 1.9 GHz), not by the queue stream.
// k-step s of a 32-key sub-tile = accumulator registers 8 s .. 8 s + 7 of product 1, i.e. lane half h carries keys
// 16 s + 8 (j >> 2) + 4 h + (j & 3), j < 8: exactly the two 16-byte chunks kq = 4 s + h and 4 s + 2 + h of a channel.
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
__device__ __forceinline__ f32x16 mfma_bf(bf16x8 a, bf16x8 b, f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
}
__device__ __forceinline__ void split8(const float (&v)[8], bf16x8& hi, bf16x8& lo) {
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const __bf16 hj = (__bf16)v[j];
        hi[j] = hj;
        lo[j] = (__bf16)(v[j] - (float)hj);
    }
}
template <int N>
__device__ __forceinline__ void p2_read(f32x4& v, const unsigned (&a2)[4]) {      // chunk N = 4 cb + g of this lane's channel
    lds_read_b128<(N >> 2) * 4096>(v, a2[N & 3]);
}
template <int W>
__device__ __forceinline__ void lds_wait2(f32x4& v0, f32x4& v1) {
    asm volatile("s_waitcnt lgkmcnt(%2)" : "+v"(v0), "+v"(v1) : "i"(W));
}
// step M = 2 cb + s: chunks 2 M, 2 M + 1; the next step's two chunks are read before this one's are waited for
template <int M>
__device__ __forceinline__ void p2_step(f32x16 (&U)[4], f32x4& c0, f32x4& c1, f32x4& n0, f32x4& n1, const unsigned (&a2)[4],
                                        const bf16x8 (&ph)[2], const bf16x8 (&pl)[2]) {
    if constexpr (M + 1 < 8) {
        p2_read<2 * M + 2>(n0, a2);
        p2_read<2 * M + 3>(n1, a2);
        lds_wait2<2>(c0, c1);
    } else {
        lds_wait2<0>(c0, c1);
    }
    const float v[8] = {c0[0], c0[1], c0[2], c0[3], c1[0], c1[1], c1[2], c1[3]};
    bf16x8 ah, al;
    split8(v, ah, al);
    constexpr int cb = M >> 1, ks = M & 1;
    U[cb] = mfma_bf(ah, ph[ks], U[cb]);
    U[cb] = mfma_bf(al, ph[ks], U[cb]);
    U[cb] = mfma_bf(ah, pl[ks], U[cb]);
}

// Diagnostic build only (tools/rowkey_small_bench.hip defines CP2_STAMPS): shader-clock stamps of one wave per
// workgroup go to a buffer nothing else reads; the shipped kernel executes none of this.
#ifdef CP2_STAMPS
__device__ unsigned long long cp2_stamps[256 * 8 * 16];
#define CP2_STAMP(i) do { __builtin_amdgcn_sched_barrier(0); if (lane == 0) cp2_stamps[(blockIdx.x * 8 + w) * 16 + (i)] = (i) == 7 ? __builtin_amdgcn_s_memrealtime() : __builtin_amdgcn_s_memtime(); __builtin_amdgcn_sched_barrier(0); } while (0)
#else
#define CP2_STAMP(i) do {} while (0)
#endif

template <bool WITH_U>
__global__ __launch_bounds__(64 * SM_WAVES) void rowkey_small_kernel(RowKeyArgs a, int tiles_per_wg, int stagger) {
    extern __shared__ __attribute__((aligned(16))) float sm[];
    const int tid = threadIdx.x, lane = tid & 63, w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int r = lane & 31, h = lane >> 5;
    float* T = sm + w * SM_TILE_F;
    float* Q = sm + SM_WAVES * SM_TILE_F;                       // shared row image
    float* mbuf = Q + SM_Q_F;                                   // [8][32] merge scratch: max, sum, count
    float* sbuf = mbuf + SM_WAVES * 32;
    int* cbuf = reinterpret_cast<int*>(sbuf + SM_WAVES * 32);
    float* ebuf = sbuf + 2 * SM_WAVES * 32 + w * 64;            // [8][64] this wave's positive logits
    const bool row_ok = r < a.R;
    const int rclamp = row_ok ? r : 0;
    CP2_STAMP(7);
    CP2_STAMP(0);

    // Every load of this kernel is an LDS-DMA (hipcc drains vmcnt to 0 around ordinary loads that sit beside DMAs).
    // (1) the positive logit of this lane's row, 4-byte pieces
    if (a.NE > 0) {
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(a.extras + (int64_t)rclamp * a.NE),
                                         (__attribute__((address_space(3))) void*)ebuf, 4, 0, 0);
    }
    // (2) the row image (B operand of product 1), 16 KB shared by the 8 waves: wave w copies rows 4 w .. 4 w + 3.
    // Chunk (row rr, channel quad cq) sits in slot rr * 32 + (cq ^ rr): a lane's 16 chunks of channels 64 h .. + 63
    // are read by ds_read_b128 without bank conflicts.  Rows beyond R alias row 0 (finite data; their MFMA columns
    // are independent of the real rows and never written out).
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int rr = 4 * w + 2 * i + h, cq = r ^ rr;
        const int rs = rr < a.R ? rr : 0;
        glds16(a.rows + (int64_t)(rs / a.RP) * a.r_sn + (int64_t)(rs % a.RP) * a.r_sx + 4 * cq, Q + (2 * w + i) * 256);
    }
    // (3) key sub-tiles through a buffer descriptor: per-lane byte offset in ONE VGPR per piece parity, the piece's
    // channel step (8 i K floats) in the scalar offset -- no 64-bit address per piece to keep alive across the loop.
    // DMA piece i fills slots 64 i + lane: channel 8 i + c0, key quad kq0 ^ (4 (i & 1))
    const __amdgpu_buffer_rsrc_t krs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.keys), 0, CH * a.K * 4, 0x00020000);
    const int c0 = 2 * (lane >> 4) + ((lane >> 3) & 1);
    const int kq0 = ((lane & 7) - (lane >> 4)) & 7;
    auto issue_tile = [&](int kbase) {
        const int keyE = kbase + 4 * kq0, keyO = kbase + 4 * (kq0 ^ 4);
        const int voffE = (c0 * a.K + (keyE < a.K ? keyE : 0)) * 4;      // K % 4 == 0: a 16-byte chunk is all or nothing;
        const int voffO = (c0 * a.K + (keyO < a.K ? keyO : 0)) * 4;      // chunks past K read key 0 and are masked later
#pragma unroll
        for (int j = 0; j < 16; ++j) {
            const int i = (j >> 1) + 8 * (j & 1);              // 0, 8, 1, 9, ...: what product 1 needs first lands first
            __builtin_amdgcn_raw_ptr_buffer_load_lds(krs, (__attribute__((address_space(3))) void*)(T + i * 256), 16,
                                                     (i & 1) ? voffO : voffE, 8 * i * a.K * 4, 0, 0);
        }
    };
    const int kfirst = blockIdx.x * tiles_per_wg * SM_KEYS + w * 32;                  // wave-uniform
    // Waves 4-7 (the second wave of each SIMD) queue their sub-tile behind the first four: those then land at about
    // half time and compute while the others' data is still streaming, instead of all eight finishing together.
    if (w >= 4) {
        for (int i = 0; i < stagger; ++i) __builtin_amdgcn_s_sleep(4);
    }
    if (kfirst < a.K) {
        issue_tile(kfirst);
        asm volatile("s_waitcnt vmcnt(16)" ::: "memory");     // everything issued before the sub-tile has landed
    } else {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    CP2_STAMP(1);
    __builtin_amdgcn_s_barrier();                             // raw barrier (no vmcnt drain): all 8 parts of the row image are in LDS
    float pos_s;
    {
        float e = 0.f;
        lds_read_b32<0>(e, lds_addr(ebuf) + 4u * (unsigned)lane);
        asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(e));
        pos_s = (row_ok && a.NE > 0) ? e * a.inv_t : INFINITY;
    }
    const unsigned qb = lds_addr(Q) + 16u * (unsigned)(r * 32 + ((h ^ (r >> 4)) << 4));

    // product 1 read addresses: T[slot(64 h + t, r >> 2) * 4 + (r & 3)], rotation (kq + (t >> 1)) & 7 has period 8 in t >> 1
    const unsigned tbase = lds_addr(T);
    unsigned a1[8];
#pragma unroll
    for (int x = 0; x < 8; ++x) a1[x] = tbase + 4u * (unsigned)(32 * h * 64 + 4 * (((r >> 2) + x) & 7) + (r & 3));
    // product 2 read addresses: the chunk of channel cb * 32 + r holding keys 8 g + 4 h .. + 3
    unsigned a2[4];
#pragma unroll
    for (int g = 0; g < 4; ++g) a2[g] = tbase + 16u * (unsigned)((r >> 1) * 16 + (r & 1) * 8 + ((2 * g + h + (r >> 1)) & 7));

    float m_run = -INFINITY, s_run = 0.f;
    int cnt = 0;
    f32x16 U[4];
#pragma unroll
    for (int cb = 0; cb < 4; ++cb) U[cb] = (f32x16){0};

    for (int ti = 0; ti < tiles_per_wg; ++ti) {
        const int kbase = kfirst + ti * SM_KEYS;
        if (kbase >= a.K) break;

        // product 1: S^T[key = rho(reg, h)][row = r]
        f32x16 acc = {0};
        {
            // B operand of product 1, re-read from the LDS row image for every sub-tile: 64 registers that are live
            // only during product 1 instead of across the whole loop
            f32x4 bq[CH / 8];                                   // bq[j][e] = this lane's row at channel 64 h + 4 j + e
            {
                const unsigned x = r & 15;
                lds_read_b128<0>(bq[0], qb + 16u * (0 ^ x));   lds_read_b128<0>(bq[1], qb + 16u * (1 ^ x));
                lds_read_b128<0>(bq[2], qb + 16u * (2 ^ x));   lds_read_b128<0>(bq[3], qb + 16u * (3 ^ x));
                lds_read_b128<0>(bq[4], qb + 16u * (4 ^ x));   lds_read_b128<0>(bq[5], qb + 16u * (5 ^ x));
                lds_read_b128<0>(bq[6], qb + 16u * (6 ^ x));   lds_read_b128<0>(bq[7], qb + 16u * (7 ^ x));
                lds_read_b128<0>(bq[8], qb + 16u * (8 ^ x));   lds_read_b128<0>(bq[9], qb + 16u * (9 ^ x));
                lds_read_b128<0>(bq[10], qb + 16u * (10 ^ x)); lds_read_b128<0>(bq[11], qb + 16u * (11 ^ x));
                lds_read_b128<0>(bq[12], qb + 16u * (12 ^ x)); lds_read_b128<0>(bq[13], qb + 16u * (13 ^ x));
                lds_read_b128<0>(bq[14], qb + 16u * (14 ^ x)); lds_read_b128<0>(bq[15], qb + 16u * (15 ^ x));
                asm volatile("s_waitcnt lgkmcnt(0)"
                             : "+v"(bq[0]), "+v"(bq[1]), "+v"(bq[2]), "+v"(bq[3]), "+v"(bq[4]), "+v"(bq[5]), "+v"(bq[6]),
                               "+v"(bq[7]), "+v"(bq[8]), "+v"(bq[9]), "+v"(bq[10]), "+v"(bq[11]), "+v"(bq[12]), "+v"(bq[13]),
                               "+v"(bq[14]), "+v"(bq[15]));
            }
            float va[8], vb[8];
            p1_read_group<0>(va, a1);
            CP2_STAMP(2);
            acc = p1_group<0>(acc, va, vb, a1, bq);
            acc = p1_group<1>(acc, vb, va, a1, bq);
            acc = p1_group<2>(acc, va, vb, a1, bq);
            acc = p1_group<3>(acc, vb, va, a1, bq);
            acc = p1_group<4>(acc, va, vb, a1, bq);
            acc = p1_group<5>(acc, vb, va, a1, bq);
            acc = p1_group<6>(acc, va, vb, a1, bq);
            acc = p1_group<7>(acc, vb, va, a1, bq);
        }

        CP2_STAMP(3);
        if (a.lnegT && a.ln_sk == 1 && row_ok) {
            // row-major logits [R][K]: registers 4 g .. 4 g + 3 are four consecutive keys -> one 16-byte store each
            // (K % 4 == 0, so a quad is valid or invalid as a whole)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int key0 = kbase + 8 * g + 4 * h;
                if (key0 < a.K) {
                    const f32x4 v = {acc[4 * g], acc[4 * g + 1], acc[4 * g + 2], acc[4 * g + 3]};
                    *reinterpret_cast<f32x4*>(a.lnegT + (int64_t)r * a.ln_sr + key0) = v;
                }
            }
        }
        float sv[16];
        float tmax = -INFINITY;
#pragma unroll
        for (int reg = 0; reg < 16; ++reg) {
            const int key = kbase + rho(reg, h);
            const bool valid = key < a.K;
            if (a.lnegT && a.ln_sk != 1 && valid && row_ok) a.lnegT[(int64_t)key * a.ln_sk + (int64_t)r * a.ln_sr] = acc[reg];
            sv[reg] = valid ? acc[reg] * a.inv_t : -INFINITY;
            tmax = fmaxf(tmax, sv[reg]);
            cnt += (sv[reg] > pos_s) ? 1 : 0;
        }
        tmax = fmaxf(tmax, __shfl_xor(tmax, 32, 64));          // at least one key of the sub-tile is valid: tmax is finite
        if (ti == 0) {
            m_run = tmax;                                       // nothing accumulated yet: no rescale
        } else if (tmax > m_run) {
            const float sc = __expf(m_run - tmax);
            s_run *= sc;
            if (WITH_U) {
#pragma unroll
                for (int cb = 0; cb < 4; ++cb) U[cb] *= sc;
            }
            m_run = tmax;
        }
        CP2_STAMP(10);
        float (&p)[16] = sv;                                    // probabilities overwrite the scaled logits
#pragma unroll
        for (int reg = 0; reg < 16; ++reg) {
            p[reg] = __expf(sv[reg] - m_run);
            s_run += p[reg];
        }
        if (WITH_U) {
            // product 2: U^T[c = cb*32 + rho(reg', h)][row] += sum_key T[c][key] * p[key][row]
            bf16x8 ph[2], pl[2];
            {
                const float v0[8] = {p[0], p[1], p[2], p[3], p[4], p[5], p[6], p[7]};
                const float v1[8] = {p[8], p[9], p[10], p[11], p[12], p[13], p[14], p[15]};
                split8(v0, ph[0], pl[0]);
                split8(v1, ph[1], pl[1]);
            }
            CP2_STAMP(8);
            f32x4 x0, x1, y0, y1;
            p2_read<0>(x0, a2); p2_read<1>(x1, a2);
            p2_step<0>(U, x0, x1, y0, y1, a2, ph, pl);  p2_step<1>(U, y0, y1, x0, x1, a2, ph, pl);
            p2_step<2>(U, x0, x1, y0, y1, a2, ph, pl);  p2_step<3>(U, y0, y1, x0, x1, a2, ph, pl);
            CP2_STAMP(9);
            p2_step<4>(U, x0, x1, y0, y1, a2, ph, pl);  p2_step<5>(U, y0, y1, x0, x1, a2, ph, pl);
            p2_step<6>(U, x0, x1, y0, y1, a2, ph, pl);  p2_step<7>(U, y0, y1, x0, x1, a2, ph, pl);
        }
        CP2_STAMP(4);
        // every LDS read of this sub-tile has been retired by a counted wait above: the region may be overwritten
        if (ti + 1 < tiles_per_wg && kbase + SM_KEYS < a.K) issue_tile(kbase + SM_KEYS);
    }

    // ---- merge the 8 waves of the workgroup, then one coalesced write of the partial state
    const float s_tot = s_run + __shfl_xor(s_run, 32, 64);
    const int cnt_tot = cnt + __shfl_xor(cnt, 32, 64);
    if (h == 0) { mbuf[w * 32 + r] = m_run; sbuf[w * 32 + r] = s_tot; cbuf[w * 32 + r] = cnt_tot; }
    if (WITH_U) {
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
        for (int cb = 0; cb < 4; ++cb)
#pragma unroll
            for (int reg = 0; reg < 16; ++reg) T[(cb * 32 + rho(reg, h)) * 32 + r] = U[cb][reg];
    }
    CP2_STAMP(5);
    __syncthreads();
    const int slot = blockIdx.x;
    // per (wave, row) weights exp(m_w - M) once, by the first 256 threads; rows' merged (max, sum, count) by the first 32
    float* fbuf = reinterpret_cast<float*>(cbuf + SM_WAVES * 32) + SM_WAVES * 64;   // [8][32], behind the positives
    if (tid < SM_WAVES * 32) {
        const int rr = tid & 31;
        float M = -INFINITY;
#pragma unroll
        for (int j = 0; j < SM_WAVES; ++j) M = fmaxf(M, mbuf[j * 32 + rr]);
        const float mj = mbuf[tid];
        fbuf[tid] = (mj == -INFINITY) ? 0.f : __expf(mj - M);
        if (tid < 32) {
            float ss = 0.f;
            int cc = 0;
#pragma unroll
            for (int j = 0; j < SM_WAVES; ++j) {
                const float mw = mbuf[j * 32 + tid];
                ss += (mw == -INFINITY) ? 0.f : sbuf[j * 32 + tid] * __expf(mw - M);
                cc += cbuf[j * 32 + tid];
            }
            if (tid < a.R) {
                a.part_m[(int64_t)slot * a.R + tid] = M;
                a.part_s[(int64_t)slot * a.R + tid] = ss;
                a.part_cnt[(int64_t)slot * a.R + tid] = cc;
            }
        }
    }
    if (WITH_U) {
        __syncthreads();
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int e4 = tid + 512 * j;                 // float4 index into [C][32]
            const int c = e4 >> 3, r0 = (e4 & 7) * 4;
            f32x4 u = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int jw = 0; jw < SM_WAVES; ++jw) {
                const f32x4 v = *reinterpret_cast<const f32x4*>(sm + jw * SM_TILE_F + e4 * 4);
                const f32x4 f = *reinterpret_cast<const f32x4*>(fbuf + jw * 32 + r0);
                u += v * f;
            }
            float* dst = a.part_U + ((int64_t)slot * CH + c) * a.R;
            if (a.R == 32) {
                *reinterpret_cast<f32x4*>(dst + r0) = u;
            } else {
#pragma unroll
                for (int e = 0; e < 4; ++e)
                    if (r0 + e < a.R) dst[r0 + e] = u[e];
            }
        }
    }
    CP2_STAMP(6);
}

#ifdef CP2_STAMPS
__global__ void cp2_stamp_end_kernel() {}
#endif

int rowkey_small_num_splits(int K, int* tiles_per_wg) {
    const int tiles = cp2_cdiv(K, SM_KEYS);
    const int tpw = cp2_cdiv(tiles, 256);              // one workgroup per CU (144 KB of LDS each)
    if (tiles_per_wg) *tiles_per_wg = tpw;
    return cp2_cdiv(tiles, tpw);
}

int rowkey_small_launch(const RowKeyArgs& a, int nsplit, bool with_u, hipStream_t stream) {
    int tpw = 1;
    if (rowkey_small_num_splits(a.K, &tpw) != nsplit) return CP2_ERR_SHAPE;
    using KFn = void (*)(RowKeyArgs, int, int);
    const int v = with_u ? 1 : 0;
    static const KFn kfns[2] = {rowkey_small_kernel<false>, rowkey_small_kernel<true>};
    const KFn kfn = kfns[v];
    const int lds = SM_LDS;
    static bool attr_set[2] = {false, false};          // (idempotent cache of a per-function attribute, not library state)
    if (!attr_set[v]) {
        hipError_t e_ = hipFuncSetAttribute(reinterpret_cast<const void*>(kfn), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
        if (e_ != hipSuccess) return (int)e_;
        attr_set[v] = true;
    }
    static int stagger = -1;                           // tuning knob (units of 256 cycles), default chosen by measurement
    if (stagger < 0) {
        const char* e = getenv("CP2_ROWKEY_STAGGER");
        stagger = e ? atoi(e) : 0;   // measured: no gain from staggering (f32 MFMA and VALU share one pipe)
    }
    CP2_LAUNCH_PROFILED(kfn, dim3(nsplit), dim3(64 * SM_WAVES), lds, stream, a, tpw, stagger);
    return cp2_launch_status();
}

#include "rowkey_small_fin.hpp"

__global__ __launch_bounds__(32 * FS2_SL) void rowkey_small_finalize_kernel(RowKeyFinArgs a, float* __restrict__ loss_mean) {
    rowkey_small_finalize_body(a, loss_mean, (int)blockIdx.x);
}

int rowkey_small_finalize_launch(const RowKeyFinArgs& a, float* loss_mean, hipStream_t stream) {
    hipLaunchKernelGGL(rowkey_small_finalize_kernel, dim3(CH / FS2_CPB), dim3(32 * FS2_SL), 0, stream, a, loss_mean);
    return cp2_launch_status();
}
