// a16 (BASELINE config 5, DenseCL local loss): thousands of per-pixel rows against the queue, split-bf16 on the matrix
// cores -- reference builder.py:866-873 (neg_local = q_local @ queue2), :906-908 + :150-176 (cross entropy, target 0).
//
// Same arithmetic as rowkey_fwd_bf16x3_kernel (infonce.hip): every fp32 operand x = hi + lo (hi = bf16(x),
// lo = bf16(x - hi)), every product = hi*hi + hi*lo + lo*hi on v_mfma_f32_32x32x16_bf16 with fp32 accumulation, the
// "transposed flash" orientation (rows on the lanes, the soft-max accumulator registers are the B operand of the
// gradient product).  What changes is how the key tiles reach the matrix cores:
//   * the queue's hi / lo split, key-major [K][C] and channel-major [C][K], is written once per call by
//     keys_split_kernel (as before); the channel-major image now stores every 16-key block in the order
//     [0-3, 8-11, 4-7, 12-15], i.e. in the accumulator's row order, so one 16-byte read is one MFMA fragment;
//   * tiles of 32 keys (4 images x 8 KB) go global -> LDS by LDS-DMA (buffer_load_dwordx4 ... lds: no staging
//     registers, the old kernel spilled at 256 VGPRs) into TWO buffers: the DMA of tile t+1 is issued before tile t is
//     computed and retired with a counted vmcnt, so the L2 / HBM latency of a fill is no longer exposed once per tile
//     (round 1: 28.5 % MFMA utilisation, the fill was synchronous between two barriers);
//   * DMA writes lane-linear, so the padding of the old images becomes an XOR swizzle applied to each lane's SOURCE
//     address and to the read address: 16-byte chunk j of key row r sits at chunk j ^ (r & 15) (key-major image,
//     256-byte rows), chunk j of channel row c at j ^ ((c >> 2) & 3) (channel-major image, 64-byte rows): both
//     products read their fragments with conflict-free ds_read_b128;
//   * fragment reads are hand-placed asm (one k-step ahead of the MFMAs that consume them) -- hipcc would wait
//     vmcnt(0) before a visible LDS read behind a DMA and drain the prefetch.
#include "infonce_common.hpp"
#include <type_traits>

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int DK = 32;                       // keys per tile
constexpr int D_T1 = DK * 256;               // one key-major image (hi or lo): 32 rows x 128 channels x 2 B
constexpr int D_T2 = CH * 64;                // one channel-major image: 128 rows x 32 keys x 2 B
constexpr int D_BUF = 2 * D_T1 + 2 * D_T2;   // T1 hi | T1 lo | T2 hi | T2 lo = 32 KB
constexpr int D_LDS = 2 * D_BUF;             // two buffers

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
template <int OFF>
__device__ __forceinline__ void lds_read_frag(bf16x8& dst, unsigned addr) {
    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(dst) : "v"(addr), "i"(OFF));
}
template <int N>
__device__ __forceinline__ void lds_wait_frag2(bf16x8& a, bf16x8& b) {      // all but the N youngest LDS reads are done
    asm volatile("s_waitcnt lgkmcnt(%2)" : "+v"(a), "+v"(b) : "i"(N));
}

// product 1, k-step ST (channels 16 ST .. 16 ST + 15) of the tile in buffer BUF: fragments of ST + 1 are read first
template <int BUF, int ST>
__device__ __forceinline__ f32x16 p1_step(f32x16 acc, bf16x8& ah, bf16x8& al, bf16x8& nh, bf16x8& nl, const unsigned (&o1)[8],
                                          const bf16x8 (&bqh)[CH / 16], const bf16x8 (&bql)[CH / 16]) {
    if constexpr (ST + 1 < CH / 16) {
        lds_read_frag<BUF * D_BUF>(nh, o1[ST + 1]);
        lds_read_frag<BUF * D_BUF + D_T1>(nl, o1[ST + 1]);
        lds_wait_frag2<2>(ah, al);
    } else {
        lds_wait_frag2<0>(ah, al);
    }
    acc = mfma_bf(ah, bqh[ST], acc);
    acc = mfma_bf(ah, bql[ST], acc);
    acc = mfma_bf(al, bqh[ST], acc);
    return acc;
}
// product 2, step M = 2 cb + ks: channels cb*32 .. +31, keys of k-step ks
template <int BUF, int M>
__device__ __forceinline__ void p2_step(f32x16 (&U)[4], bf16x8& ah, bf16x8& al, bf16x8& nh, bf16x8& nl, const unsigned (&o2)[2],
                                        const bf16x8 (&ph)[2], const bf16x8 (&pl)[2]) {
    if constexpr (M + 1 < 8) {
        constexpr int cb1 = (M + 1) >> 1, ks1 = (M + 1) & 1;
        lds_read_frag<BUF * D_BUF + 2 * D_T1 + cb1 * 2048>(nh, o2[ks1]);
        lds_read_frag<BUF * D_BUF + 2 * D_T1 + D_T2 + cb1 * 2048>(nl, o2[ks1]);
        lds_wait_frag2<2>(ah, al);
    } else {
        lds_wait_frag2<0>(ah, al);
    }
    constexpr int cb = M >> 1, ks = M & 1;
    U[cb] = mfma_bf(ah, ph[ks], U[cb]);
    U[cb] = mfma_bf(al, ph[ks], U[cb]);
    U[cb] = mfma_bf(ah, pl[ks], U[cb]);
}

template <bool WITH_U>
__global__ __launch_bounds__(256, 2) void rowkey_bf16x3_dma_kernel(RowKeyArgs a, const __bf16* __restrict__ ksplit) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smd[];
    const int tid = threadIdx.x, lane = tid & 63, wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int r = lane & 31, h = lane >> 5;
    const int row = (blockIdx.x * 4 + wid) * 32 + r;
    const bool row_ok = row < a.R;
    const int k_begin = blockIdx.y * a.keys_per_split;
    const int k_end = min(a.K, k_begin + a.keys_per_split);
    const int ntiles = (k_end - k_begin + DK - 1) / DK;

    // ---- DMA plan: 32 pieces of 1 KB per tile, wave w issues pieces w, w + 4, ...; piece p fills bytes [1024 p, +1024)
    // of the buffer.  Pieces 0-15: key-major images (rows of 256 B, 4 per piece), 16-31: channel-major (rows of 64 B, 16).
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<__bf16*>(ksplit), 0, 8 * CH * a.K, 0x00020000);
    const int CK2 = CH * a.K * 2;                                  // bytes of one of the four split arrays
    int voff[8];                                                   // per-lane byte offset of piece wid + 4 i at key 0
    {
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int p = wid + 4 * i, img = p >> 3, q = p & 7;    // img: 0 T1hi, 1 T1lo, 2 T2hi, 3 T2lo (wave-uniform)
            if (img < 2) {
                const int krow = 4 * q + (lane >> 4), chunk = (lane & 15) ^ (krow & 15);
                voff[i] = img * CK2 + (krow * CH + chunk * 8) * 2;
            } else {
                const int c = 16 * q + (lane >> 2), chunk = (lane & 3) ^ ((c >> 2) & 3);
                voff[i] = img * CK2 + (c * a.K + chunk * 8) * 2;
            }
        }
    }
    auto issue_tile = [&](int t, int buf) {
        const int k0 = k_begin + t * DK;
        unsigned char* dst = smd + buf * D_BUF;
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int p = wid + 4 * i;
            const int soff = (p < 16) ? k0 * CH * 2 : k0 * 2;      // key-major: whole rows; channel-major: along the row
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (__attribute__((address_space(3))) void*)(dst + p * 1024), 16, voff[i], soff, 0, 0);
        }
    };
    issue_tile(0, 0);

    // ---- this lane's row, split (B operand of product 1)
    bf16x8 bqh[CH / 16], bql[CH / 16];
    {
        const int rr = row_ok ? row : 0;
        const float* base = a.rows + (int64_t)(rr / a.RP) * a.r_sn + (int64_t)(rr % a.RP) * a.r_sx;
#pragma unroll
        for (int st = 0; st < CH / 16; ++st) {
            float v[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) v[j] = base[(int64_t)(16 * st + 8 * h + j) * a.r_sc];
            split8(v, bqh[st], bql[st]);
        }
    }
    const float pos_s = (row_ok && a.NE > 0) ? a.extras[(int64_t)row * a.NE] * a.inv_t : INFINITY;

    // ---- fragment addresses (buffer 0; the buffer / image selection is an immediate offset)
    const unsigned sbase = (unsigned)(size_t)(const __attribute__((address_space(3))) unsigned char*)smd;
    unsigned o1[8], o2[2];
#pragma unroll
    for (int st = 0; st < 8; ++st) o1[st] = sbase + (unsigned)(r * 256 + (((2 * st + h) ^ (r & 15)) * 16));
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) o2[ks] = sbase + (unsigned)(r * 64 + (((2 * ks + h) ^ ((r >> 2) & 3)) * 16));

    // ---- raw logits for the score statistics, row-major [row][key] (ln_sk == 1): the accumulator holds KEYS along the
    // registers and ROWS along the lanes, so a direct store is 64 four-byte requests to 32 different rows per instruction
    // -- with two workgroups per CU the texture addresser, not the matrix pipe, then bounds the launch (round 4: 370 us
    // per 1568 x 65536 chunk against 150 us without the logits).  Each wave turns its 32 x 32 tile through 4 KB of LDS of
    // its own (behind the two DMA buffers) and writes eight whole 128-byte row segments per instruction.
    const bool ln_vec = __builtin_amdgcn_readfirstlane(
        (a.lnegT != nullptr && a.ln_sk == 1 && (a.ln_sr & 3) == 0 && (((size_t)a.lnegT) & 15) == 0 && (k_begin & 3) == 0) ? 1 : 0);
    const unsigned tbase = sbase + D_LDS + wid * 4096;
    unsigned tw[4], tr[4];
#pragma unroll
    for (int g = 0; g < 4; ++g) {
        tw[g] = tbase + (unsigned)(r * 128 + (((2 * g + h) ^ (r & 7)) * 16));                 // keys 8g + 4h .. +3 of row r
        const int rr = 8 * g + (lane >> 3);
        tr[g] = tbase + (unsigned)(rr * 128 + (((lane & 7) ^ (rr & 7)) * 16));                // keys 4 (lane & 7) .. +3 of row rr
    }
    const int wave_row0 = (blockIdx.x * 4 + wid) * 32;

    float m_run = -INFINITY, s_run = 0.f;
    int cnt = 0;
    f32x16 U[4];
#pragma unroll
    for (int cb = 0; cb < 4; ++cb) U[cb] = (f32x16){0};

    auto compute = [&](auto BUFC, int k0) {
        constexpr int BUF = decltype(BUFC)::value;
        f32x16 acc = {0};
        {
            bf16x8 xh, xl, yh, yl;
            lds_read_frag<BUF * D_BUF>(xh, o1[0]);
            lds_read_frag<BUF * D_BUF + D_T1>(xl, o1[0]);
            acc = p1_step<BUF, 0>(acc, xh, xl, yh, yl, o1, bqh, bql);
            acc = p1_step<BUF, 1>(acc, yh, yl, xh, xl, o1, bqh, bql);
            acc = p1_step<BUF, 2>(acc, xh, xl, yh, yl, o1, bqh, bql);
            acc = p1_step<BUF, 3>(acc, yh, yl, xh, xl, o1, bqh, bql);
            acc = p1_step<BUF, 4>(acc, xh, xl, yh, yl, o1, bqh, bql);
            acc = p1_step<BUF, 5>(acc, yh, yl, xh, xl, o1, bqh, bql);
            acc = p1_step<BUF, 6>(acc, xh, xl, yh, yl, o1, bqh, bql);
            acc = p1_step<BUF, 7>(acc, yh, yl, xh, xl, o1, bqh, bql);
        }
        if (ln_vec) {
            f32x4 tv[4];
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const f32x4 w4 = {acc[4 * g], acc[4 * g + 1], acc[4 * g + 2], acc[4 * g + 3]};
                asm volatile("ds_write_b128 %0, %1" :: "v"(tw[g]), "v"(w4));
            }
#pragma unroll
            for (int g = 0; g < 4; ++g) asm volatile("ds_read_b128 %0, %1" : "=v"(tv[g]) : "v"(tr[g]));   // same wave: LDS keeps the order
            asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(tv[0]), "+v"(tv[1]), "+v"(tv[2]), "+v"(tv[3]));
            const int key = k0 + 4 * (lane & 7);
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int grow = wave_row0 + 8 * g + (lane >> 3);
                if (grow < a.R) {
                    float* dst = a.lnegT + (int64_t)grow * a.ln_sr + key;
                    if (key + 3 < k_end) {
                        *reinterpret_cast<f32x4*>(dst) = tv[g];
                    } else {
#pragma unroll
                        for (int j = 0; j < 4; ++j)
                            if (key + j < k_end) dst[j] = tv[g][j];
                    }
                }
            }
        }
        float sv[16];
        float tmax = -INFINITY;
#pragma unroll
        for (int reg = 0; reg < 16; ++reg) {
            const int key = k0 + rho(reg, h);
            const bool valid = key < k_end;
            if (!ln_vec && a.lnegT && valid && row_ok) a.lnegT[(int64_t)key * a.ln_sk + (int64_t)row * a.ln_sr] = acc[reg];
            sv[reg] = valid ? acc[reg] * a.inv_t : -INFINITY;
            tmax = fmaxf(tmax, sv[reg]);
            cnt += (sv[reg] > pos_s) ? 1 : 0;
        }
        tmax = fmaxf(tmax, __shfl_xor(tmax, 32, 64));
        if (tmax > m_run) {
            const float sc = __expf(m_run - tmax);              // exp(-inf) = 0 on the first tile
            s_run *= sc;
            if (WITH_U) {
#pragma unroll
                for (int cb = 0; cb < 4; ++cb) U[cb] *= sc;
            }
            m_run = tmax;
        }
#pragma unroll
        for (int reg = 0; reg < 16; ++reg) {
            sv[reg] = __expf(sv[reg] - m_run);
            s_run += sv[reg];
        }
        if (WITH_U) {
            bf16x8 ph[2], pl[2];
            {
                const float v0[8] = {sv[0], sv[1], sv[2], sv[3], sv[4], sv[5], sv[6], sv[7]};
                const float v1[8] = {sv[8], sv[9], sv[10], sv[11], sv[12], sv[13], sv[14], sv[15]};
                split8(v0, ph[0], pl[0]);
                split8(v1, ph[1], pl[1]);
            }
            bf16x8 xh, xl, yh, yl;
            lds_read_frag<BUF * D_BUF + 2 * D_T1>(xh, o2[0]);
            lds_read_frag<BUF * D_BUF + 2 * D_T1 + D_T2>(xl, o2[0]);
            p2_step<BUF, 0>(U, xh, xl, yh, yl, o2, ph, pl);  p2_step<BUF, 1>(U, yh, yl, xh, xl, o2, ph, pl);
            p2_step<BUF, 2>(U, xh, xl, yh, yl, o2, ph, pl);  p2_step<BUF, 3>(U, yh, yl, xh, xl, o2, ph, pl);
            p2_step<BUF, 4>(U, xh, xl, yh, yl, o2, ph, pl);  p2_step<BUF, 5>(U, yh, yl, xh, xl, o2, ph, pl);
            p2_step<BUF, 6>(U, xh, xl, yh, yl, o2, ph, pl);  p2_step<BUF, 7>(U, yh, yl, xh, xl, o2, ph, pl);
        }
    };

    // ---- main loop, two tiles per trip so the buffer index is a compile-time constant
    for (int t = 0; t < ntiles; t += 2) {
        if (t + 1 < ntiles) {
            issue_tile(t + 1, 1);
            asm volatile("s_waitcnt vmcnt(8)" ::: "memory");      // this wave's pieces of tile t have landed, tile t+1 is in flight
        } else {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        __builtin_amdgcn_s_barrier();                              // ... and everybody else's (raw barrier: no vmcnt drain)
        compute(std::integral_constant<int, 0>{}, k_begin + t * DK);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();                              // buffer 0 may be refilled
        if (t + 1 >= ntiles) break;
        if (t + 2 < ntiles) {
            issue_tile(t + 2, 0);
            asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
        } else {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        __builtin_amdgcn_s_barrier();
        compute(std::integral_constant<int, 1>{}, k_begin + (t + 1) * DK);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();                              // buffer 1 may be refilled
    }

    const float s_tot = s_run + __shfl_xor(s_run, 32, 64);
    const int cnt_tot = cnt + __shfl_xor(cnt, 32, 64);
    const int slot = blockIdx.y;
    if (row_ok) {
        if (h == 0) {
            a.part_m[(int64_t)slot * a.R + row] = m_run;
            a.part_s[(int64_t)slot * a.R + row] = s_tot;
            a.part_cnt[(int64_t)slot * a.R + row] = cnt_tot;
        }
        if (WITH_U) {
#pragma unroll
            for (int cb = 0; cb < 4; ++cb)
#pragma unroll
                for (int reg = 0; reg < 16; ++reg)
                    a.part_U[((int64_t)slot * CH + cb * 32 + rho(reg, h)) * a.R + row] = U[cb][reg];
        }
    }
}

int rowkey_bf16x3_dma_launch(const RowKeyArgs& a, const void* ksplit, dim3 grid, bool with_u, hipStream_t stream) {
    auto kfn = with_u ? rowkey_bf16x3_dma_kernel<true> : rowkey_bf16x3_dma_kernel<false>;
    // 4 KB per wave more when the launch also writes row-major logits (the in-LDS turn of the tiles): 2 x 80 KB = the CU's LDS
    const int lds = D_LDS + ((a.lnegT != nullptr && a.ln_sk == 1) ? 4 * 4096 : 0);
    hipError_t e_ = hipFuncSetAttribute(reinterpret_cast<const void*>(kfn), hipFuncAttributeMaxDynamicSharedMemorySize, D_LDS + 4 * 4096);
    if (e_ != hipSuccess) return (int)e_;
    CP2_LAUNCH_PROFILED(kfn, grid, dim3(256), lds, stream, a, static_cast<const __bf16*>(ksplit));
    return cp2_launch_status();
}
