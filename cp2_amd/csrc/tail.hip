// The tail of a training step's loss section as ONE launch of independent 256-thread workgroups (round 4):
//   block 0                 every scalar the step returns or logs                  (cp2_step_scalars; a15 + the loss combination)
//   next n/32 x C/32 blocks the keys' enqueue with wrap-around, pointer on the device (cp2_enqueue; a13, builder.py:569-587)
//   next 2 B blocks         IoU / masked IoU of the down-sampled region-id maps     (cp2_corr_iou*, hash form; a3-a5)
// None of the three reads what another writes -- the IoUs are logging only, the enqueue writes queue columns the loss kernels
// have already read (stream order), the scalars read the loss kernels' outputs -- so they need no order among themselves:
// round 3 launched them as three kernels of 5-9 us each, launch-bound.  Each entry point below launches the same kernel
// with only its own part switched on; cp2_step_tail switches on all three.
// ---- part 0: a15 (batch level) + the loss combination: every scalar a CP2 step returns or logs, in ONE launch.
// The reference forms them with ~30 small torch reductions and `.item()` calls per step (builder.py:1431-1448 loss,
// :1441 top-1 / top-5, :1265,1282 cross-image spread of the pooled vectors, :1553-1604 the wandb scalars); round 2 kept
// them as ATen launches (mean / std / stack / comparison kernels, ~4 us each on a GPU-bound step).  One workgroup:
//   out[ 0] loss = loss_instance + lambda * loss_dense           (builder.py:1437)
//   out[ 1] loss_instance (rows-vs-queue InfoNCE, already the batch mean)
//   out[ 2] loss_dense    = mean_n sample[n][2]
//   out[ 3] top-1 %       = 100 * mean_n [cnt_gt[n] < 1]         (builder.py:1690-1706 accuracy of the instance logits)
//   out[ 4] top-5 %       = 100 * mean_n [cnt_gt[n] < 5]
//   out[ 5] dense arg-max accuracy % = 100 * mean_n sample[n][5] (builder.py:1442-1448)
//   out[ 6] mean positive dense score = mean_n sample[n][3];  out[7] mean negative dense score = mean_n sample[n][4]
//   out[ 8] mean raw positive instance logit = mean_n extras[n][0]
//   out[ 9] mean_c std_n(q_pos[n][c])  (unbiased);  out[10] the same of k_pos
//   out[11..13] / [14..16] / [17..19]  batch means of the lower / median / upper quartiles of the positive dense, negative
//               dense and queue logits;  out[20] batch mean of the queue logits' row means   (NaN-propagating, as .mean())
// Sums run in double in a fixed order (deterministic); results are rounded to fp32 once.
#include "common.hpp"
#include <math.h>

struct StepScalarArgs {
    const float* ins_loss; const int32_t* cnt_gt; const float* extras; int NE;
    const float* sample;                         // [B][8]
    const float* q_pos; const float* k_pos;      // [B][C]
    const float* quart[3];                       // [3][B] each, or NULL
    const float* lneg_mean;                      // [B] or NULL
    float lmbd; int B, C;
    float* out;                                  // [CP2_STEP_SCALARS]
};

__device__ __forceinline__ double wave_sum_d(double v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
    return v;
}

__device__ __forceinline__ void step_scalars_body(const StepScalarArgs& a) {
    __shared__ double red[4];
    __shared__ double col[8];
    // Both [B][C] matrices of the cross-image spread are staged in LDS when they fit (the training step: 32 x 128 floats each);
    // their loads are issued FIRST, so the column sums and quartile means below run while they are in flight and the launch
    // pays one memory round trip, not one per section (12.7 -> 8 us; a first version walked global memory per channel: 24 us).
    constexpr int kTile = 4096;
    __shared__ float tile[2][kTile];
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, B = a.B;
    const int BC = B * a.C;
    const bool staged = BC <= kTile && (BC & 3) == 0 && ((reinterpret_cast<uintptr_t>(a.q_pos) | reinterpret_cast<uintptr_t>(a.k_pos)) & 15u) == 0;
    if (staged) {
        for (int e = tid; e < BC / 4; e += 256) {
            const float4 vq = reinterpret_cast<const float4*>(a.q_pos)[e], vk = reinterpret_cast<const float4*>(a.k_pos)[e];
            reinterpret_cast<float4*>(tile[0])[e] = vq;
            reinterpret_cast<float4*>(tile[1])[e] = vk;
        }
    }
    // ---- wave 0: the per-sample columns (B is a batch size: one short loop per lane)
    if (w == 0) {
        double s[8] = {0, 0, 0, 0, 0, 0, 0, 0};
        for (int n = lane; n < B; n += 64) {
            const float* sc = a.sample + (int64_t)n * 8;
            s[0] += sc[2], s[1] += sc[5], s[2] += sc[3], s[3] += sc[4];
            s[4] += a.cnt_gt[n] < 1 ? 1.0 : 0.0;
            s[5] += a.cnt_gt[n] < 5 ? 1.0 : 0.0;
            s[6] += a.extras[(int64_t)n * a.NE];
            if (a.lneg_mean) s[7] += a.lneg_mean[n];
        }
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const double t = wave_sum_d(s[j]);
            if (lane == 0) col[j] = t;
        }
    }
    // ---- waves 1-3: batch means of the quartile triples, one wave per statistic
    if (w >= 1 && a.quart[w - 1]) {
        const float* q = a.quart[w - 1];
        for (int j = 0; j < 3; ++j) {
            double s = 0;
            for (int n = lane; n < B; n += 64) s += q[(int64_t)j * B + n];
            s = wave_sum_d(s);
            if (lane == 0) a.out[11 + 3 * (w - 1) + j] = (float)(s / B);
        }
    }
    __syncthreads();
    if (tid == 0) {
        const double inv = 1.0 / B;
        const float l_ins = a.ins_loss[0], l_den = (float)(col[0] * inv);
        a.out[0] = l_ins + l_den * a.lmbd;            // the expression the fp32 graph evaluated: ins + dense * lambda
        a.out[1] = l_ins, a.out[2] = l_den;
        a.out[3] = (float)(100.0 * col[4] * inv), a.out[4] = (float)(100.0 * col[5] * inv);
        a.out[5] = (float)(100.0 * col[1] * inv);
        a.out[6] = (float)(col[2] * inv), a.out[7] = (float)(col[3] * inv), a.out[8] = (float)(col[6] * inv);
        a.out[20] = a.lneg_mean ? (float)(col[7] * inv) : 0.f;
        for (int j = 0; j < 3; ++j)
            for (int k = 0; k < 3; ++k)
                if (!a.quart[j]) a.out[11 + 3 * j + k] = 0.f;
    }
    // ---- unbiased std over the batch per channel, mean over channels (two passes, double): waves 0-1 take q_pos, waves 2-3
    // k_pos; a thread owns a channel and walks the batch (in LDS when staged)
    const int side = tid >> 7, t2 = tid & 127;
    const float* src = staged ? tile[side] : (side ? a.k_pos : a.q_pos);
    double acc = 0;
    for (int c = t2; c < a.C; c += 128) {
        double m = 0;
        for (int n = 0; n < B; ++n) m += src[(int64_t)n * a.C + c];
        m /= B;
        double ss = 0;
        for (int n = 0; n < B; ++n) { const double d = src[(int64_t)n * a.C + c] - m; ss += d * d; }
        acc += sqrt(ss / (B - 1));                     // B = 1: 0/0 = NaN, as torch.std
    }
    acc = wave_sum_d(acc);
    if (lane == 0) red[w] = acc;
    __syncthreads();
    if (tid < 2) a.out[9 + tid] = (float)((red[2 * tid] + red[2 * tid + 1]) / a.C);
}


// ---- part 1: a13, queue enqueue with wrap-around -- reference builder.py:569-587.
// keys [n,C] (row = key) are transposed into columns (ptr+i) % K of queue [C,K].  32x32 tiles go through LDS so both the
// read (along C) and the write (along K) are coalesced.  The pointer lives on the device (the reference does
// int(queue_ptr), a host synchronisation): every enqueue workgroup reads it before it stores anything, and the one that
// draws the last ticket of a caller-owned counter advances it.
struct EnqueueArgs {
    float* queue; const float* keys; int64_t* ptr; int32_t* ticket;
    int n, C, K;
    int bx, by;                                  // tiles of 32 keys x 32 channels: bx * by workgroups (0: part switched off)
};

__device__ __forceinline__ void enqueue_body(const EnqueueArgs& a, int b) {
    __shared__ float tile[32][33];
    const int i0 = (b % a.bx) * 32;  // key tile
    const int c0 = (b / a.bx) * 32;  // channel tile
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;  // 32 x 8
    const int64_t p = *a.ptr;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int i = i0 + ty + 8 * r, c = c0 + tx;
        tile[ty + 8 * r][tx] = (i < a.n && c < a.C) ? a.keys[(int64_t)i * a.C + c] : 0.0f;
    }
    __syncthreads();
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int c = c0 + ty + 8 * r, i = i0 + tx;
        if (i < a.n && c < a.C) a.queue[(int64_t)c * a.K + (p + i) % a.K] = tile[tx][ty + 8 * r];
    }
    // every thread of this workgroup holds p in a register by now (its stores were addressed with it)
    __syncthreads();
    if (threadIdx.x == 0) {
        if (atomicAdd(a.ticket, 1) == a.bx * a.by - 1) {   // all other enqueue workgroups have read the pointer: advance it
            *a.ptr = (p + a.n) % a.K;
            *a.ticket = 0;                                  // ready for the next launch (stream order makes it visible)
        }
    }
}

// ---- part 2: a3-a5, IoU of two id maps by counting keys in an LDS hash table -- tools/correlation_mapping.py:103-138
// The same counts without a sort, for maps of up to 4096 cells (the training shapes: P = 196 ... 4096): the
// 2P+1 keys go into an open-addressing table in LDS, load factor <= 1/2.  A slot is the key's bit pattern; "hit at least
// twice" is one bit per slot in a bitmap behind the table: 4 bytes + 1 bit per slot, 66 KB for config 4's 4096-cell maps
// (round 4; with a 4-byte count per slot the table stopped at 2047 cells and config 4 fell back to the 204 us
// bitonic-sort launch);
//   union = #occupied slots - 1 (the zero key is always present),  intersection = #slots of a non-zero key hit >= 2 times.
// The bitonic network above needs 45 barrier stages for 512 keys (16-18 us per launch, latency); this form needs two.
// Integer counting only: bit-equal to the sorted form (and to the reference's torch.unique arithmetic) by construction.
struct IouArgs {
    const int64_t* ids_a; const int64_t* ids_b; const float* mask_a; const float* mask_b;
    float* iou; float* iou_masked;
    int P, T, H, W, stride, Ws;
    int B;                                       // 2 B workgroups (0: part switched off)
};

__device__ __forceinline__ void corr_iou_hash_body(const IouArgs& a, int b, unsigned* tab /* dynamic LDS: [T] key bits | [T / 32] twice bits */) {
    __shared__ int red[2][4];
    constexpr unsigned kEmpty = 0xFFFFFFFFu;                             // a NaN pattern: never a key
    const int64_t* __restrict__ ids_a = a.ids_a;
    const int64_t* __restrict__ ids_b = a.ids_b;
    const float* __restrict__ mask_a = a.mask_a;
    const float* __restrict__ mask_b = a.mask_b;
    const int P = a.P, T = a.T, H = a.H, W = a.W, stride = a.stride, Ws = a.Ws;
    const int n = b >> 1;
    const bool masked = (b & 1) == 1;
    float* out = masked ? a.iou_masked : a.iou;
    if (!out) return;
    unsigned* keys = tab;
    unsigned* twice = tab + T;
    for (int i = threadIdx.x; i < T; i += 256) keys[i] = kEmpty;
    for (int i = threadIdx.x; i < T / 32; i += 256) twice[i] = 0u;
    __syncthreads();
    auto at = [&](int p) -> int64_t {
        if (stride <= 0) return (int64_t)n * P + p;
        const int off = stride >> 1;
        return ((int64_t)n * H + off + (int64_t)stride * (p / Ws)) * W + off + (int64_t)stride * (p % Ws);
    };
    const int nvalid = 2 * P + 1;
    int shift = 0;
    while ((1 << shift) < T) ++shift;
    for (int i = threadIdx.x; i < nvalid; i += 256) {
        float v = 0.0f;
        if (i >= 1 && i <= P) {
            const float idf = (float)(ids_a[at(i - 1)] + 1);
            v = masked ? __fmul_rn(idf, mask_a[(int64_t)n * P + (i - 1)]) : idf;
        } else if (i > P) {
            const float idf = (float)(ids_b[at(i - 1 - P)] + 1);
            v = masked ? __fmul_rn(idf, mask_b[(int64_t)n * P + (i - 1 - P)]) : idf;
        }
        if (v == 0.0f) v = 0.0f;                                        // -0.0 and +0.0 are one key (torch.unique compares values)
        const unsigned bits = __float_as_uint(v);
        unsigned slot = (bits * 2654435761u) >> (32 - shift);
        for (int probe = 0; probe < T; ++probe) {
            const unsigned prev = atomicCAS(&keys[slot], kEmpty, bits);
            if (prev == kEmpty) break;                                  // first occurrence
            if (prev == bits) { atomicOr(&twice[slot >> 5], 1u << (slot & 31)); break; }
            slot = (slot + 1) & (unsigned)(T - 1);
        }
    }
    __syncthreads();
    int uniq = 0, inter = 0;
    for (int i = threadIdx.x; i < T; i += 256) {
        const unsigned k = keys[i];
        if (k != kEmpty) {
            ++uniq;
            if (k != 0u && ((twice[i >> 5] >> (i & 31)) & 1u)) ++inter;
        }
    }
    uniq = wave_sum_i(uniq);
    inter = wave_sum_i(inter);
    const int w = threadIdx.x >> 6;
    if ((threadIdx.x & 63) == 0) { red[0][w] = uniq; red[1][w] = inter; }
    __syncthreads();
    if (threadIdx.x == 0) {
        int u = 0, sI = 0;
        for (int i = 0; i < 4; ++i) { u += red[0][i]; sI += red[1][i]; }
        const int uni = u - 1;                                          // minus the zero key
        out[n] = uni > 0 ? (float)((double)sI / (double)uni) : NAN;
    }
}


struct TailArgs {
    StepScalarArgs sc; int has_scal;
    EnqueueArgs enq;
    IouArgs iou;
};

__global__ __launch_bounds__(256) void step_tail_kernel(TailArgs a) {
    extern __shared__ __attribute__((aligned(16))) unsigned tail_lds[];
    int b = (int)blockIdx.x;
    if (b < a.has_scal) { step_scalars_body(a.sc); return; }
    b -= a.has_scal;
    if (b < a.enq.bx * a.enq.by) { enqueue_body(a.enq, b); return; }
    b -= a.enq.bx * a.enq.by;
    corr_iou_hash_body(a.iou, b, tail_lds);
}

static int tail_fill_scalars(TailArgs& t, const float* ins_loss, const int32_t* cnt_gt, const float* extras, int NE,
                             const float* sample_scal, const float* q_pos, const float* k_pos, const float* dense_pos_quart,
                             const float* dense_neg_quart, const float* ins_neg_quart, const float* lneg_mean, float lmbd_dense,
                             float* out, int B, int C) {
    if (!ins_loss || !cnt_gt || !extras || !sample_scal || !q_pos || !k_pos || !out) return CP2_ERR_NULL;
    if (B <= 0 || C <= 0 || NE <= 0) return CP2_ERR_SHAPE;
    t.sc = StepScalarArgs{ins_loss, cnt_gt, extras, NE, sample_scal, q_pos, k_pos, {dense_pos_quart, dense_neg_quart, ins_neg_quart},
                          lneg_mean, lmbd_dense, B, C, out};
    t.has_scal = 1;
    return CP2_OK;
}

static int tail_fill_enqueue(TailArgs& t, float* queue, const float* keys, int64_t* ptr, int32_t* ticket, int n, int C, int K) {
    if (!queue || !keys || !ptr || !ticket) return CP2_ERR_NULL;
    if (n <= 0 || C <= 0 || K <= 0) return CP2_ERR_SHAPE;
    if (n > K) return CP2_ERR_UNSUPPORTED;
    t.enq = EnqueueArgs{queue, keys, ptr, ticket, n, C, K, cp2_cdiv(n, 32), cp2_cdiv(C, 32)};
    return CP2_OK;
}

// hash-count form only: T = the power of two from 2 (2P + 1) - 2 up (load factor <= 1/2 + 1 slot: P = 4096 gives 16384 slots,
// 66 KB with the bitmap, beside the kernel's 37 KB of static LDS), 0 = unsupported (more than 4096 cells per map)
int cp2_tail_iou_table(int P) {
    int T = 128;
    while (T < 2 * (2 * P + 1) - 2) T <<= 1;
    return T <= 16384 ? T : 0;
}

static int tail_fill_iou(TailArgs& t, size_t* lds, const int64_t* ids_a, const int64_t* ids_b, const float* mask_a, const float* mask_b,
                         float* iou, float* iou_masked, int B, int P, int H, int W, int stride, int Ws) {
    if (!ids_a || !ids_b) return CP2_ERR_NULL;
    if (!iou && !iou_masked) return CP2_ERR_NULL;
    if (iou_masked && (!mask_a || !mask_b)) return CP2_ERR_NULL;
    if (B <= 0 || P <= 0) return CP2_ERR_SHAPE;
    const int T = cp2_tail_iou_table(P);
    if (T == 0) return CP2_ERR_UNSUPPORTED;
    t.iou = IouArgs{ids_a, ids_b, mask_a, mask_b, iou, iou_masked, P, T, H, W, stride, Ws, B};
    *lds = (size_t)(T + T / 32) * sizeof(unsigned);
    return CP2_OK;
}

static int tail_launch(const TailArgs& t, size_t lds, void* stream) {
    const int blocks = t.has_scal + t.enq.bx * t.enq.by + 2 * t.iou.B;
    if (blocks <= 0) return CP2_ERR_SHAPE;
    if (lds > 16 * 1024) {       // (the kernel's static LDS is 37 KB: beyond 48 KB in total the limit has to be raised)
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(step_tail_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return (int)e;
    }
    hipLaunchKernelGGL(step_tail_kernel, dim3(blocks), dim3(256), lds, cp2_stream(stream), t);
    return cp2_launch_status();
}

CP2_API int cp2_step_scalars(const float* ins_loss, const int32_t* cnt_gt, const float* extras, int NE, const float* sample_scal,
                             const float* q_pos, const float* k_pos, const float* dense_pos_quart, const float* dense_neg_quart,
                             const float* ins_neg_quart, const float* lneg_mean, float lmbd_dense, float* out, int B, int C,
                             void* stream) {
    TailArgs t{};
    int rc = tail_fill_scalars(t, ins_loss, cnt_gt, extras, NE, sample_scal, q_pos, k_pos, dense_pos_quart, dense_neg_quart,
                               ins_neg_quart, lneg_mean, lmbd_dense, out, B, C);
    return rc ? rc : tail_launch(t, 0, stream);
}

CP2_API int cp2_enqueue(float* queue, const float* keys, int64_t* ptr, int32_t* ticket, int n, int C, int K, void* stream) {
    TailArgs t{};
    int rc = tail_fill_enqueue(t, queue, keys, ptr, ticket, n, C, K);
    return rc ? rc : tail_launch(t, 0, stream);
}

// corr_iou.hip's launcher routes maps of at most 2047 cells here (larger ones: its bitonic-sort kernel)
int cp2_tail_iou_launch(const int64_t* ids_a, const int64_t* ids_b, const float* mask_a, const float* mask_b, float* iou,
                        float* iou_masked, int B, int P, int H, int W, int stride, int Ws, void* stream) {
    TailArgs t{};
    size_t lds = 0;
    int rc = tail_fill_iou(t, &lds, ids_a, ids_b, mask_a, mask_b, iou, iou_masked, B, P, H, W, stride, Ws);
    return rc ? rc : tail_launch(t, lds, stream);
}

CP2_API int cp2_step_tail(const float* ins_loss, const int32_t* cnt_gt, const float* extras, int NE, const float* sample_scal,
                          const float* q_pos, const float* k_pos, const float* dense_pos_quart, const float* dense_neg_quart,
                          const float* ins_neg_quart, const float* lneg_mean, float lmbd_dense, float* out, int B, int C,
                          float* queue, const float* keys, int64_t* queue_ptr, int32_t* ticket, int n_keys, int K,
                          const int64_t* ids_a, const int64_t* ids_b, const float* mask_a, const float* mask_b, float* iou,
                          float* iou_masked, int H, int W, int stride, void* stream) {
    TailArgs t{};
    size_t lds = 0;
    int rc = tail_fill_scalars(t, ins_loss, cnt_gt, extras, NE, sample_scal, q_pos, k_pos, dense_pos_quart, dense_neg_quart,
                               ins_neg_quart, lneg_mean, lmbd_dense, out, B, C);
    if (rc) return rc;
    if (queue) {
        rc = tail_fill_enqueue(t, queue, keys, queue_ptr, ticket, n_keys, C, K);
        if (rc) return rc;
    }
    if (ids_a) {
        if (H <= 0 || W <= 0 || stride <= 0) return CP2_ERR_SHAPE;
        const int off = stride / 2, Hs = (H - off + stride - 1) / stride, Ws = (W - off + stride - 1) / stride;
        if (Hs <= 0 || Ws <= 0) return CP2_ERR_SHAPE;
        rc = tail_fill_iou(t, &lds, ids_a, ids_b, mask_a, mask_b, iou, iou_masked, B, Hs * Ws, H, W, stride, Ws);
        if (rc) return rc;
    }
    return tail_launch(t, lds, stream);
}
