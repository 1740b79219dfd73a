// a15 (batch level) + the loss combination: every scalar a CP2 step returns or logs, in ONE launch.
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

__global__ __launch_bounds__(256) void step_scalars_kernel(StepScalarArgs a) {
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

CP2_API int cp2_step_scalars(const float* ins_loss, const int32_t* cnt_gt, const float* extras, int NE, const float* sample_scal,
                             const float* q_pos, const float* k_pos, const float* dense_pos_quart, const float* dense_neg_quart,
                             const float* ins_neg_quart, const float* lneg_mean, float lmbd_dense, float* out, int B, int C,
                             void* stream) {
    if (!ins_loss || !cnt_gt || !extras || !sample_scal || !q_pos || !k_pos || !out) return CP2_ERR_NULL;
    if (B <= 0 || C <= 0 || NE <= 0) return CP2_ERR_SHAPE;
    StepScalarArgs a{ins_loss, cnt_gt, extras, NE, sample_scal, q_pos, k_pos, {dense_pos_quart, dense_neg_quart, ins_neg_quart},
                     lneg_mean, lmbd_dense, B, C, out};
    hipLaunchKernelGGL(step_scalars_kernel, dim3(1), dim3(256), 0, cp2_stream(stream), a);
    return cp2_launch_status();
}
