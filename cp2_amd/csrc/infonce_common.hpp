// Shared pieces of the contrastive-logit kernels (infonce.hip, rowkey_small.hip, dense_stats.hip).
#pragma once
#include "common.hpp"
#include <math.h>

typedef float f32x16 __attribute__((ext_vector_type(16)));
constexpr int CH = 128;  // feature channels (MODEL dim, main.py:404-412)

// v_mfma_f32_32x32x2_f32 accumulator: register `reg` of lane half h holds row rho(reg, h) of the 32x32 tile
__device__ __forceinline__ int rho(int reg, int h) { return (reg & 3) + 8 * (reg >> 2) + 4 * h; }
__device__ __forceinline__ f32x16 mfma32(float a, float b, f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c, 0, 0, 0);
}

struct RowKeyArgs {
    const float* rows; int RP; int64_t r_sn, r_sx, r_sc; int R;  // row r -> (n=r/RP, x=r%RP), element (c,r) at n*r_sn + x*r_sx + c*r_sc
    const float* keys; int K;                                     // [CH][K]
    const float* extras; int NE; float inv_t;                     // raw extra logits [R][NE], column 0 = positive
    int keys_per_split;
    float* part_m; float* part_s; int* part_cnt; float* part_U;   // [S][R], [S][R], [S][R], [S][CH][R]
    float* lnegT; int64_t ln_sk, ln_sr;                            // optional raw logits, element (key, row) at key*ln_sk + row*ln_sr
};

// Merge of the per-split partials: lse, per-row loss, count of negatives above the positive, d loss / d row (in the
// rows' own layout) and d loss / d extra logit.
struct RowKeyFinArgs {
    const float* part_m; const float* part_s; const int* part_cnt; const float* part_U; int S;
    const float* extras; int NE; float inv_t; float grad_scale;
    int R; int RP; int64_t d_sn, d_sx, d_sc;
    float* lse; float* loss_rows; int* cnt_gt; float* drows; float* dE;
};

// dense (pixel-to-pixel) InfoNCE of one sample pair: infonce.hip
struct DenseArgs {
    const float* qd; const float* kd;            // [B][CH][P] unit vectors per pixel
    const float* mask_a; const float* mask_b;    // [B][P]
    const int64_t* pix_a; const int64_t* pix_b;  // [B][P] or NULL (all weights 1)
    const int64_t* reg_a; const int64_t* reg_b;
    float w_pixel, w_region, w_not, inv_t;
    int P;
    // forward outputs, per key pixel y: [B][P]
    float* lse; float* colsum_a; float* possum; float* allsum; float* colmax; int* argx;
    float* logits_out;                           // optional [B][P][P] raw logits (x-major), for the logging quantiles
    // backward
    const float* sample_scal;                    // [B][8]: Sa, Sb, ...
    float grad_scale; float* g_dense;            // [B][CH][P]
    // forward, split over the query pixels x: S > 1 workgroups share a (sample, key tile) and write partial column
    // statistics part[7][S][B*P] = (max, sum exp, colsum_a, possum, allsum, best value, best x); dense_post_body
    // folds them into the per-key outputs above
    int splits; float* part;
    // NegativeType reshaping of the negative pairs' raw logits (builder.py:1332-1386): L -> 2 / (1 + exp(-scale (L - centre))) - 1
    float neg_scale; const float* neg_center;    // centre per sample [B], or NULL = 0 (FIXED)
};

// Small-R form (R <= 32 rows, K % 4 == 0, 16-byte aligned keys): rowkey_small.hip
int rowkey_small_num_splits(int K, int* tiles_per_wg);
int rowkey_small_launch(const RowKeyArgs& a, int nsplit, bool with_u, hipStream_t stream);
int rowkey_small_finalize_launch(const RowKeyFinArgs& a, float* loss_mean, hipStream_t stream);

// Many-rows split-bf16 form with LDS-DMA double buffering (rowkey_bf16x3.hip); ksplit = the four arrays keys_split_kernel writes
int rowkey_bf16x3_dma_launch(const RowKeyArgs& a, const void* ksplit, dim3 grid, bool with_u, hipStream_t stream);
