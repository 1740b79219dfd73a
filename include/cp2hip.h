/*
 * cp2hip.h -- C ABI of libcp2hip.so: the MI355X (gfx950) kernels for the
 * copy-paste-contrastive (CP2) pre-training hot path.
 *
 * The reference (kimathikaai/CP2) is pure Python and has no FFI of its own; the
 * hot path is the tensor-op call sites inside builder.py `MODEL.forward_cp2` /
 * `forward_densecl` and tools/correlation_mapping.py.  Each entry point below
 * replaces one group of those call sites (cited as file:line, relative to the
 * reference root).  INTEGRATION.md shows the ctypes binding a maintainer of the
 * reference would add.
 *
 * Conventions (every function):
 *   - raw DEVICE pointers + explicit sizes; the caller owns all memory, the
 *     library allocates nothing, keeps no state (one documented exception:
 *     cp2_profile_next_launch) and never synchronises the host;
 *   - `stream` is a hipStream_t passed as void* (NULL = the null stream); every
 *     kernel is enqueued on it, so calls are hipGraph-capturable;
 *   - return value: 0 = ok, <0 = argument error (CP2_ERR_*), >0 = hipError_t of
 *     the failed launch;
 *   - fp32 arithmetic; integer / mask / index results are bit-exact with the
 *     reference, fp32 results agree to 1e-4 or better (tests/ state each bound);
 *   - tensors are dense row-major in the shape given unless strides are passed.
 */
#ifndef CP2HIP_H
#define CP2HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define CP2_OK 0
#define CP2_ERR_NULL (-1)        /* a required pointer is NULL            */
#define CP2_ERR_SHAPE (-2)       /* a size is zero/negative or inconsistent */
#define CP2_ERR_UNSUPPORTED (-3) /* size outside what the kernel supports  */
#define CP2_ERR_ALIGN (-4)       /* pointer not aligned as required        */

/* Library version (major*10000 + minor*100 + patch). */
int cp2_version(void);
/* Human-readable text for a return code of this library. */
const char* cp2_error_string(int code);

/* Measurement aid: the next launch of the dominant kernel of cp2_compose_pair, cp2_rowkey_infonce_fwd,
 * cp2_dense_infonce_fwd / _bwd, cp2_masked_quantiles(_multi) or cp2_sgd_flat made by the calling thread carries these caller-owned hipEvent_t
 * (start, stop), so hipEventElapsedTime gives that kernel's own duration.  One shot; NULL, NULL disarms.
 * The ONE exception to "the library keeps no state": a thread-local pair of event handles, consumed (and cleared) by the
 * next profiled launch of the same thread.  bench.py only; nothing is armed in normal operation. */
int cp2_profile_next_launch(void* start_event, void* stop_event);

/* ---- a1 (+ mask part of a2): copy-paste composition ---- builder.py:1146-1159
 * mask = (bg[:,0] == 0) ? 1 : 0;  out_img = img * mask + bg   (bit-exact, no FMA)
 * img, bg, out_img: [B,3,H,W]; mask_full: [B,H,W] or NULL;
 * mask_ds: [B,Hs,Ws] = mask[:, s/2::s, s/2::s] or NULL (Hs = ceil((H-s/2)/s)). */
int cp2_compose_mask(const float* img, const float* bg, float* out_img, float* mask_full,
                     float* mask_ds, int B, int H, int W, int stride, void* stream);

/* Both views in one launch (the training step's form), with three later launches folded in: the key view's rows may
 * be written in shuffle-BN order (row_b: device int64 [B] or NULL; out_b[j] = compose(img_b[row_b[j]], bg1[row_b[j]]),
 * builder.py:609-630 -- mask_ds_b stays in the original row order), and the output may be channels-last and / or bf16
 * (round-to-nearest-even of the fp32 value; what autocast's cast in front of the stem convolution produces).
 * W % 4 == 0, 16-byte aligned pointers.  out_a / out_b: [B,3,H,W] logical, fp32 or bf16 (out_bf16), NCHW or NHWC memory. */
int cp2_compose_pair(const float* img_a, const float* bg0, const float* img_b, const float* bg1, void* out_a, void* out_b,
                     float* mask_ds_a, float* mask_ds_b, const int64_t* row_b, int B, int H, int W, int stride,
                     int channels_last, int out_bf16, void* stream);

/* ---- a2: centre-tap strided down-sample ---------------- builder.py:1155-1186, loader.py:39-43
 * y[b,i,j] = x[b, s/2 + s*i, s/2 + s*j];  x: [B,H,W], y: [B,Hs,Ws]. */
int cp2_strided_gather_f32(const float* x, float* y, int B, int H, int W, int stride, void* stream);
int cp2_strided_gather_i64(const int64_t* x, int64_t* y, int B, int H, int W, int stride, void* stream);

/* ---- a12: row gather used by shuffle-BN ----------------- builder.py:630,649
 * dst[r, :] = src[idx[r], :];  src: [n_src,row_elems] f32, idx: [rows] int64 (device).
 * An idx outside [0,n_src) leaves the row untouched and sets *err_flag (device int32, may be NULL). */
int cp2_gather_rows_f32(const float* src, const int64_t* idx, float* dst, int rows, int n_src,
                        int64_t row_elems, int32_t* err_flag, void* stream);

/* ---- a3-a5: IoU of two id maps ---------------------------- tools/correlation_mapping.py:103-138,173,231
 * Per sample n: keys = float32(id+1)*mask over [0 | ids_a | ids_b];
 *   union = #distinct keys - 1;  inter = #distinct non-zero keys seen >= 2 times;
 *   iou = float32(double(inter)/double(union))  (NaN when union == 0).
 * iou uses all-ones masks, iou_masked uses mask_a/mask_b.  Either output may be NULL.
 * ids: [B,P] int64; masks: [B,P] f32 (may be NULL when iou_masked is NULL).  P <= 16383. */
int cp2_corr_iou(const int64_t* ids_a, const int64_t* ids_b, const float* mask_a, const float* mask_b,
                 float* iou, float* iou_masked, int B, int P, void* stream);
/* Same, with the centre-tap down-sampling (a2) of the id maps folded in: ids are the full-resolution [B,H,W] maps, read
 * at (s/2 + s*i, s/2 + s*j); masks are already down-sampled, [B, Hs*Ws]. */
int cp2_corr_iou_strided(const int64_t* ids_a, const int64_t* ids_b, const float* mask_a, const float* mask_b,
                         float* iou, float* iou_masked, int B, int H, int W, int stride, void* stream);
/* (Up to 2047 cells per map the keys are counted in an LDS hash table -- two barriers; above that they are sorted by a
 * bitonic network.  Identical counts either way.) */

/* ---- a11: momentum (EMA) update of the key encoder -------- builder.py:557-567
 * k[i] = k[i]*m + q[i]*one_minus_m  (two rounded products, one rounded sum; no FMA).
 * Flat form: one contiguous span of n floats (16-byte aligned). */
int cp2_ema_flat(float* k, const float* q, int64_t n, float m, float one_minus_m, void* stream);
/* Same launch; start_event / stop_event are caller-owned hipEvent_t that bracket exactly this kernel
 * (for hipEventElapsedTime after the stream has been synchronised). */
int cp2_ema_flat_timed(float* k, const float* q, int64_t n, float m, float one_minus_m, void* start_event,
                       void* stop_event, void* stream);
/* EMA that also writes k_bf16[i] = bf16(k[i]) (round to nearest even) for the updated values: n bf16, 8-byte aligned.
 * Lets the key encoder consume bf16 weights without per-tensor cast kernels.  Events as in _timed, or NULL. */
int cp2_ema_flat_shadow(float* k, const float* q, void* k_bf16, int64_t n, float m, float one_minus_m,
                        void* start_event, void* stop_event, void* stream);
/* Multi-tensor form: device tables of n_tensors pointers/sizes, plus a device chunk
 * table built by the caller: chunk c covers elements [chunk_off[c], chunk_off[c]+chunk_len[c])
 * of tensor chunk_tensor[c].  One launch for the whole encoder. */
int cp2_ema_multi(float* const* k_ptrs, const float* const* q_ptrs, const int32_t* chunk_tensor,
                  const int64_t* chunk_off, const int32_t* chunk_len, int n_chunks,
                  float m, float one_minus_m, void* stream);

/* ---- a13: queue enqueue with wrap-around ------------------ builder.py:569-587
 * queue[c, (ptr+i) % K] = keys[i, c] for i < n, then *ptr = (ptr + n) % K.
 * queue: [C,K] f32; keys: [n,C] f32 (already gathered over ranks); ptr: device int64[1].
 * The pointer is read and advanced on the device: no host synchronisation.  n <= K.
 * ticket: device int32[1] owned by the caller, zero before the first call (the kernel leaves it zero): the
 * workgroup that finishes last advances the pointer, so the enqueue is a single launch. */
int cp2_enqueue(float* queue, const float* keys, int64_t* ptr, int32_t* ticket, int n, int C, int K, void* stream);

/* ---- a7: per-pixel L2 normalise + masked pooling ---------- builder.py:1261-1268,1279-1285
 * feat: encoder output addressed as feat[n*stride_n + c*stride_c + x*stride_p] (NCHW or channels-last),
 * mask: [B,P] 0/1.  dense[B,C,P] = feat / max(|feat[:,x]|, 1e-12); inv_norm[B,P];
 * pool_partial: workspace [B, ceil(P/64), 2, C] of per-tile masked / un-masked channel sums.  C must be 128. */
int cp2_feat_normalize_pool(const float* feat, int64_t stride_n, int64_t stride_c, int64_t stride_p,
                            const float* mask, float* dense, float* inv_norm, float* pool_partial,
                            int B, int C, int P, void* stream);
/* Reduce the tile sums of the query and key maps to the pooled unit vectors
 * q_pos,q_neg,k_pos,k_neg [B,C], the query pool norms q_norms [B,2], and the raw extra logits
 * extras[B,3] = {q_pos.k_pos, q_pos.q_neg, q_pos.k_neg}      (builder.py:1264-1268,1281-1285,1395,1416-1417) */
int cp2_pool_finalize(const float* q_partial, const float* k_partial, float* q_pos, float* q_neg,
                      float* q_norms, float* k_pos, float* k_neg, float* extras, int B, int C, int P, void* stream);
/* Backward of the pooled query vectors: drow_pos [B,C] = d loss/d q_pos through the queue logits,
 * dE [B,3] = d loss/d extras  ->  ds_pos, ds_neg [B,C] = d loss / d (masked / un-masked channel sums). */
int cp2_pool_bwd(const float* drow_pos, const float* dE, const float* q_pos, const float* q_neg,
                 const float* k_pos, const float* k_neg, const float* q_norms, int include_background,
                 float* ds_pos, float* ds_neg, int B, int C, void* stream);
/* Backward of the normalisation with the pooled gradients folded in; dfeat uses the strides of feat. */
int cp2_feat_bwd(const float* dense, const float* inv_norm, const float* mask, const float* g_dense,
                 const float* ds_pos, const float* ds_neg, float* dfeat, int64_t stride_n, int64_t stride_c,
                 int64_t stride_p, int B, int C, int P, void* stream);

/* The training step's forms of a7 (round 3).
 * cp2_feat_normalize_pool_pair: query AND key map in one launch (the two cp2_feat_normalize_pool calls of a step); the
 * key side reads sample n from row k_row[n] of the key encoder's output (k_row: device int64 [B] or NULL) -- the
 * un-shuffle gather of builder.py:649 folded in.  Channels-last maps are staged through LDS (16-byte coalesced reads).
 * cp2_feat_bwd_fused: cp2_feat_bwd with two launches folded in -- the dense gradient is the sum over s < S of
 * g_part[s * split_stride + ...] in that order (what cp2_dense_infonce_bwd leaves in split_ws when g_dense is NULL; S = 1:
 * g_part is the gradient itself), and d loss / d(pooled sums) is computed per workgroup from cp2_pool_bwd's inputs
 * (dE: [B,NE], NE = 1 without / 3 with the background logits). */
int cp2_feat_normalize_pool_pair(const float* q_feat, int64_t q_sn, int64_t q_sc, int64_t q_sp, const float* k_feat,
                                 int64_t k_sn, int64_t k_sc, int64_t k_sp, const int64_t* k_row, const float* mask_a,
                                 const float* mask_b, float* q_dense, float* k_dense, float* q_inv_norm, float* q_partial,
                                 float* k_partial, int B, int C, int P, void* stream);
int cp2_feat_bwd_fused(const float* dense, const float* inv_norm, const float* mask, const float* g_part, int S,
                       int64_t split_stride, const float* drow_pos, const float* dE, int NE, const float* q_pos, const float* q_neg,
                       const float* k_pos, const float* k_neg, const float* q_norms, int include_background, float* dfeat,
                       int64_t stride_n, int64_t stride_c, int64_t stride_p, int B, int C, int P, void* stream);

/* ---- a15 at batch level + the loss combination: every scalar a CP2 step returns or logs, one launch ------------------
 * builder.py:1431-1448 (loss, accuracies), :1265,1282 (cross-image spread), :1553-1604 (logged scalars).
 * out[CP2_STEP_SCALARS]: 0 loss = loss_instance + lmbd_dense * loss_dense, 1 loss_instance, 2 loss_dense, 3 top-1 %,
 * 4 top-5 %, 5 dense arg-max accuracy %, 6 / 7 mean positive / negative dense score, 8 mean raw positive instance logit,
 * 9 / 10 mean over channels of the unbiased batch std of q_pos / k_pos, 11-13 / 14-16 / 17-19 batch means of the
 * [3,B] quartile sets (positive dense, negative dense, queue logits; NULL -> 0), 20 batch mean of lneg_mean (NULL -> 0).
 * ins_loss: device scalar; cnt_gt: int32 [B]; extras: [B,NE] (column 0 = raw positive logit); sample_scal: [B,8]. */
#define CP2_STEP_SCALARS 24
int cp2_step_scalars(const float* ins_loss, const int32_t* cnt_gt, const float* extras, int NE, const float* sample_scal,
                     const float* q_pos, const float* k_pos, const float* dense_pos_quart, const float* dense_neg_quart,
                     const float* ins_neg_quart, const float* lneg_mean, float lmbd_dense, float* out, int B, int C,
                     void* stream);
/* The tail of the step's loss section as ONE launch: cp2_step_scalars (arguments as above) + cp2_enqueue of `keys`
 * [n_keys,C] into queue [C,K] (queue NULL: no enqueue) + cp2_corr_iou_strided of the full-resolution id maps [B,H,W]
 * with the down-sampled masks [B,Hs*Ws] (ids_a NULL: no IoUs; at most 2047 down-sampled cells per map).  The three parts
 * are independent workgroups of one kernel (none reads what another writes); each of the three single entry points
 * launches the same kernel with only its own part.  builder.py:1431-1448,1553-1604 + :569-587 + :1204-1219. */
int cp2_step_tail(const float* ins_loss, const int32_t* cnt_gt, const float* extras, int NE, const float* sample_scal,
                  const float* q_pos, const float* k_pos, const float* dense_pos_quart, const float* dense_neg_quart,
                  const float* ins_neg_quart, const float* lneg_mean, float lmbd_dense, float* out, int B, int C,
                  float* queue, const float* keys, int64_t* queue_ptr, int32_t* ticket, int n_keys, int K,
                  const int64_t* ids_a, const int64_t* ids_b, const float* mask_a, const float* mask_b, float* iou,
                  float* iou_masked, int H, int W, int stride, void* stream);

/* ---- a10 / a16: rows-vs-queue InfoNCE (f32 MFMA) ---------- builder.py:1395-1428 (instance),
 *                                                            :866-873,906-908,150-176 (DenseCL local)
 * logits of row r = [extras[r,:NE] | rows[r].keys[:,j], j<K] / T, target = extras column 0;
 * loss = mean_r (logsumexp_r - extras[r,0]/T).  The queue is read once: the forward pass also
 * accumulates sum_j softmax_j * keys[:,j], so the gradient needs no second pass over the queue.
 * rows: element (c, r) at (r/RP)*r_sn + (r%RP)*r_sx + c*r_sc;  keys: [C,K] (column = key);  C = 128.
 * nsplit = cp2_rowkey_num_splits(R,K) key ranges run in parallel; workspaces part_m, part_s [nsplit,R],
 * part_cnt [nsplit,R] int32, part_U [nsplit,C,R] (NULL: no gradient), lnegT: NULL, or the raw logits rows.keys as
 * [K,R] (lneg_row_major = 0) or [R,K] (lneg_row_major = 1, the layout cp2_masked_quantiles reads fastest).
 * precision 0: f32-input MFMA (exact fp32 fma chains, logits within ~1e-6 of the reference);
 * precision 1: split-bf16 (hi*hi + hi*lo + lo*hi on bf16 MFMA, fp32 accumulate; logits within 3e-5) -- used when
 *              R > 64 with keys_split given and K % 16 == 0, otherwise the f32 kernel runs regardless.  keys_split: a workspace
 *              of 4*C*K bf16 (16-byte aligned) that receives the hi/lo split of the queue in both layouts once per call:
 *              the main kernel fills LDS from it by LDS-DMA;
 * precision 3: as 1, but keys_split already holds the split of this queue (written by an earlier precision-1 call on the
 *              same stream): the prep launch is skipped -- for callers that walk the rows in several calls. */
int cp2_rowkey_num_splits(int R, int K);
int cp2_rowkey_infonce_fwd(const float* rows, int RP, int64_t r_sn, int64_t r_sx, int64_t r_sc, int R,
                           const float* keys, int K, const float* extras, int NE, float temperature,
                           int nsplit, float* part_m, float* part_s, int32_t* part_cnt, float* part_U,
                           float* lnegT, int lneg_row_major, int precision, void* keys_split, int C, void* stream);
/* Merge the splits: lse[R], loss_rows[R], cnt_gt[R] (# negatives whose logit exceeds the positive),
 * drows (same addressing as rows, with d_* strides; NULL: skip) = grad_scale * d sum_r loss_r / d rows,
 * dE [R,NE] likewise (may be NULL), loss_mean[1] = mean_r loss_r (may be NULL). */
int cp2_rowkey_infonce_finalize(const float* part_m, const float* part_s, const int32_t* part_cnt,
                                const float* part_U, int nsplit, const float* extras, int NE,
                                float temperature, float grad_scale, int R, int RP, int64_t d_sn,
                                int64_t d_sx, int64_t d_sc, float* lse, float* loss_rows, int32_t* cnt_gt,
                                float* drows, float* dE, float* loss_mean, int C, void* stream);

/* ---- a8 / a9 (+a6, part of a15): dense pixel-to-pixel InfoNCE (f32 MFMA) -- builder.py:1289-1292,
 *                                                 1225-1243 (weights), 1392, 1431-1437, 1442-1448
 * L[n,x,y] = q_dense[n,:,x].k_dense[n,:,y] * w[n,x,y] / T;  -log_softmax over the QUERY axis x;
 * loss_n = sum_{x,y} nll[x,y] mask_a[x] mask_b[y] / sum mask_a[x] mask_b[y].  The P x P logits are never
 * written to memory.  pix_*, reg_* (all four or none): int64 [B,P] ids for the correspondence weights
 * (pixel match -> w_pixel, else known-region match -> w_region, else w_not); NULL = all weights 1.
 * Per key pixel outputs [B,P]: lse, colsum_a, possum, allsum, colmax, argx (workspaces kept for backward /
 * logging).  sample_scal [B,8] = {Sa, Sb, loss_n, mean positive score, mean negative score, label at the
 * arg-max pair, 0, 0}; the batch means (loss, arg-max accuracy) are formed by cp2_step_scalars.
 * logits_out: NULL, or [B,P,P] to also receive the raw logits q.k (only for the logging quantiles).   C = 128.
 * split_ws: NULL (one workgroup per (sample, 128-key tile) walks all query pixels), or float[7 * S * B * P] with
 * S = cp2_dense_num_splits(B, P): S workgroups share the walk and a merge kernel folds their partial statistics, so
 * small B*P still fills the chip (the fold and the per-sample scalars are one launch, one workgroup per sample).
 * negative_mode != 0 (reference NegativeType FIXED / AVERAGE / MEDIAN, builder.py:1332-1386): the raw logit L of every
 * NEGATIVE pair (mask_a[x]*mask_b[y] == 0) enters the loss as 2 / (1 + exp(-negative_scale * (L - centre))) - 1 with
 * centre = negative_center[n] (device float[B]: the sample's mean or median negative score, taken from a first
 * un-reshaped pass) or 0 when negative_center is NULL (FIXED); the logging sums always use the raw scores.
 * sample_scal NULL: the fold of the splits and the per-sample scalars are DEFERRED -- the caller finishes the call with
 * cp2_loss_post (same output arrays, same split_ws), which does that work beside the instance loss's finalize. */
int cp2_dense_num_splits(int B, int P);
int cp2_dense_infonce_fwd(const float* q_dense, const float* k_dense, const float* mask_a, const float* mask_b,
                          const int64_t* pix_a, const int64_t* pix_b, const int64_t* reg_a, const int64_t* reg_b,
                          float w_pixel, float w_region, float w_not, float temperature, float* lse,
                          float* colsum_a, float* possum, float* allsum, float* colmax, int32_t* argx,
                          float* sample_scal, float* logits_out, float* split_ws,
                          int negative_mode, float negative_scale, const float* negative_center, int B, int C,
                          int P, void* stream);
/* One launch for two independent tails of the loss section (round 4): cp2_rowkey_infonce_finalize of the INSTANCE loss
 * (its small-row form only: R <= 32, nsplit >= 16; arguments part_m .. loss_mean exactly as there) and the post-pass of a
 * cp2_dense_infonce_fwd call that was given sample_scal = NULL (d_lse .. argx: that call's per-key-pixel arrays, split_ws:
 * that call's workspace or NULL, mask_a / mask_b, sample_scal [B,8] as documented there).  Results equal the two separate
 * launches' (bit for bit up to P = 256; beyond, the per-sample sums of the dense part are taken by 1024 instead of 256
 * threads, i.e. in another fixed order). */
int cp2_loss_post(const float* part_m, const float* part_s, const int32_t* part_cnt, const float* part_U, int nsplit,
                  const float* extras, int NE, float temperature, float grad_scale, int R, int RP, int64_t d_sn,
                  int64_t d_sx, int64_t d_sc, float* lse, float* loss_rows, int32_t* cnt_gt, float* drows, float* dE,
                  float* loss_mean, const float* mask_a, const float* mask_b, float* d_lse, float* colsum_a,
                  float* possum, float* allsum, float* colmax, int32_t* argx, float* sample_scal, float* split_ws,
                  int B, int C, int P, void* stream);
/* The same two tails riding in the launch of the step's quartile statistics (round 4: one launch instead of three):
 * arguments njobs .. mean_out exactly as cp2_masked_quantiles_multi (its one-launch row form only: every N <=
 * CP2_QUANTILES_ROW_MAX, else CP2_ERR_UNSUPPORTED -- use cp2_loss_post + cp2_masked_quantiles_multi), then part_m ..
 * loss_mean and d_mask_a .. dP exactly as cp2_loss_post (fR = its R, dP = its P).  The quartile jobs may read the logits the
 * two producer kernels wrote (lnegT of cp2_rowkey_infonce_fwd, logits_out of cp2_dense_infonce_fwd): none of the three
 * parts reads what another writes.  Quartiles bit-equal to cp2_masked_quantiles_multi, the tails to cp2_loss_post. */
int cp2_step_post(int njobs, const float* const* x, const int64_t* stride_row, const int64_t* stride_elem,
                  const int* R, const int* N, const float* const* mask_a, const float* const* mask_b,
                  const int* P, const int* want, const float* q, int NQ, float* const* out, float* const* mean_out,
                  const float* part_m, const float* part_s, const int32_t* part_cnt, const float* part_U, int nsplit,
                  const float* extras, int NE, float temperature, float grad_scale, int fR, int RP, int64_t d_sn,
                  int64_t d_sx, int64_t d_sc, float* lse, float* loss_rows, int32_t* cnt_gt, float* drows, float* dE,
                  float* loss_mean, const float* d_mask_a, const float* d_mask_b, float* d_lse, float* colsum_a,
                  float* possum, float* allsum, float* colmax, int32_t* argx, float* sample_scal, float* split_ws,
                  int B, int C, int dP, void* stream);
/* g_dense [B,C,P] = grad_scale * d (sum_n loss_n) / d q_dense, recomputing the logits tile by tile.
 * split_ws: NULL (g_dense receives the gradient), or float[S * B * C * P] with S = cp2_dense_num_splits(B, P): the
 * key-pixel range is shared by S workgroups; with S > 1 g_dense must be NULL and the S partial gradients [S][B][C][P]
 * stay in split_ws for cp2_feat_bwd_fused to add in split order (deterministic); S == 1: g_dense as without split_ws. */
int cp2_dense_infonce_bwd(const float* q_dense, const float* k_dense, const float* mask_a, const float* mask_b,
                          const int64_t* pix_a, const int64_t* pix_b, const int64_t* reg_a, const int64_t* reg_b,
                          float w_pixel, float w_region, float w_not, float temperature, const float* lse,
                          const float* sample_scal, float grad_scale, float* g_dense, float* split_ws,
                          int negative_mode, float negative_scale, const float* negative_center, int B, int C,
                          int P, void* stream);

/* ---- a16: DenseCL positive selection (T18) ---------------- builder.py:818-864
 * Replaces: einsum("ncx,ncy->nxy", q_embed, k_embed).max(dim=2)[1] (:818-821), the local-similarity einsum and its
 * gather at that index (:824-835), get_correlation_map + the coordinate mix (:839-855) and the matching-positives rate
 * (:857-864) -- three b x S^2 x S^2 maps in the reference, none here.
 *   best_idx[n,x] = argmax_y <q_embed[n,:,x], k_embed[n,:,y]> (first maximum); with normalize_k the key vector is divided
 *                   by max(|k_embed[n,:,y]|, 1e-12) first, i.e. the inputs may be the RAW backbone features (the query
 *                   norm cannot change the arg-max); normalize_k = 0: the key vectors are unit vectors already;
 *   pos[n,x]      = <q_local[n,:,x], k_local[n,:,best]>; where id_q[n,x] occurs among id_k[n,:] and lmbd_coordinate > 0:
 *                   pos * one_minus_lmbd + lmbd_coordinate * sum_{y: id_k[n,y] == id_q[n,x]} <q_local[n,:,x], k_local[n,:,y]>;
 *   kvec[n,:,x]   = d pos[n,x] / d q_local[n,:,x] (NULL: not wanted);
 *   counts[2*i + {0,1}], i < B * ceil(P/32) (want_metrics; NULL otherwise): per workgroup, the number of query pixels
 *                   with an id match, and of those whose arg-max of the LOCAL similarity is their first id match.
 * q_embed / k_embed: element (n, c, p) at n*sn + c*sc + p*sp; embed_bf16 != 0: bf16, channels-last (sc == 1), CE % 64 == 0,
 * 16-byte aligned rows (bf16 MFMA: exact products, fp32 accumulation); else fp32 with any strides (f32-input MFMA).
 * k_row: NULL, or int64 [B]: sample n's key side is row k_row[n] of k_embed / k_local (the shuffle-BN un-shuffle index,
 * builder.py:649, without a gather launch); ids stay in sample order.  q_local / k_local: fp32 [B,CL,P] unit vectors,
 * CL = 128; ids_q / ids_k: int64 [B,P] or both NULL; P <= 4096. */
int cp2_densecl_match(const void* q_embed, const void* k_embed, int embed_bf16, int64_t qe_sn, int64_t qe_sc, int64_t qe_sp,
                      int64_t ke_sn, int64_t ke_sc, int64_t ke_sp, const int64_t* k_row, const float* q_local,
                      const float* k_local, const int64_t* ids_q, const int64_t* ids_k, float lmbd_coordinate,
                      float one_minus_lmbd, int normalize_k, int want_metrics, int32_t* best_idx, float* pos, float* kvec,
                      int32_t* counts, int B, int CE, int CL, int P, void* stream);

/* ---- a15: logging quantiles without a sort ---------------- tools/correlation_mapping.py:16-53, builder.py:1399-1406
 * out[j, r] = torch.nanquantile(kept elements of row r, q[j]) with linear interpolation (exact order statistics by
 * radix select).  Element i of row r is x[r*stride_row + i*stride_elem], i < N.  want < 0: every element is kept;
 * want = 1 / 0: N = P*P and element i = x_pix*P + y_pix is kept iff (mask_a[r,x_pix]*mask_b[r,y_pix] != 0) == want
 * (the reference's positive / negative dense scores).  q: device float[NQ]; out: [NQ,R].  NaN when nothing is kept. */
int cp2_masked_quantiles(const float* x, int64_t stride_row, int64_t stride_elem, int R, int N,
                         const float* mask_a, const float* mask_b, int P, int want, const float* q, int NQ,
                         float* out, void* workspace, int64_t workspace_bytes, void* stream);
/* Up to 4 such problems in one call (NQ <= 4 quantiles q shared by all jobs).  Every argument of cp2_masked_quantiles
 * becomes a HOST array of njobs entries (device pointers inside).  Rows longer than CP2_QUANTILES_ROW_MAX are cut into
 * 8192-element chunks: three chunk-parallel histogram passes (12 + 10 + 10 bits) with a per-row select after each.
 * When every job has N <= CP2_QUANTILES_ROW_MAX the call is ONE launch (one workgroup per row through all three levels,
 * the row held in registers) and needs no workspace (NULL, 0).  Rows longer than CP2_QUANTILES_ROW_MAX take the chunked
 * six-launch path; there two consecutive jobs over the SAME logits with want = 1 and want = 0 (the reference's positive
 * and negative dense scores) are served by ONE read of the logits per level.  That path needs:
 * workspace: device memory of cp2_quantiles_workspace_bytes(njobs, R, N, NQ) bytes, 16-byte aligned, ZERO before the
 * first call; every call leaves it zero again, so the same buffer serves every later call with the SAME (R, N, NQ).
 * mean_out: NULL, or a HOST array of njobs DEVICE pointers (NULL entries allowed): job j's row means [R] as torch's
 * x.mean(1) (NaN when the row holds a NaN), from the first pass of the one-launch form (want < 0 jobs only). */
#define CP2_QUANTILES_ROW_MAX 131072
int64_t cp2_quantiles_workspace_bytes(int njobs, const int* R, const int* N, int NQ);
int cp2_masked_quantiles_multi(int njobs, const float* const* x, const int64_t* stride_row, const int64_t* stride_elem,
                               const int* R, const int* N, const float* const* mask_a, const float* const* mask_b,
                               const int* P, const int* want, const float* q, int NQ, float* const* out,
                               float* const* mean_out, void* workspace, int64_t workspace_bytes, void* stream);

/* ---- f1: on-device two-crop augmentation + background erasing --- loader.py:39-43,50-118; main.py:204-225
 * One launch makes B output samples from a dataset resident in device memory: src [N,3,Hs,Ws] fp32 in [0,1]
 * (src_is_u8 = 0) or uint8 (src_is_u8 = 1, scaled by 1/255 like ToTensor); src_region [N,Hs,Ws] int64 or NULL.
 * params: device int32 [B,8] = {source index, crop top, left, height, width, flip, 0, 0} (RandomResizedCrop +
 * HorizontalFlip parameters drawn by the caller).  out_img [B,3,H,W]: bilinear, half-pixel centres, replicated edges;
 * out_pix / out_reg [B,H,W] int64 (either may be NULL): nearest neighbour, source cell floor(dst*crop/out); pixel id of
 * source cell (y,x) = y*Ws + x + 1 at id_stride 1, else the id of the centre tap of its stride block mapped back by
 * INTER_NEAREST_EXACT (loader.py:66-73); region id = src_region read at that same cell (loader.py:75-83), or the pixel id
 * when src_region is NULL.  out_img may be NULL when out_rgbx is given: out_rgbx [B,H,W] receives the view as uint8
 * (R | G<<8 | B<<16 per pixel, value*255 rounded half up) for the photometric stages below. */
int cp2_crop_resize_flip(const void* src, int src_is_u8, const int64_t* src_region, int N, int Hs, int Ws,
                         const int32_t* params, float* out_img, int64_t* out_pix, int64_t* out_reg, int B, int H,
                         int W, int id_stride, uint32_t* out_rgbx, void* stream);
/* RandomErasing(p=1, value=0) of the background view (main.py:218-224): img[b,:,top:top+h,left:left+w] = 0 exactly;
 * rects: device int32 [B,4] = {top, left, h, w} (h = 0 skips the sample).  In place. */
int cp2_erase_rect(float* img, const int32_t* rects, int B, int H, int W, void* stream);

/* ---- f1, photometric half: the image-value transforms of the reference's loaders in Pillow's arithmetic --------------
 * main.py:204-225 (background views, torchvision on PIL images), loader.py:121-152 (GaussianBlur of both kinds of view).
 * Working format: uint32 per pixel = R | G<<8 | B<<16, [B,H,W].  Host-drawn parameters in device tables.
 *
 * cp2_pil_resize_crop: RandomResizedCrop + HorizontalFlip of a background view = img.crop(box).resize((W,H), BILINEAR)
 * (Pillow Resample.c: separable, antialiased when shrinking, 22-bit integer taps, uint8 after the horizontal pass).
 * src: uint8 [N,3,Hs,Ws]; params: int32 [B,8] as cp2_crop_resize_flip; workspace: int32 scratch of
 * cp2_pil_resize_workspace_bytes(B,Hs,Ws,H,W) bytes (coefficient tables, written by a first launch). */
int cp2_pil_resize_ksize(int Hs, int Ws, int H, int W);
int64_t cp2_pil_resize_workspace_bytes(int B, int Hs, int Ws, int H, int W);
int cp2_pil_resize_crop(const unsigned char* src, int N, int Hs, int Ws, const int32_t* params, uint32_t* out_rgbx,
                        int B, int H, int W, int32_t* workspace, int64_t workspace_bytes, void* stream);
/* cp2_color_ops: ColorJitter + RandomGrayscale in place (Pillow ImageEnhance / Blend.c / Convert.c arithmetic).
 * params: int32 [B,CP2_COLOR_PARAMS]: [0..3] adjustments in application order (0 brightness, 1 contrast, 2 saturation,
 * 3 hue, -1 none), [4..6] float bits of the brightness / contrast / saturation factors, [7] uint8(hue_factor*255),
 * [8] grayscale flag, [9] arithmetic: 0 = torchvision on PIL images (the reference's background views), 1 = albumentations
 * on cv2 (its foreground views, main.py:236-237: float64 look-up tables truncated to uint8, cv2's 8-bit RGB2GRAY / RGB2HSV /
 * HSV2RGB, addWeighted; restated from the published sources), [10] float bits of the hue factor (arithmetic 1).
 * lsum: uint64 [B] scratch (zeroed by the call; receives the gray sum the contrast step needs). */
#define CP2_COLOR_PARAMS 12
int cp2_color_ops(uint32_t* img_rgbx, const int32_t* params, uint64_t* lsum, int B, int H, int W, void* stream);
/* cp2_blur_to_tensor: ImageFilter.GaussianBlur(sigma) (Pillow BoxBlur.c: three box passes per axis, 24-bit weights,
 * replicated edges) + ToTensor (uint8/255 -> CHW float) + RandomErasing(value=0) in one pass.
 * params: int32 [B,4] = {blur on, integer box radius, ww, fw}; rects: int32 [B,4] = {top,left,h,w} or NULL;
 * out: float [B,3,H,W]; rmax: the largest integer radius in params (<= CP2_BLUR_MAX_RADIUS; sizes the LDS halo). */
#define CP2_BLUR_MAX_RADIUS 4
int cp2_blur_to_tensor(const uint32_t* img_rgbx, const int32_t* params, const int32_t* rects, float* out, int B, int H,
                       int W, int rmax, void* stream);

/* ---- optimizer step of the query encoder on the flat parameter buffer ---------------- main.py:467-477, :640-642
 * torch.optim.SGD(momentum, weight_decay) (dampening 0, no Nesterov), bit-identical to torch's default multi-tensor
 * implementation on the same GPU:  g' = g + wd*p;  buf = buf*momentum + g';  p = p - lr*buf.
 * p, momentum_buf: flat fp32 device buffers (16-byte aligned) holding every parameter in a slot; p_bf16: NULL or a
 * bf16 buffer with the same element offsets that receives the new weights (the query encoder's convolutions read it
 * under bf16 autocast).  grads: HOST array of ntensors DEVICE pointers (fp32, element order of the parameter's slot;
 * NULL = no gradient, parameter untouched).  blk_tab: DEVICE int32[nblocks][4] = {tensor, flat offset, gradient
 * offset, valid floats (<= 512)} -- workgroup b updates floats [flat offset, +valid) of tensor blk_tab[b][0];
 * tensor_first_block: HOST int32[ntensors+1], the blocks of tensor t are [first[t], first[t+1]).
 * lr_dev: NULL, or a device float that overrides lr (graph capture).  momentum_buf must start as zeros. */
#define CP2_SGD_MAX_TENSORS 384
int cp2_sgd_flat(float* p, float* momentum_buf, void* p_bf16, const void* const* grads, int ntensors,
                 const int32_t* blk_tab, const int32_t* tensor_first_block, float lr, const float* lr_dev,
                 float momentum, float weight_decay, void* stream);
/* Gradient averaging over the ranks, the local half (reference main.py:456-460: DistributedDataParallel; its reducer
 * copies each parameter's gradient into a bucket, one launch per parameter).  Writes scale * grads[t] into tensor t's slot
 * of flat_grad (the layout of the flat parameter buffer, tables as cp2_sgd_flat) for t in [t_begin, t_end) in one launch;
 * grads[t] == NULL zeroes the slot.  scale = 1 / world size: the flat range is then summed over the ranks in place. */
int cp2_pack_grads(float* flat_grad, const void* const* grads, int t_begin, int t_end, const int32_t* blk_tab,
                   const int32_t* tensor_first_block, float scale, void* stream);
/* dst[i] = bf16(src[i]) (round to nearest even) for a flat buffer, n % 4 == 0: rebuilds the bf16 weight image. */
int cp2_bf16_image(const float* src, void* dst, int64_t n, void* stream);

/* ---- encoder fast path (not a reference call site): fused training-mode BatchNorm (+ residual) (+ ReLU) --------
 * for channels-last bf16 activations, replacing MIOpen's 3 BN kernels + ATen add + clamp (forward) and 3 BN kernels +
 * threshold_backward (backward) of mmseg's ResNet blocks (mmseg_/models/backbones/resnet.py:267-304).
 * x, residual, y, dy, dx, dres: [M = N*H*W, C] bf16 (device pointers, 16-byte aligned); C % 64 == 0, C <= 8192.
 * forward : y = relu?(x*scale + shift + residual?), batch statistics in fp32; running_mean / running_var (may be NULL)
 *           are updated as torch.nn.BatchNorm2d does (momentum, unbiased variance); save_mean, save_invstd: [C] out.
 *           Workspaces: part float[cp2_bn_num_partials(M,C), 2, C] (16-byte aligned), scale_shift float[2, C].
 * backward: g = dy * (y > 0) when relu (y = forward output, else NULL);  dx = gamma*invstd*(g - mean(g) - xhat*mean(g*xhat));
 *           dres (NULL or [M,C]) = g;  dgamma, dbeta: [C] fp32 (may be NULL).  Workspaces: part as above,
 *           coef float[3, C]. */
int cp2_bn_num_partials(int M, int C);
int cp2_bn_fwd(const void* x, const void* residual, const float* weight, const float* bias, float* running_mean,
               float* running_var, float momentum, float eps, int relu, void* y, float* save_mean, float* save_invstd,
               float* part, float* scale_shift, int M, int C, void* stream);
int cp2_bn_bwd(const void* x, const void* dy, const void* y, const float* weight, const float* save_mean,
               const float* save_invstd, int relu, void* dx, void* dres, float* dgamma, float* dbeta, float* part,
               float* coef, int M, int C, void* stream);


/* ---- encoder fast path (not a reference call site): weight gradient of a 1x1 stride-1 convolution ---------------
 * dw[co][ci] = sum_m dy[m][co] * x[m][ci];  dy: [M, CO] bf16, x: [M, CI] bf16 (channels-last activations viewed as
 * matrices, 16-byte aligned), dw: [CO, CI] fp32;  CO % 64 == 0, CI % 64 == 0.  part: float[S * CO * CI] with
 * S = cp2_wgrad1x1_num_splits(M, CO, CI); partial tiles are added in split order (deterministic, no atomics). */
int cp2_wgrad1x1_num_splits(int M, int CO, int CI);
int cp2_wgrad1x1(const void* dy, const void* x, float* dw, float* part, int M, int CO, int CI, void* stream);
/* The same for a KH x KW convolution (the 3x3 convolutions of the bottlenecks and of the FCN head; groups = 1, equal
 * stride / padding / dilation in both directions): dy [N,OH,OW,CO], x [N,H,W,CI] bf16 channels-last, dw fp32
 * [CO][KH][KW][CI] (the memory order of a channels-last weight), OH = (H + 2 pad - dil (KH - 1) - 1) / stride + 1.
 * part: float[S * CO * KH * KW * CI], S = cp2_wgrad_conv_num_splits(N, OH, OW, CO, CI, KH, KW).  Deterministic. */
int cp2_wgrad_conv_num_splits(int N, int OH, int OW, int CO, int CI, int KH, int KW);
int cp2_wgrad_conv(const void* dy, const void* x, float* dw, float* part, int N, int H, int W, int OH, int OW, int CO,
                   int CI, int KH, int KW, int stride, int pad, int dil, void* stream);

/* ---- encoder fast path (not a reference call site): the ResNet stem's MaxPool2d(3, stride 2, padding 1) ----------
 * (mmseg_/models/backbones/resnet.py:413) for channels-last bf16 activations: x [N,H,W,C], y [N,OH,OW,C] with
 * OH = (H - 1) / 2 + 1 (same for W), C % 8 == 0; idx: one byte per output element = position 0..8 of the maximum in
 * its window (first maximum in row-major order, NaN wins -- torch's rule), N*OH*OW*C bytes, 8-byte aligned.
 * Backward gathers: dx [N,H,W,C] = sum of dy over the (at most four) windows whose maximum is that pixel. */
int cp2_maxpool3s2_fwd(const void* x, void* y, void* idx, int N, int H, int W, int C, void* stream);
int cp2_maxpool3s2_bwd(const void* dy, const void* idx, void* dx, int N, int H, int W, int C, void* stream);

/* ---- supervised CutPaste / "mirror" pre-training ------------------------------------------------
 * cp2_cutpaste replaces the per-sample numpy / Pillow composition of datasets/pretrain_dataset.py:273-352 (cutpaste)
 * and :357-412 (__getitem__: additional patches, ToTensor).  One launch = one patch round for a whole batch:
 *   src / src_mirror: uint8 [*, H, W, 3] device arrays (the resident dataset in round 1, the previous round's dst /
 *   dst_mirror afterwards; src_mirror NULL = MirrorVariant.NONE); params: device int32 [B, CP2_CUTPASTE_PARAMS] =
 *   {src index, mirror index, class (0 = copy the sample through), patch x, patch y, patch w, patch h, paste x,
 *    paste y, rotated w, rotated h, a0..a5 (Pillow's 16.16 fixed-point reverse matrix: source pixel of rotated-patch
 *    pixel (u, v) = ((a2 + u*a0 + v*a1) >> 16, (a5 + u*a3 + v*a4) >> 16); pixels that fall outside the patch are not
 *    pasted), 0, 0, 0};
 *   dst / dst_mirror: uint8 [B, H, W, 3] or NULL; dst_f32 / dst_mirror_f32: float [B, 3, H, W] = ToTensor (u8 / 255)
 *   or NULL; mask: int64 [B, H, W], written with class inside the pasted shape and 0 elsewhere, or, with mask_or != 0,
 *   logical_or(old mask, pasted shape) as 0 / 1 (class-0 samples keep their old mask).  dst must not alias src. */
#define CP2_CUTPASTE_PARAMS 20
int cp2_cutpaste(const unsigned char* src, const unsigned char* src_mirror, const int32_t* params, unsigned char* dst,
                 unsigned char* dst_mirror, float* dst_f32, float* dst_mirror_f32, int64_t* mask, int mask_or, int B,
                 int H, int W, void* stream);
/* cp2_mirror_loss replaces networks/mirror_network.py:40-63 (MirrorModule.shared_step) after the resize of
 * networks/segment_network.py:220-231:  class_loss = cross_entropy(cat(s, t), cat(masks, masks)) (mean over all
 * pixels), compare_loss = cross_entropy(softmax(s / T, 1), softmax(t / T, 1)) with probability targets (mean over
 * N*HW), loss = class_loss + lmbd * compare_loss.  s_logits, t_logits: float [N, C, HW] (t_logits NULL =
 * MirrorVariant.NONE: class loss of s alone, compare_loss = 0); masks: int64 [N, HW] in [0, C); 2 <= C <=
 * CP2_MIRROR_MAX_CLASSES.  Outputs: out3 float[3] = {loss, class_loss, compare_loss}; grad_s / grad_t: d loss / d
 * logits (NULL = forward only; both or neither when t_logits is given -- the compare loss differentiates through its
 * target as autograd does); argmax: int64 [(t ? 2 : 1) * N, HW] or NULL; confusion: int64 [C, C] (row = ground truth,
 * column = prediction), ADDED to -- the caller zeroes it -- or NULL.  Workspace: partial double[2 *
 * cp2_mirror_loss_num_partials(N, HW)]; the partial sums are added in a fixed order (deterministic). */
#define CP2_MIRROR_MAX_CLASSES 8
int cp2_mirror_loss_num_partials(int N, int64_t HW);
int cp2_mirror_loss(const float* s_logits, const float* t_logits, const int64_t* masks, float softmax_temp,
                    float lmbd_compare_loss, float* grad_s, float* grad_t, int64_t* argmax, int64_t* confusion,
                    double* partial, float* out3, int N, int C, int64_t HW, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* CP2HIP_H */
