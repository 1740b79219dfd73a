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
 *     library allocates nothing, keeps no state and never synchronises the host;
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

/* ---- a1 (+ mask part of a2): copy-paste composition ---- builder.py:1146-1159
 * mask = (bg[:,0] == 0) ? 1 : 0;  out_img = img * mask + bg   (bit-exact, no FMA)
 * img, bg, out_img: [B,3,H,W]; mask_full: [B,H,W] or NULL;
 * mask_ds: [B,Hs,Ws] = mask[:, s/2::s, s/2::s] or NULL (Hs = ceil((H-s/2)/s)). */
int cp2_compose_mask(const float* img, const float* bg, float* out_img, float* mask_full,
                     float* mask_ds, int B, int H, int W, int stride, void* stream);

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

/* ---- a11: momentum (EMA) update of the key encoder -------- builder.py:557-567
 * k[i] = k[i]*m + q[i]*one_minus_m  (two rounded products, one rounded sum; no FMA).
 * Flat form: one contiguous span of n floats (16-byte aligned). */
int cp2_ema_flat(float* k, const float* q, int64_t n, float m, float one_minus_m, void* stream);
/* Multi-tensor form: device tables of n_tensors pointers/sizes, plus a device chunk
 * table built by the caller: chunk c covers elements [chunk_off[c], chunk_off[c]+chunk_len[c])
 * of tensor chunk_tensor[c].  One launch for the whole encoder. */
int cp2_ema_multi(float* const* k_ptrs, const float* const* q_ptrs, const int32_t* chunk_tensor,
                  const int64_t* chunk_off, const int32_t* chunk_len, int n_chunks,
                  float m, float one_minus_m, void* stream);

/* ---- a13: queue enqueue with wrap-around ------------------ builder.py:569-587
 * queue[c, (ptr+i) % K] = keys[i, c] for i < n, then *ptr = (ptr + n) % K.
 * queue: [C,K] f32; keys: [n,C] f32 (already gathered over ranks); ptr: device int64[1].
 * The pointer is read and advanced on the device: no host synchronisation.  n <= K. */
int cp2_enqueue(float* queue, const float* keys, int64_t* ptr, int n, int C, int K, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* CP2HIP_H */
