// f1: on-device two-crop augmentation and background erasing, producing the input contract of MODEL.forward
// (main.py:616-628) straight in HBM -- the reference builds it on the CPU per sample with albumentations / torchvision:
//   loader.py:50-118   A_TwoCropsTransform: RandomResizedCrop + HorizontalFlip of the image (bilinear) and of the
//                      pixel-id / region-id maps (nearest), pixel ids = arange(1 .. H*W) at `pixel_ids_stride`
//                      (loader.py:39-43 rescale_ids + INTER_NEAREST_EXACT resize back, :66-73)
//   main.py:204-225    background view: RandomResizedCrop + flip, then RandomErasing(p=1, value=0)
// The random parameters (crop box, flip, erase box) are drawn on the host exactly as the transforms do
// (cp2_amd/augment.py) and passed in a device table, so a CPU restatement can be compared bit for bit: the id maps and
// the zero rectangle are integer / exact-zero work; the bilinear image of an fp32 dataset is fp32 arithmetic in a fixed
// order (no FMA); a uint8 dataset (what cv2.imread hands the reference) is resampled in cv2.resize's INTER_LINEAR integer
// arithmetic -- 11-bit weights, int32 horizontal pass, the vertical pass's documented shifts, the 2 x 2 area special case --
// restated from OpenCV's published source (cv2 is absent here: parity-unpinned, oracle/augment_oracle.py says how).
// The whole dataset stays resident in HBM (uint8 or fp32); one thread makes one output pixel (3 channels + 2 ids).
// HBM-bound, no MFMA: 12 B/pixel of image writes + 16 B/pixel of ids, source reads served mostly by L2.
#include "common.hpp"

struct CropArgs {
    const void* src; int src_u8;                 // [N][3][Hs][Ws] fp32 in [0,1], or uint8 (scaled by 1/255 like ToTensor)
    const int64_t* src_region;                   // [N][Hs][Ws] region ids, or NULL (region id = pixel id, loader.py:84-85)
    int N, Hs, Ws;
    const int32_t* params;                       // [B][8]: source index, top, left, crop h, crop w, flip, 0, 0
    float* out_img; int64_t* out_pix; int64_t* out_reg;   // [B][3][H][W], [B][H][W] (any may be NULL)
    int B, H, W, id_stride;
    uint32_t* out_rgbx;                          // [B][H][W] R | G<<8 | B<<16, or NULL: the view in uint8 for the photometric stages
};

__device__ __forceinline__ float src_px(const CropArgs& a, int64_t plane, int y, int x) {
    const int64_t o = plane + (int64_t)y * a.Ws + x;
    return a.src_u8 ? (float)static_cast<const unsigned char*>(a.src)[o] / 255.0f : static_cast<const float*>(a.src)[o];
}

// cv2.resize INTER_LINEAR tap of destination index d (oracle/augment_oracle.py _cv2_linear_taps): left tap, right tap, weights
__device__ __forceinline__ void cv2_tap(int d, int src_size, int dst_size, int& s0, int& s1, int& w0, int& w1) {
    const double scale = (double)src_size / (double)dst_size;
    float f = (float)(((double)d + 0.5) * scale - 0.5);
    int s = (int)floorf(f);
    f -= (float)s;
    if (s < 0) { s = 0; f = 0.f; }
    if (s >= src_size - 1) { s = src_size - 1; f = 0.f; }
    s0 = s;
    s1 = min(s + 1, src_size - 1);
    w0 = (int)rintf((1.0f - f) * 2048.0f);       // saturate_cast<short>(float): round half to even
    w1 = (int)rintf(f * 2048.0f);
}

__global__ __launch_bounds__(256) void crop_resize_flip_kernel(CropArgs a) {
    const int x = blockIdx.x * 256 + threadIdx.x, y = blockIdx.y, b = blockIdx.z;
    if (x >= a.W) return;
    const int32_t* p = a.params + b * 8;
    const int n = p[0], top = p[1], left = p[2], ch = p[3], cw = p[4], flip = p[5];
    const int xr = flip ? a.W - 1 - x : x;       // HorizontalFlip acts on the resized crop
    // ---- id maps: nearest neighbour, source cell = floor(dst * crop / out) in exact integer arithmetic
    const int sy = top + (int)(((int64_t)y * ch) / a.H), sx = left + (int)(((int64_t)xr * cw) / a.W);
    int64_t pid;
    int ry = sy, rx = sx;                        // the source cell whose id this output pixel carries
    if (a.id_stride <= 1) {
        pid = (int64_t)sy * a.Ws + sx + 1;
    } else {
        // rescale_ids keeps the centre tap of every stride block; INTER_NEAREST_EXACT maps the block grid back to
        // H x W with source = floor((dst + 0.5) * small / big)
        const int s = a.id_stride, o = s / 2;
        const int hs = (a.Hs - o + s - 1) / s, ws = (a.Ws - o + s - 1) / s;
        const int i = (int)(((int64_t)(2 * sy + 1) * hs) / (2 * a.Hs)), j = (int)(((int64_t)(2 * sx + 1) * ws) / (2 * a.Ws));
        ry = o + s * i, rx = o + s * j;
        pid = (int64_t)ry * a.Ws + rx + 1;
    }
    const int64_t o_id = ((int64_t)b * a.H + y) * a.W + x;
    if (a.out_pix) a.out_pix[o_id] = pid;
    // the region map goes through the same rescale_ids + INTER_NEAREST_EXACT round trip as the pixel ids (loader.py:75-83)
    if (a.out_reg) a.out_reg[o_id] = a.src_region ? a.src_region[((int64_t)n * a.Hs + ry) * a.Ws + rx] : pid;
    if (a.src_u8) {
        // ---- image of a uint8 dataset: cv2.resize(crop, INTER_LINEAR) in its fixed-point arithmetic, then ToTensor's / 255
        const unsigned char* S = static_cast<const unsigned char*>(a.src);
        uint32_t packed = 0;
        int x0, x1, a0, a1, y0, y1, b0, b1;
        cv2_tap(xr, cw, a.W, x0, x1, a0, a1);
        cv2_tap(y, ch, a.H, y0, y1, b0, b1);
        const bool area = ch == 2 * a.H && cw == 2 * a.W;           // exact 2 x 2 shrink: INTER_AREA's fast path
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            const unsigned char* pl = S + (((int64_t)n * 3 + c) * a.Hs + top) * a.Ws + left;
            int q;
            if (area) {
                const unsigned char* r0 = pl + (int64_t)(2 * y) * a.Ws + 2 * xr;
                q = ((int)r0[0] + (int)r0[1] + (int)r0[a.Ws] + (int)r0[a.Ws + 1] + 2) >> 2;
            } else {
                const int r0 = (int)pl[(int64_t)y0 * a.Ws + x0] * a0 + (int)pl[(int64_t)y0 * a.Ws + x1] * a1;
                const int r1 = (int)pl[(int64_t)y1 * a.Ws + x0] * a0 + (int)pl[(int64_t)y1 * a.Ws + x1] * a1;
                q = (((b0 * (r0 >> 4)) >> 16) + ((b1 * (r1 >> 4)) >> 16) + 2) >> 2;
            }
            if (a.out_img) a.out_img[(((int64_t)b * 3 + c) * a.H + y) * a.W + x] = (float)q / 255.0f;
            packed |= (uint32_t)(q & 255) << (8 * c);
        }
        if (a.out_rgbx) a.out_rgbx[o_id] = packed;
        return;
    }
    // ---- image of an fp32 dataset: bilinear with half-pixel centres, edges replicated; products and sums rounded one by one
    const float fy = ((float)y + 0.5f) * ((float)ch / (float)a.H) - 0.5f;
    const float fx = ((float)xr + 0.5f) * ((float)cw / (float)a.W) - 0.5f;
    const float cy = fminf(fmaxf(fy, 0.0f), (float)(ch - 1)), cx = fminf(fmaxf(fx, 0.0f), (float)(cw - 1));
    const int y0 = (int)floorf(cy), x0 = (int)floorf(cx);
    const int y1 = min(y0 + 1, ch - 1), x1 = min(x0 + 1, cw - 1);
    const float wy = cy - (float)y0, wx = cx - (float)x0;
    uint32_t packed = 0;
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        const int64_t plane = ((int64_t)n * 3 + c) * a.Hs * a.Ws;
        const float p00 = src_px(a, plane, top + y0, left + x0), p01 = src_px(a, plane, top + y0, left + x1);
        const float p10 = src_px(a, plane, top + y1, left + x0), p11 = src_px(a, plane, top + y1, left + x1);
        const float r0 = p00 * (1.0f - wx) + p01 * wx, r1 = p10 * (1.0f - wx) + p11 * wx;
        const float v = r0 * (1.0f - wy) + r1 * wy;
        if (a.out_img) a.out_img[(((int64_t)b * 3 + c) * a.H + y) * a.W + x] = v;
        // uint8 view for the photometric stages: value * 255 rounded half up
        const int q = (int)floorf(v * 255.0f + 0.5f);
        packed |= (uint32_t)(q < 0 ? 0 : (q > 255 ? 255 : q)) << (8 * c);
    }
    if (a.out_rgbx) a.out_rgbx[o_id] = packed;
}

// RandomErasing(value=0): img[b, :, top:top+h, left:left+w] = 0 (exactly), rects [B][4] = top, left, h, w (h = 0: skip)
__global__ __launch_bounds__(256) void erase_rect_kernel(float* __restrict__ img, const int32_t* __restrict__ rects, int H, int W) {
    const int b = blockIdx.z, c = blockIdx.y;
    const int top = rects[b * 4], left = rects[b * 4 + 1], h = rects[b * 4 + 2], w = rects[b * 4 + 3];
    const int64_t n = (int64_t)h * w;
    float* plane = img + ((int64_t)b * 3 + c) * H * W;
    for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < n; e += (int64_t)gridDim.x * 256)
        plane[(int64_t)(top + (int)(e / w)) * W + left + (int)(e % w)] = 0.0f;
}

CP2_API int cp2_crop_resize_flip(const void* src, int src_is_u8, const int64_t* src_region, int N, int Hs, int Ws,
                                 const int32_t* params, float* out_img, int64_t* out_pix, int64_t* out_reg, int B, int H,
                                 int W, int id_stride, uint32_t* out_rgbx, void* stream) {
    if (!src || !params || (!out_img && !out_rgbx)) return CP2_ERR_NULL;
    if (N <= 0 || Hs <= 0 || Ws <= 0 || B <= 0 || H <= 0 || W <= 0 || id_stride < 1) return CP2_ERR_SHAPE;
    if (B > 65535 || H > 65535) return CP2_ERR_UNSUPPORTED;
    CropArgs a{src, src_is_u8, src_region, N, Hs, Ws, params, out_img, out_pix, out_reg, B, H, W, id_stride, out_rgbx};
    hipLaunchKernelGGL(crop_resize_flip_kernel, dim3(cp2_cdiv(W, 256), H, B), dim3(256), 0, cp2_stream(stream), a);
    return cp2_launch_status();
}

CP2_API int cp2_erase_rect(float* img, const int32_t* rects, int B, int H, int W, void* stream) {
    if (!img || !rects) return CP2_ERR_NULL;
    if (B <= 0 || H <= 0 || W <= 0) return CP2_ERR_SHAPE;
    if (B > 65535) return CP2_ERR_UNSUPPORTED;
    hipLaunchKernelGGL(erase_rect_kernel, dim3(cp2_cdiv((int64_t)H * W, 256 * 4), 3, B), dim3(256), 0, cp2_stream(stream), img,
                       rects, H, W);
    return cp2_launch_status();
}
