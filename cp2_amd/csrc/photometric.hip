// f1, photometric half: the image-value transforms of the reference's input pipeline on the device, in the integer /
// float arithmetic of the library the reference calls for them (Pillow, through torchvision's PIL code path):
//   main.py:209-211  RandomResizedCrop on a PIL image = img.crop(box).resize(size, BILINEAR)   -> pil_resize_kernel
//   main.py:212-214  ColorJitter(0.4, 0.4, 0.4, 0.1) p=0.8  = ImageEnhance.{Brightness,Contrast,Color}.enhance(f) and the
//                    HSV round trip of adjust_hue, in a random order                             -> color_kernel
//   main.py:215      RandomGrayscale(p=0.2) = img.convert("L") in three bands                    -> color_kernel
//   main.py:216, loader.py:121-152  GaussianBlur([0.1, 2.0]) p=0.5 = ImageFilter.GaussianBlur(sigma) (both the
//                    background AND the foreground views use Pillow's filter)                    -> blur_tensor_kernel
//   main.py:217-224  HorizontalFlip, ToTensor (uint8 / 255), RandomErasing(value=0)             -> fused into the above
// Every random parameter is drawn on the host (cp2_amd/augment.py) and arrives in a device table; the kernels are
// deterministic functions of (source pixels, parameters), checked bit for bit against oracle/augment_oracle.py, which is
// itself pinned against Pillow (tests/test_augment_photometric.py, tests/golden/make_augment_goldens.py).
//
// Working format between the stages: one uint32 per pixel = R | G<<8 | B<<16 (a 4-byte lane per pixel, 256 B per wave
// row segment), [B][H][W].  All stages are HBM / latency bound byte work; no MFMA.
//
// Pillow algorithms restated here (un-vendored third-party code, version un-pinned in the reference's
// requirements.txt; 12.2.0 in the build image):
//   Resample.c   precompute_coeffs / normalize_coeffs_8bpc / ImagingResampleHorizontal_8bpc / Vertical_8bpc
//   Blend.c      ImagingBlend (float interpolation, truncation, clipping only when alpha is outside [0,1])
//   Convert.c    rgb2l (L24), rgb2hsv_row, hsv2rgb (after colorsys)
//   BoxBlur.c    ImagingGaussianBlur = 3 x ImagingHorizontalBoxBlur per axis, 24-bit weights, edges replicated
#include "common.hpp"

namespace {

constexpr int kPrecisionBits = 32 - 8 - 2;     // Resample.c PRECISION_BITS

__device__ __forceinline__ int clip8(int v) { return v < 0 ? 0 : (v > 255 ? 255 : v); }

// ---------------------------------------------------------------------------------------------- Resample.c coefficients
// One thread = one output index of one axis of one sample: first input index, tap count, integer taps.
// ws layout: [B][2 (0 = rows / vertical, 1 = columns / horizontal)][max(H,W)][2 + KS] int32.
__global__ __launch_bounds__(256) void resize_coeffs_kernel(const int32_t* __restrict__ params, int32_t* __restrict__ ws, int B, int H,
                                                            int W, int KS, int L) {
    const int idx = blockIdx.x * 256 + threadIdx.x, axis = blockIdx.y, b = blockIdx.z;
    const int out_size = axis ? W : H;
    if (idx >= out_size) return;
    const int in_size = params[b * 8 + (axis ? 4 : 3)];          // crop w / crop h: the whole cropped image is the box
    const double scale = (double)in_size / (double)out_size;
    const double filterscale = scale < 1.0 ? 1.0 : scale;
    const double support = 1.0 * filterscale, ss = 1.0 / filterscale;
    const double center = 0.0 + ((double)idx + 0.5) * scale;
    int xmin = (int)(center - support + 0.5);
    if (xmin < 0) xmin = 0;
    int xmax = (int)(center + support + 0.5);
    if (xmax > in_size) xmax = in_size;
    xmax -= xmin;
    if (xmax > KS) xmax = KS;                                     // cannot happen for KS from cp2_pil_resize_ksize; memory safety
    int32_t* row = ws + (((int64_t)b * 2 + axis) * L + idx) * (2 + KS);
    double ww = 0.0;
    for (int x = 0; x < xmax; ++x) {
        double v = ((double)(x + xmin) - center + 0.5) * ss;
        if (v < 0.0) v = -v;
        ww += v < 1.0 ? 1.0 - v : 0.0;
    }
    row[0] = xmin, row[1] = xmax;
    for (int x = 0; x < KS; ++x) {
        double k = 0.0;
        if (x < xmax) {
            double v = ((double)(x + xmin) - center + 0.5) * ss;
            if (v < 0.0) v = -v;
            k = v < 1.0 ? 1.0 - v : 0.0;
            if (ww != 0.0) k = k / ww;
        }
        row[2 + x] = k < 0.0 ? (int)(-0.5 + k * (double)(1 << kPrecisionBits)) : (int)(0.5 + k * (double)(1 << kPrecisionBits));
    }
}

// ---------------------------------------------------------------------------------------------- crop + resize (+ flip)
// One thread = one output pixel.  For each vertical tap row the horizontal pass is evaluated and rounded to uint8 first
// (Pillow stores the intermediate image in uint8), then the vertical pass combines those.
__global__ __launch_bounds__(256) void pil_resize_kernel(const unsigned char* __restrict__ src, int N, int Hs, int Ws,
                                                         const int32_t* __restrict__ params, const int32_t* __restrict__ ws,
                                                         uint32_t* __restrict__ out, int H, int W, int KS, int L) {
    const int x = blockIdx.x * 256 + threadIdx.x, y = blockIdx.y, b = blockIdx.z;
    if (x >= W) return;
    const int32_t* p = params + b * 8;
    const int n = p[0], top = p[1], left = p[2], flip = p[5];
    const int xr = flip ? W - 1 - x : x;                           // HorizontalFlip commutes with everything after it
    const int32_t* kv = ws + (((int64_t)b * 2 + 0) * L + y) * (2 + KS);
    const int32_t* kh = ws + (((int64_t)b * 2 + 1) * L + xr) * (2 + KS);
    const int ymin = kv[0], ny = kv[1], xmin = kh[0], nx = kh[1];
    const int64_t plane = (int64_t)Hs * Ws;
    const unsigned char* base = src + (int64_t)n * 3 * plane + (int64_t)(top + ymin) * Ws + left + xmin;
    int acc[3] = {1 << (kPrecisionBits - 1), 1 << (kPrecisionBits - 1), 1 << (kPrecisionBits - 1)};
    for (int j = 0; j < ny; ++j) {
        const unsigned char* r = base + (int64_t)j * Ws;
        int h[3] = {1 << (kPrecisionBits - 1), 1 << (kPrecisionBits - 1), 1 << (kPrecisionBits - 1)};
        for (int i = 0; i < nx; ++i) {
            const int k = kh[2 + i];
            h[0] += (int)r[i] * k, h[1] += (int)r[plane + i] * k, h[2] += (int)r[2 * plane + i] * k;
        }
        const int k = kv[2 + j];
        acc[0] += clip8(h[0] >> kPrecisionBits) * k, acc[1] += clip8(h[1] >> kPrecisionBits) * k, acc[2] += clip8(h[2] >> kPrecisionBits) * k;
    }
    out[((int64_t)b * H + y) * W + x] = (uint32_t)clip8(acc[0] >> kPrecisionBits) | ((uint32_t)clip8(acc[1] >> kPrecisionBits) << 8) |
                                        ((uint32_t)clip8(acc[2] >> kPrecisionBits) << 16);
}

// ---------------------------------------------------------------------------------------------- colour operations
struct Rgb { int r, g, b; };

__device__ __forceinline__ int rgb2l(Rgb c) { return (c.r * 19595 + c.g * 38470 + c.b * 7471 + 0x8000) >> 16; }

// Blend.c: out = in1 + alpha * (in2 - in1), float arithmetic, truncated; clipped when alpha is outside [0, 1]
__device__ __forceinline__ int blend1(int deg, int px, float alpha, bool clip) {
    const float v = (float)deg + alpha * (float)(px - deg);
    if (!clip) return (int)v & 255;
    return v <= 0.0f ? 0 : (v >= 255.0f ? 255 : (int)v);
}

__device__ __forceinline__ Rgb hue_shift(Rgb c, int shift) {
    // rgb2hsv_row
    const int maxc = max(c.r, max(c.g, c.b)), minc = min(c.r, min(c.g, c.b));
    int uh = 0, us = 0;
    const int uv = maxc;
    if (minc != maxc) {
        const float cr = (float)(maxc - minc);
        const float s = cr / (float)maxc;
        const float rc = (float)(maxc - c.r) / cr, gc = (float)(maxc - c.g) / cr, bc = (float)(maxc - c.b) / cr;
        float h;
        if (c.r == maxc) h = bc - gc;
        else if (c.g == maxc) h = (float)(2.0 + (double)rc - (double)bc);
        else h = (float)(4.0 + (double)gc - (double)rc);
        const double t = (double)h / 6.0 + 1.0;                    // in (0.8, 1.9): fmod(t, 1.0) = t - floor(t), exact
        h = (float)(t - floor(t));
        uh = clip8((int)((double)h * 255.0));
        us = clip8((int)((double)s * 255.0));
    }
    uh = (uh + shift) & 255;                                       // torchvision adjust_hue: uint8 wrap-around
    // hsv2rgb
    if (us == 0) return Rgb{uv, uv, uv};
    const double hf = (double)(float)uh * 6.0 / 255.0;
    const int i = (int)floor(hf);
    const float f = (float)(hf - (double)(float)i);
    const float fs = (float)((double)(float)us / 255.0);
    const double vf = (double)(float)uv;
    const int p = clip8((int)round(vf * (1.0 - (double)fs)));
    const int q = clip8((int)round(vf * (1.0 - (double)fs * (double)f)));
    const int t = clip8((int)round(vf * (1.0 - (double)fs * (1.0 - (double)f))));
    switch (i % 6) {
        case 0: return Rgb{uv, t, p};
        case 1: return Rgb{q, uv, p};
        case 2: return Rgb{p, uv, t};
        case 3: return Rgb{p, q, uv};
        case 4: return Rgb{t, p, uv};
        default: return Rgb{uv, p, q};
    }
}

// ---- the same four adjustments + grayscale as albumentations runs them on cv2 (the reference's FOREGROUND views,
// main.py:236-237, loader.py:93-109): LUTs built in float64 and truncated, cv2's 8-bit RGB2GRAY (15-bit fixed point),
// RGB2HSV (12-bit fixed point, two division tables) / HSV2RGB (float32 sector arithmetic), addWeighted (float32, rounded
// half to even).  Restated from the published sources of albumentations / OpenCV 4.x, parity-unpinned (neither is in the
// image); the CPU restatement the tests compare with is oracle/augment_oracle.py (cv2_* / albu_*).
__device__ __forceinline__ int cv_gray(Rgb c) { return (c.r * 9798 + c.g * 19235 + c.b * 3735 + (1 << 14)) >> 15; }
__device__ __forceinline__ int lut_trunc(double v) { return v <= 0.0 ? 0 : (v >= 255.0 ? 255 : (int)v); }   // np.clip + astype(uint8)
__device__ __forceinline__ int cv_round_u8(float v) { return clip8((int)rintf(v)); }                            // saturate_cast<uchar>(float)
__device__ __forceinline__ int cv_div_table(int num, int den_times, int i) {    // saturate_cast<int>((num << 12) / (den_times * i)): half to even
    return i == 0 ? 0 : (int)rint((double)(num << 12) / ((double)den_times * (double)i));
}

__device__ __forceinline__ Rgb cv_hue(Rgb c, double factor) {
    // RGB2HSV_b, hrange 180
    const int v = max(c.r, max(c.g, c.b)), vmin = min(c.r, min(c.g, c.b)), diff = v - vmin;
    const int s = (diff * cv_div_table(255, 1, v) + (1 << 11)) >> 12;
    int h = v == c.r ? c.g - c.b : (v == c.g ? c.b - c.r + 2 * diff : c.r - c.g + 4 * diff);
    h = (h * cv_div_table(180, 6, diff) + (1 << 11)) >> 12;          // arithmetic shift
    if (h < 0) h += 180;
    h = clip8(h);
    // albumentations: lut = np.mod(arange(256) + 180 * factor, 180).astype(uint8)
    double a = fmod((double)h + 180.0 * factor, 180.0);
    if (a < 0.0) a += 180.0;
    const int hh = (int)a;
    // HSV2RGB_b -> HSV2RGB_native on (h, s / 255, v / 255), hscale = 6 / 180
    const float sf = (float)s * (1.0f / 255.0f), vf = (float)v * (1.0f / 255.0f);
    if (sf == 0.0f) { const int g = cv_round_u8(vf * 255.0f); return Rgb{g, g, g}; }
    float hf = fmodf((float)hh * (6.0f / 180.0f), 6.0f);
    int sector = (int)floorf(hf);
    hf -= (float)sector;
    if ((unsigned)sector >= 6u) { sector = 0; hf = 0.0f; }
    const float tab[4] = {vf, vf * (1.0f - sf), vf * (1.0f - sf * hf), vf * (1.0f - sf * (1.0f - hf))};
    // sector_data[sector] = tab index of (b, g, r)
    const int sb[6] = {1, 1, 3, 0, 0, 2}, sg[6] = {3, 0, 0, 2, 1, 1}, sr[6] = {0, 2, 1, 1, 3, 0};
    return Rgb{cv_round_u8(tab[sr[sector]] * 255.0f), cv_round_u8(tab[sg[sector]] * 255.0f), cv_round_u8(tab[sb[sector]] * 255.0f)};
}

__device__ __forceinline__ Rgb cv_jitter_op(Rgb c, int op, double fb, double fc, float fsat, double fhue, double mean) {
    if (op == 0) {
        if (fb == 1.0) return c;
        return Rgb{lut_trunc((double)c.r * fb), lut_trunc((double)c.g * fb), lut_trunc((double)c.b * fb)};
    }
    if (op == 1) {
        if (fc == 1.0) return c;
        if (fc == 0.0) { const int m = (int)(mean + 0.5); return Rgb{m, m, m}; }
        const double off = mean * (1.0 - fc);
        return Rgb{lut_trunc((double)c.r * fc + off), lut_trunc((double)c.g * fc + off), lut_trunc((double)c.b * fc + off)};
    }
    if (op == 2) {
        if (fsat == 1.0f) return c;
        const int g = cv_gray(c);
        if (fsat == 0.0f) return Rgb{g, g, g};
        const float beta = (float)(1.0 - (double)fsat), gb = (float)g * beta;   // albumentations passes 1 - factor (float64) to cv2
        return Rgb{cv_round_u8((float)c.r * fsat + gb), cv_round_u8((float)c.g * fsat + gb), cv_round_u8((float)c.b * fsat + gb)};
    }
    if (op == 3) return fhue == 0.0 ? c : cv_hue(c, fhue);
    return c;
}

// params row (int32 [CP2_COLOR_PARAMS = 12]): [0..3] the adjustments in application order (0 brightness, 1 contrast,
// 2 saturation, 3 hue, -1 none), [4..6] float bits of the brightness / contrast / saturation factors, [7] hue shift
// (uint8(hue_factor * 255)), [8] grayscale flag, [9] arithmetic: 0 = Pillow (torchvision on PIL images: the background
// views), 1 = albumentations on cv2 (the foreground views), [10] float bits of the hue factor (arithmetic 1), [11] reserved.
// phase 0: every adjustment in front of the contrast one (all of them when there is none, then grayscale), and the sum
//          of the L values contrast needs (ImageStat over img.convert("L")) into lsum[b];
// phase 1: samples with a contrast adjustment only -- contrast and whatever follows it, then grayscale.
__global__ __launch_bounds__(256) void color_kernel(uint32_t* __restrict__ img, const int32_t* __restrict__ params,
                                                    unsigned long long* __restrict__ lsum, int HW, int phase) {
    const int b = blockIdx.y;
    const int32_t* p = params + b * CP2_COLOR_PARAMS;
    int cpos = 4;
    for (int k = 3; k >= 0; --k)
        if (p[k] == 1) cpos = k;
    if (phase == 1 && cpos == 4) return;
    const int first = phase == 0 ? 0 : cpos, last = phase == 0 ? cpos : 4;
    const float fb = __int_as_float(p[4]), fc = __int_as_float(p[5]), fsat = __int_as_float(p[6]);
    const int shift = p[7] & 255, gray = p[8];
    const bool cv = p[9] == 1;
    const double fhue = (double)__int_as_float(p[10]);
    int mean = 0;
    double cvmean = 0.0;
    if (phase == 1) {
        cvmean = (double)lsum[b] / (double)HW;                             // float64 mean of the gray image
        mean = (int)(cvmean + 0.5);                                        // int(stat.mean[0] + 0.5)
    }
    unsigned long long local = 0;
    for (int e = blockIdx.x * 256 + threadIdx.x; e < HW; e += gridDim.x * 256) {
        const uint32_t v = img[(int64_t)b * HW + e];
        Rgb c{(int)(v & 255), (int)((v >> 8) & 255), (int)((v >> 16) & 255)};
        for (int k = first; k < last; ++k) {
            const int op = p[k];
            if (cv) {
                c = cv_jitter_op(c, op, (double)fb, (double)fc, fsat, fhue, cvmean);
            } else if (op == 0) {
                const bool clip = !(fb >= 0.0f && fb <= 1.0f);
                c = Rgb{blend1(0, c.r, fb, clip), blend1(0, c.g, fb, clip), blend1(0, c.b, fb, clip)};
            } else if (op == 1) {
                const bool clip = !(fc >= 0.0f && fc <= 1.0f);
                c = Rgb{blend1(mean, c.r, fc, clip), blend1(mean, c.g, fc, clip), blend1(mean, c.b, fc, clip)};
            } else if (op == 2) {
                const bool clip = !(fsat >= 0.0f && fsat <= 1.0f);
                const int l = rgb2l(c);
                c = Rgb{blend1(l, c.r, fsat, clip), blend1(l, c.g, fsat, clip), blend1(l, c.b, fsat, clip)};
            } else if (op == 3) {
                c = hue_shift(c, shift);
            }
        }
        if (phase == 0 && cpos < 4) local += (unsigned long long)(cv ? cv_gray(c) : rgb2l(c));
        if (gray && (phase == 1 || cpos == 4)) {
            const int l = cv ? cv_gray(c) : rgb2l(c);
            c = Rgb{l, l, l};
        }
        img[(int64_t)b * HW + e] = (uint32_t)c.r | ((uint32_t)c.g << 8) | ((uint32_t)c.b << 16);
    }
    if (phase == 0 && cpos < 4) {
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) local += __shfl_xor(local, off, 64);
        if ((threadIdx.x & 63) == 0) atomicAdd(&lsum[b], local);   // integer sum: the order does not matter
    }
}

// ---------------------------------------------------------------------------------------------- Gaussian blur + ToTensor
// params row (int32 [4]): on, integer box radius r, ww, fw (BoxBlur.c: weight of the 2r+1 inner pixels and of the two far
// pixels, 24-bit fixed point).  One workgroup = one 32 x 32 output tile; the tile with a halo of 3 (r + 1) pixels lives
// in LDS and goes through three horizontal and three vertical box passes (uint8 after each), reads clamped to the image
// (edge replication at every pass, as ImagingLineBoxBlur does).  The result is written as ToTensor would (CHW float,
// value / 255) with the RandomErasing rectangle zeroed.
constexpr int kTile = 32;

__global__ __launch_bounds__(256) void blur_tensor_kernel(const uint32_t* __restrict__ img, const int32_t* __restrict__ params,
                                                          const int32_t* __restrict__ rects, float* __restrict__ out, int H, int W,
                                                          int halo) {
    extern __shared__ uint32_t lds[];
    const int b = blockIdx.z, ty0 = blockIdx.y * kTile, tx0 = blockIdx.x * kTile;
    const int32_t* p = params + b * 4;
    const int on = p[0];
    const int r = min(p[1], halo / 3 - 1);                          // host guarantees r <= rmax; memory safety otherwise
    const uint32_t ww = (uint32_t)p[2], fw = (uint32_t)p[3];
    const int hl = on ? 3 * (r + 1) : 0;                            // this sample's halo (<= halo)
    const int RW = kTile + 2 * hl, RH = kTile + 2 * hl;
    uint32_t* A = lds;
    uint32_t* Bf = lds + (kTile + 2 * halo) * (kTile + 2 * halo);
    const uint32_t* src = img + (int64_t)b * H * W;
    const int gy0 = ty0 - hl, gx0 = tx0 - hl;
    // thread layouts without a division: loads and horizontal passes walk rows of up to 64 columns (RW <= 62) four at a
    // time; vertical passes only need the tile's 32 columns and walk them eight rows at a time
    const int hx = threadIdx.x & 63, hy = threadIdx.x >> 6;
    const int vx = threadIdx.x & 31, vy = threadIdx.x >> 5;
    if (hx < RW) {
        const int gx = min(max(gx0 + hx, 0), W - 1);
        for (int ry = hy; ry < RH; ry += 4) {
            const int gy = min(max(gy0 + ry, 0), H - 1);
            A[ry * RW + hx] = src[(int64_t)gy * W + gx];
        }
    }
    __syncthreads();
    if (on) {
        uint32_t* in = A;
        uint32_t* o = Bf;
        for (int pass = 0; pass < 6; ++pass) {
            const bool horiz = pass < 3;
            const int n = horiz ? W : H, g0 = horiz ? gx0 : gy0, R = horiz ? RW : RH;
            const int cx = horiz ? hx : vx + hl;           // column of this thread in the region
            const int step = horiz ? 4 : 8;
            // pass k of an axis only has to produce what the remaining 2 - k passes of that axis will read: the region
            // shrinks by r + 1 on both sides per pass (the last pass of an axis produces exactly the tile's extent)
            const int shrink = ((horiz ? pass : pass - 3) + 1) * (r + 1);
            const bool col_ok = horiz ? (hx >= shrink && hx < RW - shrink) : true;
            const int row_lo = horiz ? 0 : shrink, row_hi = horiz ? RH : RH - shrink;
            for (int ry = (horiz ? hy : vy) + (row_lo / step) * step; ry < row_hi; ry += step) {
                if (ry < row_lo) continue;
                const int g = horiz ? gx0 + cx : gy0 + ry;     // global coordinate along the pass axis
                if (!col_ok || g < 0 || g >= n) continue;
                const int c = g - g0;                          // local coordinate along the pass axis
                // R and B are summed side by side in one word (fields of 16 bits: at most 11 taps x 255), G in another
                uint32_t srb = 0, sg = 0, frb, fg;
                if (g - r - 1 >= 0 && g + r + 1 < n) {         // interior: no clamping (c - r - 1 >= 0 and c + r + 1 < R hold, see above)
                    const uint32_t* q = horiz ? in + ry * RW + c : in + c * RW + cx;
                    const int st = horiz ? 1 : RW;
                    for (int d = -r; d <= r; ++d) {
                        const uint32_t v = q[d * st];
                        srb += v & 0x00FF00FFu, sg += (v >> 8) & 255u;
                    }
                    const uint32_t va = q[(-r - 1) * st], vb = q[(r + 1) * st];
                    frb = (va & 0x00FF00FFu) + (vb & 0x00FF00FFu), fg = ((va >> 8) & 255u) + ((vb >> 8) & 255u);
                } else {                                        // at an image edge: replicate, as ImagingLineBoxBlur does
                    for (int d = -r; d <= r; ++d) {
                        const int l = min(max(min(max(g + d, 0), n - 1) - g0, 0), R - 1);
                        const uint32_t v = horiz ? in[ry * RW + l] : in[l * RW + cx];
                        srb += v & 0x00FF00FFu, sg += (v >> 8) & 255u;
                    }
                    const int la = min(max(min(max(g - r - 1, 0), n - 1) - g0, 0), R - 1);
                    const int lb = min(max(min(max(g + r + 1, 0), n - 1) - g0, 0), R - 1);
                    const uint32_t va = horiz ? in[ry * RW + la] : in[la * RW + cx];
                    const uint32_t vb = horiz ? in[ry * RW + lb] : in[lb * RW + cx];
                    frb = (va & 0x00FF00FFu) + (vb & 0x00FF00FFu), fg = ((va >> 8) & 255u) + ((vb >> 8) & 255u);
                }
                const uint32_t o0 = ((srb & 0xFFFFu) * ww + (frb & 0xFFFFu) * fw + (1u << 23)) >> 24;
                const uint32_t o1 = (sg * ww + fg * fw + (1u << 23)) >> 24;
                const uint32_t o2 = ((srb >> 16) * ww + (frb >> 16) * fw + (1u << 23)) >> 24;
                o[ry * RW + cx] = (o0 & 255) | ((o1 & 255) << 8) | ((o2 & 255) << 16);
            }
            __syncthreads();
            uint32_t* t = in;
            in = o, o = t;
        }
        A = in;                                                     // six passes: the result is back in the first buffer
    }
    int et = 0, el = 0, eh = 0, ew = 0;
    if (rects) et = rects[b * 4], el = rects[b * 4 + 1], eh = rects[b * 4 + 2], ew = rects[b * 4 + 3];
    const int64_t plane = (int64_t)H * W;
    for (int e = threadIdx.x; e < kTile * kTile; e += 256) {
        const int y = ty0 + e / kTile, x = tx0 + e % kTile;
        if (y >= H || x >= W) continue;
        const uint32_t v = A[(e / kTile + hl) * RW + (e % kTile) + hl];
        const bool erased = y >= et && y < et + eh && x >= el && x < el + ew;
        float* o = out + (int64_t)b * 3 * plane + (int64_t)y * W + x;
        o[0] = erased ? 0.0f : __fdiv_rn((float)(v & 255), 255.0f);
        o[plane] = erased ? 0.0f : __fdiv_rn((float)((v >> 8) & 255), 255.0f);
        o[2 * plane] = erased ? 0.0f : __fdiv_rn((float)((v >> 16) & 255), 255.0f);
    }
}

}  // namespace

CP2_API int cp2_pil_resize_ksize(int Hs, int Ws, int H, int W) {
    if (Hs <= 0 || Ws <= 0 || H <= 0 || W <= 0) return CP2_ERR_SHAPE;
    // a crop is at most the source: support = max(scale, 1) <= max(Hs / H, Ws / W, 1); ksize = ceil(support) * 2 + 1
    const int sh = (Hs + H - 1) / H, sw = (Ws + W - 1) / W;
    const int s = sh > sw ? sh : sw;
    return (s < 1 ? 1 : s) * 2 + 1;
}

CP2_API int64_t cp2_pil_resize_workspace_bytes(int B, int Hs, int Ws, int H, int W) {
    const int ks = cp2_pil_resize_ksize(Hs, Ws, H, W);
    if (ks < 0 || B <= 0) return CP2_ERR_SHAPE;
    return (int64_t)B * 2 * (H > W ? H : W) * (2 + ks) * (int64_t)sizeof(int32_t);
}

CP2_API int cp2_pil_resize_crop(const unsigned char* src, int N, int Hs, int Ws, const int32_t* params, uint32_t* out_rgbx,
                                int B, int H, int W, int32_t* workspace, int64_t workspace_bytes, void* stream) {
    if (!src || !params || !out_rgbx || !workspace) return CP2_ERR_NULL;
    if (N <= 0 || Hs <= 0 || Ws <= 0 || B <= 0 || H <= 0 || W <= 0) return CP2_ERR_SHAPE;
    if (B > 65535 || H > 65535) return CP2_ERR_UNSUPPORTED;
    if (workspace_bytes < cp2_pil_resize_workspace_bytes(B, Hs, Ws, H, W)) return CP2_ERR_SHAPE;
    const int KS = cp2_pil_resize_ksize(Hs, Ws, H, W), L = H > W ? H : W;
    hipLaunchKernelGGL(resize_coeffs_kernel, dim3(cp2_cdiv(L, 256), 2, B), dim3(256), 0, cp2_stream(stream), params, workspace, B, H,
                       W, KS, L);
    hipLaunchKernelGGL(pil_resize_kernel, dim3(cp2_cdiv(W, 256), H, B), dim3(256), 0, cp2_stream(stream), src, N, Hs, Ws, params,
                       workspace, out_rgbx, H, W, KS, L);
    return cp2_launch_status();
}

CP2_API int cp2_color_ops(uint32_t* img_rgbx, const int32_t* params, uint64_t* lsum, int B, int H, int W, void* stream) {
    if (!img_rgbx || !params || !lsum) return CP2_ERR_NULL;
    if (B <= 0 || H <= 0 || W <= 0) return CP2_ERR_SHAPE;
    if (B > 65535 || (int64_t)H * W > (1 << 30)) return CP2_ERR_UNSUPPORTED;
    const int HW = H * W;
    hipError_t e = hipMemsetAsync(lsum, 0, sizeof(uint64_t) * (size_t)B, cp2_stream(stream));
    if (e != hipSuccess) return (int)e;
    const int blocks = cp2_cdiv(HW, 256 * 4), gx = blocks < 1024 ? blocks : 1024;
    for (int phase = 0; phase < 2; ++phase)
        hipLaunchKernelGGL(color_kernel, dim3(gx, B), dim3(256), 0, cp2_stream(stream), img_rgbx, params,
                           reinterpret_cast<unsigned long long*>(lsum), HW, phase);
    return cp2_launch_status();
}

CP2_API int cp2_blur_to_tensor(const uint32_t* img_rgbx, const int32_t* params, const int32_t* rects, float* out, int B, int H,
                               int W, int rmax, void* stream) {
    if (!img_rgbx || !params || !out) return CP2_ERR_NULL;
    if (B <= 0 || H <= 0 || W <= 0 || rmax < 0) return CP2_ERR_SHAPE;
    if (B > 65535 || rmax > CP2_BLUR_MAX_RADIUS) return CP2_ERR_UNSUPPORTED;
    const int halo = 3 * (rmax + 1), side = kTile + 2 * halo;
    const size_t lds = 2 * (size_t)side * side * sizeof(uint32_t);
    hipLaunchKernelGGL(blur_tensor_kernel, dim3(cp2_cdiv(W, kTile), cp2_cdiv(H, kTile), B), dim3(256), lds, cp2_stream(stream),
                       img_rgbx, params, rects, out, H, W, halo);
    return cp2_launch_status();
}
