// Crop -> /255 -> grayscale -> antialiased bilinear resize -> repeat-last pad -> (x-0.5)/0.5,
// written straight into the packed recogniser batch.
//
// Restates utils/image/bounding_box/processing.py:128-132 (the slice), utils/image/image_io.py:126-135
// (dtype, gray, resize), utils/image/image_processing.py:54-79,125-134 (resize to the intermediate size,
// repeat_last padding) and utils/image/image_normalization.py:58.  The resize is the separable
// triangle filter Keras' `image.resize(bilinear, antialias=True)` runs on its torch backend:
// width first, then height, taps and weights in float32 with the index arithmetic of the filter
// bounds in float64, each tap accumulated with a separate multiply and add (-ffp-contract=off).
//
// A block = (crop, band of 16 output rows).  It resizes the source rows its band needs horizontally once into LDS
// (the taps of a column computed once per block), then forms the vertical sums from LDS; a band whose source rows
// do not fit the tables (very tall or wide crops) computes one output pixel per thread iteration and recomputes the
// horizontal sums of the rows a pixel needs, so there is no size limit.  Same operations in the same order either
// way.  Bound by L2 gathers of the 3-byte source pixels and the filter's float64 index arithmetic:
// algorithmic traffic = 3 B x crop pixels read + 4 (or 2) B x 64 x Wpad written.
#include <hip/hip_bf16.h>

#include "common.h"

namespace ocrhip {

namespace {

constexpr int kCropRowFloats = 4096;      // horizontally resized source rows a block keeps in LDS (16 KB)
constexpr int kCropGrayFloats = 3072;     // gray cells of those source rows before the horizontal resize (12 KB)
constexpr int kCropMaxCols = 512;         // resized columns with their taps in LDS
constexpr int kCropMaxRows = 32;          // output rows of a block with their taps in LDS
constexpr int kCropTaps = 4;              // normalised weights kept per output row (the taps beyond are recomputed)

struct Axis {
    float scale, support, invscale;
    int in, out;
    bool identity;
};

__device__ __forceinline__ Axis make_axis(int in, int out) {
    Axis a;
    a.in = in;
    a.out = out;
    a.identity = (in == out);
    a.scale = (float)in / (float)out;
    a.support = (a.scale >= 1.0f) ? a.scale : 1.0f;
    a.invscale = (a.scale >= 1.0f) ? (float)(1.0 / (double)a.scale) : 1.0f;
    return a;
}

// first source index and tap count of output coordinate i
__device__ __forceinline__ void tap_range(const Axis& a, int i, float& center, int& lo, int& cnt) {
    center = (float)((double)a.scale * ((double)i + 0.5));
    const long long l = (long long)((double)(center - a.support) + 0.5);
    const long long h = (long long)((double)(center + a.support) + 0.5);
    lo = (int)(l > 0 ? l : 0);
    cnt = (int)(h < (long long)a.in ? h : (long long)a.in) - lo;
}

__device__ __forceinline__ float tap_weight(const Axis& a, float center, int lo, int j) {
    float x = (float)(((double)((float)(j + lo) - center) + 0.5) * (double)a.invscale);
    x = fabsf(x);
    return (x < 1.0f) ? (1.0f - x) : 0.0f;
}

// `unit` = the 256 values u / 255.0f (a float division is ten instructions; a pixel tap needs three)
__device__ __forceinline__ float gray_at(const uint8_t* __restrict__ img, const float* __restrict__ unit, int W, int y, int x) {
    const uint8_t* p = img + ((int64_t)y * W + x) * 3;
    const float r = unit[p[0]], g = unit[p[1]], b = unit[p[2]];
    float v = r * 0.2989f;
    v = v + g * 0.5870f;
    v = v + b * 0.1140f;
    return v;
}

// Where a crop takes its pixels from.  ROT = false: the slice of the page that starts at (x0, y0).  ROT = true (an
// extension, the reference crops axis-aligned slices only: utils/image/bounding_box/processing.py:111-132): an
// oriented rectangle; cell (x, y) of the crop has its centre at origin + (x + 0.5) * step_x + (y + 0.5) * step_y in
// page coordinates and takes the bilinear interpolation of the page's gray values there (neighbours clamped to the
// page).  An unrotated rectangle on integer corners samples pixel centres: exactly the slice.
template <bool ROT>
struct Source {
    const uint8_t* img;
    const float* unit;      // u / 255.0f for u = 0..255 (LDS)
    int W, H, x0, y0;
    float ox, oy, sxx, sxy, syx, syy;
    __device__ __forceinline__ float at(int y, int x) const {
        if (!ROT) return gray_at(img, unit, W, y0 + y, x0 + x);
        const float fx = (float)x + 0.5f, fy = (float)y + 0.5f;
        const float px = (ox + fx * sxx) + fy * syx;
        const float py = (oy + fx * sxy) + fy * syy;
        const float qx = px - 0.5f, qy = py - 0.5f;
        const float flx = floorf(qx), fly = floorf(qy);
        const float tx = qx - flx, ty = qy - fly;
        // clamped in float first: a far-away coordinate must not overflow the integer conversion
        const int ix = (int)fminf(fmaxf(flx, -1.0f), (float)W), iy = (int)fminf(fmaxf(fly, -1.0f), (float)H);
        const int xa = min(max(ix, 0), W - 1), xb = min(max(ix + 1, 0), W - 1);
        const int ya = min(max(iy, 0), H - 1), yb = min(max(iy + 1, 0), H - 1);
        const float top = gray_at(img, unit, W, ya, xa) * (1.0f - tx) + gray_at(img, unit, W, ya, xb) * tx;
        const float bot = gray_at(img, unit, W, yb, xa) * (1.0f - tx) + gray_at(img, unit, W, yb, xb) * tx;
        return top * (1.0f - ty) + bot * ty;
    }
};

// horizontally resized gray value of source row y (crop coordinates) at resized column sx
template <bool ROT>
__device__ __forceinline__ float hsample(const Source<ROT>& src, int yy, const Axis& ax, int sx) {
    if (ax.identity) return src.at(yy, sx);
    float center;
    int lo, cnt;
    tap_range(ax, sx, center, lo, cnt);
    float total = 0.0f;
    for (int j = 0; j < cnt; ++j) total = total + tap_weight(ax, center, lo, j);
    float acc = 0.0f;
    for (int j = 0; j < cnt; ++j) {
        float w = tap_weight(ax, center, lo, j);
        if (total != 0.0f) w = w / total;
        const float term = src.at(yy, lo + j) * w;
        acc = (j == 0) ? term : (acc + term);
    }
    return acc;
}

// rects: ROT = false: (n,4) int32 x0, y0, x1, y1 in pixels; ROT = true: (n,8) float32 origin x, y, step_x (x, y),
// step_y (x, y), crop width, crop height (whole numbers)
template <bool BF16, bool ROT>
__global__ __launch_bounds__(256) void crop_resize_pack_kernel(const uint8_t* __restrict__ img, int B, int H, int W,
                                                               const void* __restrict__ rects_v,
                                                               const int* __restrict__ box_img,
                                                               const int* __restrict__ inter,
                                                               const int* __restrict__ out_w_arr,
                                                               const int64_t* __restrict__ dst_off, int out_h,
                                                               void* __restrict__ packed) {
    const int n = blockIdx.x;
    int b = box_img[n];
    b = min(max(b, 0), B - 1);
    __shared__ float s_unit[256];
    s_unit[threadIdx.x] = (float)threadIdx.x / 255.0f;          // 256 threads
    __syncthreads();
    Source<ROT> src;
    src.unit = s_unit;
    src.img = img + (int64_t)b * H * W * 3;
    src.W = W;
    src.H = H;
    int ch, cw;
    if (ROT) {
        const float* g = reinterpret_cast<const float*>(rects_v) + 8 * (int64_t)n;
        src.ox = g[0]; src.oy = g[1]; src.sxx = g[2]; src.sxy = g[3]; src.syx = g[4]; src.syy = g[5];
        src.x0 = src.y0 = 0;
        // every read is clamped to the page, so any geometry is safe; the size only has to be sane
        cw = (int)fminf(fmaxf(g[6], 1.0f), 65536.0f);
        ch = (int)fminf(fmaxf(g[7], 1.0f), 65536.0f);
    } else {
        const int* rects = reinterpret_cast<const int*>(rects_v);
        // the rectangle is clamped to the image so that a bad rectangle can never read out of bounds
        const int x0 = min(max(rects[4 * n + 0], 0), W - 1), y0 = min(max(rects[4 * n + 1], 0), H - 1);
        const int x1 = min(max(rects[4 * n + 2], x0 + 1), W), y1 = min(max(rects[4 * n + 3], y0 + 1), H);
        src.x0 = x0;
        src.y0 = y0;
        src.ox = src.oy = src.sxx = src.sxy = src.syx = src.syy = 0.0f;
        ch = y1 - y0;
        cw = x1 - x0;
    }
    const int out_w = out_w_arr[n];
    const int ih = min(max(inter[2 * n + 0], 1), out_h), iw = min(max(inter[2 * n + 1], 1), out_w);
    const Axis ax = make_axis(cw, iw), ay = make_axis(ch, ih);
    const int64_t dst = dst_off[n];

    const int rows_per_block = (out_h + gridDim.y - 1) / gridDim.y;
    const int r0 = blockIdx.y * rows_per_block;
    const int r1 = min(out_h, r0 + rows_per_block);
    if (r1 <= r0) return;
    const int total = (r1 - r0) * out_w;

    // Source rows (crop coordinates) that this block's output rows take their vertical taps from
    int ylo, yhi;
    {
        const int sya = min(r0, ih - 1), syb = min(r1 - 1, ih - 1);
        if (ay.identity) {
            ylo = sya;
            yhi = syb + 1;
        } else {
            float c;
            int lo, cnt;
            tap_range(ay, sya, c, lo, cnt);
            ylo = lo;
            tap_range(ay, syb, c, lo, cnt);
            yhi = lo + cnt;
        }
    }
    // Separable path: the taps of every resized column once (s_col), the horizontally resized source rows of the band
    // once (s_rows), then the vertical sums from LDS.  The same operations in the same order as the per-pixel path
    // below, which recomputes a row's horizontal sum for every output row that uses it (2-3 times) and a column's taps
    // for every row: identical results.  Bands that do not fit the tables take the per-pixel path.
    __shared__ float s_rows[kCropRowFloats];
    __shared__ float s_colc[kCropMaxCols], s_colt[kCropMaxCols];      // centre, sum of the tap weights
    __shared__ int s_collo[kCropMaxCols], s_colcnt[kCropMaxCols];     // first tap, tap count
    __shared__ float s_gray[kCropGrayFloats];                         // gray cells of the band's source rows
    __shared__ float s_roww[kCropMaxRows * kCropTaps];                // normalised weights of a row's first taps
    __shared__ float s_rowc[kCropMaxRows], s_rowt[kCropMaxRows];
    __shared__ int s_rowlo[kCropMaxRows], s_rowcnt[kCropMaxRows];
    const int nrows = yhi - ylo;
    if (nrows > 0 && (int64_t)nrows * iw <= kCropRowFloats && iw <= kCropMaxCols && r1 - r0 <= kCropMaxRows) {      // uniform
        if (!ax.identity) {
            for (int sx = threadIdx.x; sx < iw; sx += blockDim.x) {
                float center;
                int lo, cnt;
                tap_range(ax, sx, center, lo, cnt);
                float tw = 0.0f;
                for (int j = 0; j < cnt; ++j) tw = tw + tap_weight(ax, center, lo, j);
                s_colc[sx] = center; s_colt[sx] = tw; s_collo[sx] = lo; s_colcnt[sx] = cnt;
            }
        }
        if (!ay.identity && (int)threadIdx.x < r1 - r0) {
            const int sy = min(r0 + (int)threadIdx.x, ih - 1);
            float center;
            int lo, cnt;
            tap_range(ay, sy, center, lo, cnt);
            float tw = 0.0f;
            for (int j = 0; j < cnt; ++j) tw = tw + tap_weight(ay, center, lo, j);
            s_rowc[threadIdx.x] = center; s_rowt[threadIdx.x] = tw; s_rowlo[threadIdx.x] = lo; s_rowcnt[threadIdx.x] = cnt;
            for (int j = 0; j < min(cnt, kCropTaps); ++j) {
                float w = tap_weight(ay, center, lo, j);
                if (tw != 0.0f) w = w / tw;
                s_roww[threadIdx.x * kCropTaps + j] = w;
            }
        }
        // every cell of the band's source rows once, with independent loads in flight (the taps of neighbouring columns
        // share cells, and a thread that fetched its cells tap by tap would wait for memory once per tap)
        const bool gray_cached = (int64_t)nrows * cw <= kCropGrayFloats;          // uniform
        if (gray_cached) {
#pragma unroll 4
            for (int i = threadIdx.x; i < nrows * cw; i += blockDim.x) s_gray[i] = src.at(ylo + i / cw, i % cw);
        }
        __syncthreads();
        auto cell = [&](int yy, int x) { return gray_cached ? s_gray[(yy - ylo) * cw + x] : src.at(yy, x); };
        for (int i = threadIdx.x; i < nrows * iw; i += blockDim.x) {
            const int yy = ylo + i / iw, sx = i % iw;
            float acc;
            if (ax.identity) {
                acc = cell(yy, sx);
            } else {
                const float center = s_colc[sx], tw = s_colt[sx];
                const int lo = s_collo[sx], cnt = s_colcnt[sx];
                acc = 0.0f;
                for (int j = 0; j < cnt; ++j) {
                    float w = tap_weight(ax, center, lo, j);
                    if (tw != 0.0f) w = w / tw;
                    const float term = cell(yy, lo + j) * w;
                    acc = (j == 0) ? term : (acc + term);
                }
            }
            s_rows[i] = acc;
        }
        __syncthreads();
        for (int i = threadIdx.x; i < total; i += blockDim.x) {
            const int ry = i / out_w, ox = i % out_w;
            const int oy = r0 + ry;
            const int sy = min(oy, ih - 1), sx = min(ox, iw - 1);     // repeat_last: right, then bottom
            float v;
            if (ay.identity) {
                v = s_rows[(sy - ylo) * iw + sx];
            } else {
                const float center = s_rowc[ry], tw = s_rowt[ry];
                const int lo = s_rowlo[ry], cnt = s_rowcnt[ry];
                v = 0.0f;
                for (int j = 0; j < cnt; ++j) {
                    float w;
                    if (j < kCropTaps) {
                        w = s_roww[ry * kCropTaps + j];
                    } else {
                        w = tap_weight(ay, center, lo, j);
                        if (tw != 0.0f) w = w / tw;
                    }
                    const float term = s_rows[(lo + j - ylo) * iw + sx] * w;
                    v = (j == 0) ? term : (v + term);
                }
            }
            v = (v - 0.5f) / 0.5f;
            const int64_t o = dst + (int64_t)oy * out_w + ox;
            if (BF16)
                reinterpret_cast<__hip_bfloat16*>(packed)[o] = __float2bfloat16(v);
            else
                reinterpret_cast<float*>(packed)[o] = v;
        }
        return;
    }
    for (int i = threadIdx.x; i < total; i += blockDim.x) {
        const int oy = r0 + i / out_w, ox = i % out_w;
        const int sy = min(oy, ih - 1), sx = min(ox, iw - 1);     // repeat_last: right, then bottom
        float v;
        if (ay.identity) {
            v = hsample(src, sy, ax, sx);
        } else {
            float center;
            int lo, cnt;
            tap_range(ay, sy, center, lo, cnt);
            float total_w = 0.0f;
            for (int j = 0; j < cnt; ++j) total_w = total_w + tap_weight(ay, center, lo, j);
            v = 0.0f;
            for (int j = 0; j < cnt; ++j) {
                float w = tap_weight(ay, center, lo, j);
                if (total_w != 0.0f) w = w / total_w;
                const float term = hsample(src, lo + j, ax, sx) * w;
                v = (j == 0) ? term : (v + term);
            }
        }
        v = (v - 0.5f) / 0.5f;
        const int64_t o = dst + (int64_t)oy * out_w + ox;
        if (BF16)
            reinterpret_cast<__hip_bfloat16*>(packed)[o] = __float2bfloat16(v);
        else
            reinterpret_cast<float*>(packed)[o] = v;
    }
}

// ---- whole-page resize for the detector input -----------------------------------------------------------
// `/255` (uint8 input) -> antialiased bilinear stretch to (oh, ow) -> (x - 0.5) / 0.5, 3 channels, same filter
// arithmetic as above (width first, then height).  Restates base_image_model.py:93-131 with the EAST settings:
// image_io.py:126-128 (dtype), image_processing.py:54-79,187-199 (stretch to the next multiples of 32) and
// image_normalization.py:28-39,57.  HBM bound: 3 (or 12) B x source pixels + 12 B x output pixels.
template <bool U8>
__device__ __forceinline__ float chan_at(const void* __restrict__ img, int W, int y, int x, int c) {
    const int64_t i = ((int64_t)y * W + x) * 3 + c;
    if (U8) return (float)reinterpret_cast<const uint8_t*>(img)[i] / 255.0f;
    return reinterpret_cast<const float*>(img)[i];
}

template <bool U8>
__device__ __forceinline__ float hsample_c(const void* __restrict__ img, int W, int yy, const Axis& ax, int sx, int c) {
    if (ax.identity) return chan_at<U8>(img, W, yy, sx, c);
    float center;
    int lo, cnt;
    tap_range(ax, sx, center, lo, cnt);
    float total = 0.0f;
    for (int j = 0; j < cnt; ++j) total = total + tap_weight(ax, center, lo, j);
    float acc = 0.0f;
    for (int j = 0; j < cnt; ++j) {
        float w = tap_weight(ax, center, lo, j);
        if (total != 0.0f) w = w / total;
        const float term = chan_at<U8>(img, W, yy, lo + j, c) * w;
        acc = (j == 0) ? term : (acc + term);
    }
    return acc;
}

template <bool U8>
__global__ __launch_bounds__(256) void resize_rgb_norm_kernel(const void* __restrict__ img, int H, int W, int oh, int ow,
                                                              float* __restrict__ out, int64_t total) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;        // one output element (b, oy, ox, c)
    if (i >= total) return;
    const int c = (int)(i % 3);
    int64_t r = i / 3;
    const int ox = (int)(r % ow); r /= ow;
    const int oy = (int)(r % oh);
    const int64_t b = r / oh;
    const void* src = U8 ? (const void*)(reinterpret_cast<const uint8_t*>(img) + b * H * W * 3)
                         : (const void*)(reinterpret_cast<const float*>(img) + b * H * W * 3);
    const Axis ax = make_axis(W, ow), ay = make_axis(H, oh);
    float v;
    if (ay.identity) {
        v = hsample_c<U8>(src, W, oy, ax, ox, c);
    } else {
        float center;
        int lo, cnt;
        tap_range(ay, oy, center, lo, cnt);
        float total_w = 0.0f;
        for (int j = 0; j < cnt; ++j) total_w = total_w + tap_weight(ay, center, lo, j);
        v = 0.0f;
        for (int j = 0; j < cnt; ++j) {
            float w = tap_weight(ay, center, lo, j);
            if (total_w != 0.0f) w = w / total_w;
            const float term = hsample_c<U8>(src, W, lo + j, ax, ox, c) * w;
            v = (j == 0) ? term : (v + term);
        }
    }
    out[i] = (v - 0.5f) / 0.5f;
}

}  // namespace

int launch_resize_rgb_norm(const void* img, bool is_u8, int B, int H, int W, int oh, int ow, float* out, hipStream_t s) {
    OCR_REQUIRE(img && out && B > 0 && H > 0 && W > 0 && oh > 0 && ow > 0, "image_resize: bad arguments B=%d %dx%d -> %dx%d",
                B, H, W, oh, ow);
    const int64_t total = (int64_t)B * oh * ow * 3;
    const int64_t blocks = (total + 255) / 256;
    OCR_REQUIRE(blocks < (1ll << 31), "image_resize: too large");
    ProfScope prof("resize_rgb_norm", 0.0, (double)B * H * W * 3 * (is_u8 ? 1.0 : 4.0) + 4.0 * (double)total, s);
    if (is_u8)
        hipLaunchKernelGGL(resize_rgb_norm_kernel<true>, dim3((unsigned)blocks), dim3(256), 0, s, img, H, W, oh, ow, out, total);
    else
        hipLaunchKernelGGL(resize_rgb_norm_kernel<false>, dim3((unsigned)blocks), dim3(256), 0, s, img, H, W, oh, ow, out, total);
    OCR_HIP_CHECK(hipGetLastError());
    return OCRHIP_OK;
}

int launch_crop_resize_pack(const uint8_t* img, int B, int H, int W, const void* rects, bool rotated, const int* box_img,
                            const int* inter, const int* out_w, const int64_t* dst_off, int n, int out_h,
                            void* packed, int out_dtype, hipStream_t s) {
    OCR_REQUIRE(img && rects && box_img && inter && out_w && dst_off && packed, "crop_resize_pack: null argument");
    OCR_REQUIRE(B > 0 && H > 0 && W > 0 && n >= 0 && out_h > 0, "crop_resize_pack: bad sizes B=%d H=%d W=%d n=%d out_h=%d",
                B, H, W, n, out_h);
    OCR_REQUIRE(out_dtype == OCRHIP_F32 || out_dtype == OCRHIP_BF16, "crop_resize_pack: out_dtype must be F32 or BF16");
    if (n == 0) return OCRHIP_OK;
    dim3 grid(n, 4), block(256);
    ProfScope prof(rotated ? "crop_rotated_pack" : "crop_resize_pack", 0.0, 0.0, s);
#define OCR_CROP_LAUNCH(BF, ROT)                                                                                        \
    hipLaunchKernelGGL((crop_resize_pack_kernel<BF, ROT>), grid, block, 0, s, img, B, H, W, rects, box_img, inter, out_w, \
                       dst_off, out_h, packed)
    if (out_dtype == OCRHIP_BF16) {
        if (rotated) OCR_CROP_LAUNCH(true, true); else OCR_CROP_LAUNCH(true, false);
    } else {
        if (rotated) OCR_CROP_LAUNCH(false, true); else OCR_CROP_LAUNCH(false, false);
    }
#undef OCR_CROP_LAUNCH
    OCR_HIP_CHECK(hipGetLastError());
    return OCRHIP_OK;
}

}  // namespace ocrhip
