// Score threshold + RBOX decode + validity filter + poly -> xyxy for the EAST output map.
//
// Restates models/detection/east.py:81-85,102-187 and the poly branch of
// utils/image/bounding_box/converter.py:151-154,165-166 with the reference's dtype flow: the map
// scaling and the angle stay float32, sin/cos are taken of the float32 angle and rounded to
// float32, every other step runs in float64, the corners are divided by (W,H) and rounded to
// float32 once.  The file is compiled with -ffp-contract=off: no multiply-add is fused.
//
// Two launches over the map, one wave-ballot compaction, no atomics, row-major output order:
//   1. east_decode_flags : every pixel above the threshold is decoded; each wave publishes the
//      64-bit mask of pixels that survive the validity filter, each 256-pixel chunk its count.
//   2. east_decode_write : each chunk sums the counts of the chunks before it, the surviving
//      pixels are decoded again (same code, same bits) and written at their final position.
// HBM traffic: the 24 B/pixel map once, plus the cache lines of surviving pixels a second time,
// plus 36 B (68 B with polygons) per box written.
#include <mutex>

#include "common.h"

namespace ocrhip {

namespace {

constexpr int kChunk = 256;  // pixels per workgroup = 4 waves

struct DecodedBox {
    float poly[8];
    float xyxy[4];
    float score;
};

// m: the 6 map channels of one pixel.  Returns true when the box passes filter_polys.
__device__ __forceinline__ bool decode_pixel(const float m[6], int mx, int my, int scale_x, int scale_y, int in_w,
                                             int in_h, DecodedBox& out) {
    const float d_top = m[1] * 512.0f, d_bottom = m[2] * 512.0f;
    const float d_left = m[3] * 512.0f, d_right = m[4] * 512.0f;
    const float theta = (m[5] - 0.5f) * 3.14159274101257324219f;  // float32(pi)

    const double x = (double)(mx * scale_x), y = (double)(my * scale_y);
    const double y_min = y - (double)d_top, y_max = y + (double)d_bottom;
    const double x_min = x - (double)d_left, x_max = x + (double)d_right;

    // rotation by -theta: cos / sin of the float32 angle, correctly rounded to float32
    double sd, cd;
    sincos((double)(-theta), &sd, &cd);
    const double c = (double)(float)cd, s = (double)(float)sd;
    const double ns = -s;

    const double tx[4] = {x_min - x, x_max - x, x_max - x, x_min - x};
    const double ty[4] = {y_min - y, y_min - y, y_max - y, y_max - y};
    const double lim_x = (double)in_w, lim_y = (double)in_h;

    int outside = 0;
    float fx[4], fy[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const double px = (tx[k] * c + ty[k] * ns) + x;
        const double py = (tx[k] * s + ty[k] * c) + y;
        outside += ((px < 0.0) || (py < 0.0) || (px >= lim_x) || (py >= lim_y)) ? 1 : 0;
        fx[k] = (float)(px / lim_x);
        fy[k] = (float)(py / lim_y);
        out.poly[2 * k] = fx[k];
        out.poly[2 * k + 1] = fy[k];
    }
    const float xmin = fminf(fminf(fx[0], fx[1]), fminf(fx[2], fx[3]));
    const float xmax = fmaxf(fmaxf(fx[0], fx[1]), fmaxf(fx[2], fx[3]));
    const float ymin = fminf(fminf(fy[0], fy[1]), fminf(fy[2], fy[3]));
    const float ymax = fmaxf(fmaxf(fy[0], fy[1]), fmaxf(fy[2], fy[3]));
    out.xyxy[0] = xmin;
    out.xyxy[1] = ymin;
    out.xyxy[2] = xmin + (xmax - xmin);  // the reference rebuilds the far corner from w, h
    out.xyxy[3] = ymin + (ymax - ymin);
    out.score = m[0];
    return outside <= 1;
}

__device__ __forceinline__ void load_pixel(const float* __restrict__ maps, int64_t pixel, float m[6]) {
    const float2* p = reinterpret_cast<const float2*>(maps + pixel * 6);  // 24-byte records, 8-byte aligned
    const float2 a = p[0], b = p[1], c = p[2];
    m[0] = a.x; m[1] = a.y; m[2] = b.x; m[3] = b.y; m[4] = c.x; m[5] = c.y;
}

// grid (nchunks, B), block 256.  masks: (B, nchunks, 4) uint64, chunk_counts: (B, nchunks) int32.
__global__ __launch_bounds__(kChunk) void east_decode_flags(const float* __restrict__ maps, int mh, int mw, int in_h,
                                                            int in_w, float thr, unsigned long long* __restrict__ masks,
                                                            int* __restrict__ chunk_counts) {
    __shared__ int wave_count[4];
    const int npix = mh * mw;
    const int chunk = blockIdx.x, b = blockIdx.y, nchunks = gridDim.x;
    const int p = chunk * kChunk + threadIdx.x;
    bool valid = false;
    if (p < npix) {
        float m[6];
        load_pixel(maps, (int64_t)b * npix + p, m);
        if (m[0] > thr) {
            DecodedBox box;
            valid = decode_pixel(m, p % mw, p / mw, in_w / mw, in_h / mh, in_w, in_h, box);
        }
    }
    const unsigned long long ballot = __ballot(valid);
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    if (lane == 0) {
        masks[((int64_t)b * nchunks + chunk) * 4 + wave] = ballot;
        wave_count[wave] = __popcll(ballot);
    }
    __syncthreads();
    if (threadIdx.x == 0)
        chunk_counts[(int64_t)b * nchunks + chunk] = wave_count[0] + wave_count[1] + wave_count[2] + wave_count[3];
}

// grid (nchunks, B), block 256.
__global__ __launch_bounds__(kChunk) void east_decode_write(const float* __restrict__ maps, int mh, int mw, int in_h,
                                                            int in_w, const unsigned long long* __restrict__ masks,
                                                            const int* __restrict__ chunk_counts, int cap,
                                                            float* __restrict__ xyxy, float* __restrict__ polys,
                                                            float* __restrict__ scores, int* __restrict__ pix,
                                                            int* __restrict__ counts) {
    __shared__ int partial[4];
    __shared__ int base_s;
    const int npix = mh * mw;
    const int chunk = blockIdx.x, b = blockIdx.y, nchunks = gridDim.x;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;

    // exclusive prefix of the chunk counts of this image
    int acc = 0;
    for (int c = threadIdx.x; c < chunk; c += kChunk) acc += chunk_counts[(int64_t)b * nchunks + c];
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) acc += __shfl_down(acc, off, 64);
    if (lane == 0) partial[wave] = acc;
    __syncthreads();
    if (threadIdx.x == 0) {
        const int base = partial[0] + partial[1] + partial[2] + partial[3];
        base_s = base;
        if (chunk == nchunks - 1) counts[b] = base + chunk_counts[(int64_t)b * nchunks + chunk];
    }
    __syncthreads();

    const unsigned long long* mrow = masks + ((int64_t)b * nchunks + chunk) * 4;
    int pos = base_s;
    for (int w = 0; w < wave; ++w) pos += __popcll(mrow[w]);
    const unsigned long long mine = mrow[wave];
    if (!((mine >> lane) & 1ull)) return;
    pos += __popcll(mine & ((1ull << lane) - 1ull));
    if (pos >= cap) return;

    const int p = chunk * kChunk + threadIdx.x;
    float m[6];
    load_pixel(maps, (int64_t)b * npix + p, m);
    DecodedBox box;
    decode_pixel(m, p % mw, p / mw, in_w / mw, in_h / mh, in_w, in_h, box);

    const int64_t o = (int64_t)b * cap + pos;
    *reinterpret_cast<float4*>(xyxy + o * 4) = make_float4(box.xyxy[0], box.xyxy[1], box.xyxy[2], box.xyxy[3]);
    scores[o] = box.score;
    if (polys) {
        float4* pp = reinterpret_cast<float4*>(polys + o * 8);
        pp[0] = make_float4(box.poly[0], box.poly[1], box.poly[2], box.poly[3]);
        pp[1] = make_float4(box.poly[4], box.poly[5], box.poly[6], box.poly[7]);
    }
    if (pix) pix[o] = p;
}

}  // namespace

// Checks the arguments both entry points share and sizes the staging area: (B, nchunks, 4) 64-bit masks followed
// by (B, nchunks) int32 counts.
static int decode_check(const float* maps, int B, int mh, int mw, int in_h, int in_w, int cap, const float* xyxy,
                        const float* scores, const int* counts, int* nchunks, size_t* mask_bytes, size_t* count_bytes) {
    OCR_REQUIRE(maps && xyxy && scores && counts, "east_decode: maps, xyxy, scores and counts are required");
    OCR_REQUIRE(B >= 0 && mh > 0 && mw > 0 && cap > 0, "east_decode: bad sizes B=%d mh=%d mw=%d cap=%d", B, mh, mw, cap);
    OCR_REQUIRE(in_h >= mh && in_w >= mw && in_h % mh == 0 && in_w % mw == 0,
                "east_decode: input %dx%d is not an integer multiple of the map %dx%d", in_h, in_w, mh, mw);
    OCR_REQUIRE((int64_t)mh * mw < (1ll << 30), "east_decode: map too large");
    const int npix = mh * mw;
    *nchunks = (npix + kChunk - 1) / kChunk;
    *mask_bytes = (size_t)B * *nchunks * 4 * sizeof(unsigned long long);
    *count_bytes = (size_t)B * *nchunks * sizeof(int);
    return OCRHIP_OK;
}

// Both kernels over a staging area of mask_bytes + count_bytes bytes.  The first kernel writes every mask and
// every count the second one reads, so the staging area needs no initialisation.
static int decode_enqueue(void* stage, size_t mask_bytes, int nchunks, const float* maps, int B, int mh, int mw, int in_h,
                          int in_w, float thr, int cap, float* xyxy, float* polys, float* scores, int* pix, int* counts,
                          hipStream_t s) {
    auto* masks = reinterpret_cast<unsigned long long*>(stage);
    int* chunk_counts = reinterpret_cast<int*>(reinterpret_cast<char*>(stage) + mask_bytes);
    dim3 grid(nchunks, B), block(kChunk);
    hipLaunchKernelGGL(east_decode_flags, grid, block, 0, s, maps, mh, mw, in_h, in_w, thr, masks, chunk_counts);
    hipLaunchKernelGGL(east_decode_write, grid, block, 0, s, maps, mh, mw, in_h, in_w, masks, chunk_counts, cap, xyxy,
                       polys, scores, pix, counts);
    OCR_HIP_CHECK(hipGetLastError());
    return OCRHIP_OK;
}

int launch_east_decode(ocrhip_ctx* h, const float* maps, int B, int mh, int mw, int in_h, int in_w, float thr,
                       int cap, float* xyxy, float* polys, float* scores, int* pix, int* counts, hipStream_t s) {
    int nchunks = 0;
    size_t mask_bytes = 0, count_bytes = 0;
    OCR_TRY(decode_check(maps, B, mh, mw, in_h, in_w, cap, xyxy, scores, counts, &nchunks, &mask_bytes, &count_bytes));
    if (B == 0) return OCRHIP_OK;
    OCR_TRY(h->decode_stage.reserve(mask_bytes + count_bytes));
    // algorithmic bytes: the 24 B/pixel map; the 36 B (+32 B polygon) per box depend on the data
    ProfScope prof("east_decode", 0.0, 24.0 * (double)B * mh * mw, s);
    return decode_enqueue(h->decode_stage.ptr, mask_bytes, nchunks, maps, B, mh, mw, in_h, in_w, thr, cap, xyxy, polys,
                          scores, pix, counts, s);
}

// ---- the same decode for callers that must not allocate ---------------------------------------------------------
// The staging area is a fixed array in device memory: every GPU the library runs on holds its own copy at its own
// address until the process ends.  kStaticChunks chunks of 256 map pixels = 8.4 M map pixels per call (128 pages of
// 1024 x 1024, whose maps are 256 x 256), 36 bytes of staging per chunk.
constexpr int kStaticChunks = 32768;
constexpr size_t kStaticBytes = (size_t)kStaticChunks * (4 * sizeof(unsigned long long) + sizeof(int));
static __device__ __align__(256) unsigned char scratch[kStaticBytes];

constexpr int kMaxDevices = 64;
static void* g_scratch_ptr[kMaxDevices];          // device address of `scratch` on each GPU, set by decode_static_init
static std::mutex g_scratch_mutex;

// Called by ocrhip_create with `device` current: looks up the address of this GPU's copy of the array once.
int decode_static_init(int device) {
    OCR_REQUIRE(device >= 0 && device < kMaxDevices, "create: device index %d is beyond the %d this library tracks", device, kMaxDevices);
    std::lock_guard<std::mutex> lock(g_scratch_mutex);
    if (g_scratch_ptr[device]) return OCRHIP_OK;
    void* ptr = nullptr;
    OCR_HIP_CHECK(hipGetSymbolAddress(&ptr, HIP_SYMBOL(scratch)));
    OCR_REQUIRE(ptr != nullptr, "create: the static staging array has no address on device %d", device);
    g_scratch_ptr[device] = ptr;
    return OCRHIP_OK;
}

int launch_east_decode_static(ocrhip_ctx* h, const float* maps, int B, int mh, int mw, int in_h, int in_w, float thr,
                              int cap, float* xyxy, float* polys, float* scores, int* pix, int* counts, hipStream_t s) {
    int nchunks = 0;
    size_t mask_bytes = 0, count_bytes = 0;
    OCR_TRY(decode_check(maps, B, mh, mw, in_h, in_w, cap, xyxy, scores, counts, &nchunks, &mask_bytes, &count_bytes));
    if ((int64_t)B * nchunks > kStaticChunks) {
        set_error("east_decode_static: %d maps of %dx%d need %lld chunks of %d pixels, the static staging area holds %d "
                  "(use ocrhip_east_decode or fewer pages per call)", B, mh, mw, (long long)B * nchunks, kChunk, kStaticChunks);
        return OCRHIP_ECAP;
    }
    if (B == 0) return OCRHIP_OK;
    void* stage = nullptr;
    {
        std::lock_guard<std::mutex> lock(g_scratch_mutex);
        stage = (h->device >= 0 && h->device < kMaxDevices) ? g_scratch_ptr[h->device] : nullptr;
    }
    if (!stage) {
        set_error("east_decode_static: the static staging area of device %d was not set up", h->device);
        return OCRHIP_ESTATE;
    }
    return decode_enqueue(stage, mask_bytes, nchunks, maps, B, mh, mw, in_h, in_w, thr, cap, xyxy, polys, scores, pix,
                          counts, s);
}

}  // namespace ocrhip
