// Direct (VALU) convolutions for the layers whose reduction depth is too small for the matrix
// cores: the first 3x3 convolution of EAST (3 -> 64, fused with the `/255`, `(x-0.5)/0.5`
// normalisation when the input is uint8), the first 3x3 convolution of the CRNN (1 -> 32, fused
// with its 2x2 max-pool) and the three 1x1 sigmoid heads of EAST (32 -> 1 + 4 + 1).
//
// Restates architectures/east_arch.py:121-129,153-162, architectures/crnn_arch.py:38-41,459-473 and
// utils/image/image_normalization.py:57 (+ utils/keras/ops/core.py:271).  All three are HBM bound:
// the first EAST layer writes 256 B per input pixel, so outputs go through an LDS transpose and
// leave as full 16-byte-per-lane coalesced rows.
#include "common.h"

namespace ocrhip {

namespace {

// One wave = 64 consecutive output pixels of one row; the 4 waves of a workgroup own COUT/4 output
// channels each.  Weights are indexed by wave-uniform expressions, so they travel through scalar
// registers.  Layout of w: [ky][kx][cin][cout].
template <int CIN, int COUT, bool POOL, bool U8>
__global__ __launch_bounds__(256) void conv_first_kernel(const void* __restrict__ xin, const float* __restrict__ w,
                                                         const float* __restrict__ bias, float* __restrict__ y,
                                                         int H, int W) {
    constexpr int CPW = COUT / 4;               // channels per wave
    constexpr int LDS_STRIDE = COUT + 4;
    __shared__ float tile[64 * LDS_STRIDE];

    const int lane = threadIdx.x & 63;
    const int g = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int Ho = POOL ? H / 2 : H, Wo = POOL ? W / 2 : W;
    const int ox0 = blockIdx.x * 64, oy = blockIdx.y, b = blockIdx.z;
    const int ox = ox0 + lane;

    constexpr int P = POOL ? 2 : 1;             // pre-pool pixels per axis
    constexpr int PATCH = P + 2;
    float in[PATCH][PATCH][CIN];
    const int iy0 = oy * P - 1, ix0 = ox * P - 1;
#pragma unroll
    for (int dy = 0; dy < PATCH; ++dy) {
#pragma unroll
        for (int dx = 0; dx < PATCH; ++dx) {
            const int yy = iy0 + dy, xx = ix0 + dx;
            const bool ok = (yy >= 0) && (yy < H) && (xx >= 0) && (xx < W);
#pragma unroll
            for (int c = 0; c < CIN; ++c) {
                float v = 0.0f;
                if (ok) {
                    const int64_t idx = (((int64_t)b * H + yy) * W + xx) * CIN + c;
                    if (U8) {
                        const float u = (float)reinterpret_cast<const uint8_t*>(xin)[idx];
                        v = ((u / 255.0f) - 0.5f) / 0.5f;
                    } else {
                        v = reinterpret_cast<const float*>(xin)[idx];
                    }
                }
                in[dy][dx][c] = v;
            }
        }
    }

    float acc[P][P][CPW];
#pragma unroll
    for (int py = 0; py < P; ++py)
#pragma unroll
        for (int px = 0; px < P; ++px)
#pragma unroll
            for (int o = 0; o < CPW; ++o) acc[py][px][o] = 0.0f;

#pragma unroll
    for (int ky = 0; ky < 3; ++ky)
#pragma unroll
        for (int kx = 0; kx < 3; ++kx)
#pragma unroll
            for (int c = 0; c < CIN; ++c) {
                const float* wp = w + ((ky * 3 + kx) * CIN + c) * COUT + g * CPW;
#pragma unroll
                for (int o = 0; o < CPW; ++o) {
                    const float wv = wp[o];
#pragma unroll
                    for (int py = 0; py < P; ++py)
#pragma unroll
                        for (int px = 0; px < P; ++px)
                            acc[py][px][o] = fmaf(in[py + ky][px + kx][c], wv, acc[py][px][o]);
                }
            }

#pragma unroll
    for (int o = 0; o < CPW; ++o) {
        float v = acc[0][0][o];
        if (POOL) v = fmaxf(fmaxf(v, acc[0][P - 1][o]), fmaxf(acc[P - 1][0][o], acc[P - 1][P - 1][o]));
        v = fmaxf(v + bias[g * CPW + o], 0.0f);
        tile[lane * LDS_STRIDE + g * CPW + o] = v;
    }
    __syncthreads();

    // 64 pixels x COUT channels are contiguous in NHWC: store them as float4 rows
    const int npx = min(64, Wo - ox0);
    float* dst = y + (((int64_t)b * Ho + oy) * Wo + ox0) * COUT;
    constexpr int V4_PER_PX = COUT / 4;
    for (int i = threadIdx.x; i < npx * V4_PER_PX; i += 256) {
        const int px = i / V4_PER_PX, c4 = i % V4_PER_PX;
        const float4 v = *reinterpret_cast<const float4*>(&tile[px * LDS_STRIDE + c4 * 4]);
        reinterpret_cast<float4*>(dst)[i] = v;
    }
}

// The unpooled first layer on large images: a workgroup walks ROWS output rows of its 64 columns.  All ROWS + 2 input
// rows are requested at once (one memory round trip per workgroup instead of one per row, and a third of the loads:
// neighbouring rows share two of their three input rows), the weights stay in scalar registers, the LDS transpose is
// double buffered (one barrier per row).  Same fma order per output as conv_first_kernel.
template <int CIN, int COUT, bool U8, int ROWS>
__global__ __launch_bounds__(256) void conv_first_rows_kernel(const void* __restrict__ xin, const float* __restrict__ w,
                                                              const float* __restrict__ bias, float* __restrict__ y,
                                                              int H, int W) {
    constexpr int CPW = COUT / 4;
    constexpr int LDS_STRIDE = COUT + 4;
    __shared__ float tile[2][64 * LDS_STRIDE];
    // uint8 input: the 256 possible normalised values, computed once per workgroup with the reference's float32
    // operations (a division is ten instructions; converting a pixel's inputs in place costs a quarter of its FMAs)
    __shared__ float lut[U8 ? 256 : 1];
    if (U8) {
        lut[threadIdx.x] = (((float)threadIdx.x / 255.0f) - 0.5f) / 0.5f;
        __syncthreads();
    }

    const int lane = threadIdx.x & 63;
    const int g = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int ox0 = blockIdx.x * 64, oy0 = blockIdx.y * ROWS, b = blockIdx.z;
    const int ox = ox0 + lane;

    float in[ROWS + 2][3][CIN];
#pragma unroll
    for (int dy = 0; dy < ROWS + 2; ++dy) {
#pragma unroll
        for (int dx = 0; dx < 3; ++dx) {
            const int yy = oy0 - 1 + dy, xx = ox - 1 + dx;
            const bool ok = (yy >= 0) && (yy < H) && (xx >= 0) && (xx < W);
#pragma unroll
            for (int c = 0; c < CIN; ++c) {
                float v = 0.0f;
                if (ok) {
                    const int64_t idx = (((int64_t)b * H + yy) * W + xx) * CIN + c;
                    if (U8) {
                        v = lut[reinterpret_cast<const uint8_t*>(xin)[idx]];
                    } else {
                        v = reinterpret_cast<const float*>(xin)[idx];
                    }
                }
                in[dy][dx][c] = v;
            }
        }
    }
    float bv[CPW];
#pragma unroll
    for (int o = 0; o < CPW; ++o) bv[o] = bias[g * CPW + o];

    const int npx = min(64, W - ox0);
    constexpr int V4_PER_PX = COUT / 4;
#pragma unroll
    for (int r = 0; r < ROWS; ++r) {
        const int oy = oy0 + r;
        if (oy >= H) break;                     // uniform
        float acc[CPW];
#pragma unroll
        for (int o = 0; o < CPW; ++o) acc[o] = 0.0f;
#pragma unroll
        for (int ky = 0; ky < 3; ++ky)
#pragma unroll
            for (int kx = 0; kx < 3; ++kx)
#pragma unroll
                for (int c = 0; c < CIN; ++c) {
                    const float* wp = w + ((ky * 3 + kx) * CIN + c) * COUT + g * CPW;
#pragma unroll
                    for (int o = 0; o < CPW; ++o) acc[o] = fmaf(in[r + ky][kx][c], wp[o], acc[o]);
                }
        float* t = tile[r & 1];
#pragma unroll
        for (int o = 0; o < CPW; ++o) t[lane * LDS_STRIDE + g * CPW + o] = fmaxf(acc[o] + bv[o], 0.0f);
        __syncthreads();                        // the other buffer was read before the previous row's barrier
        float* dst = y + (((int64_t)b * H + oy) * W + ox0) * COUT;
        for (int i = threadIdx.x; i < npx * V4_PER_PX; i += 256) {
            const int px = i / V4_PER_PX, c4 = i % V4_PER_PX;
            reinterpret_cast<float4*>(dst)[i] = *reinterpret_cast<const float4*>(&t[px * LDS_STRIDE + c4 * 4]);
        }
    }
}

// The unpooled 3 -> 64 first layer on the matrix cores.  K = 27 is far too shallow for the tiled kernels (their chunks are
// 8 or 16 channels deep), but per output the f32 MFMA still does twice the work of a vector fma per cycle: a wave
// multiplies 32 pixels x 28 (27 + one zero) by 28 x 64 with 2 x 14 v_mfma_f32_32x32x2_f32, the weights held in 28
// registers for the whole workgroup.  A workgroup owns 8 rows x 64 columns: its 10 x 66 input pixels are normalised once
// (uint8 input: through the 256-entry table of the reference's float32 results) into LDS, from where a lane reads the 14
// neighbour values of its pixel that fall on its half of k; each wave turns 32 pixels x 64 channels at a time
// (4 units of one row half), adds bias, applies ReLU, transposes through its own LDS area (pixel stride 72 floats:
// conflict free both ways) and stores the 8 KB as 16-byte-per-lane coalesced rows.  Products are exact and summed in
// float32 like the vector kernel's, in pairs of k instead of one by one.
typedef float floatx16_cf __attribute__((ext_vector_type(16)));

__device__ __forceinline__ unsigned cf_bf16_pair(float lo, float hi) {
    const __bf16 a = (__bf16)lo, b = (__bf16)hi;
    return (unsigned)__builtin_bit_cast(unsigned short, a) | ((unsigned)__builtin_bit_cast(unsigned short, b) << 16);
}

// OUT16: the output tensor is bfloat16 (the bfloat16-tensor path): same sums, rounded once on the way out.
template <bool U8, bool OUT16>
__global__ __launch_bounds__(256) void conv_first_mfma_kernel(const void* __restrict__ xin, const float* __restrict__ w,
                                                              const float* __restrict__ bias, void* __restrict__ yout,
                                                              int H, int W) {
    float* y = reinterpret_cast<float*>(yout);
    constexpr int ROWS = 8, IW = 66, IN_FLOATS = (ROWS + 2) * IW * 3, TS = 72;
    __shared__ float s_in[IN_FLOATS + 4];
    __shared__ __attribute__((aligned(16))) float s_t[4][32 * TS];
    __shared__ float lut[U8 ? 256 : 1];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int m = lane & 31, lh = lane >> 5;
    const int ox0 = blockIdx.x * 64, oy0 = blockIdx.y * ROWS, b = blockIdx.z;

    if (U8) {
        lut[tid] = (((float)tid / 255.0f) - 0.5f) / 0.5f;
        __syncthreads();
    }
    // input tile: rows oy0 - 1 .. oy0 + 8, columns ox0 - 1 .. ox0 + 64, zero outside the image (the padding of the
    // normalised tensor); consecutive elements of a row are consecutive in memory
    // (all loads of the thread first: one memory round trip)
    constexpr int NLD = (IN_FLOATS + 255) / 256;
    float ld[NLD];
    int raw[NLD];
#pragma unroll
    for (int j = 0; j < NLD; ++j) {
        const int e = tid + 256 * j;
        const int r = e / (IW * 3), rem = e % (IW * 3);
        const int yy = oy0 - 1 + r, xx = ox0 - 1 + rem / 3;
        ld[j] = 0.0f;
        raw[j] = -1;
        if (e < IN_FLOATS && yy >= 0 && yy < H && xx >= 0 && xx < W) {
            const int64_t idx = (((int64_t)b * H + yy) * W + ox0 - 1) * 3 + rem;
            if (U8) raw[j] = reinterpret_cast<const uint8_t*>(xin)[idx];
            else ld[j] = reinterpret_cast<const float*>(xin)[idx];
        }
    }
#pragma unroll
    for (int j = 0; j < NLD; ++j) {
        const int e = tid + 256 * j;
        if (U8 && raw[j] >= 0) ld[j] = lut[raw[j]];
        if (e < IN_FLOATS) s_in[e] = ld[j];
    }
    // weights [k = (ky * 3 + kx) * 3 + c][64]: step s uses k = 2 s + lh, the 28th k is zero
    float bw[14][2];
#pragma unroll
    for (int st = 0; st < 14; ++st) {
        const int k = 2 * st + lh;
#pragma unroll
        for (int nt = 0; nt < 2; ++nt) bw[st][nt] = (k < 27) ? w[k * 64 + nt * 32 + m] : 0.0f;
    }
    const float bias0 = bias[m], bias1 = bias[32 + m];
    // LDS offsets of the lane's 14 neighbour values relative to (row r, column half h): k -> (ky, kx, c)
    int aoff[14];
#pragma unroll
    for (int st = 0; st < 14; ++st) {
        const int k = min(2 * st + lh, 26);
        aoff[st] = ((k / 9) * IW + (k / 3) % 3 + m) * 3 + k % 3;
    }
    __syncthreads();

    // the workgroup's output rows start at the descriptor's base: one lane offset for all stores, the unit in the scalar offset
    const int64_t first = (((int64_t)b * H + oy0) * W + ox0) * 64;
    const __amdgpu_buffer_rsrc_t y_rsrc = __builtin_amdgcn_make_buffer_rsrc(
        OUT16 ? (void*)(reinterpret_cast<uint16_t*>(yout) + first) : (void*)(y + first), 0, 0x7fffffff, 0x00020000);
    float* t = s_t[wave];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
        const int r = wave * 2 + (u >> 1), h = u & 1;              // row of the tile, half of its 64 columns
        const bool row_ok = oy0 + r < H;                           // uniform per wave
        const float* src = s_in + (r * IW + h * 32) * 3;
        floatx16_cf acc0, acc1;
#pragma unroll
        for (int v = 0; v < 16; ++v) { acc0[v] = 0.0f; acc1[v] = 0.0f; }
        // all 14 operands first, then the 28 matrix instructions back to back
        float av[14];
#pragma unroll
        for (int st = 0; st < 14; ++st) av[st] = src[aoff[st]];
        av[13] = lh ? 0.0f : av[13];
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int st = 0; st < 14; ++st) {
            acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(av[st], bw[st][0], acc0, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(av[st], bw[st][1], acc1, 0, 0, 0);
        }
        __builtin_amdgcn_sched_barrier(0);
        // register v of a lane: pixel (v & 3) + 8 (v >> 2) + 4 lh of the 32, channel m (+ 32)
#pragma unroll
        for (int v = 0; v < 16; ++v) {
            const int px = (v & 3) + 8 * (v >> 2) + 4 * lh;
            t[px * TS + m] = fmaxf(acc0[v] + bias0, 0.0f);
            t[px * TS + 32 + m] = fmaxf(acc1[v] + bias1, 0.0f);
        }
        // the wave's own transpose: LDS works off a wave's accesses in order, the fences keep the compiler from moving
        // the 16-byte reads (another type than the float writes) across them
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        const int npx = row_ok ? min(32, W - ox0 - h * 32) : 0;    // valid pixels of this half (<= 0: none)
        if constexpr (OUT16) {
            const int unit_off = (r * W + h * 32) * 128;           // bytes from the workgroup's first pixel
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int i = lane + 64 * j;                       // 16-byte piece within the unit: pixel i / 8, channel octet i % 8
                const float4 q0 = *reinterpret_cast<const float4*>(&t[(i >> 3) * TS + (i & 7) * 8]);
                const float4 q1 = *reinterpret_cast<const float4*>(&t[(i >> 3) * TS + (i & 7) * 8 + 4]);
                const unsigned p0 = cf_bf16_pair(q0.x, q0.y), p1 = cf_bf16_pair(q0.z, q0.w), p2 = cf_bf16_pair(q1.x, q1.y), p3 = cf_bf16_pair(q1.z, q1.w);
                buffer_store_b128_guarded(ocr_floatx4{__builtin_bit_cast(float, p0), __builtin_bit_cast(float, p1), __builtin_bit_cast(float, p2),
                                                      __builtin_bit_cast(float, p3)},
                                          y_rsrc, (i >> 3) < npx ? i * 16 : (int)0x80000000, unit_off);
            }
        } else {
        const int unit_off = (r * W + h * 32) * 256;               // bytes from the workgroup's first pixel
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int i = lane + 64 * j;                           // float4 index within the unit: pixel i / 16, channel quad i % 16
            const float4 q = *reinterpret_cast<const float4*>(&t[(i >> 4) * TS + (i & 15) * 4]);
            buffer_store_b128_guarded(ocr_floatx4{q.x, q.y, q.z, q.w}, y_rsrc, (i >> 4) < npx ? i * 16 : (int)0x80000000, unit_off);
        }
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    }
}

// The recogniser's first layer (1 -> 32 channels, 3x3, ReLU, 2x2 max-pool) on the matrix cores.  K = 9 (+ one zero) is five
// v_mfma_f32_32x32x2_f32 for 32 pre-pool pixels x 32 channels.  The 32 rows of a unit are 8 pooled pixels of one pooled
// row, row i = 4 g + s being sub-pixel s = 2 dy + dx of pooled pixel g: the four pre-pool results of a pooled pixel
// then sit in registers 4 q .. 4 q + 3 of ONE lane (accumulator register v of a lane holds row (v & 3) + 8 (v >> 2) +
// 4 (lane >> 5)), so the pooling is three v_max in registers, and for a fixed q the lanes of a wave hold 2 pixels x 32
// consecutive channels = 256 contiguous bytes of the NHWC output: plain coalesced stores, no transpose.  A workgroup
// owns 8 pooled rows x 64 pooled columns of one image: its 18 x 130 input values go through LDS once.  Same products,
// taps summed in the order of conv_first_kernel (the matrix instruction adds two per step), max and bias + ReLU in the
// same order: bit-identical on gfx950, which the tests pin; the instruction's internal order does not guarantee it.
template <bool OUT16>
__global__ __launch_bounds__(256) void conv_first_pool_mfma_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                                   const float* __restrict__ bias, void* __restrict__ yout,
                                                                   int H, int W) {
    float* y = reinterpret_cast<float*>(yout);
    constexpr int PR = 8, PC = 64;                     // pooled rows x columns of the workgroup
    constexpr int IH = 2 * PR + 2, IW = 2 * PC + 2;    // input rows x columns incl. the border
    __shared__ float s_in[IH * IW + 4];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int i = lane & 31, lh = lane >> 5;
    const int Ho = H / 2, Wo = W / 2;
    const int xo0 = blockIdx.x * PC, yo0 = blockIdx.y * PR, b = blockIdx.z;
    const float* img = x + (int64_t)b * H * W;

    // all loads of the thread first (one memory round trip), then the LDS writes
    constexpr int NLD = (IH * IW + 255) / 256;
    float ld[NLD];
#pragma unroll
    for (int j = 0; j < NLD; ++j) {
        const int e = tid + 256 * j;
        const int r = e / IW, c = e % IW;
        const int yy = 2 * yo0 - 1 + r, xx = 2 * xo0 - 1 + c;
        ld[j] = (e < IH * IW && yy >= 0 && yy < H && xx >= 0 && xx < W) ? img[(int64_t)yy * W + xx] : 0.0f;
    }
#pragma unroll
    for (int j = 0; j < NLD; ++j)
        if (tid + 256 * j < IH * IW) s_in[tid + 256 * j] = ld[j];
    // weights [k = ky * 3 + kx][32]: step st uses k = 2 st + lh, the 10th k is zero
    float bw[5];
#pragma unroll
    for (int st = 0; st < 5; ++st) {
        const int k = 2 * st + lh;
        bw[st] = (k < 9) ? w[k * 32 + i] : 0.0f;
    }
    const float bv = bias[i];
    // LDS offsets of the lane's five input values relative to (pooled row r, unit column u): row i = 4 g + s of the unit is
    // pre-pool pixel (2 r + (s >> 1), 16 u + 2 g + (s & 1)); k = (ky, kx) reads input (row + ky - 1, column + kx - 1), and
    // the LDS image starts at row -1, column -1
    const int g = i >> 2, sub = i & 3;
    int aoff[5];
#pragma unroll
    for (int st = 0; st < 5; ++st) {
        const int k = min(2 * st + lh, 8);
        aoff[st] = ((sub >> 1) + k / 3) * IW + 2 * g + (sub & 1) + k % 3;
    }
    __syncthreads();

    // units of the wave: pooled rows 2 wave, 2 wave + 1, eight units of 8 pooled columns each
    for (int u = 0; u < 16; ++u) {
        const int r = wave * 2 + (u >> 3), uc = u & 7;
        const int yo = yo0 + r, xo = xo0 + uc * 8;
        if (yo >= Ho || xo >= Wo) continue;               // uniform
        const float* src = s_in + (2 * r) * IW + 16 * uc;
        float av[5];
#pragma unroll
        for (int st = 0; st < 5; ++st) av[st] = src[aoff[st]];
        av[4] = lh ? 0.0f : av[4];
        floatx16_cf acc;
#pragma unroll
        for (int v = 0; v < 16; ++v) acc[v] = 0.0f;
#pragma unroll
        for (int st = 0; st < 5; ++st) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av[st], bw[st], acc, 0, 0, 0);
        float* dst = y + (((int64_t)b * Ho + yo) * Wo + xo) * 32 + i;
        unsigned* dst16 = reinterpret_cast<unsigned*>(reinterpret_cast<uint16_t*>(yout) + (((int64_t)b * Ho + yo) * Wo + xo) * 32 + i);
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            float v = fmaxf(fmaxf(acc[4 * q], acc[4 * q + 1]), fmaxf(acc[4 * q + 2], acc[4 * q + 3]));
            v = fmaxf(v + bv, 0.0f);
            const int px = 2 * q + lh;                      // pooled pixel of the unit
            if constexpr (OUT16) {
                const float other = __shfl_xor(v, 1);       // channel i ^ 1 of the same pixel: even lanes store the pair
                if (xo + px < Wo && (i & 1) == 0) dst16[px * 16] = cf_bf16_pair(v, other);
            } else {
                if (xo + px < Wo) dst[px * 32] = v;
            }
        }
    }
}

// 1x1 heads on the 32-channel merged map: out = sigmoid(W x + b), 6 channels.  w: [6][32].
__global__ __launch_bounds__(256) void east_heads_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                         const float* __restrict__ bias, float* __restrict__ maps,
                                                         int64_t npix) {
    const int64_t p = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (p >= npix) return;
    float v[32];
    const float4* xp = reinterpret_cast<const float4*>(x + p * 32);
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const float4 q = xp[i];
        v[4 * i] = q.x; v[4 * i + 1] = q.y; v[4 * i + 2] = q.z; v[4 * i + 3] = q.w;
    }
    float out[6];
#pragma unroll
    for (int o = 0; o < 6; ++o) {
        float acc = 0.0f;
#pragma unroll
        for (int c = 0; c < 32; ++c) acc = fmaf(v[c], w[o * 32 + c], acc);
        acc += bias[o];
        out[o] = 1.0f / (1.0f + expf(-acc));
    }
    float2* op = reinterpret_cast<float2*>(maps + p * 6);
    op[0] = make_float2(out[0], out[1]);
    op[1] = make_float2(out[2], out[3]);
    op[2] = make_float2(out[4], out[5]);
}

}  // namespace

int prepare_conv_direct(ConvLayer& L, const float* w_oihw, const float* bias, int cout, int cin, int ks) {
    L.cin = cin; L.cout = cout; L.ks = ks; L.cout_pad = cout; L.direct = true;
    std::vector<float> w((size_t)ks * ks * cin * cout);
    for (int o = 0; o < cout; ++o)
        for (int c = 0; c < cin; ++c)
            for (int ky = 0; ky < ks; ++ky)
                for (int kx = 0; kx < ks; ++kx)
                    w[((size_t)(ky * ks + kx) * cin + c) * cout + o] = w_oihw[(((size_t)o * cin + c) * ks + ky) * ks + kx];
    OCR_TRY(L.w_f32.reserve(w.size() * sizeof(float)));
    OCR_TRY(L.bias.reserve((size_t)cout * sizeof(float)));
    OCR_HIP_CHECK(hipMemcpy(L.w_f32.ptr, w.data(), w.size() * sizeof(float), hipMemcpyHostToDevice));
    OCR_HIP_CHECK(hipMemcpy(L.bias.ptr, bias, (size_t)cout * sizeof(float), hipMemcpyHostToDevice));
    return OCRHIP_OK;
}

int launch_conv_first(const ConvLayer& L, const void* x, bool x_is_u8, void* y, int B, int H, int W, int pool,
                      hipStream_t s, bool out_bf16) {
    OCR_REQUIRE(L.direct && L.ks == 3, "conv_first: layer is not a direct 3x3 layer");
    OCR_REQUIRE(B > 0 && H > 0 && W > 0 && B <= 65535 && H <= 65535, "conv_first: bad sizes B=%d H=%d W=%d", B, H, W);
    const float* w = L.w_f32.as<float>();
    const float* bias = L.bias.as<float>();
    const double px = (double)B * H * W;
    ProfScope prof(L.cin == 3 ? "conv_first<3,64>" : (pool ? "conv_first<1,32,pool>" : "conv_first<1,32>"), 2.0 * 9 * L.cin * L.cout * px,
                   px * L.cin * (x_is_u8 ? 1.0 : 4.0) + 4.0 * px * L.cout / (pool ? 4.0 : 1.0), s);
    if (out_bf16) {          // the bfloat16-tensor path: matrix-core kernels only
        if (L.cin == 3 && L.cout == 64 && pool == 0 && (int64_t)8 * W * 256 < (1ll << 31)) {
            dim3 grid((W + 63) / 64, (H + 7) / 8, B), block(256);
            if (x_is_u8)
                hipLaunchKernelGGL((conv_first_mfma_kernel<true, true>), grid, block, 0, s, x, w, bias, y, H, W);
            else
                hipLaunchKernelGGL((conv_first_mfma_kernel<false, true>), grid, block, 0, s, x, w, bias, y, H, W);
        } else if (L.cin == 1 && L.cout == 32 && pool == 1 && !x_is_u8 && H % 2 == 0 && W % 2 == 0) {
            dim3 grid((W / 2 + 63) / 64, (H / 2 + 7) / 8, B), block(256);
            hipLaunchKernelGGL(conv_first_pool_mfma_kernel<true>, grid, block, 0, s, reinterpret_cast<const float*>(x), w, bias, y, H, W);
        } else {
            set_error("conv_first: no bfloat16-output kernel for cin=%d cout=%d pool=%d", L.cin, L.cout, pool);
            return OCRHIP_EINVAL;
        }
        OCR_HIP_CHECK(hipGetLastError());
        return OCRHIP_OK;
    }
    if (L.cin == 3 && L.cout == 64 && pool == 0) {
        dim3 grid((W + 63) / 64, (H + 7) / 8, B), block(256);
        if (first_layer_mfma() && (int64_t)8 * W * 256 < (1ll << 31)) {
            if (x_is_u8)
                hipLaunchKernelGGL((conv_first_mfma_kernel<true, false>), grid, block, 0, s, x, w, bias, y, H, W);
            else
                hipLaunchKernelGGL((conv_first_mfma_kernel<false, false>), grid, block, 0, s, x, w, bias, y, H, W);
        } else if (x_is_u8)
            hipLaunchKernelGGL((conv_first_rows_kernel<3, 64, true, 8>), grid, block, 0, s, x, w, bias, reinterpret_cast<float*>(y), H, W);
        else
            hipLaunchKernelGGL((conv_first_rows_kernel<3, 64, false, 8>), grid, block, 0, s, x, w, bias, reinterpret_cast<float*>(y), H, W);
    } else if (L.cin == 1 && L.cout == 32 && pool == 1 && !x_is_u8) {
        OCR_REQUIRE(H % 2 == 0 && W % 2 == 0, "conv_first: pooled layer needs even H, W (got %dx%d)", H, W);
        if (first_layer_mfma()) {
            dim3 grid((W / 2 + 63) / 64, (H / 2 + 7) / 8, B), block(256);
            hipLaunchKernelGGL(conv_first_pool_mfma_kernel<false>, grid, block, 0, s, reinterpret_cast<const float*>(x), w, bias, y, H, W);
        } else {
            dim3 grid((W / 2 + 63) / 64, H / 2, B), block(256);
            hipLaunchKernelGGL((conv_first_kernel<1, 32, true, false>), grid, block, 0, s, x, w, bias, reinterpret_cast<float*>(y), H, W);
        }
    } else if (L.cin == 1 && L.cout == 32 && pool == 0 && !x_is_u8) {
        dim3 grid((W + 63) / 64, H, B), block(256);
        hipLaunchKernelGGL((conv_first_kernel<1, 32, false, false>), grid, block, 0, s, x, w, bias, reinterpret_cast<float*>(y), H, W);
    } else {
        set_error("conv_first: unsupported configuration cin=%d cout=%d pool=%d u8=%d", L.cin, L.cout, pool, (int)x_is_u8);
        return OCRHIP_EINVAL;
    }
    OCR_HIP_CHECK(hipGetLastError());
    return OCRHIP_OK;
}

int launch_east_heads(const ConvLayer& L, const float* x, float* maps, int64_t npix, hipStream_t s) {
    OCR_REQUIRE(L.cin == 32 && L.cout == 6, "east_heads: expected a 32 -> 6 layer");
    if (npix <= 0) return OCRHIP_OK;
    const int64_t blocks = (npix + 255) / 256;
    OCR_REQUIRE(blocks < (1ll << 31), "east_heads: too many pixels");
    ProfScope prof("east_heads", 2.0 * 32 * 6 * (double)npix, 4.0 * (32 + 6) * (double)npix, s);
    hipLaunchKernelGGL(east_heads_kernel, dim3((unsigned)blocks), dim3(256), 0, s, x, L.w_f32.as<float>(),
                       L.bias.as<float>(), maps, npix);
    OCR_HIP_CHECK(hipGetLastError());
    return OCRHIP_OK;
}

}  // namespace ocrhip
