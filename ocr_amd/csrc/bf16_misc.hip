// Bandwidth-bound helpers of the bfloat16-tensor path (`compute_dtype = OCRHIP_BF16`): every kernel here moves 16 bytes
// per lane, reads its input once and writes its output once.
//   upcat16     x2 align-corners bilinear upsampling + channel concatenation in front of a merge stage's 1x1 convolution
//               (architectures/east_arch.py:37-48,135-141), interpolated in float32 from bfloat16 values
//   heads16     the three sigmoid heads on the 32-channel merged map (east_arch.py:153-162): float32 arithmetic, float32 maps
//   mean_rows16 mean over the remaining feature rows (architectures/crnn_arch.py:223-226)
//   convert     float32 <-> bfloat16 copies (glue to the float32 kernels: the per-step LSTM, debug read-back)
#include "common.h"

namespace ocrhip {

namespace {

typedef unsigned uintx4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ float lo_f(unsigned u) { return __builtin_bit_cast(float, u << 16); }
__device__ __forceinline__ float hi_f(unsigned u) { return __builtin_bit_cast(float, u & 0xffff0000u); }
__device__ __forceinline__ unsigned to_bits(float f) {
    const __bf16 b = (__bf16)f;
    return (unsigned)__builtin_bit_cast(unsigned short, b);
}
__device__ __forceinline__ unsigned pack(float lo, float hi) { return to_bits(lo) | (to_bits(hi) << 16); }

// out (B,H,W,up_c + res_c): channels [0, up_c) = up (B,H/2,W/2,up_c) upsampled x2 with aligned corners, the rest = res
__global__ __launch_bounds__(256) void upcat16_kernel(const uint16_t* __restrict__ up, const uint16_t* __restrict__ res,
                                                      uint16_t* __restrict__ out, int H, int W, int up_c, int res_c, float ry,
                                                      float rx, int64_t total) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;          // one 8-channel piece
    if (i >= total) return;
    const int C = up_c + res_c, pieces = C / 8;
    const int pc = (int)(i % pieces);
    const int64_t px = i / pieces;
    const int xo = (int)(px % W), yo = (int)((px / W) % H);
    const int64_t b = px / ((int64_t)W * H);
    uintx4 v;
    if (pc * 8 >= up_c) {
        v = *reinterpret_cast<const uintx4*>(res + px * res_c + (pc * 8 - up_c));
    } else {
        const int Hs = H / 2, Ws = W / 2;
        const float sy = ry * (float)yo, sx = rx * (float)xo;
        const int iy = (int)sy, ix = (int)sx;
        const int iy1 = iy + ((iy < Hs - 1) ? 1 : 0), ix1 = ix + ((ix < Ws - 1) ? 1 : 0);
        const float ly = sy - (float)iy, lx = sx - (float)ix;
        const float hy = 1.0f - ly, hx = 1.0f - lx;
        const uint16_t* base = up + b * Hs * Ws * up_c + pc * 8;
        const uintx4 v00 = *reinterpret_cast<const uintx4*>(base + ((int64_t)iy * Ws + ix) * up_c);
        const uintx4 v01 = *reinterpret_cast<const uintx4*>(base + ((int64_t)iy * Ws + ix1) * up_c);
        const uintx4 v10 = *reinterpret_cast<const uintx4*>(base + ((int64_t)iy1 * Ws + ix) * up_c);
        const uintx4 v11 = *reinterpret_cast<const uintx4*>(base + ((int64_t)iy1 * Ws + ix1) * up_c);
        unsigned o[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const float l = hy * (hx * lo_f(v00[e]) + lx * lo_f(v01[e])) + ly * (hx * lo_f(v10[e]) + lx * lo_f(v11[e]));
            const float h = hy * (hx * hi_f(v00[e]) + lx * hi_f(v01[e])) + ly * (hx * hi_f(v10[e]) + lx * hi_f(v11[e]));
            o[e] = pack(l, h);
        }
        v = uintx4{o[0], o[1], o[2], o[3]};
    }
    *reinterpret_cast<uintx4*>(out + px * C + pc * 8) = v;
}

// maps[p][0..5] = sigmoid(W x[p] + b), x bfloat16 (npix, 32), w float32 [6][32]
__global__ __launch_bounds__(256) void heads16_kernel(const uint16_t* __restrict__ x, const float* __restrict__ w,
                                                      const float* __restrict__ bias, float* __restrict__ maps, int64_t npix) {
    const int64_t p = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (p >= npix) return;
    float v[32];
    const uintx4* xp = reinterpret_cast<const uintx4*>(x + p * 32);
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const uintx4 q = xp[i];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            v[8 * i + 2 * e] = lo_f(q[e]);
            v[8 * i + 2 * e + 1] = hi_f(q[e]);
        }
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

// x (B,H,W*C) -> y (B,W*C): mean over H, float32 sums; row_elems = W*C (a multiple of 8)
__global__ __launch_bounds__(256) void mean_rows16_kernel(const uint16_t* __restrict__ x, uint16_t* __restrict__ y, int H,
                                                          int64_t row_elems, int64_t total) {
    const int64_t i = ((int64_t)blockIdx.x * 256 + threadIdx.x) * 8;
    if (i >= total) return;
    const int64_t bidx = i / row_elems, r = i % row_elems;
    const uint16_t* p = x + bidx * H * row_elems + r;
    float s[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    for (int h = 0; h < H; ++h) {
        const uintx4 v = *reinterpret_cast<const uintx4*>(p + h * row_elems);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            s[2 * e] += lo_f(v[e]);
            s[2 * e + 1] += hi_f(v[e]);
        }
    }
    const float d = (float)H;
    *reinterpret_cast<uintx4*>(y + i) = uintx4{pack(s[0] / d, s[1] / d), pack(s[2] / d, s[3] / d), pack(s[4] / d, s[5] / d), pack(s[6] / d, s[7] / d)};
}

__global__ __launch_bounds__(256) void f32_to_bf16_kernel(const float* __restrict__ x, uint16_t* __restrict__ y, int64_t n) {
    const int64_t i = ((int64_t)blockIdx.x * 256 + threadIdx.x) * 4;
    if (i + 3 < n) {
        const float4 v = *reinterpret_cast<const float4*>(x + i);
        *reinterpret_cast<uint2*>(y + i) = make_uint2(pack(v.x, v.y), pack(v.z, v.w));
    } else {
        for (int64_t k = i; k < n; ++k) y[k] = (uint16_t)to_bits(x[k]);
    }
}

__global__ __launch_bounds__(256) void bf16_to_f32_kernel(const uint16_t* __restrict__ x, float* __restrict__ y, int64_t n) {
    const int64_t i = ((int64_t)blockIdx.x * 256 + threadIdx.x) * 4;
    if (i + 3 < n) {
        const uint2 v = *reinterpret_cast<const uint2*>(x + i);
        *reinterpret_cast<float4*>(y + i) = make_float4(lo_f(v.x), hi_f(v.x), lo_f(v.y), hi_f(v.y));
    } else {
        for (int64_t k = i; k < n; ++k) y[k] = lo_f((unsigned)x[k]);
    }
}

}  // namespace

int launch_upcat16(const uint16_t* up, const uint16_t* res, uint16_t* out, int B, int H, int W, int up_c, int res_c, hipStream_t s) {
    OCR_REQUIRE(up && res && out && B > 0 && H > 0 && W > 0 && H % 2 == 0 && W % 2 == 0 && up_c % 8 == 0 && res_c % 8 == 0,
                "upcat16: bad arguments H=%d W=%d up_c=%d res_c=%d", H, W, up_c, res_c);
    const int64_t total = (int64_t)B * H * W * ((up_c + res_c) / 8);
    const float ry = (H > 1) ? (float)(H / 2 - 1) / (float)(H - 1) : 0.f;
    const float rx = (W > 1) ? (float)(W / 2 - 1) / (float)(W - 1) : 0.f;
    ProfScope prof("upcat16", 0.0, 2.0 * (double)B * H * W * (up_c / 4.0 + res_c) + 2.0 * (double)B * H * W * (up_c + res_c), s);
    hipLaunchKernelGGL(upcat16_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, up, res, out, H, W, up_c, res_c, ry, rx, total);
    OCR_HIP_CHECK(hipGetLastError());
    return OCRHIP_OK;
}

int launch_east_heads16(const ConvLayer& L, const uint16_t* x, float* maps, int64_t npix, hipStream_t s) {
    OCR_REQUIRE(L.cin == 32 && L.cout == 6, "east_heads16: expected a 32 -> 6 layer");
    if (npix <= 0) return OCRHIP_OK;
    ProfScope prof("east_heads16", 2.0 * 32 * 6 * (double)npix, (2.0 * 32 + 4.0 * 6) * (double)npix, s);
    hipLaunchKernelGGL(heads16_kernel, dim3((unsigned)((npix + 255) / 256)), dim3(256), 0, s, x, L.w_f32.as<float>(), L.bias.as<float>(), maps, npix);
    OCR_HIP_CHECK(hipGetLastError());
    return OCRHIP_OK;
}

int launch_mean_rows16(const uint16_t* x, uint16_t* y, int B, int H, int W, int C, hipStream_t s) {
    const int64_t row_elems = (int64_t)W * C, total = (int64_t)B * row_elems;
    OCR_REQUIRE(x && y && H > 0 && row_elems % 8 == 0, "mean_rows16: bad arguments");
    if (total == 0) return OCRHIP_OK;
    ProfScope prof("mean_rows16", 0.0, 2.0 * (double)total * (H + 1), s);
    hipLaunchKernelGGL(mean_rows16_kernel, dim3((unsigned)((total / 8 + 255) / 256)), dim3(256), 0, s, x, y, H, row_elems, total);
    OCR_HIP_CHECK(hipGetLastError());
    return OCRHIP_OK;
}

int launch_f32_to_bf16(const float* x, uint16_t* y, int64_t n, hipStream_t s) {
    if (n <= 0) return OCRHIP_OK;
    hipLaunchKernelGGL(f32_to_bf16_kernel, dim3((unsigned)((n / 4 + 256) / 256)), dim3(256), 0, s, x, y, n);
    OCR_HIP_CHECK(hipGetLastError());
    return OCRHIP_OK;
}

int launch_bf16_to_f32(const uint16_t* x, float* y, int64_t n, hipStream_t s) {
    if (n <= 0) return OCRHIP_OK;
    hipLaunchKernelGGL(bf16_to_f32_kernel, dim3((unsigned)((n / 4 + 256) / 256)), dim3(256), 0, s, x, y, n);
    OCR_HIP_CHECK(hipGetLastError());
    return OCRHIP_OK;
}

}  // namespace ocrhip
