// Stand-alone max pooling with zero padding of the width, NHWC float32.
//
// The ResNet extractor of the recogniser (architectures/crnn_arch.py:389-403) pads the width with one zero
// column on both sides before a 2x2 pooling of strides (2,1) and before its first 2x2 convolution; every
// other pooling of the path is fused into the epilogue of the convolution in front of it.  The padding is a
// real zero that takes part in the maximum (ZeroPadding2D followed by a 'valid' pooling).  HBM bound:
// 4 B x (input + output elements); 16 bytes per lane along the channels.
#include "common.h"

namespace ocrhip {

namespace {

__global__ __launch_bounds__(256) void pad_maxpool_kernel(const float* __restrict__ x, float* __restrict__ y, int H, int W,
                                                          int C4, int Ho, int Wo, int ph, int pw, int sh, int sw,
                                                          int pad_w, int64_t total) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;          // one float4 of the output
    if (i >= total) return;
    const int c4 = (int)(i % C4);
    int64_t r = i / C4;
    const int xo = (int)(r % Wo); r /= Wo;
    const int yo = (int)(r % Ho);
    const int64_t b = r / Ho;
    float4 m = make_float4(0.f, 0.f, 0.f, 0.f);
    bool first = true;
    for (int dy = 0; dy < ph; ++dy) {
        const int yy = yo * sh + dy;
        for (int dx = 0; dx < pw; ++dx) {
            const int xx = xo * sw + dx - pad_w;
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (xx >= 0 && xx < W) v = reinterpret_cast<const float4*>(x)[((b * H + yy) * W + xx) * C4 + c4];
            if (first) {
                m = v;
                first = false;
            } else {
                m.x = fmaxf(m.x, v.x); m.y = fmaxf(m.y, v.y); m.z = fmaxf(m.z, v.z); m.w = fmaxf(m.w, v.w);
            }
        }
    }
    reinterpret_cast<float4*>(y)[i] = m;
}

}  // namespace

int launch_pad_maxpool(const float* x, float* y, int B, int H, int W, int C, int ph, int pw, int sh, int sw, int pad_w,
                       hipStream_t s) {
    OCR_REQUIRE(x && y && B > 0 && H > 0 && W > 0 && C > 0 && C % 4 == 0, "pad_maxpool: bad arguments B=%d H=%d W=%d C=%d", B, H, W, C);
    OCR_REQUIRE(ph >= 1 && pw >= 1 && sh >= 1 && sw >= 1 && pad_w >= 0 && H >= ph && W + 2 * pad_w >= pw,
                "pad_maxpool: bad window %dx%d strides %dx%d pad %d for %dx%d", ph, pw, sh, sw, pad_w, H, W);
    const int Ho = (H - ph) / sh + 1, Wo = (W + 2 * pad_w - pw) / sw + 1;
    const int64_t total = (int64_t)B * Ho * Wo * (C / 4);
    const int64_t blocks = (total + 255) / 256;
    OCR_REQUIRE(blocks < (1ll << 31), "pad_maxpool: too large");
    ProfScope prof("pad_maxpool", (double)total * 4 * ph * pw, 4.0 * ((double)B * H * W * C + (double)total * 4), s);
    hipLaunchKernelGGL(pad_maxpool_kernel, dim3((unsigned)blocks), dim3(256), 0, s, x, y, H, W, C / 4, Ho, Wo, ph, pw, sh, sw,
                       pad_w, total);
    OCR_HIP_CHECK(hipGetLastError());
    return OCRHIP_OK;
}

}  // namespace ocrhip
