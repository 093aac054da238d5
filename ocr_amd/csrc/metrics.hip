// Dice coefficient of two detector score maps: the frame de-duplication test of the OCR stream
// (reference: utils/distances.py:125-130, called from models/ocr/base_ocr.py:408-411 on `output[:, :, 0]`).
//   dice = 2 * sum(x * y) / (sum(x) + sum(y)),  0 when the denominator is 0 (divide_no_nan)
// One workgroup walks both maps once (a 256x256 map is 64 elements per lane); sums in double, combined in a fixed
// order, so the result does not depend on scheduling.
#include "common.h"

namespace ocrhip {

namespace {

constexpr int kThreads = 1024;

__global__ __launch_bounds__(kThreads) void dice_coeff_kernel(const float* __restrict__ x, const float* __restrict__ y,
                                                              int64_t n, int64_t stride, float* __restrict__ out) {
    __shared__ double s_sum[3][kThreads / 64];
    double inter = 0.0, sx = 0.0, sy = 0.0;
    for (int64_t i = threadIdx.x; i < n; i += kThreads) {
        const float a = x[i * stride], b = y[i * stride];
        inter += (double)a * (double)b;
        sx += (double)a;
        sy += (double)b;
    }
    for (int off = 32; off > 0; off >>= 1) {
        inter += __shfl_down(inter, off, 64);
        sx += __shfl_down(sx, off, 64);
        sy += __shfl_down(sy, off, 64);
    }
    const int wave = threadIdx.x >> 6;
    if ((threadIdx.x & 63) == 0) {
        s_sum[0][wave] = inter;
        s_sum[1][wave] = sx;
        s_sum[2][wave] = sy;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        double t[3] = {0.0, 0.0, 0.0};
        for (int w = 0; w < kThreads / 64; ++w)
            for (int k = 0; k < 3; ++k) t[k] += s_sum[k][w];
        const double den = t[1] + t[2];
        out[0] = den != 0.0 ? (float)(2.0 * t[0] / den) : 0.f;
    }
}

}  // namespace

int launch_dice_coeff(const float* x, const float* y, int64_t n, int64_t stride, float* out, hipStream_t s) {
    OCR_REQUIRE(x && y && out, "dice_coeff: null pointer");
    OCR_REQUIRE(n >= 0 && stride >= 1, "dice_coeff: n >= 0 and stride >= 1 expected");
    ProfScope prof("dice_coeff", 3.0 * (double)n, 8.0 * (double)n, s);
    hipLaunchKernelGGL(dice_coeff_kernel, dim3(1), dim3(kThreads), 0, s, x, y, n, stride, out);
    OCR_HIP_CHECK(hipGetLastError());
    return OCRHIP_OK;
}

}  // namespace ocrhip
