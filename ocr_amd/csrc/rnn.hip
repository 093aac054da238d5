// Recurrent part of the CRNN: mean over the remaining feature rows and one bidirectional LSTM
// time step (both directions in one launch).
//
// Restates architectures/crnn_arch.py:223-226 (GlobalAverage) and the Bidirectional(LSTM(256,
// return_sequences)) layers of crnn_arch.py:306-328 with Keras' cell equations and gate order
// i, f, c, o (= rows i, f, g, o of the torch checkpoint, models/weights_converter.py:305-314):
//     z = x W_ih^T + (b_ih + b_hh) + h_{t-1} W_hh^T ;  c = sig(z_f) c + sig(z_i) tanh(z_c) ;  h = sig(z_o) tanh(c)
// The input projection for all time steps is one flat GEMM (conv_mfma.hip); this file adds the
// recurrent product on the f32 matrix cores and the gate arithmetic.
//
// lstm_step tiling: a workgroup owns 32 sequences x 32 hidden units x 4 gates; its 4 waves split
// the 256-deep reduction in 4 slices of 64 and meet in LDS, wave 0 finishes the cell.  Operands go
// straight from L2 to registers in MFMA layout (16 bytes per lane; W_hh is used in checkpoint
// layout (4H, H) unchanged), there is no reuse inside a step that LDS staging could serve.
// The step is latency bound (128 dependent-free MFMAs per wave + one LDS exchange), not HBM bound.
#include "common.h"

namespace ocrhip {

namespace {

typedef float floatx16 __attribute__((ext_vector_type(16)));

constexpr int kHidden = 256;

__global__ __launch_bounds__(256) void mean_rows_kernel(const float* __restrict__ x, float* __restrict__ y, int H,
                                                        int64_t row_elems, int64_t total) {
    // x: (B,H,W,C) -> y: (B,W,C); row_elems = W*C
    const int64_t i = ((int64_t)blockIdx.x * 256 + threadIdx.x) * 4;
    if (i >= total) return;
    const int64_t bidx = i / row_elems, r = i % row_elems;
    const float* p = x + bidx * H * row_elems + r;
    float4 s = *reinterpret_cast<const float4*>(p);
    for (int h = 1; h < H; ++h) {
        const float4 v = *reinterpret_cast<const float4*>(p + h * row_elems);
        s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
    }
    const float d = (float)H;
    *reinterpret_cast<float4*>(y + i) = make_float4(s.x / d, s.y / d, s.z / d, s.w / d);
}

__device__ __forceinline__ float sigmoidf_(float x) { return 1.0f / (1.0f + expf(-x)); }

struct LstmArgs {
    const float* xw[2];      // per direction (n, T, 1024), bias included
    const float* whh[2];     // per direction (1024, 256)
    const float* h_prev;     // (2, n, 256)
    float* h_next;           // (2, n, 256)
    float* c_state;          // (2, n, 256)
    float* out;              // (n, T, 512)
    int n, T, step;
};

// grid (ceil(n/32), 8, 2), block 256
__global__ __launch_bounds__(256) void lstm_step_kernel(const LstmArgs a) {
    __shared__ float red[3 * 4 * 16 * 64];     // partial sums of waves 1..3: [wave-1][gate][v][lane]

    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int li = lane & 31, lh = lane >> 5;
    const int s0 = blockIdx.x * 32, u0 = blockIdx.y * 32, dir = blockIdx.z;
    const int t = dir ? (a.T - 1 - a.step) : a.step;
    const int n = a.n;

    floatx16 acc[4];
#pragma unroll
    for (int g = 0; g < 4; ++g)
#pragma unroll
        for (int v = 0; v < 16; ++v) acc[g][v] = 0.0f;

    if (a.step > 0) {
        const int kbase = wave * 64;
        const int seq = s0 + li;
        const float* hrow = a.h_prev + ((int64_t)dir * n + min(seq, n - 1)) * kHidden + kbase + lh * 4;
        const float* wrow = a.whh[dir] + ((int64_t)(u0 + li)) * kHidden + kbase + lh * 4;
        const bool seq_ok = seq < n;
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            float4 av = *reinterpret_cast<const float4*>(hrow + q * 8);
            if (!seq_ok) av = make_float4(0.f, 0.f, 0.f, 0.f);
            const float avv[4] = {av.x, av.y, av.z, av.w};
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const float4 bv = *reinterpret_cast<const float4*>(wrow + (int64_t)g * kHidden * kHidden + q * 8);
                const float bvv[4] = {bv.x, bv.y, bv.z, bv.w};
#pragma unroll
                for (int k = 0; k < 4; ++k)
                    acc[g] = __builtin_amdgcn_mfma_f32_32x32x2f32(avv[k], bvv[k], acc[g], 0, 0, 0);
            }
        }
        if (wave > 0) {
#pragma unroll
            for (int g = 0; g < 4; ++g)
#pragma unroll
                for (int v = 0; v < 16; ++v) red[(((wave - 1) * 4 + g) * 16 + v) * 64 + lane] = acc[g][v];
        }
    }
    __syncthreads();
    if (wave != 0) return;

    const int u = u0 + li;
#pragma unroll
    for (int v = 0; v < 16; ++v) {
        const int seq = s0 + (v & 3) + 8 * (v >> 2) + 4 * lh;
        if (seq >= n) continue;
        float z[4];
        const float* xw = a.xw[dir] + ((int64_t)seq * a.T + t) * (4 * kHidden) + u;
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            float s = acc[g][v];
            if (a.step > 0) {
                s += red[((0 * 4 + g) * 16 + v) * 64 + lane];
                s += red[((1 * 4 + g) * 16 + v) * 64 + lane];
                s += red[((2 * 4 + g) * 16 + v) * 64 + lane];
            }
            z[g] = xw[g * kHidden] + s;
        }
        const int64_t sidx = ((int64_t)dir * n + seq) * kHidden + u;
        const float c_prev = (a.step > 0) ? a.c_state[sidx] : 0.0f;
        const float c = sigmoidf_(z[1]) * c_prev + sigmoidf_(z[0]) * tanhf(z[2]);
        const float hval = sigmoidf_(z[3]) * tanhf(c);
        a.c_state[sidx] = c;
        a.h_next[sidx] = hval;
        a.out[((int64_t)seq * a.T + t) * (2 * kHidden) + dir * kHidden + u] = hval;
    }
}

}  // namespace

int launch_mean_rows(const float* x, float* y, int B, int H, int W, int C, hipStream_t s) {
    OCR_REQUIRE(x && y && B > 0 && H > 0 && W > 0 && C > 0 && C % 4 == 0, "mean_rows: bad arguments");
    const int64_t row_elems = (int64_t)W * C, total = (int64_t)B * row_elems;
    const int64_t blocks = (total / 4 + 255) / 256;
    OCR_REQUIRE(blocks < (1ll << 31), "mean_rows: too large");
    ProfScope prof("mean_rows", (double)total * H, 4.0 * (double)total * (H + 1), s);
    hipLaunchKernelGGL(mean_rows_kernel, dim3((unsigned)blocks), dim3(256), 0, s, x, y, H, row_elems, total);
    OCR_HIP_CHECK(hipGetLastError());
    return OCRHIP_OK;
}

int launch_lstm_step(const float* xw_f, const float* xw_b, const float* whh_f, const float* whh_b, const float* h_prev,
                     float* h_next, float* c_state, float* out, int n, int T, int step, hipStream_t s) {
    OCR_REQUIRE(xw_f && xw_b && whh_f && whh_b && h_prev && h_next && c_state && out, "lstm_step: null argument");
    OCR_REQUIRE(n > 0 && T > 0 && step >= 0 && step < T, "lstm_step: bad sizes n=%d T=%d step=%d", n, T, step);
    LstmArgs a;
    a.xw[0] = xw_f; a.xw[1] = xw_b; a.whh[0] = whh_f; a.whh[1] = whh_b;
    a.h_prev = h_prev; a.h_next = h_next; a.c_state = c_state; a.out = out; a.n = n; a.T = T; a.step = step;
    dim3 grid((n + 31) / 32, kHidden / 32, 2);
    hipLaunchKernelGGL(lstm_step_kernel, grid, dim3(256), 0, s, a);
    OCR_HIP_CHECK(hipGetLastError());
    return OCRHIP_OK;
}

}  // namespace ocrhip
