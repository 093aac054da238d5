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
// lstm_step tiling: a workgroup owns 64 sequences x 32 hidden units x 4 gates, one gate per wave; the
// operands are staged through LDS with full-line coalesced loads (W_hh is used in checkpoint layout
// (4H, H) unchanged) and the four gates of an element are exchanged through LDS before the cell update.
// One launch per time step covers both directions and every sequence that is still running; the step
// is bound by the f32 MFMA rate and the L2 -> LDS re-read of W_hh by every sequence tile.
#include "common.h"

namespace ocrhip {

namespace {

typedef float floatx16 __attribute__((ext_vector_type(16)));

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
    const float* xw[2];      // per direction (n, T, 4H), bias included; column = unit * 4 + gate
    const float* whh[2];     // per direction (4H, H)
    const float* h_prev;     // (2, n, H)
    float* h_next;           // (2, n, H)
    float* c_state;          // (2, n, H)
    float* out;              // (n, T, 2H)
    const int* seq_T;        // ragged mode: time steps of each sequence (non-decreasing), else nullptr
    const int* seq_row0;     // ragged mode: first row of each sequence in the flat (sum T, .) matrices
    int n, T, step, first;   // first = first sequence still active at this step (ragged mode)
    int H;                   // hidden units, a multiple of 64 (crnn_arch.py:485-486 `lstm_size`)
};

constexpr int kSeqTile = 64;     // sequences per workgroup
constexpr int kLds = 64 + 4;     // floats per staged row (64 k values + pad: conflict-free ds_read_b128)

// grid (ceil(active/64), H/32, 2), block 256.  Workgroup tile: 64 sequences x 32 hidden units x 4 gates;
// wave g owns gate g (two 32x32 accumulators, full K = H).  h_{t-1} and W_hh are staged through LDS in
// K chunks of 64 with fully coalesced 16-byte loads (a row's 256 bytes by 16 consecutive lanes); the next
// chunk is prefetched into registers during the MFMAs.  The four gates of an element meet in LDS.
__global__ __launch_bounds__(256) void lstm_step_kernel(const LstmArgs a) {
    __shared__ __attribute__((aligned(16))) float As[kSeqTile * kLds];        // h_{t-1} chunk
    __shared__ __attribute__((aligned(16))) float Bs[4 * 32 * kLds];          // W_hh chunk, rows = gate*32 + unit
    float* gates = Bs;                                                        // reused after the K loop: [gate][r][v][lane]

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int li = lane & 31, lh = lane >> 5;
    const int s0 = a.first + blockIdx.x * kSeqTile, u0 = blockIdx.y * 32, dir = blockIdx.z;
    const int n = a.n, kHidden = a.H, kChunks = a.H / 64;

    floatx16 acc[2];
#pragma unroll
    for (int r = 0; r < 2; ++r)
#pragma unroll
        for (int v = 0; v < 16; ++v) acc[r][v] = 0.0f;

    if (a.step > 0) {
        // staging map: 16 lanes per row (16 x float4 = 64 floats)
        const int srow = tid >> 4, scol = (tid & 15) * 4;
        float4 ra[4], rb[8];
        auto load_chunk = [&](int kc) {
#pragma unroll
            for (int i = 0; i < 4; ++i) {                 // 64 sequence rows
                const int seq = s0 + srow + i * 16;
                ra[i] = (seq < n) ? *reinterpret_cast<const float4*>(a.h_prev + ((int64_t)dir * n + seq) * kHidden + kc * 64 + scol)
                                  : make_float4(0.f, 0.f, 0.f, 0.f);
            }
#pragma unroll
            for (int i = 0; i < 8; ++i) {                 // 128 weight rows: gate = row / 32, unit = row % 32
                const int row = srow + i * 16;
                rb[i] = *reinterpret_cast<const float4*>(a.whh[dir] + ((int64_t)(row >> 5) * kHidden + u0 + (row & 31)) * kHidden + kc * 64 + scol);
            }
        };
        auto store_chunk = [&]() {
#pragma unroll
            for (int i = 0; i < 4; ++i) *reinterpret_cast<float4*>(&As[(srow + i * 16) * kLds + scol]) = ra[i];
#pragma unroll
            for (int i = 0; i < 8; ++i) *reinterpret_cast<float4*>(&Bs[(srow + i * 16) * kLds + scol]) = rb[i];
        };
        load_chunk(0);
        store_chunk();
        __syncthreads();
        for (int kc = 0; kc < kChunks; ++kc) {
            if (kc + 1 < kChunks) load_chunk(kc + 1);
#pragma unroll
            for (int q = 0; q < 8; ++q) {
                const float4 b4 = *reinterpret_cast<const float4*>(&Bs[(wave * 32 + li) * kLds + q * 8 + lh * 4]);
                const float bv[4] = {b4.x, b4.y, b4.z, b4.w};
#pragma unroll
                for (int r = 0; r < 2; ++r) {
                    const float4 a4 = *reinterpret_cast<const float4*>(&As[(r * 32 + li) * kLds + q * 8 + lh * 4]);
                    const float av[4] = {a4.x, a4.y, a4.z, a4.w};
#pragma unroll
                    for (int k = 0; k < 4; ++k)
                        acc[r] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[k], bv[k], acc[r], 0, 0, 0);
                }
            }
            __syncthreads();
            if (kc + 1 < kChunks) {
                store_chunk();
                __syncthreads();
            }
        }
    }
    // gate exchange: [gate][r][v][lane]
#pragma unroll
    for (int r = 0; r < 2; ++r)
#pragma unroll
        for (int v = 0; v < 16; ++v) gates[((wave * 2 + r) * 16 + v) * 64 + lane] = acc[r][v];
    __syncthreads();

    // element (r, v, lane): sequence s0 + r*32 + (v&3) + 8*(v>>2) + 4*(lane>>5), unit u0 + (lane&31);
    // wave w finishes v in [4w, 4w+4) of both r
    const int u = u0 + li;
#pragma unroll
    for (int r = 0; r < 2; ++r)
#pragma unroll
        for (int vv = 0; vv < 4; ++vv) {
            const int v = wave * 4 + vv;
            const int seq = s0 + r * 32 + (v & 3) + 8 * (v >> 2) + 4 * lh;
            if (seq >= n) continue;
            const int Ts = a.seq_T ? a.seq_T[seq] : a.T;
            const int64_t row = (a.seq_row0 ? (int64_t)a.seq_row0[seq] : (int64_t)seq * a.T) + (dir ? (Ts - 1 - a.step) : a.step);
            const float4 xw4 = *reinterpret_cast<const float4*>(a.xw[dir] + row * (4 * kHidden) + u * 4);
            const float xw[4] = {xw4.x, xw4.y, xw4.z, xw4.w};
            float z[4];
#pragma unroll
            for (int g = 0; g < 4; ++g) z[g] = xw[g] + gates[((g * 2 + r) * 16 + v) * 64 + lane];
            const int64_t sidx = ((int64_t)dir * n + seq) * kHidden + u;
            const float c_prev = (a.step > 0) ? a.c_state[sidx] : 0.0f;
            const float c = sigmoidf_(z[1]) * c_prev + sigmoidf_(z[0]) * tanhf(z[2]);
            const float hval = sigmoidf_(z[3]) * tanhf(c);
            a.c_state[sidx] = c;
            a.h_next[sidx] = hval;
            a.out[row * (2 * kHidden) + dir * kHidden + u] = hval;
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
                     float* h_next, float* c_state, float* out, int n, int T, int step, const int* seq_T,
                     const int* seq_row0, int first, int H, hipStream_t s) {
    OCR_REQUIRE(xw_f && xw_b && whh_f && whh_b && h_prev && h_next && c_state && out, "lstm_step: null argument");
    OCR_REQUIRE(n > 0 && T > 0 && step >= 0 && step < T && first >= 0 && first < n,
                "lstm_step: bad sizes n=%d T=%d step=%d first=%d", n, T, step, first);
    OCR_REQUIRE((seq_T == nullptr) == (seq_row0 == nullptr), "lstm_step: seq_T and seq_row0 go together");
    OCR_REQUIRE(H >= 64 && H % 64 == 0, "lstm_step: the hidden size must be a multiple of 64 (got %d)", H);
    LstmArgs a;
    a.H = H;
    a.seq_T = seq_T; a.seq_row0 = seq_row0; a.first = first;
    a.xw[0] = xw_f; a.xw[1] = xw_b; a.whh[0] = whh_f; a.whh[1] = whh_b;
    a.h_prev = h_prev; a.h_next = h_next; a.c_state = c_state; a.out = out; a.n = n; a.T = T; a.step = step;
    dim3 grid((n - first + kSeqTile - 1) / kSeqTile, H / 32, 2);
    hipLaunchKernelGGL(lstm_step_kernel, grid, dim3(256), 0, s, a);
    OCR_HIP_CHECK(hipGetLastError());
    return OCRHIP_OK;
}

}  // namespace ocrhip
