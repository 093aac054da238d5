// CTC greedy decoding: argmax over the raw logits, merge repeats, drop blank, left-pack, pad -1.
//
// Restates utils/text/ctc_decoder.py:14-35 (keras.ops.ctc_decode, strategy 'greedy',
// mask_index = blank): the first maximum wins, the score is -sum_{t<len} max_v logit[t] / len with the
// sum taken in time order in float32.  One workgroup per sequence, one wave per time step for the
// argmax, a ballot compaction for the collapse.  HBM bound: 4 B x T x V read + 4 B x T written.
#include "common.h"

namespace ocrhip {

namespace {

constexpr int kMaxT = 2048;

__global__ __launch_bounds__(256) void ctc_greedy_kernel(const float* __restrict__ logits, int T, int V, int ld_v,
                                                         const int* __restrict__ lengths,
                                                         const int* __restrict__ row0, int blank,
                                                         int* __restrict__ tokens, float* __restrict__ scores) {
    __shared__ int s_idx[kMaxT];
    __shared__ float s_max[kMaxT];
    __shared__ int s_wave[4];
    __shared__ int s_base;

    const int b = blockIdx.x;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    int L = lengths ? lengths[b] : T;
    L = min(max(L, 0), T);
    const float* first_row = logits + (row0 ? (int64_t)row0[b] : (int64_t)b * T) * ld_v;

    for (int t = wave; t < L; t += 4) {
        const float* row = first_row + (int64_t)t * ld_v;
        float best = -INFINITY;
        int best_i = 0x7fffffff;
        for (int v = lane; v < V; v += 64) {
            const float x = row[v];
            if (x > best || best_i == 0x7fffffff) { best = x; best_i = v; }
        }
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) {
            const float ob = __shfl_xor(best, off, 64);
            const int oi = __shfl_xor(best_i, off, 64);
            if (ob > best || (ob == best && oi < best_i)) { best = ob; best_i = oi; }
        }
        if (lane == 0) { s_idx[t] = best_i; s_max[t] = best; }
    }
    if (threadIdx.x == 0) s_base = 0;
    __syncthreads();

    int* out = tokens + (int64_t)b * T;
    for (int c0 = 0; c0 < L; c0 += 256) {
        const int t = c0 + threadIdx.x;
        int tok = -1;
        bool keep = false;
        if (t < L) {
            tok = s_idx[t];
            const int prev = (t > 0) ? s_idx[t - 1] : -1;
            keep = (tok != blank) && (tok != prev);
        }
        const unsigned long long m = __ballot(keep);
        if (lane == 0) s_wave[wave] = __popcll(m);
        __syncthreads();
        int pos = s_base;
        for (int w = 0; w < wave; ++w) pos += s_wave[w];
        if (keep) out[pos + __popcll(m & ((1ull << lane) - 1ull))] = tok;
        __syncthreads();
        if (threadIdx.x == 0) s_base += s_wave[0] + s_wave[1] + s_wave[2] + s_wave[3];
        __syncthreads();
    }
    const int n_tok = s_base;
    for (int t = n_tok + threadIdx.x; t < T; t += 256) out[t] = -1;

    if (threadIdx.x == 0 && scores) {
        float acc = 0.0f;
        for (int t = 0; t < L; ++t) acc = acc + s_max[t];
        scores[b] = (L > 0) ? (-acc / (float)L) : 0.0f;
    }
}

}  // namespace

int launch_ctc_greedy(const float* logits, int B, int T, int V, int ld_v, const int* lengths, const int* row0, int blank,
                      int* tokens, float* scores, hipStream_t s) {
    OCR_REQUIRE(row0 == nullptr || lengths != nullptr, "ctc_greedy: row offsets need lengths");
    OCR_REQUIRE(logits && tokens, "ctc_greedy: logits and tokens are required");
    OCR_REQUIRE(B >= 0 && T > 0 && V > 0 && ld_v >= V, "ctc_greedy: bad sizes B=%d T=%d V=%d ld=%d", B, T, V, ld_v);
    OCR_REQUIRE(T <= kMaxT, "ctc_greedy: T=%d exceeds the supported maximum %d", T, kMaxT);
    if (B == 0) return OCRHIP_OK;
    ProfScope prof("ctc_greedy", 0.0, 4.0 * ((double)B * T * V + (double)B * T), s);
    hipLaunchKernelGGL(ctc_greedy_kernel, dim3(B), dim3(256), 0, s, logits, T, V, ld_v, lengths, row0, blank, tokens, scores);
    OCR_HIP_CHECK(hipGetLastError());
    return OCRHIP_OK;
}

}  // namespace ocrhip
