// CTC greedy decoding: argmax over the raw logits, merge repeats, drop blank, left-pack, pad -1.
//
// Restates utils/text/ctc_decoder.py:14-35 (keras.ops.ctc_decode, strategy 'greedy',
// mask_index = blank): the first maximum wins, the score is -sum_{t<len} max_v logit[t] / len with the
// sum taken in time order in float32.  One workgroup per sequence, one wave per time step for the
// argmax, a ballot compaction for the collapse; sequences of any length are worked off in chunks of
// kChunkT steps (the previous token, the output position and the score sum carry over).
// HBM bound: 4 B x T x V read + 4 B x T written.
//
// PARTIAL variant: the class layer already reduced every row to one (maximum, column) pair per block of 128
// columns (gemm_f32.hip, ARGMAX); the argmax is finished over those pairs, 8 B x T x ceil(V / 128) read.
#include "common.h"

namespace ocrhip {

namespace {

constexpr int kChunkT = 2048;

template <bool PARTIAL>
__global__ __launch_bounds__(256) void ctc_greedy_kernel(const float* __restrict__ logits, const int* __restrict__ pidx,
                                                         int T, int V, int ld_v, const int* __restrict__ lengths,
                                                         const int* __restrict__ row0, int blank,
                                                         int* __restrict__ tokens, float* __restrict__ scores) {
    __shared__ int s_idx[kChunkT];
    __shared__ float s_max[kChunkT];
    __shared__ int s_wave[4];
    __shared__ int s_base, s_prev;
    __shared__ float s_acc;

    const int b = blockIdx.x;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    int L = lengths ? lengths[b] : T;
    L = min(max(L, 0), T);
    const int64_t first = row0 ? (int64_t)row0[b] : (int64_t)b * T;
    if (threadIdx.x == 0) { s_base = 0; s_prev = -1; s_acc = 0.0f; }
    int* out = tokens + (int64_t)b * T;

    for (int t0 = 0; t0 < L; t0 += kChunkT) {
        const int n = min(kChunkT, L - t0);
        if (PARTIAL) {
            // ld_v pairs per row, column blocks in ascending order: a later equal maximum does not win
            for (int t = threadIdx.x; t < n; t += 256) {
                const int64_t r = (first + t0 + t) * ld_v;
                float best = logits[r];
                int best_i = pidx[r];
                for (int p = 1; p < ld_v; ++p) {
                    const float x = logits[r + p];
                    if (x > best) { best = x; best_i = pidx[r + p]; }
                }
                s_idx[t] = best_i;
                s_max[t] = best;
            }
        } else {
            for (int t = wave; t < n; t += 4) {
                const float* row = logits + (first + t0 + t) * ld_v;
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
        }
        __syncthreads();

        for (int c0 = 0; c0 < n; c0 += 256) {
            const int t = c0 + threadIdx.x;
            int tok = -1;
            bool keep = false;
            if (t < n) {
                tok = s_idx[t];
                const int prev = (t > 0) ? s_idx[t - 1] : s_prev;
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
        if (threadIdx.x == 0) {
            float acc = s_acc;
            for (int t = 0; t < n; ++t) acc = acc + s_max[t];
            s_acc = acc;
            s_prev = s_idx[n - 1];
        }
        __syncthreads();
    }
    __syncthreads();
    const int n_tok = s_base;
    for (int t = n_tok + threadIdx.x; t < T; t += 256) out[t] = -1;
    if (threadIdx.x == 0 && scores) scores[b] = (L > 0) ? (-s_acc / (float)L) : 0.0f;
}

}  // namespace

int launch_ctc_greedy(const float* logits, int B, int T, int V, int ld_v, const int* lengths, const int* row0, int blank,
                      int* tokens, float* scores, hipStream_t s) {
    OCR_REQUIRE(row0 == nullptr || lengths != nullptr, "ctc_greedy: row offsets need lengths");
    OCR_REQUIRE(logits && tokens, "ctc_greedy: logits and tokens are required");
    OCR_REQUIRE(B >= 0 && T > 0 && V > 0 && ld_v >= V, "ctc_greedy: bad sizes B=%d T=%d V=%d ld=%d", B, T, V, ld_v);
    if (B == 0) return OCRHIP_OK;
    ProfScope prof("ctc_greedy", 0.0, 4.0 * ((double)B * T * V + (double)B * T), s);
    hipLaunchKernelGGL(ctc_greedy_kernel<false>, dim3(B), dim3(256), 0, s, logits, (const int*)nullptr, T, V, ld_v, lengths, row0,
                       blank, tokens, scores);
    OCR_HIP_CHECK(hipGetLastError());
    return OCRHIP_OK;
}

int launch_ctc_greedy_partial(const float* pmax, const int* pidx, int n_blocks, int B, int T, const int* lengths,
                              const int* row0, int blank, int* tokens, float* scores, hipStream_t s) {
    OCR_REQUIRE(row0 == nullptr || lengths != nullptr, "ctc_greedy: row offsets need lengths");
    OCR_REQUIRE(pmax && pidx && tokens && n_blocks > 0, "ctc_greedy: partial maxima and tokens are required");
    OCR_REQUIRE(B >= 0 && T > 0, "ctc_greedy: bad sizes B=%d T=%d", B, T);
    if (B == 0) return OCRHIP_OK;
    ProfScope prof("ctc_greedy_partial", 0.0, 4.0 * ((double)B * T * 2 * n_blocks + (double)B * T), s);
    hipLaunchKernelGGL(ctc_greedy_kernel<true>, dim3(B), dim3(256), 0, s, pmax, pidx, T, n_blocks, n_blocks, lengths, row0, blank,
                       tokens, scores);
    OCR_HIP_CHECK(hipGetLastError());
    return OCRHIP_OK;
}

}  // namespace ocrhip
