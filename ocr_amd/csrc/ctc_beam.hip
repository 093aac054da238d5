// CTC prefix beam search on the GPU: one wave per sequence, float64 path probabilities.
//
// Stands in for ctc_decode(method='beam_search') of the reference (utils/text/ctc_decoder.py:28-35 ->
// keras.ops.ctc_decode(strategy='beam_search'), the default of BaseOCR.infer, models/ocr/base_ocr.py:112-115,209-222):
// prefix search over the per-step log-softmax of the logits with blank / non-blank path probabilities in log space;
// at every step every label may extend every kept prefix and the `num_beams` most probable prefixes survive; the best
// sequence and its log probability divided by the input length are returned.
//
// Per step the wave
//   1. forms the step's log-softmax normaliser (maximum, sum of exponentials in float64),
//   2. finds the num_beams + 1 most probable non-blank labels of the step (they are the same for every prefix, and a
//      prefix's num_beams best extensions lie among them: only a repeat of its last label is worth less than
//      total + log p),
//   3. picks the num_beams best of the (prefix, label) extensions among those, plus every extension that lands on a
//      kept prefix (its probability adds to what that prefix keeps by staying),
//   4. keeps the num_beams best of {stay, picked extension} entries, in descending order.
// Prefixes are nodes of a tree (parent, label) in a per-sequence pool in device memory; the kept ones are distinct,
// so a (prefix, label) pair names at most one kept prefix.  Everything else lives in LDS; lanes work on candidates in
// parallel, selections are wave arg-max rounds in the order (value descending, index ascending): deterministic.
// Work per step ~ (num_beams + 2) scans of the V logits: latency / L2 bound, not a bandwidth kernel.
#include "common.h"

namespace ocrhip {

namespace {

constexpr int kMaxBeams = 16;
constexpr int kMaxTop = kMaxBeams + 1;
constexpr int kMaxEntries = 3 * kMaxBeams;      // stays + picked + landing extensions

__device__ __forceinline__ double shfl_xor_d(double v, int off) {
    const unsigned long long u = __builtin_bit_cast(unsigned long long, v);
    const unsigned lo = __shfl_xor((unsigned)u, off, 64), hi = __shfl_xor((unsigned)(u >> 32), off, 64);
    return __builtin_bit_cast(double, ((unsigned long long)hi << 32) | lo);
}

__device__ __forceinline__ double logaddexp_d(double a, double b) {
    if (a == -INFINITY) return b;
    if (b == -INFINITY) return a;
    const double m = a > b ? a : b;
    return m + log1p(exp(-fabs(a - b)));
}

// wave arg-max in the order (value descending, index ascending); lanes without a candidate pass idx = INT_MAX
__device__ __forceinline__ void wave_best(double& v, int& i) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        const double ov = shfl_xor_d(v, off);
        const int oi = __shfl_xor(i, off, 64);
        if (oi != 0x7fffffff && (i == 0x7fffffff || ov > v || (ov == v && oi < i))) { v = ov; i = oi; }
    }
}

__global__ __launch_bounds__(64) void ctc_beam_kernel(const float* __restrict__ logits, int T, int V, const int* __restrict__ lengths,
                                                      const int* __restrict__ row0, int blank, int K, int2* __restrict__ pool,
                                                      int* __restrict__ tokens, float* __restrict__ scores) {
    __shared__ double s_pb[kMaxBeams], s_pnb[kMaxBeams], s_total[kMaxBeams];
    __shared__ int s_node[kMaxBeams], s_parent[kMaxBeams], s_last[kMaxBeams];
    __shared__ int s_top[kMaxTop];
    __shared__ double s_toplp[kMaxTop];
    __shared__ double e_b[kMaxEntries], e_nb[kMaxEntries];
    __shared__ int e_node[kMaxEntries], e_parent[kMaxEntries], e_label[kMaxEntries];
    __shared__ double n_pb[kMaxBeams], n_pnb[kMaxBeams];
    __shared__ int n_node[kMaxBeams], n_parent[kMaxBeams], n_last[kMaxBeams];
    __shared__ int s_count;

    const int b = blockIdx.x, lane = threadIdx.x;
    int L = lengths ? lengths[b] : T;
    L = min(max(L, 0), T);
    const int64_t first = row0 ? (int64_t)row0[b] : (int64_t)b * T;
    int* out = tokens + (int64_t)b * T;
    int2* nodes = pool + (int64_t)b * ((int64_t)T * K + 1);          // node 0 = the empty prefix
    if (L == 0) {
        for (int t = lane; t < T; t += 64) out[t] = -1;
        if (lane == 0) scores[b] = 0.0f;
        return;
    }
    if (lane == 0) {
        s_pb[0] = 0.0; s_pnb[0] = -INFINITY; s_node[0] = 0; s_parent[0] = -1; s_last[0] = -1;
        nodes[0] = make_int2(-1, -1);
    }
    int N = 1, next_node = 1;
    __syncthreads();

    for (int t = 0; t < L; ++t) {
        const float* row = logits + (first + t) * V;
        // 1. log-softmax normaliser
        float mx = -INFINITY;
        for (int v = lane; v < V; v += 64) mx = fmaxf(mx, row[v]);
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) mx = fmaxf(mx, __shfl_xor(mx, off, 64));
        double sum = 0.0;
        for (int v = lane; v < V; v += 64) sum += exp((double)row[v] - (double)mx);
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) sum += shfl_xor_d(sum, off);
        const double lse = (double)mx + log(sum);
        // 2. the R most probable non-blank labels, in order (logit descending, label ascending)
        const int R = min(K + 1, V - 1);
        float bound_v = INFINITY;
        int bound_i = -1;
        for (int j = 0; j < R; ++j) {
            double best = -INFINITY;
            int best_i = 0x7fffffff;
            for (int v = lane; v < V; v += 64) {
                if (v == blank) continue;
                const float x = row[v];
                const bool after = x < bound_v || (x == bound_v && v > bound_i);      // not picked in an earlier round
                if (after && (best_i == 0x7fffffff || (double)x > best)) { best = (double)x; best_i = v; }
            }
            wave_best(best, best_i);
            if (lane == 0) { s_top[j] = best_i; s_toplp[j] = best - lse; }
            bound_v = (float)best;
            bound_i = best_i;
        }
        if (lane < N) s_total[lane] = logaddexp_d(s_pb[lane], s_pnb[lane]);
        __syncthreads();
        const double lp_blank = (double)row[blank] - lse;
        // entries 0..N-1: stay on the prefix (a blank, or a repeat of its last label)
        if (lane < N) {
            e_b[lane] = s_total[lane] + lp_blank;
            e_nb[lane] = s_last[lane] >= 0 ? s_pnb[lane] + ((double)row[s_last[lane]] - lse) : -INFINITY;
            e_node[lane] = s_node[lane];
            e_parent[lane] = s_parent[lane];
            e_label[lane] = s_last[lane];
        }
        if (lane == 0) s_count = N;
        __syncthreads();
        // 3a. extensions that land on a kept prefix q: (beam n with node == parent of q, label last of q)
        if (lane == 0) {
            for (int q = 0; q < N; ++q) {
                if (s_parent[q] < 0) continue;
                for (int n = 0; n < N; ++n)
                    if (s_node[n] == s_parent[q]) {
                        const int c = s_last[q];
                        const double base = (c == s_last[n]) ? s_pb[n] : s_total[n];
                        const double val = base + ((double)row[c] - lse);
                        if (val != -INFINITY) e_nb[q] = logaddexp_d(e_nb[q], val);
                    }
            }
        }
        // 3b. the K best (prefix, label) extensions among N x R candidates; candidate id = n * R + j
        const int ncand = N * R;
        double cbound_v = INFINITY;
        int cbound_i = -1;
        for (int k = 0; k < K; ++k) {
            double best = -INFINITY;
            int best_i = 0x7fffffff;
            for (int id = lane; id < ncand; id += 64) {
                const int n = id / R, j = id - n * R;
                const int c = s_top[j];
                const double val = ((c == s_last[n]) ? s_pb[n] : s_total[n]) + s_toplp[j];
                const bool after = val < cbound_v || (val == cbound_v && id > cbound_i);
                if (after && (best_i == 0x7fffffff || val > best)) { best = val; best_i = id; }
            }
            wave_best(best, best_i);
            if (best_i == 0x7fffffff) break;
            cbound_v = best;
            cbound_i = best_i;
            if (lane == 0 && best != -INFINITY) {
                const int n = best_i / R, c = s_top[best_i - n * R];
                // an extension that lands on a kept prefix was added in 3a
                bool lands = false;
                for (int q = 0; q < N; ++q) lands = lands || (s_parent[q] == s_node[n] && s_last[q] == c);
                if (!lands) {
                    const int e = s_count++;
                    e_b[e] = -INFINITY; e_nb[e] = best; e_node[e] = -1; e_parent[e] = s_node[n]; e_label[e] = c;
                }
            }
        }
        __syncthreads();
        // 4. the K best entries by total probability, in order (probability descending, entry ascending)
        const int ne = s_count;
        const double my = lane < ne ? logaddexp_d(e_b[lane], e_nb[lane]) : -INFINITY;
        double sbound_v = INFINITY;
        int sbound_i = -1;
        int newN = 0;
        for (int k = 0; k < K && k < ne; ++k) {
            double best = my;
            int best_i = (lane < ne && (my < sbound_v || (my == sbound_v && lane > sbound_i))) ? lane : 0x7fffffff;
            wave_best(best, best_i);
            if (best_i == 0x7fffffff) break;
            sbound_v = best;
            sbound_i = best_i;
            if (lane == 0) {
                int node = e_node[best_i];
                if (node < 0) {
                    node = next_node + newN;                  // a new prefix: one pool slot per kept entry of the step
                    nodes[node] = make_int2(e_parent[best_i], e_label[best_i]);
                }
                n_pb[k] = e_b[best_i]; n_pnb[k] = e_nb[best_i]; n_node[k] = node; n_parent[k] = e_parent[best_i]; n_last[k] = e_label[best_i];
            }
            ++newN;
        }
        next_node += K;
        __syncthreads();
        if (lane < newN) {
            s_pb[lane] = n_pb[lane]; s_pnb[lane] = n_pnb[lane]; s_node[lane] = n_node[lane];
            s_parent[lane] = n_parent[lane]; s_last[lane] = n_last[lane];
        }
        N = newN;
        __syncthreads();
    }
    // the best prefix is entry 0; walk it back to the root
    if (lane == 0) {
        int len = 0;
        for (int node = s_node[0]; node > 0; node = nodes[node].x) ++len;
        int pos = len;
        for (int node = s_node[0]; node > 0; node = nodes[node].x) out[--pos] = nodes[node].y;
        s_count = len;
        scores[b] = (float)(logaddexp_d(s_pb[0], s_pnb[0]) / (double)L);
    }
    __syncthreads();
    for (int t = s_count + lane; t < T; t += 64) out[t] = -1;
}

}  // namespace

int launch_ctc_beam(ocrhip_ctx* h, const float* logits, int B, int T, int V, const int* lengths, const int* row0, int blank,
                    int num_beams, int* tokens, float* scores, hipStream_t s) {
    OCR_REQUIRE(row0 == nullptr || lengths != nullptr, "ctc_beam_search: row offsets need lengths");
    OCR_REQUIRE(logits && tokens && scores, "ctc_beam_search: logits, tokens and scores are required");
    OCR_REQUIRE(B >= 0 && T > 0 && V > 1 && blank >= 0 && blank < V, "ctc_beam_search: bad sizes B=%d T=%d V=%d blank=%d", B, T, V, blank);
    OCR_REQUIRE(num_beams >= 1 && num_beams <= kMaxBeams, "ctc_beam_search: num_beams must be 1..%d (got %d)", kMaxBeams, num_beams);
    if (B == 0) return OCRHIP_OK;
    const size_t pool_bytes = (size_t)B * ((size_t)T * num_beams + 1) * sizeof(int2);
    OCR_TRY(h->beam_ws.reserve(pool_bytes));
    ProfScope prof("ctc_beam_search", 0.0, 4.0 * (double)B * T * V * (num_beams + 3), s);
    hipLaunchKernelGGL(ctc_beam_kernel, dim3(B), dim3(64), 0, s, logits, T, V, lengths, row0, blank, num_beams, h->beam_ws.as<int2>(),
                       tokens, scores);
    OCR_HIP_CHECK(hipGetLastError());
    return OCRHIP_OK;
}

}  // namespace ocrhip
