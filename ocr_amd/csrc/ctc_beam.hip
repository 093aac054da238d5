// CTC prefix beam search on the GPU: float64 path probabilities, two kernels.
//
// Stands in for ctc_decode(method='beam_search') of the reference (utils/text/ctc_decoder.py:28-35 ->
// keras.ops.ctc_decode(strategy='beam_search'), the default of BaseOCR.infer, models/ocr/base_ocr.py:112-115,209-222):
// prefix search over the per-step log-softmax of the logits with blank / non-blank path probabilities in log space;
// at every step every label may extend every kept prefix and the `num_beams` most probable prefixes survive; the best
// sequence and its log probability divided by the input length are returned.
//
// What a step needs of its logits row does not depend on the prefixes: the log-softmax normaliser and the
// R = num_beams + 1 most probable non-blank labels (a prefix's num_beams best extensions lie among them: only a repeat
// of its last label is worth less than total + log p).  ctc_beam_rows_kernel computes both for every (sequence, step)
// in parallel, one wave per row.  ctc_beam_kernel then walks a sequence's steps with one wave; per step it
//   1. reads the row's summary, the blank logit and the logit of every kept prefix's last label,
//   2. forms the "stay" entries (a blank, or a repeat of the last label) and adds to a kept prefix q what arrives from
//      its parent by q's last label (an extension that lands on a kept prefix),
//   3. ranks the (prefix n, label rank j) extensions.  The kept prefixes are in descending order of probability and so
//      are the labels: (n', j') with n' <= n and j' <= j precedes (n, j) unless j' is the repeat of n', so (n, j) can be
//      among the num_beams best only if (n + 1) j <= num_beams - at most 66 pairs for 16 beams, a lane each.  The rank
//      of a pair is the number of pairs that precede it in the order (value descending, pair ascending), counted
//      against all pairs at once: no selection rounds,
//   4. ranks {stay, picked extension} entries the same way and keeps the num_beams first, in order.
// Prefixes are nodes of a tree (parent, label) in a per-sequence pool in device memory (read back for the answer only);
// which prefix is which is decided by a hash of the labels, so a (prefix, label) pair names at most one kept prefix
// even when that prefix had left the kept set and come back.  Orders are total (value, then index): deterministic.
// Bound: latency of the longest sequence (a chain of ~1000 instructions per step), not bandwidth.
#include "common.h"

namespace ocrhip {

namespace {

constexpr int kMaxBeams = 16;
constexpr int kMaxTop = kMaxBeams + 1;
constexpr int kMaxPairs = 72;                   // pairs (n, j) with (n + 1) j <= 16, j <= 16: 66
constexpr int kMaxEntries = 2 * kMaxBeams;      // stays + picked extensions

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

// A prefix is known by a 64-bit hash of its labels: the same labels give the same hash whichever tree node holds them.
// (A prefix can leave the kept set while one of its extensions stays, and come back later as a new node; the extension
// then still has to be recognised as its child.  Two different prefixes sharing a hash: 2^-64 per pair.)
constexpr unsigned long long kRootHash = 0x243F6A8885A308D3ull;
__device__ __forceinline__ unsigned long long extend_hash(unsigned long long h, int label) {
    unsigned long long x = (h ^ (unsigned long long)(unsigned)(label + 1)) * 0x9E3779B97F4A7C15ull;
    x ^= x >> 29;
    return x * 0xBF58476D1CE4E5B9ull + 0x632BE59BD9B4E019ull;
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

// One wave per (sequence b, step t < length): lse[b T + t] = log sum exp of the row (float64), and the R most probable
// non-blank labels in the order (logit descending, label ascending): topi / topx [(b T + t) R + rank].
__global__ __launch_bounds__(256) void ctc_beam_rows_kernel(const float* __restrict__ logits, int B, int T, int V,
                                                           const int* __restrict__ lengths, const int* __restrict__ row0, int blank,
                                                           int R, double* __restrict__ lse_out, float* __restrict__ topx,
                                                           int* __restrict__ topi) {
    __shared__ float s_row[4][128];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int64_t idx = (int64_t)blockIdx.x * 4 + wave;
    const int b = (int)(idx / T), t = (int)(idx % T);
    // (no barrier below: a wave that has nothing to do may leave)
    if (b >= B) return;
    int L = lengths ? lengths[b] : T;
    L = min(max(L, 0), T);
    if (t >= L) return;
    const int64_t first = row0 ? (int64_t)row0[b] : (int64_t)b * T;
    const float* row = logits + (first + t) * V;

    float mx = -INFINITY;
    for (int v = lane; v < V; v += 64) mx = fmaxf(mx, row[v]);
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) mx = fmaxf(mx, __shfl_xor(mx, off, 64));
    double sum = 0.0;
    for (int v = lane; v < V; v += 64) sum += exp((double)row[v] - (double)mx);
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) sum += shfl_xor_d(sum, off);
    if (lane == 0) lse_out[idx] = (double)mx + log(sum);

    float* ox = topx + idx * R;
    int* oi = topi + idx * R;
    if (V <= 128) {
        // rank of a label = number of non-blank labels before it in the order; the row goes through LDS so that every
        // lane sees every logit (broadcast reads)
        const float x0 = lane < V ? row[lane] : -INFINITY, x1 = lane + 64 < V ? row[lane + 64] : -INFINITY;
        s_row[wave][lane] = x0;
        s_row[wave][lane + 64] = x1;
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        int r0 = 0, r1 = 0;
        const int l1 = lane + 64;
        for (int v = 0; v < V; ++v) {
            if (v == blank) continue;
            const float y = s_row[wave][v];
            r0 += (y > x0 || (y == x0 && v < lane)) ? 1 : 0;
            r1 += (y > x1 || (y == x1 && v < l1)) ? 1 : 0;
        }
        if (lane < V && lane != blank && r0 < R) { ox[r0] = x0; oi[r0] = lane; }
        if (l1 < V && l1 != blank && r1 < R) { ox[r1] = x1; oi[r1] = l1; }
    } else {
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
            if (lane == 0) { ox[j] = (float)best; oi[j] = best_i; }
            bound_v = (float)best;
            bound_i = best_i;
        }
    }
}

__global__ __launch_bounds__(64) void ctc_beam_kernel(const float* __restrict__ logits, int T, int V, const int* __restrict__ lengths,
                                                      const int* __restrict__ row0, int blank, int K, int R,
                                                      const double* __restrict__ lse_in, const float* __restrict__ topx,
                                                      const int* __restrict__ topi, int2* __restrict__ pool,
                                                      int* __restrict__ tokens, float* __restrict__ scores) {
    // kept prefixes, in descending order of probability
    __shared__ double s_pb[kMaxBeams], s_pnb[kMaxBeams], s_total[kMaxBeams];
    __shared__ int s_node[kMaxBeams], s_parent[kMaxBeams], s_last[kMaxBeams];
    __shared__ unsigned long long s_hash[kMaxBeams], s_phash[kMaxBeams];          // of the prefix, of the prefix without its last label
    // the step's labels
    __shared__ int s_top[kMaxTop];
    __shared__ double s_toplp[kMaxTop];
    // extension pairs that can be among the K best (fixed for the launch), their values of the step
    __shared__ int p_n[kMaxPairs], p_j[kMaxPairs];
    __shared__ double p_val[kMaxPairs];
    // entries: stays [0, N), then picked extensions in the order of their rank
    __shared__ double e_b[kMaxEntries], e_nb[kMaxEntries], e_tot[kMaxEntries];
    __shared__ int e_node[kMaxEntries], e_parent[kMaxEntries], e_label[kMaxEntries];
    __shared__ unsigned long long e_hash[kMaxEntries], e_phash[kMaxEntries];
    __shared__ unsigned s_newmask;
    __shared__ int s_count;
    __shared__ float s_rowl[2][128];          // the step's logits row (V <= 128), the next step's beside it

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
        int np = 0;
        for (int n = 0; n < K; ++n)
            for (int j = 0; j < R; ++j)
                if ((n + 1) * j <= K) { p_n[np] = n; p_j[np] = j; ++np; }          // ascending pair id n R + j
        s_count = np;
        s_pb[0] = 0.0; s_pnb[0] = -INFINITY; s_total[0] = 0.0; s_node[0] = 0; s_parent[0] = -1; s_last[0] = -1;
        s_hash[0] = kRootHash; s_phash[0] = 0ull;
        nodes[0] = make_int2(-1, -1);
    }
    __syncthreads();
    const int n_pairs = s_count;
    // a lane's pairs: lane, lane + 64
    const int my_n0 = lane < n_pairs ? p_n[lane] : 0, my_j0 = lane < n_pairs ? p_j[lane] : 0;
    const int my_n1 = lane + 64 < n_pairs ? p_n[lane + 64] : 0, my_j1 = lane + 64 < n_pairs ? p_j[lane + 64] : 0;
    int N = 1, next_node = 1;

    // What a step reads from memory is requested a step ahead: the row's summary, the blank logit and (V <= 128) the row
    // itself, which then waits in LDS for the lookup of the prefixes' last labels.
    const bool row_in_lds = V <= 128;
    double lse_n = lse_in[(int64_t)b * T];
    float xb_n = logits[first * V + blank];
    int ti_n = 0;
    float tx_n = 0.0f, r0_n = 0.0f, r1_n = 0.0f;
    if (lane < R) { ti_n = topi[(int64_t)b * T * R + lane]; tx_n = topx[(int64_t)b * T * R + lane]; }
    if (row_in_lds) {
        r0_n = lane < V ? logits[first * V + lane] : 0.0f;
        r1_n = lane + 64 < V ? logits[first * V + lane + 64] : 0.0f;
        s_rowl[0][lane] = r0_n;
        s_rowl[0][lane + 64] = r1_n;
    }
    __syncthreads();

    for (int t = 0; t < L; ++t) {
        const float* row = logits + (first + t) * V;
        // 1. this step's operands (requested a step ago), and the requests for the next step
        const double lse = lse_n;
        const float x_blank = xb_n;
        const int last = lane < N ? s_last[lane] : -1;
        const float x_last = last >= 0 ? (row_in_lds ? s_rowl[t & 1][last] : row[last]) : 0.0f;
        if (lane < R) {
            s_top[lane] = ti_n;
            s_toplp[lane] = (double)tx_n - lse;
        }
        if (t + 1 < L) {
            const int64_t nrow = (int64_t)b * T + t + 1;
            const float* next = row + V;
            lse_n = lse_in[nrow];
            xb_n = next[blank];
            if (lane < R) { ti_n = topi[nrow * R + lane]; tx_n = topx[nrow * R + lane]; }
            if (row_in_lds) {
                r0_n = lane < V ? next[lane] : 0.0f;
                r1_n = lane + 64 < V ? next[lane + 64] : 0.0f;
            }
        }
        const double lp_blank = (double)x_blank - lse;
        const double lp_last = (double)x_last - lse;
        // 2. stay entries, plus what lands on a kept prefix from its parent
        if (lane < N) {
            double nb = last >= 0 ? s_pnb[lane] + lp_last : -INFINITY;
            const int parent = s_parent[lane];
            const unsigned long long phash = s_phash[lane];
            int from = -1;
            if (last >= 0)
                for (int n = 0; n < N; ++n)
                    if (s_hash[n] == phash) from = n;
            if (from >= 0) {
                const double val = ((last == s_last[from]) ? s_pb[from] : s_total[from]) + lp_last;
                if (val != -INFINITY) nb = logaddexp_d(nb, val);
            }
            e_b[lane] = s_total[lane] + lp_blank;
            e_nb[lane] = nb;
            e_node[lane] = s_node[lane];
            e_parent[lane] = parent;
            e_label[lane] = last;
            e_hash[lane] = s_hash[lane];
            e_phash[lane] = phash;
        }
        if (lane == 0) s_newmask = 0u;
        __syncthreads();
        // 3. extension pairs: values, ranks, and the K first become entries unless they land on a kept prefix
        double v0 = -INFINITY, v1 = -INFINITY;
        if (lane < n_pairs && my_n0 < N) {
            const int c = s_top[my_j0];
            v0 = ((c == s_last[my_n0]) ? s_pb[my_n0] : s_total[my_n0]) + s_toplp[my_j0];
        }
        if (lane + 64 < n_pairs && my_n1 < N) {
            const int c = s_top[my_j1];
            v1 = ((c == s_last[my_n1]) ? s_pb[my_n1] : s_total[my_n1]) + s_toplp[my_j1];
        }
        if (lane < n_pairs) p_val[lane] = v0;
        if (lane + 64 < n_pairs) p_val[lane + 64] = v1;
        __syncthreads();
        int rank0 = 0, rank1 = 0;
        for (int i = 0; i < n_pairs; ++i) {
            if (p_n[i] >= N) break;              // pairs are ordered by n: the rest belongs to prefixes that do not exist yet
            const double y = p_val[i];
            rank0 += (y > v0 || (y == v0 && i < lane)) ? 1 : 0;
            rank1 += (y > v1 || (y == v1 && i < lane + 64)) ? 1 : 0;
        }
        // (pair index order = pair id order, so "i < lane" is the tie rule of the ids)
        bool new0 = lane < n_pairs && my_n0 < N && rank0 < K && v0 != -INFINITY;
        bool new1 = lane + 64 < n_pairs && my_n1 < N && rank1 < K && v1 != -INFINITY;
        if (new0) {
            const unsigned long long hash_n = s_hash[my_n0];
            const int c = s_top[my_j0];
            for (int q = 0; q < N; ++q) new0 = new0 && !(s_phash[q] == hash_n && s_last[q] == c);
            if (new0) atomicOr(&s_newmask, 1u << rank0);
        }
        if (new1) {
            const unsigned long long hash_n = s_hash[my_n1];
            const int c = s_top[my_j1];
            for (int q = 0; q < N; ++q) new1 = new1 && !(s_phash[q] == hash_n && s_last[q] == c);
            if (new1) atomicOr(&s_newmask, 1u << rank1);
        }
        __syncthreads();
        const unsigned newmask = s_newmask;
        if (new0) {
            const int e = N + __builtin_popcount(newmask & ((1u << rank0) - 1u));
            e_b[e] = -INFINITY; e_nb[e] = v0; e_node[e] = -1; e_parent[e] = s_node[my_n0]; e_label[e] = s_top[my_j0];
            e_phash[e] = s_hash[my_n0]; e_hash[e] = extend_hash(s_hash[my_n0], s_top[my_j0]);
        }
        if (new1) {
            const int e = N + __builtin_popcount(newmask & ((1u << rank1) - 1u));
            e_b[e] = -INFINITY; e_nb[e] = v1; e_node[e] = -1; e_parent[e] = s_node[my_n1]; e_label[e] = s_top[my_j1];
            e_phash[e] = s_hash[my_n1]; e_hash[e] = extend_hash(s_hash[my_n1], s_top[my_j1]);
        }
        const int ne = N + __builtin_popcount(newmask);
        __syncthreads();
        // 4. the K first entries in the order (probability descending, entry ascending) are the new prefixes
        const double my = lane < ne ? logaddexp_d(e_b[lane], e_nb[lane]) : -INFINITY;
        if (lane < ne) e_tot[lane] = my;
        __syncthreads();
        int rank = 0;
        for (int i = 0; i < ne; ++i) {
            const double y = e_tot[i];
            rank += (y > my || (y == my && i < lane)) ? 1 : 0;
        }
        const int newN = min(K, ne);
        if (lane < ne && rank < K) {
            int node = e_node[lane];
            const int parent = e_parent[lane], label = e_label[lane];
            if (node < 0) {
                node = next_node + rank;                  // a new prefix: one pool slot per kept entry of the step
                nodes[node] = make_int2(parent, label);
            }
            s_pb[rank] = e_b[lane]; s_pnb[rank] = e_nb[lane]; s_total[rank] = my;
            s_node[rank] = node; s_parent[rank] = parent; s_last[rank] = label;
            s_hash[rank] = e_hash[lane]; s_phash[rank] = e_phash[lane];
        }
        next_node += K;
        N = newN;
        if (row_in_lds) {
            s_rowl[(t + 1) & 1][lane] = r0_n;
            s_rowl[(t + 1) & 1][lane + 64] = r1_n;
        }
        __syncthreads();
    }
    // the best prefix is entry 0; walk it back to the root
    if (lane == 0) {
        int len = 0;
        for (int node = s_node[0]; node > 0; node = nodes[node].x) ++len;
        int pos = len;
        for (int node = s_node[0]; node > 0; node = nodes[node].x) out[--pos] = nodes[node].y;
        s_count = len;
        scores[b] = (float)(s_total[0] / (double)L);
    }
    __syncthreads();
    for (int t = s_count + lane; t < T; t += 64) out[t] = -1;
}

size_t align256(size_t v) { return (v + 255) & ~(size_t)255; }

}  // namespace

int launch_ctc_beam(ocrhip_ctx* h, const float* logits, int B, int T, int V, const int* lengths, const int* row0, int blank,
                    int num_beams, int* tokens, float* scores, hipStream_t s) {
    OCR_REQUIRE(row0 == nullptr || lengths != nullptr, "ctc_beam_search: row offsets need lengths");
    OCR_REQUIRE(logits && tokens && scores, "ctc_beam_search: logits, tokens and scores are required");
    OCR_REQUIRE(B >= 0 && T > 0 && V > 1 && blank >= 0 && blank < V, "ctc_beam_search: bad sizes B=%d T=%d V=%d blank=%d", B, T, V, blank);
    OCR_REQUIRE(num_beams >= 1 && num_beams <= kMaxBeams, "ctc_beam_search: num_beams must be 1..%d (got %d)", kMaxBeams, num_beams);
    if (B == 0) return OCRHIP_OK;
    const int R = std::min(num_beams + 1, V - 1);
    const size_t rows = (size_t)B * T;
    const size_t pool_bytes = align256((size_t)B * ((size_t)T * num_beams + 1) * sizeof(int2));
    const size_t lse_bytes = align256(rows * sizeof(double)), top_bytes = align256(rows * R * sizeof(float));
    OCR_TRY(h->beam_ws.reserve(pool_bytes + lse_bytes + 2 * top_bytes));
    unsigned char* ws = h->beam_ws.as<unsigned char>();
    int2* pool = reinterpret_cast<int2*>(ws);
    double* lse = reinterpret_cast<double*>(ws + pool_bytes);
    float* topx = reinterpret_cast<float*>(ws + pool_bytes + lse_bytes);
    int* topi = reinterpret_cast<int*>(ws + pool_bytes + lse_bytes + top_bytes);
    {
        ProfScope prof("ctc_beam_rows", 0.0, 4.0 * (double)rows * V + (double)rows * (8.0 + 8.0 * R), s);
        hipLaunchKernelGGL(ctc_beam_rows_kernel, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, s, logits, B, T, V, lengths, row0, blank, R,
                           lse, topx, topi);
        OCR_HIP_CHECK(hipGetLastError());
    }
    ProfScope prof("ctc_beam_search", 0.0, (double)rows * (8.0 + 8.0 * R + 4.0 * (num_beams + 1)) + 4.0 * (double)rows, s);
    hipLaunchKernelGGL(ctc_beam_kernel, dim3(B), dim3(64), 0, s, logits, T, V, lengths, row0, blank, num_beams, R, lse, topx, topi, pool,
                       tokens, scores);
    OCR_HIP_CHECK(hipGetLastError());
    return OCRHIP_OK;
}

}  // namespace ocrhip
