// Locality-aware NMS (iterative all-pairs merge) + matrix NMS, one persistent workgroup per image.
//
// Restates utils/image/bounding_box/locality_aware_nms.py:22-126 and
// non_max_suppression.py:117-138,191-220,247-259.  The reference sorts the boxes by score and
// works on an N x N float IoU matrix per iteration; here the boxes stay in input order (row-major
// over the EAST map, i.e. sorted by y, which makes 64-box tiles spatially compact) and "ranked
// better" is evaluated from the key (score, input index) directly, so no sort and no matrix exist:
//
//   merge iteration (all rows read the OLD boxes, as in the reference's single tensor expression):
//     pass A  has_upper[t] = some better-ranked u has IoU(u,t) >= thr      (reference: column max)
//             has_lower[t] = some worse-ranked  u has IoU(t,u) >= thr
//     pass B  t absorbs every worse-ranked overlapping u when !has_upper[t]   (union / average)
//             t is zeroed when a better-ranked overlapping u has !has_upper[u]
//             continue while a zeroed t had has_lower[t]
//   matrix NMS (only when nms_thr < merge_thr): the reference's sorted-NMS fixed point; its rows
//     are "alive" boxes, a box dies when an alive better-ranked box with no alive better-ranked
//     overlap of its own overlaps it; iterated until nothing dies, which selects the same boxes as a
//     greedy scan in rank order.
//
// IoU keeps the reference's operation order: union = (area_better - inter) + area_worse, 0/0 -> 0,
// compared with `>=`.  Compiled with -ffp-contract=off.
//
// Pairs are pruned by tile bounding boxes (a pair of tiles whose bounds do not intersect cannot
// contain IoU > 0); pruning is switched off when a threshold is <= 0, where every pair counts.
// Suppressed boxes leave the active list after every iteration, so the cost is dominated by the
// first iteration.  This stage is compute/latency bound (N^2 pair tests on ~20 B per box), not HBM
// bound.
#include <algorithm>

#include "common.h"

namespace ocrhip {

namespace {

constexpr int kThreads = 1024;
constexpr int kWaves = kThreads / 64;
constexpr int kCells = 64;                 // centre grid of the spatial bucketing: kCells^2 = 4 * kThreads

constexpr unsigned kHasUpper = 1u, kHasLower = 2u, kSupp = 4u, kAlive = 8u, kKill = 16u;

struct Image {
    float4* cur;          // current boxes, input order
    float4* nxt;          // next boxes
    unsigned* key;        // order-preserving score bits
    unsigned* flags;      // per box flag bits
    int* act;             // active indices (input order)
    int* act2;
    float4* bounds;       // per 64-entry tile of act: (min x1, min y1, max x2, max y2)
    float4* bounds2;      // per tile two such rectangles: its entries before and from its largest jump (see build_bounds)
    int* kept;            // indices of the kept boxes
    int* hdr;             // [0] = loop condition published by the split first iteration
    float* wcur;          // weighted merge: weight of every box (its score plus the weights of what it absorbed)
    float* wnxt;
};

__device__ __forceinline__ unsigned order_key(float s) {
    const unsigned b = __float_as_uint(s);
    return (b & 0x80000000u) ? ~b : (b | 0x80000000u);
}

__device__ __forceinline__ bool ranked_better(unsigned key_u, int u, unsigned key_t, int t) {
    return key_u > key_t || (key_u == key_t && u < t);
}

__device__ __forceinline__ float box_area(const float4 b) { return (b.w - b.y) * (b.z - b.x); }

__device__ __forceinline__ float bcast(float v, int k) {
    return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), k));
}
__device__ __forceinline__ int bcast(int v, int k) { return __builtin_amdgcn_readlane(v, k); }
__device__ __forceinline__ unsigned bcast(unsigned v, int k) {
    return (unsigned)__builtin_amdgcn_readlane((int)v, k);
}

__device__ __forceinline__ float wave_min(float v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v = fminf(v, __shfl_xor(v, off, 64));
    return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v = fmaxf(v, __shfl_xor(v, off, 64));
    return v;
}

__device__ __forceinline__ bool rects_meet(const float4 p, const float4 q) {
    return (p.x <= q.z) && (q.x <= p.z) && (p.y <= q.w) && (q.y <= p.w);
}

// Bounds of every 64-entry tile of the active list, and of its two parts.  A tile of the cell-sorted list nearly always
// holds the last candidates of one word and the first of the next one; when the two are far apart the tile's bounds
// cover everything between them, and a target tile's pair tests run over every box inside its bounds.  So the tile is
// also cut at the largest jump between the centres of consecutive entries, and both parts get bounds of their own:
// a box that meets neither part cannot overlap any box of the tile.
__device__ void build_bounds(const Image& im, const float4* boxes, const int* act, int n_act) {
    n_act = __builtin_amdgcn_readfirstlane(n_act);
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
    const int ntiles = (n_act + 63) >> 6;
    for (int T = wave; T < ntiles; T += kWaves) {
        const int e = T * 64 + lane;
        const bool valid = e < n_act;
        float4 b = make_float4(INFINITY, INFINITY, -INFINITY, -INFINITY);
        if (valid) b = boxes[act[e]];
        const float x1 = wave_min(b.x), y1 = wave_min(b.y), x2 = wave_max(b.z), y2 = wave_max(b.w);
        if (lane == 0) im.bounds[T] = make_float4(x1, y1, x2, y2);
        // jump from the previous entry (0 for the first entry and beyond the list); the cut goes before the entry with
        // the largest one (the lowest such entry; entry 32 when all centres coincide)
        const float cx = b.x + b.z, cy = b.y + b.w;
        const float px = __shfl_up(cx, 1, 64), py = __shfl_up(cy, 1, 64);
        float jump = (valid && lane > 0) ? fabsf(cx - px) + fabsf(cy - py) : 0.0f;
        if (!(jump == jump)) jump = 0.0f;
        const float top = wave_max(jump);
        const unsigned long long at = __ballot(jump == top && lane > 0);
        const int cut = (top > 0.0f && at) ? __builtin_ctzll(at) : 32;
        const bool first = lane < cut;
        const float4 inf = make_float4(INFINITY, INFINITY, -INFINITY, -INFINITY);
        const float4 ba = first ? b : inf, bb = first ? inf : b;
        const float ax1 = wave_min(ba.x), ay1 = wave_min(ba.y), ax2 = wave_max(ba.z), ay2 = wave_max(ba.w);
        const float bx1 = wave_min(bb.x), by1 = wave_min(bb.y), bx2 = wave_max(bb.z), by2 = wave_max(bb.w);
        if (lane == 0) {
            im.bounds2[2 * T] = make_float4(ax1, ay1, ax2, ay2);
            im.bounds2[2 * T + 1] = make_float4(bx1, by1, bx2, by2);
        }
    }
}

// MODE 0: merge pass A   (writes kHasUpper / kHasLower)
// MODE 1: merge pass B   (writes nxt, kSupp; sets *cond)
// MODE 2: NMS pass A     (writes kHasUpper among alive boxes)
// MODE 3: NMS pass B     (writes kKill; sets *cond)
// NSUB = 1: a wave owns a target tile (tiles tile0 + wave, + tile_stride, ...).  NSUB = 4 (MODE 0, and MODE 1 with the
// union merge; a workgroup of exactly 4 waves): the workgroup owns a target tile (tiles tile0, + tile_stride, ...) and
// wave w compares it with the candidate tiles S = w (mod 4); the four partial results - flag bits, minima and maxima,
// all independent of the order - are combined through `parts` (3 x 64 x 5 words of LDS).  A target tile's time is the
// number of candidate boxes inside its bounds, 200 on average and up to 1400: with a tile per wave a pass lasts as
// long as its slowest wave, 4.5x the average.
template <int MODE, int NSUB = 1>
__device__ void scan_pairs(const Image& im, const int* act, int n_act, float thr, bool cull, int merge_method,
                           int* cond, int tile0 = 0, int tile_stride = kWaves, float* parts = nullptr) {
    static_assert(NSUB == 1 || (NSUB == 4 && MODE <= 1), "only the merge passes are split");
    n_act = __builtin_amdgcn_readfirstlane(n_act);       // wave-uniform by construction
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
    const int ntiles = (n_act + 63) >> 6;
    const float4* boxes = im.cur;

    for (int T = tile0 + (NSUB == 1 ? wave : 0); T < ntiles; T += tile_stride) {
        const int e = T * 64 + lane;
        const bool t_valid = e < n_act;
        const int t = t_valid ? act[e] : 0;
        const float4 bt = t_valid ? boxes[t] : make_float4(0.f, 0.f, 0.f, 0.f);
        const unsigned key_t = t_valid ? im.key[t] : 0u;
        const unsigned fl_t = t_valid ? im.flags[t] : 0u;
        const float area_t = box_area(bt);
        const float4 tb = im.bounds[T];
        const float4 tba = im.bounds2[2 * T], tbb = im.bounds2[2 * T + 1];      // the tile's two parts

        bool has_upper = false, has_lower = false, hit_by_clean = false;
        float mnx = INFINITY, mny = INFINITY, mxx = 0.f, mxy = 0.f;     // union accumulators
        float sx1 = 0.f, sy1 = 0.f, sx2 = 0.f, sy2 = 0.f;               // average / weighted accumulators
        float sw = 0.f;                                                  // weighted: sum of the absorbed weights
        int n_merged = 0;
        const bool weighted = MODE == 1 && merge_method == OCRHIP_MERGE_WEIGHTED;
        const float w_t = (weighted && t_valid) ? im.wcur[t] : 0.f;

        for (int S0 = 0; S0 < ntiles; S0 += 64) {
            unsigned long long cand;
            {
                const int S = S0 + lane;
                bool c = S < ntiles;
                if (c && cull) {
                    const float4 sb = im.bounds[S];
                    c = rects_meet(sb, tb);
                    if (c) {          // part by part: the union of two distant parts meets much that neither part does
                        const float4 sa = im.bounds2[2 * S], sbb = im.bounds2[2 * S + 1];
                        c = rects_meet(sa, tba) || rects_meet(sa, tbb) || rects_meet(sbb, tba) || rects_meet(sbb, tbb);
                    }
                }
                cand = __ballot(c);
                if (NSUB > 1) cand &= 0x1111111111111111ull << wave;      // S0 is a multiple of 64: bit i is tile S0 + i
            }
            while (cand) {
                const int S = S0 + __builtin_ctzll(cand);
                cand &= cand - 1;
                const int se = S * 64 + lane;
                const int cnt = min(64, n_act - S * 64);
                const bool u_valid = se < n_act;
                const int u_l = u_valid ? act[se] : 0;
                const float4 bu_l = u_valid ? boxes[u_l] : make_float4(0.f, 0.f, 0.f, 0.f);
                const unsigned key_l = u_valid ? im.key[u_l] : 0u;
                const unsigned fl_l = u_valid ? im.flags[u_l] : 0u;
                const float area_l = box_area(bu_l);
                const float w_l = (weighted && u_valid) ? im.wcur[u_l] : 0.f;
                // only the boxes of S that reach into the bounds of one of tile T's parts can overlap one of its boxes (thr > 0)
                unsigned long long todo = (cnt >= 64) ? ~0ull : ((1ull << cnt) - 1ull);
                if (cull) todo &= __ballot(u_valid && (rects_meet(bu_l, tba) || rects_meet(bu_l, tbb)));
                while (todo) {
                    const int k = __builtin_ctzll(todo);
                    todo &= todo - 1;
                    const float ux1 = bcast(bu_l.x, k), uy1 = bcast(bu_l.y, k);
                    const float ux2 = bcast(bu_l.z, k), uy2 = bcast(bu_l.w, k);
                    const float iw = fmaxf(0.f, fminf(bt.z, ux2) - fmaxf(bt.x, ux1));
                    const float ih = fmaxf(0.f, fminf(bt.w, uy2) - fmaxf(bt.y, uy1));
                    const float inter = iw * ih;
                    if (cull && !__any(inter > 0.f)) continue;    // thr > 0: no overlap, no hit
                    const float area_u = bcast(area_l, k);
                    const unsigned key_u = bcast(key_l, k);
                    const int u = bcast(u_l, k);
                    const unsigned fl_u = bcast(fl_l, k);
                    const bool u_better = ranked_better(key_u, u, key_t, t);
                    const bool t_better = ranked_better(key_t, t, key_u, u);
                    const float a1 = u_better ? area_u : area_t;
                    const float a2 = u_better ? area_t : area_u;
                    const float uni = (a1 - inter) + a2;
                    const float iou = (uni != 0.f) ? (inter / uni) : 0.f;
                    const bool hit = t_valid && (iou >= thr);
                    if (MODE == 0) {
                        has_upper |= hit && u_better;
                        has_lower |= hit && t_better;
                    } else if (MODE == 1) {
                        hit_by_clean |= hit && u_better && !(fl_u & kHasUpper);
                        if (hit && t_better && !(fl_t & kHasUpper)) {
                            if (weighted) {
                                const float w_u = bcast(w_l, k);
                                sx1 = sx1 + w_u * ux1; sy1 = sy1 + w_u * uy1; sx2 = sx2 + w_u * ux2; sy2 = sy2 + w_u * uy2;
                                sw = sw + w_u;
                            } else {
                                mnx = fminf(mnx, ux1); mny = fminf(mny, uy1);
                                mxx = fmaxf(mxx, ux2); mxy = fmaxf(mxy, uy2);
                                sx1 = sx1 + ux1; sy1 = sy1 + uy1; sx2 = sx2 + ux2; sy2 = sy2 + uy2;
                            }
                            n_merged++;
                        }
                    } else if (MODE == 2) {
                        has_upper |= hit && u_better && (fl_u & kAlive);
                    } else {
                        hit_by_clean |= hit && u_better && (fl_u & kAlive) && !(fl_u & kHasUpper);
                    }
                }
            }
        }

        if (NSUB > 1) {
            // waves 1..3 hand their partial results to wave 0 (every wave of the workgroup walks the same tiles, so
            // the barriers match)
            if (wave > 0) {
                float* q = parts + ((wave - 1) * 64 + lane) * 5;
                q[0] = __uint_as_float((has_upper ? 1u : 0u) | (has_lower ? 2u : 0u) | (hit_by_clean ? 4u : 0u));
                q[1] = mnx; q[2] = mny; q[3] = mxx; q[4] = mxy;
            }
            __syncthreads();
            if (wave == 0) {
#pragma unroll
                for (int w = 0; w < NSUB - 1; ++w) {
                    const float* q = parts + (w * 64 + lane) * 5;
                    const unsigned bits = __float_as_uint(q[0]);
                    has_upper |= (bits & 1u) != 0; has_lower |= (bits & 2u) != 0; hit_by_clean |= (bits & 4u) != 0;
                    mnx = fminf(mnx, q[1]); mny = fminf(mny, q[2]); mxx = fmaxf(mxx, q[3]); mxy = fmaxf(mxy, q[4]);
                }
            }
            __syncthreads();
            if (wave > 0) continue;
        }
        if (!t_valid) continue;
        if (MODE == 0) {
            im.flags[t] = (fl_t & ~(kHasUpper | kHasLower | kSupp)) | (has_upper ? kHasUpper : 0u) |
                          (has_lower ? kHasLower : 0u);
        } else if (MODE == 1) {
            float4 nb;
            float w_new = w_t;
            if (merge_method == OCRHIP_MERGE_UNION) {
                nb = make_float4(fminf(bt.x, mnx), fminf(bt.y, mny), fmaxf(bt.z, mxx), fmaxf(bt.w, mxy));
            } else if (weighted) {
                // score-weighted mean of the box and what it absorbs; a box that absorbs nothing stays bit for bit
                w_new = w_t + sw;
                nb = (n_merged == 0) ? bt
                                     : make_float4((w_t * bt.x + sx1) / w_new, (w_t * bt.y + sy1) / w_new,
                                                   (w_t * bt.z + sx2) / w_new, (w_t * bt.w + sy2) / w_new);
            } else {
                const float cntf = (float)(1 + n_merged);
                nb = make_float4((bt.x + sx1) / cntf, (bt.y + sy1) / cntf, (bt.z + sx2) / cntf, (bt.w + sy2) / cntf);
            }
            if (hit_by_clean) {
                nb = make_float4(0.f, 0.f, 0.f, 0.f);
                im.flags[t] = fl_t | kSupp;
                if (fl_t & kHasLower) *cond = 1;
                w_new = 0.f;
            }
            if (weighted) im.wnxt[t] = w_new;
            im.nxt[t] = nb;
        } else if (MODE == 2) {
            im.flags[t] = (fl_t & ~(kHasUpper | kKill)) | (has_upper ? kHasUpper : 0u);
        } else {
            if (hit_by_clean && (fl_t & kAlive)) {
                im.flags[t] = fl_t | kKill;
                *cond = 1;
            }
        }
    }
}

// Ordered compaction of act -> act2 keeping the entries for which (flags & drop_bit) == 0.
// Returns the new count (same value in every thread).  Boxes dropped with zero_dropped get both
// box buffers cleared.
__device__ int compact(const Image& im, const int* act, int* act_out, int n_act, unsigned drop_bit,
                       bool zero_dropped, int* s_wave, int* s_base) {
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    if (threadIdx.x == 0) *s_base = 0;
    __syncthreads();
    for (int c0 = 0; c0 < n_act; c0 += kThreads) {
        const int e = c0 + threadIdx.x;
        int t = 0;
        bool keep = false;
        if (e < n_act) {
            t = act[e];
            keep = !(im.flags[t] & drop_bit);
            if (!keep && zero_dropped) {
                im.cur[t] = make_float4(0.f, 0.f, 0.f, 0.f);
                im.nxt[t] = make_float4(0.f, 0.f, 0.f, 0.f);
            }
        }
        const unsigned long long m = __ballot(keep);
        if (lane == 0) s_wave[wave] = __popcll(m);
        __syncthreads();
        int pos = *s_base;
        for (int w = 0; w < wave; ++w) pos += s_wave[w];
        if (keep) act_out[pos + __popcll(m & ((1ull << lane) - 1ull))] = t;
        __syncthreads();
        if (threadIdx.x == 0) {
            int tot = 0;
            for (int w = 0; w < kWaves; ++w) tot += s_wave[w];
            *s_base += tot;
        }
        __syncthreads();
    }
    return *s_base;
}

struct Params {
    const float* xyxy;
    const float* scores;
    const int* counts;
    int cap;
    float merge_thr, nms_thr;
    int merge_method, max_iter, max_out;
    float* merged;
    unsigned char* keep;
    float* out_boxes;
    float* out_scores;
    int* out_src;
    int* out_counts;
    int* n_iter;
    char* ws;
    size_t ws_stride;      // bytes per image
    // 0: everything in one launch.  The first merge iteration (all N candidates) dominates the cost, so for
    // large capacities it is spread over several workgroups per image by separate launches:
    // 1 = set-up (+ tile bounds), 2 = pass A and 3 = pass B on gridDim.y workgroups per image,
    // 4 = the rest (compaction, further iterations, NMS, outputs) on one workgroup per image.
    int phase;
};

__global__ __launch_bounds__(kThreads) void lanms_kernel(Params p) {
    __shared__ int s_cond;
    __shared__ int s_base;
    __shared__ int s_wave[kWaves];
    __shared__ int s_hist[kCells * kCells];

    const int b = blockIdx.x;
    const int cap = p.cap;
    const int n = min(p.counts[b], cap);
    const int64_t o = (int64_t)b * cap;

    if (n <= 0) {
        if (threadIdx.x == 0 && (p.phase == 0 || p.phase == 4)) {
            p.out_counts[b] = 0;
            if (p.n_iter) p.n_iter[b] = 0;
        }
        return;
    }

    Image im;
    {
        char* w = p.ws + (size_t)b * p.ws_stride;
        const size_t c = (size_t)cap;
        im.hdr = reinterpret_cast<int*>(w);                    w += 64;
        im.cur = reinterpret_cast<float4*>(w);                 w += c * sizeof(float4);
        im.nxt = reinterpret_cast<float4*>(w);                 w += c * sizeof(float4);
        im.bounds = reinterpret_cast<float4*>(w);              w += ((c + 63) / 64) * sizeof(float4);
        im.bounds2 = reinterpret_cast<float4*>(w);             w += 2 * ((c + 63) / 64) * sizeof(float4);
        im.key = reinterpret_cast<unsigned*>(w);               w += c * sizeof(unsigned);
        im.flags = reinterpret_cast<unsigned*>(w);             w += c * sizeof(unsigned);
        im.act = reinterpret_cast<int*>(w);                    w += c * sizeof(int);
        im.act2 = reinterpret_cast<int*>(w);                   w += c * sizeof(int);
        im.kept = reinterpret_cast<int*>(w);                   w += c * sizeof(int);
        im.wcur = reinterpret_cast<float*>(w);                 w += c * sizeof(float);
        im.wnxt = reinterpret_cast<float*>(w);
    }

    const bool cull_merge = p.merge_thr > 0.f;
    if (p.phase == 2 || p.phase == 3) {      // one pass of the first iteration, tiles strided over gridDim.y workgroups
        if (blockDim.x == 256) {             // workgroups of 4 waves that share a target tile (union merge)
            float* parts = reinterpret_cast<float*>(s_hist);
            if (p.phase == 2)
                scan_pairs<0, 4>(im, im.act, n, p.merge_thr, cull_merge, p.merge_method, im.hdr, blockIdx.y, gridDim.y, parts);
            else
                scan_pairs<1, 4>(im, im.act, n, p.merge_thr, cull_merge, p.merge_method, im.hdr, blockIdx.y, gridDim.y, parts);
            return;
        }
        const int tile0 = blockIdx.y * kWaves, stride = gridDim.y * kWaves;
        if (p.phase == 2)
            scan_pairs<0>(im, im.act, n, p.merge_thr, cull_merge, p.merge_method, im.hdr, tile0, stride);
        else
            scan_pairs<1>(im, im.act, n, p.merge_thr, cull_merge, p.merge_method, im.hdr, tile0, stride);
        return;
    }

    const float4* in_boxes = reinterpret_cast<const float4*>(p.xyxy) + o;
    if (p.phase != 4)
    for (int t = threadIdx.x; t < n; t += kThreads) {
        const float4 bx = in_boxes[t];
        im.cur[t] = bx;
        im.nxt[t] = bx;
        im.key[t] = p.scores ? order_key(p.scores[o + t]) : 0u;
        if (p.merge_method == OCRHIP_MERGE_WEIGHTED) im.wcur[t] = im.wnxt[t] = p.scores ? p.scores[o + t] : 1.0f;
        im.flags[t] = 0u;
        im.act[t] = t;
    }
    __syncthreads();

    // Physical order of the active list.  Results of the union merge do not depend on it (ranks come
    // from the keys; min / max / OR are order independent), so the boxes are bucketed by the grid cell of
    // their centre: all candidates of a word predict nearly the same box and land in the same cell,
    // which makes the 64-entry tiles spatially tight and the tile-bound pruning effective.  The average
    // merge sums in list order, so it keeps the input order to stay reproducible bit for bit.
    if (p.phase != 4 && p.merge_method == OCRHIP_MERGE_UNION && n > 256) {
        for (int i = threadIdx.x; i < kCells * kCells; i += kThreads) s_hist[i] = 0;
        __syncthreads();
        for (int t = threadIdx.x; t < n; t += kThreads) {
            const float4 bx = im.cur[t];
            const float cx = 0.5f * (bx.x + bx.z), cy = 0.5f * (bx.y + bx.w);
            const int ix = (cx == cx) ? (int)fminf(fmaxf(cx * kCells, 0.f), (float)(kCells - 1)) : 0;
            const int iy = (cy == cy) ? (int)fminf(fmaxf(cy * kCells, 0.f), (float)(kCells - 1)) : 0;
            const int cell = iy * kCells + ix;
            im.act2[t] = cell;
            atomicAdd(&s_hist[cell], 1);
        }
        __syncthreads();
        // exclusive scan of the histogram: 4 cells per thread
        {
            const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
            const int c0 = threadIdx.x * 4;
            const int h0 = s_hist[c0], h1 = s_hist[c0 + 1], h2 = s_hist[c0 + 2], h3 = s_hist[c0 + 3];
            const int tot = h0 + h1 + h2 + h3;
            int incl = tot;
#pragma unroll
            for (int off = 1; off < 64; off <<= 1) {
                const int v = __shfl_up(incl, off, 64);
                if (lane >= off) incl += v;
            }
            if (lane == 63) s_wave[wave] = incl;
            __syncthreads();
            int base = 0;
            for (int w = 0; w < wave; ++w) base += s_wave[w];
            const int excl = base + incl - tot;
            s_hist[c0] = excl;
            s_hist[c0 + 1] = excl + h0;
            s_hist[c0 + 2] = excl + h0 + h1;
            s_hist[c0 + 3] = excl + h0 + h1 + h2;
        }
        __syncthreads();
        for (int t = threadIdx.x; t < n; t += kThreads) {
            const int pos = atomicAdd(&s_hist[im.act2[t]], 1);
            im.act[pos] = t;
        }
        __syncthreads();
    }

    if (p.phase == 1) {                      // set-up launch: the passes of the first iteration follow as launches
        if (threadIdx.x == 0) im.hdr[0] = 0;
        build_bounds(im, im.cur, im.act, n);
        return;
    }
    const int max_iter = p.max_iter > 0 ? p.max_iter : n;
    int n_act = n;
    int iter = 0;
    bool cond = true;
    while (cond && iter < max_iter) {
        if (p.phase == 4 && iter == 0) {
            cond = im.hdr[0] != 0;           // passes A and B of this iteration ran as separate launches
        } else {
            if (threadIdx.x == 0) s_cond = 0;
            build_bounds(im, im.cur, im.act, n_act);
            __syncthreads();
            scan_pairs<0>(im, im.act, n_act, p.merge_thr, cull_merge, p.merge_method, &s_cond);
            __syncthreads();
            scan_pairs<1>(im, im.act, n_act, p.merge_thr, cull_merge, p.merge_method, &s_cond);
            __syncthreads();
            cond = s_cond != 0;
        }
        // cur <-> nxt; suppressed boxes leave the active list (they are all-zero from now on)
        float4* tmp = im.cur; im.cur = im.nxt; im.nxt = tmp;
        float* tw = im.wcur; im.wcur = im.wnxt; im.wnxt = tw;
        if (cull_merge) {
            n_act = compact(im, im.act, im.act2, n_act, kSupp, true, s_wave, &s_base);
            int* ta = im.act; im.act = im.act2; im.act2 = ta;
        } else {
            // every pair counts when thr <= 0: keep all boxes listed, but zero the suppressed ones in both buffers
            for (int e = threadIdx.x; e < n_act; e += kThreads) {
                const int t = im.act[e];
                if (im.flags[t] & kSupp) im.nxt[t] = make_float4(0.f, 0.f, 0.f, 0.f);
            }
            __syncthreads();
        }
        ++iter;
    }
    if (threadIdx.x == 0 && p.n_iter) p.n_iter[b] = iter;

    // ---- matrix NMS on the merged boxes (no re-sort: same rank)
    const bool do_nms = p.nms_thr < p.merge_thr;
    for (int t = threadIdx.x; t < n; t += kThreads) im.flags[t] = 0u;
    __syncthreads();
    if (do_nms) {
        const bool cull_nms = p.nms_thr > 0.f;
        if (!cull_nms) {            // every box takes part
            for (int t = threadIdx.x; t < n; t += kThreads) im.act[t] = t;
            n_act = n;
            __syncthreads();
        }
        for (int e = threadIdx.x; e < n_act; e += kThreads) im.flags[im.act[e]] = kAlive;
        __syncthreads();
        build_bounds(im, im.cur, im.act, n_act);
        __syncthreads();
        int guard = 0;
        bool again = true;
        while (again && guard <= n) {
            if (threadIdx.x == 0) s_cond = 0;
            __syncthreads();
            scan_pairs<2>(im, im.act, n_act, p.nms_thr, cull_nms, 0, &s_cond);
            __syncthreads();
            scan_pairs<3>(im, im.act, n_act, p.nms_thr, cull_nms, 0, &s_cond);
            __syncthreads();
            again = s_cond != 0;
            for (int e = threadIdx.x; e < n_act; e += kThreads) {
                const int t = im.act[e];
                const unsigned f = im.flags[t];
                if (f & kKill) im.flags[t] = f & ~(kAlive | kKill);
            }
            __syncthreads();
            ++guard;
        }
    }

    // ---- valid mask, input-order outputs, list of kept boxes
    if (threadIdx.x == 0) s_base = 0;
    __syncthreads();
    for (int c0 = 0; c0 < n; c0 += kThreads) {
        const int t = c0 + threadIdx.x;
        bool keep = false;
        if (t < n) {
            const float4 bx = im.cur[t];
            keep = (bx.x > 0.f) || (bx.y > 0.f) || (bx.z > 0.f) || (bx.w > 0.f);
            if (do_nms) keep = keep && (im.flags[t] & kAlive);
            if (p.merged) reinterpret_cast<float4*>(p.merged)[o + t] = bx;
        }
        const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
        const unsigned long long m = __ballot(keep);
        if (lane == 0) s_wave[wave] = __popcll(m);
        __syncthreads();
        int pos = s_base;
        for (int w = 0; w < wave; ++w) pos += s_wave[w];
        if (keep) im.kept[pos + __popcll(m & ((1ull << lane) - 1ull))] = t;
        __syncthreads();
        if (threadIdx.x == 0) {
            int tot = 0;
            for (int w = 0; w < kWaves; ++w) tot += s_wave[w];
            s_base += tot;
        }
        __syncthreads();
    }
    const int n_kept = s_base;

    // ---- rank the kept boxes among themselves and write them in rank order
    const int limit = (do_nms && p.max_out > 0) ? min(p.max_out, n_kept) : n_kept;
    for (int i = threadIdx.x; i < n_kept; i += kThreads) {
        const int t = im.kept[i];
        const unsigned key_t = im.key[t];
        int rank = 0;
        for (int j = 0; j < n_kept; ++j) {
            const int u = im.kept[j];
            rank += ranked_better(im.key[u], u, key_t, t) ? 1 : 0;
        }
        const bool in_limit = rank < limit;
        if (in_limit) {
            reinterpret_cast<float4*>(p.out_boxes)[o + rank] = im.cur[t];
            if (p.out_scores) p.out_scores[o + rank] = p.scores ? p.scores[o + t] : 1.0f;
            if (p.out_src) p.out_src[o + rank] = t;
        }
        im.flags[t] = in_limit ? kAlive : 0u;      // reuse as the final keep flag
    }
    __syncthreads();
    if (p.keep) {
        for (int t = threadIdx.x; t < n; t += kThreads) p.keep[o + t] = 0;
        __syncthreads();
        for (int i = threadIdx.x; i < n_kept; i += kThreads) {
            const int t = im.kept[i];
            p.keep[o + t] = (im.flags[t] & kAlive) ? 1 : 0;
        }
    }
    if (threadIdx.x == 0) p.out_counts[b] = limit;
}

size_t ws_bytes_per_image(int cap) {
    const size_t c = (size_t)cap;
    size_t bytes = 64 + 2 * c * sizeof(float4) + 3 * ((c + 63) / 64) * sizeof(float4) + 7 * c * sizeof(int);
    return (bytes + 255) & ~(size_t)255;
}

}  // namespace

int launch_lanms(ocrhip_ctx* h, const float* xyxy, const float* scores, const int* counts, int B, int cap,
                 float merge_thr, float nms_thr, int merge_method, int max_iter, int max_out, float* merged,
                 uint8_t* keep, float* out_boxes, float* out_scores, int* out_src, int* out_counts, int* n_iter,
                 hipStream_t s) {
    OCR_REQUIRE(xyxy && counts && out_boxes && out_counts, "lanms: xyxy, counts, out_boxes and out_counts are required");
    OCR_REQUIRE(B >= 0 && cap > 0 && cap <= (1 << 22), "lanms: bad sizes B=%d cap=%d", B, cap);
    // With a threshold <= 0 the reference's masked-matrix expressions degenerate (every entry of
    // `iou * mask >= thr` is true, diagonal and lower triangle included); not supported.
    OCR_REQUIRE(merge_thr > 0.f && nms_thr > 0.f, "lanms: thresholds must be > 0 (merge %g, nms %g)", merge_thr, nms_thr);
    OCR_REQUIRE(merge_method == OCRHIP_MERGE_UNION || merge_method == OCRHIP_MERGE_AVERAGE || merge_method == OCRHIP_MERGE_WEIGHTED,
                "lanms: unknown merge method %d", merge_method);
    OCR_REQUIRE((reinterpret_cast<uintptr_t>(xyxy) & 15) == 0 && (reinterpret_cast<uintptr_t>(out_boxes) & 15) == 0 &&
                    (merged == nullptr || (reinterpret_cast<uintptr_t>(merged) & 15) == 0),
                "lanms: box arrays must be 16-byte aligned");
    if (B == 0) return OCRHIP_OK;
    Params p;
    p.ws_stride = ws_bytes_per_image(cap);
    OCR_TRY(h->lanms_ws.reserve(p.ws_stride * (size_t)B));
    p.ws = h->lanms_ws.as<char>();
    p.xyxy = xyxy; p.scores = scores; p.counts = counts; p.cap = cap;
    p.merge_thr = merge_thr; p.nms_thr = nms_thr; p.merge_method = merge_method;
    p.max_iter = max_iter; p.max_out = max_out;
    p.merged = merged; p.keep = keep; p.out_boxes = out_boxes; p.out_scores = out_scores; p.out_src = out_src;
    p.out_counts = out_counts; p.n_iter = n_iter;
    ProfScope prof("lanms", 0.0, 0.0, s);
    // workgroups per image for the first iteration: enough tiles to share (64 boxes per wave-tile, 16 waves)
    // and no more than the chip holds at once
    int G = std::min(16, std::max(1, 256 / B));
    if (cap < 4096) G = 1;
    if (G <= 1) {
        p.phase = 0;
        hipLaunchKernelGGL(lanms_kernel, dim3(B), dim3(kThreads), 0, s, p);
    } else {
        p.phase = 1;
        hipLaunchKernelGGL(lanms_kernel, dim3(B), dim3(kThreads), 0, s, p);
        // the two passes over all candidates: with the union merge (order-independent accumulators) a workgroup of 4
        // waves per target tile, as many workgroups as the images can have tiles (up to 512 per image; the hardware
        // hands them to the CUs as others finish, which evens out the very different tile times); else 16 waves with a
        // tile each
        const bool shared_tiles = merge_method == OCRHIP_MERGE_UNION;
        const dim3 pgrid(B, shared_tiles ? std::min(512, (cap + 63) / 64) : G), pblock(shared_tiles ? 256 : kThreads);
        p.phase = 2;
        hipLaunchKernelGGL(lanms_kernel, pgrid, pblock, 0, s, p);
        p.phase = 3;
        hipLaunchKernelGGL(lanms_kernel, pgrid, pblock, 0, s, p);
        p.phase = 4;
        hipLaunchKernelGGL(lanms_kernel, dim3(B), dim3(kThreads), 0, s, p);
    }
    OCR_HIP_CHECK(hipGetLastError());
    return OCRHIP_OK;
}

}  // namespace ocrhip
