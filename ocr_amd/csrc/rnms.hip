// Greedy non-maximum suppression on quadrilaterals with rotated IoU.  Opt-in extension: the reference collapses
// quadrilaterals to axis-aligned boxes before its NMS (utils/image/bounding_box/non_max_suppression.py:27,
// converter.py:151-154); this kernel keeps the rotation of the detector's RBOX output
// (models/detection/east.py:102-151) through suppression.
//
// Per image, one workgroup repeats: take the remaining candidate with the highest score (lowest index among equal
// scores), write it out, remove every remaining candidate whose IoU with it is >= the threshold.  Removal by the
// previous winner and the search for the next one are one pass over the candidates (element i belongs to thread
// i mod 1024 for the whole kernel, so the pass needs no synchronisation of its own); a wavefront reduces its
// best (score, index) with lane shuffles, the 16 wavefronts meet in LDS.  The number of passes is the number of
// boxes kept (hundreds), each pass touches 20 bytes per candidate (alive score + axis-aligned bounds, L2 resident);
// the rotated IoU - Sutherland-Hodgman clipping of one quadrilateral by the other's 4 edges, float64, at most 8
// vertices - runs only for pairs whose bounds overlap.  Latency bound (dependent passes), like the LANMS kernel.
#include "common.h"

namespace ocrhip {

namespace {

constexpr int kThreads = 1024;
constexpr int kWaves = kThreads / 64;
constexpr int kNone = 0x7fffffff;

struct Pt {
    double x, y;
};

// twice the signed area, summed in vertex order
__device__ __forceinline__ double shoelace2(const Pt* p, int n) {
    double s = 0.0;
    for (int i = 0; i < n; ++i) {
        const Pt a = p[i], b = p[(i + 1 == n) ? 0 : i + 1];
        s = s + (a.x * b.y - b.x * a.y);
    }
    return s;
}

__device__ __forceinline__ void load_oriented(const float* __restrict__ q, Pt* out) {
    Pt t[4];
    for (int i = 0; i < 4; ++i) {
        t[i].x = (double)q[2 * i];
        t[i].y = (double)q[2 * i + 1];
    }
    const bool flip = shoelace2(t, 4) < 0.0;
    for (int i = 0; i < 4; ++i) out[i] = t[flip ? 3 - i : i];
}

// the part of `in` (n vertices) on the left of, or on, the line a -> b
__device__ __forceinline__ int clip(const Pt* in, int n, Pt a, Pt b, Pt* out) {
    const double ux = b.x - a.x, uy = b.y - a.y;
    int m = 0;
    for (int i = 0; i < n; ++i) {
        const Pt s = in[i], e = in[(i + 1 == n) ? 0 : i + 1];
        const double ds = ux * (s.y - a.y) - uy * (s.x - a.x);
        const double de = ux * (e.y - a.y) - uy * (e.x - a.x);
        if (de >= 0.0) {
            if (ds < 0.0) {
                const double t = ds / (ds - de);
                out[m].x = s.x + t * (e.x - s.x);
                out[m].y = s.y + t * (e.y - s.y);
                ++m;
            }
            out[m++] = e;
        } else if (ds >= 0.0) {
            const double t = ds / (ds - de);
            out[m].x = s.x + t * (e.x - s.x);
            out[m].y = s.y + t * (e.y - s.y);
            ++m;
        }
    }
    return m;
}

__device__ double quad_iou(const float* __restrict__ p, const Pt* Q, double area_q) {
    Pt bufa[10], bufb[10];
    load_oriented(p, bufa);
    const double area_p = 0.5 * shoelace2(bufa, 4);
    Pt* cur = bufa;
    Pt* nxt = bufb;
    int n = 4;
    for (int i = 0; i < 4 && n > 0; ++i) {
        n = clip(cur, n, Q[i], Q[(i + 1) & 3], nxt);
        Pt* t = cur;
        cur = nxt;
        nxt = t;
    }
    double inter = (n >= 3) ? 0.5 * shoelace2(cur, n) : 0.0;
    if (inter < 0.0) inter = 0.0;
    const double uni = area_p + area_q - inter;
    return uni > 0.0 ? inter / uni : 0.0;
}

__global__ __launch_bounds__(kThreads) void rotated_nms_kernel(const float* __restrict__ polys, const float* __restrict__ scores,
                                                               const int* __restrict__ counts, int cap, double iou_threshold,
                                                               int max_out, float* __restrict__ alive_ws,
                                                               float4* __restrict__ aabb_ws, float* __restrict__ out_polys,
                                                               float* __restrict__ out_scores, int* __restrict__ out_index,
                                                               int* __restrict__ out_counts) {
    __shared__ float s_best[kWaves];
    __shared__ int s_idx[kWaves];
    __shared__ int s_winner;

    const int b = blockIdx.x;
    const int n = min(max(counts[b], 0), cap);
    const float* q = polys + (int64_t)b * cap * 8;
    const float* sc = scores + (int64_t)b * cap;
    float* alive = alive_ws + (int64_t)b * cap;
    float4* aabb = aabb_ws + (int64_t)b * cap;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;

    for (int i = tid; i < n; i += kThreads) {
        const float* p = q + (int64_t)i * 8;
        float4 bb;
        bb.x = fminf(fminf(p[0], p[2]), fminf(p[4], p[6]));
        bb.y = fminf(fminf(p[1], p[3]), fminf(p[5], p[7]));
        bb.z = fmaxf(fmaxf(p[0], p[2]), fmaxf(p[4], p[6]));
        bb.w = fmaxf(fmaxf(p[1], p[3]), fmaxf(p[5], p[7]));
        aabb[i] = bb;
        const float s = sc[i];
        alive[i] = (s == s) ? s : -INFINITY;            // a NaN score never wins
    }
    // every element is written and later read by the same thread: no barrier needed for the workspace

    int kept = 0;
    int winner = kNone;                                  // the box taken in the previous pass
    Pt W[4];
    double area_w = 0.0;
    float4 wbb = make_float4(0.f, 0.f, 0.f, 0.f);
    while (kept < max_out) {
        float best = -INFINITY;
        int best_i = kNone;
        for (int i = tid; i < n; i += kThreads) {
            float s = alive[i];
            if (s == -INFINITY) continue;
            if (winner != kNone) {
                if (i == winner) {
                    alive[i] = -INFINITY;
                    continue;
                }
                const float4 bb = aabb[i];
                if (fmaxf(bb.x, wbb.x) < fminf(bb.z, wbb.z) && fmaxf(bb.y, wbb.y) < fminf(bb.w, wbb.w)) {
                    if (quad_iou(q + (int64_t)i * 8, W, area_w) >= iou_threshold) {
                        alive[i] = -INFINITY;
                        continue;
                    }
                }
            }
            if (s > best) {                               // ascending i: the first of equal scores stays
                best = s;
                best_i = i;
            }
        }
        for (int off = 32; off > 0; off >>= 1) {
            const float ob = __shfl_xor(best, off, 64);
            const int oi = __shfl_xor(best_i, off, 64);
            if (oi != kNone && (best_i == kNone || ob > best || (ob == best && oi < best_i))) {
                best = ob;
                best_i = oi;
            }
        }
        if (lane == 0) {
            s_best[wave] = best;
            s_idx[wave] = best_i;
        }
        __syncthreads();
        if (tid == 0) {
            float bs = -INFINITY;
            int bi = kNone;
            for (int w = 0; w < kWaves; ++w) {
                const int oi = s_idx[w];
                const float ob = s_best[w];
                if (oi != kNone && (bi == kNone || ob > bs || (ob == bs && oi < bi))) {
                    bs = ob;
                    bi = oi;
                }
            }
            s_winner = bi;
            if (bi != kNone) {
                const int64_t o = (int64_t)b * max_out + kept;
                for (int k = 0; k < 8; ++k) out_polys[o * 8 + k] = q[(int64_t)bi * 8 + k];
                out_scores[o] = bs;
                out_index[o] = bi;
            }
        }
        __syncthreads();
        winner = s_winner;
        if (winner == kNone) break;
        ++kept;
        load_oriented(q + (int64_t)winner * 8, W);
        area_w = 0.5 * shoelace2(W, 4);
        wbb = aabb[winner];                               // written by another thread before the first barrier above
    }
    if (tid == 0) out_counts[b] = kept;
}

}  // namespace

int launch_rotated_nms(ocrhip_ctx* h, const float* polys, const float* scores, const int* counts, int B, int cap,
                       float iou_threshold, int max_out, float* out_polys, float* out_scores, int* out_index,
                       int* out_counts, hipStream_t s) {
    OCR_REQUIRE(polys && scores && counts && out_polys && out_scores && out_index && out_counts, "rotated_nms: null pointer");
    OCR_REQUIRE(B > 0 && cap > 0 && max_out > 0, "rotated_nms: bad sizes B=%d cap=%d max_out=%d", B, cap, max_out);
    OCR_REQUIRE(iou_threshold > 0.f, "rotated_nms: the IoU threshold must be positive (pairs with disjoint bounds are skipped)");
    const size_t per = (size_t)B * cap;
    OCR_TRY(h->lanms_ws.reserve(per * (sizeof(float) + sizeof(float4)) + 16));
    float4* aabb = h->lanms_ws.as<float4>();
    float* alive = reinterpret_cast<float*>(aabb + per);
    ProfScope prof("rotated_nms", 0.0, 36.0 * (double)per, s);
    hipLaunchKernelGGL(rotated_nms_kernel, dim3(B), dim3(kThreads), 0, s, polys, scores, counts, cap, (double)iou_threshold,
                       max_out, alive, aabb, out_polys, out_scores, out_index, out_counts);
    OCR_HIP_CHECK(hipGetLastError());
    return OCRHIP_OK;
}

}  // namespace ocrhip
