// One bidirectional LSTM layer of the recogniser as ONE launch: persistent workgroups keep their slice of W_hh in LDS
// for all time steps and exchange h_t through L2.
//
// Same arithmetic as lstm_step_kernel (rnn.hip), i.e. the Bidirectional(LSTM(256, return_sequences)) layers of
// architectures/crnn_arch.py:306-328 with Keras' cell equations and gate order i, f, c, o (rows i, f, g, o of the
// torch checkpoint, models/weights_converter.py:305-314):
//     z = xw_t + h_{t-1} W_hh^T ;  c = sig(z_f) c + sig(z_i) tanh(z_c) ;  h = sig(z_o) tanh(c)
// with xw = x W_ih^T + (b_ih + b_hh) precomputed for all steps by the flat GEMM.
//
// Decomposition: workgroup = (direction, block of 32 hidden units, sequence group g of G).  It owns the 128 rows of W_hh
// of its units (4 gates x 32 units x 256 k = 128 KB of LDS, 16-byte slots XOR-swizzled so that operand reads are
// conflict free without padding) and every G-th sequence, longest first, so the sequences still running at step t are
// a prefix of its rows and all groups shrink alike.  Per step it stages h_{t-1} of its sequences through LDS in k
// chunks of 32, multiplies on v_mfma_f32_16x16x4_f32 (a wave = 8 units x 4 gates for all rows: up to 16 independent
// accumulation chains), and finishes the cell in registers: the four gates of an element sit in two lanes (l, l ^ 8)
// of one wave, exchanged by a lane permute.  The 8 workgroups that share (direction, g) hand h_t to each other
// through global memory: h is stored and loaded with agent-scope atomic accesses and one atomic counter per
// (direction, g) says when a step is complete.
// Nothing else synchronises; a group whose sequences are all finished leaves.  The grid never exceeds what the device
// holds at once (occupancy x CUs, checked by the launcher: the hand-over needs every workgroup resident); the spin
// on the counter is bounded so that a lost partner can never hang the device: the wait gives up after ~1 s and the
// workgroup runs on with whatever it finds.
//
// Bound: latency of the per-step hand-over for short batches, f32 MFMA rate for large ones
// (2 * 2 * n_active * 1024 * 256 flop per step).
#include "common.h"

namespace ocrhip {

namespace {

typedef float floatx4 __attribute__((ext_vector_type(4)));

constexpr int kH = 256;            // hidden units
constexpr int kRows = 112;         // sequences per pass of a workgroup (7 tiles of 16: what fits beside W_hh in 160 KB)
constexpr int kKC = 64;            // k values per staged chunk of h
constexpr int kAS = kKC + 4;       // floats per staged row of h
constexpr int kWFloats = 128 * kH; // W_hh slice
constexpr int kLdsBytes = (kWFloats + kRows * kAS) * 4;

struct LayerArgs {
    const float* xw[2];      // per direction (R, 1024), bias included
    const float* whh[2];     // per direction (1024, 256)
    float* hbuf;             // (2 parities, 2 directions, n, 256)
    float* cbuf;             // (2 directions, n, 256)
    float* out;              // (R, 512)
    const int* seq_T;        // time steps of each sequence, non-decreasing; nullptr: all T
    const int* seq_row0;     // first row of each sequence in the flat matrices; nullptr: s * T
    const int* step_active;  // sequences with more than t steps, per step t; nullptr: n
    unsigned* counters;      // (2 directions, G), zero at launch
    int n, T, G;
};

// logistic and tanh on the hardware exp2 / reciprocal (1 ulp each): the cell update of 14 elements per lane sits on
// the critical path of every step
__device__ __forceinline__ float sigmoidf_(float x) { return __frcp_rn(1.0f + __expf(-x)); }
__device__ __forceinline__ float tanhf_(float x) { return 2.0f * __frcp_rn(1.0f + __expf(-2.0f * x)) - 1.0f; }

// h is the only data that moves between workgroups while the kernel runs.  It is written and read with relaxed
// agent-scope atomic accesses: they go to the point where all CUs (of all XCDs) agree, so the hand-over needs no
// cache write-back or invalidation (an agent-scope release / acquire fence pair per step costs more than the step).
__device__ __forceinline__ float4 load_h4(const float* p) {
    const unsigned long long* q = reinterpret_cast<const unsigned long long*>(p);
    const unsigned long long lo = __hip_atomic_load(q, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const unsigned long long hi = __hip_atomic_load(q + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    return make_float4(__uint_as_float((unsigned)lo), __uint_as_float((unsigned)(lo >> 32)), __uint_as_float((unsigned)hi),
                       __uint_as_float((unsigned)(hi >> 32)));
}
__device__ __forceinline__ void store_h(float* p, float v) {
    __hip_atomic_store(reinterpret_cast<unsigned*>(p), __float_as_uint(v), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

__global__ __launch_bounds__(256, 1) void lstm_layer_kernel(const LayerArgs a) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    float* Ws = lds;                 // [row R = wave*32 + gate*8 + unit8][64 slots of 16 bytes, slot c stored at c ^ (R & 15)]
    float* As = lds + kWFloats;      // [row][kAS]

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l16 = lane & 15, lg = lane >> 4;
    const int G = a.G, n = a.n;

    // workgroups sharing (direction, g) sit on one XCD when the group count allows it (dispatch is round-robin over 8 XCDs)
    int sg, ub;
    if ((2 * G) % 8 == 0) {
        const int q = blockIdx.x & 7, r = blockIdx.x >> 3;
        sg = q * (2 * G / 8) + (r >> 3);
        ub = r & 7;
    } else {
        sg = blockIdx.x >> 3;
        ub = blockIdx.x & 7;
    }
    const int dir = sg / G, g = sg % G;
    const int u0 = ub * 32;
    unsigned* counter = a.counters + sg;

    // ---- W_hh slice -> LDS, once
    {
        const float* W = a.whh[dir];
        for (int f = tid; f < 128 * 64; f += 256) {
            const int R = f >> 6, c = f & 63;
            const int w = R >> 5, gate = (R >> 3) & 3, uu = R & 7;
            const float4 v = *reinterpret_cast<const float4*>(W + ((int64_t)gate * kH + u0 + 8 * w + uu) * kH + c * 4);
            *reinterpret_cast<float4*>(&Ws[(R * 64 + (c ^ (R & 15))) * 4]) = v;
        }
    }
    __syncthreads();

    // local rows: row j of pass p is the sequence of descending-length rank d = (p * kRows + j) * G + g
    const int n_local = (n > g) ? (n - g + G - 1) / G : 0;
    const int n_pass = (n_local + kRows - 1) / kRows;

    const int hi = (lane >> 3) & 1;
    const int unit = u0 + 8 * wave + (l16 & 7);
    const int srow = tid >> 4, scol = (tid & 15) * 4;         // staging map: 16 lanes per row, 16 rows per sweep

    // Everything a cell update needs besides h_{t-1} W_hh^T is fetched before the wait for h_{t-1}: the input
    // projections xw of the lane's elements, their cell state and their row numbers.
    float xwv[7][2][4], cpv[7][2];
    int rowv[7][2], sv[7][2];
    auto prefetch_cells = [&](int t, int p, int rows_here) {
        const int tiles_here = (rows_here + 15) >> 4;
#pragma unroll
        for (int rt = 0; rt < 7; ++rt) {
            sv[rt][0] = -1;
            sv[rt][1] = -1;
            if (rt >= tiles_here) continue;
#pragma unroll
            for (int k = 0; k < 2; ++k) {
                const int j = rt * 16 + 4 * lg + 2 * hi + k;
                if (j >= rows_here) continue;
                const int s = n - 1 - ((p * kRows + j) * G + g);
                const int Ts = a.seq_T ? a.seq_T[s] : a.T;
                const int row = (a.seq_row0 ? a.seq_row0[s] : s * a.T) + (dir ? (Ts - 1 - t) : t);
                const float* xw = a.xw[dir] + (int64_t)row * (4 * kH) + unit;
#pragma unroll
                for (int gq = 0; gq < 4; ++gq) xwv[rt][k][gq] = xw[gq * kH];
                cpv[rt][k] = (t > 0) ? a.cbuf[((int64_t)dir * n + s) * kH + unit] : 0.0f;
                rowv[rt][k] = row;
                sv[rt][k] = s;
            }
        }
    };

    // cell state and layer output of the elements just finished (kept in cpv / xwv[.][.][0] by the cell update)
    auto flush_cells = [&]() {
#pragma unroll
        for (int rt = 0; rt < 7; ++rt)
#pragma unroll
            for (int k = 0; k < 2; ++k) {
                const int s = sv[rt][k];
                if (s < 0) continue;
                a.cbuf[((int64_t)dir * n + s) * kH + unit] = cpv[rt][k];
                a.out[(int64_t)rowv[rt][k] * (2 * kH) + dir * kH + unit] = xwv[rt][k][0];
            }
    };

    for (int t = 0; t < a.T; ++t) {
        const int n_act = a.step_active ? a.step_active[t] : n;              // sequences still running, over all groups
        const int act_local = (n_act > g) ? (n_act - g + G - 1) / G : 0;      // prefix of this group's rows
        if (act_local == 0) break;
        prefetch_cells(t, 0, min(kRows, act_local));
        if (t > 0) {
            // wait until the 8 workgroups of (direction, g) have published h_{t-1}
            if (tid == 0) {
                const unsigned want = 8u * (unsigned)t;
                int spins = 0;
                while (__hip_atomic_load(counter, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < want && spins < (1 << 22)) {
                    __builtin_amdgcn_s_sleep(1);
                    ++spins;
                }
            }
            __syncthreads();
        }
        const float* hprev = a.hbuf + ((int64_t)((t + 1) & 1) * 2 + dir) * n * kH;
        float* hnext = a.hbuf + ((int64_t)(t & 1) * 2 + dir) * n * kH;

        for (int p = 0; p < n_pass; ++p) {
            const int rows_here = min(kRows, act_local - p * kRows);         // active rows of this pass
            if (rows_here <= 0) break;
            const int tiles_here = (rows_here + 15) >> 4;
            if (p > 0) {
                flush_cells();
                prefetch_cells(t, p, rows_here);
            }

            floatx4 acc[7][2];
#pragma unroll
            for (int rt = 0; rt < 7; ++rt)
#pragma unroll
                for (int ct = 0; ct < 2; ++ct)
#pragma unroll
                    for (int v = 0; v < 4; ++v) acc[rt][ct][v] = 0.0f;

            if (t > 0) {
                int64_t hoff[7];
#pragma unroll
                for (int i = 0; i < 7; ++i) {
                    const int j = srow + i * 16;
                    const int d = (p * kRows + j) * G + g;
                    hoff[i] = (j < rows_here) ? (int64_t)(n - 1 - d) * kH + scol : -1;
                }
                // the k chunks of h_{t-1} travel global -> registers -> LDS, two chunks ahead of their use
                float4 ra[2][7];
                auto load_chunk = [&](int kc, int slot) {
#pragma unroll
                    for (int i = 0; i < 7; ++i)
                        ra[slot][i] = (hoff[i] >= 0) ? load_h4(hprev + hoff[i] + kc * kKC) : make_float4(0.f, 0.f, 0.f, 0.f);
                };
                load_chunk(0, 0);
                load_chunk(1, 1);
#pragma unroll
                for (int kc = 0; kc < kH / kKC; ++kc) {
                    __syncthreads();                                         // previous chunk fully read
#pragma unroll
                    for (int i = 0; i < 7; ++i) *reinterpret_cast<float4*>(&As[(srow + i * 16) * kAS + scol]) = ra[kc & 1][i];
                    __syncthreads();
                    if (kc + 2 < kH / kKC) load_chunk(kc + 2, kc & 1);
#pragma unroll
                    for (int cbl = 0; cbl < kKC / 16; ++cbl) {               // 16 k values per block
                        const int c = (kc * (kKC / 16) + cbl) * 4 + lg;      // 16-byte slot of W_hh rows
                        float4 b4[2];
#pragma unroll
                        for (int ct = 0; ct < 2; ++ct) {
                            const int R = wave * 32 + ct * 16 + l16;
                            b4[ct] = *reinterpret_cast<const float4*>(&Ws[(R * 64 + (c ^ (R & 15))) * 4]);
                        }
#pragma unroll
                        for (int rt = 0; rt < 7; ++rt) {
                            if (rt < tiles_here) {
                                const float4 a4 = *reinterpret_cast<const float4*>(&As[(rt * 16 + l16) * kAS + cbl * 16 + lg * 4]);
#pragma unroll
                                for (int ct = 0; ct < 2; ++ct) {
                                    acc[rt][ct] = __builtin_amdgcn_mfma_f32_16x16x4f32(a4.x, b4[ct].x, acc[rt][ct], 0, 0, 0);
                                    acc[rt][ct] = __builtin_amdgcn_mfma_f32_16x16x4f32(a4.y, b4[ct].y, acc[rt][ct], 0, 0, 0);
                                    acc[rt][ct] = __builtin_amdgcn_mfma_f32_16x16x4f32(a4.z, b4[ct].z, acc[rt][ct], 0, 0, 0);
                                    acc[rt][ct] = __builtin_amdgcn_mfma_f32_16x16x4f32(a4.w, b4[ct].w, acc[rt][ct], 0, 0, 0);
                                }
                            }
                        }
                    }
                }
            }

            // ---- cell update.  Register v of tile (rt, ct) of a lane: row rt*16 + 4*lg + v, column l16 = (gate & 1) * 8 +
            // unit (ct 0: gates i, f; ct 1: gates g, o).  Lanes l and l ^ 8 hold the two halves of an element's gates: the
            // lane with bit 3 clear finishes rows v = 0, 1, its partner rows v = 2, 3.
#pragma unroll
            for (int rt = 0; rt < 7; ++rt) {
                if (rt >= tiles_here) continue;
                float mine[2][2], theirs[2][2];          // [ct][row of my pair]
#pragma unroll
                for (int ct = 0; ct < 2; ++ct)
#pragma unroll
                    for (int k = 0; k < 2; ++k) {
                        const float keep = hi ? acc[rt][ct][2 + k] : acc[rt][ct][k];
                        const float send = hi ? acc[rt][ct][k] : acc[rt][ct][2 + k];
                        mine[ct][k] = keep;
                        theirs[ct][k] = __shfl_xor(send, 8, 64);
                    }
#pragma unroll
                for (int k = 0; k < 2; ++k) {
                    const int s = sv[rt][k];
                    if (s < 0) continue;
                    // my column holds gate (i or g) when hi == 0 and (f or o) when hi == 1
                    const float zi = xwv[rt][k][0] + (hi ? theirs[0][k] : mine[0][k]);
                    const float zf = xwv[rt][k][1] + (hi ? mine[0][k] : theirs[0][k]);
                    const float zg = xwv[rt][k][2] + (hi ? theirs[1][k] : mine[1][k]);
                    const float zo = xwv[rt][k][3] + (hi ? mine[1][k] : theirs[1][k]);
                    const float c = sigmoidf_(zf) * cpv[rt][k] + sigmoidf_(zi) * tanhf_(zg);
                    const float hval = sigmoidf_(zo) * tanhf_(c);
                    store_h(hnext + (int64_t)s * kH + unit, hval);
                    cpv[rt][k] = c;                  // written out, with the layer output, after h_t is published
                    xwv[rt][k][0] = hval;
                }
            }
        }
        // publish h_t, then write what only this workgroup reads again
        __builtin_amdgcn_s_waitcnt(0x0F70);                 // vmcnt(0): this thread's h stores have reached the coherence point
        __syncthreads();
        if (tid == 0) __hip_atomic_fetch_add(counter, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        flush_cells();
    }
}

}  // namespace

// One bidirectional LSTM layer over n sequences in one launch of persistent workgroups.  `counters` must hold 2 * 16 unsigned
// ints (they are cleared here).  Returns OCRHIP_ESTATE when the device cannot hold the grid at once; the caller then
// uses the per-step kernel.
int launch_lstm_layer(const float* xw_f, const float* xw_b, const float* whh_f, const float* whh_b, float* hbuf, float* cbuf,
                      float* out, int n, int T, const int* seq_T, const int* seq_row0, const int* step_active,
                      unsigned* counters, hipStream_t s) {
    OCR_REQUIRE(xw_f && xw_b && whh_f && whh_b && hbuf && cbuf && out && counters, "lstm_layer: null argument");
    OCR_REQUIRE(n > 0 && T > 0, "lstm_layer: bad sizes n=%d T=%d", n, T);
    OCR_REQUIRE((seq_T == nullptr) == (seq_row0 == nullptr) && (seq_T == nullptr) == (step_active == nullptr),
                "lstm_layer: seq_T, seq_row0 and step_active go together");
    static int max_groups[64] = {};            // per device: sequence groups such that 16 * G workgroups are resident at once
    int dev = 0;
    OCR_HIP_CHECK(hipGetDevice(&dev));
    int& mg = max_groups[dev & 63];
    if (mg == 0) {
        OCR_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(&lstm_layer_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, kLdsBytes));
        hipDeviceProp_t prop;
        OCR_HIP_CHECK(hipGetDeviceProperties(&prop, dev));
        int per_cu = 0;
        OCR_HIP_CHECK(hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, reinterpret_cast<const void*>(&lstm_layer_kernel), 256, kLdsBytes));
        const int resident = per_cu * prop.multiProcessorCount;
        mg = (resident >= 16) ? std::min(16, resident / 16) : -1;
    }
    if (mg < 0) {
        set_error("lstm_layer: the device cannot hold the persistent grid at once");
        return OCRHIP_ESTATE;
    }
    LayerArgs a;
    a.xw[0] = xw_f; a.xw[1] = xw_b; a.whh[0] = whh_f; a.whh[1] = whh_b;
    a.hbuf = hbuf; a.cbuf = cbuf; a.out = out; a.seq_T = seq_T; a.seq_row0 = seq_row0; a.step_active = step_active;
    a.counters = counters; a.n = n; a.T = T;
    a.G = std::max(1, std::min(mg, (n + 15) / 16));          // no more groups than 16-row tiles of work
    OCR_HIP_CHECK(hipMemsetAsync(counters, 0, 2 * 16 * sizeof(unsigned), s));
    // a plain launch: the grid is sized to what the device holds at once (checked above), which is all the hand-over
    // needs; the cooperative-launch entry point, which performs the same check, crashes the profiler at process exit
    hipLaunchKernelGGL(lstm_layer_kernel, dim3(16 * a.G), dim3(256), kLdsBytes, s, a);
    OCR_HIP_CHECK(hipGetLastError());
    return OCRHIP_OK;
}

}  // namespace ocrhip
