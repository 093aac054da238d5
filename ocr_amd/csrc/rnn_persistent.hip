// One bidirectional LSTM layer of the recogniser as ONE launch: persistent workgroups keep their slice of W_hh in LDS
// for all time steps and exchange h_t through L2.
//
// Same arithmetic as lstm_step_kernel (rnn.hip), i.e. the Bidirectional(LSTM(256, return_sequences)) layers of
// architectures/crnn_arch.py:306-328 with Keras' cell equations and gate order i, f, c, o (rows i, f, g, o of the
// torch checkpoint, models/weights_converter.py:305-314):
//     z = xw_t + h_{t-1} W_hh^T ;  c = sig(z_f) c + sig(z_i) tanh(z_c) ;  h = sig(z_o) tanh(c)
// with xw = x W_ih^T + (b_ih + b_hh) precomputed for all steps by the flat GEMM, its columns ordered unit-major (the
// four gates of a unit are one 16-byte load).
//
// Decomposition: workgroup = (direction, block of 32 hidden units, sequence group g of G).  It owns the 128 rows of W_hh
// of its units (4 gates x 32 units x 256 k = 128 KB of LDS, 16-byte slots XOR-swizzled so that operand reads are
// conflict free without padding) and every G-th sequence, longest first, so the sequences still running at step t are
// a prefix of its rows and all groups shrink alike.  Its rows are worked off in passes of 112.  Per step and pass it
// stages h_{t-1} of the pass's sequences through LDS in k chunks of 32 (double buffered, one barrier per chunk, the
// global loads two chunks ahead), multiplies on v_mfma_f32_16x16x4_f32 (a wave = 8 units x 4 gates for all rows: up
// to 14 independent accumulation chains; the code is specialised on the number of 16-row tiles so that the chunk
// loop is branch free and the operand reads run ahead of the MFMAs), and finishes the cell in registers: the four
// gates of an element sit in two lanes (l, l ^ 8) of one wave, exchanged by a lane permute.  The 8 workgroups that
// share (direction, g) hand h_t to each other through global memory: h is stored and loaded with agent-scope atomic
// accesses and one atomic counter per (direction, g, pass) says when a pass of a step is complete.  With two or more
// passes the hand-over costs no waiting: the partners published pass p of step t-1 a pass ago.
// Nothing else synchronises; a group whose sequences are all finished leaves.  The grid never exceeds what the device
// holds at once (occupancy x CUs, checked by the launcher: the hand-over needs every workgroup resident, and launches
// of this kernel on one device are chained through an event so that two of them never share the device); the spin on
// the counter is bounded so that a lost partner can never hang the device: a wait that gives up sets an error
// word and the workgroup LEAVES (its partners see the word within 256 spins and leave too).  The layer's output is
// then incomplete; the host reads the error word with the results and fails the call (ocrhip_check_errors).
//
// Bound: latency of the per-step hand-over for short batches, f32 MFMA rate for large ones
// (2 * 2 * n_active * 1024 * 256 flop per step).
#include <mutex>

#include "common.h"

namespace ocrhip {

namespace {

typedef float floatx4 __attribute__((ext_vector_type(4)));

constexpr int kH = 256;            // hidden units
constexpr int kRows = 112;         // sequences per pass of a workgroup (7 tiles of 16: what fits beside W_hh in 160 KB)
constexpr int kKC = 32;            // k values per staged chunk of h
constexpr int kAS = kKC + 4;       // floats per staged row of h (rows 144 bytes apart: conflict-free 16-byte reads)
constexpr int kWFloats = 128 * kH; // W_hh slice
constexpr int kLdsBytes = (kWFloats + 2 * kRows * kAS) * 4;
constexpr int kMaxPass = 64;       // counters per (direction, group): passes a workgroup may need

struct LayerArgs {
    const float* xw[2];      // per direction (R, 1024), bias included; column = unit * 4 + gate
    const float* whh[2];     // per direction (1024, 256)
    float* hbuf;             // (2 parities, 2 directions, n, 256)
    float* cbuf;             // (2 directions, n, 256)
    float* out;              // (R, 512)
    const int* seq_T;        // time steps of each sequence, non-decreasing; nullptr: all T
    const int* seq_row0;     // first row of each sequence in the flat matrices; nullptr: s * T
    const int* step_active;  // sequences with more than t steps, per step t; nullptr: n
    unsigned* counters;      // (2 directions x G, kMaxPass), zero at launch
    unsigned* error;         // set to 1 by a workgroup whose wait for its partners timed out
    int n, T, G;
    int spin_cap;            // spins of a wait before it gives up
    int debug_drop;          // test hook: unit block 7 of (direction 0, group 0) never publishes step 0
};

// The cell update of 14 elements per lane sits on the critical path of every step, on a SIMD that runs one wave: what
// counts is the number of instructions, the transcendental ones (quarter rate) above all.  Hardware exp2 and reciprocal
// (1 ulp; __frcp_rn would be a ten-instruction division, __expf wraps v_exp_f32 in range scaling that 1 + e^x does
// not need), and the products of logistic and tanh share a reciprocal:
//     sig(a) tanh(b) = (1 - e^-2b) / ((1 + e^-a)(1 + e^-2b))        5 exp2 + 3 rcp per element instead of 5 + 5.
// The exponents are capped at 2^60 so that the product of two denominators stays finite (sig(-41) is 1e-18 either way).
__device__ __forceinline__ float exp2_capped(float x) { return __builtin_amdgcn_exp2f(fminf(x, 60.0f)); }
__device__ __forceinline__ void lstm_cell(float zi, float zf, float zg, float zo, float cprev, float& c, float& h) {
    const float ei = exp2_capped(-1.44269504f * zi), ef = exp2_capped(-1.44269504f * zf);
    const float eg = exp2_capped(-2.88539008f * zg), eo = exp2_capped(-1.44269504f * zo);
    const float sf = __builtin_amdgcn_rcpf(1.0f + ef);
    const float ig = (1.0f - eg) * __builtin_amdgcn_rcpf((1.0f + ei) * (1.0f + eg));
    c = sf * cprev + ig;
    const float ec = exp2_capped(-2.88539008f * c);
    h = (1.0f - ec) * __builtin_amdgcn_rcpf((1.0f + eo) * (1.0f + ec));
}

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
__device__ __forceinline__ float load_c(const float* p) {
    return __uint_as_float(__hip_atomic_load(reinterpret_cast<const unsigned*>(p), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
}
__device__ __forceinline__ void store_h(float* p, float v) {
    __hip_atomic_store(reinterpret_cast<unsigned*>(p), __float_as_uint(v), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// acc += h_{t-1}[rows of the pass] (NT tiles of 16 rows) x W_hh^T[the wave's 32 gate rows], K = 256 in chunks of 32.
// hoff[i]: offset of the float4 this thread stages for row srow + 32 i of the pass (-1: beyond the active rows).
// `between` runs once the loads of the first two chunks are on their way (the requests for the cell update's operands:
// their latency hides behind the GEMM, that of the first chunks behind their issue).
template <int NT, class Between>
__device__ __forceinline__ void gemm_pass(const float* __restrict__ hprev, const int64_t (&hoff)[4], const float* Ws, float* As,
                                          int tid, int wave, int l16, int lg, floatx4 (&acc)[7][2], Between&& between) {
    constexpr int kSweeps = (NT * 16 + 31) / 32;             // staging sweeps of 32 rows (8 lanes per row)
    const int srow = tid >> 3, scol = (tid & 7) * 4;
    float4 ra[2][kSweeps];
    auto load_chunk = [&](int kc, int slot) {
#pragma unroll
        for (int i = 0; i < kSweeps; ++i)
            ra[slot][i] = (hoff[i] >= 0) ? load_h4(hprev + hoff[i] + kc * kKC) : make_float4(0.f, 0.f, 0.f, 0.f);
    };
    load_chunk(0, 0);
    load_chunk(1, 1);
    between();
#pragma unroll
    for (int kc = 0; kc < kH / kKC; ++kc) {
        float* buf = As + (kc & 1) * (kRows * kAS);
        // the chunk's W_hh operands (static data: no need to wait for the barrier)
        float4 b4[kKC / 16][2];
#pragma unroll
        for (int cbl = 0; cbl < kKC / 16; ++cbl) {               // 16 k values per block
            const int c = (kc * (kKC / 16) + cbl) * 4 + lg;      // 16-byte slot of W_hh rows
#pragma unroll
            for (int ct = 0; ct < 2; ++ct) {
                const int R = wave * 32 + ct * 16 + l16;
                b4[cbl][ct] = *reinterpret_cast<const float4*>(&Ws[(R * 64 + (c ^ (R & 15))) * 4]);
            }
        }
        // buffer kc & 1 was last read for chunk kc - 2, i.e. before every wave's arrival at the barrier of chunk kc - 1
#pragma unroll
        for (int i = 0; i < kSweeps; ++i)
            if (srow + i * 32 < kRows) *reinterpret_cast<float4*>(&buf[(srow + i * 32) * kAS + scol]) = ra[kc & 1][i];
        __syncthreads();
        if (kc + 2 < kH / kKC) load_chunk(kc + 2, kc & 1);
        // (k block, row tile) items in sequence, the h operand of the next item requested before the MFMAs of this one
        constexpr int kItems = (kKC / 16) * NT;
        const float* arow = buf + l16 * kAS + lg * 4;
        float4 a4[2];
        a4[0] = *reinterpret_cast<const float4*>(arow);
#pragma unroll
        for (int it = 0; it < kItems; ++it) {
            const int cbl = it / NT, rt = it % NT;
            if (it + 1 < kItems) {
                const int cn = (it + 1) / NT, rn = (it + 1) % NT;
                a4[(it + 1) & 1] = *reinterpret_cast<const float4*>(arow + rn * 16 * kAS + cn * 16);
            }
            __builtin_amdgcn_sched_barrier(0);                   // keep the read ahead of the MFMAs (the scheduler sinks it)
            const float4 av = a4[it & 1];
#pragma unroll
            for (int ct = 0; ct < 2; ++ct) acc[rt][ct] = __builtin_amdgcn_mfma_f32_16x16x4f32(av.x, b4[cbl][ct].x, acc[rt][ct], 0, 0, 0);
#pragma unroll
            for (int ct = 0; ct < 2; ++ct) acc[rt][ct] = __builtin_amdgcn_mfma_f32_16x16x4f32(av.y, b4[cbl][ct].y, acc[rt][ct], 0, 0, 0);
#pragma unroll
            for (int ct = 0; ct < 2; ++ct) acc[rt][ct] = __builtin_amdgcn_mfma_f32_16x16x4f32(av.z, b4[cbl][ct].z, acc[rt][ct], 0, 0, 0);
#pragma unroll
            for (int ct = 0; ct < 2; ++ct) acc[rt][ct] = __builtin_amdgcn_mfma_f32_16x16x4f32(av.w, b4[cbl][ct].w, acc[rt][ct], 0, 0, 0);
        }
    }
}

template <bool RAGGED>
__global__ __launch_bounds__(256, 1) void lstm_layer_kernel(const LayerArgs a) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    __shared__ int bail;             // a wait of this workgroup timed out (or a partner's did): leave
    float* Ws = lds;                 // [row R = wave*32 + gate*8 + unit8][64 slots of 16 bytes, slot c stored at c ^ (R & 15)]
    float* As = lds + kWFloats;      // [2 buffers][row][kAS]

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
    unsigned* counters = a.counters + sg * kMaxPass;

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
    if (tid == 0) bail = 0;
    __syncthreads();

    // local rows: row j of pass p is the sequence of descending-length rank d = (p * kRows + j) * G + g
    const int n_local = (n > g) ? (n - g + G - 1) / G : 0;
    const int n_pass = (n_local + kRows - 1) / kRows;

    const int hi = (lane >> 3) & 1;
    const int unit = u0 + 8 * wave + (l16 & 7);

    // Everything a cell update needs besides h_{t-1} W_hh^T (the input projections xw of the lane's elements, their
    // cell state and their row numbers) is requested right after the first loads of h_{t-1}, long before its use.
    float xwv[7][2][4], cpv[7][2];
    int rowv[7][2], sv[7][2];
    // Branch free (elements beyond the active rows load sequence 0 and are ignored later): with a branch per element the
    // compiler waits for each element's loads before it issues the next element's.  The sequence data (length, first
    // row) that the addresses depend on are requested a pass ahead, so a pass pays one memory round trip, not two.
    int Tsv[7][2], r0v[7][2];
    auto prefetch_seq = [&](int p) {
#pragma unroll
        for (int rt = 0; rt < 7; ++rt)
#pragma unroll
            for (int k = 0; k < 2; ++k) {
                const int j = rt * 16 + 4 * lg + 2 * hi + k;
                const int s = max(n - 1 - ((p * kRows + j) * G + g), 0);
                if (RAGGED) {
                    Tsv[rt][k] = a.seq_T[s];
                    r0v[rt][k] = a.seq_row0[s];
                } else {
                    Tsv[rt][k] = a.T;
                    r0v[rt][k] = s * a.T;
                }
            }
    };
    auto prefetch_cells = [&](int t, int p, int rows_here) {
#pragma unroll
        for (int rt = 0; rt < 7; ++rt)
#pragma unroll
            for (int k = 0; k < 2; ++k) {
                const int j = rt * 16 + 4 * lg + 2 * hi + k;
                const bool on = j < rows_here;
                const int s = max(n - 1 - ((p * kRows + j) * G + g), 0);
                sv[rt][k] = on ? s : -1;
                const int row = r0v[rt][k] + (dir ? (Tsv[rt][k] - 1 - t) : t);
                const float4 xw4 = *reinterpret_cast<const float4*>(a.xw[dir] + (int64_t)row * (4 * kH) + unit * 4);
                xwv[rt][k][0] = xw4.x; xwv[rt][k][1] = xw4.y; xwv[rt][k][2] = xw4.z; xwv[rt][k][3] = xw4.w;
                const float cprev = a.cbuf[((int64_t)dir * n + s) * kH + unit];
                cpv[rt][k] = (t > 0) ? cprev : 0.0f;
                rowv[rt][k] = row;
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

    prefetch_seq(0);
    for (int t = 0; t < a.T; ++t) {
        const int n_act = RAGGED ? a.step_active[t] : n;              // sequences still running, over all groups
        const int act_local = (n_act > g) ? (n_act - g + G - 1) / G : 0;      // prefix of this group's rows
        if (act_local == 0) break;
        const float* hprev = a.hbuf + ((int64_t)((t + 1) & 1) * 2 + dir) * n * kH;
        float* hnext = a.hbuf + ((int64_t)(t & 1) * 2 + dir) * n * kH;

        for (int p = 0; p < n_pass; ++p) {
            const int rows_here = min(kRows, act_local - p * kRows);         // active rows of this pass
            if (rows_here <= 0) break;
            const int tiles_here = (rows_here + 15) >> 4;
            auto prefetch = [&]() {
                prefetch_cells(t, p, rows_here);
                prefetch_seq((p + 1 < n_pass && act_local > (p + 1) * kRows) ? p + 1 : 0);      // the pass that follows this one
            };

            floatx4 acc[7][2];
#pragma unroll
            for (int rt = 0; rt < 7; ++rt)
#pragma unroll
                for (int ct = 0; ct < 2; ++ct)
#pragma unroll
                    for (int v = 0; v < 4; ++v) acc[rt][ct][v] = 0.0f;

            if (t > 0) {
                // wait until the 8 workgroups of (direction, g) have published pass p of h_{t-1}
                if (tid == 0) {
                    const unsigned want = 8u * (unsigned)t;
                    int spins = 0;
                    while (__hip_atomic_load(counters + p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < want) {
                        __builtin_amdgcn_s_sleep(1);
                        ++spins;
                        const bool partner_gave_up = (spins & 255) == 0 && __hip_atomic_load(a.error, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0;
                        if (spins >= a.spin_cap || partner_gave_up) {
                            __hip_atomic_fetch_or(a.error, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                            bail = 1;
                            break;
                        }
                    }
                }
                __syncthreads();
                if (bail) return;                 // uniform: every thread reads the same LDS word after the barrier
                int64_t hoff[4];
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const int j = (tid >> 3) + i * 32;
                    const int d = (p * kRows + j) * G + g;
                    hoff[i] = (j < rows_here) ? (int64_t)(n - 1 - d) * kH + (tid & 7) * 4 : -1;
                }
                switch (tiles_here) {
                    case 1: gemm_pass<1>(hprev, hoff, Ws, As, tid, wave, l16, lg, acc, prefetch); break;
                    case 2: gemm_pass<2>(hprev, hoff, Ws, As, tid, wave, l16, lg, acc, prefetch); break;
                    case 3: gemm_pass<3>(hprev, hoff, Ws, As, tid, wave, l16, lg, acc, prefetch); break;
                    case 4: gemm_pass<4>(hprev, hoff, Ws, As, tid, wave, l16, lg, acc, prefetch); break;
                    case 5: gemm_pass<5>(hprev, hoff, Ws, As, tid, wave, l16, lg, acc, prefetch); break;
                    case 6: gemm_pass<6>(hprev, hoff, Ws, As, tid, wave, l16, lg, acc, prefetch); break;
                    default: gemm_pass<7>(hprev, hoff, Ws, As, tid, wave, l16, lg, acc, prefetch); break;
                }
            } else {
                prefetch();
            }
            // ---- cell update.  Register v of tile (rt, ct) of a lane: row rt*16 + 4*lg + v, column l16 = (gate & 1) * 8 +
            // unit (ct 0: gates i, f; ct 1: gates g, o).  Lanes l and l ^ 8 hold the two halves of an element's gates: the
            // lane with bit 3 clear finishes rows v = 0, 1, its partner rows v = 2, 3.
#pragma unroll
            for (int rt = 0; rt < 7; ++rt) {
                if (rt >= tiles_here) continue;
                float mine[2][2], theirs[2][2];          // [ct][row of my pair]
#pragma unroll
                for (int ct = 0; ct < 2; ++ct) {
                    float v0 = acc[rt][ct][0], v1 = acc[rt][ct][1], v2 = acc[rt][ct][2], v3 = acc[rt][ct][3];
                    // opaque to the optimiser, which otherwise turns the selects below into a dynamic register index
                    // (three selects each)
                    asm volatile("" : "+v"(v0), "+v"(v1), "+v"(v2), "+v"(v3));
                    const float lo[2] = {v0, v1}, up[2] = {v2, v3};
#pragma unroll
                    for (int k = 0; k < 2; ++k) {
                        mine[ct][k] = hi ? up[k] : lo[k];
                        const float send = hi ? lo[k] : up[k];
                        // lane l ^ 8 = rotation by 8 within the row of 16 lanes: a DPP move, no LDS round trip
                        theirs[ct][k] = __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(send), 0x128, 0xF, 0xF, false));
                    }
                }
#pragma unroll
                for (int k = 0; k < 2; ++k) {
                    // my column holds gate (i or g) when hi == 0 and (f or o) when hi == 1
                    const float zi = xwv[rt][k][0] + (hi ? theirs[0][k] : mine[0][k]);
                    const float zf = xwv[rt][k][1] + (hi ? mine[0][k] : theirs[0][k]);
                    const float zg = xwv[rt][k][2] + (hi ? theirs[1][k] : mine[1][k]);
                    const float zo = xwv[rt][k][3] + (hi ? mine[1][k] : theirs[1][k]);
                    float c, hval;
                    lstm_cell(zi, zf, zg, zo, cpv[rt][k], c, hval);
                    cpv[rt][k] = c;                  // written out, with the layer output, after h_t is published
                    xwv[rt][k][0] = hval;
                }
            }
            // the stores apart from the arithmetic: a branch per element in there would serialise the elements' chains of
            // dependent transcendental instructions
#pragma unroll
            for (int rt = 0; rt < 7; ++rt)
#pragma unroll
                for (int k = 0; k < 2; ++k)
                    if (sv[rt][k] >= 0) store_h(hnext + (int64_t)sv[rt][k] * kH + unit, xwv[rt][k][0]);
            // publish pass p of h_t, then write what only this workgroup reads again
            __builtin_amdgcn_s_waitcnt(0x0F70);             // vmcnt(0): this thread's h stores have reached the coherence point
            __syncthreads();
            const bool dropped = a.debug_drop && sg == 0 && ub == 7 && t == 0;
            if (tid == 0 && !dropped) __hip_atomic_fetch_add(counters + p, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            flush_cells();
        }
    }
}

// ---- The same layer with independent waves (`lstm_persistent` = 2).
// The workgroup above splits the gate columns over its waves, so all of them need the h rows of the whole pass: h goes
// through LDS and every k chunk passes a barrier, the four waves walk through a pass in lockstep, and whatever one of
// them waits for (the partners' counter, the first h loads of a pass, the acknowledgement of its h stores) all of them
// wait for, on SIMDs that hold one wave each.  Here a wave owns BLOCKS OF 16 SEQUENCES instead, with all 128 gate
// columns of the workgroup's 32 units: 8 accumulator tiles of 16x16.  Its A operand - its own 16 rows of h_{t-1} - comes
// straight from memory in the layout the matrix instruction reads (lane (row, k group) loads 16 bytes per 16 k
// values), W_hh is read from the same LDS image, and nothing is shared between waves after the image is built: no
// barrier, no staging buffer.  A workgroup has 8 waves, two per SIMD, each at its own point of its own block, so one
// wave's waits are the other's issue time.  All four gates of an element land in one lane (column tile c holds gate
// c & 3 of units 16 (c >> 2) + lane % 16), so the cell update needs no lane exchange.  The hand-over is per block: the
// 8 workgroups of a (direction, sequence group) count up one counter per block, and a wave waits for its block's
// counter only.  Same k order inside every accumulator and the same cell arithmetic as the kernel above:
// bit-identical results.
constexpr int kBlockRows = 16;
constexpr int kDeviceScope = 16;    // cache policy bits of a buffer load: sc1 = device scope, what an agent-scope atomic load carries

// One block (16 sequences) of one step for NU halves of the workgroup's 32 units, starting at half uh0: 4 NU accumulator
// tiles.  `want`: what the block's counter reaches when all partners have published h_{t-1}.  False: the wait gave up.
template <bool RAGGED, int NU>
__device__ __forceinline__ bool waves_block(const LayerArgs& a, const float* Ws, int t, int rb, int uh0, int dir, int g, int u0,
                                            int act_local, unsigned want, unsigned* counters, const float* hprev, float* hnext,
                                            int lane, bool dropped) {
    const int l16 = lane & 15, lg = lane >> 4;
    const int G = a.G, n = a.n;
    const float* wrow = Ws + l16 * 256;          // + ct * 16 * 256 + ((4 j + lg) ^ l16) * 4
    const int rows_here = min(kBlockRows, act_local - rb * kBlockRows);
    // the lane's elements: rows 4 lg + v (v = 0..3) of the block, unit l16 of each of the wave's unit halves
    int sv[4], rowv[4];
    float xwv[4][NU][4], cpv[4][NU];
#pragma unroll
    for (int v = 0; v < 4; ++v) {
        const int j = 4 * lg + v;
        const int s = max(n - 1 - ((rb * kBlockRows + j) * G + g), 0);
        sv[v] = (j < rows_here) ? s : -1;
        int Ts, r0;
        if (RAGGED) {
            Ts = a.seq_T[s];
            r0 = a.seq_row0[s];
        } else {
            Ts = a.T;
            r0 = s * a.T;
        }
        rowv[v] = r0 + (dir ? (Ts - 1 - t) : t);
    }
    floatx4 acc[4 * NU];
#pragma unroll
    for (int ct = 0; ct < 4 * NU; ++ct) acc[ct] = floatx4{0.f, 0.f, 0.f, 0.f};

    if (t > 0) {
        // wait until the partners have published this block of h_{t-1}
        int spins = 0;
        bool gave_up = false;
        while (__hip_atomic_load(counters + rb, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < want) {
            __builtin_amdgcn_s_sleep(1);
            ++spins;
            const bool partner_gave_up = (spins & 255) == 0 && __hip_atomic_load(a.error, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0;
            if (spins >= a.spin_cap || partner_gave_up) {
                gave_up = true;
                break;
            }
        }
        if (gave_up) {               // uniform in the wave: every lane read the same words
            if (lane == 0) __hip_atomic_fetch_or(a.error, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            return false;            // no barrier follows: the other waves run into the error word themselves
        }
        // the wave's 16 rows of h_{t-1}: lane (row l16, k group lg) holds k = 16 j + 4 lg .. + 3 for j = 0..15
        asm volatile("" ::: "memory");          // the loads below stay behind the wait
        const int dA = (rb * kBlockRows + l16) * G + g;
        // 16-byte loads with device scope (the cache policy of the atomic loads above: served from where all
        // XCDs agree), all 16 of a block in flight at once: at ~1.4 TB/s for the whole device these loads are
        // bound by how many of them are outstanding, not by bandwidth.  Rows beyond the active ones carry an
        // out-of-range offset and read as zeros.
        const __amdgpu_buffer_rsrc_t h_rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)hprev, 0, n * (kH * 4), 0x00020000);
        const int h_off = (l16 < rows_here) ? (max(n - 1 - dA, 0) * kH + lg * 4) * 4 : (int)0x80000000;
        float4 ha[16];
#pragma unroll
        for (int j = 0; j < 16; ++j)
            ha[j] = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(h_rsrc, h_off, 64 * j, kDeviceScope));
        // what the cell update needs besides the product, requested behind the h loads
#pragma unroll
        for (int v = 0; v < 4; ++v)
#pragma unroll
            for (int uu = 0; uu < NU; ++uu) {
                const int unit = u0 + 16 * (uh0 + uu) + l16;
                const float4 xw4 = *reinterpret_cast<const float4*>(a.xw[dir] + (int64_t)rowv[v] * (4 * kH) + unit * 4);
                xwv[v][uu][0] = xw4.x; xwv[v][uu][1] = xw4.y; xwv[v][uu][2] = xw4.z; xwv[v][uu][3] = xw4.w;
                cpv[v][uu] = load_c(a.cbuf + ((int64_t)dir * n + max(sv[v], 0)) * kH + unit);
            }
        const float* wmine = wrow + uh0 * (4 * 16 * 256);
#pragma unroll
        for (int j = 0; j < 16; ++j) {
            // W_hh operands of the step: the SIMD's other wave covers their latency
            float4 b4[4 * NU];
#pragma unroll
            for (int ct = 0; ct < 4 * NU; ++ct) b4[ct] = *reinterpret_cast<const float4*>(wmine + ct * (16 * 256) + (((4 * j + lg) ^ l16) * 4));
            const float4 av = ha[j];
#pragma unroll
            for (int ct = 0; ct < 4 * NU; ++ct) acc[ct] = __builtin_amdgcn_mfma_f32_16x16x4f32(av.x, b4[ct].x, acc[ct], 0, 0, 0);
#pragma unroll
            for (int ct = 0; ct < 4 * NU; ++ct) acc[ct] = __builtin_amdgcn_mfma_f32_16x16x4f32(av.y, b4[ct].y, acc[ct], 0, 0, 0);
#pragma unroll
            for (int ct = 0; ct < 4 * NU; ++ct) acc[ct] = __builtin_amdgcn_mfma_f32_16x16x4f32(av.z, b4[ct].z, acc[ct], 0, 0, 0);
#pragma unroll
            for (int ct = 0; ct < 4 * NU; ++ct) acc[ct] = __builtin_amdgcn_mfma_f32_16x16x4f32(av.w, b4[ct].w, acc[ct], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
        }
    } else {
#pragma unroll
        for (int v = 0; v < 4; ++v)
#pragma unroll
            for (int uu = 0; uu < NU; ++uu) {
                const int unit = u0 + 16 * (uh0 + uu) + l16;
                const float4 xw4 = *reinterpret_cast<const float4*>(a.xw[dir] + (int64_t)rowv[v] * (4 * kH) + unit * 4);
                xwv[v][uu][0] = xw4.x; xwv[v][uu][1] = xw4.y; xwv[v][uu][2] = xw4.z; xwv[v][uu][3] = xw4.w;
                cpv[v][uu] = 0.0f;
            }
    }
    // cell update: register v of column tile c = gate c & 3 of unit 16 (c >> 2) + l16, row 4 lg + v
#pragma unroll
    for (int v = 0; v < 4; ++v)
#pragma unroll
        for (int uu = 0; uu < NU; ++uu) {
            float c, hval;
            lstm_cell(xwv[v][uu][0] + acc[4 * uu + 0][v], xwv[v][uu][1] + acc[4 * uu + 1][v], xwv[v][uu][2] + acc[4 * uu + 2][v],
                      xwv[v][uu][3] + acc[4 * uu + 3][v], cpv[v][uu], c, hval);
            cpv[v][uu] = c;
            xwv[v][uu][0] = hval;
        }
#pragma unroll
    for (int v = 0; v < 4; ++v)
#pragma unroll
        for (int uu = 0; uu < NU; ++uu)
            if (sv[v] >= 0) store_h(hnext + (int64_t)sv[v] * kH + u0 + 16 * (uh0 + uu) + l16, xwv[v][uu][0]);
    // the cell state goes out before the count as well, with the scope of h: the block may be another wave's at the next
    // step (see the kernel)
#pragma unroll
    for (int v = 0; v < 4; ++v)
#pragma unroll
        for (int uu = 0; uu < NU; ++uu)
            if (sv[v] >= 0) store_h(a.cbuf + ((int64_t)dir * n + sv[v]) * kH + u0 + 16 * (uh0 + uu) + l16, cpv[v][uu]);
    // publish this part of h_t: the wave's stores have reached the coherence point, then its count
    __builtin_amdgcn_s_waitcnt(0x0F70);             // vmcnt(0)
    if (lane == 0 && !dropped) __hip_atomic_fetch_add(counters + rb, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#pragma unroll
    for (int v = 0; v < 4; ++v)
#pragma unroll
        for (int uu = 0; uu < NU; ++uu) {
            if (sv[v] < 0) continue;
            a.out[(int64_t)rowv[v] * (2 * kH) + dir * kH + u0 + 16 * (uh0 + uu) + l16] = xwv[v][uu][0];
        }
    return true;
}

template <bool RAGGED>
__global__ __launch_bounds__(512, 1) void lstm_layer_waves_kernel(const LayerArgs a) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    float* Ws = lds;                 // [row R = 16 c + j: gate c & 3 of unit 16 (c >> 2) + j][64 slots of 16 bytes, slot s stored at s ^ (R & 15)]

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int G = a.G, n = a.n;

    int sg, ub;
    if ((2 * G) % 8 == 0) {          // workgroups sharing (direction, g) on one XCD, see above
        const int q = blockIdx.x & 7, r = blockIdx.x >> 3;
        sg = q * (2 * G / 8) + (r >> 3);
        ub = r & 7;
    } else {
        sg = blockIdx.x >> 3;
        ub = blockIdx.x & 7;
    }
    const int dir = sg / G, g = sg % G;
    const int u0 = ub * 32;
    unsigned* counters = a.counters + sg * kMaxPass;

    {
        const float* W = a.whh[dir];
        for (int f = tid; f < 128 * 64; f += 512) {
            const int R = f >> 6, c = f & 63;
            const int ct = R >> 4, j = R & 15;
            const float4 v = *reinterpret_cast<const float4*>(W + ((int64_t)(ct & 3) * kH + u0 + 16 * (ct >> 2) + j) * kH + c * 4);
            *reinterpret_cast<float4*>(&Ws[(R * 64 + (c ^ (R & 15))) * 4]) = v;
        }
    }
    __syncthreads();                 // the only barrier

    // row j of block rb is the sequence of descending-length rank d = (rb * 16 + j) * G + g.
    // With 4 blocks or fewer left, a block is shared by two waves, 16 of the 32 units each: half the chain of matrix
    // instructions per step when the step's latency is all there is.  Every workgroup of (direction, g) sees the same
    // number of blocks, so all agree on the number of counts a step adds to a block's counter: `published` is what the
    // counter of any block still running has reached when step t - 1 is complete (the running blocks are a prefix that
    // only shrinks, so every block of step t has been through all steps before).
    unsigned published = 0;
    for (int t = 0; t < a.T; ++t) {
        const int n_act = RAGGED ? a.step_active[t] : n;              // sequences still running, over all groups
        const int act_local = (n_act > g) ? (n_act - g + G - 1) / G : 0;      // prefix of this group's rows
        if (act_local == 0) break;
        const int blocks_here = (act_local + kBlockRows - 1) / kBlockRows;
        const float* hprev = a.hbuf + ((int64_t)((t + 1) & 1) * 2 + dir) * n * kH;
        float* hnext = a.hbuf + ((int64_t)(t & 1) * 2 + dir) * n * kH;
        const bool dropped = a.debug_drop && sg == 0 && ub == 7 && t == 0;
        if (blocks_here > 4) {
            for (int rb = wave; rb < blocks_here; rb += 8)
                if (!waves_block<RAGGED, 2>(a, Ws, t, rb, 0, dir, g, u0, act_local, published, counters, hprev, hnext, lane, dropped)) return;
            published += 8u;
        } else {
            const int rb = wave >> 1;
            if (rb < blocks_here &&
                !waves_block<RAGGED, 1>(a, Ws, t, rb, wave & 1, dir, g, u0, act_local, published, counters, hprev, hnext, lane, dropped))
                return;
            published += 16u;
        }
    }
}

// ---- The independent-waves layer on bfloat16 operands (the bfloat16-tensor path, `compute_dtype = OCRHIP_BF16`).
// Same structure, hand-over and cell arithmetic as lstm_layer_waves_kernel; what changes is the recurrent product:
// h_{t-1} travels between workgroups as bfloat16 (half the hand-over bytes: 8 instead of 16 loads of 16 bytes per
// lane and block, 4-byte stores of unit pairs), the workgroup's W_hh slice is converted to bfloat16 while its LDS image
// is built (64 KB), and the product runs on v_mfma_f32_16x16x32_bf16: 64 matrix instructions of 16 cycles per block
// and step where the float32 kernel issues 512 of 32.  Gate pre-activations (x W_ih + b, float32), the accumulation,
// the cell state and the cell arithmetic stay float32; the layer's output rows are bfloat16 like every activation of
// this path.
typedef __bf16 bf16x8_l __attribute__((ext_vector_type(8)));
typedef unsigned uintx4_l __attribute__((ext_vector_type(4)));

struct LayerArgs16 {
    const float* xw[2];      // per direction (R, 1024) float32, bias included; column = unit * 4 + gate
    const float* whh[2];     // per direction (1024, 256) float32 as in the checkpoint
    uint16_t* hbuf;          // (2 parities, 2 directions, n, 256) bfloat16
    float* cbuf;             // (2 directions, n, 256)
    uint16_t* out;           // (R, 512) bfloat16
    const int* seq_T;
    const int* seq_row0;
    const int* step_active;
    unsigned* counters;
    unsigned* error;
    int n, T, G;
    int spin_cap;
    int debug_drop;
};

__device__ __forceinline__ unsigned bf16_pair(float lo, float hi) {
    const __bf16 a = (__bf16)lo, b = (__bf16)hi;
    return (unsigned)__builtin_bit_cast(unsigned short, a) | ((unsigned)__builtin_bit_cast(unsigned short, b) << 16);
}

// UW: hidden units of a workgroup, 32 or 64 (a 64 or 128 KB image of W_hh): 8 or 4 workgroups share a (direction, sequence
// group) and every one of them reads all of h_{t-1} of its sequences, so 64 units halve the hand-over reads.
template <bool RAGGED, int UW>
__global__ __launch_bounds__(512, 1) void lstm_layer_waves16_kernel(const LayerArgs16 a) {
    constexpr int NCT = UW / 4, NUU = UW / 16, PARTNERS = kH / UW;
    extern __shared__ __attribute__((aligned(16))) unsigned char lds16[];
    // [row R = 16 c + j: gate c & 3 of unit 16 (c >> 2) + j][32 slots of 16 bytes (8 k values), slot s stored at s ^ (R & 15)]

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l16 = lane & 15, lg = lane >> 4;
    const int G = a.G, n = a.n;

    int sg, ub;
    if ((2 * G) % 8 == 0) {
        const int q = blockIdx.x & 7, r = blockIdx.x >> 3;
        sg = q * (2 * G / 8) + r / PARTNERS;
        ub = r % PARTNERS;
    } else {
        sg = blockIdx.x / PARTNERS;
        ub = blockIdx.x % PARTNERS;
    }
    const int dir = sg / G, g = sg % G;
    const int u0 = ub * UW;
    unsigned* counters = a.counters + sg * kMaxPass;

    {
        const float* W = a.whh[dir];
        for (int f = tid; f < 4 * UW * 32; f += 512) {
            const int R = f >> 5, c = f & 31;
            const int ct = R >> 4, j = R & 15;
            const float* src = W + ((int64_t)(ct & 3) * kH + u0 + 16 * (ct >> 2) + j) * kH + c * 8;
            const float4 v0 = *reinterpret_cast<const float4*>(src), v1 = *reinterpret_cast<const float4*>(src + 4);
            *reinterpret_cast<uintx4_l*>(lds16 + (R * 32 + (c ^ (R & 15))) * 16) =
                uintx4_l{bf16_pair(v0.x, v0.y), bf16_pair(v0.z, v0.w), bf16_pair(v1.x, v1.y), bf16_pair(v1.z, v1.w)};
        }
    }
    __syncthreads();                 // the only barrier

    const unsigned char* wrow = lds16 + l16 * 512;          // + ct * 16 * 512 + ((4 j + lg) ^ l16) * 16

    for (int t = 0; t < a.T; ++t) {
        const int n_act = RAGGED ? a.step_active[t] : n;
        const int act_local = (n_act > g) ? (n_act - g + G - 1) / G : 0;
        if (act_local == 0) break;
        const int blocks_here = (act_local + kBlockRows - 1) / kBlockRows;
        const uint16_t* hprev = a.hbuf + ((int64_t)((t + 1) & 1) * 2 + dir) * n * kH;
        uint16_t* hnext = a.hbuf + ((int64_t)(t & 1) * 2 + dir) * n * kH;

        for (int rb = wave; rb < blocks_here; rb += 8) {
            const int rows_here = min(kBlockRows, act_local - rb * kBlockRows);
            int sv[4], rowv[4];
            float xwv[4][NUU][4], cpv[4][NUU];
#pragma unroll
            for (int v = 0; v < 4; ++v) {
                const int j = 4 * lg + v;
                const int s = max(n - 1 - ((rb * kBlockRows + j) * G + g), 0);
                sv[v] = (j < rows_here) ? s : -1;
                int Ts, r0;
                if (RAGGED) {
                    Ts = a.seq_T[s];
                    r0 = a.seq_row0[s];
                } else {
                    Ts = a.T;
                    r0 = s * a.T;
                }
                rowv[v] = r0 + (dir ? (Ts - 1 - t) : t);
            }
            floatx4 acc[NCT];
#pragma unroll
            for (int ct = 0; ct < NCT; ++ct) acc[ct] = floatx4{0.f, 0.f, 0.f, 0.f};

            if (t > 0) {
                const unsigned want = (unsigned)PARTNERS * (unsigned)t;
                int spins = 0;
                bool gave_up = false;
                while (__hip_atomic_load(counters + rb, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < want) {
                    __builtin_amdgcn_s_sleep(1);
                    ++spins;
                    const bool partner_gave_up = (spins & 255) == 0 && __hip_atomic_load(a.error, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0;
                    if (spins >= a.spin_cap || partner_gave_up) {
                        gave_up = true;
                        break;
                    }
                }
                if (gave_up) {
                    if (lane == 0) __hip_atomic_fetch_or(a.error, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    return;
                }
                asm volatile("" ::: "memory");
                const int dA = (rb * kBlockRows + l16) * G + g;
                // lane (row l16, k group lg) holds k = 32 j + 8 lg .. + 7 for j = 0..7: 16-byte device-scope loads
                const __amdgpu_buffer_rsrc_t h_rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)hprev, 0, n * (kH * 2), 0x00020000);
                const int h_off = (l16 < rows_here) ? (max(n - 1 - dA, 0) * kH + lg * 8) * 2 : (int)0x80000000;
                uintx4_l ha[8];
#pragma unroll
                for (int j = 0; j < 8; ++j)
                    ha[j] = __builtin_bit_cast(uintx4_l, __builtin_amdgcn_raw_buffer_load_b128(h_rsrc, h_off, 64 * j, kDeviceScope));
#pragma unroll
                for (int v = 0; v < 4; ++v)
#pragma unroll
                    for (int uu = 0; uu < NUU; ++uu) {
                        const int unit = u0 + 16 * uu + l16;
                        const float4 xw4 = *reinterpret_cast<const float4*>(a.xw[dir] + (int64_t)rowv[v] * (4 * kH) + unit * 4);
                        xwv[v][uu][0] = xw4.x; xwv[v][uu][1] = xw4.y; xwv[v][uu][2] = xw4.z; xwv[v][uu][3] = xw4.w;
                        cpv[v][uu] = a.cbuf[((int64_t)dir * n + max(sv[v], 0)) * kH + unit];
                    }
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const bf16x8_l av = __builtin_bit_cast(bf16x8_l, ha[j]);
#pragma unroll
                    for (int c0 = 0; c0 < NCT; c0 += 8) {
                        bf16x8_l b8[8];
#pragma unroll
                        for (int ct = 0; ct < 8; ++ct)
                            b8[ct] = *reinterpret_cast<const bf16x8_l*>(wrow + (c0 + ct) * (16 * 512) + (((4 * j + lg) ^ l16) * 16));
#pragma unroll
                        for (int ct = 0; ct < 8; ++ct) acc[c0 + ct] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(av, b8[ct], acc[c0 + ct], 0, 0, 0);
                    }
                }
            } else {
#pragma unroll
                for (int v = 0; v < 4; ++v)
#pragma unroll
                    for (int uu = 0; uu < NUU; ++uu) {
                        const int unit = u0 + 16 * uu + l16;
                        const float4 xw4 = *reinterpret_cast<const float4*>(a.xw[dir] + (int64_t)rowv[v] * (4 * kH) + unit * 4);
                        xwv[v][uu][0] = xw4.x; xwv[v][uu][1] = xw4.y; xwv[v][uu][2] = xw4.z; xwv[v][uu][3] = xw4.w;
                        cpv[v][uu] = 0.0f;
                    }
            }
            // cell update: register v of column tile c = gate c & 3 of unit 16 (c >> 2) + l16, row 4 lg + v
            unsigned hw[4][NUU / 2];     // the lane's packed pairs of h values per row: for unit sixteens (2p, 2p + 1), even lanes hold units 32 p + (l16, l16 + 1), odd lanes units 32 p + 16 + (l16 - 1, l16)
#pragma unroll
            for (int v = 0; v < 4; ++v) {
                float hv[NUU];
#pragma unroll
                for (int uu = 0; uu < NUU; ++uu) {
                    float c;
                    lstm_cell(xwv[v][uu][0] + acc[4 * uu + 0][v], xwv[v][uu][1] + acc[4 * uu + 1][v], xwv[v][uu][2] + acc[4 * uu + 2][v],
                              xwv[v][uu][3] + acc[4 * uu + 3][v], cpv[v][uu], c, hv[uu]);
                    cpv[v][uu] = c;
                }
#pragma unroll
                for (int pp = 0; pp < NUU / 2; ++pp) {
                    const float o0 = __shfl_xor(hv[2 * pp], 1), o1 = __shfl_xor(hv[2 * pp + 1], 1);
                    hw[v][pp] = (l16 & 1) ? bf16_pair(o1, hv[2 * pp + 1]) : bf16_pair(hv[2 * pp], o0);
                }
            }
            const int pair_unit = (l16 & 1) ? u0 + 16 + l16 - 1 : u0 + l16;          // + 32 p
#pragma unroll
            for (int v = 0; v < 4; ++v)
#pragma unroll
                for (int pp = 0; pp < NUU / 2; ++pp)
                    if (sv[v] >= 0)
                        __hip_atomic_store(reinterpret_cast<unsigned*>(hnext + (int64_t)sv[v] * kH + pair_unit + 32 * pp), hw[v][pp], __ATOMIC_RELAXED,
                                           __HIP_MEMORY_SCOPE_AGENT);
            __builtin_amdgcn_s_waitcnt(0x0F70);             // vmcnt(0): the stores have reached the coherence point
            const bool dropped = a.debug_drop && sg == 0 && ub == PARTNERS - 1 && t == 0;
            if (lane == 0 && !dropped) __hip_atomic_fetch_add(counters + rb, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#pragma unroll
            for (int v = 0; v < 4; ++v) {
                if (sv[v] < 0) continue;
#pragma unroll
                for (int uu = 0; uu < NUU; ++uu) a.cbuf[((int64_t)dir * n + sv[v]) * kH + u0 + 16 * uu + l16] = cpv[v][uu];
#pragma unroll
                for (int pp = 0; pp < NUU / 2; ++pp)
                    *reinterpret_cast<unsigned*>(a.out + (int64_t)rowv[v] * (2 * kH) + dir * kH + pair_unit + 32 * pp) = hw[v][pp];
            }
        }
    }
}

}  // namespace

// One bidirectional LSTM layer over n sequences in one launch of persistent workgroups.  `counters` must hold
// kLstmCounterCount + kLstmErrorWords unsigned ints (the counters are cleared here, the error words by the caller at the start of a forward).
// Returns OCRHIP_ESTATE when the hidden size is not 256 or the device cannot hold the grid at once; the caller then
// uses the per-step kernel.
namespace {
constexpr int kMaxDevices = 64;
std::mutex g_mutex;                         // guards the per-device state below and orders persistent launches
int g_max_groups[kMaxDevices];              // sequence groups such that 16 * G workgroups are resident at once (0: not probed, -1: none)
hipEvent_t g_last_launch[kMaxDevices];      // recorded behind the last persistent launch on the device
}  // namespace

int launch_lstm_layer(const float* xw_f, const float* xw_b, const float* whh_f, const float* whh_b, float* hbuf, float* cbuf,
                      float* out, int n, int T, int H, const int* seq_T, const int* seq_row0, const int* step_active,
                      unsigned* counters, hipStream_t s) {
    OCR_REQUIRE(xw_f && xw_b && whh_f && whh_b && hbuf && cbuf && out && counters, "lstm_layer: null argument");
    OCR_REQUIRE(n > 0 && T > 0, "lstm_layer: bad sizes n=%d T=%d", n, T);
    OCR_REQUIRE((seq_T == nullptr) == (seq_row0 == nullptr) && (seq_T == nullptr) == (step_active == nullptr),
                "lstm_layer: seq_T, seq_row0 and step_active go together");
    if (H != kH) {
        set_error("lstm_layer: the persistent kernel holds W_hh slices of %d hidden units in LDS, the model has %d (the per-step kernel runs instead)", kH, H);
        return OCRHIP_ESTATE;
    }
    int dev = 0;
    OCR_HIP_CHECK(hipGetDevice(&dev));
    OCR_REQUIRE(dev >= 0 && dev < kMaxDevices, "lstm_layer: device index %d", dev);
    std::lock_guard<std::mutex> lock(g_mutex);
    int& mg = g_max_groups[dev];
    if (mg == 0) {
        OCR_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(&lstm_layer_kernel<false>), hipFuncAttributeMaxDynamicSharedMemorySize, kLdsBytes));
        OCR_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(&lstm_layer_kernel<true>), hipFuncAttributeMaxDynamicSharedMemorySize, kLdsBytes));
        hipDeviceProp_t prop;
        OCR_HIP_CHECK(hipGetDeviceProperties(&prop, dev));
        int per_cu = 0;
        OCR_HIP_CHECK(hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, reinterpret_cast<const void*>(&lstm_layer_kernel<true>), 256, kLdsBytes));
        // the kernel of independent waves: workgroups of 8 waves with the W_hh image alone in LDS
        OCR_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(&lstm_layer_waves_kernel<false>), hipFuncAttributeMaxDynamicSharedMemorySize, kWFloats * 4));
        OCR_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(&lstm_layer_waves_kernel<true>), hipFuncAttributeMaxDynamicSharedMemorySize, kWFloats * 4));
        int per_cu_w = 0;
        OCR_HIP_CHECK(hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu_w, reinterpret_cast<const void*>(&lstm_layer_waves_kernel<true>), 512, kWFloats * 4));
        per_cu = std::min(per_cu, per_cu_w);
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
    a.counters = counters; a.error = counters + kLstmCounterCount; a.n = n; a.T = T;
    a.debug_drop = lstm_debug_drop();
    a.spin_cap = a.debug_drop ? (1 << 12) : (1 << 22);       // ~1 us per spin: seconds before a real wait gives up
    a.G = std::max(1, std::min(mg, (n + 15) / 16));          // no more groups than 16-row tiles of work
    if (((n + a.G - 1) / a.G + kRows - 1) / kRows > kMaxPass) {
        set_error("lstm_layer: more passes per workgroup than counters (n=%d)", n);
        return OCRHIP_ESTATE;
    }
    static_assert(kLstmCounterCount >= 2 * 16 * kMaxPass, "counter buffer size");
    // Two persistent grids on one device would each get part of the CUs and wait for workgroups that cannot start:
    // a launch waits for the persistent launch before it, whichever stream or handle of this process made it.
    if (g_last_launch[dev])
        OCR_HIP_CHECK(hipStreamWaitEvent(s, g_last_launch[dev], 0));
    else
        OCR_HIP_CHECK(hipEventCreateWithFlags(&g_last_launch[dev], hipEventDisableTiming));
    OCR_HIP_CHECK(hipMemsetAsync(counters, 0, kLstmCounterCount * sizeof(unsigned), s));
    // a plain launch: the grid is sized to what the device holds at once (checked above), which is all the hand-over
    // needs; the cooperative-launch entry point, which performs the same check, crashes the profiler at process exit
    // independent waves (blocks of 16 sequences, one counter per block) when the counters reach; else passes of 112
    // (a group with fewer than 8 blocks would leave waves idle: the pass kernel, whose waves share every row, is faster)
    const int blocks = ((n + a.G - 1) / a.G + kBlockRows - 1) / kBlockRows;
    const bool waves = ((lstm_persistent() == 2 && blocks >= 8) || lstm_persistent() == 3) && blocks <= kMaxPass;
    if (waves) {
        if (seq_T)
            hipLaunchKernelGGL(lstm_layer_waves_kernel<true>, dim3(16 * a.G), dim3(512), kWFloats * 4, s, a);
        else
            hipLaunchKernelGGL(lstm_layer_waves_kernel<false>, dim3(16 * a.G), dim3(512), kWFloats * 4, s, a);
    } else if (seq_T)
        hipLaunchKernelGGL(lstm_layer_kernel<true>, dim3(16 * a.G), dim3(256), kLdsBytes, s, a);
    else
        hipLaunchKernelGGL(lstm_layer_kernel<false>, dim3(16 * a.G), dim3(256), kLdsBytes, s, a);
    OCR_HIP_CHECK(hipGetLastError());
    OCR_HIP_CHECK(hipEventRecord(g_last_launch[dev], s));
    return OCRHIP_OK;
}

// The same layer on bfloat16 operands (see lstm_layer_waves16_kernel): hbuf16 (2, 2, n, 256) and out16 (R, 512) are
// bfloat16, xw and cbuf float32, whh the float32 checkpoint matrices (converted while the LDS image is built).
// OCRHIP_ESTATE under the same conditions as launch_lstm_layer, and when the handle asks for one launch per step.
int launch_lstm_layer16(const float* xw_f, const float* xw_b, const float* whh_f, const float* whh_b, uint16_t* hbuf16, float* cbuf,
                        uint16_t* out16, int n, int T, int H, const int* seq_T, const int* seq_row0, const int* step_active,
                        unsigned* counters, hipStream_t s) {
    OCR_REQUIRE(xw_f && xw_b && whh_f && whh_b && hbuf16 && cbuf && out16 && counters, "lstm_layer16: null argument");
    OCR_REQUIRE(n > 0 && T > 0, "lstm_layer16: bad sizes n=%d T=%d", n, T);
    OCR_REQUIRE((seq_T == nullptr) == (seq_row0 == nullptr) && (seq_T == nullptr) == (step_active == nullptr),
                "lstm_layer16: seq_T, seq_row0 and step_active go together");
    if (H != kH || lstm_persistent() == 0) {
        set_error("lstm_layer16: the persistent kernel is off or the model has %d hidden units (the per-step kernel runs instead)", H);
        return OCRHIP_ESTATE;
    }
    int dev = 0;
    OCR_HIP_CHECK(hipGetDevice(&dev));
    OCR_REQUIRE(dev >= 0 && dev < kMaxDevices, "lstm_layer16: device index %d", dev);
    std::lock_guard<std::mutex> lock(g_mutex);
    static int max_groups16[kMaxDevices];
    int& mg = max_groups16[dev];
    // 64 units per workgroup (`lstm16_units`, default): a 128 KB image, 4 workgroups per (direction, group), up to 32 groups
    const int UW = lstm16_units() == 32 ? 32 : 64;
    const int partners = kH / UW;
    const int kLds16 = 4 * UW * kH * 2;
    if (mg == 0) {
        OCR_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(&lstm_layer_waves16_kernel<false, 32>), hipFuncAttributeMaxDynamicSharedMemorySize, 128 * kH * 2));
        OCR_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(&lstm_layer_waves16_kernel<true, 32>), hipFuncAttributeMaxDynamicSharedMemorySize, 128 * kH * 2));
        OCR_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(&lstm_layer_waves16_kernel<false, 64>), hipFuncAttributeMaxDynamicSharedMemorySize, 256 * kH * 2));
        OCR_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(&lstm_layer_waves16_kernel<true, 64>), hipFuncAttributeMaxDynamicSharedMemorySize, 256 * kH * 2));
        hipDeviceProp_t prop;
        OCR_HIP_CHECK(hipGetDeviceProperties(&prop, dev));
        int per_cu = 0;
        OCR_HIP_CHECK(hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, reinterpret_cast<const void*>(&lstm_layer_waves16_kernel<true, 64>), 512, 256 * kH * 2));
        mg = per_cu >= 1 ? prop.multiProcessorCount : -1;          // workgroups the device holds at once (one per CU counted)
    }
    if (mg < 0) {
        set_error("lstm_layer16: the device cannot hold the persistent grid at once");
        return OCRHIP_ESTATE;
    }
    LayerArgs16 a;
    a.xw[0] = xw_f; a.xw[1] = xw_b; a.whh[0] = whh_f; a.whh[1] = whh_b;
    a.hbuf = hbuf16; a.cbuf = cbuf; a.out = out16; a.seq_T = seq_T; a.seq_row0 = seq_row0; a.step_active = step_active;
    a.counters = counters; a.error = counters + kLstmCounterCount; a.n = n; a.T = T;
    a.debug_drop = lstm_debug_drop();
    a.spin_cap = a.debug_drop ? (1 << 12) : (1 << 22);
    a.G = std::max(1, std::min(std::min(mg / (2 * partners), kLstmCounterCount / (2 * kMaxPass)), (n + 15) / 16));
    const int blocks = ((n + a.G - 1) / a.G + kBlockRows - 1) / kBlockRows;
    if (blocks > kMaxPass) {
        set_error("lstm_layer16: more blocks per workgroup than counters (n=%d)", n);
        return OCRHIP_ESTATE;
    }
    if (g_last_launch[dev])
        OCR_HIP_CHECK(hipStreamWaitEvent(s, g_last_launch[dev], 0));
    else
        OCR_HIP_CHECK(hipEventCreateWithFlags(&g_last_launch[dev], hipEventDisableTiming));
    OCR_HIP_CHECK(hipMemsetAsync(counters, 0, kLstmCounterCount * sizeof(unsigned), s));
    const dim3 grid(2 * partners * a.G);
    if (UW == 64) {
        if (seq_T)
            hipLaunchKernelGGL((lstm_layer_waves16_kernel<true, 64>), grid, dim3(512), kLds16, s, a);
        else
            hipLaunchKernelGGL((lstm_layer_waves16_kernel<false, 64>), grid, dim3(512), kLds16, s, a);
    } else if (seq_T)
        hipLaunchKernelGGL((lstm_layer_waves16_kernel<true, 32>), grid, dim3(512), kLds16, s, a);
    else
        hipLaunchKernelGGL((lstm_layer_waves16_kernel<false, 32>), grid, dim3(512), kLds16, s, a);
    OCR_HIP_CHECK(hipGetLastError());
    OCR_HIP_CHECK(hipEventRecord(g_last_launch[dev], s));
    return OCRHIP_OK;
}

}  // namespace ocrhip
