// 3x3 convolution (stride 1, zero padding 1) by the Winograd minimal-filtering algorithm F(2x2, 3x3) on the
// f32 matrix cores, NHWC float32.
//
// Same layers as conv_mfma.hip's 3x3 variant (VGG16-BN body and merge branch of EAST, architectures/
// east_arch.py:114-151; feature extractors of the recogniser, architectures/crnn_arch.py:38-41,330-477), same
// folded batch-norm, bias, ReLU, fused max-pool and residual operand, float32 data and float32 accumulation.
// Every 2x2 block of outputs is computed from its 4x4 input patch d and the 3x3 filter g as
//     Y = A^T [ (G g G^T) (.) (B^T d B) ] A,           summed over the input channels before the A transform,
//     B^T = [1 0 -1 0; 0 1 1 0; 0 -1 1 0; 0 1 0 -1],   G = [1 0 0; .5 .5 .5; .5 -.5 .5; 0 0 1],   A^T = [1 1 1 0; 0 1 -1 -1]
// which needs 16 multiplications per input channel for 4 outputs instead of 36: the matrix cores execute
// 2.25x fewer flops than the direct convolution performs.  The filter transform is done once at load time
// (in double, rounded to float32); the data transform only adds and subtracts.
//
// Tiling (one workgroup = 4 waves, two workgroups per CU): 4 rows x 32 columns of outputs = 2 x 16 blocks of 2x2,
// times 64 output channels.  The 16 Winograd components are 16 independent GEMMs (blocks x Cin) x (Cin x Cout).
//   wave = (fh, ch): components 8 fh .. 8 fh + 7 (rows 2 fh, 2 fh + 1 of the 4x4 component matrix) of 32 blocks x
//   channels 32 ch .. 32 ch + 31 = 8 accumulators of 32x32 = 128 registers per lane, so two waves fit on a SIMD and
//   a workgroup needs under 80 KB of LDS: while one wave waits (barrier, loads, prologue, epilogue) the wave of the
//   other workgroup on the same SIMD issues MFMAs.  (A lone wave per SIMD with all 16 components and every other
//   instruction hand-placed between its MFMAs was measured 8 % slower.)
//   K: chunks of 8 input channels.  Per chunk the raw input tile with its halo (6 x 34 pixels) is staged in LDS,
//   transformed into U[component][block][channel] in LDS (conflict-free 16-byte reads, block stride 12 floats), and
//   each lane loads its transformed-filter fragments straight from the lane-linear weight slab (8 coalesced 16-byte
//   loads per chunk).  U and the raw tile are double buffered; one barrier per chunk.
//   Epilogue: each wave applies the column half of the inverse transform to its two component rows, the two waves of
//   a channel half leave their partial sums of the output rows in two LDS staging areas, and the output pass adds them
//   (plus bias, residual, ReLU, optional max-pool) and writes 16-byte stores.
//
// Roofline: MFMA bound; the unit of work is the direct convolution's 2*9*Cin*Cout flop per output pixel, of
// which 16/36 are executed.
#include <type_traits>

#include <mutex>

#include "common.h"

namespace ocrhip {

namespace {

typedef float floatx16 __attribute__((ext_vector_type(16)));
typedef float floatx4 __attribute__((ext_vector_type(4)));
typedef unsigned int uintx4 __attribute__((ext_vector_type(4)));
typedef float floatx2 __attribute__((ext_vector_type(2)));

// Packed f32 arithmetic as inline assembly: one instruction for two floats at the price of one.  Written as vector
// expressions the compiler selects the same instructions but splits them into scalar ones again wherever MFMAs are
// near, which doubles the vector-ALU time between the MFMAs.
__device__ __forceinline__ floatx2 pk_fma(floatx2 a, floatx2 b, floatx2 c) {
    floatx2 d;
    asm("v_pk_fma_f32 %0, %1, %2, %3" : "=v"(d) : "v"(a), "v"(b), "v"(c));
    return d;
}
__device__ __forceinline__ floatx2 pk_add(floatx2 a, floatx2 b) {
    floatx2 d;
    asm("v_pk_add_f32 %0, %1, %2" : "=v"(d) : "v"(a), "v"(b));
    return d;
}
__device__ __forceinline__ floatx2 pk_sub(floatx2 a, floatx2 b) {
    floatx2 d;
    asm("v_pk_add_f32 %0, %1, %2 neg_lo:[0,1] neg_hi:[0,1]" : "=v"(d) : "v"(a), "v"(b));
    return d;
}     // arithmetic on these compiles to packed f32 instructions

constexpr int kCK = 8;            // input channels per chunk
constexpr int kRS = 12;           // floats per raw pixel in LDS (8 + pad: conflict-free 16-byte reads)
constexpr int kUS = 12;           // floats per (component, block) row of U
constexpr int kRawW = 34, kRawH = 6, kRawPx = kRawW * kRawH;
constexpr int kBlocks = 32;       // 2 block rows x 16 block columns
constexpr int kUC = kBlocks * kUS;            // floats per component of U
constexpr int kUFloats = 16 * kUC;            // one U buffer: 24 KB
constexpr int kRFloats = kRawPx * kRS;        // one raw-tile buffer: 9.6 KB
constexpr int kRBuf = kRFloats + 4;            // + a spare slot for the threads without a pixel
constexpr int kLoopFloats = 2 * kUFloats + 2 * kRBuf;
// CHB = channel blocks of 32 per workgroup (2 or 4); the epilogue stages 2 x 128 pixels x (32 CHB + 4) floats
constexpr int lds_bytes(int chb) {
    return 4 * (kLoopFloats > 2 * 128 * (32 * chb + 4) ? kLoopFloats : 2 * 128 * (32 * chb + 4));
}

struct WinoArgs {
    const float* x;        // (B,H,W,cin)
    const float* w;        // transformed filters, slab layout of prepare_conv_wino
    const float* bias;     // cout_pad
    const float* res;      // optional residual (POOL == 0), shaped like y
    float* y;
    int B, H, W, cin, cout, relu;
    int tiles_x, tiles_per_image, n_blocks_n;
    int vP, vB;            // vP > 0: the vB images of the batch lie side by side, vP columns apart (zeros between), and
                           // the tiles cover that one wide image; B is 1 then and W stays the width of one image
};

template <int POOL, int CHB>
__global__ __launch_bounds__(128 * CHB, CHB == 2 ? 2 : 1) void conv_wino_f32_kernel(const WinoArgs a) {
    constexpr int NT = 128 * CHB;               // threads: 2 CHB waves, two per SIMD either way
    constexpr int NCH = 32 * CHB;               // output channels of the workgroup
    constexpr int kOS = NCH + 4;                // floats per staged output pixel
    constexpr int kStage = 128 * kOS;           // one staging area of the epilogue: 4 x 32 pixels
    extern __shared__ __attribute__((aligned(16))) float lds[];      // U[2] then raw[2]: chunk j lives in buffer j & 1

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int li = lane & 31, lh = lane >> 5;
    const int fh = wave & 1;          // which 8 of the 16 components
    const int ch = wave >> 1;         // which 32 of the NCH output channels

    // Workgroups go to the 8 XCDs round-robin; the work index is permuted so that every XCD walks a contiguous range
    // of it with the channel block varying fastest: the channel blocks of one spatial tile run on the same XCD at
    // about the same time and share the input tile through its L2.
    const unsigned total = gridDim.x, xcd = blockIdx.x & 7u, slot = blockIdx.x >> 3;
    const unsigned work = xcd * (total >> 3) + min(xcd, total & 7u) + slot;
    const int nb = (int)(work % (unsigned)a.n_blocks_n);
    const int tile_all = (int)(work / (unsigned)a.n_blocks_n);
    const int b = tile_all / a.tiles_per_image;
    const int tile = tile_all % a.tiles_per_image;
    const int y0 = (tile / a.tiles_x) * 4, x0 = (tile % a.tiles_x) * 32;
    const int nK = a.cin / kCK;

    floatx16 acc[8];          // first written by the MFMAs of chunk 0, which take a literal zero as their addend

    // Global loads go through buffer descriptors: a per-lane 32-bit byte offset that never changes, the chunk as a
    // scalar offset, and the hardware's range check, which returns zeros for an offset beyond the buffer - the zero
    // padding around the image costs no instruction (lanes outside the image carry the offset 2^31).
    // weight slab of a block of 64 channels: [k chunk][component][channel half][lane][4]
    const float* wslab = a.w + (int64_t)(nb * (CHB / 2) + (ch >> 1)) * nK * (16 * 2 * 64 * 4);
    const __amdgpu_buffer_rsrc_t w_rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)wslab, 0, nK * (16 * 2 * 64 * 16), 0x00020000);
    const int w_off0 = ((fh * 8) * 128 + (ch & 1) * 64 + lane) * 16, w_off1 = w_off0 + 2048;
    auto load_w = [&](int kc, int f) {          // fragment of component 8 fh + f of chunk kc
        return __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(w_rsrc, (f & 1) ? w_off1 : w_off0,
                                                                                kc * (16 * 2 * 64 * 16) + (f >> 1) * 4096, 0));
    };

    // raw tile: 204 pixels x 2 channel quads over the threads (two or one per thread); offsets are fixed per thread;
    // threads without a pixel store to a spare slot
    constexpr int NRAW = (kRawPx * 2 + NT - 1) / NT;
    const int64_t img_floats = (int64_t)a.H * a.W * a.cin;
    const __amdgpu_buffer_rsrc_t x_rsrc = __builtin_amdgcn_make_buffer_rsrc(
        (void*)(a.x + (a.vP > 0 ? 0 : b * img_floats)), 0, (int)((a.vP > 0 ? a.vB : 1) * img_floats * 4), 0x00020000);
    float4 ra[NRAW];
    int raw_src[NRAW], raw_dst[NRAW];
#pragma unroll
    for (int i = 0; i < NRAW; ++i) {
        const int f = tid + i * NT;
        const int px = f >> 1, c4 = f & 1;
        const int gy = y0 - 1 + px / kRawW;
        int gx = x0 - 1 + px % kRawW, gb = 0;
        bool in_x = gx >= 0;
        if (a.vP > 0 && in_x) {
            gb = gx / a.vP;
            gx -= gb * a.vP;
            in_x = gb < a.vB;
        }
        const bool ok = f < kRawPx * 2 && gy >= 0 && gy < a.H && in_x && gx < a.W;
        raw_src[i] = ok ? ((((gb * a.H + gy) * a.W + gx) * a.cin + c4 * 4) * 4) : (int)0x80000000;
        raw_dst[i] = (f < kRawPx * 2) ? (px * kRS + c4 * 4) : kRFloats;
    }
    auto load_raw = [&](int kc) {
#pragma unroll
        for (int i = 0; i < NRAW; ++i)
            ra[i] = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(x_rsrc, raw_src[i], kc * (kCK * 4), 0));
    };
    auto store_raw = [&](int rbase) {
#pragma unroll
        for (int i = 0; i < NRAW; ++i) *reinterpret_cast<float4*>(&lds[rbase + raw_dst[i]]) = ra[i];
    };

    // Data transform: a thread owns (block, 4 channels, component row i):
    //   i = 0: d0 - d2,  i = 1: d1 + d2,  i = 2: d2 - d1,  i = 3: d1 - d3     (rows of B^T d) = dA + sgn * dB,
    // then (.) B along the columns: components j = 0..3 = t0 - t2, t1 + t2, t2 - t1, t1 - t3 of the row's 4 columns t.
    // With 8 waves a thread does half of that: columns c0 .. c0 + 2 (c0 = 0: j = 0, 1;  c0 = 1: j = 3, 2), as
    //   d = first - third -> j = 0 | 3,     e = second + third | second - first -> j = 1 | 2.
    constexpr int TC = (CHB == 2) ? 4 : 3;          // raw columns a thread reads
    const int t_hs = (CHB == 2) ? 0 : __builtin_amdgcn_readfirstlane(tid >> 8);
    const int t_cq = tid & 1, t_tx = (tid >> 1) & 15, t_i = (tid >> 5) & 3, t_ty = (tid >> 7) & 1;
    const int t_src = ((2 * t_ty) * kRawW + 2 * t_tx + t_hs) * kRS + t_cq * 4;
    const int rowA = (t_i == 0) ? 0 : (t_i == 2 ? 2 : 1);
    const int rowB = (t_i == 0 || t_i == 1) ? 2 : (t_i == 2 ? 1 : 3);
    const int t_offA = t_src + rowA * kRawW * kRS, t_offB = t_src + rowB * kRawW * kRS;
    const float t_sgn1 = (t_i == 1) ? 1.0f : -1.0f;
    const floatx2 t_sgn = {t_sgn1, t_sgn1};
    const int t_dst = t_i * 4 * kUC + (t_ty * 16 + t_tx) * kUS + t_cq * 4;
    // 8 waves: e = second + s0 * first + s2 * third with (s0, s2) = (0, 1) for j = 1 and (-1, 0) for j = 2
    const floatx2 t_s0 = {t_hs ? -1.0f : 0.0f, t_hs ? -1.0f : 0.0f}, t_s2 = {t_hs ? 0.0f : 1.0f, t_hs ? 0.0f : 1.0f};
    const int t_jd = t_hs ? 3 * kUC : 0, t_je = t_hs ? 2 * kUC : kUC;
    floatx4 tp[TC], tq[TC];
    floatx2 tt[TC][2];                     // B^T d row of this thread: column, channel pair
    auto tr_read = [&](int rbase) {
#pragma unroll
        for (int c = 0; c < TC; ++c) {
            tp[c] = *reinterpret_cast<const floatx4*>(&lds[rbase + t_offA + c * kRS]);
            tq[c] = *reinterpret_cast<const floatx4*>(&lds[rbase + t_offB + c * kRS]);
        }
    };
    auto tr_row = [&](int c) {
        if (c < TC) {
            tt[c][0] = pk_fma(t_sgn, tq[c].xy, tp[c].xy);
            tt[c][1] = pk_fma(t_sgn, tq[c].zw, tp[c].zw);
        }
    };
    auto put2 = [&](float* u, floatx2 lo, floatx2 hi) {       // two 8-byte stores: no need for 4 consecutive registers
        *reinterpret_cast<floatx2*>(u) = lo;
        *reinterpret_cast<floatx2*>(u + 2) = hi;
    };
    auto tr_out = [&](int ubase, int half) {        // half 0 / 1: the first / second half of the thread's U stores
        float* u = &lds[ubase + t_dst];
        if (CHB == 2) {
            if (half == 0) {
                put2(u, pk_sub(tt[0][0], tt[2][0]), pk_sub(tt[0][1], tt[2][1]));
                put2(u + kUC, pk_add(tt[1][0], tt[2][0]), pk_add(tt[1][1], tt[2][1]));
            } else {
                put2(u + 2 * kUC, pk_sub(tt[2][0], tt[1][0]), pk_sub(tt[2][1], tt[1][1]));
                put2(u + 3 * kUC, pk_sub(tt[1][0], tt[TC - 1][0]), pk_sub(tt[1][1], tt[TC - 1][1]));
            }
        } else {
            if (half == 0) {
                put2(u + t_jd, pk_sub(tt[0][0], tt[2][0]), pk_sub(tt[0][1], tt[2][1]));
            } else {
                put2(u + t_je, pk_fma(t_s2, tt[2][0], pk_fma(t_s0, tt[0][0], tt[1][0])),
                     pk_fma(t_s2, tt[2][1], pk_fma(t_s0, tt[0][1], tt[1][1])));
            }
        }
    };
    auto transform = [&](int rbase, int ubase) {
        tr_read(rbase);
#pragma unroll
        for (int c = 0; c < 4; ++c) tr_row(c);
        tr_out(ubase, 0);
        tr_out(ubase, 1);
    };

    float4 bw[8], af[2][4];              // filter fragments of a chunk; A fragments of the two groups of 4 components
    const int u_src = (fh * 8) * kUC + li * kUS + lh * 4;
    auto af_read = [&](int ubase, int g) {
#pragma unroll
        for (int i = 0; i < 4; ++i) af[g][i] = *reinterpret_cast<const float4*>(&lds[ubase + u_src + (g * 4 + i) * kUC]);
    };

    // prologue: chunk 0 transformed into U[0], raw tile 1 in raw[1], raw tile 2 on its way to registers, filter
    // fragments of chunk 0 and the A fragments of its first group in registers
    constexpr int kR0 = 2 * kUFloats, kR1 = 2 * kUFloats + kRBuf;
    load_raw(0);
#pragma unroll
    for (int f = 0; f < 8; ++f) bw[f] = load_w(0, f);
    store_raw(kR0);
    load_raw(min(1, nK - 1));
    __syncthreads();
    transform(kR0, 0);
    store_raw(kR1);
    load_raw(min(2, nK - 1));
    __syncthreads();
    af_read(0, 0);

    // Chunk kc = 32 MFMAs in two groups of 4 components x 4 k steps, one barrier in the middle, no branches; everything
    // a group needs was requested a group earlier.  On entry: U[kc & 1] complete, the A fragments of group 0 and the
    // filter fragments of the chunk in registers, raw tile kc + 1 in raw[(kc + 1) & 1], raw tile kc + 2 in flight.
    //   group 0: A fragments of group 1 requested; transform of chunk kc + 1 (raw[(kc + 1) & 1] -> U[(kc + 1) & 1]) between
    //            the MFMAs; raw tile kc + 2 into raw[kc & 1]; filter fragments 0..3 reloaded for chunk kc + 1
    //   barrier: U[(kc + 1) & 1] and raw tile kc + 2 visible; group 1 works from registers
    //   group 1: A fragments of group 0 of chunk kc + 1 and raw tile kc + 3 requested; filter fragments 4..7 reloaded
    // Past the end of K the same loads and stores run on the last chunk again and their results are not used.
    // The loop is unrolled over the buffer parity, so every LDS address is a register that never changes plus an
    // immediate: between the MFMAs the vector ALU only does the transform's arithmetic (on this hardware every other
    // vector instruction delays the next MFMA by its own duration, even when a second wave has MFMAs ready).
    auto chunk = [&](int kc, auto parity, auto first_chunk) {
        constexpr int P = decltype(parity)::value;
        constexpr bool kFirst = decltype(first_chunk)::value;
        constexpr int ucur = P * kUFloats, unxt = (1 - P) * kUFloats;
        constexpr int rnxt = kR0 + (1 - P) * kRBuf, rfill = kR0 + P * kRBuf;
        const int kn = min(kc + 1, nK - 1);
        af_read(ucur, 1);
        tr_read(rnxt);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int g = 0; g < 2; ++g) {
#pragma unroll
            for (int t = 0; t < 4; ++t) {
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const int f = g * 4 + i;
                    const float4 a4 = af[g][i], b4 = bw[f];
                    const float av = t == 0 ? a4.x : (t == 1 ? a4.y : (t == 2 ? a4.z : a4.w));
                    const float bv = t == 0 ? b4.x : (t == 1 ? b4.y : (t == 2 ? b4.z : b4.w));
                    if (kFirst && t == 0) {
                        const floatx16 zero = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
                        acc[f] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bv, zero, 0, 0, 0);
                    } else {
                        acc[f] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bv, acc[f], 0, 0, 0);
                    }
                }
                if (g == 0) {
                    if (t < 2) {
                        tr_row(2 * t);
                        tr_row(2 * t + 1);
                    }
                    if (t == 2) tr_out(unxt, 0);
                    if (t == 3) {
                        tr_out(unxt, 1);
                        store_raw(rfill);
                    }
                }
                __builtin_amdgcn_sched_barrier(0);
            }
#pragma unroll
            for (int i = 0; i < 4; ++i) bw[g * 4 + i] = load_w(kn, g * 4 + i);
            if (g == 0) {
                __syncthreads();
                af_read(unxt, 0);
                load_raw(min(kc + 3, nK - 1));
                __builtin_amdgcn_sched_barrier(0);
            }
        }
    };
    chunk(0, std::integral_constant<int, 0>{}, std::true_type{});          // nK is even and at least 2
    chunk(1, std::integral_constant<int, 1>{}, std::false_type{});
    for (int kc = 2; kc < nK; kc += 2) {
        chunk(kc, std::integral_constant<int, 0>{}, std::false_type{});
        chunk(kc + 1, std::integral_constant<int, 1>{}, std::false_type{});
    }

    // ---- inverse transform.  Register v of a lane belongs to block m = (v & 3) + 8 * (v >> 2) + 4 * (lane >> 5) of the
    // 32 (2 block rows x 16 block columns) and to output channel lane & 31 of the wave's 32.  With Z_i = (row i of the
    // component matrix) A, the outputs are Y row 0 = Z0 + Z1 + Z2 and Y row 1 = Z1 - Z2 - Z3: the wave with rows 0, 1
    // stages (Z0 + Z1, Z1), the wave with rows 2, 3 stages (Z2, -Z2 - Z3), and the output pass adds the two.
    __syncthreads();                                           // every wave is done with U
    float* O = lds + fh * kStage;
    const int oc = ch * 32 + li;
    floatx16 r0[2], r1[2];                 // this wave's share of output rows 0 and 1, columns c = 0, 1 of every block
    {
        const floatx16 z00 = acc[0] + acc[1] + acc[2], z01 = acc[1] - acc[2] - acc[3];
        const floatx16 z10 = acc[4] + acc[5] + acc[6], z11 = acc[5] - acc[6] - acc[7];
        if (fh == 0) {
            r0[0] = z00 + z10; r0[1] = z01 + z11;
            r1[0] = z10;       r1[1] = z11;
        } else {
            r0[0] = z00;        r0[1] = z01;
            r1[0] = -z00 - z10; r1[1] = -z01 - z11;
        }
    }
#pragma unroll
    for (int v = 0; v < 16; ++v) {
        const int m = (v & 3) + 8 * (v >> 2) + 4 * lh;
        const int ty = m >> 4, tx = m & 15;
#pragma unroll
        for (int c = 0; c < 2; ++c) {
            O[((2 * ty) * 32 + 2 * tx + c) * kOS + oc] = r0[c][v];
            O[((2 * ty + 1) * 32 + 2 * tx + c) * kOS + oc] = r1[c][v];
        }
    }
    __syncthreads();
    // output pass: thread -> (pixel, 4 channels); sum of the two staging areas, max-pool, bias, residual, ReLU
    constexpr int kOutW = (POOL == 1) ? 16 : 32;               // output pixels per tile row
    constexpr int kOutPx = (POOL == 0) ? 128 : (POOL == 1 ? 32 : 64);
    const int c4 = tid % (NCH / 4);
    const int co0 = nb * NCH + c4 * 4;
    const floatx4 bias4 = *reinterpret_cast<const floatx4*>(a.bias + co0);          // bias is padded to cout_pad
    const int Ho = (POOL == 0) ? a.H : a.H / 2, Wo = (POOL == 1) ? a.W / 2 : a.W;
    const int oy0 = (POOL == 0) ? y0 : (y0 >> 1), ox0 = (POOL == 1) ? (x0 >> 1) : x0;
    const bool vec_ok = (a.cout & 3) == 0 && co0 + 3 < a.cout;
    auto staged = [&](int row, int col) {
        return *reinterpret_cast<const floatx4*>(&lds[(row * 32 + col) * kOS + c4 * 4]) +
               *reinterpret_cast<const floatx4*>(&lds[kStage + (row * 32 + col) * kOS + c4 * 4]);
    };
    auto max4 = [](floatx4 p, floatx4 q) { return floatx4{fmaxf(p.x, q.x), fmaxf(p.y, q.y), fmaxf(p.z, q.z), fmaxf(p.w, q.w)}; };
    // all staged values first (one LDS round trip for the whole pass), then the stores
    constexpr int kIters = kOutPx / 16;
    floatx4 vals[kIters];
#pragma unroll
    for (int it = 0; it < kIters; ++it) {
        const int px = tid / (NCH / 4) + it * 16;
        const int py = px / kOutW, pxx = px % kOutW;          // = (it >> 1, p0 + 16 (it & 1)) or (it, p0)
        if (POOL == 0) vals[it] = staged(py, pxx);
        else if (POOL == 1) vals[it] = max4(max4(staged(2 * py, 2 * pxx), staged(2 * py, 2 * pxx + 1)), max4(staged(2 * py + 1, 2 * pxx), staged(2 * py + 1, 2 * pxx + 1)));
        else vals[it] = max4(staged(2 * py, pxx), staged(2 * py + 1, pxx));
    }
    // A thread's pixels are p0 + 16 it: the same one or two columns in successive rows.  Column validity, the image a
    // column belongs to and the column's share of the address are worked out once; the row's share is uniform.
    constexpr int NJ = (kOutW == 32) ? 2 : 1;
    const int p0 = tid / (NCH / 4);
    bool col_ok[NJ];
    int64_t col_idx[NJ];
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
        int xx = ox0 + p0 + 16 * j, ob = b;
        bool ok = true;
        if (a.vP > 0) {
            const int po = (POOL == 1) ? (a.vP >> 1) : a.vP;      // period of the images in output columns
            ob = xx / po;
            xx -= ob * po;
            ok = ob < a.vB;
        }
        col_ok[j] = ok && xx < Wo;
        col_idx[j] = ((int64_t)ob * Ho * Wo + xx) * a.cout + co0;
    }
    const int64_t row_stride = (int64_t)Wo * a.cout;
#pragma unroll
    for (int it = 0; it < kIters; ++it) {
        const int py = (kOutW == 32) ? (it >> 1) : it, j = (kOutW == 32) ? (it & 1) : 0;
        const int yy = oy0 + py;
        if (yy >= Ho) break;                  // uniform
        if (!col_ok[j]) continue;
        floatx4 val = vals[it] + bias4;
        const int64_t idx = col_idx[j] + yy * row_stride;
        if (vec_ok) {
            if (POOL == 0 && a.res) val += *reinterpret_cast<const floatx4*>(a.res + idx);
            if (a.relu) {
                val.x = fmaxf(val.x, 0.0f); val.y = fmaxf(val.y, 0.0f); val.z = fmaxf(val.z, 0.0f); val.w = fmaxf(val.w, 0.0f);
            }
            *reinterpret_cast<floatx4*>(a.y + idx) = val;
        } else {
            const float e[4] = {val.x, val.y, val.z, val.w};
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                if (co0 + k < a.cout) {
                    float o1 = e[k];
                    if (POOL == 0 && a.res) o1 += a.res[idx + k];
                    if (a.relu) o1 = fmaxf(o1, 0.0f);
                    a.y[idx + k] = o1;
                }
            }
        }
    }
}

}  // namespace

// Transformed-filter slab: [n block (64 channels)][k chunk (8)][component f = 4i + j][channel half c][h][j32][t]
// float32 with output channel = nblock * 64 + c * 32 + j32 and input channel = kchunk * 8 + 4h + t, i.e. lane
// (h * 32 + j32) of the wave that owns channel half c reads 16 consecutive bytes.
int prepare_conv_wino(ConvLayer& L, const float* w_oihw, int cout, int cin) {
    OCR_REQUIRE(cin % (2 * kCK) == 0 && cout > 0, "conv_wino: cin=%d must be a multiple of %d", cin, 2 * kCK);
    const int nN = (cout + 63) / 64, nK = cin / kCK;
    std::vector<float> w((size_t)nN * nK * 16 * 2 * 64 * 4, 0.0f);
    static const double G[4][3] = {{1, 0, 0}, {0.5, 0.5, 0.5}, {0.5, -0.5, 0.5}, {0, 0, 1}};
    for (int o = 0; o < cout; ++o)
        for (int c = 0; c < cin; ++c) {
            const float* g = w_oihw + ((size_t)o * cin + c) * 9;
            double tmp[4][3], V[4][4];
            for (int i = 0; i < 4; ++i)
                for (int k = 0; k < 3; ++k) tmp[i][k] = G[i][0] * g[0 * 3 + k] + G[i][1] * g[1 * 3 + k] + G[i][2] * g[2 * 3 + k];
            for (int i = 0; i < 4; ++i)
                for (int j = 0; j < 4; ++j) V[i][j] = tmp[i][0] * G[j][0] + tmp[i][1] * G[j][1] + tmp[i][2] * G[j][2];
            const int nbk = o / 64, chalf = (o % 64) / 32, j32 = o % 32;
            const int kc = c / kCK, hh = (c % kCK) / 4, t = c % 4;
            for (int f = 0; f < 16; ++f) {
                const size_t dst = ((((((size_t)(nbk * nK + kc) * 16 + f) * 2 + chalf) * 2 + hh) * 32 + j32) * 4) + t;
                w[dst] = (float)V[f / 4][f % 4];
            }
        }
    OCR_TRY(L.w_wino.reserve(w.size() * sizeof(float)));
    OCR_HIP_CHECK(hipMemcpy(L.w_wino.ptr, w.data(), w.size() * sizeof(float), hipMemcpyHostToDevice));
    return OCRHIP_OK;
}

int launch_conv_wino(const ConvLayer& L, const float* x, float* y, int B, int H, int W, int pool, bool relu,
                      const float* residual, hipStream_t s) {
    OCR_REQUIRE(L.ks == 3 && L.w_wino.ptr, "conv_wino: layer has no transformed filters");
    OCR_REQUIRE(x && y && B > 0 && H > 0 && W > 0, "conv_wino: bad arguments B=%d H=%d W=%d", B, H, W);
    OCR_REQUIRE(L.cout_pad % 64 == 0 && L.cin % (2 * kCK) == 0, "conv_wino: layer shape %d -> %d not supported", L.cin, L.cout);
    OCR_REQUIRE((int64_t)H * W * L.cin * 4 < (1ll << 31), "conv_wino: image of %dx%dx%d floats exceeds the 2 GB of a buffer descriptor", H, W, L.cin);
    OCR_REQUIRE(pool >= 0 && pool <= 2, "conv_wino: bad pool mode %d", pool);
    if (pool) OCR_REQUIRE(H % 2 == 0 && (pool == 2 || W % 2 == 0), "conv_wino: pooled layer needs even size, got %dx%d", H, W);
    OCR_REQUIRE(!residual || pool == 0, "conv_wino: a residual goes with an unpooled layer");
    WinoArgs a;
    a.x = x; a.w = L.w_wino.as<float>(); a.bias = L.bias.as<float>(); a.res = residual; a.y = y;
    a.B = B; a.H = H; a.W = W; a.cin = L.cin; a.cout = L.cout; a.relu = relu ? 1 : 0;
    a.tiles_x = (W + 31) / 32;
    a.vP = 0; a.vB = B;
    // A batch of narrow images whose width leaves the 32-column tiles partly empty is laid side by side as one wide
    // image, one zero column apart (two where the columns are pooled in pairs, so every image starts on an even
    // column): the gap is the zero padding of both neighbours, and the tiles of the wide image are full.
    if (B > 1 && W % 32 != 0) {
        const int period = W + (pool == 1 ? 2 : 1);
        const int64_t vw = (int64_t)B * period;
        if (vw < (1ll << 30) && (int64_t)B * H * W * L.cin * 4 < (1ll << 31) && (vw + 31) / 32 < (int64_t)a.tiles_x * B) {
            a.vP = period;
            a.B = 1;
            a.tiles_x = (int)((vw + 31) / 32);
        }
    }
    a.tiles_per_image = a.tiles_x * ((H + 3) / 4);
    // 128 channels per workgroup (8 waves, one workgroup per CU) for the deep layers: the transform and the raw tile are
    // then shared by twice as many MFMAs (measured +5 % at 256 and 512 input channels); 64 channels (4 waves, two
    // workgroups per CU) where K is short (the per-workgroup start-up and epilogue weigh more than the transform: equal
    // or slower at 64 input channels) or where 128-channel workgroups would not fill the CUs
    const int64_t wgs128 = (int64_t)a.tiles_per_image * (L.cout_pad / 128) * a.B;
    const int chb = (L.cout_pad % 128 == 0 && L.cin >= 128 && wgs128 >= 256) ? 4 : 2;
    a.n_blocks_n = L.cout_pad / (32 * chb);
    const int64_t wgs = (int64_t)a.tiles_per_image * a.n_blocks_n * a.B;
    OCR_REQUIRE(wgs < (1ll << 31), "conv_wino: too many workgroups");
    char pname[64];
    std::snprintf(pname, sizeof(pname), "conv_wino_f32<n%d,pool%d,tile4x32>", 32 * chb, pool);
    const double out_px = (double)B * H * W;
    const double out_elems = out_px * L.cout / (pool == 1 ? 4.0 : (pool == 2 ? 2.0 : 1.0));
    ProfScope prof(pname, 2.0 * 9 * L.cin * L.cout * out_px,
                   4.0 * (out_px * L.cin + out_elems * (residual ? 2.0 : 1.0) + 9.0 * L.cin * L.cout), s);
    const dim3 grid((unsigned)wgs), block(128 * chb);
    static bool lds_opt_in_dev[64] = {};        // more than 64 KB of LDS per workgroup needs an opt-in per kernel and device
    static std::mutex lds_opt_in_mutex;          // handles of several host threads launch through here
    int dev = 0;
    OCR_HIP_CHECK(hipGetDevice(&dev));
    std::lock_guard<std::mutex> lds_opt_in_lock(lds_opt_in_mutex);
    bool& lds_opt_in = lds_opt_in_dev[dev & 63];
#define OCR_WINO_FOR_ALL(X) X(0, 2) X(1, 2) X(2, 2) X(0, 4) X(1, 4) X(2, 4)
    if (!lds_opt_in) {
#define OCR_WINO_OPT_IN(P, C)                                                                                              \
    OCR_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_wino_f32_kernel<P, C>),                         \
                                      hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes(C)));
        OCR_WINO_FOR_ALL(OCR_WINO_OPT_IN)
#undef OCR_WINO_OPT_IN
        lds_opt_in = true;
    }
#define OCR_WINO_LAUNCH(P, C)                                                                                              \
    if (pool == P && chb == C) hipLaunchKernelGGL((conv_wino_f32_kernel<P, C>), grid, block, lds_bytes(C), s, a);
    OCR_WINO_FOR_ALL(OCR_WINO_LAUNCH)
#undef OCR_WINO_LAUNCH
#undef OCR_WINO_FOR_ALL
    OCR_HIP_CHECK(hipGetLastError());
    return OCRHIP_OK;
}

}  // namespace ocrhip
