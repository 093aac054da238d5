// 3x3 convolution (stride 1, zero padding 1) by the Winograd minimal-filtering algorithm F(4x4, 3x3) on the f32 matrix
// cores, NHWC float32.
//
// Same layers as conv_wino.hip (the wide 3x3 convolutions of architectures/east_arch.py:114-151 and of the feature
// extractors of architectures/crnn_arch.py:38-41,330-477), same folded batch-norm, bias, ReLU, fused 2x2 max-pool and
// residual operand.  Every 4x4 block of outputs is computed from its 6x6 input patch d and the 3x3 filter g as
//     Y = A^T [ (G g G^T) (.) (B^T d B) ] A           (interpolation points 0, +-1, +-2, infinity)
// i.e. 36 multiplications per input channel for 16 outputs instead of 144: the matrix cores execute 4x fewer flops
// than the direct convolution performs (F(2x2,3x3): 2.25x).  The price is arithmetic: the transforms multiply by
// constants up to 8 and 1/24, and the float32 result differs from the direct sum by up to ~1e-5 of the output scale
// per layer (measured: 3.8e-6 on the detector's maps after its 16 such layers, against a tolerance of 1e-4).  The
// filter transform is done once at load time in double.
//
// Tiling: a workgroup (8 waves, one per CU, two waves per SIMD) computes 16 rows x 32 columns of outputs = 4 x 8
// blocks of 4x4, times 64 output channels.  The 36 Winograd components are 36 independent GEMMs (32 blocks x Cin) x
// (Cin x 64); wave (gr, gc, ch) owns the 3x3 components (3 gr + 0..2, 3 gc + 0..2) for 32 blocks x channels
// 32 ch .. 32 ch + 31: 9 accumulators of 32x32 = 144 registers.
//   K: chunks of 8 input channels.  The raw tile with its halo (18 x 34 pixels, 40 bytes per pixel so that the blocks a
//   half-wave reads fall into different banks) is staged in LDS, transformed into U[component][block][channel] in LDS by
//   all threads (lane = (block, channel pair), wave = (half of the tile, one or two component rows): rows of B^T d,
//   then (.) B along them), and every lane loads its transformed-filter fragments straight from a lane-linear slab.
//   U and the raw tile are double buffered, one barrier per chunk; a wave transforms chunk k + 1 and then multiplies
//   chunk k, so while one wave of a SIMD is in its transform the other one has MFMAs to issue.
//   Epilogue, one block row (4 x 32 outputs) at a time: each wave applies its share of the inverse transform (A^T
//   restricted to its rows, A to its columns) and leaves its partial outputs in its own LDS staging area; the output
//   pass adds the four areas, then bias, residual, ReLU and the optional max-pool, and writes 16-byte stores.
//
// Roofline: MFMA bound; the unit of work is the direct convolution's 2*9*Cin*Cout flop per output pixel, of which
// 36/144 are executed.
#include <type_traits>

#include <mutex>

#include "common.h"

namespace ocrhip {

namespace {

typedef float floatx16 __attribute__((ext_vector_type(16)));
typedef float floatx4 __attribute__((ext_vector_type(4)));
typedef float floatx2 __attribute__((ext_vector_type(2)));

constexpr int kCK = 8;                          // input channels per chunk
constexpr int kRS = 10;                         // floats per raw pixel in LDS: 8 channels + 2, so that the 8 blocks a half-wave reads (4 pixels apart) start 8 banks apart
constexpr int kRawW = 34, kRawH = 20, kRawPx = kRawW * kRawH;      // 18 rows of one image, or 2 x 10 of a stacked pair
constexpr int kRFloats = kRawPx * kRS;
constexpr int kRBuf = kRFloats + 4;             // + a spare slot for the threads without a pixel
constexpr int kUC = 32 * kCK;                   // floats per component of U: 32 blocks x 8 channels
constexpr int kUFloats = 36 * kUC;
constexpr int kLoopFloats = 2 * kUFloats + 2 * kRBuf;
constexpr int kOS = 64 + 4;                     // floats per staged output pixel
constexpr int kPartFloats = 128 * kOS;          // one wave group's share of a block row (4 x 32 pixels x 64 channels)
constexpr int kStageFloats = 4 * kPartFloats;
constexpr int kLdsBytes = 4 * (kLoopFloats > kStageFloats ? kLoopFloats : kStageFloats);
// MODE 3 (half tiles): two U of 16 blocks and two raw tiles of 10 rows; or the staged 8 x 32 output tile
constexpr int kLoopFloats3 = 2 * (36 * 16 * kCK) + 2 * (kRawW * 10 * kRS + 4);
constexpr int kStageFloats3 = 8 * 32 * kOS;
constexpr int kLdsBytes3 = 4 * (kLoopFloats3 > kStageFloats3 ? kLoopFloats3 : kStageFloats3);
// MODE 4 (half tiles, 128 output channels, chunks of 16 input channels): twice the loop buffers; 128 + 4 floats per staged pixel
constexpr int kLoopFloats4 = 2 * kLoopFloats3;
constexpr int kStageFloats4 = 8 * 32 * (128 + 4);
constexpr int kLdsBytes4 = 4 * (kLoopFloats4 > kStageFloats4 ? kLoopFloats4 : kStageFloats4);
constexpr int kWChunkFloats = 36 * 2 * 64 * 4;  // transformed filters of one chunk of one 64-channel block
constexpr int kSplitMinCin = 512;               // layers at least this deep run as transform kernel + multiply kernel under `wino4_split`

struct Wino4Args {
    const float* x;        // (B,H,W,cin)
    const float* w;        // transformed filters, slab layout of prepare_conv_wino4
    const float* bias;     // cout_pad
    const float* res;      // optional residual (POOL == 0), shaped like y
    float* y;
    int B, H, W, cin, cout, relu;
    int tiles_x, tiles_per_image, n_blocks_n;
    float* v;              // MODE 1 / 2: transformed tiles [tile][chunk][36][32 blocks][8], laid out like U in LDS
    int chunks_per_group;  // MODE 1: chunks a workgroup transforms
    int pair;              // 1: images of 8 rows, two of them stacked in one tile (rows 0..7 and 8..15 of its outputs)
    int vP, vB;            // MODE 3 / 4, vP > 0: the vB images of the batch lie side by side, vP columns apart (zeros
                           // between), and the tiles cover that one wide image; B is 1 then, W the width of one image
};

// Share of the 3x3 components (3 GR + ii, 3 GC + jj) in the 4x4 outputs of a block: Y = At_r M At_c^T with At_r the
// columns 3 GR .. 3 GR + 2 of A^T (4x6) = [1 1 1 1 1 0; 0 1 -1 2 -2 0; 0 1 1 4 4 0; 0 1 -1 8 -8 1] and At_c likewise for GC.
// The coefficients are compile-time constants: zeros and ones cost nothing.
template <int G>
__device__ __forceinline__ void at_apply(float m0, float m1, float m2, float (&o)[4]) {
    if (G == 0) {            // columns 0..2: [1 1 1; 0 1 -1; 0 1 1; 0 1 -1]
        o[1] = m1 - m2;
        o[2] = m1 + m2;
        o[0] = m0 + o[2];
        o[3] = o[1];
    } else {                 // columns 3..5: [1 1 0; 2 -2 0; 4 4 0; 8 -8 1]
        const float s = m0 + m1, d = m0 - m1;
        o[0] = s;
        o[1] = 2.0f * d;
        o[2] = 4.0f * s;
        o[3] = 8.0f * d + m2;
    }
}
template <int GR, int GC>
__device__ __forceinline__ void inverse_part(const float (&M)[3][3], float (&Y)[4][4]) {
    float Wc[3][4];
#pragma unroll
    for (int ii = 0; ii < 3; ++ii) at_apply<GC>(M[ii][0], M[ii][1], M[ii][2], Wc[ii]);
#pragma unroll
    for (int c = 0; c < 4; ++c) {
        float col[4];
        at_apply<GR>(Wc[0][c], Wc[1][c], Wc[2][c], col);
#pragma unroll
        for (int r = 0; r < 4; ++r) Y[r][c] = col[r];
    }
}

// MODE 0: the fused kernel described above.  MODE 1 and 2 split it for deep layers (>= 512 input channels), where the
// data transform - repeated by each of the Cout / 64 workgroups of a tile and paid for in MFMA issue time - weighs
// more than writing the transformed tile to memory once:
//   MODE 1 transforms only: a workgroup owns a tile and a group of chunks (the `channel block` index of the grid) and
//          writes U[chunk] (the 36 KB LDS image) to the workspace v;
//   MODE 2 multiplies only: the MFMA operands of both sides come straight from memory (v and the filter slab, 16 bytes
//          per lane, reloaded for the next chunk right after their last use): no LDS, no barrier and no vector-ALU work
//          in the chunk loop; LDS serves the epilogue alone.
//   MODE 3 is the fused kernel with a different division of the work between the waves; it serves every fused launch
//          by default (MODE 0 remains for `wino4_wave_tiles = 0` and under MODE 2).  A workgroup of 4 waves computes 8 x 32
//          outputs (2 x 8 blocks of 4x4) x 64 channels; the multiplications run on v_mfma_f32_16x16x4_f32 and wave cq owns
//          ALL 36 components of the 16 blocks x 16 channels (36 accumulators of 4 registers: the same 144).  A lane then
//          holds the complete 6x6 product of its (block, channel) elements: the inverse transform and the max-pool
//          happen in registers, the workgroup stages its (pooled) tile in LDS once and passes ONE barrier before the
//          16-byte stores - against four partial sums per output, staged one block row at a time between eight barriers,
//          in MODE 0.  The price: a block's transformed data is read from LDS by four waves instead of two (36 16-byte
//          operand reads per wave and chunk against 27).  Half the tile of MODE 0 means half the LDS (64 KB in the loop,
//          70 KB for the staged output), so TWO workgroups share a CU, one wave of each per SIMD: while one waits (first
//          loads of a tile, a barrier, the stores of its epilogue) the other one's waves issue.  Maps of 8 rows are one
//          tile per image.  Same arithmetic as MODE 0 up to the order of the additions in the inverse transform.
template <int POOL, int MODE>
__global__ __launch_bounds__(MODE == 3 ? 256 : 512, MODE == 3 ? 2 : 1) void conv_wino4_f32_kernel(const Wino4Args a) {
    extern __shared__ __attribute__((aligned(16))) float lds[];      // U[2] then raw[2]: chunk j lives in buffer j & 1
    constexpr bool W4 = (MODE == 4);                       // wave tiles, 8 waves x 16 = 128 output channels from one transformed tile
    constexpr bool W3 = (MODE == 3 || W4);                 // the variants whose waves own whole (block, channel) tiles
    constexpr int NT = (MODE == 3) ? 256 : 512;            // threads
    constexpr int NBLK = W3 ? 16 : 32;                     // 4x4 blocks of a workgroup's tile
    constexpr int TH = NBLK / 2;                           // output rows of the tile
    constexpr int UC = NBLK * kCK, UF = 36 * UC;           // floats per component of U, per U
    constexpr int RH = W3 ? 10 : kRawH, RPX = kRawW * RH, RF = RPX * kRS, RB = RF + 4;      // raw tile
    constexpr int NSUB = W4 ? 2 : 1;                       // 8-channel sub-chunks of a K chunk: each has its own U and raw tile
    constexpr int CKK = kCK * NSUB;                        // input channels per chunk
    constexpr int NCH = W4 ? 128 : 64;                     // output channels of a workgroup
    constexpr int WCF = W4 ? 4 * kWChunkFloats : kWChunkFloats;      // floats of a chunk of the filter slab

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int li = lane & 31, lh = lane >> 5;
    const int ch = wave & 1, gc = (wave >> 1) & 1, gr = wave >> 2;

    // every XCD (own L2) walks one contiguous range of the work index, the channel block varying fastest
    const unsigned total = gridDim.x, xcd = blockIdx.x & 7u, slot = blockIdx.x >> 3;
    const unsigned work = xcd * (total >> 3) + min(xcd, total & 7u) + slot;
    const int nb = (int)(work % (unsigned)a.n_blocks_n);
    const int tile_all = (int)(work / (unsigned)a.n_blocks_n);
    const int b = tile_all / a.tiles_per_image;
    const int tile = tile_all % a.tiles_per_image;
    const int y0 = (tile / a.tiles_x) * TH, x0 = (tile % a.tiles_x) * 32;      // pair mode: b counts pairs, y0 = 0
    const int nK = a.cin / CKK;
    const int pair = a.pair;

    // bias now: a load still in flight in the epilogue would stall its guarded stores
    const int c4 = tid & (NCH / 4 - 1);
    const int co0 = nb * NCH + c4 * 4;
    floatx4 bias4 = {0.f, 0.f, 0.f, 0.f};
    if constexpr (MODE != 1) bias4 = *reinterpret_cast<const floatx4*>(a.bias + co0);          // bias is padded to cout_pad

    floatx16 acc[W3 ? 1 : 9];
    floatx4 acc3[W3 ? 36 : 1];                  // MODE 3: component f of blocks 4 (lane / 16) + v, channel 16 cq + lane % 16
    if constexpr (!W3) {          // MODE 3 / 4: the first chunk's MFMAs take zero as their addend instead
#pragma unroll
        for (int f = 0; f < 9; ++f)
#pragma unroll
            for (int v = 0; v < 16; ++v) acc[f][v] = 0.0f;
    }

    // Global loads go through buffer descriptors: a per-lane byte offset that never changes, the chunk as a scalar
    // offset, and the range check, which returns zeros beyond the buffer: the zero padding around the image costs no
    // instruction (lanes outside the image carry the offset 2^31).
    const float* wslab = a.w + (int64_t)nb * nK * WCF;
    const __amdgpu_buffer_rsrc_t w_rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)wslab, 0, nK * WCF * 4, 0x00020000);
    const int w_off = ((((3 * gr) * 6 + 3 * gc) * 2 + ch) * 64 + lane) * 16;
    auto load_w = [&](int kc, int f) {          // fragment of component (3 gr + f / 3, 3 gc + f % 3) of chunk kc
        return __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(w_rsrc, w_off, kc * (kWChunkFloats * 4) + ((f / 3) * 6 + f % 3) * 2048, 0));
    };

    // raw tile: 18 x 34 pixels (rows y0 - 1 .. y0 + 16) x 2 channel quads over the threads.  Pair mode: 20 rows = for each
    // of the two images its 8 rows between two rows of zeros, so that no block reads its neighbour image as padding.
    constexpr int NRAW = (RPX * 2 * NSUB + NT - 1) / NT;
    const int64_t img_floats = (int64_t)a.H * a.W * a.cin;
    const int n_img = a.vP > 0 ? a.vB : (pair ? min(2, a.B - 2 * b) : 1);      // a missing second image reads as zeros (range check)
    const __amdgpu_buffer_rsrc_t x_rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)(a.x + (int64_t)b * (pair ? 2 : 1) * img_floats), 0,
                                                                            (int)(n_img * img_floats * 4), 0x00020000);
    float4 ra[NRAW];
    int raw_src[NRAW], raw_dst[NRAW];
#pragma unroll
    for (int i = 0; i < NRAW; ++i) {
        // MODE 4: 2 x 680 elements over 512 threads as (sub-chunk 0, first 512), (sub-chunk 1, first 512), (sub-chunk
        // tid / 256, the remaining 168): the second differs from the first by constants in both addresses
        const int sub = W4 ? (i == 2 ? (tid >> 8) : i) : 0;
        const int f = W4 ? (i == 2 ? NT + (tid & 255) : tid) : tid + i * NT;
        const int fa = f < RPX * 2 ? sub * (RPX * 2) + f : RPX * 2 * NSUB;
        const int px = f >> 1, q = f & 1;
        const int rr = px / kRawW;
        int gx = x0 - 1 + px % kRawW;
        int img = (pair && rr >= 10) ? 1 : 0;
        const int gy = pair ? rr - 10 * img - 1 : y0 - 1 + rr;
        bool in_x = gx >= 0;
        if (W3 && a.vP > 0 && in_x) {          // column of the wide image -> (image, column); the gap reads as zeros
            img = gx / a.vP;
            gx -= img * a.vP;
            in_x = img < a.vB;
        }
        const bool ok = fa < RPX * 2 * NSUB && (pair || W3 || rr < 18) && gy >= 0 && gy < a.H && in_x && gx < a.W;
        raw_src[i] = ok ? ((((img * a.H + gy) * a.W + gx) * a.cin + sub * kCK + q * 4) * 4) : (int)0x80000000;
        raw_dst[i] = (fa < RPX * 2 * NSUB) ? (sub * RB + px * kRS + q * 4) : RF;
    }
    auto load_raw = [&](int kc) {
#pragma unroll
        for (int i = 0; i < NRAW; ++i)
            ra[i] = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(x_rsrc, raw_src[i], kc * (CKK * 4), 0));
    };
    auto store_raw = [&](int rbase) {
#pragma unroll
        for (int i = 0; i < NRAW; ++i) {          // pixels are 40 bytes apart: two 8-byte stores
            *reinterpret_cast<floatx2*>(&lds[rbase + raw_dst[i]]) = floatx2{ra[i].x, ra[i].y};
            *reinterpret_cast<floatx2*>(&lds[rbase + raw_dst[i] + 2]) = floatx2{ra[i].z, ra[i].w};
        }
    };

    // Data transform.  A wave task = (block-row half, component rows): its lanes = (block row within the half, block
    // column, channel pair).  A lane forms its rows of B^T d for its block (6 columns, each a combination of 3 or 4 raw
    // rows) and multiplies by B along each row: components (i, 0..5) of 2 channels,
    //   B^T = [4 0 -5 0 1 0; 0 -4 -4 1 1 0; 0 4 -4 -1 1 0; 0 -2 -1 2 1 0; 0 2 -1 -2 1 0; 0 4 0 -5 0 1].
    // Rows 1 and 2 are (d4 - 4 d2) +- (d3 - 4 d1), rows 3 and 4 are (d4 - d2) +- 2 (d3 - d1): a task that forms both
    // rows of such a pair reads its four raw rows once.  Eight tasks, one per wave: waves 0..3 take the row pairs (1, 2)
    // and (3, 4) of the two halves, waves 4..7 the single rows 0 and 5, so that every SIMD (waves s and s + 4) gets one
    // pair task and one single task.  The roles are uniform in a wave: the code is specialised per role.
    const int t_p = lane & 3, t_bx = (lane >> 2) & 7, t_bylo = lane >> 5;
    auto transform = [&](int rbase, int ubase) {
        // role 0: rows 1, 2; 1: rows 3, 4; 2: row 0; 3: row 5.  MODE 3: one half, the four roles over the four waves;
        // MODE 4: waves 0..3 transform the first 8 channels of the chunk, waves 4..7 the other 8
        const int half = W3 ? 0 : (wave & 1), role = W3 ? (wave & 3) : (wave >> 1);
        if constexpr (W4) {
            rbase += (wave >> 2) * RB;
            ubase += (wave >> 2) * UF;
        }
        const int by = half * 2 + t_bylo;
        const int m = by * 8 + t_bx;
        const float* src = &lds[rbase + ((by * 4 + (pair ? 2 * (by >> 1) : 0)) * kRawW + t_bx * 4) * kRS + t_p * 2];
        constexpr int RW = kRawW * kRS;
        auto rd = [&](int r, int c) { return *reinterpret_cast<const floatx2*>(src + r * RW + c * kRS); };
        // (.) B along a row t[0..5] of B^T d, written as components (i, 0..5)
        auto row_out = [&](const floatx2 (&t)[6], int i) {
            const floatx2 e1 = t[4] - 4.0f * t[2], e2 = t[3] - 4.0f * t[1], e3 = t[4] - t[2], e4 = t[3] - t[1];
            floatx2 o[6];
            o[0] = 4.0f * t[0] + (t[4] - 5.0f * t[2]);
            o[1] = e1 + e2;
            o[2] = e1 - e2;
            o[3] = e3 + 2.0f * e4;
            o[4] = e3 - 2.0f * e4;
            o[5] = 4.0f * t[1] + (t[5] - 5.0f * t[3]);
            if constexpr (W3) {
                // MODE 3: components in pairs, U[f / 2][block][channel pair][f % 2][2 channels]: a lane's operands for two
                // components are one 16-byte read, and a wave's reads tile 1 KB without gaps
                float* dst3 = &lds[ubase + (i * 3) * (2 * UC) + m * (2 * kCK) + t_p * 4];
#pragma unroll
                for (int j = 0; j < 6; ++j) *reinterpret_cast<floatx2*>(dst3 + (j >> 1) * (2 * UC) + (j & 1) * 2) = o[j];
            } else {
                // U[component][block m][8 channels]; the two 16-byte halves of a block swap places in blocks 8..15 and
                // 24..31, which makes the MFMA operand reads (16 lanes = 16 blocks, one half) conflict free without padding
                float* dst = &lds[ubase + (i * 6) * kUC + m * kCK + (((t_p >> 1) ^ ((m >> 3) & 1)) * 4) + (t_p & 1) * 2];
#pragma unroll
                for (int j = 0; j < 6; ++j) *reinterpret_cast<floatx2*>(dst + j * kUC) = o[j];
            }
        };
        if (role >= 2) {                               // 4 d_a - 5 d_b + d_c with rows (0, 2, 4) or (1, 3, 5)
            const int ra_ = role == 2 ? 0 : RW;
            floatx2 t[6];
#pragma unroll
            for (int c = 0; c < 6; ++c) {
                const floatx2 da = *reinterpret_cast<const floatx2*>(src + ra_ + c * kRS);
                const floatx2 db = *reinterpret_cast<const floatx2*>(src + ra_ + 2 * RW + c * kRS);
                const floatx2 dc = *reinterpret_cast<const floatx2*>(src + ra_ + 4 * RW + c * kRS);
                t[c] = 4.0f * da + (dc - 5.0f * db);
            }
            row_out(t, role == 2 ? 0 : 5);
        } else {                                       // (d4 - q d2) +- p (d3 - q d1): q = 4, p = 1 (rows 1, 2); q = 1, p = 2 (rows 3, 4)
            const float q = role == 0 ? 4.0f : 1.0f, pp = role == 0 ? 1.0f : 2.0f;
            floatx2 ta[6], tb[6];
#pragma unroll
            for (int c = 0; c < 6; ++c) {
                const floatx2 e1 = rd(4, c) - q * rd(2, c);
                const floatx2 e2 = rd(3, c) - q * rd(1, c);
                ta[c] = e1 + pp * e2;
                tb[c] = e1 - pp * e2;
            }
            row_out(ta, 1 + 2 * role);
            row_out(tb, 2 + 2 * role);
        }
    };

    float4 bw[9], af[MODE == 2 ? 9 : 3];
    const __amdgpu_buffer_rsrc_t v_rsrc = __builtin_amdgcn_make_buffer_rsrc(
        (void*)(a.v + (int64_t)tile_all * nK * kUFloats), 0, MODE == 0 ? 0 : nK * kUFloats * 4, 0x00020000);
    const int u_src = ((3 * gr) * 6 + 3 * gc) * kUC + li * kCK + ((lh ^ ((li >> 3) & 1)) * 4);
    auto af_read = [&](int ubase, int g) {       // component row 3 gr + g, columns 3 gc + 0..2
#pragma unroll
        for (int jj = 0; jj < 3; ++jj) af[jj] = *reinterpret_cast<const float4*>(&lds[ubase + u_src + (g * 6 + jj) * kUC]);
    };

    constexpr int kR0 = 2 * NSUB * UF, kR1 = kR0 + NSUB * RB;
    if constexpr (MODE == 1) {
        // transform only: chunks [nb * chunks_per_group, ...) of this tile -> v
        const int k_lo = nb * a.chunks_per_group, k_hi = min(nK, k_lo + a.chunks_per_group);
        load_raw(k_lo);
        for (int kc = k_lo; kc < k_hi; ++kc) {
            store_raw(kR0);
            if (kc + 1 < k_hi) load_raw(kc + 1);
            __syncthreads();
            transform(kR0, 0);
            __syncthreads();
            float4* dstv = reinterpret_cast<float4*>(a.v + ((int64_t)tile_all * nK + kc) * kUFloats);
            for (int f = tid; f < kUFloats / 4; f += 512) dstv[f] = *reinterpret_cast<const float4*>(&lds[f * 4]);
        }
        return;
    }
    auto load_v = [&](int kc, int f) {          // A fragment of component (3 gr + f / 3, 3 gc + f % 3) of chunk kc from v
        return __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(v_rsrc, u_src * 4, kc * (kUFloats * 4) + ((f / 3) * 6 + f % 3) * (kUC * 4), 0));
    };
    if constexpr (MODE == 2) {
#pragma unroll
        for (int f = 0; f < 9; ++f) {
            af[f] = load_v(0, f);
            bw[f] = load_w(0, f);
        }
        for (int kc = 0; kc < nK; ++kc) {
            const int kn = min(kc + 1, nK - 1);
#pragma unroll
            for (int g = 0; g < 3; ++g) {
#pragma unroll
                for (int t = 0; t < 4; ++t)
#pragma unroll
                    for (int jj = 0; jj < 3; ++jj) {
                        const float4 a4 = af[g * 3 + jj], b4 = bw[g * 3 + jj];
                        const float av = t == 0 ? a4.x : (t == 1 ? a4.y : (t == 2 ? a4.z : a4.w));
                        const float bv = t == 0 ? b4.x : (t == 1 ? b4.y : (t == 2 ? b4.z : b4.w));
                        acc[g * 3 + jj] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bv, acc[g * 3 + jj], 0, 0, 0);
                    }
#pragma unroll
                for (int jj = 0; jj < 3; ++jj) {
                    af[g * 3 + jj] = load_v(kn, g * 3 + jj);
                    bw[g * 3 + jj] = load_w(kn, g * 3 + jj);
                }
                __builtin_amdgcn_sched_barrier(0);
            }
        }
    }
    // MODE 0 prologue: chunk 0 transformed into U[0], raw tile 1 in raw[1], raw tile 2 on its way, filter fragments of chunk 0
    if constexpr (MODE == 0) {
    {
        float4 rb[NRAW];                         // raw tiles 0 and 1 requested together: one memory round trip
        load_raw(0);
#pragma unroll
        for (int i = 0; i < NRAW; ++i)
            rb[i] = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(x_rsrc, raw_src[i], kCK * 4, 0));
#pragma unroll
        for (int f = 0; f < 9; ++f) bw[f] = load_w(0, f);
        store_raw(kR0);
#pragma unroll
        for (int i = 0; i < NRAW; ++i) {
            *reinterpret_cast<floatx2*>(&lds[kR1 + raw_dst[i]]) = floatx2{rb[i].x, rb[i].y};
            *reinterpret_cast<floatx2*>(&lds[kR1 + raw_dst[i] + 2]) = floatx2{rb[i].z, rb[i].w};
        }
    }
    load_raw(min(2, nK - 1));
    __syncthreads();
    transform(kR0, 0);
    __syncthreads();

    // Chunk kc.  On entry: U[kc & 1] complete, the chunk's filter fragments in registers, raw tile kc + 1 in
    // raw[(kc + 1) & 1], raw tile kc + 2 in flight (past the end of K: the last chunk again, unused).
    auto chunk = [&](int kc, auto parity) {
        constexpr int P = decltype(parity)::value;
        constexpr int ucur = P * kUFloats, unxt = (1 - P) * kUFloats;
        constexpr int rnxt = kR0 + (1 - P) * kRBuf, rfill = kR0 + P * kRBuf;
        const int kn = min(kc + 1, nK - 1);
        if (kc + 1 < nK) {                          // uniform
            transform(rnxt, unxt);
            store_raw(rfill);
            load_raw(min(kc + 3, nK - 1));
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int g = 0; g < 3; ++g) {
            af_read(ucur, g);
#pragma unroll
            for (int t = 0; t < 4; ++t)
#pragma unroll
                for (int jj = 0; jj < 3; ++jj) {
                    const float4 a4 = af[jj], b4 = bw[g * 3 + jj];
                    const float av = t == 0 ? a4.x : (t == 1 ? a4.y : (t == 2 ? a4.z : a4.w));
                    const float bv = t == 0 ? b4.x : (t == 1 ? b4.y : (t == 2 ? b4.z : b4.w));
                    acc[g * 3 + jj] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bv, acc[g * 3 + jj], 0, 0, 0);
                }
#pragma unroll
            for (int jj = 0; jj < 3; ++jj) bw[g * 3 + jj] = load_w(kn, g * 3 + jj);
            __builtin_amdgcn_sched_barrier(0);
        }
        __syncthreads();
    };
    for (int kc = 0; kc < nK; kc += 2) {          // nK is even
        chunk(kc, std::integral_constant<int, 0>{});
        chunk(kc + 1, std::integral_constant<int, 1>{});
    }
    }       // MODE == 0

    // ---- MODE 3: fused, every wave owns all 36 components of 16 blocks x 16 channels (see the kernel's header)
    if constexpr (W3) {
        const int cq = wave;                              // quarter of the 64 channels
        const int l16 = lane & 15, q = lane >> 4;
        // Filter slab of MODE 3: [chunk][f2 = 0..17][cq][lane][4 floats] = components 2 f2, 2 f2 + 1 x input channels
        // 2 q, 2 q + 1 of the chunk, for output channel 16 cq + lane % 16: one 16-byte load feeds four MFMAs.
        // MODE 4: [chunk of 16][sub-chunk][f2][cq = 0..7][lane][4 floats]: step = 18 sub + f2
        const int w3_off = (cq * 64 + lane) * 16;
        auto load_w3 = [&](int kc, int step) {
            return __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(w_rsrc, w3_off, kc * (WCF * 4) + step * (NCH * 64), 0));
        };
        constexpr int RING = W4 ? 6 : 9;                  // 18 steps per sub-chunk are a multiple of both
        float4 bw3[RING];                                 // fragment ring: step s (two components) sits in slot s % RING, reloaded RING steps ahead
        {
            float4 rb[NRAW];                             // raw tiles 0 and 1 requested together: one memory round trip
            load_raw(0);
#pragma unroll
            for (int i = 0; i < NRAW; ++i)
                rb[i] = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(x_rsrc, raw_src[i], CKK * 4, 0));
#pragma unroll
            for (int f = 0; f < RING; ++f) bw3[f] = load_w3(0, f);
            store_raw(kR0);
#pragma unroll
            for (int i = 0; i < NRAW; ++i) {
            *reinterpret_cast<floatx2*>(&lds[kR1 + raw_dst[i]]) = floatx2{rb[i].x, rb[i].y};
            *reinterpret_cast<floatx2*>(&lds[kR1 + raw_dst[i] + 2]) = floatx2{rb[i].z, rb[i].w};
        }
        }
        load_raw(min(2, nK - 1));
        __syncthreads();
        transform(kR0, 0);
        __syncthreads();

        // operands of a lane within a component pair of U: block lane % 16, channels 2 q and 2 q + 1 of both
        // components (k of the two MFMAs that consume the chunk's 8 channels: k = lane / 16, channel = 2 k + step)
        const int u3 = l16 * (2 * kCK) + 4 * q;
        auto chunk3 = [&](int kc, auto parity, auto first_chunk) {
            constexpr int P = decltype(parity)::value;
            constexpr bool FIRST = decltype(first_chunk)::value;       // nothing accumulated yet: 144 register clears saved
            const floatx4 zero4 = {0.f, 0.f, 0.f, 0.f};
            constexpr int ucur = P * NSUB * UF, unxt = (1 - P) * NSUB * UF;
            constexpr int rnxt = kR0 + (1 - P) * NSUB * RB, rfill = kR0 + P * NSUB * RB;
            constexpr int NG = 9 * NSUB;                // groups of 4 components; the sub-chunks' U lie back to back
            const int kn = min(kc + 1, nK - 1);
            // MODE 4: the two waves of a SIMD (w and w + 4) belong to one workgroup and meet at every barrier; the second
            // one transforms AFTER its multiplications, so that one of them has MFMAs to issue while the other transforms
            const bool late = W4 && wave >= 4;          // uniform
            // The loads of the raw tile go out BEFORE the transform: loads return in order, so the first filter fragment
            // requested after them waits for them too, and it is needed RING steps into the multiplications.
            if (kc + 1 < nK && !late) {                 // uniform
                store_raw(rfill);
                load_raw(min(kc + 3, nK - 1));
                __builtin_amdgcn_sched_barrier(0);
                transform(rnxt, unxt);
            }
            __builtin_amdgcn_sched_barrier(0);
            const float* ub = &lds[ucur + u3];
            float4 a4[2][2];                            // operands of a group of 4 components, read one group ahead
#pragma unroll
            for (int c = 0; c < 2; ++c) a4[0][c] = *reinterpret_cast<const float4*>(ub + c * (2 * UC));
#pragma unroll
            for (int gg = 0; gg < NG; ++gg) {
                const int g = gg % 9;
                if (gg + 1 < NG) {
#pragma unroll
                    for (int c = 0; c < 2; ++c) a4[(gg + 1) & 1][c] = *reinterpret_cast<const float4*>(ub + (2 * (gg + 1) + c) * (2 * UC));
                }
                __builtin_amdgcn_sched_barrier(0);
                const floatx2 av[4] = {floatx2{a4[gg & 1][0].x, a4[gg & 1][0].y}, floatx2{a4[gg & 1][0].z, a4[gg & 1][0].w},
                                       floatx2{a4[gg & 1][1].x, a4[gg & 1][1].y}, floatx2{a4[gg & 1][1].z, a4[gg & 1][1].w}};
                const float4 b0 = bw3[(2 * g) % RING], b1 = bw3[(2 * g + 1) % RING];
                // the two MFMAs of a component (its 8 channels) sit four instructions apart
                const bool fresh = FIRST && gg < 9;
                acc3[4 * g + 0] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[0].x, b0.x, fresh ? zero4 : acc3[4 * g + 0], 0, 0, 0);
                acc3[4 * g + 1] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[1].x, b0.z, fresh ? zero4 : acc3[4 * g + 1], 0, 0, 0);
                acc3[4 * g + 2] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[2].x, b1.x, fresh ? zero4 : acc3[4 * g + 2], 0, 0, 0);
                acc3[4 * g + 3] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[3].x, b1.z, fresh ? zero4 : acc3[4 * g + 3], 0, 0, 0);
                acc3[4 * g + 0] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[0].y, b0.y, acc3[4 * g + 0], 0, 0, 0);
                acc3[4 * g + 1] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[1].y, b0.w, acc3[4 * g + 1], 0, 0, 0);
                acc3[4 * g + 2] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[2].y, b1.y, acc3[4 * g + 2], 0, 0, 0);
                acc3[4 * g + 3] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[3].y, b1.w, acc3[4 * g + 3], 0, 0, 0);
                // the two slots just used: refilled with the fragments RING steps on (this chunk's second half, or the
                // next chunk's first half)
#pragma unroll
                for (int d = 0; d < 2; ++d) {
                    const int st = 2 * gg + d + RING;
                    bw3[(2 * g + d) % RING] = (st < 2 * NG) ? load_w3(kc, st) : load_w3(kn, st - 2 * NG);
                }
                __builtin_amdgcn_sched_barrier(0);
            }
            if constexpr (W4) {
                if (kc + 1 < nK && late) {
                    store_raw(rfill);
                    load_raw(min(kc + 3, nK - 1));
                    __builtin_amdgcn_sched_barrier(0);
                    transform(rnxt, unxt);
                }
            }
            __syncthreads();
        };
        chunk3(0, std::integral_constant<int, 0>{}, std::true_type{});
        chunk3(1, std::integral_constant<int, 1>{}, std::false_type{});
        for (int kc = 2; kc < nK; kc += 2) {          // nK is even
            chunk3(kc, std::integral_constant<int, 0>{}, std::false_type{});
            chunk3(kc + 1, std::integral_constant<int, 1>{}, std::false_type{});
        }

        // ---- epilogue: Y = A^T M A per (block, channel) in registers, max-pool in registers, the (pooled) tile staged in
        // LDS once (every wave has passed the loop's last barrier: the loop's buffers are free), one barrier, 16-byte stores.
        // A^T = [1 1 1 1 1 0; 0 1 -1 2 -2 0; 0 1 1 4 4 0; 0 1 -1 8 -8 1].
        constexpr int PW = (POOL == 1) ? 2 : 4, PH = (POOL == 0) ? 4 : 2;      // outputs a block keeps after pooling
        constexpr int kTW = 8 * PW, kTH = (NBLK / 8) * PH;                     // the tile after pooling
        constexpr int OS = NCH + 4;                                            // floats per staged pixel
        // two of a lane's four blocks at a time (registers v and v + 1 of every accumulator: neighbours in a block row),
        // so that the additions are packed ones
        auto at4 = [](floatx2 m0, floatx2 m1, floatx2 m2, floatx2 m3, floatx2 m4, floatx2 m5, floatx2 (&o)[4]) {
            const floatx2 s1 = m1 + m2, d1 = m1 - m2, s2 = m3 + m4, d2 = m3 - m4;
            o[0] = (m0 + s1) + s2;
            o[1] = d1 + 2.0f * d2;
            o[2] = s1 + 4.0f * s2;
            o[3] = (d1 + 8.0f * d2) + m5;
        };
        const int ch3 = cq * 16 + l16;
#pragma unroll
        for (int vp = 0; vp < 2; ++vp) {
            auto M = [&](int f) { return floatx2{acc3[f][2 * vp], acc3[f][2 * vp + 1]}; };
            floatx2 Wt[6][4];
#pragma unroll
            for (int i = 0; i < 6; ++i) at4(M(i * 6 + 0), M(i * 6 + 1), M(i * 6 + 2), M(i * 6 + 3), M(i * 6 + 4), M(i * 6 + 5), Wt[i]);
            floatx2 Y[4][4];
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                floatx2 col[4];
                at4(Wt[0][c], Wt[1][c], Wt[2][c], Wt[3][c], Wt[4][c], Wt[5][c], col);
#pragma unroll
                for (int r = 0; r < 4; ++r) Y[r][c] = col[r];
            }
#pragma unroll
            for (int e = 0; e < 2; ++e) {
                const int m = 4 * q + 2 * vp + e, by = m >> 3, bx = m & 7;
                // the four lane groups of a wave (q) write pixels a multiple of 64 floats apart: each rotates its channels
                // by a different quarter of 64 so that they use different LDS banks
                const int pos = (ch3 + 16 * ((bx >> 2) & 1) + 32 * (by & 1)) & (NCH - 1);
                float* O = &lds[((by * PH) * kTW + bx * PW) * OS + pos];
#pragma unroll
                for (int r = 0; r < PH; ++r)
#pragma unroll
                    for (int c = 0; c < PW; ++c) {
                        float o;
                        if (POOL == 0) o = Y[r][c][e];
                        else if (POOL == 1) o = fmaxf(fmaxf(Y[2 * r][2 * c][e], Y[2 * r][2 * c + 1][e]), fmaxf(Y[2 * r + 1][2 * c][e], Y[2 * r + 1][2 * c + 1][e]));
                        else o = fmaxf(Y[2 * r][c][e], Y[2 * r + 1][c][e]);
                        O[(r * kTW + c) * OS] = o;
                    }
            }
        }
        __syncthreads();
        const int Ho3 = (POOL == 0) ? a.H : a.H / 2, Wo3 = (POOL == 1) ? a.W / 2 : a.W;
        const int ox0 = (POOL == 1) ? (x0 >> 1) : x0;
        if ((a.cout & 3) == 0) {                     // uniform
            // Every store of a lane is (its pixel of the first 16 or 32 staged ones) + (a step that only depends on the
            // iteration): the lane's byte offset is ONE register for the whole pass, the step goes into the scalar offset
            // of a buffer store, and pixels or channels outside the map get the out-of-range offset 2^31 instead of a
            // branch (the range check looks at the vector offset only).  No 64-bit index arithmetic per store.
            constexpr int PXI = NT / (NCH / 4);                                   // pixels per iteration
            constexpr int NIT = kTW * kTH / PXI;
            static_assert(PXI == 16 && kTW % PXI == 0, "a pass covers 16 pixels of one row");
            const int lp = tid / (NCH / 4), lx = lp;                              // the lane's pixel within a pass
            const int Y0 = (POOL == 0) ? y0 : (y0 >> 1);
            const bool wide = a.vP > 0;                                           // uniform
            // the tile's first pixel, the block's first channel (side by side images: the batch's first pixel)
            const int64_t first = (wide ? 0 : (((int64_t)b * Ho3 + Y0) * Wo3 + ox0) * a.cout) + nb * NCH;
            const __amdgpu_buffer_rsrc_t y_rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)(a.y + first), 0, 0x7fffffff, 0x00020000);
            const __amdgpu_buffer_rsrc_t r_rsrc = __builtin_amdgcn_make_buffer_rsrc(
                (void*)((POOL == 0 && a.res) ? a.res + first : a.y), 0, (POOL == 0 && a.res) ? 0x7fffffff : 0, 0x00020000);
            const int pix_bytes = a.cout * 4, row_bytes = Wo3 * pix_bytes;
            const bool ch_ok = co0 + 3 < a.cout;
            // A lane's column takes one of kTW / 16 values over the passes.  Side by side images: a column of the wide
            // (pooled) image is (image, column); columns of the gap are dropped like those beyond the map.
            int col_off[kTW / PXI];
            bool col_ok[kTW / PXI];
#pragma unroll
            for (int j = 0; j < kTW / PXI; ++j) {
                const int X = ox0 + j * PXI + lx;
                if (wide) {
                    const int po = (POOL == 1) ? (a.vP >> 1) : a.vP;
                    const int gb = X / po, col = X - gb * po;
                    col_ok[j] = ch_ok && gb < a.vB && col < Wo3;
                    col_off[j] = ((gb * Ho3 + Y0) * Wo3 + col) * pix_bytes + c4 * 16;          // the launcher keeps the batch's output below 2 GB
                } else {
                    col_ok[j] = ch_ok && X < Wo3;
                    col_off[j] = (j * PXI + lx) * pix_bytes + c4 * 16;
                }
            }
#pragma unroll
            for (int it = 0; it < NIT; ++it) {
                const int py = (it * PXI) / kTW, px_it = (it * PXI) % kTW;        // constants after unrolling
                const int pxx = px_it + lx;
                const int by = py / PH, bx = pxx / PW;
                const bool ok = col_ok[px_it / PXI] && Y0 + py < Ho3;
                const int voff = ok ? col_off[px_it / PXI] : (int)0x80000000;
                const int soff = py * row_bytes;
                const int rot = 16 * ((bx >> 2) & 1) + 32 * (by & 1);
                floatx4 val = *reinterpret_cast<const floatx4*>(&lds[(lp + it * PXI) * OS + ((c4 * 4 + rot) & (NCH - 1))]) + bias4;
                if (POOL == 0 && a.res) val += __builtin_bit_cast(floatx4, __builtin_amdgcn_raw_buffer_load_b128(r_rsrc, voff, soff, 0));
                if (a.relu) {
                    val.x = fmaxf(val.x, 0.0f); val.y = fmaxf(val.y, 0.0f); val.z = fmaxf(val.z, 0.0f); val.w = fmaxf(val.w, 0.0f);
                }
                buffer_store_b128_guarded<kStreamingStore>(val, y_rsrc, voff, soff);
            }
            return;
        }
        // a channel count that is not a multiple of 4: element by element
#pragma unroll 4
        for (int it = 0; it < kTW * kTH / (NT / (NCH / 4)); ++it) {
            const int px = tid / (NCH / 4) + it * (NT / (NCH / 4));
            const int py = px / kTW, pxx = px % kTW;
            const int by = py / PH, bx = pxx / PW;
            const int yy = ((POOL == 0) ? y0 : (y0 >> 1)) + py, xx = ox0 + pxx;
            if (yy >= Ho3 || xx >= Wo3) continue;
            const int rot = 16 * ((bx >> 2) & 1) + 32 * (by & 1);
            const floatx4 val = *reinterpret_cast<const floatx4*>(&lds[px * OS + ((c4 * 4 + rot) & (NCH - 1))]) + bias4;
            const int64_t idx = (((int64_t)b * Ho3 + yy) * Wo3 + xx) * a.cout + co0;
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
        return;
    }

    if constexpr (!W3) {
    // ---- inverse transform and output, one block row (4 x 32 outputs) at a time.  Register v of a lane belongs to block
    // m = (v & 3) + 8 * (v >> 2) + 4 * (lane >> 5), i.e. block row v >> 2, and to output channel lane & 31 of the wave's 32.
    // The wave's 3x3 components contribute
    //     Y[r][c] += sum_ii sum_jj A^T[r][3 gr + ii] M[ii][jj] A^T[c][3 gc + jj]
    // to the 4x4 outputs of a block: every wave writes its share of the block row to its own staging area, and the
    // output pass adds the four areas (in a fixed order) with bias, residual, ReLU and max-pool: 16-byte stores.
    const int part = gr * 2 + gc;
    const int oc = ch * 32 + li;
    const bool vec_ok = (a.cout & 3) == 0 && co0 + 3 < a.cout;
    auto max4 = [](floatx4 p, floatx4 q) { return floatx4{fmaxf(p.x, q.x), fmaxf(p.y, q.y), fmaxf(p.z, q.z), fmaxf(p.w, q.w)}; };
    auto staged = [&](int row, int col) {
        const float* q = &lds[(row * 32 + col) * kOS + c4 * 4];
        return (*reinterpret_cast<const floatx4*>(q) + *reinterpret_cast<const floatx4*>(q + kPartFloats)) +
               (*reinterpret_cast<const floatx4*>(q + 2 * kPartFloats) + *reinterpret_cast<const floatx4*>(q + 3 * kPartFloats));
    };
    const int Ho = (POOL == 0) ? a.H : a.H / 2, Wo = (POOL == 1) ? a.W / 2 : a.W;
    constexpr int kOutW = (POOL == 1) ? 16 : 32, kOutH = (POOL == 0) ? 4 : 2;       // output pixels of a block row
    constexpr int kIters = kOutW * kOutH / 32;
#pragma unroll
    for (int by = 0; by < 4; ++by) {
        if (by > 0) __syncthreads();             // the output pass of the previous block row has read the staging areas
        float* O = &lds[part * kPartFloats + oc];
        // one specialised copy of the block row's arithmetic per (gr, gc): the accumulators are used where they are
        // (a switch per block instead made the compiler copy them into common registers first)
        auto block_row = [&](auto GR, auto GC) {
#pragma unroll
            for (int vv = 0; vv < 4; ++vv) {
                const int v = by * 4 + vv;
                const int bx = vv + 4 * lh;
                float M[3][3];
#pragma unroll
                for (int f = 0; f < 9; ++f) M[f / 3][f % 3] = acc[f][v];
                float Y[4][4];
                inverse_part<decltype(GR)::value, decltype(GC)::value>(M, Y);
#pragma unroll
                for (int r = 0; r < 4; ++r)
#pragma unroll
                    for (int c = 0; c < 4; ++c) O[(r * 32 + bx * 4 + c) * kOS] = Y[r][c];
            }
        };
        using I0 = std::integral_constant<int, 0>;
        using I1 = std::integral_constant<int, 1>;
        switch (part) {
            case 0: block_row(I0{}, I0{}); break;
            case 1: block_row(I0{}, I1{}); break;
            case 2: block_row(I1{}, I0{}); break;
            default: block_row(I1{}, I1{}); break;
        }
        __syncthreads();
        floatx4 vals[kIters];
#pragma unroll
        for (int it = 0; it < kIters; ++it) {
            const int px = (tid >> 4) + it * 32;
            const int py = px / kOutW, pxx = px % kOutW;
            if (POOL == 0) vals[it] = staged(py, pxx);
            else if (POOL == 1) vals[it] = max4(max4(staged(2 * py, 2 * pxx), staged(2 * py, 2 * pxx + 1)), max4(staged(2 * py + 1, 2 * pxx), staged(2 * py + 1, 2 * pxx + 1)));
            else vals[it] = max4(staged(2 * py, pxx), staged(2 * py + 1, pxx));
        }
        const int ob = pair ? 2 * b + (by >> 1) : b;                   // image of this block row
        if (ob >= a.B) break;                                          // uniform: the pair's second image does not exist
        const int oy0 = pair ? (by & 1) * kOutH : ((POOL == 0) ? y0 : (y0 >> 1)) + by * kOutH, ox0 = (POOL == 1) ? (x0 >> 1) : x0;
#pragma unroll
        for (int it = 0; it < kIters; ++it) {
            const int px = (tid >> 4) + it * 32;
            const int yy = oy0 + px / kOutW, xx = ox0 + px % kOutW;
            if (yy >= Ho || xx >= Wo) continue;
            floatx4 val = vals[it] + bias4;
            const int64_t idx = (((int64_t)ob * Ho + yy) * Wo + xx) * a.cout + co0;
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
    }       // !W3
}

}  // namespace

// Transformed-filter slab: [n block (64 channels)][k chunk (8)][component f = 6 i + j][channel half c][h][j32][t]
// float32 with output channel = nblock * 64 + c * 32 + j32 and input channel = kchunk * 8 + 4 h + t: lane (h * 32 + j32)
// of a wave that owns channel half c reads 16 consecutive bytes per component.
int prepare_conv_wino4(ConvLayer& L, const float* w_oihw, int cout, int cin) {
    OCR_REQUIRE(cin % (2 * kCK) == 0 && cout > 0, "conv_wino4: cin=%d must be a multiple of %d", cin, 2 * kCK);
    const int nN = (cout + 63) / 64, nK = cin / kCK;
    std::vector<float> w((size_t)nN * nK * kWChunkFloats, 0.0f);
    static const double G[6][3] = {{1.0 / 4, 0, 0},           {-1.0 / 6, -1.0 / 6, -1.0 / 6}, {-1.0 / 6, 1.0 / 6, -1.0 / 6},
                                   {1.0 / 24, 1.0 / 12, 1.0 / 6}, {1.0 / 24, -1.0 / 12, 1.0 / 6}, {0, 0, 1}};
    for (int o = 0; o < cout; ++o)
        for (int c = 0; c < cin; ++c) {
            const float* g = w_oihw + ((size_t)o * cin + c) * 9;
            double tmp[6][3], V[6][6];
            for (int i = 0; i < 6; ++i)
                for (int k = 0; k < 3; ++k) tmp[i][k] = G[i][0] * g[0 * 3 + k] + G[i][1] * g[1 * 3 + k] + G[i][2] * g[2 * 3 + k];
            for (int i = 0; i < 6; ++i)
                for (int j = 0; j < 6; ++j) V[i][j] = tmp[i][0] * G[j][0] + tmp[i][1] * G[j][1] + tmp[i][2] * G[j][2];
            const int nbk = o / 64, chalf = (o % 64) / 32, j32 = o % 32;
            const int kc = c / kCK, hh = (c % kCK) / 4, t = c % 4;
            for (int f = 0; f < 36; ++f) {
                const size_t dst = ((((((size_t)(nbk * nK + kc) * 36 + f) * 2 + chalf) * 2 + hh) * 32 + j32) * 4) + t;
                w[dst] = (float)V[f / 6][f % 6];
            }
        }
    OCR_TRY(L.w_wino4.reserve(w.size() * sizeof(float)));
    OCR_HIP_CHECK(hipMemcpy(L.w_wino4.ptr, w.data(), w.size() * sizeof(float), hipMemcpyHostToDevice));
    {
        // the same transformed filters in the layout of MODE 3 (the fused layers):
        // [n block][k chunk][f2 = f / 2][cq][lane = 16 q + j][t = 2 (f % 2) + step]: output channel 64 nblock + 16 cq + j,
        // input channel 8 kchunk + 2 q + step
        std::vector<float> w3((size_t)nN * nK * kWChunkFloats, 0.0f);
        for (int o = 0; o < cout; ++o)
            for (int c = 0; c < cin; ++c) {
                const int nbk = o / 64, cq = (o % 64) / 16, j = o % 16;
                const int kc = c / kCK, q = (c % kCK) / 2, step = c % 2;
                const int chalf = (o % 64) / 32, j32 = o % 32, hh = (c % kCK) / 4, t = c % 4;
                for (int f = 0; f < 36; ++f) {
                    const size_t src = ((((((size_t)(nbk * nK + kc) * 36 + f) * 2 + chalf) * 2 + hh) * 32 + j32) * 4) + t;
                    const size_t dst = (((((size_t)(nbk * nK + kc) * 18 + f / 2) * 4 + cq) * 64 + (16 * q + j)) * 4) + 2 * (f % 2) + step;
                    w3[dst] = w[src];
                }
            }
        OCR_TRY(L.w_wino4b.reserve(w3.size() * sizeof(float)));
        OCR_HIP_CHECK(hipMemcpy(L.w_wino4b.ptr, w3.data(), w3.size() * sizeof(float), hipMemcpyHostToDevice));
        if (cout % 128 == 0 && cin % 32 == 0) {          // an even number of 16-channel chunks
            // and in the layout of MODE 4 (workgroups of 128 output channels, chunks of 16 input channels):
            // [n block of 128][k chunk of 16][sub-chunk][f2][cq = 0..7][lane = 16 q + j][t = 2 (f % 2) + step]:
            // output channel 128 nblock + 16 cq + j, input channel 16 kchunk + 8 sub + 2 q + step
            std::vector<float> w4(w3.size(), 0.0f);
            const int nK16 = cin / 16;
            for (int o = 0; o < cout; ++o)
                for (int c = 0; c < cin; ++c) {
                    const int cq3 = (o % 64) / 16, j = o % 16, q = (c % kCK) / 2, step = c % 2;
                    const int nb4 = o / 128, cq4 = (o % 128) / 16, kc4 = c / 16, sub = (c % 16) / 8;
                    for (int f = 0; f < 36; ++f) {
                        const size_t src = (((((size_t)((o / 64) * nK + c / kCK) * 18 + f / 2) * 4 + cq3) * 64 + (16 * q + j)) * 4) + 2 * (f % 2) + step;
                        const size_t dst = ((((((size_t)(nb4 * nK16 + kc4) * 2 + sub) * 18 + f / 2) * 8 + cq4) * 64 + (16 * q + j)) * 4) + 2 * (f % 2) + step;
                        w4[dst] = w3[src];
                    }
                }
            OCR_TRY(L.w_wino4c.reserve(w4.size() * sizeof(float)));
            OCR_HIP_CHECK(hipMemcpy(L.w_wino4c.ptr, w4.data(), w4.size() * sizeof(float), hipMemcpyHostToDevice));
        }
    }
    return OCRHIP_OK;
}

// Side by side: a batch of narrow images whose width leaves the 32-column tiles partly empty is tiled as ONE wide
// image, the images `period` columns apart with zeros between them (one column; two where columns are pooled in
// pairs, so that every image starts on an even column).  The gap is the zero padding of both neighbours and the tiles
// of the wide image are full.  An output then also sees the rounding noise of the neighbour image's pixels that
// share its 6x6 patch (their exact contribution is zero): results agree with the image-by-image launch to the
// algorithm's own error, not bit for bit.  Wave-tile modes only; the batch's input and output must fit the 2 GB of a
// buffer descriptor.  Returns the period, 0 when the launch stays image by image.
static int wino4_side_by_side(const ConvLayer& L, int B, int H, int W, int pool) {
    if (!wino4_side_by_side_enabled() || !L.w_wino4b.ptr || !wino4_wave_tiles() || (wino4_split() && L.cin >= kSplitMinCin)) return 0;
    if (B < 2 || W % 32 == 0 || (L.cout & 3) != 0 || (pool == 1 && (W & 1))) return 0;
    const int period = W + (pool == 1 ? 2 : 1);
    const int64_t vw = (int64_t)B * period;
    const int64_t in_bytes = (int64_t)B * H * W * L.cin * 4, out_bytes = (int64_t)B * H * W * L.cout * 4;      // (unpooled bound)
    if (vw >= (1ll << 30) || in_bytes >= (1ll << 31) || out_bytes >= (1ll << 31)) return 0;
    if ((vw + 31) / 32 >= (int64_t)((W + 31) / 32) * B) return 0;               // no tile saved
    return period;
}

// Whether the F(4x4,3x3) kernel serves this launch (its tiles reasonably full; maps of 8 rows go in pairs under MODE 0).
bool conv_wino4_fits(const ConvLayer& L, int B, int H, int W, int pool) {
    if (!L.w_wino4.ptr || (int64_t)H * W * L.cin * 4 >= (1ll << 31)) return false;
    const int period = wino4_side_by_side(L, B, H, W, pool);
    if (period > 0) {
        const int64_t vw = (int64_t)B * period;
        const double fill = (double)B * W * H / ((double)((vw + 31) / 32 * 32) * ((H + 7) / 8 * 8));
        return fill >= 0.7;
    }
    if (H == 8) return (double)W / ((W + 31) / 32 * 32) >= 0.7;          // one half tile, or two images stacked per tile
    const int th = (L.w_wino4b.ptr && wino4_wave_tiles()) ? 8 : 16;       // rows of the tiles the launch will use
    const double fill = (double)H * W / ((double)((H + th - 1) / th * th) * ((W + 31) / 32 * 32));
    return fill >= 0.7;
}

int launch_conv_wino4(const ConvLayer& L, const float* x, float* y, int B, int H, int W, int pool, bool relu,
                      const float* residual, hipStream_t s) {
    OCR_REQUIRE(L.ks == 3 && L.w_wino4.ptr, "conv_wino4: layer has no transformed filters");
    OCR_REQUIRE(x && y && B > 0 && H > 0 && W > 0, "conv_wino4: bad arguments B=%d H=%d W=%d", B, H, W);
    OCR_REQUIRE(L.cout_pad % 64 == 0 && L.cin % (2 * kCK) == 0, "conv_wino4: layer shape %d -> %d not supported", L.cin, L.cout);
    OCR_REQUIRE((int64_t)H * W * L.cin * 4 < (1ll << 31), "conv_wino4: image of %dx%dx%d floats exceeds the 2 GB of a buffer descriptor", H, W, L.cin);
    OCR_REQUIRE(pool >= 0 && pool <= 2, "conv_wino4: bad pool mode %d", pool);
    if (pool) OCR_REQUIRE(H % 2 == 0 && (pool == 2 || W % 2 == 0), "conv_wino4: pooled layer needs even size, got %dx%d", H, W);
    OCR_REQUIRE(!residual || pool == 0, "conv_wino4: a residual goes with an unpooled layer");
    Wino4Args a;
    a.x = x; a.w = L.w_wino4.as<float>(); a.bias = L.bias.as<float>(); a.res = residual; a.y = y;
    a.B = B; a.H = H; a.W = W; a.cin = L.cin; a.cout = L.cout; a.relu = relu ? 1 : 0;
    a.tiles_x = (W + 31) / 32;
    a.pair = (H == 8) ? 1 : 0;
    a.vP = 0; a.vB = B;
    OCR_REQUIRE(!a.pair || (int64_t)2 * H * W * L.cin * 4 < (1ll << 31), "conv_wino4: image pair too large");
    a.tiles_per_image = a.tiles_x * (a.pair ? 1 : (H + 15) / 16);
    a.n_blocks_n = L.cout_pad / 64;
    const int64_t wgs = (int64_t)a.tiles_per_image * a.n_blocks_n * (a.pair ? (B + 1) / 2 : B);
    OCR_REQUIRE(wgs < (1ll << 31), "conv_wino4: too many workgroups");
    static bool lds_opt_in_dev[64] = {};        // more than 64 KB of LDS per workgroup needs an opt-in per kernel and device
    static std::mutex lds_opt_in_mutex;          // handles of several host threads launch through here
    int dev = 0;
    OCR_HIP_CHECK(hipGetDevice(&dev));
    std::lock_guard<std::mutex> lds_opt_in_lock(lds_opt_in_mutex);
    bool& lds_opt_in = lds_opt_in_dev[dev & 63];
#define OCR_W4_FOR_ALL(X) X(0, 0) X(1, 0) X(2, 0) X(0, 1) X(0, 2) X(1, 2) X(2, 2) X(0, 3) X(1, 3) X(2, 3) X(0, 4) X(1, 4) X(2, 4)
    if (!lds_opt_in) {
#define OCR_W4_OPT_IN(P, M)                                                                                               \
    OCR_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_wino4_f32_kernel<P, M>),                       \
                                      hipFuncAttributeMaxDynamicSharedMemorySize, M == 3 ? kLdsBytes3 : (M == 4 ? kLdsBytes4 : kLdsBytes)));
        OCR_W4_FOR_ALL(OCR_W4_OPT_IN)
#undef OCR_W4_OPT_IN
        lds_opt_in = true;
    }
    // Deep layers: transform once into the workspace, then multiply from it (see MODE 1 / 2 of the kernel)
    DevBuf* ws = current_workspace();
    const int nK = L.cin / kCK;
    const int64_t tiles_total = (int64_t)a.tiles_per_image * (a.pair ? (B + 1) / 2 : B);
    const int64_t v_bytes = tiles_total * nK * kUFloats * 4;
    int mode = 0;
    if (ws && wino4_split() && L.cin >= kSplitMinCin && v_bytes <= (6ll << 30) && ws->reserve((size_t)v_bytes) == OCRHIP_OK) mode = 2;
    int64_t wgs_launch = wgs;
    // fused layers: every wave owns whole (block, channel) tiles of half a tile, two workgroups per CU
    if (mode == 0 && L.w_wino4b.ptr && wino4_wave_tiles()) {
        mode = 3;
        a.w = L.w_wino4b.as<float>();
        a.pair = 0;
        const int period = wino4_side_by_side(L, B, H, W, pool);
        if (period > 0) {
            a.vP = period;
            a.B = 1;
            a.tiles_x = (int)(((int64_t)B * period + 31) / 32);
        }
        a.tiles_per_image = a.tiles_x * ((H + 7) / 8);
        // layers with a multiple of 128 output channels: one workgroup of 8 waves per tile and 128 channels, so that
        // the data transform of a tile is shared by twice as many multiplications
        if (L.w_wino4c.ptr && wino4_wave_tiles() >= 2) {
            mode = 4;
            a.w = L.w_wino4c.as<float>();
            a.n_blocks_n = L.cout / 128;
        }
        wgs_launch = (int64_t)a.tiles_per_image * a.n_blocks_n * a.B;
        OCR_REQUIRE(wgs_launch < (1ll << 31), "conv_wino4: too many workgroups");
    }
    a.v = mode == 2 ? ws->as<float>() : nullptr;
    a.chunks_per_group = 8;
    const double out_px = (double)B * H * W;
    const double out_elems = out_px * L.cout / (pool == 1 ? 4.0 : (pool == 2 ? 2.0 : 1.0));
    if (mode == 2) {
        Wino4Args t = a;
        t.n_blocks_n = (nK + t.chunks_per_group - 1) / t.chunks_per_group;
        const int64_t twgs = tiles_total * t.n_blocks_n;
        OCR_REQUIRE(twgs < (1ll << 31), "conv_wino4: too many workgroups");
        ProfScope prof("conv_wino4_transform", 0.0, 4.0 * out_px * L.cin + (double)v_bytes, s);      // HBM bound: input read, tiles written
        hipLaunchKernelGGL((conv_wino4_f32_kernel<0, 1>), dim3((unsigned)twgs), dim3(512), kLdsBytes, s, t);
    }
    // profile name: the fused kernel and the multiply-only kernel are different kernels
    char pname[64];
    static const bool per_layer = getenv("OCRHIP_PROFILE_LAYERS") != nullptr;      // development aid: one profile line per layer shape
    if (per_layer)
        std::snprintf(pname, sizeof(pname), "conv_wino4%s_f32<p%d,c%d-%d,h%d>", mode == 2 ? "_mul" : (mode >= 3 ? "_h" : ""), pool, L.cin, L.cout, H);
    else
        std::snprintf(pname, sizeof(pname), "conv_wino4%s_f32<n%d,pool%d,tile%dx32>", mode == 2 ? "_mul" : (mode >= 3 ? "_h" : ""), mode == 4 ? 128 : 64, pool, mode >= 3 ? 8 : 16);
    ProfScope prof(pname, 2.0 * 9 * L.cin * L.cout * out_px,
                   4.0 * ((mode == 2 ? 0.0 : out_px * L.cin) + out_elems * (residual ? 2.0 : 1.0) + 9.0 * L.cin * L.cout) + (mode == 2 ? (double)v_bytes : 0.0), s);
#define OCR_W4_LAUNCH(P, M)                                                                                               \
    if (pool == P && mode == M && M != 1)                                                                                 \
        hipLaunchKernelGGL((conv_wino4_f32_kernel<P, M>), dim3((unsigned)wgs_launch), dim3(M == 3 ? 256 : 512),           \
                           M == 3 ? kLdsBytes3 : (M == 4 ? kLdsBytes4 : kLdsBytes), s, a);
    OCR_W4_FOR_ALL(OCR_W4_LAUNCH)
#undef OCR_W4_LAUNCH
#undef OCR_W4_FOR_ALL
    OCR_HIP_CHECK(hipGetLastError());
    return OCRHIP_OK;
}

}  // namespace ocrhip
