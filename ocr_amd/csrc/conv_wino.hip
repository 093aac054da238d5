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
// Tiling (one workgroup = 4 waves, one workgroup per CU):
//   spatial tile 8 rows x 32 columns of outputs = 4 x 16 blocks of 2x2, times 64 output channels;
//   the 16 Winograd components are 16 independent GEMMs (blocks x Cin) x (Cin x Cout); a wave owns all 16
//   components of 32 blocks x 32 channels: 16 accumulators of 32x32 (256 accumulator registers per lane), so the
//   inverse transform needs no exchange: a lane holds the 16 components of its (block, channel) pairs.
//   K: chunks of 8 input channels.  Per chunk the raw input tile with its halo (10 x 34 pixels) is staged in
//   LDS, transformed into U[component][block][channel] in LDS (conflict-free 16-byte reads, block stride 12
//   floats), and each lane loads its transformed-filter fragments straight from the lane-linear weight slab
//   (16 coalesced 16-byte loads per chunk).  U and the raw tile are double buffered (128 KB of LDS) so the
//   transform of the next chunk and all its loads are issued between the MFMAs of the current one.
//
// Roofline: MFMA bound; the unit of work is the direct convolution's 2*9*Cin*Cout flop per output pixel, of
// which 16/36 are executed.
#include <type_traits>

#include "common.h"

namespace ocrhip {

namespace {

typedef float floatx16 __attribute__((ext_vector_type(16)));

constexpr int kCK = 8;            // input channels per chunk
constexpr int kRS = 12;           // floats per raw pixel in LDS (8 + pad: conflict-free 16-byte reads)
constexpr int kUS = 12;           // floats per (component, block) row of U
constexpr int kRawW = 34, kRawH = 10, kRawPx = kRawW * kRawH;

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

constexpr int kUFloats = 16 * 64 * kUS;       // one U buffer: 48 KB
constexpr int kRFloats = kRawPx * kRS;        // one raw-tile buffer: 16 KB
constexpr int kLdsBytes = (2 * kUFloats + 2 * kRFloats + 4) * 4;

template <int POOL>
__global__ __launch_bounds__(256, 1) void conv_wino_f32_kernel(const WinoArgs a) {
    extern __shared__ __attribute__((aligned(16))) float lds[];      // U[2] then raw[2]: chunk j lives in buffer j & 1

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int li = lane & 31, lh = lane >> 5;
    const int th = wave & 1;          // which 2 of the 4 block rows
    const int ch = wave >> 1;         // which 32 of the 64 output channels

    // Workgroups go to the 8 XCDs round-robin; the work index is permuted so that every XCD walks a contiguous range of
    // it with the channel block varying fastest: the channel blocks of one spatial tile run on the same XCD at about the
    // same time and share the input tile through its L2 (measured: same speed as channel-block-major order, which keeps
    // the filters in L2 instead, and 1.2x less memory-side traffic).
    const unsigned total = gridDim.x, xcd = blockIdx.x & 7u, slot = blockIdx.x >> 3;
    const unsigned work = xcd * (total >> 3) + min(xcd, total & 7u) + slot;
    const int nb = (int)(work % (unsigned)a.n_blocks_n);
    const int tile_all = (int)(work / (unsigned)a.n_blocks_n);
    const int b = tile_all / a.tiles_per_image;
    const int tile = tile_all % a.tiles_per_image;
    const int y0 = (tile / a.tiles_x) * 8, x0 = (tile % a.tiles_x) * 32;
    const int nK = a.cin / kCK;

    floatx16 acc[16];
#pragma unroll
    for (int f = 0; f < 16; ++f)
#pragma unroll
        for (int v = 0; v < 16; ++v) acc[f][v] = 0.0f;

    // weight slab of this channel block: [k chunk][component][channel half][lane][4]
    const float4* wslab = reinterpret_cast<const float4*>(a.w) + (int64_t)nb * nK * (16 * 2 * 64);
    const float4* wlane = wslab + ch * 64 + lane;

    // raw tile: 340 pixels x 2 channel quads over 256 threads; addresses are fixed per thread
    // All loads and stores are unconditional (pixels outside the image read the first pixel of the image and are
    // replaced by zeros on the way to LDS; threads without a pixel store to a spare slot): the chunk loop stays one
    // basic block, which keeps the compiler's count of outstanding loads exact.
    float4 ra[3];
    int64_t raw_src[3];
    int raw_dst[3];
    bool raw_ok[3];
    constexpr int kDummy = 2 * kUFloats + 2 * kRFloats;     // 16 spare bytes behind the buffers
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        const int f = tid + i * 256;
        const int px = f >> 1, c4 = f & 1;
        const int gy = y0 - 1 + px / kRawW;
        int gx = x0 - 1 + px % kRawW, gb = b;
        bool in_x = gx >= 0;
        if (a.vP > 0 && in_x) {
            gb = gx / a.vP;
            gx -= gb * a.vP;
            in_x = gb < a.vB;
        }
        raw_ok[i] = f < kRawPx * 2 && gy >= 0 && gy < a.H && in_x && gx < a.W;
        raw_src[i] = raw_ok[i] ? ((((int64_t)gb * a.H + gy) * a.W + gx) * a.cin + c4 * 4) : (int64_t)b * a.H * a.W * a.cin;
        raw_dst[i] = (f < kRawPx * 2) ? (px * kRS + c4 * 4) : -1;
    }
    auto load_raw = [&](int kc) {
#pragma unroll
        for (int i = 0; i < 3; ++i) ra[i] = *reinterpret_cast<const float4*>(a.x + raw_src[i] + kc * kCK);
    };
    auto store_raw_one = [&](int rbase, int i) {
        const int idx = raw_dst[i] >= 0 ? rbase + raw_dst[i] : kDummy;
        const float4 v = ra[i];
        *reinterpret_cast<float4*>(&lds[idx]) = raw_ok[i] ? v : make_float4(0.f, 0.f, 0.f, 0.f);
    };

    // Data transform: a thread owns (block, 4 channels, 2 of the 4 component rows i), one row per pass `ii`:
    //   i = 0: d0 - d2,  i = 1: d1 + d2,  i = 2: d2 - d1,  i = 3: d1 - d3     (rows of B^T d) = dA + sgn * dB,
    // then (.) B along the columns.
    const int t_cq = tid & 1, t_tx = (tid >> 1) & 15, t_half = (tid >> 5) & 1, t_ty = tid >> 6;
    const int t_src = ((2 * t_ty) * kRawW + 2 * t_tx) * kRS + t_cq * 4;
    int t_offA[2], t_offB[2], t_dst[2];
    float t_sgn[2];
#pragma unroll
    for (int ii = 0; ii < 2; ++ii) {
        const int i = t_half * 2 + ii;
        const int rowA = (i == 0) ? 0 : (i == 2 ? 2 : 1);
        const int rowB = (i == 0 || i == 1) ? 2 : (i == 2 ? 1 : 3);
        t_offA[ii] = t_src + rowA * kRawW * kRS;
        t_offB[ii] = t_src + rowB * kRawW * kRS;
        t_sgn[ii] = (i == 1) ? 1.0f : -1.0f;
        t_dst[ii] = i * 4 * 64 * kUS + (t_ty * 16 + t_tx) * kUS + t_cq * 4;
    }
    float4 trA[4], trB[4], tt[4], tu[4];       // transform registers: raw rows, B^T d row, finished components
    auto tr_read = [&](int rbase, int ii, int c) {
        trA[c] = *reinterpret_cast<const float4*>(&lds[rbase + t_offA[ii] + c * kRS]);
        trB[c] = *reinterpret_cast<const float4*>(&lds[rbase + t_offB[ii] + c * kRS]);
    };
    // an empty volatile asm that "rewrites" a value: keeps the producer before it and the consumers behind it, which
    // pins the slice of work to its slot (the instruction scheduler would otherwise hoist it next to its loads)
    auto pin = [](float4& v) { asm volatile("" : "+v"(v.x), "+v"(v.y), "+v"(v.z), "+v"(v.w)); };
    auto tr_row = [&](int ii, int c) {
        pin(trA[c]);
        pin(trB[c]);
        const float sg = t_sgn[ii];
        tt[c] = make_float4(fmaf(sg, trB[c].x, trA[c].x), fmaf(sg, trB[c].y, trA[c].y), fmaf(sg, trB[c].z, trA[c].z),
                            fmaf(sg, trB[c].w, trA[c].w));
    };
    auto tr_col = [&](int j) {
        pin(tt[j == 0 ? 0 : 1]);
        pin(tt[j == 3 ? 3 : 2]);
        if (j == 0) tu[0] = make_float4(tt[0].x - tt[2].x, tt[0].y - tt[2].y, tt[0].z - tt[2].z, tt[0].w - tt[2].w);
        if (j == 1) tu[1] = make_float4(tt[1].x + tt[2].x, tt[1].y + tt[2].y, tt[1].z + tt[2].z, tt[1].w + tt[2].w);
        if (j == 2) tu[2] = make_float4(tt[2].x - tt[1].x, tt[2].y - tt[1].y, tt[2].z - tt[1].z, tt[2].w - tt[1].w);
        if (j == 3) tu[3] = make_float4(tt[1].x - tt[3].x, tt[1].y - tt[3].y, tt[1].z - tt[3].z, tt[1].w - tt[3].w);
    };
    auto tr_write = [&](int ubase, int ii, int j) {
        *reinterpret_cast<float4*>(&lds[ubase + t_dst[ii] + j * 64 * kUS]) = tu[j];
    };

    float4 bw[16];
    float4 af[2][8];                     // A fragments of the current and of the next half (8 components) of a chunk
    const int u_src = (th * 32 + li) * kUS + lh * 4;
    auto af_read = [&](int ubase, int half, int first, int set) {      // components 8 half + first .. + 3
#pragma unroll
        for (int i = first; i < first + 4; ++i)
            af[set][i] = *reinterpret_cast<const float4*>(&lds[ubase + (half * 8 + i) * 64 * kUS + u_src]);
    };

    // prologue: chunk 0 transformed into U[0], raw tile of chunk 1 in raw[1], filter fragments of chunk 0 in registers
    constexpr int kR0 = 2 * kUFloats, kR1 = 2 * kUFloats + kRFloats;
    load_raw(0);
#pragma unroll
    for (int f = 0; f < 8; ++f) bw[f] = wlane[f * 128];         // fragments 8..15 are loaded by the chunk itself
#pragma unroll
    for (int i = 0; i < 3; ++i) store_raw_one(kR0, i);
    if (nK > 1) load_raw(1);
    __syncthreads();
#pragma unroll
    for (int ii = 0; ii < 2; ++ii) {
#pragma unroll
        for (int c = 0; c < 4; ++c) tr_read(kR0, ii, c);
#pragma unroll
        for (int c = 0; c < 4; ++c) tr_row(ii, c);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            tr_col(j);
            tr_write(0, ii, j);
        }
    }
    if (nK > 1) {
#pragma unroll
        for (int i = 0; i < 3; ++i) store_raw_one(kR1, i);
    }
    __syncthreads();
    af_read(0, 0, 0, 0);
    af_read(0, 0, 4, 0);
    // every load of the prologue has landed before the loop: the waits inside the loop are then derived from the
    // loop's own loads only (a pending prologue load would make the wait at the loop head conservative for every trip)
    __builtin_amdgcn_s_waitcnt(0x0F70);      // vmcnt(0)

    // Chunk kc: 64 MFMAs on U[kc & 1] in 2 halves of 32 (8 components x 4 k steps; an accumulator is used again 8 MFMAs
    // later: with only 4 in flight the dependent-accumulator latency caps a lone wave at 79 % of the MFMA rate).  All other
    // work of the wave is sliced into the gaps between them, at most ~6 instructions per gap (a lone wave issues one
    // instruction per 4 cycles; what does not fit into the 64 cycles of an MFMA delays the next one): the A fragments of
    // the next half, the transform of chunk kc+1 (raw[(kc+1) & 1] -> U[(kc+1) & 1], rows 0 and 1 of this thread's pair),
    // the filter fragments (components 8..15 of this chunk at its start, 0..7 of chunk kc+1 after their last use), the raw
    // tile of chunk kc+2 (global -> registers -> raw[kc & 1]) and two barriers: slot 0 (raw tile kc+1 visible, U[(kc+1) & 1]
    // no longer read) and slot 44 (U[(kc+1) & 1] complete, so the A fragments of its first half can be fetched early).
    auto chunk = [&](int kc, auto more_c, auto more2_c) {
        constexpr bool more = decltype(more_c)::value, more2 = decltype(more2_c)::value;
        const int ucur = (kc & 1) * kUFloats, unxt = ((kc + 1) & 1) * kUFloats;
        const int rnxt = 2 * kUFloats + ((kc + 1) & 1) * kRFloats, rfill = 2 * kUFloats + (kc & 1) * kRFloats;
        const float4* wcur = wlane + (int64_t)kc * (16 * 2 * 64);
        const float4* wnext = wcur + 16 * 2 * 64;
#pragma unroll
        for (int half = 0; half < 2; ++half)
#pragma unroll
        for (int t = 0; t < 4; ++t)
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int sl = half * 32 + t * 8 + i, f = half * 8 + i;
            const float4 a4 = af[half][i], b4 = bw[f];
            const float av = t == 0 ? a4.x : (t == 1 ? a4.y : (t == 2 ? a4.z : a4.w));
            const float bv = t == 0 ? b4.x : (t == 1 ? b4.y : (t == 2 ? b4.z : b4.w));
            acc[f] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bv, acc[f], 0, 0, 0);
            // ---- side work of this gap
            if (more && sl == 0) __syncthreads();
            if (sl == 1) af_read(ucur, 1, 0, 1);
            if (sl == 3) af_read(ucur, 1, 4, 1);
            if (sl == 2 || sl == 4 || sl == 6 || sl == 8) {
                const int q = 8 + (sl - 2);
                bw[q] = wcur[q * 128];
                bw[q + 1] = wcur[(q + 1) * 128];
            }
            if (more2 && sl == 11) load_raw(kc + 2);
            if (more) {
                if (sl == 5 || sl == 7) tr_read(rnxt, 0, (sl - 5) >> 1);
                if (sl == 9 || sl == 10) tr_read(rnxt, 0, sl - 7);
                if (sl >= 12 && sl <= 15) tr_row(0, sl - 12);
                if (sl >= 17 && sl <= 20) { tr_col(sl - 17); tr_write(unxt, 0, sl - 17); }
                if (sl >= 21 && sl <= 24) tr_read(rnxt, 1, sl - 21);
                if (sl >= 26 && sl <= 29) tr_row(1, sl - 26);
                if (sl == 31 || (sl >= 34 && sl <= 36)) { const int j = sl == 31 ? 0 : sl - 33; tr_col(j); tr_write(unxt, 1, j); }
                if (sl >= 37 && sl <= 40) {
                    const int q = 2 * (sl - 37);
                    bw[q] = wnext[q * 128];
                    bw[q + 1] = wnext[(q + 1) * 128];
                }
                if (sl == 44) __syncthreads();
                if (sl == 46) af_read(unxt, 0, 0, 0);
                if (sl == 48) af_read(unxt, 0, 4, 0);
            }
            if (more2 && (sl == 50 || sl == 52 || sl == 54)) store_raw_one(rfill, (sl - 50) >> 1);
            __builtin_amdgcn_sched_barrier(0);
        }
    };
    {
        int kc = 0;
        for (; kc + 2 < nK; ++kc) chunk(kc, std::true_type{}, std::true_type{});
        if (kc + 1 < nK) {
            chunk(kc, std::true_type{}, std::false_type{});
            ++kc;
        }
        chunk(kc, std::false_type{}, std::false_type{});
    }

    // ---- inverse transform and optional pool in registers, then through LDS so that the tile leaves as 16-byte
    // stores (4x fewer store instructions than one dword per lane; the store tail is issue bound).  Register v of a
    // lane belongs to block m = (v & 3) + 8 * (v >> 2) + 4 * (lane >> 5) of the wave's 32 (2 block rows x 16 block
    // columns) and to output channel lane & 31 of the wave's 32.
    constexpr int kOS = 68;                                    // floats per staged pixel (64 channels + pad)
    constexpr int kOutW = (POOL == 1) ? 16 : 32;               // staged pixels per row
    constexpr int kOutPx = (POOL == 0) ? 256 : (POOL == 1 ? 64 : 128);
    __syncthreads();                                           // every wave is done with U
    float* O = lds;
    const int oc = ch * 32 + li;
#pragma unroll
    for (int v = 0; v < 16; ++v) {
        const int m = (v & 3) + 8 * (v >> 2) + 4 * lh;
        const int ty = th * 2 + (m >> 4), tx = m & 15;
        float s[2][4];       // A^T M : rows 0,1 x 4 columns
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            s[0][j] = acc[j][v] + acc[4 + j][v] + acc[8 + j][v];
            s[1][j] = acc[4 + j][v] - acc[8 + j][v] - acc[12 + j][v];
        }
        float o[2][2];
#pragma unroll
        for (int r = 0; r < 2; ++r) {
            o[r][0] = s[r][0] + s[r][1] + s[r][2];
            o[r][1] = s[r][1] - s[r][2] - s[r][3];
        }
        if (POOL == 0) {
#pragma unroll
            for (int r = 0; r < 2; ++r)
#pragma unroll
                for (int c = 0; c < 2; ++c) O[((2 * ty + r) * 32 + 2 * tx + c) * kOS + oc] = o[r][c];
        } else if (POOL == 1) {
            O[(ty * 16 + tx) * kOS + oc] = fmaxf(fmaxf(o[0][0], o[0][1]), fmaxf(o[1][0], o[1][1]));
        } else {
#pragma unroll
            for (int c = 0; c < 2; ++c) O[(ty * 32 + 2 * tx + c) * kOS + oc] = fmaxf(o[0][c], o[1][c]);
        }
    }
    __syncthreads();
    // output pass: thread -> (pixel, 4 channels); bias, residual, ReLU here
    const int c4 = tid & 15;
    const int co0 = nb * 64 + c4 * 4;
    const float4 bias4 = *reinterpret_cast<const float4*>(a.bias + co0);          // bias is padded to cout_pad
    const int Ho = (POOL == 0) ? a.H : a.H / 2, Wo = (POOL == 1) ? a.W / 2 : a.W;
    const int oy0 = (POOL == 0) ? y0 : (y0 >> 1), ox0 = (POOL == 1) ? (x0 >> 1) : x0;
    const bool vec_ok = (a.cout & 3) == 0 && co0 + 3 < a.cout;
#pragma unroll 4
    for (int px = tid >> 4; px < kOutPx; px += 16) {
        const int yy = oy0 + px / kOutW;
        int xx = ox0 + px % kOutW, ob = b;
        if (a.vP > 0) {
            const int po = (POOL == 1) ? (a.vP >> 1) : a.vP;      // period of the images in output columns
            ob = xx / po;
            xx -= ob * po;
            if (ob >= a.vB) continue;
        }
        if (yy >= Ho || xx >= Wo) continue;
        float4 val = *reinterpret_cast<const float4*>(&O[px * kOS + c4 * 4]);
        val.x += bias4.x; val.y += bias4.y; val.z += bias4.z; val.w += bias4.w;
        const int64_t idx = (((int64_t)ob * Ho + yy) * Wo + xx) * a.cout + co0;
        if (vec_ok) {
            if (POOL == 0 && a.res) {
                const float4 rr = *reinterpret_cast<const float4*>(a.res + idx);
                val.x += rr.x; val.y += rr.y; val.z += rr.z; val.w += rr.w;
            }
            if (a.relu) {
                val.x = fmaxf(val.x, 0.0f); val.y = fmaxf(val.y, 0.0f); val.z = fmaxf(val.z, 0.0f); val.w = fmaxf(val.w, 0.0f);
            }
            *reinterpret_cast<float4*>(a.y + idx) = val;
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
    OCR_REQUIRE(cin % kCK == 0 && cout > 0, "conv_wino: cin=%d must be a multiple of %d", cin, kCK);
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
    OCR_REQUIRE(L.cout_pad % 64 == 0 && L.cin % kCK == 0, "conv_wino: layer shape %d -> %d not supported", L.cin, L.cout);
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
        if (vw < (1ll << 30) && (vw + 31) / 32 < (int64_t)a.tiles_x * B) {
            a.vP = period;
            a.B = 1;
            a.tiles_x = (int)((vw + 31) / 32);
        }
    }
    a.tiles_per_image = a.tiles_x * ((H + 7) / 8);
    a.n_blocks_n = L.cout_pad / 64;
    const int64_t wgs = (int64_t)a.tiles_per_image * a.n_blocks_n * a.B;
    OCR_REQUIRE(wgs < (1ll << 31), "conv_wino: too many workgroups");
    char pname[64];
    std::snprintf(pname, sizeof(pname), "conv_wino_f32<n64,pool%d,tile8x32>", pool);
    const double out_px = (double)B * H * W;
    const double out_elems = out_px * L.cout / (pool == 1 ? 4.0 : (pool == 2 ? 2.0 : 1.0));
    ProfScope prof(pname, 2.0 * 9 * L.cin * L.cout * out_px,
                   4.0 * (out_px * L.cin + out_elems * (residual ? 2.0 : 1.0) + 9.0 * L.cin * L.cout), s);
    const dim3 grid((unsigned)wgs), block(256);
    static bool lds_opt_in_dev[64] = {};        // more than 64 KB of LDS per workgroup needs an opt-in per kernel and device
    int dev = 0;
    OCR_HIP_CHECK(hipGetDevice(&dev));
    bool& lds_opt_in = lds_opt_in_dev[dev & 63];
    if (!lds_opt_in) {
        OCR_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_wino_f32_kernel<0>), hipFuncAttributeMaxDynamicSharedMemorySize, kLdsBytes));
        OCR_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_wino_f32_kernel<1>), hipFuncAttributeMaxDynamicSharedMemorySize, kLdsBytes));
        OCR_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_wino_f32_kernel<2>), hipFuncAttributeMaxDynamicSharedMemorySize, kLdsBytes));
        lds_opt_in = true;
    }
    if (pool == 0)
        hipLaunchKernelGGL(conv_wino_f32_kernel<0>, grid, block, kLdsBytes, s, a);
    else if (pool == 1)
        hipLaunchKernelGGL(conv_wino_f32_kernel<1>, grid, block, kLdsBytes, s, a);
    else
        hipLaunchKernelGGL(conv_wino_f32_kernel<2>, grid, block, kLdsBytes, s, a);
    OCR_HIP_CHECK(hipGetLastError());
    return OCRHIP_OK;
}

}  // namespace ocrhip
