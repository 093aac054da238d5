// bfloat16 tensors end to end: 3x3 convolutions and flat GEMMs on v_mfma_f32_32x32x16_bf16 with bfloat16 activations
// in HBM (NHWC), float32 accumulation.  This is the path of `compute_dtype = OCRHIP_BF16` for the detector
// (architectures/east_arch.py:114-151) and the VGG recogniser (architectures/crnn_arch.py:38-41,306-328,427-531); the
// reference computes in float32, outputs are compared with the float32 oracle at 1e-2.
//
// Work of a workgroup (8 waves): 256 output pixels (8 rows x 32 columns; flat mode: 256 matrix rows) x BN output
// channels (256, 128 or 64), K in chunks of CK = 64 (or 32) input channels.
//   * Activations: the chunk's halo tile (10 x 34 pixels x CK channels) is copied ONCE per chunk straight into LDS by
//     buffer loads with an LDS destination (no registers, no conversion: the tensor already is bfloat16) and serves all
//     nine taps; two buffers, chunk c + 1 is requested when chunk c starts.  The image border, the gap column between
//     crops tiled side by side and rows beyond the matrix are out-of-range offsets of the buffer descriptor: zeros.
//   * Weights: one (tap, chunk) block of BN x CK per step, prepared at load as the byte image LDS holds, copied the
//     same way into two buffers, the block of step s + 1 while step s multiplies.  Every workgroup of the launch walks
//     the same blocks at about the same time: they are served by the L2.
//   * Both images are rows of CK channels (128 or 64 bytes); the 16-byte piece a lane reads (8 channels: k-slice kk,
//     lane half h) is stored at piece (2 kk + h) ^ f(row), f chosen so that the 16 lanes of a ds_read_b128 phase (rows
//     distinct modulo 16) hit 16 different 16-byte bank groups.  An LDS-destination load writes lane-linear, so the
//     permutation is applied to the SOURCE address (activations) or to the prepared image (weights).
//   * One barrier per step.  The weights are the matrix instruction's row operand and the pixels its column operand:
//     a lane then holds 4 consecutive output channels of ONE pixel per accumulator quad, i.e. 8 contiguous bytes of the
//     NHWC bfloat16 output, which it writes into an LDS staging tile; the output pass reads the tile back in 16-byte
//     pieces along the channels, applies the fused max-pool, and stores whole 128-byte lines.
//
// Roofline: MFMA (dense bfloat16 peak 2.5 PFLOP/s); per step and workgroup 2 * 256 * BN * CK flop against
// BN * CK * 2 bytes of weights from the L2 and 1/9 of a halo tile from HBM.
#include <algorithm>
#include <mutex>

#include "common.h"

namespace ocrhip {

namespace {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float floatx16 __attribute__((ext_vector_type(16)));
typedef unsigned uintx2 __attribute__((ext_vector_type(2)));
typedef unsigned uintx4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) void* lds_ptr_t;

struct TileArgs {
    const uint16_t* x;          // conv: (B, nin, H, Wc, cin) bf16; flat: rows of cin-channel pixels / matrix rows
    const uint16_t* w;          // prepared slab, see prepare_conv_bf16t
    const float* bias;          // cout_pad floats
    void* y;                    // bf16 (or float32) output
    int B;                      // conv: separately tiled outer images
    int nin;                    // conv: inner images side by side in an outer image (crops), gap columns apart
    int H, Wc, gap;             // conv: rows, width of an inner image, zero columns between inner images
    int cin, cout, relu;
    int tiles_x, tiles_per_image, n_tiles_n;
    long long M;                // flat: matrix rows
    int seg_chunks;             // flat: chunks per segment of a row (plain GEMM: all of them)
    int seg_off[4];             // flat: byte offset of segment s from the row's base
    int fh, fw;                 // flat: row m is pixel (m / ((fh-1)(fw-1)), y, x) of (n, fh, fw, cin) images (2x2 VALID); fh = 0: row m starts at m * row_bytes
    int row_bytes;
    unsigned x_bytes;           // flat: bytes of x (offsets beyond it read as zeros)
    // FIRST: the layer in front (3 -> 64, ReLU) is computed into the activation image from the page itself
    const void* page;           // (B, H, W, 3) uint8 or float32 (normalised)
    const void* w1;             // bfloat16 fragments of the first layer's weights, see prepare_first_pair_bf16t
    const float* b1;            // 64 floats
    int page_u8;
};

__device__ __forceinline__ int swz(int row, int np) {
    // 3 (np = 8) or 2 (np = 4) bits: the piece permutation of a row
    return np == 8 ? ((((row >> 1) & 3) << 1) | ((row >> 3) & 1)) : ((((row >> 2) & 1) << 1) | ((row >> 3) & 1));
}

__device__ __forceinline__ unsigned bf16_bits(float f) {
    const __bf16 b = (__bf16)f;
    return (unsigned)__builtin_bit_cast(unsigned short, b);
}
__device__ __forceinline__ unsigned pack2(float lo, float hi) { return bf16_bits(lo) | (bf16_bits(hi) << 16); }
// element-wise maximum of two packed bfloat16 pairs
__device__ __forceinline__ unsigned max2(unsigned p, unsigned q) {
    const float pl = __builtin_bit_cast(float, p << 16), ph = __builtin_bit_cast(float, p & 0xffff0000u);
    const float ql = __builtin_bit_cast(float, q << 16), qh = __builtin_bit_cast(float, q & 0xffff0000u);
    return (__builtin_bit_cast(unsigned, fmaxf(pl, ql)) >> 16) | (__builtin_bit_cast(unsigned, fmaxf(ph, qh)) & 0xffff0000u);
}
__device__ __forceinline__ uintx4 max8(uintx4 p, uintx4 q) {
    return uintx4{max2(p.x, q.x), max2(p.y, q.y), max2(p.z, q.z), max2(p.w, q.w)};
}

// One activation image: instruction i of `ni` (1 KB each) is issued by wave i % 8 with that wave's k-th source offset.
// (No template parameter may reach the arguments of the LDS-destination load: the host pass of the compiler re-checks a
// dependent call against the host's target features and silently drops the kernel's launch stub.)
constexpr int kPatchBytes = 12 * 36 * 3 * 2 + 32 + 256;    // FIRST: the page pixels under a tile, bfloat16 (2592 bytes), and the first layer's bias
constexpr int kMaxAW = 6;
__device__ __forceinline__ void copy_rows(__amdgpu_buffer_rsrc_t rsrc, unsigned char* dst, const int (&src)[kMaxAW], int soff, int wave, int ni) {
#pragma unroll
    for (int k = 0; k < kMaxAW; ++k) {
        const int i = wave + 8 * k;
        if (i < ni) __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (lds_ptr_t)(dst + i * 1024), 16, src[k], soff, 0, 0);
    }
}

// KS: 3 = 3x3 convolution, padding 1; 1 = flat GEMM over rows (with row segments: 2x2 VALID convolutions).
// POOL (KS 3): 0 none, 1 max 2x2, 2 max over row pairs.  OUT_F32 (KS 1): float32 output rows instead of bfloat16.
// FIRST (KS 3, one chunk of 64 channels): the input tensor is the output of the detector's first layer (3 -> 64 + ReLU,
// architectures/east_arch.py:114-118), and instead of copying its halo tile from memory the workgroup computes it: the
// 12 x 36 x 3 page pixels under the tile go to LDS as bfloat16, every wave multiplies 32 halo pixels x 32 (27 + zeros)
// by the 32 x 64 weights with four matrix instructions and writes bias + ReLU of the result, rounded to bfloat16,
// straight into the activation image (zeros where the halo pixel lies outside the picture: the padding of THIS layer).
// The 64-channel tensor at full resolution - 128 bytes per pixel written and read back with halo - never exists.
template <int KS, int CK, int BN, int POOL, bool OUT_F32, bool FIRST = false>
__global__ __launch_bounds__(512) void conv_bf16t_kernel(const TileArgs a) {
    constexpr int TAPS = KS == 3 ? 9 : 1;
    constexpr int ROWB = CK * 2, NP = ROWB / 16, PXI = 64 / NP;
    constexpr int TH = 8, TW = 32, HW = TW + 2;
    constexpr int NPA = KS == 3 ? (TH + 2) * HW : 256;
    constexpr int NA_INSTR = (NPA + PXI - 1) / PXI, NAW = (NA_INSTR + 7) / 8;
    constexpr int A_BYTES = NA_INSTR * 1024;
    constexpr int NB_INSTR = BN * ROWB / 1024, B_BYTES = NB_INSTR * 1024;
    constexpr int WN = BN / 64, WM = 8 / WN, MB = 8 / WM;
    constexpr int KK = CK / 16;
    constexpr int SROW = BN * 2 + 16;
    extern __shared__ __attribute__((aligned(1024))) unsigned char lds[];
    constexpr int A_OFF = 0;

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int r = lane & 31, h = lane >> 5;
    const int wm = wave / WN, wn = wave % WN;

    const unsigned total = gridDim.x, xcd = blockIdx.x & 7u, slot = blockIdx.x >> 3;
    const unsigned work = xcd * (total >> 3) + min(xcd, total & 7u) + slot;
    const int nt = (int)(work % (unsigned)a.n_tiles_n);
    const int tile_all = (int)(work / (unsigned)a.n_tiles_n);
    const int nChunks = (KS == 3 ? a.cin : a.row_bytes / 2) / CK;
    const int nSteps = nChunks * TAPS;
    // a layer of one chunk needs one activation buffer: its workgroups take less LDS (the launcher sizes it) and two of
    // them share a compute unit, one's prologue and epilogue behind the other's multiplications
    const int B_OFF = (nChunks == 1 ? 1 : 2) * A_BYTES;

    int b = 0, y0 = 0, x0 = 0;
    long long m0 = 0;
    if (KS == 3) {
        b = tile_all / a.tiles_per_image;
        const int tile = tile_all % a.tiles_per_image;
        y0 = (tile / a.tiles_x) * TH;
        x0 = (tile % a.tiles_x) * TW;
    } else {
        m0 = (long long)tile_all * 256;
    }
    const int period = a.Wc + a.gap;                    // columns of an inner image in the wide image
    const int Wv = a.nin * period;                      // width of the wide image

    // ---- descriptors and the per-lane source offsets of the activation copies (fixed for the whole kernel)
    const long long outer_in = (long long)a.nin * a.H * a.Wc * a.cin * 2;
    const unsigned x_bytes = KS == 3 ? (unsigned)outer_in : a.x_bytes;
    const __amdgpu_buffer_rsrc_t x_rsrc = __builtin_amdgcn_make_buffer_rsrc(
        (void*)((const unsigned char*)a.x + (KS == 3 ? (long long)b * outer_in : 0ll)), 0, x_bytes, 0x00020000);
    const unsigned char* wslab = (const unsigned char*)a.w + (long long)nt * nSteps * B_BYTES;
    const __amdgpu_buffer_rsrc_t w_rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)wslab, 0, (unsigned)(nSteps * B_BYTES), 0x00020000);
    static_assert(NAW <= kMaxAW, "activation copies per wave");
    int a_src[kMaxAW];
#pragma unroll
    for (int k = NAW; k < kMaxAW; ++k) a_src[k] = (int)0x80000000;
#pragma unroll
    for (int k = 0; k < NAW; ++k) {
        const int i = wave + 8 * k;
        const int p = i * PXI + lane / NP, j = lane % NP;
        const int q = j ^ swz(p, NP);
        bool ok = p < NPA;
        long long off = 0;
        if (KS == 3) {
            const int gy = y0 - 1 + p / HW, vx = x0 - 1 + p % HW;
            ok = ok && gy >= 0 && gy < a.H && vx >= 0 && vx < Wv;
            const int img = ok ? vx / period : 0;
            const int cx = vx - img * period;
            ok = ok && cx < a.Wc;
            off = ((long long)(img * a.H + gy) * a.Wc + cx) * a.cin * 2 + q * 16;
        } else {
            const long long m = m0 + p;
            ok = ok && m < a.M;
            if (a.fh > 0) {
                const int ow = a.fw - 1, oh = a.fh - 1;
                const int xx = (int)(m % ow), yy = (int)((m / ow) % oh);
                const long long n = m / ((long long)ow * oh);
                off = ((n * a.fh + yy) * a.fw + xx) * (long long)a.cin * 2 + q * 16;
            } else {
                off = m * a.row_bytes + q * 16;
            }
        }
        a_src[k] = ok ? (int)off : (int)0x80000000;
    }
    auto a_soff = [&](int c) {          // scalar byte offset of chunk c from a row's first channel
        if (KS == 3) return c * ROWB;
        const int seg = c / a.seg_chunks;
        const int so = seg == 0 ? a.seg_off[0] : (seg == 1 ? a.seg_off[1] : (seg == 2 ? a.seg_off[2] : a.seg_off[3]));
        return so + (c - seg * a.seg_chunks) * ROWB;
    };
    auto issue_b = [&](int s, int buf) {
#pragma unroll
        for (int k = 0; k < (NB_INSTR + 7) / 8; ++k) {
            const int i = wave + 8 * k;
            if (i < NB_INSTR)
                __builtin_amdgcn_raw_ptr_buffer_load_lds(w_rsrc, (lds_ptr_t)(lds + B_OFF + buf * B_BYTES + i * 1024), 16, lane * 16,
                                                        s * B_BYTES + i * 1024, 0, 0);
        }
    };

    // ---- operand read addresses
    int w_base[2];
#pragma unroll
    for (int nb = 0; nb < 2; ++nb) {
        const int n = wn * 64 + nb * 32 + r;
        const int sw = swz(n, NP);
        w_base[nb] = n * ROWB + (((sw >> 1) * 2 + (h ^ (sw & 1))) << 4);
    }
    int x_pix[MB];              // pixel of the lane in the activation image at tap (0,0)
#pragma unroll
    for (int mb = 0; mb < MB; ++mb) x_pix[mb] = KS == 3 ? (wm * MB + mb) * HW + r : (wm * MB + mb) * 32 + r;

    floatx16 acc[MB][2];
#pragma unroll
    for (int mb = 0; mb < MB; ++mb)
#pragma unroll
        for (int nb = 0; nb < 2; ++nb)
#pragma unroll
            for (int v = 0; v < 16; ++v) acc[mb][nb][v] = 0.0f;

    if constexpr (FIRST) {
        static_assert(KS == 3 && CK == 64 && !OUT_F32, "fused first layer");
        issue_b(0, 0);
        constexpr int PW = TW + 4, PH = TH + 4, PN = PH * PW * 3;
        // the first layer's weights as the row operand (prepared in that order: fragment (nb, kk) of lane l at
        // ((nb * 2 + kk) * 64 + l) * 16 bytes; channel nb * 32 + r, k = kk * 16 + h * 8 + e, zero for k >= 27) and its bias,
        // requested before the page pixels: one memory round trip for all three
        bf16x8 w1f[2][2];
#pragma unroll
        for (int nb = 0; nb < 2; ++nb)
#pragma unroll
            for (int kk = 0; kk < 2; ++kk)
                w1f[nb][kk] = *reinterpret_cast<const bf16x8*>(reinterpret_cast<const unsigned char*>(a.w1) + ((nb * 2 + kk) * 64 + lane) * 16);
        uint16_t* patch = reinterpret_cast<uint16_t*>(lds + B_OFF + 2 * B_BYTES);
        float* b1s = reinterpret_cast<float*>(lds + B_OFF + 2 * B_BYTES + kPatchBytes - 256);          // the bias, 64 floats
        if (tid < 64) b1s[tid] = a.b1[tid];
        for (int e = tid; e < PN; e += 512) {
            const int pr = e / (PW * 3), rem = e - pr * (PW * 3);
            const int gy = y0 - 2 + pr, gx = x0 - 2 + rem / 3;
            float v = 0.0f;
            if (gy >= 0 && gy < a.H && gx >= 0 && gx < a.Wc) {
                const long long idx = (((long long)b * a.H + gy) * a.Wc + gx) * 3 + rem % 3;
                v = a.page_u8 ? ((((float)reinterpret_cast<const uint8_t*>(a.page)[idx]) / 255.0f) - 0.5f) / 0.5f
                              : reinterpret_cast<const float*>(a.page)[idx];
            }
            patch[e] = (uint16_t)bf16_bits(v);
        }
        __syncthreads();
        for (int mt = wave; mt * 32 < NPA; mt += 8) {
            const int p = mt * 32 + r;
            const int pc = min(p, NPA - 1);
            const int py = pc / HW, pxx = pc - py * HW;
            const int base = (py * PW + pxx) * 3;
            floatx16 f[2];
#pragma unroll
            for (int nb = 0; nb < 2; ++nb)
#pragma unroll
                for (int v = 0; v < 16; ++v) f[nb][v] = 0.0f;
#pragma unroll
            for (int kk = 0; kk < 2; ++kk) {
                bf16x8 xf1;
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    const int k0 = min(kk * 16 + e, 26), k1 = min(kk * 16 + 8 + e, 26);          // lane half 0 / 1
                    const int off = h ? (k1 / 9) * (PW * 3) + k1 % 9 : (k0 / 9) * (PW * 3) + k0 % 9;
                    xf1[e] = __builtin_bit_cast(__bf16, patch[base + off]);
                }
#pragma unroll
                for (int nb = 0; nb < 2; ++nb) f[nb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w1f[nb][kk], xf1, f[nb], 0, 0, 0);
            }
            // register v of f[nb]: channel nb * 32 + (v & 3) + 8 (v >> 2) + 4 h of halo pixel p
            const int gy = y0 - 1 + py, gx = x0 - 1 + pxx;
            const bool inside = gy >= 0 && gy < a.H && gx >= 0 && gx < a.Wc;
            if (p < NPA) {
                const int sw = swz(p, NP);
#pragma unroll
                for (int nb = 0; nb < 2; ++nb)
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        const int co = nb * 32 + 8 * j + 4 * h;
                        const float4 bias = *reinterpret_cast<const float4*>(b1s + co);
                        uintx2 o = uintx2{0u, 0u};
                        if (inside)
                            o = uintx2{pack2(fmaxf(f[nb][4 * j] + bias.x, 0.0f), fmaxf(f[nb][4 * j + 1] + bias.y, 0.0f)),
                                       pack2(fmaxf(f[nb][4 * j + 2] + bias.z, 0.0f), fmaxf(f[nb][4 * j + 3] + bias.w, 0.0f))};
                        *reinterpret_cast<uintx2*>(lds + A_OFF + p * ROWB + (((co >> 3) ^ sw) << 4) + (co & 7) * 2) = o;
                    }
            }
        }
        __syncthreads();
    } else {
        copy_rows(x_rsrc, lds + A_OFF, a_src, a_soff(0), wave, NA_INSTR);
        issue_b(0, 0);
    }
    for (int c = 0; c < nChunks; ++c) {
        const int a_buf = A_OFF + (c & 1) * A_BYTES;
#pragma unroll
        for (int t = 0; t < TAPS; ++t) {
            const int s = c * TAPS + t;
            // my copies have landed; behind the barrier everybody's have, and everybody is done with step s - 1
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            if (t == 0 && c + 1 < nChunks) copy_rows(x_rsrc, lds + A_OFF + ((c + 1) & 1) * A_BYTES, a_src, a_soff(c + 1), wave, NA_INSTR);
            if (s + 1 < nSteps) issue_b(s + 1, (s + 1) & 1);
            const int b_buf = B_OFF + (s & 1) * B_BYTES;
            const int tap_off = KS == 3 ? (t / 3) * HW + (t % 3) : 0;
            int x_base[MB];
#pragma unroll
            for (int mb = 0; mb < MB; ++mb) {
                const int p = x_pix[mb] + tap_off;
                const int sw = swz(p, NP);
                x_base[mb] = a_buf + p * ROWB + (((sw >> 1) * 2 + (h ^ (sw & 1))) << 4);
            }
#pragma unroll
            for (int kk = 0; kk < KK; ++kk) {
                bf16x8 wf[2], xf[MB];
#pragma unroll
                for (int nb = 0; nb < 2; ++nb) wf[nb] = *reinterpret_cast<const bf16x8*>(lds + b_buf + (w_base[nb] ^ (kk << 5)));
#pragma unroll
                for (int mb = 0; mb < MB; ++mb) xf[mb] = *reinterpret_cast<const bf16x8*>(lds + (x_base[mb] ^ (kk << 5)));
#pragma unroll
                for (int mb = 0; mb < MB; ++mb)
#pragma unroll
                    for (int nb = 0; nb < 2; ++nb)
                        acc[mb][nb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wf[nb], xf[mb], acc[mb][nb], 0, 0, 0);
            }
        }
    }

    // ---- epilogue.  Register v of acc[mb][nb]: output channel n0 + wn*64 + nb*32 + (v&3) + 8*(v>>2) + 4*h of pixel
    // (wm*MB + mb)*32 + r of the tile.
    const int n0 = nt * BN;
    if constexpr (OUT_F32) {
        float* y = reinterpret_cast<float*>(a.y);
        if ((a.cout & 3) == 0) {          // uniform
            // float32 rows whose length is a multiple of 16 bytes: every wave passes its 32 rows x 64 channels through
            // a staging area of its own (row stride 272 bytes: the lanes' 16-byte writes of a quad spread over the
            // banks) and stores 256 contiguous bytes per row: whole 128-byte lines instead of 32-byte pieces
            __builtin_amdgcn_s_barrier();          // every wave is done with the operand images
            constexpr int FROW = 64 * 4 + 16;
            unsigned char* mine = lds + wave * (32 * FROW);
#pragma unroll
            for (int mb = 0; mb < MB; ++mb) {
#pragma unroll
                for (int nb = 0; nb < 2; ++nb)
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        const int cl = nb * 32 + 8 * j + 4 * h;
                        const float4 bias = *reinterpret_cast<const float4*>(a.bias + n0 + wn * 64 + cl);
                        float4 o = make_float4(acc[mb][nb][4 * j] + bias.x, acc[mb][nb][4 * j + 1] + bias.y, acc[mb][nb][4 * j + 2] + bias.z,
                                               acc[mb][nb][4 * j + 3] + bias.w);
                        if (a.relu) { o.x = fmaxf(o.x, 0.0f); o.y = fmaxf(o.y, 0.0f); o.z = fmaxf(o.z, 0.0f); o.w = fmaxf(o.w, 0.0f); }
                        *reinterpret_cast<float4*>(mine + r * FROW + cl * 4) = o;
                    }
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                __builtin_amdgcn_wave_barrier();
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
                const long long mrow = m0 + (wm * MB + mb) * 32;
                const int cbase = n0 + wn * 64;
#pragma unroll
                for (int it = 0; it < 8; ++it) {
                    const int row = it * 4 + (lane >> 4), pc = lane & 15;
                    const float4 v = *reinterpret_cast<const float4*>(mine + row * FROW + pc * 16);
                    if (mrow + row < a.M && cbase + pc * 4 < a.cout) *reinterpret_cast<float4*>(y + (mrow + row) * a.cout + cbase + pc * 4) = v;
                }
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                __builtin_amdgcn_wave_barrier();
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            }
            return;
        }
#pragma unroll
        for (int mb = 0; mb < MB; ++mb) {
            const long long m = m0 + (wm * MB + mb) * 32 + r;
#pragma unroll
            for (int nb = 0; nb < 2; ++nb)
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const int co = n0 + wn * 64 + nb * 32 + 8 * j + 4 * h;
                    const float4 bias = *reinterpret_cast<const float4*>(a.bias + co);          // bias is padded to the tile
                    float o[4] = {acc[mb][nb][4 * j] + bias.x, acc[mb][nb][4 * j + 1] + bias.y, acc[mb][nb][4 * j + 2] + bias.z,
                                  acc[mb][nb][4 * j + 3] + bias.w};
                    if (a.relu) {
#pragma unroll
                        for (int e = 0; e < 4; ++e) o[e] = fmaxf(o[e], 0.0f);
                    }
                    if (m < a.M) {
                        float* dst = y + m * a.cout + co;
#pragma unroll
                        for (int e = 0; e < 4; ++e)
                            if (co + e < a.cout) dst[e] = o[e];
                    }
                }
        }
        return;
    } else {
        __builtin_amdgcn_s_barrier();          // every wave is done with the operand images: LDS becomes the staging tile
#pragma unroll
        for (int mb = 0; mb < MB; ++mb) {
            const int pl = (wm * MB + mb) * 32 + r;
#pragma unroll
            for (int nb = 0; nb < 2; ++nb)
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const int cl = wn * 64 + nb * 32 + 8 * j + 4 * h;
                    const float4 bias = *reinterpret_cast<const float4*>(a.bias + n0 + cl);
                    float o0 = acc[mb][nb][4 * j] + bias.x, o1 = acc[mb][nb][4 * j + 1] + bias.y;
                    float o2 = acc[mb][nb][4 * j + 2] + bias.z, o3 = acc[mb][nb][4 * j + 3] + bias.w;
                    if (a.relu) { o0 = fmaxf(o0, 0.0f); o1 = fmaxf(o1, 0.0f); o2 = fmaxf(o2, 0.0f); o3 = fmaxf(o3, 0.0f); }
                    *reinterpret_cast<uintx2*>(lds + pl * SROW + cl * 2) = uintx2{pack2(o0, o1), pack2(o2, o3)};
                }
        }
        __syncthreads();
        uint16_t* y = reinterpret_cast<uint16_t*>(a.y);
        constexpr int PPX = BN / 8;                                   // 16-byte pieces per pixel
        constexpr int OUTPX = POOL == 0 ? 256 : (POOL == 1 ? 64 : 128);
        constexpr int OW = POOL == 1 ? 16 : 32;                       // output columns of the tile
        const int Ho = POOL ? a.H / 2 : a.H;
        const int Wo = POOL == 1 ? a.Wc / 2 : a.Wc;                   // inner image width of the output
        const int operiod = POOL == 1 ? period / 2 : period;
        const long long outer_out = (long long)a.nin * Ho * Wo * a.cout;
#pragma unroll
        for (int it = 0; it < OUTPX * PPX / 512; ++it) {
            const int id = it * 512 + tid;
            const int opx = id / PPX, pc = id % PPX;
            uintx4 v;
            bool ok;
            long long dst;
            if (KS == 3) {
                const int ty = opx / OW, tx = opx % OW;
                if (POOL == 0) {
                    v = *reinterpret_cast<const uintx4*>(lds + (ty * 32 + tx) * SROW + pc * 16);
                } else if (POOL == 1) {
                    const unsigned char* p0 = lds + ((2 * ty) * 32 + 2 * tx) * SROW + pc * 16;
                    v = max8(max8(*reinterpret_cast<const uintx4*>(p0), *reinterpret_cast<const uintx4*>(p0 + SROW)),
                             max8(*reinterpret_cast<const uintx4*>(p0 + 32 * SROW), *reinterpret_cast<const uintx4*>(p0 + 33 * SROW)));
                } else {
                    const unsigned char* p0 = lds + ((2 * ty) * 32 + tx) * SROW + pc * 16;
                    v = max8(*reinterpret_cast<const uintx4*>(p0), *reinterpret_cast<const uintx4*>(p0 + 32 * SROW));
                }
                const int oy = (POOL ? y0 / 2 : y0) + ty, ovx = (POOL == 1 ? x0 / 2 : x0) + tx;
                const int img = ovx / operiod, ocx = ovx - img * operiod;
                ok = oy < Ho && img < a.nin && ocx < Wo && n0 + pc * 8 < a.cout;
                dst = (long long)b * outer_out + ((long long)(img * Ho + oy) * Wo + ocx) * a.cout + n0 + pc * 8;
            } else {
                v = *reinterpret_cast<const uintx4*>(lds + opx * SROW + pc * 16);
                const long long m = m0 + opx;
                ok = m < a.M && n0 + pc * 8 < a.cout;
                dst = m * a.cout + n0 + pc * 8;
            }
            if (ok) *reinterpret_cast<uintx4*>(y + dst) = v;
        }
    }
}

template <int KS, int CK, int BN>
constexpr int lds_bytes(bool one_chunk) {
    constexpr int ROWB = CK * 2, NP = ROWB / 16, PXI = 64 / NP;
    constexpr int NPA = KS == 3 ? 10 * 34 : 256;
    constexpr int A_BYTES = (NPA + PXI - 1) / PXI * 1024, B_BYTES = BN * ROWB;
    const int loop = (one_chunk ? 1 : 2) * A_BYTES + 2 * B_BYTES;
    constexpr int stage = 256 * (BN * 2 + 16), stage32 = 8 * 32 * (64 * 4 + 16);
    return std::max(loop, std::max(stage, stage32));
}

std::mutex g_opt_in_mutex;

template <int KS, int CK, int BN, int POOL, bool OUT_F32, bool FIRST = false>
int launch_variant_t(const TileArgs& a, int64_t wgs, hipStream_t s) {
    static bool opted[64] = {};
    int dev = 0;
    OCR_HIP_CHECK(hipGetDevice(&dev));
    constexpr int bytes_max = lds_bytes<KS, CK, BN>(false) + (FIRST ? kPatchBytes : 0);
    const int n_chunks = (KS == 3 ? a.cin : a.row_bytes / 2) / CK;
    const int bytes = lds_bytes<KS, CK, BN>(n_chunks == 1) + (FIRST ? kPatchBytes : 0);
    {
        std::lock_guard<std::mutex> lock(g_opt_in_mutex);
        if (!opted[dev & 63]) {
            OCR_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_bf16t_kernel<KS, CK, BN, POOL, OUT_F32, FIRST>),
                                              hipFuncAttributeMaxDynamicSharedMemorySize, bytes_max));
            opted[dev & 63] = true;
        }
    }
    hipLaunchKernelGGL((conv_bf16t_kernel<KS, CK, BN, POOL, OUT_F32, FIRST>), dim3((unsigned)wgs), dim3(512), bytes, s, a);
    OCR_HIP_CHECK(hipGetLastError());
    return OCRHIP_OK;
}

int tile_bn(int cout) { return cout % 256 == 0 ? 256 : (cout > 64 ? 128 : 64); }

}  // namespace

// Slab: [n tile][chunk][tap][row n of the tile][piece][8] bfloat16, piece pp of row n holding the channels of logical
// piece pp ^ swz(n) - byte for byte what a step's weight buffer holds in LDS.  Conv (ks 3): chunk over the input
// channels, tap inside; flat (ks 1, 2): K = ks*ks*cin in the order (tap, channel), one "tap" per step.
int prepare_conv_bf16t(ConvLayer& L, const float* w_oihw, int cout, int cin, int ks) {
    OCR_REQUIRE(cin % 32 == 0 && (ks == 1 || ks == 2 || ks == 3), "conv_bf16t: cin=%d ks=%d not supported", cin, ks);
    const int CK = cin % 64 == 0 ? 64 : 32;
    const int BN = tile_bn(cout);
    const int nN = (cout + BN - 1) / BN, taps = ks * ks;
    const int NP = CK / 8;
    const bool conv = ks == 3;
    const int K = conv ? cin : taps * cin;
    const int nChunks = K / CK, T = conv ? 9 : 1;
    std::vector<uint16_t> slab((size_t)nN * nChunks * T * BN * CK, 0);
    for (int ntile = 0; ntile < nN; ++ntile)
        for (int c = 0; c < nChunks; ++c)
            for (int t = 0; t < T; ++t) {
                uint16_t* blk = slab.data() + ((size_t)(ntile * nChunks + c) * T + t) * BN * CK;
                for (int n = 0; n < BN; ++n) {
                    const int o = ntile * BN + n;
                    if (o >= cout) continue;
                    const int sw = NP == 8 ? ((((n >> 1) & 3) << 1) | ((n >> 3) & 1)) : ((((n >> 2) & 1) << 1) | ((n >> 3) & 1));
                    for (int pp = 0; pp < NP; ++pp)
                        for (int e = 0; e < 8; ++e) {
                            const int k = c * CK + (pp ^ sw) * 8 + e;
                            const int ch = conv ? k : k % cin, tap = conv ? t : k / cin;
                            const float v = w_oihw[((size_t)o * cin + ch) * taps + tap];
                            const __bf16 bv = (__bf16)v;
                            blk[(n * NP + pp) * 8 + e] = __builtin_bit_cast(unsigned short, bv);
                        }
                }
            }
    OCR_TRY(L.w_bf16t.reserve(slab.size() * sizeof(uint16_t)));
    OCR_HIP_CHECK(hipMemcpy(L.w_bf16t.ptr, slab.data(), slab.size() * sizeof(uint16_t), hipMemcpyHostToDevice));
    // the tile kernels read the bias as float4 up to the end of the last tile
    if (L.bias.bytes < (size_t)nN * BN * sizeof(float)) {
        std::vector<float> bpad((size_t)nN * BN, 0.0f);
        if (L.bias.ptr) OCR_HIP_CHECK(hipMemcpy(bpad.data(), L.bias.ptr, std::min(L.bias.bytes, (size_t)L.cout_pad * sizeof(float)), hipMemcpyDeviceToHost));
        OCR_TRY(L.bias.reserve(bpad.size() * sizeof(float)));
        OCR_HIP_CHECK(hipMemcpy(L.bias.ptr, bpad.data(), bpad.size() * sizeof(float), hipMemcpyHostToDevice));
    }
    return OCRHIP_OK;
}

// 3x3 convolution, padding 1, on bfloat16 NHWC: x (B, nin, H, Wc, cin) -> y (B, nin, Ho, Wo, cout).  gap > 0 tiles the
// nin inner images of an outer image side by side, `gap` zero columns apart (gap 2 under 2x2 pooling).
int launch_conv3_bf16t(const ConvLayer& L, const uint16_t* x, uint16_t* y, int B, int nin, int H, int Wc, int gap, int pool,
                       bool relu, hipStream_t s) {
    OCR_REQUIRE(L.ks == 3 && L.w_bf16t.ptr, "conv3_bf16t: layer has no bfloat16 tile weights");
    OCR_REQUIRE(x && y && B > 0 && nin > 0 && H > 0 && Wc > 0, "conv3_bf16t: bad arguments");
    OCR_REQUIRE(L.cout % 8 == 0, "conv3_bf16t: output channels must be a multiple of 8");
    if (pool) OCR_REQUIRE(H % 2 == 0 && (pool == 2 || (Wc % 2 == 0 && gap % 2 == 0)), "conv3_bf16t: pooled layer needs even sizes");
    OCR_REQUIRE((int64_t)nin * H * Wc * L.cin * 2 < (1ll << 31), "conv3_bf16t: an outer image exceeds 2 GB");
    const int BN = tile_bn(L.cout), CK = L.cin % 64 == 0 ? 64 : 32;
    TileArgs a{};
    a.x = x; a.w = L.w_bf16t.as<uint16_t>(); a.bias = L.bias.as<float>(); a.y = y;
    a.B = B; a.nin = nin; a.H = H; a.Wc = Wc; a.gap = nin > 1 ? gap : 0; a.cin = L.cin; a.cout = L.cout; a.relu = relu ? 1 : 0;
    const int64_t Wv = (int64_t)nin * (Wc + a.gap);
    a.tiles_x = (int)((Wv + 31) / 32);
    a.tiles_per_image = a.tiles_x * ((H + 7) / 8);
    a.n_tiles_n = (L.cout + BN - 1) / BN;
    const int64_t wgs = (int64_t)a.tiles_per_image * a.n_tiles_n * B;
    OCR_REQUIRE(wgs < (1ll << 31), "conv3_bf16t: too many workgroups");
    char pname[64];
    static const bool per_layer = getenv("OCRHIP_PROFILE_LAYERS") != nullptr;
    if (per_layer)
        std::snprintf(pname, sizeof(pname), "conv_bf16t<p%d,c%d-%d,h%d>", pool, L.cin, L.cout, H);
    else
        std::snprintf(pname, sizeof(pname), "conv_bf16t<ks3,n%d,k%d,pool%d>", BN, CK, pool);
    const double out_px = (double)B * nin * H * Wc;
    const double out_elems = out_px * L.cout / (pool == 1 ? 4.0 : (pool == 2 ? 2.0 : 1.0));
    ProfScope prof(pname, 2.0 * 9 * L.cin * L.cout * out_px, 2.0 * (out_px * L.cin + out_elems + 9.0 * L.cin * L.cout), s);
#define OCR_T_CASE(CK_, BN_, POOL_) \
    if (CK == CK_ && BN == BN_ && pool == POOL_) return launch_variant_t<3, CK_, BN_, POOL_, false>(a, wgs, s);
    OCR_T_CASE(64, 256, 0) OCR_T_CASE(64, 256, 1) OCR_T_CASE(64, 256, 2)
    OCR_T_CASE(64, 128, 0) OCR_T_CASE(64, 128, 1) OCR_T_CASE(64, 128, 2)
    OCR_T_CASE(64, 64, 0) OCR_T_CASE(64, 64, 1)
    OCR_T_CASE(32, 64, 0) OCR_T_CASE(32, 64, 1)
#undef OCR_T_CASE
    set_error("conv3_bf16t: no kernel variant for cin=%d cout=%d pool=%d", L.cin, L.cout, pool);
    return OCRHIP_EINVAL;
}

// The 3 -> 64 first layer as matrix-instruction row operands: [nb][kk][lane = h * 32 + r][8] bfloat16, element e = weight
// of channel nb * 32 + r at k = kk * 16 + h * 8 + e, k = (ky * 3 + kx) * 3 + c, zero for k >= 27.
int prepare_first_pair_bf16t(ConvLayer& first, const float* w_oihw) {
    OCR_REQUIRE(first.cin == 3 && first.cout == 64 && first.ks == 3, "first_pair: expected the 3 -> 64 3x3 layer");
    std::vector<uint16_t> t(2 * 2 * 64 * 8, 0);
    for (int nb = 0; nb < 2; ++nb)
        for (int kk = 0; kk < 2; ++kk)
            for (int l = 0; l < 64; ++l)
                for (int e = 0; e < 8; ++e) {
                    const int k = kk * 16 + (l >> 5) * 8 + e, o = nb * 32 + (l & 31);
                    if (k >= 27) continue;
                    const __bf16 v = (__bf16)w_oihw[((size_t)o * 3 + k % 3) * 9 + k / 3];
                    t[(((nb * 2 + kk) * 64) + l) * 8 + e] = __builtin_bit_cast(unsigned short, v);
                }
    OCR_TRY(first.w_bf16t.reserve(t.size() * sizeof(uint16_t)));
    OCR_HIP_CHECK(hipMemcpy(first.w_bf16t.ptr, t.data(), t.size() * sizeof(uint16_t), hipMemcpyHostToDevice));
    return OCRHIP_OK;
}

// The detector's first two layers in one launch: page (B, H, W, 3) uint8 or normalised float32 -> first (3 -> 64, ReLU;
// its float32 weights [k][64] and bias) -> L (64 -> 64, ReLU, 2x2 max-pool) -> y (B, H/2, W/2, 64) bfloat16.
int launch_conv3_first_pair_bf16t(const ConvLayer& first, const ConvLayer& L, const void* page, bool page_u8, uint16_t* y, int B, int H,
                                  int W, hipStream_t s) {
    OCR_REQUIRE(first.direct && first.cin == 3 && first.cout == 64 && first.w_bf16t.ptr, "conv3_first_pair: the first layer must be the prepared 3 -> 64 layer");
    OCR_REQUIRE(L.ks == 3 && L.cin == 64 && L.cout == 64 && L.w_bf16t.ptr, "conv3_first_pair: the second layer must be a 64 -> 64 bfloat16 tile layer");
    OCR_REQUIRE(page && y && B > 0 && H > 0 && W > 0 && H % 2 == 0 && W % 2 == 0, "conv3_first_pair: bad arguments");
    OCR_REQUIRE((int64_t)H * W * 64 * 2 < (1ll << 31), "conv3_first_pair: an image exceeds 2 GB");
    TileArgs a{};
    a.x = reinterpret_cast<const uint16_t*>(page); a.w = L.w_bf16t.as<uint16_t>(); a.bias = L.bias.as<float>(); a.y = y;
    a.B = B; a.nin = 1; a.H = H; a.Wc = W; a.gap = 0; a.cin = 64; a.cout = 64; a.relu = 1;
    a.page = page; a.w1 = first.w_bf16t.ptr; a.b1 = first.bias.as<float>(); a.page_u8 = page_u8 ? 1 : 0;
    a.tiles_x = (W + 31) / 32;
    a.tiles_per_image = a.tiles_x * ((H + 7) / 8);
    a.n_tiles_n = 1;
    const int64_t wgs = (int64_t)a.tiles_per_image * B;
    OCR_REQUIRE(wgs < (1ll << 31), "conv3_first_pair: too many workgroups");
    const double px = (double)B * H * W;
    ProfScope prof("conv_bf16t<first pair,n64,pool1>", 2.0 * 9 * 64 * 64 * px + 2.0 * 27 * 64 * px * (340.0 / 256.0),
                   px * (page_u8 ? 3.0 : 12.0) + 2.0 * px * 64 / 4.0, s);
    return launch_variant_t<3, 64, 64, 1, false, true>(a, wgs, s);
}

// Flat GEMM y (M, cout) = x rows . W^T + bias.  ks 1: row m = the cin values at x + m*cin.  ks 2: x is (n, fh, fw, cin),
// row m = the 2x2 VALID patch of output pixel m of (n, fh-1, fw-1): four segments of cin values.
int launch_gemm_bf16t(const ConvLayer& L, const uint16_t* x, void* y, bool y_f32, int64_t M, int n, int fh, int fw, bool relu,
                      hipStream_t s) {
    OCR_REQUIRE((L.ks == 1 || L.ks == 2) && L.w_bf16t.ptr, "gemm_bf16t: layer has no bfloat16 tile weights");
    OCR_REQUIRE(x && y && M > 0, "gemm_bf16t: bad arguments");
    OCR_REQUIRE(y_f32 || L.cout % 8 == 0, "gemm_bf16t: bfloat16 output rows need a multiple of 8 channels");
    const int BN = tile_bn(L.cout), CK = L.cin % 64 == 0 ? 64 : 32;
    TileArgs a{};
    a.x = x; a.w = L.w_bf16t.as<uint16_t>(); a.bias = L.bias.as<float>(); a.y = y;
    a.B = 1; a.nin = 1; a.H = 1; a.Wc = 1; a.gap = 0; a.cin = L.cin; a.cout = L.cout; a.relu = relu ? 1 : 0;
    a.M = M;
    a.seg_chunks = L.cin / CK;
    a.row_bytes = L.ks * L.ks * L.cin * 2;
    if (L.ks == 2) {
        OCR_REQUIRE(n > 0 && fh > 1 && fw > 1 && M == (int64_t)n * (fh - 1) * (fw - 1), "gemm_bf16t: 2x2 layer sizes");
        OCR_REQUIRE((int64_t)n * fh * fw * L.cin * 2 < (1ll << 31), "gemm_bf16t: input exceeds 2 GB");
        a.fh = fh; a.fw = fw;
        a.x_bytes = (unsigned)((int64_t)n * fh * fw * L.cin * 2);
        a.seg_off[0] = 0; a.seg_off[1] = L.cin * 2; a.seg_off[2] = fw * L.cin * 2; a.seg_off[3] = (fw + 1) * L.cin * 2;
    } else {
        OCR_REQUIRE(M * L.cin * 2 < (1ll << 31), "gemm_bf16t: input exceeds 2 GB");
        a.fh = 0; a.fw = 0;
        a.x_bytes = (unsigned)(M * L.cin * 2);
    }
    a.tiles_x = 1; a.tiles_per_image = 1;
    a.n_tiles_n = (L.cout + BN - 1) / BN;
    const int64_t wgs = (M + 255) / 256 * a.n_tiles_n;
    OCR_REQUIRE(wgs < (1ll << 31), "gemm_bf16t: too many workgroups");
    char pname[64];
    std::snprintf(pname, sizeof(pname), "gemm_bf16t<n%d,k%d,%s>", BN, CK, y_f32 ? "f32" : "bf16");
    const double K = (double)L.ks * L.ks * L.cin;
    ProfScope prof(pname, 2.0 * K * L.cout * (double)M, 2.0 * ((double)M * K + K * L.cout) + (y_f32 ? 4.0 : 2.0) * (double)M * L.cout, s);
#define OCR_G_CASE(CK_, BN_, F32_) \
    if (CK == CK_ && BN == BN_ && y_f32 == F32_) return launch_variant_t<1, CK_, BN_, 0, F32_>(a, wgs, s);
    OCR_G_CASE(64, 256, false) OCR_G_CASE(64, 256, true)
    OCR_G_CASE(64, 128, false) OCR_G_CASE(64, 128, true)
    OCR_G_CASE(64, 64, false) OCR_G_CASE(64, 64, true)
#undef OCR_G_CASE
    set_error("gemm_bf16t: no kernel variant for cin=%d cout=%d", L.cin, L.cout);
    return OCRHIP_EINVAL;
}

}  // namespace ocrhip
