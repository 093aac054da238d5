// Flat GEMM on the f32 matrix cores (v_mfma_f32_32x32x2_f32): y[M, N] = act(A[M, K] W[N, K]^T + bias).
//
// Carries the contractions of the recogniser that have no spatial reuse to exploit: the LSTM input projections and
// the Dense layers of architectures/crnn_arch.py:306-328,517-519 (K = 256 or 512, N = 2048 / 256 / vocabulary) and
// its 2x2 VALID convolutions (crnn_arch.py:38-41: the last layer of the VGG extractor; crnn_arch.py:330-425: conv4_1 /
// conv4_2 of the ResNet one).  A 2x2 convolution over NHWC data is a GEMM whose A row is not one contiguous run but
// KS runs: output pixel (b, r, c) reads, for each kernel row dy, the 2 * Cin floats that start at input pixel
// (b, r + dy, c).  The kernel therefore addresses A as `row base + segment(k)`: K is cut into segments of `seg_len`
// floats that lie `seg_stride` floats apart (one segment for a plain GEMM), and the row base is any function of the
// row number (here: the pixel's offset).  No im2col copy, no half-empty spatial tile: every row of the 256-row M
// tile is a real output pixel.
//
// Same arithmetic as conv_mfma.hip (exact float32 products accumulated in float32 in k order); what differs is the
// blocking: a workgroup (4 waves) owns 256 rows x 128 columns (a wave 64 x 128: 8 accumulator tiles, so an A operand
// is reused by 4 MFMAs and a B operand by 2), K runs in chunks of 16 through a double-buffered LDS image (one barrier
// per chunk, the global loads of chunk k + 1 issued before the MFMAs of chunk k), and two workgroups share a CU.
//
// ARGMAX variant (the class layer Dense(vocab) in front of greedy CTC decoding, architectures/crnn_arch.py:517-519 +
// utils/text/ctc_decoder.py:28-32): instead of the 256 x 128 output tile the epilogue writes, per row, the largest
// value of the tile and its column (the first column among equal values) - the logits never reach memory.  The CTC
// kernel finishes the argmax over the column blocks.  Same values (acc + bias), so the same tokens and scores.
//
// Roofline: MFMA bound (2 M N K flop against 64 flop/clk/SIMD).
#include <type_traits>

#include "common.h"

namespace ocrhip {

namespace {

typedef float floatx16 __attribute__((ext_vector_type(16)));

constexpr int kBM = 256, kBN = 128, kCK = 16, kPS = kCK + 4;
constexpr int kAFloats = kBM * kPS, kBFloats = kCK * kBN;

struct GemmArgs {
    const float* x;
    const float* w;          // slabs [n block][k chunk][q][h][128 columns][4]: k = chunk * 16 + 8 q + 4 h + t
    const float* bias;       // n_pad floats
    float* y;                // (M, N)
    float* pmax;             // ARGMAX: (M, n_blocks) largest value of the row within the column block ...
    int* pidx;               //         ... and its column
    int64_t M;
    int N, K, relu;
    int n_blocks;
    // A addressing: row m = (b * Hc + r) * Wc + c  ->  base ((b * H + r) * W + c) * cin; k -> (k / seg_len) * seg_stride + k % seg_len
    int Hc, Wc, H, W, cin;
    int seg_len, seg_stride;
    int64_t x_elems;         // floats in x: where its buffer descriptors end
    int pos_per_tile;        // MEAN: output positions (b, c) per M tile, each Hc consecutive rows of it
};

// MODE 0: y = act(A W^T + bias).  MODE 1 (ARGMAX): see above.  MODE 2 (MEAN): the mean over the Hc output rows of every
// (b, c) - GlobalAverage over the height that follows the extractor's last convolution (architectures/crnn_arch.py:
// 223-226, 427-477) - leaves the kernel instead of the rows themselves: the M tile holds pos_per_tile positions with
// their Hc rows next to each other, the epilogue passes 32 columns at a time through LDS, sums the rows of a position
// in row order and divides by Hc (the arithmetic of the separate kernel), and writes y (B * Wc, N).
template <int MODE>
__global__ __launch_bounds__(256, 2) void gemm_f32_kernel(const GemmArgs a) {
    constexpr bool ARGMAX = MODE == 1, MEAN = MODE == 2;
    __shared__ __attribute__((aligned(16))) float As[2][kAFloats];
    __shared__ __attribute__((aligned(16))) float Bs[2][kBFloats];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int li = lane & 31, lh = lane >> 5;

    // every XCD (own L2) walks one contiguous range of the work index, in which the column blocks of an M tile are
    // adjacent: the A tile is fetched from HBM once
    const unsigned total = gridDim.x, xcd = blockIdx.x & 7u, slot = blockIdx.x >> 3;
    const unsigned work = xcd * (total >> 3) + min(xcd, total & 7u) + slot;
    const int nb = (int)(work % (unsigned)a.n_blocks);
    const int64_t tile = (int64_t)(work / (unsigned)a.n_blocks);
    const int64_t m0 = tile * kBM;
    const int nK = a.K / kCK;
    const int64_t pos0 = tile * a.pos_per_tile, n_pos = a.M / a.Hc;          // MEAN

    // the 4 rows this thread stages (row = tid / 4 + 64 i, 16-byte column tid % 4).  Loads and stores go through buffer
    // descriptors that start at the tile's first row (so that 32-bit offsets do for any size of x and y): a per-lane
    // byte offset that never changes, the k chunk / output row as a scalar offset, and the range check instead of
    // guards - rows beyond M carry an offset of 2^31: their loads return zeros and their stores are dropped.
    auto row_offset = [&](int64_t m) {                         // element offset of the A row of output row m (M < 2^31)
        const unsigned mu = (unsigned)m;
        const unsigned c = mu % (unsigned)a.Wc, br = mu / (unsigned)a.Wc;
        const unsigned r = br % (unsigned)a.Hc, b = br / (unsigned)a.Hc;
        return (((int64_t)b * a.H + r) * a.W + c) * a.cin;
    };
    auto mean_row_offset = [&](int ml) {                       // MEAN: tile row ml = position pos0 + ml / Hc, output row ml % Hc
        const unsigned pos = (unsigned)(pos0 + ml / a.Hc), r = (unsigned)(ml % a.Hc);
        const unsigned c = pos % (unsigned)a.Wc, b = pos / (unsigned)a.Wc;
        return (((int64_t)b * a.H + r) * a.W + c) * a.cin;
    };
    const int64_t x_first = MEAN ? mean_row_offset(0) : row_offset(m0);
    const int64_t x_left = (a.x_elems - x_first) * 4;
    const __amdgpu_buffer_rsrc_t x_rsrc = __builtin_amdgcn_make_buffer_rsrc(
        (void*)(a.x + x_first), 0, (int)(x_left < 0x7fffffffll ? x_left : 0x7fffffffll), 0x00020000);
    int abase[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int ml = (tid >> 2) + 64 * i;
        const int64_t m = m0 + ml;
        if (MEAN)
            abase[i] = (ml < a.pos_per_tile * a.Hc && pos0 + ml / a.Hc < n_pos) ? (int)((mean_row_offset(ml) - x_first + (tid & 3) * 4) * 4) : (int)0x80000000;
        else
            abase[i] = (m < a.M) ? (int)((row_offset(m) - x_first + (tid & 3) * 4) * 4) : (int)0x80000000;
    }
    const float* wslab = a.w + (int64_t)nb * nK * kBFloats;

    floatx16 acc[2][4];
#pragma unroll
    for (int r = 0; r < 2; ++r)
#pragma unroll
        for (int nt = 0; nt < 4; ++nt)
#pragma unroll
            for (int v = 0; v < 16; ++v) acc[r][nt][v] = 0.0f;

    // bias now: a load still in flight in the epilogue would make every guarded store wait for the one before it
    float biasv[4];
#pragma unroll
    for (int nt = 0; nt < 4; ++nt) biasv[nt] = a.bias[nb * kBN + nt * 32 + li];

    // (scalars, not a two-element array: the compiler moves such an array to LDS, i.e. waits for the load at once)
    float4 ra[4], rb0, rb1;
    auto load_chunk = [&](int kc) {
        const float4* src = reinterpret_cast<const float4*>(wslab + (int64_t)kc * kBFloats);
        rb0 = src[tid];
        rb1 = src[tid + 256];
        const int k0 = kc * kCK;
        const int koff = ((k0 / a.seg_len) * a.seg_stride + k0 % a.seg_len) * 4;      // uniform: a scalar offset
#pragma unroll
        for (int i = 0; i < 4; ++i)
            ra[i] = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(x_rsrc, abase[i], koff, 0));
    };
    auto store_chunk = [&](int buf) {
        reinterpret_cast<float4*>(Bs[buf])[tid] = rb0;
        reinterpret_cast<float4*>(Bs[buf])[tid + 256] = rb1;
#pragma unroll
        for (int i = 0; i < 4; ++i) *reinterpret_cast<float4*>(&As[buf][((tid >> 2) + 64 * i) * kPS + (tid & 3) * 4]) = ra[i];
    };

    load_chunk(0);
    store_chunk(0);
    __syncthreads();

    for (int kc = 0; kc < nK; ++kc) {
        const int buf = kc & 1;
        if (kc + 1 < nK) load_chunk(kc + 1);
#pragma unroll
        for (int q = 0; q < kCK / 8; ++q) {
            float av[2][4], bv[4][4];
#pragma unroll
            for (int r = 0; r < 2; ++r) {
                const float4 t4 = *reinterpret_cast<const float4*>(&As[buf][(wave * 64 + r * 32 + li) * kPS + q * 8 + lh * 4]);
                av[r][0] = t4.x; av[r][1] = t4.y; av[r][2] = t4.z; av[r][3] = t4.w;
            }
#pragma unroll
            for (int nt = 0; nt < 4; ++nt) {
                const float4 t4 = *reinterpret_cast<const float4*>(&Bs[buf][(((q * 2 + lh) * kBN) + nt * 32 + li) * 4]);
                bv[nt][0] = t4.x; bv[nt][1] = t4.y; bv[nt][2] = t4.z; bv[nt][3] = t4.w;
            }
#pragma unroll
            for (int t = 0; t < 4; ++t)
#pragma unroll
                for (int r = 0; r < 2; ++r)
#pragma unroll
                    for (int nt = 0; nt < 4; ++nt)
                        acc[r][nt] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[r][t], bv[nt][t], acc[r][nt], 0, 0, 0);
        }
        // the LDS writes below stay behind the MFMAs, and the latency of the loads with them (left alone the compiler
        // merges the two `kc + 1 < nK` regions and writes the weight slab to LDS right after loading it)
        asm volatile("" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
        if (kc + 1 < nK) {
            // the other buffer was last read for chunk kc - 1, before every wave's arrival at the previous barrier
            store_chunk(buf ^ 1);
            __syncthreads();
        }
    }

    // accumulator register v of a lane: row (v & 3) + 8 * (v >> 2) + 4 * (lane >> 5) of its 32-row tile, column lane & 31
    // Stores: the lane's offset within the wave's 64 x 128 part of the tile is one VGPR for the whole epilogue; the row
    // (r, v) and the column tile nt go into the scalar offset, which costs no vector instruction.  The range check of a
    // buffer access looks at the vector offset only, so in a tile that crosses the end of the rows or of the columns the
    // lanes that must not write get the out-of-range offset 2^31 per store.
    const bool full = (m0 + kBM <= a.M) && ((nb + 1) * kBN <= a.N);      // uniform: no guards on the stores
    auto write_out = [&](auto guarded) {
        const int64_t row0 = m0 + wave * 64;                             // descriptor from the wave's first row: 32-bit offsets do
        const __amdgpu_buffer_rsrc_t y_rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)(a.y + row0 * a.N), 0, 0x7fffffff, 0x00020000);
        const int row_bytes = a.N * 4;
        const int lane_off = (4 * lh) * row_bytes + (nb * kBN + li) * 4;
#pragma unroll
        for (int nt = 0; nt < 4; ++nt) {
            const bool co_ok = nb * kBN + nt * 32 + li < a.N;
#pragma unroll
            for (int r = 0; r < 2; ++r) {
#pragma unroll
                for (int v = 0; v < 16; ++v) {
                    const int dm = (v & 3) + 8 * (v >> 2);
                    float o = acc[r][nt][v] + biasv[nt];
                    if (a.relu) o = fmaxf(o, 0.0f);
                    const bool ok = !decltype(guarded)::value || (co_ok && row0 + r * 32 + 4 * lh + dm < a.M);
                    __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, o), y_rsrc, ok ? lane_off : (int)0x80000000,
                                                          (r * 32 + dm) * row_bytes + nt * 128, 0);
                }
            }
        }
    };
    if (MEAN) {
        constexpr int SS = 33;                                  // floats per staged row: the column reads of a position's rows hit 32 banks
        float* S = &As[0][0];
        static_assert(2 * kAFloats >= kBM * SS, "staging area");
        __syncthreads();                                        // every wave is done with the operand images
        const int col = tid & 31;
#pragma unroll
        for (int nt = 0; nt < 4; ++nt) {
#pragma unroll
            for (int r = 0; r < 2; ++r)
#pragma unroll
                for (int v = 0; v < 16; ++v) {
                    float o = acc[r][nt][v] + biasv[nt];
                    if (a.relu) o = fmaxf(o, 0.0f);
                    S[(wave * 64 + r * 32 + (v & 3) + 8 * (v >> 2) + 4 * lh) * SS + li] = o;
                }
            __syncthreads();
            const int co = nb * kBN + nt * 32 + col;
            for (int p = tid >> 5; p < a.pos_per_tile; p += 8) {
                float sum = S[(p * a.Hc) * SS + col];
                for (int r = 1; r < a.Hc; ++r) sum += S[(p * a.Hc + r) * SS + col];
                if (pos0 + p < n_pos && co < a.N) a.y[(pos0 + p) * a.N + co] = sum / (float)a.Hc;
            }
            __syncthreads();
        }
        return;
    }
    if (ARGMAX) {
#pragma unroll
        for (int r = 0; r < 2; ++r) {
            const int64_t mrow = m0 + wave * 64 + r * 32 + 4 * lh;
#pragma unroll
            for (int v = 0; v < 16; ++v) {
                float best = -INFINITY;
                int best_i = 0x7fffffff;
#pragma unroll
                for (int nt = 0; nt < 4; ++nt) {                       // ascending columns: a later equal value does not win
                    const int co = nb * kBN + nt * 32 + li;
                    const float o = acc[r][nt][v] + biasv[nt];
                    if (co < a.N && (o > best || best_i == 0x7fffffff)) { best = o; best_i = co; }
                }
#pragma unroll
                for (int off = 16; off > 0; off >>= 1) {               // the 32 lanes that hold the row
                    const float ob = __shfl_xor(best, off, 64);
                    const int oi = __shfl_xor(best_i, off, 64);
                    if (oi != 0x7fffffff && (best_i == 0x7fffffff || ob > best || (ob == best && oi < best_i))) { best = ob; best_i = oi; }
                }
                const int64_t m = mrow + (v & 3) + 8 * (v >> 2);
                if (li == 0 && m < a.M) {
                    a.pmax[m * a.n_blocks + nb] = best;
                    a.pidx[m * a.n_blocks + nb] = best_i;
                }
            }
        }
        return;
    }
    if (full)
        write_out(std::false_type{});
    else
        write_out(std::true_type{});
}

}  // namespace

// Weights (cout, cin, ks, ks) of a 1x1 or 2x2 layer -> the GEMM's slabs.  GEMM k = (ky * ks + kx) * cin + c: the order
// in which an NHWC input row pair lies in memory.
int prepare_gemm_f32(ConvLayer& L, const float* w_oihw, const float* bias, int cout, int cin, int ks) {
    OCR_REQUIRE(ks == 1 || ks == 2, "gemm_f32: kernel size %d not supported", ks);
    const int K = ks * ks * cin;
    OCR_REQUIRE(cin % kCK == 0, "gemm_f32: cin=%d must be a multiple of %d", cin, kCK);
    const int nN = (cout + kBN - 1) / kBN, nK = K / kCK, taps = ks * ks;
    std::vector<float> w((size_t)nN * nK * kBFloats, 0.0f), bpad((size_t)nN * kBN, 0.0f);
    for (int o = 0; o < cout; ++o) {
        if (bias) bpad[o] = bias[o];
        for (int k = 0; k < K; ++k) {
            const int tap = k / cin, c = k % cin;
            const int kc = k / kCK, q = (k % kCK) / 8, hh = (k % 8) / 4, t = k % 4;
            const size_t dst = ((((size_t)(o / kBN) * nK + kc) * (kCK / 8) + q) * 2 + hh) * kBN * 4 + (size_t)(o % kBN) * 4 + t;
            w[dst] = w_oihw[((size_t)o * cin + c) * taps + tap];
        }
    }
    OCR_TRY(L.w_gemm.reserve(w.size() * sizeof(float)));
    OCR_TRY(L.bias_gemm.reserve(bpad.size() * sizeof(float)));
    OCR_HIP_CHECK(hipMemcpy(L.w_gemm.ptr, w.data(), w.size() * sizeof(float), hipMemcpyHostToDevice));
    OCR_HIP_CHECK(hipMemcpy(L.bias_gemm.ptr, bpad.data(), bpad.size() * sizeof(float), hipMemcpyHostToDevice));
    return OCRHIP_OK;
}

// x: (B, H, W, cin) NHWC; y: (B, H - ks + 1, W - ks + 1, cout).  ks == 1 is the flat GEMM over B * H * W rows.
int launch_gemm_f32(const ConvLayer& L, const float* x, float* y, int B, int H, int W, bool relu, hipStream_t s) {
    OCR_REQUIRE(L.w_gemm.ptr && (L.ks == 1 || L.ks == 2), "gemm_f32: layer not prepared");
    OCR_REQUIRE(x && y && B > 0 && H >= L.ks && W >= L.ks, "gemm_f32: bad arguments B=%d H=%d W=%d", B, H, W);
    GemmArgs a;
    a.x = x; a.w = L.w_gemm.as<float>(); a.bias = L.bias_gemm.as<float>(); a.y = y; a.pmax = nullptr; a.pidx = nullptr;
    a.N = L.cout; a.K = L.ks * L.ks * L.cin; a.relu = relu ? 1 : 0;
    a.n_blocks = (L.cout + kBN - 1) / kBN;
    a.H = H; a.W = W; a.cin = L.cin;
    a.Hc = H - L.ks + 1; a.Wc = W - L.ks + 1;
    a.M = (int64_t)B * a.Hc * a.Wc;
    a.seg_len = L.ks * L.cin;             // one kernel row: ks pixels of cin channels, contiguous in NHWC
    a.seg_stride = W * L.cin;
    a.x_elems = (int64_t)B * H * W * L.cin;
    a.pos_per_tile = 0;
    OCR_REQUIRE((int64_t)B * H * W * L.cin < (1ll << 40) && a.M < (1ll << 31), "gemm_f32: input too large");
    OCR_REQUIRE((int64_t)(64 + 40) * a.N * 4 < (1ll << 31), "gemm_f32: N=%d too wide", a.N);          // a wave's store offsets are 32-bit
    const int64_t wgs = (a.M + kBM - 1) / kBM * a.n_blocks;
    OCR_REQUIRE(wgs < (1ll << 31), "gemm_f32: too many workgroups");
    char pname[64];
    std::snprintf(pname, sizeof(pname), "gemm_f32<ks%d,tile256x128>", L.ks);
    ProfScope prof(pname, 2.0 * (double)a.M * a.N * a.K,
                   4.0 * ((double)B * H * W * L.cin + (double)a.M * a.N + (double)a.N * a.K), s);
    hipLaunchKernelGGL(gemm_f32_kernel<0>, dim3((unsigned)wgs), dim3(256), 0, s, a);
    OCR_HIP_CHECK(hipGetLastError());
    return OCRHIP_OK;
}

// x: (B, H, W, cin) NHWC through the 2x2 (or 1x1) layer L, then the mean over the output rows: y (B, W - ks + 1, cout).
int launch_gemm_f32_mean_rows(const ConvLayer& L, const float* x, float* y, int B, int H, int W, bool relu, hipStream_t s) {
    OCR_REQUIRE(L.w_gemm.ptr && (L.ks == 1 || L.ks == 2), "gemm_f32_mean_rows: layer not prepared");
    OCR_REQUIRE(x && y && B > 0 && H >= L.ks && W >= L.ks, "gemm_f32_mean_rows: bad arguments B=%d H=%d W=%d", B, H, W);
    GemmArgs a;
    a.x = x; a.w = L.w_gemm.as<float>(); a.bias = L.bias_gemm.as<float>(); a.y = y; a.pmax = nullptr; a.pidx = nullptr;
    a.N = L.cout; a.K = L.ks * L.ks * L.cin; a.relu = relu ? 1 : 0;
    a.n_blocks = (L.cout + kBN - 1) / kBN;
    a.H = H; a.W = W; a.cin = L.cin;
    a.Hc = H - L.ks + 1; a.Wc = W - L.ks + 1;
    OCR_REQUIRE(a.Hc >= 1 && a.Hc <= 16, "gemm_f32_mean_rows: %d output rows (1..16 supported)", a.Hc);
    a.M = (int64_t)B * a.Hc * a.Wc;
    a.pos_per_tile = kBM / a.Hc;
    a.seg_len = L.ks * L.cin;
    a.seg_stride = W * L.cin;
    a.x_elems = (int64_t)B * H * W * L.cin;
    OCR_REQUIRE((int64_t)B * H * W * L.cin < (1ll << 40) && a.M < (1ll << 31), "gemm_f32_mean_rows: input too large");
    const int64_t n_pos = (int64_t)B * a.Wc;
    const int64_t wgs = (n_pos + a.pos_per_tile - 1) / a.pos_per_tile * a.n_blocks;
    OCR_REQUIRE(wgs < (1ll << 31), "gemm_f32_mean_rows: too many workgroups");
    char pname[64];
    std::snprintf(pname, sizeof(pname), "gemm_f32<ks%d,mean rows,tile256x128>", L.ks);
    ProfScope prof(pname, 2.0 * (double)a.M * a.N * a.K, 4.0 * ((double)B * H * W * L.cin + (double)n_pos * a.N + (double)a.N * a.K), s);
    hipLaunchKernelGGL(gemm_f32_kernel<2>, dim3((unsigned)wgs), dim3(256), 0, s, a);
    OCR_HIP_CHECK(hipGetLastError());
    return OCRHIP_OK;
}

int gemm_f32_column_blocks(const ConvLayer& L) { return (L.cout + kBN - 1) / kBN; }

// x: (M, cin) -> per row and column block of 128 the largest value of x W^T + bias and its column (see ARGMAX above).
int launch_gemm_f32_argmax(const ConvLayer& L, const float* x, int64_t M, float* pmax, int* pidx, hipStream_t s) {
    OCR_REQUIRE(L.w_gemm.ptr && L.ks == 1, "gemm_f32_argmax: layer not prepared");
    OCR_REQUIRE(x && pmax && pidx && M > 0 && M < (1ll << 31), "gemm_f32_argmax: bad arguments");
    GemmArgs a;
    a.x = x; a.w = L.w_gemm.as<float>(); a.bias = L.bias_gemm.as<float>(); a.y = nullptr; a.pmax = pmax; a.pidx = pidx;
    a.N = L.cout; a.K = L.cin; a.relu = 0;
    a.n_blocks = gemm_f32_column_blocks(L);
    a.H = 1; a.W = (int)M; a.cin = L.cin; a.Hc = 1; a.Wc = (int)M; a.M = M;
    a.seg_len = L.cin; a.seg_stride = (int)M * L.cin; a.x_elems = M * L.cin; a.pos_per_tile = 0;
    const int64_t wgs = (a.M + kBM - 1) / kBM * a.n_blocks;
    OCR_REQUIRE(wgs < (1ll << 31), "gemm_f32_argmax: too many workgroups");
    ProfScope prof("gemm_f32_argmax<tile256x128>", 2.0 * (double)a.M * a.N * a.K,
                   4.0 * ((double)M * L.cin + 2.0 * (double)a.M * a.n_blocks + (double)a.N * a.K), s);
    hipLaunchKernelGGL(gemm_f32_kernel<1>, dim3((unsigned)wgs), dim3(256), 0, s, a);
    OCR_HIP_CHECK(hipGetLastError());
    return OCRHIP_OK;
}

}  // namespace ocrhip
