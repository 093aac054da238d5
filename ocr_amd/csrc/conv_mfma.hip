// Implicit-GEMM convolution / GEMM on the f32 matrix cores (v_mfma_f32_32x32x2_f32), NHWC float32.
//
// Carries every dense contraction of the path except the two 1- and 3-channel input layers:
// the VGG16-BN body and the merge branch of EAST (architectures/east_arch.py:114-151), the VGG
// extractor of the CRNN (architectures/crnn_arch.py:38-41,427-477) and, as a flat GEMM, the LSTM
// input projections and the Dense layers (crnn_arch.py:306-328,517-519).  Batch-norm is folded
// into weights and bias at load time; bias, ReLU and the 2x2 / (2,1) max-pool that follows a
// convolution are applied in the epilogue, and for the merge branch the x2 align-corners bilinear
// upsampling and the channel concatenation (east_arch.py:37-48,136-137) happen while the input
// tile is staged, so neither the pooled, the upsampled nor the concatenated tensor is ever written.
//
// f32 MFMA multiplies exact float32 products and accumulates them in float32 in k order, i.e. the
// result is a float32 fma chain: no reduced-precision step anywhere.
//
// Tiling (one workgroup = 4 waves = 256 threads):
//   M: a spatial tile of 8 rows x 32 columns of output pixels (KS = 3, 2; three smaller shapes exist
//      for small maps, see Geo) or 256 consecutive rows of the flat (B*H*W, Cin) matrix (KS = 1);
//      each wave owns 2 x 32 pixels,
//   N: NT x 32 output channels,
//   K: chunks of CK input channels; per chunk the input tile with its halo ((8+KS-1) x (32+KS-1)
//      pixels) and the weight slab (KS*KS x CK x N) are staged in LDS once and reused by all taps.
// Operands are read from LDS as 16-byte vectors: a lane takes 4 consecutive channels of its pixel
// (A) or of its output channel (B) and feeds them to 4 successive MFMAs; the k slot of lane half h
// is channel 8q + 4h + t, the same for A and B, so any k permutation cancels.  The pixel stride in
// LDS is CK + 4 floats, which spreads the 16 lanes of a ds_read_b128 group over all 16-byte slots
// of the 256-byte bank row (conflict free); the weight image is lane-linear.
//
// Roofline: MFMA bound.  2 * KS*KS * Cin * Cout flop per output pixel against 64 flop/clk/SIMD.
#include "common.h"

namespace ocrhip {

namespace {

typedef float floatx16 __attribute__((ext_vector_type(16)));

struct ConvArgs {
    const float* x;        // (B,H,W,cin_x) NHWC, or (M,cin) for KS == 1
    const float* w;        // slab layout, see prepare_conv_mfma
    const float* bias;     // cout_pad
    float* y;
    const float* up;       // UPCAT: (B,H/2,W/2,up_c)
    const float* res;      // optional residual, same shape as y, added before the ReLU (unpooled 3x3 / 2x2 layers)
    int B, H, W;           // input spatial size (KS == 1: H*W*B = M rows)
    int cin, cout;         // cin = total reduction channels (incl. upsampled ones)
    int up_c;
    int relu;
    float up_ry, up_rx;    // (Hs-1)/(H-1), (Ws-1)/(W-1)
    int tiles_x, tiles_per_image;
    int n_blocks_n;        // output-channel blocks; work index = (image * tiles_per_image + tile) * n_blocks_n + channel block
};

// Spatial tile of a workgroup: TW = 32 or 16 columns; an MFMA M tile (32 pixels) is 1 row x 32 or
// 2 rows x 16; every wave owns MT (2 or 1) M tiles stacked vertically, the 4 waves are stacked too:
// TH = 4 * MT * (32 / TW) rows.  (32,2) = 8 x 32 is the default; the others keep small feature maps
// (recogniser: 16, 8 or 4 rows, widths that are odd multiples of 16) from wasting half-empty tiles.
template <int KS, int TWc, int MT>
struct Geo {
    static constexpr int CK = (KS == 1) ? 32 : 16;
    static constexpr int TAPS = KS * KS;
    static constexpr int TW = TWc, RM = 32 / TWc, TH = 4 * MT * RM;
    static constexpr int HT = TH + KS - 1, WT = TW + KS - 1;
    static constexpr int NPIX = (KS == 1) ? 256 : HT * WT;
    static constexpr int PS = CK + 4;
    static constexpr int PAD = (KS == 3) ? 1 : 0;
};

template <int KS, int NT, int POOL, bool UPCAT, int TWc = 32, int MT = 2>
__global__ __launch_bounds__(256, 2) void conv_mfma_f32_kernel(const ConvArgs a) {
    using G = Geo<KS, TWc, MT>;
    constexpr int RM = G::RM;
    static_assert(KS != 1 || (TWc == 32 && MT == 2), "the flat GEMM uses the default tile");
    static_assert(POOL == 0 || TWc == 16 || MT == 2, "pooling needs both rows of a 2x2 window in one wave");
    constexpr int CK = G::CK, TAPS = G::TAPS, WT = G::WT, NPIX = G::NPIX, PS = G::PS, PAD = G::PAD;
    constexpr int BN = NT * 32;
    constexpr int A_FLOATS = NPIX * PS;
    constexpr int B_FLOATS = TAPS * CK * BN;
    __shared__ __attribute__((aligned(16))) float As[A_FLOATS];
    __shared__ __attribute__((aligned(16))) float Bs[B_FLOATS];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int li = lane & 31, lh = lane >> 5;

    // Workgroups are handed to the 8 XCDs (each with its own L2) round-robin in launch order.  The work
    // index is permuted so that every XCD walks one contiguous range of it: the channel blocks of a
    // spatial tile, and the tiles next to it, then run on the same XCD at about the same time, and the
    // input tile (with its halo) is fetched from HBM once instead of once per channel block.
    const unsigned total = gridDim.x, xcd = blockIdx.x & 7u, slot = blockIdx.x >> 3;
    const unsigned work = xcd * (total >> 3) + min(xcd, total & 7u) + slot;
    const int nb = (int)(work % (unsigned)a.n_blocks_n);
    const int tile_all = (int)(work / (unsigned)a.n_blocks_n);
    const int nK = a.cin / CK;

    // tile origin
    int b = 0, y0 = 0, x0 = 0;
    int64_t m0 = 0;
    if (KS == 1) {
        m0 = (int64_t)tile_all * 256;
    } else {
        b = tile_all / a.tiles_per_image;
        const int tile = tile_all % a.tiles_per_image;
        y0 = (tile / a.tiles_x) * G::TH;
        x0 = (tile % a.tiles_x) * G::TW;
    }
    const int64_t M = (int64_t)a.B * a.H * a.W;
    const int cin_x = a.cin - a.up_c;      // channels that come from a.x

    floatx16 acc[MT][NT];
#pragma unroll
    for (int r = 0; r < MT; ++r)
#pragma unroll
        for (int nt = 0; nt < NT; ++nt)
#pragma unroll
            for (int v = 0; v < 16; ++v) acc[r][nt][v] = 0.0f;

    const float* wslab = a.w + (int64_t)nb * nK * B_FLOATS;
    // bias now: a load still in flight in the epilogue would make every guarded store wait for the one before it
    float biasv[NT];
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) biasv[nt] = a.bias[nb * BN + nt * 32 + li];          // bias is padded to cout_pad

    // Software pipeline over the K chunks: the global loads of chunk k+1 are issued into registers
    // before the matrix work of chunk k and written to LDS after it, so their latency hides behind
    // 144 MFMAs per wave.
    constexpr int V4 = CK / 4;
    constexpr int A_V4 = NPIX * V4, B_V4 = B_FLOATS / 4;
    constexpr int A_IT = (A_V4 + 255) / 256, B_IT = (B_V4 + 255) / 256;
    float4 ra[A_IT], rb[B_IT];

    auto load_chunk = [&](int kc) {
        const int c0 = kc * CK;
        const float4* src = reinterpret_cast<const float4*>(wslab + (int64_t)kc * B_FLOATS);
#pragma unroll
        for (int i = 0; i < B_IT; ++i) {
            const int f = tid + i * 256;
            rb[i] = (B_V4 % 256 == 0 || f < B_V4) ? src[f] : make_float4(0.f, 0.f, 0.f, 0.f);
        }
        if (KS == 1) {
            const bool from_up = UPCAT && (c0 < a.up_c);
#pragma unroll
            for (int i = 0; i < A_IT; ++i) {
                const int f = tid + i * 256;
                const int px = f / V4, c4 = f % V4;
                const int64_t m = m0 + px;
                float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
                if (m < M) {
                    if (!from_up) {
                        v = *reinterpret_cast<const float4*>(a.x + m * cin_x + (c0 - a.up_c) + c4 * 4);
                    } else {
                        // x2 bilinear upsampling with aligned corners of the (H/2, W/2) source
                        const int Hs = a.H / 2, Ws = a.W / 2;
                        const int xo = (int)(m % a.W);
                        const int yo = (int)((m / a.W) % a.H);
                        const int bo = (int)(m / ((int64_t)a.W * a.H));
                        const float sy = a.up_ry * (float)yo, sx = a.up_rx * (float)xo;
                        const int iy = (int)sy, ix = (int)sx;
                        const int iy1 = iy + ((iy < Hs - 1) ? 1 : 0), ix1 = ix + ((ix < Ws - 1) ? 1 : 0);
                        const float ly = sy - (float)iy, lx = sx - (float)ix;
                        const float hy = 1.0f - ly, hx = 1.0f - lx;
                        const float* base = a.up + (int64_t)bo * Hs * Ws * a.up_c + c0 + c4 * 4;
                        const float4 v00 = *reinterpret_cast<const float4*>(base + ((int64_t)iy * Ws + ix) * a.up_c);
                        const float4 v01 = *reinterpret_cast<const float4*>(base + ((int64_t)iy * Ws + ix1) * a.up_c);
                        const float4 v10 = *reinterpret_cast<const float4*>(base + ((int64_t)iy1 * Ws + ix) * a.up_c);
                        const float4 v11 = *reinterpret_cast<const float4*>(base + ((int64_t)iy1 * Ws + ix1) * a.up_c);
                        v.x = hy * (hx * v00.x + lx * v01.x) + ly * (hx * v10.x + lx * v11.x);
                        v.y = hy * (hx * v00.y + lx * v01.y) + ly * (hx * v10.y + lx * v11.y);
                        v.z = hy * (hx * v00.z + lx * v01.z) + ly * (hx * v10.z + lx * v11.z);
                        v.w = hy * (hx * v00.w + lx * v01.w) + ly * (hx * v10.w + lx * v11.w);
                    }
                }
                ra[i] = v;
            }
        } else {
#pragma unroll
            for (int i = 0; i < A_IT; ++i) {
                const int f = tid + i * 256;
                const int px = f / V4, c4 = f % V4;
                const int gy = y0 - PAD + px / WT, gx = x0 - PAD + px % WT;
                float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
                if ((A_V4 % 256 == 0 || f < A_V4) && gy >= 0 && gy < a.H && gx >= 0 && gx < a.W)
                    v = *reinterpret_cast<const float4*>(a.x + (((int64_t)b * a.H + gy) * a.W + gx) * a.cin + c0 + c4 * 4);
                ra[i] = v;
            }
        }
    };

    auto store_chunk = [&]() {
#pragma unroll
        for (int i = 0; i < B_IT; ++i) {
            const int f = tid + i * 256;
            if (B_V4 % 256 == 0 || f < B_V4) reinterpret_cast<float4*>(Bs)[f] = rb[i];
        }
#pragma unroll
        for (int i = 0; i < A_IT; ++i) {
            const int f = tid + i * 256;
            if (A_V4 % 256 == 0 || f < A_V4) *reinterpret_cast<float4*>(&As[(f / V4) * PS + (f % V4) * 4]) = ra[i];
        }
    };

    load_chunk(0);
    store_chunk();
    __syncthreads();

    for (int kc = 0; kc < nK; ++kc) {
        if (kc + 1 < nK) load_chunk(kc + 1);

        // ---- matrix work of this chunk
#pragma unroll
        for (int tap = 0; tap < TAPS; ++tap) {
            const int ky = tap / KS, kx = tap % KS;
#pragma unroll
            for (int q = 0; q < CK / 8; ++q) {
                float av[MT][4], bv[NT][4];
#pragma unroll
                for (int r = 0; r < MT; ++r) {
                    int px;
                    if (KS == 1)
                        px = wave * 64 + r * 32 + li;
                    else
                        px = ((wave * MT + r) * RM + li / TWc + ky) * WT + (li % TWc + kx);
                    const float4 t4 = *reinterpret_cast<const float4*>(&As[px * PS + q * 8 + lh * 4]);
                    av[r][0] = t4.x; av[r][1] = t4.y; av[r][2] = t4.z; av[r][3] = t4.w;
                }
#pragma unroll
                for (int nt = 0; nt < NT; ++nt) {
                    const float4 t4 = *reinterpret_cast<const float4*>(
                        &Bs[((((tap * (CK / 8) + q) * 2 + lh) * BN) + nt * 32 + li) * 4]);
                    bv[nt][0] = t4.x; bv[nt][1] = t4.y; bv[nt][2] = t4.z; bv[nt][3] = t4.w;
                }
#pragma unroll
                for (int t = 0; t < 4; ++t)
#pragma unroll
                    for (int r = 0; r < MT; ++r)
#pragma unroll
                        for (int nt = 0; nt < NT; ++nt)
                            acc[r][nt] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[r][t], bv[nt][t], acc[r][nt], 0, 0, 0);
            }
        }

        if (kc + 1 < nK) {
            __syncthreads();   // every wave has finished reading this chunk from LDS
            store_chunk();
            __syncthreads();
        }
    }

    // ---- epilogue: bias, ReLU, optional max-pool, store.  Accumulator register v of a lane holds
    // output channel (lane & 31) of pixel column (v & 3) + 8 * (v >> 2) + 4 * (lane >> 5).
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
        const int co = nb * BN + nt * 32 + li;
        const bool co_ok = co < a.cout;
        const float bias = biasv[nt];
        if (KS == 1) {
#pragma unroll
            for (int r = 0; r < MT; ++r)
#pragma unroll
                for (int v = 0; v < 16; ++v) {
                    const int64_t m = m0 + wave * 64 + r * 32 + (v & 3) + 8 * (v >> 2) + 4 * lh;
                    float o = acc[r][nt][v] + bias;
                    if (a.relu) o = fmaxf(o, 0.0f);
                    if (co_ok && m < M) a.y[m * a.cout + co] = o;
                }
        } else {
            const int Hc = a.H - (KS - 1) + 2 * PAD, Wc = a.W - (KS - 1) + 2 * PAD;   // conv output size
            // accumulator register v of a lane is pixel m = (v & 3) + 8 * (v >> 2) + 4 * (lane >> 5) of its M tile,
            // i.e. row m / TW, column m % TW of the tile rows [(wave * MT + r) * RM, +RM)
            if (POOL == 0) {
#pragma unroll
                for (int r = 0; r < MT; ++r) {
#pragma unroll
                    for (int v = 0; v < 16; ++v) {
                        const int m = (v & 3) + 8 * (v >> 2) + 4 * lh;
                        const int yy = y0 + (wave * MT + r) * RM + m / TWc, xx = x0 + m % TWc;
                        float o = acc[r][nt][v] + bias;
                        if (co_ok && yy < Hc && xx < Wc) {
                            const int64_t idx = (((int64_t)b * Hc + yy) * Wc + xx) * a.cout + co;
                            if (a.res) o += a.res[idx];
                            if (a.relu) o = fmaxf(o, 0.0f);
                            a.y[idx] = o;
                        }
                    }
                }
            } else if (TWc == 32) {
                // rows 2w and 2w+1 are the two M tiles of this wave; column pairs are adjacent registers
                const int Hp = Hc / 2, Wp = Wc / 2;
                const int yy = (y0 >> 1) + wave;
                if (POOL == 1) {
#pragma unroll
                    for (int v = 0; v < 16; v += 2) {
                        const int xx = (x0 >> 1) + (((v & 3) + 8 * (v >> 2) + 4 * lh) >> 1);
                        float o = fmaxf(fmaxf(acc[0][nt][v], acc[0][nt][v + 1]), fmaxf(acc[MT - 1][nt][v], acc[MT - 1][nt][v + 1])) + bias;
                        if (a.relu) o = fmaxf(o, 0.0f);
                        if (co_ok && yy < Hp && xx < Wp) a.y[(((int64_t)b * Hp + yy) * Wp + xx) * a.cout + co] = o;
                    }
                } else {
#pragma unroll
                    for (int v = 0; v < 16; ++v) {
                        const int xx = x0 + (v & 3) + 8 * (v >> 2) + 4 * lh;
                        float o = fmaxf(acc[0][nt][v], acc[MT - 1][nt][v]) + bias;
                        if (a.relu) o = fmaxf(o, 0.0f);
                        if (co_ok && yy < Hp && xx < Wc) a.y[(((int64_t)b * Hp + yy) * Wc + xx) * a.cout + co] = o;
                    }
                }
            } else {
                // TW = 16: an M tile holds two rows of 16 pixels; pixel m and m + 16 are registers v and v + 8
                const int Hp = Hc / 2, Wp = Wc / 2;
#pragma unroll
                for (int r = 0; r < MT; ++r) {
                    const int yy = (y0 >> 1) + wave * MT + r;
                    if (POOL == 1) {
#pragma unroll
                        for (int v = 0; v < 8; v += 2) {
                            const int m = (v & 3) + 8 * (v >> 2) + 4 * lh;
                            const int xx = (x0 >> 1) + (m >> 1);
                            float o = fmaxf(fmaxf(acc[r][nt][v], acc[r][nt][v + 1]), fmaxf(acc[r][nt][v + 8], acc[r][nt][v + 9])) + bias;
                            if (a.relu) o = fmaxf(o, 0.0f);
                            if (co_ok && yy < Hp && xx < Wp) a.y[(((int64_t)b * Hp + yy) * Wp + xx) * a.cout + co] = o;
                        }
                    } else {
#pragma unroll
                        for (int v = 0; v < 8; ++v) {
                            const int xx = x0 + (v & 3) + 8 * (v >> 2) + 4 * lh;
                            float o = fmaxf(acc[r][nt][v], acc[r][nt][v + 8]) + bias;
                            if (a.relu) o = fmaxf(o, 0.0f);
                            if (co_ok && yy < Hp && xx < Wc) a.y[(((int64_t)b * Hp + yy) * Wc + xx) * a.cout + co] = o;
                        }
                    }
                }
            }
        }
    }
}

template <int KS, int NT, int POOL, bool UPCAT, int TWc = 32, int MT = 2>
void launch_variant(const ConvArgs& a, dim3 grid, hipStream_t s) {
    hipLaunchKernelGGL((conv_mfma_f32_kernel<KS, NT, POOL, UPCAT, TWc, MT>), grid, dim3(256), 0, s, a);
}

// Tile shapes built for a (kernel size, N tile, pool) combination, and the choice among them: the best
// fill of the output map, with a penalty for the shapes that re-stage the weight slab for half the work.
struct TileShape { int tw, mt; };
inline int tile_rows(TileShape t) { return 4 * t.mt * (32 / t.tw); }

TileShape pick_tile(int ks, int NT, int pool, int Hc, int Wc) {
    static const TileShape all[4] = {{32, 2}, {16, 2}, {16, 1}, {32, 1}};
    TileShape best = all[0];
    double best_score = -1.0;
    for (const TileShape t : all) {
        if (NT != 2 && !(t.tw == 32 && t.mt == 2)) continue;            // only built for the 64-channel N tile
        if (pool != 0 && t.tw == 32 && t.mt == 1) continue;
        if (ks == 2 && t.tw == 16 && t.mt == 2) continue;
        const int th = tile_rows(t);
        const double covered = (double)((Hc + th - 1) / th * th) * ((Wc + t.tw - 1) / t.tw * t.tw);
        double score = (double)Hc * Wc / covered;
        if (t.mt == 1) score *= 0.85;
        if (t.tw == 16) score *= 0.97;
        if (score > best_score) { best_score = score; best = t; }
    }
    return best;
}

}  // namespace

// Weight slab layout: [n block][k chunk][tap][q][h][j][t] float32 with
//   output channel = nblock * BN + j,  input channel = kchunk * CK + 8q + 4h + t,  tap = ky * KS + kx,
// BN = 64 when cout is a multiple of 64 (after padding to 32), else 32.  cout is zero-padded.
int prepare_conv_mfma(ConvLayer& L, const float* w_oihw, const float* bias, int cout, int cin, int ks) {
    OCR_REQUIRE(ks == 1 || ks == 2 || ks == 3, "conv_mfma: kernel size %d not supported", ks);
    const int CK = (ks == 1) ? 32 : 16;
    OCR_REQUIRE(cin % CK == 0, "conv_mfma: cin=%d must be a multiple of %d for a %dx%d kernel", cin, CK, ks, ks);
    L.cin = cin; L.cout = cout; L.ks = ks; L.direct = false;
    const int cout32 = (cout + 31) / 32 * 32;
    const int BN = (cout32 % 64 == 0) ? 64 : 32;
    L.cout_pad = (cout32 + BN - 1) / BN * BN;
    const int taps = ks * ks, nK = cin / CK, nN = L.cout_pad / BN;
    std::vector<float> w((size_t)nN * nK * taps * CK * BN, 0.0f);
    for (int nbk = 0; nbk < nN; ++nbk)
        for (int kc = 0; kc < nK; ++kc)
            for (int tap = 0; tap < taps; ++tap)
                for (int q = 0; q < CK / 8; ++q)
                    for (int hh = 0; hh < 2; ++hh)
                        for (int j = 0; j < BN; ++j)
                            for (int t = 0; t < 4; ++t) {
                                const int o = nbk * BN + j, c = kc * CK + q * 8 + hh * 4 + t;
                                if (o >= cout) continue;
                                const size_t dst = (((((size_t)(nbk * nK + kc) * taps + tap) * (CK / 8) + q) * 2 + hh) * BN + j) * 4 + t;
                                w[dst] = w_oihw[((size_t)o * cin + c) * taps + tap];
                            }
    std::vector<float> bpad((size_t)L.cout_pad, 0.0f);
    if (bias)
        for (int o = 0; o < cout; ++o) bpad[o] = bias[o];
    OCR_TRY(L.w_f32.reserve(w.size() * sizeof(float)));
    OCR_TRY(L.bias.reserve(bpad.size() * sizeof(float)));
    OCR_HIP_CHECK(hipMemcpy(L.w_f32.ptr, w.data(), w.size() * sizeof(float), hipMemcpyHostToDevice));
    OCR_HIP_CHECK(hipMemcpy(L.bias.ptr, bpad.data(), bpad.size() * sizeof(float), hipMemcpyHostToDevice));
    if (ks == 3 && BN == 64 && cin % 16 == 0) OCR_TRY(prepare_conv_wino(L, w_oihw, cout, cin));
    if (ks == 3 && BN == 64 && cin % 16 == 0) OCR_TRY(prepare_conv_wino4(L, w_oihw, cout, cin));
    if (cin % 16 == 0) OCR_TRY(prepare_conv_bf16(L, w_oihw, cout, cin, ks));
    if (cin % 32 == 0) OCR_TRY(prepare_conv_bf16t(L, w_oihw, cout, cin, ks));
    if (ks != 3 && cin % 16 == 0) OCR_TRY(prepare_gemm_f32(L, w_oihw, bias, cout, cin, ks));
    return OCRHIP_OK;
}

int launch_conv_mfma(const ConvLayer& L, const float* x, float* y, int B, int H, int W, int pool, bool relu,
                     const float* up_src, int up_c, hipStream_t s, const float* residual) {
    OCR_REQUIRE(!L.direct && L.w_f32.ptr, "conv_mfma: layer not prepared");
    OCR_REQUIRE(x && y && B > 0 && H > 0 && W > 0, "conv_mfma: bad arguments B=%d H=%d W=%d", B, H, W);
    const int ks = L.ks;
    if (compute_bf16()) return launch_conv_bf16(L, x, y, B, H, W, pool, relu, up_src, up_c, s, residual);
    if (ks == 3 && !up_src && conv3x3_winograd() == 2 && conv_wino4_fits(L, B, H, W, pool))
        return launch_conv_wino4(L, x, y, B, H, W, pool, relu, residual, s);
    if (ks == 3 && L.w_wino.ptr && !up_src && conv3x3_winograd() && (int64_t)H * W * L.cin * 4 < (1ll << 31))
        return launch_conv_wino(L, x, y, B, H, W, pool, relu, residual, s);
    if (ks != 3 && L.w_gemm.ptr && !up_src && !residual && pool == 0 && flat_gemm()) return launch_gemm_f32(L, x, y, B, H, W, relu, s);
    const int BN = (L.cout_pad % 64 == 0) ? 64 : 32;
    const int NT = BN / 32;
    ConvArgs a;
    OCR_REQUIRE(!residual || (L.ks != 1 && pool == 0), "conv_mfma: a residual goes with an unpooled spatial layer");
    a.x = x; a.w = L.w_f32.as<float>(); a.bias = L.bias.as<float>(); a.y = y; a.up = up_src; a.res = residual;
    a.B = B; a.H = H; a.W = W; a.cin = L.cin; a.cout = L.cout; a.up_c = up_src ? up_c : 0; a.relu = relu ? 1 : 0;
    a.up_ry = 0.f; a.up_rx = 0.f; a.tiles_x = 1; a.tiles_per_image = 1; a.n_blocks_n = 1;
    dim3 grid;
    TileShape shape{32, 2};
    if (ks == 1) {
        OCR_REQUIRE(pool == 0, "conv_mfma: 1x1 layers have no fused pool");
        if (up_src) {
            OCR_REQUIRE(up_c > 0 && up_c % 32 == 0 && up_c < L.cin && H % 2 == 0 && W % 2 == 0,
                        "conv_mfma: bad upsample-concat configuration up_c=%d cin=%d H=%d W=%d", up_c, L.cin, H, W);
            a.up_ry = (H > 1) ? (float)(H / 2 - 1) / (float)(H - 1) : 0.f;
            a.up_rx = (W > 1) ? (float)(W / 2 - 1) / (float)(W - 1) : 0.f;
        }
        const int64_t M = (int64_t)B * H * W;
        const int64_t mt = (M + 255) / 256;
        OCR_REQUIRE(mt < (1ll << 31), "conv_mfma: too many rows");
        a.n_blocks_n = L.cout_pad / BN;
        OCR_REQUIRE(mt * a.n_blocks_n < (1ll << 31), "conv_mfma: too many workgroups");
        grid = dim3((unsigned)(mt * a.n_blocks_n), 1, 1);
    } else {
        OCR_REQUIRE(!up_src, "conv_mfma: upsample-concat is only fused into 1x1 layers");
        const int pad = (ks == 3) ? 1 : 0;
        const int Hc = H - (ks - 1) + 2 * pad, Wc = W - (ks - 1) + 2 * pad;
        OCR_REQUIRE(Hc > 0 && Wc > 0, "conv_mfma: input %dx%d too small for a %dx%d kernel", H, W, ks, ks);
        if (pool) OCR_REQUIRE(Hc % 2 == 0 && (pool == 2 || Wc % 2 == 0), "conv_mfma: pooled layer needs even size, got %dx%d", Hc, Wc);
        shape = pick_tile(ks, NT, pool, Hc, Wc);
        a.tiles_x = (Wc + shape.tw - 1) / shape.tw;
        const int tiles_y = (Hc + tile_rows(shape) - 1) / tile_rows(shape);
        a.tiles_per_image = a.tiles_x * tiles_y;
        a.n_blocks_n = L.cout_pad / BN;
        const int64_t wgs = (int64_t)a.tiles_per_image * a.n_blocks_n * B;
        OCR_REQUIRE(wgs < (1ll << 31), "conv_mfma: too many workgroups");
        grid = dim3((unsigned)wgs, 1, 1);
    }

    char pname[64];
    std::snprintf(pname, sizeof(pname), "conv_mfma_f32<ks%d,n%d,pool%d,up%d,tile%dx%d>", ks, BN, pool, (int)(up_src != nullptr),
                  ks == 1 ? 256 : tile_rows(shape), ks == 1 ? 1 : shape.tw);
    double out_px, in_elems;
    if (ks == 1) {
        out_px = (double)B * H * W;
        in_elems = out_px * L.cin;
    } else {
        const int pad = (ks == 3) ? 1 : 0;
        out_px = (double)B * (H - (ks - 1) + 2 * pad) * (W - (ks - 1) + 2 * pad);
        in_elems = (double)B * H * W * L.cin;
    }
    const double out_elems = out_px * L.cout / (pool == 1 ? 4.0 : (pool == 2 ? 2.0 : 1.0));
    ProfScope prof(pname, 2.0 * ks * ks * L.cin * L.cout * out_px,
                   4.0 * (in_elems + out_elems * (residual ? 2.0 : 1.0) + (double)ks * ks * L.cin * L.cout), s);

#define OCR_CONV_CASE(KS_, NT_, POOL_, UP_, TW_, MT_)                                            \
    if (ks == KS_ && NT == NT_ && pool == POOL_ && (up_src != nullptr) == UP_ && shape.tw == TW_ && \
        shape.mt == MT_) {                                                                          \
        launch_variant<KS_, NT_, POOL_, UP_, TW_, MT_>(a, grid, s);                                  \
        OCR_HIP_CHECK(hipGetLastError());                                                           \
        return OCRHIP_OK;                                                                           \
    }
    OCR_CONV_CASE(3, 2, 0, false, 32, 2)
    OCR_CONV_CASE(3, 2, 0, false, 16, 2)
    OCR_CONV_CASE(3, 2, 0, false, 16, 1)
    OCR_CONV_CASE(3, 2, 0, false, 32, 1)
    OCR_CONV_CASE(3, 2, 1, false, 32, 2)
    OCR_CONV_CASE(3, 2, 1, false, 16, 2)
    OCR_CONV_CASE(3, 2, 1, false, 16, 1)
    OCR_CONV_CASE(3, 2, 2, false, 32, 2)
    OCR_CONV_CASE(3, 2, 2, false, 16, 2)
    OCR_CONV_CASE(3, 2, 2, false, 16, 1)
    OCR_CONV_CASE(3, 1, 0, false, 32, 2)
    OCR_CONV_CASE(2, 2, 0, false, 32, 2)
    OCR_CONV_CASE(2, 2, 0, false, 32, 1)
    OCR_CONV_CASE(2, 2, 0, false, 16, 1)
    OCR_CONV_CASE(1, 2, 0, false, 32, 2)
    OCR_CONV_CASE(1, 1, 0, false, 32, 2)
    OCR_CONV_CASE(1, 2, 0, true, 32, 2)
    OCR_CONV_CASE(1, 1, 0, true, 32, 2)
#undef OCR_CONV_CASE
    set_error("conv_mfma: no kernel variant for ks=%d NT=%d pool=%d upcat=%d", ks, NT, pool, (int)(up_src != nullptr));
    return OCRHIP_EINVAL;
}

}  // namespace ocrhip
