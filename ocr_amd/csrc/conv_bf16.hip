// bfloat16 variant of the implicit-GEMM convolution / GEMM (v_mfma_f32_32x32x16_bf16), NHWC float32 tensors.
//
// Same layers, tiling (8 rows x 32 columns of output pixels or 256 matrix rows, times 32 or 64 output channels per
// 4-wave workgroup), epilogue (folded batch-norm bias, ReLU, fused max-pool, residual) and fused x2 align-corners
// upsample + concat as conv_mfma.hip.  Only the multiplication differs: activations and weights are rounded to
// bfloat16 (nearest even) on their way into LDS, products are accumulated in float32, and activations stay float32
// in HBM.  This is the reduced-precision option of the path (`compute_dtype = OCRHIP_BF16`; the reference runs in
// float32, architectures/east_arch.py:95-175 and architectures/crnn_arch.py:427-531 under Keras' default policy);
// its outputs are compared with the float32 oracle at 1e-2.
//
// K: chunks of 16 input channels = the K of one MFMA.  A lane's operand is 8 consecutive channels (16 bytes) of its
// pixel (A) or output channel (B): channels 8h..8h+7 for lane half h.  LDS pixel stride 48 bytes: the 16 lanes of a
// ds_read_b128 phase hit 16 different 16-byte slots.  The weight image is lane-linear.
//
// Roofline: with 16x less MFMA time per chunk than the float32 kernel the staging (global -> registers -> bf16 -> LDS)
// dominates; the kernel is LDS / issue bound, far from the 2.5 PFLOP/s bf16 MFMA peak.
#include <hip/hip_bf16.h>

#include "common.h"

namespace ocrhip {

namespace {

typedef float floatx16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

struct ConvArgsB {
    const float* x;             // (B,H,W,cin_x) NHWC, or (M,cin) for KS == 1
    const uint16_t* w;          // bf16 slab, see prepare_conv_bf16
    const float* bias;          // cout_pad
    float* y;
    const float* up;            // UPCAT: (B,H/2,W/2,up_c)
    const float* res;           // optional residual (unpooled spatial layers)
    int B, H, W;
    int cin, cout, up_c, relu;
    float up_ry, up_rx;
    int tiles_x, tiles_per_image, n_blocks_n;
};

__device__ __forceinline__ uint2 pack_bf16x4(float4 v) {
    const __hip_bfloat16 a = __float2bfloat16(v.x), b = __float2bfloat16(v.y), c = __float2bfloat16(v.z), d = __float2bfloat16(v.w);
    const uint16_t ua = *reinterpret_cast<const uint16_t*>(&a), ub = *reinterpret_cast<const uint16_t*>(&b);
    const uint16_t uc = *reinterpret_cast<const uint16_t*>(&c), ud = *reinterpret_cast<const uint16_t*>(&d);
    return make_uint2((uint32_t)ua | ((uint32_t)ub << 16), (uint32_t)uc | ((uint32_t)ud << 16));
}

template <int KS, int NT, int POOL, bool UPCAT>
__global__ __launch_bounds__(256, 2) void conv_bf16_kernel(const ConvArgsB a) {
    constexpr int CK = 16, TAPS = KS * KS, MT = 2;
    constexpr int TH = 8, TW = 32, HT = TH + KS - 1, WT = TW + KS - 1;
    constexpr int NPIX = (KS == 1) ? 256 : HT * WT;
    constexpr int PAD = (KS == 3) ? 1 : 0;
    constexpr int BN = NT * 32;
    constexpr int PSB = 48;                                   // bytes per pixel in LDS (32 + pad)
    constexpr int B_VEC = TAPS * 2 * BN;                      // 16-byte vectors of the weight image
    __shared__ __attribute__((aligned(16))) unsigned char As[NPIX * PSB];
    __shared__ __attribute__((aligned(16))) uint4 Bs[B_VEC];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int li = lane & 31, lh = lane >> 5;

    const unsigned total = gridDim.x, xcd = blockIdx.x & 7u, slot = blockIdx.x >> 3;
    const unsigned work = xcd * (total >> 3) + min(xcd, total & 7u) + slot;
    const int nb = (int)(work % (unsigned)a.n_blocks_n);
    const int tile_all = (int)(work / (unsigned)a.n_blocks_n);
    const int nK = a.cin / CK;

    int b = 0, y0 = 0, x0 = 0;
    int64_t m0 = 0;
    if (KS == 1) {
        m0 = (int64_t)tile_all * 256;
    } else {
        b = tile_all / a.tiles_per_image;
        const int tile = tile_all % a.tiles_per_image;
        y0 = (tile / a.tiles_x) * TH;
        x0 = (tile % a.tiles_x) * TW;
    }
    const int64_t M = (int64_t)a.B * a.H * a.W;
    const int cin_x = a.cin - a.up_c;

    floatx16 acc[MT][NT];
#pragma unroll
    for (int r = 0; r < MT; ++r)
#pragma unroll
        for (int nt = 0; nt < NT; ++nt)
#pragma unroll
            for (int v = 0; v < 16; ++v) acc[r][nt][v] = 0.0f;

    const uint4* wslab = reinterpret_cast<const uint4*>(a.w) + (int64_t)nb * nK * B_VEC;

    constexpr int A_V4 = NPIX * 4;                            // float4 items of the input tile (4 per pixel)
    constexpr int A_IT = (A_V4 + 255) / 256, B_IT = (B_VEC + 255) / 256;
    float4 ra[A_IT];
    uint4 rb[B_IT];

    auto load_chunk = [&](int kc) {
        const int c0 = kc * CK;
        const uint4* src = wslab + (int64_t)kc * B_VEC;
#pragma unroll
        for (int i = 0; i < B_IT; ++i) {
            const int f = tid + i * 256;
            rb[i] = (B_VEC % 256 == 0 || f < B_VEC) ? src[f] : make_uint4(0, 0, 0, 0);
        }
        if (KS == 1) {
            const bool from_up = UPCAT && (c0 < a.up_c);
#pragma unroll
            for (int i = 0; i < A_IT; ++i) {
                const int f = tid + i * 256;
                const int px = f >> 2, c4 = f & 3;
                const int64_t m = m0 + px;
                float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
                if (m < M) {
                    if (!from_up) {
                        v = *reinterpret_cast<const float4*>(a.x + m * cin_x + (c0 - a.up_c) + c4 * 4);
                    } else {
                        // x2 bilinear upsampling with aligned corners of the (H/2, W/2) source, in float32
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
                const int px = f >> 2, c4 = f & 3;
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
            if (B_VEC % 256 == 0 || f < B_VEC) Bs[f] = rb[i];
        }
#pragma unroll
        for (int i = 0; i < A_IT; ++i) {
            const int f = tid + i * 256;
            if (A_V4 % 256 == 0 || f < A_V4) *reinterpret_cast<uint2*>(&As[(f >> 2) * PSB + (f & 3) * 8]) = pack_bf16x4(ra[i]);
        }
    };

    load_chunk(0);
    store_chunk();
    __syncthreads();

    for (int kc = 0; kc < nK; ++kc) {
        if (kc + 1 < nK) load_chunk(kc + 1);
#pragma unroll
        for (int tap = 0; tap < TAPS; ++tap) {
            const int ky = tap / KS, kx = tap % KS;
            bf16x8 av[MT], bv[NT];
#pragma unroll
            for (int r = 0; r < MT; ++r) {
                const int px = (KS == 1) ? (wave * 64 + r * 32 + li) : ((wave * MT + r + ky) * WT + (li + kx));
                av[r] = *reinterpret_cast<const bf16x8*>(&As[px * PSB + lh * 16]);
            }
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) bv[nt] = *reinterpret_cast<const bf16x8*>(&Bs[(tap * 2 + lh) * BN + nt * 32 + li]);
#pragma unroll
            for (int r = 0; r < MT; ++r)
#pragma unroll
                for (int nt = 0; nt < NT; ++nt)
                    acc[r][nt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(av[r], bv[nt], acc[r][nt], 0, 0, 0);
        }
        if (kc + 1 < nK) {
            __syncthreads();
            store_chunk();
            __syncthreads();
        }
    }

    // ---- epilogue (as in conv_mfma.hip): register v of a lane = output channel (lane & 31) of pixel column
    // (v & 3) + 8 * (v >> 2) + 4 * (lane >> 5) of its 32-pixel M tile; the wave's two M tiles are rows 2w and 2w + 1
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
        const int co = nb * BN + nt * 32 + li;
        const bool co_ok = co < a.cout;
        const float bias = a.bias[co];
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
            const int Hc = a.H - (KS - 1) + 2 * PAD, Wc = a.W - (KS - 1) + 2 * PAD;
            if (POOL == 0) {
#pragma unroll
                for (int r = 0; r < MT; ++r)
#pragma unroll
                    for (int v = 0; v < 16; ++v) {
                        const int yy = y0 + wave * MT + r, xx = x0 + (v & 3) + 8 * (v >> 2) + 4 * lh;
                        float o = acc[r][nt][v] + bias;
                        if (co_ok && yy < Hc && xx < Wc) {
                            const int64_t idx = (((int64_t)b * Hc + yy) * Wc + xx) * a.cout + co;
                            if (a.res) o += a.res[idx];
                            if (a.relu) o = fmaxf(o, 0.0f);
                            a.y[idx] = o;
                        }
                    }
            } else {
                const int Hp = Hc / 2, Wp = Wc / 2;
                const int yy = (y0 >> 1) + wave;
                if (POOL == 1) {
#pragma unroll
                    for (int v = 0; v < 16; v += 2) {
                        const int xx = (x0 >> 1) + (((v & 3) + 8 * (v >> 2) + 4 * lh) >> 1);
                        float o = fmaxf(fmaxf(acc[0][nt][v], acc[0][nt][v + 1]), fmaxf(acc[1][nt][v], acc[1][nt][v + 1])) + bias;
                        if (a.relu) o = fmaxf(o, 0.0f);
                        if (co_ok && yy < Hp && xx < Wp) a.y[(((int64_t)b * Hp + yy) * Wp + xx) * a.cout + co] = o;
                    }
                } else {
#pragma unroll
                    for (int v = 0; v < 16; ++v) {
                        const int xx = x0 + (v & 3) + 8 * (v >> 2) + 4 * lh;
                        float o = fmaxf(acc[0][nt][v], acc[1][nt][v]) + bias;
                        if (a.relu) o = fmaxf(o, 0.0f);
                        if (co_ok && yy < Hp && xx < Wc) a.y[(((int64_t)b * Hp + yy) * Wc + xx) * a.cout + co] = o;
                    }
                }
            }
        }
    }
}

template <int KS, int NT, int POOL, bool UPCAT>
void launch_variant_b(const ConvArgsB& a, dim3 grid, hipStream_t s) {
    hipLaunchKernelGGL((conv_bf16_kernel<KS, NT, POOL, UPCAT>), grid, dim3(256), 0, s, a);
}

}  // namespace

// bf16 weight slab: [n block][k chunk (16)][tap][h][j][8] with output channel = nblock * BN + j and input channel =
// kchunk * 16 + 8h + e; BN as in prepare_conv_mfma (L.cout_pad is already set).
int prepare_conv_bf16(ConvLayer& L, const float* w_oihw, int cout, int cin, int ks) {
    OCR_REQUIRE(cin % 16 == 0, "conv_bf16: cin=%d must be a multiple of 16", cin);
    const int BN = (L.cout_pad % 64 == 0) ? 64 : 32;
    const int taps = ks * ks, nK = cin / 16, nN = L.cout_pad / BN;
    std::vector<uint16_t> w((size_t)nN * nK * taps * 2 * BN * 8, 0);
    for (int o = 0; o < cout; ++o)
        for (int c = 0; c < cin; ++c)
            for (int tap = 0; tap < taps; ++tap) {
                const int nbk = o / BN, j = o % BN, kc = c / 16, hh = (c % 16) / 8, e = c % 8;
                const __hip_bfloat16 v = __float2bfloat16(w_oihw[((size_t)o * cin + c) * taps + tap]);
                w[(((((size_t)(nbk * nK + kc) * taps + tap) * 2 + hh) * BN + j) * 8) + e] = *reinterpret_cast<const uint16_t*>(&v);
            }
    OCR_TRY(L.w_bf16.reserve(w.size() * sizeof(uint16_t)));
    OCR_HIP_CHECK(hipMemcpy(L.w_bf16.ptr, w.data(), w.size() * sizeof(uint16_t), hipMemcpyHostToDevice));
    return OCRHIP_OK;
}

int launch_conv_bf16(const ConvLayer& L, const float* x, float* y, int B, int H, int W, int pool, bool relu,
                     const float* up_src, int up_c, hipStream_t s, const float* residual) {
    OCR_REQUIRE(!L.direct && L.w_bf16.ptr, "conv_bf16: layer has no bfloat16 weights");
    OCR_REQUIRE(x && y && B > 0 && H > 0 && W > 0, "conv_bf16: bad arguments B=%d H=%d W=%d", B, H, W);
    const int ks = L.ks;
    const int BN = (L.cout_pad % 64 == 0) ? 64 : 32;
    const int NT = BN / 32;
    OCR_REQUIRE(!residual || (ks != 1 && pool == 0), "conv_bf16: a residual goes with an unpooled spatial layer");
    ConvArgsB a;
    a.x = x; a.w = L.w_bf16.as<uint16_t>(); a.bias = L.bias.as<float>(); a.y = y; a.up = up_src; a.res = residual;
    a.B = B; a.H = H; a.W = W; a.cin = L.cin; a.cout = L.cout; a.up_c = up_src ? up_c : 0; a.relu = relu ? 1 : 0;
    a.up_ry = 0.f; a.up_rx = 0.f; a.tiles_x = 1; a.tiles_per_image = 1; a.n_blocks_n = L.cout_pad / BN;
    int64_t wgs;
    double out_px, in_elems;
    if (ks == 1) {
        OCR_REQUIRE(pool == 0, "conv_bf16: 1x1 layers have no fused pool");
        if (up_src) {
            OCR_REQUIRE(up_c > 0 && up_c % 16 == 0 && up_c < L.cin && H % 2 == 0 && W % 2 == 0,
                        "conv_bf16: bad upsample-concat configuration up_c=%d cin=%d H=%d W=%d", up_c, L.cin, H, W);
            a.up_ry = (H > 1) ? (float)(H / 2 - 1) / (float)(H - 1) : 0.f;
            a.up_rx = (W > 1) ? (float)(W / 2 - 1) / (float)(W - 1) : 0.f;
        }
        const int64_t M = (int64_t)B * H * W;
        wgs = (M + 255) / 256 * a.n_blocks_n;
        out_px = (double)M;
        in_elems = out_px * L.cin;
    } else {
        OCR_REQUIRE(!up_src, "conv_bf16: upsample-concat is only fused into 1x1 layers");
        const int pad = (ks == 3) ? 1 : 0;
        const int Hc = H - (ks - 1) + 2 * pad, Wc = W - (ks - 1) + 2 * pad;
        OCR_REQUIRE(Hc > 0 && Wc > 0, "conv_bf16: input %dx%d too small for a %dx%d kernel", H, W, ks, ks);
        if (pool) OCR_REQUIRE(Hc % 2 == 0 && (pool == 2 || Wc % 2 == 0), "conv_bf16: pooled layer needs even size, got %dx%d", Hc, Wc);
        a.tiles_x = (Wc + 31) / 32;
        a.tiles_per_image = a.tiles_x * ((Hc + 7) / 8);
        wgs = (int64_t)a.tiles_per_image * a.n_blocks_n * B;
        out_px = (double)B * Hc * Wc;
        in_elems = (double)B * H * W * L.cin;
    }
    OCR_REQUIRE(wgs < (1ll << 31), "conv_bf16: too many workgroups");
    const dim3 grid((unsigned)wgs);
    char pname[64];
    std::snprintf(pname, sizeof(pname), "conv_bf16<ks%d,n%d,pool%d,up%d>", ks, BN, pool, (int)(up_src != nullptr));
    const double out_elems = out_px * L.cout / (pool == 1 ? 4.0 : (pool == 2 ? 2.0 : 1.0));
    ProfScope prof(pname, 2.0 * ks * ks * L.cin * L.cout * out_px,
                   4.0 * (in_elems + out_elems * (residual ? 2.0 : 1.0)) + 2.0 * ks * ks * L.cin * L.cout, s);
#define OCR_CONVB_CASE(KS_, NT_, POOL_, UP_)                                             \
    if (ks == KS_ && NT == NT_ && pool == POOL_ && (up_src != nullptr) == UP_) {        \
        launch_variant_b<KS_, NT_, POOL_, UP_>(a, grid, s);                              \
        OCR_HIP_CHECK(hipGetLastError());                                                \
        return OCRHIP_OK;                                                                \
    }
    OCR_CONVB_CASE(3, 2, 0, false)
    OCR_CONVB_CASE(3, 2, 1, false)
    OCR_CONVB_CASE(3, 2, 2, false)
    OCR_CONVB_CASE(3, 1, 0, false)
    OCR_CONVB_CASE(2, 2, 0, false)
    OCR_CONVB_CASE(1, 2, 0, false)
    OCR_CONVB_CASE(1, 1, 0, false)
    OCR_CONVB_CASE(1, 2, 0, true)
    OCR_CONVB_CASE(1, 1, 0, true)
#undef OCR_CONVB_CASE
    set_error("conv_bf16: no kernel variant for ks=%d NT=%d pool=%d upcat=%d", ks, NT, pool, (int)(up_src != nullptr));
    return OCRHIP_EINVAL;
}

}  // namespace ocrhip
