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
//   (16 coalesced 16-byte loads per chunk, issued before the data transform of that chunk).
//
// Roofline: MFMA bound; the unit of work is the direct convolution's 2*9*Cin*Cout flop per output pixel, of
// which 16/36 are executed.
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
};

template <int POOL>
__global__ __launch_bounds__(256, 1) void conv_wino_f32_kernel(const WinoArgs a) {
    __shared__ __attribute__((aligned(16))) float U[16 * 64 * kUS];       // 48 KB
    __shared__ __attribute__((aligned(16))) float Rw[kRawPx * kRS];       // 16 KB

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int li = lane & 31, lh = lane >> 5;
    const int th = wave & 1;          // which 2 of the 4 block rows
    const int ch = wave >> 1;         // which 32 of the 64 output channels

    // every XCD walks a contiguous range of the work index (see conv_mfma.hip)
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

    float4 ra[3];
    auto load_raw = [&](int kc) {
#pragma unroll
        for (int i = 0; i < 3; ++i) {
            const int f = tid + i * 256;
            const int px = f >> 1, c4 = f & 1;
            const int gy = y0 - 1 + px / kRawW, gx = x0 - 1 + px % kRawW;
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (f < kRawPx * 2 && gy >= 0 && gy < a.H && gx >= 0 && gx < a.W)
                v = *reinterpret_cast<const float4*>(a.x + (((int64_t)b * a.H + gy) * a.W + gx) * a.cin + kc * kCK + c4 * 4);
            ra[i] = v;
        }
    };
    auto store_raw = [&]() {
#pragma unroll
        for (int i = 0; i < 3; ++i) {
            const int f = tid + i * 256;
            if (f < kRawPx * 2) *reinterpret_cast<float4*>(&Rw[(f >> 1) * kRS + (f & 1) * 4]) = ra[i];
        }
    };
    // data transform of one (block, 4 channels, half of the component rows) per thread
    const int t_cq = tid & 1, t_tx = (tid >> 1) & 15, t_half = (tid >> 5) & 1, t_ty = tid >> 6;
    auto transform = [&]() {
        float4 d[3][4];
#pragma unroll
        for (int r = 0; r < 3; ++r)
#pragma unroll
            for (int c = 0; c < 4; ++c)
                d[r][c] = *reinterpret_cast<const float4*>(&Rw[((2 * t_ty + t_half + r) * kRawW + 2 * t_tx + c) * kRS + t_cq * 4]);
        // rows of B^T d: half 0 -> i = 0: d0 - d2, i = 1: d1 + d2;  half 1 (rows start at d1) -> i = 2: d2 - d1, i = 3: d1 - d3
#pragma unroll
        for (int ii = 0; ii < 2; ++ii) {
            float4 t[4];
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                const float4 p = d[0][c], q = d[1][c], r = d[2][c];
                float4 o;
                if (t_half == 0) {
                    if (ii == 0) { o.x = p.x - r.x; o.y = p.y - r.y; o.z = p.z - r.z; o.w = p.w - r.w; }
                    else         { o.x = q.x + r.x; o.y = q.y + r.y; o.z = q.z + r.z; o.w = q.w + r.w; }
                } else {
                    if (ii == 0) { o.x = q.x - p.x; o.y = q.y - p.y; o.z = q.z - p.z; o.w = q.w - p.w; }
                    else         { o.x = p.x - r.x; o.y = p.y - r.y; o.z = p.z - r.z; o.w = p.w - r.w; }
                }
                t[c] = o;
            }
            const int i = t_half * 2 + ii;
            float4 u[4];
            u[0] = make_float4(t[0].x - t[2].x, t[0].y - t[2].y, t[0].z - t[2].z, t[0].w - t[2].w);
            u[1] = make_float4(t[1].x + t[2].x, t[1].y + t[2].y, t[1].z + t[2].z, t[1].w + t[2].w);
            u[2] = make_float4(t[2].x - t[1].x, t[2].y - t[1].y, t[2].z - t[1].z, t[2].w - t[1].w);
            u[3] = make_float4(t[1].x - t[3].x, t[1].y - t[3].y, t[1].z - t[3].z, t[1].w - t[3].w);
#pragma unroll
            for (int j = 0; j < 4; ++j)
                *reinterpret_cast<float4*>(&U[((i * 4 + j) * 64 + t_ty * 16 + t_tx) * kUS + t_cq * 4]) = u[j];
        }
    };

    float4 bw[16];
    auto load_b = [&](int kc) {
        const float4* p = wlane + (int64_t)kc * (16 * 2 * 64);
#pragma unroll
        for (int f = 0; f < 16; ++f) bw[f] = p[f * 128];
    };
    auto mfma_phase = [&]() {
#pragma unroll
        for (int fg = 0; fg < 4; ++fg) {
            float4 af[4];
#pragma unroll
            for (int i = 0; i < 4; ++i)
                af[i] = *reinterpret_cast<const float4*>(&U[((fg * 4 + i) * 64 + th * 32 + li) * kUS + lh * 4]);
#pragma unroll
            for (int t = 0; t < 4; ++t)
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const int f = fg * 4 + i;
                    const float av = t == 0 ? af[i].x : (t == 1 ? af[i].y : (t == 2 ? af[i].z : af[i].w));
                    const float bv = t == 0 ? bw[f].x : (t == 1 ? bw[f].y : (t == 2 ? bw[f].z : bw[f].w));
                    acc[f] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bv, acc[f], 0, 0, 0);
                }
        }
    };

    load_raw(0);
    load_b(0);
    store_raw();
    __syncthreads();
    transform();
    __syncthreads();

    for (int kc = 0; kc < nK; ++kc) {
        if (kc + 1 < nK) load_raw(kc + 1);          // lands during the matrix work
        mfma_phase();
        if (kc + 1 < nK) {
            load_b(kc + 1);                         // lands during the data transform
            store_raw();
            __syncthreads();                        // U of chunk kc fully read, raw tile of chunk kc + 1 visible
            transform();
            __syncthreads();
        }
    }

    // ---- inverse transform, bias, ReLU, optional pool, store.  Register v of a lane belongs to block
    // m = (v & 3) + 8 * (v >> 2) + 4 * (lane >> 5) of the wave's 32 (2 block rows x 16 block columns) and to
    // output channel lane & 31 of the wave's 32.
    const int co = nb * 64 + ch * 32 + li;
    const bool co_ok = co < a.cout;
    const float bias = a.bias[co];
    const int Hc = a.H, Wc = a.W;
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
        const int yy = y0 + 2 * ty, xx = x0 + 2 * tx;
        if (!co_ok) continue;
        if (POOL == 0) {
#pragma unroll
            for (int r = 0; r < 2; ++r)
#pragma unroll
                for (int c = 0; c < 2; ++c) {
                    if (yy + r < Hc && xx + c < Wc) {
                        const int64_t idx = (((int64_t)b * Hc + yy + r) * Wc + xx + c) * a.cout + co;
                        float val = o[r][c] + bias;
                        if (a.res) val += a.res[idx];
                        if (a.relu) val = fmaxf(val, 0.0f);
                        a.y[idx] = val;
                    }
                }
        } else if (POOL == 1) {
            const int Hp = Hc / 2, Wp = Wc / 2;
            if ((yy >> 1) < Hp && (xx >> 1) < Wp) {
                float val = fmaxf(fmaxf(o[0][0], o[0][1]), fmaxf(o[1][0], o[1][1])) + bias;
                if (a.relu) val = fmaxf(val, 0.0f);
                a.y[(((int64_t)b * Hp + (yy >> 1)) * Wp + (xx >> 1)) * a.cout + co] = val;
            }
        } else {
            const int Hp = Hc / 2;
#pragma unroll
            for (int c = 0; c < 2; ++c) {
                if ((yy >> 1) < Hp && xx + c < Wc) {
                    float val = fmaxf(o[0][c], o[1][c]) + bias;
                    if (a.relu) val = fmaxf(val, 0.0f);
                    a.y[(((int64_t)b * Hp + (yy >> 1)) * Wc + xx + c) * a.cout + co] = val;
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
    a.tiles_per_image = a.tiles_x * ((H + 7) / 8);
    a.n_blocks_n = L.cout_pad / 64;
    const int64_t wgs = (int64_t)a.tiles_per_image * a.n_blocks_n * B;
    OCR_REQUIRE(wgs < (1ll << 31), "conv_wino: too many workgroups");
    char pname[64];
    std::snprintf(pname, sizeof(pname), "conv_wino_f32<n64,pool%d,tile8x32>", pool);
    const double out_px = (double)B * H * W;
    const double out_elems = out_px * L.cout / (pool == 1 ? 4.0 : (pool == 2 ? 2.0 : 1.0));
    ProfScope prof(pname, 2.0 * 9 * L.cin * L.cout * out_px,
                   4.0 * (out_px * L.cin + out_elems * (residual ? 2.0 : 1.0) + 9.0 * L.cin * L.cout), s);
    const dim3 grid((unsigned)wgs), block(256);
    if (pool == 0)
        hipLaunchKernelGGL(conv_wino_f32_kernel<0>, grid, block, 0, s, a);
    else if (pool == 1)
        hipLaunchKernelGGL(conv_wino_f32_kernel<1>, grid, block, 0, s, a);
    else
        hipLaunchKernelGGL(conv_wino_f32_kernel<2>, grid, block, 0, s, a);
    OCR_HIP_CHECK(hipGetLastError());
    return OCRHIP_OK;
}

}  // namespace ocrhip
