// C ABI of libocrhip.so: handle lifetime, weight loading (batch-norm folding, device layouts) and
// the two network graphs.  The graphs follow architectures/east_arch.py:95-175 and
// architectures/crnn_arch.py:38-41,306-328,427-531 layer by layer; each layer is one launch of a
// kernel from conv_direct.hip / conv_mfma.hip / rnn.hip.
#include <algorithm>
#include <cmath>
#include <cstring>

#include "common.h"

namespace ocrhip {

static thread_local char g_error[512] = "";

void set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_error, sizeof(g_error), fmt, ap);
    va_end(ap);
}

static thread_local Profiler* g_prof = nullptr;
Profiler* current_profiler() { return g_prof; }
void set_current_profiler(Profiler* p) { g_prof = p; }

static thread_local int g_compute_bf16 = 0;
int compute_bf16() { return g_compute_bf16; }
void set_compute_bf16(int on) { g_compute_bf16 = on; }
// sets the bfloat16 switch for the duration of one API call
struct Bf16Scope {
    explicit Bf16Scope(int dtype) { set_compute_bf16(dtype == OCRHIP_BF16 ? 1 : 0); }
    ~Bf16Scope() { set_compute_bf16(0); }
};
static thread_local int g_lstm_persistent = 2;
int lstm_persistent() { return g_lstm_persistent; }
void set_lstm_persistent(int on) { g_lstm_persistent = on; }
static thread_local int g_lstm16_units = 64;
int lstm16_units() { return g_lstm16_units; }
void set_lstm16_units(int u) { g_lstm16_units = u; }
static thread_local int g_lstm_debug_drop = 0;
int lstm_debug_drop() { return g_lstm_debug_drop; }
void set_lstm_debug_drop(int on) { g_lstm_debug_drop = on; }
static thread_local int g_conv3x3_winograd = 2;
int conv3x3_winograd() { return g_conv3x3_winograd; }
void set_conv3x3_winograd(int on) { g_conv3x3_winograd = on; }
static thread_local DevBuf* g_workspace = nullptr;
DevBuf* current_workspace() { return g_workspace; }
void set_current_workspace(DevBuf* ws) { g_workspace = ws; }
static thread_local int g_wino4_split = 0;
int wino4_split() { return g_wino4_split; }
void set_wino4_split(int on) { g_wino4_split = on; }
static thread_local int g_wino4_side_by_side = 1;
int wino4_side_by_side_enabled() { return g_wino4_side_by_side; }
void set_wino4_side_by_side(int on) { g_wino4_side_by_side = on; }
static thread_local int g_fuse_ctc = 1;
static thread_local int g_fuse_mean_rows = 1;
static thread_local int g_wino4_wave_tiles = 2;
int wino4_wave_tiles() { return g_wino4_wave_tiles; }
void set_wino4_wave_tiles(int on) { g_wino4_wave_tiles = on; }
static thread_local int g_first_layer_mfma = 1;
int first_layer_mfma() { return g_first_layer_mfma; }
void set_first_layer_mfma(int on) { g_first_layer_mfma = on; }
static thread_local int g_flat_gemm = 1;
int flat_gemm() { return g_flat_gemm; }
void set_flat_gemm(int on) { g_flat_gemm = on; }

hipEvent_t Profiler::get_event() {
    if (!pool.empty()) {
        hipEvent_t e = pool.back();
        pool.pop_back();
        return e;
    }
    hipEvent_t e = nullptr;
    if (hipEventCreate(&e) != hipSuccess) return nullptr;
    return e;
}

void Profiler::recycle() {
    for (auto& r : records) {
        if (r.e0) pool.push_back(r.e0);
        if (r.e1) pool.push_back(r.e1);
    }
    records.clear();
}

ProfScope::ProfScope(const char* name, double flops, double bytes, hipStream_t stream) : p(current_profiler()), s(stream), idx(0) {
    if (!p || !p->enabled || (!p->only.empty() && p->only != name)) {
        p = nullptr;
        return;
    }
    ProfRecord r;
    r.name = name;
    r.flops = flops;
    r.bytes = bytes;
    r.e0 = p->get_event();
    r.e1 = p->get_event();
    if (!r.e0 || !r.e1) {
        p = nullptr;
        return;
    }
    (void)hipEventRecord(r.e0, s);
    idx = p->records.size();
    p->records.push_back(r);
}

ProfScope::~ProfScope() {
    if (p) (void)hipEventRecord(p->records[idx].e1, s);
}

int DevBuf::reserve(size_t need) {
    if (need <= bytes && ptr) return OCRHIP_OK;
    if (ptr) {
        OCR_HIP_CHECK(hipFree(ptr));
        ptr = nullptr;
        bytes = 0;
    }
    if (need == 0) need = 256;
    OCR_HIP_CHECK(hipMalloc(&ptr, need));
    bytes = need;
    return OCRHIP_OK;
}

int PinnedRing::upload(void* dst, const void* src, size_t n, hipStream_t s) {
    const int k = next;
    next = (next + 1) % kSlots;
    if (done[k]) OCR_HIP_CHECK(hipEventSynchronize(done[k]));          // the copy that last used this slot has finished
    else OCR_HIP_CHECK(hipEventCreateWithFlags(&done[k], hipEventDisableTiming));
    if (bytes[k] < n) {
        if (host[k]) OCR_HIP_CHECK(hipHostFree(host[k]));
        host[k] = nullptr;
        bytes[k] = 0;
        OCR_HIP_CHECK(hipHostMalloc(&host[k], n, hipHostMallocDefault));
        bytes[k] = n;
    }
    std::memcpy(host[k], src, n);
    OCR_HIP_CHECK(hipMemcpyAsync(dst, host[k], n, hipMemcpyHostToDevice, s));
    OCR_HIP_CHECK(hipEventRecord(done[k], s));
    return OCRHIP_OK;
}

void PinnedRing::release() {
    for (int k = 0; k < kSlots; ++k) {
        if (host[k]) (void)hipHostFree(host[k]);
        if (done[k]) (void)hipEventDestroy(done[k]);
        host[k] = nullptr; done[k] = nullptr; bytes[k] = 0;
    }
}

void DevBuf::release() {
    if (ptr) (void)hipFree(ptr);
    ptr = nullptr;
    bytes = 0;
}

namespace {

constexpr double kBnEps = 1e-5;

struct Weights {
    const ocrhip_tensor_desc* w;
    int n;
    const ocrhip_tensor_desc* find(const std::string& name) const {
        for (int i = 0; i < n; ++i)
            if (w[i].name && name == w[i].name) return &w[i];
        return nullptr;
    }
};

int64_t numel(const ocrhip_tensor_desc& t) {
    int64_t k = 1;
    for (int i = 0; i < t.ndim; ++i) k *= t.shape[i];
    return k;
}

int get(const Weights& W, const std::string& name, std::initializer_list<int64_t> shape, const float** out) {
    const ocrhip_tensor_desc* t = W.find(name);
    if (!t || !t->data) {
        set_error("load_weights: tensor '%s' is missing", name.c_str());
        return OCRHIP_EINVAL;
    }
    int64_t want = 1;
    for (int64_t s : shape) want *= s;
    if (numel(*t) != want) {
        set_error("load_weights: tensor '%s' has %lld elements, expected %lld", name.c_str(), (long long)numel(*t),
                  (long long)want);
        return OCRHIP_EINVAL;
    }
    *out = t->data;
    return OCRHIP_OK;
}

// conv (+ optional bias) followed by inference batch-norm -> folded weight / bias
int fold_conv_bn(const Weights& W, const std::string& conv, const std::string& bn, bool has_bias, int cout, int cin,
                 int ks, std::vector<float>& w_out, std::vector<float>& b_out, double eps = kBnEps) {
    const float *w = nullptr, *b = nullptr;
    OCR_TRY(get(W, conv + ".weight", {cout, cin, ks, ks}, &w));
    if (has_bias) OCR_TRY(get(W, conv + ".bias", {cout}, &b));
    const size_t per_out = (size_t)cin * ks * ks;
    w_out.assign(w, w + (size_t)cout * per_out);
    b_out.assign((size_t)cout, 0.0f);
    if (b) b_out.assign(b, b + cout);
    if (!bn.empty()) {
        const float *g = nullptr, *beta = nullptr, *mean = nullptr, *var = nullptr;
        OCR_TRY(get(W, bn + ".weight", {cout}, &g));
        OCR_TRY(get(W, bn + ".bias", {cout}, &beta));
        OCR_TRY(get(W, bn + ".running_mean", {cout}, &mean));
        OCR_TRY(get(W, bn + ".running_var", {cout}, &var));
        for (int o = 0; o < cout; ++o) {
            const double sc = (double)g[o] / std::sqrt((double)var[o] + eps);
            for (size_t i = 0; i < per_out; ++i) w_out[(size_t)o * per_out + i] = (float)((double)w_out[(size_t)o * per_out + i] * sc);
            b_out[o] = (float)(((double)b_out[o] - (double)mean[o]) * sc + (double)beta[o]);
        }
    }
    return OCRHIP_OK;
}

const int kEastVggIdx[13] = {0, 3, 7, 10, 14, 17, 20, 24, 27, 30, 34, 37, 40};
const int kEastVggCout[13] = {64, 64, 128, 128, 256, 256, 256, 512, 512, 512, 512, 512, 512};
const int kEastVggPool[13] = {0, 1, 0, 1, 0, 0, 1, 0, 0, 1, 0, 0, 1};
const int kMergeCin[7] = {1024, 128, 384, 64, 192, 32, 32};
const int kMergeCout[7] = {128, 128, 64, 64, 32, 32, 32};
const int kMergeKs[7] = {1, 3, 1, 3, 1, 3, 3};

void record(std::map<std::string, Activation>& taps, const char* name, const void* p, int64_t b, int64_t h, int64_t w,
            int64_t c, bool bf16 = false) {
    Activation a;
    a.ptr = p;
    a.bf16 = bf16;
    a.shape[0] = b; a.shape[1] = h; a.shape[2] = w; a.shape[3] = c;
    taps[name] = a;
}

int read_activation(const std::map<std::string, Activation>& taps, const char* name, float* dst, int64_t cap,
                    int64_t shape[4]) {
    OCR_REQUIRE(name && shape, "get_activation: null argument");
    auto it = taps.find(name);
    OCR_REQUIRE(it != taps.end(), "get_activation: no activation named '%s' (run a forward first)", name);
    int64_t k = 1;
    for (int i = 0; i < 4; ++i) {
        shape[i] = it->second.shape[i];
        k *= shape[i];
    }
    if (!dst) return OCRHIP_OK;
    if (cap < k) {
        set_error("get_activation: destination holds %lld elements, %lld needed", (long long)cap, (long long)k);
        return OCRHIP_ECAP;
    }
    OCR_HIP_CHECK(hipDeviceSynchronize());
    if (it->second.bf16) {
        std::vector<uint16_t> raw((size_t)k);
        OCR_HIP_CHECK(hipMemcpy(raw.data(), it->second.ptr, (size_t)k * sizeof(uint16_t), hipMemcpyDeviceToHost));
        for (int64_t i = 0; i < k; ++i) {
            const uint32_t bits = (uint32_t)raw[(size_t)i] << 16;
            std::memcpy(dst + i, &bits, sizeof(float));
        }
        return OCRHIP_OK;
    }
    OCR_HIP_CHECK(hipMemcpy(dst, it->second.ptr, (size_t)k * sizeof(float), hipMemcpyDeviceToHost));
    return OCRHIP_OK;
}

}  // namespace
}  // namespace ocrhip

using namespace ocrhip;

#define OCR_HANDLE(h)                                               \
    do {                                                            \
        if (!(h)) {                                                 \
            ocrhip::set_error("null handle");                       \
            return OCRHIP_EINVAL;                                   \
        }                                                           \
        OCR_HIP_CHECK(hipSetDevice((h)->device));                   \
        ocrhip::set_current_profiler(&(h)->prof);                   \
        ocrhip::set_conv3x3_winograd((h)->conv3x3_winograd);        \
        ocrhip::set_lstm_persistent((h)->lstm_persistent);          \
        ocrhip::set_lstm16_units((h)->lstm16_units);                \
        ocrhip::set_lstm_debug_drop((h)->lstm_debug_drop);          \
        ocrhip::set_flat_gemm((h)->flat_gemm);                      \
        ocrhip::g_fuse_ctc = (h)->fuse_ctc;                         \
        ocrhip::g_fuse_mean_rows = (h)->fuse_mean_rows;             \
        ocrhip::set_current_workspace(&(h)->wino_ws);               \
        ocrhip::set_wino4_split((h)->wino4_split);                  \
        ocrhip::set_wino4_wave_tiles((h)->wino4_wave_tiles);        \
        ocrhip::set_wino4_side_by_side((h)->wino4_side_by_side);    \
        ocrhip::set_first_layer_mfma((h)->first_layer_mfma);        \
    } while (0)

extern "C" {

int ocrhip_version(void) { return 100; }

const char* ocrhip_last_error(void) { return ocrhip::g_error; }

int ocrhip_create(int device, ocrhip_handle* out) {
    OCR_REQUIRE(out, "create: null output pointer");
    int count = 0;
    OCR_HIP_CHECK(hipGetDeviceCount(&count));
    OCR_REQUIRE(device >= 0 && device < count, "create: device %d not available (%d devices)", device, count);
    OCR_HIP_CHECK(hipSetDevice(device));
    hipDeviceProp_t prop;
    OCR_HIP_CHECK(hipGetDeviceProperties(&prop, device));
    if (std::strncmp(prop.gcnArchName, "gfx950", 6) != 0) {
        set_error("create: device %d is %s; this library contains gfx950 code only", device, prop.gcnArchName);
        return OCRHIP_ESTATE;
    }
    OCR_TRY(decode_static_init(device));
    ocrhip_ctx* h = new ocrhip_ctx();
    h->device = device;
    *out = h;
    return OCRHIP_OK;
}

int ocrhip_destroy(ocrhip_handle h) {
    if (!h) return OCRHIP_OK;
    (void)hipSetDevice(h->device);
    (void)hipDeviceSynchronize();
    auto free_layer = [](ConvLayer& L) { L.w_f32.release(); L.w_wino.release(); L.w_wino4.release(); L.w_wino4b.release(); L.w_wino4c.release(); L.w_bf16.release(); L.w_bf16t.release(); L.bias.release(); L.w_gemm.release(); L.bias_gemm.release(); };
    for (auto& L : h->crnn.rconv) free_layer(L);
    for (auto& L : h->east.vgg) free_layer(L);
    for (auto& L : h->east.merge) free_layer(L);
    free_layer(h->east.head);
    for (auto& b : h->east.act) b.release();
    for (auto& b : h->east.pool) b.release();
    for (auto& L : h->crnn.conv) free_layer(L);
    for (int l = 0; l < 2; ++l) {
        for (int d = 0; d < 2; ++d) {
            free_layer(h->crnn.wih[l][d]);
            h->crnn.whh[l][d].release();
        }
        free_layer(h->crnn.linear[l]);
    }
    free_layer(h->crnn.pred);
    for (auto& b : h->crnn.act) b.release();
    for (int i = 0; i < CrnnNet::kAux; ++i) {
        for (auto& b : h->crnn.act_aux[i]) b.release();
        if (h->crnn.aux_stream[i]) (void)hipStreamDestroy(h->crnn.aux_stream[i]);
        if (h->crnn.aux_join[i]) (void)hipEventDestroy(h->crnn.aux_join[i]);
    }
    if (h->crnn.aux_fork) (void)hipEventDestroy(h->crnn.aux_fork);
    for (DevBuf* b : {&h->crnn.feats, &h->crnn.xw, &h->crnn.hbuf, &h->crnn.cbuf, &h->crnn.seq, &h->crnn.lin, &h->crnn.logits, &h->crnn.seq16, &h->crnn.meta, &h->crnn.lstm_counters, &h->crnn.pmax, &h->crnn.pidx})
        b->release();
    h->crnn.ring.release();
    if (h->crnn.lstm_err_host) (void)hipHostFree(h->crnn.lstm_err_host);
    h->decode_stage.release();
    h->lanms_ws.release();
    h->beam_ws.release();
    h->wino_ws.release();
    ocrhip::set_current_workspace(nullptr);
    h->prof.recycle();
    for (hipEvent_t e : h->prof.pool) (void)hipEventDestroy(e);
    ocrhip::set_current_profiler(nullptr);
    delete h;
    return OCRHIP_OK;
}

// ------------------------------------------------------------------------------------------ EAST

int ocrhip_east_load_weights(ocrhip_handle h, const ocrhip_tensor_desc* w, int n) {
    OCR_HANDLE(h);
    OCR_REQUIRE(w && n > 0, "east_load_weights: no tensors");
    Weights W{w, n};
    EastNet& net = h->east;
    net.loaded = false;
    std::vector<float> wf, bf;
    int cin = 3;
    for (int i = 0; i < 13; ++i) {
        const std::string conv = "extractor.features." + std::to_string(kEastVggIdx[i]);
        const std::string bn = "extractor.features." + std::to_string(kEastVggIdx[i] + 1);
        OCR_TRY(fold_conv_bn(W, conv, bn, true, kEastVggCout[i], cin, 3, wf, bf));
        if (i == 0) {
            OCR_TRY(prepare_conv_direct(net.vgg[i], wf.data(), bf.data(), kEastVggCout[i], cin, 3));
            OCR_TRY(prepare_first_pair_bf16t(net.vgg[i], wf.data()));
        } else
            OCR_TRY(prepare_conv_mfma(net.vgg[i], wf.data(), bf.data(), kEastVggCout[i], cin, 3));
        cin = kEastVggCout[i];
    }
    for (int k = 0; k < 7; ++k) {
        const std::string conv = "merge.conv" + std::to_string(k + 1), bn = "merge.bn" + std::to_string(k + 1);
        OCR_TRY(fold_conv_bn(W, conv, bn, true, kMergeCout[k], kMergeCin[k], kMergeKs[k], wf, bf));
        OCR_TRY(prepare_conv_mfma(net.merge[k], wf.data(), bf.data(), kMergeCout[k], kMergeCin[k], kMergeKs[k]));
    }
    {
        // heads: output.conv1 (score), output.conv2 (4 distances), output.conv3 (angle) -> [6][32]
        std::vector<float> hw(6 * 32), hb(6);
        const int rows[3] = {1, 4, 1};
        int r0 = 0;
        for (int k = 0; k < 3; ++k) {
            const float *pw = nullptr, *pb = nullptr;
            const std::string name = "output.conv" + std::to_string(k + 1);
            OCR_TRY(get(W, name + ".weight", {rows[k], 32, 1, 1}, &pw));
            OCR_TRY(get(W, name + ".bias", {rows[k]}, &pb));
            std::memcpy(hw.data() + r0 * 32, pw, sizeof(float) * rows[k] * 32);
            std::memcpy(hb.data() + r0, pb, sizeof(float) * rows[k]);
            r0 += rows[k];
        }
        ConvLayer& L = net.head;
        L.cin = 32; L.cout = 6; L.ks = 1; L.cout_pad = 6; L.direct = true;
        OCR_TRY(L.w_f32.reserve(hw.size() * sizeof(float)));
        OCR_TRY(L.bias.reserve(hb.size() * sizeof(float)));
        OCR_HIP_CHECK(hipMemcpy(L.w_f32.ptr, hw.data(), hw.size() * sizeof(float), hipMemcpyHostToDevice));
        OCR_HIP_CHECK(hipMemcpy(L.bias.ptr, hb.data(), hb.size() * sizeof(float), hipMemcpyHostToDevice));
    }
    net.loaded = true;
    return OCRHIP_OK;
}

// The detector on bfloat16 tensors: the first layer and the heads compute in float32 (3 and 32 input channels are no
// matrix work), every other layer is a bfloat16 tile kernel with float32 accumulation; activations are bfloat16 in HBM.
// The x2 upsampling + concatenation of a merge stage is one bandwidth-bound pass in front of its 1x1 convolution.
static int east_forward_bf16(EastNet& net, const void* d_img, int B, int H, int W, int in_dtype, float* d_maps, bool fuse_first, hipStream_t s) {
    const size_t px = (size_t)B * H * W;
    unsigned char* act0 = net.act[0].as<unsigned char>();          // 256 bytes per input pixel
    uint16_t* T = reinterpret_cast<uint16_t*>(act0);               // upsample + concat of a merge stage: <= 24 bytes per input pixel
    uint16_t* X = reinterpret_cast<uint16_t*>(act0 + px * 64);     // <= 128 bytes per input pixel
    uint16_t* Y = net.act[1].as<uint16_t>();                       // 128 bytes per input pixel
    const int pool_c[5] = {64, 128, 256, 512, 512};
    const uint16_t* cur = X;
    int ch = H, cw = W, np = 0;
    int i0 = 1;
    if (fuse_first && kEastVggPool[1] && (int64_t)H * W * 128 < (1ll << 31)) {
        // layers 0 and 1 in one launch: the 64-channel tensor at full resolution is never written
        uint16_t* dst = net.pool[0].as<uint16_t>();
        OCR_TRY(launch_conv3_first_pair_bf16t(net.vgg[0], net.vgg[1], d_img, in_dtype == OCRHIP_U8, dst, B, H, W, s));
        ch /= 2; cw /= 2;
        record(net.taps, "pool1", dst, B, ch, cw, pool_c[0], true);
        np = 1;
        cur = dst;
        i0 = 2;
    } else {
        OCR_TRY(launch_conv_first(net.vgg[0], d_img, in_dtype == OCRHIP_U8, X, B, H, W, 0, s, true));
    }
    for (int i = i0; i < 13; ++i) {
        uint16_t* dst = kEastVggPool[i] ? net.pool[np].as<uint16_t>() : ((cur == X) ? Y : X);
        OCR_TRY(launch_conv3_bf16t(net.vgg[i], cur, dst, B, 1, ch, cw, 0, kEastVggPool[i] ? 1 : 0, true, s));
        if (kEastVggPool[i]) {
            ch /= 2; cw /= 2;
            const std::string name = "pool" + std::to_string(np + 1);
            record(net.taps, name.c_str(), dst, B, ch, cw, pool_c[np], true);
            ++np;
        }
        cur = dst;
    }
    const uint16_t* up = net.pool[4].as<uint16_t>();
    int up_c = 512;
    int mh = H / 32, mw = W / 32;
    int k = 0;
    for (int stage = 0; stage < 3; ++stage) {
        mh *= 2; mw *= 2;
        const uint16_t* res = net.pool[3 - stage].as<uint16_t>();
        OCR_TRY(launch_upcat16(up, res, T, B, mh, mw, up_c, kMergeCin[k] - up_c, s));
        OCR_TRY(launch_gemm_bf16t(net.merge[k], T, X, false, (int64_t)B * mh * mw, 0, 0, 0, true, s));
        ++k;
        OCR_TRY(launch_conv3_bf16t(net.merge[k], X, Y, B, 1, mh, mw, 0, 0, true, s));
        ++k;
        const uint16_t* last = Y;
        if (stage == 2) {
            OCR_TRY(launch_conv3_bf16t(net.merge[k], Y, X, B, 1, mh, mw, 0, 0, true, s));
            ++k;
            last = X;
        }
        const std::string name = "merge" + std::to_string(k);
        record(net.taps, name.c_str(), last, B, mh, mw, kMergeCout[k - 1], true);
        up = last;
        up_c = kMergeCout[k - 1];
    }
    OCR_TRY(launch_east_heads16(net.head, up, d_maps, (int64_t)B * mh * mw, s));
    return OCRHIP_OK;
}

int ocrhip_east_forward(ocrhip_handle h, const void* d_img, int B, int H, int W, int in_dtype, int compute_dtype,
                        float* d_maps, ocrhip_stream stream) {
    OCR_HANDLE(h);
    EastNet& net = h->east;
    if (!net.loaded) {
        set_error("east_forward: weights not loaded");
        return OCRHIP_ESTATE;
    }
    OCR_REQUIRE(d_img && d_maps, "east_forward: null buffer");
    OCR_REQUIRE(B > 0 && H > 0 && W > 0 && H % 32 == 0 && W % 32 == 0, "east_forward: H and W must be positive multiples of 32 (B=%d H=%d W=%d)", B, H, W);
    OCR_REQUIRE(in_dtype == OCRHIP_U8 || in_dtype == OCRHIP_F32, "east_forward: in_dtype must be U8 or F32");
    OCR_REQUIRE(compute_dtype == OCRHIP_F32 || compute_dtype == OCRHIP_BF16, "east_forward: compute dtype %d is neither F32 nor BF16", compute_dtype);
    Bf16Scope bf16_scope(compute_dtype);
    hipStream_t s = (hipStream_t)stream;
    const size_t px = (size_t)B * H * W;
    OCR_TRY(net.act[0].reserve(px * 64 * sizeof(float)));
    OCR_TRY(net.act[1].reserve(px * 32 * sizeof(float)));
    const int pool_c[5] = {64, 128, 256, 512, 512};
    for (int i = 0; i < 5; ++i) OCR_TRY(net.pool[i].reserve((px >> (2 * (i + 1))) * pool_c[i] * sizeof(float)));
    float* A = net.act[0].as<float>();
    float* Bf = net.act[1].as<float>();
    net.taps.clear();
    if (compute_dtype == OCRHIP_BF16 && h->bf16_tensors) return east_forward_bf16(net, d_img, B, H, W, in_dtype, d_maps, h->bf16_fuse_first != 0, s);

    // VGG16-BN body; a convolution followed by 'M' writes the pooled tensor directly
    OCR_TRY(launch_conv_first(net.vgg[0], d_img, in_dtype == OCRHIP_U8, A, B, H, W, 0, s));
    const float* cur = A;
    int ch = H, cw = W, np = 0;
    for (int i = 1; i < 13; ++i) {
        float* dst;
        if (kEastVggPool[i])
            dst = net.pool[np].as<float>();
        else
            dst = (cur == A) ? Bf : A;
        OCR_TRY(launch_conv_mfma(net.vgg[i], cur, dst, B, ch, cw, kEastVggPool[i] ? 1 : 0, true, nullptr, 0, s));
        if (kEastVggPool[i]) {
            ch /= 2; cw /= 2;
            const std::string name = "pool" + std::to_string(np + 1);
            record(net.taps, name.c_str(), dst, B, ch, cw, pool_c[np]);
            ++np;
        }
        cur = dst;
    }
    // merge branch: [upsample x2 + concat + 1x1] then 3x3 (two of them at the last stage)
    const float* up = net.pool[4].as<float>();
    int up_c = 512;
    int mh = H / 32, mw = W / 32;
    int k = 0;
    for (int stage = 0; stage < 3; ++stage) {
        mh *= 2; mw *= 2;
        const float* res = net.pool[3 - stage].as<float>();
        OCR_TRY(launch_conv_mfma(net.merge[k], res, A, B, mh, mw, 0, true, up, up_c, s));
        ++k;
        OCR_TRY(launch_conv_mfma(net.merge[k], A, Bf, B, mh, mw, 0, true, nullptr, 0, s));
        ++k;
        const float* last = Bf;
        if (stage == 2) {
            OCR_TRY(launch_conv_mfma(net.merge[k], Bf, A, B, mh, mw, 0, true, nullptr, 0, s));
            ++k;
            last = A;
        }
        const std::string name = "merge" + std::to_string(k);
        record(net.taps, name.c_str(), last, B, mh, mw, kMergeCout[k - 1]);
        up = last;
        up_c = kMergeCout[k - 1];
    }
    OCR_TRY(launch_east_heads(net.head, up, d_maps, (int64_t)B * mh * mw, s));
    return OCRHIP_OK;
}

int ocrhip_image_resize_norm(ocrhip_handle h, const void* d_img, int in_dtype, int B, int H, int W, int out_h, int out_w,
                             float* d_out, ocrhip_stream stream) {
    OCR_HANDLE(h);
    OCR_REQUIRE(in_dtype == OCRHIP_U8 || in_dtype == OCRHIP_F32, "image_resize_norm: input must be uint8 or float32");
    return launch_resize_rgb_norm(d_img, in_dtype == OCRHIP_U8, B, H, W, out_h, out_w, d_out, (hipStream_t)stream);
}

int ocrhip_dice_coeff(ocrhip_handle h, const float* d_x, const float* d_y, int64_t n, int64_t stride, float* d_out,
                      ocrhip_stream stream) {
    OCR_HANDLE(h);
    return launch_dice_coeff(d_x, d_y, n, stride, d_out, (hipStream_t)stream);
}

int ocrhip_east_get_activation(ocrhip_handle h, const char* name, float* dst, int64_t dst_capacity, int64_t shape[4]) {
    OCR_HANDLE(h);
    return read_activation(h->east.taps, name, dst, dst_capacity, shape);
}

int ocrhip_east_decode(ocrhip_handle h, const float* d_maps, int B, int mh, int mw, int in_h, int in_w,
                       float obj_threshold, int cap, float* d_xyxy, float* d_polys, float* d_scores, int* d_pix,
                       int* d_counts, ocrhip_stream stream) {
    OCR_HANDLE(h);
    return launch_east_decode(h, d_maps, B, mh, mw, in_h, in_w, obj_threshold, cap, d_xyxy, d_polys, d_scores, d_pix,
                              d_counts, (hipStream_t)stream);
}

int ocrhip_east_decode_static(ocrhip_handle h, const float* d_maps, int B, int mh, int mw, int in_h, int in_w,
                              float obj_threshold, int cap, float* d_xyxy, float* d_polys, float* d_scores, int* d_pix,
                              int* d_counts, ocrhip_stream stream) {
    // no OCR_HANDLE: nothing here may create events or touch the handle's buffers
    OCR_REQUIRE(h, "east_decode_static: null handle");
    OCR_HIP_CHECK(hipSetDevice(h->device));
    return launch_east_decode_static(h, d_maps, B, mh, mw, in_h, in_w, obj_threshold, cap, d_xyxy, d_polys, d_scores,
                                     d_pix, d_counts, (hipStream_t)stream);
}

int ocrhip_lanms(ocrhip_handle h, const float* d_xyxy, const float* d_scores, const int* d_counts, int B, int cap,
                 float merge_threshold, float nms_threshold, int merge_method, int max_iter, int max_out, float* d_merged,
                 uint8_t* d_keep, float* d_out_boxes, float* d_out_scores, int* d_out_src, int* d_out_counts,
                 int* d_n_iter, ocrhip_stream stream) {
    OCR_HANDLE(h);
    return launch_lanms(h, d_xyxy, d_scores, d_counts, B, cap, merge_threshold, nms_threshold, merge_method, max_iter,
                        max_out, d_merged, d_keep, d_out_boxes, d_out_scores, d_out_src, d_out_counts, d_n_iter,
                        (hipStream_t)stream);
}

int ocrhip_crop_resize_pack(ocrhip_handle h, const uint8_t* d_img, int B, int H, int W, const int* d_rects,
                            const int* d_box_img, const int* d_inter, const int* d_out_w, const int64_t* d_dst_off,
                            int n, int out_h, void* d_packed, int out_dtype, ocrhip_stream stream) {
    OCR_HANDLE(h);
    return launch_crop_resize_pack(d_img, B, H, W, d_rects, false, d_box_img, d_inter, d_out_w, d_dst_off, n, out_h,
                                   d_packed, out_dtype, (hipStream_t)stream);
}

int ocrhip_crop_rotated_pack(ocrhip_handle h, const uint8_t* d_img, int B, int H, int W, const float* d_geom,
                             const int* d_box_img, const int* d_inter, const int* d_out_w, const int64_t* d_dst_off,
                             int n, int out_h, void* d_packed, int out_dtype, ocrhip_stream stream) {
    OCR_HANDLE(h);
    return launch_crop_resize_pack(d_img, B, H, W, d_geom, true, d_box_img, d_inter, d_out_w, d_dst_off, n, out_h,
                                   d_packed, out_dtype, (hipStream_t)stream);
}

int ocrhip_rotated_nms(ocrhip_handle h, const float* d_polys, const float* d_scores, const int* d_counts, int B, int cap,
                       float iou_threshold, int max_out, float* d_out_polys, float* d_out_scores, int* d_out_index,
                       int* d_out_counts, ocrhip_stream stream) {
    OCR_HANDLE(h);
    return launch_rotated_nms(h, d_polys, d_scores, d_counts, B, cap, iou_threshold, max_out, d_out_polys, d_out_scores,
                              d_out_index, d_out_counts, (hipStream_t)stream);
}

// ------------------------------------------------------------------------------------------ CRNN

// ResNet extractor as the reference builds it (architectures/crnn_arch.py:330-425): layers in execution order.
// The batch-norms inside the BasicBlocks (and after their 1x1 downsample) use the Conv2DBN default epsilon
// 1e-3 (current_blocks.py:222), the stage convolutions pass 1e-5 (crnn_arch.py:416).
struct ResLayer { std::string conv, bn; int cin, cout, ks; double eps; };

static std::vector<ResLayer> resnet_layers() {
    const std::string pre = "FeatureExtraction.ConvNet.";
    std::vector<ResLayer> L;
    L.push_back({pre + "conv0_1", pre + "bn0_1", 1, 32, 3, 1e-5});
    L.push_back({pre + "conv0_2", pre + "bn0_2", 32, 64, 3, 1e-5});
    const int blocks[4] = {1, 2, 5, 3}, chans[4] = {128, 256, 512, 512};
    int cin = 64;
    for (int st = 0; st < 4; ++st) {
        const std::string lay = pre + "layer" + std::to_string(st + 1) + ".";
        const int c = chans[st];
        if (cin != c) L.push_back({lay + "0.downsample.0", lay + "0.downsample.1", cin, c, 1, 1e-3});
        for (int b = 0; b < blocks[st]; ++b) {
            const std::string blk = lay + std::to_string(b) + ".";
            L.push_back({blk + "conv1", blk + "bn1", b == 0 ? cin : c, c, 3, 1e-3});
            L.push_back({blk + "conv2", blk + "bn2", c, c, 3, 1e-3});
        }
        if (st < 3) {
            const std::string k = std::to_string(st + 1);
            L.push_back({pre + "conv" + k, pre + "bn" + k, c, c, 3, 1e-5});
        }
        cin = c;
    }
    L.push_back({pre + "conv4_1", pre + "bn4_1", 512, 512, 2, 1e-5});
    L.push_back({pre + "conv4_2", pre + "bn4_2", 512, 512, 2, 1e-5});
    return L;
}

int ocrhip_crnn_load_weights(ocrhip_handle h, const ocrhip_tensor_desc* w, int n) {
    OCR_HANDLE(h);
    OCR_REQUIRE(w && n > 0, "crnn_load_weights: no tensors");
    Weights W{w, n};
    CrnnNet& net = h->crnn;
    net.loaded = false;
    std::vector<float> wf, bf;
    net.backbone = W.find("FeatureExtraction.ConvNet.conv0_1.weight") ? 1 : 0;
    net.feat_c = net.backbone ? 512 : 256;
    if (net.backbone) {
        const std::vector<ResLayer> layers = resnet_layers();
        OCR_REQUIRE(layers.size() == 32, "crnn_load_weights: internal layer table");
        for (size_t i = 0; i < layers.size(); ++i) {
            const ResLayer& r = layers[i];
            OCR_TRY(fold_conv_bn(W, r.conv, r.bn, false, r.cout, r.cin, r.ks, wf, bf, r.eps));
            if (i == 0)
                OCR_TRY(prepare_conv_direct(net.rconv[i], wf.data(), bf.data(), r.cout, r.cin, r.ks));
            else
                OCR_TRY(prepare_conv_mfma(net.rconv[i], wf.data(), bf.data(), r.cout, r.cin, r.ks));
        }
    } else {
        // (Sequential index, cin, cout, kernel, bias, batch-norm index or -1), crnn_arch.py:38-41
        const int cfg[7][6] = {{0, 1, 32, 3, 1, -1},    {3, 32, 64, 3, 1, -1},   {6, 64, 128, 3, 1, -1}, {8, 128, 128, 3, 1, -1},
                               {11, 128, 256, 3, 0, 12}, {14, 256, 256, 3, 0, 15}, {18, 256, 256, 2, 1, -1}};
        for (int i = 0; i < 7; ++i) {
            const std::string conv = "FeatureExtraction.ConvNet." + std::to_string(cfg[i][0]);
            const std::string bn = cfg[i][5] >= 0 ? "FeatureExtraction.ConvNet." + std::to_string(cfg[i][5]) : std::string();
            OCR_TRY(fold_conv_bn(W, conv, bn, cfg[i][4] != 0, cfg[i][2], cfg[i][1], cfg[i][3], wf, bf));
            if (i == 0)
                OCR_TRY(prepare_conv_direct(net.conv[i], wf.data(), bf.data(), cfg[i][2], cfg[i][1], cfg[i][3]));
            else
                OCR_TRY(prepare_conv_mfma(net.conv[i], wf.data(), bf.data(), cfg[i][2], cfg[i][1], cfg[i][3]));
        }
    }
    {
        // hidden units of the BiLSTM layers (`lstm_size`, architectures/crnn_arch.py:485-486): read off the recurrent matrix
        const ocrhip_tensor_desc* t = W.find("SequenceModeling.0.rnn.weight_hh_l0");
        OCR_REQUIRE(t && t->data && t->ndim == 2 && t->shape[1] >= 64 && t->shape[1] % 64 == 0 && t->shape[0] == 4 * t->shape[1],
                    "crnn_load_weights: SequenceModeling.0.rnn.weight_hh_l0 must be (4H, H) with H a multiple of 64");
        net.hidden = (int)t->shape[1];
    }
    const int Hd = net.hidden;
    for (int l = 0; l < 2; ++l) {
        const std::string pre = "SequenceModeling." + std::to_string(l);
        for (int d = 0; d < 2; ++d) {
            const std::string suf = d ? "_reverse" : "";
            const float *wih = nullptr, *whh = nullptr, *bih = nullptr, *bhh = nullptr;
            const int in_c = l == 0 ? net.feat_c : Hd;
            OCR_TRY(get(W, pre + ".rnn.weight_ih_l0" + suf, {4 * Hd, in_c}, &wih));
            OCR_TRY(get(W, pre + ".rnn.weight_hh_l0" + suf, {4 * Hd, Hd}, &whh));
            OCR_TRY(get(W, pre + ".rnn.bias_ih_l0" + suf, {4 * Hd}, &bih));
            OCR_TRY(get(W, pre + ".rnn.bias_hh_l0" + suf, {4 * Hd}, &bhh));
            // rows of W_ih reordered from (gate, unit) to (unit, gate): the recurrence reads the four gate inputs of a unit
            // as one 16-byte load
            std::vector<float> bsum(4 * Hd), wperm((size_t)4 * Hd * in_c);
            for (int gq = 0; gq < 4; ++gq)
                for (int u = 0; u < Hd; ++u) {
                    bsum[u * 4 + gq] = bih[gq * Hd + u] + bhh[gq * Hd + u];
                    std::copy(wih + (size_t)(gq * Hd + u) * in_c, wih + (size_t)(gq * Hd + u + 1) * in_c, wperm.begin() + (size_t)(u * 4 + gq) * in_c);
                }
            OCR_TRY(prepare_conv_mfma(net.wih[l][d], wperm.data(), bsum.data(), 4 * Hd, in_c, 1));
            OCR_TRY(net.whh[l][d].reserve((size_t)4 * Hd * Hd * sizeof(float)));
            OCR_HIP_CHECK(hipMemcpy(net.whh[l][d].ptr, whh, (size_t)4 * Hd * Hd * sizeof(float), hipMemcpyHostToDevice));
        }
        const float *lw = nullptr, *lb = nullptr;
        OCR_TRY(get(W, pre + ".linear.weight", {Hd, 2 * Hd}, &lw));
        OCR_TRY(get(W, pre + ".linear.bias", {Hd}, &lb));
        OCR_TRY(prepare_conv_mfma(net.linear[l], lw, lb, Hd, 2 * Hd, 1));
    }
    {
        const ocrhip_tensor_desc* t = W.find("Prediction.weight");
        OCR_REQUIRE(t && t->data && t->ndim == 2 && t->shape[1] == Hd && t->shape[0] > 0,
                    "crnn_load_weights: Prediction.weight must be (vocab, %d)", Hd);
        const int V = (int)t->shape[0];
        const float* pb = nullptr;
        OCR_TRY(get(W, "Prediction.bias", {V}, &pb));
        OCR_TRY(prepare_conv_mfma(net.pred, t->data, pb, V, Hd, 1));
        net.vocab = V;
        net.vocab_pad = net.pred.cout_pad;
    }
    net.loaded = true;
    return OCRHIP_OK;
}

int ocrhip_crnn_vocab_size(ocrhip_handle h) { return (h && h->crnn.loaded) ? h->crnn.vocab : 0; }

int ocrhip_crnn_seq_len(ocrhip_handle h, int wpad) { return (h && h->crnn.loaded && wpad > 0) ? h->crnn.seq_len(wpad) : 0; }

// ResNet extractor: (n,64,W,1) -> feats (n,T,512), T = W/4 + 1.  Three rotating buffers: A / B ping-pong and R
// for the 1x1 downsample that the first block of a stage adds back.  The reference adds a skip connection only
// there (crnn_arch.py:357); the other blocks are plain conv-bn-relu-conv-bn-relu.  The two 2x2 poolings of
// strides 2 are fused into the convolution in front of them.
static int crnn_features_resnet(CrnnNet& net, const float* x, int n, int Wpad, float* feats, hipStream_t s) {
    OCR_REQUIRE(Wpad % 4 == 0, "crnn: the ResNet extractor needs a width that is a multiple of 4 (got %d)", Wpad);
    const size_t per_buf = (size_t)n * Wpad * 2048 * sizeof(float) + (size_t)n * 8 * 3 * 512 * sizeof(float);
    for (int i = 0; i < 3; ++i) OCR_TRY(net.act[i].reserve(per_buf));
    float* A = net.act[0].as<float>();
    float* Bf = net.act[1].as<float>();
    float* R = net.act[2].as<float>();
    const ConvLayer* L = net.rconv;
    int li = 0;
    OCR_TRY(launch_conv_first(L[li++], x, false, A, n, 64, Wpad, 0, s));                                  // (n,64,W,32)
    OCR_TRY(launch_conv_mfma(L[li++], A, Bf, n, 64, Wpad, 1, true, nullptr, 0, s));                      // (n,32,W/2,64)
    float *cur = Bf, *other = A;
    int H = 32, W = Wpad / 2;
    const int blocks[4] = {1, 2, 5, 3}, chans[4] = {128, 256, 512, 512};
    int cin = 64;
    for (int st = 0; st < 4; ++st) {
        const int c = chans[st];
        const bool down = cin != c;
        if (down) OCR_TRY(launch_conv_mfma(L[li++], cur, R, 1, 1, n * H * W, 0, false, nullptr, 0, s));  // 1x1, bn, no relu
        for (int b = 0; b < blocks[st]; ++b) {
            OCR_TRY(launch_conv_mfma(L[li++], cur, other, n, H, W, 0, true, nullptr, 0, s));
            OCR_TRY(launch_conv_mfma(L[li++], other, cur, n, H, W, 0, true, nullptr, 0, s, (down && b == 0) ? R : nullptr));
        }
        if (st == 0) {                      // conv1 + max-pool 2x2
            OCR_TRY(launch_conv_mfma(L[li++], cur, other, n, H, W, 1, true, nullptr, 0, s));
            std::swap(cur, other); H /= 2; W /= 2;
        } else if (st == 1) {               // conv2, then width padding + max-pool 2x2 with strides (2,1)
            OCR_TRY(launch_conv_mfma(L[li++], cur, other, n, H, W, 0, true, nullptr, 0, s));
            OCR_TRY(launch_pad_maxpool(other, cur, n, H, W, c, 2, 2, 2, 1, 1, s));
            H /= 2; W += 1;
        } else if (st == 2) {               // conv3
            OCR_TRY(launch_conv_mfma(L[li++], cur, other, n, H, W, 0, true, nullptr, 0, s));
            std::swap(cur, other);
        }
        cin = c;
    }
    // conv4_1: 2x2, strides (2,1), on the width-padded map.  Every pair of rows is its own 2-row image for the
    // stride-1 kernel: (n,8,W+2,512) viewed as (4n,2,W+2,512) -> (4n,1,W+1,512) = (n,4,W+1,512).
    OCR_REQUIRE(H == 8, "crnn: internal size error (H=%d)", H);
    OCR_TRY(launch_pad_maxpool(cur, other, n, H, W, 512, 1, 1, 1, 1, 1, s));
    OCR_TRY(launch_conv_mfma(L[li++], other, cur, n * 4, 2, W + 2, 0, true, nullptr, 0, s));
    OCR_REQUIRE(li == 31 && W == net.seq_len(Wpad), "crnn: internal layer count error");
    if (g_fuse_mean_rows && flat_gemm() && !compute_bf16() && L[li].w_gemm.ptr) {
        // the last convolution and the mean over its 3 output rows in one kernel: (n,3,W,512) never reaches memory
        OCR_TRY(launch_gemm_f32_mean_rows(L[li], cur, feats, n, 4, W + 1, true, s));
    } else {
        OCR_TRY(launch_conv_mfma(L[li], cur, other, n, 4, W + 1, 0, true, nullptr, 0, s));                // (n,3,W,512)
        OCR_TRY(launch_mean_rows(other, feats, n, 3, W, 512, s));
    }
    return OCRHIP_OK;
}

// Feature extractor for a batch of equal-width crops: (n,64,Wpad,1) -> feats (n,T,feat_c).
// VGG: T = Wpad/4 - 1, 256 features.
// `act`: the two activation buffers of the stream `s` (the net's own, or those of a side stream).
static int crnn_features(CrnnNet& net, DevBuf* act, const float* x, int n, int Wpad, float* feats, hipStream_t s) {
    if (net.backbone) return crnn_features_resnet(net, x, n, Wpad, feats, s);
    const int Hin = 64;
    const int T = Wpad / 4 - 1;
    const size_t nW = (size_t)n * Wpad;
    OCR_TRY(act[0].reserve(nW * 512 * sizeof(float)));          // (n,32,W/2,32) is the largest
    OCR_TRY(act[1].reserve(nW * 256 * sizeof(float)));          // (n,16,W/4,64)
    float* A = act[0].as<float>();
    float* Bf = act[1].as<float>();
    OCR_TRY(launch_conv_first(net.conv[0], x, false, A, n, Hin, Wpad, 1, s));                       // (n,32,W/2,32)
    OCR_TRY(launch_conv_mfma(net.conv[1], A, Bf, n, 32, Wpad / 2, 1, true, nullptr, 0, s));         // (n,16,W/4,64)
    OCR_TRY(launch_conv_mfma(net.conv[2], Bf, A, n, 16, Wpad / 4, 0, true, nullptr, 0, s));         // (n,16,W/4,128)
    OCR_TRY(launch_conv_mfma(net.conv[3], A, Bf, n, 16, Wpad / 4, 2, true, nullptr, 0, s));         // (n,8,W/4,128)
    OCR_TRY(launch_conv_mfma(net.conv[4], Bf, A, n, 8, Wpad / 4, 0, true, nullptr, 0, s));          // (n,8,W/4,256)
    OCR_TRY(launch_conv_mfma(net.conv[5], A, Bf, n, 8, Wpad / 4, 2, true, nullptr, 0, s));          // (n,4,W/4,256)
    if (g_fuse_mean_rows && flat_gemm() && !compute_bf16() && net.conv[6].w_gemm.ptr) {
        // the 2x2 convolution and the mean over its 3 output rows in one kernel: (n,3,T,256) never reaches memory
        OCR_TRY(launch_gemm_f32_mean_rows(net.conv[6], Bf, feats, n, 4, Wpad / 4, true, s));
    } else {
        OCR_TRY(launch_conv_mfma(net.conv[6], Bf, A, n, 4, Wpad / 4, 0, true, nullptr, 0, s));      // (n,3,T,256)
        OCR_TRY(launch_mean_rows(A, feats, n, 3, T, net.feat_c, s));
    }
    return OCRHIP_OK;
}

static int crnn_report_timeout(CrnnNet& net, hipStream_t s);

// 2 x [BiLSTM(256) -> Dense(256)] -> Dense(vocab) over R = sum of T rows of `feats`.
// Uniform mode: seq_T == nullptr, every sequence has Tmax steps.  Ragged mode: host_T (non-decreasing)
// and the device arrays d_T / d_row0 describe the sequences.
static int crnn_sequence(CrnnNet& net, const float* feats, int n, int64_t R, int Tmax, const int* host_T,
                         const int* d_T, const int* d_row0, const int* d_active, float* logits, hipStream_t s) {
    const int Hd = net.hidden;
    OCR_REQUIRE(R < (1ll << 31), "crnn: too many time steps in one batch");
    OCR_TRY(net.xw.reserve((size_t)2 * R * 4 * Hd * sizeof(float)));
    OCR_TRY(net.hbuf.reserve((size_t)2 * 2 * n * Hd * sizeof(float)));
    OCR_TRY(net.cbuf.reserve((size_t)2 * n * Hd * sizeof(float)));
    OCR_TRY(net.seq.reserve((size_t)R * 2 * Hd * sizeof(float)));
    OCR_TRY(net.lin.reserve((size_t)2 * R * Hd * sizeof(float)));
    if (!net.lstm_counters.ptr) {
        OCR_TRY(net.lstm_counters.reserve((kLstmCounterCount + kLstmErrorWords) * sizeof(unsigned)));
        OCR_HIP_CHECK(hipMemsetAsync(net.lstm_counters.ptr, 0, (kLstmCounterCount + kLstmErrorWords) * sizeof(unsigned), s));
    }
    if (!net.lstm_err_host) {
        OCR_HIP_CHECK(hipHostMalloc(reinterpret_cast<void**>(&net.lstm_err_host), CrnnNet::kErrSlots * sizeof(unsigned), hipHostMallocDefault));
        for (int i = 0; i < CrnnNet::kErrSlots; ++i) net.lstm_err_host[i] = 0;
    }
    // every forward starts from clean error words (in stream order: an earlier call's time-out cannot make this call's
    // kernels leave early) and reports into a status slot of its own
    OCR_HIP_CHECK(hipMemsetAsync(net.lstm_counters.as<unsigned>() + kLstmCounterCount, 0, kLstmErrorWords * sizeof(unsigned), s));
    const float* layer_in = feats;
    for (int l = 0; l < 2; ++l) {
        float* xw_f = net.xw.as<float>();
        float* xw_b = xw_f + (size_t)R * 4 * Hd;
        OCR_TRY(launch_conv_mfma(net.wih[l][0], layer_in, xw_f, 1, 1, (int)R, 0, false, nullptr, 0, s));
        OCR_TRY(launch_conv_mfma(net.wih[l][1], layer_in, xw_b, 1, 1, (int)R, 0, false, nullptr, 0, s));
        float* hb0 = net.hbuf.as<float>();
        float* hb1 = hb0 + (size_t)2 * n * Hd;
        {
            // one timer scope for the whole recurrence of a layer (Tmax launches, both directions each)
            ProfScope prof("lstm_layer(T steps)", 2.0 * 2 * 4.0 * Hd * Hd * (double)(R - n),
                           4.0 * (2.0 * R * 4 * Hd + (double)R * 2 * Hd) + 4.0 * 2 * 4 * Hd * Hd * Tmax, s);
            int rc = OCRHIP_ESTATE;
            if (lstm_persistent())
                rc = launch_lstm_layer(xw_f, xw_b, net.whh[l][0].as<float>(), net.whh[l][1].as<float>(), hb0, net.cbuf.as<float>(),
                                       net.seq.as<float>(), n, Tmax, Hd, d_T, d_row0, d_active, net.lstm_counters.as<unsigned>(), s);
            if (rc != OCRHIP_OK && rc != OCRHIP_ESTATE) return rc;
            int first = 0;
            for (int step = 0; rc == OCRHIP_ESTATE && step < Tmax; ++step) {          // one launch per time step instead
                if (host_T)
                    while (first < n && host_T[first] <= step) ++first;     // sequences shorter than step+1 are done
                const float* hp = (step & 1) ? hb1 : hb0;
                float* hn = (step & 1) ? hb0 : hb1;
                OCR_TRY(launch_lstm_step(xw_f, xw_b, net.whh[l][0].as<float>(), net.whh[l][1].as<float>(), hp, hn,
                                         net.cbuf.as<float>(), net.seq.as<float>(), n, Tmax, step, d_T, d_row0, first, Hd, s));
            }
        }
        float* lin = net.lin.as<float>() + (size_t)l * R * Hd;
        OCR_TRY(launch_conv_mfma(net.linear[l], net.seq.as<float>(), lin, 1, 1, (int)R, 0, false, nullptr, 0, s));
        record(net.taps, l == 0 ? "seq0" : "seq1", lin, 1, R, Hd, 1);
        layer_in = lin;
    }
    if (logits) {
        OCR_TRY(launch_conv_mfma(net.pred, layer_in, logits, 1, 1, (int)R, 0, false, nullptr, 0, s));
    } else {
        // greedy decoding only: the class layer leaves one (maximum, class) pair per row and block of 128 classes
        const size_t pairs = (size_t)R * gemm_f32_column_blocks(net.pred);
        OCR_TRY(net.pmax.reserve(pairs * sizeof(float)));
        OCR_TRY(net.pidx.reserve(pairs * sizeof(int)));
        OCR_TRY(launch_gemm_f32_argmax(net.pred, layer_in, R, net.pmax.as<float>(), net.pidx.as<int>(), s));
    }
    // the error word of this call's recurrences travels to the host behind the results, into the call's own slot
    if (net.calls_started - net.calls_checked >= (uint64_t)CrnnNet::kErrSlots) {      // ring full: wait for the calls in flight
        OCR_HIP_CHECK(hipStreamSynchronize(s));
        OCR_TRY(crnn_report_timeout(net, s));
    }
    unsigned* slot = net.lstm_err_host + net.calls_started % CrnnNet::kErrSlots;
    *slot = CrnnNet::kErrPending;
    OCR_HIP_CHECK(hipMemcpyAsync(slot, net.lstm_counters.as<unsigned>() + kLstmCounterCount, sizeof(unsigned), hipMemcpyDeviceToHost, s));
    ++net.calls_started;
    return OCRHIP_OK;
}

// Tokens without a logits request: the class layer can hand per-block maxima to the CTC kernel when it runs on the flat
// float32 GEMM (the bfloat16 variant and flat_gemm = 0 go through the convolution kernels, which write logits).
static bool crnn_can_fuse_ctc(const CrnnNet& net) {
    return g_fuse_ctc && flat_gemm() && !compute_bf16() && net.pred.w_gemm.ptr != nullptr;
}

// A persistent recurrence of an earlier call on this handle gave up waiting for its partner workgroups: that call's
// outputs are incomplete.  Every forward has a status slot of its own (pinned, written behind its results); the slots
// of the calls whose status has arrived are read in call order.  On an error the stream is drained and EVERY forward
// enqueued before this check counts as failed (the caller cannot tell from one return code which of its pipelined
// calls it was), reported once.
static int crnn_report_timeout(CrnnNet& net, hipStream_t s) {
    if (!net.lstm_err_host) return OCRHIP_OK;
    bool failed = false;
    while (net.calls_checked < net.calls_started) {
        const unsigned v = *reinterpret_cast<volatile unsigned*>(net.lstm_err_host + net.calls_checked % CrnnNet::kErrSlots);
        if (v == CrnnNet::kErrPending) break;          // that call has not finished yet
        if (v != 0) { failed = true; break; }
        ++net.calls_checked;
    }
    if (!failed) return OCRHIP_OK;
    OCR_HIP_CHECK(hipStreamSynchronize(s));
    net.calls_checked = net.calls_started;
    set_error("lstm hand-over timed out: a persistent BiLSTM launch did not have the whole GPU (another process or stream held "
              "compute units); the outputs of every recogniser call enqueued before this check are invalid.  Retry, or "
              "ocrhip_set_option(h, \"lstm_persistent\", 0)");
    return OCRHIP_EHIP;
}

// ---- the VGG recogniser on bfloat16 tensors (`compute_dtype = OCRHIP_BF16`, option bf16_tensors = 1)
// Crops of one width are tiled side by side in every 3x3 layer (one zero column apart, two under 2x2 pooling): the
// 32-column tiles are full whatever the crop width.
static int crnn_features16(CrnnNet& net, DevBuf* act, const float* x, int n, int Wpad, uint16_t* feats16, hipStream_t s) {
    const int T = Wpad / 4 - 1;
    const size_t nW = (size_t)n * Wpad;
    OCR_TRY(act[0].reserve(nW * 512 * sizeof(float)));
    OCR_TRY(act[1].reserve(nW * 256 * sizeof(float)));
    uint16_t* A = act[0].as<uint16_t>();
    uint16_t* Bf = act[1].as<uint16_t>();
    OCR_TRY(launch_conv_first(net.conv[0], x, false, A, n, 64, Wpad, 1, s, true));                      // (n,32,W/2,32)
    OCR_TRY(launch_conv3_bf16t(net.conv[1], A, Bf, 1, n, 32, Wpad / 2, 2, 1, true, s));                  // (n,16,W/4,64)
    OCR_TRY(launch_conv3_bf16t(net.conv[2], Bf, A, 1, n, 16, Wpad / 4, 1, 0, true, s));                  // (n,16,W/4,128)
    OCR_TRY(launch_conv3_bf16t(net.conv[3], A, Bf, 1, n, 16, Wpad / 4, 1, 2, true, s));                  // (n,8,W/4,128)
    OCR_TRY(launch_conv3_bf16t(net.conv[4], Bf, A, 1, n, 8, Wpad / 4, 1, 0, true, s));                   // (n,8,W/4,256)
    OCR_TRY(launch_conv3_bf16t(net.conv[5], A, Bf, 1, n, 8, Wpad / 4, 1, 2, true, s));                   // (n,4,W/4,256)
    OCR_TRY(launch_gemm_bf16t(net.conv[6], Bf, A, false, (int64_t)n * 3 * T, n, 4, Wpad / 4, true, s));  // (n,3,T,256)
    OCR_TRY(launch_mean_rows16(A, feats16, n, 3, T, net.feat_c, s));
    return OCRHIP_OK;
}

static int crnn_sequence16(CrnnNet& net, const uint16_t* feats16, int n, int64_t R, int Tmax, const int* host_T,
                           const int* d_T, const int* d_row0, const int* d_active, float* logits, hipStream_t s) {
    const int Hd = net.hidden;
    OCR_REQUIRE(R < (1ll << 31), "crnn: too many time steps in one batch");
    OCR_TRY(net.xw.reserve((size_t)2 * R * 4 * Hd * sizeof(float)));
    OCR_TRY(net.hbuf.reserve((size_t)2 * 2 * n * Hd * sizeof(float)));
    OCR_TRY(net.cbuf.reserve((size_t)2 * n * Hd * sizeof(float)));
    OCR_TRY(net.seq.reserve((size_t)R * 2 * Hd * sizeof(float)));
    OCR_TRY(net.lin.reserve((size_t)2 * R * Hd * sizeof(float)));
    OCR_TRY(net.seq16.reserve((size_t)R * 2 * Hd * sizeof(uint16_t)));
    if (!net.lstm_counters.ptr) {
        OCR_TRY(net.lstm_counters.reserve((kLstmCounterCount + kLstmErrorWords) * sizeof(unsigned)));
        OCR_HIP_CHECK(hipMemsetAsync(net.lstm_counters.ptr, 0, (kLstmCounterCount + kLstmErrorWords) * sizeof(unsigned), s));
    }
    if (!net.lstm_err_host) {
        OCR_HIP_CHECK(hipHostMalloc(reinterpret_cast<void**>(&net.lstm_err_host), CrnnNet::kErrSlots * sizeof(unsigned), hipHostMallocDefault));
        for (int i = 0; i < CrnnNet::kErrSlots; ++i) net.lstm_err_host[i] = 0;
    }
    OCR_HIP_CHECK(hipMemsetAsync(net.lstm_counters.as<unsigned>() + kLstmCounterCount, 0, kLstmErrorWords * sizeof(unsigned), s));
    const uint16_t* layer_in = feats16;
    uint16_t* seq16 = net.seq16.as<uint16_t>();
    for (int l = 0; l < 2; ++l) {
        float* xw_f = net.xw.as<float>();
        float* xw_b = xw_f + (size_t)R * 4 * Hd;
        OCR_TRY(launch_gemm_bf16t(net.wih[l][0], layer_in, xw_f, true, R, 0, 0, 0, false, s));
        OCR_TRY(launch_gemm_bf16t(net.wih[l][1], layer_in, xw_b, true, R, 0, 0, 0, false, s));
        {
            ProfScope prof("lstm_layer16(T steps)", 2.0 * 2 * 4.0 * Hd * Hd * (double)(R - n),
                           4.0 * (2.0 * R * 4 * Hd) + 2.0 * (double)R * 2 * Hd + 2.0 * 2 * 4 * Hd * Hd * Tmax, s);
            int rc = launch_lstm_layer16(xw_f, xw_b, net.whh[l][0].as<float>(), net.whh[l][1].as<float>(), net.hbuf.as<uint16_t>(),
                                         net.cbuf.as<float>(), seq16, n, Tmax, Hd, d_T, d_row0, d_active, net.lstm_counters.as<unsigned>(), s);
            if (rc != OCRHIP_OK && rc != OCRHIP_ESTATE) return rc;
            if (rc == OCRHIP_ESTATE) {
                // one float32 launch per time step (other hidden sizes, lstm_persistent = 0, a device that cannot hold the
                // persistent grid): float32 h, the layer's output rounded to bfloat16 afterwards
                float* hb0 = net.hbuf.as<float>();
                float* hb1 = hb0 + (size_t)2 * n * Hd;
                int first = 0;
                for (int step = 0; step < Tmax; ++step) {
                    if (host_T)
                        while (first < n && host_T[first] <= step) ++first;
                    const float* hp = (step & 1) ? hb1 : hb0;
                    float* hn = (step & 1) ? hb0 : hb1;
                    OCR_TRY(launch_lstm_step(xw_f, xw_b, net.whh[l][0].as<float>(), net.whh[l][1].as<float>(), hp, hn,
                                             net.cbuf.as<float>(), net.seq.as<float>(), n, Tmax, step, d_T, d_row0, first, Hd, s));
                }
                OCR_TRY(launch_f32_to_bf16(net.seq.as<float>(), seq16, R * 2 * Hd, s));
            }
        }
        uint16_t* lin = net.lin.as<uint16_t>() + (size_t)l * R * Hd;
        OCR_TRY(launch_gemm_bf16t(net.linear[l], seq16, lin, false, R, 0, 0, 0, false, s));
        record(net.taps, l == 0 ? "seq0" : "seq1", lin, 1, R, Hd, 1, true);
        layer_in = lin;
    }
    OCR_TRY(launch_gemm_bf16t(net.pred, layer_in, logits, true, R, 0, 0, 0, false, s));
    if (net.calls_started - net.calls_checked >= (uint64_t)CrnnNet::kErrSlots) {
        OCR_HIP_CHECK(hipStreamSynchronize(s));
        OCR_TRY(crnn_report_timeout(net, s));
    }
    unsigned* slot = net.lstm_err_host + net.calls_started % CrnnNet::kErrSlots;
    *slot = CrnnNet::kErrPending;
    OCR_HIP_CHECK(hipMemcpyAsync(slot, net.lstm_counters.as<unsigned>() + kLstmCounterCount, sizeof(unsigned), hipMemcpyDeviceToHost, s));
    ++net.calls_started;
    return OCRHIP_OK;
}

// whether a forward call runs on the bfloat16-tensor kernels: bfloat16 asked for, the option on, the VGG extractor
// (the ResNet extractor keeps the float32-tensor kernels with rounded operands) and bfloat16 tile weights for every layer
static bool crnn_use_bf16_tensors(const ocrhip_ctx* h, int compute_dtype) {
    const CrnnNet& net = h->crnn;
    return compute_dtype == OCRHIP_BF16 && h->bf16_tensors && net.backbone == 0 && net.conv[1].w_bf16t.ptr && net.pred.w_bf16t.ptr &&
           net.wih[0][0].w_bf16t.ptr && net.hidden % 32 == 0;
}

int ocrhip_crnn_forward(ocrhip_handle h, const void* d_crops, int n, int Wpad, int compute_dtype, float* d_logits,
                        int* d_tokens, float* d_scores, ocrhip_stream stream) {
    OCR_HANDLE(h);
    CrnnNet& net = h->crnn;
    if (!net.loaded) {
        set_error("crnn_forward: weights not loaded");
        return OCRHIP_ESTATE;
    }
    OCR_REQUIRE(d_crops && n > 0 && Wpad >= 8 && Wpad % 4 == 0, "crnn_forward: bad arguments n=%d Wpad=%d", n, Wpad);
    OCR_TRY(crnn_report_timeout(net, (hipStream_t)stream));
    OCR_REQUIRE(compute_dtype == OCRHIP_F32 || compute_dtype == OCRHIP_BF16, "crnn_forward: compute dtype %d is neither F32 nor BF16", compute_dtype);
    Bf16Scope bf16_scope(compute_dtype);
    OCR_REQUIRE(d_logits || d_tokens, "crnn_forward: ask for logits and/or tokens");
    OCR_REQUIRE(!d_tokens || d_scores, "crnn_forward: d_scores is required with d_tokens");
    hipStream_t s = (hipStream_t)stream;
    const int Fc = net.feat_c;
    const int T = net.seq_len(Wpad);
    OCR_REQUIRE(T >= 1, "crnn_forward: crop too narrow");
    const int64_t R = (int64_t)n * T;
    OCR_TRY(net.feats.reserve((size_t)R * Fc * sizeof(float)));
    float* feats = net.feats.as<float>();
    net.taps.clear();
    if (crnn_use_bf16_tensors(h, compute_dtype)) {
        uint16_t* feats16 = net.feats.as<uint16_t>();
        OCR_TRY(crnn_features16(net, net.act, reinterpret_cast<const float*>(d_crops), n, Wpad, feats16, s));
        record(net.taps, "features", feats16, n, T, Fc, 1, true);
        float* lg = d_logits;
        if (!lg) {
            OCR_TRY(net.logits.reserve((size_t)R * net.vocab * sizeof(float)));
            lg = net.logits.as<float>();
        }
        OCR_TRY(crnn_sequence16(net, feats16, n, R, T, nullptr, nullptr, nullptr, nullptr, lg, s));
        if (d_tokens) OCR_TRY(launch_ctc_greedy(lg, n, T, net.vocab, net.vocab, nullptr, nullptr, 0, d_tokens, d_scores, s));
        return OCRHIP_OK;
    }
    OCR_TRY(crnn_features(net, net.act, reinterpret_cast<const float*>(d_crops), n, Wpad, feats, s));
    record(net.taps, "features", feats, n, T, Fc, 1);
    float* logits = d_logits;
    const bool fused = !logits && crnn_can_fuse_ctc(net);
    if (!logits && !fused) {
        OCR_TRY(net.logits.reserve((size_t)R * net.vocab * sizeof(float)));
        logits = net.logits.as<float>();
    }
    OCR_TRY(crnn_sequence(net, feats, n, R, T, nullptr, nullptr, nullptr, nullptr, logits, s));
    if (fused)
        OCR_TRY(launch_ctc_greedy_partial(net.pmax.as<float>(), net.pidx.as<int>(), gemm_f32_column_blocks(net.pred), n, T, nullptr,
                                          nullptr, 0, d_tokens, d_scores, s));
    else if (d_tokens)
        OCR_TRY(launch_ctc_greedy(logits, n, T, net.vocab, net.vocab, nullptr, nullptr, 0, d_tokens, d_scores, s));
    return OCRHIP_OK;
}

int ocrhip_crnn_forward_packed(ocrhip_handle h, const void* d_packed, const int* out_w, int n, int compute_dtype,
                               int t_stride, float* d_logits, int* d_tokens, float* d_scores, int* d_lengths,
                               ocrhip_stream stream) {
    OCR_HANDLE(h);
    CrnnNet& net = h->crnn;
    if (!net.loaded) {
        set_error("crnn_forward_packed: weights not loaded");
        return OCRHIP_ESTATE;
    }
    OCR_REQUIRE(d_packed && out_w && n > 0, "crnn_forward_packed: bad arguments n=%d", n);
    OCR_TRY(crnn_report_timeout(net, (hipStream_t)stream));
    OCR_REQUIRE(compute_dtype == OCRHIP_F32 || compute_dtype == OCRHIP_BF16, "crnn_forward_packed: compute dtype %d is neither F32 nor BF16", compute_dtype);
    Bf16Scope bf16_scope(compute_dtype);
    OCR_REQUIRE(d_logits || d_tokens, "crnn_forward_packed: ask for logits and/or tokens");
    OCR_REQUIRE(!d_tokens || d_scores, "crnn_forward_packed: d_scores is required with d_tokens");
    hipStream_t s = (hipStream_t)stream;
    const int Fc = net.feat_c;
    std::vector<int> meta((size_t)2 * n);        // [T of each crop | first row of each crop | crops with more than t steps, per t]
    int64_t R = 0;
    for (int i = 0; i < n; ++i) {
        OCR_REQUIRE(out_w[i] >= 8 && out_w[i] % 4 == 0 && (i == 0 || out_w[i] >= out_w[i - 1]),
                    "crnn_forward_packed: widths must be multiples of 4, >= 8 and non-decreasing (crop %d: %d)", i, out_w[i]);
        meta[i] = net.seq_len(out_w[i]);
        meta[n + i] = (int)R;
        R += meta[i];
        OCR_REQUIRE(R < (1ll << 31), "crnn_forward_packed: too many time steps in one batch");
    }
    const int Tmax = meta[n - 1];
    OCR_REQUIRE(!d_tokens || t_stride >= Tmax, "crnn_forward_packed: t_stride %d is smaller than the longest sequence %d", t_stride, Tmax);
    OCR_REQUIRE(Tmax >= 1, "crnn_forward_packed: crops too narrow");
    meta.resize((size_t)2 * n + Tmax);
    for (int t = 0, first = 0; t < Tmax; ++t) {
        while (first < n && meta[first] <= t) ++first;
        meta[(size_t)2 * n + t] = n - first;
    }
    OCR_TRY(net.meta.reserve(meta.size() * sizeof(int)));
    OCR_TRY(net.ring.upload(net.meta.ptr, meta.data(), meta.size() * sizeof(int), s));
    const int* d_T = net.meta.as<int>();
    const int* d_row0 = d_T + n;
    const int* d_active = d_row0 + n;
    if (d_lengths) OCR_HIP_CHECK(hipMemcpyAsync(d_lengths, d_T, (size_t)n * sizeof(int), hipMemcpyDeviceToDevice, s));

    OCR_TRY(net.feats.reserve((size_t)R * Fc * sizeof(float)));
    float* feats = net.feats.as<float>();
    net.taps.clear();
    // feature extractor: one pass per group of equal width, split so that a pass stays within the workspace budget
    const int64_t max_elems = net.backbone ? (1ll << 18) : (1ll << 20);      // n_group * Wpad per pass (2 GB per buffer at most)
    const float* x = reinterpret_cast<const float*>(d_packed);
    const bool tensors16 = crnn_use_bf16_tensors(h, compute_dtype);
    // Groups of equal width.  Where they are many and small (the lines of grouped text: a few crops per width), a group's
    // launches fill a fraction of the GPU, and groups go round-robin over the caller's stream and kAux side streams, each
    // with activation buffers of its own (sized before the first launch: no buffer grows while a launch may still read it)
    int n_groups = 0;
    size_t max_nw = 0;
    for (int i = 0; i < n;) {
        int j = i;
        while (j < n && out_w[j] == out_w[i]) ++j;
        max_nw = std::max(max_nw, (size_t)std::min<int64_t>((int64_t)(j - i) * out_w[i], std::max<int64_t>(max_elems, out_w[i])));
        ++n_groups;
        i = j;
    }
    const bool side = h->crnn_side_streams && !net.backbone && !h->wino4_split && n_groups >= 4 && n / n_groups < 256;
    if (side) {
        if (!net.aux_fork) OCR_HIP_CHECK(hipEventCreateWithFlags(&net.aux_fork, hipEventDisableTiming));
        for (int a = 0; a < CrnnNet::kAux; ++a) {
            if (!net.aux_stream[a]) OCR_HIP_CHECK(hipStreamCreateWithFlags(&net.aux_stream[a], hipStreamNonBlocking));
            if (!net.aux_join[a]) OCR_HIP_CHECK(hipEventCreateWithFlags(&net.aux_join[a], hipEventDisableTiming));
            OCR_TRY(net.act_aux[a][0].reserve(max_nw * 512 * sizeof(float)));
            OCR_TRY(net.act_aux[a][1].reserve(max_nw * 256 * sizeof(float)));
        }
        OCR_TRY(net.act[0].reserve(max_nw * 512 * sizeof(float)));
        OCR_TRY(net.act[1].reserve(max_nw * 256 * sizeof(float)));
        OCR_HIP_CHECK(hipEventRecord(net.aux_fork, s));          // the crops and the uploads above are in place
        for (int a = 0; a < CrnnNet::kAux; ++a) OCR_HIP_CHECK(hipStreamWaitEvent(net.aux_stream[a], net.aux_fork, 0));
    }
    int64_t off = 0;
    int g = 0;
    for (int i = 0; i < n; ++g) {
        const int w = out_w[i];
        int j = i;
        while (j < n && out_w[j] == w) ++j;
        const int lane_id = side ? g % (CrnnNet::kAux + 1) : 0;          // 0: the caller's stream
        hipStream_t gs = lane_id ? net.aux_stream[lane_id - 1] : s;
        DevBuf* act = lane_id ? net.act_aux[lane_id - 1] : net.act;
        const int per_pass = (int)std::max<int64_t>(1, max_elems / w);
        for (int c0 = i; c0 < j; c0 += per_pass) {
            const int c1 = std::min(j, c0 + per_pass);
            if (tensors16)
                OCR_TRY(crnn_features16(net, act, x + off, c1 - c0, w, net.feats.as<uint16_t>() + (size_t)meta[n + c0] * Fc, gs));
            else
                OCR_TRY(crnn_features(net, act, x + off, c1 - c0, w, feats + (size_t)meta[n + c0] * Fc, gs));
            off += (int64_t)(c1 - c0) * 64 * w;
        }
        i = j;
    }
    if (side)
        for (int a = 0; a < CrnnNet::kAux; ++a) {
            OCR_HIP_CHECK(hipEventRecord(net.aux_join[a], net.aux_stream[a]));
            OCR_HIP_CHECK(hipStreamWaitEvent(s, net.aux_join[a], 0));
        }
    record(net.taps, "features", feats, 1, R, Fc, 1, tensors16);
    if (tensors16) {
        float* lg = d_logits;
        if (!lg) {
            OCR_TRY(net.logits.reserve((size_t)R * net.vocab * sizeof(float)));
            lg = net.logits.as<float>();
        }
        OCR_TRY(crnn_sequence16(net, net.feats.as<uint16_t>(), n, R, Tmax, meta.data(), d_T, d_row0, d_active, lg, s));
        if (d_tokens) OCR_TRY(launch_ctc_greedy(lg, n, t_stride, net.vocab, net.vocab, d_T, d_row0, 0, d_tokens, d_scores, s));
        return OCRHIP_OK;
    }
    float* logits = d_logits;
    const bool fused = !logits && crnn_can_fuse_ctc(net);
    if (!logits && !fused) {
        OCR_TRY(net.logits.reserve((size_t)R * net.vocab * sizeof(float)));
        logits = net.logits.as<float>();
    }
    OCR_TRY(crnn_sequence(net, feats, n, R, Tmax, meta.data(), d_T, d_row0, d_active, logits, s));
    if (fused)
        OCR_TRY(launch_ctc_greedy_partial(net.pmax.as<float>(), net.pidx.as<int>(), gemm_f32_column_blocks(net.pred), n, t_stride, d_T,
                                          d_row0, 0, d_tokens, d_scores, s));
    else if (d_tokens)
        OCR_TRY(launch_ctc_greedy(logits, n, t_stride, net.vocab, net.vocab, d_T, d_row0, 0, d_tokens, d_scores, s));
    return OCRHIP_OK;
}

int ocrhip_check_errors(ocrhip_handle h, ocrhip_stream stream, int synchronize) {
    OCR_HANDLE(h);
    if (synchronize) OCR_HIP_CHECK(hipStreamSynchronize((hipStream_t)stream));
    return crnn_report_timeout(h->crnn, (hipStream_t)stream);
}

int ocrhip_crnn_get_activation(ocrhip_handle h, const char* name, float* dst, int64_t dst_capacity, int64_t shape[4]) {
    OCR_HANDLE(h);
    return read_activation(h->crnn.taps, name, dst, dst_capacity, shape);
}

int ocrhip_ctc_greedy(ocrhip_handle h, const float* d_logits, int B, int T, int V, const int* d_lengths, int blank,
                      int* d_tokens, float* d_scores, ocrhip_stream stream) {
    OCR_HANDLE(h);
    return launch_ctc_greedy(d_logits, B, T, V, V, d_lengths, nullptr, blank, d_tokens, d_scores, (hipStream_t)stream);
}

int ocrhip_ctc_beam_search(ocrhip_handle h, const float* d_logits, int B, int T, int V, const int* d_lengths, const int* d_row0,
                           int blank, int num_beams, int* d_tokens, float* d_scores, ocrhip_stream stream) {
    OCR_HANDLE(h);
    return launch_ctc_beam(h, d_logits, B, T, V, d_lengths, d_row0, blank, num_beams, d_tokens, d_scores, (hipStream_t)stream);
}

int ocrhip_set_option(ocrhip_handle h, const char* name, int value) {
    OCR_HANDLE(h);
    OCR_REQUIRE(name, "set_option: null name");
    if (std::strcmp(name, "conv3x3_winograd") == 0) {
        OCR_REQUIRE(value >= 0 && value <= 2, "set_option: conv3x3_winograd is 0, 1 or 2");
        h->conv3x3_winograd = value;
        return OCRHIP_OK;
    }
    if (std::strcmp(name, "lstm_persistent") == 0) {
        OCR_REQUIRE(value >= 0 && value <= 3, "set_option: lstm_persistent is 0, 1, 2 or 3");
        h->lstm_persistent = value;
        return OCRHIP_OK;
    }
    if (std::strcmp(name, "bf16_fuse_first") == 0) {
        OCR_REQUIRE(value == 0 || value == 1, "set_option: bf16_fuse_first is 0 or 1");
        h->bf16_fuse_first = value;
        return OCRHIP_OK;
    }
    if (std::strcmp(name, "crnn_side_streams") == 0) {
        OCR_REQUIRE(value == 0 || value == 1, "set_option: crnn_side_streams is 0 or 1");
        h->crnn_side_streams = value;
        return OCRHIP_OK;
    }
    if (std::strcmp(name, "lstm16_units") == 0) {
        OCR_REQUIRE(value == 32 || value == 64, "set_option: lstm16_units is 32 or 64");
        h->lstm16_units = value;
        return OCRHIP_OK;
    }
    if (std::strcmp(name, "lstm_debug_drop_handover") == 0) {
        OCR_REQUIRE(value == 0 || value == 1, "set_option: lstm_debug_drop_handover is 0 or 1");
        h->lstm_debug_drop = value;
        return OCRHIP_OK;
    }
    if (std::strcmp(name, "wino4_split") == 0) {
        OCR_REQUIRE(value == 0 || value == 1, "set_option: wino4_split is 0 or 1");
        h->wino4_split = value;
        return OCRHIP_OK;
    }
    if (std::strcmp(name, "first_layer_mfma") == 0) {
        OCR_REQUIRE(value == 0 || value == 1, "set_option: first_layer_mfma is 0 or 1");
        h->first_layer_mfma = value;
        return OCRHIP_OK;
    }
    if (std::strcmp(name, "wino4_wave_tiles") == 0) {
        OCR_REQUIRE(value >= 0 && value <= 2, "set_option: wino4_wave_tiles is 0, 1 or 2");
        h->wino4_wave_tiles = value;
        return OCRHIP_OK;
    }
    if (std::strcmp(name, "wino4_side_by_side") == 0) {
        OCR_REQUIRE(value == 0 || value == 1, "set_option: wino4_side_by_side is 0 or 1");
        h->wino4_side_by_side = value;
        return OCRHIP_OK;
    }
    if (std::strcmp(name, "fuse_mean_rows") == 0) {
        OCR_REQUIRE(value == 0 || value == 1, "set_option: fuse_mean_rows is 0 or 1");
        h->fuse_mean_rows = value;
        return OCRHIP_OK;
    }
    if (std::strcmp(name, "fuse_ctc") == 0) {
        OCR_REQUIRE(value == 0 || value == 1, "set_option: fuse_ctc is 0 or 1");
        h->fuse_ctc = value;
        return OCRHIP_OK;
    }
    if (std::strcmp(name, "bf16_tensors") == 0) {
        OCR_REQUIRE(value == 0 || value == 1, "set_option: bf16_tensors is 0 or 1");
        h->bf16_tensors = value;
        return OCRHIP_OK;
    }
    if (std::strcmp(name, "flat_gemm") == 0) {
        OCR_REQUIRE(value == 0 || value == 1, "set_option: flat_gemm is 0 or 1");
        h->flat_gemm = value;
        return OCRHIP_OK;
    }
    set_error("set_option: unknown option '%s'", name);
    return OCRHIP_EINVAL;
}

int ocrhip_profile_enable(ocrhip_handle h, int enable) {
    OCR_HANDLE(h);
    OCR_HIP_CHECK(hipDeviceSynchronize());
    h->prof.recycle();
    h->prof.enabled = enable != 0;
    return OCRHIP_OK;
}

int ocrhip_profile_only(ocrhip_handle h, const char* name) {
    OCR_HANDLE(h);
    OCR_HIP_CHECK(hipDeviceSynchronize());
    h->prof.only = name ? name : "";
    return OCRHIP_OK;
}

int ocrhip_profile_collect(ocrhip_handle h, ocrhip_kernel_stat* out, int cap, int* n) {
    OCR_HANDLE(h);
    OCR_REQUIRE(n && (out || cap == 0), "profile_collect: null argument");
    OCR_HIP_CHECK(hipDeviceSynchronize());
    std::map<std::string, ocrhip_kernel_stat> agg;
    for (auto& r : h->prof.records) {
        float ms = 0.f;
        if (hipEventElapsedTime(&ms, r.e0, r.e1) != hipSuccess) continue;
        auto it = agg.find(r.name);
        if (it == agg.end()) {
            ocrhip_kernel_stat st;
            std::memset(&st, 0, sizeof(st));
            std::snprintf(st.name, sizeof(st.name), "%s", r.name.c_str());
            it = agg.insert({r.name, st}).first;
        }
        it->second.launches += 1;
        it->second.total_ms += ms;
        it->second.flops += r.flops;
        it->second.bytes += r.bytes;
    }
    std::vector<ocrhip_kernel_stat> v;
    for (auto& kv : agg) v.push_back(kv.second);
    std::sort(v.begin(), v.end(), [](const ocrhip_kernel_stat& a, const ocrhip_kernel_stat& b) { return a.total_ms > b.total_ms; });
    *n = (int)v.size();
    for (int i = 0; i < (int)v.size() && i < cap; ++i) out[i] = v[i];
    return OCRHIP_OK;
}

}  // extern "C"
