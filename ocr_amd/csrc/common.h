// Shared declarations of libocrhip.so (host side + launch helpers).  gfx950 only.
#pragma once

#include <hip/hip_runtime.h>

#include <cstdarg>
#include <cstdint>
#include <cstdio>
#include <map>
#include <string>
#include <vector>

#include "../../include/ocrhip.h"

namespace ocrhip {

void set_error(const char* fmt, ...);

#define OCR_HIP_CHECK(expr)                                                                   \
    do {                                                                                      \
        hipError_t _e = (expr);                                                               \
        if (_e != hipSuccess) {                                                               \
            ocrhip::set_error("%s failed: %s (%s:%d)", #expr, hipGetErrorString(_e), __FILE__, \
                              __LINE__);                                                      \
            return OCRHIP_EHIP;                                                               \
        }                                                                                     \
    } while (0)

#define OCR_REQUIRE(cond, ...)              \
    do {                                    \
        if (!(cond)) {                      \
            ocrhip::set_error(__VA_ARGS__); \
            return OCRHIP_EINVAL;           \
        }                                   \
    } while (0)

#define OCR_TRY(expr)               \
    do {                            \
        int _rc = (expr);           \
        if (_rc != OCRHIP_OK) return _rc; \
    } while (0)

// Event-pair records of the stage timers (ocrhip_profile_*).
struct ProfRecord {
    std::string name;
    double flops, bytes;
    hipEvent_t e0, e1;
};
struct Profiler {
    bool enabled = false;
    std::string only;          // when not empty: only launches recorded under this name are timed
    std::vector<ProfRecord> records;
    std::vector<hipEvent_t> pool;
    hipEvent_t get_event();
    void recycle();
};
Profiler* current_profiler();
void set_current_profiler(Profiler* p);

#if defined(__HIPCC__)
// 16-byte buffer store whose data registers stay untouched for two more cycles.  The store reads the upper part of its
// 128-bit data after it has issued; a vector instruction right behind it that overwrites those registers (the compiler
// likes to reuse them for the next address) changed what reached memory.  The compiler only knows this hazard for stores
// without a scalar offset register; the s_nop takes the data as an operand, so nothing can be scheduled between the two
// that writes them.  (Stores of 8 bytes or less are not affected.)
typedef float ocr_floatx4 __attribute__((ext_vector_type(4)));
// AUX: cache policy bits of the store; kStreamingStore marks data that the kernel does not read again (a layer's
// output): those lines are the first to leave the L2, which keeps the filters and the neighbours' halo pixels there.
constexpr int kStreamingStore = 2;          // nt
template <int AUX = 0>
__device__ __forceinline__ void buffer_store_b128_guarded(ocr_floatx4 data, __amdgpu_buffer_rsrc_t rsrc, int voffset, int soffset) {
    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(__attribute__((__vector_size__(4 * sizeof(unsigned)))) unsigned, data), rsrc,
                                           voffset, soffset, AUX);
    asm volatile("s_nop 1" ::"v"(data));
}
#endif

// RAII: records an event before and after the launches made in its scope (no-op when disabled).
struct ProfScope {
    Profiler* p;
    hipStream_t s;
    size_t idx;
    ProfScope(const char* name, double flops, double bytes, hipStream_t stream);
    ~ProfScope();
};

// Grow-only device buffer owned by a handle.
struct DevBuf {
    void* ptr = nullptr;
    size_t bytes = 0;
    int reserve(size_t need);
    void release();
    template <class T>
    T* as() const { return reinterpret_cast<T*>(ptr); }
};

// Small host->device parameter uploads without blocking the host: a ring of pinned slots, each
// guarded by an event recorded after the copy that last used it.
struct PinnedRing {
    static constexpr int kSlots = 4;
    void* host[kSlots] = {nullptr, nullptr, nullptr, nullptr};
    size_t bytes[kSlots] = {0, 0, 0, 0};
    hipEvent_t done[kSlots] = {nullptr, nullptr, nullptr, nullptr};
    int next = 0;
    int upload(void* dst, const void* src, size_t n, hipStream_t s);
    void release();
};

// One convolution / GEMM layer after batch-norm folding, in the device layouts the kernels read.
struct ConvLayer {
    int cin = 0, cout = 0, ks = 0;
    int cout_pad = 0;       // cout rounded up to the kernel's N tile
    DevBuf w_f32;           // MFMA slab layout (see conv_mfma.hip) or direct layout (see conv_direct.hip)
    DevBuf w_bf16;          // bfloat16 slab of the same layer (see conv_bf16.hip), when cin % 16 == 0
    DevBuf w_bf16t;         // bfloat16 LDS images per (output tile, chunk, tap) for the bfloat16-tensor kernels (see conv_bf16t.hip), when cin % 32 == 0
    DevBuf w_wino;          // 3x3 layers with >= 64 output channels: Winograd-transformed filters (see conv_wino.hip)
    DevBuf w_wino4;         // the same layers: F(4x4,3x3)-transformed filters (see conv_wino4.hip)
    DevBuf w_wino4c;        // of those, layers with a multiple of 128 output channels: the layout of its MODE 4
    DevBuf w_wino4b;        // layers with < 256 input channels: the same filters in the layout of that kernel's MODE 3
    DevBuf bias;            // cout_pad floats, folded
    DevBuf w_gemm, bias_gemm;   // 1x1 and 2x2 layers: slabs and bias for the 256 x 128 flat GEMM (see gemm_f32.hip)
    bool direct = false;    // small-Cin direct kernel layout
};

struct Activation {
    const void* ptr = nullptr;
    int64_t shape[4] = {0, 0, 0, 0};
    bool bf16 = false;       // a bfloat16 tensor (the bfloat16-tensor path); read back as float32
};

struct EastNet {
    bool loaded = false;
    ConvLayer vgg[13];
    ConvLayer merge[7];
    ConvLayer head;          // 32 -> 6 (score, 4 geo, angle), direct layout [6][32] + bias[6]
    DevBuf act[2];           // ping-pong activations
    DevBuf pool[5];          // pool1..pool5 taps (pool2..4 feed the merge branch)
    std::map<std::string, Activation> taps;
};

struct LstmDir {
    DevBuf w_ih;             // GEMM slab layout, 256 -> 1024
    DevBuf bias;             // b_ih + b_hh, 1024
    DevBuf w_hh;             // (1024,256) float32 as in the checkpoint
};

struct CrnnNet {
    bool loaded = false;
    int vocab = 0, vocab_pad = 0, hidden = 256;
    int backbone = 0;        // 0: VGG extractor (256 features, T = W/4 - 1); 1: ResNet extractor (512 features, T = W/4 + 1)
    int feat_c = 256;
    int seq_len(int wpad) const { return wpad / 4 + (backbone ? 1 : -1); }
    ConvLayer conv[7];       // VGG: ConvNet.0,3,6,8,11,14,18
    ConvLayer rconv[32];     // ResNet: in execution order, see crnn_load_resnet
    ConvLayer wih[2][2];     // [layer][dir] input projections feat_c / 256 -> 1024 (bias = b_ih + b_hh)
    DevBuf whh[2][2];        // [layer][dir] (1024,256) float32
    ConvLayer linear[2];     // 512 -> 256
    ConvLayer pred;          // 256 -> vocab
    DevBuf act[3];
    // many small groups of equal-width crops (lines of text): the feature extractor of a group runs on one of kAux side
    // streams with activation buffers of its own, so that launches which fill a fraction of the GPU overlap
    static constexpr int kAux = 3;
    DevBuf act_aux[kAux][2];
    hipStream_t aux_stream[kAux] = {nullptr, nullptr, nullptr};
    hipEvent_t aux_fork = nullptr, aux_join[kAux] = {nullptr, nullptr, nullptr};
    DevBuf feats, xw, hbuf, cbuf, seq, lin, logits, meta, lstm_counters;
    DevBuf seq16;            // bfloat16-tensor path: output rows of a BiLSTM layer (R, 2H)
    DevBuf pmax, pidx;       // greedy decoding without logits: per row and block of 128 classes the maximum and its class
    static constexpr int kErrSlots = 16;
    static constexpr unsigned kErrPending = 0xFFFFFFFFu;
    unsigned* lstm_err_host = nullptr;   // pinned ring: the error word of forward number k lands in slot k % kErrSlots behind its results
    uint64_t calls_started = 0, calls_checked = 0;   // forwards enqueued / forwards whose status slot has been read
    PinnedRing ring;
    std::map<std::string, Activation> taps;
};

// ctc_beam.hip: CTC prefix beam search, one wave per sequence (row0 as for launch_ctc_greedy)
int launch_ctc_beam(ocrhip_ctx* h, const float* logits, int B, int T, int V, const int* lengths, const int* row0, int blank,
                    int num_beams, int* tokens, float* scores, hipStream_t s);

}  // namespace ocrhip

struct ocrhip_ctx {
    int device = 0;
    ocrhip::EastNet east;
    ocrhip::CrnnNet crnn;
    ocrhip::DevBuf decode_stage;   // staging of the decode kernel
    ocrhip::DevBuf lanms_ws;       // scratch of the LANMS kernel
    ocrhip::DevBuf beam_ws;        // prefix tree pools of the CTC beam search
    int lstm_debug_drop = 0;       // test hook: the persistent recurrence withholds one hand-over and times out quickly
    int lstm_persistent = 2;       // recurrence of a BiLSTM layer in one persistent launch: independent waves on blocks of 16 sequences where a workgroup has at least 8 blocks, workgroup-wide passes of 112 sequences otherwise (2), passes always (1), independent waves always (3); or one launch per step (0)
    int bf16_fuse_first = 1;       // bfloat16-tensor detector: the first two layers (3 -> 64, 64 -> 64 + pool) in one launch
    int crnn_side_streams = 1;     // recogniser: feature extractor of small width groups on side streams (1) or all on the caller's stream (0)
    int lstm16_units = 64;         // bfloat16 recurrence: hidden units per workgroup (128 KB of W_hh in LDS, 4 workgroups share a sequence group) or 32
    int conv3x3_winograd = 2;      // 3x3 convolutions by Winograd F(2x2,3x3) (1) or as direct implicit GEMM (0)
    int first_layer_mfma = 1;      // the detector's 3 -> 64 first layer on the matrix cores (1) or on the vector ALU (0); identical results
    int wino4_wave_tiles = 2;      // fused F(4x4,3x3) layers: waves own whole (block, channel) tiles of 8 x 32 outputs, inverse transform in registers: workgroups of 4 waves x 16 channels, two per CU (1); the same, and workgroups of 8 waves x 16 channels where the layer has a multiple of 128 output channels (2); or the 3x3-component split with staged partial sums on 16 x 32 tiles (0)
    int wino4_side_by_side = 1;    // fused F(4x4,3x3) layers: a batch of images whose width leaves the 32-column tiles partly empty is tiled as one wide image, a zero column apart (1) or image by image (0)
    int wino4_split = 0;           // F(4x4,3x3) layers with >= 512 input channels: fused like the others (0), or transform once into `wino_ws` and multiply from it (1)
    ocrhip::DevBuf wino_ws;        // transformed tiles of the layer in flight
    int fuse_mean_rows = 1;        // float32 recogniser: the mean over the feature rows in the epilogue of the extractor's last convolution (1) or as a kernel of its own (0)
    int fuse_ctc = 1;              // tokens without a logits request: class layer reduced to per-block maxima in its epilogue (1) or logits written and re-read (0)
    int bf16_tensors = 1;          // compute_dtype = OCRHIP_BF16: bfloat16 activations in HBM, tile kernels, bfloat16 recurrence (1); or the float32-tensor kernels with operands rounded on their way into LDS and a float32 recurrence (0)
    int flat_gemm = 1;             // 1x1 / 2x2 layers on the 256 x 128 flat GEMM (1) or on the implicit-GEMM convolution kernel (0)
    ocrhip::Profiler prof;
};

namespace ocrhip {

// ---- kernels (one launcher per .hip file) -------------------------------------------------------

// conv_mfma.hip: implicit-GEMM convolution on f32 MFMA, NHWC float32.
//   ks 3 (pad 1), 2 (valid) or 1 (flat GEMM over B*H*W rows).
//   pool: 0 none, 1 max 2x2, 2 max (2,1) (rows only) fused into the epilogue.
//   up_src != nullptr: the first up_c input channels are the x2 align-corners bilinear upsampling of
//   up_src (B,H/2,W/2,up_c) and the remaining cin - up_c channels come from x (B,H,W,cin-up_c).
int prepare_conv_mfma(ConvLayer& L, const float* w_oihw, const float* bias, int cout, int cin, int ks);
//   residual != nullptr (unpooled ks 3 / 2 only): a tensor shaped like y, added before the ReLU.
int launch_conv_mfma(const ConvLayer& L, const float* x, float* y, int B, int H, int W, int pool, bool relu,
                     const float* up_src, int up_c, hipStream_t s, const float* residual = nullptr);

// conv_bf16.hip: the same layers with bfloat16 operands (float32 accumulation and tensors); launch_conv_mfma forwards to
// it while compute_bf16() is set (an API call with compute_dtype = OCRHIP_BF16 sets it for its duration).
int prepare_conv_bf16(ConvLayer& L, const float* w_oihw, int cout, int cin, int ks);
int launch_conv_bf16(const ConvLayer& L, const float* x, float* y, int B, int H, int W, int pool, bool relu,
                     const float* up_src, int up_c, hipStream_t s, const float* residual);
int compute_bf16();
void set_compute_bf16(int on);

// conv_bf16t.hip: bfloat16 tensors in HBM (the path of compute_dtype = OCRHIP_BF16 for the detector and the VGG
// recogniser): 3x3 convolutions (optionally over `nin` inner images tiled side by side, `gap` zero columns apart) and
// flat GEMMs (ks 1; ks 2 = 2x2 VALID convolution over (n, fh, fw, cin) images, rows = output pixels).
int prepare_conv_bf16t(ConvLayer& L, const float* w_oihw, int cout, int cin, int ks);
int prepare_first_pair_bf16t(ConvLayer& first, const float* w_oihw);
int launch_conv3_first_pair_bf16t(const ConvLayer& first, const ConvLayer& L, const void* page, bool page_u8, uint16_t* y, int B, int H,
                                  int W, hipStream_t s);
int launch_conv3_bf16t(const ConvLayer& L, const uint16_t* x, uint16_t* y, int B, int nin, int H, int Wc, int gap, int pool,
                       bool relu, hipStream_t s);
int launch_gemm_bf16t(const ConvLayer& L, const uint16_t* x, void* y, bool y_f32, int64_t M, int n, int fh, int fw, bool relu,
                      hipStream_t s);

// conv_wino.hip: the same 3x3 layers by Winograd F(2x2,3x3) (cout_pad % 64 == 0, cin % 8 == 0); launch_conv_mfma
// forwards to it when the handle's "conv3x3_winograd" option is on (the default).
int prepare_conv_wino(ConvLayer& L, const float* w_oihw, int cout, int cin);
int launch_conv_wino(const ConvLayer& L, const float* x, float* y, int B, int H, int W, int pool, bool relu,
                     const float* residual, hipStream_t s);
// conv_wino4.hip: F(4x4,3x3) for the launches whose maps fill its 16 x 32 tiles (option "conv3x3_winograd" = 2)
int prepare_conv_wino4(ConvLayer& L, const float* w_oihw, int cout, int cin);
bool conv_wino4_fits(const ConvLayer& L, int B, int H, int W, int pool);
int launch_conv_wino4(const ConvLayer& L, const float* x, float* y, int B, int H, int W, int pool, bool relu,
                      const float* residual, hipStream_t s);
DevBuf* current_workspace();         // grow-only scratch of the handle in use on this thread (transformed tiles), or nullptr
void set_current_workspace(DevBuf* ws);
int wino4_wave_tiles();              // option of the handle in use on this thread
int first_layer_mfma();
void set_first_layer_mfma(int on);
void set_wino4_wave_tiles(int on);
int wino4_side_by_side_enabled();    // option of the handle in use on this thread
void set_wino4_side_by_side(int on);
int wino4_split();                   // option of the handle in use on this thread
void set_wino4_split(int on);
int conv3x3_winograd();              // option of the handle in use on this thread
void set_conv3x3_winograd(int on);

// gemm_f32.hip: 1x1 and 2x2 layers as a flat GEMM with segmented A rows; launch_conv_mfma forwards to it when the
// handle's "flat_gemm" option is on (the default) and the layer has neither pooling, upsampling nor a residual.
int prepare_gemm_f32(ConvLayer& L, const float* w_oihw, const float* bias, int cout, int cin, int ks);
int launch_gemm_f32(const ConvLayer& L, const float* x, float* y, int B, int H, int W, bool relu, hipStream_t s);
int launch_gemm_f32_mean_rows(const ConvLayer& L, const float* x, float* y, int B, int H, int W, bool relu, hipStream_t s);
// Dense(vocab) in front of greedy CTC: per row and block of 128 columns the largest value and its column, no logits
int gemm_f32_column_blocks(const ConvLayer& L);
int launch_gemm_f32_argmax(const ConvLayer& L, const float* x, int64_t M, float* pmax, int* pidx, hipStream_t s);
int flat_gemm();                     // option of the handle in use on this thread
void set_flat_gemm(int on);

// conv_direct.hip: first layers (Cin 3 or 1) and the 32->6 sigmoid heads.
int prepare_conv_direct(ConvLayer& L, const float* w_oihw, const float* bias, int cout, int cin, int ks);
//   out_bf16: y is a bfloat16 tensor (the bfloat16-tensor path; matrix-core kernels only)
int launch_conv_first(const ConvLayer& L, const void* x, bool x_is_u8, void* y, int B, int H, int W, int pool,
                      hipStream_t s, bool out_bf16 = false);
int launch_east_heads(const ConvLayer& L, const float* x, float* maps, int64_t npix, hipStream_t s);

// bf16_misc.hip: bandwidth-bound pieces of the bfloat16-tensor path
int launch_upcat16(const uint16_t* up, const uint16_t* res, uint16_t* out, int B, int H, int W, int up_c, int res_c, hipStream_t s);
int launch_east_heads16(const ConvLayer& L, const uint16_t* x, float* maps, int64_t npix, hipStream_t s);
int launch_mean_rows16(const uint16_t* x, uint16_t* y, int B, int H, int W, int C, hipStream_t s);
int launch_f32_to_bf16(const float* x, uint16_t* y, int64_t n, hipStream_t s);
int launch_bf16_to_f32(const uint16_t* x, float* y, int64_t n, hipStream_t s);

// decode.hip
int launch_east_decode(ocrhip_ctx* h, const float* maps, int B, int mh, int mw, int in_h, int in_w, float thr,
                       int cap, float* xyxy, float* polys, float* scores, int* pix, int* counts, hipStream_t s);

// decode.hip: the same decode with its staging area in a fixed device array instead of the handle's grow-only buffer
// (no allocation in the call; ECAP beyond the array's size).  decode_static_init looks the array's address up once per
// device (ocrhip_create calls it with that device current).
int decode_static_init(int device);
int launch_east_decode_static(ocrhip_ctx* h, const float* maps, int B, int mh, int mw, int in_h, int in_w, float thr,
                              int cap, float* xyxy, float* polys, float* scores, int* pix, int* counts, hipStream_t s);

// lanms.hip
int launch_lanms(ocrhip_ctx* h, const float* xyxy, const float* scores, const int* counts, int B, int cap,
                 float merge_thr, float nms_thr, int merge_method, int max_iter, int max_out, float* merged,
                 uint8_t* keep, float* out_boxes, float* out_scores, int* out_src, int* out_counts, int* n_iter,
                 hipStream_t s);

// crop.hip
int launch_rotated_nms(ocrhip_ctx* h, const float* polys, const float* scores, const int* counts, int B, int cap,
                       float iou_threshold, int max_out, float* out_polys, float* out_scores, int* out_index,
                       int* out_counts, hipStream_t s);
int launch_crop_resize_pack(const uint8_t* img, int B, int H, int W, const void* rects, bool rotated, const int* box_img,
                            const int* inter, const int* out_w, const int64_t* dst_off, int n, int out_h,
                            void* packed, int out_dtype, hipStream_t s);

// pool.hip: zero padding of the width by pad_w on both sides, then max pooling (window ph x pw, strides sh x sw,
// no further padding).  A 1x1 window with unit strides is a plain zero-padding copy.
int launch_pad_maxpool(const float* x, float* y, int B, int H, int W, int C, int ph, int pw, int sh, int sw, int pad_w,
                       hipStream_t s);

// crop.hip: whole-page `/255` (uint8 only) -> antialiased stretch -> (x-0.5)/0.5, (B,H,W,3) -> float32 (B,oh,ow,3)
int launch_resize_rgb_norm(const void* img, bool is_u8, int B, int H, int W, int oh, int ow, float* out, hipStream_t s);
int launch_dice_coeff(const float* x, const float* y, int64_t n, int64_t stride, float* out, hipStream_t s);

// rnn.hip
int launch_mean_rows(const float* x, float* y, int B, int H, int W, int C, hipStream_t s);
int launch_lstm_step(const float* xw_f, const float* xw_b, const float* whh_f, const float* whh_b, const float* h_prev,
                     float* h_next, float* c_state, float* out, int n, int T, int step, const int* seq_T,
                     const int* seq_row0, int first, int H, hipStream_t s);

// rnn_persistent.hip: a whole bidirectional layer in one launch of persistent workgroups (hbuf: 2 parities x 2 directions x n x 256,
// counters: kLstmCounterCount unsigned ints cleared per launch, followed by kLstmErrorWords error words the kernel
// sets when a hand-over between workgroups times out and the caller clears at the start of every forward).  OCRHIP_ESTATE when the hidden size is not 256, the device cannot
// hold the grid at once or a workgroup would need more passes than it has counters (use the per-step kernel).
constexpr int kLstmCounterCount = 2 * 32 * 64;
constexpr int kLstmErrorWords = 16;
int launch_lstm_layer(const float* xw_f, const float* xw_b, const float* whh_f, const float* whh_b, float* hbuf, float* cbuf,
                      float* out, int n, int T, int H, const int* seq_T, const int* seq_row0, const int* step_active,
                      unsigned* counters, hipStream_t s);
int launch_lstm_layer16(const float* xw_f, const float* xw_b, const float* whh_f, const float* whh_b, uint16_t* hbuf16, float* cbuf,
                        uint16_t* out16, int n, int T, int H, const int* seq_T, const int* seq_row0, const int* step_active,
                        unsigned* counters, hipStream_t s);
int lstm_debug_drop();               // test hook of the handle in use on this thread: withhold one hand-over (see ocrhip.h)
void set_lstm_debug_drop(int on);
int lstm_persistent();               // option of the handle in use on this thread
void set_lstm_persistent(int on);
int lstm16_units();                  // hidden units per workgroup of the bfloat16 recurrence (32 or 64), option of the handle in use
void set_lstm16_units(int u);

// ctc.hip
// row0 != nullptr: sequence b starts at row row0[b] of a flat (sum T, ld_v) logits matrix; T is then the
// stride of the token rows.
int launch_ctc_greedy(const float* logits, int B, int T, int V, int ld_v, const int* lengths, const int* row0, int blank,
                      int* tokens, float* scores, hipStream_t s);
// the same decoding from the (maximum, column) pairs per block of 128 columns that launch_gemm_f32_argmax writes
int launch_ctc_greedy_partial(const float* pmax, const int* pidx, int n_blocks, int B, int T, const int* lengths,
                              const int* row0, int blank, int* tokens, float* scores, hipStream_t s);

// ctc_beam.hip: CTC prefix beam search, one wave per sequence (row0 as for launch_ctc_greedy)
int launch_ctc_beam(ocrhip_ctx* h, const float* logits, int B, int T, int V, const int* lengths, const int* row0, int blank,
                    int num_beams, int* tokens, float* scores, hipStream_t s);

}  // namespace ocrhip
