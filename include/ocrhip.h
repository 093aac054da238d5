/*
 * ocrhip.h - C ABI of libocrhip.so: the OCR inference hot path on MI355X (gfx950).
 *
 *   image -> EAST-VGG16 -> RBOX decode -> locality-aware NMS -> crop / resize -> CRNN -> CTC greedy
 *
 * Every entry point replaces a piece of Python / Keras / NumPy in the reference repository; the
 * reference interface each one stands in for is cited as path:line (paths relative to the
 * reference checkout).  Conventions:
 *
 *   - return 0 (OCRHIP_OK) or a negative OCRHIP_E* code; ocrhip_last_error() gives the text
 *     (thread-local);
 *   - every pointer named "d_*" or documented as "device" is device memory of the handle's GPU;
 *     everything else is host memory;
 *   - calls are asynchronous on the given stream (a hipStream_t passed as void*, NULL = the
 *     default stream); nothing synchronises the device unless stated;
 *   - the library never frees or keeps caller buffers; outputs are caller-allocated;
 *   - one handle per (device, host thread); a handle owns the network weights and a grow-only
 *     activation workspace (hipMalloc happens in *_load_weights and when a forward call needs a
 *     larger workspace than any call before it).
 */
#ifndef OCRHIP_H
#define OCRHIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct ocrhip_ctx* ocrhip_handle;
typedef void* ocrhip_stream; /* hipStream_t */

enum {
    OCRHIP_OK = 0,
    OCRHIP_EINVAL = -1, /* bad argument / shape the kernels do not support               */
    OCRHIP_ECAP = -2,   /* a caller-provided capacity is too small                        */
    OCRHIP_EHIP = -3,   /* a HIP runtime call failed                                      */
    OCRHIP_ESTATE = -4  /* weights not loaded / handle not usable                         */
};

enum { OCRHIP_F32 = 0, OCRHIP_BF16 = 1, OCRHIP_U8 = 2 };
/* UNION and AVERAGE are the reference's merge methods (locality_aware_nms.py:99-126).  WEIGHTED is an opt-in extension
 * with no counterpart there: the score-weighted merge of the original EAST LANMS inside the reference's iteration (a
 * box becomes the mean of itself and the boxes it absorbs, weighted by score, and takes over their weights). */
enum { OCRHIP_MERGE_UNION = 0, OCRHIP_MERGE_AVERAGE = 1, OCRHIP_MERGE_WEIGHTED = 2 };

/* A named host tensor handed to *_load_weights.  Names and layouts are those of the original
 * torch checkpoints the reference converts from (models/weights_converter.py:252-322):
 * conv OIHW, linear (out,in), LSTM (4H,in) with gate rows i,f,g,o, batch-norm
 * weight/bias/running_mean/running_var. */
typedef struct {
    const char* name;
    const float* data; /* host, float32, C-contiguous */
    int ndim;
    int64_t shape[4];
} ocrhip_tensor_desc;

int ocrhip_version(void);
const char* ocrhip_last_error(void);

/* Handle lifetime.  Replaces model construction through the runtime switch
 * (models/interfaces/base_model.py:179-189, utils/keras/runtimes/__init__.py:23-36). */
int ocrhip_create(int device, ocrhip_handle* out);
int ocrhip_destroy(ocrhip_handle h);

/* Options of a handle (no counterpart in the reference; Keras picks its convolution algorithm itself).
 *   "conv3x3_winograd": how the 3x3 convolutions with >= 64 output channels are computed, always in float32.
 *   2 (default): Winograd F(4x4,3x3) (4x fewer multiplications; results within ~1e-5 of the output scale of the direct
 *   sum per layer) for the launches whose feature map fills that kernel's 16x32 tiles, F(2x2,3x3) for the others;
 *   1: F(2x2,3x3) everywhere (2.25x fewer multiplications, results within float32 rounding of the direct sum);
 *   0: a direct implicit GEMM (a float32 fma chain over the taps and channels).
 *   "lstm_persistent": 1 and 2 (default) run the recurrence of a BiLSTM layer as one launch of persistent workgroups
 *   (W_hh resident in LDS, h_t handed over through memory); 0 launches one kernel per time step.  1: the four waves of
 *   a workgroup share passes of 112 sequences (h staged through LDS, one counter per pass).  2: where a workgroup has
 *   at least 8 blocks of 16 sequences (about 2000 crops on 256 compute units), its eight waves own blocks with all gate
 *   columns each, read h straight from memory and hand over per block, without any barrier; smaller batches as 1.
 *   3: the independent waves whatever the batch size.  Same arithmetic in all of them, 1 / 2 / 3 bit-identical
 *   (hidden sizes other than 256 always take the per-step kernel).  A persistent launch needs the whole GPU: its
 *   workgroups wait for each other, so launches of one process are chained (one at a time per device), and if another
 *   process or stream holds compute units for seconds a wait gives up, the kernel leaves, and the error surfaces through
 *   ocrhip_check_errors / the next ocrhip_crnn_forward* call on the handle (OCRHIP_EHIP, "lstm hand-over timed out").
 *   "bf16_fuse_first": 1 (default) | 0: the bfloat16-tensor detector computes its first layer (3 -> 64) inside the second
 *   layer's kernel, per tile, from the page pixels (bfloat16 operands like every other layer of that path), so the
 *   full-resolution 64-channel tensor is never written; 0 runs the float32 first-layer kernel and stores its result.
 *   "crnn_side_streams": 1 (default) | 0: ocrhip_crnn_forward_packed runs the feature extractor of a batch with many
 *   small groups of equal-width crops (4 or more groups of fewer than 256 crops on average: grouped lines of text) on
 *   the caller's stream and three side streams of the handle, joined before the BiLSTM; same results, the launches of
 *   different groups overlap.  0: everything on the caller's stream.
 *   "lstm16_units": 64 (default) | 32: hidden units a workgroup of the bfloat16 recurrence (compute_dtype OCRHIP_BF16)
 *   keeps of W_hh in LDS.  Every workgroup reads all of h_{t-1} of its sequences, so 64 units (4 workgroups per sequence
 *   group instead of 8) halve the hand-over reads; same results.
 *   "lstm_debug_drop_handover": test hook, 0 (default) | 1: one workgroup of the persistent recurrence withholds its
 *   first hand-over and waits give up after 4096 spins, which exercises the time-out path above.
 *   "wino4_split": 0 (default) runs every F(4x4,3x3) layer fused (a workgroup transforms its input tile itself); 1 runs
 *   the layers with >= 512 input channels as two kernels (transform the input tiles once into a workspace of the
 *   handle, then multiply from it: no LDS, barrier or vector arithmetic in the multiply loop).  Results agree to the
 *   order of the inverse transform's additions.
 *   "wino4_wave_tiles": how the fused F(4x4,3x3) launches (all layers; with wino4_split = 1 those below 512 input channels)
 *   divide a tile between the waves.  1: tiles of 8 x 32 outputs, two workgroups of 4 waves x 16 channels per compute
 *   unit, every wave owns whole (block, channel) tiles (16x16x4 matrix instructions, all 36 Winograd components in one
 *   lane), so the inverse transform and the max-pool run in registers and a tile passes one barrier on its way out.
 *   2 (default): the same, and layers with a multiple of 128 output channels (and of 32 input channels) run one
 *   workgroup of 8 waves x 16 channels per tile, so that a tile is transformed once per 128 output channels; results
 *   identical to 1.  0: tiles of 16 x 32, one workgroup of 8 waves, waves own 3x3 of the components and sum their
 *   partial outputs through LDS.  Same arithmetic up to the order of the inverse transform's additions (all within
 *   the 1e-4 tolerance of the parity tests).
 *   "wino4_side_by_side": 1 (default): a fused F(4x4,3x3) launch over a batch of images whose width is not a multiple
 *   of the 32-column tile (the recogniser's crops) tiles the batch as one wide image, the images one zero column apart
 *   (two under 2x2 pooling), so the tiles are full.  An output then shares Winograd patches with its neighbour image:
 *   results equal the image-by-image launch (0) to the algorithm's rounding error (~1e-5 of the output scale), not bit
 *   for bit, and depend on which images share a launch to that extent.
 *   "first_layer_mfma": 1 (default) runs the detector's first convolution (3 -> 64 channels) on the matrix cores, 0 on
 *   the vector ALU; the same 27 exact products per output, summed in pairs of taps by the matrix instruction
 *   (v_mfma_f32_32x32x2_f32 adds two products per step) and one by one by the vector kernel: bit-identical on gfx950 as
 *   pinned by the tests (random and large mixed-sign inputs), not by construction.
 *   "flat_gemm": 1 (default) runs the Dense layers, the LSTM input projections and the 2x2 convolutions as a flat
 *   GEMM with 256 x 128 tiles; 0 runs them on the implicit-GEMM convolution kernel (256 x 64 / spatial tiles).  Same
 *   arithmetic (a float32 fma chain over k in both).
 *   "fuse_mean_rows": 1 (default) | 0: the float32 recogniser takes the mean over the 3 feature rows in the epilogue of the
 *   extractor's last (2x2) convolution instead of writing the rows and averaging them in a kernel of its own; the same
 *   sums in the same order.
 *   "fuse_ctc": 1 (default): when ocrhip_crnn_forward* is asked for tokens but not for logits, the class layer
 *   Dense(vocab) reduces every row to one (maximum, class) pair per block of 128 classes in its epilogue and the CTC
 *   kernel finishes the argmax over those pairs: the (rows, vocab) logits are never written or read.  0: logits go
 *   through a buffer of the handle.  Same tokens and scores bit for bit. */
int ocrhip_set_option(ocrhip_handle h, const char* name, int value);

/* ------------------------------------------------------------------------------------ EAST --
 * Weights of EASTVGG (architectures/east_arch.py:95-175), keys of SakuraRiven's east_vgg16.pth:
 *   extractor.features.{0,3,7,10,14,17,20,24,27,30,34,37,40}.{weight,bias}   conv 3x3
 *   extractor.features.{idx+1}.{weight,bias,running_mean,running_var}        batch-norm
 *   merge.conv{1..7}.{weight,bias}, merge.bn{1..7}.{weight,bias,running_mean,running_var}
 *   output.conv{1,2,3}.{weight,bias}
 * Batch-norm (eps 1e-5) is folded into the preceding convolution at load time. */
int ocrhip_east_load_weights(ocrhip_handle h, const ocrhip_tensor_desc* w, int n);

/* Detector input preparation for pages whose sides are not multiples of 32: replaces the dtype conversion,
 * the stretch and the normalisation of BaseImageModel.get_image (models/interfaces/base_image_model.py:93-131;
 * utils/image/image_io.py:126-128; utils/image/image_processing.py:54-79,187-199 with size=None, multiples=32,
 * antialias=True; utils/image/image_normalization.py:28-39,57):  x/255 (uint8 input only) -> antialiased bilinear
 * resize, width first then height -> (x - 0.5) / 0.5.
 *   d_img : device (B,H,W,3) uint8 raw pixels or float32 in [0,1]
 *   d_out : device (B,out_h,out_w,3) float32, ready for ocrhip_east_forward(in_dtype = OCRHIP_F32) */
int ocrhip_image_resize_norm(ocrhip_handle h, const void* d_img, int in_dtype, int B, int H, int W, int out_h,
                             int out_w, float* d_out, ocrhip_stream stream);

/* Dice coefficient of two score maps: replaces `dice_coeff(output, prev_output)` in the frame de-duplication of
 * BaseOCR.stream_fn (models/ocr/base_ocr.py:402,408-411; utils/distances.py:125-130):
 *   d_out[0] = 2 * sum(x * y) / (sum(x) + sum(y)), 0 when the denominator is 0; sums in float64.
 *   d_x, d_y : device float32, n elements each taken every `stride` floats (stride 6 walks channel 0 of two
 *              (mh,mw,6) detector maps in place)
 *   d_out    : device float32[1] */
int ocrhip_dice_coeff(ocrhip_handle h, const float* d_x, const float* d_y, int64_t n, int64_t stride, float* d_out,
                      ocrhip_stream stream);

/* EAST forward: replaces `compiled_infer(inputs)` in BaseDetector.infer
 * (models/detection/base_detector.py:72-76) and, for in_dtype == OCRHIP_U8, also the
 * `/255` + 'east' normalisation `(x - 0.5) / 0.5` (utils/image/image_normalization.py:57,
 * models/interfaces/base_image_model.py:127-131).
 *   d_img : device, (B,H,W,3) NHWC, uint8 raw pixels or float32 already normalised; H, W % 32 == 0
 *   d_maps: device, (B,H/4,W/4,6) float32 = [score, d_top, d_bottom, d_left, d_right, angle],
 *           sigmoid outputs exactly as the Keras model returns them
 *   compute_dtype: OCRHIP_F32 (float32 products and sums, the reference's precision) or OCRHIP_BF16 (operands of
 *           the MFMA convolutions rounded to bfloat16, float32 accumulation and tensors; within 1e-2 of float32) */
int ocrhip_east_forward(ocrhip_handle h, const void* d_img, int B, int H, int W, int in_dtype,
                        int compute_dtype, float* d_maps, ocrhip_stream stream);

/* Debug / test access to an activation of the last ocrhip_east_forward call (synchronises).
 *   name: "pool1".."pool5", "merge2", "merge4", "merge7"
 *   shape receives (B,H,W,C); *n_elems the element count; dst (host, float32) may be NULL to query. */
int ocrhip_east_get_activation(ocrhip_handle h, const char* name, float* dst, int64_t dst_capacity,
                               int64_t shape[4]);

/* Score threshold + RBOX decode + validity filter + /(W,H) + poly -> xyxy.  Replaces
 * restore_polys_from_map / restore_polys / filter_polys (models/detection/east.py:102-187), the map
 * scaling in EAST.decode_output (east.py:81-85) and the poly branch of convert_box_format
 * (utils/image/bounding_box/converter.py:151-154,165-166).
 *   d_maps  : device (B,mh,mw,6) float32 from ocrhip_east_forward
 *   in_h/w  : network input size (the reference's `inputs.shape[1:-1]`), scale = in / map
 *   cap     : slots per image in the outputs; image b owns [b*cap, b*cap + counts[b])
 *   d_xyxy  : device (B*cap,4) float32 relative xyxy            (required)
 *   d_polys : device (B*cap,8) float32 relative corners, or NULL
 *   d_scores: device (B*cap)   float32                           (required)
 *   d_pix   : device (B*cap)   int32 map pixel index y*mw+x, or NULL
 *   d_counts: device (B)       int32 number of boxes found; if counts[b] > cap only the first cap
 *             (row-major order) were stored - the caller checks this after synchronising.
 * Output order inside an image is row-major over the map. */
int ocrhip_east_decode(ocrhip_handle h, const float* d_maps, int B, int mh, int mw, int in_h, int in_w,
                       float obj_threshold, int cap, float* d_xyxy, float* d_polys, float* d_scores,
                       int* d_pix, int* d_counts, ocrhip_stream stream);

/* ocrhip_east_decode for callers that cannot allocate device memory (code under HIP graph capture, tight serving
 * loops, harnesses that forbid hipMalloc).  No counterpart in the reference.  Same arguments, same results bit for
 * bit; the difference is where the kernels' staging area lives (the per-wave survivor masks and per-chunk counts
 * the two decode kernels hand to each other): ocrhip_east_decode keeps it in a grow-only buffer of the handle
 * (hipMalloc when a call needs more than any call before it), this entry point keeps it in a fixed 1.125 MB array in
 * device memory that every GPU the library runs on holds until the process ends.  The call makes no hipMalloc /
 * hipFree / event creation and records no stage timer.
 *   Limit: B * ceil(mh*mw / 256) <= 32768 (8.4 M map pixels per call, e.g. 128 pages of 1024x1024); beyond it
 *   the call returns OCRHIP_ECAP and launches nothing.
 *   All handles and streams of one GPU share that GPU's array: calls that overlap on one GPU (two streams, two
 *   host threads) race on it; serialise them or use ocrhip_east_decode.  Nothing in the array has to start at zero
 *   (the first kernel writes every word the second reads), so a call is not disturbed by what the last one left. */
int ocrhip_east_decode_static(ocrhip_handle h, const float* d_maps, int B, int mh, int mw, int in_h, int in_w,
                              float obj_threshold, int cap, float* d_xyxy, float* d_polys, float* d_scores,
                              int* d_pix, int* d_counts, ocrhip_stream stream);

/* Locality-aware NMS followed (when nms_threshold < merge_threshold) by matrix NMS.  Replaces
 * lanms / self_merging / _merge_boxes (utils/image/bounding_box/locality_aware_nms.py:22-126),
 * fast_nms / self_suppression (non_max_suppression.py:117-138,191-220) and _sort_boxes (:247-253).
 * Rank = score descending, ties towards the lower input index.
 *   d_xyxy, d_scores, d_counts, cap: as written by ocrhip_east_decode (any xyxy boxes work)
 *   max_iter  : <= 0 means "number of boxes" (the reference default)
 *   max_out   : <= 0 means unlimited (the reference's max_output_size=None)
 *   d_merged  : device (B*cap,4) merged boxes in INPUT order (suppressed boxes are 0), or NULL
 *   d_keep    : device (B*cap) uint8 valid mask in INPUT order, or NULL
 *   d_out_boxes / d_out_scores / d_out_src: device (B*cap,4)/(B*cap)/(B*cap): the kept boxes of image b
 *               in RANK order at [b*cap, b*cap + out_counts[b]); src = input index inside the image
 *   d_out_counts: device (B) int32;  d_n_iter: device (B) int32 merge iterations run, or NULL */
int ocrhip_lanms(ocrhip_handle h, const float* d_xyxy, const float* d_scores, const int* d_counts, int B,
                 int cap, float merge_threshold, float nms_threshold, int merge_method, int max_iter,
                 int max_out, float* d_merged, uint8_t* d_keep, float* d_out_boxes, float* d_out_scores,
                 int* d_out_src, int* d_out_counts, int* d_n_iter, ocrhip_stream stream);

/* ------------------------------------------------------------------------------ crop / pack --
 * Crop + /255 + grayscale + antialiased bilinear resize + repeat-last padding + 'easyocr'
 * normalisation, packed for the recogniser.  Replaces crop_box
 * (utils/image/bounding_box/processing.py:111-132), the tail of load_image
 * (utils/image/image_io.py:122-135), resize_image / pad_image
 * (utils/image/image_processing.py:17-142) and the normalisation
 * (utils/image/image_normalization.py:58) for the CRNN arguments of models/ocr/crnn.py:39-45.
 *   d_img    : device (B,H,W,3) uint8
 *   d_rects  : device (n,4) int32 absolute x0,y0,x1,y1 already resolved to the slice the reference
 *              takes (0 <= x0 < x1 <= W ...); sizes are computed on the host with the reference's
 *              float32 arithmetic
 *   d_box_img: device (n) int32 image index of each crop
 *   d_inter  : device (n,2) int32 (h,w) the crop is resized to before padding
 *   d_out_w  : device (n) int32 padded width of each crop (multiple of 64)
 *   d_dst_off: device (n) int64 element offset of each crop's (out_h,out_w) plane in d_packed
 *   d_packed : device, float32 (or bf16) planes, row-major (out_h, out_w), value (x-0.5)/0.5 */
int ocrhip_crop_resize_pack(ocrhip_handle h, const uint8_t* d_img, int B, int H, int W, const int* d_rects,
                            const int* d_box_img, const int* d_inter, const int* d_out_w,
                            const int64_t* d_dst_off, int n, int out_h, void* d_packed, int out_dtype,
                            ocrhip_stream stream);

/* Opt-in extensions that keep the rotation of the detected quadrilaterals.  They have no counterpart in the
 * reference, which collapses quadrilaterals to axis-aligned boxes before NMS
 * (utils/image/bounding_box/non_max_suppression.py:27, converter.py:151-154) and crops axis-aligned slices
 * (utils/image/bounding_box/processing.py:111-132; rotation raises NotImplementedError, converter.py:204).
 *
 * ocrhip_rotated_nms: greedy NMS on the quadrilaterals of ocrhip_east_decode (d_polys) with rotated IoU: repeatedly
 * takes the remaining candidate with the highest score (lowest index among equals) and drops every remaining one
 * whose IoU with it is >= iou_threshold (> 0).  IoU by clipping one convex quadrilateral with the other, float64.
 *   d_polys (B,cap,8) float32 x,y of 4 corners; d_scores (B,cap); d_counts (B) candidates per image
 *   d_out_polys (B,max_out,8), d_out_scores (B,max_out), d_out_index (B,max_out) int32 candidate index, in the order
 *   taken (descending score); d_out_counts (B) int32 number kept (at most max_out)
 *
 * ocrhip_crop_rotated_pack: ocrhip_crop_resize_pack for oriented rectangles.  d_geom (n,8) float32 per crop:
 * origin x, y (the text's top-left corner, pixels), step along the text (x, y), step across the text (x, y), crop
 * width cw and height ch (whole numbers): cell (x, y) of the cw x ch crop is the bilinear interpolation of the
 * page's gray values at origin + (x + 0.5) * step_along + (y + 0.5) * step_across (neighbours clamped to the page);
 * the crop then takes the same resize / pad / normalise path as an axis-aligned one.  An unrotated rectangle on
 * integer corners gives bit for bit the output of ocrhip_crop_resize_pack. */
int ocrhip_rotated_nms(ocrhip_handle h, const float* d_polys, const float* d_scores, const int* d_counts, int B, int cap,
                       float iou_threshold, int max_out, float* d_out_polys, float* d_out_scores, int* d_out_index,
                       int* d_out_counts, ocrhip_stream stream);
int ocrhip_crop_rotated_pack(ocrhip_handle h, const uint8_t* d_img, int B, int H, int W, const float* d_geom,
                             const int* d_box_img, const int* d_inter, const int* d_out_w, const int64_t* d_dst_off,
                             int n, int out_h, void* d_packed, int out_dtype, ocrhip_stream stream);

/* ------------------------------------------------------------------------------------ CRNN --
 * Weights of CRNNCTC with the VGG extractor (architectures/crnn_arch.py:38-41,306-328,427-531),
 * keys of EasyOCR's recogniser checkpoints without the "module." prefix:
 *   FeatureExtraction.ConvNet.{0,3,6,8,18}.{weight,bias}, FeatureExtraction.ConvNet.{11,14}.weight,
 *   FeatureExtraction.ConvNet.{12,15}.{weight,bias,running_mean,running_var},
 *   SequenceModeling.{0,1}.rnn.{weight_ih_l0,weight_hh_l0,bias_ih_l0,bias_hh_l0}[_reverse],
 *   SequenceModeling.{0,1}.linear.{weight,bias}, Prediction.{weight,bias}
 * The vocabulary size is the row count of Prediction.weight; the LSTM width H (`lstm_size`, crnn_arch.py:485-486;
 * 256 in the gen-2 EasyOCR models, 512 in gen-1) is read off SequenceModeling.0.rnn.weight_hh_l0 (4H, H), a multiple of 64.
 * ResNet extractor (crnn_arch.py:330-425, `backbone='resnet'`): chosen when FeatureExtraction.ConvNet.conv0_1.weight
 * is present; tensors conv0_{1,2}, bn0_{1,2}, layer{1..4}.{b}.{conv1,bn1,conv2,bn2}, layer{1,2,3}.0.downsample.{0,1},
 * conv{1,2,3}, bn{1,2,3}, conv4_{1,2}, bn4_{1,2} under FeatureExtraction.ConvNet (EasyOCR's ResNet_FeatureExtractor
 * keys).  It yields 512 features per step, so SequenceModeling.0.rnn.weight_ih_l0* is (1024, 512).  As in the
 * reference graph, only the first block of layer1-3 has a skip connection and the block batch-norms use eps 1e-3. */
int ocrhip_crnn_load_weights(ocrhip_handle h, const ocrhip_tensor_desc* w, int n);

/* CRNN forward + optional fused CTC greedy.  Replaces `compiled_infer(inp[None])` per text row in
 * BaseOCR.infer (models/ocr/base_ocr.py:171-201) for a whole batch of equal-width crops, and (when
 * d_tokens != NULL) pad_batch + ctc_decode (base_ocr.py:209-222, utils/text/ctc_decoder.py:14-35).
 *   d_crops : device (n,64,Wpad,1) float32 normalised crops (ocrhip_crop_resize_pack output)
 *   d_logits: device (n,T,V) float32 raw logits, T = ocrhip_crnn_seq_len(h, Wpad), or NULL
 *   d_tokens: device (n,T) int32 greedy tokens padded with -1, or NULL
 *   d_scores: device (n) float32 -sum_t max_v logit / T, or NULL (required with d_tokens) */
int ocrhip_crnn_forward(ocrhip_handle h, const void* d_crops, int n, int Wpad, int compute_dtype,
                        float* d_logits, int* d_tokens, float* d_scores, ocrhip_stream stream);
int ocrhip_crnn_vocab_size(ocrhip_handle h);

/* Time steps the loaded recogniser produces for a crop of padded width wpad: wpad/4 - 1 with the VGG
 * extractor (3 valid columns lost by its last 2x2 convolution, crnn_arch.py:38-41), wpad/4 + 1 with the
 * ResNet extractor (crnn_arch.py:389-423).  0 when no weights are loaded. */
int ocrhip_crnn_seq_len(ocrhip_handle h, int wpad);

/* CRNN forward + CTC greedy for a whole batch of crops of DIFFERENT widths in one call: what
 * BaseOCR.infer does for all the rows of an image (one forward per row, pad_batch of the logits,
 * ctc_decode with lengths; models/ocr/base_ocr.py:171-222, utils/sequence_utils.py:16-62).  The
 * feature extractor runs once per group of equal width; the recurrent layers, the Dense layers and
 * the CTC decoding run once over all sequences (ragged: sequence i has T_i = ocrhip_crnn_seq_len(h, out_w[i]) steps).
 *   d_packed : device, crop i is a (64, out_w[i]) float32 plane; planes back to back in crop order
 *              (the layout ocrhip_crop_resize_pack writes for cumulative dst offsets)
 *   out_w    : HOST (n) int32 padded widths, non-decreasing
 *   t_stride : row stride of d_tokens, >= max_i T_i
 *   d_logits : device (sum_i T_i, V) float32, sequences back to back, or NULL
 *   d_tokens : device (n, t_stride) int32 padded with -1, or NULL;  d_scores: device (n) float32
 *   d_lengths: device (n) int32 receives T_i, or NULL */
int ocrhip_crnn_forward_packed(ocrhip_handle h, const void* d_packed, const int* out_w, int n, int compute_dtype,
                               int t_stride, float* d_logits, int* d_tokens, float* d_scores, int* d_lengths,
                               ocrhip_stream stream);

/* Asynchronous failures of earlier calls on the handle.  The persistent BiLSTM kernel cannot return a code when a
 * hand-over between its workgroups times out; it sets a device word that is cleared, in stream order, at the start of
 * every ocrhip_crnn_forward* call (a time-out of one call never makes a later call's kernels leave early) and copied
 * to the host behind that call's results, into a status slot of the call's own (a ring of 16; the 17th call in flight
 * waits for the first).  This call reads the slots that have arrived, in call order: OCRHIP_OK when none carries an
 * error.  On an error it waits for `stream`, returns OCRHIP_EHIP with "lstm hand-over timed out" once, and EVERY
 * recogniser call enqueued on the handle before this check counts as failed (one return code cannot say which of
 * several pipelined calls it was): run them again.  synchronize != 0 waits for `stream` first (so every call enqueued
 * so far is covered); with 0 the caller has already waited (event / stream synchronisation) for the outputs it is
 * about to use.  ocrhip_crnn_forward* perform the same check on entry.  The reference has no counterpart: its Keras
 * LSTM raises synchronously. */
int ocrhip_check_errors(ocrhip_handle h, ocrhip_stream stream, int synchronize);

/* Debug / test access to an activation of the last ocrhip_crnn_forward call (synchronises).
 *   name: "features" (n,T,256), "seq0", "seq1" (n,T,256) */
int ocrhip_crnn_get_activation(ocrhip_handle h, const char* name, float* dst, int64_t dst_capacity,
                               int64_t shape[4]);

/* CTC greedy decoding of padded logits.  Replaces ctc_decode(method='greedy')
 * (utils/text/ctc_decoder.py:14-35 -> keras.ops.ctc_decode): argmax over the raw logits (first
 * maximum wins), merge repeats, drop `blank`, left-pack, pad with -1;
 * score = -sum_{t<len} max_v logit[t] / len.
 *   d_logits : device (B,T,V) float32;  d_lengths: device (B) int32 or NULL (= T)
 *   d_tokens : device (B,T) int32;      d_scores : device (B) float32 */
int ocrhip_ctc_greedy(ocrhip_handle h, const float* d_logits, int B, int T, int V, const int* d_lengths,
                      int blank, int* d_tokens, float* d_scores, ocrhip_stream stream);

/* CTC prefix beam search.  Replaces ctc_decode(method='beam_search') (utils/text/ctc_decoder.py:28-35 ->
 * keras.ops.ctc_decode(strategy='beam_search', beam_width=num_beams, top_paths=1)), the default decoding of
 * BaseOCR.infer (models/ocr/base_ocr.py:112-115,209-222): per-step log-softmax of the logits, blank / non-blank path
 * probabilities in log space (float64), every label may extend every kept prefix, the num_beams most probable
 * prefixes survive each step; the best prefix is returned with score = log p(prefix) / length.  One wave per sequence.
 *   d_logits : device float32; d_row0 == NULL: (B,T,V) padded; else sequence b starts at row d_row0[b] of a flat
 *              (sum of lengths, V) matrix (the layout ocrhip_crnn_forward_packed writes) and T is the row stride of
 *              d_tokens (>= the longest sequence)
 *   d_lengths: device (B) int32 or NULL (= T; required with d_row0);  num_beams: 1..16
 *   d_tokens : device (B,T) int32 padded with -1;  d_scores: device (B) float32 */
int ocrhip_ctc_beam_search(ocrhip_handle h, const float* d_logits, int B, int T, int V, const int* d_lengths,
                           const int* d_row0, int blank, int num_beams, int* d_tokens, float* d_scores,
                           ocrhip_stream stream);

/* ------------------------------------------------------------------------------ stage timers --
 * Per-kernel device timing with HIP events recorded on the caller's stream around every launch the
 * library makes.  Replaces the hierarchical @timer / Timer instrumentation that wraps every stage of
 * the reference (loggers/time_logging.py:27-136, e.g. models/detection/east.py:64,102,153).
 *   ocrhip_profile_enable : enable != 0 starts recording (and drops earlier records), 0 stops
 *   ocrhip_profile_only   : restricts the recording to launches of one kernel name (as ocrhip_profile_collect reports
 *                           it); NULL or "" times every launch again.  Two events per launch cost a few microseconds
 *                           of stream time each: a throughput measurement times only the kernel it wants to price.
 *   ocrhip_profile_collect: synchronises the device, aggregates the records per kernel name, writes
 *                           at most `cap` entries sorted by total time and sets *n to the number of
 *                           distinct kernels; flops / bytes are the ALGORITHMIC work of the launches
 *                           (a direct convolution's 2*k*k*Cin*Cout flop per output pixel; bytes = inputs
 *                           read once + outputs written once + weights), not counter readings. */
typedef struct {
    char name[64];
    int launches;
    double total_ms;
    double flops;
    double bytes;
} ocrhip_kernel_stat;

int ocrhip_profile_enable(ocrhip_handle h, int enable);
int ocrhip_profile_only(ocrhip_handle h, const char* name);
int ocrhip_profile_collect(ocrhip_handle h, ocrhip_kernel_stat* out, int cap, int* n);

#ifdef __cplusplus
}
#endif
#endif /* OCRHIP_H */
