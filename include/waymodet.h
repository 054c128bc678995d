/* libwaymotrack - detector custom ops (C ABI) of the Cascade R-CNN X152-FPN hot path.
 *
 * The reference reaches these operators through detectron2 0.1.3 / torchvision 0.6 (not vendored):
 *   detnet/nn/detectron2_det/__init__.py:56,116 builds and calls the detectron2 GeneralizedRCNN whose layers
 *   (printed in logs/12442/job.log:336-1221) contain ROIAlign (job.log:1138-1143), DeformConv (job.log:412-415),
 *   the cascade box heads (job.log:1146-1218) and torchvision.ops.nms (box_utils.py:3 imports the same op).
 * Semantics are restated in SURVEY.md App. C ("parity unpinned": detectron2 cannot be imported in the build
 * container; the ops are checked against a float64/float32 PyTorch restatement in tests/test_gpu_detops.py).
 *
 * All pointers are DEVICE pointers; `stream` is a hipStream_t passed as void*; calls are stream-ordered and never
 * synchronise.  Feature maps are NHWC float32 ("channels_last" storage of an (N,C,H,W) torch tensor).
 * Status codes and wt_last_error() as in waymotrack.h.
 */
#ifndef WAYMODET_H
#define WAYMODET_H
#include <stddef.h>
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

/* ROIPooler = level assignment + ROIAlign(output 7x7 (pooled), spatial_scale 1/stride, sampling_ratio 0, aligned=True)
 * over up to 4 FPN levels in one launch (detectron2 ROIPooler; job.log:1137-1143).
 *   feats      : HOST array of n_levels DEVICE pointers; feats[l] = (N, H_l, W_l, C) NHWC float32;
 *                heights / widths / scales: HOST arrays, scales[l] = 1/stride_l
 *   rois       : (R, 5) float32 [batch_index, x1, y1, x2, y2] in image coordinates
 *   level assignment: floor(canonical_level + log2(sqrt(area)/canonical_size + 1e-8)) clamped to
 *                     [min_level, min_level + n_levels - 1]   (detectron2 assign_boxes_to_levels)
 *   out        : (R, pooled, pooled, C) NHWC float32 */
int wd_roi_pool_fpn_f32(const float* const* feats, const int32_t* heights, const int32_t* widths,
                        const float* scales, int n_levels, int channels, int batch,
                        const float* rois, int n_rois, int pooled, int min_level, int canonical_level,
                        float canonical_size, float* out, void* stream);

/* torchvision.ops.nms / detectron2 batched_nms restated: boxes (n,4) xyxy float32 ALREADY sorted by descending
 * score, idxs (n) int32 group id (NULL = one group); box j is suppressed by a kept box i < j of the same group
 * when IoU(i,j) > iou_threshold.  keep_mask (n) uint8 receives 1 for kept boxes; n_keep (device int32) their count.
 * workspace: wd_nms_workspace(n) bytes. */
size_t wd_nms_workspace(int n);
/* Static-shape survivor list: the first `cap` entries i (in mask order) with keep_mask[i] != 0 (and valid[i] != 0 when given) as
 * out_idx[k] = order ? order[i] : i; unused slots are -1, *count = min(#survivors, cap) (device int32).  Replaces the
 * `keep[:post_nms_topk]` / `[:topk]` slices behind detectron2's batched_nms without a host round trip. */
int wd_select_kept(const uint8_t* keep_mask, const uint8_t* valid, const int64_t* order, int n, int cap, int64_t* out_idx,
                   int32_t* count, void* stream);
int wd_nms_sorted_f32(const float* boxes, const int32_t* idxs, int n, float iou_threshold,
                      uint8_t* keep_mask, int32_t* n_keep, void* workspace, size_t workspace_bytes, void* stream);

/* wd_gemm_nt_f32 for large shapes (box-head fc1, job.log:1146-1218): 128 x 128 tiles, fragments by ds_read_b128 from an
 * XOR-swizzled LDS image, deterministic two-pass split-K through `workspace` (wd_gemm_nt_workspace bytes; 0 = the shape runs on
 * the v1 kernel and needs none). */
size_t wd_gemm_nt_workspace(int M, int N, int K);
int wd_gemm_nt_ws_f32(const float* A, const float* Bt, const float* bias, const float* residual, int relu, int M, int N, int K,
                      float* C, void* workspace, size_t workspace_bytes, void* stream);

/* 1x1 convolution with residual (detectron2 BottleneckBlock conv3 + shortcut + ReLU, job.log:412-415) as ONE hipBLASLt GEMM:
 * out (m, n) = relu?(a (m, k) . w (n, k)^T + residual (m, n) + bias (n)), row-major; residual may alias out.  beta = 1 carries
 * the residual, the RELU_BIAS epilogue the folded FrozenBN shift.  The first call per shape (outside a stream capture) times the
 * heuristic's candidates and caches the fastest; hipBLASLt itself is resolved with dlopen at first use. */
int wd_gemm_lt_f32(const float* a, const float* w, const float* bias, const float* residual, float* out, int m, int n, int k,
                   int relu, void* workspace, size_t workspace_bytes, void* stream);
int wd_gemm_lt_plan_info(int m, int n, int k, int relu, int has_bias, int has_residual, float* best_us, int* candidates);
/* fp32-equivalent GEMM on the bf16 matrix cores (csrc/det_gemm_split.hip): every f32 operand is carried EXACTLY as three bfloat16
 * planes (hi + mid + lo by successive round-to-nearest subtraction), the six cross terms with i + j <= 2 run on
 * v_mfma_f32_32x32x16_bf16 with f32 accumulation.  Accuracy: the dropped cross terms are bounded by ~2^-25 |a.b| per product (the order of
 * one f32 product rounding); the MEASURED error against float64 is not above the exact-f32 kernels' on the tested distributions
 * (tests/test_gpu_gemm_split.py, incl. operands chosen to maximise the small planes).  Semantics that differ from an f32 GEMM: an operand
 * that is +-inf or NaN makes its output elements NaN (f32: +-inf where no NaN is involved), and operands below 2^-110 in magnitude lose
 * their lowest plane (bfloat16 underflow).  Finite post-ReLU activations and weights - the detector's operands - are unaffected.
 * Replaces the 1x1 convolutions of detectron2's BottleneckBlock (conv1 / conv3 / shortcut, job.log:534-546), the FPN lateral convs
 * (job.log:1093-1108) and - wd_conv_split_f32 - the dense 3x3 convolutions of FPN output / RPN / box heads (job.log:1109-1160).
 *   wd_gemm_split_pack_weight : w (N, K) f32 row-major -> `packed` (wd_gemm_split_packed_bytes(N, K) bytes, MFMA fragment order); once per weight.
 *                               For a 3x3 convolution pass w as (N, 9 * C) with k = (kh * 3 + kw) * C + c.
 *   wd_gemm_split_f32         : out (M, N) = relu?(a (M, K; row stride lda) . w^T + bias (N) + residual (M, N; row stride ldc)); residual may
 *                               alias out.  K % 64 == 0, N % 32 == 0 (full speed: N % 256 == 0), lda / ldc % 4 == 0, every pointer 16-byte
 *                               aligned, M * lda < 2^31: wd_gemm_split_supported() answers that question for host code that wants to fall back.
 *   wd_conv_split_f32         : x (batch, H, W, C) NHWC, ksize 1 or 3, any stride / pad (zero padding), C % 64 == 0;
 *                               out (batch, Ho, Wo, N) NHWC with the same fused epilogue. */
size_t wd_gemm_split_packed_bytes(int N, int K);
int wd_gemm_split_pack_weight(const float* w, int N, int K, void* packed, void* stream);
/* the same for any 2-D view: element (n, k) = w[n * stride_n + k * stride_k] (the transpose of a weight for the backward-data GEMM of training) */
int wd_gemm_split_pack_weight_strided(const float* w, int N, int K, long stride_n, long stride_k, void* packed, void* stream);
/* Many weights in ONE launch (a training step re-packs every trainable weight after the optimizer step, in the forward and the backward-data
 * orientation).  Descriptors live in DEVICE memory.  A source is addressed through strides, so a convolution weight (N, C, ks, ks) is read where it
 * lies: element (n, k) with k = (kh * ks + kw) * C + c is src[n * s_n + c * s_c + kh' * s_kh + kw' * s_kw], (kh', kw') = (ks-1-kh, ks-1-kw) when flip
 * (the weight of the backward-data convolution: flipped taps, channel roles swapped through s_n / s_c).  A 2-D (N, K) weight: C = K, ksize = 1.
 * K % 64 == 0, C % 8 == 0.  first_block = running sum of blocks(i) = ceil(ceil32(N_i) * (K_i / 8) / 256) over the descriptors before i (ascending);
 * total_blocks = the sum over all.  dst: wd_gemm_split_packed_bytes(N, K) bytes, 16-byte aligned. */
typedef struct WdSplitPackDesc {
    const float* src;
    void* dst;
    long s_n, s_c, s_kh, s_kw;
    long first_block;
    int N, K, C, ksize, flip, reserved;
} WdSplitPackDesc;
int wd_gemm_split_pack_batch(const WdSplitPackDesc* descs_device, int count, long total_blocks, void* stream);
/* Shapes with few output tiles (FPN p5 / p6 convolutions, the box-head FC) are cut into K slices whose partial tiles meet in `workspace`
 * (wd_gemm_split_workspace(M, N, K) bytes; M = batch * Ho * Wo, K = ksize^2 * C for the convolution) and are summed in slice order by a second
 * launch: deterministic.  workspace may be NULL / smaller: the call then runs unsliced. */
size_t wd_gemm_split_workspace(long M, int N, int K);
int wd_gemm_split_f32(const float* a, long lda, const void* packed_w, const float* bias, const float* residual, float* out, long ldc,
                      int M, int N, int K, int relu, void* workspace, size_t workspace_bytes, void* stream);
int wd_conv_split_f32(const float* x, int batch, int H, int W, int C, const void* packed_w, int ksize, int stride, int pad, const float* bias,
                      const float* residual, float* out, int N, int relu, void* workspace, size_t workspace_bytes, void* stream);
int wd_gemm_split_supported(const float* a, long lda, const float* bias, const float* residual, const float* out, long ldc, long M, int N, int K);
/* Round 6 - activation planes: the A operand of a split-operand GEMM stored PRE-SPLIT by the kernel that produced it, so that the consumer pulls it
 * into LDS with LDS-DMA (global_load_lds_dwordx4) instead of loading f32, splitting on the vector units and writing LDS once per N tile.
 * Layout of an (M, K) activation matrix, K % 32 == 0: [ceil(M / 32) row blocks][K / 32][3 planes: hi, mid, lo][2048 bytes], the 2048 bytes being the
 * LDS image of 32 rows x 32 bfloat16 the kernel's fragment reads expect (64-byte rows, the four 16-byte slots of row r XOR-swizzled with (r >> 2) & 3);
 * 6 bytes per element, wd_split_planes_bytes(M, K) in total, 16-byte aligned.  hi + mid + lo reproduces the f32 value exactly (same semantics as above).
 *   wd_split_planes_pack_f32 / _unpack_f32 : f32 (M, K; row stride) <-> planes (stand-alone producer / consumer; exact both ways)
 *   wd_gemm_split_io : the GEMM with every operand in either form - exactly one of io->a (f32, row stride lda) / io->a_planes; at most one of
 *                      io->residual (f32, row stride ldc) / io->residual_planes ((M, N) planes); at least one of io->out (f32, ldc) / io->out_planes
 *                      ((M, N) planes: the A operand of the next GEMM).  ldc == 0 means N.  A planes residual may alias out_planes. */
typedef struct WdSplitIO {
    const float* a;
    long lda;
    const void* a_planes;
    const float* residual;
    const void* residual_planes;
    float* out;
    void* out_planes;
    long ldc;
} WdSplitIO;
size_t wd_split_planes_bytes(long M, int K);
int wd_split_planes_pack_f32(const float* a, long lda, long M, int K, void* planes, void* stream);
int wd_split_planes_unpack_f32(const void* planes, long M, int K, float* out, long ldo, void* stream);
int wd_gemm_split_io(const WdSplitIO* io, const void* packed_w, const float* bias, int M, int N, int K, int relu, void* workspace,
                     size_t workspace_bytes, void* stream);
/* wd_nms_sorted_f32 on n_seg (<= 8) independent row ranges in one pair of launches: detectron2's per-level batched_nms of the RPN
 * (find_top_rpn_proposals) with the levels' suppression chains in parallel workgroups.  Every range is sorted by descending
 * score; idxs may still mark rows that must not suppress (group -1).  n_keep: n_seg device ints. */
int wd_nms_segmented_f32(const float* boxes, const int32_t* idxs, const int32_t* seg_offsets, int n_seg, float iou_threshold,
                         uint8_t* keep_mask, int32_t* n_keep, void* workspace, size_t workspace_bytes, void* stream);

/* --- fused, static-shape tail of the detector (csrc/det_tail.hip): one launch per step, no host round trip ------------
 * RPN.predict_proposals (detectron2 proposal_utils.find_top_rpn_proposals, called from detectron2_det/__init__.py:74 via the
 * model forward): per FPN level the k best objectness logits (torch.topk(sorted=True); ties: lower anchor index first) and
 * apply_deltas + clip of exactly those anchors.  Level l's rows start at sum_{m<l} min(k, n_m).  group = level, or -1 for an
 * empty box (dropped by find_top_rpn_proposals); valid = 0 for an empty box.  Device pointers in host arrays. */
size_t wd_rpn_topk_workspace(const int* n_per_level, int n_levels, int k);
int wd_rpn_topk_decode_f32(const float* const* logits, const float* const* deltas, const float* const* anchors,
                           const int* n_per_level, int n_levels, int k, float img_h, float img_w, float scale_clamp,
                           float* out_boxes, float* out_scores, int32_t* out_group, uint8_t* out_valid, void* workspace,
                           size_t workspace_bytes, void* stream);
/* Stable descending score sort of n <= 8192 candidate rows, gathered into sorted order (the `scores.sort(descending=True)` in
 * front of batched_nms); out_order[t] = source row of sorted row t. */
int wd_sort_candidates_f32(const float* boxes, const float* scores, const int32_t* group, const uint8_t* valid,
                           const uint8_t* valid2 /* nullable: ANDed into valid (e.g. a keep mask) */, int n, float* out_boxes, float* out_scores, int32_t* out_group, uint8_t* out_valid, int64_t* out_order,
                           void* stream);
/* fast_rcnn_inference_single_image candidates: every (row, class) pair of the last cascade stage, score = (s0 + s1 + s2) / 3 of
 * the three stages' softmax (rows x (classes + 1)); a pair is real when its row < *n_valid, box and scores are finite and the
 * class score > score_thresh; real pairs sort first (stable), boxes are clipped to the image.  group = class or -1. */
int wd_box_candidates_f32(const float* boxes, const float* s0, const float* s1, const float* s2, const int32_t* n_valid, int rows,
                          int num_classes, float score_thresh, float img_h, float img_w, float* out_boxes, float* out_scores,
                          int32_t* out_group, uint8_t* out_valid, int64_t* out_order, void* stream);
/* First `cap` kept & valid sorted candidates -> fixed-size outputs (unused rows zero) + device count: the proposal list
 * (out_scores / out_class NULL) or the final detections (class = order % num_classes). */
int wd_gather_kept_f32(const uint8_t* keep, const uint8_t* valid, const float* boxes, const float* scores, const int64_t* order,
                       int n, int cap, int num_classes, float* out_boxes, float* out_scores, int64_t* out_class, int32_t* count,
                       void* stream);
/* Detectron2Det.predict (detectron2_det/__init__.py:119-131) + COCODetection.load_prediction (detnet/data/coco.py:229-252) on the
 * device: pixel boxes of the (possibly h-flipped) in_w x in_h network input -> integer [x, y, w, h] of the out_w x out_h image,
 * 5-decimal score (xywhs: 5 rows of n doubles) and category id = class + 1 (0 for slots >= *count). */
int wd_detections_to_wire(const float* boxes, const float* scores, const int64_t* classes, const int32_t* count, int n, int in_w,
                          int in_h, int hflip, int out_w, int out_h, double* xywhs, int32_t* category, void* stream);

/* detectron2 DeformConv / ModulatedDeformConv forward (job.log:412-415), kernel 3x3, dilation 1,
 * deformable_groups 1:
 *   x      : (N, H, W, C_in) NHWC float32
 *   offset : (N, H_out, W_out, 18) NHWC float32, channel 2k = dy, 2k+1 = dx of tap k = kh*3+kw;
 *            NULL = no deformation: an ordinary grouped 3x3 convolution (res2 blocks, job.log:367-370)
 *   mask   : (N, H_out, W_out, 9) or NULL (non-modulated)
 *   weight : packed by wd_deform_pack_weight from the (C_out, C_in/groups, 3, 3) OIHW tensor
 *   scale/bias (C_out) or NULL: fused per-channel affine (FrozenBatchNorm2d) ; relu != 0 fuses ReLU
 *   y      : (N, H_out, W_out, C_out) NHWC float32,  H_out = (H + 2*pad - 3)/stride + 1 */
size_t wd_deform_packed_weight_floats(int c_in, int c_out, int groups);
int wd_deform_pack_weight(const float* weight_oihw, int c_in, int c_out, int groups, float* packed, void* stream);
/* Name of the kernel wd_deform_conv3x3_f32 dispatches for a shape (host-only; profiling / bench labels). */
const char* wd_deform_conv3x3_variant(int c_in, int groups, int stride, int pad, int has_offset);
int wd_deform_conv3x3_f32(const float* x, const float* offset, const float* mask, const float* packed_weight,
                          const float* scale, const float* bias, int relu,
                          int batch, int h, int w, int c_in, int c_out, int groups, int stride, int pad,
                          float* y, void* stream);
/* Same, with a per-layer hint: far_offsets != 0 marks a layer whose learned offsets move many samples further than the 2-pixel halo
 * of the persistent kernel's input patch (calibrated once per layer by the caller, detnet/nn/cascade_rcnn.py); such layers take the
 * per-tile fallback kernel (offset spread 2 px: 144 vs 171 us on res4). */
int wd_deform_conv3x3_hint_f32(const float* x, const float* offset, const float* mask, const float* packed_weight,
                               const float* scale, const float* bias, int relu, int batch, int h, int w, int c_in,
                               int c_out, int groups, int stride, int pad, int far_offsets, float* y, void* stream);
/* The persistent kernel (stride 1, 16 or 32 channels per group) reads a per-layer sampling table: one 16-byte entry per (pixel, tap) - the 4
 * corner slots inside the kernel's 14x14 input patch + the bilinear fractions, or, for a sample whose corners leave the patch, its image
 * coordinates (the kernel then fetches that sample from global memory itself) - followed by one flag per 8x8 tile.  The table depends on the
 * offsets only, not on the channel group: wd_deform_offsets_table_f32 builds it inside the offset conv's epilogue launch (it replaces
 * wd_tap_shift_add_f32 there: same offsets, same arithmetic), wd_deform_conv3x3_tab_f32 consumes it (table may be NULL = the library builds
 * it from `offset` into an internal scratch buffer with one extra launch; ignored by the other kernel variants).  Same batch / h / w on both
 * calls.  far_offsets != 0 (wd_deform_conv3x3_hint_f32) still selects the round-1 per-tile kernel; since round 3 the persistent kernel is
 * faster at every offset spread measured, so callers normally pass 0. */
size_t wd_deform_table_bytes(int batch, int h, int w);
int wd_deform_offsets_table_f32(const float* partial, int ld, const float* bias, int batch, int h, int w, float* offsets,
                                void* table, void* stream);
int wd_deform_conv3x3_tab_f32(const float* x, const float* offset, const float* mask, const float* packed_weight,
                              const float* scale, const float* bias, int relu, int batch, int h, int w, int c_in,
                              int c_out, int groups, int stride, int pad, int far_offsets, const void* table, float* y,
                              void* stream);

/* Box-head FC / 1x1 convolution as GEMM on f32-input MFMA:  C = act(A (M,K) * B^T + bias [+ residual])
 *   A row-major (M,K) float32; Bt row-major (N,K) float32 (a torch Linear / 1x1-conv weight as stored);
 *   bias (N) or NULL; residual (M,N) or NULL; relu != 0 fuses ReLU; C row-major (M,N). */
int wd_gemm_nt_f32(const float* A, const float* Bt, const float* bias, const float* residual, int relu,
                   int M, int N, int K, float* C, void* stream);

/* GroupNorm(groups, C, eps) + optional ReLU, in place, on n items of (HW, C) NHWC float32 (HW <= 64): the norm of
 * FastRCNNConvFCHead (job.log:1149-1150).  Biased variance, like torch.nn.GroupNorm. */
int wd_groupnorm_relu_nhwc_f32(float* x, const float* gamma, const float* beta, int n, int hw, int c, int groups,
                               float eps, int relu, void* stream);
/* The same into a separate output (y == x allowed), and its backward for the training graph (round 4; torch's group_norm copies channels_last
 * maps to NCHW and back): dx (n, HW, C), dgamma (C), dbeta (C) are overwritten; x = the forward INPUT, dy = the gradient of the (ReLU'd) output. */
int wd_groupnorm_relu_out_nhwc_f32(const float* x, float* y, const float* gamma, const float* beta, int n, int hw, int c, int groups,
                                   float eps, int relu, void* stream);
int wd_groupnorm_relu_bwd_nhwc_f32(const float* x, const float* dy, const float* gamma, const float* beta, int n, int hw, int c, int groups,
                                   float eps, int relu, float* dx, float* dgamma, float* dbeta, void* stream);

/* Fused image pre-processing in front of the detector (one HBM pass): replaces TTA.pre_process
 * (detnet/nn/tta.py:179-190 ResizeTTA = F.interpolate(scale_factor, bilinear, align_corners=False); :147-156
 * HFlipTTA / VFlipTTA = torch.flip), Detectron2Det.forward's RGB->BGR swap (detnet/nn/detectron2_det/__init__.py:70-74)
 * and detectron2's (x - PIXEL_MEAN) / PIXEL_STD + zero padding to a multiple of `divisor` (size_divisibility 32).
 *   src        : DEVICE pointer; WD_LAYOUT_NCHW_F32 = (N,3,H,W) float32 0..255 (what Detectron2Det.predict receives),
 *                WD_LAYOUT_NHWC_U8 = (N,H,W,3) uint8 (decoded camera frames)
 *   order of operations = the reference's: resize by `scale` (1.0 = none), flips, channel swap (swap_rb != 0),
 *                normalisation with mean3 / std3 (HOST pointers, indexed by OUTPUT channel; NULL = 0 / 1), padding
 *   out        : (N, Hp, Wp, 3) NHWC float32, Ho = floor(H*scale), Hp = ceil(Ho/divisor)*divisor (same for W);
 *                wd_preprocess_out_shape returns Ho, Wo, Hp, Wp (host-only helper, no GPU needed). */
#define WD_LAYOUT_NCHW_F32 0
#define WD_LAYOUT_NHWC_U8 1
int wd_preprocess_out_shape(int h, int w, double scale, int divisor, int* ho, int* wo, int* hp, int* wp);
int wd_preprocess_f32(const void* src, int src_layout, int batch, int h, int w, double scale, int hflip, int vflip,
                      int swap_rb, const float* mean3, const float* std3, int divisor, float* out, void* stream);

/* detectron2 Box2BoxTransform.apply_deltas (weights wx, wy, ww, wh; dw / dh clamped to scale_clamp = log(1000/16)) followed by
 * Boxes.clip to [0, clip_w] x [0, clip_h] (clip_w <= 0: no clipping), one launch, arithmetic identical to the torch sequence.
 *   deltas, boxes : (m, 4) float32, 16-byte aligned; index (n) int64 or NULL: row i uses deltas[index[i]], boxes[index[i]]
 *   out           : (n, 4) float32 xyxy */
int wd_decode_boxes_f32(const float* deltas, const float* boxes, const int64_t* index, int n, float wx, float wy, float ww,
                        float wh, float scale_clamp, float clip_w, float clip_h, float* out, void* stream);

/* ---- backward (training fwd+bwd) ---------------------------------------------------
 * ROIPooler backward: grad_out (R, pooled, pooled, C) is scattered (+=) into grad_feats[l] (same shapes as the
 * forward feats; the caller zero-initialises them).  grad_feats: HOST array of DEVICE pointers. */
int wd_roi_pool_fpn_bwd_f32(float* const* grad_feats, const int32_t* heights, const int32_t* widths, const float* scales,
                            int n_levels, int channels, int batch, const float* rois, int n_rois, int pooled, int min_level,
                            int canonical_level, float canonical_size, const float* grad_out, void* stream);
/* Deformable conv backward building blocks (detectron2 deformable_im2col / col2im / col2im_coord restated):
 *   im2col : col[g][p][k][ci] (g = group, p = N*Ho*Wo output pixels, k = 9 taps, ci = channel inside the group; group-major
 *            so that the per-group GEMMs are strided-batched library GEMMs without permute copies; groups = 1 gives the
 *            plain [p][k][c]) from x and offset;
 *   col2im : dcol (same layout) -> dx (N,H,W,C) += and doffset (N,Ho,Wo,18) += (both zero-initialised by the caller).
 * The weight / column GEMMs between them (dW[g] = dY[g]^T col[g], dcol[g] = dY[g] W[g]) are plain library GEMMs. */
int wd_deform_im2col_f32(const float* x, const float* offset, int batch, int h, int w, int c, int groups, int stride, int pad,
                         float* col, void* stream);
int wd_deform_col2im_f32(const float* dcol, const float* x, const float* offset, int batch, int h, int w, int c, int groups,
                         int stride, int pad, float* dx, float* doffset, void* stream);
/* Fused form of the same backward for pad 1 / stride 1 or 2 / C_out = C_in / 16 or 32 channels per group (every DeformConv of res3 and res4;
 * round 4): the column slab is never written to HBM.  h, w = the INPUT size; dy / offset / doffset have the output size ((h - 1) / stride + 1).
 *   wd_deform_dw_f32 : dw (C, C/groups, 3, 3) OIHW, the layout of the weight = sum over output pixels of dY[p][o] * col[p][tap][ci]
 *                      (the im2col + dW GEMM pair; dw is overwritten).  dy (N,H,W,C) NHWC. */
/*   y_act / scale (both may be NULL): the backward of the block's fused epilogue y = relu(conv * scale + bias) applied to dy as it is loaded,
 *                      dy_eff = dy * (y_act > 0) * scale[channel] (y_act (N,H,W,C) = the forward output; instead of a wd_act_bwd_f32 pass).
 *   wd_deform_dxoff_f32 : dx (N,H,W,C) and doffset (N,H,W,18), both overwritten (the dcol GEMM + col2im pair).  weight = the (C, C/groups,
 *                      3, 3) OIHW tensor; tables (wd_deform_bwd_tables_bytes) and packed_weight (C * 9 * C/groups floats) are scratch. */
size_t wd_deform_bwd_tables_bytes(int batch, int h, int w, int stride);
int wd_deform_dxoff_f32(const float* x, const float* offset, const float* dy, const float* y_act, const float* scale, const float* weight, int batch,
                        int h, int w, int c, int groups, int stride, unsigned char* tables, float* packed_weight, float* dx, float* doffset,
                        void* stream);
size_t wd_deform_dw_scratch_floats(int batch, int h, int w, int c, int groups, int stride);     /* per-workgroup partial sums (reduced by a second launch) */
int wd_deform_dw_f32(const float* x, const float* offset, const float* dy, const float* y_act, const float* scale, int batch, int h, int w, int c,
                     int groups, int stride, float* scratch, float* dw, void* stream);

/* 3x3 convolution (pad 1) with few output channels as "library GEMM + shift-add" (the 18-channel offset conv in front of
 * every DeformConv, job.log:412): partial (N,H,W,ld) holds, per INPUT pixel, partial[tap*n_out + n] = sum_c x[c]*w[n][c][tap]
 * (one GEMM against the (9*n_out, C) re-ordered weight, ld >= 9*n_out); this op writes
 *   out (N,Ho,Wo,n_out)[y][x][n] = bias[n] + sum_{kh,kw} partial[y*stride+kh-1][x*stride+kw-1][(kh*3+kw)*n_out + n]. */
int wd_tap_shift_add_f32(const float* partial, int ld, int n_out, const float* bias, int batch, int h, int w, int stride,
                         float* out, void* stream);

/* Nearest-neighbour x2 upsampling of an NHWC float32 map: dst (N, 2H, 2W, C)[y][x] = src (N, H, W, C)[y / 2][x / 2] - the FPN top-down
 * pathway (detectron2 fpn.py: F.interpolate(scale_factor=2, mode="nearest")).  C % 4 == 0. */
int wd_upsample2x_nhwc_f32(const float* src, int batch, int h, int w, int c, float* dst, void* stream);

/* In-place epilogue behind a library GEMM: y[m][n] = act(y[m][n] + bias[n]); y row-major (M,N), N % 4 == 0. */
int wd_bias_relu_f32(float* y, const float* bias, long m, int n, int relu, void* stream);

/* Backward of the fused epilogue y = act(z * scale[n] + bias[n]) with respect to z (training; what autograd derives from
 * detectron2's FrozenBatchNorm2d + F.relu_ behind a convolution): g[m][n] = dy[m][n] * (relu ? y[m][n] > 0 : 1) * (scale ? scale[n] : 1).
 * Row-major (M, N) float32, N % 4 == 0; g may alias dy. */
int wd_act_bwd_f32(const float* dy, const float* y, const float* scale, long m, int n, int relu, float* g, void* stream);

/* ImageOps.autocontrast(image) of PIL with cutoff 0 - the reference's AutoContrast transform (detnet/trainer/transforms/vision.py:1069-1075,
 * detnet/inference.py:171; README.md:37 runs with --auto-contrast=1) - on an (h, w, 3) uint8 DEVICE image, in place, bit-exact with PIL
 * (per channel lut[v] = clamp(int(v * (255.0 / (hi - lo)) + (-lo * scale)), 0, 255) in float64, identity when hi <= lo).
 * workspace24: 24 bytes of device memory (min / max of the three channels). */
int wd_autocontrast_u8(uint8_t* img, int h, int w, void* workspace24, void* stream);

/* GPU JPEG decode: replaces `PIL.Image.open(path).convert('RGB')` of the reference's loader
 * (detnet/data/coco.py image read + detnet/inference.py:170 ToRGB), i.e. libjpeg-turbo at its defaults (baseline Huffman,
 * JDCT_ISLOW, fancy upsampling, jdcolor YCbCr -> RGB); bit-exact with it.  Entropy decoding runs on the GPU by
 * self-synchronising 1024-bit subsequences with candidate sets (csrc/jpeg_core.h); only marker parsing and byte unstuffing stay on the host.
 *   data / n   : HOST pointer to the file bytes
 *   rgb        : DEVICE buffer of `capacity` bytes, receives (height, width, 3) uint8 RGB (grayscale files replicated,
 *                as convert('RGB') does) - the layout wd_preprocess_f32 takes as WD_LAYOUT_NHWC_U8
 *   sync_rounds: optional, number of iteration launches behind the candidate sets (2 unless the stream does not synchronise
 *                inside a subsequence, e.g. quality-100 noise)
 * The call returns when the image is complete (it synchronises `stream`).  Supported: 8-bit baseline / extended
 * sequential Huffman, one interleaved scan, grayscale or YCbCr 4:4:4 / 4:2:2 / 4:2:0, restart intervals.  Anything else
 * (progressive, arithmetic, CMYK, multi-scan, truncated data) is WT_ERR_INVALID with the reason in wt_last_error().
 * wd_jpeg_info parses the headers only (host-only, no GPU needed); restart_interval = 0 when the file has no DRI. */
int wd_jpeg_info(const uint8_t* data, int64_t n, int32_t* width, int32_t* height, int32_t* components, int32_t* h_samp,
                 int32_t* v_samp, int32_t* restart_interval);
int wd_jpeg_decode_rgb_u8(const uint8_t* data, int64_t n, uint8_t* rgb, int64_t capacity, int32_t* width, int32_t* height,
                          int32_t* sync_rounds, void* stream);
/* Statistics of the calling thread's last decode: out4 = {synchronisation launches, most iterations a workgroup needed inside
 * one launch, subsequence decodes in total over all launches, subsequences} (host-only). */
int wd_jpeg_last_stats(int32_t* out4);

#ifdef __cplusplus
}
#endif
#endif
