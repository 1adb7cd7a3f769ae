/*
 * brcnn_hip.h -- C ABI of libbrcnn_hip.so: the MI355X (gfx950) kernels of the
 * Boosting R-CNN hot path.
 *
 * Conventions (the same the reference's operator seam uses):
 *   - every pointer is a DEVICE pointer unless the name ends in `_host`;
 *   - the caller owns and pre-allocates every buffer (mmcv does
 *     `output = input.new_zeros(...)` before calling its native op);
 *   - `stream` is a hipStream_t passed as void* (NULL = the null stream); all
 *     work is enqueued on it, nothing synchronises unless stated;
 *   - return value: 0 on success, -22 (EINVAL) for a rejected argument,
 *     -(1000 + hipError_t) when a HIP call failed, -62 (BRCNN_EHANDOVER) when an
 *     EARLIER conv launch reported a lost stream-K hand-over (its results are
 *     invalid; see brcnn_conv_handover_status).  No exceptions cross the ABI;
 *   - thread-safe for calls on distinct streams with distinct workspaces
 *     (brcnn_conv_set_workspace); the `brcnn_*_set_*` tuning hooks are
 *     process-wide test / benchmarking switches: set them from one thread while
 *     no other thread is inside the library.
 *
 * Each entry cites the reference interface it replaces.  The reference tree has
 * no native code: those interfaces are the `mmcv.ops` Python functions
 * (mmcv-full 1.4.0, un-vendored) whose call sites are given as
 * /root/reference file:line.
 */
#ifndef BRCNN_HIP_H
#define BRCNN_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define BRCNN_LAYOUT_NCHW 0 /* (N,C,H,W) contiguous, the reference's layout      */
#define BRCNN_LAYOUT_NHWC 1 /* (N,H,W,C) contiguous == torch.channels_last       */

#define BRCNN_DT_F32 0
#define BRCNN_DT_BF16 1          /* bf16 operands, fp32 accumulate, bf16 result */
#define BRCNN_DT_BF16_OUT_F32 2  /* bf16 operands, fp32 accumulate, fp32 result  */
#define BRCNN_DT_F16 3           /* IEEE fp16 operands, fp32 accumulate, fp16 result (the recipes' `fp16 =
                                    dict(loss_scale=512.)`, mmdet/apis/train.py:115-119) */
#define BRCNN_DT_F16_OUT_F32 4   /* fp16 operands, fp32 accumulate, fp32 result  */

#define BRCNN_MAX_LEVELS 8
#define BRCNN_MAX_IMAGES 64      /* images per call of the whole-batch train-step entries */

/* library / device info -------------------------------------------------------- */
int brcnn_version(void);

/* Measurement aid: a one-lane kernel that spins for `wall_ticks_100mhz` ticks of the constant
 * 100 MHz counter and writes {elapsed wall ticks, elapsed shader cycles} to out2 (device, 2 x
 * int64).  Launched on a second stream beside a workload it gives the effective shader clock
 * under that load (cycles / ticks * 0.1 GHz) without a profiler attached (tools/clock_under_load.py). */
int brcnn_clock_probe(int64_t *out2, int64_t wall_ticks_100mhz, void *stream);

/* number of HIP devices visible, or a negative status */
int brcnn_device_count(void);

/* ------------------------------------------------------------------------------
 * RoIAlign.  Replaces mmcv.ops.roi_align_forward / roi_align_backward, reached at
 * mmdet/models/roi_heads/roi_extractors/base_roi_extractor.py:54-60 (construction,
 * `getattr(ops, 'RoIAlign')`) and single_level_roi_extractor.py:103 (call).
 *   input  (N,C,H,W) logical shape, memory per `layout`
 *   rois   (K,5) fp32 [batch_index, x1, y1, x2, y2]
 *   output (K,C,ph,pw) logical shape, memory per `layout` ((K,ph,pw,C) for NHWC)
 *   pool_mode 0 = max (argmax_y/argmax_x (K,C,ph,pw) required), 1 = avg
 *   sampling_ratio 0 = adaptive grid ceil(roi_size / pooled_size)
 * -------------------------------------------------------------------------- */
int brcnn_roi_align_forward(const float *input, const float *rois, float *output,
                            float *argmax_y, float *argmax_x, int batch, int channels,
                            int height, int width, int n_rois, int pooled_h, int pooled_w,
                            float spatial_scale, int sampling_ratio, int pool_mode, int aligned,
                            int layout, void *stream);

/* avg-mode backward; grad_input must be zero-filled by the caller. */
int brcnn_roi_align_backward(const float *grad_output, const float *rois, float *grad_input,
                             int batch, int channels, int height, int width, int n_rois,
                             int pooled_h, int pooled_w, float spatial_scale, int sampling_ratio,
                             int aligned, int layout, void *stream);

/* Fused SingleRoIExtractor.forward (single_level_roi_extractor.py:57-115 with
 * map_roi_levels :36-55): every RoI picks its pyramid level
 * clamp(floor(log2(sqrt(w*h)/finest_scale + 1e-6)), 0, L-1) in-kernel and is
 * pooled from that level's map; NHWC only, avg, aligned.
 *   feats[l]  (N,H_l,W_l,C) device pointers (host array of L pointers)
 *   output    (K,ph,pw,C)
 *   levels_out optional (K) int32: the level each RoI was mapped to */
int brcnn_roi_extract_forward(const void *const *feats_host, const int *heights_host,
                              const int *widths_host, const float *scales_host, int num_levels,
                              const float *rois, void *output, int32_t *levels_out, int batch,
                              int channels, int n_rois, int pooled_h, int pooled_w,
                              int sampling_ratio, float finest_scale,
                              int dtype /* of feats and output: BRCNN_DT_F32 | BRCNN_DT_BF16 */,
                              void *stream);
/* The same with `order_ws` (n_rois int32, caller-owned; NULL = visit the RoIs as given): from ~12000 RoIs on the
 * kernel visits them sorted by (image, level, 12-row band of the RoI's centre) -- one small counting-sort launch --
 * so that the bin rows an XCD has in flight stay inside what its L2 holds.  Results are those of
 * brcnn_roi_extract_forward bit for bit. */
int brcnn_roi_extract_forward_ordered(const void *const *feats_host, const int *heights_host,
                              const int *widths_host, const float *scales_host, int num_levels,
                              const float *rois, void *output, int32_t *levels_out, int batch,
                              int channels, int n_rois, int pooled_h, int pooled_w,
                              int sampling_ratio, float finest_scale,
                              int dtype /* of feats and output */, int32_t *order_ws, void *stream);

/* Round 5: the same with `prep_ws` -- brcnn_roi_extract_prep_workspace_bytes(n_rois) bytes of caller-owned scratch,
 * 256-byte aligned, or NULL.  With it the level mapping (single_level_roi_extractor.py:36-55), the RoI geometry and the
 * per-axis bilinear weights are computed once per RoI by a small first launch and handed to the gather through a
 * 576-byte record per RoI, instead of once per bin row inside the gather.  Results: those of the footprint form (bit
 * for bit on the bin rows that take its streaming path; the others run the reference's sample loop).  Measured
 * (profiles/r05_notes.md) faster only from ~1000 RoIs per image on, so the form is OFF by default
 * (brcnn_roi_align_set_exact(31) switches it on): brcnn_roi_extract_prep_workspace_bytes returns 0 while it is off. */
size_t brcnn_roi_extract_prep_workspace_bytes(int n_rois);
int brcnn_roi_extract_forward_prepared(const void *const *feats_host, const int *heights_host,
                              const int *widths_host, const float *scales_host, int num_levels,
                              const float *rois, void *output, int32_t *levels_out, int batch,
                              int channels, int n_rois, int pooled_h, int pooled_w,
                              int sampling_ratio, float finest_scale,
                              int dtype /* of feats and output */, int32_t *order_ws, void *prep_ws,
                              size_t prep_bytes, void *stream);
/* the RoI count from which a non-NULL `order_ws` is used (below it the RoIs are visited as given) */
int brcnn_roi_extract_order_min_rois(void);

int brcnn_roi_extract_backward(float *const *grad_feats_host, const int *heights_host,
                               const int *widths_host, const float *scales_host, int num_levels,
                               const float *rois, const float *grad_output, int batch,
                               int channels, int n_rois, int pooled_h, int pooled_w,
                               int sampling_ratio, float finest_scale, void *stream);

/* The same gradient as a GATHER: every pixel of every grad_feats level is written exactly once (no
 * pre-zeroing, no atomics, a fixed summation order: deterministic).  One workgroup per 8x8 tile of a
 * (level, image) map collects the RoIs whose footprint touches the tile and accumulates
 * dX[h,w] += sum_bins Wy[ph][h] * Wx[pw][w] * dY[roi,ph,pw] / count with the separable bilinear footprint
 * weights.  workspace: brcnn_roi_extract_backward_workspace_bytes(n_rois).  pooled_h, pooled_w <= 7.
 * dtype: element type of grad_output AND of the grad_feats maps (BRCNN_DT_F32, or the 16-bit compute dtype of
 * the pyramid in training: fp32 accumulation, one rounding at the store). */
size_t brcnn_roi_extract_backward_workspace_bytes(int n_rois);
/* ... plus room for the fp32 partials of the hit-chunked form: a map with at most 32 tiles of 8 x 8 pixels per image (the
 * coarse pyramid levels) is touched by most of its level's RoIs in every tile, and one workgroup per tile walks those hits
 * one after the other -- that chain, not the bytes, is the launch (512 RoIs / image of 450-800 px: 2.0 ms, of 16-110 px:
 * 0.22 ms).  With a workspace of this size the gather entries give such tiles n_rois / batch / 48 (<= 16) workgroups, each
 * taking every n-th hit in RoI order, and add their partials in a fixed order in a second launch (reproducible; not the
 * one-workgroup chain's association).  With the smaller workspace above they run the one-workgroup form. */
size_t brcnn_roi_extract_backward_workspace_bytes_ex(int n_rois, int batch, int channels, int num_levels,
                                                     const int *heights_host, const int *widths_host);
int brcnn_roi_extract_backward_gather(void *const *grad_feats_host, const int *heights_host,
                                      const int *widths_host, const float *scales_host, int num_levels,
                                      const float *rois, const void *grad_output, int batch, int channels,
                                      int n_rois, int pooled_h, int pooled_w, int sampling_ratio,
                                      float finest_scale, void *workspace, size_t workspace_bytes,
                                      int dtype, void *stream);
/* ... with one more operand per level: grad_feats[l] = gather + addends[l] (NULL array or NULL entries: none), the
 * addend in the maps' dtype, added in fp32 before the store's one rounding.  What autograd does with a separate add
 * over the whole map when a pyramid level has a second consumer besides the RoI extractor (here: the RPN branch's
 * gradient, kept from its own earlier backward pass -- detectors.py, early_rpn_backward). */
int brcnn_roi_extract_backward_gather_add(void *const *grad_feats_host, const void *const *addends_host,
                                          const int *heights_host, const int *widths_host,
                                          const float *scales_host, int num_levels, const float *rois,
                                          const void *grad_output, int batch, int channels, int n_rois,
                                          int pooled_h, int pooled_w, int sampling_ratio, float finest_scale,
                                          void *workspace, size_t workspace_bytes, int dtype, void *stream);
/* NHWC RoIAlign forward variants: 0 (default) = footprint form, column streaming (one wavefront per row of bins
 * reads every pixel of the ROW's footprint once, rows reduced first, then spread over the bins with their x weights;
 * bin rows dealt to the XCDs in contiguous eighths; equal to the reference to fp32 round-off), 1 = the reference's
 * sample-order accumulation (bit-identical to mmcv's CPU kernel; ~1.5x the L2 reads), 2 = the footprint form of
 * rounds 1-2 (per-bin loop, round-robin rows), 3 = column streaming with round-robin rows (tuning / A-B hooks). */
int brcnn_roi_align_set_exact(int exact);

/* The policy switches of the library as ONE documented struct (round 5; VERDICT r04: the integer hooks had become what
 * product defaults depend on).  Process-wide, unsynchronised: read / written from one thread while no other thread is
 * inside the library.  Every choice yields the same results bit for bit EXCEPT conv_split_k (the association of the K
 * sum changes: results then depend on the tile count, i.e. on the batch size), roi_exact_order (the footprint form
 * equals the exact form to fp32 round-off) and wgrad_slab_reduction = 0 (fp32 atomics: order-dependent sums).
 * `size` must be sizeof(brcnn_tuning) (a caller built against another layout gets BRCNN_EINVAL).
 * The integer hooks (brcnn_conv_set_tile*, brcnn_roi_align_set_exact) stay as the fine-grained test / tuning interface;
 * brcnn_set_tuning is expressed through them. */
typedef struct brcnn_tuning {
    int size;
    int conv_stream_k;                /* chained stream-K schedule of the conv kernels: 0 off, 1 by the heuristic (default), 2 wherever the tile count allows */
    int conv_split_k;                 /* split-K of launches with fewer tiles than CUs: 0 off (default), 1 heuristic, 2 forced */
    int conv_eight_phase_16bit;       /* eight-phase 256x256 kernel for bf16 / fp16: 0 never, 1 heuristic (default) */
    int conv_persistent_1x1;          /* persistent LDS-resident-weights kernel for the short-K 1x1 layers: 0 never, 1 heuristic (default), 2 forced */
    int conv_eight_phase_f32;         /* fp32 eight-phase kernel: 0 never, 1 heuristic (default), 2 forced, 128 / 256 forced with that tile height */
    int wgrad_slab_reduction;         /* weight gradient over the M slices: 0 fp32 atomics, 1 slabs + fixed-order second stage (default: deterministic) */
    int wgrad_eight_phase;            /* eight-phase weight-gradient kernel: 0 never, 1 heuristic (default), 2 wherever the shape allows */
    int wgrad_reduce_in_launch;       /* its slab reduction: 0 separate launches (default), 1 inside the producing launch (measured slower) */
    int wgrad_generation_percent;     /* two-buffer weight-gradient launches: percent of one generation of resident workgroups (default 75: they share the device with the main stream) */
    int wgrad_eight_phase_cu_percent; /* eight-phase weight-gradient launches: percent of the CUs (default 75) */
    int roi_exact_order;              /* RoIAlign forward: 0 footprint form (default), 1 the reference's sample order bit for bit */
    int roi_rows_per_wave;            /* bin rows per wavefront: 0 heuristic (default), 1, 7 */
    int roi_visit_order;              /* band-ordered visit of the RoIs: 0 off, 1 from brcnn_roi_extract_order_min_rois() RoIs on (default), 2 always */
    int roi_prepared_records;         /* per-RoI records written by a first launch: 0 off (default), 1 on where the caller gives the scratch */
} brcnn_tuning;
int brcnn_get_tuning(brcnn_tuning *t);          /* t->size set by the caller */
int brcnn_set_tuning(const brcnn_tuning *t);

/* Whole-batch candidate preparation / collection around brcnn_nms for fixed per-image slots of
 * `slot` candidates (the device-resident form of mmcv.ops.batched_nms below split_thr, called per
 * image by atss_rpn_head.py:756 and bbox_nms.py:86): brcnn_nms_prepare compacts the valid rows of
 * every slot in order (zeros behind them), writes `nms_boxes = box + float(id) * (max + 1)` with
 * max = the slot's largest surviving coordinate (mmcv's class / level separation, same fp32
 * operations) and the slot's [begin, end) range; brcnn_nms_collect gathers the first
 * min(num[b], max_keep) survivors of each slot into dets (batch, max_keep, 5) / ids_kept (zeros / -1
 * behind them).  ids int64, valid one byte per row. */
int brcnn_nms_prepare(const float *boxes, const float *scores, const int64_t *ids, const uint8_t *valid,
                      float *c_boxes, float *c_scores, int64_t *c_ids, float *nms_boxes, int32_t *ranges,
                      int batch, int slot, void *stream);
int brcnn_nms_collect(const int64_t *keep, const int32_t *num, const float *c_boxes, const float *c_scores,
                      const int64_t *c_ids, float *dets, int64_t *ids_kept, int batch, int slot,
                      int max_keep, void *stream);

/* The same for mmcv batched_nms at or above split_thr (its per-id branch, mmcv/ops/nms.py; the training
 * proposal stage has 15 150 candidates per image, atss_rpn_head.py:756, and the 80-class second stage
 * 256 x 80): candidate columns [sum level_sizes[:l], +level_sizes[l]) of a slot carry id l, so after
 * the order-preserving compaction every (image, id) is one segment -- ranges is (batch*num_levels, 2).
 * brcnn_nms_collect_sorted then re-sorts an image's survivors of all ids by score (descending, ties by
 * ascending candidate position) and keeps the first max_keep: dets (batch, max_keep, 5) zero padded,
 * ids_kept (batch, max_keep) (-1 padded) or NULL, n_kept (batch).  new_scores5: rows of 5 floats aligned
 * with `keep` whose column 4 replaces the score (soft-NMS picks), or NULL.  slot <= 16384. */
int brcnn_nms_prepare_levels(const float *boxes, const float *scores, const int64_t *ids, const uint8_t *valid,
                             float *c_boxes, float *c_scores, int64_t *c_ids, float *nms_boxes,
                             int32_t *ranges, int batch, int slot, int num_levels,
                             const int *level_sizes_host, void *stream);
int brcnn_nms_collect_sorted(const int64_t *keep, const int32_t *num, const int32_t *ranges,
                             const float *c_boxes, const float *c_scores, const int64_t *c_ids,
                             const float *new_scores5, float *dets, int64_t *ids_kept, int32_t *n_kept,
                             int batch, int slot, int num_levels, int max_keep, void *stream);

/* ------------------------------------------------------------------------------
 * NMS.  Replaces mmcv.ops.nms (ext `nms(boxes, scores, iou_threshold, offset)`),
 * reached through mmcv batched_nms at atss_rpn_head.py:756, rpn_head.py:245 and
 * core/post_processing/bbox_nms.py:86.
 * Segmented form: `num_segments` independent problems (images, or levels of the
 * split_thr path) in one launch; segment s owns boxes [seg_begin[s], seg_end[s])
 * (for back-to-back segments pass offsets and offsets+1; gaps are allowed).  Semantics per segment = mmcv nms_cpu: order by score
 * descending (ties: ascending index), greedy suppress when
 * inter/(area_i+area_j-inter) > iou_threshold.
 *   boxes (n,4) fp32, scores (n) fp32, seg_begin/seg_end (S) int32 on device
 *   keep  (n) int64: for segment s the kept ORIGINAL global indices in score
 *         order are written at keep[seg_begin[s] ...]; num_keep (S) int32
 *   max_keep > 0 stops each segment after that many survivors (mmcv `max_num`)
 *   max_segment_len: an upper bound on the longest segment (sizes the bit mask:
 *         n * ceil(max_segment_len/64) * 8 bytes)
 *   workspace: brcnn_nms_workspace_bytes(n, S, max_segment_len) bytes of scratch
 * -------------------------------------------------------------------------- */
size_t brcnn_nms_workspace_bytes(int64_t n, int num_segments, int64_t max_segment_len);
int brcnn_nms(const float *boxes, const float *scores, const int32_t *seg_begin,
              const int32_t *seg_end, int num_segments, int64_t n, int64_t max_segment_len, float iou_threshold,
              int offset, int max_keep, int64_t *keep, int32_t *num_keep, void *workspace,
              size_t workspace_bytes, void *stream);

/* ------------------------------------------------------------------------------
 * Soft-NMS.  Replaces mmcv.ops.soft_nms (ext `softnms`, CPU-only in mmcv),
 * selected by `nms=dict(type='soft_nms', ...)`
 * (configs/boosting_rcnn/boosting_rcnn_r2_101_dcn_pafpn_mstrain_3x_coco.py:27)
 * through bbox_nms.py:86.  Sequential pick-max / decay / discard semantics of
 * softnms_cpu reproduced exactly, one workgroup per segment.
 *   dets (n,5) out [x1,y1,x2,y2,decayed score] in pick order per segment
 *   inds (n) int64 out: original global indices in pick order
 *   method 0 naive, 1 linear, 2 gaussian
 * -------------------------------------------------------------------------- */
size_t brcnn_softnms_workspace_bytes(int64_t n, int num_segments);
int brcnn_softnms(const float *boxes, const float *scores, const int32_t *seg_begin,
                  const int32_t *seg_end, int num_segments, int64_t n, float iou_threshold, float sigma, float min_score,
                  int method, int offset, float *dets, int64_t *inds, int32_t *num_keep,
                  void *workspace, size_t workspace_bytes, void *stream);

/* ------------------------------------------------------------------------------
 * Sigmoid focal loss, element-wise (reduction 'none').  Replaces
 * mmcv.ops.sigmoid_focal_loss_forward/_backward reached at
 * mmdet/models/losses/focal_loss.py:86.  input (n,c) fp32, target (n) int64 in
 * [0,c] (c = background), weight (c) or NULL.
 * -------------------------------------------------------------------------- */
int brcnn_sigmoid_focal_loss_forward(const float *input, const int64_t *target,
                                     const float *weight, float *output, int64_t n, int64_t c,
                                     float gamma, float alpha, void *stream);
int brcnn_sigmoid_focal_loss_backward(const float *input, const int64_t *target,
                                      const float *weight, float *grad_input, int64_t n,
                                      int64_t c, float gamma, float alpha, void *stream);

/* ------------------------------------------------------------------------------
 * Convolution / linear stack (no mmcv counterpart: torch.nn.Conv2d/Linear +
 * BatchNorm(eval)/ReLU/residual in mmdet/models/backbones/resnet.py:263-302,
 * necks/pafpn.py:100-158, dense_heads/atss_rpn_head.py:207-215,
 * roi_heads/bbox_heads/convfc_bbox_head.py:154-192).
 * Implicit-GEMM on MFMA, NHWC activations:
 *   x  (N,H,W,Cin)   w (Cout,KH,KW,Cin)   y (N,Ho,Wo,Cout)
 *   y = act( conv(x,w) * scale[c] + shift[c] + residual ),  act = relu if `relu`
 *   scale/shift/residual may be NULL (scale -> 1, shift -> 0).
 *   dtype BRCNN_DT_F32: fp32 operands on v_mfma_f32_32x32x2_f32 (exact fp32 FMA
 *   chain); BRCNN_DT_BF16: bf16 operands, fp32 accumulate, bf16 output.
 * A linear layer is the 1x1 case with H=W=1.
 * -------------------------------------------------------------------------- */
int brcnn_conv2d_nhwc(const void *x, const void *w, const float *scale, const float *shift,
                      const void *residual, void *y, int batch, int height, int width, int cin,
                      int cout, int kh, int kw, int stride, int pad, int relu, int dtype,
                      void *stream);

/* Stride-1 convolution whose output is scattered with stride 2: output pixel (a, b) of the
 * (height + 2 pad - kh + 1) x (width + 2 pad - kw + 1) result goes to pixel
 * (2 (a - origin) + ph, 2 (b - origin) + pw) of y (batch, out_height, out_width, cout); pixels that
 * fall outside y are dropped, pixels of y of other parities are not touched.  It is one parity
 * class of the data gradient of a stride-2 convolution (x = dy, w = the class's taps as
 * (Cin_of_the_layer, KH', KW', Cout_of_the_layer)): the four classes of a 3x3 / pad 1 layer have
 * 1 + 2 + 2 + 4 = 9 taps, against 36 for the full kernel over the zero-stuffed dy that
 * brcnn_conv2d_dgrad_nhwc_multi runs (the reference gets the same effect from cuDNN's strided
 * dgrad).  cin % 32 == 0 (fp32) / % 64 (bf16). */
int brcnn_conv2d_nhwc_scatter2(const void *x, const void *w, void *y, int batch, int height, int width,
                               int cin, int cout, int kh, int kw, int pad, int out_height,
                               int out_width, int ph, int pw, int origin, int dtype, void *stream);

/* Scratch of the convolution launches of ONE stream -- the stream-K hand-over slots (fp32 accumulators of the tiles
 * cut between two workgroups), their flags, and the slabs of the sliced 16-bit weight gradient -- is CALLER-OWNED like
 * every other buffer: allocate brcnn_conv_workspace_bytes() bytes (256-byte aligned), register them for the stream
 * once, keep them alive until brcnn_conv_set_workspace(stream, NULL, 0).  The registration enqueues a memset of the
 * flag area on `stream`.  A stream that never registered gets a library-side hipMalloc on its first convolution
 * (fallback for plain C callers; released by the NULL call, which synchronises that stream). */
size_t brcnn_conv_workspace_bytes(void);
int brcnn_conv_set_workspace(void *stream, void *workspace, size_t bytes);

/* 0, or BRCNN_EHANDOVER (-62) if a K tail of a stream-K launch gave up waiting for its head since the last call
 * (reported once, then cleared).  Every conv launch checks the same word first, so the error also surfaces as the
 * status of the NEXT convolution; call this after a synchronisation to vet the launches already finished. */
#define BRCNN_EHANDOVER (-62)
int brcnn_conv_handover_status(void);

/* Tuning hook: force the workgroup tile (wm: 2 -> 128 rows, 4 -> 256 rows; nt: 1 -> 64
 * columns, 2 -> 128 columns; 0 -> built-in heuristic).  Process-wide; used by the
 * benchmarking scripts (tools/conv_bench.py).  (-1, 0/1/2): register-staged / heuristic / LDS-DMA
 * staging.  (-2, 0 / 1 / 2 / 128 / 256): the eight-phase fp32 kernel (csrc/conv_pp_f32.hip: 256- or
 * 128-row x 256-column tiles, two wave groups alternating MFMA and load blocks, chained stream-K) never /
 * by the heuristic / forced, forced with 128- / 256-row tiles.  (-3, 0 / 1 / 2): its 256 x 128 form for layers with 128
 * output channels never / by the heuristic / forced.  (-4, 0 / 1): test hook -- 1 = every conv launch (all dtypes) takes the
 * general set-up and read-out instead of the straight-line forms of full tiles and plain 1x1 layers.  Every choice gives the
 * same bits. */
int brcnn_conv_set_tile(int wm, int nt);

/* Tuning hook of the bf16 kernel: 0 heuristic; 11 / 21 / 22 = 64x64 / 128x64 / 128x128 tile on
 * 4 waves; 81 / 82 / 164 / 42 = 128x64 / 128x128 (8 waves) / 128x128 (16 waves) / 256x128 (8
 * waves); 381 / 382 / 3164 / 322 / 342 = the same tiles on a 3-deep LDS ring with counted
 * vmcnt waits, 482 = 4-deep (measured slower than two double-buffered workgroups per CU:
 * profiles/r01_notes.md).  -1 / -2: spread the LDS-DMA pieces of the next K tile between the MFMA
 * groups / issue them in front of the tile (default).  8844: the 256x256 eight-phase kernel
 * (csrc/conv_pp_bf16.hip).  -3 / -4 / -5: chained stream-K schedule (a tile cut between two workgroups
 * continues from the stored fp32 accumulators; bit-identical) off / by the heuristic / wherever the
 * tile count allows it.  -6 / -7: eight-phase kernel never / by the heuristic.  -11 / -12 (test hook): stream-K heads
 * stop / resume publishing and tails give up after 256 polls -- a lost hand-over on demand, to exercise
 * BRCNN_EHANDOVER.  -15 / -16 / -17: the persistent kernel of the plain K = 64 / 128 1x1 layers (csrc/conv1x1_stream_bf16.hip:
 * weights resident in LDS, x / residual tiles through LDS-DMA rings) never / by the heuristic (default) / wherever the shape
 * allows; bit-identical. */
int brcnn_conv_set_tile_bf16(int mtnt);

/* The same convolution over `num_segments` feature maps that share one set of weights (the
 * RetinaRPN tower and heads run over 5 pyramid levels, atss_rpn_head.py:296-297): x and y hold
 * the segments back to back ((N,H_s,W_s,Cin) then the next), one launch covers all of them. */
int brcnn_conv2d_nhwc_multi(const void *x, const void *w, const float *scale, const float *shift,
                            const void *residual, void *y, int batch, int num_segments,
                            const int *heights_host, const int *widths_host, int cin, int cout,
                            int kh, int kw, int stride, int pad, int relu, int dtype,
                            void *stream);

/* Grouped convolution (ResNeXt bottleneck conv2, mmdet/models/backbones/resnext.py:10-84).
 * x (N,H,W,Cin) fp32; w_tiles (Cout,KH,KW,window) block-diagonal per 64-channel output tile
 * (tile t = co/64 reads input channels [t*window, (t+1)*window), window = 64*cg_in/cg_out,
 * a multiple of 32; Cout % 64 == 0); same fused epilogue as brcnn_conv2d_nhwc. */
int brcnn_conv2d_nhwc_grouped(const void *x, const void *w_tiles, const float *scale,
                              const float *shift, const void *residual, void *y, int batch,
                              int height, int width, int cin, int cout, int kh, int kw, int stride,
                              int pad, int window, int relu, int dtype, void *stream);
/* its backward: dgrad with w_t_tiles = the grouped packing of the flipped, (co<->ci)-transposed
 * per-group filters; wgrad into dw_tiles (Cout,KH,KW,window) (zero-filled by the caller, fp32
 * atomics), of which the block-diagonal entries are the parameter gradient. */
int brcnn_conv2d_dgrad_nhwc_grouped(const void *dy, const void *w_t_tiles, void *dx, int batch,
                                    int in_height, int in_width, int out_height, int out_width, int cin,
                                    int cout, int kh, int kw, int stride, int pad, int window, int dtype,
                                    void *stream);
int brcnn_conv2d_wgrad_nhwc_grouped(const void *x, const void *dy, void *dw_tiles, int batch, int height,
                                    int width, int cin, int cout, int kh, int kw, int stride, int pad,
                                    int window, int dtype, void *stream);

/* Backward of the convolution (autograd of the trainable convs / FCs; the reference gets these
 * from cuDNN/cuBLAS through torch autograd).
 * dgrad: dx (N,H,W,Cin) from dy (N,Ho,Wo,Cout) and w_t (Cin,KH,KW,Cout) =
 *   flipped + transposed weights (w_t[ci,a,b,co] = w[co,KH-1-a,KW-1-b,ci]); in_* are the
 *   forward-input sizes (of dx), out_* the forward-output sizes (of dy).
 * wgrad: dw (Cout,KH,KW,Cin) += sum over pixels of dy x im2col(x); dw must be zero-filled
 *   by the caller (fp32 atomics across the slices of the pixel dimension). */
int brcnn_conv2d_dgrad_nhwc_multi(const void *dy, const void *w_t, void *dx, int batch,
                                  int num_segments, const int *in_heights_host,
                                  const int *in_widths_host, const int *out_heights_host,
                                  const int *out_widths_host, int cin, int cout, int kh, int kw,
                                  int stride, int pad, int dtype, void *stream);
int brcnn_conv2d_wgrad_nhwc_multi(const void *x, const void *dy, void *dw, int batch,
                                  int num_segments, const int *heights_host,
                                  const int *widths_host, int cin, int cout, int kh, int kw,
                                  int stride, int pad, int dtype, void *stream);
/* Deferred second stage of the sliced 16-bit weight gradients (csrc/wgrad_defer.hip).  A sliced launch of
 * brcnn_conv2d_wgrad_nhwc_multi / _grouped (bf16, fp16) writes one fp32 slab per slice of the pixel dimension and adds the
 * slabs into dw with a second launch.  After brcnn_wgrad_defer_begin(stream, arena, bytes, max_items) the launches on
 * `stream` take their slabs from the caller-owned `arena` (device memory, 256-byte aligned, >= 1 MiB; a launch whose slabs
 * exceed it keeps the immediate form) and the second stages of up to `max_items` (<= 64; <= 0: 64) launches wait for
 * brcnn_wgrad_defer_flush(stream): ONE table-driven launch with the same additions in the same order -- dw is bit for bit
 * the immediate result, but complete only after the flush.  A full arena / table flushes by itself before the next
 * producing launch.  arena NULL: deferral off (pending items are flushed first).  _flush returns the number of items
 * reduced (>= 0) or an error; _pending the number waiting.  The reference has no counterpart (cuDNN's weight gradient is
 * one call, mmdet/models/backbones/resnet.py:263-302 via torch.nn.Conv2d). */
int brcnn_wgrad_defer_begin(void *stream, void *arena, size_t bytes, int max_items);
int brcnn_wgrad_defer_flush(void *stream);
int brcnn_wgrad_defer_pending(void *stream);
int brcnn_wgrad_defer_stats(void *stream, long long *flushes, long long *items);
/* per-step operand preparation of a trainable conv: weight (Cout,Cin,KH,KW) fp32 (the reference's
 * parameter layout) -> fwd (Cout,KH,KW,Cin) and / or dgrad (Cin,KH,KW,Cout) with flipped taps, in
 * `dtype`; either output may be NULL. */
int brcnn_pack_conv_weights(const float *weight, void *fwd, void *dgrad, int cout, int cin, int kh, int kw,
                            int dtype, void *stream);
/* tuning hook of the bf16 wgrad kernel: 0 heuristic, 1 = 64x64 output tile, 2 = 128x128, 4 = 256x256 (16 waves);
 * 10 / 11: the M slices of a tile are reduced with fp32 atomics / through per-workgroup slabs and a fixed-order
 * second stage (default: reproducible bit for bit, faster) */
int brcnn_conv_set_tile_wgrad_bf16(int wt);

/* ResNet stem: 7x7 / stride 2 / pad 3 convolution of the 3-channel NCHW image + folded BN +
 * ReLU (resnet.py:599-611,631-636) -> y (N,Ho,Wo,Cout) NHWC.  w_packed fp32: (Cout,7,1,32) with
 * w_packed[co,kh,0,kw*4+c] = w[co,c,kh,kw]; 16-bit: (Cout,4,1,64) with w_packed[co,t,0,r*32+kw*4+c] =
 * w[co,c,2t+r,kw] (two filter rows per 128-byte K row); zeros elsewhere.  workspace:
 * brcnn_stem_workspace_bytes() bytes (zero-bordered NHWC4 repack of the image). */
size_t brcnn_stem_workspace_bytes(int batch, int height, int width);
int brcnn_stem7x7s2_nchw(const float *img, const void *w_packed, const float *scale,
                         const float *shift, void *y, void *workspace, int batch, int height,
                         int width, int cout, int relu, int dtype, void *stream);

/* The same stem followed by its 3x3/s2/p1 max-pool (resnet.py:611,631-636) in ONE launch, 64 output channels, ReLU
 * always: y (N,Hp,Wp,64) NHWC in `dtype` (BRCNN_DT_F32 / BF16 / F16).  The image tile of a workgroup's 7 x 8 pooled
 * outputs is staged in LDS once; the conv output never reaches memory.  w_packed: K index of filter row kh, tap kw,
 * channel c = kh*22 + kw*3 + c (fp32; stored [k/2][co/32][k%2][co%32], 154 x 64 floats) or kh*32 + kw*4 + c (16-bit;
 * stored [co][232]), zeros elsewhere.  cout must be 64. */
int brcnn_stem7x7s2_pool_nchw(const float *img, const void *w_packed, const float *scale, const float *shift,
                              void *y, int batch, int height, int width, int cout, int dtype, void *stream);

/* Tail of a frozen stage-1 Bottleneck (resnet.py Bottleneck.forward:263-302) in one fp32 launch:
 * y = relu(bn3(conv3_1x1(relu(bn2(conv2_3x3(x))))) + identity), x (N,H,W,64), w2 (64,3,3,64), w3 (256,1,1,64) packed
 * (Cout,KH,KW,Cin), scale / shift = the folded eval-mode BatchNorms (NULL: 1 / 0), identity / y (N,H,W,256); N*H*W must be
 * a multiple of 64.  The 64-channel intermediate stays in LDS; bit-identical to the two brcnn_conv2d_nhwc launches. */
int brcnn_bottleneck_tail_f32(const float *x, const float *w2, const float *scale2, const float *shift2, const float *w3,
                              const float *scale3, const float *shift3, const float *identity, float *y, int batch,
                              int height, int width, void *stream);
/* ... and with 16-bit tensors (dtype BRCNN_DT_BF16 / BRCNN_DT_F16: x, w2, w3, identity, y; fp32 accumulation, the
 * intermediate rounded to the 16-bit type as the two-launch form stores it); N*H*W must be a multiple of 128. */
int brcnn_bottleneck_tail_16(const void *x, const void *w2, const float *scale2, const float *shift2, const void *w3,
                             const float *scale3, const float *shift3, const void *identity, void *y, int batch,
                             int height, int width, int dtype, void *stream);

/* 3x3/s2/p1 max-pool of the ResNet stem (resnet.py:611), NHWC fp32/bf16 */
int brcnn_maxpool3x3s2_nhwc(const void *x, void *y, int batch, int height, int width,
                            int channels, int dtype, void *stream);
/* its backward (a trainable stem: frozen_stages < 0): dx[n,h,w,c] = sum of dy over the windows whose first
 * maximum (row-major scan, torch's argmax rule) is (h,w); x / y are the forward's input / output */
int brcnn_maxpool3x3s2_nhwc_backward(const void *x, const void *y, const void *dy, void *dx, int batch,
                                     int height, int width, int channels, int dtype, void *stream);

/* GroupNorm(+ReLU) over NHWC (RPN tower ConvModule norm, atss_rpn_head.py:118,150-190):
 * y = relu?((x-mean_g)/sqrt(var_g+eps)*gamma[c]+beta[c]); stats per (n, group). */
int brcnn_groupnorm_nhwc(const void *x, const float *gamma, const float *beta, void *y,
                         void *stats_ws /* batch*groups*2 doubles of device scratch */,
                         int batch, int hw, int channels, int groups, float eps, int relu,
                         int dtype, void *stream);

int brcnn_groupnorm_nhwc_multi(const void *x, const float *gamma, const float *beta, void *y,
                               void *stats_ws /* batch*num_segments*groups*2 doubles */,
                               int batch, int num_segments, const int *hw_host, int channels,
                               int groups, float eps, int relu, int dtype, void *stream);

/* Backward of brcnn_groupnorm_nhwc_multi (training of the RPN tower: GroupNorm(32) + ReLU of
 * mmcv's ConvModule, atss_rpn_head.py:150-170; torch's native group-norm backward needs NCHW
 * and costs two layout copies per layer and direction).  `x` is the forward input, `stats` the
 * forward's stats_ws (mean / rstd per (segment, image, group) after the forward call), `relu`
 * the forward's flag (the clipped positions are recomputed from x).  Writes dx (same layout and
 * dtype as x), dgamma / dbeta (channels, fp32, overwritten).  Deterministic except for the
 * double-precision group sums (atomics). */
size_t brcnn_groupnorm_nhwc_multi_backward_workspace_bytes(int batch, int num_segments, const int *hw_host,
                                                           int channels, int groups);
int brcnn_groupnorm_nhwc_multi_backward(const void *dy, const void *x, const void *stats, const float *gamma,
                                        const float *beta, void *dx, float *dgamma, float *dbeta,
                                        void *workspace, size_t workspace_bytes, int batch, int num_segments,
                                        const int *hw_host, int channels, int groups, int relu, int dtype,
                                        void *stream);

/* Element-wise tail of a trainable conv block in training (eval-mode BatchNorm = per-channel
 * affine whose gamma / beta still train, resnet.py:263-302 with norm_eval=True):
 *   forward   out = [relu](z * scale[c] + shift[c] [+ residual])          (rows, C) NHWC rows
 *   backward  dpre = dout * (out > 0);  dz = dpre * scale[c];  dres = dpre (optional);
 *             dscale[c] = sum_rows dpre * z;  dshift[c] = sum_rows dpre    (fp32; two-stage,
 *             deterministic: per-strip partials in `workspace`, then a column reduction)
 * C % 4 == 0 (fp32) / C % 8 == 0 (bf16); backward needs C/vec to be a power of two or >= 256. */
int brcnn_bn_act_forward(const void *z, const float *scale, const float *shift, const void *residual,
                         void *out, int64_t rows, int channels, int relu, int dtype, void *stream);
size_t brcnn_bn_act_backward_workspace_bytes(int64_t rows, int channels, int dtype);
int brcnn_bn_act_backward(const void *dout, const void *out, const void *z, const float *scale,
                          void *dz, void *dres, float *dscale, float *dshift, void *workspace,
                          size_t workspace_bytes, int64_t rows, int channels, int relu, int dtype,
                          void *stream);

/* Training forward of conv -> eval-mode BatchNorm [-> + residual] [-> ReLU] in one launch (the trainable
 * Bottleneck layers, backbones/resnet.py:263-302 with norm_eval=True): z_out receives the raw conv output
 * (what brcnn_bn_eval_act_backward needs for dgamma and for the recomputed ReLU mask), y the activation;
 * both (rows, cout) in `dtype`.  mean / var NULL: gamma / beta are a plain per-channel scale / shift.
 * 16-bit dtypes (BRCNN_DT_BF16 / BRCNN_DT_F16), cout % 8 == 0; geometry as brcnn_conv2d_nhwc_multi. */
int brcnn_conv2d_bn_act_nhwc_multi(const void *x, const void *w, const float *gamma, const float *beta,
                                   const float *mean, const float *var, float eps, const void *residual,
                                   void *z_out, void *y, int batch, int num_segments,
                                   const int *heights_host, const int *widths_host, int cin, int cout,
                                   int kh, int kw, int stride, int pad, int relu, int dtype, void *stream);

/* Data gradient of a conv whose input came out of conv -> eval-BN -> [ReLU] (conv2 / conv3 of a Bottleneck,
 * backbones/resnet.py:263-302), with THAT BatchNorm's backward folded into the epilogue: what leaves is
 * dz_prev = (dx masked by the producer's ReLU, z_prev * scale + shift > 0) * scale -- the gradient of the
 * producer's raw conv output -- plus dgamma / dbeta of the producer's BatchNorm (per-row-tile partial sums in
 * `workspace`, summed in a fixed order).  dy (batch, out_h, out_w, cout), w_t (cin, kh, kw, cout) flipped /
 * transposed as for brcnn_conv2d_dgrad_nhwc_multi, z_prev / dz_prev (batch, in_h, in_w, cin).  16-bit dtypes,
 * cin % 64 == 0, cout % 64 == 0.  Values equal brcnn_conv2d_dgrad_nhwc_multi followed by
 * brcnn_bn_eval_act_backward (dz bit for bit; dgamma / dbeta up to the summation order).
 * Residual producer (bn3 of the previous Bottleneck, whose output is this conv's input AND the identity of this
 * block): dskip = the identity branch's gradient (added to the data gradient first, as
 * brcnn_conv2d_nhwc's residual operand does), prev_out = the producer's output (ReLU mask: > 0 instead of the
 * recomputed z * scale + shift), dres receives the masked sum -- the previous block's identity gradient.  All
 * three NULL for a producer without residual. */
size_t brcnn_conv2d_dgrad_bn_backward_workspace_bytes(int batch, int in_height, int in_width, int cin);
int brcnn_conv2d_dgrad_bn_backward_nhwc(const void *dy, const void *w_t, const void *z_prev, const float *gamma,
                                        const float *beta, const float *mean, const float *var, float eps, int relu,
                                        const void *dskip, const void *prev_out, void *dres,
                                        void *dz_prev, float *dgamma, float *dbeta, void *workspace,
                                        size_t workspace_bytes, int batch, int in_height, int in_width,
                                        int out_height, int out_width, int cin, int cout, int kh, int kw, int stride,
                                        int pad, int dtype, void *stream);

/* The same pair with the eval-mode BatchNorm parameters themselves (norm_eval=True,
 * backbones/resnet.py:648-657): scale = gamma / sqrt(var + eps), shift = beta - mean * scale are formed
 * inside the kernels (the reference spends ~5 element-wise torch launches per layer and direction on
 * them), and the backward returns dgamma = (sum dpre*z - mean * sum dpre) / sqrt(var + eps),
 * dbeta = sum dpre directly. */
int brcnn_bn_eval_act_forward(const void *z, const float *gamma, const float *beta, const float *mean,
                              const float *var, float eps, const void *residual, void *out, int64_t rows,
                              int channels, int relu, int dtype, void *stream);
/* `out` (the forward output, read for the ReLU mask) may be NULL when the forward had NO residual and
 * `beta` is given: the mask is then recomputed as z * scale + shift > 0 and one of the four streams
 * of the backward pass is not read at all. */
int brcnn_bn_eval_act_backward(const void *dout, const void *out, const void *z, const float *gamma,
                               const float *beta, const float *mean, const float *var, float eps, void *dz, void *dres,
                               float *dgamma, float *dbeta, void *workspace, size_t workspace_bytes,
                               int64_t rows, int channels, int relu, int dtype, void *stream);
/* Deferred second stage.  Both backward entries above end with a small launch that turns the per-strip partial sums
 * in `workspace` into dgamma / dbeta (16 - 128 workgroups, ~7 us): a backward pass issues ~40 of them, serial on
 * the launching stream, and nothing reads their results before the optimizer.  With defer_second_stage != 0 the `_ex`
 * forms record that stage instead (workspace, dgamma, dbeta, mean, var must stay valid until the flush) and
 * brcnn_bn_reduce_flush(stream) runs everything recorded for `stream` in one launch per 48 layers -- same sums in the
 * same order, same bits; returns the number of stages it ran, < 0 on error.  brcnn_bn_reduce_pending: recorded, not yet
 * run (all streams). */
int brcnn_bn_eval_act_backward_ex(const void *dout, const void *out, const void *z, const float *gamma,
                                  const float *beta, const float *mean, const float *var, float eps, void *dz, void *dres,
                                  float *dgamma, float *dbeta, void *workspace, size_t workspace_bytes,
                                  int64_t rows, int channels, int relu, int dtype, void *stream, int defer_second_stage);
int brcnn_conv2d_dgrad_bn_backward_nhwc_ex(const void *dy, const void *w_t, const void *z_prev, const float *gamma,
                                           const float *beta, const float *mean, const float *var, float eps, int relu,
                                           const void *dskip, const void *prev_out, void *dres,
                                           void *dz_prev, float *dgamma, float *dbeta, void *workspace,
                                           size_t workspace_bytes, int batch, int in_height, int in_width,
                                           int out_height, int out_width, int cin, int cout, int kh, int kw, int stride,
                                           int pad, int dtype, void *stream, int defer_second_stage);
int brcnn_bn_reduce_flush(void *stream);
int brcnn_bn_reduce_pending(void);

/* Res2Net / DCNv2 rows (the r2_101 recipes): NHWC average pooling with torch.nn.AvgPool2d's
 * window / divisor rules (res2net.py:52-54, 173-178), and mmcv's modulated deformable im2col
 * (ModulatedDeformConv2dPack, deform_groups 1): col (N*Ho*Wo, KH*KW*channels_padded) with K
 * order (tap, c); `offset_mask` is the raw conv_offset output (N,Ho,Wo,>=3*KH*KW) NHWC with
 * om_stride floats per pixel: [2t] = dy, [2t+1] = dx of tap t, [2*KH*KW + t] = mask logit. */
int brcnn_avgpool_nhwc(const float *x, float *y, int batch, int height, int width, int channels,
                       int kernel, int stride, int pad, int ceil_mode, int count_include_pad,
                       void *stream);
int brcnn_deform_im2col_nhwc(const float *x, const float *offset_mask, float *col, int batch,
                             int height, int width, int channels, int kh, int kw, int stride, int pad,
                             int dilation, int om_stride, int channels_padded, void *stream);
/* backward of the above: dx (zero-filled by the caller, fp32 atomics) and d_offset_mask (N,Ho,Wo,
 * 3*KH*KW; every entry written) from dcol; the mask sigmoid's derivative is included. */
int brcnn_deform_col2im_nhwc(const float *x, const float *offset_mask, const float *dcol, float *dx,
                             float *d_offset_mask, int batch, int height, int width, int channels, int kh,
                             int kw, int stride, int pad, int dilation, int om_stride, int channels_padded,
                             void *stream);

/* FPN top-down path: dst[n,y,x,c] += src[n, y*Hs/Hd, x*Ws/Wd, c]  (nearest,
 * F.interpolate(size=...) at necks/pafpn.py:113-115, fpn.py:178-181) */
int brcnn_upsample_nearest_add_nhwc(void *dst, const void *src, int batch, int hd, int wd,
                                    int hs, int ws, int channels, int dtype, void *stream);
/* differentiable form: out = dst + nearest_upsample(src) into a separate tensor, and the gradient w.r.t.
 * src (every source pixel sums the gradient of the destination pixels that read it; d/d dst is dout) */
int brcnn_upsample_nearest_add_nhwc_out(const void *dst, const void *src, void *out, int batch, int hd,
                                        int wd, int hs, int ws, int channels, int dtype, void *stream);
int brcnn_upsample_nearest_add_nhwc_backward(const void *dout, void *dsrc, int batch, int hd, int wd, int hs,
                                             int ws, int channels, int dtype, void *stream);

/* column sums of a (rows, channels) fp32 / bf16 tensor into fp32 (bias gradients of the convs / FCs:
 * db = sum_rows dy), two fixed-order stages (deterministic); channels / 4 a power of two or >= 256 */
size_t brcnn_colsum_workspace_bytes(int64_t rows, int channels);
int brcnn_colsum(const void *x, float *out, void *workspace, size_t workspace_bytes, int64_t rows,
                 int channels, int dtype, void *stream);

/* layout shuffles between the reference's NCHW boundary and the NHWC interior */
int brcnn_nchw_to_nhwc(const float *src, void *dst, int batch, int channels, int hw,
                       int dst_dtype, void *stream);
int brcnn_nhwc_to_nchw(const void *src, float *dst, int batch, int channels, int hw,
                       int src_dtype, void *stream);

/* ------------------------------------------------------------------------------
 * RetinaRPN proposal stage (ATSSRPNHead._get_bboxes_single,
 * atss_rpn_head.py:688-760, with AnchorGenerator.single_level_grid_anchors
 * core/anchor/anchor_generator.py:336-381 and delta2bbox
 * core/bbox/coder/delta_xywh_bbox_coder.py:145-272 fused in).
 *
 * brcnn_rpn_score: score[r*A + a] = sqrt(sigmoid(cls)*sigmoid(iou)) over `rows` pixels of
 *   NHWC head output; consecutive pixels are cls_stride / iou_stride floats apart (A when the
 *   tensor is dense, 54 when the fused cls|reg|iou head output is read in place).
 *   bbox_pred likewise has pred_stride floats per pixel and is multiplied by pred_scale
 *   (the level's learnable Scale) before decoding.
 * brcnn_rpn_decode: for `count` selected anchors (flat index into (H*W*A) of one
 *   level) regenerate the anchor from base_anchors (A,4) + stride, apply
 *   delta2bbox(means 0, stds `std4`, wh_ratio_clip) with max_shape clipping; writes
 *   proposals (count,4) and valid (count) uint8 = w > min_size && h > min_size.
 * brcnn_rpn_topk: for every (image, level) the `k` best scores of the level's `n` anchors in
 *   (score descending, index ascending) order -- the reference sorts the whole level and keeps
 *   nms_pre (atss_rpn_head.py:727-737); exact radix select + an in-LDS sort of the winners.
 *   score_levels / out_score / out_idx are HOST arrays of device pointers, one per level:
 *   score (batch, n_l), out_score / out_idx (batch, min(k, n_l)).  A level with n_l <= k is
 *   passed through in index order, as the reference does.  k <= 4096.  Levels of >= 32768 anchors
 *   are selected in up to 8 parts and merged (two launches; device scratch in `workspace`).
 * -------------------------------------------------------------------------- */
int brcnn_rpn_score(const float *cls, const float *iou, float *score, int64_t rows,
                    int num_anchors, int cls_stride, int iou_stride, void *stream);
int brcnn_rpn_decode(const int64_t *topk_inds, const float *bbox_pred, int pred_stride,
                     float pred_scale, const float *base_anchors,
                     int batch, int count, int height, int width, int num_anchors, int stride_w,
                     int stride_h, const float *means4_host, const float *stds4_host,
                     double wh_ratio_clip, float max_h, float max_w, float min_size,
                     float *proposals, uint8_t *valid, void *stream);

/* brcnn_rpn_decode for all pyramid levels in one launch: level l's picked anchors (topk_inds[l],
 * (batch, counts[l])) are decoded into columns [sum counts[:l], +counts[l]) of proposals
 * (batch, T, 4) / valid (batch, T) / ids (batch, T) (ids = the level index: the `level_ids` of
 * atss_rpn_head.py:738-740 that batched_nms separates by).  Host arrays of device pointers. */
int brcnn_rpn_decode_levels(const int64_t *const *topk_inds, const float *const *bbox_pred,
                            const int *pred_strides, const float *pred_scales,
                            const float *const *base_anchors, int batch, int num_levels, const int *counts,
                            const int *heights, const int *widths, int num_anchors, const int *strides_w,
                            const int *strides_h, const float *means4_host, const float *stds4_host,
                            double wh_ratio_clip, float max_h, float max_w, float min_size,
                            float *proposals, uint8_t *valid, int64_t *ids, void *stream);
/* the same with the per-level scales read from DEVICE memory (num_levels floats): the training
 * proposal stage decodes with the live Scale parameters without reading them back to the host;
 * max_shape_dev (batch, 2) [h, w] device = per-image clip border (img_shape differs inside a
 * training batch), or NULL to use max_h / max_w */
int brcnn_rpn_decode_levels_dscale(const int64_t *const *topk_inds, const float *const *bbox_pred,
                                   const int *pred_strides, const float *pred_scales_dev,
                                   const float *max_shape_dev, const float *const *base_anchors, int batch, int num_levels, const int *counts,
                                   const int *heights, const int *widths, int num_anchors, const int *strides_w,
                                   const int *strides_h, const float *means4_host, const float *stds4_host,
                                   double wh_ratio_clip, float max_h, float max_w, float min_size,
                                   float *proposals, uint8_t *valid, int64_t *ids, void *stream);

/* Second-stage candidates of a whole batch in one launch: the score fusion of
 * prob_roi_head.py:232-240 and ProbConvFCBBoxHead.get_bboxes (convfc_bbox_head.py:294-330) up to
 * the NMS call.  probs (batch*per_image, C+1) softmax outputs, bbox_pred (batch*per_image, 4C),
 * proposals (batch, per_image, 5) [x1,y1,x2,y2,prior] zero padded with num[b] real rows,
 * max_shape (batch, 2) [h, w], scale_factor (batch, 4) or NULL.  Writes, in (proposal, class)
 * order, boxes (batch, per_image*C, 4), scores = sqrt(prob * prior), labels = class, valid =
 * score > score_thr and the row is real. */
int brcnn_rcnn_decode(const float *probs, const float *bbox_pred, const float *proposals, const int32_t *num,
                      const float *max_shape, const float *scale_factor, int batch, int per_image,
                      int num_classes, float score_thr, const float *means4_host, const float *stds4_host,
                      double wh_ratio_clip, float *boxes, float *scores, int64_t *labels, uint8_t *valid,
                      void *stream);
size_t brcnn_rpn_topk_workspace_bytes(const int *n_host, int num_levels, int batch, int k);
int brcnn_rpn_topk(const float *const *score_levels, const int *n_host, int num_levels,
                   int batch, int k, float *const *out_score, int64_t *const *out_idx,
                   void *workspace, size_t workspace_bytes, void *stream);

/* ------------------------------------------------------------------------------
 * Input front door: the reference's Resize -> RandomFlip -> Normalize -> Pad
 * transforms (mmdet/datasets/pipelines/transforms.py:31-315,318-470,700-739,625-697; the image
 * arithmetic is mmcv.imresize = cv2.resize INTER_LINEAR on uint8, mmcv.imflip,
 * mmcv.imnormalize, mmcv.impad) of ONE decoded image as one pass.
 *   src  (src_h, src_w, 3) uint8, BGR, dense        dst (3, pad_h, pad_w) fp32, CHW
 *   the image is resized to (new_h, new_w) with OpenCV's 8-bit fixed-point bilinear, flipped
 *   (flip: 0 none, 1 horizontal, 2 vertical, 3 diagonal), converted BGR->RGB when to_rgb,
 *   normalised (v - mean[c]) * fp32(1/std[c]) (mean/std in the OUTPUT channel order, host
 *   arrays of 3) and zero-padded at the bottom / right to (pad_h, pad_w).
 * -------------------------------------------------------------------------- */
int brcnn_preprocess_u8(const uint8_t *src, int src_h, int src_w, float *dst, int new_h,
                        int new_w, int pad_h, int pad_w, int flip, const float *mean3_host,
                        const float *std3_host, int to_rgb, void *stream);

/* ------------------------------------------------------------------------------
 * Train step: whole-batch target assignment, RoI sampling and the fused losses.
 * The reference runs these per image / per level as chains of small torch ops with host
 * synchronisations (nonzero / unique / .item()); each entry below serves the whole batch.
 * Ground truth is passed flat: gts (sum G, 4) [+ gt_labels (sum G) int64], image b owning rows
 * [gt_offsets_host[b], gt_offsets_host[b+1]) (HOST array of batch+1 ints; batch <= BRCNN_MAX_IMAGES).
 *
 * brcnn_assign_max_iou -- MaxIoUAssigner.assign (mmdet/core/bbox/assigners/max_iou_assigner.py:61-213)
 *   over BboxOverlaps2D (core/bbox/iou_calculators/iou2d_calculator.py:30-261, mode 'iou').
 *   boxes: image b's n rows start at boxes + b*box_batch_stride (0: shared anchors), box_row_stride
 *   floats per row (5 for proposals with a score column); num_boxes (batch) real rows or NULL.
 *   num_levels > 0 describes RPN anchors (level_start_host[L+1] first anchor of each level,
 *   level_width_host[L] cells per row, anchors_per_cell): valid_hw (batch, L, 2) device
 *   [valid_h, valid_w] in cells reproduces AnchorGenerator.valid_flags
 *   (core/anchor/anchor_generator.py:383-440), img_hw (batch, 2) + allowed_border >= 0
 *   anchor_inside_flags (core/anchor/utils.py:21-47); either may be NULL.  Boxes that fail take no
 *   part and come out as -1.  neg_iou_thr as [neg_iou_lo, neg_iou_hi) (a float thr is [0, thr)).
 *   gt_max_ws (sum G) uint32 scratch for match_low_quality.  Writes gt_inds (batch, n) int32
 *   (-1 ignore, 0 negative, k matched to gt k-1), max_overlaps (batch, n) or NULL, counts (batch, 2)
 *   [#positive, #negative] or NULL.  Index-exact with the reference on the host (first maximum on
 *   ties, later ground truth overriding earlier in the low-quality pass).
 *
 * brcnn_rcnn_sample -- RandomSampler.sample (core/bbox/samplers/base_sampler.py:35-102,
 *   random_sampler.py:32-82), SamplingResult (sampling_result.py:26-55), the prior extraction of
 *   ProbRoIHead.forward_train (models/roi_heads/prob_roi_head.py:51-64), bbox2roi
 *   (core/bbox/transforms.py:59-78) and BBoxHead._get_target_single (bbox_heads/bbox_head.py:122-196).
 *   proposals (batch, per_image, 5) zero padded with num_props[b] real rows; gt_inds / max_overlaps
 *   from brcnn_assign_max_iou on them.  perm (batch, num_expected_pos + num) int32: image b's first
 *   num_expected_pos entries are torch.randperm(n_pos)[:num_expected_pos] when n_pos exceeds it, the
 *   rest randperm(n_neg)[:expected_neg] likewise (drawn on the host: the reference's seeded stream);
 *   row_offsets_host (batch+1): first output row of each image (host knows the counts).
 *   list_ws (batch, 2, list_stride) int32 scratch, list_stride >= per_image + max G.
 *   Writes rois (N,5), labels (N) int64 (background = num_classes), bbox_targets (N,4) (encoded
 *   with means / stds, or the matched gt box when reg_decoded_bbox), priors (N), ious (N) or NULL
 *   (`quality`), pos_flags (N) int32 or NULL; rows of an image are [positives | negatives], each
 *   ascending in candidate index.
 *
 * brcnn_rpn_loss_* -- ATSSRPNHead.loss / loss_single (models/dense_heads/atss_rpn_head.py:299-464)
 *   for every RPN loss configuration of the recipes: reg_decoded_bbox=True with IoULoss(mode 'log')
 *   [+ MSELoss aug] or reg_decoded_bbox=False with CIoULoss on the raw deltas (:361-374, the FPN
 *   recipe); FocalLoss or VarifocalLoss (:393-400, the VOC recipe); sigmoid-BCE IoU branch.
 *   y (rows, ystride) is the fused head output of all levels, level-major rows
 *   (level l: batch*H_l*W_l rows), channels [cls A | reg 4A | iou A | padding]; scales (L) the
 *   per-level Scale parameters (device); gt_inds (batch, anchors per image) from
 *   brcnn_assign_max_iou.  cfg20_host = [focal_gamma, focal_alpha, pos_weight, iou_gamma(self.gamma),
 *   mean4, std4, max_ratio |ln(wh_ratio_clip)|, with_aug, lw_cls, lw_bbox, lw_aug, lw_iou,
 *   cls_mode (0 FocalLoss, 1 VarifocalLoss iou_weighted, 2 VarifocalLoss not iou_weighted; gamma /
 *   alpha are then the varifocal ones), reg_mode (0 decoded IoU-log, 1 CIoU on raw deltas)].
 *   forward: sums (L, 6) [focal, iou-loss, mse, bce, iou_target, n_pos] and totals (2)
 *   [n_pos, sum iou_target] of THIS rank; the caller averages totals over ranks (reduce_mean,
 *   :440-444,458-460) and calls finalize: losses3 = [loss_rpn_cls, loss_rpn_bbox, loss_rpn_iou]
 *   summed over levels, per_level (3, L), coef2 = the two reciprocal normalisers the backward uses.
 *   backward: grad3 (3) device = dLoss/dlosses3; writes dy (rows, ystride) and dscales (L).
 *   Deterministic (fixed-order partial sums, no float atomics).
 *
 * brcnn_boost_loss_* -- ProbRoIHead._bbox_forward_train_boost + norm_loss
 *   (models/roi_heads/prob_roi_head.py:107-154) over ProbConvFCBBoxHead.loss
 *   (bbox_heads/convfc_bbox_head.py:332-418; CrossEntropyLoss + L1Loss, reduction 'none').
 *   cfg6_host = [gamma, alpha, iou_gamma, loss_cls weight, loss_bbox weight, reg_norm == 'mean'].
 *   out3 = [loss_cls, loss_bbox, acc]; backward: dcls (n, C+1), dbbox (n, 4C | 4).
 * -------------------------------------------------------------------------- */
/* bbox_overlaps (mmdet/core/bbox/iou_calculators/iou2d_calculator.py:75-261): mode 0 'iou', 1 'iof',
 * 2 'giou'; out (n1, n2), or (n1) when is_aligned (n1 == n2).  stride = floats per box row (4, or 5 with a
 * score column, which BboxOverlaps2D.__call__ strips, :30-66).  union / enclosing area clamped at eps. */
int brcnn_bbox_overlaps(const float *bboxes1, int stride1, int n1, const float *bboxes2, int stride2, int n2,
                        int mode, int is_aligned, float eps, float *out, void *stream);
int brcnn_assign_max_iou(const float *boxes, int64_t box_batch_stride, int box_row_stride,
                         const int32_t *num_boxes, int n, int batch, const float *gts,
                         const int *gt_offsets_host, int num_levels, const int *level_start_host,
                         const int *level_width_host, int anchors_per_cell, const int32_t *valid_hw,
                         const float *img_hw, float allowed_border, float pos_iou_thr, float neg_iou_lo,
                         float neg_iou_hi, float min_pos_iou, int match_low_quality, uint32_t *gt_max_ws,
                         int32_t *gt_inds, float *max_overlaps, int32_t *counts, void *stream);
int brcnn_rcnn_sample(const float *proposals, const int32_t *num_props, int per_image, int batch,
                      const int32_t *gt_inds, const float *max_overlaps, const float *gts,
                      const int64_t *gt_labels, const int *gt_offsets_host, int add_gt_as_proposals,
                      int num, int num_expected_pos, float neg_pos_ub, const int32_t *perm,
                      const int *row_offsets_host, int num_classes, int reg_decoded_bbox,
                      const float *means4_host, const float *stds4_host, int32_t *list_ws,
                      int list_stride, float *rois, int64_t *labels, float *bbox_targets, float *priors,
                      float *ious, int32_t *pos_flags, void *stream);
size_t brcnn_rpn_loss_workspace_bytes(int batch, int num_levels, const int *heights, const int *widths,
                                      int anchors_per_cell);
int brcnn_rpn_loss_forward(const float *y, int ystride, int batch, int num_levels, const int *heights,
                           const int *widths, const int *strides_w, const int *strides_h,
                           const float *const *base_anchors, int anchors_per_cell, const float *scales,
                           const int32_t *gt_inds, const float *gts, const int *gt_offsets_host,
                           const float *cfg20_host, void *workspace, size_t workspace_bytes, float *sums,
                           float *totals, void *stream);
int brcnn_rpn_loss_finalize(const float *sums, const float *totals, int num_levels, const float *cfg20_host,
                            float *losses3, float *per_level, float *coef2, void *stream);
int brcnn_rpn_loss_backward(const float *y, int ystride, int batch, int num_levels, const int *heights,
                            const int *widths, const int *strides_w, const int *strides_h,
                            const float *const *base_anchors, int anchors_per_cell, const float *scales,
                            const int32_t *gt_inds, const float *gts, const int *gt_offsets_host,
                            const float *cfg20_host, const float *grad3, const float *coef2,
                            void *workspace, size_t workspace_bytes, float *dy, float *dscales,
                            void *stream);
size_t brcnn_boost_loss_workspace_bytes(int n);
int brcnn_boost_loss_forward(const float *cls_score, const float *bbox_pred, const int64_t *labels,
                             const float *priors, const float *ious, const float *bbox_targets, int n,
                             int num_classes, int reg_class_agnostic, const float *cfg6_host,
                             void *workspace, size_t workspace_bytes, float *out3, float *coef2,
                             void *stream);
int brcnn_boost_loss_backward(const float *cls_score, const float *bbox_pred, const int64_t *labels,
                              const float *priors, const float *ious, const float *bbox_targets, int n,
                              int num_classes, int reg_class_agnostic, const float *cfg6_host,
                              const float *grad3, const float *coef2, float *dcls, float *dbbox,
                              void *stream);
/* the same kernels with two more switches, cfg8_host = cfg6 + [plain, beta]:
 *   plain != 0: the boosted weights are LABEL WEIGHTS of the head's own loss -- loss_cls = sum(w * ce) / max(#{w > 0}, 1)
 *     (DyProbRoIHead._bbox_forward_train_boost, models/roi_heads/prob_roi_head.py:604-623 over
 *     bbox_heads/bbox_head.py loss; also gamma = 0 for its un-boosted branch) instead of norm_loss;
 *   beta > 0: SmoothL1Loss(beta) for the box term (models/losses/smooth_l1_loss.py:9-32; Dynamic R-CNN moves beta
 *     every update_iter_interval iterations), beta <= 0: L1Loss. */
int brcnn_boost_loss_forward_ex(const float *cls_score, const float *bbox_pred, const int64_t *labels,
                                const float *priors, const float *ious, const float *bbox_targets, int n,
                                int num_classes, int reg_class_agnostic, const float *cfg8_host,
                                void *workspace, size_t workspace_bytes, float *out3, float *coef2,
                                void *stream);
int brcnn_boost_loss_backward_ex(const float *cls_score, const float *bbox_pred, const int64_t *labels,
                                 const float *priors, const float *ious, const float *bbox_targets, int n,
                                 int num_classes, int reg_class_agnostic, const float *cfg8_host,
                                 const float *grad3, const float *coef2, float *dcls, float *dbbox,
                                 void *stream);

/* ------------------------------------------------------------------------------
 * Optimizer step of the train loop: SGD with momentum and weight decay after clipping the global gradient
 * norm (torch.optim.SGD + mmcv OptimizerHook grad_clip; configs/_base_/schedules/schedule_1x.py:2-3,
 * configs/boosting_rcnn/boosting_rcnn_r50_pafpn_1x_utdac.py:130), as three stages over tables of tensors
 * passed by value (HOST arrays of device pointers; fp32 dense tensors):
 *   ctl3 = [ ||g||_2 / loss_scale, applied factor, skipped ]; with skip_nonfinite the step is skipped when the norm
 *   is not finite (GradScaler.step, the fp16 recipes); without it the update runs as clip_grad_norm_ + SGD would
 *   d = g * factor + wd * w;  buf = has_buf ? momentum * buf + d : d;  w -= lr * buf   (dampening 0, no nesterov)
 * brcnn_pack_conv_weights_batch: forward (Cout,KH,KW,Cin) and data-gradient (Cin,KH,KW,Cout, taps flipped)
 * operands of many conv weights in one launch per 64 tensors (dims_host: cout, cin, kh, kw each;
 * channels_last_host[i] != 0: master weight i is stored with torch.channels_last strides -- the layout the
 * weight-gradient kernel writes, so that its result IS the parameter's gradient without a layout copy; NULL =
 * all contiguous); with ctl3 the packing follows the step it belongs to (skipped together).
 * -------------------------------------------------------------------------- */
size_t brcnn_sgd_workspace_bytes(int num_tensors, const int64_t *numel_host);
int brcnn_sgd_step(float *const *params, const float *const *grads, float *const *bufs, const int64_t *numel_host,
                   const float *lr_host, const float *wd_host, const int *has_buf_host, int num_tensors,
                   float momentum, float max_norm, float inv_scale, int skip_nonfinite, void *workspace,
                   size_t workspace_bytes, float *ctl3, void *stream);
int brcnn_pack_conv_weights_batch(const float *const *weights, void *const *fwd, void *const *dgrad,
                                  const int *dims_host, const int *channels_last_host, int num, int dtype,
                                  const float *ctl3, void *stream);
/* The first FC of the box head (convfc_bbox_head.py:154-192 flattens the RoI features as (C, ph, pw)): master weight
 * (out_features, channels, positions) fp32 -> forward operand (out_features, positions, channels) in `dtype` -- the K
 * order of the NHWC RoI features -- and, when dgrad != NULL, the data-gradient operand (positions * channels,
 * out_features).  channels % 64 == 0, positions <= 255; ctl3 as above. */
int brcnn_pack_fc_weight_permuted(const float *weight, void *fwd, void *dgrad, int out_features, int channels,
                                  int positions, int dtype, const float *ctl3, void *stream);

#ifdef __cplusplus
}
#endif
#endif /* BRCNN_HIP_H */
