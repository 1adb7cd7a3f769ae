/*
 * brcnn_oracle.c -- TEST INFRASTRUCTURE ONLY.
 *
 * CPU restatement (plain C, scalar, single thread unless noted) of the native
 * operators the Boosting R-CNN hot path reaches through `mmcv.ops`.  Nothing
 * under oracle/ is imported by the product package; only tests/,
 * __graft_entry__.smoke() and bench.py's cpu_baseline leg load this library.
 *
 * The reference repo (/root/reference) holds no native code: the operators
 * live in the un-vendored dependency mmcv-full (pinned 1.3.8..1.4.0 by
 * mmdet/__init__.py:19-20, README.md:17 says 1.4.0).  Each function below
 * restates the published mmcv 1.4.0 CPU algorithm and cites the reference call
 * site that reaches it.  Pinning: the known-answer vectors in
 * tests/golden/kat_mmcv_ops.json (re-derived in float64 by
 * tests/golden/make_golden.py) -- mmdet's own tests hold no value test for
 * these ops, so beyond those vectors parity is "unpinned" against mmcv itself
 * and is anchored on the reference's call sites.
 *
 * Build: see oracle/Makefile (gcc -O2 -ffp-contract=off: no FMA contraction so
 * that the fp32 operation order below is exactly what is evaluated).
 */
#include <math.h>
#include <float.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define ORC_API __attribute__((visibility("default")))

/* ------------------------------------------------------------------------- */
/* RoIAlign (avg / max, aligned or legacy)                                    */
/* reference call sites: roi_extractors/base_roi_extractor.py:54-60 builds    */
/* mmcv.ops.RoIAlign(output_size=7, sampling_ratio=0, aligned=True default);  */
/* single_level_roi_extractor.py:103 calls it per pyramid level.              */
/* Algorithm: mmcv ops/csrc/pytorch/cpu/roi_align.cpp (ROIAlignForward with   */
/* pre_calc_for_bilinear_interpolate).                                        */
/* ------------------------------------------------------------------------- */

typedef struct {
    int pos1, pos2, pos3, pos4;
    float w1, w2, w3, w4;
} orc_precalc_t;

static void orc_precalc(int height, int width, int ph_n, int pw_n, int gh, int gw,
                        float roi_start_h, float roi_start_w, float bin_h, float bin_w,
                        orc_precalc_t *pc)
{
    int idx = 0;
    for (int ph = 0; ph < ph_n; ph++)
        for (int pw = 0; pw < pw_n; pw++)
            for (int iy = 0; iy < gh; iy++) {
                const float yy = roi_start_h + ph * bin_h +
                                 (float)(iy + .5f) * bin_h / (float)gh;
                for (int ix = 0; ix < gw; ix++) {
                    const float xx = roi_start_w + pw * bin_w +
                                     (float)(ix + .5f) * bin_w / (float)gw;
                    float x = xx, y = yy;
                    orc_precalc_t p;
                    if (y < -1.0f || y > (float)height || x < -1.0f || x > (float)width) {
                        p.pos1 = p.pos2 = p.pos3 = p.pos4 = 0;
                        p.w1 = p.w2 = p.w3 = p.w4 = 0.f;
                        pc[idx++] = p;
                        continue;
                    }
                    if (y <= 0) y = 0;
                    if (x <= 0) x = 0;
                    int y_low = (int)y, x_low = (int)x, y_high, x_high;
                    if (y_low >= height - 1) { y_high = y_low = height - 1; y = (float)y_low; }
                    else y_high = y_low + 1;
                    if (x_low >= width - 1) { x_high = x_low = width - 1; x = (float)x_low; }
                    else x_high = x_low + 1;
                    float ly = y - y_low, lx = x - x_low;
                    float hy = 1.f - ly, hx = 1.f - lx;
                    p.w1 = hy * hx; p.w2 = hy * lx; p.w3 = ly * hx; p.w4 = ly * lx;
                    p.pos1 = y_low * width + x_low;
                    p.pos2 = y_low * width + x_high;
                    p.pos3 = y_high * width + x_low;
                    p.pos4 = y_high * width + x_high;
                    pc[idx++] = p;
                }
            }
}

static void orc_roi_geom(const float *roi, float scale, int aligned, int ph_n, int pw_n,
                         int sampling_ratio, float *start_h, float *start_w,
                         float *bin_h, float *bin_w, int *gh, int *gw)
{
    float offset = aligned ? 0.5f : 0.0f;
    float roi_start_w = roi[1] * scale - offset;
    float roi_start_h = roi[2] * scale - offset;
    float roi_end_w = roi[3] * scale - offset;
    float roi_end_h = roi[4] * scale - offset;
    float roi_width = roi_end_w - roi_start_w;
    float roi_height = roi_end_h - roi_start_h;
    if (!aligned) {
        roi_width = roi_width > 1.f ? roi_width : 1.f;
        roi_height = roi_height > 1.f ? roi_height : 1.f;
    }
    *bin_h = roi_height / (float)ph_n;
    *bin_w = roi_width / (float)pw_n;
    *gh = (sampling_ratio > 0) ? sampling_ratio : (int)ceilf(roi_height / (float)ph_n);
    *gw = (sampling_ratio > 0) ? sampling_ratio : (int)ceilf(roi_width / (float)pw_n);
    if (*gh < 0) *gh = 0;
    if (*gw < 0) *gw = 0;
    *start_h = roi_start_h;
    *start_w = roi_start_w;
}

/* input (N,C,H,W) fp32 NCHW, rois (K,5) [batch,x1,y1,x2,y2], output (K,C,ph,pw).
 * pool_mode 0 = max (argmax_y/x written), 1 = avg.  Returns 0. */
ORC_API int orc_roi_align_forward(const float *input, const float *rois, float *output,
                                  float *argmax_y, float *argmax_x,
                                  int channels, int height, int width, int n_rois,
                                  int ph_n, int pw_n, float spatial_scale,
                                  int sampling_ratio, int pool_mode, int aligned)
{
    for (int n = 0; n < n_rois; n++) {
        const float *roi = rois + n * 5;
        int b = (int)roi[0];
        float sh, sw, bh, bw; int gh, gw;
        orc_roi_geom(roi, spatial_scale, aligned, ph_n, pw_n, sampling_ratio,
                     &sh, &sw, &bh, &bw, &gh, &gw);
        int cnt_i = gh * gw; if (cnt_i < 1) cnt_i = 1;
        const float count = (float)cnt_i;
        size_t npc = (size_t)gh * gw * ph_n * pw_n;
        orc_precalc_t *pc = (orc_precalc_t *)malloc((npc ? npc : 1) * sizeof(orc_precalc_t));
        orc_precalc(height, width, ph_n, pw_n, gh, gw, sh, sw, bh, bw, pc);
        for (int c = 0; c < channels; c++) {
            const float *in = input + ((size_t)b * channels + c) * height * width;
            size_t obase = ((size_t)n * channels + c) * ph_n * pw_n;
            int pi = 0;
            for (int ph = 0; ph < ph_n; ph++)
                for (int pw = 0; pw < pw_n; pw++) {
                    float out = 0.f, maxval = -10000.f, my = -1.f, mx = -1.f;
                    for (int iy = 0; iy < gh; iy++) {
                        const float y = sh + ph * bh + (float)(iy + .5f) * bh / (float)gh;
                        for (int ix = 0; ix < gw; ix++) {
                            const float x = sw + pw * bw + (float)(ix + .5f) * bw / (float)gw;
                            orc_precalc_t p = pc[pi++];
                            float val = p.w1 * in[p.pos1] + p.w2 * in[p.pos2] +
                                        p.w3 * in[p.pos3] + p.w4 * in[p.pos4];
                            if (val > maxval) { maxval = val; my = y; mx = x; }
                            out += val;
                        }
                    }
                    if (pool_mode == 0) {
                        output[obase + ph * pw_n + pw] = maxval;
                        if (argmax_y) argmax_y[obase + ph * pw_n + pw] = my;
                        if (argmax_x) argmax_x[obase + ph * pw_n + pw] = mx;
                    } else {
                        output[obase + ph * pw_n + pw] = out / count;
                    }
                }
        }
        free(pc);
    }
    return 0;
}

/* avg-mode backward: grad_input (N,C,H,W) must be zeroed by the caller (mmcv:
 * `grad_input = grad_output.new_zeros(ctx.input_shape)`); accumulates in roi
 * order.  Algorithm: mmcv cpu/roi_align.cpp ROIAlignBackward. */
ORC_API int orc_roi_align_backward(const float *grad_output, const float *rois,
                                   float *grad_input, int channels, int height, int width,
                                   int n_rois, int ph_n, int pw_n, float spatial_scale,
                                   int sampling_ratio, int aligned)
{
    for (int n = 0; n < n_rois; n++) {
        const float *roi = rois + n * 5;
        int b = (int)roi[0];
        float sh, sw, bh, bw; int gh, gw;
        orc_roi_geom(roi, spatial_scale, aligned, ph_n, pw_n, sampling_ratio,
                     &sh, &sw, &bh, &bw, &gh, &gw);
        int cnt_i = gh * gw; if (cnt_i < 1) cnt_i = 1;
        const float count = (float)cnt_i;
        size_t npc = (size_t)gh * gw * ph_n * pw_n;
        orc_precalc_t *pc = (orc_precalc_t *)malloc((npc ? npc : 1) * sizeof(orc_precalc_t));
        orc_precalc(height, width, ph_n, pw_n, gh, gw, sh, sw, bh, bw, pc);
        for (int c = 0; c < channels; c++) {
            float *gi = grad_input + ((size_t)b * channels + c) * height * width;
            size_t obase = ((size_t)n * channels + c) * ph_n * pw_n;
            int pi = 0;
            for (int ph = 0; ph < ph_n; ph++)
                for (int pw = 0; pw < pw_n; pw++) {
                    const float g = grad_output[obase + ph * pw_n + pw];
                    for (int iy = 0; iy < gh; iy++)
                        for (int ix = 0; ix < gw; ix++) {
                            orc_precalc_t p = pc[pi++];
                            /* the reference skips samples outside the map (x_low<0) */
                            if (p.w1 == 0.f && p.w2 == 0.f && p.w3 == 0.f && p.w4 == 0.f) continue;
                            gi[p.pos1] += g * p.w1 / count;
                            gi[p.pos2] += g * p.w2 / count;
                            gi[p.pos3] += g * p.w3 / count;
                            gi[p.pos4] += g * p.w4 / count;
                        }
                }
        }
        free(pc);
    }
    return 0;
}

/* ------------------------------------------------------------------------- */
/* NMS.  reference call sites: atss_rpn_head.py:756, rpn_head.py:245,         */
/* core/post_processing/bbox_nms.py:86 (all through mmcv batched_nms -> nms). */
/* Algorithm: mmcv cpu/nms.cpp nms_cpu: order by score descending, greedy     */
/* suppress `inter / (area_i + area_j - inter) > thr` (strict).               */
/* Tie rule (the reference's sort is unspecified on ties): descending score,  */
/* ascending original index -- shared with the HIP kernel.                    */
/* ------------------------------------------------------------------------- */

typedef struct { float s; int64_t i; } orc_si_t;

static int orc_cmp_desc(const void *a, const void *b)
{
    const orc_si_t *x = (const orc_si_t *)a, *y = (const orc_si_t *)b;
    if (x->s > y->s) return -1;
    if (x->s < y->s) return 1;
    return (x->i < y->i) ? -1 : (x->i > y->i);
}

/* writes the sorted permutation (stable, descending) into order[n] */
ORC_API int orc_argsort_desc(const float *scores, int64_t n, int64_t *order)
{
    orc_si_t *t = (orc_si_t *)malloc((n ? n : 1) * sizeof(orc_si_t));
    for (int64_t i = 0; i < n; i++) { t[i].s = scores[i]; t[i].i = i; }
    qsort(t, n, sizeof(orc_si_t), orc_cmp_desc);
    for (int64_t i = 0; i < n; i++) order[i] = t[i].i;
    free(t);
    return 0;
}

/* boxes (n,4), scores (n); keep[n] receives kept ORIGINAL indices in score order;
 * returns the number kept. */
ORC_API int64_t orc_nms(const float *boxes, const float *scores, int64_t n,
                        float iou_threshold, int offset, int64_t *keep)
{
    if (n == 0) return 0;
    int64_t *order = (int64_t *)malloc(n * sizeof(int64_t));
    float *areas = (float *)malloc(n * sizeof(float));
    unsigned char *sel = (unsigned char *)malloc(n);
    orc_argsort_desc(scores, n, order);
    for (int64_t i = 0; i < n; i++) {
        const float *b = boxes + i * 4;
        areas[i] = (b[2] - b[0] + offset) * (b[3] - b[1] + offset);
        sel[i] = 1;
    }
    for (int64_t _i = 0; _i < n; _i++) {
        if (!sel[_i]) continue;
        int64_t i = order[_i];
        float ix1 = boxes[i * 4], iy1 = boxes[i * 4 + 1], ix2 = boxes[i * 4 + 2],
              iy2 = boxes[i * 4 + 3], iarea = areas[i];
        for (int64_t _j = _i + 1; _j < n; _j++) {
            if (!sel[_j]) continue;
            int64_t j = order[_j];
            float xx1 = fmaxf(ix1, boxes[j * 4]), yy1 = fmaxf(iy1, boxes[j * 4 + 1]);
            float xx2 = fminf(ix2, boxes[j * 4 + 2]), yy2 = fminf(iy2, boxes[j * 4 + 3]);
            float w = fmaxf(0.f, xx2 - xx1 + offset), h = fmaxf(0.f, yy2 - yy1 + offset);
            float inter = w * h;
            float ovr = inter / (iarea + areas[j] - inter);
            if (ovr > iou_threshold) sel[_j] = 0;
        }
    }
    int64_t k = 0;
    for (int64_t _i = 0; _i < n; _i++) if (sel[_i]) keep[k++] = order[_i];
    free(order); free(areas); free(sel);
    return k;
}

/* ------------------------------------------------------------------------- */
/* Soft-NMS.  reference call sites: nms=dict(type='soft_nms', ...) in         */
/* configs/boosting_rcnn/boosting_rcnn_r2_101_dcn_pafpn_mstrain_3x_coco.py:27 */
/* -> bbox_nms.py:86 -> mmcv batched_nms -> soft_nms.                         */
/* Algorithm: mmcv cpu/nms.cpp softnms_cpu (sequential pick-max / swap /      */
/* decay / swap-with-last discard).  method 0 naive, 1 linear, 2 gaussian.    */
/* dets (n,5) receives [x1,y1,x2,y2,decayed score] in pick order; inds[n]     */
/* the original indices; returns the number of survivors.                     */
/* ------------------------------------------------------------------------- */
ORC_API int64_t orc_softnms(const float *boxes, const float *scores, int64_t n, float *dets,
                            int64_t *inds, float iou_threshold, float sigma, float min_score,
                            int method, int offset)
{
    if (n == 0) return 0;
    float *x1 = (float *)malloc(n * 4), *y1 = (float *)malloc(n * 4), *x2 = (float *)malloc(n * 4),
          *y2 = (float *)malloc(n * 4), *sc = (float *)malloc(n * 4), *ar = (float *)malloc(n * 4);
    for (int64_t i = 0; i < n; i++) {
        x1[i] = boxes[i * 4]; y1[i] = boxes[i * 4 + 1]; x2[i] = boxes[i * 4 + 2]; y2[i] = boxes[i * 4 + 3];
        sc[i] = scores[i];
        ar[i] = (x2[i] - x1[i] + offset) * (y2[i] - y1[i] + offset);
        inds[i] = i;
    }
    int64_t nboxes = n;
    for (int64_t i = 0; i < nboxes; i++) {
        float max_score = sc[i];
        int64_t max_pos = i, pos = i + 1;
        while (pos < nboxes) {
            if (max_score < sc[pos]) { max_score = sc[pos]; max_pos = pos; }
            pos++;
        }
        float ix1 = dets[i * 5 + 0] = x1[max_pos];
        float iy1 = dets[i * 5 + 1] = y1[max_pos];
        float ix2 = dets[i * 5 + 2] = x2[max_pos];
        float iy2 = dets[i * 5 + 3] = y2[max_pos];
        float iscore = dets[i * 5 + 4] = sc[max_pos];
        float iarea = ar[max_pos];
        int64_t iind = inds[max_pos];
        x1[max_pos] = x1[i]; y1[max_pos] = y1[i]; x2[max_pos] = x2[i]; y2[max_pos] = y2[i];
        sc[max_pos] = sc[i]; ar[max_pos] = ar[i]; inds[max_pos] = inds[i];
        x1[i] = ix1; y1[i] = iy1; x2[i] = ix2; y2[i] = iy2; sc[i] = iscore; ar[i] = iarea; inds[i] = iind;

        pos = i + 1;
        while (pos < nboxes) {
            float xx1 = fmaxf(ix1, x1[pos]), yy1 = fmaxf(iy1, y1[pos]);
            float xx2 = fminf(ix2, x2[pos]), yy2 = fminf(iy2, y2[pos]);
            float w = fmaxf(0.f, xx2 - xx1 + offset), h = fmaxf(0.f, yy2 - yy1 + offset);
            float inter = w * h;
            float ovr = inter / (iarea + ar[pos] - inter);
            float weight = 1.f;
            if (method == 0) { if (ovr >= iou_threshold) weight = 0.f; }
            else if (method == 1) { if (ovr >= iou_threshold) weight = 1.f - ovr; }
            else if (method == 2) { weight = expf(-(ovr * ovr) / sigma); }
            sc[pos] *= weight;
            if (sc[pos] < min_score) {
                x1[pos] = x1[nboxes - 1]; y1[pos] = y1[nboxes - 1];
                x2[pos] = x2[nboxes - 1]; y2[pos] = y2[nboxes - 1];
                sc[pos] = sc[nboxes - 1]; ar[pos] = ar[nboxes - 1]; inds[pos] = inds[nboxes - 1];
                nboxes--; pos--;
            }
            pos++;
        }
    }
    free(x1); free(y1); free(x2); free(y2); free(sc); free(ar);
    return nboxes;
}

/* ------------------------------------------------------------------------- */
/* Sigmoid focal loss (per element, reduction 'none').                        */
/* reference call site: losses/focal_loss.py:86 (CUDA path, mmcv op) and the  */
/* python fallback py_sigmoid_focal_loss focal_loss.py:12-57 (CPU path).      */
/* Algorithm: mmcv sigmoid_focal_loss_cuda_kernel.cuh (closed form; equal to  */
/* the python form up to fp32 rounding).  target[n] in [0, C]; C = background.*/
/* ------------------------------------------------------------------------- */
ORC_API int orc_sigmoid_focal_loss_forward(const float *input, const int64_t *target,
                                           const float *weight, float *output, int64_t n,
                                           int64_t c_n, float gamma, float alpha)
{
    for (int64_t idx = 0; idx < n * c_n; idx++) {
        int64_t i = idx / c_n, c = idx % c_n, t = target[i];
        float flag_p = (t == c), flag_n = (t != c);
        float p = 1.f / (1.f + expf(-input[idx]));
        float term_p = powf(1.f - p, gamma) * logf(fmaxf(p, FLT_MIN));
        float term_n = powf(p, gamma) * logf(fmaxf(1.f - p, FLT_MIN));
        float o = 0.f;
        o += -flag_p * alpha * term_p;
        o += -flag_n * (1.f - alpha) * term_n;
        if (weight) o *= weight[t];
        output[idx] = o;
    }
    return 0;
}

ORC_API int orc_sigmoid_focal_loss_backward(const float *input, const int64_t *target,
                                            const float *weight, float *grad_input, int64_t n,
                                            int64_t c_n, float gamma, float alpha)
{
    for (int64_t idx = 0; idx < n * c_n; idx++) {
        int64_t i = idx / c_n, c = idx % c_n, t = target[i];
        float flag_p = (t == c), flag_n = (t != c);
        float p = 1.f / (1.f + expf(-input[idx]));
        float term_p = powf(1.f - p, gamma) * (1.f - p - gamma * p * logf(fmaxf(p, FLT_MIN)));
        float term_n = powf(p, gamma) * (gamma * (1.f - p) * logf(fmaxf(1.f - p, FLT_MIN)) - p);
        float g = 0.f;
        g += -flag_p * alpha * term_p;
        g += -flag_n * (1.f - alpha) * term_n;
        if (weight) g *= weight[t];
        grad_input[idx] = g;
    }
    return 0;
}

/* ------------------------------------------------------------------------- */
/* OpenMP-free multi-RoI helper used only by bench.py's cpu_baseline: the     */
/* same forward split over [k0,k1) so that Python threads can shard RoIs.     */
/* ------------------------------------------------------------------------- */
ORC_API int orc_roi_align_forward_range(const float *input, const float *rois, float *output,
                                        int channels, int height, int width, int k0, int k1,
                                        int ph_n, int pw_n, float spatial_scale,
                                        int sampling_ratio, int aligned)
{
    return orc_roi_align_forward(input, rois + (size_t)k0 * 5,
                                 output + (size_t)k0 * channels * ph_n * pw_n, 0, 0, channels,
                                 height, width, k1 - k0, ph_n, pw_n, spatial_scale,
                                 sampling_ratio, 1, aligned);
}

/* ------------------------------------------------------------------------- */
/* Input front door: Resize -> RandomFlip -> Normalize -> Pad of one decoded  */
/* uint8 BGR image (mmdet/datasets/pipelines/transforms.py:31-315, 318-470,   */
/* 700-739, 625-697).  The image arithmetic lives in mmcv 1.4.0 -> OpenCV     */
/* (both absent): cv::resize INTER_LINEAR on CV_8U (imgproc/resize.cpp:       */
/* resizeGeneric_, HResizeLinear<uchar,int,short>, VResizeLinear<uchar,int,   */
/* short,FixedPtCast<int,uchar,22>> with INTER_RESIZE_COEF_BITS = 11),        */
/* mmcv.imnormalize (fp32 subtract / multiply by fp32(1/std)).                */
/* PARITY UNPINNED AGAINST cv2 for the resize: no cv2 and no reference golden */
/* vector of it here; restated from the published algorithm, cross-checked    */
/* against the independent numpy restatement in the package and pinned only   */
/* by paper cases (tests/golden/kat_mmcv_ops.json resize_hand /               */
/* preprocess_hand: integer and non-integer scale factors, 1-pixel sources,   */
/* round-half-up, flips, BGR->RGB with RGB-ordered mean / std, padding).      */
/* ------------------------------------------------------------------------- */
static void orc_axis_coeff(int d, double scale, int src, int *s, int *c0, int *c1)
{
    float f = (float)(((double)d + 0.5) * scale - 0.5);
    int si = (int)floorf(f);
    f -= (float)si;
    if (si < 0) { f = 0.f; si = 0; }
    if (si >= src - 1) { f = 0.f; si = src - 1; }
    *s = si;
    *c0 = (int)rintf((1.f - f) * 2048.f);
    *c1 = (int)rintf(f * 2048.f);
}

ORC_API int orc_preprocess_u8(const uint8_t *src, int sh, int sw, float *dst, int nh, int nw,
                              int ph, int pw, int flip, const float *mean3, const float *std3,
                              int to_rgb)
{
    if (!src || !dst || sh <= 0 || sw <= 0 || nh <= 0 || nw <= 0 || ph < nh || pw < nw) return -22;
    const double scale_x = 1.0 / ((double)nw / (double)sw);
    const double scale_y = 1.0 / ((double)nh / (double)sh);
    float stdinv[3];
    for (int c = 0; c < 3; c++) stdinv[c] = (float)(1.0 / (double)std3[c]);
    const size_t plane = (size_t)ph * pw;
    for (int y = 0; y < ph; y++)
        for (int x = 0; x < pw; x++) {
            float *o = dst + (size_t)y * pw + x;
            if (x >= nw || y >= nh) { o[0] = o[plane] = o[2 * plane] = 0.f; continue; }
            const int rx = (flip & 1) ? nw - 1 - x : x;
            const int ry = (flip & 2) ? nh - 1 - y : y;
            int sx, a0, a1, sy, b0, b1;
            orc_axis_coeff(rx, scale_x, sw, &sx, &a0, &a1);
            orc_axis_coeff(ry, scale_y, sh, &sy, &b0, &b1);
            const int sx1 = sx + 1 < sw ? sx + 1 : sw - 1;
            const int sy1 = sy + 1 < sh ? sy + 1 : sh - 1;
            const uint8_t *r0 = src + (size_t)sy * sw * 3, *r1 = src + (size_t)sy1 * sw * 3;
            for (int c = 0; c < 3; c++) {
                const int sc = to_rgb ? 2 - c : c;
                const int h0 = (int)r0[sx * 3 + sc] * a0 + (int)r0[sx1 * 3 + sc] * a1;
                const int h1 = (int)r1[sx * 3 + sc] * a0 + (int)r1[sx1 * 3 + sc] * a1;
                int v = (((b0 * (h0 >> 4)) >> 16) + ((b1 * (h1 >> 4)) >> 16) + 2) >> 2;
                if (v < 0) v = 0;
                if (v > 255) v = 255;
                o[c * plane] = ((float)v - mean3[c]) * stdinv[c];
            }
        }
    return 0;
}
