// Fused losses of the train step, forward and backward, for gfx950.
//
//   rpn_loss_*    ATSSRPNHead.loss / loss_single (mmdet/models/dense_heads/atss_rpn_head.py:299-464)
//                 for the recipe family reg_decoded_bbox=True + IoULoss(log) + MSELoss aug + FocalLoss:
//                 per anchor, straight from the fused (cls | reg | iou) head output of all pyramid
//                 levels and the assignment (`gt_inds`): sigmoid focal loss over every non-ignored
//                 anchor (losses/focal_loss.py:137-182 -> mmcv sigmoid_focal_loss); for positives the
//                 anchor is regenerated (anchor_generator.py:336-381), the scaled deltas decoded
//                 (delta_xywh_bbox_coder.py:145-272, no border clip), iou_target = IoU(decoded, gt)
//                 (detached), -log(IoU) * clamp(iou_target^g, 1e-12) (losses/iou_loss.py:14-50,
//                 491-534), MSE(deltas, encode(anchor, gt)) * the same weight (losses/mse_loss.py),
//                 BCE-with-logits(iou_pred, iou_target) (losses/cross_entropy_loss.py:61-113).
//                 Normalisers (num_total_samples, sum of iou_target; :440-460) stay on the device:
//                 forward = per-block partial sums -> fixed-order reduction (deterministic) ->
//                 [optional all-reduce of the 2 normalisers by the caller] -> finalize.
//   boost_loss_*  ProbRoIHead._bbox_forward_train_boost + norm_loss (roi_heads/prob_roi_head.py:107-154)
//                 over ProbConvFCBBoxHead.loss (bbox_heads/convfc_bbox_head.py:332-418):
//                 L_i = lw * CE_i (softmax), w_i = (1 - prior_i)^g [* alpha] [* |iou_i - p_label|^ig],
//                 loss_cls = sum_i L_i * (w_i * sum L / sum(w L)) / N  (the factor is detached),
//                 loss_bbox = lw_b * sum_pos |pred[label] - target| / N  ('bbox_num') or / (4 n_pos)
//                 ('mean'), acc = top-1 %; backward: dcls = (softmax - onehot) * lw * w_i * c / N.
// HBM streams: rpn forward reads 54 floats per pixel once, backward writes them once.
#include "common.h"
#include <float.h>

namespace {

struct GtTable {
    int off[BRCNN_MAX_IMAGES + 1];
};

struct RpnLevels {
    int num;
    int row0[BRCNN_MAX_LEVELS + 1];     // first row of each level in the concatenated head output
    int hw[BRCNN_MAX_LEVELS], width[BRCNN_MAX_LEVELS];
    int stride_w[BRCNN_MAX_LEVELS], stride_h[BRCNN_MAX_LEVELS];
    int start[BRCNN_MAX_LEVELS + 1];    // first anchor of each level in the per-image order
    int blk0[BRCNN_MAX_LEVELS + 1];     // first workgroup of each level
    const float* base[BRCNN_MAX_LEVELS];// (A, 4) base anchors
};

struct RpnLossParams {
    const float* y;            // (rows, ystride): [cls A | reg 4A | iou A | pad]
    int ystride, A, batch, anchors_per_image;
    const float* scales;       // (L) the learnable per-level Scale of rpn_reg (device)
    RpnLevels lv;
    const int* gt_inds;        // (B, anchors_per_image)
    const float* gts;
    GtTable gt;
    float focal_gamma, focal_alpha, pos_weight;
    float iou_gamma;           // ATSSRPNHead.gamma
    float mean[4], std[4], max_ratio;
    int with_aug;
    float lw_cls, lw_bbox, lw_aug, lw_iou;
    int cls_mode;              // 0 FocalLoss, 1 VarifocalLoss (iou_weighted), 2 VarifocalLoss (not iou_weighted)
    int reg_mode;              // 0 decoded boxes: -log(IoU) [+ MSE aug], 1 reg_decoded_bbox=False: CIoU(deltas, encoded targets)
};

enum { S_FOCAL = 0, S_IOU, S_MSE, S_BCE, S_IOUT, S_NPOS, S_N };

__device__ __forceinline__ float sigmoidf_(float x) { return 1.f / (1.f + expf(-x)); }

__device__ __forceinline__ float focal_fwd(float x, bool is_pos, float gamma, float alpha) {
    const float p = sigmoidf_(x);
    return is_pos ? -alpha * powf(1.f - p, gamma) * logf(fmaxf(p, FLT_MIN))
                  : -(1.f - alpha) * powf(p, gamma) * logf(fmaxf(1.f - p, FLT_MIN));
}

__device__ __forceinline__ float focal_bwd(float x, bool is_pos, float gamma, float alpha) {
    const float p = sigmoidf_(x);
    return is_pos ? -alpha * powf(1.f - p, gamma) * (1.f - p - gamma * p * logf(fmaxf(p, FLT_MIN)))
                  : -(1.f - alpha) * powf(p, gamma) * (gamma * (1.f - p) * logf(fmaxf(1.f - p, FLT_MIN)) - p);
}

__device__ __forceinline__ float pow_gamma(float x, float g) {      // tensor ** python float, as torch lowers it
    if (g == 0.5f) return sqrtf(x);
    if (g == 1.f) return x;
    if (g == 2.f) return x * x;
    return powf(x, g);
}

// VarifocalLoss element (losses/varifocal_loss.py:10-58): BCE-with-logits(x, t) * (t [or 1] for t > 0,
// alpha * |sigmoid(x) - t|^gamma for t <= 0); t is detached
__device__ __forceinline__ float bce_logits(float x, float t) { return fmaxf(x, 0.f) - x * t + log1pf(expf(-fabsf(x))); }
__device__ __forceinline__ float vfl_fwd(float x, float t, float gamma, float alpha, bool iou_weighted) {
    if (t > 0.f) return bce_logits(x, t) * (iou_weighted ? t : 1.f);
    const float s = sigmoidf_(x);
    return bce_logits(x, t) * (alpha * powf(fabsf(s - t), gamma));
}
__device__ __forceinline__ float vfl_bwd(float x, float t, float gamma, float alpha, bool iou_weighted) {
    const float s = sigmoidf_(x);
    if (t > 0.f) return (s - t) * (iou_weighted ? t : 1.f);
    const float u = s - t, au = fabsf(u);
    const float fw = alpha * powf(au, gamma);
    const float dfw = au > 0.f ? alpha * gamma * powf(au, gamma - 1.f) * (u > 0.f ? 1.f : -1.f) * s * (1.f - s) : 0.f;
    return (s - t) * fw + bce_logits(x, t) * dfw;
}

// Complete-IoU loss of (p, q) as the reference applies it to RAW deltas (reg_decoded_bbox=False with
// loss_bbox=CIoULoss: atss_rpn_head.py:361-374, losses/iou_loss.py:175-236, eps 1e-6); `grad` (optional)
// receives d loss / d p with torch's conventions (ties of max / min split evenly, clamp passes at the bound)
__device__ __forceinline__ float ciou_terms(const float4 p, const float4 q, float eps, float* grad) {
    const float ltx = fmaxf(p.x, q.x), lty = fmaxf(p.y, q.y), rbx = fminf(p.z, q.z), rby = fminf(p.w, q.w);
    const float wr = rbx - ltx, hr = rby - lty;
    const float w = wr < 0.f ? 0.f : wr, h = hr < 0.f ? 0.f : hr;
    const float ov = w * h;
    const float bw = p.z - p.x, bh = p.w - p.y;
    const float ap = bw * bh, ag = (q.z - q.x) * (q.w - q.y);
    const float un = ap + ag - ov + eps;
    const float iou = ov / un;
    const float cwr = fmaxf(p.z, q.z) - fminf(p.x, q.x), chr = fmaxf(p.w, q.w) - fminf(p.y, q.y);
    const float cw = cwr < 0.f ? 0.f : cwr, ch = chr < 0.f ? 0.f : chr;
    const float c2 = cw * cw + ch * ch + eps;
    const float w1 = bw, h1 = bh + eps, w2 = q.z - q.x, h2 = q.w - q.y + eps;
    const float sx = (q.x + q.z) - (p.x + p.z), sy = (q.y + q.w) - (p.y + p.w);
    const float rho2 = sx * sx / 4.f + sy * sy / 4.f;
    const float F = 0.40528473456935116f;            // 4 / pi^2
    const float r1 = w1 / h1;
    const float at = atanf(w2 / h2) - atanf(r1);
    const float v = F * (at * at);
    const float alpha = iou > 0.5f ? v / (1.f - iou + v) : 0.f;
    const float ci = iou - (rho2 / c2 + alpha * v);
    const float loss = 1.f - fminf(fmaxf(ci, -1.f), 1.f);
    if (grad) {
        float gx1 = 0.f, gy1 = 0.f, gx2 = 0.f, gy2 = 0.f;
        const float g_ci = (ci >= -1.f && ci <= 1.f) ? -1.f : 0.f;
        const float g_rho2 = -g_ci / c2, g_c2 = g_ci * rho2 / (c2 * c2), g_v = -g_ci * alpha;
        const float g_ov = g_ci / un, g_un = -g_ci * ov / (un * un);
        const float g_ovt = g_ov - g_un;
        const float g_w = wr >= 0.f ? g_ovt * h : 0.f, g_h = hr >= 0.f ? g_ovt * w : 0.f;
        const float kx1 = p.x > q.x ? 1.f : (p.x == q.x ? 0.5f : 0.f), ky1 = p.y > q.y ? 1.f : (p.y == q.y ? 0.5f : 0.f);
        const float kx2 = p.z < q.z ? 1.f : (p.z == q.z ? 0.5f : 0.f), ky2 = p.w < q.w ? 1.f : (p.w == q.w ? 0.5f : 0.f);
        gx1 += -g_w * kx1 - g_un * bh;
        gy1 += -g_h * ky1 - g_un * bw;
        gx2 += g_w * kx2 + g_un * bh;
        gy2 += g_h * ky2 + g_un * bw;
        const float g_cw = cwr >= 0.f ? g_c2 * 2.f * cw : 0.f, g_ch = chr >= 0.f ? g_c2 * 2.f * ch : 0.f;
        const float mx1 = p.x < q.x ? 1.f : (p.x == q.x ? 0.5f : 0.f), my1 = p.y < q.y ? 1.f : (p.y == q.y ? 0.5f : 0.f);
        const float mx2 = p.z > q.z ? 1.f : (p.z == q.z ? 0.5f : 0.f), my2 = p.w > q.w ? 1.f : (p.w == q.w ? 0.5f : 0.f);
        gx1 += -g_cw * mx1; gx2 += g_cw * mx2;
        gy1 += -g_ch * my1; gy2 += g_ch * my2;
        const float grx = g_rho2 * (-sx * 0.5f), gry = g_rho2 * (-sy * 0.5f);
        gx1 += grx; gx2 += grx; gy1 += gry; gy2 += gry;
        const float g_r1 = -(g_v * F * 2.f * at) / (r1 * r1 + 1.f);
        const float g_w1 = g_r1 / h1, g_h1 = -g_r1 * w1 / (h1 * h1);
        gx2 += g_w1; gx1 -= g_w1; gy2 += g_h1; gy1 -= g_h1;
        grad[0] = gx1; grad[1] = gy1; grad[2] = gx2; grad[3] = gy2;
    }
    return loss;
}

struct PosTerms {
    float4 d;        // scaled deltas (what the loss sees: Scale(rpn_reg(x)))
    float4 raw;      // raw head output
    float4 enc;      // encode(anchor, gt)
    float4 box;      // decoded prediction
    float4 gt;
    float pw, ph, gw, gh;       // anchor size, decoded size
    bool clamp_w, clamp_h;      // dw / dh outside +-max_ratio (no gradient)
    float iou, w;               // iou_target, clamp(iou_target^g, 1e-12)
};

__device__ __forceinline__ void pos_terms(const RpnLossParams& p, int l, int cell, int a, const float* yrow,
                                          float scale, const float4 g, PosTerms& t) {
    const int cx = cell % p.lv.width[l], cy = cell / p.lv.width[l];
    const float sx = (float)(cx * p.lv.stride_w[l]), sy = (float)(cy * p.lv.stride_h[l]);
    const float4 ba = *reinterpret_cast<const float4*>(p.lv.base[l] + a * 4);
    const float ax1 = ba.x + sx, ay1 = ba.y + sy, ax2 = ba.z + sx, ay2 = ba.w + sy;
    const float* r = yrow + p.A + a * 4;
    t.raw = make_float4(r[0], r[1], r[2], r[3]);
    t.d = make_float4(t.raw.x * scale, t.raw.y * scale, t.raw.z * scale, t.raw.w * scale);
    const float dx = t.d.x * p.std[0] + p.mean[0], dy = t.d.y * p.std[1] + p.mean[1];
    float dw = t.d.z * p.std[2] + p.mean[2], dh = t.d.w * p.std[3] + p.mean[3];
    const float px = (ax1 + ax2) * 0.5f, py = (ay1 + ay2) * 0.5f;
    t.pw = ax2 - ax1; t.ph = ay2 - ay1;
    t.clamp_w = dw < -p.max_ratio || dw > p.max_ratio;
    t.clamp_h = dh < -p.max_ratio || dh > p.max_ratio;
    dw = fminf(fmaxf(dw, -p.max_ratio), p.max_ratio);
    dh = fminf(fmaxf(dh, -p.max_ratio), p.max_ratio);
    t.gw = t.pw * expf(dw); t.gh = t.ph * expf(dh);
    const float gx = px + t.pw * dx, gy = py + t.ph * dy;
    t.box = make_float4(gx - t.gw * 0.5f, gy - t.gh * 0.5f, gx + t.gw * 0.5f, gy + t.gh * 0.5f);
    t.gt = g;
    // encode(anchor, gt): bbox2delta (delta_xywh_bbox_coder.py:99-141)
    const float ggx = (g.x + g.z) * 0.5f, ggy = (g.y + g.w) * 0.5f, ggw = g.z - g.x, ggh = g.w - g.y;
    t.enc.x = ((ggx - px) / t.pw - p.mean[0]) / p.std[0];
    t.enc.y = ((ggy - py) / t.ph - p.mean[1]) / p.std[1];
    t.enc.z = (logf(ggw / t.pw) - p.mean[2]) / p.std[2];
    t.enc.w = (logf(ggh / t.ph) - p.mean[3]) / p.std[3];
    float4 q = g;
    if (p.reg_mode == 1) {
        // reg_decoded_bbox=False: the target box of iou_target is decode(anchor, encoded target) (:362-367)
        const float ex = t.enc.x * p.std[0] + p.mean[0], ey = t.enc.y * p.std[1] + p.mean[1];
        const float ew = fminf(fmaxf(t.enc.z * p.std[2] + p.mean[2], -p.max_ratio), p.max_ratio);
        const float eh = fminf(fmaxf(t.enc.w * p.std[3] + p.mean[3], -p.max_ratio), p.max_ratio);
        const float qw = t.pw * expf(ew), qh = t.ph * expf(eh), qx = px + t.pw * ex, qy = py + t.ph * ey;
        q = make_float4(qx - qw * 0.5f, qy - qh * 0.5f, qx + qw * 0.5f, qy + qh * 0.5f);
    }
    // iou_target = bbox_overlaps(pred, target box, is_aligned=True)
    const float a1 = (t.box.z - t.box.x) * (t.box.w - t.box.y), a2 = (q.z - q.x) * (q.w - q.y);
    float w = fminf(t.box.z, q.z) - fmaxf(t.box.x, q.x), h = fminf(t.box.w, q.w) - fmaxf(t.box.y, q.y);
    w = w < 0.f ? 0.f : w;
    h = h < 0.f ? 0.f : h;
    const float ov = w * h;
    const float un = fmaxf(a1 + a2 - ov, 1e-6f);
    t.iou = ov / un;
    t.w = fmaxf(pow_gamma(t.iou, p.iou_gamma), 1e-12f);
}

__device__ __forceinline__ void locate(const RpnLossParams& p, int& l, long long& e) {
    l = 0;
#pragma unroll
    for (int k = 1; k < BRCNN_MAX_LEVELS; k++)
        if (k < p.lv.num && (int)blockIdx.x >= p.lv.blk0[k]) l = k;
    e = (long long)(blockIdx.x - p.lv.blk0[l]) * 256 + threadIdx.x;
}

__device__ __forceinline__ float block_sum(float v, float* s_red) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) v += __shfl_xor(v, d, 64);
    __syncthreads();
    if (lane == 0) s_red[wave] = v;
    __syncthreads();
    return s_red[0] + s_red[1] + s_red[2] + s_red[3];
}

__global__ __launch_bounds__(256) void rpn_loss_fwd_kernel(const RpnLossParams p, float* __restrict__ partials) {
    __shared__ float s_red[4];
    int l;
    long long e;
    locate(p, l, e);
    const long long n_l = (long long)p.batch * p.lv.hw[l] * p.A;
    float acc[S_N] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    if (e < n_l) {
        const int a = (int)(e % p.A);
        const long long row = e / p.A;
        const int b = (int)(row / p.lv.hw[l]), cell = (int)(row - (long long)b * p.lv.hw[l]);
        const float* yrow = p.y + (size_t)(p.lv.row0[l] + row) * p.ystride;
        const int gi = p.gt_inds[(size_t)b * p.anchors_per_image + p.lv.start[l] + cell * p.A + a];
        const bool pos = gi > 0;
        float iou_t = 0.f;
        if (pos) {
            PosTerms t;
            const float4 g = *reinterpret_cast<const float4*>(p.gts + (size_t)(p.gt.off[b] + gi - 1) * 4);
            pos_terms(p, l, cell, a, yrow, p.scales[l], g, t);
            iou_t = t.iou;
            if (p.reg_mode == 1) {
                acc[S_IOU] = ciou_terms(t.d, t.enc, 1e-6f, nullptr) * t.w;
            } else {
                acc[S_IOU] = -logf(fmaxf(t.iou, 1e-6f)) * t.w;
                const float e0 = t.d.x - t.enc.x, e1 = t.d.y - t.enc.y, e2 = t.d.z - t.enc.z, e3 = t.d.w - t.enc.w;
                acc[S_MSE] = p.with_aug ? (e0 * e0 * t.w + e1 * e1 * t.w + e2 * e2 * t.w + e3 * e3 * t.w) : 0.f;
            }
            const float x = yrow[5 * p.A + a];
            acc[S_BCE] = bce_logits(x, t.iou);
            acc[S_IOUT] = t.iou;
            acc[S_NPOS] = 1.f;
        }
        if (p.cls_mode) {
            // VarifocalLoss is called without label weights (:393-397): anchors outside the image count as negatives
            acc[S_FOCAL] = vfl_fwd(yrow[a], iou_t, p.focal_gamma, p.focal_alpha, p.cls_mode == 1);
        } else if (gi >= 0) {
            float f = focal_fwd(yrow[a], pos, p.focal_gamma, p.focal_alpha);
            if (pos && p.pos_weight > 0.f) f *= p.pos_weight;
            acc[S_FOCAL] = f;
        }
    }
#pragma unroll
    for (int k = 0; k < S_N; k++) {
        const float s = block_sum(acc[k], s_red);
        if (threadIdx.x == 0) partials[(size_t)blockIdx.x * S_N + k] = s;
    }
}

// fixed-order reduction of the per-block partial sums: sums (L, S_N) and totals [num_pos, sum iou_target]
__global__ __launch_bounds__(256) void rpn_loss_reduce_kernel(const float* __restrict__ partials, RpnLevels lv,
                                                             float* __restrict__ sums, float* __restrict__ totals) {
    __shared__ float s_red[4];
    __shared__ float s_tot[2];
    if (threadIdx.x < 2) s_tot[threadIdx.x] = 0.f;
    for (int l = 0; l < lv.num; l++) {
        for (int k = 0; k < S_N; k++) {
            float v = 0.f;
            for (int blk = lv.blk0[l] + threadIdx.x; blk < lv.blk0[l + 1]; blk += 256) v += partials[(size_t)blk * S_N + k];
            const float s = block_sum(v, s_red);
            if (threadIdx.x == 0) {
                sums[l * S_N + k] = s;
                if (k == S_NPOS) s_tot[0] += s;
                if (k == S_IOUT) s_tot[1] += s;
            }
        }
    }
    __syncthreads();
    if (threadIdx.x < 2) totals[threadIdx.x] = s_tot[threadIdx.x];
}

// losses3 = [loss_cls, loss_bbox, loss_iou] summed over levels, per_level (3, L), coef = [1/nts, 1/baf]
// totals hold the rank MEANS of (num_pos, sum iou_target) (atss_rpn_head.py:440-444,458-460)
__global__ void rpn_loss_finalize_kernel(const float* __restrict__ sums, const float* __restrict__ totals, int L,
                                         int with_aug, float lw_cls, float lw_bbox, float lw_aug, float lw_iou,
                                         float* __restrict__ losses3, float* __restrict__ per_level,
                                         float* __restrict__ coef) {
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    const float nts = fmaxf(totals[0], 1.f), baf = fmaxf(totals[1], 1.f);
    float t0 = 0.f, t1 = 0.f, t2 = 0.f;
    for (int l = 0; l < L; l++) {
        const float* s = sums + l * S_N;
        const float lc = lw_cls * (s[S_FOCAL] / nts);
        float lb = lw_bbox * s[S_IOU];
        if (with_aug) lb = (lb + lw_aug * s[S_MSE]) * 0.5f;
        lb = lb / baf;
        const float li = lw_iou * (s[S_BCE] / nts);
        per_level[l] = lc; per_level[L + l] = lb; per_level[2 * L + l] = li;
        t0 += lc; t1 += lb; t2 += li;
    }
    losses3[0] = t0; losses3[1] = t1; losses3[2] = t2;
    coef[0] = 1.f / nts; coef[1] = 1.f / baf;
}

// gradient of (g[0] * loss_cls + g[1] * loss_bbox + g[2] * loss_iou) w.r.t. the raw head output; also the
// per-block partial of d/dScale_l = sum dreg * raw
__global__ __launch_bounds__(256) void rpn_loss_bwd_kernel(const RpnLossParams p, const float* __restrict__ g3,
                                                          const float* __restrict__ coef, float* __restrict__ dy,
                                                          float* __restrict__ dscale_partials) {
    __shared__ float s_red[4];
    int l;
    long long e;
    locate(p, l, e);
    const long long n_l = (long long)p.batch * p.lv.hw[l] * p.A;
    float dsc = 0.f;
    if (e < n_l) {
        const int a = (int)(e % p.A);
        const long long row = e / p.A;
        const int b = (int)(row / p.lv.hw[l]), cell = (int)(row - (long long)b * p.lv.hw[l]);
        const float* yrow = p.y + (size_t)(p.lv.row0[l] + row) * p.ystride;
        float* drow = dy + (size_t)(p.lv.row0[l] + row) * p.ystride;
        const int gi = p.gt_inds[(size_t)b * p.anchors_per_image + p.lv.start[l] + cell * p.A + a];
        const float inv_nts = coef[0], inv_baf = coef[1];
        float dcls = 0.f, diou = 0.f;
        float4 dreg = make_float4(0.f, 0.f, 0.f, 0.f);
        const bool pos = gi > 0;
        float iou_t = 0.f;
        if (pos) {
            PosTerms t;
            const float4 g = *reinterpret_cast<const float4*>(p.gts + (size_t)(p.gt.off[b] + gi - 1) * 4);
            const float scale = p.scales[l];
            pos_terms(p, l, cell, a, yrow, scale, g, t);
            iou_t = t.iou;
            const float x = yrow[5 * p.A + a];
            diou = g3[2] * p.lw_iou * inv_nts * (sigmoidf_(x) - t.iou);
            const float gb = g3[1] * inv_baf * (p.with_aug ? 0.5f : 1.f);
            if (p.reg_mode == 1) {
                // CIoU(deltas, encoded targets) * w: the deltas ARE the loss's boxes, no decode in the chain
                float gr[4];
                ciou_terms(t.d, t.enc, 1e-6f, gr);
                const float k = gb * p.lw_bbox * t.w;
                dreg = make_float4(k * gr[0], k * gr[1], k * gr[2], k * gr[3]);
            } else {
                // ---- -log(clamp(IoU, 1e-6)) * w through the aligned IoU and the decode
                float gx1 = 0.f, gy1 = 0.f, gx2 = 0.f, gy2 = 0.f;
                {
                    const float bw = t.box.z - t.box.x, bh = t.box.w - t.box.y;
                    const float a1 = bw * bh, a2 = (g.z - g.x) * (g.w - g.y);
                    const float ltx = fmaxf(t.box.x, g.x), lty = fmaxf(t.box.y, g.y);
                    const float rbx = fminf(t.box.z, g.z), rby = fminf(t.box.w, g.w);
                    const float wr = rbx - ltx, hr = rby - lty;
                    const float w = wr < 0.f ? 0.f : wr, h = hr < 0.f ? 0.f : hr;
                    const float ov = w * h, ur = a1 + a2 - ov;
                    const float un = fmaxf(ur, 1e-6f);
                    const float iou = ov / un;
                    if (iou >= 1e-6f) {
                        const float gl = -(gb * p.lw_bbox * t.w) / iou;             // dL/dIoU
                        const float g_ov = gl / un, g_un = ur >= 1e-6f ? -gl * ov / (un * un) : 0.f;
                        const float g_ovt = g_ov - g_un;                             // union = a1 + a2 - ov
                        const float g_a1 = g_un;
                        const float g_w = wr >= 0.f ? g_ovt * h : 0.f, g_h = hr >= 0.f ? g_ovt * w : 0.f;
                        // max / min split ties evenly, as torch.max / torch.min of two tensors do
                        const float kx1 = t.box.x > g.x ? 1.f : (t.box.x == g.x ? 0.5f : 0.f);
                        const float ky1 = t.box.y > g.y ? 1.f : (t.box.y == g.y ? 0.5f : 0.f);
                        const float kx2 = t.box.z < g.z ? 1.f : (t.box.z == g.z ? 0.5f : 0.f);
                        const float ky2 = t.box.w < g.w ? 1.f : (t.box.w == g.w ? 0.5f : 0.f);
                        gx1 = -g_w * kx1 - g_a1 * bh;
                        gy1 = -g_h * ky1 - g_a1 * bw;
                        gx2 = g_w * kx2 + g_a1 * bh;
                        gy2 = g_h * ky2 + g_a1 * bw;
                    }
                }
                // decode: x1 = gx - gw/2, x2 = gx + gw/2, gx = px + pw*dx, gw = pw*exp(clamp(dw))
                dreg.x = (gx1 + gx2) * t.pw * p.std[0];
                dreg.y = (gy1 + gy2) * t.ph * p.std[1];
                dreg.z = t.clamp_w ? 0.f : (gx2 - gx1) * 0.5f * t.gw * p.std[2];
                dreg.w = t.clamp_h ? 0.f : (gy2 - gy1) * 0.5f * t.gh * p.std[3];
                if (p.with_aug) {
                    const float k = gb * p.lw_aug * 2.f * t.w;
                    dreg.x += k * (t.d.x - t.enc.x);
                    dreg.y += k * (t.d.y - t.enc.y);
                    dreg.z += k * (t.d.z - t.enc.z);
                    dreg.w += k * (t.d.w - t.enc.w);
                }
            }
            dsc = dreg.x * t.raw.x + dreg.y * t.raw.y + dreg.z * t.raw.z + dreg.w * t.raw.w;
            dreg.x *= scale; dreg.y *= scale; dreg.z *= scale; dreg.w *= scale;
        }
        if (p.cls_mode) {
            dcls = g3[0] * p.lw_cls * inv_nts * vfl_bwd(yrow[a], iou_t, p.focal_gamma, p.focal_alpha, p.cls_mode == 1);
        } else if (gi >= 0) {
            float f = focal_bwd(yrow[a], pos, p.focal_gamma, p.focal_alpha);
            if (pos && p.pos_weight > 0.f) f *= p.pos_weight;
            dcls = g3[0] * p.lw_cls * inv_nts * f;
        }
        drow[a] = dcls;
        float* dr = drow + p.A + a * 4;
        dr[0] = dreg.x; dr[1] = dreg.y; dr[2] = dreg.z; dr[3] = dreg.w;
        drow[5 * p.A + a] = diou;
        if (a == 0)
            for (int c = 6 * p.A; c < p.ystride; c++) drow[c] = 0.f;       // channel padding of the head conv
    }
    const float s = block_sum(dsc, s_red);
    if (threadIdx.x == 0) dscale_partials[blockIdx.x] = s;
}

__global__ __launch_bounds__(256) void rpn_dscale_reduce_kernel(const float* __restrict__ partials, RpnLevels lv,
                                                               float* __restrict__ dscale) {
    __shared__ float s_red[4];
    for (int l = 0; l < lv.num; l++) {
        float v = 0.f;
        for (int blk = lv.blk0[l] + threadIdx.x; blk < lv.blk0[l + 1]; blk += 256) v += partials[blk];
        const float s = block_sum(v, s_red);
        if (threadIdx.x == 0) dscale[l] = s;
    }
}

int fill_rpn_params(RpnLossParams& p, const float* y, int ystride, int batch, int num_levels, const int* heights,
                    const int* widths, const int* strides_w, const int* strides_h, const float* const* base_anchors,
                    int A, const float* scales, const int32_t* gt_inds, const float* gts, const int* gt_offsets_host,
                    const float* cfg) {
    if (!y || !scales || !gt_inds || !heights || !widths || !strides_w || !strides_h || !base_anchors || !cfg ||
        num_levels <= 0 || num_levels > BRCNN_MAX_LEVELS || batch <= 0 || batch > BRCNN_MAX_IMAGES || A <= 0 ||
        ystride < 6 * A || !gt_offsets_host)
        return BRCNN_EINVAL;
    p.y = y; p.ystride = ystride; p.A = A; p.batch = batch; p.scales = scales; p.gt_inds = gt_inds; p.gts = gts;
    for (int b = 0; b <= batch; b++) p.gt.off[b] = gt_offsets_host[b];
    if (p.gt.off[batch] > 0 && !gts) return BRCNN_EINVAL;
    p.lv.num = num_levels;
    int row = 0, start = 0, blk = 0;
    for (int l = 0; l < num_levels; l++) {
        if (heights[l] <= 0 || widths[l] <= 0 || !base_anchors[l]) return BRCNN_EINVAL;
        p.lv.row0[l] = row; p.lv.start[l] = start; p.lv.blk0[l] = blk;
        p.lv.hw[l] = heights[l] * widths[l]; p.lv.width[l] = widths[l];
        p.lv.stride_w[l] = strides_w[l]; p.lv.stride_h[l] = strides_h[l];
        p.lv.base[l] = base_anchors[l];
        row += batch * p.lv.hw[l];
        start += p.lv.hw[l] * A;
        blk += brcnn_cdiv((long long)batch * p.lv.hw[l] * A, 256);
    }
    p.lv.row0[num_levels] = row; p.lv.start[num_levels] = start; p.lv.blk0[num_levels] = blk;
    p.anchors_per_image = start;
    // cfg (20 floats): [focal_gamma, focal_alpha, pos_weight, iou_gamma, mean4, std4, max_ratio, with_aug, lw_cls, lw_bbox,
    //                   lw_aug, lw_iou, cls_mode, reg_mode]
    p.focal_gamma = cfg[0]; p.focal_alpha = cfg[1]; p.pos_weight = cfg[2]; p.iou_gamma = cfg[3];
    for (int k = 0; k < 4; k++) { p.mean[k] = cfg[4 + k]; p.std[k] = cfg[8 + k]; }
    p.max_ratio = cfg[12]; p.with_aug = cfg[13] != 0.f;
    p.lw_cls = cfg[14]; p.lw_bbox = cfg[15]; p.lw_aug = cfg[16]; p.lw_iou = cfg[17];
    p.cls_mode = (int)cfg[18]; p.reg_mode = (int)cfg[19];
    if (p.cls_mode < 0 || p.cls_mode > 2 || p.reg_mode < 0 || p.reg_mode > 1) return BRCNN_EINVAL;
    if (p.reg_mode == 1) p.with_aug = 0;       // the aug MSE term only exists on the decoded branch (:329-359)
    return 0;
}

// ------------------------------------------------------------------------------------------------
enum { B_L = 0, B_WL, B_L1, B_CORRECT, B_NPOS, B_WPOS, B_N };

struct BoostParams {
    const float* cls;          // (N, C+1)
    const float* bbox;         // (N, 4C) or (N, 4) when class agnostic
    const long long* labels;   // (N), background = C
    const float* priors;       // (N)
    const float* ious;         // (N) or NULL (`quality`)
    const float* targets;      // (N, 4)
    int N, C, agnostic;
    float gamma, alpha, iou_gamma, lw_cls, lw_bbox;
    int reg_mean;              // reg_norm == 'mean'
    // plain: the boosted weights are label weights of the head's own loss (DyProbRoIHead / BoostRoIHead,
    // prob_roi_head.py:438-468,604-623 over bbox_head.py loss: avg_factor = max(#{w > 0}, 1)) instead of norm_loss
    int plain;
    float beta;                // SmoothL1Loss beta of the box term (<= 0: L1Loss)
};

// smooth_l1_loss (mmdet/models/losses/smooth_l1_loss.py:9-32): 0.5 d^2 / beta below beta, d - 0.5 beta above; beta <= 0: |d|
__device__ __forceinline__ float box_term(float d, float beta) {
    const float a = fabsf(d);
    if (beta > 0.f && a < beta) return 0.5f * a * a / beta;
    return beta > 0.f ? a - 0.5f * beta : a;
}

// one wavefront per row: lanes over the classes
__device__ __forceinline__ void row_softmax(const float* row, int nc, int lane, float& mx, float& se, int& amax) {
    mx = -FLT_MAX;
    amax = 0x7fffffff;
    for (int c = lane; c < nc; c += 64) {
        const float v = row[c];
        if (v > mx) { mx = v; amax = c; }
    }
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) {
        const float om = __shfl_xor(mx, d, 64);
        const int oa = __shfl_xor(amax, d, 64);
        if (om > mx || (om == mx && oa < amax)) { mx = om; amax = oa; }
    }
    se = 0.f;
    for (int c = lane; c < nc; c += 64) se += expf(row[c] - mx);
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) se += __shfl_xor(se, d, 64);
}

__device__ __forceinline__ float boost_weight(const BoostParams& p, int i, float p_label) {
    float w = powf(1.f - p.priors[i], p.gamma);
    if (p.gamma == 0.5f) w = sqrtf(1.f - p.priors[i]);
    if (p.ious) w = powf(fabsf(p.ious[i] - p_label), p.iou_gamma) * w;
    if (p.alpha != 0.f) w *= p.alpha;
    return w;
}

__global__ __launch_bounds__(256) void boost_loss_fwd_kernel(const BoostParams p, float* __restrict__ partials) {
    __shared__ float s_part[4][B_N];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int i = blockIdx.x * 4 + wave;
    float v[B_N] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    if (i < p.N) {
        const int nc = p.C + 1;
        const float* row = p.cls + (size_t)i * nc;
        float mx, se;
        int amax;
        row_softmax(row, nc, lane, mx, se, amax);
        const int lab = (int)p.labels[i];
        const float xl = row[lab];
        const float ce = logf(se) - (xl - mx);                    // -log_softmax[label]
        const float L = p.lw_cls * ce;
        const float w = boost_weight(p, i, expf(xl - mx) / se);
        v[B_L] = L;
        v[B_WL] = w * L;
        v[B_WPOS] = w > 0.f ? 1.f : 0.f;
        v[B_CORRECT] = amax == lab ? 1.f : 0.f;
        if (lab >= 0 && lab < p.C) {
            v[B_NPOS] = 1.f;
            const float* bp = p.bbox + (size_t)i * (p.agnostic ? 4 : 4 * p.C) + (p.agnostic ? 0 : 4 * lab);
            const float* tg = p.targets + (size_t)i * 4;
            v[B_L1] = box_term(bp[0] - tg[0], p.beta) + box_term(bp[1] - tg[1], p.beta) + box_term(bp[2] - tg[2], p.beta) +
                      box_term(bp[3] - tg[3], p.beta);
        }
    }
    if (lane == 0)
        for (int k = 0; k < B_N; k++) s_part[wave][k] = v[k];
    __syncthreads();
    if (threadIdx.x < B_N)
        partials[(size_t)blockIdx.x * B_N + threadIdx.x] =
            s_part[0][threadIdx.x] + s_part[1][threadIdx.x] + s_part[2][threadIdx.x] + s_part[3][threadIdx.x];
}

// out3 = [loss_cls, loss_bbox, acc]; coef = [c / N, lw_bbox / norm]
__global__ __launch_bounds__(256) void boost_loss_finalize_kernel(const float* __restrict__ partials, int nblocks,
                                                                 BoostParams p, float* __restrict__ out3,
                                                                 float* __restrict__ coef) {
    __shared__ float s_red[4];
    __shared__ float s_sum[B_N];
    for (int k = 0; k < B_N; k++) {
        float v = 0.f;
        for (int blk = threadIdx.x; blk < nblocks; blk += 256) v += partials[(size_t)blk * B_N + k];
        const float s = block_sum(v, s_red);
        if (threadIdx.x == 0) s_sum[k] = s;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        const float n = (float)p.N;
        // norm_loss: weights * (loss.sum() / (weights * loss).sum()), mean over n; plain: weighted sum / #{w > 0}
        const float c = p.plain ? n / fmaxf(s_sum[B_WPOS], 1.f) : s_sum[B_L] / s_sum[B_WL];
        out3[0] = c * s_sum[B_WL] / n;
        const float norm = p.reg_mean ? 4.f * s_sum[B_NPOS] : n;
        out3[1] = s_sum[B_NPOS] > 0.f ? p.lw_bbox * s_sum[B_L1] / norm : 0.f;
        out3[2] = s_sum[B_CORRECT] * (100.f / n);
        coef[0] = c / n;
        coef[1] = s_sum[B_NPOS] > 0.f ? p.lw_bbox / norm : 0.f;
    }
}

__global__ __launch_bounds__(256) void boost_loss_bwd_kernel(const BoostParams p, const float* __restrict__ g3,
                                                            const float* __restrict__ coef, float* __restrict__ dcls,
                                                            float* __restrict__ dbbox) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int i = blockIdx.x * 4 + wave;
    if (i >= p.N) return;
    const int nc = p.C + 1;
    const float* row = p.cls + (size_t)i * nc;
    float mx, se;
    int amax;
    row_softmax(row, nc, lane, mx, se, amax);
    const int lab = (int)p.labels[i];
    const float w = boost_weight(p, i, expf(row[lab] - mx) / se);
    const float k = g3[0] * p.lw_cls * w * coef[0];
    for (int c = lane; c < nc; c += 64) {
        const float sm = expf(row[c] - mx) / se;
        dcls[(size_t)i * nc + c] = k * (sm - (c == lab ? 1.f : 0.f));
    }
    const int nb = p.agnostic ? 4 : 4 * p.C;
    const bool pos = lab >= 0 && lab < p.C;
    const int c0 = p.agnostic ? 0 : 4 * lab;
    const float kb = g3[1] * coef[1];
    for (int c = lane; c < nb; c += 64) {
        float gv = 0.f;
        if (pos && c >= c0 && c < c0 + 4) {
            const float d = p.bbox[(size_t)i * nb + c] - p.targets[(size_t)i * 4 + (c - c0)];
            gv = (p.beta > 0.f && fabsf(d) < p.beta) ? kb * d / p.beta : (d > 0.f ? kb : (d < 0.f ? -kb : 0.f));
        }
        dbbox[(size_t)i * nb + c] = gv;
    }
}

int fill_boost(BoostParams& p, const float* cls, const float* bbox, const int64_t* labels, const float* priors,
               const float* ious, const float* targets, int n, int num_classes, int agnostic, const float* cfg) {
    if (!cls || !bbox || !labels || !priors || !targets || !cfg || n <= 0 || num_classes <= 0) return BRCNN_EINVAL;
    p.cls = cls; p.bbox = bbox; p.labels = (const long long*)labels; p.priors = priors; p.ious = ious; p.targets = targets;
    p.N = n; p.C = num_classes; p.agnostic = agnostic ? 1 : 0;
    // cfg: [gamma, alpha, iou_gamma, lw_cls, lw_bbox, reg_mean, plain label weights, smooth-L1 beta]
    p.gamma = cfg[0]; p.alpha = cfg[1]; p.iou_gamma = cfg[2]; p.lw_cls = cfg[3]; p.lw_bbox = cfg[4];
    p.reg_mean = cfg[5] != 0.f;
    p.plain = cfg[6] != 0.f;
    p.beta = cfg[7];
    return 0;
}

}  // namespace

BRCNN_API size_t brcnn_rpn_loss_workspace_bytes(int batch, int num_levels, const int* heights, const int* widths,
                                               int anchors_per_cell) {
    size_t blocks = 0;
    for (int l = 0; l < num_levels; l++)
        blocks += brcnn_cdiv((long long)batch * heights[l] * widths[l] * anchors_per_cell, 256);
    return blocks * S_N * sizeof(float) + 256;
}

BRCNN_API int brcnn_rpn_loss_forward(const float* y, int ystride, int batch, int num_levels, const int* heights,
                                     const int* widths, const int* strides_w, const int* strides_h,
                                     const float* const* base_anchors, int anchors_per_cell, const float* scales,
                                     const int32_t* gt_inds, const float* gts, const int* gt_offsets_host,
                                     const float* cfg20_host, void* workspace, size_t workspace_bytes, float* sums,
                                     float* totals, void* stream) {
    RpnLossParams p;
    if (int st = fill_rpn_params(p, y, ystride, batch, num_levels, heights, widths, strides_w, strides_h, base_anchors,
                                 anchors_per_cell, scales, gt_inds, gts, gt_offsets_host, cfg20_host))
        return st;
    const int blocks = p.lv.blk0[num_levels];
    if (!workspace || !sums || !totals || workspace_bytes < (size_t)blocks * S_N * sizeof(float)) return BRCNN_EINVAL;
    hipStream_t s = (hipStream_t)stream;
    hipLaunchKernelGGL(rpn_loss_fwd_kernel, dim3(blocks), dim3(256), 0, s, p, (float*)workspace);
    BRCNN_LAUNCH_CHECK();
    hipLaunchKernelGGL(rpn_loss_reduce_kernel, dim3(1), dim3(256), 0, s, (const float*)workspace, p.lv, sums, totals);
    BRCNN_LAUNCH_CHECK();
    return 0;
}

BRCNN_API int brcnn_rpn_loss_finalize(const float* sums, const float* totals, int num_levels, const float* cfg20_host,
                                      float* losses3, float* per_level, float* coef2, void* stream) {
    if (!sums || !totals || !cfg20_host || !losses3 || !per_level || !coef2 || num_levels <= 0 ||
        num_levels > BRCNN_MAX_LEVELS)
        return BRCNN_EINVAL;
    hipLaunchKernelGGL(rpn_loss_finalize_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, sums, totals, num_levels,
                       (cfg20_host[13] != 0.f && cfg20_host[19] == 0.f) ? 1 : 0, cfg20_host[14], cfg20_host[15], cfg20_host[16], cfg20_host[17],
                       losses3, per_level, coef2);
    BRCNN_LAUNCH_CHECK();
    return 0;
}

BRCNN_API int brcnn_rpn_loss_backward(const float* y, int ystride, int batch, int num_levels, const int* heights,
                                      const int* widths, const int* strides_w, const int* strides_h,
                                      const float* const* base_anchors, int anchors_per_cell, const float* scales,
                                      const int32_t* gt_inds, const float* gts, const int* gt_offsets_host,
                                      const float* cfg20_host, const float* grad3, const float* coef2,
                                      void* workspace, size_t workspace_bytes, float* dy, float* dscales,
                                      void* stream) {
    RpnLossParams p;
    if (int st = fill_rpn_params(p, y, ystride, batch, num_levels, heights, widths, strides_w, strides_h, base_anchors,
                                 anchors_per_cell, scales, gt_inds, gts, gt_offsets_host, cfg20_host))
        return st;
    const int blocks = p.lv.blk0[num_levels];
    if (!workspace || !grad3 || !coef2 || !dy || !dscales || workspace_bytes < (size_t)blocks * sizeof(float))
        return BRCNN_EINVAL;
    hipStream_t s = (hipStream_t)stream;
    hipLaunchKernelGGL(rpn_loss_bwd_kernel, dim3(blocks), dim3(256), 0, s, p, grad3, coef2, dy, (float*)workspace);
    BRCNN_LAUNCH_CHECK();
    hipLaunchKernelGGL(rpn_dscale_reduce_kernel, dim3(1), dim3(256), 0, s, (const float*)workspace, p.lv, dscales);
    BRCNN_LAUNCH_CHECK();
    return 0;
}

BRCNN_API size_t brcnn_boost_loss_workspace_bytes(int n) {
    return (size_t)brcnn_cdiv(n > 0 ? n : 1, 4) * B_N * sizeof(float) + 256;
}

BRCNN_API int brcnn_boost_loss_forward_ex(const float* cls_score, const float* bbox_pred, const int64_t* labels,
                                          const float* priors, const float* ious, const float* bbox_targets, int n,
                                          int num_classes, int reg_class_agnostic, const float* cfg8_host,
                                          void* workspace, size_t workspace_bytes, float* out3, float* coef2,
                                          void* stream) {
    BoostParams p;
    if (int st = fill_boost(p, cls_score, bbox_pred, labels, priors, ious, bbox_targets, n, num_classes,
                            reg_class_agnostic, cfg8_host))
        return st;
    const int blocks = brcnn_cdiv(n, 4);
    if (!workspace || !out3 || !coef2 || workspace_bytes < (size_t)blocks * B_N * sizeof(float)) return BRCNN_EINVAL;
    hipStream_t s = (hipStream_t)stream;
    hipLaunchKernelGGL(boost_loss_fwd_kernel, dim3(blocks), dim3(256), 0, s, p, (float*)workspace);
    BRCNN_LAUNCH_CHECK();
    hipLaunchKernelGGL(boost_loss_finalize_kernel, dim3(1), dim3(256), 0, s, (const float*)workspace, blocks, p, out3,
                       coef2);
    BRCNN_LAUNCH_CHECK();
    return 0;
}

BRCNN_API int brcnn_boost_loss_forward(const float* cls_score, const float* bbox_pred, const int64_t* labels,
                                       const float* priors, const float* ious, const float* bbox_targets, int n,
                                       int num_classes, int reg_class_agnostic, const float* cfg6_host,
                                       void* workspace, size_t workspace_bytes, float* out3, float* coef2,
                                       void* stream) {
    if (!cfg6_host) return BRCNN_EINVAL;
    const float cfg8[8] = {cfg6_host[0], cfg6_host[1], cfg6_host[2], cfg6_host[3], cfg6_host[4], cfg6_host[5], 0.f, 0.f};
    return brcnn_boost_loss_forward_ex(cls_score, bbox_pred, labels, priors, ious, bbox_targets, n, num_classes,
                                       reg_class_agnostic, cfg8, workspace, workspace_bytes, out3, coef2, stream);
}

BRCNN_API int brcnn_boost_loss_backward_ex(const float* cls_score, const float* bbox_pred, const int64_t* labels,
                                           const float* priors, const float* ious, const float* bbox_targets, int n,
                                           int num_classes, int reg_class_agnostic, const float* cfg8_host,
                                           const float* grad3, const float* coef2, float* dcls, float* dbbox,
                                           void* stream) {
    BoostParams p;
    if (int st = fill_boost(p, cls_score, bbox_pred, labels, priors, ious, bbox_targets, n, num_classes,
                            reg_class_agnostic, cfg8_host))
        return st;
    if (!grad3 || !coef2 || !dcls || !dbbox) return BRCNN_EINVAL;
    hipLaunchKernelGGL(boost_loss_bwd_kernel, dim3(brcnn_cdiv(n, 4)), dim3(256), 0, (hipStream_t)stream, p, grad3, coef2,
                       dcls, dbbox);
    BRCNN_LAUNCH_CHECK();
    return 0;
}

BRCNN_API int brcnn_boost_loss_backward(const float* cls_score, const float* bbox_pred, const int64_t* labels,
                                        const float* priors, const float* ious, const float* bbox_targets, int n,
                                        int num_classes, int reg_class_agnostic, const float* cfg6_host,
                                        const float* grad3, const float* coef2, float* dcls, float* dbbox,
                                        void* stream) {
    if (!cfg6_host) return BRCNN_EINVAL;
    const float cfg8[8] = {cfg6_host[0], cfg6_host[1], cfg6_host[2], cfg6_host[3], cfg6_host[4], cfg6_host[5], 0.f, 0.f};
    return brcnn_boost_loss_backward_ex(cls_score, bbox_pred, labels, priors, ious, bbox_targets, n, num_classes,
                                        reg_class_agnostic, cfg8, grad3, coef2, dcls, dbbox, stream);
}
