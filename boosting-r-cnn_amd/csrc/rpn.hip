// RetinaRPN proposal stage kernels (ATSSRPNHead._get_bboxes_single,
// mmdet/models/dense_heads/atss_rpn_head.py:688-760).
//
//   rpn_score : score = sqrt(sigmoid(cls) * sigmoid(iou))       (:712-725)   HBM stream
//   rpn_decode: anchors regenerated on the fly (AnchorGenerator.single_level_grid_anchors,
//               mmdet/core/anchor/anchor_generator.py:336-381: base_anchor + (x*stride_w,
//               y*stride_h), row-major cells, A anchors contiguous per cell) -- the 201 600 x 4
//               anchor tensor is never materialised -- then delta2bbox
//               (mmdet/core/bbox/coder/delta_xywh_bbox_coder.py:145-272) with the reference's
//               operation order (no FMA contraction), border clip to [0,W]x[0,H], and the
//               min_bbox_size validity flag (:747-754).
// Head outputs are NHWC: cls/iou (N,H,W,A), bbox_pred (N,H,W,4A) == the reference's
// permute(1,2,0).reshape(-1[,4]) order, so the flat anchor index is the same.
#include "common.h"

namespace {

// cls / iou: `rows` pixels of A channels each, consecutive pixels `cls_stride` / `iou_stride`
// floats apart (the fused 54-channel head output is read in place); score (rows, A) dense.
__global__ __launch_bounds__(256) void rpn_score_kernel(const float* __restrict__ cls,
                                                       const float* __restrict__ iou,
                                                       float* __restrict__ score, long long rows,
                                                       int A, int cls_stride, int iou_stride) {
    const long long n = rows * A;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n;
         i += (long long)gridDim.x * blockDim.x) {
        const long long r = i / A;
        const int c = (int)(i - r * A);
        const float a = 1.f / (1.f + expf(-cls[r * cls_stride + c]));
        const float b = 1.f / (1.f + expf(-iou[r * iou_stride + c]));
        score[i] = sqrtf(a * b);
    }
}

struct DecodeParams {
    float mean[4], std[4];
    float max_ratio, max_h, max_w, min_size;
    float pred_scale;     // the level's learnable `Scale` (atss_rpn_head.py:211), 1 if pre-applied
    int pred_stride;      // floats between consecutive pixels of bbox_pred (4A when dense)
    int clip;
};

// decode of one picked anchor (delta_xywh_bbox_coder.py:145-272 on the grid anchor the flat index
// denotes): returns the clipped box, *ok = both sides longer than min_size
__device__ __forceinline__ float4 decode_one(long long idx, int b, const float* __restrict__ bbox_pred,
                                             const float* __restrict__ base_anchors, int hwA, int width, int A,
                                             int stride_w, int stride_h, int pred_stride, float pred_scale,
                                             const DecodeParams& dp, bool* ok) {
    const int a = (int)(idx % A);
    const long long cell = idx / A;
    const int cx = (int)(cell % width), cy = (int)(cell / width);
    const float sx = (float)(cx * stride_w), sy = (float)(cy * stride_h);
    const float4 ba = *reinterpret_cast<const float4*>(base_anchors + a * 4);
    const float x1 = ba.x + sx, y1 = ba.y + sy, x2 = ba.z + sx, y2 = ba.w + sy;
    const float* dptr = bbox_pred + ((size_t)b * (hwA / A) + cell) * pred_stride + a * 4;
    float4 d = make_float4(dptr[0], dptr[1], dptr[2], dptr[3]);
    if (pred_scale != 1.f) {
        d.x *= pred_scale; d.y *= pred_scale; d.z *= pred_scale; d.w *= pred_scale;
    }
    const float dx = d.x * dp.std[0] + dp.mean[0];
    const float dy = d.y * dp.std[1] + dp.mean[1];
    float dw = d.z * dp.std[2] + dp.mean[2];
    float dh = d.w * dp.std[3] + dp.mean[3];
    const float px = (x1 + x2) * 0.5f, py = (y1 + y2) * 0.5f;
    const float pw = x2 - x1, ph = y2 - y1;
    const float dxw = pw * dx, dyh = ph * dy;
    dw = fminf(fmaxf(dw, -dp.max_ratio), dp.max_ratio);
    dh = fminf(fmaxf(dh, -dp.max_ratio), dp.max_ratio);
    const float gw = pw * expf(dw), gh = ph * expf(dh);
    const float gx = px + dxw, gy = py + dyh;
    float ox1 = gx - gw * 0.5f, oy1 = gy - gh * 0.5f;
    float ox2 = gx + gw * 0.5f, oy2 = gy + gh * 0.5f;
    if (dp.clip) {
        ox1 = ox1 < 0.f ? 0.f : ox1; ox1 = ox1 > dp.max_w ? dp.max_w : ox1;
        oy1 = oy1 < 0.f ? 0.f : oy1; oy1 = oy1 > dp.max_h ? dp.max_h : oy1;
        ox2 = ox2 < 0.f ? 0.f : ox2; ox2 = ox2 > dp.max_w ? dp.max_w : ox2;
        oy2 = oy2 < 0.f ? 0.f : oy2; oy2 = oy2 > dp.max_h ? dp.max_h : oy2;
    }
    *ok = (ox2 - ox1) > dp.min_size && (oy2 - oy1) > dp.min_size;
    return make_float4(ox1, oy1, ox2, oy2);
}

// inds: (batch, count) flat anchor indices of one level; bbox_pred: (batch, H*W*A, 4)
__global__ __launch_bounds__(256) void rpn_decode_kernel(
    const int64_t* __restrict__ inds, const float* __restrict__ bbox_pred,
    const float* __restrict__ base_anchors, int batch, int count, int hwA, int width, int A,
    int stride_w, int stride_h, DecodeParams dp, float* __restrict__ proposals,
    uint8_t* __restrict__ valid) {
    const long long total = (long long)batch * count;
    for (long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x; t < total;
         t += (long long)gridDim.x * blockDim.x) {
        bool ok;
        const float4 o = decode_one(inds[t], (int)(t / count), bbox_pred, base_anchors, hwA, width, A, stride_w,
                                    stride_h, dp.pred_stride, dp.pred_scale, dp, &ok);
        *reinterpret_cast<float4*>(proposals + (size_t)t * 4) = o;
        if (valid) valid[t] = ok ? 1 : 0;
    }
}

// all pyramid levels of the proposal stage in one launch: the picked anchors of level l occupy
// columns [col0[l], col0[l+1]) of the (batch, T) candidate slots; also writes the level id column
struct DecodeLevels {
    int num;
    const int64_t* inds[BRCNN_MAX_LEVELS];
    const float* pred[BRCNN_MAX_LEVELS];
    const float* base[BRCNN_MAX_LEVELS];
    int pred_stride[BRCNN_MAX_LEVELS], hwA[BRCNN_MAX_LEVELS], width[BRCNN_MAX_LEVELS];
    int stride_w[BRCNN_MAX_LEVELS], stride_h[BRCNN_MAX_LEVELS], col0[BRCNN_MAX_LEVELS + 1];
    float pred_scale[BRCNN_MAX_LEVELS];
    const float* scale_dev;            // (num) device-resident scales (training: the Scale parameters), or NULL
    const float* max_shape_dev;        // (batch, 2) [h, w] per-image clip border (img_shape), or NULL = dp.max_h/w
};

__global__ __launch_bounds__(256) void rpn_decode_levels_kernel(DecodeLevels lv, int batch, int A, DecodeParams dp,
                                                               float* __restrict__ proposals,
                                                               uint8_t* __restrict__ valid, int64_t* __restrict__ ids) {
    const int T = lv.col0[lv.num];
    const long long total = (long long)batch * T;
    for (long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x; t < total;
         t += (long long)gridDim.x * blockDim.x) {
        const int b = (int)(t / T), col = (int)(t - (long long)b * T);
        int l = 0;
#pragma unroll
        for (int i = 1; i < BRCNN_MAX_LEVELS; i++)
            if (i < lv.num && col >= lv.col0[i]) l = i;
        const int count = lv.col0[l + 1] - lv.col0[l];
        const long long idx = lv.inds[l][(size_t)b * count + (col - lv.col0[l])];
        bool ok;
        DecodeParams d = dp;
        if (lv.max_shape_dev) { d.max_h = lv.max_shape_dev[2 * b]; d.max_w = lv.max_shape_dev[2 * b + 1]; }
        const float4 o = decode_one(idx, b, lv.pred[l], lv.base[l], lv.hwA[l], lv.width[l], A, lv.stride_w[l],
                                    lv.stride_h[l], lv.pred_stride[l], lv.scale_dev ? lv.scale_dev[l] : lv.pred_scale[l],
                                    d, &ok);
        *reinterpret_cast<float4*>(proposals + (size_t)t * 4) = o;
        valid[t] = ok ? 1 : 0;
        ids[t] = l;
    }
}


// ---- second-stage candidates of a whole batch (prob_roi_head.py:232-240 score fusion,
// convfc_bbox_head.py:294-330 get_bboxes up to the NMS call): per (image, proposal, class)
//     score = sqrt(softmax_c * prior),  box = delta2bbox(proposal, deltas[4c:4c+4]) clipped to the
//     image and divided by its scale factor,  valid = score > thr and the proposal row is real.
// `probs` are the softmax outputs (all C+1 columns, the background column is not emitted).
__global__ __launch_bounds__(256) void rcnn_decode_kernel(const float* __restrict__ probs, const float* __restrict__ bbox_pred,
                                                         const float* __restrict__ props, const int* __restrict__ num,
                                                         const float* __restrict__ max_shape, const float* __restrict__ scale,
                                                         int B, int K, int C, float score_thr, DecodeParams dp,
                                                         float* __restrict__ boxes, float* __restrict__ scores,
                                                         int64_t* __restrict__ labels, uint8_t* __restrict__ valid) {
    const long long total = (long long)B * K * C;
    for (long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x; t < total;
         t += (long long)gridDim.x * blockDim.x) {
        const int c = (int)(t % C);
        const long long bk = t / C;
        const int b = (int)(bk / K), k = (int)(bk - (long long)b * K);
        const float* pr = props + bk * 5;            // x1, y1, x2, y2, prior
        const float s = sqrtf(probs[bk * (C + 1) + c] * pr[4]);
        const float4 d = *reinterpret_cast<const float4*>(bbox_pred + bk * 4 * C + 4 * c);
        const float x1 = pr[0], y1 = pr[1], x2 = pr[2], y2 = pr[3];
        const float dx = d.x * dp.std[0] + dp.mean[0];
        const float dy = d.y * dp.std[1] + dp.mean[1];
        float dw = d.z * dp.std[2] + dp.mean[2];
        float dh = d.w * dp.std[3] + dp.mean[3];
        const float px = (x1 + x2) * 0.5f, py = (y1 + y2) * 0.5f;
        const float pw = x2 - x1, ph = y2 - y1;
        const float dxw = pw * dx, dyh = ph * dy;
        dw = fminf(fmaxf(dw, -dp.max_ratio), dp.max_ratio);
        dh = fminf(fmaxf(dh, -dp.max_ratio), dp.max_ratio);
        const float gw = pw * expf(dw), gh = ph * expf(dh);
        const float gx = px + dxw, gy = py + dyh;
        float ox1 = gx - gw * 0.5f, oy1 = gy - gh * 0.5f;
        float ox2 = gx + gw * 0.5f, oy2 = gy + gh * 0.5f;
        const float mh = max_shape[2 * b], mw = max_shape[2 * b + 1];
        ox1 = ox1 < 0.f ? 0.f : ox1; ox1 = ox1 > mw ? mw : ox1;
        oy1 = oy1 < 0.f ? 0.f : oy1; oy1 = oy1 > mh ? mh : oy1;
        ox2 = ox2 < 0.f ? 0.f : ox2; ox2 = ox2 > mw ? mw : ox2;
        oy2 = oy2 < 0.f ? 0.f : oy2; oy2 = oy2 > mh ? mh : oy2;
        if (scale) {
            const float* sf = scale + 4 * b;
            ox1 = ox1 / sf[0]; oy1 = oy1 / sf[1]; ox2 = ox2 / sf[2]; oy2 = oy2 / sf[3];
        }
        *reinterpret_cast<float4*>(boxes + t * 4) = make_float4(ox1, oy1, ox2, oy2);
        scores[t] = s;
        labels[t] = c;
        valid[t] = (s > score_thr && k < num[b]) ? 1 : 0;
    }
}


// ---- per-(image, level) top-k of the proposal scores (atss_rpn_head.py:727-737 sorts the
// level and keeps nms_pre; the shared tie rule is descending score, ascending index) --------
// One 1024-thread workgroup per (image, level).  Four 8-bit radix-select passes over the
// order-preserving key find the k-th key exactly; one more pass collects the k winners as
// (key << 32 | index) composites -- ties at the threshold are taken in index order -- and a
// bitonic sort of the <= 4096 composites in LDS puts them in (score desc, index asc) order.
// A job = one (row of a level) or, for a long level, one PART of its rows (stage 1: the level's
// top-k is contained in the union of its parts' top-k) or the merge of those parts (stage 2, which
// maps candidate positions back to anchor indices through `idx_map`).  For equal scores the
// candidate position order equals the anchor index order (parts are index ranges in order and
// each part's winners are (score desc, index asc) sorted), so the tie rule survives the split.
constexpr int TOPK_MAX_JOBS = 24;
struct TopkLevels {
    const float* score[TOPK_MAX_JOBS];
    float* out_score[TOPK_MAX_JOBS];
    int64_t* out_idx[TOPK_MAX_JOBS];
    const int64_t* idx_map[TOPK_MAX_JOBS];
    long long row_stride[TOPK_MAX_JOBS], out_stride[TOPK_MAX_JOBS], map_stride[TOPK_MAX_JOBS];
    int n[TOPK_MAX_JOBS], idx_add[TOPK_MAX_JOBS];
};

// descending key: larger score <=> smaller key; -0.0 == +0.0
__device__ __forceinline__ unsigned desc_key(float x) {
    unsigned u = __float_as_uint(x);
    if (u == 0x80000000u) u = 0u;
    const unsigned asc = u ^ ((u >> 31) ? 0xffffffffu : 0x80000000u);
    return ~asc;
}

__global__ __launch_bounds__(1024) void rpn_topk_kernel(TopkLevels lv, int k, int KP) {
    extern __shared__ unsigned long long sel[];     // [KP]
    __shared__ int hist[256];
    __shared__ unsigned s_prefix;
    __shared__ int s_need, s_cnt, s_ties;
    __shared__ int wsum[16];
    const int lvl = blockIdx.y, b = blockIdx.x, tid = threadIdx.x;
    const int n = lv.n[lvl];
    const int kk = min(k, n);
    const float* __restrict__ sc = lv.score[lvl] + (size_t)b * lv.row_stride[lvl];
    float* __restrict__ os = lv.out_score[lvl] + (size_t)b * lv.out_stride[lvl];
    int64_t* __restrict__ oi = lv.out_idx[lvl] + (size_t)b * lv.out_stride[lvl];
    const int64_t* __restrict__ imap = lv.idx_map[lvl] ? lv.idx_map[lvl] + (size_t)b * lv.map_stride[lvl] : nullptr;
    const int iadd = lv.idx_add[lvl];
    (void)kk;
    if (n <= k) {       // the reference keeps the level unsorted in this case
        for (int i = tid; i < n; i += 1024) { os[i] = sc[i]; oi[i] = imap ? imap[i] : (int64_t)(i + iadd); }
        return;
    }
    // the job's keys stay in registers over the five passes (every job the host builds is <= 32 768 long: 32 per thread,
    // loaded once, all loads in flight); longer rows fall back to re-reading the scores.  Round 6: each pass was
    // `for i: key(sc[i]) -> LDS atomic`, one dependent load per iteration -- 5 x 20-32 serial L2 latencies per launch.
    constexpr int EPT = 32;
    const bool cached = n <= EPT * 1024;
    unsigned dkr[EPT];
    if (cached) {
#pragma unroll
        for (int j = 0; j < EPT; j++) dkr[j] = desc_key(sc[min(tid + j * 1024, n - 1)]);
    }
    auto for_each_key = [&](auto&& f) {
        if (cached) {
#pragma unroll
            for (int j = 0; j < EPT; j++) {
                const int i = tid + j * 1024;
                if (i < n) f(i, dkr[j]);
            }
        } else {
            for (int i = tid; i < n; i += 1024) f(i, desc_key(sc[i]));
        }
    };
    unsigned prefix = 0u, mask = 0u;
    int need = k;
    for (int pass = 0; pass < 4; pass++) {
        const int shift = 24 - 8 * pass;
        if (tid < 256) hist[tid] = 0;
        __syncthreads();
        for_each_key([&](int, unsigned dk) {
            if ((dk & mask) == prefix) atomicAdd(&hist[(dk >> shift) & 255u], 1);
        });
        __syncthreads();
        // the digit whose cumulative count first reaches `need`: a 256-wide scan (a single thread
        // walking the bins cost ~25 us per launch over the four passes)
        if (tid < 256) {
            const int h = hist[tid];
            int incl = h;
#pragma unroll
            for (int d = 1; d < 64; d <<= 1) {
                const int o = __shfl_up(incl, d, 64);
                if ((tid & 63) >= d) incl += o;
            }
            if ((tid & 63) == 63) wsum[tid >> 6] = incl;
        }
        __syncthreads();
        if (tid < 256) {
            const int h = hist[tid];
            int incl = h;
#pragma unroll
            for (int d = 1; d < 64; d <<= 1) {
                const int o = __shfl_up(incl, d, 64);
                if ((tid & 63) >= d) incl += o;
            }
            int woff = 0;
            for (int w = 0; w < (tid >> 6); w++) woff += wsum[w];
            incl += woff;
            const int excl = incl - h;
            const bool hit = (excl < need && incl >= need) || (tid == 255 && incl < need);
            if (hit) {
                s_prefix = prefix | ((unsigned)tid << shift);
                s_need = need - excl;
                s_ties = h;
            }
        }
        __syncthreads();
        prefix = s_prefix;
        need = s_need;
        mask |= 0xffu << shift;
        __syncthreads();
    }
    const unsigned T = prefix;          // key of the k-th element; `need` of the s_ties equal keys are in
    const int ties = s_ties;
    const int nless = k - need;
    if (tid == 0) s_cnt = 0;
    for (int i = k + tid; i < KP; i += 1024) sel[i] = ~0ull;
    __syncthreads();
    for_each_key([&](int i, unsigned dk) {
        if (dk < T) {
            const int pos = atomicAdd(&s_cnt, 1);
            sel[pos] = ((unsigned long long)dk << 32) | (unsigned)i;
        } else if (dk == T && ties == need) {
            const int pos = nless + atomicAdd(&s_ties, -1) - 1;      // any order: all ties are in
            sel[pos] = ((unsigned long long)dk << 32) | (unsigned)i;
        }
    });
    if (ties != need) {                 // more ties than room: the lowest indices win
        const int lane = tid & 63, wave = tid >> 6;
        int base = 0;
        for (int c0 = 0; c0 < n && base < need; c0 += 1024) {
            const int i = c0 + tid;
            const bool f = i < n && desc_key(sc[i]) == T;
            const unsigned long long bal = __ballot(f);
            if (lane == 0) wsum[wave] = __popcll(bal);
            __syncthreads();
            int off = __popcll(bal & ((1ull << lane) - 1ull)), tot = 0;
            for (int w = 0; w < 16; w++) {
                if (w < wave) off += wsum[w];
                tot += wsum[w];
            }
            if (f && base + off < need) sel[nless + base + off] = ((unsigned long long)T << 32) | (unsigned)i;
            base += tot;
            __syncthreads();
        }
    }
    // bitonic sort, ascending composites
    for (int size = 2; size <= KP; size <<= 1)
        for (int stride = size >> 1; stride > 0; stride >>= 1) {
            __syncthreads();
            for (int t = tid; t < (KP >> 1); t += 1024) {
                const int pos = 2 * t - (t & (stride - 1));
                const unsigned long long a = sel[pos], c = sel[pos + stride];
                const bool up = (pos & size) == 0;
                if ((a > c) == up) { sel[pos] = c; sel[pos + stride] = a; }
            }
        }
    __syncthreads();
    for (int j = tid; j < k; j += 1024) {
        const unsigned idx = (unsigned)(sel[j] & 0xffffffffull);
        oi[j] = imap ? imap[idx] : (int64_t)idx + iadd;
        os[j] = sc[idx];
    }
}

}  // namespace

BRCNN_API int brcnn_rpn_score(const float* cls, const float* iou, float* score, int64_t rows,
                              int num_anchors, int cls_stride, int iou_stride, void* stream) {
    if (rows < 0 || num_anchors <= 0 || cls_stride < num_anchors || iou_stride < num_anchors)
        return BRCNN_EINVAL;
    if (rows == 0) return 0;
    if (!cls || !iou || !score) return BRCNN_EINVAL;
    long long g = (rows * num_anchors + 255) / 256;
    if (g > 4096) g = 4096;
    hipLaunchKernelGGL(rpn_score_kernel, dim3((int)g), dim3(256), 0, (hipStream_t)stream, cls, iou,
                       score, (long long)rows, num_anchors, cls_stride, iou_stride);
    BRCNN_LAUNCH_CHECK();
    return 0;
}

BRCNN_API int brcnn_rpn_decode(const int64_t* topk_inds, const float* bbox_pred,
                               int pred_stride, float pred_scale, const float* base_anchors,
                               int batch, int count, int height, int width, int num_anchors,
                               int stride_w, int stride_h,
                               const float* means4_host, const float* stds4_host,
                               double wh_ratio_clip, float max_h, float max_w, float min_size,
                               float* proposals, uint8_t* valid, void* stream) {
    if (batch < 0 || count < 0 || height <= 0 || width <= 0 || num_anchors <= 0 ||
        pred_stride < 4 * num_anchors ||
        !means4_host || !stds4_host || !(wh_ratio_clip > 0.0))
        return BRCNN_EINVAL;
    if (batch == 0 || count == 0) return 0;
    if (!topk_inds || !bbox_pred || !base_anchors || !proposals) return BRCNN_EINVAL;
    DecodeParams dp;
    for (int i = 0; i < 4; i++) { dp.mean[i] = means4_host[i]; dp.std[i] = stds4_host[i]; }
    dp.max_ratio = (float)fabs(log(wh_ratio_clip));
    dp.clip = (max_h > 0.f && max_w > 0.f) ? 1 : 0;
    dp.max_h = max_h; dp.max_w = max_w; dp.min_size = min_size;
    dp.pred_scale = pred_scale; dp.pred_stride = pred_stride;
    const long long total = (long long)batch * count;
    long long g = (total + 255) / 256;
    if (g > 4096) g = 4096;
    hipLaunchKernelGGL(rpn_decode_kernel, dim3((int)g), dim3(256), 0, (hipStream_t)stream,
                       topk_inds, bbox_pred, base_anchors, batch, count, height * width * num_anchors,
                       width, num_anchors, stride_w, stride_h, dp, proposals, valid);
    BRCNN_LAUNCH_CHECK();
    return 0;
}

BRCNN_API int brcnn_rcnn_decode(const float* probs, const float* bbox_pred, const float* proposals, const int32_t* num,
                                const float* max_shape, const float* scale_factor, int batch, int per_image,
                                int num_classes, float score_thr, const float* means4_host, const float* stds4_host,
                                double wh_ratio_clip, float* boxes, float* scores, int64_t* labels, uint8_t* valid,
                                void* stream) {
    if (!probs || !bbox_pred || !proposals || !num || !max_shape || !boxes || !scores || !labels || !valid ||
        batch <= 0 || per_image <= 0 || num_classes <= 0 || !means4_host || !stds4_host || !(wh_ratio_clip > 0.0))
        return BRCNN_EINVAL;
    DecodeParams dp = {};
    for (int i = 0; i < 4; i++) { dp.mean[i] = means4_host[i]; dp.std[i] = stds4_host[i]; }
    dp.max_ratio = (float)fabs(log(wh_ratio_clip));
    const long long total = (long long)batch * per_image * num_classes;
    long long g = (total + 255) / 256;
    if (g > 4096) g = 4096;
    hipLaunchKernelGGL(rcnn_decode_kernel, dim3((int)g), dim3(256), 0, (hipStream_t)stream, probs, bbox_pred, proposals,
                       num, max_shape, scale_factor, batch, per_image, num_classes, score_thr, dp, boxes, scores, labels,
                       valid);
    BRCNN_LAUNCH_CHECK();
    return 0;
}

static int decode_levels_impl(const int64_t* const* topk_inds, const float* const* bbox_pred,
                              const int* pred_strides, const float* pred_scales, const float* pred_scales_dev,
                              const float* max_shape_dev, const float* const* base_anchors, int batch, int num_levels, const int* counts,
                              const int* heights, const int* widths, int num_anchors, const int* strides_w,
                              const int* strides_h, const float* means4_host, const float* stds4_host,
                              double wh_ratio_clip, float max_h, float max_w, float min_size,
                              float* proposals, uint8_t* valid, int64_t* ids, void* stream) {
    if (batch < 0 || num_levels <= 0 || num_levels > BRCNN_MAX_LEVELS || num_anchors <= 0 || !topk_inds || !bbox_pred ||
        !pred_strides || (!pred_scales && !pred_scales_dev) || !base_anchors || !counts || !heights || !widths || !strides_w || !strides_h ||
        !means4_host || !stds4_host || !(wh_ratio_clip > 0.0) || !proposals || !valid || !ids)
        return BRCNN_EINVAL;
    DecodeLevels lv = {};
    lv.num = num_levels;
    lv.scale_dev = pred_scales_dev;
    lv.max_shape_dev = max_shape_dev;
    int T = 0;
    for (int l = 0; l < num_levels; l++) {
        if (counts[l] <= 0 || heights[l] <= 0 || widths[l] <= 0 || pred_strides[l] < 4 * num_anchors || !topk_inds[l] ||
            !bbox_pred[l] || !base_anchors[l])
            return BRCNN_EINVAL;
        lv.inds[l] = topk_inds[l]; lv.pred[l] = bbox_pred[l]; lv.base[l] = base_anchors[l];
        lv.pred_stride[l] = pred_strides[l]; lv.pred_scale[l] = pred_scales ? pred_scales[l] : 1.f;
        lv.hwA[l] = heights[l] * widths[l] * num_anchors; lv.width[l] = widths[l];
        lv.stride_w[l] = strides_w[l]; lv.stride_h[l] = strides_h[l];
        lv.col0[l] = T;
        T += counts[l];
    }
    for (int l = num_levels; l <= BRCNN_MAX_LEVELS; l++) lv.col0[l] = T;
    if (batch == 0) return 0;
    DecodeParams dp;
    for (int i = 0; i < 4; i++) { dp.mean[i] = means4_host[i]; dp.std[i] = stds4_host[i]; }
    dp.max_ratio = (float)fabs(log(wh_ratio_clip));
    dp.clip = ((max_h > 0.f && max_w > 0.f) || max_shape_dev) ? 1 : 0;
    dp.max_h = max_h; dp.max_w = max_w; dp.min_size = min_size;
    dp.pred_scale = 1.f; dp.pred_stride = 0;
    const long long total = (long long)batch * T;
    long long g = (total + 255) / 256;
    if (g > 4096) g = 4096;
    hipLaunchKernelGGL(rpn_decode_levels_kernel, dim3((int)g), dim3(256), 0, (hipStream_t)stream, lv, batch, num_anchors,
                       dp, proposals, valid, ids);
    BRCNN_LAUNCH_CHECK();
    return 0;
}


namespace {
constexpr int TOPK_SPLIT_MIN = 32768;      // levels at least this long are selected in parts
constexpr int TOPK_PART = 20000;
inline int topk_parts(int n, int k) {
    if (n < TOPK_SPLIT_MIN) return 1;
    int p = (n + TOPK_PART - 1) / TOPK_PART;
    if (p > 8) p = 8;
    while (p > 1 && (n + p - 1) / p <= k) p--;           // every part must be longer than k
    return p;
}
}  // namespace

BRCNN_API size_t brcnn_rpn_topk_workspace_bytes(const int* n_host, int num_levels, int batch, int k) {
    if (!n_host || num_levels <= 0 || batch <= 0 || k <= 0) return 0;
    size_t b = 0;
    for (int l = 0; l < num_levels; l++) {
        const int p = topk_parts(n_host[l], k);
        if (p > 1) b += (size_t)batch * p * k * (sizeof(float) + sizeof(int64_t));
    }
    return b + 256;
}

BRCNN_API int brcnn_rpn_topk(const float* const* score_levels, const int* n_host, int num_levels,
                             int batch, int k, float* const* out_score, int64_t* const* out_idx,
                             void* workspace, size_t workspace_bytes, void* stream) {
    if (num_levels <= 0 || num_levels > BRCNN_MAX_LEVELS || batch < 0 || k <= 0 || k > 4096 ||
        !score_levels || !n_host || !out_score || !out_idx)
        return BRCNN_EINVAL;
    if (batch == 0) return 0;
    if (workspace_bytes < brcnn_rpn_topk_workspace_bytes(n_host, num_levels, batch, k) || !workspace)
        return BRCNN_EINVAL;
    TopkLevels s1 = {}, s2 = {};
    int j1 = 0, j2 = 0;
    char* ws = (char*)workspace;
    for (int l = 0; l < num_levels; l++) {
        const int n = n_host[l];
        if (n <= 0 || !score_levels[l] || !out_score[l] || !out_idx[l]) return BRCNN_EINVAL;
        const int kk = n < k ? n : k;
        const int parts = topk_parts(n, k);
        if (parts == 1) {
            if (j1 >= TOPK_MAX_JOBS) return BRCNN_EINVAL;
            s1.score[j1] = score_levels[l]; s1.row_stride[j1] = n; s1.n[j1] = n;
            s1.out_score[j1] = out_score[l]; s1.out_idx[j1] = out_idx[l]; s1.out_stride[j1] = kk;
            j1++;
            continue;
        }
        float* tmp_s = (float*)ws;
        ws += (size_t)batch * parts * k * sizeof(float);
        int64_t* tmp_i = (int64_t*)ws;
        ws += (size_t)batch * parts * k * sizeof(int64_t);
        const int len = (n + parts - 1) / parts;
        for (int q = 0; q < parts; q++) {
            if (j1 >= TOPK_MAX_JOBS) return BRCNN_EINVAL;
            const int beg = q * len, cnt = (beg + len <= n ? len : n - beg);
            s1.score[j1] = score_levels[l] + beg; s1.row_stride[j1] = n; s1.n[j1] = cnt; s1.idx_add[j1] = beg;
            s1.out_score[j1] = tmp_s + (size_t)q * k; s1.out_idx[j1] = tmp_i + (size_t)q * k;
            s1.out_stride[j1] = (long long)parts * k;
            j1++;
        }
        s2.score[j2] = tmp_s; s2.row_stride[j2] = (long long)parts * k; s2.n[j2] = parts * k;
        s2.idx_map[j2] = tmp_i; s2.map_stride[j2] = (long long)parts * k;
        s2.out_score[j2] = out_score[l]; s2.out_idx[j2] = out_idx[l]; s2.out_stride[j2] = k;
        j2++;
    }
    int KP = 2;
    while (KP < k) KP <<= 1;
    hipLaunchKernelGGL(rpn_topk_kernel, dim3(batch, j1), dim3(1024), (size_t)KP * 8, (hipStream_t)stream, s1, k, KP);
    BRCNN_LAUNCH_CHECK();
    if (j2 > 0) {
        hipLaunchKernelGGL(rpn_topk_kernel, dim3(batch, j2), dim3(1024), (size_t)KP * 8, (hipStream_t)stream, s2, k,
                           KP);
        BRCNN_LAUNCH_CHECK();
    }
    return 0;
}

BRCNN_API int brcnn_rpn_decode_levels(const int64_t* const* topk_inds, const float* const* bbox_pred,
                                      const int* pred_strides, const float* pred_scales,
                                      const float* const* base_anchors, int batch, int num_levels, const int* counts,
                                      const int* heights, const int* widths, int num_anchors, const int* strides_w,
                                      const int* strides_h, const float* means4_host, const float* stds4_host,
                                      double wh_ratio_clip, float max_h, float max_w, float min_size,
                                      float* proposals, uint8_t* valid, int64_t* ids, void* stream) {
    if (!pred_scales) return BRCNN_EINVAL;
    return decode_levels_impl(topk_inds, bbox_pred, pred_strides, pred_scales, nullptr, nullptr, base_anchors, batch, num_levels,
                              counts, heights, widths, num_anchors, strides_w, strides_h, means4_host, stds4_host,
                              wh_ratio_clip, max_h, max_w, min_size, proposals, valid, ids, stream);
}

BRCNN_API int brcnn_rpn_decode_levels_dscale(const int64_t* const* topk_inds, const float* const* bbox_pred,
                                             const int* pred_strides, const float* pred_scales_dev,
                                             const float* max_shape_dev, const float* const* base_anchors, int batch,
                                             int num_levels,
                                             const int* counts, const int* heights, const int* widths, int num_anchors,
                                             const int* strides_w, const int* strides_h, const float* means4_host,
                                             const float* stds4_host, double wh_ratio_clip, float max_h, float max_w,
                                             float min_size, float* proposals, uint8_t* valid, int64_t* ids,
                                             void* stream) {
    if (!pred_scales_dev) return BRCNN_EINVAL;
    return decode_levels_impl(topk_inds, bbox_pred, pred_strides, nullptr, pred_scales_dev, max_shape_dev, base_anchors, batch,
                              num_levels, counts, heights, widths, num_anchors, strides_w, strides_h, means4_host,
                              stds4_host, wh_ratio_clip, max_h, max_w, min_size, proposals, valid, ids, stream);
}
