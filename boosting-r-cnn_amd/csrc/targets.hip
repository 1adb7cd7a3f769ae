// Whole-batch target assignment and RoI sampling of the train step, for gfx950.
//
//   assign_*   MaxIoUAssigner.assign / assign_wrt_overlaps
//              (mmdet/core/bbox/assigners/max_iou_assigner.py:61-213) on top of bbox_overlaps
//              (mmdet/core/bbox/iou_calculators/iou2d_calculator.py:75-261), for every image of the
//              batch in one or two launches.  The (num_gts x num_boxes) IoU matrix is never
//              materialised: a lane owns one box and walks the image's ground truth (LDS resident);
//              the per-gt maxima of the low-quality pass are wave-reduced and merged with integer
//              atomic max (IoUs are >= 0, so their bit patterns order like the values) and the
//              second launch recomputes the same IoUs bit for bit.  RPN anchors take their validity
//              (AnchorGenerator.valid_flags, core/anchor/anchor_generator.py:383-440 and
//              anchor_inside_flags, core/anchor/utils.py:21-47) from the level geometry instead of
//              a flag tensor; invalid anchors take no part (anchor_head.py:199-204) and come out
//              as "ignore" (-1), which is what `unmap` leaves behind for them (:252-262).
//   rcnn_sample  RandomSampler.sample (core/bbox/samplers/base_sampler.py:35-102,
//              random_sampler.py:32-82) + SamplingResult (sampling_result.py:26-55) + the prior
//              extraction of ProbRoIHead.forward_train (roi_heads/prob_roi_head.py:51-64) +
//              BBoxHead._get_target_single (bbox_heads/bbox_head.py:122-196) + bbox2roi
//              (core/bbox/transforms.py:59-78): one workgroup per image builds the ordered lists of
//              positive / negative candidates, applies the HOST-drawn permutations (torch.randperm
//              on the host keeps the reference's seeded stream), sorts the picks (`.unique()`) and
//              writes rois, labels, encoded box targets, priors and IoUs of the sampled rows.
//
// Index / integer outputs are exact; floating point follows the reference's operation order
// (-ffp-contract=off, correctly rounded division), so IoUs and thresholds compare bit for bit.
#include "common.h"
#include <float.h>

namespace {

struct GtTable {
    int off[BRCNN_MAX_IMAGES + 1];     // image b's ground truth = rows [off[b], off[b+1]) of the flat list
};

struct AnchorGeom {                    // pyramid geometry of the RPN anchors of ONE image
    int num_levels;                    // 0: plain box list, no validity test
    int start[BRCNN_MAX_LEVELS + 1];   // first anchor of each level in the per-image order
    int width[BRCNN_MAX_LEVELS];       // cells per row
    int A;                             // anchors per cell
};

struct AssignParams {
    const float* boxes;                // (B?, n, row_stride) [x1,y1,x2,y2,...]
    long long batch_stride;            // floats between images (0: every image uses the same boxes)
    int row_stride;                    // floats per box row (4 or 5)
    const int* num_boxes;              // (B) real rows per image, or NULL = n
    int n;
    const float* gts;                  // (sum G, 4)
    GtTable gt;
    AnchorGeom geom;
    const int* valid_hw;               // (B, L, 2) [valid_h, valid_w] in cells, or NULL = all valid
    const float* img_hw;               // (B, 2) image height / width for allowed_border >= 0, or NULL
    float border;
    float pos_thr, neg_lo, neg_hi, min_pos;
    int low_quality;
    unsigned* gt_max;                  // (sum G) bit patterns of the per-gt maxima (low-quality pass)
    int* gt_inds;                      // (B, n): -1 ignore / invalid, 0 negative, k > 0 matched to gt k-1
    float* max_overlaps;               // (B, n) or NULL
    int* counts;                       // (B, 2) [#positive, #negative] or NULL (zeroed by the caller)
};

// bbox_overlaps(gt, box) for one pair, the reference's operation order (iou2d_calculator.py:232-261):
// area1 = gt, area2 = box, union = area1 + area2 - overlap, max(union, eps), overlap / union
__device__ __forceinline__ float pair_iou(const float4 g, const float ga, const float4 b, const float ba) {
    const float ltx = fmaxf(g.x, b.x), lty = fmaxf(g.y, b.y);
    const float rbx = fminf(g.z, b.z), rby = fminf(g.w, b.w);
    float w = rbx - ltx, h = rby - lty;
    w = w < 0.f ? 0.f : w;
    h = h < 0.f ? 0.f : h;
    const float ov = w * h;
    float un = ga + ba - ov;
    un = fmaxf(un, 1e-6f);
    return ov / un;
}

// the same in two steps: overlap first (most pairs are disjoint and 0 / union == 0 exactly, so the
// correctly rounded division is only paid where boxes intersect)
__device__ __forceinline__ float pair_overlap(const float4 g, const float4 b) {
    const float ltx = fmaxf(g.x, b.x), lty = fmaxf(g.y, b.y);
    const float rbx = fminf(g.z, b.z), rby = fminf(g.w, b.w);
    float w = rbx - ltx, h = rby - lty;
    w = w < 0.f ? 0.f : w;
    h = h < 0.f ? 0.f : h;
    return w * h;
}
__device__ __forceinline__ float iou_from_overlap(float ov, float ga, float ba) {
    float un = ga + ba - ov;
    un = fmaxf(un, 1e-6f);
    return ov / un;
}

__device__ __forceinline__ bool box_is_valid(const AssignParams& p, int b, int i, const float4 bx) {
    if (p.num_boxes && i >= p.num_boxes[b]) return false;
    if (p.geom.num_levels > 0 && p.valid_hw) {
        int l = 0;
#pragma unroll
        for (int k = 1; k < BRCNN_MAX_LEVELS; k++)
            if (k < p.geom.num_levels && i >= p.geom.start[k]) l = k;
        const int cell = (i - p.geom.start[l]) / p.geom.A;
        const int cx = cell % p.geom.width[l], cy = cell / p.geom.width[l];
        const int* v = p.valid_hw + ((size_t)b * p.geom.num_levels + l) * 2;
        if (cy >= v[0] || cx >= v[1]) return false;
    }
    if (p.img_hw && p.border >= 0.f) {
        const float ih = p.img_hw[2 * b], iw = p.img_hw[2 * b + 1];
        if (!(bx.x >= -p.border && bx.y >= -p.border && bx.z < iw + p.border && bx.w < ih + p.border)) return false;
    }
    return true;
}

constexpr int GT_CHUNK = 256;

// PASS 0: per-gt maxima over the valid boxes (only needed by the low-quality rule).
// PASS 1: the assignment itself.
template <int PASS>
__global__ __launch_bounds__(256) void assign_kernel(const AssignParams p) {
    __shared__ float4 s_gt[GT_CHUNK];
    __shared__ float s_ga[GT_CHUNK];
    __shared__ unsigned s_gmax[GT_CHUNK];
    const int b = blockIdx.y, tid = threadIdx.x, lane = tid & 63;
    const int i = blockIdx.x * 256 + tid;
    const int g0 = p.gt.off[b], G = p.gt.off[b + 1] - g0;
    float4 bx = make_float4(0.f, 0.f, 0.f, 0.f);
    bool valid = false;
    if (i < p.n) {
        const float* src = p.boxes + (size_t)b * p.batch_stride + (size_t)i * p.row_stride;
        bx = make_float4(src[0], src[1], src[2], src[3]);
        valid = box_is_valid(p, b, i, bx);
    }
    const float ba = (bx.z - bx.x) * (bx.w - bx.y);
    float mx = -FLT_MAX;
    int arg = 0, lowq = 0;
    for (int c0 = 0; c0 < G; c0 += GT_CHUNK) {
        const int cn = min(GT_CHUNK, G - c0);
        __syncthreads();
        if (tid < cn) {
            const float4 g = *reinterpret_cast<const float4*>(p.gts + (size_t)(g0 + c0 + tid) * 4);
            s_gt[tid] = g;
            s_ga[tid] = (g.z - g.x) * (g.w - g.y);
            if (PASS == 1 && p.low_quality) s_gmax[tid] = p.gt_max[g0 + c0 + tid];
            if (PASS == 0) s_gmax[tid] = 0u;
        }
        __syncthreads();
        for (int k = 0; k < cn; k++) {
            const float ov = valid ? pair_overlap(s_gt[k], bx) : 0.f;
            if (PASS == 0) {
                // wave maximum -> one atomic per wave and gt; nothing to do while every lane is at 0
                // (gt_max starts at 0, the smallest IoU)
                if (__ballot(ov > 0.f) == 0ull) continue;         // (wave-uniform) no lane overlaps this gt
                float m = ov > 0.f ? iou_from_overlap(ov, s_ga[k], ba) : 0.f;
#pragma unroll
                for (int d = 32; d >= 1; d >>= 1) m = fmaxf(m, __shfl_xor(m, d, 64));
                if (lane == 0) atomicMax(&s_gmax[k], __float_as_uint(m));        // workgroup maximum in LDS
            } else {
                const float iou = !valid ? -1.f : (ov > 0.f ? iou_from_overlap(ov, s_ga[k], ba) : 0.f);
                if (iou > mx) { mx = iou; arg = c0 + k; }      // first maximum, as torch.max(dim=0) on the host
                if (p.low_quality) {
                    const float gm = __uint_as_float(s_gmax[k]);
                    if (gm >= p.min_pos && iou == gm) lowq = c0 + k + 1;     // later gts override earlier ones
                }
            }
        }
        if (PASS == 0) {
            // one global atomic per workgroup and gt, and only when it can still raise the maximum (a few
            // hundred addresses take every update of the launch: unconditional atomics serialise on them)
            __syncthreads();
            if (tid < cn) {
                const unsigned v = s_gmax[tid];
                if (v > __atomic_load_n(p.gt_max + g0 + c0 + tid, __ATOMIC_RELAXED)) atomicMax(p.gt_max + g0 + c0 + tid, v);
            }
        }
    }
    if (PASS == 1) {
        int a = -1;
        float mo = 0.f;
        if (valid) {
            if (G == 0) {
                a = 0;                                           // no ground truth: everything is background
            } else {
                mo = mx;
                if (mx >= p.neg_lo && mx < p.neg_hi) a = 0;
                if (mx >= p.pos_thr) a = arg + 1;
                if (lowq) a = lowq;
            }
        }
        if (i < p.n) {
            p.gt_inds[(size_t)b * p.n + i] = a;
            if (p.max_overlaps) p.max_overlaps[(size_t)b * p.n + i] = mo;
        }
        if (p.counts) {
            const unsigned long long mp = __ballot(a > 0), mn = __ballot(a == 0);
            if (lane == 0) {
                if (mp) atomicAdd(p.counts + 2 * b, __popcll(mp));
                if (mn) atomicAdd(p.counts + 2 * b + 1, __popcll(mn));
            }
        }
    }
}

// ------------------------------------------------------------------------------------------------
struct SampleParams {
    const float* props;        // (B, K, 5) [x1,y1,x2,y2,score] zero padded
    const int* num_props;      // (B)
    int K;
    const int* gt_inds;        // (B, K) from the assignment
    const float* max_overlaps; // (B, K) or NULL (only for `ious`)
    const float* gts;          // (sum G, 4)
    const long long* gt_labels;// (sum G)
    GtTable gt;
    int add_gt;                // add_gt_as_proposals
    int num, num_pos;          // sampler.num, int(num * pos_fraction)
    float neg_pos_ub;          // < 0: unbounded
    const int* perm;           // (B, num_pos + num) host-drawn picks: [0,num_pos) positives, then negatives
    int row0[BRCNN_MAX_IMAGES + 1];
    int num_classes;
    int reg_decoded;
    float mean[4], std[4];
    int* lists;                // (B, 2, K + Gmax) scratch: ordered positive / negative candidate indices
    int list_stride;
    float* rois;               // (N, 5)
    long long* labels;         // (N)
    float* bbox_targets;       // (N, 4)
    float* priors;             // (N)
    float* ious;               // (N) or NULL
    int* pos_flags;            // (N) 1 = positive row, or NULL
};

constexpr int SAMPLE_MAX = 2048;       // sampler.num is at most this

__device__ __forceinline__ void bitonic_sort_int(int* v, int P, int tid, int nthreads) {
    for (int k = 2; k <= P; k <<= 1) {
        for (int j = k >> 1; j > 0; j >>= 1) {
            for (int t = tid; t < (P >> 1); t += nthreads) {
                const int lo = ((t & ~(j - 1)) << 1) | (t & (j - 1));
                const int hi = lo | j;
                const int a = v[lo], c = v[hi];
                const bool up = (lo & k) == 0;
                if ((a > c) == up) { v[lo] = c; v[hi] = a; }
            }
            __syncthreads();
        }
    }
}

__global__ __launch_bounds__(1024) void rcnn_sample_kernel(const SampleParams p) {
    __shared__ int wsum_p[16], wsum_n[16];
    __shared__ int s_base[2];
    __shared__ int s_pos[SAMPLE_MAX], s_neg[SAMPLE_MAX];
    const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int g0 = p.gt.off[b], G = p.gt.off[b + 1] - g0;
    const int Gadd = (p.add_gt && G > 0) ? G : 0;
    const int nprops = p.num_props[b];
    const int nall = Gadd + nprops;
    int* list_p = p.lists + (size_t)b * 2 * p.list_stride;
    int* list_n = list_p + p.list_stride;
    const int* gi_row = p.gt_inds + (size_t)b * p.K;
    if (tid < 2) s_base[tid] = 0;
    __syncthreads();
    // ordered lists of the positive / negative candidates (torch.nonzero order)
    for (int t0 = 0; t0 < nall; t0 += 1024) {
        const int t = t0 + tid;
        int gi = -1;
        if (t < nall) gi = t < Gadd ? t + 1 : gi_row[t - Gadd];
        const int vp = gi > 0 ? 1 : 0, vn = gi == 0 ? 1 : 0;
        int ip = vp, in = vn;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
            const int op = __shfl_up(ip, d, 64), on = __shfl_up(in, d, 64);
            if (lane >= d) { ip += op; in += on; }
        }
        if (lane == 63) { wsum_p[wave] = ip; wsum_n[wave] = in; }
        __syncthreads();
        int offp = 0, offn = 0, totp = 0, totn = 0;
        for (int w = 0; w < 16; w++) {
            if (w < wave) { offp += wsum_p[w]; offn += wsum_n[w]; }
            totp += wsum_p[w];
            totn += wsum_n[w];
        }
        if (vp) list_p[s_base[0] + offp + ip - 1] = t;
        if (vn) list_n[s_base[1] + offn + in - 1] = t;
        __syncthreads();
        if (tid == 0) { s_base[0] += totp; s_base[1] += totn; }
        __syncthreads();
    }
    const int npos = s_base[0], nneg = s_base[1];
    const int spos = min(npos, p.num_pos);
    int eneg = p.num - spos;
    if (p.neg_pos_ub >= 0.f) {
        const int ub = (int)(p.neg_pos_ub * (float)max(1, spos));
        eneg = min(eneg, ub);
    }
    const int sneg = min(nneg, eneg);
    const int* perm_p = p.perm + (size_t)b * (p.num_pos + p.num);
    const int* perm_n = perm_p + p.num_pos;
    int Pp = 1, Pn = 1;
    while (Pp < spos) Pp <<= 1;
    while (Pn < sneg) Pn <<= 1;
    __threadfence_block();
    for (int j = tid; j < Pp; j += 1024)
        s_pos[j] = j < spos ? list_p[npos > p.num_pos ? min(perm_p[j], npos - 1) : j] : 0x7fffffff;
    for (int j = tid; j < Pn; j += 1024)
        s_neg[j] = j < sneg ? list_n[nneg > eneg ? min(perm_n[j], nneg - 1) : j] : 0x7fffffff;
    __syncthreads();
    if (npos > p.num_pos) bitonic_sort_int(s_pos, Pp, tid, 1024);      // `.unique()`: ascending
    if (nneg > eneg) bitonic_sort_int(s_neg, Pn, tid, 1024);
    __syncthreads();
    const int r0 = p.row0[b], rows = p.row0[b + 1] - r0;
    const float* prow = p.props + (size_t)b * p.K * 5;
    for (int j = tid; j < spos + sneg && j < rows; j += 1024) {
        const bool is_pos = j < spos;
        const int idx = is_pos ? s_pos[j] : s_neg[j - spos];
        float4 bx;
        int gi;
        float mo;
        if (idx < Gadd) {
            bx = *reinterpret_cast<const float4*>(p.gts + (size_t)(g0 + idx) * 4);
            gi = idx + 1;
            mo = 1.f;
        } else {
            const float* s = prow + (size_t)(idx - Gadd) * 5;
            bx = make_float4(s[0], s[1], s[2], s[3]);
            gi = gi_row[idx - Gadd];
            mo = p.max_overlaps ? p.max_overlaps[(size_t)b * p.K + idx - Gadd] : 0.f;
        }
        const size_t r = (size_t)(r0 + j);
        float* ro = p.rois + r * 5;
        ro[0] = (float)b; ro[1] = bx.x; ro[2] = bx.y; ro[3] = bx.z; ro[4] = bx.w;
        float4 tg = make_float4(0.f, 0.f, 0.f, 0.f);
        long long lab = p.num_classes;
        if (is_pos) {
            const float4 g = *reinterpret_cast<const float4*>(p.gts + (size_t)(g0 + gi - 1) * 4);
            lab = p.gt_labels[g0 + gi - 1];
            if (p.reg_decoded) {
                tg = g;
            } else {                                             // bbox2delta (delta_xywh_bbox_coder.py:99-141)
                const float px = (bx.x + bx.z) * 0.5f, py = (bx.y + bx.w) * 0.5f;
                const float pw = bx.z - bx.x, ph = bx.w - bx.y;
                const float gx = (g.x + g.z) * 0.5f, gy = (g.y + g.w) * 0.5f;
                const float gw = g.z - g.x, gh = g.w - g.y;
                tg.x = ((gx - px) / pw - p.mean[0]) / p.std[0];
                tg.y = ((gy - py) / ph - p.mean[1]) / p.std[1];
                tg.z = (logf(gw / pw) - p.mean[2]) / p.std[2];
                tg.w = (logf(gh / ph) - p.mean[3]) / p.std[3];
            }
        }
        p.labels[r] = lab;
        *reinterpret_cast<float4*>(p.bbox_targets + r * 4) = tg;
        // prior (prob_roi_head.py:51-64): rows [0, G) are taken to be the ground truth itself (0), the
        // rest index the proposal list at (sampled index - G), python-style wrap-around included
        float prior = 0.f;
        if (!(is_pos && j < G)) {
            int q = idx - G;
            if (q < 0) q += nprops;
            const float s = (q >= 0 && q < nprops) ? prow[(size_t)q * 5 + 4] : 0.f;
            prior = is_pos ? s : 1.f - s;
        }
        p.priors[r] = prior;
        if (p.ious) p.ious[r] = is_pos ? mo : 1.f - mo;
        if (p.pos_flags) p.pos_flags[r] = is_pos ? 1 : 0;
    }
}

// bbox_overlaps (iou2d_calculator.py:75-261) as a table: out[i, j] (or out[i] when aligned) for modes
// 0 iou / 1 iof / 2 giou, the reference's operation order
__global__ __launch_bounds__(256) void bbox_overlaps_kernel(const float* __restrict__ b1, int s1, int n1,
                                                           const float* __restrict__ b2, int s2, int n2, int mode,
                                                           int aligned, float eps, float* __restrict__ out) {
    const long long total = aligned ? n1 : (long long)n1 * n2;
    for (long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x; idx < total;
         idx += (long long)gridDim.x * blockDim.x) {
        const int i = aligned ? (int)idx : (int)(idx / n2), j = aligned ? (int)idx : (int)(idx - (long long)i * n2);
        const float* p = b1 + (size_t)i * s1;
        const float* q = b2 + (size_t)j * s2;
        const float4 a = make_float4(p[0], p[1], p[2], p[3]), b = make_float4(q[0], q[1], q[2], q[3]);
        const float a1 = (a.z - a.x) * (a.w - a.y), a2 = (b.z - b.x) * (b.w - b.y);
        float w = fminf(a.z, b.z) - fmaxf(a.x, b.x), h = fminf(a.w, b.w) - fmaxf(a.y, b.y);
        w = w < 0.f ? 0.f : w;
        h = h < 0.f ? 0.f : h;
        const float ov = w * h;
        float un = mode == 1 ? a1 : a1 + a2 - ov;
        un = fmaxf(un, eps);
        float r = ov / un;
        if (mode == 2) {
            float ew = fmaxf(a.z, b.z) - fminf(a.x, b.x), eh = fmaxf(a.w, b.w) - fminf(a.y, b.y);
            ew = ew < 0.f ? 0.f : ew;
            eh = eh < 0.f ? 0.f : eh;
            const float ea = fmaxf(ew * eh, eps);
            r = r - (ea - un) / ea;
        }
        out[idx] = r;
    }
}

int fill_gt_table(GtTable& t, const int* gt_offsets_host, int batch) {
    if (!gt_offsets_host || batch <= 0 || batch > BRCNN_MAX_IMAGES) return BRCNN_EINVAL;
    for (int b = 0; b <= batch; b++) {
        t.off[b] = gt_offsets_host[b];
        if (b && t.off[b] < t.off[b - 1]) return BRCNN_EINVAL;
    }
    return 0;
}

}  // namespace

BRCNN_API int brcnn_bbox_overlaps(const float* bboxes1, int stride1, int n1, const float* bboxes2, int stride2, int n2,
                                  int mode, int is_aligned, float eps, float* out, void* stream) {
    if (n1 < 0 || n2 < 0 || mode < 0 || mode > 2 || stride1 < 4 || stride2 < 4 || (is_aligned && n1 != n2))
        return BRCNN_EINVAL;
    const long long total = is_aligned ? n1 : (long long)n1 * n2;
    if (total == 0) return 0;
    if (!bboxes1 || !bboxes2 || !out) return BRCNN_EINVAL;
    long long g = (total + 255) / 256;
    if (g > 8192) g = 8192;
    hipLaunchKernelGGL(bbox_overlaps_kernel, dim3((int)g), dim3(256), 0, (hipStream_t)stream, bboxes1, stride1, n1,
                       bboxes2, stride2, n2, mode, is_aligned ? 1 : 0, eps, out);
    BRCNN_LAUNCH_CHECK();
    return 0;
}

BRCNN_API int brcnn_assign_max_iou(const float* boxes, int64_t box_batch_stride, int box_row_stride,
                                   const int32_t* num_boxes, int n, int batch, const float* gts,
                                   const int* gt_offsets_host, int num_levels, const int* level_start_host,
                                   const int* level_width_host, int anchors_per_cell, const int32_t* valid_hw,
                                   const float* img_hw, float allowed_border, float pos_iou_thr, float neg_iou_lo,
                                   float neg_iou_hi, float min_pos_iou, int match_low_quality, uint32_t* gt_max_ws,
                                   int32_t* gt_inds, float* max_overlaps, int32_t* counts, void* stream) {
    if (!boxes || !gt_inds || n <= 0 || box_row_stride < 4 || num_levels < 0 || num_levels > BRCNN_MAX_LEVELS)
        return BRCNN_EINVAL;
    AssignParams p;
    if (int st = fill_gt_table(p.gt, gt_offsets_host, batch)) return st;
    const int total_gt = p.gt.off[batch];
    if (total_gt > 0 && !gts) return BRCNN_EINVAL;
    if (match_low_quality && total_gt > 0 && !gt_max_ws) return BRCNN_EINVAL;
    p.boxes = boxes; p.batch_stride = box_batch_stride; p.row_stride = box_row_stride;
    p.num_boxes = num_boxes; p.n = n; p.gts = gts;
    p.geom.num_levels = num_levels;
    p.geom.A = anchors_per_cell > 0 ? anchors_per_cell : 1;
    for (int l = 0; l < num_levels; l++) {
        if (!level_start_host || !level_width_host || level_width_host[l] <= 0) return BRCNN_EINVAL;
        p.geom.start[l] = level_start_host[l];
        p.geom.width[l] = level_width_host[l];
    }
    if (num_levels) p.geom.start[num_levels] = level_start_host[num_levels];
    p.valid_hw = num_levels ? valid_hw : nullptr;
    p.img_hw = img_hw; p.border = allowed_border;
    p.pos_thr = pos_iou_thr; p.neg_lo = neg_iou_lo; p.neg_hi = neg_iou_hi; p.min_pos = min_pos_iou;
    p.low_quality = match_low_quality ? 1 : 0;
    p.gt_max = gt_max_ws; p.gt_inds = gt_inds; p.max_overlaps = max_overlaps; p.counts = counts;
    hipStream_t s = (hipStream_t)stream;
    const dim3 grid(brcnn_cdiv(n, 256), batch);
    if (counts) BRCNN_HIP_CHECK(hipMemsetAsync(counts, 0, sizeof(int32_t) * 2 * batch, s));
    if (p.low_quality && total_gt > 0) {
        BRCNN_HIP_CHECK(hipMemsetAsync(gt_max_ws, 0, sizeof(uint32_t) * total_gt, s));
        hipLaunchKernelGGL(assign_kernel<0>, grid, dim3(256), 0, s, p);
        BRCNN_LAUNCH_CHECK();
    }
    hipLaunchKernelGGL(assign_kernel<1>, grid, dim3(256), 0, s, p);
    BRCNN_LAUNCH_CHECK();
    return 0;
}

BRCNN_API int brcnn_rcnn_sample(const float* proposals, const int32_t* num_props, int per_image, int batch,
                                const int32_t* gt_inds, const float* max_overlaps, const float* gts,
                                const int64_t* gt_labels, const int* gt_offsets_host, int add_gt_as_proposals,
                                int num, int num_expected_pos, float neg_pos_ub, const int32_t* perm,
                                const int* row_offsets_host, int num_classes, int reg_decoded_bbox,
                                const float* means4_host, const float* stds4_host, int32_t* list_ws,
                                int list_stride, float* rois, int64_t* labels, float* bbox_targets, float* priors,
                                float* ious, int32_t* pos_flags, void* stream) {
    if (!proposals || !num_props || !gt_inds || !perm || !row_offsets_host || !list_ws || !rois || !labels ||
        !bbox_targets || !priors || per_image <= 0 || num <= 0 || num > SAMPLE_MAX || num_expected_pos < 0 ||
        num_expected_pos > num || !means4_host || !stds4_host)
        return BRCNN_EINVAL;
    SampleParams p;
    if (int st = fill_gt_table(p.gt, gt_offsets_host, batch)) return st;
    int gmax = 0;
    for (int b = 0; b < batch; b++) gmax = p.gt.off[b + 1] - p.gt.off[b] > gmax ? p.gt.off[b + 1] - p.gt.off[b] : gmax;
    if (p.gt.off[batch] > 0 && (!gts || !gt_labels)) return BRCNN_EINVAL;
    if (list_stride < per_image + gmax) return BRCNN_EINVAL;
    for (int b = 0; b <= batch; b++) p.row0[b] = row_offsets_host[b];
    p.props = proposals; p.num_props = num_props; p.K = per_image; p.gt_inds = gt_inds; p.max_overlaps = max_overlaps;
    p.gts = gts; p.gt_labels = (const long long*)gt_labels; p.add_gt = add_gt_as_proposals ? 1 : 0;
    p.num = num; p.num_pos = num_expected_pos; p.neg_pos_ub = neg_pos_ub; p.perm = perm;
    p.num_classes = num_classes; p.reg_decoded = reg_decoded_bbox ? 1 : 0;
    for (int k = 0; k < 4; k++) { p.mean[k] = means4_host[k]; p.std[k] = stds4_host[k]; }
    p.lists = list_ws; p.list_stride = list_stride;
    p.rois = rois; p.labels = (long long*)labels; p.bbox_targets = bbox_targets; p.priors = priors; p.ious = ious;
    p.pos_flags = pos_flags;
    hipLaunchKernelGGL(rcnn_sample_kernel, dim3(batch), dim3(1024), 0, (hipStream_t)stream, p);
    BRCNN_LAUNCH_CHECK();
    return 0;
}
