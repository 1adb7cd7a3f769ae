// Soft-NMS (segmented) for gfx950.
//
// Replaces mmcv.ops.soft_nms (ext `softnms`; CPU-only in mmcv: CUDA tensors are copied to
// the host), selected by `nms=dict(type='soft_nms', iou_threshold=0.7, min_score=0.0)` in
// configs/boosting_rcnn/boosting_rcnn_r2_101_dcn_pafpn_mstrain_3x_coco.py:27 and reached
// through mmdet/core/post_processing/bbox_nms.py:86.
//
// The algorithm (softnms_cpu) is a chain of dependent steps: pick the current max (first
// position on ties), swap it to the front, decay every remaining score, discard scores below
// min_score by swap-with-last.  Parallelism exists only across segments ((image, class)
// problems) and inside one step, so: one 256-thread workgroup per segment; per step a
// block-wide arg-max, a parallel decay sweep and -- only when something fell below min_score
// -- an exact parallel emulation of the sequential swap-with-last compaction (holes in
// ascending position are filled by the survivors of the tail in descending position), so
// that positions, and with them the tie-breaking of later arg-max steps, match the
// sequential reference bit for bit.  Latency-bound by construction; reported as such.
#include "common.h"

namespace {

constexpr int TPB = 256;

struct Best { float s; int pos; };

__device__ __forceinline__ Best better(Best a, Best b) {
    // larger score wins; equal scores: smaller position (the reference scans left to right
    // with a strict `max_score < sc[pos]`)
    if (b.s > a.s || (b.s == a.s && b.pos < a.pos)) return b;
    return a;
}

__device__ __forceinline__ Best wave_best(Best v) {
    for (int off = 32; off > 0; off >>= 1) {
        Best o;
        o.s = __shfl_xor(v.s, off);
        o.pos = __shfl_xor(v.pos, off);
        v = better(v, o);
    }
    return v;
}

__device__ __forceinline__ int block_sum(int v, int* sh) {
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off);
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    __syncthreads();
    if (lane == 0) sh[wave] = v;
    __syncthreads();
    return sh[0] + sh[1] + sh[2] + sh[3];
}

// exclusive prefix sum across the block of one int per thread; returns (exclusive, total)
__device__ __forceinline__ int block_excl_scan(int v, int* sh, int* total) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    int incl = v;
    for (int off = 1; off < 64; off <<= 1) {
        int t = __shfl_up(incl, off);
        if (lane >= off) incl += t;
    }
    __syncthreads();
    if (lane == 63) sh[wave] = incl;
    __syncthreads();
    int base = 0;
    for (int w = 0; w < wave; w++) base += sh[w];
    *total = sh[0] + sh[1] + sh[2] + sh[3];
    return base + incl - v;
}

__global__ __launch_bounds__(TPB) void softnms_kernel(
    const float* __restrict__ boxes, const float* __restrict__ scores,
    const int32_t* __restrict__ seg_begin, const int32_t* __restrict__ seg_end,
    float* __restrict__ x1, float* __restrict__ y1,
    float* __restrict__ x2, float* __restrict__ y2, float* __restrict__ sc, float* __restrict__ ar,
    int32_t* __restrict__ id, int32_t* __restrict__ holes, int32_t* __restrict__ donors,
    float* __restrict__ dets, int64_t* __restrict__ inds, int32_t* __restrict__ num_keep,
    float thr, float sigma, float min_score, int method, int offset) {
    __shared__ int sh_i[4];
    __shared__ Best sh_b[4];
    __shared__ float sh_box[5];
    const int seg = blockIdx.x, tid = threadIdx.x;
    const int beg = seg_begin[seg];
    int nboxes = seg_end[seg] - beg;
    x1 += beg; y1 += beg; x2 += beg; y2 += beg; sc += beg; ar += beg; id += beg;
    holes += beg; donors += beg;
    for (int p = tid; p < nboxes; p += TPB) {
        const float4 b = *reinterpret_cast<const float4*>(boxes + (size_t)(beg + p) * 4);
        x1[p] = b.x; y1[p] = b.y; x2[p] = b.z; y2[p] = b.w;
        sc[p] = scores[beg + p];
        ar[p] = (b.z - b.x + offset) * (b.w - b.y + offset);
        id[p] = beg + p;
    }
    __syncthreads();

    // arg-max over [from, nboxes): block-wide, first position on ties
    auto block_argmax = [&](int from) -> Best {
        Best bst; bst.s = -INFINITY; bst.pos = 0x7fffffff;
        for (int p = from + tid; p < nboxes; p += TPB) {
            Best c; c.s = sc[p]; c.pos = p;
            bst = better(bst, c);
        }
        bst = wave_best(bst);
        __syncthreads();
        if ((tid & 63) == 0) sh_b[tid >> 6] = bst;
        __syncthreads();
        return better(better(sh_b[0], sh_b[1]), better(sh_b[2], sh_b[3]));
    };
    Best m = block_argmax(0);
    for (int i = 0; i < nboxes; i++) {
        // ---- 1. the pick: `m` = arg-max over [i, nboxes), known from the previous sweep ------------
        // the reference starts from position i and only moves on a strictly larger score;
        // NaN / -inf corner: nothing compares greater -> position i
        const int max_pos = (m.pos == 0x7fffffff || !(m.s > sc[i])) ? i : m.pos;
        __syncthreads();
        // ---- 2. swap to the front, emit --------------------------------------------------
        if (tid == 0) {
            const float ix1 = x1[max_pos], iy1 = y1[max_pos], ix2 = x2[max_pos], iy2 = y2[max_pos];
            const float isc = sc[max_pos], iar = ar[max_pos];
            const int iid = id[max_pos];
            x1[max_pos] = x1[i]; y1[max_pos] = y1[i]; x2[max_pos] = x2[i]; y2[max_pos] = y2[i];
            sc[max_pos] = sc[i]; ar[max_pos] = ar[i]; id[max_pos] = id[i];
            x1[i] = ix1; y1[i] = iy1; x2[i] = ix2; y2[i] = iy2; sc[i] = isc; ar[i] = iar; id[i] = iid;
            float* d = dets + (size_t)(beg + i) * 5;
            d[0] = ix1; d[1] = iy1; d[2] = ix2; d[3] = iy2; d[4] = isc;
            inds[beg + i] = (int64_t)iid;
            sh_box[0] = ix1; sh_box[1] = iy1; sh_box[2] = ix2; sh_box[3] = iy2; sh_box[4] = iar;
        }
        __syncthreads();
        const float ix1 = sh_box[0], iy1 = sh_box[1], ix2 = sh_box[2], iy2 = sh_box[3],
                    iarea = sh_box[4];
        // ---- 3. decay sweep, fused with the arg-max of the next pick ------------------------
        int dead = 0;
        Best nxt; nxt.s = -INFINITY; nxt.pos = 0x7fffffff;
        for (int p = i + 1 + tid; p < nboxes; p += TPB) {
            const float xx1 = fmaxf(ix1, x1[p]), yy1 = fmaxf(iy1, y1[p]);
            const float xx2 = fminf(ix2, x2[p]), yy2 = fminf(iy2, y2[p]);
            const float w = fmaxf(0.f, xx2 - xx1 + offset), h = fmaxf(0.f, yy2 - yy1 + offset);
            const float inter = w * h;
            const float ovr = inter / (iarea + ar[p] - inter);
            float weight = 1.f;
            if (method == 0) { if (ovr >= thr) weight = 0.f; }
            else if (method == 1) { if (ovr >= thr) weight = 1.f - ovr; }
            else if (method == 2) { weight = expf(-(ovr * ovr) / sigma); }
            const float s = sc[p] * weight;
            sc[p] = s;
            dead += (s < min_score) ? 1 : 0;
            Best c; c.s = s; c.pos = p;
            nxt = better(nxt, c);
        }
        // one reduction for both: dead count and next arg-max
        nxt = wave_best(nxt);
        for (int off = 32; off > 0; off >>= 1) dead += __shfl_xor(dead, off);
        __syncthreads();
        if ((tid & 63) == 0) { sh_b[tid >> 6] = nxt; sh_i[tid >> 6] = dead; }
        __syncthreads();
        m = better(better(sh_b[0], sh_b[1]), better(sh_b[2], sh_b[3]));
        const int ndead = sh_i[0] + sh_i[1] + sh_i[2] + sh_i[3];
        if (ndead == 0) continue;   // uniform
        // ---- 4. swap-with-last compaction, emulated exactly --------------------------------
        const int tail = nboxes - (i + 1);
        const int alive_total = tail - ndead;
        const int B = i + 1 + alive_total;       // new nboxes
        // contiguous chunk per thread over [i+1, nboxes)
        const int per = (tail + TPB - 1) / TPB;
        const int lo = i + 1 + tid * per, hi = min(nboxes, lo + per);
        int nh = 0, nd = 0;
        for (int p = lo; p < hi; p++) {
            const bool dd = sc[p] < min_score;
            if (p < B) nh += dd ? 1 : 0;
            else nd += dd ? 0 : 1;
        }
        int tot_h, tot_d;
        const int hbase = block_excl_scan(nh, sh_i, &tot_h);
        const int dbase = block_excl_scan(nd, sh_i, &tot_d);
        // holes ascending; donors ranked from the end (descending position)
        int hr = hbase, dr = dbase;
        for (int p = lo; p < hi; p++) {
            const bool dd = sc[p] < min_score;
            if (p < B) { if (dd) holes[hr++] = p; }
            else if (!dd) { donors[tot_d - 1 - dr] = p; dr++; }
        }
        __syncthreads();
        for (int r = tid; r < tot_h; r += TPB) {
            const int p = holes[r], q = donors[r];
            x1[p] = x1[q]; y1[p] = y1[q]; x2[p] = x2[q]; y2[p] = y2[q];
            sc[p] = sc[q]; ar[p] = ar[q]; id[p] = id[q];
        }
        nboxes = B;
        __syncthreads();
        m = block_argmax(i + 1);        // the compaction moved boxes: positions of the maximum changed
    }
    if (tid == 0) num_keep[seg] = nboxes;
}

inline size_t align_up(size_t x) { return (x + 255) & ~(size_t)255; }

}  // namespace

BRCNN_API size_t brcnn_softnms_workspace_bytes(int64_t n, int num_segments) {
    if (n <= 0) return 256;
    return 9 * align_up((size_t)n * 4);
}

BRCNN_API int brcnn_softnms(const float* boxes, const float* scores, const int32_t* seg_begin,
                            const int32_t* seg_end, int num_segments, int64_t n, float iou_threshold, float sigma,
                            float min_score, int method, int offset, float* dets, int64_t* inds,
                            int32_t* num_keep, void* workspace, size_t workspace_bytes,
                            void* stream) {
    if (n < 0 || num_segments <= 0 || method < 0 || method > 2 || (offset != 0 && offset != 1) ||
        !seg_begin || !seg_end || !num_keep)
        return BRCNN_EINVAL;
    hipStream_t s = (hipStream_t)stream;
    if (n == 0) {
        BRCNN_HIP_CHECK(hipMemsetAsync(num_keep, 0, sizeof(int32_t) * num_segments, s));
        return 0;
    }
    if (!boxes || !scores || !dets || !inds || !workspace || n > 0x7fffffffLL) return BRCNN_EINVAL;
    if (workspace_bytes < brcnn_softnms_workspace_bytes(n, num_segments)) return BRCNN_EINVAL;
    char* p = (char*)workspace;
    const size_t st = align_up((size_t)n * 4);
    float* x1 = (float*)(p + 0 * st); float* y1 = (float*)(p + 1 * st);
    float* x2 = (float*)(p + 2 * st); float* y2 = (float*)(p + 3 * st);
    float* sc = (float*)(p + 4 * st); float* ar = (float*)(p + 5 * st);
    int32_t* id = (int32_t*)(p + 6 * st);
    int32_t* holes = (int32_t*)(p + 7 * st);
    int32_t* donors = (int32_t*)(p + 8 * st);
    hipLaunchKernelGGL(softnms_kernel, dim3(num_segments), dim3(TPB), 0, s, boxes, scores,
                       seg_begin, seg_end, x1, y1, x2, y2, sc, ar, id, holes, donors, dets, inds, num_keep,
                       iou_threshold, sigma, min_score, method, offset);
    BRCNN_LAUNCH_CHECK();
    return 0;
}
