// Greedy NMS (segmented) for gfx950.
//
// Replaces mmcv.ops.nms (ext `nms`) reached through mmcv batched_nms at
// mmdet/models/dense_heads/atss_rpn_head.py:756, rpn_head.py:245 and
// mmdet/core/post_processing/bbox_nms.py:86.  Per segment the semantics are mmcv's
// nms_cpu: order by score descending (ties: ascending original index), greedy, suppress
// when inter / (area_i + area_j - inter) > thr evaluated in fp32 in exactly that operation
// order (file compiled with -ffp-contract=off, correctly rounded division).
//
// Pipeline (all on the caller's stream, no host synchronisation):
//   1. rocPRIM segmented radix sort (stable, descending) of (score, index) pairs;
//   2. gather boxes into sorted order + areas;
//   3. 64x64 IoU tiles -> bit mask, one wavefront per tile (wave64 == one 64-bit word per
//      lane), upper triangle only;
//   4. one wavefront per segment walks the mask 64 rows at a time: the in-chunk dependency
//      is resolved with wave-uniform scalar ops on the diagonal words (v_readlane), the
//      rows of the survivors are OR-ed into the LDS-resident `removed` vector.
#include <cstdlib>
#include "common.h"
#include <cstring>
#include <rocprim/rocprim.hpp>

namespace {

// both buffers: positions outside every segment are never written by the segmented sort,
// so the output index buffer must hold valid indices there too
__global__ void iota_kernel(int32_t* __restrict__ v, int32_t* __restrict__ v2, int64_t n) {
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) { v[i] = (int32_t)i; v2[i] = (int32_t)i; }
}

__global__ void gather_boxes_kernel(const float* __restrict__ boxes,
                                    const int32_t* __restrict__ sorted_idx,
                                    float* __restrict__ sboxes, float* __restrict__ sareas,
                                    int64_t n, int offset) {
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float4 b = *reinterpret_cast<const float4*>(boxes + (size_t)sorted_idx[i] * 4);
    *reinterpret_cast<float4*>(sboxes + (size_t)i * 4) = b;
    sareas[i] = (b.z - b.x + offset) * (b.w - b.y + offset);
}

// grid: (col_tile, row_tile, segment); block: 64 threads (one wavefront).
__global__ __launch_bounds__(64) void nms_mask_kernel(const float* __restrict__ sboxes,
                                                      const float* __restrict__ sareas,
                                                      const int32_t* __restrict__ seg_begin,
                                                      const int32_t* __restrict__ seg_end,
                                                      unsigned long long* __restrict__ mask,
                                                      int words, float thr, int offset) {
    const int col_t = blockIdx.x, row_t = blockIdx.y, seg = blockIdx.z;
    if (col_t < row_t) return;
    const int beg = seg_begin[seg], len = seg_end[seg] - beg;
    if (row_t * 64 >= len || col_t * 64 >= len) return;
    __shared__ float4 cb[64];
    __shared__ float ca[64];
    const int lane = threadIdx.x;
    const int cj = col_t * 64 + lane;
    if (cj < len) {
        cb[lane] = *reinterpret_cast<const float4*>(sboxes + (size_t)(beg + cj) * 4);
        ca[lane] = sareas[beg + cj];
    }
    __syncthreads();
    const int ri = row_t * 64 + lane;
    if (ri >= len) return;
    const float4 a = *reinterpret_cast<const float4*>(sboxes + (size_t)(beg + ri) * 4);
    const float aarea = sareas[beg + ri];
    const int ncol = min(64, len - col_t * 64);
    unsigned long long bits = 0ull;
    const int jstart = (row_t == col_t) ? lane + 1 : 0;
    for (int j = jstart; j < ncol; j++) {
        const float4 b = cb[j];
        const float xx1 = fmaxf(a.x, b.x), yy1 = fmaxf(a.y, b.y);
        const float xx2 = fminf(a.z, b.z), yy2 = fminf(a.w, b.w);
        const float w = fmaxf(0.f, xx2 - xx1 + offset), h = fmaxf(0.f, yy2 - yy1 + offset);
        const float inter = w * h;
        const float ovr = inter / (aarea + ca[j] - inter);
        if (ovr > thr) bits |= 1ull << j;
    }
    mask[(size_t)(beg + ri) * words + col_t] = bits;
}

// One 256-thread workgroup per segment.  Wave 0 resolves the greedy chain of the current
// 64-row chunk (wave-uniform scalar walk over the diagonal mask words) and publishes the
// survivor bits; then every thread ORs the survivors' words of its own mask columns into the
// LDS-resident `removed` vector (coalesced across the workgroup, 16 loads in flight per thread).
__global__ __launch_bounds__(256) void nms_reduce_kernel(const unsigned long long* __restrict__ mask,
                                                         const int32_t* __restrict__ sorted_idx,
                                                         const int32_t* __restrict__ seg_begin,
                                                         const int32_t* __restrict__ seg_end,
                                                         int64_t* __restrict__ keep,
                                                         int32_t* __restrict__ num_keep, int words,
                                                         int max_keep) {
    extern __shared__ unsigned long long remv[];     // [words] + 1 word for the survivor bits
    __shared__ int sh_count;
    const int seg = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int beg = seg_begin[seg], len = seg_end[seg] - beg;
    const int nchunk = (len + 63) >> 6;
    unsigned long long* kept_sh = remv + words;
    for (int w = tid; w < nchunk; w += 256) remv[w] = 0ull;
    if (tid == 0) sh_count = 0;
    __syncthreads();
    int count = 0;
    // the diagonal word of the NEXT chunk does not depend on `removed`: fetched one chunk ahead
    // (handed from `diag_next` to `diag_cur` through a register move AFTER its wait, so that the use inside the chain
    // is not a pending load for the compiler's wait-count pass -- it would drain the row prefetch there otherwise)
    unsigned long long diag_next = 0ull, diag_cur = 0ull;
    if (wave == 0 && lane < len) diag_next = mask[(size_t)(beg + lane) * words];
    {
        unsigned nlo, nhi;
        asm volatile("v_mov_b32 %0, %2\n\tv_mov_b32 %1, %3"
                     : "=v"(nlo), "=v"(nhi)
                     : "v"((unsigned)diag_next), "v"((unsigned)(diag_next >> 32)));
        diag_cur = ((unsigned long long)nhi << 32) | nlo;
    }
    for (int c = 0; c < nchunk; c++) {
        // rows of this chunk x the next 64 columns, ALL 64 rows (wave = which 16 rows, lane = column), issued before
        // the sequential resolution of the diagonal so that their latency hides under it; the survivors' words are
        // picked out of the registers afterwards
        const unsigned long long* mchunk = mask + (size_t)(beg + c * 64) * words;
        const int w0 = c + 1 + lane;
        unsigned long long pre[16];
#pragma unroll
        for (int j = 0; j < 16; j++) {
            const int r = wave * 16 + j;
            pre[j] = (w0 < nchunk && c * 64 + r < len) ? mchunk[(size_t)r * words + w0] : 0ull;
        }
        if (wave == 0) {
            const int row = c * 64 + lane;
            const unsigned long long diag = diag_cur;
            if (c + 1 < nchunk && row + 64 < len) diag_next = mask[(size_t)(beg + row + 64) * words + c + 1];
            else diag_next = 0ull;
            // the greedy chain of the chunk on the scalar unit: `cur` (removed bits) and `kept` live in SGPRs, one
            // iteration per SURVIVOR (find-first-set over the not-yet-removed rows), the survivor's diagonal word
            // comes out of lane b of `diag`
            const unsigned long long cur_v = remv[c];
            // (the builtin returns int: cast before widening)
            unsigned long long cur = ((unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((int)(cur_v >> 32)) << 32) |
                                     (unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((int)cur_v);
            const int nb = min(64, len - c * 64);
            const unsigned long long valid = nb >= 64 ? ~0ull : ((1ull << nb) - 1ull);
            unsigned long long kept = 0ull;
            const unsigned lo = (unsigned)diag, hi = (unsigned)(diag >> 32);
            unsigned long long rem = ~cur & valid;
            while (rem) {
                const int b = __builtin_ctzll(rem);
                kept |= 1ull << b;
                const unsigned dlo = __builtin_amdgcn_readlane(lo, b);
                const unsigned dhi = __builtin_amdgcn_readlane(hi, b);
                cur |= ((unsigned long long)dhi << 32) | dlo;
                // rows above b that are still alive (b itself leaves the candidate set)
                rem = ~cur & valid & ~((2ull << b) - 1ull);
            }
            const bool mine = (kept >> lane) & 1ull;
            const int rank = __popcll(kept & ((1ull << lane) - 1ull));
            if (mine) {
                const int pos = count + rank;
                if (max_keep <= 0 || pos < max_keep) keep[beg + pos] = (int64_t)sorted_idx[beg + row];
            }
            count += __popcll(kept);
            if (lane == 0) { *kept_sh = kept; sh_count = count; }
        }
        __syncthreads();
        count = sh_count;
        if (max_keep > 0 && count >= max_keep) { count = max_keep; break; }
        const unsigned long long k2 = *kept_sh;
        {
            unsigned long long acc = 0ull;
            const unsigned kw = (unsigned)(k2 >> (wave * 16)) & 0xffffu;
#pragma unroll
            for (int j = 0; j < 16; j++) acc |= ((kw >> j) & 1u) ? pre[j] : 0ull;
            if (acc) atomicOr(&remv[w0], acc);
        }
        // columns beyond the prefetched 64 (segments of more than 4096 + 64 c candidates): a thread owns a column,
        // the survivors' words are independent loads issued 16 deep (a spent bit set re-reads the last row: OR is
        // idempotent)
        for (int w = c + 65 + tid; w < nchunk; w += 256) {
            unsigned long long acc = 0ull, k = k2;
            int b = 0;
            while (k) {
                unsigned long long v[16];
#pragma unroll
                for (int j = 0; j < 16; j++) {
                    if (k) { b = __ffsll((long long)k) - 1; k &= k - 1; }
                    v[j] = mchunk[(size_t)b * words + w];
                }
#pragma unroll
                for (int j = 0; j < 16; j++) acc |= v[j];
            }
            if (acc) atomicOr(&remv[w], acc);
        }
        // the prefetched diagonal word has landed by now: settle the load counter here, not in front of the chain
        {
            unsigned nlo, nhi;
            asm volatile("v_mov_b32 %0, %2\n\tv_mov_b32 %1, %3"
                         : "=v"(nlo), "=v"(nhi)
                         : "v"((unsigned)diag_next), "v"((unsigned)(diag_next >> 32)));
            diag_cur = ((unsigned long long)nhi << 32) | nlo;
        }
        __syncthreads();
    }
    if (tid == 0) num_keep[seg] = count;
}

struct NmsWs {
    float* keys_out;
    int32_t* idx_in;
    int32_t* idx_out;
    float* sboxes;
    float* sareas;
    unsigned long long* mask;
    void* sort_tmp;
    size_t sort_tmp_bytes;
    size_t total;
};

inline size_t align_up(size_t x) { return (x + 255) & ~(size_t)255; }

size_t sort_temp_bytes(int64_t n, int S) {
    size_t bytes = 0;
    (void)rocprim::segmented_radix_sort_pairs_desc<rocprim::default_config, const float*, float*,
                                             const int32_t*, int32_t*, const int32_t*>(
        nullptr, bytes, nullptr, nullptr, nullptr, nullptr, (unsigned)n, (unsigned)S, nullptr,
        nullptr, 0, 32, (hipStream_t)0, false);
    return bytes;
}

NmsWs carve(void* base, int64_t n, int S, int words) {
    NmsWs w;
    size_t off = 0;
    char* p = (char*)base;
    auto take = [&](size_t bytes) { void* r = p ? p + off : nullptr; off += align_up(bytes); return r; };
    w.keys_out = (float*)take(n * 4);
    w.idx_in = (int32_t*)take(n * 4);
    w.idx_out = (int32_t*)take(n * 4);
    w.sboxes = (float*)take(n * 16);
    w.sareas = (float*)take(n * 4);
    w.mask = (unsigned long long*)take((size_t)n * words * 8);
    w.sort_tmp_bytes = sort_temp_bytes(n, S);
    w.sort_tmp = take(w.sort_tmp_bytes);
    w.total = off;
    return w;
}


struct LevelBounds {
    int num;                               // 0: one segment per image
    int col0[BRCNN_MAX_LEVELS + 1];        // candidate columns [col0[l], col0[l+1]) belong to id / level l
};

// ---- whole-batch candidate preparation for batched NMS (postprocess.batched_nms_images) ------
// One workgroup per image slot of T candidates: order-preserving compaction of the valid ones
// (rejected rows leave zeros behind), the image's largest surviving coordinate, and the
// class / level separation of mmcv's batched_nms on the same fp32 values:
//     boxes_for_nms = box + float(id) * (max_coordinate + 1)
// plus the image's [begin, end) range for the segmented NMS.  Replaces ~30 small torch launches
// (cumsum, where, 3 scatters, amax, ...) per call.
__global__ __launch_bounds__(1024) void nms_prepare_kernel(const float* __restrict__ boxes, const float* __restrict__ scores,
                                                          const long long* __restrict__ ids,
                                                          const unsigned char* __restrict__ valid, float* __restrict__ c_boxes,
                                                          float* __restrict__ c_scores, long long* __restrict__ c_ids,
                                                          float* __restrict__ nms_boxes, int* __restrict__ ranges, int T,
                                                          const LevelBounds lb) {
    __shared__ int wsum[16];
    __shared__ int s_lcnt[BRCNN_MAX_LEVELS];
    __shared__ float wmax[16];
    __shared__ int s_base;
    __shared__ float s_max;
    const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const size_t row0 = (size_t)b * T;
    if (tid == 0) s_base = 0;
    if (tid < BRCNN_MAX_LEVELS) s_lcnt[tid] = 0;
    float vmax = -3.402823466e+38f;             // torch.finfo(float32).min, the reference's fill value
    __syncthreads();
    // pass 1: positions of the survivors + the largest coordinate among them
    for (int t0 = 0; t0 < T; t0 += 1024) {
        const int t = t0 + tid;
        const int v = (t < T && valid[row0 + t]) ? 1 : 0;
        // inclusive scan inside the wave, then across the 16 waves
        int incl = v;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
            const int o = __shfl_up(incl, d, 64);
            if (lane >= d) incl += o;
        }
        if (lane == 63) wsum[wave] = incl;
        __syncthreads();
        int woff = 0;
        for (int w = 0; w < wave; w++) woff += wsum[w];
        int total = 0;
        for (int w = 0; w < 16; w++) total += wsum[w];
        const int pos = s_base + woff + incl - 1;
        if (v) {
            const float4 bx = *reinterpret_cast<const float4*>(boxes + (row0 + t) * 4);
            *reinterpret_cast<float4*>(c_boxes + (row0 + pos) * 4) = bx;
            c_scores[row0 + pos] = scores[row0 + t];
            c_ids[row0 + pos] = ids[row0 + t];
            vmax = fmaxf(vmax, fmaxf(fmaxf(bx.x, bx.y), fmaxf(bx.z, bx.w)));
            if (lb.num > 0) {
                int l = 0;
#pragma unroll
                for (int k = 1; k < BRCNN_MAX_LEVELS; k++)
                    if (k < lb.num && t >= lb.col0[k]) l = k;
                atomicAdd(&s_lcnt[l], 1);
            }
        }
        __syncthreads();
        if (tid == 0) s_base += total;
        __syncthreads();
    }
    const int cnt = s_base;
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) vmax = fmaxf(vmax, __shfl_xor(vmax, d, 64));
    if (lane == 0) wmax[wave] = vmax;
    __syncthreads();
    if (tid == 0) {
        float m = wmax[0];
        for (int w = 1; w < 16; w++) m = fmaxf(m, wmax[w]);
        s_max = m;
        if (lb.num > 0) {
            // the survivors of a level stay contiguous: one [begin, end) per (image, level)
            int beg = b * T;
            for (int l = 0; l < lb.num; l++) {
                ranges[2 * (b * lb.num + l)] = beg;
                beg += s_lcnt[l];
                ranges[2 * (b * lb.num + l) + 1] = beg;
            }
        } else {
            ranges[2 * b] = b * T;
            ranges[2 * b + 1] = b * T + cnt;
        }
    }
    __syncthreads();
    const float step = s_max + 1.0f;
    // pass 2: zero the tail of the slot, offset boxes for the segmented NMS
    for (int t = tid; t < T; t += 1024) {
        if (t < cnt) {
            const float4 bx = *reinterpret_cast<const float4*>(c_boxes + (row0 + t) * 4);
            const float off = (float)c_ids[row0 + t] * step;
            *reinterpret_cast<float4*>(nms_boxes + (row0 + t) * 4) = make_float4(bx.x + off, bx.y + off, bx.z + off, bx.w + off);
        } else {
            *reinterpret_cast<float4*>(c_boxes + (row0 + t) * 4) = make_float4(0.f, 0.f, 0.f, 0.f);
            *reinterpret_cast<float4*>(nms_boxes + (row0 + t) * 4) = make_float4(0.f, 0.f, 0.f, 0.f);
            c_scores[row0 + t] = 0.f;
            c_ids[row0 + t] = 0;
        }
    }
}

// survivors of the segmented NMS back into fixed (B, K) slots: dets = [box, score] of keep[b*T + k]
// for k < min(num[b], K), zero rows / id -1 behind them
__global__ __launch_bounds__(256) void nms_collect_kernel(const long long* __restrict__ keep, const int* __restrict__ num,
                                                         const float* __restrict__ c_boxes, const float* __restrict__ c_scores,
                                                         const long long* __restrict__ c_ids, float* __restrict__ dets,
                                                         long long* __restrict__ ids_kept, int B, int T, int K) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= B * K) return;
    const int b = i / K, k = i - b * K;
    float* d = dets + (size_t)i * 5;
    if (k < num[b]) {
        const long long src = keep[(size_t)b * T + k];
        const float4 bx = *reinterpret_cast<const float4*>(c_boxes + src * 4);
        d[0] = bx.x; d[1] = bx.y; d[2] = bx.z; d[3] = bx.w; d[4] = c_scores[src];
        ids_kept[i] = c_ids[src];
    } else {
        d[0] = d[1] = d[2] = d[3] = d[4] = 0.f;
        ids_kept[i] = -1;
    }
}


// ---- segments of at most 8192 candidates: sort + gather in one launch ---------------------
// One 1024-thread workgroup per segment: (score, position) pairs as 64-bit composites
// [~order_key(score) | position] sorted ascending by a bitonic network in LDS -- i.e. descending
// score, ties by ascending position, the order of the stable rocPRIM sort -- then the sorted
// global indices, the boxes in that order and their areas are written directly (replaces iota +
// segmented radix sort + gather: 154 -> ~45 us for 8 x 4693 RPN candidates).
__device__ __forceinline__ unsigned order_key(float f) {
    const unsigned u = __float_as_uint(f);
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);      // ascending in f
}

template <int THREADS>
__global__ __launch_bounds__(THREADS) void seg_sort_gather_kernel(const float* __restrict__ boxes,
                                                              const float* __restrict__ scores,
                                                              const int32_t* __restrict__ seg_begin,
                                                              const int32_t* __restrict__ seg_end,
                                                              int32_t* __restrict__ idx_out, float* __restrict__ sboxes,
                                                              float* __restrict__ sareas, int offset) {
    extern __shared__ unsigned long long comp[];
    const int seg = blockIdx.x, tid = threadIdx.x;
    const int beg = seg_begin[seg], len = seg_end[seg] - beg;
    if (len <= 0) return;
    int P = 64;
    while (P < len) P <<= 1;
    for (int i = tid; i < P; i += THREADS)
        comp[i] = i < len ? (((unsigned long long)(~order_key(scores[beg + i])) << 32) | (unsigned)i) : ~0ull;
    __syncthreads();
    for (int k = 2; k <= P; k <<= 1) {
        for (int j = k >> 1; j > 0; j >>= 1) {
            for (int t = tid; t < (P >> 1); t += THREADS) {
                const int lo = ((t & ~(j - 1)) << 1) | (t & (j - 1));     // index with bit j clear
                const int hi = lo | j;
                const unsigned long long a = comp[lo], b = comp[hi];
                const bool up = (lo & k) == 0;
                if ((a > b) == up) { comp[lo] = b; comp[hi] = a; }
            }
            __syncthreads();
        }
    }
    for (int i = tid; i < len; i += THREADS) {
        const int src = beg + (int)(comp[i] & 0xffffffffu);
        idx_out[beg + i] = src;
        const float4 b = *reinterpret_cast<const float4*>(boxes + (size_t)src * 4);
        *reinterpret_cast<float4*>(sboxes + (size_t)(beg + i) * 4) = b;
        sareas[beg + i] = (b.z - b.x + offset) * (b.w - b.y + offset);
    }
}

// mmcv batched_nms above split_thr (the `for id in torch.unique(idxs)` branch, mmcv/ops/nms.py): after the
// per-(image, id) NMS the survivors of all ids of an image are re-sorted by score (descending, ties by
// ascending candidate position) and the first K kept.  One workgroup per image: (score, position)
// composites of the survivors in LDS, bitonic sort, gather.  `new_scores` (flat rows of 5, column 4) are
// the decayed scores of soft-NMS picks, aligned with `keep`.
__global__ __launch_bounds__(1024) void nms_collect_sorted_kernel(const long long* __restrict__ keep,
                                                                 const int* __restrict__ num, const int* __restrict__ ranges,
                                                                 const float* __restrict__ c_boxes,
                                                                 const float* __restrict__ c_scores,
                                                                 const long long* __restrict__ c_ids,
                                                                 const float* __restrict__ new_scores,
                                                                 float* __restrict__ dets, long long* __restrict__ ids_kept,
                                                                 int* __restrict__ n_kept, int T, int L, int K) {
    extern __shared__ unsigned long long comp[];
    __shared__ int s_off[BRCNN_MAX_LEVELS + 1];
    const int b = blockIdx.x, tid = threadIdx.x;
    if (tid == 0) {
        int o = 0;
        for (int l = 0; l < L; l++) { s_off[l] = o; o += num[b * L + l]; }
        s_off[L] = o;
    }
    __syncthreads();
    const int cnt = s_off[L];
    int P = 64;
    while (P < cnt) P <<= 1;
    for (int l = 0; l < L; l++) {
        const int beg = ranges[2 * (b * L + l)], n_l = s_off[l + 1] - s_off[l];
        for (int i = tid; i < n_l; i += 1024) {
            const long long src = keep[beg + i];
            const float sc = new_scores ? new_scores[(size_t)(beg + i) * 5 + 4] : c_scores[src];
            comp[s_off[l] + i] = ((unsigned long long)(~order_key(sc)) << 32) | (unsigned)(src - (long long)b * T);
        }
    }
    for (int i = cnt + tid; i < P; i += 1024) comp[i] = ~0ull;
    __syncthreads();
    for (int k = 2; k <= P; k <<= 1) {
        for (int j = k >> 1; j > 0; j >>= 1) {
            for (int t = tid; t < (P >> 1); t += 1024) {
                const int lo = ((t & ~(j - 1)) << 1) | (t & (j - 1));
                const int hi = lo | j;
                const unsigned long long a = comp[lo], c = comp[hi];
                const bool up = (lo & k) == 0;
                if ((a > c) == up) { comp[lo] = c; comp[hi] = a; }
            }
            __syncthreads();
        }
    }
    const int kept = cnt < K ? cnt : K;
    if (tid == 0) n_kept[b] = kept;
    for (int k = tid; k < K; k += 1024) {
        float* d = dets + ((size_t)b * K + k) * 5;
        if (k < kept) {
            const unsigned long long c = comp[k];
            const size_t src = (size_t)b * T + (unsigned)(c & 0xffffffffu);
            const float4 bx = *reinterpret_cast<const float4*>(c_boxes + src * 4);
            // the score back from its order key
            const unsigned key = ~(unsigned)(c >> 32);
            const unsigned u = (key & 0x80000000u) ? (key & 0x7fffffffu) : ~key;
            d[0] = bx.x; d[1] = bx.y; d[2] = bx.z; d[3] = bx.w; d[4] = __uint_as_float(u);
            if (ids_kept) ids_kept[(size_t)b * K + k] = c_ids[src];
        } else {
            d[0] = d[1] = d[2] = d[3] = d[4] = 0.f;
            if (ids_kept) ids_kept[(size_t)b * K + k] = -1;
        }
    }
}

}  // namespace

static int g_nms_lds_sort = getenv("BRCNN_NMS_RADIX") ? 0 : 1;     // BRCNN_NMS_RADIX=1: always the rocPRIM sort (A/B)

BRCNN_API size_t brcnn_nms_workspace_bytes(int64_t n, int num_segments, int64_t max_segment_len) {
    if (n <= 0 || num_segments <= 0) return 256;
    if (max_segment_len <= 0 || max_segment_len > n) max_segment_len = n;
    const int words = (int)((max_segment_len + 63) / 64);
    return carve(nullptr, n, num_segments, words).total;
}

BRCNN_API int brcnn_nms(const float* boxes, const float* scores, const int32_t* seg_begin,
                        const int32_t* seg_end, int num_segments, int64_t n, int64_t max_segment_len, float iou_threshold,
                        int offset, int max_keep, int64_t* keep, int32_t* num_keep, void* workspace,
                        size_t workspace_bytes, void* stream) {
    if (n < 0 || num_segments <= 0 || (offset != 0 && offset != 1) || !seg_begin || !seg_end || !num_keep)
        return BRCNN_EINVAL;
    hipStream_t s = (hipStream_t)stream;
    if (n == 0) {
        BRCNN_HIP_CHECK(hipMemsetAsync(num_keep, 0, sizeof(int32_t) * num_segments, s));
        return 0;
    }
    if (!boxes || !scores || !keep || !workspace || max_segment_len <= 0 || max_segment_len > n)
        return BRCNN_EINVAL;
    if (n > 0x7fffffffLL) return BRCNN_EINVAL;
    const int words = (int)((max_segment_len + 63) / 64);
    NmsWs w = carve(workspace, n, num_segments, words);
    if (w.total > workspace_bytes) return BRCNN_EINVAL;

    // own sort for every segment length the pipeline produces (<= 16384 candidates: the composites of a segment fit
    // the 160 KB of LDS); rocPRIM's segmented radix sort stays as the fallback for longer segments only
    if (max_segment_len <= 16384 && g_nms_lds_sort) {
        int P = 64;
        while (P < max_segment_len) P <<= 1;
        static bool attr_done = false;
        if (!attr_done) {
            BRCNN_HIP_CHECK(hipFuncSetAttribute((const void*)seg_sort_gather_kernel<1024>,
                                                hipFuncAttributeMaxDynamicSharedMemorySize, 16384 * 8));
            attr_done = true;
        }
        // rows outside every segment are never read by the mask / reduce kernels
        if (max_segment_len <= 512)        // short segments (per-class second-stage sets): four wavefronts each
            hipLaunchKernelGGL(seg_sort_gather_kernel<256>, dim3(num_segments), dim3(256), (size_t)P * 8, s, boxes, scores,
                               seg_begin, seg_end, w.idx_out, w.sboxes, w.sareas, offset);
        else
            hipLaunchKernelGGL(seg_sort_gather_kernel<1024>, dim3(num_segments), dim3(1024), (size_t)P * 8, s, boxes, scores,
                               seg_begin, seg_end, w.idx_out, w.sboxes, w.sareas, offset);
        BRCNN_LAUNCH_CHECK();
    } else {
        hipLaunchKernelGGL(iota_kernel, dim3(brcnn_cdiv(n, 256)), dim3(256), 0, s, w.idx_in, w.idx_out, n);
        BRCNN_LAUNCH_CHECK();
        size_t tmp = w.sort_tmp_bytes;
        BRCNN_HIP_CHECK((rocprim::segmented_radix_sort_pairs_desc(
            w.sort_tmp, tmp, scores, w.keys_out, (const int32_t*)w.idx_in, w.idx_out, (unsigned)n,
            (unsigned)num_segments, seg_begin, seg_end, 0, 32, s, false)));
        hipLaunchKernelGGL(gather_boxes_kernel, dim3(brcnn_cdiv(n, 256)), dim3(256), 0, s, boxes,
                           (const int32_t*)w.idx_out, w.sboxes, w.sareas, n, offset);
        BRCNN_LAUNCH_CHECK();
    }
    hipLaunchKernelGGL(nms_mask_kernel, dim3(words, words, num_segments), dim3(64), 0, s,
                       (const float*)w.sboxes, (const float*)w.sareas, seg_begin, seg_end, w.mask, words,
                       iou_threshold, offset);
    BRCNN_LAUNCH_CHECK();
    hipLaunchKernelGGL(nms_reduce_kernel, dim3(num_segments), dim3(256), (size_t)(words + 1) * 8, s,
                       (const unsigned long long*)w.mask, (const int32_t*)w.idx_out, seg_begin, seg_end,
                       keep, num_keep, words, max_keep);
    BRCNN_LAUNCH_CHECK();
    return 0;
}

BRCNN_API int brcnn_nms_prepare(const float* boxes, const float* scores, const int64_t* ids, const uint8_t* valid,
                                float* c_boxes, float* c_scores, int64_t* c_ids, float* nms_boxes, int32_t* ranges,
                                int batch, int slot, void* stream) {
    if (!boxes || !scores || !ids || !valid || !c_boxes || !c_scores || !c_ids || !nms_boxes || !ranges ||
        batch <= 0 || slot <= 0 || (long long)batch * slot >= 0x7fffffffLL)
        return BRCNN_EINVAL;
    LevelBounds lb;
    lb.num = 0;
    hipLaunchKernelGGL(nms_prepare_kernel, dim3(batch), dim3(1024), 0, (hipStream_t)stream, boxes, scores,
                       (const long long*)ids, valid, c_boxes, c_scores, (long long*)c_ids, nms_boxes, ranges, slot, lb);
    BRCNN_LAUNCH_CHECK();
    return 0;
}

BRCNN_API int brcnn_nms_prepare_levels(const float* boxes, const float* scores, const int64_t* ids, const uint8_t* valid,
                                       float* c_boxes, float* c_scores, int64_t* c_ids, float* nms_boxes,
                                       int32_t* ranges, int batch, int slot, int num_levels,
                                       const int* level_sizes_host, void* stream) {
    if (!boxes || !scores || !ids || !valid || !c_boxes || !c_scores || !c_ids || !nms_boxes || !ranges ||
        batch <= 0 || slot <= 0 || (long long)batch * slot >= 0x7fffffffLL || num_levels <= 0 ||
        num_levels > BRCNN_MAX_LEVELS || !level_sizes_host)
        return BRCNN_EINVAL;
    LevelBounds lb;
    lb.num = num_levels;
    int c = 0;
    for (int l = 0; l < num_levels; l++) {
        lb.col0[l] = c;
        if (level_sizes_host[l] < 0) return BRCNN_EINVAL;
        c += level_sizes_host[l];
    }
    lb.col0[num_levels] = c;
    if (c != slot) return BRCNN_EINVAL;
    hipLaunchKernelGGL(nms_prepare_kernel, dim3(batch), dim3(1024), 0, (hipStream_t)stream, boxes, scores,
                       (const long long*)ids, valid, c_boxes, c_scores, (long long*)c_ids, nms_boxes, ranges, slot, lb);
    BRCNN_LAUNCH_CHECK();
    return 0;
}

BRCNN_API int brcnn_nms_collect_sorted(const int64_t* keep, const int32_t* num, const int32_t* ranges,
                                       const float* c_boxes, const float* c_scores, const int64_t* c_ids,
                                       const float* new_scores5, float* dets, int64_t* ids_kept, int32_t* n_kept,
                                       int batch, int slot, int num_levels, int max_keep, void* stream) {
    if (!keep || !num || !ranges || !c_boxes || !c_scores || !c_ids || !dets || !n_kept || batch <= 0 || slot <= 0 ||
        slot > 16384 || num_levels <= 0 || num_levels > BRCNN_MAX_LEVELS || max_keep <= 0 || max_keep > slot)
        return BRCNN_EINVAL;
    int P = 64;
    while (P < slot) P <<= 1;
    static bool attr_done = false;
    if (!attr_done) {
        BRCNN_HIP_CHECK(hipFuncSetAttribute((const void*)nms_collect_sorted_kernel,
                                            hipFuncAttributeMaxDynamicSharedMemorySize, 16384 * 8));
        attr_done = true;
    }
    hipLaunchKernelGGL(nms_collect_sorted_kernel, dim3(batch), dim3(1024), (size_t)P * 8, (hipStream_t)stream,
                       (const long long*)keep, num, ranges, c_boxes, c_scores, (const long long*)c_ids, new_scores5, dets,
                       (long long*)ids_kept, n_kept, slot, num_levels, max_keep);
    BRCNN_LAUNCH_CHECK();
    return 0;
}

BRCNN_API int brcnn_nms_collect(const int64_t* keep, const int32_t* num, const float* c_boxes, const float* c_scores,
                                const int64_t* c_ids, float* dets, int64_t* ids_kept, int batch, int slot,
                                int max_keep, void* stream) {
    if (!keep || !num || !c_boxes || !c_scores || !c_ids || !dets || !ids_kept || batch <= 0 || slot <= 0 ||
        max_keep <= 0 || max_keep > slot)
        return BRCNN_EINVAL;
    const int total = batch * max_keep;
    hipLaunchKernelGGL(nms_collect_kernel, dim3((total + 255) / 256), dim3(256), 0, (hipStream_t)stream,
                       (const long long*)keep, num, c_boxes, c_scores, (const long long*)c_ids, dets,
                       (long long*)ids_kept, batch, slot, max_keep);
    BRCNN_LAUNCH_CHECK();
    return 0;
}
