// RoIAlign forward/backward for gfx950 (CDNA4).
//
// Replaces mmcv.ops.roi_align_{forward,backward} (call sites:
// mmdet/models/roi_heads/roi_extractors/base_roi_extractor.py:54-60,
// single_level_roi_extractor.py:103).  Semantics = mmcv ROIAlign (aligned flag,
// adaptive sampling grid, avg/max pooling); arithmetic order per output element is
// the reference's: val = w1*v1 + w2*v2 + w3*v3 + w4*v4, sum over (iy, ix), / count,
// compiled with -ffp-contract=off so results are bit-identical to the C oracle.
//
// HBM-bound gather.  NHWC kernel: one 64-lane wavefront per RoI bin, channels across
// lanes (float4 per lane => one 1 KiB coalesced line per bilinear corner at C=256),
// 4 bins per 256-thread workgroup, consecutive workgroups walk the bins of one RoI so
// the corner lines shared by neighbouring bins/samples are L1/L2 hits.  NCHW kernel:
// one thread per output element (the reference's own layout; uncoalesced by nature).
#include "common.h"

namespace {

// NHWC element types: fp32 or bf16 (as unsigned short), 4 channels per lane
__device__ __forceinline__ float4 ld4(const float* p) { return *reinterpret_cast<const float4*>(p); }
__device__ __forceinline__ void st4(float* p, float4 v) { *reinterpret_cast<float4*>(p) = v; }
__device__ __forceinline__ float4 ld4(const bf16_t* p) {
    const uint2 u = *reinterpret_cast<const uint2*>(p);
    return make_float4(__uint_as_float(u.x << 16), __uint_as_float(u.x & 0xffff0000u),
                       __uint_as_float(u.y << 16), __uint_as_float(u.y & 0xffff0000u));
}
__device__ __forceinline__ unsigned f2bf(float v) { return brcnn_f2b(v); }
__device__ __forceinline__ void st4(bf16_t* p, float4 v) {
    uint2 u;
    u.x = brcnn_pk2b(v.x, v.y);
    u.y = brcnn_pk2b(v.z, v.w);
    *reinterpret_cast<uint2*>(p) = u;
}

__device__ __forceinline__ float4 ld4(const f16_t* p) {
    const uint2 u = *reinterpret_cast<const uint2*>(p);
    return make_float4(brcnn_h2f((unsigned short)(u.x & 0xffffu)), brcnn_h2f((unsigned short)(u.x >> 16)),
                       brcnn_h2f((unsigned short)(u.y & 0xffffu)), brcnn_h2f((unsigned short)(u.y >> 16)));
}
__device__ __forceinline__ void st4(f16_t* p, float4 v) {
    uint2 u;
    u.x = brcnn_pk2h(v.x, v.y);
    u.y = brcnn_pk2h(v.z, v.w);
    *reinterpret_cast<uint2*>(p) = u;
}

struct RoiGeom {
    float start_h, start_w, bin_h, bin_w;
    int gh, gw;
    float count;
    int batch;
};

__device__ __forceinline__ RoiGeom roi_geom(const float* __restrict__ roi, float scale, int aligned,
                                            int ph_n, int pw_n, int sampling_ratio) {
    RoiGeom g;
    g.batch = (int)roi[0];
    float offset = aligned ? 0.5f : 0.0f;
    float roi_start_w = roi[1] * scale - offset;
    float roi_start_h = roi[2] * scale - offset;
    float roi_end_w = roi[3] * scale - offset;
    float roi_end_h = roi[4] * scale - offset;
    float roi_width = roi_end_w - roi_start_w;
    float roi_height = roi_end_h - roi_start_h;
    if (!aligned) {
        roi_width = fmaxf(roi_width, 1.f);
        roi_height = fmaxf(roi_height, 1.f);
    }
    g.bin_h = roi_height / (float)ph_n;
    g.bin_w = roi_width / (float)pw_n;
    g.gh = (sampling_ratio > 0) ? sampling_ratio : (int)ceilf(roi_height / (float)ph_n);
    g.gw = (sampling_ratio > 0) ? sampling_ratio : (int)ceilf(roi_width / (float)pw_n);
    if (g.gh < 0) g.gh = 0;
    if (g.gw < 0) g.gw = 0;
    int c = g.gh * g.gw;
    g.count = (float)(c < 1 ? 1 : c);
    g.start_h = roi_start_h;
    g.start_w = roi_start_w;
    return g;
}

struct Tap {
    int p1, p2, p3, p4;   // y*W+x of the four corners
    float w1, w2, w3, w4;
    bool valid;
};

__device__ __forceinline__ Tap bilinear_tap(float y, float x, int height, int width) {
    Tap t;
    if (y < -1.0f || y > (float)height || x < -1.0f || x > (float)width) {
        t.valid = false;
        t.p1 = t.p2 = t.p3 = t.p4 = 0;
        t.w1 = t.w2 = t.w3 = t.w4 = 0.f;
        return t;
    }
    t.valid = true;
    if (y <= 0) y = 0;
    if (x <= 0) x = 0;
    int y_low = (int)y, x_low = (int)x, y_high, x_high;
    if (y_low >= height - 1) { y_high = y_low = height - 1; y = (float)y_low; }
    else y_high = y_low + 1;
    if (x_low >= width - 1) { x_high = x_low = width - 1; x = (float)x_low; }
    else x_high = x_low + 1;
    float ly = y - y_low, lx = x - x_low;
    float hy = 1.f - ly, hx = 1.f - lx;
    t.w1 = hy * hx; t.w2 = hy * lx; t.w3 = ly * hx; t.w4 = ly * lx;
    t.p1 = y_low * width + x_low;
    t.p2 = y_low * width + x_high;
    t.p3 = y_high * width + x_low;
    t.p4 = y_high * width + x_high;
    return t;
}

struct LevelTable {
    const float* feat[BRCNN_MAX_LEVELS];
    float* gfeat[BRCNN_MAX_LEVELS];
    int height[BRCNN_MAX_LEVELS];
    int width[BRCNN_MAX_LEVELS];
    float scale[BRCNN_MAX_LEVELS];
    int num_levels;
    float finest_scale;
};

// SingleRoIExtractor.map_roi_levels (single_level_roi_extractor.py:36-55):
//   scale = sqrt((x2-x1)*(y2-y1)); lvl = floor(log2(scale / finest_scale + 1e-6)) clamped.
__device__ __forceinline__ int map_roi_level(const float* __restrict__ roi, float finest_scale,
                                             int num_levels) {
    float s = sqrtf((roi[3] - roi[1]) * (roi[4] - roi[2]));
    float l = floorf(log2f(s / finest_scale + 1e-6f));
    // clamp(min=0, max=L-1).long(); NaN (negative area) -> reference yields an
    // undefined long; we send it to level 0.
    if (!(l > 0.f)) l = 0.f;
    if (l > (float)(num_levels - 1)) l = (float)(num_levels - 1);
    return (int)l;
}

// ---------------------------------------------------------------------------------------
// NHWC forward: one wave per (roi, ph, pw) bin; lane owns channels [4*lane + 256*j, +4).
// ---------------------------------------------------------------------------------------
template <bool MULTI, typename T = float>
__global__ __launch_bounds__(256) void roi_align_fwd_nhwc_kernel(
    const T* __restrict__ input, LevelTable lv, const float* __restrict__ rois,
    T* __restrict__ output, int32_t* __restrict__ levels_out, int channels, int height,
    int width, int n_rois, int ph_n, int pw_n, float spatial_scale, int sampling_ratio,
    int aligned) {
    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    const long long bin = (long long)blockIdx.x * 4 + wave;
    const int bins_per_roi = ph_n * pw_n;
    if (bin >= (long long)n_rois * bins_per_roi) return;
    const int k = (int)(bin / bins_per_roi);
    const int r = (int)(bin - (long long)k * bins_per_roi);
    const int ph = r / pw_n, pw = r - ph * pw_n;
    const float* roi = rois + (size_t)k * 5;

    const T* feat = input;
    if (MULTI) {
        int l = map_roi_level(roi, lv.finest_scale, lv.num_levels);
        feat = reinterpret_cast<const T*>(lv.feat[l]);
        height = lv.height[l];
        width = lv.width[l];
        spatial_scale = lv.scale[l];
        if (levels_out && r == 0 && lane == 0) levels_out[k] = l;
    }
    const RoiGeom g = roi_geom(roi, spatial_scale, aligned, ph_n, pw_n, sampling_ratio);
    const T* base = feat + (size_t)g.batch * height * width * channels;
    T* out = output + (size_t)bin * channels;

    for (int c0 = lane * 4; c0 < channels; c0 += 256) {
        float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
        for (int iy = 0; iy < g.gh; iy++) {
            const float y = g.start_h + ph * g.bin_h + (float)(iy + .5f) * g.bin_h / (float)g.gh;
            for (int ix = 0; ix < g.gw; ix++) {
                const float x = g.start_w + pw * g.bin_w + (float)(ix + .5f) * g.bin_w / (float)g.gw;
                const Tap t = bilinear_tap(y, x, height, width);
                if (!t.valid) continue;   // contributes exactly +0.f in the reference
                const float4 v1 = ld4(base + (size_t)t.p1 * channels + c0);
                const float4 v2 = ld4(base + (size_t)t.p2 * channels + c0);
                const float4 v3 = ld4(base + (size_t)t.p3 * channels + c0);
                const float4 v4 = ld4(base + (size_t)t.p4 * channels + c0);
                acc.x += t.w1 * v1.x + t.w2 * v2.x + t.w3 * v3.x + t.w4 * v4.x;
                acc.y += t.w1 * v1.y + t.w2 * v2.y + t.w3 * v3.y + t.w4 * v4.y;
                acc.z += t.w1 * v1.z + t.w2 * v2.z + t.w3 * v3.z + t.w4 * v4.z;
                acc.w += t.w1 * v1.w + t.w2 * v2.w + t.w3 * v3.w + t.w4 * v4.w;
            }
        }
        acc.x /= g.count; acc.y /= g.count; acc.z /= g.count; acc.w /= g.count;
        st4(out + c0, acc);
    }
}


// ---------------------------------------------------------------------------------------
// NHWC forward, footprint form (default).  The sampling grid of a bin is a product grid and the
// bilinear weight of a sample is hy*hx / ly*hx / ..., validity and border clamping are per axis,
// so   sum_samples sum_corners w * v  ==  sum_{rows} sum_{cols} Wy[row] * Wx[col] * v[row,col]
// with Wy[row] = sum over the bin's valid y-samples of the weight they put on that row (same for
// Wx).  A bin then reads every pixel of its footprint ONCE ((rows x cols) <= (gh+1) x (gw+1)
// loads instead of 4*gh*gw): the kernel is bound by L2->CU bytes, and this halves them on the
// 7x7-bins-of-1..2-px RoIs the level mapping produces.  Lane l evaluates y-sample l and x-sample l
// and then holds Wy[row_min + l], Wx[col_min + l]; the accumulation order differs from the
// reference's sample order, so results agree to fp32 round-off (not bit for bit -- the exact-order
// kernel above stays selectable with brcnn_roi_align_set_exact).  Grids or footprints wider than
// 64 (bins larger than ~60 px) take the sample loop.
// ---------------------------------------------------------------------------------------
struct AxisSample { int lo, hi; float wlo, whi; bool valid; };

__device__ __forceinline__ AxisSample axis_sample(float v, int size) {
    AxisSample a;
    a.valid = !(v < -1.0f || v > (float)size);
    if (v <= 0) v = 0;
    int lo = (int)v, hi;
    if (lo >= size - 1) { hi = lo = size - 1; v = (float)lo; }
    else hi = lo + 1;
    a.lo = lo; a.hi = hi;
    a.whi = v - (float)lo;
    a.wlo = 1.f - a.whi;
    return a;
}

// per-axis footprint weights: returns (first index, extent); lane l ends up with W[first + l]
__device__ __forceinline__ void axis_weights(float start, float bin, int pidx, int g, int size, int lane,
                                             int& first, int& extent, float& W) {
    const float v = start + pidx * bin + (float)(lane + .5f) * bin / (float)g;
    AxisSample a = axis_sample(v, size);
    const bool live = lane < g && a.valid;
    const unsigned long long m = __ballot(live);
    if (m == 0ull) { first = 0; extent = 0; W = 0.f; return; }
    const int f = __ffsll((long long)m) - 1, l = 63 - __clzll((long long)m);
    first = __shfl(a.lo, f);
    extent = __shfl(a.hi, l) - first + 1;
    const int mine = first + lane;
    float w = 0.f;
    for (int i = f; i <= l; i++) {           // valid samples are a contiguous run (monotonic coordinate)
        const int lo = __shfl(a.lo, i), hi = __shfl(a.hi, i);
        const float wl = __shfl(a.wlo, i), wh = __shfl(a.whi, i);
        if (lo == mine) w += wl;
        if (hi == mine) w += wh;
    }
    W = w;
}

// one wave per (roi, ph) ROW of bins: the level mapping (sqrt / log2), the RoI geometry (divisions,
// ceil) and the y-axis weights are computed once per row instead of once per bin -- with 1..2 px
// bins the per-bin address / weight arithmetic, not the bytes, is what bounds this kernel
// RPW: bin rows per wavefront.  1 (few RoIs: as many waves as possible) or pooled_h (thousands of RoIs: the level
// mapping, the RoI geometry and the x-axis weights of the seven bins -- two thirds of a bin row's instructions with
// 1-2 px bins -- are computed once per RoI instead of once per bin row).  perm (or NULL): the order in which the RoIs
// are visited (roi_order_kernel: by level, image and row band, so that the bin rows an XCD works on at one time stay
// inside what its L2 holds); results land at the RoI's own index.
template <bool MULTI, typename T = float, int RPW = 1>
__global__ __launch_bounds__(256) void roi_align_fwd_nhwc_fp_kernel(
    const T* __restrict__ input, LevelTable lv, const float* __restrict__ rois,
    T* __restrict__ output, int32_t* __restrict__ levels_out, int channels, int height,
    int width, int n_rois, int ph_n, int pw_n, float spatial_scale, int sampling_ratio,
    int aligned, int stream_c, const int32_t* __restrict__ perm) {
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    // XCD x (= blockIdx % 8, the dispatcher's round robin) takes a CONTIGUOUS eighth of the bin rows: RoIs arrive
    // image by image, so the maps an XCD's L2 has to hold are one image's (batch 8) instead of the whole batch's
    int bid = blockIdx.x;
    if (stream_c & 2) {
        const int nwg = gridDim.x, q = nwg >> 3, r = nwg & 7, xcd = bid & 7, loc = bid >> 3;
        bid = ((xcd < r) ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + loc;
    }
    const int upr = (ph_n + RPW - 1) / RPW;                  // wave units per RoI
    const long long unit = (long long)bid * 4 + wave;
    if (unit >= (long long)n_rois * upr) return;
    const int kk = (int)(unit / upr);
    const int ph_begin = (int)(unit - (long long)kk * upr) * RPW;
    const int ph_end = min(ph_n, ph_begin + RPW);
    const int k = perm ? perm[kk] : kk;
    const float* roi = rois + (size_t)k * 5;
    const T* feat = input;
    if (MULTI) {
        int l = map_roi_level(roi, lv.finest_scale, lv.num_levels);
        feat = reinterpret_cast<const T*>(lv.feat[l]);
        height = lv.height[l];
        width = lv.width[l];
        spatial_scale = lv.scale[l];
        if (levels_out && ph_begin == 0 && lane == 0) levels_out[k] = l;
    }
    const RoiGeom g = roi_geom(roi, spatial_scale, aligned, ph_n, pw_n, sampling_ratio);
    const T* base = feat + (size_t)g.batch * height * width * channels;
    const bool small = g.gh <= 64 && g.gw <= 64;
    // x-axis weights of ALL bins of the row at once when a bin has <= 8 samples and <= 8 footprint
    // columns (bins up to 7 px): lane group q = lane/8 is bin pw = q, slot j = lane%8
    const int q = lane >> 3, jx = lane & 7;
    int x0_all = 0, nx_all = 0;
    float Wx_all = 0.f;
    bool vecx = small && g.gw <= 8 && pw_n <= 8;
    if (vecx) {
        const float v = g.start_w + q * g.bin_w + (float)(jx + .5f) * g.bin_w / (float)g.gw;
        const AxisSample a = axis_sample(v, width);
        const bool live = jx < g.gw && q < pw_n && a.valid;
        const unsigned mg = (unsigned)((__ballot(live) >> (q * 8)) & 0xffull);
        if (mg) {
            const int f = __ffs((int)mg) - 1, l = 31 - __clz((int)mg);
            x0_all = __shfl(a.lo, q * 8 + f);
            nx_all = __shfl(a.hi, q * 8 + l) - x0_all + 1;
        }
        const int mine = x0_all + jx;
        float w = 0.f;
#pragma unroll
        for (int i = 0; i < 8; i++) {
            const int lo = __shfl(a.lo, q * 8 + i), hi = __shfl(a.hi, q * 8 + i);
            const float wl = __shfl(a.wlo, q * 8 + i), wh = __shfl(a.whi, q * 8 + i);
            if ((mg >> i) & 1u) {
                if (lo == mine) w += wl;
                if (hi == mine) w += wh;
            }
        }
        Wx_all = w;
        if (__ballot(nx_all > 8) != 0ull) vecx = false;
    }
    // ---- one bin row
    auto do_row = [&](int ph) {
    T* out_row = output + ((size_t)k * ph_n + ph) * pw_n * channels;
    int y0 = 0, ny = 0;
    float Wy = 0.f;
    if (small) axis_weights(g.start_h, g.bin_h, ph, g.gh, height, lane, y0, ny, Wy);
    // ---- column streaming (the common case: every bin of the row has <= 8 footprint columns, the row <= 4 footprint
    // rows): the footprint of the WHOLE bin row is one dense patch of ny x (X1 - X0) pixels; each pixel is loaded once
    // (adjacent bins share their border columns: with 1..2 px bins that is a third of the per-bin loads, and the
    // fixed 2 x 4 batches of the per-bin loop padded a 3 x 3 footprint to 16 loads), CB columns x ny rows in flight
    // per batch.  Rows are reduced first (column sum = sum_r Wy[r] v[r][x]), then the column sum goes into the bins
    // whose footprint holds column x with that bin's x weight -- all of it wave-uniform control flow.
    if (vecx && ny <= 4 && pw_n <= 8 && (stream_c & 1)) {
        ny = __builtin_amdgcn_readfirstlane(ny);
        constexpr int CB = 2;
        int bx0[8], bnx[8];
        int X0 = 0x7fffffff, X1 = 0;
#pragma unroll
        for (int b = 0; b < 8; b++) {
            bx0[b] = __builtin_amdgcn_readlane(x0_all, b * 8);
            bnx[b] = b < pw_n ? __builtin_amdgcn_readlane(nx_all, b * 8) : 0;
            if (bnx[b] > 0) { X0 = min(X0, bx0[b]); X1 = max(X1, bx0[b] + bnx[b]); }
        }
        float wy[4];
#pragma unroll
        for (int r = 0; r < 4; r++) wy[r] = r < ny ? __int_as_float(__builtin_amdgcn_readlane(__float_as_int(Wy), r)) : 0.f;
        for (int c0 = lane * 4; c0 < channels; c0 += 256) {
            float4 acc[8];
#pragma unroll
            for (int b = 0; b < 8; b++) acc[b] = make_float4(0.f, 0.f, 0.f, 0.f);
            if (ny > 0) {
                // (wave-uniform pixel address + the lane's channel offset: scalar base, one shared offset register)
                const int y0u = __builtin_amdgcn_readfirstlane(y0), wu = __builtin_amdgcn_readfirstlane(width);
                const unsigned long long bq = (unsigned long long)base;
                const T* bu = (const T*)(((unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((int)(bq >> 32)) << 32) |
                                         (unsigned)__builtin_amdgcn_readfirstlane((int)bq));
                const T* r0 = bu + (size_t)y0u * wu * channels;
                const size_t rstride = (size_t)wu * channels;
                const unsigned lane_off = (unsigned)c0;
                for (int cx = X0; cx < X1; cx += CB) {
                    float4 v[4][CB];
#pragma unroll
                    for (int r = 0; r < 4; r++)
#pragma unroll
                        for (int j = 0; j < CB; j++)
                            if (r < ny && cx + j < X1) v[r][j] = ld4(r0 + r * rstride + (size_t)(cx + j) * channels + lane_off);
#pragma unroll
                    for (int j = 0; j < CB; j++) {
                        if (cx + j >= X1) break;
                        float4 cs = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
                        for (int r = 0; r < 4; r++)
                            if (r < ny) {
                                cs.x += wy[r] * v[r][j].x; cs.y += wy[r] * v[r][j].y;
                                cs.z += wy[r] * v[r][j].z; cs.w += wy[r] * v[r][j].w;
                            }
#pragma unroll
                        for (int b = 0; b < 8; b++) {
                            const int jj = cx + j - bx0[b];
                            if (jj >= 0 && jj < bnx[b]) {
                                const float w = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(Wx_all), b * 8 + jj));
                                acc[b].x += w * cs.x; acc[b].y += w * cs.y; acc[b].z += w * cs.z; acc[b].w += w * cs.w;
                            }
                        }
                    }
                }
            }
            // (x * (1 / count) instead of x / count: 28 IEEE divisions per bin row were a tenth of the wave's instructions;
            // one more rounding, inside the footprint form's stated round-off)
            const float inv = 1.f / g.count;
#pragma unroll
            for (int b = 0; b < 8; b++)
                if (b < pw_n) {
                    float4 o = acc[b];
                    o.x *= inv; o.y *= inv; o.z *= inv; o.w *= inv;
                    st4(out_row + (size_t)b * channels + c0, o);
                }
        }
        return;
    }
    for (int pw = 0; pw < pw_n; pw++) {
        T* out = out_row + (size_t)pw * channels;
        int x0 = 0, nx = 0;
        float Wx = 0.f;
        if (vecx) {
            x0 = __shfl(x0_all, pw * 8);
            nx = __shfl(nx_all, pw * 8);
            Wx = __shfl(Wx_all, pw * 8 + (lane & 7));      // lane l < 8 holds the weight of column x0 + l
        } else if (small) {
            axis_weights(g.start_w, g.bin_w, pw, g.gw, width, lane, x0, nx, Wx);
        }
        if (small && ny <= 64 && nx <= 64) {
            for (int c0 = lane * 4; c0 < channels; c0 += 256) {
                float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
                // 2 rows x 4 columns of the footprint per batch: 8 independent 16-byte loads in
                // flight; positions past the footprint re-read its last pixel with weight 0
                for (int ry = 0; ry < ny; ry += 2) {
                    const float wy0 = __shfl(Wy, ry);
                    const float wy1 = (ry + 1 < ny) ? __shfl(Wy, ry + 1) : 0.f;
                    const T* r0 = base + (size_t)(y0 + ry) * width * channels + c0;
                    const T* r1 = base + (size_t)(y0 + min(ry + 1, ny - 1)) * width * channels + c0;
                    for (int rx = 0; rx < nx; rx += 4) {
                        float4 v0[4], v1[4];
                        float wx[4];
#pragma unroll
                        for (int j = 0; j < 4; j++) {
                            const int xx = x0 + min(rx + j, nx - 1);
                            v0[j] = ld4(r0 + (size_t)xx * channels);
                            v1[j] = ld4(r1 + (size_t)xx * channels);
                            wx[j] = (rx + j < nx) ? __shfl(Wx, rx + j) : 0.f;
                        }
#pragma unroll
                        for (int j = 0; j < 4; j++) {
                            const float a = wy0 * wx[j], b = wy1 * wx[j];
                            acc.x += a * v0[j].x; acc.y += a * v0[j].y; acc.z += a * v0[j].z; acc.w += a * v0[j].w;
                            acc.x += b * v1[j].x; acc.y += b * v1[j].y; acc.z += b * v1[j].z; acc.w += b * v1[j].w;
                        }
                    }
                }
                acc.x /= g.count; acc.y /= g.count; acc.z /= g.count; acc.w /= g.count;
                st4(out + c0, acc);
            }
            continue;
        }
        for (int c0 = lane * 4; c0 < channels; c0 += 256) {      // huge bins: the sample loop
            float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
            for (int iy = 0; iy < g.gh; iy++) {
                const float y = g.start_h + ph * g.bin_h + (float)(iy + .5f) * g.bin_h / (float)g.gh;
                for (int ix = 0; ix < g.gw; ix++) {
                    const float x = g.start_w + pw * g.bin_w + (float)(ix + .5f) * g.bin_w / (float)g.gw;
                    const Tap t = bilinear_tap(y, x, height, width);
                    if (!t.valid) continue;
                    const float4 v1 = ld4(base + (size_t)t.p1 * channels + c0);
                    const float4 v2 = ld4(base + (size_t)t.p2 * channels + c0);
                    const float4 v3 = ld4(base + (size_t)t.p3 * channels + c0);
                    const float4 v4 = ld4(base + (size_t)t.p4 * channels + c0);
                    acc.x += t.w1 * v1.x + t.w2 * v2.x + t.w3 * v3.x + t.w4 * v4.x;
                    acc.y += t.w1 * v1.y + t.w2 * v2.y + t.w3 * v3.y + t.w4 * v4.y;
                    acc.z += t.w1 * v1.z + t.w2 * v2.z + t.w3 * v3.z + t.w4 * v4.z;
                    acc.w += t.w1 * v1.w + t.w2 * v2.w + t.w3 * v3.w + t.w4 * v4.w;
                }
            }
            acc.x /= g.count; acc.y /= g.count; acc.z /= g.count; acc.w /= g.count;
            st4(out + c0, acc);
        }
    }
    };
    for (int ph = ph_begin; ph < ph_end; ph++) do_row(ph);
}

// ---------------------------------------------------------------------------------------
// Round 5: prepared records.  The footprint kernel above spends ~700 instructions per wave on the level mapping
// (sqrt, log2), the RoI geometry (divisions, ceil) and the x-axis weights of the seven bins BEFORE its first load --
// the same values for all seven bin rows of a RoI, recomputed by seven waves; one wave walking all seven rows (RPW)
// shares them but leaves a seventh of the waves to hide the gather latency and was 10-80 % slower.  Here a small
// first launch (one wave per RoI) writes them to a 576-byte record, and the streaming launch (one wave per bin row,
// as before) starts from one coalesced 256-byte load of the x weights and a few scalar loads.
//
//   record of RoI k, 32-bit words:  [0..63]  x weight by lane (lane = 8 * bin + slot)
//                                   [64] level  [65] image  [66] flags: bit ph = bin row ph takes the streaming form
//                                   [67] 1 / sample count (float)
//                                   [68..75] first footprint column of bin b   [76..83] footprint columns of bin b
//                                   [84..91] first footprint row of bin row ph [92..99] footprint rows of bin row ph
//                                   [100..131] y weights, 4 per bin row
// A bin row that does not fit the streaming form (a bin wider than 8 footprint columns, more than 4 footprint rows:
// bins beyond ~3 px, which the level mapping makes rare) takes the sample loop in the reference's own order.
// ---------------------------------------------------------------------------------------
constexpr int REC_WORDS = 144;

__global__ __launch_bounds__(256) void roi_prep_kernel(LevelTable lv, const float* __restrict__ rois, int32_t* __restrict__ recs,
                                                       int32_t* __restrict__ levels_out, int n_rois, int ph_n, int pw_n,
                                                       int sampling_ratio) {
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int k = blockIdx.x * 4 + wave;
    if (k >= n_rois) return;
    const float* roi = rois + (size_t)k * 5;
    const int l = map_roi_level(roi, lv.finest_scale, lv.num_levels);
    const int height = lv.height[l], width = lv.width[l];
    const RoiGeom g = roi_geom(roi, lv.scale[l], 1, ph_n, pw_n, sampling_ratio);
    const bool small = g.gh <= 64 && g.gw <= 64;
    const int q = lane >> 3, jx = lane & 7;
    int x0_all = 0, nx_all = 0;
    float Wx_all = 0.f;
    bool vecx = small && g.gw <= 8 && pw_n <= 8 && ph_n <= 8;
    if (vecx) {
        const float v = g.start_w + q * g.bin_w + (float)(jx + .5f) * g.bin_w / (float)g.gw;
        const AxisSample a = axis_sample(v, width);
        const bool live = jx < g.gw && q < pw_n && a.valid;
        const unsigned mg = (unsigned)((__ballot(live) >> (q * 8)) & 0xffull);
        if (mg) {
            const int f = __ffs((int)mg) - 1, lst = 31 - __clz((int)mg);
            x0_all = __shfl(a.lo, q * 8 + f);
            nx_all = __shfl(a.hi, q * 8 + lst) - x0_all + 1;
        }
        const int mine = x0_all + jx;
        float w = 0.f;
#pragma unroll
        for (int i = 0; i < 8; i++) {
            const int lo = __shfl(a.lo, q * 8 + i), hi = __shfl(a.hi, q * 8 + i);
            const float wl = __shfl(a.wlo, q * 8 + i), wh = __shfl(a.whi, q * 8 + i);
            if ((mg >> i) & 1u) {
                if (lo == mine) w += wl;
                if (hi == mine) w += wh;
            }
        }
        Wx_all = w;
        if (__ballot(nx_all > 8) != 0ull) vecx = false;
    }
    int32_t* rec = recs + (size_t)k * REC_WORDS;
    rec[lane] = __float_as_int(Wx_all);
    if (jx == 0) {
        rec[68 + q] = x0_all;
        rec[76 + q] = q < pw_n ? nx_all : 0;
    }
    unsigned flags = 0;
    for (int ph = 0; ph < ph_n && ph < 8; ph++) {
        int y0 = 0, ny = 0;
        float Wy = 0.f;
        if (small) axis_weights(g.start_h, g.bin_h, ph, g.gh, height, lane, y0, ny, Wy);
        if (lane == 0) {
            rec[84 + ph] = y0;
            rec[92 + ph] = ny;
        }
        if (lane < 4) rec[100 + ph * 4 + lane] = __float_as_int(lane < ny ? Wy : 0.f);
        if (vecx && ny <= 4) flags |= 1u << ph;
    }
    if (lane == 0) {
        rec[64] = l;
        rec[65] = g.batch;
        rec[66] = (int)flags;
        rec[67] = __float_as_int(1.f / g.count);
        if (levels_out) levels_out[k] = l;
    }
}

template <typename T>
__global__ __launch_bounds__(256) void roi_align_fwd_rec_kernel(LevelTable lv, const float* __restrict__ rois,
                                                                const int32_t* __restrict__ recs, T* __restrict__ output,
                                                                int channels, int n_rois, int ph_n, int pw_n, int sampling_ratio,
                                                                const int32_t* __restrict__ perm) {
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    // XCD x (= blockIdx % 8) takes a CONTIGUOUS eighth of the bin rows (one image's maps per L2 at batch 8)
    int bid = blockIdx.x;
    {
        const int nwg = gridDim.x, q = nwg >> 3, r = nwg & 7, xcd = bid & 7, loc = bid >> 3;
        bid = ((xcd < r) ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + loc;
    }
    const long long unit = (long long)bid * 4 + wave;
    if (unit >= (long long)n_rois * ph_n) return;
    const int kk = (int)(unit / ph_n);
    const int ph = (int)(unit - (long long)kk * ph_n);
    const int k = __builtin_amdgcn_readfirstlane(perm ? perm[kk] : kk);
    const int32_t* rec = recs + (size_t)k * REC_WORDS;
    const int l = __builtin_amdgcn_readfirstlane(rec[64]);
    const int batch = __builtin_amdgcn_readfirstlane(rec[65]);
    const unsigned flags = (unsigned)__builtin_amdgcn_readfirstlane(rec[66]);
    const int height = lv.height[l], width = lv.width[l];
    const T* base = reinterpret_cast<const T*>(lv.feat[l]) + (size_t)batch * height * width * channels;
    T* out_row = output + ((size_t)k * ph_n + ph) * pw_n * channels;
    if (!((flags >> ph) & 1u)) {
        // not the streaming form: the sample loop, the reference's operation order
        const float* roi = rois + (size_t)k * 5;
        const RoiGeom g = roi_geom(roi, lv.scale[l], 1, ph_n, pw_n, sampling_ratio);
        for (int pw = 0; pw < pw_n; pw++) {
            T* out = out_row + (size_t)pw * channels;
            for (int c0 = lane * 4; c0 < channels; c0 += 256) {
                float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
                for (int iy = 0; iy < g.gh; iy++) {
                    const float y = g.start_h + ph * g.bin_h + (float)(iy + .5f) * g.bin_h / (float)g.gh;
                    for (int ix = 0; ix < g.gw; ix++) {
                        const float x = g.start_w + pw * g.bin_w + (float)(ix + .5f) * g.bin_w / (float)g.gw;
                        const Tap t = bilinear_tap(y, x, height, width);
                        if (!t.valid) continue;
                        const float4 v1 = ld4(base + (size_t)t.p1 * channels + c0);
                        const float4 v2 = ld4(base + (size_t)t.p2 * channels + c0);
                        const float4 v3 = ld4(base + (size_t)t.p3 * channels + c0);
                        const float4 v4 = ld4(base + (size_t)t.p4 * channels + c0);
                        acc.x += t.w1 * v1.x + t.w2 * v2.x + t.w3 * v3.x + t.w4 * v4.x;
                        acc.y += t.w1 * v1.y + t.w2 * v2.y + t.w3 * v3.y + t.w4 * v4.y;
                        acc.z += t.w1 * v1.z + t.w2 * v2.z + t.w3 * v3.z + t.w4 * v4.z;
                        acc.w += t.w1 * v1.w + t.w2 * v2.w + t.w3 * v3.w + t.w4 * v4.w;
                    }
                }
                acc.x /= g.count; acc.y /= g.count; acc.z /= g.count; acc.w /= g.count;
                st4(out + c0, acc);
            }
        }
        return;
    }
    const float Wx_all = __int_as_float(rec[lane]);
    const float inv = __int_as_float(__builtin_amdgcn_readfirstlane(rec[67]));
    int bx0[8], bnx[8];
    int X0 = 0x7fffffff, X1 = 0;
#pragma unroll
    for (int b = 0; b < 8; b++) {
        bx0[b] = __builtin_amdgcn_readfirstlane(rec[68 + b]);
        bnx[b] = b < pw_n ? __builtin_amdgcn_readfirstlane(rec[76 + b]) : 0;
        if (bnx[b] > 0) { X0 = min(X0, bx0[b]); X1 = max(X1, bx0[b] + bnx[b]); }
    }
    const int y0u = __builtin_amdgcn_readfirstlane(rec[84 + ph]);
    const int ny = __builtin_amdgcn_readfirstlane(rec[92 + ph]);
    float wy[4];
#pragma unroll
    for (int r = 0; r < 4; r++) wy[r] = __int_as_float(__builtin_amdgcn_readfirstlane(rec[100 + ph * 4 + r]));
    constexpr int CB = 2;
    for (int c0 = lane * 4; c0 < channels; c0 += 256) {
        float4 acc[8];
#pragma unroll
        for (int b = 0; b < 8; b++) acc[b] = make_float4(0.f, 0.f, 0.f, 0.f);
        if (ny > 0) {
            const T* r0 = base + (size_t)y0u * width * channels;
            const size_t rstride = (size_t)width * channels;
            const unsigned lane_off = (unsigned)c0;
            for (int cx = X0; cx < X1; cx += CB) {
                float4 v[4][CB];
#pragma unroll
                for (int r = 0; r < 4; r++)
#pragma unroll
                    for (int j = 0; j < CB; j++)
                        if (r < ny && cx + j < X1) v[r][j] = ld4(r0 + r * rstride + (size_t)(cx + j) * channels + lane_off);
#pragma unroll
                for (int j = 0; j < CB; j++) {
                    if (cx + j >= X1) break;
                    float4 cs = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
                    for (int r = 0; r < 4; r++)
                        if (r < ny) {
                            cs.x += wy[r] * v[r][j].x; cs.y += wy[r] * v[r][j].y;
                            cs.z += wy[r] * v[r][j].z; cs.w += wy[r] * v[r][j].w;
                        }
#pragma unroll
                    for (int b = 0; b < 8; b++) {
                        const int jj = cx + j - bx0[b];
                        if (jj >= 0 && jj < bnx[b]) {
                            const float w = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(Wx_all), b * 8 + jj));
                            acc[b].x += w * cs.x; acc[b].y += w * cs.y; acc[b].z += w * cs.z; acc[b].w += w * cs.w;
                        }
                    }
                }
            }
        }
#pragma unroll
        for (int b = 0; b < 8; b++)
            if (b < pw_n) {
                float4 o = acc[b];
                o.x *= inv; o.y *= inv; o.z *= inv; o.w *= inv;
                st4(out_row + (size_t)b * channels + c0, o);
            }
    }
}

int g_roi_exact = 0;     // 1: exact sample-order kernel (bit-identical to the reference's CPU order)
int g_roi_stream_c = 3;  // footprint kernel: bit 0 column streaming over the bin row's patch (else the per-bin loop), bit 1 XCD-contiguous rows

// NHWC backward (avg): same decomposition, atomicAdd of g*w/count to the four corners.
template <bool MULTI>
__global__ __launch_bounds__(256) void roi_align_bwd_nhwc_kernel(
    const float* __restrict__ grad_output, LevelTable lv, const float* __restrict__ rois,
    float* __restrict__ grad_input, int channels, int height, int width, int n_rois, int ph_n,
    int pw_n, float spatial_scale, int sampling_ratio, int aligned) {
    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    const long long bin = (long long)blockIdx.x * 4 + wave;
    const int bins_per_roi = ph_n * pw_n;
    if (bin >= (long long)n_rois * bins_per_roi) return;
    const int k = (int)(bin / bins_per_roi);
    const int r = (int)(bin - (long long)k * bins_per_roi);
    const int ph = r / pw_n, pw = r - ph * pw_n;
    const float* roi = rois + (size_t)k * 5;
    float* gfeat = grad_input;
    if (MULTI) {
        int l = map_roi_level(roi, lv.finest_scale, lv.num_levels);
        gfeat = lv.gfeat[l];
        height = lv.height[l];
        width = lv.width[l];
        spatial_scale = lv.scale[l];
    }
    const RoiGeom g = roi_geom(roi, spatial_scale, aligned, ph_n, pw_n, sampling_ratio);
    float* base = gfeat + (size_t)g.batch * height * width * channels;
    const float* go = grad_output + (size_t)bin * channels;
    for (int c = lane; c < channels; c += 64) {
        const float gv = go[c];
        for (int iy = 0; iy < g.gh; iy++) {
            const float y = g.start_h + ph * g.bin_h + (float)(iy + .5f) * g.bin_h / (float)g.gh;
            for (int ix = 0; ix < g.gw; ix++) {
                const float x = g.start_w + pw * g.bin_w + (float)(ix + .5f) * g.bin_w / (float)g.gw;
                const Tap t = bilinear_tap(y, x, height, width);
                if (!t.valid) continue;
                atomicAdd(base + (size_t)t.p1 * channels + c, gv * t.w1 / g.count);
                atomicAdd(base + (size_t)t.p2 * channels + c, gv * t.w2 / g.count);
                atomicAdd(base + (size_t)t.p3 * channels + c, gv * t.w3 / g.count);
                atomicAdd(base + (size_t)t.p4 * channels + c, gv * t.w4 / g.count);
            }
        }
    }
}

// ---------------------------------------------------------------------------------------
// NCHW forward/backward: one thread per output element (n, c, ph, pw), grid-stride.
// ---------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void roi_align_fwd_nchw_kernel(
    const float* __restrict__ input, const float* __restrict__ rois, float* __restrict__ output,
    float* __restrict__ argmax_y, float* __restrict__ argmax_x, long long total, int channels,
    int height, int width, int ph_n, int pw_n, float spatial_scale, int sampling_ratio,
    int pool_mode, int aligned) {
    for (long long index = (long long)blockIdx.x * blockDim.x + threadIdx.x; index < total;
         index += (long long)gridDim.x * blockDim.x) {
        const int pw = (int)(index % pw_n);
        const int ph = (int)((index / pw_n) % ph_n);
        const int c = (int)((index / pw_n / ph_n) % channels);
        const int k = (int)(index / pw_n / ph_n / channels);
        const float* roi = rois + (size_t)k * 5;
        const RoiGeom g = roi_geom(roi, spatial_scale, aligned, ph_n, pw_n, sampling_ratio);
        const float* in = input + ((size_t)g.batch * channels + c) * height * width;
        float out = 0.f, maxval = -10000.f, my = -1.f, mx = -1.f;
        for (int iy = 0; iy < g.gh; iy++) {
            const float y = g.start_h + ph * g.bin_h + (float)(iy + .5f) * g.bin_h / (float)g.gh;
            for (int ix = 0; ix < g.gw; ix++) {
                const float x = g.start_w + pw * g.bin_w + (float)(ix + .5f) * g.bin_w / (float)g.gw;
                const Tap t = bilinear_tap(y, x, height, width);
                float val = t.w1 * in[t.p1] + t.w2 * in[t.p2] + t.w3 * in[t.p3] + t.w4 * in[t.p4];
                if (val > maxval) { maxval = val; my = y; mx = x; }
                out += val;
            }
        }
        if (pool_mode == 0) {
            output[index] = maxval;
            argmax_y[index] = my;
            argmax_x[index] = mx;
        } else {
            output[index] = out / g.count;
        }
    }
}

__global__ __launch_bounds__(256) void roi_align_bwd_nchw_kernel(
    const float* __restrict__ grad_output, const float* __restrict__ rois,
    float* __restrict__ grad_input, long long total, int channels, int height, int width,
    int ph_n, int pw_n, float spatial_scale, int sampling_ratio, int aligned) {
    for (long long index = (long long)blockIdx.x * blockDim.x + threadIdx.x; index < total;
         index += (long long)gridDim.x * blockDim.x) {
        const int pw = (int)(index % pw_n);
        const int ph = (int)((index / pw_n) % ph_n);
        const int c = (int)((index / pw_n / ph_n) % channels);
        const int k = (int)(index / pw_n / ph_n / channels);
        const float* roi = rois + (size_t)k * 5;
        const RoiGeom g = roi_geom(roi, spatial_scale, aligned, ph_n, pw_n, sampling_ratio);
        float* gi = grad_input + ((size_t)g.batch * channels + c) * height * width;
        const float gv = grad_output[index];
        for (int iy = 0; iy < g.gh; iy++) {
            const float y = g.start_h + ph * g.bin_h + (float)(iy + .5f) * g.bin_h / (float)g.gh;
            for (int ix = 0; ix < g.gw; ix++) {
                const float x = g.start_w + pw * g.bin_w + (float)(ix + .5f) * g.bin_w / (float)g.gw;
                const Tap t = bilinear_tap(y, x, height, width);
                if (!t.valid) continue;
                atomicAdd(gi + t.p1, gv * t.w1 / g.count);
                atomicAdd(gi + t.p2, gv * t.w2 / g.count);
                atomicAdd(gi + t.p3, gv * t.w3 / g.count);
                atomicAdd(gi + t.p4, gv * t.w4 / g.count);
            }
        }
    }
}

int check_common(int batch, int channels, int height, int width, int n_rois, int ph, int pw) {
    if (batch < 0 || channels <= 0 || height <= 0 || width <= 0 || n_rois < 0 || ph <= 0 || pw <= 0)
        return BRCNN_EINVAL;
    return 0;
}

}  // namespace

BRCNN_API int brcnn_roi_align_forward(const float* input, const float* rois, float* output,
                                      float* argmax_y, float* argmax_x, int batch, int channels,
                                      int height, int width, int n_rois, int pooled_h,
                                      int pooled_w, float spatial_scale, int sampling_ratio,
                                      int pool_mode, int aligned, int layout, void* stream) {
    if (check_common(batch, channels, height, width, n_rois, pooled_h, pooled_w)) return BRCNN_EINVAL;
    if (pool_mode != 0 && pool_mode != 1) return BRCNN_EINVAL;
    if (n_rois == 0) return 0;
    if (!input || !rois || !output) return BRCNN_EINVAL;
    hipStream_t s = (hipStream_t)stream;
    const long long total = (long long)n_rois * channels * pooled_h * pooled_w;
    if (layout == BRCNN_LAYOUT_NHWC) {
        if (pool_mode != 1 || (channels & 3)) return BRCNN_EINVAL;
        LevelTable lv = {};
        const long long bins = (long long)n_rois * pooled_h * pooled_w;
        if (g_roi_exact)
            hipLaunchKernelGGL((roi_align_fwd_nhwc_kernel<false, float>), dim3(brcnn_cdiv(bins, 4)), dim3(256), 0,
                               s, input, lv, rois, output, (int32_t*)nullptr, channels, height, width,
                               n_rois, pooled_h, pooled_w, spatial_scale, sampling_ratio, aligned);
        else
            hipLaunchKernelGGL((roi_align_fwd_nhwc_fp_kernel<false, float>), dim3(brcnn_cdiv((long long)n_rois * pooled_h, 4)), dim3(256), 0,
                               s, input, lv, rois, output, (int32_t*)nullptr, channels, height, width,
                               n_rois, pooled_h, pooled_w, spatial_scale, sampling_ratio, aligned, g_roi_stream_c, nullptr);
    } else if (layout == BRCNN_LAYOUT_NCHW) {
        if (pool_mode == 0 && (!argmax_y || !argmax_x)) return BRCNN_EINVAL;
        int grid = brcnn_cdiv(total, 256);
        if (grid > 65536) grid = 65536;
        hipLaunchKernelGGL(roi_align_fwd_nchw_kernel, dim3(grid), dim3(256), 0, s, input, rois,
                           output, argmax_y, argmax_x, total, channels, height, width, pooled_h,
                           pooled_w, spatial_scale, sampling_ratio, pool_mode, aligned);
    } else {
        return BRCNN_EINVAL;
    }
    BRCNN_LAUNCH_CHECK();
    return 0;
}

BRCNN_API int brcnn_roi_align_backward(const float* grad_output, const float* rois,
                                       float* grad_input, int batch, int channels, int height,
                                       int width, int n_rois, int pooled_h, int pooled_w,
                                       float spatial_scale, int sampling_ratio, int aligned,
                                       int layout, void* stream) {
    if (check_common(batch, channels, height, width, n_rois, pooled_h, pooled_w)) return BRCNN_EINVAL;
    if (n_rois == 0) return 0;
    if (!grad_output || !rois || !grad_input) return BRCNN_EINVAL;
    hipStream_t s = (hipStream_t)stream;
    const long long total = (long long)n_rois * channels * pooled_h * pooled_w;
    if (layout == BRCNN_LAYOUT_NHWC) {
        LevelTable lv = {};
        const long long bins = (long long)n_rois * pooled_h * pooled_w;
        hipLaunchKernelGGL(roi_align_bwd_nhwc_kernel<false>, dim3(brcnn_cdiv(bins, 4)), dim3(256), 0,
                           s, grad_output, lv, rois, grad_input, channels, height, width, n_rois,
                           pooled_h, pooled_w, spatial_scale, sampling_ratio, aligned);
    } else if (layout == BRCNN_LAYOUT_NCHW) {
        int grid = brcnn_cdiv(total, 256);
        if (grid > 65536) grid = 65536;
        hipLaunchKernelGGL(roi_align_bwd_nchw_kernel, dim3(grid), dim3(256), 0, s, grad_output, rois,
                           grad_input, total, channels, height, width, pooled_h, pooled_w,
                           spatial_scale, sampling_ratio, aligned);
    } else {
        return BRCNN_EINVAL;
    }
    BRCNN_LAUNCH_CHECK();
    return 0;
}

static int fill_levels(LevelTable& lv, const float* const* feats, float* const* gfeats,
                       const int* heights, const int* widths, const float* scales, int num_levels,
                       float finest_scale) {
    if (num_levels <= 0 || num_levels > BRCNN_MAX_LEVELS || !heights || !widths || !scales)
        return BRCNN_EINVAL;
    lv.num_levels = num_levels;
    lv.finest_scale = finest_scale;
    for (int l = 0; l < num_levels; l++) {
        lv.feat[l] = feats ? feats[l] : nullptr;
        lv.gfeat[l] = gfeats ? gfeats[l] : nullptr;
        lv.height[l] = heights[l];
        lv.width[l] = widths[l];
        lv.scale[l] = scales[l];
        if (heights[l] <= 0 || widths[l] <= 0) return BRCNN_EINVAL;
    }
    return 0;
}

// Visiting order of the RoIs for the footprint kernel: a counting sort by (image, level, 12-row band of the RoI's centre
// on its own level) -- one workgroup, LDS histogram + scan + scatter.  Random proposals of one image touch its whole
// level-0 map (17 MB at 800 x 1344, fp32) again and again through a 4 MB L2; visited band by band, the bin rows an XCD
// has in flight stay inside a few rows of one map.  The order inside a bucket is whatever the atomics give: results do
// not depend on it.
constexpr int ORDER_BANDS = 16, ORDER_BAND_ROWS = 12;
__global__ __launch_bounds__(1024) void roi_order_kernel(const float* __restrict__ rois, int n_rois, LevelTable lv, int batch,
                                                        int32_t* __restrict__ perm) {
    __shared__ int cnt[BRCNN_MAX_IMAGES * BRCNN_MAX_LEVELS * ORDER_BANDS];
    __shared__ int part[1024];
    const int nb = batch * lv.num_levels * ORDER_BANDS;
    auto bucket = [&](int i) {
        const float* roi = rois + (size_t)i * 5;
        const int l = map_roi_level(roi, lv.finest_scale, lv.num_levels);
        int img = (int)roi[0];
        img = img < 0 ? 0 : (img >= batch ? batch - 1 : img);
        int band = (int)((roi[2] + roi[4]) * 0.5f * lv.scale[l]) / ORDER_BAND_ROWS;
        band = band < 0 ? 0 : (band >= ORDER_BANDS ? ORDER_BANDS - 1 : band);
        return (img * lv.num_levels + l) * ORDER_BANDS + band;
    };
    for (int i = threadIdx.x; i < nb; i += 1024) cnt[i] = 0;
    __syncthreads();
    for (int i = threadIdx.x; i < n_rois; i += 1024) atomicAdd(&cnt[bucket(i)], 1);
    __syncthreads();
    // exclusive scan: thread t owns the consecutive buckets [t per, (t + 1) per)
    const int per = (nb + 1023) / 1024;
    int sum = 0;
    for (int j = 0; j < per; j++) {
        const int bi = threadIdx.x * per + j;
        if (bi < nb) sum += cnt[bi];
    }
    part[threadIdx.x] = sum;
    __syncthreads();
    for (int d = 1; d < 1024; d <<= 1) {
        const int v = threadIdx.x >= d ? part[threadIdx.x - d] : 0;
        __syncthreads();
        part[threadIdx.x] += v;
        __syncthreads();
    }
    int run = part[threadIdx.x] - sum;
    for (int j = 0; j < per; j++) {
        const int bi = threadIdx.x * per + j;
        if (bi < nb) { const int c = cnt[bi]; cnt[bi] = run; run += c; }
    }
    __syncthreads();
    for (int i = threadIdx.x; i < n_rois; i += 1024) perm[atomicAdd(&cnt[bucket(i)], 1)] = i;
}

constexpr int ROI_ORDER_MIN_ROIS = 12288;        // brcnn_roi_extract_order_min_rois(): what the caller sizes its scratch by
int g_roi_prep = 0;      // ... (30 / 31): prepared-record form off / on where the caller provides its scratch.  OFF: measured
                         // slower below 1000 RoIs / image (profiles/r05_notes.md)
int g_roi_rpw = 0;       // tuning hook (set_exact(10 / 11 / 17)): rows per wave by the heuristic / 1 / all
int g_roi_order = 1;     // ... (20 / 21 / 22): never / where a workspace is given and the RoI count pays for the sort / always

template <typename T>
static int extract_forward_impl(const void* const* feats_host, const int* heights_host, const int* widths_host,
                                const float* scales_host, int num_levels, const float* rois, void* output,
                                int32_t* levels_out, int batch, int channels, int n_rois, int pooled_h, int pooled_w,
                                int sampling_ratio, float finest_scale, int32_t* order_ws, hipStream_t s,
                                int32_t* prep_ws = nullptr, size_t prep_bytes = 0) {
    LevelTable lv = {};
    if (fill_levels(lv, (const float* const*)feats_host, nullptr, heights_host, widths_host, scales_host,
                    num_levels, finest_scale))
        return BRCNN_EINVAL;
    if (n_rois == 0) return 0;
    if (!rois || !output) return BRCNN_EINVAL;
    if (g_roi_exact) {
        const long long bins = (long long)n_rois * pooled_h * pooled_w;
        hipLaunchKernelGGL((roi_align_fwd_nhwc_kernel<true, T>), dim3(brcnn_cdiv(bins, 4)), dim3(256), 0, s, (const T*)nullptr,
                           lv, rois, (T*)output, levels_out, channels, 0, 0, n_rois, pooled_h, pooled_w, 0.f,
                           sampling_ratio, 1);
        BRCNN_LAUNCH_CHECK();
        return 0;
    }
    // many thousands of RoIs: visit them band by band (r04, 8 images x 2000 RoIs: fabric fetch per launch 1077 -> 318 MB by
    // the PMC counters, time 358-382 -> 344 us; at 1000 / image the gain is inside the run-to-run spread, at 512 / image
    // the ~4 us sort eats it).  One wave per bin row at every size: a wave that
    // walks all seven rows of its RoI shares the level / geometry / x-weight arithmetic but leaves a seventh of the waves
    // to hide the gather latency -- measured 20-80 % slower (hook 17; profiles/r04_notes.md)
    const int32_t* perm = nullptr;
    if (order_ws && batch <= BRCNN_MAX_IMAGES && (g_roi_order == 2 || (g_roi_order == 1 && n_rois >= ROI_ORDER_MIN_ROIS))) {
        hipLaunchKernelGGL(roi_order_kernel, dim3(1), dim3(1024), 0, s, rois, n_rois, lv, batch, order_ws);
        BRCNN_LAUNCH_CHECK();
        perm = order_ws;
    }
    if (prep_ws && g_roi_prep && pooled_h <= 8 && pooled_w <= 8 && prep_bytes >= (size_t)n_rois * REC_WORDS * sizeof(int32_t)) {
        // prepared records: level mapping, geometry and axis weights once per RoI (one small launch), then one wave per
        // bin row that starts from its RoI's record
        hipLaunchKernelGGL(roi_prep_kernel, dim3(brcnn_cdiv((long long)n_rois, 4)), dim3(256), 0, s, lv, rois, prep_ws, levels_out,
                           n_rois, pooled_h, pooled_w, sampling_ratio);
        BRCNN_LAUNCH_CHECK();
        hipLaunchKernelGGL((roi_align_fwd_rec_kernel<T>), dim3(brcnn_cdiv((long long)n_rois * pooled_h, 4)), dim3(256), 0, s, lv,
                           rois, prep_ws, (T*)output, channels, n_rois, pooled_h, pooled_w, sampling_ratio, perm);
        BRCNN_LAUNCH_CHECK();
        return 0;
    }
    const bool all_rows = pooled_h == 7 && g_roi_rpw == 17;
    if (all_rows)
        hipLaunchKernelGGL((roi_align_fwd_nhwc_fp_kernel<true, T, 7>), dim3(brcnn_cdiv((long long)n_rois, 4)), dim3(256), 0, s,
                           (const T*)nullptr, lv, rois, (T*)output, levels_out, channels, 0, 0, n_rois, pooled_h, pooled_w,
                           0.f, sampling_ratio, 1, g_roi_stream_c, perm);
    else
        hipLaunchKernelGGL((roi_align_fwd_nhwc_fp_kernel<true, T, 1>), dim3(brcnn_cdiv((long long)n_rois * pooled_h, 4)),
                           dim3(256), 0, s, (const T*)nullptr, lv, rois, (T*)output, levels_out, channels, 0, 0, n_rois,
                           pooled_h, pooled_w, 0.f, sampling_ratio, 1, g_roi_stream_c, perm);
    BRCNN_LAUNCH_CHECK();
    return 0;
}

// `order_ws`: n_rois int32 of caller-owned scratch for the visiting order of the RoIs (or NULL: RoIs are visited as given)
BRCNN_API int brcnn_roi_extract_forward_ordered(const void* const* feats_host, const int* heights_host,
                                                const int* widths_host, const float* scales_host,
                                                int num_levels, const float* rois, void* output,
                                                int32_t* levels_out, int batch, int channels, int n_rois,
                                                int pooled_h, int pooled_w, int sampling_ratio,
                                                float finest_scale, int dtype, int32_t* order_ws, void* stream) {
    if (!brcnn_elem_ok(dtype)) return BRCNN_EINVAL;
    if (!feats_host || channels <= 0 || (channels & 3) || n_rois < 0 || pooled_h <= 0 || pooled_w <= 0)
        return BRCNN_EINVAL;
    hipStream_t s = (hipStream_t)stream;
    if (dtype == BRCNN_DT_BF16)
        return extract_forward_impl<bf16_t>(feats_host, heights_host, widths_host, scales_host, num_levels, rois, output, levels_out,
                                            batch, channels, n_rois, pooled_h, pooled_w, sampling_ratio, finest_scale, order_ws, s);
    if (dtype == BRCNN_DT_F16)
        return extract_forward_impl<f16_t>(feats_host, heights_host, widths_host, scales_host, num_levels, rois, output, levels_out,
                                           batch, channels, n_rois, pooled_h, pooled_w, sampling_ratio, finest_scale, order_ws, s);
    return extract_forward_impl<float>(feats_host, heights_host, widths_host, scales_host, num_levels, rois, output, levels_out,
                                       batch, channels, n_rois, pooled_h, pooled_w, sampling_ratio, finest_scale, order_ws, s);
}

// bytes of caller-owned scratch the prepared-record form wants (one 576-byte record per RoI); 0 while that form is
// switched off (the default): the caller then passes NULL
BRCNN_API size_t brcnn_roi_extract_prep_workspace_bytes(int n_rois) {
    return (n_rois > 0 && g_roi_prep) ? (size_t)n_rois * REC_WORDS * sizeof(int32_t) : 0;
}

// RoI count from which the library visits the RoIs in band order when `order_ws` is given (what the caller sizes that
// scratch by: below it a NULL `order_ws` costs nothing)
BRCNN_API int brcnn_roi_extract_order_min_rois(void) { return ROI_ORDER_MIN_ROIS; }

// brcnn_roi_extract_forward_ordered + `prep_ws` (brcnn_roi_extract_prep_workspace_bytes(n_rois) bytes, caller-owned, or
// NULL): the per-RoI level mapping / geometry / axis weights are computed once per RoI by a small first launch instead of
// once per bin row inside the gather.  Same results as the footprint form (bit for bit on the bin rows that take the
// streaming form; the rare others run the reference's sample loop).
BRCNN_API int brcnn_roi_extract_forward_prepared(const void* const* feats_host, const int* heights_host,
                                                 const int* widths_host, const float* scales_host,
                                                 int num_levels, const float* rois, void* output,
                                                 int32_t* levels_out, int batch, int channels, int n_rois,
                                                 int pooled_h, int pooled_w, int sampling_ratio,
                                                 float finest_scale, int dtype, int32_t* order_ws, void* prep_ws,
                                                 size_t prep_bytes, void* stream) {
    if (!brcnn_elem_ok(dtype)) return BRCNN_EINVAL;
    if (!feats_host || channels <= 0 || (channels & 3) || n_rois < 0 || pooled_h <= 0 || pooled_w <= 0)
        return BRCNN_EINVAL;
    if (prep_ws && ((uintptr_t)prep_ws & 255)) return BRCNN_EINVAL;
    hipStream_t s = (hipStream_t)stream;
    if (dtype == BRCNN_DT_BF16)
        return extract_forward_impl<bf16_t>(feats_host, heights_host, widths_host, scales_host, num_levels, rois, output, levels_out,
                                            batch, channels, n_rois, pooled_h, pooled_w, sampling_ratio, finest_scale, order_ws, s,
                                            (int32_t*)prep_ws, prep_bytes);
    if (dtype == BRCNN_DT_F16)
        return extract_forward_impl<f16_t>(feats_host, heights_host, widths_host, scales_host, num_levels, rois, output, levels_out,
                                           batch, channels, n_rois, pooled_h, pooled_w, sampling_ratio, finest_scale, order_ws, s,
                                           (int32_t*)prep_ws, prep_bytes);
    return extract_forward_impl<float>(feats_host, heights_host, widths_host, scales_host, num_levels, rois, output, levels_out,
                                       batch, channels, n_rois, pooled_h, pooled_w, sampling_ratio, finest_scale, order_ws, s,
                                       (int32_t*)prep_ws, prep_bytes);
}

BRCNN_API int brcnn_roi_extract_forward(const void* const* feats_host, const int* heights_host,
                                        const int* widths_host, const float* scales_host,
                                        int num_levels, const float* rois, void* output,
                                        int32_t* levels_out, int batch, int channels, int n_rois,
                                        int pooled_h, int pooled_w, int sampling_ratio,
                                        float finest_scale, int dtype, void* stream) {
    return brcnn_roi_extract_forward_ordered(feats_host, heights_host, widths_host, scales_host, num_levels, rois, output,
                                             levels_out, batch, channels, n_rois, pooled_h, pooled_w, sampling_ratio,
                                             finest_scale, dtype, nullptr, stream);
}

BRCNN_API int brcnn_roi_extract_backward(float* const* grad_feats_host, const int* heights_host,
                                         const int* widths_host, const float* scales_host,
                                         int num_levels, const float* rois,
                                         const float* grad_output, int batch, int channels,
                                         int n_rois, int pooled_h, int pooled_w,
                                         int sampling_ratio, float finest_scale, void* stream) {
    if (!grad_feats_host || channels <= 0 || n_rois < 0 || pooled_h <= 0 || pooled_w <= 0)
        return BRCNN_EINVAL;
    LevelTable lv = {};
    if (fill_levels(lv, nullptr, grad_feats_host, heights_host, widths_host, scales_host,
                    num_levels, finest_scale))
        return BRCNN_EINVAL;
    if (n_rois == 0) return 0;
    if (!rois || !grad_output) return BRCNN_EINVAL;
    const long long bins = (long long)n_rois * pooled_h * pooled_w;
    hipLaunchKernelGGL(roi_align_bwd_nhwc_kernel<true>, dim3(brcnn_cdiv(bins, 4)), dim3(256), 0,
                       (hipStream_t)stream, grad_output, lv, rois, (float*)nullptr, channels, 0, 0,
                       n_rois, pooled_h, pooled_w, 0.f, sampling_ratio, 1);
    BRCNN_LAUNCH_CHECK();
    return 0;
}


// ---------------------------------------------------------------------------------------
// Multi-level backward as a GATHER (no atomics, deterministic, every gradient pixel written once).
//   roi_record_kernel: per RoI its (level, image) key and the pixel rectangle its samples can touch.
//   roi_grad_gather_kernel: one workgroup per 8x8 tile of one (level, image) map.  All RoI records are
//     filtered 256 at a time against the tile (ballot compaction keeps RoI order -> fixed summation
//     order); per hit a wave owning 2 x 8 pixels derives the separable footprint weights
//         dX[h, w] += sum_ph sum_pw Wy[ph][h] * Wx[pw][w] * dY[roi, ph, pw] / count,
//         Wy[ph][h] = sum over the bin's samples of the bilinear weight their two taps give row h
//     (28 + 28 lanes compute the weights of the wave's 4 columns / 4 rows; a dY row is loaded once per wave
//     and bin, bins with all-zero weights are skipped) and accumulates 16 pixels x 4 channels per lane.
// The scatter form above issues 4 * gh * gw fp32 atomics per bin and channel (0.07 of HBM peak, order
// dependent); this form reads each dY row ~4x from L2 / MALL and writes each dX pixel once.
struct RoiRec { int key, y01, x01, pad; };
constexpr int GT_TILE = 8;

struct GatherLevels {
    int num_levels, batch;
    int tiles_x[BRCNN_MAX_LEVELS], tiles_y[BRCNN_MAX_LEVELS];
    int blk0[BRCNN_MAX_LEVELS + 1];
    // hit chunks (round 6).  A tile of a COARSE level is touched by most RoIs of that level (level 3 at 800 x 1344: six
    // tiles per image, ~2/3 of the level's RoIs each): its hits are a serial chain of ~6 us each in one workgroup, and the
    // longest chain IS the launch (512 RoIs / image of 450-800 px: 2.0 ms; of 16-110 px: 0.22 ms).  Levels with at most
    // GT_CHUNK_TILES tiles per image get `chunks[l]` workgroups per tile; workgroup c takes the tile's hits c, c + chunks,
    // ... (in RoI order) and leaves an fp32 partial; roi_grad_chunk_sum_kernel adds the partials in chunk order: a fixed
    // association, so the result is reproducible (not the one-workgroup chain's bits).
    // A tile with at most GT_DIRECT_HITS hits is finished by its chunk-0 workgroup alone (every chunk workgroup scans the
    // same records and sees the same count, so the decision needs no communication): no partials, the other chunks exit.
    int chunks[BRCNN_MAX_LEVELS];
    int ptile0[BRCNN_MAX_LEVELS + 1];       // first partial-tile index of the level's tiles (chunked levels only)
    float* partial;                         // [partial tile][chunk][64 pixels][channels] fp32
    int* direct;                            // [partial tile]: 1 = written by chunk 0 itself (zeroed before the launch)
};
constexpr int GT_CHUNK_TILES = 32;
constexpr int GT_DIRECT_HITS = 12;

__global__ __launch_bounds__(256) void roi_record_kernel(const float* __restrict__ rois, int n_rois, LevelTable lv, int batch,
                                                        int ph_n, int pw_n, int sampling_ratio, RoiRec* __restrict__ recs,
                                                        int* __restrict__ ranges) {
    const int k = blockIdx.x * 256 + threadIdx.x;
    const bool live = k < n_rois;
    const float* roi = rois + (size_t)(live ? k : 0) * 5;
    const int l = map_roi_level(roi, lv.finest_scale, lv.num_levels);
    const RoiGeom g = roi_geom(roi, lv.scale[l], 1, ph_n, pw_n, sampling_ratio);
    const int H = lv.height[l], W = lv.width[l];
    RoiRec r;
    r.pad = 0;
    const float end_h = g.start_h + g.bin_h * (float)ph_n, end_w = g.start_w + g.bin_w * (float)pw_n;
    const bool empty = g.gh <= 0 || g.gw <= 0 || g.batch < 0 || g.batch >= batch || end_h < -1.f || g.start_h > (float)H ||
                       end_w < -1.f || g.start_w > (float)W || !(end_h == end_h) || !(end_w == end_w);
    int rb = -1;        // image whose [first, last) RoI index range this record widens
    if (empty) {
        r.key = -1; r.y01 = r.x01 = 0;
    } else {
        const int y0 = max(0, (int)floorf(g.start_h)), y1 = min(H - 1, (int)floorf(end_h) + 1);
        const int x0 = max(0, (int)floorf(g.start_w)), x1 = min(W - 1, (int)floorf(end_w) + 1);
        r.key = l * batch + g.batch;
        r.y01 = (y0 << 16) | max(y1, 0);
        r.x01 = (x0 << 16) | max(x1, 0);
        if (y1 < y0 || x1 < x0) r.key = -1;
        if (r.key >= 0) rb = g.batch;
    }
    // the image's RoI index range: RoIs arrive image by image (bbox2roi), so a tile of image b scans [first, last) instead
    // of every record; any other order only widens the range.  One atomic pair per wave and image (round 6: one pair
    // per RoI on `batch` addresses serialised to ~60 us for 4096 RoIs): k grows with the lane, so the lanes of one
    // image contribute their lowest / highest lane's index.
    if (ranges) {
        if (!live) rb = -1;
        const int lane = threadIdx.x & 63;
        unsigned long long todo = __ballot(rb >= 0);
        while (todo) {
            const int leader = __ffsll((long long)todo) - 1;
            const int lb = __shfl(rb, leader, 64);
            const unsigned long long same = __ballot(rb == lb);
            if (lane == leader) {
                const int kb = k - lane;
                atomicMin(&ranges[lb], kb + __ffsll((long long)same) - 1);
                atomicMax(&ranges[1024 + lb], kb + 64 - __clzll((long long)same));
            }
            todo &= ~same;
        }
    }
    if (!live) return;
    recs[k] = r;
}

template <typename T>       // element type of dY and of the gradient maps (fp32, or the 16-bit compute dtype in training)
__global__ __launch_bounds__(256) void roi_grad_gather_kernel(const T* __restrict__ grad_output, LevelTable lv,
                                                             GatherLevels gl, const float* __restrict__ rois,
                                                             const RoiRec* __restrict__ recs, const int* __restrict__ ranges,
                                                             int n_rois, int channels, int ph_n, int pw_n, int sampling_ratio) {
    constexpr int MAXHIT = 512;
    __shared__ int s_hits[MAXHIT];
    __shared__ int s_wcnt[4];
    __shared__ int s_nhit;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    int l = 0;
#pragma unroll
    for (int i = 1; i < BRCNN_MAX_LEVELS; i++)
        if (i < gl.num_levels && (int)blockIdx.x >= gl.blk0[i]) l = i;
    const int H = lv.height[l], W = lv.width[l];
    int t = blockIdx.x - gl.blk0[l];
    const int nchunk = gl.chunks[l];
    const int chunk = t % nchunk;           // (the chunk workgroups of a tile are neighbours)
    t /= nchunk;
    const int tile_lin = t;                 // tile index inside the level (image-major)
    const int per_img = gl.tiles_x[l] * gl.tiles_y[l];
    const int b = t / per_img;
    t -= b * per_img;
    const int ty = t / gl.tiles_x[l], tx = t - ty * gl.tiles_x[l];
    const int h0 = ty * GT_TILE, w0 = tx * GT_TILE;                 // tile origin
    const int h1 = min(H - 1, h0 + GT_TILE - 1), w1 = min(W - 1, w0 + GT_TILE - 1);
    const int key = l * gl.batch + b;
    const int wr0 = h0 + 4 * (wave >> 1), wc0 = w0 + 4 * (wave & 1);  // this wave's 4 x 4 pixel quadrant
    T* gout = reinterpret_cast<T*>(lv.gfeat[l]) + (size_t)b * H * W * channels;
    const float scale = lv.scale[l];

    for (int cbase = 0; cbase < channels; cbase += 256) {
        const int c = cbase + lane * 4;
        const bool c_ok = c < channels;
        float4 acc[16];
#pragma unroll
        for (int p = 0; p < 16; p++) acc[p] = make_float4(0.f, 0.f, 0.f, 0.f);
        // (the records of this image only: same hits in the same order as a scan over all records)
        const int r_lo = ranges ? max(ranges[b], 0) : 0, r_hi = ranges ? min(ranges[1024 + b], n_rois) : n_rois;
        int hit0 = 0;                       // index (over the whole scan) of the first hit of the running batch
        bool direct = false;
        if (nchunk > 1 && r_lo >= r_hi) {   // no RoI of this image at all: chunk 0 writes the zeros (+ addend)
            if (chunk != 0) return;
            direct = true;
        }
        for (int base = r_lo; base < r_hi; ) {
            // ---- collect up to MAXHIT RoIs (in index order) that touch this tile
            if (tid == 0) s_nhit = 0;
            __syncthreads();
            int scanned = base;
            while (scanned < r_hi) {
                const int k = scanned + tid;
                bool hit = false;
                if (k < r_hi) {
                    const RoiRec r = recs[k];
                    if (r.key == key) {
                        const int y0 = r.y01 >> 16, y1 = r.y01 & 0xffff, x0 = r.x01 >> 16, x1 = r.x01 & 0xffff;
                        hit = !(y1 < h0 || y0 > h1 || x1 < w0 || x0 > w1);
                    }
                }
                const unsigned long long m = __ballot(hit);
                if (lane == 0) s_wcnt[wave] = __popcll(m);
                __syncthreads();
                const int total = s_wcnt[0] + s_wcnt[1] + s_wcnt[2] + s_wcnt[3];
                const int cur = s_nhit;
                if (cur + total > MAXHIT) { __syncthreads(); break; }       // next round takes this chunk
                int off = cur;
                for (int w_ = 0; w_ < wave; w_++) off += s_wcnt[w_];
                if (hit) s_hits[off + __popcll(m & ((1ull << lane) - 1ull))] = k;
                __syncthreads();
                if (tid == 0) s_nhit = cur + total;
                scanned += 256;
                __syncthreads();
            }
            const int nhit = s_nhit;
            if (nchunk > 1 && base == r_lo) {       // first batch: does it hold every hit of the tile, and few enough?
                direct = scanned >= r_hi && nhit <= GT_DIRECT_HITS;
                if (direct && chunk != 0) return;   // (workgroup-uniform: nothing was written, no barrier is pending)
            }
            base = scanned;
            // ---- accumulate the hits into the wave's 16 pixels
            for (int hi = 0; hi < nhit; hi++) {
                if (nchunk > 1 && !direct && (hit0 + hi) % nchunk != chunk) continue;       // (workgroup-uniform)
                const int k = s_hits[hi];
                const float* roi = rois + (size_t)k * 5;
                const RoiGeom g = roi_geom(roi, scale, 1, ph_n, pw_n, sampling_ratio);
                // Wx[col][pw] on lanes 0..27, Wy[row][ph] on lanes 32..59 (4 columns / rows x 7 bins each).  The weights
                // STAY in the lanes that computed them: the loops below fetch them with v_readlane (compile-time lane
                // numbers) and the zero tests are one ballot.  (Until round 6 they went through LDS: a write, a wave
                // barrier, and -- what the listing showed -- 28 + 28 broadcast reads per hit, each behind its own wait
                // because the `||` chain of the zero tests short-circuits.)
                float w = 0.f;
                {
                    const int half = lane >> 5, li = lane & 31;
                    const int pix = li / 7, bin = li - pix * 7;
                    if (li < 28 && bin < (half ? ph_n : pw_n)) {
                        const int p = (half ? wr0 : wc0) + pix;
                        const float start = half ? g.start_h : g.start_w, bsz = half ? g.bin_h : g.bin_w;
                        const int gn = half ? g.gh : g.gw, size = half ? H : W;
                        for (int i = 0; i < gn; i++) {
                            const float v = start + bin * bsz + (float)(i + .5f) * bsz / (float)gn;
                            const AxisSample a = axis_sample(v, size);
                            if (a.valid) {
                                if (a.lo == p) w += a.wlo;
                                if (a.hi == p) w += a.whi;
                            }
                        }
                        if (half) w = w / g.count;
                    }
                }
                const int wbits = __float_as_int(w);
                // bins with a non-zero weight on any of the quadrant's four columns / rows (wave-uniform)
                const unsigned long long nz = __ballot(w != 0.f);
                const unsigned nzx = (unsigned)(nz & 0xfffffffull), nzy = (unsigned)((nz >> 32) & 0xfffffffull);
                const unsigned pwm = (nzx | (nzx >> 7) | (nzx >> 14) | (nzx >> 21)) & 0x7fu;
                const unsigned phm = (nzy | (nzy >> 7) | (nzy >> 14) | (nzy >> 21)) & 0x7fu;
                if (!pwm || !phm) continue;
                // (a lane beyond the channel count reads channel 0 and never stores: no branch around the loads)
                const T* go = grad_output + (size_t)k * ph_n * pw_n * channels + (c_ok ? c : 0);
#pragma unroll
                for (int ph = 0; ph < 7; ph++) {
                    if (!((phm >> ph) & 1u)) continue;          // (also every ph >= ph_n: those lanes hold 0)
                    float wys[4];
#pragma unroll
                    for (int r = 0; r < 4; r++) wys[r] = __int_as_float(__builtin_amdgcn_readlane(wbits, 32 + r * 7 + ph));
                    // the dY rows of ALL bins of this bin row are requested together, unconditionally and in straight-line
                    // code: seven loads in flight per wave (round 6, second pass: the first form had `c_ok ? load : 0`,
                    // which the compiler turned into seven exec-masked blocks with a vmcnt(0) each -- seven memory latencies
                    // in a row per bin row, found by the loads-in-flight scan of check_isa), then accumulated in bin order
                    // where the x weights are not all zero: the summation order is unchanged
                    float4 gv[7];
#pragma unroll
                    for (int j = 0; j < 7; j++) {
                        const int pwj = j < pw_n ? j : pw_n - 1;
                        gv[j] = ld4(go + (size_t)(ph * pw_n + pwj) * channels);
                    }
                    // separable accumulation (round 6: the gather was VALU-bound -- 16 pixels x 7 bins x 4 channels of
                    // multiply-adds per bin row and lane, 560 VALU operations, most of them on zero weights):
                    //   t[q]  = sum_pw Wx[q][pw] * dY[ph][pw]       (the wave's four columns: 4 x 7 x 4 FMAs)
                    //   acc[r][q] += Wy[r][ph] * t[q]                (16 x 4 FMAs)
                    // -- the products Wy Wx dY summed over pw first, then over ph: the same terms in another association
                    // (fp32 round-off against the one-product-per-term form; the tests compare with the C oracle at 1e-5)
                    float4 tq[4];
#pragma unroll
                    for (int q = 0; q < 4; q++) tq[q] = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
                    for (int j = 0; j < 7; j++) {
                        if (!((pwm >> j) & 1u)) continue;
#pragma unroll
                        for (int q = 0; q < 4; q++) {
                            const float wx = __int_as_float(__builtin_amdgcn_readlane(wbits, q * 7 + j));
                            tq[q].x += wx * gv[j].x;
                            tq[q].y += wx * gv[j].y;
                            tq[q].z += wx * gv[j].z;
                            tq[q].w += wx * gv[j].w;
                        }
                    }
#pragma unroll
                    for (int r = 0; r < 4; r++)
#pragma unroll
                        for (int q = 0; q < 4; q++) {
                            acc[r * 4 + q].x += wys[r] * tq[q].x;
                            acc[r * 4 + q].y += wys[r] * tq[q].y;
                            acc[r * 4 + q].z += wys[r] * tq[q].z;
                            acc[r * 4 + q].w += wys[r] * tq[q].w;
                        }
                }
            }
            hit0 += nhit;
            __syncthreads();
        }
        if (nchunk > 1 && direct && tid == 0 && cbase == 0) gl.direct[gl.ptile0[l] + tile_lin] = 1;
        if (nchunk > 1 && !direct) {
            if (c_ok) {
                float* pt = gl.partial + (((size_t)(gl.ptile0[l] + tile_lin) * nchunk + chunk) * 64) * channels + c;
#pragma unroll
                for (int p = 0; p < 16; p++) {
                    const int pix = (4 * (wave >> 1) + (p >> 2)) * GT_TILE + 4 * (wave & 1) + (p & 3);
                    *reinterpret_cast<float4*>(pt + (size_t)pix * channels) = acc[p];
                }
            }
            continue;
        }
        if (c_ok) {
            // lv.feat[l] (the `addends` of brcnn_roi_extract_backward_gather_add): a gradient of the same map from another
            // branch, added in fp32 before the one rounding of the store
            const T* gadd = lv.feat[l] ? reinterpret_cast<const T*>(lv.feat[l]) + (size_t)b * H * W * channels : nullptr;
#pragma unroll
            for (int p = 0; p < 16; p++) {
                const int py = wr0 + (p >> 2), px = wc0 + (p & 3);
                if (py < H && px < W) {
                    const size_t off = ((size_t)py * W + px) * channels + c;
                    float4 v = acc[p];
                    if (gadd) {
                        const float4 a = ld4(gadd + off);
                        v.x += a.x; v.y += a.y; v.z += a.z; v.w += a.w;
                    }
                    st4(gout + off, v);
                }
            }
        }
    }
}

// second stage of the chunked levels: one workgroup per tile, partials added in chunk order (+ the level's addend)
BRCNN_API size_t brcnn_roi_extract_backward_workspace_bytes(int n_rois);
static int g_roi_gather_chunks = -1;      // tuning hook (brcnn_roi_align_set_exact(40 + n)): -1 heuristic, n chunks per coarse tile

template <typename T>
__global__ __launch_bounds__(256) void roi_grad_chunk_sum_kernel(LevelTable lv, GatherLevels gl, int channels) {
    int l = 0;
#pragma unroll
    for (int i = 1; i < BRCNN_MAX_LEVELS; i++)
        if (i < gl.num_levels && (int)blockIdx.x >= gl.ptile0[i]) l = i;
    if (gl.direct[blockIdx.x]) return;              // finished by its chunk-0 workgroup
    const int nchunk = gl.chunks[l];
    const int H = lv.height[l], W = lv.width[l];
    const int tile_lin = blockIdx.x - gl.ptile0[l];
    const int per_img = gl.tiles_x[l] * gl.tiles_y[l];
    const int b = tile_lin / per_img, t = tile_lin - b * per_img;
    const int h0 = (t / gl.tiles_x[l]) * GT_TILE, w0 = (t % gl.tiles_x[l]) * GT_TILE;
    T* gout = reinterpret_cast<T*>(lv.gfeat[l]) + (size_t)b * H * W * channels;
    const T* gadd = lv.feat[l] ? reinterpret_cast<const T*>(lv.feat[l]) + (size_t)b * H * W * channels : nullptr;
    const float* pt = gl.partial + ((size_t)(gl.ptile0[l] + tile_lin) * nchunk * 64) * channels;
    const int cv = channels >> 2;                   // float4 columns
    for (int i = threadIdx.x; i < 64 * cv; i += 256) {
        const int pix = i / cv, c = (i - pix * cv) * 4;
        const int py = h0 + pix / GT_TILE, px = w0 + pix % GT_TILE;
        if (py >= H || px >= W) continue;
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        for (int k = 0; k < nchunk; k++) {
            const float4 a = *reinterpret_cast<const float4*>(pt + ((size_t)k * 64 + pix) * channels + c);
            v.x += a.x; v.y += a.y; v.z += a.z; v.w += a.w;
        }
        const size_t off = ((size_t)py * W + px) * channels + c;
        if (gadd) {
            const float4 a = ld4(gadd + off);
            v.x += a.x; v.y += a.y; v.z += a.z; v.w += a.w;
        }
        st4(gout + off, v);
    }
}

// hit chunks per tile of the coarse levels for `n_rois` RoIs over `batch` images (1: the one-workgroup form)
static int gather_chunks(int n_rois, int batch) {
    if (g_roi_gather_chunks >= 0) return g_roi_gather_chunks < 1 ? 1 : g_roi_gather_chunks;
    const int per_img = n_rois / (batch > 0 ? batch : 1);
    int ch = per_img / 48;
    if (ch > 16) ch = 16;
    return ch < 2 ? 1 : ch;
}

BRCNN_API size_t brcnn_roi_extract_backward_workspace_bytes_ex(int n_rois, int batch, int channels, int num_levels,
                                                               const int* heights_host, const int* widths_host) {
    size_t bytes = brcnn_roi_extract_backward_workspace_bytes(n_rois);
    const int ch = gather_chunks(n_rois, batch);
    if (ch > 1 && heights_host && widths_host && num_levels > 0 && num_levels <= BRCNN_MAX_LEVELS && channels > 0) {
        size_t tiles = 0;
        for (int l = 0; l < num_levels; l++) {
            const size_t per_img = (size_t)((heights_host[l] + GT_TILE - 1) / GT_TILE) * ((widths_host[l] + GT_TILE - 1) / GT_TILE);
            if (per_img <= (size_t)GT_CHUNK_TILES) tiles += per_img * batch;
        }
        bytes = ((bytes + 255) & ~(size_t)255) + ((tiles * sizeof(int) + 255) & ~(size_t)255) + tiles * ch * 64 * channels * sizeof(float);
    }
    return bytes;
}

BRCNN_API size_t brcnn_roi_extract_backward_workspace_bytes(int n_rois) {
    // records + per-image [first, last) RoI index ranges (2 x BRCNN_GATHER_MAX_BATCH ints)
    return (size_t)(n_rois > 0 ? n_rois : 1) * sizeof(RoiRec) + 256 + 2 * 1024 * sizeof(int);
}

BRCNN_API int brcnn_roi_extract_backward_gather(void* const* grad_feats_host, const int* heights_host,
                                                const int* widths_host, const float* scales_host, int num_levels,
                                                const float* rois, const void* grad_output, int batch, int channels,
                                                int n_rois, int pooled_h, int pooled_w, int sampling_ratio,
                                                float finest_scale, void* workspace, size_t workspace_bytes,
                                                int dtype, void* stream) {
    return brcnn_roi_extract_backward_gather_add(grad_feats_host, nullptr, heights_host, widths_host, scales_host, num_levels,
                                                 rois, grad_output, batch, channels, n_rois, pooled_h, pooled_w,
                                                 sampling_ratio, finest_scale, workspace, workspace_bytes, dtype, stream);
}

BRCNN_API int brcnn_roi_extract_backward_gather_add(void* const* grad_feats_host, const void* const* addends_host,
                                                    const int* heights_host, const int* widths_host,
                                                    const float* scales_host, int num_levels, const float* rois,
                                                    const void* grad_output, int batch, int channels, int n_rois,
                                                    int pooled_h, int pooled_w, int sampling_ratio, float finest_scale,
                                                    void* workspace, size_t workspace_bytes, int dtype, void* stream) {
    if (!brcnn_elem_ok(dtype)) return BRCNN_EINVAL;
    if (!grad_feats_host || channels <= 0 || (channels & 3) || n_rois < 0 || pooled_h <= 0 || pooled_w <= 0 || pooled_h > 7 ||
        pooled_w > 7 || batch <= 0 || !workspace || workspace_bytes < (size_t)(n_rois > 0 ? n_rois : 1) * sizeof(RoiRec))
        return BRCNN_EINVAL;
    LevelTable lv = {};
    if (fill_levels(lv, reinterpret_cast<const float* const*>(addends_host), reinterpret_cast<float* const*>(grad_feats_host),
                    heights_host, widths_host, scales_host, num_levels, finest_scale))
        return BRCNN_EINVAL;
    if (n_rois > 0 && (!rois || !grad_output)) return BRCNN_EINVAL;
    GatherLevels gl = {};
    gl.num_levels = num_levels;
    gl.batch = batch;
    int blk = 0;
    for (int l = 0; l < num_levels; l++) {
        if (!grad_feats_host[l] || heights_host[l] >= 65536 || widths_host[l] >= 65536) return BRCNN_EINVAL;
        gl.tiles_y[l] = (heights_host[l] + GT_TILE - 1) / GT_TILE;
        gl.tiles_x[l] = (widths_host[l] + GT_TILE - 1) / GT_TILE;
        gl.blk0[l] = blk;
        blk += gl.tiles_x[l] * gl.tiles_y[l] * batch;
    }
    for (int l = num_levels; l <= BRCNN_MAX_LEVELS; l++) gl.blk0[l] = blk;
    // hit chunks for the coarse levels, when the caller's workspace holds their partials (.._workspace_bytes_ex)
    for (int l = 0; l < BRCNN_MAX_LEVELS; l++) gl.chunks[l] = 1;
    const size_t base_bytes = (brcnn_roi_extract_backward_workspace_bytes(n_rois) + 255) & ~(size_t)255;
    int ptiles = 0;
    {
        const int ch = gather_chunks(n_rois, batch);
        size_t tiles = 0;
        for (int l = 0; l < num_levels; l++)
            if (gl.tiles_x[l] * gl.tiles_y[l] <= GT_CHUNK_TILES) tiles += (size_t)gl.tiles_x[l] * gl.tiles_y[l] * batch;
        const size_t flag_bytes = (tiles * sizeof(int) + 255) & ~(size_t)255;
        if (ch > 1 && n_rois > 0 && tiles > 0 &&
            workspace_bytes >= base_bytes + flag_bytes + tiles * ch * 64 * channels * sizeof(float)) {
            gl.direct = (int*)((char*)workspace + base_bytes);
            gl.partial = (float*)((char*)workspace + base_bytes + flag_bytes);
            BRCNN_HIP_CHECK(hipMemsetAsync(gl.direct, 0, tiles * sizeof(int), (hipStream_t)stream));
            blk = 0;
            for (int l = 0; l < num_levels; l++) {
                const int per_level = gl.tiles_x[l] * gl.tiles_y[l] * batch;
                const bool coarse = gl.tiles_x[l] * gl.tiles_y[l] <= GT_CHUNK_TILES;
                gl.chunks[l] = coarse ? ch : 1;
                gl.ptile0[l] = ptiles;
                if (coarse) ptiles += per_level;
                gl.blk0[l] = blk;
                blk += per_level * gl.chunks[l];
            }
            for (int l = num_levels; l <= BRCNN_MAX_LEVELS; l++) { gl.blk0[l] = blk; gl.ptile0[l] = ptiles; }
            // (levels that are not chunked keep ptile0 = the running count: the sum kernel's level search only sees
            // chunked tiles because a fine level adds none)
        }
    }
    hipStream_t s = (hipStream_t)stream;
    RoiRec* recs = (RoiRec*)workspace;
    int* ranges = nullptr;
    const size_t rec_bytes = ((size_t)(n_rois > 0 ? n_rois : 1) * sizeof(RoiRec) + 255) & ~(size_t)255;
    if (batch <= 1024 && workspace_bytes >= rec_bytes + 2 * 1024 * sizeof(int)) {
        ranges = (int*)((char*)workspace + rec_bytes);
        BRCNN_HIP_CHECK(hipMemsetAsync(ranges, 0x7f, 1024 * sizeof(int), s));          // first = large
        BRCNN_HIP_CHECK(hipMemsetAsync(ranges + 1024, 0, 1024 * sizeof(int), s));      // last = 0
    }
    if (n_rois > 0) {
        hipLaunchKernelGGL(roi_record_kernel, dim3(brcnn_cdiv(n_rois, 256)), dim3(256), 0, s, rois, n_rois, lv, batch,
                           pooled_h, pooled_w, sampling_ratio, recs, ranges);
        BRCNN_LAUNCH_CHECK();
    }
    if (dtype == BRCNN_DT_F32)
        hipLaunchKernelGGL(roi_grad_gather_kernel<float>, dim3(blk), dim3(256), 0, s, (const float*)grad_output, lv, gl, rois,
                           recs, ranges, n_rois, channels, pooled_h, pooled_w, sampling_ratio);
    else if (dtype == BRCNN_DT_BF16)
        hipLaunchKernelGGL(roi_grad_gather_kernel<bf16_t>, dim3(blk), dim3(256), 0, s, (const bf16_t*)grad_output, lv, gl, rois,
                           recs, ranges, n_rois, channels, pooled_h, pooled_w, sampling_ratio);
    else
        hipLaunchKernelGGL(roi_grad_gather_kernel<f16_t>, dim3(blk), dim3(256), 0, s, (const f16_t*)grad_output, lv, gl, rois,
                           recs, ranges, n_rois, channels, pooled_h, pooled_w, sampling_ratio);
    BRCNN_LAUNCH_CHECK();
    if (ptiles > 0) {
        if (dtype == BRCNN_DT_F32)
            hipLaunchKernelGGL(roi_grad_chunk_sum_kernel<float>, dim3(ptiles), dim3(256), 0, s, lv, gl, channels);
        else if (dtype == BRCNN_DT_BF16)
            hipLaunchKernelGGL(roi_grad_chunk_sum_kernel<bf16_t>, dim3(ptiles), dim3(256), 0, s, lv, gl, channels);
        else
            hipLaunchKernelGGL(roi_grad_chunk_sum_kernel<f16_t>, dim3(ptiles), dim3(256), 0, s, lv, gl, channels);
        BRCNN_LAUNCH_CHECK();
    }
    return 0;
}

BRCNN_API int brcnn_roi_align_set_exact(int exact) {
    // 0: footprint kernel (column streaming, XCD-contiguous bin rows), 1: exact sample order, 2: footprint kernel with
    // the per-bin loop and round-robin rows (the r02 form), 3: column streaming with round-robin rows
    // 10 / 11 / 17: bin rows per wavefront by the heuristic / one / all seven; 20 / 21 / 22: RoI visiting order off / by the heuristic / forced
    if (exact == 10 || exact == 11 || exact == 17) { g_roi_rpw = exact == 10 ? 0 : exact; return 0; }
    if (exact >= 20 && exact <= 22) { g_roi_order = exact - 20; return 0; }
    if (exact == 30 || exact == 31) { g_roi_prep = exact - 30; return 0; }
    if (exact >= 39 && exact <= 56) { g_roi_gather_chunks = exact - 40; return 0; }    // 39: heuristic, 40 / 41: off, 42..56: chunks per coarse tile
    g_roi_exact = exact == 1 ? 1 : 0;
    g_roi_stream_c = exact == 2 ? 0 : exact == 3 ? 1 : 3;
    return 0;
}

// ---- the policy switches as one documented struct (include/brcnn_hip.h: brcnn_tuning) ---------------------------
namespace brcnn_conv {
int tuning_get_stream_k(); int tuning_get_split_k(); int tuning_get_eight_phase_16(); int tuning_get_persistent_1x1();
int tuning_get_eight_phase_f32(); int tuning_get_wgrad_slabs(); int tuning_get_wgrad_generation_percent();
int tuning_get_wgrad_eight_phase(); int tuning_get_wgrad_reduce_in_launch(); int tuning_get_wgrad_cu_percent();
}  // namespace brcnn_conv

BRCNN_API int brcnn_get_tuning(brcnn_tuning* t) {
    if (!t || t->size != (int)sizeof(brcnn_tuning)) return BRCNN_EINVAL;
    t->conv_stream_k = brcnn_conv::tuning_get_stream_k();
    t->conv_split_k = brcnn_conv::tuning_get_split_k();
    t->conv_eight_phase_16bit = brcnn_conv::tuning_get_eight_phase_16();
    t->conv_persistent_1x1 = brcnn_conv::tuning_get_persistent_1x1();
    t->conv_eight_phase_f32 = brcnn_conv::tuning_get_eight_phase_f32();
    t->wgrad_slab_reduction = brcnn_conv::tuning_get_wgrad_slabs();
    t->wgrad_eight_phase = brcnn_conv::tuning_get_wgrad_eight_phase();
    t->wgrad_reduce_in_launch = brcnn_conv::tuning_get_wgrad_reduce_in_launch();
    t->wgrad_generation_percent = brcnn_conv::tuning_get_wgrad_generation_percent();
    t->wgrad_eight_phase_cu_percent = brcnn_conv::tuning_get_wgrad_cu_percent();
    t->roi_exact_order = g_roi_exact;
    t->roi_rows_per_wave = g_roi_rpw == 0 ? 0 : (g_roi_rpw == 17 ? 7 : 1);
    t->roi_visit_order = g_roi_order;
    t->roi_prepared_records = g_roi_prep;
    return 0;
}

BRCNN_API int brcnn_set_tuning(const brcnn_tuning* t) {
    if (!t || t->size != (int)sizeof(brcnn_tuning)) return BRCNN_EINVAL;
    auto in = [](int v, int lo, int hi) { return v >= lo && v <= hi; };
    if (!in(t->conv_stream_k, 0, 2) || !in(t->conv_split_k, 0, 2) || !in(t->conv_eight_phase_16bit, 0, 1) ||
        !in(t->conv_persistent_1x1, 0, 2) ||
        !(in(t->conv_eight_phase_f32, 0, 2) || t->conv_eight_phase_f32 == 128 || t->conv_eight_phase_f32 == 256) ||
        !in(t->wgrad_slab_reduction, 0, 1) || !in(t->wgrad_eight_phase, 0, 2) || !in(t->wgrad_reduce_in_launch, 0, 1) ||
        !in(t->wgrad_generation_percent, 10, 400) || !in(t->wgrad_eight_phase_cu_percent, 10, 400) ||
        !in(t->roi_exact_order, 0, 1) || !(t->roi_rows_per_wave == 0 || t->roi_rows_per_wave == 1 || t->roi_rows_per_wave == 7) ||
        !in(t->roi_visit_order, 0, 2) || !in(t->roi_prepared_records, 0, 1))
        return BRCNN_EINVAL;
    int rc = 0;
    rc |= brcnn_conv_set_tile_bf16(-3 - t->conv_stream_k);
    rc |= brcnn_conv_set_tile_bf16(-8 - t->conv_split_k);
    rc |= brcnn_conv_set_tile_bf16(-6 - t->conv_eight_phase_16bit);
    rc |= brcnn_conv_set_tile_bf16(-15 - t->conv_persistent_1x1);
    rc |= brcnn_conv_set_tile(-2, t->conv_eight_phase_f32);
    rc |= brcnn_conv_set_tile_wgrad_bf16(10 + t->wgrad_slab_reduction);
    rc |= brcnn_conv_set_tile_wgrad_bf16(20 + t->wgrad_eight_phase);
    rc |= brcnn_conv_set_tile_wgrad_bf16(30 + t->wgrad_reduce_in_launch);
    rc |= brcnn_conv_set_tile_wgrad_bf16(2000 + t->wgrad_generation_percent);
    rc |= brcnn_conv_set_tile_wgrad_bf16(4000 + t->wgrad_eight_phase_cu_percent);
    rc |= brcnn_roi_align_set_exact(t->roi_exact_order);
    rc |= brcnn_roi_align_set_exact(t->roi_rows_per_wave == 0 ? 10 : (t->roi_rows_per_wave == 7 ? 17 : 11));
    rc |= brcnn_roi_align_set_exact(20 + t->roi_visit_order);
    rc |= brcnn_roi_align_set_exact(30 + t->roi_prepared_records);
    return rc ? BRCNN_EINVAL : 0;
}
