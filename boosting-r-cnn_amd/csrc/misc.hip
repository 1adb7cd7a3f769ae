// Small NHWC kernels around the conv stack (all HBM-bound streams, float4 per lane):
//   max-pool 3x3/s2/p1 (resnet.py:611), GroupNorm(+ReLU) of the RetinaRPN tower
//   (atss_rpn_head.py:118,150-190 -> ConvModule norm), FPN nearest-upsample-add
//   (necks/pafpn.py:113-115), NCHW<->NHWC shuffles at the reference's tensor boundary,
//   library info.
#include "common.h"

namespace {

// element types of the NHWC activations: fp32 or bf16 (stored as unsigned short)
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
// two adjacent channel quads as ONE 16-byte access of a 16-bit tensor (fp32: two 16-byte accesses)
__device__ __forceinline__ void ld8(const float* p, float4& a, float4& b) { a = ld4(p); b = ld4(p + 4); }
__device__ __forceinline__ void st8(float* p, float4 a, float4 b) { st4(p, a); st4(p + 4, b); }
__device__ __forceinline__ void ld8(const bf16_t* p, float4& a, float4& b) {
    const uint4 u = *reinterpret_cast<const uint4*>(p);
    a = make_float4(__uint_as_float(u.x << 16), __uint_as_float(u.x & 0xffff0000u),
                    __uint_as_float(u.y << 16), __uint_as_float(u.y & 0xffff0000u));
    b = make_float4(__uint_as_float(u.z << 16), __uint_as_float(u.z & 0xffff0000u),
                    __uint_as_float(u.w << 16), __uint_as_float(u.w & 0xffff0000u));
}
__device__ __forceinline__ void st8(bf16_t* p, float4 a, float4 b) {
    uint4 u;
    u.x = brcnn_pk2b(a.x, a.y); u.y = brcnn_pk2b(a.z, a.w);
    u.z = brcnn_pk2b(b.x, b.y); u.w = brcnn_pk2b(b.z, b.w);
    *reinterpret_cast<uint4*>(p) = u;
}
__device__ __forceinline__ void ld8(const f16_t* p, float4& a, float4& b) {
    const uint4 u = *reinterpret_cast<const uint4*>(p);
    a = make_float4(brcnn_h2f((unsigned short)(u.x & 0xffffu)), brcnn_h2f((unsigned short)(u.x >> 16)),
                    brcnn_h2f((unsigned short)(u.y & 0xffffu)), brcnn_h2f((unsigned short)(u.y >> 16)));
    b = make_float4(brcnn_h2f((unsigned short)(u.z & 0xffffu)), brcnn_h2f((unsigned short)(u.z >> 16)),
                    brcnn_h2f((unsigned short)(u.w & 0xffffu)), brcnn_h2f((unsigned short)(u.w >> 16)));
}
__device__ __forceinline__ void st8(f16_t* p, float4 a, float4 b) {
    uint4 u;
    u.x = brcnn_pk2h(a.x, a.y);
    u.y = brcnn_pk2h(a.z, a.w);
    u.z = brcnn_pk2h(b.x, b.y);
    u.w = brcnn_pk2h(b.z, b.w);
    *reinterpret_cast<uint4*>(p) = u;
}
// one element from fp32 (weight packing)
__device__ __forceinline__ void st1(float* p, float v) { *p = v; }
__device__ __forceinline__ void st1(bf16_t* p, float v) { *p = (bf16_t)f2bf(v); }
__device__ __forceinline__ void st1(f16_t* p, float v) { p->v = brcnn_f2h(v); }

// one output pixel x 4 channels per thread (fp32: measured 6 % faster than the paired form below)
__global__ __launch_bounds__(256) void maxpool3x3s2_f32_kernel(const float* __restrict__ x, float* __restrict__ y, int N,
                                                              int H, int W, int C, int Ho, int Wo) {
    const int c4n = C >> 2;
    const long long total = (long long)N * Ho * Wo * c4n;
    for (long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x; idx < total;
         idx += (long long)gridDim.x * blockDim.x) {
        const int c4 = (int)(idx % c4n);
        long long r = idx / c4n;
        const int wo = (int)(r % Wo); r /= Wo;
        const int ho = (int)(r % Ho);
        const int n = (int)(r / Ho);
        float4 m = make_float4(-INFINITY, -INFINITY, -INFINITY, -INFINITY);
        for (int kh = 0; kh < 3; kh++) {
            const int hi = ho * 2 - 1 + kh;
            if (hi < 0 || hi >= H) continue;
            for (int kw = 0; kw < 3; kw++) {
                const int wi = wo * 2 - 1 + kw;
                if (wi < 0 || wi >= W) continue;
                const float4 v = ld4(x + (((size_t)n * H + hi) * W + wi) * C + c4 * 4);
                m.x = fmaxf(m.x, v.x); m.y = fmaxf(m.y, v.y);
                m.z = fmaxf(m.z, v.z); m.w = fmaxf(m.w, v.w);
            }
        }
        st4(y + (size_t)idx * 4, m);
    }
}

// 16 bytes of channels (4 fp32 / 8 bf16) of TWO horizontally adjacent output pixels per thread: the
// pair shares the middle input column, so 15 loads serve 2 outputs (9 each before); 32-bit index
// arithmetic.  max of bf16 values is exact in fp32, so the bf16 result is bit-identical.
struct Vec8 { float v[8]; };
__device__ __forceinline__ void vmax(Vec8& m, const Vec8& x, int n) {
#pragma unroll
    for (int e = 0; e < 8; e++) if (e < n) m.v[e] = fmaxf(m.v[e], x.v[e]);
}
__device__ __forceinline__ Vec8 ldv(const float* p) {
    const float4 a = *reinterpret_cast<const float4*>(p);
    Vec8 r; r.v[0] = a.x; r.v[1] = a.y; r.v[2] = a.z; r.v[3] = a.w; r.v[4] = r.v[5] = r.v[6] = r.v[7] = 0.f;
    return r;
}
__device__ __forceinline__ Vec8 ldv(const bf16_t* p) {
    const uint4 u = *reinterpret_cast<const uint4*>(p);
    const unsigned w[4] = {u.x, u.y, u.z, u.w};
    Vec8 r;
#pragma unroll
    for (int e = 0; e < 4; e++) { r.v[2 * e] = __uint_as_float(w[e] << 16); r.v[2 * e + 1] = __uint_as_float(w[e] & 0xffff0000u); }
    return r;
}
__device__ __forceinline__ Vec8 ldv(const f16_t* p) {
    const uint4 u = *reinterpret_cast<const uint4*>(p);
    const unsigned w[4] = {u.x, u.y, u.z, u.w};
    Vec8 r;
#pragma unroll
    for (int e = 0; e < 4; e++) {
        r.v[2 * e] = brcnn_h2f((unsigned short)(w[e] & 0xffffu));
        r.v[2 * e + 1] = brcnn_h2f((unsigned short)(w[e] >> 16));
    }
    return r;
}
__device__ __forceinline__ void stv(f16_t* p, const Vec8& m) {       // (max of fp16 values: exact, like bf16)
    uint4 u;
    u.x = brcnn_pk2h(m.v[0], m.v[1]);
    u.y = brcnn_pk2h(m.v[2], m.v[3]);
    u.z = brcnn_pk2h(m.v[4], m.v[5]);
    u.w = brcnn_pk2h(m.v[6], m.v[7]);
    *reinterpret_cast<uint4*>(p) = u;
}
__device__ __forceinline__ void stv(float* p, const Vec8& m) { *reinterpret_cast<float4*>(p) = make_float4(m.v[0], m.v[1], m.v[2], m.v[3]); }
__device__ __forceinline__ void stv(bf16_t* p, const Vec8& m) {
    uint4 u;
    u.x = (__float_as_uint(m.v[0]) >> 16) | (__float_as_uint(m.v[1]) & 0xffff0000u);
    u.y = (__float_as_uint(m.v[2]) >> 16) | (__float_as_uint(m.v[3]) & 0xffff0000u);
    u.z = (__float_as_uint(m.v[4]) >> 16) | (__float_as_uint(m.v[5]) & 0xffff0000u);
    u.w = (__float_as_uint(m.v[6]) >> 16) | (__float_as_uint(m.v[7]) & 0xffff0000u);
    *reinterpret_cast<uint4*>(p) = u;
}

template <typename T>
__global__ __launch_bounds__(256) void maxpool3x3s2_kernel(const T* __restrict__ x,
                                                          T* __restrict__ y, int N, int H,
                                                          int W, int C, int Ho, int Wo) {
    constexpr int VEC = 16 / sizeof(T);
    const int cvn = C / VEC, wp = (Wo + 1) >> 1;
    const unsigned total = (unsigned)N * Ho * wp * cvn;
    for (unsigned idx = blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += gridDim.x * blockDim.x) {
        const int cv = idx % cvn;
        unsigned r = idx / cvn;
        const int j = r % wp; r /= wp;
        const int ho = r % Ho;
        const int n = r / Ho;
        Vec8 m0, m1;
#pragma unroll
        for (int e = 0; e < 8; e++) m0.v[e] = m1.v[e] = -INFINITY;
#pragma unroll
        for (int kh = 0; kh < 3; kh++) {
            const int hi = ho * 2 - 1 + kh;
            if (hi < 0 || hi >= H) continue;
            const T* row = x + ((size_t)n * H + hi) * W * C + cv * VEC;
#pragma unroll
            for (int kw = 0; kw < 5; kw++) {
                const int wi = j * 4 - 1 + kw;
                if (wi < 0 || wi >= W) continue;
                const Vec8 v = ldv(row + (size_t)wi * C);
                if (kw <= 2) vmax(m0, v, VEC);
                if (kw >= 2) vmax(m1, v, VEC);
            }
        }
        T* out = y + (((size_t)n * Ho + ho) * Wo + 2 * j) * C + cv * VEC;
        stv(out, m0);
        if (2 * j + 1 < Wo) stv(out + C, m1);
    }
}

// max-pool 3x3/s2/p1 backward: one input pixel x 4 channels per thread gathers the gradient of the (at most 4)
// windows that contain it and whose FIRST maximum (row-major scan, torch's argmax rule) it is -- no atomics
template <typename T>
__global__ __launch_bounds__(256) void maxpool3x3s2_bwd_kernel(const T* __restrict__ x, const T* __restrict__ y,
                                                              const T* __restrict__ dy, T* __restrict__ dx, int N, int H,
                                                              int W, int C, int Ho, int Wo) {
    const int cvn = C >> 2;
    const long long total = (long long)N * H * W * cvn;
    for (long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x; idx < total;
         idx += (long long)gridDim.x * blockDim.x) {
        const int cv = (int)(idx % cvn);
        long long r = idx / cvn;
        const int w = (int)(r % W); r /= W;
        const int h = (int)(r % H);
        const int n = (int)(r / H);
        const float4 me = ld4(x + (((size_t)n * H + h) * W + w) * C + cv * 4);
        float g[4] = {0.f, 0.f, 0.f, 0.f};
        const float mv[4] = {me.x, me.y, me.z, me.w};
        for (int ho = (h >> 1); ho <= ((h + 1) >> 1); ho++) {
            if (ho < 0 || ho >= Ho) continue;
            for (int wo = (w >> 1); wo <= ((w + 1) >> 1); wo++) {
                if (wo < 0 || wo >= Wo) continue;
                const size_t o = (((size_t)n * Ho + ho) * Wo + wo) * C + cv * 4;
                const float4 ym = ld4(y + o), gy = ld4(dy + o);
                const float yv[4] = {ym.x, ym.y, ym.z, ym.w}, gv[4] = {gy.x, gy.y, gy.z, gy.w};
                bool mine[4];
#pragma unroll
                for (int e = 0; e < 4; e++) mine[e] = mv[e] == yv[e];
                if (!(mine[0] || mine[1] || mine[2] || mine[3])) continue;
                // an earlier position of the window holding the same maximum takes the gradient instead
                for (int kh = 0; kh < 3; kh++) {
                    const int hi = ho * 2 - 1 + kh;
                    if (hi < 0 || hi >= H) continue;
                    for (int kw = 0; kw < 3; kw++) {
                        const int wi = wo * 2 - 1 + kw;
                        if (wi < 0 || wi >= W) continue;
                        if (hi > h || (hi == h && wi >= w)) continue;
                        const float4 v = ld4(x + (((size_t)n * H + hi) * W + wi) * C + cv * 4);
                        if (v.x == yv[0]) mine[0] = false;
                        if (v.y == yv[1]) mine[1] = false;
                        if (v.z == yv[2]) mine[2] = false;
                        if (v.w == yv[3]) mine[3] = false;
                    }
                }
#pragma unroll
                for (int e = 0; e < 4; e++)
                    if (mine[e]) g[e] += gv[e];
            }
        }
        st4(dx + (((size_t)n * H + h) * W + w) * C + cv * 4, make_float4(g[0], g[1], g[2], g[3]));
    }
}

// ---- GroupNorm over one or several back-to-back NHWC segments (pyramid levels) -----------
struct GnSegs {
    int nseg;
    int hw[BRCNN_MAX_LEVELS];
    long long row0[BRCNN_MAX_LEVELS + 1];   // first row of segment s in the concatenated (rows, C)
};

// pass 1: per (segment, n, group) sum / sum of squares, accumulated in double.
// grid (chunks, N * nseg); 256 threads = 64 channel quads x 4 row lanes: every thread streams
// float4 rows (16 loads in flight per thread), partial sums are folded to double every 32 rows,
// the 4 row lanes are combined through LDS and one double atomic pair per (group) leaves the
// workgroup.  C <= 256, C % 4 == 0, (C/G) % 4 == 0 or 4 % (C/G) == 0 handled generally below.
template <typename T, int V>
__global__ __launch_bounds__(256) void gn_stats_kernel(const T* __restrict__ x,
                                                      double* __restrict__ stats, GnSegs sg, int N,
                                                      int C, int G, int rows_per_block) {
    // V channels per lane: 16-byte loads for both element widths where C % 8 == 0 (round 6: the 16-bit form read 8 bytes
    // per lane and ran at 2.4 TB/s, the fp32 form at 4.2); LPR lanes cover 256 channels of a row, RL row lanes share the
    // block's strip
    constexpr int LPR = 256 / V, RL = 256 / LPR;
    __shared__ double red[RL][256][2];    // [row lane][channel][sum, sumsq]
    const int seg = blockIdx.y / N, n = blockIdx.y - seg * N;
    const int HW = sg.hw[seg];
    const int row0 = blockIdx.x * rows_per_block;
    if (row0 >= HW) return;
    const int row1 = min(HW, row0 + rows_per_block);
    const T* xs = x + (size_t)(sg.row0[seg] + (long long)n * HW) * C;
    const int cvn = C / V;
    const int q = threadIdx.x % LPR, rl = threadIdx.x / LPR;
    for (int q0 = q; q0 < cvn; q0 += LPR) {
        double ds[V], dss[V];
#pragma unroll
        for (int e = 0; e < V; e++) ds[e] = dss[e] = 0.0;
        for (int rb = row0 + rl; rb < row1; rb += RL * 32) {
            float s[V], ss[V];
#pragma unroll
            for (int e = 0; e < V; e++) s[e] = ss[e] = 0.f;
            // eight rows per batch: the loads are unconditional (a row beyond the strip re-reads its last row and is
            // weighted 0) -- with the load under `if (r < row1)` the compiler waited for every load before issuing the
            // next (one 16-byte load in flight per wave: 4.2 TB/s fp32, 3.2 TB/s bf16 at the tower's size)
#pragma unroll
            for (int i0 = 0; i0 < 32; i0 += 8) {
                float xi[8][V];
#pragma unroll
                for (int i = 0; i < 8; i++) {
                    const int r = min(rb + RL * (i0 + i), row1 - 1);
                    if constexpr (V == 8) {
                        float4 x0, x1;
                        ld8(xs + (size_t)r * C + q0 * 8, x0, x1);
                        xi[i][0] = x0.x; xi[i][1] = x0.y; xi[i][2] = x0.z; xi[i][3] = x0.w;
                        xi[i][4] = x1.x; xi[i][5] = x1.y; xi[i][6] = x1.z; xi[i][7] = x1.w;
                    } else {
                        const float4 v = ld4(xs + (size_t)r * C + q0 * 4);
                        xi[i][0] = v.x; xi[i][1] = v.y; xi[i][2] = v.z; xi[i][3] = v.w;
                    }
                }
#pragma unroll
                for (int i = 0; i < 8; i++) {
                    const bool ok = rb + RL * (i0 + i) < row1;
#pragma unroll
                    for (int e = 0; e < V; e++) {
                        const float v = ok ? xi[i][e] : 0.f;
                        s[e] += v; ss[e] += v * v;
                    }
                }
            }
#pragma unroll
            for (int e = 0; e < V; e++) { ds[e] += s[e]; dss[e] += ss[e]; }
        }
#pragma unroll
        for (int e = 0; e < V; e++) {
            red[rl][q0 * V + e][0] = ds[e];
            red[rl][q0 * V + e][1] = dss[e];
        }
    }
    __syncthreads();
    const int cpg = C / G;
    for (int g = threadIdx.x; g < G; g += 256) {
        double a = 0.0, b2 = 0.0;
        for (int c = g * cpg; c < (g + 1) * cpg; c++)
#pragma unroll
            for (int l = 0; l < RL; l++) { a += red[l][c][0]; b2 += red[l][c][1]; }
        atomicAdd(&stats[((size_t)blockIdx.y * G + g) * 2 + 0], a);
        atomicAdd(&stats[((size_t)blockIdx.y * G + g) * 2 + 1], b2);
    }
}

// pass 2: the (sum, sumsq) doubles of every (segment, n, group) become (mean, rstd) floats in
// place (first 8 bytes of the 16-byte entry), so that the streaming pass does no fp64 math
__global__ void gn_finalize_kernel(double* __restrict__ stats, GnSegs sg, int N, int C, int G,
                                   float eps) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= sg.nseg * N * G) return;
    const int seg = i / (N * G);
    const double inv_cnt = 1.0 / ((double)sg.hw[seg] * (C / G));
    const double mean = stats[(size_t)i * 2] * inv_cnt;
    double var = stats[(size_t)i * 2 + 1] * inv_cnt - mean * mean;
    if (var < 0.0) var = 0.0;
    const float rstd = (float)(1.0 / sqrt(var + (double)eps));
    reinterpret_cast<float2*>(stats + (size_t)i * 2)[0] = make_float2((float)mean, rstd);
}

// pass 3: y = (x - mean) * rstd * gamma + beta (+ReLU); Q channel quads (16 bytes) per thread
template <typename T, int Q>
__global__ __launch_bounds__(256) void gn_apply_kernel(const T* __restrict__ x,
                                                      const double* __restrict__ stats,
                                                      const float* __restrict__ gamma,
                                                      const float* __restrict__ beta,
                                                      T* __restrict__ y, GnSegs sg, int N, int C,
                                                      int G, int relu) {
    const int cvn = C / (4 * Q), cpg = C / G;
    // 32-bit index arithmetic (the host checks rows * C < 2^31): the row -> (segment, image) decode
    // was two 64-bit divisions per thread and iteration
    const unsigned total = (unsigned)(sg.row0[sg.nseg] * cvn);
    for (unsigned idx = blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += gridDim.x * blockDim.x) {
        const unsigned row = idx / (unsigned)cvn;
        const int cv = (int)(idx - row * (unsigned)cvn);
        int seg = 0;
#pragma unroll
        for (int t = 1; t < BRCNN_MAX_LEVELS; t++)
            if (t < sg.nseg && row >= (unsigned)sg.row0[t]) seg = t;
        const int n = (int)((row - (unsigned)sg.row0[seg]) / (unsigned)sg.hw[seg]);
        const size_t sbase = ((size_t)(seg * N + n)) * G;
        static_assert(Q == 1 || Q == 2, "one or two channel quads per thread");
        float4 v[2];
        if constexpr (Q == 2) ld8(x + (size_t)idx * 8, v[0], v[1]);
        else v[0] = ld4(x + (size_t)idx * 4);
#pragma unroll
        for (int q = 0; q < Q; q++) {
            const int c0 = (cv * Q + q) * 4;
            const float in[4] = {v[q].x, v[q].y, v[q].z, v[q].w};
            float out[4];
#pragma unroll
            for (int e = 0; e < 4; e++) {
                const int c = c0 + e;
                const float2 mr = reinterpret_cast<const float2*>(stats + (sbase + c / cpg) * 2)[0];
                float o = (in[e] - mr.x) * mr.y * gamma[c] + beta[c];
                if (relu) o = fmaxf(o, 0.f);
                out[e] = o;
            }
            v[q] = make_float4(out[0], out[1], out[2], out[3]);
        }
        if constexpr (Q == 2) st8(y + (size_t)idx * 8, v[0], v[1]);
        else st4(y + (size_t)idx * 4, v[0]);
    }
}


// Row-strip forms of the two apply passes for 16-bit tensors: grid (row chunks, N * nseg) like the statistics
// kernels, 256 threads = 32 lanes of 8 channels (one 16-byte access) x 8 row lanes.  A thread keeps its 8
// channels' statistics / affine in registers for all of its rows, so the inner loop is load - ~10 flops per
// element - store: the flat kernels above re-derive (segment, image, group) and re-fetch the statistics per
// element and are instruction / latency bound on 16-bit data (1.9 TB/s).
template <typename T>
__global__ __launch_bounds__(256) void gn_apply_rows_kernel(const T* __restrict__ x, const double* __restrict__ stats,
                                                           const float* __restrict__ gamma,
                                                           const float* __restrict__ beta, T* __restrict__ y, GnSegs sg,
                                                           int N, int C, int G, int rows_per_block, int relu) {
    const int seg = blockIdx.y / N, n = blockIdx.y - seg * N;
    const int HW = sg.hw[seg];
    const int row0 = blockIdx.x * rows_per_block;
    if (row0 >= HW) return;
    const int row1 = min(HW, row0 + rows_per_block);
    const size_t base = (size_t)(sg.row0[seg] + (long long)n * HW) * C;
    const int c8n = C >> 3, cpg = C / G;
    const int v = threadIdx.x & 31, rl = threadIdx.x >> 5;
    for (int v0 = v; v0 < c8n; v0 += 32) {
        float mean[8], rstd[8], gm[8], bt[8];
#pragma unroll
        for (int e = 0; e < 8; e++) {
            const int c = v0 * 8 + e;
            const float2 mr = reinterpret_cast<const float2*>(stats + ((size_t)blockIdx.y * G + c / cpg) * 2)[0];
            mean[e] = mr.x; rstd[e] = mr.y; gm[e] = gamma[c]; bt[e] = beta[c];
        }
        const T* xs = x + base + v0 * 8;
        T* ys = y + base + v0 * 8;
#pragma unroll 4
        for (int r = row0 + rl; r < row1; r += 8) {
            float4 a, b;
            ld8(xs + (size_t)r * C, a, b);
            float in[8] = {a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w};
#pragma unroll
            for (int e = 0; e < 8; e++) {
                float o = (in[e] - mean[e]) * rstd[e] * gm[e] + bt[e];
                in[e] = relu ? fmaxf(o, 0.f) : o;
            }
            st8(ys + (size_t)r * C, make_float4(in[0], in[1], in[2], in[3]), make_float4(in[4], in[5], in[6], in[7]));
        }
    }
}

template <typename T>
__global__ __launch_bounds__(256) void gn_bwd_apply_rows_kernel(const T* __restrict__ x, const T* __restrict__ dy,
                                                               const double* __restrict__ stats,
                                                               const double* __restrict__ gsum,
                                                               const float* __restrict__ gamma,
                                                               const float* __restrict__ beta, T* __restrict__ dx,
                                                               GnSegs sg, int N, int C, int G, int rows_per_block, int relu) {
    const int seg = blockIdx.y / N, n = blockIdx.y - seg * N;
    const int HW = sg.hw[seg];
    const int row0 = blockIdx.x * rows_per_block;
    if (row0 >= HW) return;
    const int row1 = min(HW, row0 + rows_per_block);
    const size_t base = (size_t)(sg.row0[seg] + (long long)n * HW) * C;
    const int c8n = C >> 3, cpg = C / G;
    const float inv_d = 1.f / ((float)HW * (float)cpg);
    const int v = threadIdx.x & 31, rl = threadIdx.x >> 5;
    for (int v0 = v; v0 < c8n; v0 += 32) {
        float mean[8], rstd[8], gm[8], bt[8], s1[8], s2[8];
#pragma unroll
        for (int e = 0; e < 8; e++) {
            const int c = v0 * 8 + e;
            const size_t gi = ((size_t)blockIdx.y * G + c / cpg) * 2;
            const float2 mr = reinterpret_cast<const float2*>(stats + gi)[0];
            mean[e] = mr.x; rstd[e] = mr.y; gm[e] = gamma[c]; bt[e] = beta[c];
            s1[e] = (float)gsum[gi]; s2[e] = (float)gsum[gi + 1];
        }
        const T* xs = x + base + v0 * 8;
        const T* ds = dy + base + v0 * 8;
        T* os = dx + base + v0 * 8;
#pragma unroll 4
        for (int r = row0 + rl; r < row1; r += 8) {
            float4 xa, xb, da, db;
            ld8(xs + (size_t)r * C, xa, xb);
            ld8(ds + (size_t)r * C, da, db);
            const float xi[8] = {xa.x, xa.y, xa.z, xa.w, xb.x, xb.y, xb.z, xb.w};
            const float di[8] = {da.x, da.y, da.z, da.w, db.x, db.y, db.z, db.w};
            float out[8];
#pragma unroll
            for (int e = 0; e < 8; e++) {
                const float xh = (xi[e] - mean[e]) * rstd[e];
                const float g = (relu && xh * gm[e] + bt[e] <= 0.f) ? 0.f : di[e];
                out[e] = rstd[e] * (g * gm[e] - (s1[e] + xh * s2[e]) * inv_d);
            }
            st8(os + (size_t)r * C, make_float4(out[0], out[1], out[2], out[3]), make_float4(out[4], out[5], out[6], out[7]));
        }
    }
}


// ---- GroupNorm(+ReLU) backward over the same multi-segment NHWC layout ----------------------
// With xh = (x - mean) * rstd, g = dy (zeroed where the ReLU clipped):
//     dbeta_c = sum g,  dgamma_c = sum g*xh                     (over images and pixels)
//     s1 = sum_{c in group, pixels} g*gamma_c,  s2 = sum g*gamma_c*xh     (per segment, image, group)
//     dx = rstd * (g*gamma_c - s1/D - xh*s2/D),   D = pixels * channels per group
// pass 1 (same streaming layout as gn_stats_kernel): per workgroup the per-channel sums (A = sum g,
// B = sum g*xh) of its row chunk; they go to a partial buffer [workgroup][2][C] for the
// deterministic second stage (dgamma / dbeta), and, weighted by gamma, as two double atomics per
// group into gsum (s1, s2).
template <typename T>
__global__ __launch_bounds__(256) void gn_bwd_reduce_kernel(const T* __restrict__ x, const T* __restrict__ dy,
                                                           const double* __restrict__ stats,
                                                           const float* __restrict__ gamma,
                                                           const float* __restrict__ beta, double* __restrict__ gsum,
                                                           float* __restrict__ part, GnSegs sg, int N, int C, int G,
                                                           int rows_per_block, int relu) {
    // V channels per lane: 16-byte loads for both element widths (8 x 16-bit / 4 x fp32); LPR lanes cover 256 channels
    // of a row, RL rows are in flight per pass over the block's row strip
    constexpr int V = sizeof(T) == 2 ? 8 : 4, LPR = 256 / V, RL = 256 / LPR;
    __shared__ float red[RL][256][2];
    const int seg = blockIdx.y / N, n = blockIdx.y - seg * N;
    const int HW = sg.hw[seg];
    const int row0 = blockIdx.x * rows_per_block;
    float* my_part = part + ((size_t)blockIdx.y * gridDim.x + blockIdx.x) * 2 * C;
    const int cpg = C / G;
    if (row0 >= HW) {       // chunk beyond this (smaller) segment: its partial must still be defined
        for (int c = threadIdx.x; c < 2 * C; c += 256) my_part[c] = 0.f;
        return;
    }
    const int row1 = min(HW, row0 + rows_per_block);
    const size_t base = (size_t)(sg.row0[seg] + (long long)n * HW) * C;
    const T* xs = x + base;
    const T* ds = dy + base;
    const int cvn = C / V;
    const int q = threadIdx.x % LPR, rl = threadIdx.x / LPR;
    for (int q0 = q; q0 < cvn; q0 += LPR) {
        float mean[V], rstd[V], gm[V], bt[V];
#pragma unroll
        for (int e = 0; e < V; e++) {
            const int c = q0 * V + e;
            const float2 mr = reinterpret_cast<const float2*>(stats + ((size_t)blockIdx.y * G + c / cpg) * 2)[0];
            mean[e] = mr.x; rstd[e] = mr.y; gm[e] = gamma[c]; bt[e] = beta[c];
        }
        float a[V], b[V];
#pragma unroll
        for (int e = 0; e < V; e++) a[e] = b[e] = 0.f;
#pragma unroll 4
        for (int r = row0 + rl; r < row1; r += RL) {
            float xi[V], di[V];
            if constexpr (V == 8) {
                float4 x0, x1, d0, d1;
                ld8(xs + (size_t)r * C + q0 * 8, x0, x1);
                ld8(ds + (size_t)r * C + q0 * 8, d0, d1);
                xi[0] = x0.x; xi[1] = x0.y; xi[2] = x0.z; xi[3] = x0.w; xi[4] = x1.x; xi[5] = x1.y; xi[6] = x1.z; xi[7] = x1.w;
                di[0] = d0.x; di[1] = d0.y; di[2] = d0.z; di[3] = d0.w; di[4] = d1.x; di[5] = d1.y; di[6] = d1.z; di[7] = d1.w;
            } else {
                const float4 xv = ld4(xs + (size_t)r * C + q0 * 4);
                const float4 dv = ld4(ds + (size_t)r * C + q0 * 4);
                xi[0] = xv.x; xi[1] = xv.y; xi[2] = xv.z; xi[3] = xv.w;
                di[0] = dv.x; di[1] = dv.y; di[2] = dv.z; di[3] = dv.w;
            }
#pragma unroll
            for (int e = 0; e < V; e++) {
                const float xh = (xi[e] - mean[e]) * rstd[e];
                const float g = (relu && xh * gm[e] + bt[e] <= 0.f) ? 0.f : di[e];
                a[e] += g;
                b[e] += g * xh;
            }
        }
#pragma unroll
        for (int e = 0; e < V; e++) {
            red[rl][q0 * V + e][0] = a[e];
            red[rl][q0 * V + e][1] = b[e];
        }
    }
    __syncthreads();
    for (int c = threadIdx.x; c < C; c += 256) {
        float A = 0.f, B = 0.f;
#pragma unroll
        for (int w = 0; w < RL; w++) { A += red[w][c][0]; B += red[w][c][1]; }
        my_part[c] = A;
        my_part[C + c] = B;
        red[0][c][0] = A * gamma[c];
        red[0][c][1] = B * gamma[c];
    }
    __syncthreads();
    for (int g = threadIdx.x; g < G; g += 256) {
        double s1 = 0.0, s2 = 0.0;
        for (int c = g * cpg; c < (g + 1) * cpg; c++) { s1 += red[0][c][0]; s2 += red[0][c][1]; }
        atomicAdd(&gsum[((size_t)blockIdx.y * G + g) * 2 + 0], s1);
        atomicAdd(&gsum[((size_t)blockIdx.y * G + g) * 2 + 1], s2);
    }
}

// second stage of dgamma / dbeta: column sums of the [num_parts][2][C] partials.  Grid (32-column blocks,
// GN_PARAM_SLICES row slices): a workgroup sums its slice of the partial rows (32 columns x 8 row lanes, four rows
// in flight per lane, combined through LDS) into part2[slice][2C]; gn_bwd_param_final_kernel adds the slices in a
// fixed order.  (One workgroup per 32 columns over all 2560 partial rows of the five-level tower -- 16 workgroups
// on a 256-CU part -- took 155 us per layer.)
constexpr int GN_PARAM_SLICES = 32;
__global__ __launch_bounds__(256) void gn_bwd_param_kernel(const float* __restrict__ part, float* __restrict__ part2,
                                                          int num_parts, int C) {
    __shared__ float red[8][33];
    const int cols = 2 * C;
    const int cl = threadIdx.x & 31, lane = threadIdx.x >> 5;
    const int col = blockIdx.x * 32 + cl;
    const int per = (num_parts + GN_PARAM_SLICES - 1) / GN_PARAM_SLICES;
    const int r0 = blockIdx.y * per, r1 = min(num_parts, r0 + per);
    float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
    if (col < cols) {
        int r = r0 + lane;
        for (; r + 24 < r1; r += 32) {
            a0 += part[(size_t)r * cols + col];
            a1 += part[(size_t)(r + 8) * cols + col];
            a2 += part[(size_t)(r + 16) * cols + col];
            a3 += part[(size_t)(r + 24) * cols + col];
        }
        for (; r < r1; r += 8) a0 += part[(size_t)r * cols + col];
    }
    red[lane][cl] = (a0 + a1) + (a2 + a3);
    __syncthreads();
    if (lane == 0 && col < cols) {
        float t = 0.f;
#pragma unroll
        for (int l = 0; l < 8; l++) t += red[l][cl];
        part2[(size_t)blockIdx.y * cols + col] = t;
    }
}

__global__ __launch_bounds__(256) void gn_bwd_param_final_kernel(const float* __restrict__ part2, float* __restrict__ dgamma,
                                                                float* __restrict__ dbeta, int C) {
    const int cols = 2 * C;
    const int col = blockIdx.x * 256 + threadIdx.x;
    if (col >= cols) return;
    float t = 0.f;
#pragma unroll 8
    for (int s = 0; s < GN_PARAM_SLICES; s++) t += part2[(size_t)s * cols + col];
    if (col < C) dbeta[col] = t;
    else dgamma[col - C] = t;
}

// Q channel quads (16 bytes of a 16-bit tensor with Q = 2) per thread; the group statistics are fetched once per
// quad when the quad lies inside one group (channels per group a multiple of 4: every GroupNorm of the recipes)
template <typename T, int Q>
__global__ __launch_bounds__(256) void gn_bwd_apply_kernel(const T* __restrict__ x, const T* __restrict__ dy,
                                                          const double* __restrict__ stats,
                                                          const double* __restrict__ gsum,
                                                          const float* __restrict__ gamma,
                                                          const float* __restrict__ beta, T* __restrict__ dx,
                                                          GnSegs sg, int N, int C, int G, int relu) {
    const int cvn = C / (4 * Q), cpg = C / G;
    const unsigned total = (unsigned)(sg.row0[sg.nseg] * cvn);      // rows * C < 2^31 checked by the host
    for (unsigned idx = blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += gridDim.x * blockDim.x) {
        const unsigned row = idx / (unsigned)cvn;
        const int cv = (int)(idx - row * (unsigned)cvn);
        int seg = 0;
#pragma unroll
        for (int t = 1; t < BRCNN_MAX_LEVELS; t++)
            if (t < sg.nseg && row >= (unsigned)sg.row0[t]) seg = t;
        const int n = (int)((row - (unsigned)sg.row0[seg]) / (unsigned)sg.hw[seg]);
        const size_t sbase = ((size_t)(seg * N + n)) * G;
        const float inv_d = 1.f / ((float)sg.hw[seg] * (float)cpg);
        float4 xq[2], dq[2];
        if constexpr (Q == 2) {
            ld8(x + (size_t)idx * 8, xq[0], xq[1]);
            ld8(dy + (size_t)idx * 8, dq[0], dq[1]);
        } else {
            xq[0] = ld4(x + (size_t)idx * 4);
            dq[0] = ld4(dy + (size_t)idx * 4);
        }
#pragma unroll
        for (int q = 0; q < Q; q++) {
            const int c0 = (cv * Q + q) * 4;
            const float xi[4] = {xq[q].x, xq[q].y, xq[q].z, xq[q].w}, di[4] = {dq[q].x, dq[q].y, dq[q].z, dq[q].w};
            const bool one = (c0 + 3) / cpg == c0 / cpg;
            size_t gi = (sbase + c0 / cpg) * 2;
            float2 mr = reinterpret_cast<const float2*>(stats + gi)[0];
            float s1 = (float)gsum[gi], s2 = (float)gsum[gi + 1];
            float out[4];
#pragma unroll
            for (int e = 0; e < 4; e++) {
                const int c = c0 + e;
                if (!one) {
                    gi = (sbase + c / cpg) * 2;
                    mr = reinterpret_cast<const float2*>(stats + gi)[0];
                    s1 = (float)gsum[gi]; s2 = (float)gsum[gi + 1];
                }
                const float xh = (xi[e] - mr.x) * mr.y;
                const float gmm = gamma[c];
                const float g = (relu && xh * gmm + beta[c] <= 0.f) ? 0.f : di[e];
                out[e] = mr.y * (g * gmm - (s1 + xh * s2) * inv_d);
            }
            xq[q] = make_float4(out[0], out[1], out[2], out[3]);
        }
        if constexpr (Q == 2) st8(dx + (size_t)idx * 8, xq[0], xq[1]);
        else st4(dx + (size_t)idx * 4, xq[0]);
    }
}


// Shader-clock probe: one lane spins for `wall_ticks` ticks of the constant 100 MHz counter and
// reports how many shader cycles (s_memtime) passed meanwhile: launched on a second stream next
// to a workload, out[1] / out[0] * 0.1 is the effective shader clock in GHz under that load.
__global__ void clock_probe_kernel(long long* __restrict__ out, long long wall_ticks) {
    const long long w0 = wall_clock64(), c0 = clock64();
    long long w = w0;
    while (w - w0 < wall_ticks) {
        __builtin_amdgcn_s_sleep(32);
        w = wall_clock64();
    }
    out[0] = w - w0;
    out[1] = clock64() - c0;
}

template <typename T>
__global__ __launch_bounds__(256) void upsample_add_kernel(T* __restrict__ dst,
                                                          const T* __restrict__ src, int N,
                                                          int Hd, int Wd, int Hs, int Ws, int C) {
    const int c4n = C >> 2;
    const long long total = (long long)N * Hd * Wd * c4n;
    const float sh = (float)Hs / (float)Hd, sw = (float)Ws / (float)Wd;
    for (long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x; idx < total;
         idx += (long long)gridDim.x * blockDim.x) {
        const int c4 = (int)(idx % c4n);
        long long r = idx / c4n;
        const int wd = (int)(r % Wd); r /= Wd;
        const int hd = (int)(r % Hd);
        const int n = (int)(r / Hd);
        const int hs = min((int)floorf(hd * sh), Hs - 1);
        const int ws = min((int)floorf(wd * sw), Ws - 1);
        const float4 s = ld4(src + (((size_t)n * Hs + hs) * Ws + ws) * C + c4 * 4);
        float4 d = ld4(dst + (size_t)idx * 4);
        d.x += s.x; d.y += s.y; d.z += s.z; d.w += s.w;
        st4(dst + (size_t)idx * 4, d);
    }
}

// out = dst + nearest_upsample(src) written to a separate tensor (the differentiable form), and its
// backward w.r.t. src: every source pixel sums the gradient of the destination pixels that read it
// (the same floor(dst * scale) mapping, so the candidates around src * Hd / Hs are tested with it)
template <typename T>
__global__ __launch_bounds__(256) void upsample_add_out_kernel(const T* __restrict__ dst, const T* __restrict__ src,
                                                              T* __restrict__ out, int N, int Hd, int Wd, int Hs,
                                                              int Ws, int C) {
    const int c4n = C >> 2;
    const long long total = (long long)N * Hd * Wd * c4n;
    const float sh = (float)Hs / (float)Hd, sw = (float)Ws / (float)Wd;
    for (long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x; idx < total;
         idx += (long long)gridDim.x * blockDim.x) {
        const int c4 = (int)(idx % c4n);
        long long r = idx / c4n;
        const int wd = (int)(r % Wd); r /= Wd;
        const int hd = (int)(r % Hd);
        const int n = (int)(r / Hd);
        const int hs = min((int)floorf(hd * sh), Hs - 1);
        const int ws = min((int)floorf(wd * sw), Ws - 1);
        const float4 s = ld4(src + (((size_t)n * Hs + hs) * Ws + ws) * C + c4 * 4);
        float4 d = ld4(dst + (size_t)idx * 4);
        d.x += s.x; d.y += s.y; d.z += s.z; d.w += s.w;
        st4(out + (size_t)idx * 4, d);
    }
}

template <typename T>
__global__ __launch_bounds__(256) void upsample_add_bwd_kernel(const T* __restrict__ dout, T* __restrict__ dsrc, int N,
                                                              int Hd, int Wd, int Hs, int Ws, int C) {
    const int c4n = C >> 2;
    const long long total = (long long)N * Hs * Ws * c4n;
    const float sh = (float)Hs / (float)Hd, sw = (float)Ws / (float)Wd;
    for (long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x; idx < total;
         idx += (long long)gridDim.x * blockDim.x) {
        const int c4 = (int)(idx % c4n);
        long long r = idx / c4n;
        const int ws = (int)(r % Ws); r /= Ws;
        const int hs = (int)(r % Hs);
        const int n = (int)(r / Hs);
        const int h0 = max(0, (int)((long long)hs * Hd / Hs) - 1), h1 = min(Hd - 1, (int)((long long)(hs + 1) * Hd / Hs) + 1);
        const int w0 = max(0, (int)((long long)ws * Wd / Ws) - 1), w1 = min(Wd - 1, (int)((long long)(ws + 1) * Wd / Ws) + 1);
        float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
        for (int hd = h0; hd <= h1; hd++) {
            if (min((int)floorf(hd * sh), Hs - 1) != hs) continue;
            for (int wd = w0; wd <= w1; wd++) {
                if (min((int)floorf(wd * sw), Ws - 1) != ws) continue;
                const float4 g = ld4(dout + (((size_t)n * Hd + hd) * Wd + wd) * C + c4 * 4);
                acc.x += g.x; acc.y += g.y; acc.z += g.z; acc.w += g.w;
            }
        }
        st4(dsrc + (size_t)idx * 4, acc);
    }
}

// column sums of a (rows, C) tensor (bias gradients): strips of rows per workgroup, then a fixed-order
// second stage -- deterministic, reads bf16 directly
template <typename T>
__global__ __launch_bounds__(256) void colsum_partial_kernel(const T* __restrict__ x, float* __restrict__ partial,
                                                            long long rows, int C, int rows_per_block) {
    __shared__ float red[256][4];
    const int c4n = C >> 2;
    const int CW = c4n < 256 ? c4n : 256;                 // channel vectors per block (power of two or 256)
    const int cv = blockIdx.y * CW + (threadIdx.x % CW);
    const int rl = threadIdx.x / CW, RL = 256 / CW;
    const long long r0 = (long long)blockIdx.x * rows_per_block, r1 = min(rows, r0 + (long long)rows_per_block);
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    if (cv < c4n && rl < RL)
        // eight rows per batch, loaded unconditionally (a row beyond the strip re-reads the strip's last row and is not
        // added): the plain `for r: acc += x[r]` loop kept ONE load in flight per wave; same sums in the same order
        for (long long r = r0 + rl; r < r1; r += 8 * RL) {
            float4 v[8];
#pragma unroll
            for (int u = 0; u < 8; u++) v[u] = ld4(x + (size_t)min(r + (long long)u * RL, r1 - 1) * C + cv * 4);
#pragma unroll
            for (int u = 0; u < 8; u++)
                if (r + (long long)u * RL < r1) { acc.x += v[u].x; acc.y += v[u].y; acc.z += v[u].z; acc.w += v[u].w; }
        }
    red[threadIdx.x][0] = acc.x; red[threadIdx.x][1] = acc.y; red[threadIdx.x][2] = acc.z; red[threadIdx.x][3] = acc.w;
    __syncthreads();
    if (rl == 0 && cv < c4n) {
#pragma unroll
        for (int e = 0; e < 4; e++) {
            float a = 0.f;
            for (int k = 0; k < RL; k++) a += red[k * CW + threadIdx.x][e];
            partial[(size_t)blockIdx.x * C + cv * 4 + e] = a;
        }
    }
}

__global__ __launch_bounds__(1024) void colsum_final_kernel(const float* __restrict__ partial, float* __restrict__ out,
                                                           int strips, int C) {
    // 16 columns x 64 strip lanes per workgroup (C / 16 workgroups: the launch is latency sized)
    __shared__ float red[64][16];
    const int col = blockIdx.x * 16 + (threadIdx.x & 15), sl = threadIdx.x >> 4;
    float a = 0.f;
    if (col < C)
        for (int s = sl; s < strips; s += 64) a += partial[(size_t)s * C + col];
    red[sl][threadIdx.x & 15] = a;
    __syncthreads();
    if (sl == 0 && col < C) {
        float t = 0.f;
#pragma unroll
        for (int k = 0; k < 64; k++) t += red[k][threadIdx.x];
        out[col] = t;
    }
}

// (N, R, Cc) -> (N, Cc, R) tiled transpose through LDS: 32x32 tiles, 256 threads
__global__ __launch_bounds__(256) void transpose_kernel(const float* __restrict__ src,
                                                       float* __restrict__ dst, int R, int Cc) {
    __shared__ float tile[32][33];
    const int n = blockIdx.z;
    const int r0 = blockIdx.y * 32, c0 = blockIdx.x * 32;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;   // ty 0..7
    const float* s = src + (size_t)n * R * Cc;
    float* d = dst + (size_t)n * R * Cc;
    for (int j = ty; j < 32; j += 8) {
        const int r = r0 + j, c = c0 + tx;
        if (r < R && c < Cc) tile[j][tx] = s[(size_t)r * Cc + c];
    }
    __syncthreads();
    for (int j = ty; j < 32; j += 8) {
        const int c = c0 + j, r = r0 + tx;
        if (r < R && c < Cc) d[(size_t)c * R + r] = tile[tx][j];
    }
}

// Per-step weight preparation of a trainable conv: the fp32 master parameter (Cout,Cin,KH,KW) is
// written once as the forward operand (Cout,KH,KW,Cin) and once as the data-gradient operand
// (Cin,KH,KW,Cout) with flipped taps, in fp32 or bf16 -- one launch instead of the five
// permute / flip / cast / contiguous kernels of the torch formulation.
template <typename T>
__global__ __launch_bounds__(256) void pack_conv_weights_kernel(const float* __restrict__ w, T* __restrict__ fwd,
                                                               T* __restrict__ dgrad, int Cout, int Cin, int KH,
                                                               int KW) {
    const long long total = (long long)Cout * Cin * KH * KW;
    for (long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x; idx < total;
         idx += (long long)gridDim.x * blockDim.x) {
        // idx enumerates the FORWARD layout (co, a, b, ci): coalesced writes of `fwd`
        const int ci = (int)(idx % Cin);
        long long r = idx / Cin;
        const int b = (int)(r % KW); r /= KW;
        const int a = (int)(r % KH);
        const int co = (int)(r / KH);
        const float v = w[(((size_t)co * Cin + ci) * KH + a) * KW + b];
        if (fwd) st1(fwd + idx, v);
        if (dgrad) st1(dgrad + (((size_t)ci * KH + (KH - 1 - a)) * KW + (KW - 1 - b)) * Cout + co, v);
    }
}

inline int stream_grid(long long total) {
    long long g = (total + 255) / 256;
    if (g > 256 * 16) g = 256 * 16;
    if (g < 1) g = 1;
    return (int)g;
}

}  // namespace

BRCNN_API int brcnn_version(void) { return 100; }

BRCNN_API int brcnn_device_count(void) {
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess) return -(1000 + (int)e);
    return n;
}

BRCNN_API int brcnn_maxpool3x3s2_nhwc(const void* x, void* y, int batch, int height, int width,
                                      int channels, int dtype, void* stream) {
    if (!x || !y || batch <= 0 || height <= 0 || width <= 0 || channels <= 0 || (channels & 3) ||
        !brcnn_elem_ok(dtype))
        return BRCNN_EINVAL;
    const int Ho = (height + 2 - 3) / 2 + 1, Wo = (width + 2 - 3) / 2 + 1;
    const int vec = dtype == BRCNN_DT_F32 ? 4 : 8;
    if (channels % vec) return BRCNN_EINVAL;
    const long long total = (long long)batch * Ho * ((Wo + 1) / 2) * (channels / vec);
    if (total >= 0x7fffffffLL) return BRCNN_EINVAL;
    if (dtype == BRCNN_DT_F32)
        hipLaunchKernelGGL(maxpool3x3s2_f32_kernel, dim3(stream_grid((long long)batch * Ho * Wo * (channels >> 2))),
                           dim3(256), 0, (hipStream_t)stream, (const float*)x, (float*)y, batch, height, width,
                           channels, Ho, Wo);
    else if (dtype == BRCNN_DT_BF16)
        hipLaunchKernelGGL(maxpool3x3s2_kernel<bf16_t>, dim3(stream_grid(total)), dim3(256), 0,
                           (hipStream_t)stream, (const bf16_t*)x, (bf16_t*)y, batch, height, width,
                           channels, Ho, Wo);
    else
        hipLaunchKernelGGL(maxpool3x3s2_kernel<f16_t>, dim3(stream_grid(total)), dim3(256), 0,
                           (hipStream_t)stream, (const f16_t*)x, (f16_t*)y, batch, height, width,
                           channels, Ho, Wo);
    BRCNN_LAUNCH_CHECK();
    return 0;
}

BRCNN_API int brcnn_maxpool3x3s2_nhwc_backward(const void* x, const void* y, const void* dy, void* dx, int batch,
                                               int height, int width, int channels, int dtype, void* stream) {
    if (!x || !y || !dy || !dx || batch <= 0 || height <= 0 || width <= 0 || channels <= 0 || (channels & 3) ||
        !brcnn_elem_ok(dtype))
        return BRCNN_EINVAL;
    const int Ho = (height + 2 - 3) / 2 + 1, Wo = (width + 2 - 3) / 2 + 1;
    const long long total = (long long)batch * height * width * (channels >> 2);
    if (dtype == BRCNN_DT_F32)
        hipLaunchKernelGGL(maxpool3x3s2_bwd_kernel<float>, dim3(stream_grid(total)), dim3(256), 0, (hipStream_t)stream,
                           (const float*)x, (const float*)y, (const float*)dy, (float*)dx, batch, height, width, channels,
                           Ho, Wo);
    else if (dtype == BRCNN_DT_BF16)
        hipLaunchKernelGGL(maxpool3x3s2_bwd_kernel<bf16_t>, dim3(stream_grid(total)), dim3(256), 0, (hipStream_t)stream,
                           (const bf16_t*)x, (const bf16_t*)y, (const bf16_t*)dy, (bf16_t*)dx, batch, height, width,
                           channels, Ho, Wo);
    else
        hipLaunchKernelGGL(maxpool3x3s2_bwd_kernel<f16_t>, dim3(stream_grid(total)), dim3(256), 0, (hipStream_t)stream,
                           (const f16_t*)x, (const f16_t*)y, (const f16_t*)dy, (f16_t*)dx, batch, height, width,
                           channels, Ho, Wo);
    BRCNN_LAUNCH_CHECK();
    return 0;
}

constexpr int GN_APPLY_ROWS = 128;      // rows of one image per workgroup of the row-strip apply kernels

template <typename T>
static int gn_forward_16(const void* x, const float* gamma, const float* beta, void* y, void* stats_ws, const GnSegs& sg,
                         int batch, int num_segments, int channels, int groups, float eps, int relu, int chunks, int rpb,
                         long long total, int nstat, hipStream_t s) {
    if (channels & 7)
        hipLaunchKernelGGL((gn_stats_kernel<T, 4>), dim3(chunks, batch * num_segments), dim3(256), 0, s, (const T*)x,
                           (double*)stats_ws, sg, batch, channels, groups, rpb);
    else
        hipLaunchKernelGGL((gn_stats_kernel<T, 8>), dim3(chunks, batch * num_segments), dim3(256), 0, s, (const T*)x,
                           (double*)stats_ws, sg, batch, channels, groups, rpb);
    BRCNN_LAUNCH_CHECK();
    hipLaunchKernelGGL(gn_finalize_kernel, dim3((nstat + 255) / 256), dim3(256), 0, s, (double*)stats_ws, sg, batch,
                       channels, groups, eps);
    BRCNN_LAUNCH_CHECK();
    if (channels & 7) {
        hipLaunchKernelGGL((gn_apply_kernel<T, 1>), dim3(stream_grid(total)), dim3(256), 0, s, (const T*)x,
                           (const double*)stats_ws, gamma, beta, (T*)y, sg, batch, channels, groups, relu);
    } else {
        int max_hw = 0;
        for (int i = 0; i < sg.nseg; i++) max_hw = sg.hw[i] > max_hw ? sg.hw[i] : max_hw;
        hipLaunchKernelGGL(gn_apply_rows_kernel<T>, dim3((max_hw + GN_APPLY_ROWS - 1) / GN_APPLY_ROWS, batch * num_segments),
                           dim3(256), 0, s, (const T*)x, (const double*)stats_ws, gamma, beta, (T*)y, sg, batch, channels,
                           groups, GN_APPLY_ROWS, relu);
    }
    BRCNN_LAUNCH_CHECK();
    return 0;
}

BRCNN_API int brcnn_groupnorm_nhwc_multi(const void* x, const float* gamma, const float* beta,
                                         void* y, void* stats_ws, int batch, int num_segments,
                                         const int* hw_host, int channels, int groups, float eps,
                                         int relu, int dtype, void* stream) {
    if (!x || !y || !gamma || !beta || !stats_ws || batch <= 0 || num_segments <= 0 ||
        num_segments > BRCNN_MAX_LEVELS || !hw_host || channels <= 0 || channels > 256 || groups <= 0 ||
        channels % groups || (channels & 3) || !brcnn_elem_ok(dtype))
        return BRCNN_EINVAL;
    GnSegs sg = {};
    sg.nseg = num_segments;
    long long rows = 0;
    int max_hw = 0;
    for (int i = 0; i < num_segments; i++) {
        if (hw_host[i] <= 0) return BRCNN_EINVAL;
        sg.hw[i] = hw_host[i];
        sg.row0[i] = rows;
        rows += (long long)batch * hw_host[i];
        if (hw_host[i] > max_hw) max_hw = hw_host[i];
    }
    for (int i = num_segments; i <= BRCNN_MAX_LEVELS; i++) sg.row0[i] = rows;
    if (rows * channels >= 0x7fffffffLL) return BRCNN_EINVAL;
    hipStream_t s = (hipStream_t)stream;
    BRCNN_HIP_CHECK(hipMemsetAsync(stats_ws, 0,
                                   (size_t)batch * num_segments * groups * 2 * sizeof(double), s));
    int chunks = (max_hw + 255) / 256;
    if (chunks > 512) chunks = 512;
    const int rpb = (max_hw + chunks - 1) / chunks;
    chunks = (max_hw + rpb - 1) / rpb;
    const long long total = rows * (channels >> 2);
    const int nstat = num_segments * batch * groups;
    if (dtype == BRCNN_DT_F32) {
        hipLaunchKernelGGL((gn_stats_kernel<float, 4>), dim3(chunks, batch * num_segments), dim3(256), 0, s,
                           (const float*)x, (double*)stats_ws, sg, batch, channels, groups, rpb);
        BRCNN_LAUNCH_CHECK();
        hipLaunchKernelGGL(gn_finalize_kernel, dim3((nstat + 255) / 256), dim3(256), 0, s,
                           (double*)stats_ws, sg, batch, channels, groups, eps);
        BRCNN_LAUNCH_CHECK();
        if (!(channels & 7))
            hipLaunchKernelGGL(gn_apply_rows_kernel<float>, dim3((max_hw + GN_APPLY_ROWS - 1) / GN_APPLY_ROWS, batch * num_segments),
                               dim3(256), 0, s, (const float*)x, (const double*)stats_ws, gamma, beta, (float*)y, sg, batch, channels,
                               groups, GN_APPLY_ROWS, relu);
        else
        hipLaunchKernelGGL((gn_apply_kernel<float, 1>), dim3(stream_grid(total)), dim3(256), 0, s,
                           (const float*)x, (const double*)stats_ws, gamma, beta, (float*)y, sg, batch,
                           channels, groups, relu);
    } else if (dtype == BRCNN_DT_BF16) {
        if (int st = gn_forward_16<bf16_t>(x, gamma, beta, y, stats_ws, sg, batch, num_segments, channels, groups, eps,
                                           relu, chunks, rpb, total, nstat, s))
            return st;
    } else {
        if (int st = gn_forward_16<f16_t>(x, gamma, beta, y, stats_ws, sg, batch, num_segments, channels, groups, eps,
                                          relu, chunks, rpb, total, nstat, s))
            return st;
    }
    BRCNN_LAUNCH_CHECK();
    return 0;
}


static int gn_setup(GnSegs& sg, long long& rows, int& max_hw, int batch, int num_segments, const int* hw_host) {
    sg = GnSegs{};
    sg.nseg = num_segments;
    rows = 0;
    max_hw = 0;
    for (int i = 0; i < num_segments; i++) {
        if (hw_host[i] <= 0) return BRCNN_EINVAL;
        sg.hw[i] = hw_host[i];
        sg.row0[i] = rows;
        rows += (long long)batch * hw_host[i];
        if (hw_host[i] > max_hw) max_hw = hw_host[i];
    }
    for (int i = num_segments; i <= BRCNN_MAX_LEVELS; i++) sg.row0[i] = rows;
    return 0;
}

static void gn_bwd_chunks(int max_hw, int* chunks, int* rpb) {
    int c = (max_hw + 255) / 256;
    if (c > 64) c = 64;
    *rpb = (max_hw + c - 1) / c;
    *chunks = (max_hw + *rpb - 1) / *rpb;
}

BRCNN_API size_t brcnn_groupnorm_nhwc_multi_backward_workspace_bytes(int batch, int num_segments, const int* hw_host,
                                                                    int channels, int groups) {
    if (batch <= 0 || num_segments <= 0 || num_segments > BRCNN_MAX_LEVELS || !hw_host || channels <= 0 || groups <= 0)
        return 0;
    int max_hw = 0;
    for (int i = 0; i < num_segments; i++) max_hw = hw_host[i] > max_hw ? hw_host[i] : max_hw;
    int chunks, rpb;
    gn_bwd_chunks(max_hw, &chunks, &rpb);
    return (size_t)batch * num_segments * groups * 2 * sizeof(double) +
           ((size_t)batch * num_segments * chunks + GN_PARAM_SLICES) * 2 * channels * sizeof(float);
}

BRCNN_API int brcnn_groupnorm_nhwc_multi_backward(const void* dy, const void* x, const void* stats,
                                                  const float* gamma, const float* beta, void* dx, float* dgamma,
                                                  float* dbeta, void* workspace, size_t workspace_bytes, int batch,
                                                  int num_segments, const int* hw_host, int channels, int groups,
                                                  int relu, int dtype, void* stream) {
    if (!dy || !x || !stats || !gamma || !beta || !dx || !dgamma || !dbeta || !workspace || batch <= 0 ||
        num_segments <= 0 || num_segments > BRCNN_MAX_LEVELS || !hw_host || channels <= 0 || channels > 256 ||
        groups <= 0 || channels % groups || (channels & 3) || !brcnn_elem_ok(dtype) ||
        (dtype != BRCNN_DT_F32 && (channels & 7)))          // 16-bit rows are read 8 channels (16 bytes) per lane
        return BRCNN_EINVAL;
    if (workspace_bytes < brcnn_groupnorm_nhwc_multi_backward_workspace_bytes(batch, num_segments, hw_host, channels, groups))
        return BRCNN_EINVAL;
    GnSegs sg;
    long long rows;
    int max_hw;
    if (gn_setup(sg, rows, max_hw, batch, num_segments, hw_host)) return BRCNN_EINVAL;
    if (rows * channels >= 0x7fffffffLL) return BRCNN_EINVAL;
    int chunks, rpb;
    gn_bwd_chunks(max_hw, &chunks, &rpb);
    hipStream_t s = (hipStream_t)stream;
    double* gsum = (double*)workspace;
    const size_t gbytes = (size_t)batch * num_segments * groups * 2 * sizeof(double);
    float* part = (float*)((char*)workspace + gbytes);
    BRCNN_HIP_CHECK(hipMemsetAsync(gsum, 0, gbytes, s));
    const long long total = rows * (channels >> 2);
    const int num_parts = batch * num_segments * chunks;
    if (dtype == BRCNN_DT_F32)
        hipLaunchKernelGGL(gn_bwd_reduce_kernel<float>, dim3(chunks, batch * num_segments), dim3(256), 0, s,
                           (const float*)x, (const float*)dy, (const double*)stats, gamma, beta, gsum, part, sg, batch,
                           channels, groups, rpb, relu);
    else if (dtype == BRCNN_DT_BF16)
        hipLaunchKernelGGL(gn_bwd_reduce_kernel<bf16_t>, dim3(chunks, batch * num_segments), dim3(256), 0, s,
                           (const bf16_t*)x, (const bf16_t*)dy, (const double*)stats, gamma, beta, gsum, part, sg,
                           batch, channels, groups, rpb, relu);
    else
        hipLaunchKernelGGL(gn_bwd_reduce_kernel<f16_t>, dim3(chunks, batch * num_segments), dim3(256), 0, s,
                           (const f16_t*)x, (const f16_t*)dy, (const double*)stats, gamma, beta, gsum, part, sg,
                           batch, channels, groups, rpb, relu);
    BRCNN_LAUNCH_CHECK();
    float* part2 = part + (size_t)num_parts * 2 * channels;
    hipLaunchKernelGGL(gn_bwd_param_kernel, dim3((2 * channels + 31) / 32, GN_PARAM_SLICES), dim3(256), 0, s, part, part2,
                       num_parts, channels);
    BRCNN_LAUNCH_CHECK();
    hipLaunchKernelGGL(gn_bwd_param_final_kernel, dim3((2 * channels + 255) / 256), dim3(256), 0, s, part2, dgamma, dbeta,
                       channels);
    BRCNN_LAUNCH_CHECK();
    // one channel quad per thread: the two-quad (16-byte) form measured 35 % SLOWER here (295 vs 208 us per
    // backward of the tower tensor in bf16), although it is the faster one in the forward apply kernel
    if (dtype == BRCNN_DT_F32 && !(channels & 7))
        hipLaunchKernelGGL(gn_bwd_apply_rows_kernel<float>, dim3((max_hw + GN_APPLY_ROWS - 1) / GN_APPLY_ROWS, batch * num_segments), dim3(256), 0, s, (const float*)x, (const float*)dy,
                               (const double*)stats, gsum, gamma, beta, (float*)dx, sg, batch, channels, groups,
                               GN_APPLY_ROWS, relu);
    else if (dtype == BRCNN_DT_F32)
        hipLaunchKernelGGL((gn_bwd_apply_kernel<float, 1>), dim3(stream_grid(total)), dim3(256), 0, s, (const float*)x,
                           (const float*)dy, (const double*)stats, gsum, gamma, beta, (float*)dx, sg, batch, channels,
                           groups, relu);
    else if ((channels & 7) == 0) {      // 16-bit, whole 16-byte channel vectors: the row-strip form
        const dim3 grid((max_hw + GN_APPLY_ROWS - 1) / GN_APPLY_ROWS, batch * num_segments);
        if (dtype == BRCNN_DT_BF16)
            hipLaunchKernelGGL(gn_bwd_apply_rows_kernel<bf16_t>, grid, dim3(256), 0, s, (const bf16_t*)x, (const bf16_t*)dy,
                               (const double*)stats, gsum, gamma, beta, (bf16_t*)dx, sg, batch, channels, groups,
                               GN_APPLY_ROWS, relu);
        else
            hipLaunchKernelGGL(gn_bwd_apply_rows_kernel<f16_t>, grid, dim3(256), 0, s, (const f16_t*)x, (const f16_t*)dy,
                               (const double*)stats, gsum, gamma, beta, (f16_t*)dx, sg, batch, channels, groups,
                               GN_APPLY_ROWS, relu);
    } else if (dtype == BRCNN_DT_BF16)
        hipLaunchKernelGGL((gn_bwd_apply_kernel<bf16_t, 1>), dim3(stream_grid(total)), dim3(256), 0, s, (const bf16_t*)x,
                           (const bf16_t*)dy, (const double*)stats, gsum, gamma, beta, (bf16_t*)dx, sg, batch, channels,
                           groups, relu);
    else
        hipLaunchKernelGGL((gn_bwd_apply_kernel<f16_t, 1>), dim3(stream_grid(total)), dim3(256), 0, s, (const f16_t*)x,
                           (const f16_t*)dy, (const double*)stats, gsum, gamma, beta, (f16_t*)dx, sg, batch, channels,
                           groups, relu);
    BRCNN_LAUNCH_CHECK();
    return 0;
}


BRCNN_API int brcnn_clock_probe(int64_t* out2, int64_t wall_ticks_100mhz, void* stream) {
    if (!out2 || wall_ticks_100mhz <= 0) return BRCNN_EINVAL;
    hipLaunchKernelGGL(clock_probe_kernel, dim3(1), dim3(1), 0, (hipStream_t)stream, (long long*)out2,
                       (long long)wall_ticks_100mhz);
    BRCNN_LAUNCH_CHECK();
    return 0;
}

BRCNN_API int brcnn_groupnorm_nhwc(const void* x, const float* gamma, const float* beta, void* y,
                                   void* stats_ws, int batch, int hw, int channels, int groups,
                                   float eps, int relu, int dtype, void* stream) {
    const int hws[1] = {hw};
    return brcnn_groupnorm_nhwc_multi(x, gamma, beta, y, stats_ws, batch, 1, hws, channels, groups,
                                      eps, relu, dtype, stream);
}

BRCNN_API int brcnn_upsample_nearest_add_nhwc(void* dst, const void* src, int batch, int hd, int wd,
                                              int hs, int ws, int channels, int dtype,
                                              void* stream) {
    if (!dst || !src || batch <= 0 || hd <= 0 || wd <= 0 || hs <= 0 || ws <= 0 || channels <= 0 ||
        (channels & 3) || !brcnn_elem_ok(dtype))
        return BRCNN_EINVAL;
    const long long total = (long long)batch * hd * wd * (channels >> 2);
    if (dtype == BRCNN_DT_F32)
        hipLaunchKernelGGL(upsample_add_kernel<float>, dim3(stream_grid(total)), dim3(256), 0,
                           (hipStream_t)stream, (float*)dst, (const float*)src, batch, hd, wd, hs, ws,
                           channels);
    else if (dtype == BRCNN_DT_BF16)
        hipLaunchKernelGGL(upsample_add_kernel<bf16_t>, dim3(stream_grid(total)), dim3(256), 0,
                           (hipStream_t)stream, (bf16_t*)dst, (const bf16_t*)src, batch, hd, wd, hs, ws,
                           channels);
    else
        hipLaunchKernelGGL(upsample_add_kernel<f16_t>, dim3(stream_grid(total)), dim3(256), 0,
                           (hipStream_t)stream, (f16_t*)dst, (const f16_t*)src, batch, hd, wd, hs, ws,
                           channels);
    BRCNN_LAUNCH_CHECK();
    return 0;
}

BRCNN_API int brcnn_upsample_nearest_add_nhwc_out(const void* dst, const void* src, void* out, int batch, int hd,
                                                  int wd, int hs, int ws, int channels, int dtype, void* stream) {
    if (!dst || !src || !out || batch <= 0 || hd <= 0 || wd <= 0 || hs <= 0 || ws <= 0 || channels <= 0 ||
        (channels & 3) || !brcnn_elem_ok(dtype))
        return BRCNN_EINVAL;
    const long long total = (long long)batch * hd * wd * (channels >> 2);
    if (dtype == BRCNN_DT_F32)
        hipLaunchKernelGGL(upsample_add_out_kernel<float>, dim3(stream_grid(total)), dim3(256), 0, (hipStream_t)stream,
                           (const float*)dst, (const float*)src, (float*)out, batch, hd, wd, hs, ws, channels);
    else if (dtype == BRCNN_DT_BF16)
        hipLaunchKernelGGL(upsample_add_out_kernel<bf16_t>, dim3(stream_grid(total)), dim3(256), 0, (hipStream_t)stream,
                           (const bf16_t*)dst, (const bf16_t*)src, (bf16_t*)out, batch, hd, wd, hs, ws, channels);
    else
        hipLaunchKernelGGL(upsample_add_out_kernel<f16_t>, dim3(stream_grid(total)), dim3(256), 0, (hipStream_t)stream,
                           (const f16_t*)dst, (const f16_t*)src, (f16_t*)out, batch, hd, wd, hs, ws, channels);
    BRCNN_LAUNCH_CHECK();
    return 0;
}

BRCNN_API int brcnn_upsample_nearest_add_nhwc_backward(const void* dout, void* dsrc, int batch, int hd, int wd, int hs,
                                                       int ws, int channels, int dtype, void* stream) {
    if (!dout || !dsrc || batch <= 0 || hd <= 0 || wd <= 0 || hs <= 0 || ws <= 0 || channels <= 0 || (channels & 3) ||
        !brcnn_elem_ok(dtype))
        return BRCNN_EINVAL;
    const long long total = (long long)batch * hs * ws * (channels >> 2);
    if (dtype == BRCNN_DT_F32)
        hipLaunchKernelGGL(upsample_add_bwd_kernel<float>, dim3(stream_grid(total)), dim3(256), 0, (hipStream_t)stream,
                           (const float*)dout, (float*)dsrc, batch, hd, wd, hs, ws, channels);
    else if (dtype == BRCNN_DT_BF16)
        hipLaunchKernelGGL(upsample_add_bwd_kernel<bf16_t>, dim3(stream_grid(total)), dim3(256), 0, (hipStream_t)stream,
                           (const bf16_t*)dout, (bf16_t*)dsrc, batch, hd, wd, hs, ws, channels);
    else
        hipLaunchKernelGGL(upsample_add_bwd_kernel<f16_t>, dim3(stream_grid(total)), dim3(256), 0, (hipStream_t)stream,
                           (const f16_t*)dout, (f16_t*)dsrc, batch, hd, wd, hs, ws, channels);
    BRCNN_LAUNCH_CHECK();
    return 0;
}

BRCNN_API size_t brcnn_colsum_workspace_bytes(int64_t rows, int channels) {
    if (rows <= 0 || channels <= 0) return 256;
    long long strips = (rows + 63) / 64;
    if (strips > 2048) strips = 2048;
    return (size_t)strips * channels * sizeof(float) + 256;
}

BRCNN_API int brcnn_colsum(const void* x, float* out, void* workspace, size_t workspace_bytes, int64_t rows,
                           int channels, int dtype, void* stream) {
    if (!x || !out || !workspace || rows < 0 || channels <= 0 || (channels & 3) ||
        !brcnn_elem_ok(dtype))
        return BRCNN_EINVAL;
    const int c4n = channels >> 2;
    if (c4n < 256 && (c4n & (c4n - 1))) return BRCNN_EINVAL;      // channel-vector count: a power of two or >= 256
    hipStream_t s = (hipStream_t)stream;
    if (rows == 0) {
        BRCNN_HIP_CHECK(hipMemsetAsync(out, 0, (size_t)channels * sizeof(float), s));
        return 0;
    }
    long long strips = (rows + 63) / 64;          // ~2048 workgroups of >= 64 rows: enough to fill the chip
    if (strips > 2048) strips = 2048;
    const int rpb = (int)((rows + strips - 1) / strips);
    strips = (rows + rpb - 1) / rpb;
    if (workspace_bytes < (size_t)strips * channels * sizeof(float)) return BRCNN_EINVAL;
    const int CW = c4n < 256 ? c4n : 256;
    const dim3 grid((unsigned)strips, (c4n + CW - 1) / CW);
    if (dtype == BRCNN_DT_F32)
        hipLaunchKernelGGL(colsum_partial_kernel<float>, grid, dim3(256), 0, s, (const float*)x, (float*)workspace,
                           (long long)rows, channels, rpb);
    else if (dtype == BRCNN_DT_BF16)
        hipLaunchKernelGGL(colsum_partial_kernel<bf16_t>, grid, dim3(256), 0, s, (const bf16_t*)x, (float*)workspace,
                           (long long)rows, channels, rpb);
    else
        hipLaunchKernelGGL(colsum_partial_kernel<f16_t>, grid, dim3(256), 0, s, (const f16_t*)x, (float*)workspace,
                           (long long)rows, channels, rpb);
    BRCNN_LAUNCH_CHECK();
    hipLaunchKernelGGL(colsum_final_kernel, dim3((channels + 15) / 16), dim3(1024), 0, s, (const float*)workspace, out,
                       (int)strips, channels);
    BRCNN_LAUNCH_CHECK();
    return 0;
}

BRCNN_API int brcnn_nchw_to_nhwc(const float* src, void* dst, int batch, int channels, int hw,
                                 int dst_dtype, void* stream) {
    if (!src || !dst || batch <= 0 || channels <= 0 || hw <= 0 || dst_dtype != BRCNN_DT_F32)
        return BRCNN_EINVAL;
    // src viewed as (N, R=C, Cc=HW) -> dst (N, HW, C)
    hipLaunchKernelGGL(transpose_kernel, dim3((hw + 31) / 32, (channels + 31) / 32, batch),
                       dim3(256), 0, (hipStream_t)stream, src, (float*)dst, channels, hw);
    BRCNN_LAUNCH_CHECK();
    return 0;
}

BRCNN_API int brcnn_nhwc_to_nchw(const void* src, float* dst, int batch, int channels, int hw,
                                 int src_dtype, void* stream) {
    if (!src || !dst || batch <= 0 || channels <= 0 || hw <= 0 || src_dtype != BRCNN_DT_F32)
        return BRCNN_EINVAL;
    hipLaunchKernelGGL(transpose_kernel, dim3((channels + 31) / 32, (hw + 31) / 32, batch),
                       dim3(256), 0, (hipStream_t)stream, (const float*)src, dst, hw, channels);
    BRCNN_LAUNCH_CHECK();
    return 0;
}


BRCNN_API int brcnn_pack_conv_weights(const float* weight, void* fwd, void* dgrad, int cout, int cin, int kh,
                                      int kw, int dtype, void* stream) {
    if (!weight || (!fwd && !dgrad) || cout <= 0 || cin <= 0 || kh <= 0 || kw <= 0 ||
        !brcnn_elem_ok(dtype))
        return BRCNN_EINVAL;
    const long long total = (long long)cout * cin * kh * kw;
    if (dtype == BRCNN_DT_F32)
        hipLaunchKernelGGL(pack_conv_weights_kernel<float>, dim3(stream_grid(total)), dim3(256), 0, (hipStream_t)stream,
                           weight, (float*)fwd, (float*)dgrad, cout, cin, kh, kw);
    else if (dtype == BRCNN_DT_BF16)
        hipLaunchKernelGGL(pack_conv_weights_kernel<bf16_t>, dim3(stream_grid(total)), dim3(256), 0,
                           (hipStream_t)stream, weight, (bf16_t*)fwd, (bf16_t*)dgrad, cout, cin, kh, kw);
    else
        hipLaunchKernelGGL(pack_conv_weights_kernel<f16_t>, dim3(stream_grid(total)), dim3(256), 0,
                           (hipStream_t)stream, weight, (f16_t*)fwd, (f16_t*)dgrad, cout, cin, kh, kw);
    BRCNN_LAUNCH_CHECK();
    return 0;
}
