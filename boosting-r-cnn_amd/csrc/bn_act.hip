// Element-wise tail of a trainable conv block in training (resnet.py Bottleneck.forward:263-302
// with norm_eval=True: BatchNorm in eval mode is the per-channel affine scale = gamma/sqrt(var+eps),
// shift = beta - mean*scale whose gamma / beta still train), fused into one pass each way:
//   forward   out = [relu]( z * scale[c] + shift[c] [+ res] )
//   backward  dpre = dout * (out > 0)            (relu mask from the saved output)
//             dz = dpre * scale[c],  dres = dpre,  dscale[c] = sum_m dpre*z,  dshift[c] = sum_m dpre
// HBM-bound streams over (rows, C) NHWC tensors, 16 bytes per lane; the per-channel sums are
// reduced per thread over a strip of rows, across the row lanes of a workgroup through LDS, and
// leave as per-strip partials that a second launch sums in a fixed order (deterministic).  The reference reaches the same
// arithmetic through ~9 separate torch kernels per block (mul, add, add, relu; threshold_backward,
// mul, mul, 2 x sum).
#include <type_traits>
#include <mutex>
#include <vector>
#include "common.h"

namespace {

template <typename T> struct Vec;
template <> struct Vec<float> { static constexpr int N = 4; };
template <> struct Vec<bf16_t> { static constexpr int N = 8; };
template <> struct Vec<f16_t> { static constexpr int N = 8; };

__device__ __forceinline__ void ldv(const float* p, float v[4]) {
    const float4 t = *reinterpret_cast<const float4*>(p);
    v[0] = t.x; v[1] = t.y; v[2] = t.z; v[3] = t.w;
}
__device__ __forceinline__ void ldv(const bf16_t* p, float v[8]) {
    const uint4 u = *reinterpret_cast<const uint4*>(p);
    const unsigned w[4] = {u.x, u.y, u.z, u.w};
#pragma unroll
    for (int e = 0; e < 4; e++) {
        v[2 * e] = __uint_as_float(w[e] << 16);
        v[2 * e + 1] = __uint_as_float(w[e] & 0xffff0000u);
    }
}
__device__ __forceinline__ unsigned f2bf(float v) { return brcnn_f2b(v); }
__device__ __forceinline__ void ldv(const f16_t* p, float v[8]) {
    const uint4 u = *reinterpret_cast<const uint4*>(p);
    const unsigned w[4] = {u.x, u.y, u.z, u.w};
#pragma unroll
    for (int e = 0; e < 4; e++) {
        v[2 * e] = brcnn_h2f((unsigned short)(w[e] & 0xffffu));
        v[2 * e + 1] = brcnn_h2f((unsigned short)(w[e] >> 16));
    }
}
__device__ __forceinline__ void stv(f16_t* p, const float v[8]) {
    uint4 u;
    u.x = brcnn_pk2h(v[0], v[1]);
    u.y = brcnn_pk2h(v[2], v[3]);
    u.z = brcnn_pk2h(v[4], v[5]);
    u.w = brcnn_pk2h(v[6], v[7]);
    *reinterpret_cast<uint4*>(p) = u;
}
__device__ __forceinline__ void stv(float* p, const float v[4]) {
    *reinterpret_cast<float4*>(p) = make_float4(v[0], v[1], v[2], v[3]);
}
__device__ __forceinline__ void stv(bf16_t* p, const float v[8]) {
    uint4 u;
    u.x = brcnn_pk2b(v[0], v[1]);
    u.y = brcnn_pk2b(v[2], v[3]);
    u.z = brcnn_pk2b(v[4], v[5]);
    u.w = brcnn_pk2b(v[6], v[7]);
    *reinterpret_cast<uint4*>(p) = u;
}

// eval-mode BatchNorm statistics (resnet.py:648-657): scale = gamma / sqrt(var + eps),
// shift = beta - mean * scale, the operation order of the torch expression they replace
struct BnStats {
    const float* mean;      // NULL: `scale` / `shift` are the affine itself
    const float* var;
    float eps;
};

__device__ __forceinline__ void bn_affine(const float* gamma, const float* beta, const BnStats& bn, int c, float& sc,
                                          float& sh) {
    if (bn.mean) {
        sc = gamma[c] / sqrtf(bn.var[c] + bn.eps);
        sh = beta ? beta[c] - bn.mean[c] * sc : 0.f;
    } else {
        sc = gamma[c];
        sh = beta ? beta[c] : 0.f;
    }
}

// the launch makes gridDim.x * 256 a multiple of C / V, so a thread meets the same channels in every
// iteration and holds their scale / shift in registers
template <typename T>
__global__ __launch_bounds__(256) void bn_act_fwd_kernel(const T* __restrict__ z,
                                                        const float* __restrict__ scale,
                                                        const float* __restrict__ shift, const BnStats bn,
                                                        const T* __restrict__ res, T* __restrict__ out,
                                                        long long rows, int C, int relu) {
    constexpr int V = Vec<T>::N;
    const int cvn = C / V;
    const long long total = rows * cvn;
    const long long first = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    // the workgroup's distinct channels (all C when C / V <= 256, else a window of 256 vectors) are
    // derived once, cooperatively, and handed out through LDS
    __shared__ float s_sc[256 * V], s_sh[256 * V];
    const int nvec = cvn < 256 ? cvn : 256;
    const int base = (int)(((long long)blockIdx.x * blockDim.x) % cvn);
    for (int i = threadIdx.x; i < nvec * V; i += 256) bn_affine(scale, shift, bn, base * V + i, s_sc[i], s_sh[i]);
    __syncthreads();
    const int v0 = (int)(threadIdx.x % nvec) * V;
    float sc[V], sh[V];
#pragma unroll
    for (int e = 0; e < V; e++) { sc[e] = s_sc[v0 + e]; sh[e] = s_sh[v0 + e]; }
    for (long long idx = first; idx < total; idx += (long long)gridDim.x * blockDim.x) {
        float v[V], r[V];
        ldv(z + idx * V, v);
        if (res) ldv(res + idx * V, r);
#pragma unroll
        for (int e = 0; e < V; e++) {
            float o = v[e] * sc[e] + sh[e];
            if (res) o += r[e];
            if (relu) o = fmaxf(o, 0.f);
            v[e] = o;
        }
        stv(out + idx * V, v);
    }
}

// grid (row strips, channel chunks of 256 vectors); block 256 threads = RL row lanes x CW
// channel vectors (CW = min(256, C/V), power of two)
template <typename T>
__global__ __launch_bounds__(256) void bn_act_bwd_kernel(const T* __restrict__ dout,
                                                        const T* __restrict__ out,
                                                        const T* __restrict__ z,
                                                        const float* __restrict__ scale,
                                                        const float* __restrict__ shift, const BnStats bn,
                                                        T* __restrict__ dz, T* __restrict__ dres,
                                                        float* __restrict__ partial, long long rows,
                                                        int C, int relu, int rows_per_block, int CW) {
    // `out` NULL with relu: the forward had no residual, the mask is recomputed as z * scale + shift > 0
    // (the forward's own fp32 expression) and the `out` stream is not read at all
    constexpr int V = Vec<T>::N;
    __shared__ float red[2][256][V];
    const bool zmask = relu && out == nullptr;
    const int cvn = C / V;
    const int cv = blockIdx.y * CW + (threadIdx.x % CW);
    const int rl = threadIdx.x / CW, RL = 256 / CW;
    const long long r0 = (long long)blockIdx.x * rows_per_block;
    const long long r1 = min(rows, r0 + rows_per_block);
    float ss[V], sh[V], sc[V], sf[V];
    {   // the CW channel vectors of this workgroup, derived once and shared by its row lanes
        float* s_sc = &red[0][0][0];
        float* s_sf = &red[1][0][0];
        for (int i = threadIdx.x; i < CW * V; i += 256) {
            const int c = blockIdx.y * CW * V + i;
            float f = 0.f, v = 0.f;
            if (c < C) bn_affine(scale, zmask ? shift : nullptr, bn, c, v, f);
            s_sc[i] = v;
            s_sf[i] = f;
        }
        __syncthreads();
#pragma unroll
        for (int e = 0; e < V; e++) {
            ss[e] = 0.f; sh[e] = 0.f;
            sc[e] = s_sc[(threadIdx.x % CW) * V + e];
            sf[e] = s_sf[(threadIdx.x % CW) * V + e];
        }
        __syncthreads();
    }
    if (cv < cvn) {
        // U rows of the strip in flight per lane (with ~512 workgroups the loads of ONE row per lane do not
        // cover the HBM latency; more workgroups would grow the per-strip partials the second stage reads).
        // Whole groups of U rows run WITHOUT per-row guards -- every load of the group is issued before the first use
        // -- and the strip's last rows one at a time.  (r04: the guarded form, `if (rr < r1)` around every row of the
        // group, compiled for fp16 into exec-masked blocks whose partial sums were re-paired through swizzled packed
        // adds; under a co-running weight-gradient kernel two of the eight lanes of dgamma were intermittently wrong:
        // tests/test_stress_gpu.py)  The per-lane accumulation order -- rows r, r + RL, r + 2 RL, ... -- is
        // the same in both loops, so the sums are the guarded form's bits.
        constexpr int U = 4;
        // (the mask source is a launch constant: deciding it per load made the compiler branch around every `out` load
        // and drain the queue behind it)
        auto rows = [&](auto n_c, auto out_c, long long r) {
            constexpr int N = decltype(n_c)::value;
            constexpr bool OUT = decltype(out_c)::value;        // ReLU mask from the saved output (residual form)
            float g[N][V], o[OUT ? N : 1][V], zz[N][V];
#pragma unroll
            for (int u = 0; u < N; u++) {
                const long long idx = ((r + (long long)u * RL) * cvn + cv) * V;
                ldv(dout + idx, g[u]);
                if constexpr (OUT) ldv(out + idx, o[u]);
                ldv(z + idx, zz[u]);
            }
#pragma unroll
            for (int u = 0; u < N; u++) {
                const long long idx = ((r + (long long)u * RL) * cvn + cv) * V;
                float gz[V];
#pragma unroll
                for (int e = 0; e < V; e++) {
                    float ov;
                    if constexpr (OUT) ov = o[u][e];
                    else ov = zz[u][e] * sc[e] + sf[e];
                    const float d = (!relu || ov > 0.f) ? g[u][e] : 0.f;
                    g[u][e] = d;
                    ss[e] += d * zz[u][e];
                    sh[e] += d;
                    gz[e] = d * sc[e];
                }
                stv(dz + idx, gz);
                if (dres) stv(dres + idx, g[u]);
            }
        };
        long long r = r0 + rl;
        if (relu && !zmask) {
            for (; r + (long long)(U - 1) * RL < r1; r += (long long)RL * U) rows(std::integral_constant<int, U>{}, std::true_type{}, r);
            for (; r < r1; r += RL) rows(std::integral_constant<int, 1>{}, std::true_type{}, r);
        } else {
            for (; r + (long long)(U - 1) * RL < r1; r += (long long)RL * U) rows(std::integral_constant<int, U>{}, std::false_type{}, r);
            for (; r < r1; r += RL) rows(std::integral_constant<int, 1>{}, std::false_type{}, r);
        }
    }
#pragma unroll
    for (int e = 0; e < V; e++) { red[0][threadIdx.x][e] = ss[e]; red[1][threadIdx.x][e] = sh[e]; }
    __syncthreads();
    if (rl == 0 && cv < cvn) {
#pragma unroll
        for (int e = 0; e < V; e++) {
            float a = 0.f, b = 0.f;
            for (int k = 0; k < RL; k++) { a += red[0][k * CW + threadIdx.x][e]; b += red[1][k * CW + threadIdx.x][e]; }
            partial[((size_t)blockIdx.x * 2 + 0) * C + cv * V + e] = a;      // [strip][dscale | dshift][C]
            partial[((size_t)blockIdx.x * 2 + 1) * C + cv * V + e] = b;
        }
    }
}

// second stage: deterministic column sums of the per-strip partials
__global__ __launch_bounds__(1024) void bn_act_reduce_kernel(const float* __restrict__ partial,
                                                            float* __restrict__ dscale,
                                                            float* __restrict__ dshift, int strips, int C) {
    __shared__ float red[16][64];
    const int col = blockIdx.x * 64 + (threadIdx.x & 63), sl = threadIdx.x >> 6;
    const int which = blockIdx.y;
    float a = 0.f;
    if (col < C)
        for (int s = sl; s < strips; s += 16) a += partial[((size_t)s * 2 + which) * C + col];
    red[sl][threadIdx.x & 63] = a;
    __syncthreads();
    if (sl == 0 && col < C) {
        float t = 0.f;
#pragma unroll
        for (int k = 0; k < 16; k++) t += red[k][threadIdx.x];
        (which ? dshift : dscale)[col] = t;
    }
}

// second stage when the affine is an eval-mode BatchNorm: the gradients of gamma / beta themselves,
//   dbeta = sum dshift,   dgamma = (sum dscale - mean * sum dshift) / sqrt(var + eps)
// (16 columns x 64 strip lanes per workgroup: C / 16 workgroups, the launch is latency sized)
__global__ __launch_bounds__(1024) void bn_eval_reduce_kernel(const float* __restrict__ partial, const BnStats bn,
                                                             float* __restrict__ dgamma, float* __restrict__ dbeta,
                                                             int strips, int C) {
    __shared__ float red[2][64][16];
    const int col = blockIdx.x * 16 + (threadIdx.x & 15), sl = threadIdx.x >> 4;
    float a = 0.f, b = 0.f;
    if (col < C)
        for (int s = sl; s < strips; s += 64) {
            a += partial[((size_t)s * 2 + 0) * C + col];
            b += partial[((size_t)s * 2 + 1) * C + col];
        }
    red[0][sl][threadIdx.x & 15] = a;
    red[1][sl][threadIdx.x & 15] = b;
    __syncthreads();
    if (sl == 0 && col < C) {
        float ta = 0.f, tb = 0.f;
#pragma unroll
        for (int k = 0; k < 64; k++) { ta += red[0][k][threadIdx.x]; tb += red[1][k][threadIdx.x]; }
        dgamma[col] = (ta - bn.mean[col] * tb) / sqrtf(bn.var[col] + bn.eps);
        dbeta[col] = tb;
    }
}

// ... of MANY layers in one launch (brcnn_bn_reduce_flush): the second stages of a backward pass are ~40 launches of
// 16 - 128 workgroups each, serial on the main stream (0.29 ms per bf16 train step of bench.py); nothing reads dgamma /
// dbeta before the optimizer, so they are recorded and run together.  Same columns per workgroup, same sums in the same
// order as bn_eval_reduce_kernel: same bits.
constexpr int RED_TAB = 48;
struct RedEntry {
    const float* partial;
    const float* mean;
    const float* var;
    float* dgamma;
    float* dbeta;
    float eps;
    int strips, C, blk0;
};
struct RedTable {
    int count;
    RedEntry e[RED_TAB];
};
__global__ __launch_bounds__(1024) void bn_eval_reduce_batch_kernel(const RedTable t) {
    __shared__ float red[2][64][16];
    int lo = 0, hi = t.count - 1;
    while (lo < hi) {
        const int mid = (lo + hi + 1) >> 1;
        if (t.e[mid].blk0 <= (int)blockIdx.x) lo = mid; else hi = mid - 1;
    }
    const RedEntry& e = t.e[lo];
    const int C = e.C, strips = e.strips;
    const int col = ((int)blockIdx.x - e.blk0) * 16 + (threadIdx.x & 15), sl = threadIdx.x >> 4;
    float a = 0.f, b = 0.f;
    if (col < C)
        for (int s = sl; s < strips; s += 64) {
            a += e.partial[((size_t)s * 2 + 0) * C + col];
            b += e.partial[((size_t)s * 2 + 1) * C + col];
        }
    red[0][sl][threadIdx.x & 15] = a;
    red[1][sl][threadIdx.x & 15] = b;
    __syncthreads();
    if (sl == 0 && col < C) {
        float ta = 0.f, tb = 0.f;
#pragma unroll
        for (int k = 0; k < 64; k++) { ta += red[0][k][threadIdx.x]; tb += red[1][k][threadIdx.x]; }
        e.dgamma[col] = (ta - e.mean[col] * tb) / sqrtf(e.var[col] + e.eps);
        e.dbeta[col] = tb;
    }
}

struct BwdPlan { int CW, chunks; long long strips, rpb; };
inline bool bwd_plan(long long rows, int channels, int V, BwdPlan* pl) {
    const int cvn = channels / V;
    // short maps (stage 4: 8 400 rows x 512 / 2048 channels) split the channels too, so that ~1024 workgroups
    // of >= 32 rows exist; long maps keep whole rows per workgroup and ~512 strips (the second stage reads
    // strips x 2 x C partials, more strips cost there what they gain here)
    const bool short_map = rows < 16384;
    const int cw_max = short_map ? 64 : 256;
    const int wg_target = short_map ? 1024 : 512;
    const int min_rows = short_map ? 32 : 64;
    int CW = 1;
    while (CW < cvn && CW < cw_max) CW <<= 1;
    if (cvn < cw_max && CW != cvn) return false;      // channel-vector count must be a power of two
    pl->CW = CW;
    pl->chunks = (cvn + CW - 1) / CW;
    long long strips = wg_target / pl->chunks;
    if (strips < 1) strips = 1;
    long long rpb = (rows + strips - 1) / strips;
    if (rpb < min_rows) rpb = min_rows;
    pl->rpb = rpb;
    pl->strips = rows > 0 ? (rows + rpb - 1) / rpb : 0;
    return true;
}

inline int stream_grid(long long total) {
    long long g = (total + 255) / 256;
    return (int)(g > 16384 ? 16384 : (g < 1 ? 1 : g));
}

// forward grid: gridDim.x * 256 must be a multiple of the channel-vector count (a power of two or a
// multiple handled by rounding the grid up)
inline int fwd_grid(long long total, int cvn, bool bn_mode) {
    long long g = stream_grid(total);
    // BN form: every thread derives its scale / shift first (correctly rounded sqrt + divide per channel),
    // so fewer, longer-running threads (4 workgroups per CU) amortise that prologue
    const long long cap = bn_mode ? 2048 : 4096;
    if (g > cap) g = cap;
    long long step = 1;
    while ((step * 256) % cvn) step++;
    g = (g + step - 1) / step * step;
    return (int)g;
}

int forward_impl(const void* z, const float* scale, const float* shift, const BnStats bn, const void* residual,
                 void* out, int64_t rows, int channels, int relu, int dtype, void* stream) {
    if (!z || !scale || !shift || !out || rows < 0 || channels <= 0 ||
        !brcnn_elem_ok(dtype))
        return BRCNN_EINVAL;
    const int V = dtype == BRCNN_DT_F32 ? 4 : 8;
    if (channels % V) return BRCNN_EINVAL;
    if (rows == 0) return 0;
    const int cvn = channels / V;
    const long long total = rows * cvn;
    const int grid = fwd_grid(total, cvn, bn.mean != nullptr);
    if (dtype == BRCNN_DT_F32)
        hipLaunchKernelGGL(bn_act_fwd_kernel<float>, dim3(grid), dim3(256), 0, (hipStream_t)stream,
                           (const float*)z, scale, shift, bn, (const float*)residual, (float*)out, (long long)rows,
                           channels, relu);
    else if (dtype == BRCNN_DT_BF16)
        hipLaunchKernelGGL(bn_act_fwd_kernel<bf16_t>, dim3(grid), dim3(256), 0, (hipStream_t)stream,
                           (const bf16_t*)z, scale, shift, bn, (const bf16_t*)residual, (bf16_t*)out, (long long)rows,
                           channels, relu);
    else
        hipLaunchKernelGGL(bn_act_fwd_kernel<f16_t>, dim3(grid), dim3(256), 0, (hipStream_t)stream,
                           (const f16_t*)z, scale, shift, bn, (const f16_t*)residual, (f16_t*)out, (long long)rows,
                           channels, relu);
    BRCNN_LAUNCH_CHECK();
    return 0;
}

}  // namespace

// second stage over per-tile partials written by another kernel (the data-gradient epilogue of
// conv_igemm_bf16.hip, MODE 2): partials (strips, 2 [sum d*z | sum d], C)
// deferred second stages: (stream, entry), in issue order
namespace {
struct RedPending { hipStream_t s; RedEntry e; };
std::mutex g_red_mutex;
std::vector<RedPending> g_red_pending;
}

int brcnn_bn_eval_reduce_launch(const float* partials, int strips, const float* mean, const float* var, float eps,
                                float* dgamma, float* dbeta, int channels, hipStream_t s, int defer) {
    if (defer) {
        std::lock_guard<std::mutex> lock(g_red_mutex);
        RedPending p;
        p.s = s;
        p.e = RedEntry{partials, mean, var, dgamma, dbeta, eps, strips, channels, 0};
        g_red_pending.push_back(p);
        return 0;
    }
    BnStats bn = {mean, var, eps};
    hipLaunchKernelGGL(bn_eval_reduce_kernel, dim3((channels + 15) / 16), dim3(1024), 0, s, partials, bn, dgamma, dbeta,
                       strips, channels);
    BRCNN_LAUNCH_CHECK();
    return 0;
}

BRCNN_API int brcnn_bn_act_forward(const void* z, const float* scale, const float* shift,
                                   const void* residual, void* out, int64_t rows, int channels,
                                   int relu, int dtype, void* stream) {
    BnStats bn = {nullptr, nullptr, 0.f};
    return forward_impl(z, scale, shift, bn, residual, out, rows, channels, relu, dtype, stream);
}

BRCNN_API int brcnn_bn_eval_act_forward(const void* z, const float* gamma, const float* beta, const float* mean,
                                        const float* var, float eps, const void* residual, void* out, int64_t rows,
                                        int channels, int relu, int dtype, void* stream) {
    if (!mean || !var) return BRCNN_EINVAL;
    BnStats bn = {mean, var, eps};
    return forward_impl(z, gamma, beta, bn, residual, out, rows, channels, relu, dtype, stream);
}

BRCNN_API size_t brcnn_bn_act_backward_workspace_bytes(int64_t rows, int channels, int dtype) {
    BwdPlan pl;
    const int V = dtype == BRCNN_DT_F32 ? 4 : 8;
    if (rows < 0 || channels <= 0 || channels % V || !bwd_plan(rows, channels, V, &pl)) return 0;
    return (size_t)(pl.strips > 0 ? pl.strips : 1) * 2 * channels * sizeof(float);
}

static int backward_impl(const void* dout, const void* out, const void* z, const float* scale, const float* shift,
                         const BnStats bn, void* dz, void* dres, float* dscale, float* dshift, void* workspace,
                         size_t workspace_bytes, int64_t rows, int channels, int relu, int dtype, void* stream, int defer = 0) {
    if (!dout || !z || !scale || !dz || !dscale || !dshift || !workspace || rows < 0 || channels <= 0 ||
        (relu && !out && !shift) || !brcnn_elem_ok(dtype))
        return BRCNN_EINVAL;
    const int V = dtype == BRCNN_DT_F32 ? 4 : 8;
    if (channels % V) return BRCNN_EINVAL;
    BwdPlan pl;
    if (!bwd_plan(rows, channels, V, &pl)) return BRCNN_EINVAL;
    if (workspace_bytes < brcnn_bn_act_backward_workspace_bytes(rows, channels, dtype)) return BRCNN_EINVAL;
    hipStream_t s = (hipStream_t)stream;
    if (rows == 0) {
        BRCNN_HIP_CHECK(hipMemsetAsync(dscale, 0, (size_t)channels * sizeof(float), s));
        BRCNN_HIP_CHECK(hipMemsetAsync(dshift, 0, (size_t)channels * sizeof(float), s));
        return 0;
    }
    if (dtype == BRCNN_DT_F32)
        hipLaunchKernelGGL(bn_act_bwd_kernel<float>, dim3((unsigned)pl.strips, pl.chunks), dim3(256), 0, s,
                           (const float*)dout, (const float*)out, (const float*)z, scale, shift, bn, (float*)dz, (float*)dres,
                           (float*)workspace, (long long)rows, channels, relu, (int)pl.rpb, pl.CW);
    else if (dtype == BRCNN_DT_BF16)
        hipLaunchKernelGGL(bn_act_bwd_kernel<bf16_t>, dim3((unsigned)pl.strips, pl.chunks), dim3(256), 0, s,
                           (const bf16_t*)dout, (const bf16_t*)out, (const bf16_t*)z, scale, shift, bn, (bf16_t*)dz,
                           (bf16_t*)dres, (float*)workspace, (long long)rows, channels, relu, (int)pl.rpb, pl.CW);
    else
        hipLaunchKernelGGL(bn_act_bwd_kernel<f16_t>, dim3((unsigned)pl.strips, pl.chunks), dim3(256), 0, s,
                           (const f16_t*)dout, (const f16_t*)out, (const f16_t*)z, scale, shift, bn, (f16_t*)dz,
                           (f16_t*)dres, (float*)workspace, (long long)rows, channels, relu, (int)pl.rpb, pl.CW);
    BRCNN_LAUNCH_CHECK();
    if (bn.mean)
        return brcnn_bn_eval_reduce_launch((const float*)workspace, (int)pl.strips, bn.mean, bn.var, bn.eps, dscale, dshift,
                                           channels, s, defer);
    else
        hipLaunchKernelGGL(bn_act_reduce_kernel, dim3((channels + 63) / 64, 2), dim3(1024), 0, s,
                           (const float*)workspace, dscale, dshift, (int)pl.strips, channels);
    BRCNN_LAUNCH_CHECK();
    return 0;
}

BRCNN_API int brcnn_bn_act_backward(const void* dout, const void* out, const void* z, const float* scale,
                                    void* dz, void* dres, float* dscale, float* dshift, void* workspace,
                                    size_t workspace_bytes, int64_t rows, int channels, int relu, int dtype,
                                    void* stream) {
    BnStats bn = {nullptr, nullptr, 0.f};
    return backward_impl(dout, out, z, scale, nullptr, bn, dz, dres, dscale, dshift, workspace, workspace_bytes, rows, channels,
                         relu, dtype, stream);
}

BRCNN_API int brcnn_bn_eval_act_backward(const void* dout, const void* out, const void* z, const float* gamma,
                                         const float* beta, const float* mean, const float* var, float eps, void* dz, void* dres,
                                         float* dgamma, float* dbeta, void* workspace, size_t workspace_bytes,
                                         int64_t rows, int channels, int relu, int dtype, void* stream) {
    return brcnn_bn_eval_act_backward_ex(dout, out, z, gamma, beta, mean, var, eps, dz, dres, dgamma, dbeta, workspace,
                                         workspace_bytes, rows, channels, relu, dtype, stream, 0);
}

BRCNN_API int brcnn_bn_eval_act_backward_ex(const void* dout, const void* out, const void* z, const float* gamma,
                                            const float* beta, const float* mean, const float* var, float eps, void* dz,
                                            void* dres, float* dgamma, float* dbeta, void* workspace, size_t workspace_bytes,
                                            int64_t rows, int channels, int relu, int dtype, void* stream,
                                            int defer_second_stage) {
    if (!mean || !var) return BRCNN_EINVAL;
    BnStats bn = {mean, var, eps};
    if (!out && dres) return BRCNN_EINVAL;       // a residual forward needs its output for the mask
    return backward_impl(dout, out, z, gamma, beta, bn, dz, dres, dgamma, dbeta, workspace, workspace_bytes, rows, channels,
                         relu, dtype, stream, rows > 0 ? defer_second_stage : 0);
}

BRCNN_API int brcnn_bn_reduce_pending(void) {
    std::lock_guard<std::mutex> lock(g_red_mutex);
    return (int)g_red_pending.size();
}

BRCNN_API int brcnn_bn_reduce_flush(void* stream) {
    std::vector<RedPending> take;
    {
        std::lock_guard<std::mutex> lock(g_red_mutex);
        std::vector<RedPending> keep;
        for (const RedPending& p : g_red_pending) (p.s == (hipStream_t)stream ? take : keep).push_back(p);
        g_red_pending.swap(keep);
    }
    hipStream_t s = (hipStream_t)stream;
    for (size_t first = 0; first < take.size(); first += RED_TAB) {
        RedTable t;
        t.count = (int)(take.size() - first < (size_t)RED_TAB ? take.size() - first : (size_t)RED_TAB);
        int blk = 0;
        for (int i = 0; i < t.count; i++) {
            t.e[i] = take[first + i].e;
            t.e[i].blk0 = blk;
            blk += (t.e[i].C + 15) / 16;
        }
        hipLaunchKernelGGL(bn_eval_reduce_batch_kernel, dim3(blk), dim3(1024), 0, s, t);
        BRCNN_LAUNCH_CHECK();
    }
    return (int)take.size();
}
