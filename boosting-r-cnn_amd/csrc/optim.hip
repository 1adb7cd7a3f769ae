// Fused optimizer step of the train loop, for gfx950.
//
// The reference steps with torch.optim.SGD(lr, momentum 0.9, weight_decay 1e-4) after mmcv's OptimizerHook clipped
// the global gradient norm to 35 (configs/_base_/schedules/schedule_1x.py:2-3, boosting_rcnn_r50_pafpn_1x_utdac.py:130
// `optimizer_config = dict(grad_clip=dict(max_norm=35, norm_type=2))`): per step ~40 multi-tensor launches for the
// norm, the clip and the update, plus -- on this path -- one weight-packing launch per trainable conv at the next
// forward.  Here the whole step is
//   sqnorm (two fixed-order stages over a table of gradient tensors)  ->  clip coefficient on the device
//   update (one pass over every parameter: g * coef [/ loss scale] + wd * w, momentum, w -= lr * buf; skipped
//           as a whole when the norm is not finite -- the GradScaler rule of the fp16 recipes)
//   pack   (the conv weights' forward / data-gradient operands of the NEXT step, (Cout,KH,KW,Cin) and flipped
//           (Cin,KH,KW,Cout) in the compute dtype, through 32 x 32 LDS tiles)
// with the tensor tables passed by value (the gradient pointers change every step), a few launches in all.
// HBM-bound streams: update reads w, g, buf and writes w, buf once (16 B + 8 B per parameter).
#include "common.h"

namespace {

constexpr int TAB = 64;            // tensors per launch (kernel-argument table: 64 x 48 B)

struct SgdEntry {
    float* w;
    const float* g;
    float* buf;
    int blk0;          // first workgroup of this tensor (each workgroup: 1024 elements)
    int n;
    float lr, wd;
    int has_buf;       // 0: first step of this parameter (buf = d_p, as torch initialises it)
};
struct SgdTable {
    int count, nblocks;
    SgdEntry e[TAB];
};

__global__ __launch_bounds__(256) void sqnorm_partial_kernel(const SgdTable t, float* __restrict__ partial) {
    __shared__ float red[4];
    int lo = 0, hi = t.count - 1;
    while (lo < hi) {
        const int mid = (lo + hi + 1) >> 1;
        if (t.e[mid].blk0 <= (int)blockIdx.x) lo = mid; else hi = mid - 1;
    }
    const SgdEntry& e = t.e[lo];
    const int base = ((int)blockIdx.x - e.blk0) * 1024;
    float s = 0.f;
    float v[4];
#pragma unroll
    for (int j = 0; j < 4; j++) {       // (unconditional loads, all four in flight; an index past the end re-reads the last element, weight 0)
        const int i = base + j * 256 + threadIdx.x;
        const float g = e.g[min(i, e.n - 1)];
        v[j] = i < e.n ? g : 0.f;
    }
#pragma unroll
    for (int j = 0; j < 4; j++) s += v[j] * v[j];
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) s += __shfl_xor(s, d, 64);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) partial[blockIdx.x] = red[0] + red[1] + red[2] + red[3];
}

// out[0] = total gradient norm (of the UNSCALED gradients), out[1] = factor the update multiplies g by
// (inv_scale * min(1, max_norm / (norm + 1e-6)); 0 when the norm is not finite), out[2] = 1 if the step is skipped
__global__ __launch_bounds__(1024) void sqnorm_final_kernel(const float* __restrict__ partial, int n, float inv_scale,
                                                           float max_norm, int skip_nonfinite, float* __restrict__ out) {
    __shared__ double red[16];
    double s = 0.0;
    for (int i = threadIdx.x; i < n; i += 1024) s += (double)partial[i];
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) s += __shfl_xor(s, d, 64);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) {
        double tsum = 0.0;
        for (int k = 0; k < 16; k++) tsum += red[k];
        const float norm = (float)sqrt(tsum) * inv_scale;
        const bool finite = norm == norm && norm <= 3.402823466e+38f;
        float coef = inv_scale;
        if (max_norm > 0.f) {
            const float c = max_norm / (norm + 1e-6f);           // torch.nn.utils.clip_grad_norm_
            coef = inv_scale * (c < 1.f ? c : (c != c ? c : 1.f));   // clamp(max=1) keeps a NaN factor, as torch does
        }
        out[0] = norm;
        // skip_nonfinite (loss-scaled fp16 recipes): GradScaler.step's rule.  Without it the update runs as
        // clip_grad_norm_ + SGD would: an inf / NaN gradient reaches the weights and the divergence is visible
        const bool skip = !finite && skip_nonfinite;
        out[1] = skip ? 0.f : coef;
        out[2] = skip ? 1.f : 0.f;
    }
}

__global__ __launch_bounds__(256) void sgd_update_kernel(const SgdTable t, const float* __restrict__ ctl, float momentum) {
    const float coef = ctl ? ctl[1] : 1.f;
    if (ctl && ctl[2] != 0.f) return;                           // non-finite gradients: the whole step is skipped
    int lo = 0, hi = t.count - 1;
    while (lo < hi) {
        const int mid = (lo + hi + 1) >> 1;
        if (t.e[mid].blk0 <= (int)blockIdx.x) lo = mid; else hi = mid - 1;
    }
    const SgdEntry& e = t.e[lo];
    const int base = ((int)blockIdx.x - e.blk0) * 1024;
    // four consecutive elements per thread as 16-byte accesses where the tensor allows it (all but a handful do): the
    // three loads of a thread go out together.  The scalar form below has each load under `if (i < n)` inside the j loop,
    // which the compiler serialises (one 4-byte load in flight per wave); same arithmetic per element either way.
    if ((e.n & 3) == 0 && ((((uintptr_t)e.w) | ((uintptr_t)e.g) | ((uintptr_t)e.buf)) & 15) == 0) {
        const int i = base + 4 * (int)threadIdx.x;
        if (i < e.n) {
            const float4 w4 = *reinterpret_cast<const float4*>(e.w + i);
            const float4 g4 = *reinterpret_cast<const float4*>(e.g + i);
            float4 b4 = make_float4(0.f, 0.f, 0.f, 0.f);
            if (momentum != 0.f && e.has_buf) b4 = *reinterpret_cast<const float4*>(e.buf + i);
            const float w[4] = {w4.x, w4.y, w4.z, w4.w}, g[4] = {g4.x, g4.y, g4.z, g4.w}, bo[4] = {b4.x, b4.y, b4.z, b4.w};
            float wn[4], bn[4];
#pragma unroll
            for (int k = 0; k < 4; k++) {
                float d = g[k] * coef;
                if (e.wd != 0.f) d = d + e.wd * w[k];
                float b = d;
                if (momentum != 0.f && e.has_buf) b = momentum * bo[k] + d;
                bn[k] = b;
                wn[k] = w[k] - e.lr * b;
            }
            if (momentum != 0.f) *reinterpret_cast<float4*>(e.buf + i) = make_float4(bn[0], bn[1], bn[2], bn[3]);
            *reinterpret_cast<float4*>(e.w + i) = make_float4(wn[0], wn[1], wn[2], wn[3]);
        }
        return;
    }
#pragma unroll
    for (int j = 0; j < 4; j++) {
        const int i = base + j * 256 + threadIdx.x;
        if (i < e.n) {
            const float w = e.w[i];
            float d = e.g[i] * coef;
            if (e.wd != 0.f) d = d + e.wd * w;                   // d_p = d_p.add(p, alpha=weight_decay)
            float b = d;
            if (momentum != 0.f) {
                if (e.has_buf) b = momentum * e.buf[i] + d;      // buf.mul_(momentum).add_(d_p)
                e.buf[i] = b;
            }
            e.w[i] = w - e.lr * b;                               // p.add_(buf, alpha=-lr)
        }
    }
}

struct PackEntry {
    const float* w;    // (Cout, Cin, KH, KW) fp32 master weight; src_cl: stored channels-last, i.e. (Cout, KH, KW, Cin)
    int src_cl;
    void* fwd;         // (Cout, KH, KW, Cin)
    void* dgrad;       // (Cin, KH, KW, Cout), taps flipped; may be NULL
    int cout, cin, kh, kw;
    int blk0;          // first workgroup; workgroups of a tensor = tiles_ci * tiles_co * taps
};
struct PackTable {
    int count, nblocks;
    PackEntry e[TAB];
};

__device__ __forceinline__ void st1(float* p, float v) { *p = v; }
__device__ __forceinline__ void st1(bf16_t* p, float v) { *p = brcnn_f2b(v); }
__device__ __forceinline__ void st1(f16_t* p, float v) { p->v = brcnn_f2h(v); }

template <typename T>
__global__ __launch_bounds__(256) void pack_batch_kernel(const PackTable t, const float* __restrict__ ctl) {
    __shared__ float tile[32][33];
    if (ctl && ctl[2] != 0.f) return;                           // step skipped: the packed copies are still current
    int lo = 0, hi = t.count - 1;
    while (lo < hi) {
        const int mid = (lo + hi + 1) >> 1;
        if (t.e[mid].blk0 <= (int)blockIdx.x) lo = mid; else hi = mid - 1;
    }
    const PackEntry& e = t.e[lo];
    const int taps = e.kh * e.kw;
    const int tci = (e.cin + 31) >> 5, tco = (e.cout + 31) >> 5;
    int r = (int)blockIdx.x - e.blk0;
    const int tap = r % taps; r /= taps;
    const int ci0 = (r % tci) * 32, co0 = (r / tci) * 32;
    (void)tco;
    const int a = tap / e.kw, b = tap - a * e.kw;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
    {                                                             // rows = co, columns = ci
        // the four loads of a thread go out together (clamped index + select: under the bounds test each waited for the
        // one before it)
        float v[4];
#pragma unroll
        for (int u = 0; u < 4; u++) {
            const int co = min(co0 + ty + 8 * u, e.cout - 1), ci = min(ci0 + tx, e.cin - 1);
            const size_t src = e.src_cl ? ((size_t)co * taps + tap) * e.cin + ci : ((size_t)co * e.cin + ci) * taps + tap;
            v[u] = e.w[src];
        }
#pragma unroll
        for (int u = 0; u < 4; u++) tile[ty + 8 * u][tx] = (co0 + ty + 8 * u < e.cout && ci0 + tx < e.cin) ? v[u] : 0.f;
    }
    __syncthreads();
    T* fwd = reinterpret_cast<T*>(e.fwd);
    T* dg = reinterpret_cast<T*>(e.dgrad);
    if (fwd)
        for (int j = ty; j < 32; j += 8) {
            const int co = co0 + j, ci = ci0 + tx;
            if (co < e.cout && ci < e.cin) st1(fwd + ((size_t)co * taps + tap) * e.cin + ci, tile[j][tx]);
        }
    if (dg)
        for (int j = ty; j < 32; j += 8) {                        // rows = ci, columns = co
            const int ci = ci0 + j, co = co0 + tx;
            if (co < e.cout && ci < e.cin)
                st1(dg + (((size_t)ci * e.kh + (e.kh - 1 - a)) * e.kw + (e.kw - 1 - b)) * e.cout + co, tile[tx][j]);
        }
}


// ---- the first FC of the box head: master weight (out, C, P) [the reference flattens RoI features as (C, ph, pw)] ->
// forward operand (out, P, C) [the NHWC RoI features' K order] and data-gradient operand (P C, out).  Two streaming
// passes instead of an aten permute copy (51 MB read + 51 MB written in fp32 at 4096 x 12544) followed by the generic
// packing launch (32 x 32 tiles, 2-byte stores: 0.7 TB/s on this shape):
//   fc_perm_kernel: one workgroup per output row and 64-channel slab; the (64, P) block is read as it lies, turned in
//     LDS, and leaves as P rows of 64 converted channels (128-byte pieces);
//   transpose16_kernel: (rows, cols) -> (cols, rows) of 16-bit / 32-bit elements through 64 x 64 LDS tiles.
template <typename T>
__global__ __launch_bounds__(256) void fc_perm_kernel(const float* __restrict__ w, T* __restrict__ fwd, int out, int c, int pn,
                                                      const float* __restrict__ ctl) {
    extern __shared__ float fc_tile[];          // [64][pn + 1]
    if (ctl && ctl[2] != 0.f) return;
    const int slabs = c >> 6;
    const int co = blockIdx.x / slabs, c0 = (blockIdx.x - co * slabs) * 64;
    const float* src = w + ((size_t)co * c + c0) * pn;              // 64 x pn contiguous floats
    const int n = 64 * pn, pitch = pn + 1;
    for (int i = threadIdx.x; i < n; i += 256) {
        const int cc = i / pn, p = i - cc * pn;
        fc_tile[cc * pitch + p] = src[i];
    }
    __syncthreads();
    T* dst = fwd + (size_t)co * pn * c + c0;
    for (int i = threadIdx.x; i < n; i += 256) {
        const int p = i >> 6, cc = i & 63;
        st1(dst + (size_t)p * c + cc, fc_tile[cc * pitch + p]);
    }
}

template <typename T>
__global__ __launch_bounds__(256) void transpose16_kernel(const T* __restrict__ src, T* __restrict__ dst, int rows, int cols,
                                                          const float* __restrict__ ctl) {
    __shared__ T tt[64][66];
    if (ctl && ctl[2] != 0.f) return;
    const int tiles_c = (cols + 63) >> 6;
    const int r0 = (blockIdx.x / tiles_c) * 64, c0 = (blockIdx.x % tiles_c) * 64;
    const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
    for (int j = ty; j < 64; j += 4)
        if (r0 + j < rows && c0 + tx < cols) tt[j][tx] = src[(size_t)(r0 + j) * cols + c0 + tx];
    __syncthreads();
    for (int j = ty; j < 64; j += 4)
        if (c0 + j < cols && r0 + tx < rows) dst[(size_t)(c0 + j) * rows + r0 + tx] = tt[tx][j];
}

}  // namespace

BRCNN_API size_t brcnn_sgd_workspace_bytes(int num_tensors, const int64_t* numel_host) {
    size_t blocks = 0;
    for (int i = 0; i < num_tensors; i++) blocks += (size_t)((numel_host[i] + 1023) / 1024);
    return blocks * sizeof(float) + 256;
}

// params / grads / bufs: HOST arrays of device pointers (fp32, dense); lr / weight_decay per tensor; has_buf per tensor.
// max_norm <= 0: no clipping (the norm is still computed when ctl3 is given).  ctl3 (device, 3 floats) receives
// [grad norm, applied factor, skipped]; inv_scale = 1 / loss scale; skip_nonfinite != 0: a non-finite norm skips the step.
BRCNN_API int brcnn_sgd_step(float* const* params, const float* const* grads, float* const* bufs, const int64_t* numel_host,
                             const float* lr_host, const float* wd_host, const int* has_buf_host, int num_tensors,
                             float momentum, float max_norm, float inv_scale, int skip_nonfinite, void* workspace,
                             size_t workspace_bytes, float* ctl3, void* stream) {
    if (num_tensors <= 0) return 0;
    if (!params || !grads || !numel_host || !lr_host || !wd_host || !has_buf_host || !workspace || !ctl3 ||
        (momentum != 0.f && !bufs))
        return BRCNN_EINVAL;
    if (workspace_bytes < brcnn_sgd_workspace_bytes(num_tensors, numel_host) - 256) return BRCNN_EINVAL;
    hipStream_t s = (hipStream_t)stream;
    float* partial = (float*)workspace;
    // pass 1: squared norm, tables of TAB tensors
    int total_blocks = 0;
    for (int first = 0; first < num_tensors; first += TAB) {
        SgdTable t;
        t.count = num_tensors - first < TAB ? num_tensors - first : TAB;
        int blk = 0;
        for (int i = 0; i < t.count; i++) {
            const int k = first + i;
            if (numel_host[k] <= 0 || numel_host[k] > 0x7fffffffLL || !params[k] || !grads[k]) return BRCNN_EINVAL;
            t.e[i].w = params[k]; t.e[i].g = grads[k]; t.e[i].buf = bufs ? bufs[k] : nullptr;
            t.e[i].blk0 = blk; t.e[i].n = (int)numel_host[k]; t.e[i].lr = lr_host[k]; t.e[i].wd = wd_host[k];
            t.e[i].has_buf = has_buf_host[k];
            blk += (int)((numel_host[k] + 1023) / 1024);
        }
        t.nblocks = blk;
        hipLaunchKernelGGL(sqnorm_partial_kernel, dim3(blk), dim3(256), 0, s, t, partial + total_blocks);
        BRCNN_LAUNCH_CHECK();
        total_blocks += blk;
    }
    hipLaunchKernelGGL(sqnorm_final_kernel, dim3(1), dim3(1024), 0, s, (const float*)partial, total_blocks, inv_scale,
                       max_norm, skip_nonfinite, ctl3);
    BRCNN_LAUNCH_CHECK();
    for (int first = 0; first < num_tensors; first += TAB) {
        SgdTable t;
        t.count = num_tensors - first < TAB ? num_tensors - first : TAB;
        int blk = 0;
        for (int i = 0; i < t.count; i++) {
            const int k = first + i;
            t.e[i].w = params[k]; t.e[i].g = grads[k]; t.e[i].buf = bufs ? bufs[k] : nullptr;
            t.e[i].blk0 = blk; t.e[i].n = (int)numel_host[k]; t.e[i].lr = lr_host[k]; t.e[i].wd = wd_host[k];
            t.e[i].has_buf = has_buf_host[k];
            blk += (int)((numel_host[k] + 1023) / 1024);
        }
        t.nblocks = blk;
        hipLaunchKernelGGL(sgd_update_kernel, dim3(blk), dim3(256), 0, s, t, (const float*)ctl3, momentum);
        BRCNN_LAUNCH_CHECK();
    }
    return 0;
}

// forward / data-gradient operands of `num` conv weights in one launch per TAB tensors.  dims_host: 4 ints per tensor
// (cout, cin, kh, kw); dgrad[i] may be NULL.  ctl3: the optimizer's control block (packing is skipped with the step) or NULL.
// channels_last_host (num ints, or NULL = all 0): tensor i is stored with torch.channels_last strides.
BRCNN_API int brcnn_pack_conv_weights_batch(const float* const* weights, void* const* fwd, void* const* dgrad,
                                            const int* dims_host, const int* channels_last_host, int num, int dtype,
                                            const float* ctl3, void* stream) {
    if (num <= 0) return 0;
    if (!weights || !fwd || !dgrad || !dims_host || !brcnn_elem_ok(dtype)) return BRCNN_EINVAL;
    hipStream_t s = (hipStream_t)stream;
    for (int first = 0; first < num; first += TAB) {
        PackTable t;
        t.count = num - first < TAB ? num - first : TAB;
        long long blk = 0;
        for (int i = 0; i < t.count; i++) {
            const int k = first + i;
            const int* d = dims_host + 4 * k;
            if (!weights[k] || (!fwd[k] && !dgrad[k]) || d[0] <= 0 || d[1] <= 0 || d[2] <= 0 || d[3] <= 0) return BRCNN_EINVAL;
            t.e[i].w = weights[k]; t.e[i].fwd = fwd[k]; t.e[i].dgrad = dgrad[k];
            t.e[i].src_cl = channels_last_host ? (channels_last_host[k] != 0) : 0;
            t.e[i].cout = d[0]; t.e[i].cin = d[1]; t.e[i].kh = d[2]; t.e[i].kw = d[3];
            t.e[i].blk0 = (int)blk;
            blk += (long long)((d[0] + 31) / 32) * ((d[1] + 31) / 32) * d[2] * d[3];
            if (blk > 0x7fffffffLL) return BRCNN_EINVAL;
        }
        t.nblocks = (int)blk;
        if (dtype == BRCNN_DT_F32)
            hipLaunchKernelGGL(pack_batch_kernel<float>, dim3((unsigned)blk), dim3(256), 0, s, t, ctl3);
        else if (dtype == BRCNN_DT_BF16)
            hipLaunchKernelGGL(pack_batch_kernel<bf16_t>, dim3((unsigned)blk), dim3(256), 0, s, t, ctl3);
        else
            hipLaunchKernelGGL(pack_batch_kernel<f16_t>, dim3((unsigned)blk), dim3(256), 0, s, t, ctl3);
        BRCNN_LAUNCH_CHECK();
    }
    return 0;
}

BRCNN_API int brcnn_pack_fc_weight_permuted(const float* weight, void* fwd, void* dgrad, int out_features, int channels,
                                            int positions, int dtype, const float* ctl3, void* stream) {
    if (!weight || !fwd || out_features <= 0 || channels <= 0 || (channels & 63) || positions <= 0 || positions > 255 ||
        !brcnn_elem_ok(dtype) || (long long)out_features * (channels / 64) > 0x7fffffffLL)
        return BRCNN_EINVAL;
    hipStream_t s = (hipStream_t)stream;
    const unsigned blocks = (unsigned)out_features * (unsigned)(channels / 64);
    const size_t lds = (size_t)64 * (positions + 1) * sizeof(float);
    const int k = channels * positions;
    const unsigned tb = (unsigned)(((out_features + 63) / 64) * ((k + 63) / 64));
    if (dtype == BRCNN_DT_F32) {
        hipLaunchKernelGGL(fc_perm_kernel<float>, dim3(blocks), dim3(256), lds, s, weight, (float*)fwd, out_features, channels, positions, ctl3);
        if (dgrad) hipLaunchKernelGGL(transpose16_kernel<float>, dim3(tb), dim3(256), 0, s, (const float*)fwd, (float*)dgrad, out_features, k, ctl3);
    } else if (dtype == BRCNN_DT_BF16) {
        hipLaunchKernelGGL(fc_perm_kernel<bf16_t>, dim3(blocks), dim3(256), lds, s, weight, (bf16_t*)fwd, out_features, channels, positions, ctl3);
        if (dgrad) hipLaunchKernelGGL(transpose16_kernel<bf16_t>, dim3(tb), dim3(256), 0, s, (const bf16_t*)fwd, (bf16_t*)dgrad, out_features, k, ctl3);
    } else {
        hipLaunchKernelGGL(fc_perm_kernel<f16_t>, dim3(blocks), dim3(256), lds, s, weight, (f16_t*)fwd, out_features, channels, positions, ctl3);
        if (dgrad) hipLaunchKernelGGL(transpose16_kernel<f16_t>, dim3(tb), dim3(256), 0, s, (const f16_t*)fwd, (f16_t*)dgrad, out_features, k, ctl3);
    }
    BRCNN_LAUNCH_CHECK();
    return 0;
}
