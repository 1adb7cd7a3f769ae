// Res2Net-DCN rows of the hot path (configs/boosting_rcnn/boosting_rcnn_r2_101_*: Res2Net-101
// with DCNv2 in stages 2-4):
//   * average pooling of NHWC maps (res2net.py:52-54 AvgPool2d(3, stride, padding=1) on the last
//     split of a stage-opening block; res2net.py:173-178 AvgPool2d(stride, stride, ceil_mode=True,
//     count_include_pad=False) in the avg_down shortcut; resnet.py deep-stem variants),
//   * the modulated deformable im2col of mmcv's ModulatedDeformConv2dPack (mmcv 1.4.0
//     ops/csrc/common/cuda/modulated_deform_conv_cuda_kernel.cuh: dmcn_im2col_bilinear and
//     modulated_deformable_im2col_gpu_kernel), deform_groups = 1: column row m = (n, ho, wo),
//     column index (tap, c) -- the K order of this repo's packed weights -- so the GEMM that
//     follows is the ordinary MFMA linear kernel with its fused BN / ReLU epilogue.
// Both are HBM-bound streams, 16 bytes of channels per lane.
#include "common.h"

namespace {

__global__ __launch_bounds__(256) void avgpool_nhwc_kernel(const float* __restrict__ x, float* __restrict__ y,
                                                          int N, int H, int W, int C, int Ho, int Wo, int k,
                                                          int stride, int pad, int count_include_pad) {
    const int c4n = C >> 2;
    const long long total = (long long)N * Ho * Wo * c4n;
    for (long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x; idx < total;
         idx += (long long)gridDim.x * blockDim.x) {
        const int c4 = (int)(idx % c4n);
        long long r = idx / c4n;
        const int wo = (int)(r % Wo); r /= Wo;
        const int ho = (int)(r % Ho);
        const int n = (int)(r / Ho);
        // torch AvgPool2d window: [hs, he) clipped to the padded extent, divisor per count_include_pad
        int hs = ho * stride - pad, ws = wo * stride - pad;
        int he = min(hs + k, H + pad), we = min(ws + k, W + pad);
        const int pool_size = (he - hs) * (we - ws);
        hs = max(hs, 0); ws = max(ws, 0);
        he = min(he, H); we = min(we, W);
        float4 a = make_float4(0.f, 0.f, 0.f, 0.f);
        for (int h = hs; h < he; h++)
            for (int w = ws; w < we; w++) {
                const float4 v = *reinterpret_cast<const float4*>(x + (((size_t)n * H + h) * W + w) * C + c4 * 4);
                a.x += v.x; a.y += v.y; a.z += v.z; a.w += v.w;
            }
        const float div = (float)(count_include_pad ? pool_size : (he - hs) * (we - ws));
        a.x /= div; a.y /= div; a.z /= div; a.w /= div;
        *reinterpret_cast<float4*>(y + (size_t)idx * 4) = a;
    }
}

__device__ __forceinline__ float4 ld4z(const float* x, int H, int W, int C, int n, int h, int w, int c, bool ok) {
    if (!ok) return make_float4(0.f, 0.f, 0.f, 0.f);
    return *reinterpret_cast<const float4*>(x + (((size_t)n * H + h) * W + w) * C + c);
}

// one thread: one (output pixel, tap, 4 channels)
__global__ __launch_bounds__(256) void deform_im2col_nhwc_kernel(const float* __restrict__ x,
                                                                const float* __restrict__ om,
                                                                float* __restrict__ col, int N, int H, int W,
                                                                int C, int Ho, int Wo, int KH, int KW, int stride,
                                                                int pad, int dilation, int om_stride, int Cpad) {
    const int c4n = Cpad >> 2, taps = KH * KW;
    const long long total = (long long)N * Ho * Wo * taps * c4n;
    for (long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x; idx < total;
         idx += (long long)gridDim.x * blockDim.x) {
        const int c = (int)(idx % c4n) * 4;
        long long r = idx / c4n;
        const int tap = (int)(r % taps); r /= taps;
        const long long m = r;
        const int wo = (int)(m % Wo);
        const int ho = (int)((m / Wo) % Ho);
        const int n = (int)(m / ((long long)Wo * Ho));
        float4 out = make_float4(0.f, 0.f, 0.f, 0.f);
        if (c < C) {
            const int i = tap / KW, j = tap - i * KW;
            const float* o = om + (size_t)m * om_stride;
            const float off_h = o[2 * tap], off_w = o[2 * tap + 1];
            const float mask = 1.f / (1.f + expf(-o[2 * taps + tap]));
            const float h_im = (float)(ho * stride - pad + i * dilation) + off_h;
            const float w_im = (float)(wo * stride - pad + j * dilation) + off_w;
            if (h_im > -1.f && w_im > -1.f && h_im < (float)H && w_im < (float)W) {
                const int h_low = (int)floorf(h_im), w_low = (int)floorf(w_im);
                const int h_high = h_low + 1, w_high = w_low + 1;
                const float lh = h_im - (float)h_low, lw = w_im - (float)w_low;
                const float hh = 1.f - lh, hw = 1.f - lw;
                const float4 v1 = ld4z(x, H, W, C, n, h_low, w_low, c, h_low >= 0 && w_low >= 0);
                const float4 v2 = ld4z(x, H, W, C, n, h_low, w_high, c, h_low >= 0 && w_high <= W - 1);
                const float4 v3 = ld4z(x, H, W, C, n, h_high, w_low, c, h_high <= H - 1 && w_low >= 0);
                const float4 v4 = ld4z(x, H, W, C, n, h_high, w_high, c, h_high <= H - 1 && w_high <= W - 1);
                const float w1 = hh * hw, w2 = hh * lw, w3 = lh * hw, w4 = lh * lw;
                out.x = (w1 * v1.x + w2 * v2.x + w3 * v3.x + w4 * v4.x) * mask;
                out.y = (w1 * v1.y + w2 * v2.y + w3 * v3.y + w4 * v4.y) * mask;
                out.z = (w1 * v1.z + w2 * v2.z + w3 * v3.z + w4 * v4.z) * mask;
                out.w = (w1 * v1.w + w2 * v2.w + w3 * v3.w + w4 * v4.w) * mask;
            }
        }
        *reinterpret_cast<float4*>(col + (size_t)idx * 4) = out;
    }
}

// Backward of the modulated deformable im2col (mmcv modulated_deformable_col2im_gpu_kernel and
// modulated_deformable_col2im_coord_gpu_kernel): one wavefront per (output pixel, tap), lanes over
// channels.  dx gets dcol * mask * corner weight (fp32 atomics, like the RoIAlign backward); the
// offset / mask gradients are channel sums, reduced across the wave and written once:
//   d off_h = mask * sum_c dcol_c * d val_c / d h,   d off_w likewise,
//   d mask_logit = mask (1 - mask) * sum_c dcol_c * val_c          (sigmoid applied in the forward)
__global__ __launch_bounds__(256) void deform_col2im_nhwc_kernel(const float* __restrict__ x,
                                                                const float* __restrict__ om,
                                                                const float* __restrict__ dcol,
                                                                float* __restrict__ dx, float* __restrict__ dom,
                                                                int N, int H, int W, int C, int Ho, int Wo, int KH,
                                                                int KW, int stride, int pad, int dilation,
                                                                int om_stride, int Cpad) {
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int taps = KH * KW;
    const long long item = (long long)blockIdx.x * 4 + wave;
    if (item >= (long long)N * Ho * Wo * taps) return;
    const int tap = (int)(item % taps);
    const long long m = item / taps;
    const int wo = (int)(m % Wo);
    const int ho = (int)((m / Wo) % Ho);
    const int n = (int)(m / ((long long)Wo * Ho));
    const int i = tap / KW, j = tap - i * KW;
    const float* o = om + (size_t)m * om_stride;
    const float off_h = o[2 * tap], off_w = o[2 * tap + 1];
    const float mask = 1.f / (1.f + expf(-o[2 * taps + tap]));
    const float h_im = (float)(ho * stride - pad + i * dilation) + off_h;
    const float w_im = (float)(wo * stride - pad + j * dilation) + off_w;
    const bool inside = h_im > -1.f && w_im > -1.f && h_im < (float)H && w_im < (float)W;
    float g_h = 0.f, g_w = 0.f, g_m = 0.f;
    if (inside) {
        const int h_low = (int)floorf(h_im), w_low = (int)floorf(w_im);
        const int h_high = h_low + 1, w_high = w_low + 1;
        const float lh = h_im - (float)h_low, lw = w_im - (float)w_low;
        const float hh = 1.f - lh, hw = 1.f - lw;
        const bool ok1 = h_low >= 0 && w_low >= 0, ok2 = h_low >= 0 && w_high <= W - 1;
        const bool ok3 = h_high <= H - 1 && w_low >= 0, ok4 = h_high <= H - 1 && w_high <= W - 1;
        const float* dc = dcol + ((size_t)m * taps + tap) * Cpad;
        for (int c = lane * 4; c < C; c += 256) {
            const float4 g = *reinterpret_cast<const float4*>(dc + c);
            const float4 v1 = ld4z(x, H, W, C, n, h_low, w_low, c, ok1);
            const float4 v2 = ld4z(x, H, W, C, n, h_low, w_high, c, ok2);
            const float4 v3 = ld4z(x, H, W, C, n, h_high, w_low, c, ok3);
            const float4 v4 = ld4z(x, H, W, C, n, h_high, w_high, c, ok4);
            const float gg[4] = {g.x, g.y, g.z, g.w};
            const float a1[4] = {v1.x, v1.y, v1.z, v1.w}, a2[4] = {v2.x, v2.y, v2.z, v2.w};
            const float a3[4] = {v3.x, v3.y, v3.z, v3.w}, a4[4] = {v4.x, v4.y, v4.z, v4.w};
#pragma unroll
            for (int e = 0; e < 4; e++) {
                const float val = hh * hw * a1[e] + hh * lw * a2[e] + lh * hw * a3[e] + lh * lw * a4[e];
                g_m += gg[e] * val;
                g_h += gg[e] * (-hw * a1[e] - lw * a2[e] + hw * a3[e] + lw * a4[e]);
                g_w += gg[e] * (-hh * a1[e] + hh * a2[e] - lh * a3[e] + lh * a4[e]);
                const float gx = gg[e] * mask;
                if (ok1) atomicAdd(dx + (((size_t)n * H + h_low) * W + w_low) * C + c + e, gx * hh * hw);
                if (ok2) atomicAdd(dx + (((size_t)n * H + h_low) * W + w_high) * C + c + e, gx * hh * lw);
                if (ok3) atomicAdd(dx + (((size_t)n * H + h_high) * W + w_low) * C + c + e, gx * lh * hw);
                if (ok4) atomicAdd(dx + (((size_t)n * H + h_high) * W + w_high) * C + c + e, gx * lh * lw);
            }
        }
    }
#pragma unroll
    for (int d = 32; d > 0; d >>= 1) {
        g_h += __shfl_xor(g_h, d);
        g_w += __shfl_xor(g_w, d);
        g_m += __shfl_xor(g_m, d);
    }
    if (lane == 0) {
        float* go = dom + (size_t)m * om_stride;
        go[2 * tap] = g_h * mask;
        go[2 * tap + 1] = g_w * mask;
        go[2 * taps + tap] = g_m * mask * (1.f - mask);
    }
}

inline int stream_grid(long long total) {
    long long g = (total + 255) / 256;
    return (int)(g > 32768 ? 32768 : (g < 1 ? 1 : g));
}

}  // namespace

BRCNN_API int brcnn_avgpool_nhwc(const float* x, float* y, int batch, int height, int width, int channels,
                                 int kernel, int stride, int pad, int ceil_mode, int count_include_pad,
                                 void* stream) {
    if (!x || !y || batch <= 0 || height <= 0 || width <= 0 || channels <= 0 || (channels & 3) || kernel <= 0 ||
        stride <= 0 || pad < 0 || pad > kernel / 2)
        return BRCNN_EINVAL;
    auto osz = [&](int in) {
        int o = ceil_mode ? (in + 2 * pad - kernel + stride - 1) / stride + 1 : (in + 2 * pad - kernel) / stride + 1;
        if (ceil_mode && (o - 1) * stride >= in + pad) o--;      // last window must start inside the input
        return o;
    };
    const int Ho = osz(height), Wo = osz(width);
    if (Ho <= 0 || Wo <= 0) return BRCNN_EINVAL;
    const long long total = (long long)batch * Ho * Wo * (channels >> 2);
    hipLaunchKernelGGL(avgpool_nhwc_kernel, dim3(stream_grid(total)), dim3(256), 0, (hipStream_t)stream, x, y, batch,
                       height, width, channels, Ho, Wo, kernel, stride, pad, count_include_pad);
    BRCNN_LAUNCH_CHECK();
    return 0;
}

BRCNN_API int brcnn_deform_im2col_nhwc(const float* x, const float* offset_mask, float* col, int batch,
                                       int height, int width, int channels, int kh, int kw, int stride, int pad,
                                       int dilation, int om_stride, int channels_padded, void* stream) {
    if (!x || !offset_mask || !col || batch <= 0 || height <= 0 || width <= 0 || channels <= 0 || (channels & 3) ||
        kh <= 0 || kw <= 0 || stride <= 0 || pad < 0 || dilation <= 0 || om_stride < 3 * kh * kw ||
        channels_padded < channels || (channels_padded & 3))
        return BRCNN_EINVAL;
    const int Ho = (height + 2 * pad - (dilation * (kh - 1) + 1)) / stride + 1;
    const int Wo = (width + 2 * pad - (dilation * (kw - 1) + 1)) / stride + 1;
    if (Ho <= 0 || Wo <= 0) return BRCNN_EINVAL;
    const long long total = (long long)batch * Ho * Wo * kh * kw * (channels_padded >> 2);
    hipLaunchKernelGGL(deform_im2col_nhwc_kernel, dim3(stream_grid(total)), dim3(256), 0, (hipStream_t)stream, x,
                       offset_mask, col, batch, height, width, channels, Ho, Wo, kh, kw, stride, pad, dilation,
                       om_stride, channels_padded);
    BRCNN_LAUNCH_CHECK();
    return 0;
}


BRCNN_API int brcnn_deform_col2im_nhwc(const float* x, const float* offset_mask, const float* dcol, float* dx,
                                       float* d_offset_mask, int batch, int height, int width, int channels,
                                       int kh, int kw, int stride, int pad, int dilation, int om_stride,
                                       int channels_padded, void* stream) {
    if (!x || !offset_mask || !dcol || !dx || !d_offset_mask || batch <= 0 || height <= 0 || width <= 0 ||
        channels <= 0 || (channels & 3) || kh <= 0 || kw <= 0 || stride <= 0 || pad < 0 || dilation <= 0 ||
        om_stride != 3 * kh * kw || channels_padded < channels || (channels_padded & 3))
        return BRCNN_EINVAL;
    const int Ho = (height + 2 * pad - (dilation * (kh - 1) + 1)) / stride + 1;
    const int Wo = (width + 2 * pad - (dilation * (kw - 1) + 1)) / stride + 1;
    if (Ho <= 0 || Wo <= 0) return BRCNN_EINVAL;
    const long long items = (long long)batch * Ho * Wo * kh * kw;
    hipLaunchKernelGGL(deform_col2im_nhwc_kernel, dim3((unsigned)((items + 3) / 4)), dim3(256), 0, (hipStream_t)stream,
                       x, offset_mask, dcol, dx, d_offset_mask, batch, height, width, channels, Ho, Wo, kh, kw, stride,
                       pad, dilation, om_stride, channels_padded);
    BRCNN_LAUNCH_CHECK();
    return 0;
}
