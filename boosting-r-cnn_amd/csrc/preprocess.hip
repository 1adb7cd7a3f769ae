// Input front door: Resize -> RandomFlip -> Normalize -> Pad of one decoded
// uint8 BGR image (mmdet/datasets/pipelines/transforms.py Resize:31 / RandomFlip:318 /
// Normalize:700 / Pad:625 over mmcv.imresize = cv2.resize INTER_LINEAR, mmcv.imflip,
// mmcv.imnormalize, mmcv.impad) fused into one HBM-bound pass:
//   read  src  (H, W, 3) u8 HWC      (4 taps x 3 B per output pixel, L1/L2-resident rows)
//   write dst  (3, PH, PW) fp32 CHW  (12 B per pixel, coalesced per plane)
// The bilinear arithmetic is OpenCV's 8-bit path: sample position fx = (float)((dx + 0.5) *
// scale - 0.5) with scale = 1 / (dst / src) in double, 11-bit coefficients cvRound(w * 2048),
// horizontal sums in int, vertical combine (((b0 * (S0 >> 4)) >> 16) + ((b1 * (S1 >> 4)) >> 16)
// + 2) >> 2; then (float(v) - mean[c]) * fp32(1 / std[c]) on the RGB (or BGR) channel order.
#include "common.h"

namespace {

struct PreParams {
    int sh, sw, nh, nw, ph, pw, flip, to_rgb;
    double scale_x, scale_y;
    float mean[3], stdinv[3];
};

__device__ __forceinline__ void axis_coeff(int d, double scale, int src, int& s, int& c0, int& c1) {
    float f = (float)(((double)d + 0.5) * scale - 0.5);
    s = (int)floorf(f);
    f -= (float)s;
    if (s < 0) { f = 0.f; s = 0; }
    if (s >= src - 1) { f = 0.f; s = src - 1; }
    c0 = (int)rintf((1.f - f) * 2048.f);
    c1 = (int)rintf(f * 2048.f);
}

__global__ __launch_bounds__(256) void preprocess_u8_kernel(const uint8_t* __restrict__ src,
                                                           float* __restrict__ dst, PreParams p) {
    const int x = blockIdx.x * 64 + (threadIdx.x & 63);
    const int y = blockIdx.y * 4 + (threadIdx.x >> 6);
    if (x >= p.pw || y >= p.ph) return;
    const size_t plane = (size_t)p.ph * p.pw;
    float* o = dst + (size_t)y * p.pw + x;
    if (x >= p.nw || y >= p.nh) {
        o[0] = 0.f; o[plane] = 0.f; o[2 * plane] = 0.f;
        return;
    }
    const int rx = (p.flip & 1) ? p.nw - 1 - x : x;       // pixel of the resized image shown here
    const int ry = (p.flip & 2) ? p.nh - 1 - y : y;
    int sx, a0, a1, sy, b0, b1;
    axis_coeff(rx, p.scale_x, p.sw, sx, a0, a1);
    axis_coeff(ry, p.scale_y, p.sh, sy, b0, b1);
    const int sx1 = min(sx + 1, p.sw - 1);
    const int sy0 = min(max(sy, 0), p.sh - 1), sy1 = min(max(sy + 1, 0), p.sh - 1);
    const uint8_t* r0 = src + (size_t)sy0 * p.sw * 3;
    const uint8_t* r1 = src + (size_t)sy1 * p.sw * 3;
#pragma unroll
    for (int c = 0; c < 3; c++) {
        const int sc = p.to_rgb ? 2 - c : c;
        const int h0 = (int)r0[sx * 3 + sc] * a0 + (int)r0[sx1 * 3 + sc] * a1;
        const int h1 = (int)r1[sx * 3 + sc] * a0 + (int)r1[sx1 * 3 + sc] * a1;
        int v = (((b0 * (h0 >> 4)) >> 16) + ((b1 * (h1 >> 4)) >> 16) + 2) >> 2;
        v = min(max(v, 0), 255);
        o[c * plane] = ((float)v - p.mean[c]) * p.stdinv[c];
    }
}

}  // namespace

BRCNN_API int brcnn_preprocess_u8(const uint8_t* src, int src_h, int src_w, float* dst, int new_h,
                                  int new_w, int pad_h, int pad_w, int flip,
                                  const float* mean3_host, const float* std3_host, int to_rgb,
                                  void* stream) {
    if (!src || !dst || src_h <= 0 || src_w <= 0 || new_h <= 0 || new_w <= 0 || pad_h < new_h ||
        pad_w < new_w || flip < 0 || flip > 3 || !mean3_host || !std3_host)
        return BRCNN_EINVAL;
    PreParams p;
    p.sh = src_h; p.sw = src_w; p.nh = new_h; p.nw = new_w; p.ph = pad_h; p.pw = pad_w;
    p.flip = flip; p.to_rgb = to_rgb ? 1 : 0;
    p.scale_x = 1.0 / ((double)new_w / (double)src_w);
    p.scale_y = 1.0 / ((double)new_h / (double)src_h);
    for (int c = 0; c < 3; c++) {
        if (!(std3_host[c] != 0.f)) return BRCNN_EINVAL;
        p.mean[c] = mean3_host[c];
        p.stdinv[c] = (float)(1.0 / (double)std3_host[c]);
    }
    hipLaunchKernelGGL(preprocess_u8_kernel, dim3((pad_w + 63) / 64, (pad_h + 3) / 4), dim3(256), 0,
                       (hipStream_t)stream, src, dst, p);
    BRCNN_LAUNCH_CHECK();
    return 0;
}
