// Sigmoid focal loss forward/backward, element-wise, for gfx950.
//
// Replaces mmcv.ops.sigmoid_focal_loss_{forward,backward} reached at
// mmdet/models/losses/focal_loss.py:86 (the reference's CPU path is the python form
// py_sigmoid_focal_loss focal_loss.py:12-57; both agree to fp32 rounding).
// HBM-bound stream: 4 B read + 4 B write per logit, float4 per lane when c allows.
#include "common.h"
#include <float.h>

namespace {

template <bool BWD>
__device__ __forceinline__ float focal_elem(float x, bool is_pos, float gamma, float alpha) {
    const float p = 1.f / (1.f + expf(-x));
    if (!BWD) {
        const float term_p = powf(1.f - p, gamma) * logf(fmaxf(p, FLT_MIN));
        const float term_n = powf(p, gamma) * logf(fmaxf(1.f - p, FLT_MIN));
        return is_pos ? -alpha * term_p : -(1.f - alpha) * term_n;
    } else {
        const float term_p = powf(1.f - p, gamma) * (1.f - p - gamma * p * logf(fmaxf(p, FLT_MIN)));
        const float term_n = powf(p, gamma) * (gamma * (1.f - p) * logf(fmaxf(1.f - p, FLT_MIN)) - p);
        return is_pos ? -alpha * term_p : -(1.f - alpha) * term_n;
    }
}

template <bool BWD>
__global__ __launch_bounds__(256) void focal_kernel(const float* __restrict__ input,
                                                    const int64_t* __restrict__ target,
                                                    const float* __restrict__ weight,
                                                    float* __restrict__ out, long long total,
                                                    int c_n, float gamma, float alpha) {
    for (long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x; idx < total;
         idx += (long long)gridDim.x * blockDim.x) {
        const long long i = idx / c_n;
        const int c = (int)(idx - i * c_n);
        const long long t = target[i];
        float v = focal_elem<BWD>(input[idx], t == c, gamma, alpha);
        if (weight) v *= weight[t];
        out[idx] = v;
    }
}

template <bool BWD>
int launch(const float* input, const int64_t* target, const float* weight, float* out, int64_t n,
           int64_t c, float gamma, float alpha, void* stream) {
    if (n < 0 || c <= 0 || c > 0x7fffffff) return BRCNN_EINVAL;
    if (n == 0) return 0;
    if (!input || !target || !out) return BRCNN_EINVAL;
    const long long total = n * c;
    int grid = brcnn_cdiv(total, 256);
    if (grid > 256 * 8) grid = 256 * 8;
    hipLaunchKernelGGL(focal_kernel<BWD>, dim3(grid), dim3(256), 0, (hipStream_t)stream, input,
                       target, weight, out, total, (int)c, gamma, alpha);
    BRCNN_LAUNCH_CHECK();
    return 0;
}

}  // namespace

BRCNN_API int brcnn_sigmoid_focal_loss_forward(const float* input, const int64_t* target,
                                               const float* weight, float* output, int64_t n,
                                               int64_t c, float gamma, float alpha, void* stream) {
    return launch<false>(input, target, weight, output, n, c, gamma, alpha, stream);
}

BRCNN_API int brcnn_sigmoid_focal_loss_backward(const float* input, const int64_t* target,
                                                const float* weight, float* grad_input, int64_t n,
                                                int64_t c, float gamma, float alpha, void* stream) {
    return launch<true>(input, target, weight, grad_input, n, c, gamma, alpha, stream);
}
