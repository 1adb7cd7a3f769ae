// Implicit-GEMM convolution / linear layer on the gfx950 matrix cores, NHWC.
//
// Covers every conv and FC of the Boosting R-CNN hot path (ResNet bottlenecks
// mmdet/models/backbones/resnet.py:263-302 with eval-mode BN folded into the epilogue,
// PAFPN laterals / 3x3 / stride-2 convs necks/pafpn.py:100-158, the RetinaRPN tower and its
// three 3x3 heads dense_heads/atss_rpn_head.py:207-215, the 12544->1024->1024->(C+1 | 4C)
// FCs roi_heads/bbox_heads/convfc_bbox_head.py:154-192).
//
//   GEMM view:  D[m, co] = sum_k A[m, k] * W[co, k]
//     m  = (n, ho, wo) output pixel, k = (kh, kw, ci), A gathered on the fly (no im2col buffer),
//     W  = weights in (Cout, KH, KW, Cin) order == (Cout, K) row-major: A and W are both
//     K-contiguous, so one staging scheme serves both operands.
//
// fp32 path: v_mfma_f32_32x32x2_f32 -- an exact fp32 FMA chain (no TF32 on gfx950) at the
// fp32 vector rate (157 TF/s chip peak), so the results track the reference CPU path to fp32
// round-off.  Workgroup = 256 threads = 4 waves (2x2), tile 128(M) x 128(N) x 32(K), each
// wave 64x64 = 2x2 MFMA tiles (64 accumulator VGPRs).  Operands are staged
// global -> registers -> LDS (issue the next tile's loads before the MFMA block, write them
// after it), LDS rows padded to 36 floats so the ds_read_b128 fragment reads are
// bank-conflict free; each lane reads 4 consecutive k per b128 and the K order inside an
// 8-deep group is permuted identically for A and W (lane half h takes k = 4h..4h+3), which
// leaves every product paired correctly and needs one LDS read per 4 MFMAs per operand.
// Epilogue fused: per-channel scale/shift (folded BN or bias), residual add, ReLU.
// Block ids are remapped so that each XCD's L2 sees a contiguous run of tiles (the N tiles
// of one M tile share the gathered A rows).
#include "common.h"

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int BM = 128, BN = 128, BK = 32, LDS_STRIDE = 36;

struct ConvParams {
    const float* x;
    const float* w;
    const float* scale;
    const float* shift;
    const float* residual;
    float* y;
    int batch, H, W, Cin, Cout, KH, KW, stride, pad, Ho, Wo;
    int M, K;
    int relu;
    int tiles_m, tiles_n;
};

__device__ __forceinline__ int xcd_remap(int bid, int nwg) {
    // bijective: XCD x (= bid % 8) owns a contiguous chunk of logical tile ids
    const int q = nwg >> 3, r = nwg & 7;
    const int xcd = bid & 7, loc = bid >> 3;
    const int base = (xcd < r) ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
    return base + loc;
}

template <bool FAST>   // FAST: Cin % 32 == 0 (a K tile never straddles a filter tap)
__global__ __launch_bounds__(256, 2) void conv_igemm_f32_kernel(ConvParams p) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* As = smem;                          // [2][BM][LDS_STRIDE]
    float* Bs = smem + 2 * BM * LDS_STRIDE;    // [2][BN][LDS_STRIDE]

    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;

    const int nwg = p.tiles_m * p.tiles_n;
    const int tile = xcd_remap(blockIdx.x, nwg);
    const int tile_m = tile / p.tiles_n, tile_n = tile - tile_m * p.tiles_n;
    const int m0 = tile_m * BM, n0 = tile_n * BN;

    // ---- staging assignment: float4 column c4 of rows r0 + 32*j -------------------------
    const int c4 = tid & 7;
    const int r0 = tid >> 3;
    int a_base[4], a_hw[4];   // a_hw packs (hi0 + 4096) << 16 | (wi0 + 4096); a_base < 0: row invalid
    const float* b_ptr[4];
    bool b_ok[4];
#pragma unroll
    for (int j = 0; j < 4; j++) {
        const int m = m0 + r0 + 32 * j;
        if (m < p.M) {
            const int n = m / (p.Ho * p.Wo);
            const int rem = m - n * (p.Ho * p.Wo);
            const int ho = rem / p.Wo, wo = rem - ho * p.Wo;
            a_base[j] = n * p.H * p.W;
            a_hw[j] = ((ho * p.stride - p.pad + 4096) << 16) | (wo * p.stride - p.pad + 4096);
        } else {
            a_base[j] = -1;
            a_hw[j] = 0;
        }
        const int co = n0 + r0 + 32 * j;
        b_ok[j] = co < p.Cout;
        b_ptr[j] = p.w + (size_t)(b_ok[j] ? co : 0) * p.K;
    }

    f32x16 acc[2][2];
#pragma unroll
    for (int a = 0; a < 2; a++)
#pragma unroll
        for (int b = 0; b < 2; b++)
#pragma unroll
            for (int r = 0; r < 16; r++) acc[a][b][r] = 0.f;

    float4 ra[4], rb[4];
    const int nk = (p.K + BK - 1) / BK;

    auto load_tile = [&](int kt) {
        const int k0 = kt * BK;
        if (FAST) {
            const int tap = k0 / p.Cin;
            const int ci = k0 - tap * p.Cin + c4 * 4;
            const int kh = tap / p.KW, kw = tap - kh * p.KW;
#pragma unroll
            for (int j = 0; j < 4; j++) {
                const int hi = (a_hw[j] >> 16) - 4096 + kh;
                const int wi = (a_hw[j] & 0xffff) - 4096 + kw;
                const bool ok = a_base[j] >= 0 && hi >= 0 && hi < p.H && wi >= 0 && wi < p.W;
                if (ok)
                    ra[j] = *reinterpret_cast<const float4*>(
                        p.x + ((size_t)(a_base[j] + hi * p.W + wi)) * p.Cin + ci);
                else
                    ra[j] = make_float4(0.f, 0.f, 0.f, 0.f);
                if (b_ok[j])
                    rb[j] = *reinterpret_cast<const float4*>(b_ptr[j] + k0 + c4 * 4);
                else
                    rb[j] = make_float4(0.f, 0.f, 0.f, 0.f);
            }
        } else {
#pragma unroll
            for (int j = 0; j < 4; j++) {
                float va[4], vb[4];
#pragma unroll
                for (int e = 0; e < 4; e++) {
                    const int k = k0 + c4 * 4 + e;
                    va[e] = 0.f;
                    vb[e] = 0.f;
                    if (k < p.K) {
                        const int tap = k / p.Cin, ci = k - tap * p.Cin;
                        const int kh = tap / p.KW, kw = tap - kh * p.KW;
                        const int hi = (a_hw[j] >> 16) - 4096 + kh;
                        const int wi = (a_hw[j] & 0xffff) - 4096 + kw;
                        if (a_base[j] >= 0 && hi >= 0 && hi < p.H && wi >= 0 && wi < p.W)
                            va[e] = p.x[((size_t)(a_base[j] + hi * p.W + wi)) * p.Cin + ci];
                        if (b_ok[j]) vb[e] = b_ptr[j][k];
                    }
                }
                ra[j] = make_float4(va[0], va[1], va[2], va[3]);
                rb[j] = make_float4(vb[0], vb[1], vb[2], vb[3]);
            }
        }
    };
    auto store_tile = [&](int buf) {
        float* as = As + buf * BM * LDS_STRIDE;
        float* bs = Bs + buf * BN * LDS_STRIDE;
#pragma unroll
        for (int j = 0; j < 4; j++) {
            *reinterpret_cast<float4*>(as + (r0 + 32 * j) * LDS_STRIDE + c4 * 4) = ra[j];
            *reinterpret_cast<float4*>(bs + (r0 + 32 * j) * LDS_STRIDE + c4 * 4) = rb[j];
        }
    };

    load_tile(0);
    store_tile(0);
    __syncthreads();

    const int li = lane & 31, lh = lane >> 5;
    int cur = 0;
    for (int kt = 0; kt < nk; kt++) {
        if (kt + 1 < nk) load_tile(kt + 1);
        const float* as = As + cur * BM * LDS_STRIDE + (wm * 64 + li) * LDS_STRIDE + lh * 4;
        const float* bs = Bs + cur * BN * LDS_STRIDE + (wn * 64 + li) * LDS_STRIDE + lh * 4;
#pragma unroll
        for (int kk = 0; kk < BK / 8; kk++) {
            const float4 a0 = *reinterpret_cast<const float4*>(as + kk * 8);
            const float4 a1 = *reinterpret_cast<const float4*>(as + 32 * LDS_STRIDE + kk * 8);
            const float4 b0 = *reinterpret_cast<const float4*>(bs + kk * 8);
            const float4 b1 = *reinterpret_cast<const float4*>(bs + 32 * LDS_STRIDE + kk * 8);
            const float av0[4] = {a0.x, a0.y, a0.z, a0.w}, av1[4] = {a1.x, a1.y, a1.z, a1.w};
            const float bv0[4] = {b0.x, b0.y, b0.z, b0.w}, bv1[4] = {b1.x, b1.y, b1.z, b1.w};
#pragma unroll
            for (int e = 0; e < 4; e++) {
                acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(av0[e], bv0[e], acc[0][0], 0, 0, 0);
                acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(av0[e], bv1[e], acc[0][1], 0, 0, 0);
                acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(av1[e], bv0[e], acc[1][0], 0, 0, 0);
                acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(av1[e], bv1[e], acc[1][1], 0, 0, 0);
            }
        }
        if (kt + 1 < nk) {
            store_tile(cur ^ 1);
            __syncthreads();
            cur ^= 1;
        }
    }

    // ---- epilogue: scale/shift, residual, relu; D layout col = lane&31, row = (r&3)+8*(r>>2)+4*(lane>>5)
#pragma unroll
    for (int tn = 0; tn < 2; tn++) {
        const int co = n0 + wn * 64 + tn * 32 + li;
        if (co >= p.Cout) continue;
        const float sc = p.scale ? p.scale[co] : 1.f;
        const float sh = p.shift ? p.shift[co] : 0.f;
#pragma unroll
        for (int tm = 0; tm < 2; tm++) {
#pragma unroll
            for (int r = 0; r < 16; r++) {
                const int m = m0 + wm * 64 + tm * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
                if (m >= p.M) continue;
                float v = acc[tm][tn][r];
                if (p.scale) v = v * sc;
                v = v + sh;
                const size_t o = (size_t)m * p.Cout + co;
                if (p.residual) v += p.residual[o];
                if (p.relu) v = fmaxf(v, 0.f);
                p.y[o] = v;
            }
        }
    }
}

}  // namespace

BRCNN_API int brcnn_conv2d_nhwc(const void* x, const void* w, const float* scale, const float* shift,
                                const void* residual, void* y, int batch, int height, int width,
                                int cin, int cout, int kh, int kw, int stride, int pad, int relu,
                                int dtype, void* stream) {
    if (!x || !w || !y || batch <= 0 || height <= 0 || width <= 0 || cin <= 0 || cout <= 0 ||
        kh <= 0 || kw <= 0 || stride <= 0 || pad < 0)
        return BRCNN_EINVAL;
    if (dtype != BRCNN_DT_F32) return BRCNN_EINVAL;   // bf16 path: see conv_igemm_bf16.hip
    ConvParams p;
    p.x = (const float*)x; p.w = (const float*)w; p.scale = scale; p.shift = shift;
    p.residual = (const float*)residual; p.y = (float*)y;
    p.batch = batch; p.H = height; p.W = width; p.Cin = cin; p.Cout = cout; p.KH = kh; p.KW = kw;
    p.stride = stride; p.pad = pad;
    p.Ho = (height + 2 * pad - kh) / stride + 1;
    p.Wo = (width + 2 * pad - kw) / stride + 1;
    if (p.Ho <= 0 || p.Wo <= 0) return BRCNN_EINVAL;
    if (height + pad >= 4096 || width + pad >= 4096) return BRCNN_EINVAL;
    const long long M = (long long)batch * p.Ho * p.Wo;
    if (M > 0x7fffffffLL || (long long)batch * height * width > 0x7fffffffLL) return BRCNN_EINVAL;
    p.M = (int)M;
    p.K = kh * kw * cin;
    p.relu = relu;
    p.tiles_m = (p.M + BM - 1) / BM;
    p.tiles_n = (cout + BN - 1) / BN;
    const size_t lds = (size_t)2 * (BM + BN) * LDS_STRIDE * sizeof(float);
    const int grid = p.tiles_m * p.tiles_n;
    hipStream_t s = (hipStream_t)stream;
    if (cin % 32 == 0) {
        static bool attr_fast = false;
        if (!attr_fast) {
            BRCNN_HIP_CHECK(hipFuncSetAttribute((const void*)conv_igemm_f32_kernel<true>,
                                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
            attr_fast = true;
        }
        hipLaunchKernelGGL(conv_igemm_f32_kernel<true>, dim3(grid), dim3(256), lds, s, p);
    } else {
        static bool attr_gen = false;
        if (!attr_gen) {
            BRCNN_HIP_CHECK(hipFuncSetAttribute((const void*)conv_igemm_f32_kernel<false>,
                                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
            attr_gen = true;
        }
        hipLaunchKernelGGL(conv_igemm_f32_kernel<false>, dim3(grid), dim3(256), lds, s, p);
    }
    BRCNN_LAUNCH_CHECK();
    return 0;
}
