// Weight gradient of the NHWC convolution / linear layer on the gfx950 matrix cores.
//
//   dW[co, k] = sum_m dY[m, co] * A[m, k]        k = (kh, kw, ci), A = on-the-fly im2col of x
//
// (autograd of every trainable conv / FC of the hot path: resnet.py stages 2-4, necks/pafpn.py,
// atss_rpn_head.py tower + heads, convfc_bbox_head.py FCs; the reference gets it from
// cuDNN/cuBLAS through torch autograd.)
//
// GEMM with the reduction over the M = N*Ho*Wo output pixels (up to 537 600) and a small
// (Cout x K) result, so the reduction is split: grid = (co tiles x k tiles) x S slices of M,
// every workgroup reduces its slice into a 128x128 fp32 tile on v_mfma_f32_32x32x2_f32 and
// adds it to dW with fp32 atomics (dW zero-filled by the caller).  Both operands are
// "reduction-major" in memory ((M,Cout) and (M,K) rows), which is exactly the MFMA operand
// order lane -> column: the LDS tiles are stored [m][128 columns] unpadded (ds_write_b128 of
// coalesced rows, conflict-free ds_read_b32 of 32 consecutive columns), one read per operand
// per MFMA.
#include "common.h"

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
constexpr int OOB = 0x7fffffff;
constexpr int TM = 32;          // reduction (m) rows per LDS tile
constexpr int TC = 128;         // columns per operand tile

struct WgradParams {
    const float* dy;    // (M, Cout)
    const float* x;     // (N, H, W, Cin) segments back to back
    float* dw;          // (Cout, K) fp32, accumulated atomically
    int Cin, Cout, KH, KW, stride, pad, M, K;
    int tiles_co, tiles_k, slices, rows_per_slice;
    unsigned dy_bytes, x_bytes;
    int nseg;
    int seg_m0[BRCNN_MAX_LEVELS + 1];
    int seg_H[BRCNN_MAX_LEVELS], seg_W[BRCNN_MAX_LEVELS], seg_Ho[BRCNN_MAX_LEVELS], seg_Wo[BRCNN_MAX_LEVELS];
    long long seg_xoff[BRCNN_MAX_LEVELS];
};

__global__ __launch_bounds__(256, 2) void conv_wgrad_f32_kernel(WgradParams p) {
    __shared__ __attribute__((aligned(16))) float Ya[2][TM][TC];   // dY tile  [m][co]
    __shared__ __attribute__((aligned(16))) float Xa[2][TM][TC];   // im2col tile [m][k]

    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int li = lane & 31, lh = lane >> 5;

    int b = blockIdx.x;
    const int slice = b % p.slices; b /= p.slices;
    const int tk = b % p.tiles_k, tco = b / p.tiles_k;
    const int co0 = tco * TC, k0 = tk * TC;
    const int m_begin = slice * p.rows_per_slice;
    const int m_end = min(p.M, m_begin + p.rows_per_slice);
    if (m_begin >= m_end) return;

    const __amdgpu_buffer_rsrc_t rsrc_y = __builtin_amdgcn_make_buffer_rsrc((void*)p.dy, 0, (int)p.dy_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rsrc_x = __builtin_amdgcn_make_buffer_rsrc((void*)p.x, 0, (int)p.x_bytes, 0x00020000);

    // staging: thread owns float4 column c4 (0..31) of rows r0 + 8*j (j = 0..3) of both tiles
    const int c4 = tid & 31;
    const int r0 = tid >> 5;
    const int co = co0 + c4 * 4;
    const bool co_ok = co < p.Cout;                 // Cout % 4 == 0 enforced by the host
    const int k = k0 + c4 * 4;
    const bool k_ok = k < p.K;                      // K % 4 == 0 (Cin % 4 == 0)
    const int tap = k_ok ? k / p.Cin : 0;
    const int ci = k - tap * p.Cin;
    const int kh = tap / p.KW, kw = tap - kh * p.KW;

    f32x16 acc[2][2];
#pragma unroll
    for (int a = 0; a < 2; a++)
#pragma unroll
        for (int c = 0; c < 2; c++)
#pragma unroll
            for (int r = 0; r < 16; r++) acc[a][c][r] = 0.f;

    float4 ry[4], rx[4];
    auto load_tile = [&](int mt) {
#pragma unroll
        for (int j = 0; j < 4; j++) {
            const int m = mt + r0 + 8 * j;
            int offy = OOB, offx = OOB;
            if (m < m_end) {
                if (co_ok) offy = (m * p.Cout + co) * 4;
                if (k_ok) {
                    int sg = 0;
#pragma unroll
                    for (int t = 1; t < BRCNN_MAX_LEVELS; t++)
                        if (t < p.nseg && m >= p.seg_m0[t]) sg = t;
                    const int ml = m - p.seg_m0[sg];
                    const int Ho = p.seg_Ho[sg], Wo = p.seg_Wo[sg], H = p.seg_H[sg], W = p.seg_W[sg];
                    const int n = ml / (Ho * Wo);
                    const int rem = ml - n * (Ho * Wo);
                    const int ho = rem / Wo, wo = rem - ho * Wo;
                    const int hi = ho * p.stride - p.pad + kh, wi = wo * p.stride - p.pad + kw;
                    if ((unsigned)hi < (unsigned)H && (unsigned)wi < (unsigned)W)
                        offx = ((int)p.seg_xoff[sg] + ((n * H + hi) * W + wi) * p.Cin + ci) * 4;
                }
            }
            const u32x4 vy = __builtin_amdgcn_raw_buffer_load_b128(rsrc_y, offy, 0, 0);
            const u32x4 vx = __builtin_amdgcn_raw_buffer_load_b128(rsrc_x, offx, 0, 0);
            ry[j] = make_float4(__uint_as_float(vy.x), __uint_as_float(vy.y), __uint_as_float(vy.z), __uint_as_float(vy.w));
            rx[j] = make_float4(__uint_as_float(vx.x), __uint_as_float(vx.y), __uint_as_float(vx.z), __uint_as_float(vx.w));
        }
    };
    auto store_tile = [&](int buf) {
#pragma unroll
        for (int j = 0; j < 4; j++) {
            *reinterpret_cast<float4*>(&Ya[buf][r0 + 8 * j][c4 * 4]) = ry[j];
            *reinterpret_cast<float4*>(&Xa[buf][r0 + 8 * j][c4 * 4]) = rx[j];
        }
    };
    // one MFMA step consumes reduction rows (2s, 2s+1): lane half lh takes row 2s + lh
    auto mfma_steps = [&](int buf, int s0, int s1) {
#pragma unroll
        for (int s = s0; s < s1; s++) {
            const int row = 2 * s + lh;
            const float a0 = Ya[buf][row][wm * 64 + li], a1 = Ya[buf][row][wm * 64 + 32 + li];
            const float b0 = Xa[buf][row][wn * 64 + li], b1 = Xa[buf][row][wn * 64 + 32 + li];
            acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b0, acc[0][0], 0, 0, 0);
            acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b1, acc[0][1], 0, 0, 0);
            acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b0, acc[1][0], 0, 0, 0);
            acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b1, acc[1][1], 0, 0, 0);
        }
    };

    load_tile(m_begin);
    store_tile(0);
    __syncthreads();
    int cur = 0;
    for (int mt = m_begin; mt < m_end; mt += TM) {
        const bool more = mt + TM < m_end;
        if (more) load_tile(mt + TM);
        mfma_steps(cur, 0, 8);
        if (more) store_tile(cur ^ 1);
        mfma_steps(cur, 8, 16);
        __syncthreads();
        cur ^= 1;
    }

    // D[row = co][col = k]:  col = lane&31, row = (r&3) + 8*(r>>2) + 4*(lane>>5)
#pragma unroll
    for (int tn = 0; tn < 2; tn++) {
        const int kk = k0 + wn * 64 + tn * 32 + li;
        if (kk >= p.K) continue;
#pragma unroll
        for (int tm = 0; tm < 2; tm++) {
#pragma unroll
            for (int r = 0; r < 16; r++) {
                const int c = co0 + wm * 64 + tm * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
                if (c < p.Cout) atomicAdd(p.dw + (size_t)c * p.K + kk, acc[tm][tn][r]);
            }
        }
    }
}

}  // namespace

BRCNN_API int brcnn_conv2d_wgrad_nhwc_multi(const void* x, const void* dy, void* dw, int batch,
                                            int num_segments, const int* heights_host,
                                            const int* widths_host, int cin, int cout, int kh,
                                            int kw, int stride, int pad, int dtype, void* stream) {
    if (!x || !dy || !dw || batch <= 0 || cin <= 0 || cout <= 0 || kh <= 0 || kw <= 0 ||
        stride <= 0 || pad < 0 || num_segments <= 0 || num_segments > BRCNN_MAX_LEVELS ||
        !heights_host || !widths_host || dtype != BRCNN_DT_F32 || (cin & 3) || (cout & 3))
        return BRCNN_EINVAL;
    WgradParams p = {};
    p.dy = (const float*)dy; p.x = (const float*)x; p.dw = (float*)dw;
    p.Cin = cin; p.Cout = cout; p.KH = kh; p.KW = kw; p.stride = stride; p.pad = pad;
    p.nseg = num_segments;
    long long m_total = 0, x_off = 0;
    for (int s = 0; s < num_segments; s++) {
        const int H = heights_host[s], W = widths_host[s];
        const int Ho = (H + 2 * pad - kh) / stride + 1, Wo = (W + 2 * pad - kw) / stride + 1;
        if (H <= 0 || W <= 0 || Ho <= 0 || Wo <= 0) return BRCNN_EINVAL;
        p.seg_H[s] = H; p.seg_W[s] = W; p.seg_Ho[s] = Ho; p.seg_Wo[s] = Wo;
        p.seg_m0[s] = (int)m_total;
        p.seg_xoff[s] = x_off;
        m_total += (long long)batch * Ho * Wo;
        x_off += (long long)batch * H * W * cin;
    }
    for (int s = num_segments; s <= BRCNN_MAX_LEVELS; s++) p.seg_m0[s] = (int)m_total;
    if (m_total * cout * 4 >= 0x7fffffffLL || x_off * 4 >= 0x7fffffffLL) return BRCNN_EINVAL;
    p.M = (int)m_total;
    p.K = kh * kw * cin;
    p.dy_bytes = (unsigned)(m_total * cout * 4);
    p.x_bytes = (unsigned)(x_off * 4);
    p.tiles_co = (cout + TC - 1) / TC;
    p.tiles_k = (p.K + TC - 1) / TC;
    // enough slices of M to fill the chip ~4x over, each at least 256 rows deep
    const int tiles = p.tiles_co * p.tiles_k;
    int slices = (2048 + tiles - 1) / tiles;
    const int max_slices = (p.M + 255) / 256;
    if (slices > max_slices) slices = max_slices;
    if (slices < 1) slices = 1;
    int rps = (p.M + slices - 1) / slices;
    rps = (rps + TM - 1) / TM * TM;
    p.slices = (p.M + rps - 1) / rps;
    p.rows_per_slice = rps;
    hipLaunchKernelGGL(conv_wgrad_f32_kernel, dim3(tiles * p.slices), dim3(256), 0,
                       (hipStream_t)stream, p);
    BRCNN_LAUNCH_CHECK();
    return 0;
}
