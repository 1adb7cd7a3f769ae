// bf16 implicit-GEMM convolution / linear layer on the gfx950 matrix cores, NHWC
// (BASELINE.json configs[2..4]: "bf16 MFMA backbone").
//
// Same decomposition as the fp32 kernel (conv_igemm.hip): D[m,co] = sum_k A[m,k] W[co,k],
// operand tiles staged by LDS-DMA (`buffer_load_dwordx4 ... lds`), unpadded 128-byte LDS rows
// with the source-side XOR swizzle (physical 16-B chunk c' of row r holds logical chunk
// c' ^ ((r>>1)&7)) -- a 128-byte row is 64 bf16, i.e. the K tile is 64 deep and one 16-B chunk
// is exactly the 8-element K slice a lane feeds to v_mfma_f32_32x32x16_bf16
// (lane l: A[i = l&31][k = 8*(l>>5) .. +7]).  fp32 accumulation; the epilogue applies the fp32
// per-channel scale/shift (folded BN / bias), the bf16 residual, ReLU, and rounds to bf16
// (round-to-nearest-even) or writes fp32 (head outputs that feed the fp32 post-processing).
// bf16 MFMA issues in 32 cycles, a K tile carries only MT*NT*4 MFMAs per wave, so the larger
// 128x128 tile (16 MFMAs per wave and barrier) is the default here; the kernel is bound by the
// L2->LDS stream and the per-tile barrier, not by the matrix pipes.
#include "conv_common.h"

namespace {
using namespace brcnn_conv;

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __attribute__((address_space(3))) void* lds_ptr_t;
constexpr int BKE = 64;     // K tile in elements (128 bytes)

__device__ __forceinline__ unsigned short f2bf(float v) {
    unsigned u = __float_as_uint(v);
    u += 0x7fffu + ((u >> 16) & 1u);
    return (unsigned short)(u >> 16);
}
__device__ __forceinline__ float bf2f(unsigned short h) { return __uint_as_float(((unsigned)h) << 16); }

template <int MT, int NT, bool RES, bool OUTF32>
__global__ __launch_bounds__(256, 2) void conv_igemm_bf16_dma_kernel(ConvParams p) {
    constexpr int BM = 64 * MT, BN = 64 * NT;
    constexpr int AG = BM / 8 / 4;      // 8-row groups of the A tile per wave
    constexpr int BG = BN / 8 / 4;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* As = smem;                   // [2][BM][32 dwords = 128 B]
    float* Bs = smem + 2 * BM * 32;     // [2][BN][32 dwords]

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1;
    const int li = lane & 31, lh = lane >> 5;

    const int nwg = p.tiles_m * p.tiles_n;
    const int tile = xcd_remap(blockIdx.x, nwg);
    const int tile_m = tile / p.tiles_n, tile_n = tile - tile_m * p.tiles_n;
    const int m0 = tile_m * BM, n0 = tile_n * BN;

    const __amdgpu_buffer_rsrc_t rsrc_x = __builtin_amdgcn_make_buffer_rsrc((void*)p.x, 0, (int)p.x_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rsrc_w = __builtin_amdgcn_make_buffer_rsrc((void*)p.w, 0, (int)p.w_bytes, 0x00020000);

    const int rg = lane >> 3, pc = lane & 7;
    int a_base[AG], a_hw[AG], a_H[AG], a_W[AG], a_lc[AG];
    int b_off[BG];
#pragma unroll
    for (int j = 0; j < AG; j++) {
        const int r = (wave * AG + j) * 8 + rg;
        a_lc[j] = (pc ^ ((r >> 1) & 7)) * 8;          // logical k offset (elements) of this lane's chunk
        const int m = m0 + r;
        if (m < p.M) {
            int sg = 0;
#pragma unroll
            for (int t = 1; t < BRCNN_MAX_LEVELS; t++)
                if (t < p.nseg && m >= p.seg_m0[t]) sg = t;
            const int ml = m - p.seg_m0[sg];
            const int Ho = p.seg_Ho[sg], Wo = p.seg_Wo[sg];
            a_H[j] = p.seg_H[sg];
            a_W[j] = p.seg_W[sg];
            const int n = ml / (Ho * Wo);
            const int rem = ml - n * (Ho * Wo);
            const int ho = rem / Wo, wo = rem - ho * Wo;
            a_base[j] = (int)p.seg_xoff[sg] + n * a_H[j] * a_W[j] * p.pitch;
            a_hw[j] = ((ho * p.stride - p.pad + 4096) << 16) | (wo * p.stride - p.pad + 4096);
        } else {
            a_base[j] = -1;
            a_hw[j] = 0;
            a_H[j] = a_W[j] = 0;
        }
    }
#pragma unroll
    for (int j = 0; j < BG; j++) {
        const int r = (wave * BG + j) * 8 + rg;
        const int co = n0 + r;
        b_off[j] = (co < p.Cout) ? co * p.K + (pc ^ ((r >> 1) & 7)) * 8 : -1;
    }

    f32x16 acc[MT][NT];
#pragma unroll
    for (int a = 0; a < MT; a++)
#pragma unroll
        for (int b = 0; b < NT; b++)
#pragma unroll
            for (int r = 0; r < 16; r++) acc[a][b][r] = 0.f;

    const int nk = p.K / BKE;

    auto dma_tile = [&](int kt, int buf) {
        const int k0 = kt * BKE;
        const int tap = k0 / p.Cin;
        const int ci0 = k0 - tap * p.Cin;
        const int kh = tap / p.KW, kw = tap - kh * p.KW;
#pragma unroll
        for (int j = 0; j < AG; j++) {
            const int hi = (a_hw[j] >> 16) - 4096 + kh;
            const int wi = (a_hw[j] & 0xffff) - 4096 + kw;
            const bool ok = (a_base[j] >= 0) & ((unsigned)hi < (unsigned)a_H[j]) &
                            ((unsigned)wi < (unsigned)a_W[j]);
            const int off = ok ? (a_base[j] + (hi * a_W[j] + wi) * p.pitch + ci0 + a_lc[j]) * 2 : OOB;
            float* dst = As + buf * BM * 32 + (wave * AG + j) * 8 * 32;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc_x, (lds_ptr_t)dst, 16, off, 0, 0, 0);
        }
#pragma unroll
        for (int j = 0; j < BG; j++) {
            const int off = (b_off[j] >= 0) ? (b_off[j] + k0) * 2 : OOB;
            float* dst = Bs + buf * BN * 32 + (wave * BG + j) * 8 * 32;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc_w, (lds_ptr_t)dst, 16, off, 0, 0, 0);
        }
    };

    dma_tile(0, 0);

    // residual prefetch (bf16), D layout: col = lane&31, row = (r&3) + 8*(r>>2) + 4*(lane>>5)
    const unsigned short* __restrict__ res = reinterpret_cast<const unsigned short*>(p.residual);
    float rv[MT][NT][16];
    if (RES) {
#pragma unroll
        for (int tn = 0; tn < NT; tn++) {
            const int co = n0 + wn * 32 * NT + tn * 32 + li;
#pragma unroll
            for (int tm = 0; tm < MT; tm++) {
                const int mb = m0 + wm * 32 * MT + tm * 32 + 4 * lh;
#pragma unroll
                for (int r = 0; r < 16; r++) {
                    const int m = mb + (r & 3) + 8 * (r >> 2);
                    rv[tm][tn][r] = (co < p.Cout && m < p.M) ? bf2f(res[(size_t)m * p.Cout + co]) : 0.f;
                }
            }
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();

    const int sw = (li >> 1) & 7;
    int cur = 0;
    for (int kt = 0; kt < nk; kt++) {
        if (kt + 1 < nk) dma_tile(kt + 1, cur ^ 1);
        const float* as = As + cur * BM * 32 + (wm * 32 * MT + li) * 32;
        const float* bs = Bs + cur * BN * 32 + (wn * 32 * NT + li) * 32;
#pragma unroll
        for (int kk = 0; kk < BKE / 16; kk++) {
            const int ch = ((2 * kk + lh) ^ sw) * 4;
            bf16x8 av[MT], bv[NT];
#pragma unroll
            for (int t = 0; t < MT; t++) {
                const uint4 a = *reinterpret_cast<const uint4*>(as + t * 32 * 32 + ch);
                av[t] = *reinterpret_cast<const bf16x8*>(&a);
            }
#pragma unroll
            for (int t = 0; t < NT; t++) {
                const uint4 b = *reinterpret_cast<const uint4*>(bs + t * 32 * 32 + ch);
                bv[t] = *reinterpret_cast<const bf16x8*>(&b);
            }
#pragma unroll
            for (int tm = 0; tm < MT; tm++)
#pragma unroll
                for (int t = 0; t < NT; t++)
                    acc[tm][t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(av[tm], bv[t], acc[tm][t], 0, 0, 0);
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        cur ^= 1;
    }

    unsigned short* __restrict__ yh = reinterpret_cast<unsigned short*>(p.y);
    float* __restrict__ yf = p.y;
#pragma unroll
    for (int tn = 0; tn < NT; tn++) {
        const int co = n0 + wn * 32 * NT + tn * 32 + li;
        const bool cok = co < p.Cout;
        const float sc = (p.scale && cok) ? p.scale[co] : 1.f;
        const float sh = (p.shift && cok) ? p.shift[co] : 0.f;
#pragma unroll
        for (int tm = 0; tm < MT; tm++) {
            const int mb = m0 + wm * 32 * MT + tm * 32 + 4 * lh;
#pragma unroll
            for (int r = 0; r < 16; r++) {
                const int m = mb + (r & 3) + 8 * (r >> 2);
                float v = acc[tm][tn][r];
                if (p.scale) v = v * sc;
                v = v + sh;
                if (RES) v = v + rv[tm][tn][r];
                if (p.relu) v = fmaxf(v, 0.f);
                if (cok && m < p.M) {
                    if (OUTF32) yf[(size_t)m * p.Cout + co] = v;
                    else yh[(size_t)m * p.Cout + co] = f2bf(v);
                }
            }
        }
    }
}

template <int MT, int NT, bool RES, bool OUTF32>
int launch(const ConvParams& p, hipStream_t s) {
    const size_t lds = (size_t)2 * (64 * MT + 64 * NT) * 32 * sizeof(float);
    static bool attr_done = false;
    if (!attr_done) {
        BRCNN_HIP_CHECK(hipFuncSetAttribute((const void*)conv_igemm_bf16_dma_kernel<MT, NT, RES, OUTF32>,
                                            hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        attr_done = true;
    }
    hipLaunchKernelGGL((conv_igemm_bf16_dma_kernel<MT, NT, RES, OUTF32>), dim3(p.tiles_m * p.tiles_n),
                       dim3(256), lds, s, p);
    BRCNN_LAUNCH_CHECK();
    return 0;
}

template <int MT, int NT>
int launch2(ConvParams& p, hipStream_t s) {
    p.tiles_m = (p.M + 64 * MT - 1) / (64 * MT);
    p.tiles_n = (p.Cout + 64 * NT - 1) / (64 * NT);
    if (p.out_f32) return p.residual ? launch<MT, NT, true, true>(p, s) : launch<MT, NT, false, true>(p, s);
    return p.residual ? launch<MT, NT, true, false>(p, s) : launch<MT, NT, false, false>(p, s);
}

int g_bf16_tile = 0;   // tuning hook: 0 heuristic, 11 / 21 / 22 = MT NT

}  // namespace

namespace brcnn_conv {
int dispatch_conv_bf16(ConvParams& p, hipStream_t s) {
    if (p.dilate != 1) return BRCNN_EINVAL;      // inference path only this round
    int t = g_bf16_tile;
    if (t == 0) {
        // enough 128x128 tiles to give every CU >= 4 workgroups -> the big tile; else smaller
        const long long t22 = (long long)((p.M + 127) / 128) * ((p.Cout + 127) / 128);
        t = (p.Cout <= 64) ? 21 : (t22 >= 1024 ? 22 : (t22 >= 256 ? 21 : 11));
    }
    if (t == 22 && p.Cout > 64) return launch2<2, 2>(p, s);
    if (t == 21 || (t == 22 && p.Cout <= 64)) return launch2<2, 1>(p, s);
    return launch2<1, 1>(p, s);
}
}  // namespace brcnn_conv

BRCNN_API int brcnn_conv_set_tile_bf16(int mtnt) {
    if (mtnt != 0 && mtnt != 11 && mtnt != 21 && mtnt != 22) return BRCNN_EINVAL;
    g_bf16_tile = mtnt;
    return 0;
}
