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
// round-off.  A workgroup is WM x 2 waves, each wave owning a 64 x (32*NT) output tile
// (2 x NT MFMA tiles); block tile = (64*WM) x (64*NT), K tile 32:
//     WM=2 NT=2 : 128x128, 256 threads, 2 workgroups / CU      (default)
//     WM=2 NT=1 : 128x64  for Cout <= 64
//     WM=4 NT=2 : 256x128, 512 threads, 1 workgroup / CU       (large M: fewer L2->LDS bytes
//                                                               per FLOP)
// Operands are staged global -> registers -> LDS (the next tile's loads are issued before the
// MFMA block and written after it); LDS rows are padded to 36 floats so the ds_read_b128
// fragment reads are bank-conflict free; each lane reads 4 consecutive k per b128 and the K
// order inside an 8-deep group is permuted identically for A and W (lane half h takes
// k = 4h..4h+3), which keeps every product paired correctly and needs one LDS read per 4
// MFMAs per operand.  Epilogue fused: per-channel scale/shift (folded BN or bias), residual
// add, ReLU; the residual tile is prefetched into registers before the K loop so that its
// latency hides under the MFMAs.  Block ids are remapped so that each XCD's L2 sees a
// contiguous run of tiles (the N tiles of one M tile share the gathered A rows).
// Several feature maps that share one set of weights (pyramid levels) run as ONE launch: the
// output rows of the segments are concatenated and each row carries its own geometry.
#include "conv_common.h"

namespace {
using namespace brcnn_conv;

// FAST: Cin % 32 == 0 (a K tile never straddles a filter tap).  RES: residual operand present.
template <bool FAST, int WM, int NT, bool RES>
__global__ __launch_bounds__(128 * WM, 2) void conv_igemm_f32_kernel(ConvParams p) {
    constexpr int THREADS = 128 * WM;
    constexpr int BM = 64 * WM, BN = 64 * NT;
    constexpr int RPP = THREADS / 8;            // tile rows staged per pass (8 float4 per row)
    constexpr int AJ = BM / RPP;                // == 4
    constexpr int BJ = (BN + RPP - 1) / RPP;    // 1, 2 or 4
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* As = smem;                          // [2][BM][LDS_STRIDE]
    float* Bs = smem + 2 * BM * LDS_STRIDE;    // [2][BN][LDS_STRIDE]

    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int li = lane & 31, lh = lane >> 5;
    const float* __restrict__ xin = p.x;

    const int nwg = p.tiles_m * p.tiles_n;
    const int tile = xcd_remap(blockIdx.x, nwg);
    const int tile_m = tile / p.tiles_n, tile_n = tile - tile_m * p.tiles_n;
    const int m0 = tile_m * BM, n0 = tile_n * BN;

    // ---- staging assignment: float4 column c4 of rows r0 + RPP*j --------------------------
    const int c4 = tid & 7;
    const int r0 = tid >> 3;
    int a_base[AJ];                // element offset of the row's image; < 0: row outside M
    int a_hw[AJ], a_H[AJ], a_W[AJ];   // a_hw packs (hi0 + 4096) << 16 | (wi0 + 4096)
    const float* b_ptr[BJ];
    int b_off[BJ];                 // element offset of the weight row; < 0: outside Cout
    bool b_ok[BJ];
    const __amdgpu_buffer_rsrc_t rsrc_x = __builtin_amdgcn_make_buffer_rsrc((void*)p.x, 0, (int)p.x_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rsrc_w = __builtin_amdgcn_make_buffer_rsrc((void*)p.w, 0, (int)p.w_bytes, 0x00020000);
#pragma unroll
    for (int j = 0; j < AJ; j++) {
        const int m = m0 + r0 + RPP * j;
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
    for (int j = 0; j < BJ; j++) {
        const int row = r0 + RPP * j;
        const int co = n0 + row;
        b_ok[j] = (row < BN) && (co < p.Cout);
        b_ptr[j] = p.w + (size_t)(b_ok[j] ? co : 0) * p.K;
        b_off[j] = b_ok[j] ? co * p.K + c4 * 4 : -1;
    }

    f32x16 acc[2][NT];
#pragma unroll
    for (int a = 0; a < 2; a++)
#pragma unroll
        for (int b = 0; b < NT; b++)
#pragma unroll
            for (int r = 0; r < 16; r++) acc[a][b][r] = 0.f;

    float4 ra[AJ], rb[BJ];
    const int nk = (p.K + BK - 1) / BK;

    auto load_tile = [&](int kt) {
        int k0 = kt * BK;
        if (FAST) {
            // (chunk-major K order of the LDS-DMA kernels: K tile kt = channel chunk kt / T, tap kt % T)
            const int T = p.KH * p.KW;
            const int cc = kt / T, tap = kt - cc * T;
            const int ci = cc * BK + c4 * 4;
            const int kh = tap / p.KW, kw = tap - kh * p.KW;
            k0 = tap * p.Cin + cc * BK;
#pragma unroll
            for (int j = 0; j < AJ; j++) {
                int hi = (a_hw[j] >> 16) - 4096 + kh;
                int wi = (a_hw[j] & 0xffff) - 4096 + kw;
                bool ok = a_base[j] >= 0;
                if (p.dilate > 1) {      // zero-stuffed input: only multiples of `dilate` exist
                    brcnn_undilate(p.dilate, hi, wi, ok);
                }
                ok = ok & ((unsigned)hi < (unsigned)a_H[j]) & ((unsigned)wi < (unsigned)a_W[j]);
                const int off = ok ? (a_base[j] + (hi * a_W[j] + wi) * p.pitch + ci) * 4 : OOB;
                const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(rsrc_x, off, 0, 0);
                ra[j] = make_float4(__uint_as_float(v.x), __uint_as_float(v.y), __uint_as_float(v.z),
                                    __uint_as_float(v.w));
            }
#pragma unroll
            for (int j = 0; j < BJ; j++) {
                const int off = (b_off[j] >= 0) ? (b_off[j] + k0) * 4 : OOB;
                const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(rsrc_w, off, 0, 0);
                rb[j] = make_float4(__uint_as_float(v.x), __uint_as_float(v.y), __uint_as_float(v.z),
                                    __uint_as_float(v.w));
            }
        } else {
#pragma unroll
            for (int j = 0; j < AJ; j++) {
                float va[4];
#pragma unroll
                for (int e = 0; e < 4; e++) {
                    const int k = k0 + c4 * 4 + e;
                    va[e] = 0.f;
                    if (k < p.K) {
                        const int tap = k / p.Cin, ci = k - tap * p.Cin;
                        const int kh = tap / p.KW, kw = tap - kh * p.KW;
                        const int hi = (a_hw[j] >> 16) - 4096 + kh;
                        const int wi = (a_hw[j] & 0xffff) - 4096 + kw;
                        if (a_base[j] >= 0 && hi >= 0 && hi < a_H[j] && wi >= 0 && wi < a_W[j])
                            va[e] = xin[(long long)a_base[j] + ((long long)(hi * a_W[j] + wi)) * p.pitch + ci];
                    }
                }
                ra[j] = make_float4(va[0], va[1], va[2], va[3]);
            }
#pragma unroll
            for (int j = 0; j < BJ; j++) {
                float vb[4];
#pragma unroll
                for (int e = 0; e < 4; e++) {
                    const int k = k0 + c4 * 4 + e;
                    vb[e] = (k < p.K && b_ok[j]) ? b_ptr[j][k] : 0.f;
                }
                rb[j] = make_float4(vb[0], vb[1], vb[2], vb[3]);
            }
        }
    };
    auto store_tile = [&](int buf) {
        float* as = As + buf * BM * LDS_STRIDE;
        float* bs = Bs + buf * BN * LDS_STRIDE;
#pragma unroll
        for (int j = 0; j < AJ; j++)
            *reinterpret_cast<float4*>(as + (r0 + RPP * j) * LDS_STRIDE + c4 * 4) = ra[j];
#pragma unroll
        for (int j = 0; j < BJ; j++)
            if (BJ * RPP == BN || r0 + RPP * j < BN)
                *reinterpret_cast<float4*>(bs + (r0 + RPP * j) * LDS_STRIDE + c4 * 4) = rb[j];
    };

    load_tile(0);

    // ---- residual prefetch: D layout col = lane&31, row = (r&3) + 8*(r>>2) + 4*(lane>>5) ----
    const float* __restrict__ res = p.residual;
    float rv[2][NT][16];
    if (RES) {
#pragma unroll
        for (int tn = 0; tn < NT; tn++) {
            const int co = n0 + wn * 32 * NT + tn * 32 + li;
#pragma unroll
            for (int tm = 0; tm < 2; tm++) {
                const int mb = m0 + wm * 64 + tm * 32 + 4 * lh;
#pragma unroll
                for (int r = 0; r < 16; r++) {
                    const int m = mb + (r & 3) + 8 * (r >> 2);
                    rv[tm][tn][r] = (co < p.Cout && m < p.M) ? res[(size_t)m * p.Cout + co] : 0.f;
                }
            }
        }
    }

    store_tile(0);
    __syncthreads();

    auto mfma_group = [&](const float* as, const float* bs, int kk) {
        const float4 a0 = *reinterpret_cast<const float4*>(as + kk * 8);
        const float4 a1 = *reinterpret_cast<const float4*>(as + 32 * LDS_STRIDE + kk * 8);
        const float av[2][4] = {{a0.x, a0.y, a0.z, a0.w}, {a1.x, a1.y, a1.z, a1.w}};
        float bv[NT][4];
#pragma unroll
        for (int t = 0; t < NT; t++) {
            const float4 b = *reinterpret_cast<const float4*>(bs + t * 32 * LDS_STRIDE + kk * 8);
            bv[t][0] = b.x; bv[t][1] = b.y; bv[t][2] = b.z; bv[t][3] = b.w;
        }
#pragma unroll
        for (int e = 0; e < 4; e++)
#pragma unroll
            for (int tm = 0; tm < 2; tm++)
#pragma unroll
                for (int t = 0; t < NT; t++)
                    acc[tm][t] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[tm][e], bv[t][e], acc[tm][t],
                                                                      0, 0, 0);
    };

    // K loop.  Per iteration: the next tile's buffer loads are issued first, half of the MFMAs
    // run, the (by then landed) registers are written to the other LDS buffer, the second half
    // of the MFMAs runs, one barrier.  No branches inside: the scheduler is free to slot the
    // address arithmetic and the LDS writes into the 64-cycle shadows of the MFMAs.
    int cur = 0;
    for (int kt = 0; kt < nk; kt++) {
        const bool more = kt + 1 < nk;
        if (more) load_tile(kt + 1);
        const float* as = As + cur * BM * LDS_STRIDE + (wm * 64 + li) * LDS_STRIDE + lh * 4;
        const float* bs = Bs + cur * BN * LDS_STRIDE + (wn * 32 * NT + li) * LDS_STRIDE + lh * 4;
        mfma_group(as, bs, 0);
        mfma_group(as, bs, 1);
        if (more) store_tile(cur ^ 1);
        mfma_group(as, bs, 2);
        mfma_group(as, bs, 3);
        __syncthreads();
        cur ^= 1;
    }

    // ---- epilogue: scale/shift, residual, relu ----------------------------------------------
    float* __restrict__ yout = p.y;
#pragma unroll
    for (int tn = 0; tn < NT; tn++) {
        const int co = n0 + wn * 32 * NT + tn * 32 + li;
        const bool cok = co < p.Cout;
        const float sc = (p.scale && cok) ? p.scale[co] : 1.f;
        const float sh = (p.shift && cok) ? p.shift[co] : 0.f;
#pragma unroll
        for (int tm = 0; tm < 2; tm++) {
            const int mb = m0 + wm * 64 + tm * 32 + 4 * lh;
#pragma unroll
            for (int r = 0; r < 16; r++) {
                const int m = mb + (r & 3) + 8 * (r >> 2);
                float v = acc[tm][tn][r];
                if (p.scale) v = v * sc;
                v = v + sh;
                if (RES) v = v + rv[tm][tn][r];
                if (p.relu) v = fmaxf(v, 0.f);
                if (cok && m < p.M) yout[(size_t)m * p.Cout + co] = v;
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------
// FAST-path kernel with LDS-DMA staging (`buffer_load_dwordx4 ... lds`): the operand tiles go
// HBM/L2 -> LDS without passing through VGPRs, so the K loop has no ds_write pass and no
// staging registers; one `s_waitcnt vmcnt(0)` + barrier per K tile.  A DMA instruction writes
// the wave's 64 x 16 B contiguously, so the LDS tile is unpadded [row][8 chunks of 16 B] and the
// bank-conflict-free read pattern comes from an XOR swizzle applied on the SOURCE side: the
// lane that lands at physical chunk c' of row r fetches logical chunk c' ^ ((r >> 1) & 7); the
// fragment reads apply the same involution.  Tile 128 x (64*NT) x 32, 4 waves (2x2).
template <int MT, int NT, bool RES>
__global__ __launch_bounds__(256, 2) void conv_igemm_f32_dma_kernel(ConvParams p) {
    constexpr int BM = 64 * MT, BN = 64 * NT;
    constexpr int AG = BM / 8 / 4;      // 8-row groups of the A tile per wave (4)
    constexpr int BG = BN / 8 / 4;      // 8-row groups of the W tile per wave (4 or 2)
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* As = smem;                   // [2][BM][32]
    float* Bs = smem + 2 * BM * 32;     // [2][BN][32]
    typedef __attribute__((address_space(3))) void* lds_ptr_t;

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

    // ---- DMA assignment: wave w moves row groups w*AG+j; lane -> (row in group, physical chunk)
    const int rg = lane >> 3, pc = lane & 7;
    const bool plain = !p.no_fast && p.KH == 1 && p.KW == 1 && p.stride == 1 && p.pad == 0 && p.nseg == 1 && p.dilate <= 1;
    int a_base[AG], a_hw[AG], a_H[AG], a_W[AG], a_lc[AG];
    int b_off[BG];
#pragma unroll
    for (int j = 0; j < AG; j++) {
        const int r = (wave * AG + j) * 8 + rg;
        a_lc[j] = (pc ^ ((r >> 1) & 7)) * 4;          // logical k offset (floats) of this lane's chunk
        const int m = m0 + r;
        if (plain) {
            // a 1x1 stride-1 layer on one map reads row m of x for row m of the output: no map lookup, no divisions
            a_base[j] = m < p.M ? (int)p.seg_xoff[0] + m * p.pitch : -1;
            a_hw[j] = (4096 << 16) | 4096;
            a_H[j] = a_W[j] = 1;
        } else if (m < p.M) {
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
        b_off[j] = (co < p.Cout) ? co * p.K + (pc ^ ((r >> 1) & 7)) * 4 : -1;
    }

    f32x16 acc[MT][NT];
#pragma unroll
    for (int a = 0; a < MT; a++)
#pragma unroll
        for (int b = 0; b < NT; b++)
#pragma unroll
            for (int r = 0; r < 16; r++) acc[a][b][r] = 0.f;

    const int nk = (p.K + BK - 1) / BK;

    // K tiles are staged in order, so the filter tap (kh, kw) and the channel offset of the next
    // tile are carried as scalar state instead of being re-derived by two integer divisions
    int d_k0 = 0, d_ci0 = 0, d_kh = 0, d_kw = 0;
    // plain (1x1 stride-1, one map) layers: the byte offset of every piece is its offset in K tile 0 plus 128 bytes per K
    // tile -- one add per piece and tile instead of the tap / bounds arithmetic (a row beyond M starts at OOB and the sum
    // stays beyond the buffer's extent: x and w are < 2 GiB)
    unsigned a_off4[AG], b_off4[BG];
#pragma unroll
    for (int j = 0; j < AG; j++) a_off4[j] = a_base[j] >= 0 ? (unsigned)(a_base[j] + a_lc[j] + tile_n * p.gstep) * 4u : (unsigned)OOB;
#pragma unroll
    for (int j = 0; j < BG; j++) b_off4[j] = b_off[j] >= 0 ? (unsigned)b_off[j] * 4u : (unsigned)OOB;
    unsigned d_plain = 0;
    auto dma_tile_plain = [&](int buf) {
        const unsigned step = d_plain;
        d_plain += BK * 4;
#pragma unroll
        for (int j = 0; j < AG; j++) {
            float* dst = As + buf * BM * 32 + (wave * AG + j) * 8 * 32;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc_x, (lds_ptr_t)dst, 16, (int)(a_off4[j] + step), 0, 0, 0);
        }
#pragma unroll
        for (int j = 0; j < BG; j++) {
            float* dst = Bs + buf * BN * 32 + (wave * BG + j) * 8 * 32;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc_w, (lds_ptr_t)dst, 16, (int)(b_off4[j] + step), 0, 0, 0);
        }
    };
    auto dma_tile = [&](int kt, int buf) {
        (void)kt;
        if (plain) { dma_tile_plain(buf); return; }
        // (channel chunk by channel chunk, the filter taps inside a chunk: see conv_pp_f32.hip; every fp32 kernel
        // visits K in this order, so their results stay bit-identical to each other)
        const int ci0 = d_ci0, kh = d_kh, kw = d_kw;
        const int k0 = (kh * p.KW + kw) * p.Cin + ci0;
        (void)d_k0;
        if (++d_kw == p.KW) {
            d_kw = 0;
            if (++d_kh == p.KH) { d_kh = 0; d_ci0 += BK; }
        }
#pragma unroll
        for (int j = 0; j < AG; j++) {
            int hi = (a_hw[j] >> 16) - 4096 + kh;
            int wi = (a_hw[j] & 0xffff) - 4096 + kw;
            bool ok = a_base[j] >= 0;
            if (p.dilate > 1) {
                brcnn_undilate(p.dilate, hi, wi, ok);
            }
            ok = ok & ((unsigned)hi < (unsigned)a_H[j]) & ((unsigned)wi < (unsigned)a_W[j]);
            const int off = ok ? (a_base[j] + (hi * a_W[j] + wi) * p.pitch + ci0 + a_lc[j] + tile_n * p.gstep) * 4 : OOB;
            float* dst = As + buf * BM * 32 + (wave * AG + j) * 8 * 32;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc_x, (lds_ptr_t)dst, 16, off, 0, 0, 0);
        }
#pragma unroll
        for (int j = 0; j < BG; j++) {
            const int off = (b_off[j] >= 0) ? (b_off[j] + k0) * 4 : OOB;
            float* dst = Bs + buf * BN * 32 + (wave * BG + j) * 8 * 32;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc_w, (lds_ptr_t)dst, 16, off, 0, 0, 0);
        }
    };

    dma_tile(0, 0);

    // residual tile, fetched at the top of the LAST K tile -- no LDS-DMA is in flight any more, so
    // nothing has to wait for it before the epilogue, and its latency hides under that tile's
    // MFMAs.  Channel counts that are multiples of 4 use the row mapping of the vectorised
    // epilogue (lane -> row it*8 + lane/8, 4 channels at 4*(lane%8): one 16-byte load per 4
    // values); ragged counts (fused heads) the accumulator (D) layout col = lane&31,
    // row = (r&3) + 8*(r>>2) + 4*(lane>>5).
    const float* __restrict__ res = p.residual;
    const bool vec = (p.Cout & 3) == 0;
    const int vrow = lane >> 3, vcol = (lane & 7) * 4;
    float rv[MT][NT][16];
    // the block tile lies inside the output (wave-uniform): residual loads and the read-out without guards
    const bool full = !p.no_fast && vec && !p.scatter && m0 + BM <= p.M && n0 + BN <= p.Cout;
    const size_t row0 = (size_t)(m0 + wm * 32 * MT + vrow) * p.Cout + n0 + wn * 32 * NT + vcol;
    auto load_residual = [&]() {
        if (full) {
            const float* __restrict__ rrow = res + row0;
#pragma unroll
            for (int tn = 0; tn < NT; tn++)
#pragma unroll
                for (int tm = 0; tm < MT; tm++)
#pragma unroll
                    for (int it = 0; it < 4; it++) {
                        const float4 q = *reinterpret_cast<const float4*>(rrow + (size_t)(tm * 32 + it * 8) * p.Cout + tn * 32);
                        rv[tm][tn][4 * it + 0] = q.x; rv[tm][tn][4 * it + 1] = q.y;
                        rv[tm][tn][4 * it + 2] = q.z; rv[tm][tn][4 * it + 3] = q.w;
                    }
            return;
        }
#pragma unroll
        for (int tn = 0; tn < NT; tn++) {
#pragma unroll
            for (int tm = 0; tm < MT; tm++) {
                if (vec) {
                    const int co = n0 + wn * 32 * NT + tn * 32 + vcol;
#pragma unroll
                    for (int it = 0; it < 4; it++) {
                        const int m = m0 + wm * 32 * MT + tm * 32 + it * 8 + vrow;
                        float4 q = make_float4(0.f, 0.f, 0.f, 0.f);
                        if (co < p.Cout && m < p.M) q = *reinterpret_cast<const float4*>(res + (size_t)m * p.Cout + co);
                        rv[tm][tn][4 * it + 0] = q.x; rv[tm][tn][4 * it + 1] = q.y;
                        rv[tm][tn][4 * it + 2] = q.z; rv[tm][tn][4 * it + 3] = q.w;
                    }
                } else {
                    const int co = n0 + wn * 32 * NT + tn * 32 + li;
                    const int mb = m0 + wm * 32 * MT + tm * 32 + 4 * lh;
#pragma unroll
                    for (int r = 0; r < 16; r++) {
                        const int m = mb + (r & 3) + 8 * (r >> 2);
                        rv[tm][tn][r] = (co < p.Cout && m < p.M) ? res[(size_t)m * p.Cout + co] : 0.f;
                    }
                }
            }
        }
    };
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();

    // fragment addressing: row R = base + li, logical chunk c = 2*kk + lh, physical c ^ ((R>>1)&7);
    // all row bases are multiples of 32, so (R>>1)&7 == (li>>1)&7.
    // The fragment reads are INLINE ASM on purpose: a compiler-visible LDS read after a
    // `buffer_load ... lds` makes the waitcnt pass put `s_waitcnt vmcnt(0)` in front of it (it
    // cannot tell the two buffers apart), which serialises the prefetch of tile kt+1 with the
    // MFMAs of tile kt inside every wave.  With opaque reads the only vmcnt wait of an iteration
    // is the explicit one before the barrier, so the DMA lands under the MFMA block.
    const int sw = (li >> 1) & 7;
    unsigned chb[BK / 8];
#pragma unroll
    for (int kk = 0; kk < BK / 8; kk++) chb[kk] = (unsigned)(((2 * kk + lh) ^ sw) * 16);
    const unsigned a_lane = (unsigned)(size_t)(lds_ptr_t)(As + (wm * 32 * MT + li) * 32);
    const unsigned b_lane = (unsigned)(size_t)(lds_ptr_t)(Bs + (wn * 32 * NT + li) * 32);
    f32x4 av[2][MT], bv[2][NT];
    auto frag_read = [&](int slot, unsigned a_addr, unsigned b_addr) {
        asm volatile("ds_read_b128 %0, %1" : "=v"(av[slot][0]) : "v"(a_addr) : "memory");
        if constexpr (MT == 2)
            asm volatile("ds_read_b128 %0, %1 offset:4096" : "=v"(av[slot][1]) : "v"(a_addr) : "memory");
        asm volatile("ds_read_b128 %0, %1" : "=v"(bv[slot][0]) : "v"(b_addr) : "memory");
        if constexpr (NT == 2)
            asm volatile("ds_read_b128 %0, %1 offset:4096" : "=v"(bv[slot][1]) : "v"(b_addr) : "memory");
    };
    auto frag_wait = [&](int slot) {
        if constexpr (MT == 2 && NT == 2)
            asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(av[slot][0]), "+v"(av[slot][1]), "+v"(bv[slot][0]), "+v"(bv[slot][1]) :: "memory");
        else if constexpr (MT == 2)
            asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(av[slot][0]), "+v"(av[slot][1]), "+v"(bv[slot][0]) :: "memory");
        else if constexpr (NT == 2)
            asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(av[slot][0]), "+v"(bv[slot][0]), "+v"(bv[slot][1]) :: "memory");
        else
            asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(av[slot][0]), "+v"(bv[slot][0]) :: "memory");
    };
    auto compute_tile = [&](int cur, bool prefetch) {
        const unsigned a_cur = a_lane + cur * (BM * 32 * 4);
        const unsigned b_cur = b_lane + cur * (BN * 32 * 4);
        __builtin_amdgcn_s_setprio(1);
        frag_read(0, a_cur + chb[0], b_cur + chb[0]);
        if (prefetch) dma_tile(0, cur ^ 1);      // issued under the latency of the first fragment read
        frag_wait(0);
        __builtin_amdgcn_s_setprio(0);
#pragma unroll
        for (int kk = 0; kk < BK / 8; kk++) {
            const int sl = kk & 1;
            if (kk + 1 < BK / 8) frag_read(sl ^ 1, a_cur + chb[kk + 1], b_cur + chb[kk + 1]);
#pragma unroll
            for (int e = 0; e < 4; e++)
#pragma unroll
                for (int tm = 0; tm < MT; tm++)
#pragma unroll
                    for (int t = 0; t < NT; t++)
                        acc[tm][t] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[sl][tm][e], bv[sl][t][e],
                                                                          acc[tm][t], 0, 0, 0);
            if (kk + 1 < BK / 8) frag_wait(sl ^ 1);
        }
    };
    int cur = 0;
    for (int kt = 0; kt + 1 < nk; kt++) {
        compute_tile(cur, true);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        cur ^= 1;
    }
    // last K tile: no prefetch, no barrier (nothing in LDS is read afterwards)
    if (RES) load_residual();
    compute_tile(cur, false);

    // ---- epilogue.  Scale / shift in the accumulator layout; for channel counts that are
    // multiples of 4 each wave then transposes one 32x32 tile at a time through a private 4 KiB
    // slab of the idle operand buffer (the one the last K tile did not read), so that residual
    // loads and result stores are 16-byte pieces of NHWC rows: 4 wide stores per tile instead
    // of 16 scalar ones (the store tail is issue bound).
    float* __restrict__ yout = p.y;
    float* slab = (wave < 2 ? As + (cur ^ 1) * BM * 32 : Bs + (cur ^ 1) * BN * 32) + (wave & 1) * 1024;
    if (full) {
        // straight-line read-out (the short-K 1x1 layers spend as many SIMD cycles on VALU as on MFMA, most of them in the
        // guards of the general form below): same arithmetic per element -- x * 1 is x, so the scale multiply is unconditional
        float* __restrict__ yrow = yout + row0;
#pragma unroll
        for (int tn = 0; tn < NT; tn++) {
            const int co = n0 + wn * 32 * NT + tn * 32 + li;
            const float sc = p.scale ? p.scale[co] : 1.f;
            const float sh = p.shift ? p.shift[co] : 0.f;
#pragma unroll
            for (int tm = 0; tm < MT; tm++) {
#pragma unroll
                for (int r = 0; r < 16; r += 2) {       // (v_pk_mul_f32 / v_pk_add_f32: two values per instruction)
                    const brcnn_f32x2 a = brcnn_f32x2{acc[tm][tn][r], acc[tm][tn][r + 1]} * brcnn_f32x2{sc, sc} + brcnn_f32x2{sh, sh};
                    slab[((r & 3) + 8 * (r >> 2) + 4 * lh) * 32 + li] = a.x;
                    slab[(((r + 1) & 3) + 8 * ((r + 1) >> 2) + 4 * lh) * 32 + li] = a.y;
                }
                __builtin_amdgcn_wave_barrier();
#pragma unroll
                for (int it = 0; it < 4; it++) {
                    float4 v = *reinterpret_cast<const float4*>(slab + (it * 8 + vrow) * 32 + vcol);
                    if (RES) {
                        v.x += rv[tm][tn][4 * it + 0]; v.y += rv[tm][tn][4 * it + 1];
                        v.z += rv[tm][tn][4 * it + 2]; v.w += rv[tm][tn][4 * it + 3];
                    }
                    if (p.relu) {       // med3(v, 0, inf) = max(v, 0) in one instruction (fmaxf costs a canonicalising second one)
                        v.x = brcnn_relu1(v.x); v.y = brcnn_relu1(v.y); v.z = brcnn_relu1(v.z); v.w = brcnn_relu1(v.w);
                    }
                    *reinterpret_cast<float4*>(yrow + (size_t)(tm * 32 + it * 8) * p.Cout + tn * 32) = v;
                }
                __builtin_amdgcn_wave_barrier();
            }
        }
        return;
    }
#pragma unroll
    for (int tn = 0; tn < NT; tn++) {
        const int co = n0 + wn * 32 * NT + tn * 32 + li;
        const bool cok = co < p.Cout;
        const float sc = (p.scale && cok) ? p.scale[co] : 1.f;
        const float sh = (p.shift && cok) ? p.shift[co] : 0.f;
#pragma unroll
        for (int tm = 0; tm < MT; tm++) {
            if (vec) {
#pragma unroll
                for (int r = 0; r < 16; r++) {
                    float v = acc[tm][tn][r];
                    if (p.scale) v = v * sc;
                    slab[((r & 3) + 8 * (r >> 2) + 4 * lh) * 32 + li] = v + sh;
                }
                __builtin_amdgcn_wave_barrier();
                const int cv = n0 + wn * 32 * NT + tn * 32 + vcol;
#pragma unroll
                for (int it = 0; it < 4; it++) {
                    const int row = it * 8 + vrow;
                    const int m = m0 + wm * 32 * MT + tm * 32 + row;
                    float4 v = *reinterpret_cast<const float4*>(slab + row * 32 + vcol);
                    if (RES) {
                        v.x += rv[tm][tn][4 * it + 0]; v.y += rv[tm][tn][4 * it + 1];
                        v.z += rv[tm][tn][4 * it + 2]; v.w += rv[tm][tn][4 * it + 3];
                    }
                    if (p.relu) {
                        v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f);
                    }
                    if (cv < p.Cout && m < p.M) {
                        const long long ro = out_row_offset(p, m);
                        if (ro >= 0) *reinterpret_cast<float4*>(yout + ro + cv) = v;
                    }
                }
                __builtin_amdgcn_wave_barrier();
            } else {
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
                        const long long ro = out_row_offset(p, m);
                        if (ro >= 0) yout[ro + co] = v;
                    }
                }
            }
        }
    }
}

template <int MT, int NT, bool RES>
int launch_dma(const ConvParams& p, hipStream_t s) {
    const size_t lds = (size_t)2 * (64 * MT + 64 * NT) * 32 * sizeof(float);
    static bool attr_done = false;
    if (!attr_done) {
        BRCNN_HIP_CHECK(hipFuncSetAttribute((const void*)conv_igemm_f32_dma_kernel<MT, NT, RES>,
                                            hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        attr_done = true;
    }
    hipLaunchKernelGGL((conv_igemm_f32_dma_kernel<MT, NT, RES>), dim3(p.tiles_m * p.tiles_n), dim3(256), lds,
                       s, p);
    BRCNN_LAUNCH_CHECK();
    return 0;
}

template <bool FAST, int WM, int NT, bool RES>
int launch_conv(const ConvParams& p, hipStream_t s) {
    const size_t lds = (size_t)2 * (64 * WM + 64 * NT) * LDS_STRIDE * sizeof(float);
    static bool attr_done = false;
    if (!attr_done) {
        BRCNN_HIP_CHECK(hipFuncSetAttribute((const void*)conv_igemm_f32_kernel<FAST, WM, NT, RES>,
                                            hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        attr_done = true;
    }
    hipLaunchKernelGGL((conv_igemm_f32_kernel<FAST, WM, NT, RES>), dim3(p.tiles_m * p.tiles_n),
                       dim3(128 * WM), lds, s, p);
    BRCNN_LAUNCH_CHECK();
    return 0;
}

template <bool FAST, int WM, int NT>
int launch_res(const ConvParams& p, hipStream_t s) {
    return p.residual ? launch_conv<FAST, WM, NT, true>(p, s) : launch_conv<FAST, WM, NT, false>(p, s);
}

int g_use_dma = 1;                      // LDS-DMA staged kernel for the FAST path
int g_force_wm = 0, g_force_nt = 0;   // tuning hooks (brcnn_conv_set_tile): 0 = heuristic

int g_no_fast = 0;          // test hook (brcnn_conv_set_tile(-4, 0 / 1)): 1 = no straight-line read-out, no plain-layer set-up (every kernel, every dtype)
int g_pp_f32_n128 = 1;      // tuning hook (brcnn_conv_set_tile(-3, 0 / 1 / 2)): the 256 x 128 eight-phase tile never / heuristic / forced
int g_pp_f32_mode = 1;      // tuning hook (brcnn_conv_set_tile(-2, 0 / 1 / 2 / 128 / 256)): eight-phase fp32 kernel never / heuristic / forced (tile rows by the heuristic / 128 / 256)

int dispatch_conv(ConvParams& p, hipStream_t s) {
    const bool fast = (p.Cin % 32 == 0);
    // The 256 x 256 eight-phase kernel (conv_pp_f32.hip; bit-identical results) where its tiles cover the device: whole
    // 256-channel column tiles and at least ~0.8 generations of tiles (one workgroup per CU; from 256 tiles on the
    // chained stream-K schedule evens out the last generation)
    if (fast && g_pp_f32_mode && g_force_wm == 0 && g_force_nt == 0 && !p.gstep && (p.Cout % 256) == 0 && p.K >= 256 &&
        p.KH * p.KW <= 32 && !(p.dilate > 1 && p.residual)) {
        // (128-row tiles where 256-row ones would leave CUs idle: the stage-3 / stage-4 maps)
        const long long t48 = (long long)((p.M + 127) / 128) * (p.Cout / 256);
        p.pp_rows = g_pp_f32_mode >= 128 ? g_pp_f32_mode : 0;
        // (a K = 256 layer with a residual stream is eight K tiles deep and output-bound: measured 6 % behind the
        // 64 x 64 kernel, tools/conv_bench.py "s3 1x1 256->1024 +res")
        // ... and the few-tile layers of the 25 x 42 maps (66-132 such tiles on 256 CUs) with a K loop of >= 64 tiles: split-K
        // pieces summed at the end fill the device (stage-4 3x3: 444 -> 380 us, 2048 -> 512 1x1: 185 -> 171 us; reproducible,
        // equal to the unsplit sum to fp32 round-off -- the one place where the kernel choice changes the association)
        const bool few = sk_par_enabled() && t48 >= 64 && t48 < 208 && p.K / 32 >= 64 && p.dilate <= 1;
        if (g_pp_f32_mode >= 2 || ((t48 >= 208 || few) && (p.K >= 512 || !p.residual))) return dispatch_conv_pp_f32(p, s);
    }
    // ... and its 256 x 128 form (r04) for the layers with 128 output channels (stage 2: 3x3 128->128, 1x1 256/512->128)
    if (fast && g_pp_f32_mode && g_pp_f32_n128 && g_force_wm == 0 && g_force_nt == 0 && !p.gstep && (p.Cout % 256) != 0 &&
        (p.Cout % 128) == 0 && p.K >= 256 && p.KH * p.KW <= 32 && !(p.dilate > 1 && p.residual)) {
        const long long t = (long long)((p.M + 255) / 256) * (p.Cout / 128);
        p.pp_rows = 0;
        p.pp_cols = 128;
        // (measured, batch 8: stage-2 3x3 128->128 1.43 -> 1.38 ms for the four layers, 105 -> 112 TF/s; the K = 256 / 512 1x1
        // layers are output-bound and no faster than on the 64 x 64 tile: 3x3-deep K loops only)
        if (g_pp_f32_n128 >= 2 || (t >= 208 && p.K >= 1024)) return dispatch_conv_pp_f32(p, s);
        p.pp_cols = 0;
    }
    // Measured on MI355X (profiles/r01_conv_tiles.txt): with LDS-DMA staging the SMALLEST tile,
    // 64x64 (4 waves x one 32x32 MFMA tile, 32 KiB LDS -> up to 5 resident workgroups / CU,
    // 4x the tiles of 128x128 -> hardly any tail quantisation), is the fastest on every layer
    // of the network, the 12544-deep FC included: the kernel is bound by MFMA issue and
    // latency hiding, not by L2->LDS bytes, so it is the default.  The other shapes stay
    // reachable through brcnn_conv_set_tile (tools/conv_bench.py).
    if (fast && g_use_dma && g_force_wm == 0 && g_force_nt == 0) {
        p.tiles_m = (p.M + 63) / 64;
        p.tiles_n = (p.Cout + 63) / 64;
        return p.residual ? launch_dma<1, 1, true>(p, s) : launch_dma<1, 1, false>(p, s);
    }
    int nt = (fast || p.Cout <= 64) ? 1 : 2;
    if (g_force_nt == 1 || (g_force_nt == 2 && p.Cout > 64)) nt = g_force_nt;
    int wm = 2;
    if (g_force_wm == 4 && fast && nt == 2) wm = 4;
    p.tiles_m = (p.M + 64 * wm - 1) / (64 * wm);
    p.tiles_n = (p.Cout + 64 * nt - 1) / (64 * nt);
    if (!fast) return nt == 1 ? launch_res<false, 2, 1>(p, s) : launch_res<false, 2, 2>(p, s);
    if (wm == 2 && g_use_dma) {
        if (g_force_wm == 1) {
            p.tiles_m = (p.M + 63) / 64;
            p.tiles_n = (p.Cout + 63) / 64;
            return p.residual ? launch_dma<1, 1, true>(p, s) : launch_dma<1, 1, false>(p, s);
        }
        if (nt == 1) return p.residual ? launch_dma<2, 1, true>(p, s) : launch_dma<2, 1, false>(p, s);
        return p.residual ? launch_dma<2, 2, true>(p, s) : launch_dma<2, 2, false>(p, s);
    }
    if (nt == 1) return launch_res<true, 2, 1>(p, s);
    return wm == 4 ? launch_res<true, 4, 2>(p, s) : launch_res<true, 2, 2>(p, s);
}

}  // namespace

namespace brcnn_conv {
int tuning_get_eight_phase_f32() { return g_pp_f32_mode; }
}  // namespace brcnn_conv

BRCNN_API int brcnn_conv_set_tile(int wm, int nt) {
    if (wm == -1) { g_use_dma = nt; return 0; }   // (-1, 0/1/2): register-staged / heuristic / always LDS-DMA
    if (wm == -3) { if (nt < 0 || nt > 2) return BRCNN_EINVAL; g_pp_f32_n128 = nt; return 0; }
    if (wm == -4) { if (nt != 0 && nt != 1) return BRCNN_EINVAL; g_no_fast = nt; return 0; }
    if (wm == -2) { if (nt != 0 && nt != 1 && nt != 2 && nt != 128 && nt != 256) return BRCNN_EINVAL; g_pp_f32_mode = nt; return 0; }
    if ((wm != 0 && wm != 1 && wm != 2 && wm != 4) || nt < 0 || nt > 2) return BRCNN_EINVAL;
    g_force_wm = wm;
    g_force_nt = nt;
    return 0;
}

BRCNN_API int brcnn_conv2d_nhwc(const void* x, const void* w, const float* scale, const float* shift,
                                const void* residual, void* y, int batch, int height, int width,
                                int cin, int cout, int kh, int kw, int stride, int pad, int relu,
                                int dtype, void* stream) {
    const int hs[1] = {height}, ws[1] = {width};
    return brcnn_conv2d_nhwc_multi(x, w, scale, shift, residual, y, batch, 1, hs, ws, cin, cout, kh,
                                   kw, stride, pad, relu, dtype, stream);
}

struct TrainTail {
    void* z_out; const float* mean; const float* var; float eps;
    const void* tail_z; float* partials; int tail_relu;        // data gradient + producer's BatchNorm backward
    const void* tail_mask; void* dres;                         // ... of a residual producer (see ConvParams)
};

int brcnn_bn_eval_reduce_launch(const float* partials, int strips, const float* mean, const float* var, float eps,
                                float* dgamma, float* dbeta, int channels, hipStream_t s, int defer);     // bn_act.hip

static int conv_setup_and_launch(const void* x, const void* w, const float* scale, const float* shift,
                                 const void* residual, void* y, int batch, int num_segments,
                                 const int* heights_host, const int* widths_host,
                                 const int* out_heights_host, const int* out_widths_host, int cin,
                                 int cout, int kh, int kw, int stride, int pad, int dilate, int relu,
                                 int dtype, void* stream, int pitch = 0, const int* scatter = nullptr,
                                 const TrainTail* tail = nullptr, int* tiles_m_out = nullptr, int stem2 = 0) {
    if (!x || !w || !y || batch <= 0 || cin <= 0 || cout <= 0 || kh <= 0 || kw <= 0 || stride <= 0 ||
        pad < 0 || dilate < 1 || num_segments <= 0 || num_segments > BRCNN_MAX_LEVELS ||
        !heights_host || !widths_host)
        return BRCNN_EINVAL;
    const int esize = (dtype == BRCNN_DT_F32) ? 4 : 2;
    const bool bf16 = brcnn_is16(dtype);          // 16-bit operands (bf16 or fp16), fp32 accumulate
    if (dtype != BRCNN_DT_F32 && !bf16) return BRCNN_EINVAL;
    if (bf16 && cin % 64 != 0) return BRCNN_EINVAL;
    if (dilate > 1 && (cin % 32 != 0 || stride != 1)) return BRCNN_EINVAL;
    if (bf16 && (cout & 7) && !brcnn_out_f32(dtype) && residual) return BRCNN_EINVAL;
    ConvParams p = {};
    p.no_fast = g_no_fast;
    p.x = (const float*)x; p.w = (const float*)w; p.scale = scale; p.shift = shift;
    p.residual = (const float*)residual; p.y = (float*)y;
    p.batch = batch; p.Cin = cin; p.Cout = cout; p.KH = kh; p.KW = kw;
    if (pitch <= 0) pitch = cin;
    p.stride = stride; p.pad = pad; p.pitch = pitch; p.nseg = num_segments; p.dilate = dilate;
    p.stem2 = stem2;
    long long m_total = 0, x_off = 0;
    for (int sgi = 0; sgi < num_segments; sgi++) {
        const int H = heights_host[sgi], W = widths_host[sgi];
        if (H <= 0 || W <= 0 || H * dilate + pad >= 4096 || W * dilate + pad >= 4096) return BRCNN_EINVAL;
        int Ho = (H + 2 * pad - kh) / stride + 1, Wo = (W + 2 * pad - kw) / stride + 1;
        if (out_heights_host && out_widths_host) { Ho = out_heights_host[sgi]; Wo = out_widths_host[sgi]; }
        if (Ho <= 0 || Wo <= 0 || Ho >= 4096 || Wo >= 4096) return BRCNN_EINVAL;
        p.seg_H[sgi] = H; p.seg_W[sgi] = W; p.seg_Ho[sgi] = Ho; p.seg_Wo[sgi] = Wo;
        p.seg_m0[sgi] = (int)m_total;
        p.seg_xoff[sgi] = x_off;
        m_total += (long long)batch * Ho * Wo;
        x_off += (long long)batch * H * W * pitch;
        if (m_total > 0x7fffffffLL) return BRCNN_EINVAL;
    }
    for (int sgi = num_segments; sgi <= BRCNN_MAX_LEVELS; sgi++) p.seg_m0[sgi] = (int)m_total;
    if (x_off * esize >= 0x7fffffffLL || (long long)cout * kh * kw * cin * esize >= 0x7fffffffLL) return BRCNN_EINVAL;
    p.x_bytes = (unsigned)(x_off * esize);
    p.w_bytes = (unsigned)((long long)cout * kh * kw * cin * esize);
    p.out_f32 = brcnn_out_f32(dtype) ? 1 : 0;
    p.f16 = brcnn_isf16(dtype) ? 1 : 0;
    p.M = (int)m_total;
    p.K = kh * kw * cin;
    p.relu = relu;
    if (scatter) {      // {out_h, out_w, ph, pw, origin, rows, cols}: single map, LDS-DMA kernels only
        if (num_segments != 1 || cin % 32 != 0 || residual) return BRCNN_EINVAL;
        p.scatter = 1;
        p.sc_H = scatter[0]; p.sc_W = scatter[1]; p.sc_ph = scatter[2]; p.sc_pw = scatter[3];
        p.sc_o = scatter[4]; p.sc_na = scatter[5]; p.sc_nb = scatter[6];
    }
    if (tail) {         // dual store: 16-bit kernels, whole 16-byte channel pieces
        if (!bf16 || (cout & 7) || brcnn_out_f32(dtype) || scatter || (!tail->z_out == !tail->tail_z) || !scale ||
            !shift || (tail->mean != nullptr) != (tail->var != nullptr) ||
            (tail->tail_z && (!tail->partials || (residual != nullptr) != (tail->tail_mask != nullptr) ||
                              (tail->tail_mask != nullptr) != (tail->dres != nullptr))))
            return BRCNN_EINVAL;
        p.z_out = tail->z_out; p.bn_mean = tail->mean; p.bn_var = tail->var; p.bn_eps = tail->eps;
        p.tail_z = tail->tail_z; p.tail_partials = tail->partials; p.tail_relu = tail->tail_relu;
        p.tail_mask = tail->tail_mask; p.tail_dres = tail->dres;
    }
    if (bf16) {
        const int st = dispatch_conv_bf16(p, (hipStream_t)stream);
        if (tiles_m_out) *tiles_m_out = p.tiles_m;
        return st;
    }
    return dispatch_conv(p, (hipStream_t)stream);
}

// Training forward of conv -> eval-mode BatchNorm (-> + residual) (-> ReLU) in ONE launch: the raw conv
// output z (which the BatchNorm backward needs for dgamma) and the activation y leave the same epilogue
// (resnet.py Bottleneck.forward:263-302 with norm_eval=True).  16-bit dtypes only.
BRCNN_API int brcnn_conv2d_bn_act_nhwc_multi(const void* x, const void* w, const float* gamma, const float* beta,
                                             const float* mean, const float* var, float eps, const void* residual,
                                             void* z_out, void* y, int batch, int num_segments,
                                             const int* heights_host, const int* widths_host, int cin, int cout,
                                             int kh, int kw, int stride, int pad, int relu, int dtype, void* stream) {
    if (!brcnn_is16(dtype) || brcnn_out_f32(dtype)) return BRCNN_EINVAL;
    const TrainTail tail = {z_out, mean, var, eps, nullptr, nullptr, 0, nullptr, nullptr};
    return conv_setup_and_launch(x, w, gamma, beta, residual, y, batch, num_segments, heights_host, widths_host,
                                 nullptr, nullptr, cin, cout, kh, kw, stride, pad, 1, relu, dtype, stream, 0, nullptr,
                                 &tail);
}

// Data gradient of a conv whose INPUT came out of conv -> eval-BN -> [ReLU] (the next layer up in a Bottleneck),
// with that BatchNorm's backward folded into the epilogue: dz_prev = (dx masked by the producer's ReLU) * scale
// leaves instead of dx, dgamma / dbeta of the producer's BatchNorm come from per-row-tile partial sums (fixed
// order, deterministic).  Saves the write + read of dx and a launch per layer.  Single map, 16-bit dtypes.
BRCNN_API size_t brcnn_conv2d_dgrad_bn_backward_workspace_bytes(int batch, int in_height, int in_width, int cin) {
    const long long m = (long long)batch * in_height * in_width;
    return (size_t)((m + 63) / 64) * 2 * (size_t)cin * sizeof(float);
}

BRCNN_API int brcnn_conv2d_dgrad_bn_backward_nhwc_ex(const void* dy, const void* w_t, const void* z_prev,
                                                  const float* gamma, const float* beta, const float* mean,
                                                  const float* var, float eps, int relu, const void* dskip,
                                                  const void* prev_out, void* dres, void* dz_prev,
                                                  float* dgamma, float* dbeta, void* workspace, size_t workspace_bytes,
                                                  int batch, int in_height, int in_width, int out_height, int out_width,
                                                  int cin, int cout, int kh, int kw, int stride, int pad, int dtype,
                                                  void* stream, int defer_second_stage) {
    if (!z_prev || !gamma || !beta || !mean || !var || !dz_prev || !dgamma || !dbeta || !workspace ||
        !brcnn_is16(dtype) || brcnn_out_f32(dtype) || pad > kh - 1 || pad > kw - 1 || (cin & 63))
        return BRCNN_EINVAL;
    // residual producer (bn3 of the previous block): identity gradient, producer output and dres go together
    if ((dskip != nullptr) != (prev_out != nullptr) || (dskip != nullptr) != (dres != nullptr)) return BRCNN_EINVAL;
    if (workspace_bytes < brcnn_conv2d_dgrad_bn_backward_workspace_bytes(batch, in_height, in_width, cin))
        return BRCNN_EINVAL;
    const TrainTail tail = {nullptr, mean, var, eps, z_prev, (float*)workspace, relu, prev_out, dres};
    const int ih[1] = {in_height}, iw[1] = {in_width}, oh[1] = {out_height}, ow[1] = {out_width};
    // roles swap as in brcnn_conv2d_dgrad_nhwc_multi: the kernel's "input" is dy, its "output" the input gradient
    int tiles_m = 0;
    const int st = conv_setup_and_launch(dy, w_t, gamma, beta, dskip, dz_prev, batch, 1, oh, ow, ih, iw, cout, cin, kh,
                                         kw, 1, kh - 1 - pad, stride, 0, dtype, stream, 0, nullptr, &tail, &tiles_m);
    if (st) return st;
    return brcnn_bn_eval_reduce_launch((const float*)workspace, tiles_m, mean, var, eps, dgamma, dbeta, cin,
                                       (hipStream_t)stream, defer_second_stage);
}

BRCNN_API int brcnn_conv2d_dgrad_bn_backward_nhwc(const void* dy, const void* w_t, const void* z_prev,
                                                  const float* gamma, const float* beta, const float* mean,
                                                  const float* var, float eps, int relu, const void* dskip,
                                                  const void* prev_out, void* dres, void* dz_prev,
                                                  float* dgamma, float* dbeta, void* workspace, size_t workspace_bytes,
                                                  int batch, int in_height, int in_width, int out_height, int out_width,
                                                  int cin, int cout, int kh, int kw, int stride, int pad, int dtype,
                                                  void* stream) {
    return brcnn_conv2d_dgrad_bn_backward_nhwc_ex(dy, w_t, z_prev, gamma, beta, mean, var, eps, relu, dskip, prev_out, dres,
                                                  dz_prev, dgamma, dbeta, workspace, workspace_bytes, batch, in_height,
                                                  in_width, out_height, out_width, cin, cout, kh, kw, stride, pad, dtype,
                                                  stream, 0);
}

BRCNN_API int brcnn_conv2d_nhwc_scatter2(const void* x, const void* w, void* y, int batch, int height, int width,
                                         int cin, int cout, int kh, int kw, int pad, int out_height,
                                         int out_width, int ph, int pw, int origin, int dtype, void* stream) {
    if (out_height <= 0 || out_width <= 0 || ph < 0 || ph > 1 || pw < 0 || pw > 1 || origin < 0 ||
        !brcnn_elem_ok(dtype))
        return BRCNN_EINVAL;
    const int Ho = height + 2 * pad - kh + 1, Wo = width + 2 * pad - kw + 1;
    int na = (out_height - ph + 1) / 2, nb = (out_width - pw + 1) / 2;      // pixels of this parity class
    if (na > Ho - origin) na = Ho - origin;                                  // rows the launch produces
    if (nb > Wo - origin) nb = Wo - origin;
    if (na <= 0 || nb <= 0) return 0;
    const int sc[7] = {out_height, out_width, ph, pw, origin, na, nb};
    const int hs[1] = {height}, ws[1] = {width};
    return conv_setup_and_launch(x, w, nullptr, nullptr, nullptr, y, batch, 1, hs, ws, nullptr, nullptr, cin, cout,
                                 kh, kw, 1, pad, 1, 0, dtype, stream, 0, sc);
}

// Grouped convolution (ResNeXt, resnext.py:10-84: conv2 of every bottleneck has `groups` = 32 / 64).
// An output tile of 64 channels covers 64/cg_out whole groups whose input channels form ONE
// contiguous window of `window` = 64 * cg_in / cg_out channels, so the tile is a dense 64 x
// (KH*KW*window) GEMM on block-diagonal weights: the 64x64 LDS-DMA kernel runs unchanged with
// its A operand shifted by tile_n * window channels.  `w_tiles` (Cout, KH, KW, window) holds,
// for output channel co of tile t = co / 64, the filter taps at window position
// (co / cg_out) * cg_in - t * window + ci and zeros elsewhere (brcnn.ops.pack_grouped_weight).
static int grouped_launch(const void* x, const void* w_tiles, const float* scale, const float* shift,
                          const void* residual, void* y, int batch, int height, int width, int cin, int cout,
                          int kh, int kw, int stride, int pad, int dilate, int Ho, int Wo, int window, int relu,
                          int dtype, void* stream) {
    if (!x || !w_tiles || !y || batch <= 0 || height <= 0 || width <= 0 || cin <= 0 || cout <= 0 || kh <= 0 ||
        kw <= 0 || stride <= 0 || pad < 0 || dilate < 1 || !brcnn_elem_ok(dtype) ||
        window <= 0 || (window % (dtype == BRCNN_DT_F32 ? 32 : 64)) ||
        (cout % 64) || (cout / 64) * window != cin || height * dilate + pad >= 4096 || width * dilate + pad >= 4096)
        return BRCNN_EINVAL;
    const int esz = dtype == BRCNN_DT_F32 ? 4 : 2;
    if (Ho <= 0 || Wo <= 0 || Ho >= 4096 || Wo >= 4096) return BRCNN_EINVAL;
    ConvParams p = {};
    p.no_fast = g_no_fast;
    p.x = (const float*)x; p.w = (const float*)w_tiles; p.scale = scale; p.shift = shift;
    p.residual = (const float*)residual; p.y = (float*)y;
    p.batch = batch; p.Cin = window; p.Cout = cout; p.KH = kh; p.KW = kw;
    p.stride = stride; p.pad = pad; p.pitch = cin; p.nseg = 1; p.dilate = dilate; p.gstep = window;
    p.seg_H[0] = height; p.seg_W[0] = width; p.seg_Ho[0] = Ho; p.seg_Wo[0] = Wo;
    p.seg_m0[0] = 0; p.seg_xoff[0] = 0;
    const long long m_total = (long long)batch * Ho * Wo, x_elems = (long long)batch * height * width * cin;
    for (int sgi = 1; sgi <= BRCNN_MAX_LEVELS; sgi++) p.seg_m0[sgi] = (int)m_total;
    p.K = kh * kw * window;
    if (x_elems * esz >= 0x7fffffffLL || (long long)cout * p.K * esz >= 0x7fffffffLL || m_total > 0x7fffffffLL)
        return BRCNN_EINVAL;
    p.x_bytes = (unsigned)(x_elems * esz);
    p.w_bytes = (unsigned)((long long)cout * p.K * esz);
    p.M = (int)m_total;
    p.relu = relu;
    p.f16 = dtype == BRCNN_DT_F16 ? 1 : 0;
    if (dtype != BRCNN_DT_F32) return dispatch_conv_bf16(p, (hipStream_t)stream);
    p.tiles_m = (p.M + 63) / 64;
    p.tiles_n = cout / 64;
    return p.residual ? launch_dma<1, 1, true>(p, (hipStream_t)stream) : launch_dma<1, 1, false>(p, (hipStream_t)stream);
}

BRCNN_API int brcnn_conv2d_nhwc_grouped(const void* x, const void* w_tiles, const float* scale,
                                        const float* shift, const void* residual, void* y, int batch,
                                        int height, int width, int cin, int cout, int kh, int kw,
                                        int stride, int pad, int window, int relu, int dtype,
                                        void* stream) {
    if (stride <= 0 || kh <= 0 || kw <= 0) return BRCNN_EINVAL;
    const int Ho = (height + 2 * pad - kh) / stride + 1, Wo = (width + 2 * pad - kw) / stride + 1;
    return grouped_launch(x, w_tiles, scale, shift, residual, y, batch, height, width, cin, cout, kh, kw, stride,
                          pad, 1, Ho, Wo, window, relu, dtype, stream);
}

// data gradient of the grouped conv: the same tiles on the zero-stuffed dy with the per-group
// flipped / transposed weights (packed as a grouped weight from the dy channels to the dx channels)
BRCNN_API int brcnn_conv2d_dgrad_nhwc_grouped(const void* dy, const void* w_t_tiles, void* dx, int batch,
                                              int in_height, int in_width, int out_height, int out_width,
                                              int cin, int cout, int kh, int kw, int stride, int pad,
                                              int window, int dtype, void* stream) {
    if (pad > kh - 1 || pad > kw - 1) return BRCNN_EINVAL;
    return grouped_launch(dy, w_t_tiles, nullptr, nullptr, nullptr, dx, batch, out_height, out_width, cout, cin,
                          kh, kw, 1, kh - 1 - pad, stride, in_height, in_width, window, 0, dtype, stream);
}

BRCNN_API int brcnn_conv2d_nhwc_multi(const void* x, const void* w, const float* scale,
                                      const float* shift, const void* residual, void* y, int batch,
                                      int num_segments, const int* heights_host,
                                      const int* widths_host, int cin, int cout, int kh, int kw,
                                      int stride, int pad, int relu, int dtype, void* stream) {
    return conv_setup_and_launch(x, w, scale, shift, residual, y, batch, num_segments, heights_host,
                                 widths_host, nullptr, nullptr, cin, cout, kh, kw, stride, pad, 1,
                                 relu, dtype, stream);
}

// Data gradient of conv(x (N,H,W,Cin), w, stride, pad) -> y (N,Ho,Wo,Cout):
//   dx[n,hi,wi,ci] = sum_{kh,kw,co} dy[n,(hi+pad-kh)/s,(wi+pad-kw)/s,co] * w[co,kh,kw,ci]
// = a stride-1 convolution of the zero-stuffed dy with the flipped, (co<->ci)-transposed
// weights `w_t` (Cin,KH,KW,Cout) [w_t[ci,a,b,co] = w[co,KH-1-a,KW-1-b,ci]] and padding K-1-pad.
BRCNN_API int brcnn_conv2d_dgrad_nhwc_multi(const void* dy, const void* w_t, void* dx, int batch,
                                            int num_segments, const int* in_heights_host,
                                            const int* in_widths_host, const int* out_heights_host,
                                            const int* out_widths_host, int cin, int cout, int kh,
                                            int kw, int stride, int pad, int dtype, void* stream) {
    if (pad > kh - 1 || pad > kw - 1 || !out_heights_host || !out_widths_host) return BRCNN_EINVAL;
    // roles swap: the kernel's "input" is dy (Ho,Wo,Cout), its "output" is dx (H,W,Cin)
    return conv_setup_and_launch(dy, w_t, nullptr, nullptr, nullptr, dx, batch, num_segments,
                                 out_heights_host, out_widths_host, in_heights_host, in_widths_host,
                                 cout, cin, kh, kw, 1, kh - 1 - pad, stride, 0, dtype, stream);
}

// ---------------------------------------------------------------------------------------------
// ResNet stem (7x7 / stride 2 / pad 3 on the 3-channel image, resnet.py:599-611) on the
// vector-load path: the NCHW image is re-packed once to a zero-bordered (N, H+6, W+6, 4)
// buffer; a filter row (7 taps x 3 channels) is then 28 of the 32 contiguous floats that
// start at pixel (ho*2+kh, wo*2), so the stem is a KH=7, KW=1, "Cin"=32 convolution whose
// input pixels are 4 floats apart (pitch 4) -- no per-element gather, no bounds tests.
// Weights arrive packed as (Cout, 7, 1, 32) with zeros at tap 7 / channel 3.
// 16-bit: a 128-byte K row holds 64 elements = the 8-pixel windows of TWO filter rows, so the stem is four K tiles deep
// (rows 0|1, 2|3, 4|5, 6|none) instead of seven half-empty ones: weights (Cout, 4, 1, 64) = [co, t, 0, r*32 + kw*4 + c]
// for filter row 2t + r; ConvParams::stem2 makes the upper half of a tile's pieces read image row + 1 (round 6:
// the kernel is bound by its LDS-DMA bytes, which fall by 3/7).
namespace {
// EXTRA = zero pixels appended to each row beyond the 3+3 border (so that the last window's
// full 32-float / 64-bf16 K row stays inside the row); bf16 variant packs 4 x bf16 per pixel.
template <int BF16>      // 0 fp32, 1 bf16, 2 fp16
__global__ __launch_bounds__(256) void stem_pack_kernel(const float* __restrict__ img,
                                                       void* __restrict__ out, int N, int H, int W,
                                                       int Wp) {
    const int Hp = H + 6;
    const long long total = (long long)N * Hp * Wp;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total;
         i += (long long)gridDim.x * blockDim.x) {
        const int wp = (int)(i % Wp);
        const int hp = (int)((i / Wp) % Hp);
        const int n = (int)(i / ((long long)Wp * Hp));
        const int h = hp - 3, w = wp - 3;
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (h >= 0 && h < H && w >= 0 && w < W) {
            const size_t b = ((size_t)n * 3 * H + h) * W + w;
            v.x = img[b]; v.y = img[b + (size_t)H * W]; v.z = img[b + 2 * (size_t)H * W];
        }
        if (BF16 == 2) {
            uint2 u;
            u.x = brcnn_pk2h(v.x, v.y);
            u.y = (unsigned)brcnn_f2h(v.z);
            reinterpret_cast<uint2*>(out)[i] = u;
        } else if (BF16) {
            uint2 u;
            u.x = brcnn_pk2b(v.x, v.y);
            u.y = (unsigned)brcnn_f2b(v.z);
            reinterpret_cast<uint2*>(out)[i] = u;
        } else {
            reinterpret_cast<float4*>(out)[i] = v;
        }
    }
}
}  // namespace

BRCNN_API size_t brcnn_stem_workspace_bytes(int batch, int height, int width) {
    return (size_t)batch * (height + 6) * (width + 14) * 4 * sizeof(float) + 256;
}

BRCNN_API int brcnn_stem7x7s2_nchw(const float* img, const void* w_packed, const float* scale,
                                   const float* shift, void* y, void* workspace, int batch,
                                   int height, int width, int cout, int relu, int dtype, void* stream) {
    if (!img || !w_packed || !y || !workspace || batch <= 0 || height < 7 || width < 7 || cout <= 0 ||
        !brcnn_elem_ok(dtype))
        return BRCNN_EINVAL;
    hipStream_t s = (hipStream_t)stream;
    const bool bf = dtype != BRCNN_DT_F32;
    const int kelems = bf ? 64 : 32;                 // K row = 128 bytes of 4-element pixels (16-bit: two filter rows of 8)
    const int Hp = height + 6, Wp = width + 6;
    const long long total = (long long)batch * Hp * Wp;
    long long g = (total + 255) / 256;
    if (g > 8192) g = 8192;
    if (dtype == BRCNN_DT_F16) hipLaunchKernelGGL(stem_pack_kernel<2>, dim3((int)g), dim3(256), 0, s, img, workspace, batch, height, width, Wp);
    else if (bf) hipLaunchKernelGGL(stem_pack_kernel<1>, dim3((int)g), dim3(256), 0, s, img, workspace, batch, height, width, Wp);
    else hipLaunchKernelGGL(stem_pack_kernel<0>, dim3((int)g), dim3(256), 0, s, img, workspace, batch, height, width, Wp);
    BRCNN_LAUNCH_CHECK();
    const int Ho = (height + 6 - 7) / 2 + 1, Wo = (width + 6 - 7) / 2 + 1;
    const int hs[1] = {Hp}, ws[1] = {Wp}, ohs[1] = {Ho}, ows[1] = {Wo};
    return conv_setup_and_launch(workspace, w_packed, scale, shift, nullptr, y, batch, 1, hs, ws, ohs,
                                 ows, kelems, cout, bf ? 4 : 7, 1, 2, 0, 1, relu, dtype, stream, 4, nullptr, nullptr, nullptr,
                                 bf ? 1 : 0);
}
