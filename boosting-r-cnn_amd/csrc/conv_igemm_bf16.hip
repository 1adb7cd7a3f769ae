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
#include <algorithm>
#include <mutex>
#include <vector>
#include "conv_common.h"

namespace {
using namespace brcnn_conv;

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef __attribute__((address_space(3))) void* lds_ptr_t;
constexpr int BKE = 64;     // K tile in elements (128 bytes)

__device__ __forceinline__ unsigned short f2bf(float v) { return brcnn_f2b(v); }
__device__ __forceinline__ float bf2f(unsigned short h) { return __uint_as_float(((unsigned)h) << 16); }
// ET: element type of the 16-bit operands / result, 0 = bf16, 1 = IEEE fp16 (v_mfma_f32_32x32x16_f16); the
// staging path moves bytes and is the same for both
template <int ET> __device__ __forceinline__ float e2f(unsigned short h) { return ET ? brcnn_h2f(h) : bf2f(h); }
template <int ET> __device__ __forceinline__ unsigned short f2e(float v) { return ET ? brcnn_f2h(v) : f2bf(v); }
template <int ET> __device__ __forceinline__ unsigned pk2e(float lo, float hi) { return ET ? brcnn_pk2h(lo, hi) : brcnn_pk2b(lo, hi); }

// WM x 2 waves, each MT x NT MFMA tiles: block tile (32*MT*WM) x (64*NT).  WM = 2: 4 waves,
// two workgroups per CU; WM = 4: 8 waves, 256-row tiles -- 1.5x the FLOPs per staged byte of the
// 128x128 tile, which is what the 64 B/clk/CU LDS-DMA path needs.
// SK: chained stream-K schedule (ConvParams::sk_*): the workgroup's item (tile, K range, hand-over slot) comes from
// the launch's item table.
template <int MT, int NT, bool RES, bool OUTF32, int WM, int WNW = 2, int ST = 2, int ET = 0, int MODE = 0, bool SK = false>
__global__ __launch_bounds__(64 * WM * WNW, (ST == 2 && MT * NT <= 4 && (WM * WNW == 4 || MT == 1)) ? 2 : 1) void conv_igemm_bf16_dma_kernel(ConvParams p) {
    constexpr int NW = WNW * WM;        // waves per workgroup (WM along M x WNW along N)
    constexpr int BM = 32 * MT * WM, BN = 32 * NT * WNW;
    constexpr int AG = BM / 8 / NW;     // 8-row groups of the A tile per wave
    constexpr int BG = BN / 8 / NW;
    static_assert(AG >= 1 && BG >= 1 && AG * 8 * NW == BM && BG * 8 * NW == BN, "tile / wave split");
    static_assert(!SK || ST == 2, "stream-K schedule: the two-buffer K loop");
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* As = smem;                   // [2][BM][32 dwords = 128 B]
    const int nk = p.K / BKE;
    const int nbuf = (SK || nk >= ST) ? ST : nk; // ring of ST K-tile buffers (fewer when the K loop is shorter)
    float* Bs = smem + nbuf * BM * 32;  // [nbuf][BN][32 dwords]

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave / WNW, wn = wave % WNW;
    const int li = lane & 31, lh = lane >> 5;

    const int nwg = p.tiles_m * p.tiles_n;
    int tile, kb = 0, ke = nk, sk_slot = 0;     // this work item: K tiles [kb, ke) of output tile `tile`
    if constexpr (SK) {
        const int4 item = p.sk_items[blockIdx.x];       // wave-uniform: scalar loads
        tile = __builtin_amdgcn_readfirstlane(item.x);
        kb = __builtin_amdgcn_readfirstlane(item.y);
        ke = __builtin_amdgcn_readfirstlane(item.z);
        sk_slot = __builtin_amdgcn_readfirstlane(item.w);
        if (tile < 0) return;                            // padding entry of the table
    } else {
        tile = xcd_remap(blockIdx.x, nwg);
    }
    const bool finish = !SK || ke == nk;        // this item ends with the epilogue (else: the partial tile is published)
    const int tile_m = tile / p.tiles_n, tile_n = tile - tile_m * p.tiles_n;
    const int m0 = tile_m * BM, n0 = tile_n * BN;

    const __amdgpu_buffer_rsrc_t rsrc_x = __builtin_amdgcn_make_buffer_rsrc((void*)p.x, 0, (int)p.x_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rsrc_w = __builtin_amdgcn_make_buffer_rsrc((void*)p.w, 0, (int)p.w_bytes, 0x00020000);

    const int rg = lane >> 3, pc = lane & 7;
    const bool plain = !p.no_fast && p.KH == 1 && p.KW == 1 && p.stride == 1 && p.pad == 0 && p.nseg == 1 && p.dilate <= 1;
    int a_base[AG], a_hw[AG], a_H[AG], a_W[AG], a_lc[AG];
    int b_off[BG];
#pragma unroll
    for (int j = 0; j < AG; j++) {
        const int r = (wave * AG + j) * 8 + rg;
        a_lc[j] = (pc ^ ((r >> 1) & 7)) * 8;          // logical k offset (elements) of this lane's chunk
        const int m = m0 + r;
        if (plain) {
            // a 1x1 stride-1 layer on one map reads row m of x for row m of the output: no map lookup, no divisions
            // (the short-K layers of the backbone are VALU-bound, and this set-up was a fifth of their instructions)
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
        b_off[j] = (co < p.Cout) ? co * p.K + (pc ^ ((r >> 1) & 7)) * 8 : -1;
    }

    f32x16 acc[MT][NT];
#pragma unroll
    for (int a = 0; a < MT; a++)
#pragma unroll
        for (int b = 0; b < NT; b++)
#pragma unroll
            for (int r = 0; r < 16; r++) acc[a][b][r] = 0.f;

    constexpr int PIECES = AG + BG;     // LDS-DMA instructions per wave and K tile
    struct TileK { int k0, ci0, kh, kw; };
    // K tiles are staged in order: the filter tap and channel offset of the next tile are carried
    // as scalar state instead of being re-derived by two integer divisions per tile
    // Order: channel chunk by channel chunk, the filter taps INSIDE a chunk (K tile kt = chunk kt / (KH KW), tap
    // kt % (KH KW)) -- the taps of a 3x3 filter then read shifted windows of the same 128-byte pieces of the same
    // pixels back to back and hit in L2 (conv_pp_f32.hip has the measurement); every 16-bit kernel uses this order,
    // so their results stay bit-identical to each other.  k0 = the tile's column in the [Cout][kh][kw][ci] weights.
    TileK d_next = {0, 0, 0, 0};
    if constexpr (SK) {
        if (kb > 0) {
            const int T = p.KH * p.KW, cc = kb / T, tap = kb - cc * T;
            d_next.ci0 = cc * BKE;
            d_next.kh = tap / p.KW;
            d_next.kw = tap - d_next.kh * p.KW;
            d_next.k0 = tap * p.Cin + d_next.ci0;
        }
    }
    auto dma_setup = [&](int kt) {
        (void)kt;
        const TileK t = d_next;
        if (++d_next.kw == p.KW) {
            d_next.kw = 0;
            if (++d_next.kh == p.KH) { d_next.kh = 0; d_next.ci0 += BKE; }
        }
        d_next.k0 = (d_next.kh * p.KW + d_next.kw) * p.Cin + d_next.ci0;
        return t;
    };
    // byte offsets of the pieces in K tile 0: the weight rows always, the input rows of a plain (1x1 stride-1, one map)
    // layer -- one add per piece and K tile instead of the tap / bounds arithmetic (an absent row starts at OOB and the sum
    // stays beyond the buffer's extent: x and w are < 2 GiB)
    unsigned a_off2[AG], b_off2[BG];
#pragma unroll
    for (int j = 0; j < AG; j++) a_off2[j] = a_base[j] >= 0 ? (unsigned)(a_base[j] + a_lc[j] + tile_n * p.gstep) * 2u : (unsigned)OOB;
#pragma unroll
    for (int j = 0; j < BG; j++) b_off2[j] = b_off[j] >= 0 ? (unsigned)b_off[j] * 2u : (unsigned)OOB;
    auto dma_piece = [&](const TileK& t, int buf, int j) {
        if (j < AG && plain) {
            float* dst = As + buf * BM * 32 + (wave * AG + j) * 8 * 32;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc_x, (lds_ptr_t)dst, 16, (int)(a_off2[j] + (unsigned)t.ci0 * 2u), 0, 0, 0);
        } else if (j < AG) {
            int hi = (a_hw[j] >> 16) - 4096 + t.kh;
            int wi = (a_hw[j] & 0xffff) - 4096 + t.kw;
            bool ok = a_base[j] >= 0;
            if (p.dilate > 1) {      // zero-stuffed input (data gradient of a strided conv)
                brcnn_undilate(p.dilate, hi, wi, ok);
            }
            int lc = a_lc[j];
            if (p.stem2) {          // (wave-uniform) the stem's K tile kh: image rows 2 kh and 2 kh + 1, 8 pixels of 4 elements each
                hi += t.kh + (lc >> 5);
                lc &= 31;
            }
            ok = ok & ((unsigned)hi < (unsigned)a_H[j]) & ((unsigned)wi < (unsigned)a_W[j]);
            const int off = ok ? (a_base[j] + (hi * a_W[j] + wi) * p.pitch + t.ci0 + lc + tile_n * p.gstep) * 2 : OOB;
            float* dst = As + buf * BM * 32 + (wave * AG + j) * 8 * 32;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc_x, (lds_ptr_t)dst, 16, off, 0, 0, 0);
        } else {
            const int jb = j - AG;
            float* dst = Bs + buf * BN * 32 + (wave * BG + jb) * 8 * 32;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc_w, (lds_ptr_t)dst, 16, (int)(b_off2[jb] + (unsigned)t.k0 * 2u), 0, 0, 0);
        }
    };
    auto dma_tile = [&](int kt, int buf) {
        const TileK t = dma_setup(kt);
#pragma unroll
        for (int j = 0; j < PIECES; j++) dma_piece(t, buf, j);
    };

    if (ST == 2) dma_tile(kb, 0);
    if constexpr (SK) {
        if (kb > 0) {
            // the K head of this tile: published by a workgroup of the launch's first round.  One lane polls
            // (relaxed, with a bound: a lost flag must not hang the device), one agent-scope acquire, then plain loads.
            if (tid == 0) {
                int spins = 0;
                while (__hip_atomic_load(p.sk_flags + sk_slot, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != p.sk_epoch &&
                       ++spins < p.sk_spin_limit)
                    __builtin_amdgcn_s_sleep(4);
                // a hand-over that never arrives must not end as a silent wrong result: the host-mapped error word
                // makes the next launch on any stream (and brcnn_conv_handover_status) return BRCNN_EHANDOVER
                if (spins >= p.sk_spin_limit)
                    __hip_atomic_store(p.sk_err, p.sk_epoch | 0x80000000u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
            }
            __syncthreads();
            const f32x4* src = reinterpret_cast<const f32x4*>(p.sk_ws) + (size_t)sk_slot * (BM * BN / 4) + wave * 64 + lane;
#pragma unroll
            for (int a = 0; a < MT; a++)
#pragma unroll
                for (int b = 0; b < NT; b++)
#pragma unroll
                    for (int g = 0; g < 4; g++) {
                        const f32x4 v = src[((a * NT + b) * 4 + g) * (NW * 64)];
                        acc[a][b][4 * g + 0] = v.x; acc[a][b][4 * g + 1] = v.y;
                        acc[a][b][4 * g + 2] = v.z; acc[a][b][4 * g + 3] = v.w;
                    }
        }
    }

    // residual rows (bf16, 16 B = 8 channels of one pixel per lane, the read-out mapping of the
    // epilogue), issued before the K loop so that their latency hides behind it
    constexpr int LPR = 4 * NT;                   // lanes per output row (8 channels each)
    constexpr int RPI = 64 / LPR;                 // rows per read-out iteration
    const int cw0 = n0 + wn * 32 * NT;            // first channel of this wave
    const bool vec_ok = (p.Cout & 7) == 0;
    const int rl = lane / LPR, cl = (lane % LPR) * 8;
    // (MODE 2 launches with RES: the "residual" rows prefetched here are the producer's z tile)
    const unsigned short* __restrict__ res = reinterpret_cast<const unsigned short*>(MODE == 2 ? p.tail_z : (const void*)p.residual);
    static_assert(!RES || MT * (32 / RPI) <= 16, "residual prefetch registers: use the non-residual form for 4x4 register tiles");
    uint4 rq[MT][32 / RPI];
    // MODE 0, tile inside the output: scale / shift of the lane's 8 read-out channels are requested here, in front of the
    // K loop, and applied behind the LDS transposition -- on the one- and two-tile K loops of the 1x1 layers the load was a
    // second full memory latency between the last MFMA and the first store
    const bool full_tile = !p.no_fast && vec_ok && !p.scatter && m0 + BM <= p.M && n0 + BN <= p.Cout;
    float4 pre_sc[2], pre_sh[2];
    pre_sc[0] = pre_sc[1] = make_float4(1.f, 1.f, 1.f, 1.f);
    pre_sh[0] = pre_sh[1] = make_float4(0.f, 0.f, 0.f, 0.f);
    auto load_pre = [&]() {
        if (p.scale) { pre_sc[0] = *reinterpret_cast<const float4*>(p.scale + cw0 + cl); pre_sc[1] = *reinterpret_cast<const float4*>(p.scale + cw0 + cl + 4); }
        if (p.shift) { pre_sh[0] = *reinterpret_cast<const float4*>(p.shift + cw0 + cl); pre_sh[1] = *reinterpret_cast<const float4*>(p.shift + cw0 + cl + 4); }
    };
    // (early only where registers are free: not on the 16-wave tiles with 128 VGPRs per lane, and not next to the residual
    // prefetch -- measured: with both, the residual layers lose more occupancy than the latency is worth, 89 -> 103 us)
    constexpr bool PRE_EARLY = WM * WNW < 16 && !RES;
    if constexpr (MODE == 0 && !OUTF32 && PRE_EARLY) {
        if (full_tile && finish) load_pre();
    }
    // MODE 3: two more tiles, the producer's z and its output (the ReLU mask); they are requested at the start of
    // the epilogue (behind the K loop, whose fragment registers they would otherwise compete with: 135 VGPRs and one
    // workgroup per CU instead of two) and arrive while the affine vectors are derived and the first slab is written
    uint4 rq2[MODE == 3 ? MT : 1][MODE == 3 ? 32 / RPI : 1], rq3[MODE == 3 ? MT : 1][MODE == 3 ? 32 / RPI : 1];
    if (RES) {
#pragma unroll
        for (int tm = 0; tm < MT; tm++)
#pragma unroll
            for (int it = 0; it < 32 / RPI; it++) {
                const int m = m0 + wm * 32 * MT + tm * 32 + it * RPI + rl, co = cw0 + cl;
                rq[tm][it] = make_uint4(0, 0, 0, 0);
                const bool ok = vec_ok && m < p.M && co < p.Cout && finish;
                if (ok) rq[tm][it] = *reinterpret_cast<const uint4*>(res + (size_t)m * p.Cout + co);
            }
    }
    if (ST == 2) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
    }

    // Fragment reads are inline asm: a compiler-visible LDS read after `buffer_load ... lds` gets
    // an `s_waitcnt vmcnt(0)` put in front of it by the waitcnt pass (it cannot tell the two LDS
    // buffers apart), which would serialise the prefetch of tile kt+1 with the MFMAs of tile kt
    // in every wave.  Opaque reads leave the explicit wait before the barrier as the only one.
    const int sw = (li >> 1) & 7;
    unsigned chb[BKE / 16];
#pragma unroll
    for (int kk = 0; kk < BKE / 16; kk++) chb[kk] = (unsigned)(((2 * kk + lh) ^ sw) * 16);
    const unsigned a_lane = (unsigned)(size_t)(lds_ptr_t)(As + (wm * 32 * MT + li) * 32);
    const unsigned b_lane = (unsigned)(size_t)(lds_ptr_t)(Bs + (wn * 32 * NT + li) * 32);
    f32x4 av[2][MT], bv[2][NT];
    auto lds_read = [&](f32x4& d, unsigned addr, int t) {      // t-th 32-row MFMA tile: +4096 B
        if (t == 0) asm volatile("ds_read_b128 %0, %1" : "=v"(d) : "v"(addr) : "memory");
        else if (t == 1) asm volatile("ds_read_b128 %0, %1 offset:4096" : "=v"(d) : "v"(addr) : "memory");
        else if (t == 2) asm volatile("ds_read_b128 %0, %1 offset:8192" : "=v"(d) : "v"(addr) : "memory");
        else asm volatile("ds_read_b128 %0, %1 offset:12288" : "=v"(d) : "v"(addr) : "memory");
    };
    auto frag_read = [&](int slot, unsigned a_addr, unsigned b_addr) {
#pragma unroll
        for (int t = 0; t < MT; t++) lds_read(av[slot][t], a_addr, t);
#pragma unroll
        for (int t = 0; t < NT; t++) lds_read(bv[slot][t], b_addr, t);
    };
    // the wait, then one empty asm per fragment register: volatile asms keep their order, so every
    // use of a fragment is placed after the wait
    auto frag_wait = [&](int slot) {
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
        for (int t = 0; t < MT; t++) asm volatile("" : "+v"(av[slot][t]));
#pragma unroll
        for (int t = 0; t < NT; t++) asm volatile("" : "+v"(bv[slot][t]));
    };
    static_assert(MT <= 4 && NT <= 4, "fragment readers cover up to 4 MFMA tiles per wave and axis");
    // `fill_kt` >= 0: the LDS-DMA pieces of that K tile go into ring slot `fill_slot`, spread
    // between the MFMA groups (p.il) so that their issue time overlaps MFMAs already queued in
    // the matrix pipe, or all in front of the tile's first fragment read.
    auto compute_tile = [&](int slot, int fill_kt, int fill_slot) {
        const bool issue = fill_kt >= 0;
        const bool spread = issue && p.il;
        TileK ft = {0, 0, 0, 0};
        if (issue) ft = dma_setup(fill_kt);
        const unsigned a_cur = a_lane + slot * (BM * 32 * 4);
        const unsigned b_cur = b_lane + slot * (BN * 32 * 4);
        __builtin_amdgcn_s_setprio(1);       // the prefetch issue wins over other waves' MFMA streams
        frag_read(0, a_cur + chb[0], b_cur + chb[0]);
        if (issue && !spread) {              // under the latency of the first fragment read
#pragma unroll
            for (int j = 0; j < PIECES; j++) dma_piece(ft, fill_slot, j);
        }
        frag_wait(0);
        __builtin_amdgcn_s_setprio(0);
#pragma unroll
        for (int kk = 0; kk < BKE / 16; kk++) {
            const int sl = kk & 1;
            if (kk + 1 < BKE / 16) frag_read(sl ^ 1, a_cur + chb[kk + 1], b_cur + chb[kk + 1]);
#pragma unroll
            for (int tm = 0; tm < MT; tm++)
#pragma unroll
                for (int t = 0; t < NT; t++)
                    if constexpr (ET)
                        acc[tm][t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(
                            __builtin_bit_cast(f16x8, bv[sl][t]), __builtin_bit_cast(f16x8, av[sl][tm]), acc[tm][t], 0, 0, 0);
                    else
                        acc[tm][t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(
                            __builtin_bit_cast(bf16x8, bv[sl][t]), __builtin_bit_cast(bf16x8, av[sl][tm]), acc[tm][t], 0, 0, 0);
            if (spread) {
#pragma unroll
                for (int j = 0; j < PIECES; j++)
                    if (j * (BKE / 16) / PIECES == kk) dma_piece(ft, fill_slot, j);
            }
            if (kk + 1 < BKE / 16) frag_wait(sl ^ 1);
        }
    };
    if constexpr (ST == 2) {
        int cur = 0;
        for (int kt = kb; kt < ke; kt++) {
            compute_tile(cur, kt + 1 < ke ? kt + 1 : -1, cur ^ 1);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
            cur ^= 1;
        }
    } else {
        // ST-deep ring: tiles kt+1 .. kt+ST-2 stay in flight while tile kt is multiplied.  One
        // counted wait (this wave's pieces of tile kt have landed) and one raw barrier (every
        // wave's pieces have, and nobody still reads slot kt-1) per K tile; slot kt-1 is refilled
        // with tile kt+ST-1 after the barrier.
#pragma unroll
        for (int t = 0; t < ST - 1; t++)
            if (t < nk) dma_tile(t, t);
        int slot = 0;
        for (int kt = 0; kt < nk; kt++) {
            if (nk - 1 - kt >= ST - 2) asm volatile("s_waitcnt vmcnt(%0)" ::"n"((ST - 2) * PIECES) : "memory");
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            asm volatile("s_barrier" ::: "memory");
            int fill = slot + ST - 1;
            if (fill >= ST) fill -= ST;
            compute_tile(slot, kt + ST - 1 < nk ? kt + ST - 1 : -1, fill);
            slot = slot + 1 == ST ? 0 : slot + 1;
        }
        asm volatile("s_barrier" ::: "memory");      // the epilogue slabs overwrite the ring
    }

    if constexpr (SK) {
        if (!finish) {
            // the K head of a tile that a later workgroup finishes: accumulators in register order (one coalesced
            // 1 KiB store per wave instruction), then publish: stores drained in every wave, one agent-scope release,
            // the flag (cdna_hip_programming.md 6 G16; the second wait restates the one behind buffer_wbl2)
            f32x4* dst = reinterpret_cast<f32x4*>(p.sk_ws) + (size_t)sk_slot * (BM * BN / 4) + wave * 64 + lane;
#pragma unroll
            for (int a = 0; a < MT; a++)
#pragma unroll
                for (int b = 0; b < NT; b++)
#pragma unroll
                    for (int g = 0; g < 4; g++) {
                        f32x4 v;
                        v.x = acc[a][b][4 * g + 0]; v.y = acc[a][b][4 * g + 1];
                        v.z = acc[a][b][4 * g + 2]; v.w = acc[a][b][4 * g + 3];
                        dst[((a * NT + b) * 4 + g) * (NW * 64)] = v;
                    }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
            if (tid == 0) {
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                if (!p.sk_drop_publish)       // (test hook: a lost hand-over)
                    __hip_atomic_store(p.sk_flags + sk_slot, p.sk_epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
            return;
        }
    }
    // ---- epilogue.  The MFMA ran as D^T = W A^T, so lane l holds pixel m = l&31 and, per
    // register group g, four consecutive channels co = 8g + 4(l>>5) + (0..3).  Scale/shift are
    // applied in the accumulators; each wave transposes one 32-row slab at a time through its
    // private LDS region (pitch 32*NT*4+16 B: ds_write_b128 / ds_read_b128 conflict-free) so
    // that a lane ends up with 8 consecutive channels of one pixel: the residual is read and
    // the result written as full 16-byte (bf16) / 32-byte (fp32) pieces of whole NHWC rows.
    constexpr int PITCH = 32 * NT + 4;            // floats
    float* cs = smem + wave * 32 * PITCH;
    if constexpr (MODE == 3) {
#pragma unroll
        for (int tm = 0; tm < MT; tm++)
#pragma unroll
            for (int it = 0; it < 32 / RPI; it++) {
                const int m = m0 + wm * 32 * MT + tm * 32 + it * RPI + rl, co = cw0 + cl;
                rq2[tm][it] = rq3[tm][it] = make_uint4(0, 0, 0, 0);
                if (vec_ok && m < p.M && co < p.Cout) {
                    rq2[tm][it] = *reinterpret_cast<const uint4*>(reinterpret_cast<const unsigned short*>(p.tail_z) + (size_t)m * p.Cout + co);
                    rq3[tm][it] = *reinterpret_cast<const uint4*>(reinterpret_cast<const unsigned short*>(p.tail_mask) + (size_t)m * p.Cout + co);
                }
            }
    }
    // DUAL (training): the raw tile leaves as z, the affine is applied in the read-out layout where a lane
    // owns the same 8 channels in every iteration.  The wave's 32*NT channels are derived one per lane
    // (correctly rounded divide / sqrt: ~100 instructions each) and handed out through the wave's slab.
    // MODE 2 (data gradient + the BatchNorm backward of the input's producer): the same per-lane channel
    // vectors; the lane also accumulates sum d and sum d*z of its 8 channels over its rows.
    constexpr bool DUAL = MODE == 1;
    constexpr bool dual = MODE != 0;
    const float* __restrict__ acc_scale = dual ? nullptr : p.scale;
    const float* __restrict__ acc_shift = dual ? nullptr : p.shift;
    float sc8[8], sh8[8];
    float sum_dz[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f}, sum_d[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    if constexpr (MODE != 0) {
        static_assert(NT <= 2, "one channel per lane");
        if (lane < 32 * NT) {
            const int co = cw0 + lane;
            float a = 1.f, b = 0.f;
            if (co < p.Cout) {
                if (p.bn_mean) {       // the operation order of bn_act.hip's bn_affine (the backward recomputes it)
                    a = p.scale[co] / sqrtf(p.bn_var[co] + p.bn_eps);
                    b = p.shift[co] - p.bn_mean[co] * a;
                } else {
                    a = p.scale ? p.scale[co] : 1.f;
                    b = p.shift ? p.shift[co] : 0.f;
                }
            }
            cs[lane] = a;
            cs[64 + lane] = b;
        }
        __builtin_amdgcn_s_waitcnt(0xc07f);
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int e = 0; e < 8; e++) { sc8[e] = cs[cl + e]; sh8[e] = cs[64 + cl + e]; }
        __builtin_amdgcn_s_waitcnt(0xc07f);
        __builtin_amdgcn_wave_barrier();
    }
    if constexpr (MODE <= 1 && !OUTF32) {
        // the block tile lies inside the output (wave-uniform): the straight-line read-out (conv_common.h)
        if (full_tile) {
            const size_t row0 = (size_t)(m0 + wm * 32 * MT + rl) * p.Cout + cw0 + cl;
            unsigned short* __restrict__ yrow = reinterpret_cast<unsigned short*>(p.y) + row0;
            unsigned short* __restrict__ zrow = reinterpret_cast<unsigned short*>(p.z_out) + row0;
            const unsigned floor2 = p.relu ? 0u : 0x80008000u;
            brcnn_f32x2 sc8p[4], sh8p[4];
            if constexpr (MODE == 0) {      // (the prefetched vectors; applied in the read-out layout: the same products and sums)
                if constexpr (!PRE_EARLY) load_pre();
                sc8p[0] = brcnn_f32x2{pre_sc[0].x, pre_sc[0].y}; sc8p[1] = brcnn_f32x2{pre_sc[0].z, pre_sc[0].w};
                sc8p[2] = brcnn_f32x2{pre_sc[1].x, pre_sc[1].y}; sc8p[3] = brcnn_f32x2{pre_sc[1].z, pre_sc[1].w};
                sh8p[0] = brcnn_f32x2{pre_sh[0].x, pre_sh[0].y}; sh8p[1] = brcnn_f32x2{pre_sh[0].z, pre_sh[0].w};
                sh8p[2] = brcnn_f32x2{pre_sh[1].x, pre_sh[1].y}; sh8p[3] = brcnn_f32x2{pre_sh[1].z, pre_sh[1].w};
            } else {
#pragma unroll
                for (int e = 0; e < 4; e++) { sc8p[e] = brcnn_f32x2{sc8[2 * e], sc8[2 * e + 1]}; sh8p[e] = brcnn_f32x2{sh8[2 * e], sh8[2 * e + 1]}; }
            }
#pragma unroll
            for (int tm = 0; tm < MT; tm++) {
#pragma unroll
                for (int tn = 0; tn < NT; tn++)
#pragma unroll
                    for (int g = 0; g < 4; g++) {
                        brcnn_f32x2 lo = {acc[tm][tn][4 * g + 0], acc[tm][tn][4 * g + 1]};
                        brcnn_f32x2 hi = {acc[tm][tn][4 * g + 2], acc[tm][tn][4 * g + 3]};
                        if constexpr (MODE != 0) {        // the general form's x * 1 + 0 (a -0 leaves as +0)
                            lo = lo + brcnn_f32x2{0.f, 0.f};
                            hi = hi + brcnn_f32x2{0.f, 0.f};
                        }
                        *reinterpret_cast<float4*>(cs + li * PITCH + tn * 32 + 8 * g + 4 * lh) = make_float4(lo.x, lo.y, hi.x, hi.y);
                    }
                __builtin_amdgcn_s_waitcnt(0xc07f);
                __builtin_amdgcn_wave_barrier();
#pragma unroll
                for (int it = 0; it < 32 / RPI; it++) {
                    const int row = it * RPI + rl;
                    const float4 lo = *reinterpret_cast<const float4*>(cs + row * PITCH + cl);
                    const float4 hi = *reinterpret_cast<const float4*>(cs + row * PITCH + cl + 4);
                    brcnn_f32x2 v[4] = {{lo.x, lo.y}, {lo.z, lo.w}, {hi.x, hi.y}, {hi.z, hi.w}};
                    const size_t off = (size_t)(tm * 32 + it * RPI) * p.Cout;
                    if constexpr (MODE == 0) {
#pragma unroll
                        for (int e = 0; e < 4; e++) v[e] = v[e] * sc8p[e] + sh8p[e];
                    }
                    if constexpr (DUAL) {
                        // the affine sees the STORED (rounded) z, as in the general form
                        const unsigned zw[4] = {brcnn_pk2<ET>(v[0]), brcnn_pk2<ET>(v[1]), brcnn_pk2<ET>(v[2]), brcnn_pk2<ET>(v[3])};
                        *reinterpret_cast<uint4*>(zrow + off) = make_uint4(zw[0], zw[1], zw[2], zw[3]);
#pragma unroll
                        for (int e = 0; e < 4; e++) v[e] = brcnn_unpk2<ET>(zw[e]) * sc8p[e] + sh8p[e];
                    }
                    if (RES) {
                        const unsigned rr[4] = {rq[tm][it].x, rq[tm][it].y, rq[tm][it].z, rq[tm][it].w};
#pragma unroll
                        for (int e = 0; e < 4; e++) v[e] += brcnn_unpk2<ET>(rr[e]);
                    }
                    uint4 o;
                    o.x = brcnn_relu_pk(brcnn_pk2<ET>(v[0]), floor2);
                    o.y = brcnn_relu_pk(brcnn_pk2<ET>(v[1]), floor2);
                    o.z = brcnn_relu_pk(brcnn_pk2<ET>(v[2]), floor2);
                    o.w = brcnn_relu_pk(brcnn_pk2<ET>(v[3]), floor2);
                    *reinterpret_cast<uint4*>(yrow + off) = o;
                }
                __builtin_amdgcn_wave_barrier();
            }
            return;
        }
    }
    bool done = false;
    if constexpr (MODE >= 2 && !OUTF32) {
        // ... and of the data-gradient launches that run a BatchNorm backward (the same arithmetic per element and the same
        // order of the per-channel sums as the general form below)
        if (full_tile) {
            const size_t row0 = (size_t)(m0 + wm * 32 * MT + rl) * p.Cout + cw0 + cl;
            unsigned short* __restrict__ yrow = reinterpret_cast<unsigned short*>(p.y) + row0;
            unsigned short* __restrict__ drow = reinterpret_cast<unsigned short*>(p.tail_dres) + row0;
            const unsigned floor2 = p.relu ? 0u : 0x80008000u;
            const bool mask_on = p.tail_relu != 0;
            brcnn_f32x2 sc8p[4], sh8p[4], sdz[4], sd[4];
#pragma unroll
            for (int e = 0; e < 4; e++) {
                sc8p[e] = brcnn_f32x2{sc8[2 * e], sc8[2 * e + 1]}; sh8p[e] = brcnn_f32x2{sh8[2 * e], sh8[2 * e + 1]};
                sdz[e] = brcnn_f32x2{0.f, 0.f}; sd[e] = brcnn_f32x2{0.f, 0.f};
            }
#pragma unroll
            for (int tm = 0; tm < MT; tm++) {
#pragma unroll
                for (int tn = 0; tn < NT; tn++)
#pragma unroll
                    for (int g = 0; g < 4; g++) {
                        const brcnn_f32x2 lo = brcnn_f32x2{acc[tm][tn][4 * g + 0], acc[tm][tn][4 * g + 1]} + brcnn_f32x2{0.f, 0.f};
                        const brcnn_f32x2 hi = brcnn_f32x2{acc[tm][tn][4 * g + 2], acc[tm][tn][4 * g + 3]} + brcnn_f32x2{0.f, 0.f};
                        *reinterpret_cast<float4*>(cs + li * PITCH + tn * 32 + 8 * g + 4 * lh) = make_float4(lo.x, lo.y, hi.x, hi.y);
                    }
                __builtin_amdgcn_s_waitcnt(0xc07f);
                __builtin_amdgcn_wave_barrier();
#pragma unroll
                for (int it = 0; it < 32 / RPI; it++) {
                    const int row = it * RPI + rl;
                    const float4 lo = *reinterpret_cast<const float4*>(cs + row * PITCH + cl);
                    const float4 hi = *reinterpret_cast<const float4*>(cs + row * PITCH + cl + 4);
                    brcnn_f32x2 v[4] = {{lo.x, lo.y}, {lo.z, lo.w}, {hi.x, hi.y}, {hi.z, hi.w}};
                    const size_t off = (size_t)(tm * 32 + it * RPI) * p.Cout;
                    if constexpr (MODE == 2) {
                        const unsigned zw[4] = {rq[tm][it].x, rq[tm][it].y, rq[tm][it].z, rq[tm][it].w};
#pragma unroll
                        for (int e = 0; e < 4; e++) {
                            const brcnn_f32x2 zz = brcnn_unpk2<ET>(zw[e]);
                            const brcnn_f32x2 g = brcnn_unpk2<ET>(brcnn_pk2<ET>(v[e]));
                            const brcnn_f32x2 pre = zz * sc8p[e] + sh8p[e];
                            brcnn_f32x2 d;
                            d.x = (!mask_on || pre.x > 0.f) ? g.x : 0.f;
                            d.y = (!mask_on || pre.y > 0.f) ? g.y : 0.f;
                            sdz[e] += d * zz;
                            sd[e] += d;
                            v[e] = d * sc8p[e];
                        }
                    } else {
                        const unsigned rr[4] = {rq[tm][it].x, rq[tm][it].y, rq[tm][it].z, rq[tm][it].w};
                        const unsigned zw[4] = {rq2[tm][it].x, rq2[tm][it].y, rq2[tm][it].z, rq2[tm][it].w};
                        const unsigned ow[4] = {rq3[tm][it].x, rq3[tm][it].y, rq3[tm][it].z, rq3[tm][it].w};
                        unsigned dq[4];
#pragma unroll
                        for (int e = 0; e < 4; e++) {
                            if (RES) v[e] += brcnn_unpk2<ET>(rr[e]);
                            const brcnn_f32x2 zz = brcnn_unpk2<ET>(zw[e]);
                            const brcnn_f32x2 oo = brcnn_unpk2<ET>(ow[e]);
                            const brcnn_f32x2 g = brcnn_unpk2<ET>(brcnn_pk2<ET>(v[e]));
                            brcnn_f32x2 d;
                            d.x = (!mask_on || oo.x > 0.f) ? g.x : 0.f;
                            d.y = (!mask_on || oo.y > 0.f) ? g.y : 0.f;
                            sdz[e] += d * zz;
                            sd[e] += d;
                            dq[e] = brcnn_pk2<ET>(d);
                            v[e] = d * sc8p[e];
                        }
                        *reinterpret_cast<uint4*>(drow + off) = make_uint4(dq[0], dq[1], dq[2], dq[3]);
                    }
                    uint4 o;
                    o.x = brcnn_relu_pk(brcnn_pk2<ET>(v[0]), floor2);
                    o.y = brcnn_relu_pk(brcnn_pk2<ET>(v[1]), floor2);
                    o.z = brcnn_relu_pk(brcnn_pk2<ET>(v[2]), floor2);
                    o.w = brcnn_relu_pk(brcnn_pk2<ET>(v[3]), floor2);
                    *reinterpret_cast<uint4*>(yrow + off) = o;
                }
                __builtin_amdgcn_wave_barrier();
            }
#pragma unroll
            for (int e = 0; e < 4; e++) {
                sum_dz[2 * e] = sdz[e].x; sum_dz[2 * e + 1] = sdz[e].y;
                sum_d[2 * e] = sd[e].x; sum_d[2 * e + 1] = sd[e].y;
            }
            done = true;
        }
    }
    if (!done) {
    float4 scv[NT][4], shv[NT][4];
#pragma unroll
    for (int tn = 0; tn < NT; tn++)
#pragma unroll
        for (int g = 0; g < 4; g++) {
            const int co = cw0 + tn * 32 + 8 * g + 4 * lh;
            float sc4[4], sh4[4];
#pragma unroll
            for (int e = 0; e < 4; e++) {
                const bool ok = co + e < p.Cout;
                sc4[e] = (acc_scale && ok) ? acc_scale[co + e] : 1.f;
                sh4[e] = (acc_shift && ok) ? acc_shift[co + e] : 0.f;
            }
            scv[tn][g] = make_float4(sc4[0], sc4[1], sc4[2], sc4[3]);
            shv[tn][g] = make_float4(sh4[0], sh4[1], sh4[2], sh4[3]);
        }
    unsigned short* __restrict__ yh = reinterpret_cast<unsigned short*>(p.y);
    float* __restrict__ yf = p.y;
#pragma unroll
    for (int tm = 0; tm < MT; tm++) {
        const int mw = m0 + wm * 32 * MT + tm * 32;
#pragma unroll
        for (int tn = 0; tn < NT; tn++)
#pragma unroll
            for (int g = 0; g < 4; g++) {
                float4 v;
                v.x = acc[tm][tn][4 * g + 0] * scv[tn][g].x + shv[tn][g].x;
                v.y = acc[tm][tn][4 * g + 1] * scv[tn][g].y + shv[tn][g].y;
                v.z = acc[tm][tn][4 * g + 2] * scv[tn][g].z + shv[tn][g].z;
                v.w = acc[tm][tn][4 * g + 3] * scv[tn][g].w + shv[tn][g].w;
                *reinterpret_cast<float4*>(cs + li * PITCH + tn * 32 + 8 * g + 4 * lh) = v;
            }
        __builtin_amdgcn_s_waitcnt(0xc07f);      // lgkmcnt(0): the slab is wave-private
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int it = 0; it < 32 / RPI; it++) {
            const int row = it * RPI + rl;
            const int m = mw + row, co = cw0 + cl;
            const float4 lo = *reinterpret_cast<const float4*>(cs + row * PITCH + cl);
            const float4 hi = *reinterpret_cast<const float4*>(cs + row * PITCH + cl + 4);
            float v[8] = {lo.x, lo.y, lo.z, lo.w, hi.x, hi.y, hi.z, hi.w};
            if (m >= p.M || co >= p.Cout) continue;
            const long long ro = out_row_offset(p, m);
            if (ro < 0) continue;
            if (vec_ok) {
                if constexpr (DUAL) {
                    uint4 zq;
                    zq.x = pk2e<ET>(v[0], v[1]);
                    zq.y = pk2e<ET>(v[2], v[3]);
                    zq.z = pk2e<ET>(v[4], v[5]);
                    zq.w = pk2e<ET>(v[6], v[7]);
                    *reinterpret_cast<uint4*>(reinterpret_cast<unsigned short*>(p.z_out) + ro + co) = zq;
                    // the affine sees the STORED (rounded) z: bit-identical to the conv kernel followed by
                    // bn_act_fwd_kernel, and the mask the backward recomputes from z is the forward's own
                    const unsigned zw[4] = {zq.x, zq.y, zq.z, zq.w};
#pragma unroll
                    for (int e = 0; e < 4; e++) {
                        v[2 * e] = e2f<ET>((unsigned short)(zw[e] & 0xffffu)) * sc8[2 * e] + sh8[2 * e];
                        v[2 * e + 1] = e2f<ET>((unsigned short)(zw[e] >> 16)) * sc8[2 * e + 1] + sh8[2 * e + 1];
                    }
                }
                if constexpr (MODE == 2) {
                    // d = the data gradient as the two-kernel path stores it (rounded), masked by the producer's
                    // ReLU; out = d * scale (bn_act_bwd_kernel's arithmetic)
                    const unsigned zw[4] = {rq[tm][it].x, rq[tm][it].y, rq[tm][it].z, rq[tm][it].w};
#pragma unroll
                    for (int e = 0; e < 8; e++) {
                        const float zz = e2f<ET>((unsigned short)((e & 1) ? (zw[e >> 1] >> 16) : (zw[e >> 1] & 0xffffu)));
                        const float g = e2f<ET>(f2e<ET>(v[e]));
                        const float pre = zz * sc8[e] + sh8[e];
                        const float d = (!p.tail_relu || pre > 0.f) ? g : 0.f;
                        sum_dz[e] += d * zz;
                        sum_d[e] += d;
                        v[e] = d * sc8[e];
                    }
                }
                if (RES && MODE != 2) {
                    const unsigned rr[4] = {rq[tm][it].x, rq[tm][it].y, rq[tm][it].z, rq[tm][it].w};
#pragma unroll
                    for (int e = 0; e < 4; e++) {
                        v[2 * e] += e2f<ET>((unsigned short)(rr[e] & 0xffffu));
                        v[2 * e + 1] += e2f<ET>((unsigned short)(rr[e] >> 16));
                    }
                }
                if constexpr (MODE == 3) {
                    // v = data gradient + identity gradient = d(previous block's output), rounded as the plain launch
                    // stores it; masked by that output's ReLU it is the identity gradient of the previous block
                    // (tail_dres) and, times scale, the gradient of its conv3 output z (bn_act_bwd_kernel's arithmetic)
                    const unsigned zw[4] = {rq2[tm][it].x, rq2[tm][it].y, rq2[tm][it].z, rq2[tm][it].w};
                    const unsigned ow[4] = {rq3[tm][it].x, rq3[tm][it].y, rq3[tm][it].z, rq3[tm][it].w};
                    float d8[8];
#pragma unroll
                    for (int e = 0; e < 8; e++) {
                        const float zz = e2f<ET>((unsigned short)((e & 1) ? (zw[e >> 1] >> 16) : (zw[e >> 1] & 0xffffu)));
                        const float oo = e2f<ET>((unsigned short)((e & 1) ? (ow[e >> 1] >> 16) : (ow[e >> 1] & 0xffffu)));
                        const float g = e2f<ET>(f2e<ET>(v[e]));
                        const float d = (!p.tail_relu || oo > 0.f) ? g : 0.f;
                        sum_dz[e] += d * zz;
                        sum_d[e] += d;
                        d8[e] = d;
                        v[e] = d * sc8[e];
                    }
                    uint4 dq;
                    dq.x = pk2e<ET>(d8[0], d8[1]);
                    dq.y = pk2e<ET>(d8[2], d8[3]);
                    dq.z = pk2e<ET>(d8[4], d8[5]);
                    dq.w = pk2e<ET>(d8[6], d8[7]);
                    *reinterpret_cast<uint4*>(reinterpret_cast<unsigned short*>(p.tail_dres) + ro + co) = dq;
                }
                if (p.relu) {
#pragma unroll
                    for (int e = 0; e < 8; e++) v[e] = fmaxf(v[e], 0.f);
                }
                if (OUTF32) {
                    float4* dst = reinterpret_cast<float4*>(yf + ro + co);
                    dst[0] = make_float4(v[0], v[1], v[2], v[3]);
                    dst[1] = make_float4(v[4], v[5], v[6], v[7]);
                } else {
                    uint4 o;
                    o.x = pk2e<ET>(v[0], v[1]);
                    o.y = pk2e<ET>(v[2], v[3]);
                    o.z = pk2e<ET>(v[4], v[5]);
                    o.w = pk2e<ET>(v[6], v[7]);
                    *reinterpret_cast<uint4*>(yh + ro + co) = o;
                }
            } else {       // ragged channel count (fused heads: 54, 21): element-wise tail
#pragma unroll
                for (int e = 0; e < 8; e++) {
                    if (co + e >= p.Cout) break;
                    float t = v[e];
                    if (RES) t += e2f<ET>(res[(size_t)m * p.Cout + co + e]);
                    if (p.relu) t = fmaxf(t, 0.f);
                    if (OUTF32) yf[ro + co + e] = t;
                    else yh[ro + co + e] = f2e<ET>(t);
                }
            }
        }
        __builtin_amdgcn_wave_barrier();
    }
    }   // !done
    if constexpr (MODE >= 2) {
        // lanes that share the channel vector (equal lane % LPR) hold different rows: butterfly over the
        // row bits, then the WM waves of one channel range add up through their slabs in a fixed order
#pragma unroll
        for (int d = LPR; d < 64; d <<= 1)
#pragma unroll
            for (int e = 0; e < 8; e++) {
                sum_dz[e] += __shfl_xor(sum_dz[e], d, 64);
                sum_d[e] += __shfl_xor(sum_d[e], d, 64);
            }
        if (lane < LPR) {
#pragma unroll
            for (int e = 0; e < 8; e++) { cs[lane * 8 + e] = sum_dz[e]; cs[64 + lane * 8 + e] = sum_d[e]; }
        }
        __syncthreads();
        if (wm == 0 && lane < 32 * NT) {
            const int co = cw0 + lane;
            float a = 0.f, b = 0.f;
#pragma unroll
            for (int k = 0; k < WM; k++) {
                const float* o = smem + (k * WNW + wn) * 32 * PITCH;
                a += o[lane];
                b += o[64 + lane];
            }
            if (co < p.Cout) {
                p.tail_partials[((size_t)tile_m * 2 + 0) * p.Cout + co] = a;
                p.tail_partials[((size_t)tile_m * 2 + 1) * p.Cout + co] = b;
            }
        }
    }
}

// ---- chained stream-K schedule: host side ---------------------------------------------------------------------
// Per (device, stream): the hand-over slots (fp32 accumulators of one tile each), their flags and the launch epoch.
// Launches on one stream are serialised, so a slot is free again when the next launch starts.  The device is part of
// the key: the default stream is handle 0 on EVERY device, and a workspace lives in one device's memory.
struct SkStream { hipStream_t stream; float* ws; unsigned* flags; unsigned epoch; float* slabs; char* base; int owned; int device; };
struct SkTable { long long tiles; int nk, slots, blocks; int4* items; int par; };
constexpr size_t SK_WS_BYTES = (size_t)128 << 20;
constexpr int SK_MAX_SLOTS = 2048;
constexpr size_t SK_FLAG_BYTES = (size_t)64 << 10;          // SK_MAX_SLOTS words, padded
constexpr size_t WGRAD_SLAB_BYTES = (size_t)160 << 20;      // == WGRAD_WS_BYTES of conv_wgrad_bf16.hip
constexpr size_t CONV_WS_BYTES = SK_WS_BYTES + SK_FLAG_BYTES + WGRAD_SLAB_BYTES;
std::mutex g_sk_mutex;
std::vector<SkStream> g_sk_streams;
std::vector<SkTable> g_sk_tables;
// hand-over time-outs: ONE host-mapped word every K tail can write (system scope) and every launch wrapper reads
unsigned* g_sk_err_host = nullptr;
unsigned* g_sk_err_dev = nullptr;
int g_sk_spin_limit = 1 << 24;       // ~5 s of polling; test hook -11 / -12: 256 polls and heads that do not publish / back
int g_sk_drop_publish = 0;
int g_sk_mode = 1;      // tuning hook (set_tile_bf16(-3 / -4 / -5)): 0 off, 1 heuristic, 2 wherever the tile count allows
int g_num_cus = 0;

static int sk_error_word() {
    if (g_sk_err_host) return 0;
    BRCNN_HIP_CHECK(hipHostMalloc((void**)&g_sk_err_host, 64, hipHostMallocMapped | hipHostMallocCoherent));
    *g_sk_err_host = 0;
    BRCNN_HIP_CHECK(hipHostGetDevicePointer((void**)&g_sk_err_dev, g_sk_err_host, 0));
    return 0;
}

// 0, or BRCNN_EHANDOVER once per reported time-out (the word is cleared: the caller has been told)
static int sk_take_error() {
    if (!g_sk_err_host) return 0;
    const unsigned v = __atomic_exchange_n(g_sk_err_host, 0u, __ATOMIC_RELAXED);
    return v ? BRCNN_EHANDOVER : 0;
}

// The scratch of one stream's conv launches: [stream-K accumulator slots | their flags | weight-gradient slabs].
// The CALLER provides it (brcnn_conv_set_workspace: SURVEY 8b "caller owns all buffers"); a stream without one gets a
// library allocation on first use (fallback for plain C callers that never registered; released by
// brcnn_conv_set_workspace(stream, NULL, 0)).  g_sk_mutex held by the caller.
static void sk_carve(SkStream& e, char* base) {
    e.base = base;
    e.ws = reinterpret_cast<float*>(base);
    e.flags = reinterpret_cast<unsigned*>(base + SK_WS_BYTES);
    e.slabs = reinterpret_cast<float*>(base + SK_WS_BYTES + SK_FLAG_BYTES);
}

int sk_stream_state(hipStream_t s, SkStream** out) {
    if (int rc = sk_error_word()) return rc;
    int dev = 0;
    BRCNN_HIP_CHECK(hipGetDevice(&dev));        // the device the caller is about to launch on
    for (auto& e : g_sk_streams)
        if (e.stream == s && e.device == dev) { *out = &e; return 0; }
    if (g_sk_streams.capacity() < 64) g_sk_streams.reserve(64);       // pointers handed out stay valid
    if (g_sk_streams.size() >= 64) return BRCNN_EINVAL;
    SkStream e = {s, nullptr, nullptr, 0, nullptr, nullptr, 1, dev};
    char* base = nullptr;
    BRCNN_HIP_CHECK(hipMalloc((void**)&base, CONV_WS_BYTES));
    sk_carve(e, base);
    BRCNN_HIP_CHECK(hipMemsetAsync(e.flags, 0, SK_FLAG_BYTES, s));    // ordered before the stream's first stream-K launch
    g_sk_streams.push_back(e);
    *out = &g_sk_streams.back();
    return 0;
}

}  // namespace

namespace brcnn_conv {
float* conv_ws_wgrad_slabs(hipStream_t s) {
    std::lock_guard<std::mutex> lock(g_sk_mutex);
    SkStream* st = nullptr;
    if (sk_stream_state(s, &st)) return nullptr;
    return st->slabs;
}
// arrival counters of the weight gradient's in-launch slab reduction: the upper half of the (zero-initialised) flag area,
// 8192 words; every counter is back at zero when the launch that used it ends (the last arriver resets it)
unsigned* conv_ws_wgrad_counters(hipStream_t s) {
    std::lock_guard<std::mutex> lock(g_sk_mutex);
    SkStream* st = nullptr;
    if (sk_stream_state(s, &st)) return nullptr;
    return st->flags + 8192;
}
}  // namespace brcnn_conv

BRCNN_API size_t brcnn_conv_workspace_bytes(void) { return CONV_WS_BYTES; }

BRCNN_API int brcnn_conv_set_workspace(void* stream, void* workspace, size_t bytes) {
    hipStream_t s = (hipStream_t)stream;
    std::lock_guard<std::mutex> lock(g_sk_mutex);
    if (int rc = sk_error_word()) return rc;
    if (workspace != nullptr && (bytes < CONV_WS_BYTES || ((uintptr_t)workspace & 255))) return BRCNN_EINVAL;
    int dev = 0;
    BRCNN_HIP_CHECK(hipGetDevice(&dev));        // (the current device: the one `workspace` was allocated on)
    SkStream* e = nullptr;
    for (auto& c : g_sk_streams)
        if (c.stream == s && c.device == dev) e = &c;
    if (workspace == nullptr) {             // release: the stream falls back to a library allocation on its next use
        if (e) {
            if (e->owned && e->base) {
                BRCNN_HIP_CHECK(hipStreamSynchronize(s));
                BRCNN_HIP_CHECK(hipFree(e->base));
            }
            const unsigned epoch = e->epoch;
            *e = g_sk_streams.back();
            g_sk_streams.pop_back();
            (void)epoch;
        }
        return 0;
    }
    if (e == nullptr) {
        if (g_sk_streams.capacity() < 64) g_sk_streams.reserve(64);
        if (g_sk_streams.size() >= 64) return BRCNN_EINVAL;
        g_sk_streams.push_back({s, nullptr, nullptr, 0, nullptr, nullptr, 0, dev});
        e = &g_sk_streams.back();
    } else if (e->owned && e->base) {
        BRCNN_HIP_CHECK(hipStreamSynchronize(s));
        BRCNN_HIP_CHECK(hipFree(e->base));
    }
    e->owned = 0;
    sk_carve(*e, (char*)workspace);
    BRCNN_HIP_CHECK(hipMemsetAsync(e->flags, 0, SK_FLAG_BYTES, s));
    return 0;
}

namespace {

// The item table of (tiles, nk, slots).  The iteration space tiles x nk is cut into `slots` equal ranges (what a
// persistent stream-K workgroup would walk, backwards: the K head of its last tile first, whole tiles, the K tail of
// its first tile last); the items are then laid out for the hardware's in-order dispatch of an ordinary grid (block
// b goes to XCD b % 8, each XCD starts its blocks in order as slots free up): per XCD, all K heads first, shortest
// first; the whole tiles; the K tails last, longest first -- so the slot that finished the shortest head picks up the
// longest tail and every slot ends up with one range's worth of work.  A head and its tail stay on the XCD that owns
// the tile (xcd_remap's contiguous chunks), so the operand rows and the hand-over stay in one L2.
int sk_table(long long tiles, int nk, int slots, const SkTable** out) {
    for (auto& t : g_sk_tables)
        if (t.tiles == tiles && t.nk == nk && t.slots == slots && !t.par) { *out = &t; return 0; }
    struct Item { int tile, kb, ke, slot; };
    std::vector<Item> heads[8], wholes[8], tails[8];
    std::vector<int> slot_of((size_t)tiles, -1);
    const long long iters = tiles * nk, q = iters / slots, r = iters % slots;
    int nslots = 0;
    const int cq = slots >> 3, cr = slots & 7;
    for (int w = 0; w < slots; w++) {
        // the XCD whose chunk of logical workgroups holds w (xcd_remap)
        int x = 0;
        for (int base = 0; x < 8; x++) {
            const int cnt = cq + (x < cr ? 1 : 0);
            if (w < base + cnt) break;
            base += cnt;
        }
        const long long sw = q * w + (w < r ? w : r), ew = sw + q + (w < r ? 1 : 0);
        long long end = ew;
        while (end > sw) {
            const long long tile = (end - 1) / nk, t0 = tile * nk;
            const int kb = (int)((sw > t0 ? sw : t0) - t0), ke = (int)(end - t0);
            Item it = {(int)tile, kb, ke, 0};
            if (kb > 0 && ke < nk) return BRCNN_EINVAL;         // three-way split: the caller requires tiles >= slots
            if (kb > 0 || ke < nk) {
                if (slot_of[tile] < 0) slot_of[tile] = nslots++;
                it.slot = slot_of[tile];
                (kb > 0 ? tails : heads)[x].push_back(it);
            } else {
                wholes[x].push_back(it);
            }
            end = t0 + kb;
        }
    }
    if (nslots > SK_MAX_SLOTS) return BRCNN_EINVAL;
    size_t maxlen = 0;
    std::vector<Item> queue[8];
    for (int x = 0; x < 8; x++) {
        std::stable_sort(heads[x].begin(), heads[x].end(), [](const Item& a, const Item& b) { return a.ke - a.kb < b.ke - b.kb; });
        std::stable_sort(wholes[x].begin(), wholes[x].end(), [](const Item& a, const Item& b) { return a.tile < b.tile; });
        std::stable_sort(tails[x].begin(), tails[x].end(), [](const Item& a, const Item& b) { return a.ke - a.kb > b.ke - b.kb; });
        queue[x] = heads[x];
        queue[x].insert(queue[x].end(), wholes[x].begin(), wholes[x].end());
        queue[x].insert(queue[x].end(), tails[x].begin(), tails[x].end());
        if (queue[x].size() > maxlen) maxlen = queue[x].size();
    }
    std::vector<int4> host(maxlen * 8);
    for (size_t loc = 0; loc < maxlen; loc++)
        for (int x = 0; x < 8; x++) {
            int4 v = make_int4(-1, 0, 0, 0);
            if (loc < queue[x].size()) v = make_int4(queue[x][loc].tile, queue[x][loc].kb, queue[x][loc].ke, queue[x][loc].slot);
            host[loc * 8 + x] = v;
        }
    SkTable t = {tiles, nk, slots, (int)(maxlen * 8), nullptr, 0};
    BRCNN_HIP_CHECK(hipMalloc((void**)&t.items, host.size() * sizeof(int4)));
    BRCNN_HIP_CHECK(hipMemcpy(t.items, host.data(), host.size() * sizeof(int4), hipMemcpyHostToDevice));
    if (g_sk_tables.capacity() < 256) g_sk_tables.reserve(256);
    if (g_sk_tables.size() >= 256) { (void)hipFree(t.items); return BRCNN_EINVAL; }
    g_sk_tables.push_back(t);
    *out = &g_sk_tables.back();
    return 0;
}

// Split-K table for launches with FEWER tiles than resident workgroups (the 8 x 25 x 42 maps of stage 4 / P4-P6 at
// batch 8: 66-132 tiles of the eight-phase kernels on 256 CUs).  The chained hand-over cannot help there -- a tile's
// K loop stays one serial chain --, so the pieces of a tile run side by side from zero and are summed at the end:
// tiles x nk is cut into `slots` equal ranges (boundaries one K tile away from a tile edge snap onto it); a range
// holds the END of one tile and / or the START or a MIDDLE of the next.  Every piece but the one holding the tile's
// last K tile stores its accumulators to ws[range]; that last piece (w = range | n_prev << 16 | 1 << 30) adds the
// n_prev ranges before its own, in K order.  Deterministic, but not the unsplit chain's bits: the launchers take this
// table only where the heuristic (or the forcing hook) says so, and `brcnn_conv_set_tile_bf16(-8)` switches it off.
// Layout: all storing pieces first (longest first), then the finishing ones (longest first); block b goes to XCD
// b % 8 and every XCD starts its blocks in order, so no finishing piece can occupy a CU before every storing piece of
// its XCD has been started.
int sk_table_par(long long tiles, int nk, int slots, const SkTable** out) {
    for (auto& t : g_sk_tables)
        if (t.tiles == tiles && t.nk == nk && t.slots == slots && t.par) { *out = &t; return 0; }
    struct Item { int tile, kb, ke, w; };
    std::vector<Item> storing, finishing;
    const long long iters = tiles * nk;
    std::vector<long long> bound((size_t)slots + 1);
    for (int r = 0; r <= slots; r++) {
        long long b = iters * r / slots;
        const long long rem = b % nk;
        if (rem == 1) b -= 1;
        else if (rem == nk - 1) b += 1;
        bound[r] = b;
    }
    std::vector<int> first_range((size_t)tiles, -1);
    for (int r = 0; r < slots; r++) {
        long long sw = bound[r];
        const long long ew = bound[r + 1];
        while (sw < ew) {
            const long long tile = sw / nk, t0 = tile * nk;
            const long long pe = (ew < t0 + nk) ? ew : t0 + nk;
            const int kb = (int)(sw - t0), ke = (int)(pe - t0);
            if (first_range[tile] < 0) first_range[tile] = r;
            if (ke == nk) {
                const int n_prev = r - first_range[tile];
                if (n_prev > 255) return BRCNN_EINVAL;
                finishing.push_back({(int)tile, kb, ke, r | (n_prev << 16) | (1 << 30)});
            } else {
                storing.push_back({(int)tile, kb, ke, r | (1 << 30)});
            }
            sw = pe;
        }
    }
    auto longer = [](const Item& a, const Item& b) { return a.ke - a.kb > b.ke - b.kb; };
    std::stable_sort(storing.begin(), storing.end(), longer);
    std::stable_sort(finishing.begin(), finishing.end(), longer);
    std::vector<int4> host;
    for (auto& it : storing) host.push_back(make_int4(it.tile, it.kb, it.ke, it.w));
    while (host.size() % 8) host.push_back(make_int4(-1, 0, 0, 0));       // the finishing pieces start on a fresh round of XCDs
    for (auto& it : finishing) host.push_back(make_int4(it.tile, it.kb, it.ke, it.w));
    SkTable t = {tiles, nk, slots, (int)host.size(), nullptr, 1};
    BRCNN_HIP_CHECK(hipMalloc((void**)&t.items, host.size() * sizeof(int4)));
    BRCNN_HIP_CHECK(hipMemcpy(t.items, host.data(), host.size() * sizeof(int4), hipMemcpyHostToDevice));
    if (g_sk_tables.capacity() < 256) g_sk_tables.reserve(256);
    if (g_sk_tables.size() >= 256) { (void)hipFree(t.items); return BRCNN_EINVAL; }
    g_sk_tables.push_back(t);
    *out = &g_sk_tables.back();
    return 0;
}

// OFF by default: it is the one schedule whose result is not the unsplit chain's bits, so with it the value of a conv
// would depend on the tile count -- i.e. on the batch size (the batched and the per-image paths stop agreeing bit for
// bit, tests/test_golden_gpu.py) -- for +0.7 % on the fp32 inference pass (25.20 -> 25.02 ms; stage-4 3x3 402 -> 369 us,
// first FC 476 -> 413 us, 2048 -> 512 1x1 184 -> 166 us); bf16 gains nothing (tools/experiments/splitk_try.py).
int g_sk_par = 0;       // tuning hook (set_tile_bf16(-8 / -9 / -10)): split-K of few-tile launches off / heuristic / forced

// the schedule of one launch, or sk_wgs = 0: `slots` = resident workgroups of this kernel on the whole device;
// `min_nk`: shortest K loop (in K tiles) the heuristic cuts for this tile shape (0: the 128 x 128 rule below)
static int sk_plan(ConvParams& p, int slots, int bm, int bn, hipStream_t s, int min_nk = 0, int bke = BKE, double max_eff = 0.9) {
    p.sk_wgs = 0;
    if (int e = sk_take_error()) return e;          // a K tail of an EARLIER launch gave up waiting for its head
    if (g_sk_mode == 0 || slots <= 0 || slots > SK_MAX_SLOTS) return 0;
    {
        // under stream capture (brcnn/graphs.py: the trunk's forward / backward as HIP graphs) the launch is replayed with
        // the epoch baked into its arguments, and a flag left behind by the previous replay would pass for this one's:
        // captured launches take the plain schedule (same bits: the chained form only moves work between workgroups)
        hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
        if (hipStreamIsCapturing(s, &cs) == hipSuccess && cs != hipStreamCaptureStatusNone) return 0;
    }
    const long long tiles = (long long)p.tiles_m * p.tiles_n;
    const int nk = p.K / bke;
    if (nk < 2 || tiles * nk >= 0x7fffffffLL) return 0;
    if ((size_t)(slots + 8) * bm * bn * sizeof(float) > SK_WS_BYTES) return 0;
    if (tiles < slots) {
        // fewer tiles than resident workgroups: split-K with a sum at the end (eight-phase kernels only: min_nk > 0),
        // where at least a fifth of the device would idle and every piece keeps >= min_nk / 2 K tiles
        if (min_nk <= 0 || g_sk_par == 0) return 0;
        if (g_sk_par == 1 && (tiles * 5 > (long long)slots * 4 || tiles * nk / slots < min_nk / 2 || tiles * nk / slots < 4)) return 0;
        if (tiles * nk / slots < 2) return 0;
        std::lock_guard<std::mutex> lock(g_sk_mutex);
        SkStream* st = nullptr;
        int rc = sk_stream_state(s, &st);
        if (rc) return rc;
        const SkTable* tab = nullptr;
        rc = sk_table_par(tiles, nk, slots, &tab);
        if (rc) return rc == BRCNN_EINVAL ? 0 : rc;
        if (++st->epoch == 0) st->epoch = 1;
        p.sk_wgs = tab->blocks;
        p.sk_items = tab->items;
        p.sk_ws = st->ws;
        p.sk_flags = st->flags;
        p.sk_epoch = st->epoch;
        p.sk_err = g_sk_err_dev;
        p.sk_spin_limit = g_sk_spin_limit;
        p.sk_drop_publish = g_sk_drop_publish;
        return 0;
    }
    // every slot must own at least one whole tile's worth of iterations: a tile then straddles two ranges at most
    if (g_sk_mode == 1) {
        // where it pays (tools/conv_bench_bf16.py, profiles/r03_notes.md): the 128 x 128 tile with a last generation of
        // workgroups that leaves most of the device idle and a K loop of >= 32 tiles -- every slot pays one hand-over
        // (a tile of fp32 accumulators each way, a second prologue), about a tenth of a tile's time; the smaller tiles
        // run three to five workgroups per CU, whose last generation speeds up by itself when its neighbours are gone
        const double gens = (double)tiles / slots;
        const double eff = gens / (double)(long long)(gens + 0.999999);
        if (min_nk > 0) {
            if (eff >= max_eff || nk < min_nk) return 0;
        } else if (eff >= 0.9 || nk < 32 || bm != 128 || bn != 128) {
            return 0;
        }
    }
    std::lock_guard<std::mutex> lock(g_sk_mutex);
    SkStream* st = nullptr;
    int rc = sk_stream_state(s, &st);
    if (rc) return rc;
    const SkTable* tab = nullptr;
    rc = sk_table(tiles, nk, slots, &tab);
    if (rc) return rc == BRCNN_EINVAL ? 0 : rc;
    if (++st->epoch == 0) st->epoch = 1;
    p.sk_wgs = tab->blocks;
    p.sk_items = tab->items;
    p.sk_ws = st->ws;
    p.sk_flags = st->flags;
    p.sk_epoch = st->epoch;
    p.sk_err = g_sk_err_dev;
    p.sk_spin_limit = g_sk_spin_limit;
    p.sk_drop_publish = g_sk_drop_publish;
    return 0;
}

template <int MT, int NT, bool RES, bool OUTF32, int WM = 2, int WNW = 2, int ST = 2, int ET = 0, int MODE = 0>
int launch(ConvParams& p, hipStream_t s) {
    const size_t lds_stage = (size_t)(32 * MT * WM + 32 * NT * WNW) * 32 * sizeof(float);
    const size_t lds_full = ST * lds_stage;
    const size_t lds_epi = (size_t)WNW * WM * 32 * (32 * NT + 4) * sizeof(float);
    const int nk = p.K / BKE;
    if constexpr (ST == 2 && MT * NT <= 4) {
        // persistent launch of exactly the resident workgroups (the occupancy query can be one high near an SGPR
        // edge -- MI355X_MICROARCH.md; a surplus workgroup only starts late, lower ranges never wait for higher ones)
        static int occ = -1;
        static bool sk_attr = false;
        const size_t lds_sk = lds_full > lds_epi ? lds_full : lds_epi;
        if (!sk_attr) {
            BRCNN_HIP_CHECK(hipFuncSetAttribute((const void*)conv_igemm_bf16_dma_kernel<MT, NT, RES, OUTF32, WM, WNW, ST, ET, MODE, true>,
                                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_sk));
            int n = 0;
            BRCNN_HIP_CHECK(hipOccupancyMaxActiveBlocksPerMultiprocessor(
                &n, (const void*)conv_igemm_bf16_dma_kernel<MT, NT, RES, OUTF32, WM, WNW, ST, ET, MODE, true>, 64 * WM * WNW, lds_sk));
            if (g_num_cus == 0) {
                int dev = 0;
                hipDeviceProp_t prop;
                BRCNN_HIP_CHECK(hipGetDevice(&dev));
                BRCNN_HIP_CHECK(hipGetDeviceProperties(&prop, dev));
                g_num_cus = prop.multiProcessorCount;
            }
            occ = n;
            sk_attr = true;
        }
        const int rc = sk_plan(p, occ * g_num_cus, 32 * MT * WM, 32 * NT * WNW, s);
        if (rc) return rc;
        if (p.sk_wgs > 0) {
            hipLaunchKernelGGL((conv_igemm_bf16_dma_kernel<MT, NT, RES, OUTF32, WM, WNW, ST, ET, MODE, true>), dim3(p.sk_wgs),
                               dim3(64 * WM * WNW), lds_sk, s, p);
            BRCNN_LAUNCH_CHECK();
            return 0;
        }
    }
    const size_t lds_k = (nk < ST ? nk : ST) * lds_stage;              // operand ring (shorter for a short K loop)
    const size_t lds = lds_k > lds_epi ? lds_k : lds_epi;               // the epilogue slabs reuse the same space
    static bool attr_done = false;
    if (!attr_done) {
        const size_t lds_max = lds_full > lds_epi ? lds_full : lds_epi;
        BRCNN_HIP_CHECK(hipFuncSetAttribute((const void*)conv_igemm_bf16_dma_kernel<MT, NT, RES, OUTF32, WM, WNW, ST, ET, MODE>,
                                            hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_max));
        attr_done = true;
    }
    hipLaunchKernelGGL((conv_igemm_bf16_dma_kernel<MT, NT, RES, OUTF32, WM, WNW, ST, ET, MODE>), dim3(p.tiles_m * p.tiles_n),
                       dim3(64 * WM * WNW), lds, s, p);
    BRCNN_LAUNCH_CHECK();
    return 0;
}

template <int MT, int NT, int WM = 2, int WNW = 2, int ST = 2, int ET = 0>
int launch2(ConvParams& p, hipStream_t s) {
    p.tiles_m = (p.M + 32 * MT * WM - 1) / (32 * MT * WM);
    p.tiles_n = (p.Cout + 32 * NT * WNW - 1) / (32 * NT * WNW);
    if (p.z_out || p.tail_z) {      // training epilogues: the production tiles of the backbone layers only
        if constexpr (ST == 2 && NT <= 2 && MT * NT <= 2 && ((WM == 4 && WNW == 2) || (WM == 2 && WNW == 2))) {
            if (p.out_f32) return BRCNN_EINVAL;
            if (p.tail_z && p.tail_mask)
                return (p.residual && p.tail_dres) ? launch<MT, NT, true, false, WM, WNW, ST, ET, 3>(p, s) : BRCNN_EINVAL;
            if (p.tail_z) return p.residual ? BRCNN_EINVAL : launch<MT, NT, true, false, WM, WNW, ST, ET, 2>(p, s);
            return p.residual ? launch<MT, NT, true, false, WM, WNW, ST, ET, 1>(p, s)
                              : launch<MT, NT, false, false, WM, WNW, ST, ET, 1>(p, s);
        } else {
            return BRCNN_EINVAL;
        }
    }
    if constexpr (MT * NT > 4) {        // large register tiles: bf16 output, no residual operand (callers check)
        return launch<MT, NT, false, false, WM, WNW, ST, ET>(p, s);
    } else {
        if (p.out_f32)
            return p.residual ? launch<MT, NT, true, true, WM, WNW, ST, ET>(p, s) : launch<MT, NT, false, true, WM, WNW, ST, ET>(p, s);
        return p.residual ? launch<MT, NT, true, false, WM, WNW, ST, ET>(p, s) : launch<MT, NT, false, false, WM, WNW, ST, ET>(p, s);
    }
}

// The 256 x 256 eight-phase kernel (conv_pp_bf16.hip) where it wins (tools/conv_bench_bf16.py, profiles/r03_notes.md):
// whole 256-channel column tiles, at least half a generation of tiles (one workgroup per CU; on 132 tiles it still
// beats the small tiles' two full generations: 58 vs 70 us), K >= 512 (on the 1 - 4 K-tile layers the prologue and the
// partly filled second generation cost more than the schedule returns).  g_pp_mode: 0 never, 1 heuristic.
int g_pp_mode = 1;
static bool pp_wins(const ConvParams& p) {
    if (g_pp_mode == 0 || p.gstep || (p.Cout % 256) || p.K < 512 || (p.K % 64) || p.KH * p.KW > 32 || p.scatter) return false;
    if (p.tail_z && p.tail_mask) return false;
    const long long t88 = (long long)((p.M + 255) / 256) * (p.Cout / 256);
    return t88 >= 128;
}

// The 256 x 128 two-group kernel (conv_pp128_bf16.hip, round 6).  Measured (tools/conv_bench_bf16.py, profiles/r06_notes.md):
// per CU it runs at 0.8 of the 256 x 256 kernel's rate on a full chip (912 vs 1149 TF/s: 48 KB of LDS-DMA per 128 MFMAs
// instead of 64 KB per 256 -- the chip-wide LDS-DMA rate of ~10.8 TB/s is what bounds both), and on the mid-size maps its
// 264 / 525 tiles sit just above one / two generations of 256 workgroups, where the chained stream-K hand-over (128 KB of
// fp32 accumulators each way per slot) costs what the second generation would.  It wins where neither of the other kernels
// has a shape: few 256-row tiles AND a long K loop -- the stage-4 3x3 layers (M = 8400, N = 512, K = 4608: 71.5 -> 63.5 us).
// The heuristic takes exactly those; `brcnn_conv_set_tile_bf16(8842)` forces it wherever its shape rules allow.
// g_pp128_mode: 0 never, 1 heuristic.  g_pp128_min_k: shortest K the heuristic takes.
int g_pp128_mode = 1;
int g_pp128_min_k = 4096;
int g_pp128_max_t88 = 128;      // N % 256 == 0 layers: 256 x 256 tiles from this count on (where pp_wins takes them)
static bool pp128_ok(const ConvParams& p) {
    return !(p.gstep || (p.Cout % 128) || p.K < 192 || (p.K % 64) || p.KH * p.KW > 32 || p.scatter || (p.tail_z && p.tail_mask));
}
static bool pp128_wins(const ConvParams& p) {
    if (g_pp128_mode == 0 || !pp128_ok(p) || p.K < g_pp128_min_k || p.KH * p.KW < 9) return false;     // (measured on 3x3 layers only)
    const long long tm = (p.M + 255) / 256;
    if ((p.Cout % 256) == 0 && tm * (p.Cout / 256) >= g_pp128_max_t88) return false;
    const long long t84 = tm * (p.Cout / 128);
    return t84 >= 128 && t84 <= 264;          // (one generation of workgroups: more go to the smaller tiles, several per CU)
}

int g_bf16_il = 0;     // tuning hook (set_tile_bf16(-1 / -2)): spread the LDS-DMA pieces between the MFMA groups
int g_bf16_tile = 0;   // tuning hook: 0 heuristic, 11 / 21 / 22 = MT NT (4 waves), 42 = 256x128 (8 waves)

}  // namespace

namespace brcnn_conv {
int sk_plan_pp(ConvParams& p, int slots, int bm, int bn, hipStream_t s) { return sk_plan(p, slots, bm, bn, s, 16); }
bool sk_par_enabled() { return g_sk_par != 0 && g_sk_mode != 0; }
// fp32 (K tiles of 32 values, 16x the MFMA time per tile): a hand-over is cheap against a tile, any idle CU is not
int sk_plan_pp_f32(ConvParams& p, int slots, int bm, int bn, hipStream_t s) { return sk_plan(p, slots, bm, bn, s, 8, 32, 0.97); }

// fp16 operands: the production tile shapes only (the tuning-hook variants stay bf16)
static int dispatch_conv_f16(ConvParams& p, hipStream_t s) {
    p.il = 0;
    if (const int rc = conv1x1_stream_try(p, s, 1)) return rc < 0 ? rc : 0;
    if (p.gstep) return launch2<1, 1, 4, 2, 2, 1>(p, s);
    if (pp128_wins(p)) return dispatch_conv_pp128_bf16(p, s);
    if (pp_wins(p)) return dispatch_conv_pp_bf16(p, s);
    const long long t22 = (long long)((p.M + 127) / 128) * ((p.Cout + 127) / 128);
    const long long t44 = (long long)((p.M + 255) / 256) * ((p.Cout + 255) / 256);
    const bool fill44 = p.Cout >= 256 && p.K >= 1024 && !p.residual && !p.out_f32 && !p.z_out && !p.tail_z && t44 >= 512 &&
                        (double)t44 / (double)(((t44 + 255) / 256) * 256) >= 0.85;
    if (fill44 && p.Cout > 128) return launch2<2, 2, 4, 4, 2, 1>(p, s);
    if (p.Cout <= 64) return launch2<2, 1, 2, 2, 2, 1>(p, s);
    if (t22 < 256) return launch2<1, 1, 2, 2, 2, 1>(p, s);
    const bool sk82 = g_sk_mode == 1 && (p.Cout % 128) == 0 && p.K >= 2048 && t22 >= 512 &&
                      (double)t22 / (double)(((t22 + 511) / 512) * 512) < 0.9;
    if (p.M >= 65536 || sk82) return launch2<1, 2, 4, 2, 2, 1>(p, s);
    if (p.M >= 16384) return launch2<1, 1, 4, 2, 2, 1>(p, s);
    return launch2<2, 1, 2, 2, 2, 1>(p, s);
}

int dispatch_conv_bf16(ConvParams& p, hipStream_t s) {
    if (p.f16) return dispatch_conv_f16(p, s);
    p.il = g_bf16_il;
    if (g_bf16_tile == 0) {
        if (const int rc = conv1x1_stream_try(p, s, 0)) return rc < 0 ? rc : 0;
    }
    if (p.gstep) return launch2<1, 1, 4, 2>(p, s);      // grouped conv: 64-channel N tiles (128x64 on 8 waves)
    int t = ((p.z_out || p.tail_z) && g_bf16_tile != 8844 && g_bf16_tile != 8842) ? 0 : g_bf16_tile;
    if (t == 0 && pp128_wins(p)) return dispatch_conv_pp128_bf16(p, s);
    if (t == 0 && pp_wins(p)) return dispatch_conv_pp_bf16(p, s);
    if (t == 0) {
        // Measured per layer shape (tools/conv_bench_bf16.py, profiles/r01_conv_tiles_bf16.txt): the more
        // waves share the LDS-DMA issue of a K tile, the better -- 128x128 on 8 waves of 32x64 beats the
        // same tile on 4 waves of 64x64 by ~10 %, 128x64 on 8 waves wins on the mid-size maps; the
        // small 4-wave tiles keep the few-tile layers (FC, stage 4) spread over the chip.
        const long long t22 = (long long)((p.M + 127) / 128) * ((p.Cout + 127) / 128);
        // 256x256 on 16 waves moves half the LDS-DMA bytes per FLOP of the 128x128 tile (980 vs 710
        // TF/s per full generation), but is one workgroup per CU: only when its tile count fills
        // whole generations of 256 (the fused RPN tower: 700 tiles)
        const long long t44 = (long long)((p.M + 255) / 256) * ((p.Cout + 255) / 256);
        const bool fill44 = p.Cout >= 256 && p.K >= 1024 && !p.residual && !p.out_f32 && !p.z_out && !p.tail_z && t44 >= 512 &&
                            (double)t44 / (double)(((t44 + 255) / 256) * 256) >= 0.85;
        // 128x128 under the stream-K schedule where its tile count sits just above a multiple of the 512 slots (the
        // stage-3 3x3 layers: 526 tiles) and K is long: 69 -> 65 us there, 214 -> 188 us on the 4608-deep layer
        const bool sk82 = g_sk_mode == 1 && (p.Cout % 128) == 0 && p.K >= 2048 && t22 >= 512 &&
                          (double)t22 / (double)(((t22 + 511) / 512) * 512) < 0.9;
        if (fill44) t = 2244;
        else if (p.Cout <= 64) t = 21;
        else if (t22 < 256) t = 11;
        else if (p.M >= 65536 || sk82) t = 82;
        else if (p.M >= 16384) t = 81;
        else t = 21;
    }
    if (t == 8842 && pp128_ok(p)) return dispatch_conv_pp128_bf16(p, s);
    if (t == 8842) t = 82;
    if (t == 8844 && p.Cout > 128 && p.K >= 128 && p.KH * p.KW <= 32 && !(p.tail_z && p.tail_mask)) return dispatch_conv_pp_bf16(p, s);
    if (t == 8844) t = 82;
    if (t == 342 && p.Cout > 64) return launch2<2, 2, 4, 2, 3>(p, s);    // 3-stage ring variants
    if (t == 382 && p.Cout > 64) return launch2<1, 2, 4, 2, 3>(p, s);
    if (t == 3164 && p.Cout > 64) return launch2<1, 1, 4, 4, 3>(p, s);
    if (t == 322 && p.Cout > 64) return launch2<2, 2, 2, 2, 3>(p, s);
    if (t == 482 && p.Cout > 64) return launch2<1, 2, 4, 2, 4>(p, s);
    if (t == 381 || ((t == 342 || t == 382 || t == 3164 || t == 322 || t == 482) && p.Cout <= 64)) return launch2<1, 1, 4, 2, 3>(p, s);
    if (t == 2244 && p.Cout > 128) return launch2<2, 2, 4, 4>(p, s);   // 256x256 on 16 waves of 64x64
    if (t == 2144 && p.Cout > 64) return launch2<2, 1, 4, 4>(p, s);    // 256x128 on 16 waves of 64x32
    if (t == 2244 || t == 2144) t = 82;
    if (t == 42 && p.Cout > 64) return launch2<2, 2, 4>(p, s);
    if (t == 82 && p.Cout > 64) return launch2<1, 2, 4>(p, s);       // 128x128 on 8 waves of 32x64
    if (t == 164 && p.Cout > 64) return launch2<1, 1, 4, 4>(p, s);   // 128x128 on 16 waves of 32x32
    if (t == 81 || ((t == 82 || t == 164) && p.Cout <= 64)) return launch2<1, 1, 4, 2>(p, s);   // 128x64 on 8 waves

    if ((t == 22 || t == 42) && p.Cout > 64) return launch2<2, 2>(p, s);
    if (t == 21 || ((t == 22 || t == 42) && p.Cout <= 64)) return launch2<2, 1>(p, s);
    return launch2<1, 1>(p, s);
}
}  // namespace brcnn_conv

BRCNN_API int brcnn_conv_handover_status(void) {
    std::lock_guard<std::mutex> lock(g_sk_mutex);
    return sk_take_error();
}

namespace brcnn_conv {
int tuning_get_stream_k() { return g_sk_mode; }
int tuning_get_split_k() { return g_sk_par; }
int tuning_get_eight_phase_16() { return g_pp_mode; }
}  // namespace brcnn_conv

BRCNN_API int brcnn_conv_set_tile_bf16(int mtnt) {
    if (mtnt == -1 || mtnt == -2) { g_bf16_il = (mtnt == -1); return 0; }
    if (mtnt <= -3 && mtnt >= -5) { g_sk_mode = -3 - mtnt; return 0; }       // stream-K: -3 off, -4 heuristic, -5 forced
    if (mtnt <= -8 && mtnt >= -10) { g_sk_par = -8 - mtnt; return 0; }       // split-K of few-tile launches: -8 off, -9 heuristic, -10 forced
    if (mtnt == -6 || mtnt == -7) { g_pp_mode = mtnt == -7; return 0; }      // eight-phase kernel: -6 never, -7 heuristic
    if (mtnt == -18 || mtnt == -19) { g_pp128_mode = mtnt == -19; return 0; }   // 256 x 128 two-group kernel: -18 never, -19 heuristic
    if (mtnt <= -1000 && mtnt > -2000) { g_pp128_min_k = -1000 - mtnt; return 0; }      // ... its shortest K (-1000 - K)
    if (mtnt <= -2000 && mtnt > -3000) { g_pp128_max_t88 = -2000 - mtnt; return 0; }    // ... 256 x 256 tiles from this count on
    // test hook: -11 = the K heads of the following stream-K launches do not publish and the tails give up after 256
    // polls (a lost hand-over, to exercise BRCNN_EHANDOVER); -12 = back to normal
    if (mtnt == -15 || mtnt == -16 || mtnt == -17) return conv1x1_stream_set(-15 - mtnt);      // persistent short-K 1x1 kernel never / heuristic / forced
    if (mtnt == -11 || mtnt == -12) { g_sk_drop_publish = mtnt == -11; g_sk_spin_limit = mtnt == -11 ? 256 : 1 << 24; return 0; }
    const int ok[] = {0, 11, 21, 22, 42, 82, 81, 164, 342, 382, 3164, 322, 482, 381, 2244, 2144, 8844, 8842};
    bool found = false;
    for (int v : ok) found |= (v == mtnt);
    if (!found) return BRCNN_EINVAL;
    g_bf16_tile = mtnt;
    return 0;
}
