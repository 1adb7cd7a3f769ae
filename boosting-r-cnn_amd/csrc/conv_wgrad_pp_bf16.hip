// 16-bit weight gradient on a 256 (co) x 256 (k) tile with the eight-phase, two-group schedule of conv_pp_bf16.hip.
//
//   dW[co, k] (fp32) += sum_m dY[m, co] * A[m, k]      m = (n, ho, wo) pixels,  k = (kh, kw, ci)
//
// conv_wgrad_bf16.hip runs {stage 64 reduction rows, read fragments, MFMA, vmcnt(0), barrier} per step with every
// wave in the same phase; its 256 x 256 form (16 waves) reaches 830-910 TFLOP/s on the tower layers where the forward
// kernel of the same shape -- the eight-phase schedule -- reaches 1100+.  This kernel is that schedule on the weight
// gradient's operands:
//   * both operands are REDUCTION-major in memory ((M, Cout) and the im2col rows (M, K)), so they are staged as they lie
//     -- [64 m][64 columns] blocks of 128-byte rows, LDS-DMA, source-side XOR swizzle of the 16-byte chunks -- and the
//     MFMA fragments (8 consecutive m per lane) come out of the transposing LDS load `ds_read_b64_tr_b16`, two per
//     fragment;
//   * a half-tile slot (16 KB) = two column blocks: dY half h = co [128 h, 128 h + 128), x half h = k [128 h, +128);
//     wave group wm (0 / 1) owns the co block wm of either half (two 32-row MFMA tiles each: 4 x 32 rows = 128 co),
//     wave wn the 32 k columns (wn & 1) of block wn >> 1 of either half (2 x 32 = 64 k): 4 x 2 MFMA tiles, 128
//     accumulator registers, exactly the forward kernel's wave tile with  rows := co  and  columns := k;
//   * per 64-row step four phases (one 64 x 32 quadrant each: 8 MFMAs), the second wave group one barrier behind the
//     first; eight slots = two steps in LDS, five half-tiles in flight, counted vmcnt(10) only;
//   * the x operand is the im2col gather: a lane stages the same two tile rows in every step, so its pixel decode
//     (image, ho, wo) advances by 64 rows per step (one conditional wrap on maps at least 64 wide, the full decode
//     otherwise and across map boundaries), and its (tap, channel) column split is loop invariant.
// The M slices of a (co, k) tile leave as fp32 slabs in register order; wgrad_pp_reduce_kernel adds them in slice order
// into dW (fixed order: reproducible), with 256-thread workgroups and eight loads in flight per thread.
#include <type_traits>
#include "conv_common.h"

namespace {
using namespace brcnn_conv;

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) void* lds_ptr_t;
constexpr int TM = 64;              // reduction rows per step
constexpr int TILE = 256;
constexpr int SLOT = 16384;         // one half-tile slot: 2 blocks x 64 rows x 128 B
constexpr int BLKB = 8192;          // bytes of one [64][64] block
constexpr int MAX_SLICES = 320;

template <int N> using ic = std::integral_constant<int, N>;

struct WgradPPParams {
    const unsigned short* dy;   // (M, Cout)
    const unsigned short* x;    // NHWC segments back to back
    float* dw;                  // (Cout, K) fp32, accumulated into
    float* slab;                // slices > 1: [slice][tile][32 register groups][512 threads] float4
    // in-launch reduction of the slabs (fuse_group > 0): the slices of a tile form groups of fuse_group; the LAST workgroup
    // of a group to arrive (ticket on counters[tile * (ngroups + 1) + group]) adds the group's slabs in slice order into the
    // group's first slab, then takes a ticket on the tile's counter (index ngroups); the last group of the tile adds the
    // group sums in group order into dw.  Fixed association whoever does the adding: reproducible.  No workgroup waits.
    unsigned* counters;
    int fuse_group, ngroups;
    int Cin, Cout, KH, KW, stride, pad, M, K;
    int tiles_co, tiles_k, slices;
    unsigned dy_bytes, x_bytes;
    int nseg;
    // slice s reduces rows [slice_m0[s], slice_m0[s + 1]); the host cuts every map (pyramid level) into its OWN slices, so a
    // workgroup's rows lie in one map and its geometry is wave-uniform (scalar registers, no per-lane map search)
    int slice_m0[MAX_SLICES + 1];
    int seg_m0[BRCNN_MAX_LEVELS + 1];
    int seg_H[BRCNN_MAX_LEVELS], seg_W[BRCNN_MAX_LEVELS], seg_Ho[BRCNN_MAX_LEVELS], seg_Wo[BRCNN_MAX_LEVELS];
    long long seg_xoff[BRCNN_MAX_LEVELS];
    unsigned seg_mhw[BRCNN_MAX_LEVELS], seg_shw[BRCNN_MAX_LEVELS], seg_mw[BRCNN_MAX_LEVELS], seg_sw[BRCNN_MAX_LEVELS];
};

__device__ __forceinline__ unsigned fastdiv(unsigned x, unsigned magic, unsigned shift) {
    return (unsigned)(((unsigned long long)__umulhi(x, magic) + x) >> shift);
}

// ET: 0 bf16, 1 fp16.  PLAIN: 1x1 filter, stride 1, no padding, one map: x is the plain (M, Cin) matrix.
template <int ET, bool PLAIN>
__global__ __launch_bounds__(512, 2) void conv_wgrad_pp_bf16_kernel(WgradPPParams p) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 2, wn = wave & 3;

    const int tiles = p.tiles_co * p.tiles_k;
    const int nwg = tiles * p.slices;
    int b = xcd_remap(blockIdx.x, nwg);
    const int slice = b / tiles;
    b -= slice * tiles;
    const int tk = b % p.tiles_k, tco = b / p.tiles_k;
    const int co0 = tco * TILE, k0 = tk * TILE;
    const int m_begin = p.slice_m0[slice];
    const int m_end = p.slice_m0[slice + 1];
    if (m_begin >= m_end) return;
    int sg = 0;                         // the map of this slice (scalar)
#pragma unroll
    for (int t = 1; t < BRCNN_MAX_LEVELS; t++)
        if (t < p.nseg && m_begin >= p.seg_m0[t]) sg = t;
    const int gH = p.seg_H[sg], gW = p.seg_W[sg], gHo = p.seg_Ho[sg], gWo = p.seg_Wo[sg];
    const int nt = (m_end - m_begin + TM - 1) / TM;             // steps of this slice

    const __amdgpu_buffer_rsrc_t rsrc_y = __builtin_amdgcn_make_buffer_rsrc((void*)p.dy, 0, (int)p.dy_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rsrc_x = __builtin_amdgcn_make_buffer_rsrc((void*)p.x, 0, (int)p.x_bytes, 0x00020000);

    // ---- staging.  A wave fills, per slot, rows [16 (wave & 3), +16) of block wave >> 2 with two DMA instructions
    // (8 rows x 128 B each): LDS byte 2048 wave + 1024 j inside the slot.  Lane (rg_row, pc): row 16 (wave & 3) + 8 j +
    // rg_row, physical chunk pc = logical chunk pc ^ ((row >> 1) & 7).
    const int rg_row = lane >> 3, pc = lane & 7;
    int row[2], lcol[2];
#pragma unroll
    for (int j = 0; j < 2; j++) {
        row[j] = 16 * (wave & 3) + 8 * j + rg_row;
        lcol[j] = (pc ^ ((row[j] >> 1) & 7)) * 8;            // column inside the 64-column block
    }
    // dY: byte offset of (row, column) relative to the step's first row, or -1
    int y_off[2][2];
#pragma unroll
    for (int h = 0; h < 2; h++)
#pragma unroll
        for (int j = 0; j < 2; j++) {
            const int co = co0 + h * 128 + wm * 64 + lcol[j];
            y_off[h][j] = co < p.Cout ? (row[j] * p.Cout + co) * 2 : -1;
        }
    // x: a wave stages ONE 64-column block (block wm) of either half, and a block lies inside one filter tap (Cin is a
    // multiple of 64): the tap (kh, kw) and the block's first channel are wave-uniform -- scalar registers -- and a lane
    // adds its own 8-column piece lcol[j] (folded into its pixel offset below).  x_s[h]: byte offset of the half's
    // (tap, first channel) relative to a pixel's (kh, kw) = (pad, pad) position, x_dh / x_dw: kh - pad, kw - pad;
    // x_ok: the block's columns are inside K.
    int x_s[2], x_dh[2], x_dw[2];
    bool x_ok[2];
#pragma unroll
    for (int h = 0; h < 2; h++) {
        const int k = k0 + h * 128 + wm * 64;                  // scalar
        x_ok[h] = k < p.K;
        const int tap = x_ok[h] ? k / p.Cin : 0;
        const int kh = tap / p.KW, kw = tap - kh * p.KW;
        x_dh[h] = kh - p.pad;
        x_dw[h] = kw - p.pad;
        x_s[h] = PLAIN ? k * 2 : (k - tap * p.Cin) * 2;        // + the tap's pixel displacement, added per map width below
    }
    const unsigned lds0 = (unsigned)(size_t)(lds_ptr_t)smem;
    const unsigned st_dst = lds0 + (unsigned)wave * 2048u;
    auto stage_Y = [&](int slot, int h, int ti) {
        const int left = ti < nt ? m_end - (m_begin + ti * TM) : 0;       // valid rows of the step (scalar)
        const int base = (m_begin + ti * TM) * p.Cout * 2;
#pragma unroll
        for (int j = 0; j < 2; j++) {
            const int off = (row[j] < left && y_off[h][j] >= 0) ? base + y_off[h][j] : OOB;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc_y, (lds_ptr_t)(size_t)(st_dst + slot * SLOT + j * 1024), 16, off, 0, 0, 0);
        }
    };
    // Pixel state of the lane's two rows at the step whose x halves are staged next, in INPUT coordinates: hi0 / wi0 =
    // ho stride - pad, wo stride - pad (the (kh, kw) = (0, 0) tap) and pix = byte offset of input pixel (ho stride,
    // wo stride) + the lane's column piece.  A step moves a row 64 output pixels on: on maps at least 64 wide that is
    // at most one wrap to the next output row (and possibly to the next image) -- selects, no branch; narrower maps and
    // the first step behind a map boundary decode m afresh.  One map: its geometry is launch-uniform (scalar registers).
    int s_hi0[2], s_wi0[2], s_pix[2];
    auto decode = [&](int m, int j) {
        const int ml = m - p.seg_m0[sg];
        const int n = (int)fastdiv((unsigned)ml, p.seg_mhw[sg], p.seg_shw[sg]);
        const int rem = ml - n * (gHo * gWo);
        const int ho = (int)fastdiv((unsigned)rem, p.seg_mw[sg], p.seg_sw[sg]);
        const int wo = rem - ho * gWo;
        s_hi0[j] = ho * p.stride - p.pad;
        s_wi0[j] = wo * p.stride - p.pad;
        s_pix[j] = ((int)p.seg_xoff[sg] + ((n * gH + ho * p.stride) * gW + wo * p.stride) * p.Cin + lcol[j]) * 2;
    };
    int x_step = 0;                     // the step the pixel state stands at
    if constexpr (!PLAIN) {
#pragma unroll
        for (int j = 0; j < 2; j++) {
            s_hi0[j] = s_wi0[j] = s_pix[j] = 0;
            if (m_begin + row[j] < m_end) decode(m_begin + row[j], j);
        }
    }
    // (scalar constants of the map)
    const int c_sw = TM * p.stride * p.Cin * 2;                       // bytes: 64 output columns on
    const int c_wlim = gWo * p.stride - p.pad, c_hlim = gHo * p.stride - p.pad;
    const int c_wwrap = gWo * p.stride, c_hwrap = gHo * p.stride;
    const int c_rowjump = (p.stride * gW - gWo * p.stride) * p.Cin * 2;   // first column of the next output row
    const int c_imgjump = (gH - gHo * p.stride) * gW * p.Cin * 2;      // first row of the next image
    auto advance_x = [&]() {            // the pixel state moves 64 rows on
        x_step++;
        if constexpr (!PLAIN) {
            if (gWo < TM) {             // narrow map (launch-uniform branch): decode afresh
#pragma unroll
                for (int j = 0; j < 2; j++) {
                    const int m = m_begin + x_step * TM + row[j];
                    if (m < m_end) decode(m, j);
                }
            } else {
#pragma unroll
                for (int j = 0; j < 2; j++) {
                    int wi = s_wi0[j] + TM * p.stride;
                    int pix = s_pix[j] + c_sw;
                    const bool wrap = wi >= c_wlim;                  // past the last output column
                    const int hi = s_hi0[j] + (wrap ? p.stride : 0);
                    const bool img = wrap && hi >= c_hlim;           // past the last output row: next image
                    wi -= wrap ? c_wwrap : 0;
                    pix += wrap ? c_rowjump : 0;
                    pix += img ? c_imgjump : 0;
                    s_hi0[j] = img ? hi - c_hwrap : hi;
                    s_wi0[j] = wi;
                    s_pix[j] = pix;
                }
            }
        }
    };
    auto stage_X = [&](int slot, int h, int ti) {
        const int left = ti < nt ? m_end - (m_begin + ti * TM) : 0;
        if constexpr (PLAIN) {
            const int base = (m_begin + ti * TM) * p.Cin * 2 + x_s[h];
#pragma unroll
            for (int j = 0; j < 2; j++) {
                const int off = (row[j] < left && x_ok[h]) ? base + (row[j] * p.Cin + lcol[j]) * 2 : OOB;
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc_x, (lds_ptr_t)(size_t)(st_dst + slot * SLOT + j * 1024), 16, off, 0, 0, 0);
            }
        } else {
#pragma unroll
            for (int j = 0; j < 2; j++) {
                const int hi = s_hi0[j] + p.pad + x_dh[h], wi = s_wi0[j] + p.pad + x_dw[h];
                const bool ok = (row[j] < left) & x_ok[h] & ((unsigned)hi < (unsigned)gH) & ((unsigned)wi < (unsigned)gW);
                const int off = ok ? s_pix[j] + (x_dh[h] * gW + x_dw[h]) * p.Cin * 2 + x_s[h] : OOB;
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc_x, (lds_ptr_t)(size_t)(st_dst + slot * SLOT + j * 1024), 16, off, 0, 0, 0);
            }
        }
    };

    // ---- fragment reads (transposing): MFMA tile columns col0 .. col0 + 31 of a block, reduction rows 16 ks + 8 (lane >> 5) + 4 q + (0..3)
    const int g4 = lane >> 4, t16 = lane & 15;
    auto frag_addr = [&](int blk, int col0, int q) -> unsigned {
        const int col = col0 + 16 * (g4 & 1) + 4 * (t16 & 3);
        const int r = 8 * (g4 >> 1) + 4 * q + (t16 >> 2);
        return lds0 + (unsigned)(blk * BLKB + (r * 64 + (((col >> 3) ^ ((r >> 1) & 7)) << 3) + (col & 4)) * 2);
    };
    unsigned y_rd[2][2], x_rd[2];
#pragma unroll
    for (int q = 0; q < 2; q++) {
        y_rd[0][q] = frag_addr(wm, 0, q);
        y_rd[1][q] = frag_addr(wm, 32, q);
        x_rd[q] = frag_addr(wn >> 1, (wn & 1) * 32, q);
    }
    s16x4 Yr[2][4][2], X0r[4][2], X1r[4][2];        // [MFMA tile][ks][half of the 8 reduction rows]
    auto rd = [&](s16x4& d, unsigned addr, auto off_c) {
        constexpr int OFF = decltype(off_c)::value;
        if constexpr (OFF < 65536) {
            asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(d) : "v"(addr), "n"(OFF) : "memory");
        } else {
            const unsigned hi = addr + 65536u;
            asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(d) : "v"(hi), "n"(OFF - 65536) : "memory");
        }
    };
    auto read_Y = [&](auto slot_c) {
        constexpr int S = decltype(slot_c)::value;
#pragma unroll
        for (int ks = 0; ks < 4; ks++)
#pragma unroll
            for (int q = 0; q < 2; q++) {
                if (ks == 0) { rd(Yr[0][0][q], y_rd[0][q], ic<S * SLOT>{}); rd(Yr[1][0][q], y_rd[1][q], ic<S * SLOT>{}); }
                if (ks == 1) { rd(Yr[0][1][q], y_rd[0][q], ic<S * SLOT + 2048>{}); rd(Yr[1][1][q], y_rd[1][q], ic<S * SLOT + 2048>{}); }
                if (ks == 2) { rd(Yr[0][2][q], y_rd[0][q], ic<S * SLOT + 4096>{}); rd(Yr[1][2][q], y_rd[1][q], ic<S * SLOT + 4096>{}); }
                if (ks == 3) { rd(Yr[0][3][q], y_rd[0][q], ic<S * SLOT + 6144>{}); rd(Yr[1][3][q], y_rd[1][q], ic<S * SLOT + 6144>{}); }
            }
    };
    auto read_X = [&](s16x4 (&Xr)[4][2], auto slot_c) {
        constexpr int S = decltype(slot_c)::value;
#pragma unroll
        for (int q = 0; q < 2; q++) {
            rd(Xr[0][q], x_rd[q], ic<S * SLOT>{});
            rd(Xr[1][q], x_rd[q], ic<S * SLOT + 2048>{});
            rd(Xr[2][q], x_rd[q], ic<S * SLOT + 4096>{});
            rd(Xr[3][q], x_rd[q], ic<S * SLOT + 6144>{});
        }
    };
    auto pack = [&](const s16x4& lo, const s16x4& hi) -> bf16x8 {
        union { s16x4 h[2]; bf16x8 f; } u;
        u.h[0] = lo;
        u.h[1] = hi;
        return u.f;
    };

    f32x16 acc[4][2];
#pragma unroll
    for (int a = 0; a < 4; a++)
#pragma unroll
        for (int c = 0; c < 2; c++)
#pragma unroll
            for (int r = 0; r < 16; r++) acc[a][c][r] = 0.f;

    // quadrant: co tiles (a0, a0 + 1) x k tile c: D[co][k] += dY^T x
    auto mfma_quad = [&](auto a0_c, auto c_c, s16x4 (&Xr)[4][2]) {
        constexpr int A0 = decltype(a0_c)::value, C = decltype(c_c)::value;
        __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int ks = 0; ks < 4; ks++)
#pragma unroll
            for (int t = 0; t < 2; t++) {
                if constexpr (ET)
                    acc[A0 + t][C] = __builtin_amdgcn_mfma_f32_32x32x16_f16(
                        __builtin_bit_cast(f16x8, pack(Yr[t][ks][0], Yr[t][ks][1])),
                        __builtin_bit_cast(f16x8, pack(Xr[ks][0], Xr[ks][1])), acc[A0 + t][C], 0, 0, 0);
                else
                    acc[A0 + t][C] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(pack(Yr[t][ks][0], Yr[t][ks][1]),
                                                                            pack(Xr[ks][0], Xr[ks][1]), acc[A0 + t][C], 0, 0, 0);
            }
        __builtin_amdgcn_s_setprio(0);
    };
    auto barrier = [&]() { asm volatile("s_barrier" ::: "memory"); };
    auto frags_ready = [&]() {
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
    };
    auto dma_wait = [&]() { asm volatile("s_waitcnt vmcnt(10)" ::: "memory"); };

    // ---- prologue: steps 0 (slots 0-3: X0 Y0 X1 Y1) and 1 (slots 4-7: X1 Y0 X0 Y1), in read order
    stage_X(0, 0, 0);
    stage_Y(1, 0, 0);
    stage_X(2, 1, 0);
    stage_Y(3, 1, 0);
    advance_x();
    stage_X(4, 1, 1);
    stage_Y(5, 0, 1);
    stage_X(6, 0, 1);
    stage_Y(7, 1, 1);
    asm volatile("s_waitcnt vmcnt(12)" ::: "memory");
    barrier();
    read_X(X0r, ic<0>{});
    if (wm == 1) barrier();

    const int iters = (nt + 1) >> 1;
    for (int it = 0; it < iters; it++) {
        const int kt = 2 * it;
        const bool odd_ok = kt + 1 < nt;
        // phase 1: Y0 of the even step (slot 1) x X0; stage slot 7 = Y1 of step kt+1 (the prologue staged it for it == 0)
        read_Y(ic<1>{});
        if (it > 0) stage_Y(7, 1, kt + 1);
        dma_wait();
        barrier();
        frags_ready();
        mfma_quad(ic<0>{}, ic<0>{}, X0r);
        barrier();
        // phase 2: x X1 (slot 2); stage slot 0 = X0 of step kt+2
        read_X(X1r, ic<2>{});
        advance_x();
        stage_X(0, 0, kt + 2);
        dma_wait();
        barrier();
        frags_ready();
        mfma_quad(ic<0>{}, ic<1>{}, X1r);
        barrier();
        // phase 3: Y1 (slot 3) x X1; stage slot 1 = Y0 of step kt+2
        read_Y(ic<3>{});
        stage_Y(1, 0, kt + 2);
        dma_wait();
        barrier();
        frags_ready();
        mfma_quad(ic<2>{}, ic<1>{}, X1r);
        barrier();
        // phase 4: Y1 x X0; read X1 of the odd step (slot 4); stage slot 2 = X1 of step kt+2
        read_X(X1r, ic<4>{});
        stage_X(2, 1, kt + 2);
        dma_wait();
        barrier();
        frags_ready();
        mfma_quad(ic<2>{}, ic<0>{}, X0r);
        barrier();
        // phase 5: Y0 of the odd step (slot 5) x X1; stage slot 3 = Y1 of step kt+2
        read_Y(ic<5>{});
        stage_Y(3, 1, kt + 2);
        dma_wait();
        barrier();
        frags_ready();
        if (odd_ok) mfma_quad(ic<0>{}, ic<1>{}, X1r);
        barrier();
        // phase 6: x X0 (slot 6); stage slot 4 = X1 of step kt+3
        read_X(X0r, ic<6>{});
        advance_x();
        stage_X(4, 1, kt + 3);
        dma_wait();
        barrier();
        frags_ready();
        if (odd_ok) mfma_quad(ic<0>{}, ic<0>{}, X0r);
        barrier();
        // phase 7: Y1 (slot 7) x X0; stage slot 5 = Y0 of step kt+3
        read_Y(ic<7>{});
        stage_Y(5, 0, kt + 3);
        dma_wait();
        barrier();
        frags_ready();
        if (odd_ok) mfma_quad(ic<2>{}, ic<0>{}, X0r);
        barrier();
        // phase 8: Y1 x X1; read X0 of the next even step (slot 0); stage slot 6 = X0 of step kt+3
        read_X(X0r, ic<0>{});
        stage_X(6, 0, kt + 3);
        dma_wait();
        barrier();
        frags_ready();
        if (odd_ok) mfma_quad(ic<2>{}, ic<1>{}, X1r);
        barrier();
    }
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    if (wm == 0) barrier();

    // ---- epilogue.  acc[a][c]: co = co0 + (a >> 1) 128 + wm 64 + (a & 1) 32 + row,  k = k0 + c 128 + (wn >> 1) 64 + (wn & 1) 32 + (lane & 31),
    // row = (r & 3) + 8 (r >> 2) + 4 (lane >> 5)
    if (p.slab) {
        const size_t tile_f4 = (size_t)(TILE * TILE / 4);
        f32x4* slab4 = reinterpret_cast<f32x4*>(p.slab);
        f32x4* dst = slab4 + ((size_t)slice * tiles + b) * tile_f4 + tid;
#pragma unroll
        for (int a = 0; a < 4; a++)
#pragma unroll
            for (int c = 0; c < 2; c++)
#pragma unroll
                for (int g = 0; g < 4; g++) {
                    f32x4 v;
                    v.x = acc[a][c][4 * g + 0]; v.y = acc[a][c][4 * g + 1];
                    v.z = acc[a][c][4 * g + 2]; v.w = acc[a][c][4 * g + 3];
                    dst[((a * 2 + c) * 4 + g) * 512] = v;
                }
        if (p.fuse_group <= 0) return;
        // ---- in-launch reduction (MI355X_MICROARCH.md, inter-workgroup visibility: stores -> every wave drains ->
        // barrier -> one lane: agent release, drained, relaxed ticket; the reader: ticket -> agent acquire -> barrier -> plain
        // loads).  The role word lives in the (now idle) dynamic LDS: a second __shared__ object would make the compiler
        // drain the DMA queue in front of every fragment read of the main loop.
        int* role = reinterpret_cast<int*>(smem);
        const int G = p.fuse_group, NG = p.ngroups;
        const int grp = slice / G;
        const int gs = min(G, p.slices - grp * G);
        unsigned* cnt = p.counters + (size_t)b * (NG + 1);
        auto arrive = [&](unsigned* c, int last_ticket) -> bool {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
            if (tid == 0) {
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                const unsigned t = __hip_atomic_fetch_add(c, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                const int last = (int)t == last_ticket;
                if (last) {
                    __hip_atomic_store(c, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);      // ready for the next launch
                    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
                }
                role[0] = last;
            }
            __syncthreads();
            const bool r = role[0] != 0;
            __syncthreads();
            return r;
        };
        if (!arrive(cnt + grp, gs - 1)) return;
        // sum of `count` slabs first, first + step, ... (slice order) for this thread's 32 float4 positions; `to_dw`: into dw
        // with the accumulator layout, else into the first slab
        auto reduce = [&](int first, int step, int count, bool to_dw) {
            const size_t sstride = (size_t)tiles * tile_f4 * step;
            f32x4* src0 = slab4 + ((size_t)first * tiles + b) * tile_f4 + tid;
            for (int v0 = 0; v0 < 32; v0 += 8) {
                f32x4 s4[8];
#pragma unroll
                for (int u = 0; u < 8; u++) s4[u] = src0[(size_t)(v0 + u) * 512];
                for (int i = 1; i < count; i++) {
                    f32x4 t4[8];
#pragma unroll
                    for (int u = 0; u < 8; u++) t4[u] = src0[(size_t)i * sstride + (size_t)(v0 + u) * 512];
#pragma unroll
                    for (int u = 0; u < 8; u++) { s4[u].x += t4[u].x; s4[u].y += t4[u].y; s4[u].z += t4[u].z; s4[u].w += t4[u].w; }
                }
#pragma unroll
                for (int u = 0; u < 8; u++) {
                    const int v = v0 + u;
                    if (!to_dw) {
                        src0[(size_t)v * 512] = s4[u];
                    } else {
                        const int a = v >> 3, c = (v >> 2) & 1, g = v & 3;
                        const int kk = k0 + c * 128 + (wn >> 1) * 64 + (wn & 1) * 32 + (lane & 31);
                        const int co = co0 + (a >> 1) * 128 + wm * 64 + (a & 1) * 32 + 8 * g + 4 * (lane >> 5);
                        if (kk < p.K) {
                            const float sv[4] = {s4[u].x, s4[u].y, s4[u].z, s4[u].w};
#pragma unroll
                            for (int j = 0; j < 4; j++)
                                if (co + j < p.Cout) p.dw[(size_t)(co + j) * p.K + kk] += sv[j];
                        }
                    }
                }
            }
        };
        reduce(grp * G, 1, gs, NG == 1);
        if (NG == 1) return;
        if (!arrive(cnt + NG, NG - 1)) return;
        reduce(0, G, NG, true);
        return;
    }
    const int li = lane & 31, lh = lane >> 5;
#pragma unroll
    for (int c = 0; c < 2; c++) {
        const int kk = k0 + c * 128 + (wn >> 1) * 64 + (wn & 1) * 32 + li;
        if (kk >= p.K) continue;
#pragma unroll
        for (int a = 0; a < 4; a++)
#pragma unroll
            for (int r = 0; r < 16; r++) {
                const int co = co0 + (a >> 1) * 128 + wm * 64 + (a & 1) * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
                if (co < p.Cout) p.dw[(size_t)co * p.K + kk] += acc[a][c][r];      // one slice: this workgroup owns the tile
            }
    }
}

// second stage: blockIdx.x = (tile, register group v = (a, c, g), half of the 512 first-stage threads); a thread adds
// its float4 over `count` slices first, first + stride, ... in that order (eight loads in flight).  TO_DW: into dw with
// the first stage's accumulator layout; else (first pass over groups of slices, blockIdx.y = group) into the group's
// first slab.
template <bool TO_DW>
__global__ __launch_bounds__(256) void wgrad_pp_reduce_kernel(float* __restrict__ slab, float* __restrict__ dw, int tiles_k,
                                                              int tiles, int slices, int stride, int group, int Cout, int K) {
    const int half = blockIdx.x & 1, v = (blockIdx.x >> 1) & 31, tile = blockIdx.x >> 6;
    const int t512 = half * 256 + threadIdx.x;
    const int first = blockIdx.y * group * stride;
    int count = (slices - first + stride - 1) / stride;
    if (count > group) count = group;
    const size_t step = (size_t)tiles * (TILE * TILE / 4) * stride;
    f32x4* src = reinterpret_cast<f32x4*>(slab) + ((size_t)first * tiles + tile) * (size_t)(TILE * TILE / 4) + (size_t)v * 512 + t512;
    f32x4 s = src[0];
    int i = 1;
    for (; i + 7 < count; i += 8) {
        f32x4 t[8];
#pragma unroll
        for (int u = 0; u < 8; u++) t[u] = src[(size_t)(i + u) * step];
#pragma unroll
        for (int u = 0; u < 8; u++) { s.x += t[u].x; s.y += t[u].y; s.z += t[u].z; s.w += t[u].w; }
    }
    for (; i < count; i++) {
        const f32x4 t = src[(size_t)i * step];
        s.x += t.x; s.y += t.y; s.z += t.z; s.w += t.w;
    }
    if constexpr (!TO_DW) {
        src[0] = s;
    } else {
        const int a = v >> 3, c = (v >> 2) & 1, g = v & 3;
        const int wave = t512 >> 6, lane = t512 & 63;
        const int wm = wave >> 2, wn = wave & 3;
        const int tk = tile % tiles_k, tco = tile / tiles_k;
        const int kk = tk * TILE + c * 128 + (wn >> 1) * 64 + (wn & 1) * 32 + (lane & 31);
        const int co = tco * TILE + (a >> 1) * 128 + wm * 64 + (a & 1) * 32 + 8 * g + 4 * (lane >> 5);
        if (kk >= K) return;
        const float sv[4] = {s.x, s.y, s.z, s.w};
        // dw += sum: the four old values are requested together (rows clamped into the tensor), then the stores -- as
        // `dw[..] += sv[j]` under `if (co + j < Cout)` the four read-modify-writes ran one memory latency after the other
        float old[4];
#pragma unroll
        for (int j = 0; j < 4; j++) old[j] = dw[(size_t)min(co + j, Cout - 1) * K + kk];
#pragma unroll
        for (int j = 0; j < 4; j++)
            if (co + j < Cout) dw[(size_t)(co + j) * K + kk] = old[j] + sv[j];
    }
}

constexpr int NOT_TAKEN = 1 << 20;     // launch(): the slabs would not fit the stream's workspace
int g_wgrad_pp_mode = 1;        // tuning hook (brcnn_conv_set_tile_wgrad_bf16(20 / 21 / 22)): never / heuristic / wherever the shape allows
int g_wgrad_pp_slot_pct = 75;   // ... (4000 + n): n percent of the CUs per launch (the launches share the device with the main stream)
int g_wgrad_pp_two_pass = 24;
int g_wgrad_pp_fuse = 0;         // ... (30 / 31): slab reduction as separate launches / inside the producing launch.  Measured (r04_notes.md):
                                 // the in-launch form costs ~70 us per launch (one workgroup pulling 1-3 MB is latency-bound) -> off
int g_wgrad_pp_launches = 0;     // launches taken so far (tests: hook 29 returns and clears it)

void magic_for(unsigned d, unsigned* magic, unsigned* shift) {
    unsigned l = 0;
    while ((1ull << l) < d) l++;
    *magic = (unsigned)(((1ull << 32) * ((1ull << l) - d)) / d + 1);
    *shift = l;
}

template <int ET, bool PLAIN>
int launch(WgradPPParams& p, hipStream_t s) {
    constexpr size_t lds = 8 * SLOT;
    static bool attr_done = false;
    static int num_cus = 0;
    if (!attr_done) {
        BRCNN_HIP_CHECK(hipFuncSetAttribute((const void*)conv_wgrad_pp_bf16_kernel<ET, PLAIN>,
                                            hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        int dev = 0;
        hipDeviceProp_t prop;
        BRCNN_HIP_CHECK(hipGetDevice(&dev));
        BRCNN_HIP_CHECK(hipGetDeviceProperties(&prop, dev));
        num_cus = prop.multiProcessorCount;
        attr_done = true;
    }
    const int tiles = p.tiles_co * p.tiles_k;
    // M slices: one generation of workgroups (one per CU: 128 KiB of LDS), at least 1024 reduction rows each; every
    // map is cut into its own slices (rows per slice: the smallest multiple of 64 with which the maps' slice counts fit)
    const int slots = num_cus * g_wgrad_pp_slot_pct / 100;
    int budget = slots / tiles;
    const int max_slices = (p.M + 1023) / 1024;
    if (budget > max_slices) budget = max_slices;
    if (budget > MAX_SLICES) budget = MAX_SLICES;
    if (budget < p.nseg) budget = p.nseg;
    int rps = ((p.M + budget - 1) / budget + TM - 1) / TM * TM;
    for (;; rps += TM) {
        int n = 0;
        for (int sgi = 0; sgi < p.nseg; sgi++) n += (p.seg_m0[sgi + 1] - p.seg_m0[sgi] + rps - 1) / rps;
        if (n <= budget || n <= p.nseg) break;
    }
    p.slices = 0;
    for (int sgi = 0; sgi < p.nseg; sgi++) {
        const int rows = p.seg_m0[sgi + 1] - p.seg_m0[sgi];
        const int n = (rows + rps - 1) / rps;
        // equal parts of the map, multiples of 64 rows
        const int part = ((rows + n - 1) / n + TM - 1) / TM * TM;
        for (int m = p.seg_m0[sgi]; m < p.seg_m0[sgi + 1]; m += part) p.slice_m0[p.slices++] = m;
    }
    p.slice_m0[p.slices] = p.M;
    p.slab = nullptr;
    p.fuse_group = 0;
    p.ngroups = 0;
    bool deferred = false;
    if (p.slices > 1) {
        if ((size_t)tiles * p.slices * TILE * TILE * sizeof(float) > ((size_t)160 << 20)) return NOT_TAKEN;
        if (!g_wgrad_pp_fuse) {      // second stage postponed to the stream's batched reduction (wgrad_defer.hip)?
            int derr = 0;
            p.slab = wgrad_defer_slabs(s, (size_t)tiles * p.slices * TILE * TILE * sizeof(float), &derr);
            if (derr) return derr;
            deferred = p.slab != nullptr;
        }
        if (!p.slab) p.slab = conv_ws_wgrad_slabs(s);
        if (!p.slab) return BRCNN_EINVAL;
        if (g_wgrad_pp_fuse) {      // groups of ~sqrt(slices): two serial passes of <= ~6 slabs each on the last arrivers
            int group = 2;
            while (group * group < p.slices) group++;
            const int ngroups = (p.slices + group - 1) / group;
            if ((long long)tiles * (ngroups + 1) <= 8192) {
                p.counters = conv_ws_wgrad_counters(s);
                if (!p.counters) return BRCNN_EINVAL;
                p.fuse_group = group;
                p.ngroups = ngroups;
            }
        }
    }
    g_wgrad_pp_launches++;
    hipLaunchKernelGGL((conv_wgrad_pp_bf16_kernel<ET, PLAIN>), dim3(tiles * p.slices), dim3(512), lds, s, p);
    BRCNN_LAUNCH_CHECK();
    if (deferred) {
        int group = p.slices;
        if (p.slices > g_wgrad_pp_two_pass) {
            group = 4;
            while (group * group < p.slices) group++;
        }
        wgrad_defer_push(s, p.slab, p.dw, tiles, p.tiles_k, p.slices, group, p.Cout, p.K, 0);
    } else if (p.slab && p.fuse_group == 0) {
        int stride = 1, count = p.slices;
        if (p.slices > g_wgrad_pp_two_pass) {
            int group = 4;
            while (group * group < p.slices) group++;
            const int ngroups = (p.slices + group - 1) / group;
            hipLaunchKernelGGL((wgrad_pp_reduce_kernel<false>), dim3(tiles * 64, ngroups), dim3(256), 0, s, p.slab, p.dw,
                               p.tiles_k, tiles, p.slices, 1, group, p.Cout, p.K);
            BRCNN_LAUNCH_CHECK();
            stride = group;
            count = ngroups;
        }
        hipLaunchKernelGGL((wgrad_pp_reduce_kernel<true>), dim3(tiles * 64, 1), dim3(256), 0, s, p.slab, p.dw, p.tiles_k,
                           tiles, p.slices, stride, count, p.Cout, p.K);
        BRCNN_LAUNCH_CHECK();
    }
    return 0;
}

}  // namespace

namespace brcnn_conv {

// 1: the shape is this kernel's (launched); 0: not taken (the caller goes on to conv_wgrad_bf16.hip); < 0: error
int wgrad_pp_bf16_try(const void* x, const void* dy, void* dw, int batch, int num_segments, const int* heights_host,
                      const int* widths_host, int cin, int cout, int kh, int kw, int stride, int pad, hipStream_t stream,
                      int f16) {
    if (g_wgrad_pp_mode == 0) return 0;
    const int K = kh * kw * cin;
    if ((cin & 63) || (cout & 7) || kh > 127 || kw > 255) return 0;
    const int tiles_co = (cout + TILE - 1) / TILE, tiles_k = (K + TILE - 1) / TILE;
    if (g_wgrad_pp_mode == 1) {
        // whole 256-wide tiles on both axes (at most a seventh of the MFMA work on padding) and enough reduction rows
        if (cout < 256 || K < 256) return 0;
        if ((long long)tiles_co * TILE * tiles_k * TILE * 6 > (long long)cout * K * 7) return 0;
    }
    WgradPPParams p = {};
    p.dy = (const unsigned short*)dy; p.x = (const unsigned short*)x; p.dw = (float*)dw;
    p.Cin = cin; p.Cout = cout; p.KH = kh; p.KW = kw; p.stride = stride; p.pad = pad;
    p.nseg = num_segments;
    long long m_total = 0, x_off = 0;
    for (int s = 0; s < num_segments; s++) {
        const int H = heights_host[s], W = widths_host[s];
        const int Ho = (H + 2 * pad - kh) / stride + 1, Wo = (W + 2 * pad - kw) / stride + 1;
        if (H <= 0 || W <= 0 || Ho <= 0 || Wo <= 0 || H > 32767 || W > 32767) return BRCNN_EINVAL;
        p.seg_H[s] = H; p.seg_W[s] = W; p.seg_Ho[s] = Ho; p.seg_Wo[s] = Wo;
        p.seg_m0[s] = (int)m_total;
        p.seg_xoff[s] = x_off;
        magic_for((unsigned)(Ho * Wo), &p.seg_mhw[s], &p.seg_shw[s]);
        magic_for((unsigned)Wo, &p.seg_mw[s], &p.seg_sw[s]);
        m_total += (long long)batch * Ho * Wo;
        x_off += (long long)batch * H * W * cin;
    }
    for (int s = num_segments; s <= BRCNN_MAX_LEVELS; s++) p.seg_m0[s] = (int)m_total;
    if (num_segments > 64) return 0;
    if (m_total * cout * 2 >= 0x7fffffffLL || x_off * 2 >= 0x7fffffffLL) return BRCNN_EINVAL;
    if (g_wgrad_pp_mode == 1 && m_total < 4096) return 0;
    p.M = (int)m_total;
    p.K = K;
    p.dy_bytes = (unsigned)(m_total * cout * 2);
    p.x_bytes = (unsigned)(x_off * 2);
    p.tiles_co = tiles_co;
    p.tiles_k = tiles_k;
    const bool plain = kh == 1 && kw == 1 && stride == 1 && pad == 0 && num_segments == 1;
    int rc;
    if (f16) rc = plain ? launch<1, true>(p, stream) : launch<1, false>(p, stream);
    else rc = plain ? launch<0, true>(p, stream) : launch<0, false>(p, stream);
    if (rc == NOT_TAKEN) return 0;
    return rc ? rc : 1;
}

int tuning_get_wgrad_eight_phase() { return g_wgrad_pp_mode; }
int tuning_get_wgrad_reduce_in_launch() { return g_wgrad_pp_fuse; }
int tuning_get_wgrad_cu_percent() { return g_wgrad_pp_slot_pct; }
int wgrad_pp_set(int v) {
    if (v >= 20 && v <= 22) { g_wgrad_pp_mode = v - 20; return 0; }
    if (v == 29) { const int n = g_wgrad_pp_launches; g_wgrad_pp_launches = 0; return n; }
    if (v == 30 || v == 31) { g_wgrad_pp_fuse = v - 30; return 0; }
    if (v >= 4010 && v <= 4400) { g_wgrad_pp_slot_pct = v - 4000; return 0; }
    if (v >= 5001 && v <= 5999) { g_wgrad_pp_two_pass = v - 5000; return 0; }
    return BRCNN_EINVAL;
}

}  // namespace brcnn_conv
