// 16-bit implicit-GEMM convolution on a 256 x 256 x 64 tile with an eight-phase, two-group ("ping-pong") schedule.
//
// Same decomposition, operand layout and numerics as conv_igemm_bf16.hip (D[m,co] = sum_k A[m,k] W[co,k], LDS-DMA
// staging of 128-byte rows with the source-side XOR swizzle, v_mfma_f32_32x32x16 as D^T = W A^T, the K order of the
// unsplit chain -- so every output is bit-identical to the other tile shapes); what differs is the schedule inside
// the workgroup.  The two-buffer kernel runs {stage next K tile, read fragments, MFMA, vmcnt(0), barrier} per K tile:
// every wave is in the same phase, so the matrix pipes idle while fragments are read and the LDS idles during the
// MFMAs (profiles/design_history_r01_r05.md 4.4: no unit saturated, phases do not overlap).  Here
//   * 8 waves = 2 groups (rows 0-127 / 128-255 of the tile) x 4 column strips; a wave owns 128 x 64 outputs
//     (4 x 2 MFMA tiles, 128 accumulator registers);
//   * a K tile is four phases, one 64 x 32 quadrant of the wave's outputs each (8 MFMAs = 256 matrix-pipe cycles);
//     a phase is  {read the quadrant's new fragments, issue ONE half-tile (16 KB) of LDS-DMA}  barrier
//     {MFMAs}  barrier;  the second group runs one barrier behind the first, so on every SIMD one wave is in its
//     MFMA block while the other reads fragments and issues DMA (MI355X_MICROARCH.md: matrix beside memory);
//   * the LDS holds two K tiles as eight 16 KB half-tile slots, each read in exactly one phase of the eight
//     (slot s in phase s) and re-staged two phases later with the data it needs six phases after that: five
//     half-tiles (80 KB per CU) are in flight, the only wait is a counted `vmcnt(10)` per phase, never a drain.
// Hazards (cdna_hip_programming.md 5, "Read a staged buffer one phase AFTER the wait that retires it"): the slot
// read in phase q was staged in phase q-6 and waited for (by the staging waves, vmcnt(10)) in phase q-1 before that
// phase's first barrier; it is overwritten in phase q+2, two barriers after the slower group's read was waited for.
#include <type_traits>
#include "conv_common.h"
#include "conv_pp_epilogue.h"

namespace {
using namespace brcnn_conv;

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef __attribute__((address_space(3))) void* lds_ptr_t;
constexpr int BKE = 64;
constexpr int BM = 256, BN = 256, MT = 4, NT = 2, WNW = 4, NW = 8;
constexpr int SLOT = 16384;         // bytes of one half-tile slot: 128 rows x 128 B


template <int N> using ic = std::integral_constant<int, N>;

// DIL: zero-stuffed input (p.dilate > 1, the data gradient of a strided conv): the general address form
// SK: chained stream-K schedule (ConvParams::sk_*, conv_igemm_bf16.hip): the workgroup's item (tile, K tiles [kb, ke),
// hand-over slot) comes from the launch's table; a K head stores its accumulators, a K tail starts from them.
template <bool RES, bool OUTF32, int ET, bool DIL, int MODE = 0, bool SK = false>
__global__ __launch_bounds__(512, 2) void conv_pp_bf16_kernel(ConvParams p) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int nk = p.K / BKE;
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 2, wn = wave & 3;
    const int li = lane & 31, lh = lane >> 5;

    const int nwg = p.tiles_m * p.tiles_n;
    int tile, kb = 0, ke = nk, sk_slot = 0, sk_nprev = 0;
    bool sk_par = false;        // split-K pieces summed at the end (table bit 30) instead of the chained hand-over
    if constexpr (SK) {
        const int4 item = p.sk_items[blockIdx.x];
        tile = __builtin_amdgcn_readfirstlane(item.x);
        kb = __builtin_amdgcn_readfirstlane(item.y);
        ke = __builtin_amdgcn_readfirstlane(item.z);
        const int sk_w = __builtin_amdgcn_readfirstlane(item.w);
        sk_slot = sk_w & 0xffff;
        sk_nprev = (sk_w >> 16) & 0xff;
        sk_par = ((sk_w >> 30) & 1) != 0;
        if (tile < 0) return;
    } else {
        tile = xcd_remap(blockIdx.x, nwg);
    }
    const bool finish = !SK || ke == nk;
    const int tile_m = tile / p.tiles_n, tile_n = tile - tile_m * p.tiles_n;
    const int m0 = tile_m * BM, n0 = tile_n * BN;

    const __amdgpu_buffer_rsrc_t rsrc_x = __builtin_amdgcn_make_buffer_rsrc((void*)p.x, 0, (int)p.x_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rsrc_w = __builtin_amdgcn_make_buffer_rsrc((void*)p.w, 0, (int)p.w_bytes, 0x00020000);

    // ---- staging: per half-tile a wave fills LDS rows [16 wave, 16 wave + 16) with two DMA instructions (8 rows x
    // 128 B each).  LDS row R of A half h holds tile row (R>>6)*128 + (2h + ((R>>5)&1))*32 + (R&31): the two 32-row
    // MFMA tiles 2h, 2h+1 of both wave groups; LDS row R of B half h holds output column (R>>5)*64 + h*32 + (R&31).
    const int rg = lane >> 3, pc = lane & 7;
    // Per staged A row: the byte offset of its (kh, kw) = (0, 0) tap, the byte stride of an input row, and one validity
    // bit per filter tap (row inside M, tap inside the map) -- a stage is then one multiply-add, one add and a select
    // per DMA instruction instead of the unpack / four compares / multiply chain (the load block of a phase has to
    // fit beside the other group's 256 cycles of MFMAs).  Zero-stuffed inputs (dilate > 1: data gradient of a
    // strided conv) keep the general form.
    int a_off[2][2], b_off[2][2], lc[2];
    int a_ws[DIL ? 1 : 2][2];           // !DIL: byte stride of an input row
    unsigned a_mask[2][2];
    int a_hw[DIL ? 2 : 1][2], a_HW[DIL ? 2 : 1][2];         // DIL only: packed (hi0, wi0), (H, W)
#pragma unroll
    for (int j = 0; j < 2; j++) lc[j] = (pc ^ ((4 * j + (lane >> 4)) & 7)) * 8;
#pragma unroll
    for (int h = 0; h < 2; h++)
#pragma unroll
        for (int j = 0; j < 2; j++) {
            const int R = 16 * wave + 8 * j + rg;
            const int m = m0 + (R >> 6) * 128 + (2 * h + ((R >> 5) & 1)) * 32 + (R & 31);
            a_off[h][j] = 0;
            a_mask[h][j] = 0u;
            if constexpr (DIL) { a_hw[h][j] = 0; a_HW[h][j] = 0; }
            else a_ws[h][j] = 0;
            if (m < p.M) {
                int sg = 0;
#pragma unroll
                for (int t = 1; t < BRCNN_MAX_LEVELS; t++)
                    if (t < p.nseg && m >= p.seg_m0[t]) sg = t;
                const int ml = m - p.seg_m0[sg];
                const int Ho = p.seg_Ho[sg], Wo = p.seg_Wo[sg], H = p.seg_H[sg], W = p.seg_W[sg];
                const int n = ml / (Ho * Wo);
                const int rem = ml - n * (Ho * Wo);
                const int ho = rem / Wo, wo = rem - ho * Wo;
                const int hi0 = ho * p.stride - p.pad, wi0 = wo * p.stride - p.pad;
                const int base = (int)p.seg_xoff[sg] + n * H * W * p.pitch;
                if constexpr (DIL) {
                    a_HW[h][j] = (H << 16) | W;
                    a_off[h][j] = base;
                    a_hw[h][j] = ((hi0 + 4096) << 16) | (wi0 + 4096);
                    a_mask[h][j] = 1u;
                } else {
                    a_off[h][j] = (base + (hi0 * W + wi0) * p.pitch + lc[j] + tile_n * p.gstep) * 2;
                    a_ws[h][j] = W * p.pitch * 2;
                    unsigned mk = 0u;
                    for (int kh = 0; kh < p.KH; kh++)
                        for (int kw = 0; kw < p.KW; kw++)
                            if ((unsigned)(hi0 + kh) < (unsigned)H && (unsigned)(wi0 + kw) < (unsigned)W) mk |= 1u << (kh * p.KW + kw);
                    a_mask[h][j] = mk;
                }
            }
            const int co = n0 + (R >> 5) * 64 + h * 32 + (R & 31);
            b_off[h][j] = (co < p.Cout) ? (co * p.K + lc[j]) * 2 : OOB;
        }
    const unsigned lds0 = (unsigned)(size_t)(lds_ptr_t)smem;
    const unsigned st_dst = lds0 + (unsigned)wave * 2048u;        // this wave's rows inside a slot

    // K tiles are visited channel chunk by channel chunk, the filter taps INSIDE a chunk (K tile kt = chunk kt / T,
    // tap kt % T; T = KH KW): the nine taps of a 3x3 filter read shifted windows of the same 128-byte pieces of the
    // same pixels back to back, so the workgroups of an XCD keep (tile + halo) x 128 B each in L2 instead of cycling
    // through all channels of the tile between two taps (tap-major order, PMC on the 8 x 140 x 160 x 256 map: 3.6x
    // the input fetched from the fabric).  The weights stay [Cout][kh][kw][ci]: a tile's column is tap Cin + chunk BKE.
    const int T = p.KH * p.KW;
    const unsigned Tmagic = T > 1 ? 0xFFFFFFFFu / (unsigned)T + 1u : 0u;
    auto chunk_of = [&](int kt) { return T > 1 ? (int)__umulhi((unsigned)kt, Tmagic) : kt; };   // kt / T, kt < 2^16
    int tA_ci0 = 0, tA_kh = 0, tA_kw = 0;      // filter tap / channel offset of the K tile whose A halves are staged next
    if constexpr (SK) {
        if (kb > 0) {
            const int cc = chunk_of(kb), tap = kb - cc * T;
            tA_ci0 = cc * BKE;
            tA_kh = tap / p.KW;
            tA_kw = tap - tA_kh * p.KW;
        }
    }
    auto advance_tap = [&]() {
        if (++tA_kw == p.KW) {
            tA_kw = 0;
            if (++tA_kh == p.KH) { tA_kh = 0; tA_ci0 += BKE; }
        }
    };
    auto stage_A = [&](int slot, int h, bool valid) {
        if constexpr (!DIL) {
            const int tap = tA_kh * p.KW + tA_kw;                   // scalar
            const int s_off = (tA_kw * p.pitch + tA_ci0) * 2;
#pragma unroll
            for (int j = 0; j < 2; j++) {
                const bool ok = valid & (((a_mask[h][j] >> tap) & 1u) != 0u);
                const int off = ok ? a_off[h][j] + tA_kh * a_ws[h][j] + s_off : OOB;
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc_x, (lds_ptr_t)(size_t)(st_dst + slot * SLOT + j * 1024), 16, off, 0, 0, 0);
            }
        } else {
#pragma unroll
        for (int j = 0; j < 2; j++) {
            int hi = (a_hw[h][j] >> 16) - 4096 + tA_kh;
            int wi = (a_hw[h][j] & 0xffff) - 4096 + tA_kw;
            bool ok = valid & (a_mask[h][j] != 0u);
            brcnn_undilate(p.dilate, hi, wi, ok);
            const int H = a_HW[h][j] >> 16, W = a_HW[h][j] & 0xffff;
            ok = ok & ((unsigned)hi < (unsigned)H) & ((unsigned)wi < (unsigned)W);
            const int off = ok ? (a_off[h][j] + (hi * W + wi) * p.pitch + tA_ci0 + lc[j] + tile_n * p.gstep) * 2 : OOB;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc_x, (lds_ptr_t)(size_t)(st_dst + slot * SLOT + j * 1024), 16, off, 0, 0, 0);
        }
        }
    };
    auto stage_B = [&](int slot, int h, int kt) {
        const int cc = chunk_of(kt);
        const int koff = ((kt - cc * T) * p.Cin + cc * BKE) * 2;       // byte offset of the K tile inside a weight row
#pragma unroll
        for (int j = 0; j < 2; j++) {
            // (an out-of-range row offset plus the K offset stays out of range and below 2^32)
            const int off = kt < ke ? (int)((unsigned)b_off[h][j] + (unsigned)koff) : OOB;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc_w, (lds_ptr_t)(size_t)(st_dst + slot * SLOT + j * 1024), 16, off, 0, 0, 0);
        }
    };

    // ---- fragment reads: lane (li, lh) reads row li of a 32-row MFMA tile, chunk (2 kk + lh) ^ ((li >> 1) & 7)
    const int sw = (li >> 1) & 7;
    unsigned a_rd[4], b_rd[4];
#pragma unroll
    for (int kk = 0; kk < 4; kk++) {
        const unsigned ch = (unsigned)(((2 * kk + lh) ^ sw) * 16);
        a_rd[kk] = lds0 + (unsigned)(wm * 64 + li) * 128u + ch;
        b_rd[kk] = lds0 + (unsigned)(wn * 32 + li) * 128u + ch;
    }
    f32x4 Ar[2][4], B0r[4], B1r[4];
    // slots 4..7 lie beyond the 16-bit offset field: their reads add 64 KiB to the address register
    auto rd = [&](f32x4& d, unsigned addr, auto off_c) {
        constexpr int OFF = decltype(off_c)::value;
        if constexpr (OFF < 65536) {
            asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(d) : "v"(addr), "n"(OFF) : "memory");
        } else {
            const unsigned hi = addr + 65536u;
            asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(d) : "v"(hi), "n"(OFF - 65536) : "memory");
        }
    };
    auto read_A = [&](auto slot_c) {
        constexpr int S = decltype(slot_c)::value;
#pragma unroll
        for (int kk = 0; kk < 4; kk++) {
            rd(Ar[0][kk], a_rd[kk], ic<S * SLOT>{});
            rd(Ar[1][kk], a_rd[kk], ic<S * SLOT + 4096>{});
        }
    };
    auto read_B = [&](f32x4 (&Br)[4], auto slot_c) {
        constexpr int S = decltype(slot_c)::value;
#pragma unroll
        for (int kk = 0; kk < 4; kk++) rd(Br[kk], b_rd[kk], ic<S * SLOT>{});
    };

    f32x16 acc[MT][NT];
#pragma unroll
    for (int a = 0; a < MT; a++)
#pragma unroll
        for (int b = 0; b < NT; b++)
#pragma unroll
            for (int r = 0; r < 16; r++) acc[a][b][r] = 0.f;

    // quadrant (MFMA tiles tm0, tm0+1) x tn: 8 MFMAs, the two accumulators alternate
    auto mfma_quad = [&](auto tm0_c, auto tn_c, f32x4 (&Br)[4]) {
        constexpr int TM0 = decltype(tm0_c)::value, TN = decltype(tn_c)::value;
        __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int kk = 0; kk < 4; kk++)
#pragma unroll
            for (int t = 0; t < 2; t++) {
                if constexpr (ET)
                    acc[TM0 + t][TN] = __builtin_amdgcn_mfma_f32_32x32x16_f16(
                        __builtin_bit_cast(f16x8, Br[kk]), __builtin_bit_cast(f16x8, Ar[t][kk]), acc[TM0 + t][TN], 0, 0, 0);
                else
                    acc[TM0 + t][TN] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(
                        __builtin_bit_cast(bf16x8, Br[kk]), __builtin_bit_cast(bf16x8, Ar[t][kk]), acc[TM0 + t][TN], 0, 0, 0);
            }
        __builtin_amdgcn_s_setprio(0);
    };
    auto barrier = [&]() { asm volatile("s_barrier" ::: "memory"); };
    // the fragment registers of this phase are complete: wait, then pin every later use below the wait
    auto frags_ready = [&]() {
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
    };
    auto dma_wait = [&]() { asm volatile("s_waitcnt vmcnt(10)" ::: "memory"); };

    // ---- prologue: K tiles 0 (slots 0-3: B0 A0 B1 A1) and 1 (slots 4-7: B1 A0 B0 A1), in read order
    stage_B(0, 0, kb);
    stage_A(1, 0, true);
    stage_B(2, 1, kb);
    stage_A(3, 1, true);
    advance_tap();
    stage_B(4, 1, kb + 1);
    stage_A(5, 0, kb + 1 < ke);
    stage_B(6, 0, kb + 1);
    stage_A(7, 1, kb + 1 < ke);
    advance_tap();                                   // -> K tile kb + 2
    if constexpr (SK) {
        if (kb > 0 && !sk_par) {
            // the K head of this tile, published by a workgroup of the launch's first round: one lane polls (bounded),
            // one agent-scope acquire, then plain loads into the accumulators (the DMA above stays in flight: these
            // are ordinary loads the compiler counts itself, issued after it)
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
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     // accumulators and all eight stages landed
        }
    }
    asm volatile("s_waitcnt vmcnt(12)" ::: "memory");      // slots 0, 1 of this wave have landed
    barrier();
    read_B(B0r, ic<0>{});                            // "phase 0": B0 of K tile 0
    if (wm == 1) barrier();                          // the second group runs one barrier behind

    const int iters = (ke - kb + 1) >> 1;
    for (int it = 0; it < iters; it++) {
        const int kt = kb + 2 * it;                  // even K tile of this iteration; kt + 1 the odd one
        const bool odd_ok = kt + 1 < ke;
        // phase 1: A0 of the even tile (slot 1) x B0; stage slot 7 = A1 of the odd tile kt+1 (already there in the
        // first iteration: the prologue staged it, and the tap state stands at tile 2)
        read_A(ic<1>{});
        // (no DMA is issued there: the prologue's eight stages stand for phases -6 .. 1, so vmcnt(10) still means
        // "everything up to five stages back has landed")
        if (it > 0) { stage_A(7, 1, odd_ok); advance_tap(); }
        dma_wait();
        barrier();
        frags_ready();
        mfma_quad(ic<0>{}, ic<0>{}, B0r);
        barrier();
        // phase 2: x B1 (slot 2); stage slot 0 = B0 of tile kt+2
        read_B(B1r, ic<2>{});
        stage_B(0, 0, kt + 2);
        dma_wait();
        barrier();
        frags_ready();
        mfma_quad(ic<0>{}, ic<1>{}, B1r);
        barrier();
        // phase 3: A1 (slot 3) x B1; stage slot 1 = A0 of tile kt+2
        read_A(ic<3>{});
        stage_A(1, 0, kt + 2 < ke);
        dma_wait();
        barrier();
        frags_ready();
        mfma_quad(ic<2>{}, ic<1>{}, B1r);
        barrier();
        // phase 4: A1 x B0; read B1 of the odd tile (slot 4); stage slot 2 = B1 of tile kt+2
        read_B(B1r, ic<4>{});
        stage_B(2, 1, kt + 2);
        dma_wait();
        barrier();
        frags_ready();
        mfma_quad(ic<2>{}, ic<0>{}, B0r);
        barrier();
        // phase 5: A0 of the odd tile (slot 5) x B1; stage slot 3 = A1 of tile kt+2
        read_A(ic<5>{});
        stage_A(3, 1, kt + 2 < ke);
        advance_tap();
        dma_wait();
        barrier();
        frags_ready();
        if (odd_ok) mfma_quad(ic<0>{}, ic<1>{}, B1r);
        barrier();
        // phase 6: x B0 (slot 6); stage slot 4 = B1 of tile kt+3
        read_B(B0r, ic<6>{});
        stage_B(4, 1, kt + 3);
        dma_wait();
        barrier();
        frags_ready();
        if (odd_ok) mfma_quad(ic<0>{}, ic<0>{}, B0r);
        barrier();
        // phase 7: A1 (slot 7) x B0; stage slot 5 = A0 of tile kt+3
        read_A(ic<7>{});
        stage_A(5, 0, kt + 3 < ke);
        dma_wait();
        barrier();
        frags_ready();
        if (odd_ok) mfma_quad(ic<2>{}, ic<0>{}, B0r);
        barrier();
        // phase 8: A1 x B1; read B0 of the next even tile (slot 0); stage slot 6 = B0 of tile kt+3
        read_B(B0r, ic<0>{});
        stage_B(6, 0, kt + 3);
        dma_wait();
        barrier();
        frags_ready();
        if (odd_ok) mfma_quad(ic<2>{}, ic<1>{}, B1r);
        barrier();
    }
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    if (wm == 0) barrier();
    barrier();                                       // every wave is past its last fragment read and DMA: the slabs may land
    if constexpr (SK) {
        if (!finish) {
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
        if (sk_par && sk_nprev > 0) {
            // split-K (launches with fewer tiles than CUs): the pieces of a tile ran side by side from zero; the piece
            // that holds the tile's last K tile adds the partial sums of the others -- the slots before its own, in K
            // order: a fixed association, so the result is reproducible (not the unsplit chain's bits)
            for (int j = sk_nprev; j >= 1; j--) {
                const int sl = sk_slot - j;
                if (tid == 0) {
                    int spins = 0;
                    while (__hip_atomic_load(p.sk_flags + sl, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != p.sk_epoch &&
                           ++spins < p.sk_spin_limit)
                        __builtin_amdgcn_s_sleep(4);
                    // a hand-over that never arrives must not end as a silent wrong result: the host-mapped error word
                    // makes the next launch on any stream (and brcnn_conv_handover_status) return BRCNN_EHANDOVER
                    if (spins >= p.sk_spin_limit)
                        __hip_atomic_store(p.sk_err, p.sk_epoch | 0x80000000u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
                }
                __syncthreads();
                const f32x4* src = reinterpret_cast<const f32x4*>(p.sk_ws) + (size_t)sl * (BM * BN / 4) + wave * 64 + lane;
#pragma unroll
                for (int a = 0; a < MT; a++)
#pragma unroll
                    for (int b = 0; b < NT; b++)
#pragma unroll
                        for (int g = 0; g < 4; g++) {
                            const f32x4 v = src[((a * NT + b) * 4 + g) * (NW * 64)];
                            acc[a][b][4 * g + 0] += v.x; acc[a][b][4 * g + 1] += v.y;
                            acc[a][b][4 * g + 2] += v.z; acc[a][b][4 * g + 3] += v.w;
                        }
            }
        }
    }

    pp_epilogue<RES, OUTF32, ET, MODE, MT, NT, 2>(p, smem, acc, tid, wave, wm, wn, m0, n0, tile_m);
}

template <bool RES, bool OUTF32, int ET, bool DIL, int MODE = 0>
int launch_pp2(ConvParams& p, hipStream_t s) {
    constexpr size_t lds = 8 * SLOT;
    static bool attr_done = false;
    static int num_cus = 0;
    if (!attr_done) {
        BRCNN_HIP_CHECK(hipFuncSetAttribute((const void*)conv_pp_bf16_kernel<RES, OUTF32, ET, DIL, MODE, false>,
                                            hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        BRCNN_HIP_CHECK(hipFuncSetAttribute((const void*)conv_pp_bf16_kernel<RES, OUTF32, ET, DIL, MODE, true>,
                                            hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        int dev = 0;
        hipDeviceProp_t prop;
        BRCNN_HIP_CHECK(hipGetDevice(&dev));
        BRCNN_HIP_CHECK(hipGetDeviceProperties(&prop, dev));
        num_cus = prop.multiProcessorCount;
        attr_done = true;
    }
    // one workgroup per CU (128 KiB of LDS): chained stream-K where the tile count leaves much of the last generation idle
    const int rc = sk_plan_pp(p, num_cus, BM, BN, s);
    if (rc) return rc;
    if (p.sk_wgs > 0) {
        hipLaunchKernelGGL((conv_pp_bf16_kernel<RES, OUTF32, ET, DIL, MODE, true>), dim3(p.sk_wgs), dim3(512), lds, s, p);
    } else {
        hipLaunchKernelGGL((conv_pp_bf16_kernel<RES, OUTF32, ET, DIL, MODE, false>), dim3(p.tiles_m * p.tiles_n), dim3(512), lds, s, p);
    }
    BRCNN_LAUNCH_CHECK();
    return 0;
}

template <bool RES, bool OUTF32, int ET>
int launch_pp(ConvParams& p, hipStream_t s) {
    return p.dilate > 1 ? launch_pp2<RES, OUTF32, ET, true>(p, s) : launch_pp2<RES, OUTF32, ET, false>(p, s);
}

// training epilogues (16-bit result, whole 16-byte channel pieces): MODE 1 dual store, MODE 2 data gradient + the
// producer's BatchNorm backward (conv_igemm_bf16.hip launch2's rules)
template <int ET>
int launch_pp_train(ConvParams& p, hipStream_t s) {
    if (p.out_f32 || (p.Cout & 7) || (p.tail_z && p.tail_mask)) return BRCNN_EINVAL;
    if (p.tail_z) {
        if (p.residual) return BRCNN_EINVAL;
        return p.dilate > 1 ? launch_pp2<false, false, ET, true, 2>(p, s) : launch_pp2<false, false, ET, false, 2>(p, s);
    }
    if (p.dilate > 1) return BRCNN_EINVAL;
    return p.residual ? launch_pp2<true, false, ET, false, 1>(p, s) : launch_pp2<false, false, ET, false, 1>(p, s);
}

}  // namespace

namespace brcnn_conv {
// 256 x 256 tile, eight-phase schedule; plain epilogue (scale / shift, residual, ReLU, bf16 / fp16 or fp32 result)
// and the training epilogues of the 16-bit backbone layers (dual store; data gradient + BatchNorm backward)
int dispatch_conv_pp_bf16(ConvParams& p, hipStream_t s) {
    if (p.K < 2 * BKE || (p.K % BKE) || p.KH * p.KW > 32) return BRCNN_EINVAL;
    p.tiles_m = (p.M + BM - 1) / BM;
    p.tiles_n = (p.Cout + BN - 1) / BN;
    if (p.z_out || p.tail_z) return p.f16 ? launch_pp_train<1>(p, s) : launch_pp_train<0>(p, s);
    if (p.f16) {
        if (p.out_f32) return p.residual ? launch_pp<true, true, 1>(p, s) : launch_pp<false, true, 1>(p, s);
        return p.residual ? launch_pp<true, false, 1>(p, s) : launch_pp<false, false, 1>(p, s);
    }
    if (p.out_f32) return p.residual ? launch_pp<true, true, 0>(p, s) : launch_pp<false, true, 0>(p, s);
    return p.residual ? launch_pp<true, false, 0>(p, s) : launch_pp<false, false, 0>(p, s);
}
}  // namespace brcnn_conv
