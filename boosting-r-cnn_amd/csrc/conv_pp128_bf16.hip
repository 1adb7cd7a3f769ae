// 16-bit implicit-GEMM convolution on a 256 x 128 x 64 tile with the two-group ("ping-pong") schedule of
// conv_pp_bf16.hip, for the layers that kernel's 256 x 256 tile does not serve (round 6):
//   * 128 output channels (stage 2 of the ResNet: its 3x3 convs and their data gradients ran on the two-buffer
//     128 x 128 kernel at MfmaUtil 0.15 -- every wave in the same phase, matrix pipes idle during the fragment reads);
//   * 256 / 512 output channels on maps with too few 256-row tiles for 256 CUs (stage 3: M = 33 600 is 132 tiles of
//     256 x 256 -- half the chip -- and 264 of 256 x 128; stage 4: 66 -> 132 / 264).
// Same decomposition, operand layout and numerics as the other 16-bit kernels (D[m,co] = sum_k A[m,k] W[co,k], LDS-DMA
// staging of 128-byte rows with the source-side XOR swizzle, v_mfma_f32_32x32x16 as D^T = W A^T, K tiles in the unsplit
// chain's order), so every output is bit-identical to theirs.  What differs:
//   * 8 waves = 4 row groups (64 rows each) x 2 column strips (64 columns each); a wave owns 64 x 64 outputs (2 x 2 MFMA
//     tiles, 64 accumulator registers).  The two waves of a SIMD (wave w and w + 4: row groups {0,1} / {2,3}) run one
//     barrier apart: while one issues its MFMAs the other reads fragments and issues LDS-DMA;
//   * a K tile is THREE full 16 KB slots -- B (128 output columns x 64 k), A0 and A1 (MFMA tile 0 / 1 of the four row
//     groups: 4 x 32 rows x 64 k) -- and TWO phases of 8 MFMAs: phase a = A0 x {B0, B1}, phase b = A1 x {B0, B1} (a wave's
//     two column tiles).  With the eight-phase kernel's half-tile pairing a 64-row wave tile would have meant phases of
//     4 MFMAs (128 matrix-pipe cycles) between two barriers each; here a phase keeps its 256 cycles;
//   * the LDS holds three K tiles as nine slots (144 KB).  Phase a of K tile u reads A0(u) and stages A0(u + 2) into the
//     slot read two phases earlier; phase b reads A1(u) and B(u + 1) (into the other B register set) and stages
//     A1(u + 2) and B(u + 3).  Four to five slots (64 - 80 KB per CU) are in flight; the only waits on the DMA queue are
//     counted (`vmcnt(8)` in phase a, `vmcnt(10)` in phase b), never a drain;
//   * per K tile a workgroup stages 48 KB for 128 MFMAs (the 256 x 256 tile: 64 KB for 256): 0.75 of the CU's LDS-DMA
//     rate at full matrix-pipe rate instead of 0.5 -- the price of the narrower tile.
// Hazards (cdna_hip_programming.md 5): the slot read in phase q was staged in phase q-4 (A0) / q-4 (A1, B) and waited
// for (by the staging waves) in phase q-1 before that phase's first barrier; it is overwritten in phase q+2, two
// barriers after the slower group's read was waited for.
#include <type_traits>
#include "conv_common.h"
#include "conv_pp_epilogue.h"

namespace {
using namespace brcnn_conv;

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef __attribute__((address_space(3))) void* lds_ptr_t;
constexpr int BKE = 64;
constexpr int MT = 2, NT = 2, WG = 4, WNW = 2, NW = 8;
constexpr int BM = WG * 32 * MT, BN = WNW * 32 * NT;       // 256 x 128
constexpr int SLOT = 16384;         // bytes of one slot: 128 rows x 128 B
constexpr int NSLOT = 9;

template <int N> using ic = std::integral_constant<int, N>;

// DIL: zero-stuffed input (p.dilate > 1, the data gradient of a strided conv): the general address form
// SK: chained stream-K schedule (ConvParams::sk_*, conv_igemm_bf16.hip): the workgroup's item (tile, K tiles [kb, ke),
// hand-over slot) comes from the launch's table; a K head stores its accumulators, a K tail starts from them.
template <bool RES, bool OUTF32, int ET, bool DIL, int MODE = 0, bool SK = false>
__global__ __launch_bounds__(512, 2) void conv_pp128_bf16_kernel(ConvParams p) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int nk = p.K / BKE;
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1;
    const int pg = wave >> 2;                       // phase group: the two waves of a SIMD run one barrier apart
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

    // ---- staging: per slot a wave fills LDS rows [16 wave, 16 wave + 16) with two DMA instructions (8 rows x 128 B
    // each).  LDS row R of slot A_h holds tile row (R >> 5) * 64 + h * 32 + (R & 31): MFMA tile h of the four row groups;
    // LDS row R of a B slot holds output column n0 + R.
    const int rg = lane >> 3, pc = lane & 7;
    // Per staged A row: the byte offset of its (kh, kw) = (0, 0) tap, the byte stride of an input row, and one validity
    // bit per filter tap (conv_pp_bf16.hip); zero-stuffed inputs keep the general form.
    int a_off[2][2], b_off[2], lc[2];
    int a_ws[DIL ? 1 : 2][2];           // !DIL: byte stride of an input row
    unsigned a_mask[2][2];
    int a_hw[DIL ? 2 : 1][2], a_HW[DIL ? 2 : 1][2];         // DIL only: packed (hi0, wi0), (H, W)
#pragma unroll
    for (int j = 0; j < 2; j++) lc[j] = (pc ^ ((4 * j + (lane >> 4)) & 7)) * 8;
#pragma unroll
    for (int j = 0; j < 2; j++) {
        const int R = 16 * wave + 8 * j + rg;
        const int co = n0 + R;
        b_off[j] = (co < p.Cout) ? (co * p.K + lc[j]) * 2 : OOB;
    }
#pragma unroll
    for (int h = 0; h < 2; h++)
#pragma unroll
        for (int j = 0; j < 2; j++) {
            const int R = 16 * wave + 8 * j + rg;
            const int m = m0 + (R >> 5) * 64 + h * 32 + (R & 31);
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
        }
    const unsigned lds0 = (unsigned)(size_t)(lds_ptr_t)smem;
    const unsigned st_dst = lds0 + (unsigned)wave * 2048u;        // this wave's rows inside a slot

    // K tiles are visited channel chunk by channel chunk, the filter taps INSIDE a chunk (conv_pp_bf16.hip): K tile kt =
    // chunk kt / T, tap kt % T; the weights stay [Cout][kh][kw][ci]: a tile's column is tap Cin + chunk BKE.
    const int T = p.KH * p.KW;
    const unsigned Tmagic = T > 1 ? 0xFFFFFFFFu / (unsigned)T + 1u : 0u;
    auto chunk_of = [&](int kt) { return T > 1 ? (int)__umulhi((unsigned)kt, Tmagic) : kt; };   // kt / T, kt < 2^16
    int tA_ci0 = 0, tA_kh = 0, tA_kw = 0;      // filter tap / channel offset of the K tile whose A slots are staged next
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
    auto stage_B = [&](int slot, int kt) {
        const int cc = chunk_of(kt);
        const int koff = ((kt - cc * T) * p.Cin + cc * BKE) * 2;       // byte offset of the K tile inside a weight row
#pragma unroll
        for (int j = 0; j < 2; j++) {
            // (an out-of-range row offset plus the K offset stays out of range and below 2^32)
            const int off = kt < ke ? (int)((unsigned)b_off[j] + (unsigned)koff) : OOB;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc_w, (lds_ptr_t)(size_t)(st_dst + slot * SLOT + j * 1024), 16, off, 0, 0, 0);
        }
    };

    // ---- fragment reads: lane (li, lh) reads row li of a 32-row MFMA tile, chunk (2 kk + lh) ^ ((li >> 1) & 7)
    const int sw = (li >> 1) & 7;
    unsigned a_rd[4], b_rd[4];
#pragma unroll
    for (int kk = 0; kk < 4; kk++) {
        const unsigned ch = (unsigned)(((2 * kk + lh) ^ sw) * 16);
        a_rd[kk] = lds0 + (unsigned)(wm * 32 + li) * 128u + ch;
        b_rd[kk] = lds0 + (unsigned)(wn * 64 + li) * 128u + ch;
    }
    f32x4 Ar[2][4], Br[2][2][4];        // A tile h; B register set s, column tile tn
    // slots 4..8 lie beyond the 16-bit offset field: their reads add 64 KiB / 128 KiB to the address register
    auto rd = [&](f32x4& d, unsigned addr, auto off_c) {
        constexpr int OFF = decltype(off_c)::value;
        if constexpr (OFF < 65536) {
            asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(d) : "v"(addr), "n"(OFF) : "memory");
        } else if constexpr (OFF < 131072) {
            const unsigned hi = addr + 65536u;
            asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(d) : "v"(hi), "n"(OFF - 65536) : "memory");
        } else {
            const unsigned hi = addr + 131072u;
            asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(d) : "v"(hi), "n"(OFF - 131072) : "memory");
        }
    };
    auto read_A = [&](auto h_c, auto slot_c) {
        constexpr int H = decltype(h_c)::value, S = decltype(slot_c)::value;
#pragma unroll
        for (int kk = 0; kk < 4; kk++) rd(Ar[H][kk], a_rd[kk], ic<S * SLOT>{});
    };
    auto read_B = [&](auto set_c, auto slot_c) {
        constexpr int BS = decltype(set_c)::value, S = decltype(slot_c)::value;
#pragma unroll
        for (int kk = 0; kk < 4; kk++) {
            rd(Br[BS][0][kk], b_rd[kk], ic<S * SLOT>{});
            rd(Br[BS][1][kk], b_rd[kk], ic<S * SLOT + 4096>{});
        }
    };

    f32x16 acc[MT][NT];
#pragma unroll
    for (int a = 0; a < MT; a++)
#pragma unroll
        for (int b = 0; b < NT; b++)
#pragma unroll
            for (int r = 0; r < 16; r++) acc[a][b][r] = 0.f;

    // A tile h x both column tiles of B register set s: 8 MFMAs, the two accumulators alternate
    auto mfma_pair = [&](auto h_c, auto set_c) {
        constexpr int H = decltype(h_c)::value, BS = decltype(set_c)::value;
        __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int kk = 0; kk < 4; kk++)
#pragma unroll
            for (int t = 0; t < 2; t++) {
                if constexpr (ET)
                    acc[H][t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(
                        __builtin_bit_cast(f16x8, Br[BS][t][kk]), __builtin_bit_cast(f16x8, Ar[H][kk]), acc[H][t], 0, 0, 0);
                else
                    acc[H][t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(
                        __builtin_bit_cast(bf16x8, Br[BS][t][kk]), __builtin_bit_cast(bf16x8, Ar[H][kk]), acc[H][t], 0, 0, 0);
            }
        __builtin_amdgcn_s_setprio(0);
    };
    auto barrier = [&]() { asm volatile("s_barrier" ::: "memory"); };
    // the fragment registers of this phase are complete: wait, then pin every later use below the wait
    auto frags_ready = [&]() {
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
    };

    // ---- prologue: B(0) A0(0) A1(0) B(1) A0(1) A1(1) B(2), in read order (K tile t lives in slots 3 (t % 3) + {0, 1, 2})
    stage_B(0, kb);
    stage_A(1, 0, true);
    stage_A(2, 1, true);
    advance_tap();
    stage_B(3, kb + 1);
    stage_A(4, 0, kb + 1 < ke);
    stage_A(5, 1, kb + 1 < ke);
    advance_tap();                                   // -> K tile kb + 2
    stage_B(6, kb + 2);
    if constexpr (SK) {
        if (kb > 0 && !sk_par) {
            // the K head of this tile, published by a workgroup of the launch's first round: one lane polls (bounded),
            // one agent-scope acquire, then plain loads into the accumulators
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
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     // accumulators and all seven stages landed
        }
    }
    asm volatile("s_waitcnt vmcnt(10)" ::: "memory");      // slots 0, 1 of this wave have landed
    barrier();
    read_B(ic<0>{}, ic<0>{});                        // "phase 0": B of K tile 0 into register set 0
    if (pg == 1) barrier();                          // the second group runs one barrier behind

    // K tile u = kb + 6 it + J: phase a reads A0(u) and stages A0(u + 2); phase b reads A1(u), B(u + 1) and stages
    // A1(u + 2), B(u + 3).  Six K tiles per iteration make the slot (u % 3) and the B register set (u % 2) constants.
    auto ktile = [&](auto j_c, int kt0, bool first) {
        constexpr int J = decltype(j_c)::value;
        constexpr int S0 = 3 * (J % 3), S2 = 3 * ((J + 2) % 3), S1 = 3 * ((J + 1) % 3);
        constexpr int BS = J % 2;
        const int u = kt0 + J;
        const bool live = u < ke;
        // phase a
        read_A(ic<0>{}, ic<S0 + 1>{});
        // (the prologue has staged A0 / A1 of K tiles 0 and 1 and B of 0..2, and the tap state stands at tile 2)
        stage_A(S2 + 1, 0, u + 2 < ke);
        asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
        barrier();
        frags_ready();
        if (live) mfma_pair(ic<0>{}, ic<BS>{});
        barrier();
        // phase b
        read_A(ic<1>{}, ic<S0 + 2>{});
        read_B(ic<1 - BS>{}, ic<S1>{});
        stage_A(S2 + 2, 1, u + 2 < ke);
        advance_tap();
        stage_B(S0, u + 3);
        asm volatile("s_waitcnt vmcnt(10)" ::: "memory");
        barrier();
        frags_ready();
        if (live) mfma_pair(ic<1>{}, ic<BS>{});
        barrier();
        (void)first;
    };
    const int iters = (ke - kb + 5) / 6;
    for (int it = 0; it < iters; it++) {
        const int kt0 = kb + 6 * it;
        ktile(ic<0>{}, kt0, it == 0);
        ktile(ic<1>{}, kt0, false);
        ktile(ic<2>{}, kt0, false);
        ktile(ic<3>{}, kt0, false);
        ktile(ic<4>{}, kt0, false);
        ktile(ic<5>{}, kt0, false);
    }
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    if (pg == 0) barrier();
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
            // split-K (launches with fewer tiles than CUs): the piece that holds the tile's last K tile adds the partial
            // sums of the others -- the slots before its own, in K order: a fixed association (conv_pp_bf16.hip)
            for (int j = sk_nprev; j >= 1; j--) {
                const int sl = sk_slot - j;
                if (tid == 0) {
                    int spins = 0;
                    while (__hip_atomic_load(p.sk_flags + sl, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != p.sk_epoch &&
                           ++spins < p.sk_spin_limit)
                        __builtin_amdgcn_s_sleep(4);
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
    pp_epilogue<RES, OUTF32, ET, MODE, MT, NT, WG>(p, smem, acc, tid, wave, wm, wn, m0, n0, tile_m);
}

template <bool RES, bool OUTF32, int ET, bool DIL, int MODE = 0>
int launch_pp2(ConvParams& p, hipStream_t s) {
    constexpr size_t lds = NSLOT * SLOT;
    static bool attr_done = false;
    static int num_cus = 0;
    if (!attr_done) {
        BRCNN_HIP_CHECK(hipFuncSetAttribute((const void*)conv_pp128_bf16_kernel<RES, OUTF32, ET, DIL, MODE, false>,
                                            hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        BRCNN_HIP_CHECK(hipFuncSetAttribute((const void*)conv_pp128_bf16_kernel<RES, OUTF32, ET, DIL, MODE, true>,
                                            hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        int dev = 0;
        hipDeviceProp_t prop;
        BRCNN_HIP_CHECK(hipGetDevice(&dev));
        BRCNN_HIP_CHECK(hipGetDeviceProperties(&prop, dev));
        num_cus = prop.multiProcessorCount;
        attr_done = true;
    }
    // one workgroup per CU (144 KiB of LDS): chained stream-K where the tile count leaves much of the last generation idle
    const int rc = sk_plan_pp(p, num_cus, BM, BN, s);
    if (rc) return rc;
    if (p.sk_wgs > 0) {
        hipLaunchKernelGGL((conv_pp128_bf16_kernel<RES, OUTF32, ET, DIL, MODE, true>), dim3(p.sk_wgs), dim3(512), lds, s, p);
    } else {
        hipLaunchKernelGGL((conv_pp128_bf16_kernel<RES, OUTF32, ET, DIL, MODE, false>), dim3(p.tiles_m * p.tiles_n), dim3(512), lds, s, p);
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
// 256 x 128 tile, two-group schedule with three slots per K tile; plain epilogue (scale / shift, residual, ReLU, bf16 /
// fp16 or fp32 result) and the training epilogues of the 16-bit backbone layers
int dispatch_conv_pp128_bf16(ConvParams& p, hipStream_t s) {
    if (p.K < 3 * BKE || (p.K % BKE) || p.KH * p.KW > 32) return BRCNN_EINVAL;
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
