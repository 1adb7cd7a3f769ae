// Persistent streaming kernel for the short-K 1x1 layers (16-bit operands): y[M, N] = act(x[M, K] W[N, K]^T * scale + shift
// + residual), K = 64 or 128, stride 1, one map (the "plain" layers of conv_igemm_bf16.hip).
//
// Why: with one output tile per workgroup these layers run set-up, one operand latency, a handful of MFMAs and the
// read-out in series; after round 4's instruction diet they sit at 46-81 % of the streaming rate with half of the wave
// cycles spent waiting (profiles/r04_notes.md).  Here a workgroup OWNS a 128-column block of the output and walks down a
// strip of 64-row tiles: the weight block is staged into LDS once, the x tiles arrive through a ring of D slots filled
// D-1 tiles ahead by LDS-DMA, scale / shift are fetched once, and the read-out of tile i (LDS transposition, residual,
// conversion, stores) runs while the loads of tiles i+1 .. i+D-1 are in flight.  One counted vmcnt wait and one
// barrier per tile.  The arithmetic per output element, the MFMA operand layout and the K order are those of
// conv_igemm_bf16.hip: same bits (tests/test_bf16_gpu.py).
//
// Every LDS access inside the tile loop is inline asm: a compiler-visible LDS access next to `buffer_load ... lds`
// makes the wait-count pass drain vmcnt in front of it (it cannot tell the ring from the slabs), which would serialise
// the prefetch with the read-out.  Loads into registers (residual) are ordinary code: the compiler counts those itself.
#include "conv_common.h"

namespace {
using namespace brcnn_conv;

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef __attribute__((address_space(3))) void* lds_ptr_t;

constexpr int SBM = 64, SBN = 128;             // rows per step, columns per workgroup
constexpr int SPITCH = 68;                     // floats per slab row (32 * NT + 4)
constexpr int SLAB_BYTES = 32 * SPITCH * 4;    // one wave's transposition slab

// ring depths: as deep as the 160 KB of LDS allow -- the tile time is set by bytes in flight per CU over the loaded memory
// latency (~4 us at 4.5 TB/s: first version with 70 KB in flight ran at 2 us per tile)
template <int KT, bool RES> constexpr int ring_depth() { return KT == 1 ? (RES ? 4 : 3) : (RES ? 3 : 5); }
// the residual tiles (64 rows x 256 bytes) come through their own LDS-DMA ring: registers would expose the load latency
// in every tile (requested at the top of tile i, needed a few hundred cycles later), and a compiler-visible load makes the
// wait-count pass drain the whole queue at its use
template <int KT> constexpr int res_depth() { return KT == 1 ? 4 : 2; }
constexpr int R_SLOT = SBM * SBN * 2;
template <int KT, bool RES> constexpr size_t stream_lds_bytes() {
    return (size_t)KT * SBN * 128 + (size_t)ring_depth<KT, RES>() * KT * SBM * 128 + 4 * SLAB_BYTES + (RES ? (size_t)res_depth<KT>() * R_SLOT : 0);
}

template <int N> __device__ __forceinline__ void wait_vmcnt() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }

// KT: K / 64.  RES: residual operand.  ET: 0 bf16, 1 fp16.
template <int KT, bool RES, int ET>
__global__ __launch_bounds__(256, 1) void conv1x1_stream_kernel(ConvParams p) {
    constexpr int D = ring_depth<KT, RES>(), DR = res_depth<KT>();
    constexpr int PA = 2 * KT;                  // LDS-DMA pieces of one x tile per wave
    constexpr int PR = RES ? 4 : 0;             // ... of one residual tile
    constexpr int PS = 4;                       // stores per wave and tile
    // issue order of tile i's iteration: x pieces of tile i+D-1, residual pieces of tile i+DR-1, MFMA, read-out, stores.
    // Vector memory operations issued BEHIND the pieces tile i needs (they may still be in flight at the top of iteration i):
    constexpr int BEHIND_A = PR + PS + (D - 2) * (PA + PR + PS);
    constexpr int BEHIND_R = PS + (DR - 2) * (PA + PR + PS);
    constexpr int STEADY = (RES && BEHIND_R < BEHIND_A) ? BEHIND_R : BEHIND_A;
    constexpr int DEEP = (RES && DR > D) ? DR : D;              // iterations before / after which the counts above hold
    static_assert(STEADY < 64, "vmcnt is a 6-bit counter");
    extern __shared__ __attribute__((aligned(16))) float smem[];
    char* lds = reinterpret_cast<char*>(smem);
    constexpr int W_BYTES = KT * SBN * 128, A_SLOT = KT * SBM * 128;
    const unsigned ws_base = (unsigned)(size_t)(lds_ptr_t)lds;
    const unsigned as_base = ws_base + W_BYTES;
    const unsigned slab_base = as_base + D * A_SLOT;
    const unsigned rs_base = slab_base + 4 * SLAB_BYTES;
    constexpr int RS_OFF = W_BYTES + D * A_SLOT + 4 * SLAB_BYTES;

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1;
    const int li = lane & 31, lh = lane >> 5;

    // blockIdx -> (column block, strip): the column blocks of one strip sit on the same XCD (blockIdx % 8) so that the
    // second reader of an x tile finds it in that XCD's L2
    const int ncb = p.Cout / SBN;
    const int xcd = blockIdx.x & 7, q = blockIdx.x >> 3;
    const int cb = q % ncb, strip = (q / ncb) * 8 + xcd;
    const int strips = p.st_strips;
    if (strip >= strips) return;
    const int ntile_all = (p.M + SBM - 1) / SBM;
    const int t_begin = (int)((long long)ntile_all * strip / strips), t_end = (int)((long long)ntile_all * (strip + 1) / strips);
    const int nt = t_end - t_begin;
    if (nt <= 0) return;
    const int n0 = cb * SBN;

    const __amdgpu_buffer_rsrc_t rsrc_x = __builtin_amdgcn_make_buffer_rsrc((void*)p.x, 0, (int)p.x_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rsrc_w = __builtin_amdgcn_make_buffer_rsrc((void*)p.w, 0, (int)p.w_bytes, 0x00020000);
    const int rg = lane >> 3, pc = lane & 7;

    // ---- weights of the column block: KT x 16 pieces of 8 rows x 128 bytes, four per wave and K tile
#pragma unroll
    for (int j = 0; j < 4 * KT; j++) {
        const int idx = wave + 4 * j, kt = idx / 16, g = idx % 16;
        const int r = g * 8 + rg;
        const unsigned off = (unsigned)(((n0 + r) * p.K + kt * 64 + (pc ^ ((r >> 1) & 7)) * 8) * 2);
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc_w, (lds_ptr_t)(lds + kt * (SBN * 128) + g * 1024), 16, (int)off, 0, 0, 0);
    }
    // ---- x tiles: a lane stages the same (row, chunk) of every tile; byte offset = tile base + rel
    unsigned a_rel[PA];
    int a_row[PA];
    unsigned a_dst[PA];
#pragma unroll
    for (int j = 0; j < PA; j++) {
        const int idx = wave + 4 * j, kt = idx / 8, g = idx % 8;
        const int r = g * 8 + rg;
        a_row[j] = r;
        a_rel[j] = (unsigned)((r * p.pitch + kt * 64 + (pc ^ ((r >> 1) & 7)) * 8) * 2);
        a_dst[j] = (unsigned)(W_BYTES + kt * (SBM * 128) + g * 1024);
    }
    const unsigned x0_bytes = (unsigned)p.seg_xoff[0] * 2u;
    auto dma_tile = [&](int t, int slot) {
        const int m0 = t * SBM;
        const unsigned base = x0_bytes + (unsigned)m0 * (unsigned)p.pitch * 2u;
#pragma unroll
        for (int j = 0; j < PA; j++) {
            const unsigned off = (m0 + a_row[j] < p.M) ? base + a_rel[j] : (unsigned)OOB;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc_x, (lds_ptr_t)(lds + a_dst[j] + slot * A_SLOT), 16, (int)off, 0, 0, 0);
        }
    };
    // residual tile = 16 pieces of 4 rows x 256 bytes, four per wave; lane -> (row 4q + lane / 16, physical 16-byte chunk
    // lane % 16) fetches logical chunk phys ^ 8 (row & 1): the read-out's 16 lanes of two rows then cover all 64 banks
    const __amdgpu_buffer_rsrc_t rsrc_r = __builtin_amdgcn_make_buffer_rsrc((void*)p.residual, 0, RES ? (int)((unsigned)p.M * (unsigned)p.Cout * 2u) : 0, 0x00020000);
    unsigned r_rel[4];
#pragma unroll
    for (int j = 0; j < 4; j++) {
        const int qd = wave + 4 * j, row = 4 * qd + (lane >> 4), ph = lane & 15;
        r_rel[j] = (unsigned)((row * p.Cout + n0 + ((ph ^ ((row & 1) * 8)) * 8)) * 2);
    }
    auto dma_res = [&](int t, int slot) {
        const unsigned base = (unsigned)(t * SBM) * (unsigned)p.Cout * 2u;
#pragma unroll
        for (int j = 0; j < 4; j++)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc_r, (lds_ptr_t)(lds + RS_OFF + slot * R_SLOT + (wave + 4 * j) * 1024), 16, (int)(base + r_rel[j]), 0, 0, 0);
    };
#pragma unroll
    for (int s = 0; s < D - 1; s++)
        if (s < nt) dma_tile(t_begin + s, s);
    if (RES) {
#pragma unroll
        for (int s = 0; s < DR - 1; s++)
            if (s < nt) dma_res(t_begin + s, s);
    }

    // ---- per-workgroup constants of the read-out: lane -> row rl of an 8-row group, 8 channels at cl
    const int rl = lane >> 3, cl = (lane & 7) * 8;
    const int cw0 = n0 + wn * 64;
    brcnn_f32x2 sc8p[4], sh8p[4];
    {
        float4 a0 = make_float4(1.f, 1.f, 1.f, 1.f), a1 = a0, b0 = make_float4(0.f, 0.f, 0.f, 0.f), b1 = b0;
        if (p.scale) { a0 = *reinterpret_cast<const float4*>(p.scale + cw0 + cl); a1 = *reinterpret_cast<const float4*>(p.scale + cw0 + cl + 4); }
        if (p.shift) { b0 = *reinterpret_cast<const float4*>(p.shift + cw0 + cl); b1 = *reinterpret_cast<const float4*>(p.shift + cw0 + cl + 4); }
        sc8p[0] = brcnn_f32x2{a0.x, a0.y}; sc8p[1] = brcnn_f32x2{a0.z, a0.w}; sc8p[2] = brcnn_f32x2{a1.x, a1.y}; sc8p[3] = brcnn_f32x2{a1.z, a1.w};
        sh8p[0] = brcnn_f32x2{b0.x, b0.y}; sh8p[1] = brcnn_f32x2{b0.z, b0.w}; sh8p[2] = brcnn_f32x2{b1.x, b1.y}; sh8p[3] = brcnn_f32x2{b1.z, b1.w};
    }
    const unsigned floor2 = p.relu ? 0u : 0x80008000u;
    unsigned short* __restrict__ yh = reinterpret_cast<unsigned short*>(p.y);

    // fragment addresses (conv_igemm_bf16.hip's layout: row R, logical 16-byte chunk c at physical c ^ ((R >> 1) & 7))
    const int sw = (li >> 1) & 7;
    unsigned chb[4];
#pragma unroll
    for (int kk = 0; kk < 4; kk++) chb[kk] = (unsigned)(((2 * kk + lh) ^ sw) * 16);
    const unsigned a_lane = as_base + (unsigned)((wm * 32 + li) * 128);
    const unsigned b_lane = ws_base + (unsigned)((wn * 64 + li) * 128);
    const unsigned cs = slab_base + (unsigned)(wave * SLAB_BYTES);
    const unsigned cs_w = cs + (unsigned)(li * (SPITCH * 4) + lh * 16);
    const unsigned cs_r = cs + (unsigned)(rl * (SPITCH * 4) + cl * 4);

    // residual chunk of this lane in a ring slot: row wm * 32 + it * 8 + rl, logical chunk wn * 8 + lane % 8
    const unsigned rs_lane = rs_base + (unsigned)((wm * 32 + rl) * 256 + (((wn * 8 + (lane & 7)) ^ ((rl & 1) * 8)) * 16));
    int slot = 0, rslot = 0;
    for (int i = 0; i < nt; i++) {
        const int t = t_begin + i;
        // tile i's pieces have landed in this wave (everything issued behind them may still be in flight) ...
        if (i >= DEEP - 1 && i + DEEP - 2 < nt) wait_vmcnt<STEADY>();
        else wait_vmcnt<0>();
        // ... and in every wave; nobody reads slot i-1 any more
        asm volatile("s_barrier" ::: "memory");
        const int m_w = t * SBM + wm * 32;                       // first row of this wave
        const bool whole = t * SBM + SBM <= p.M;                 // (only the last tile of the last strip is not)
        if (i + D - 1 < nt) {
            int fill = slot + D - 1;
            if (fill >= D) fill -= D;
            dma_tile(t + D - 1, fill);
        }
        if (RES && i + DR - 1 < nt) {
            int fill = rslot + DR - 1;
            if (fill >= DR) fill -= DR;
            dma_res(t + DR - 1, fill);
        }
        // ---- MFMA: D^T = W x^T, K tiles and 16-wide steps in order
        f32x16 acc[2];
#pragma unroll
        for (int tn = 0; tn < 2; tn++)
#pragma unroll
            for (int r = 0; r < 16; r++) acc[tn][r] = 0.f;
        const unsigned a_cur = a_lane + (unsigned)(slot * A_SLOT);
#pragma unroll
        for (int kt = 0; kt < KT; kt++) {
            f32x4 av[4], bv[4][2];
#pragma unroll
            for (int kk = 0; kk < 4; kk++) {
                asm volatile("ds_read_b128 %0, %1" : "=v"(av[kk]) : "v"(a_cur + kt * (SBM * 128) + chb[kk]) : "memory");
                asm volatile("ds_read_b128 %0, %1" : "=v"(bv[kk][0]) : "v"(b_lane + kt * (SBN * 128) + chb[kk]) : "memory");
                asm volatile("ds_read_b128 %0, %1 offset:4096" : "=v"(bv[kk][1]) : "v"(b_lane + kt * (SBN * 128) + chb[kk]) : "memory");
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
            for (int kk = 0; kk < 4; kk++) {
                asm volatile("" : "+v"(av[kk]), "+v"(bv[kk][0]), "+v"(bv[kk][1]));
#pragma unroll
                for (int tn = 0; tn < 2; tn++)
                    if constexpr (ET)
                        acc[tn] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, bv[kk][tn]), __builtin_bit_cast(f16x8, av[kk]), acc[tn], 0, 0, 0);
                    else
                        acc[tn] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, bv[kk][tn]), __builtin_bit_cast(bf16x8, av[kk]), acc[tn], 0, 0, 0);
            }
        }
        // ---- read-out: accumulators -> the wave's slab (lane = pixel li, 4 channels per group) -> rows of 8 channels
#pragma unroll
        for (int tn = 0; tn < 2; tn++)
#pragma unroll
            for (int g = 0; g < 4; g++) {
                f32x4 v;
                v.x = acc[tn][4 * g + 0]; v.y = acc[tn][4 * g + 1]; v.z = acc[tn][4 * g + 2]; v.w = acc[tn][4 * g + 3];
                asm volatile("ds_write_b128 %0, %1" ::"v"(cs_w + (unsigned)(tn * 128 + g * 32)), "v"(v) : "memory");
            }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_wave_barrier();
        f32x4 lo[4], hi[4];
#pragma unroll
        for (int it = 0; it < 4; it++) {
            asm volatile("ds_read_b128 %0, %1" : "=v"(lo[it]) : "v"(cs_r + (unsigned)(it * 8 * SPITCH * 4)) : "memory");
            asm volatile("ds_read_b128 %0, %1 offset:16" : "=v"(hi[it]) : "v"(cs_r + (unsigned)(it * 8 * SPITCH * 4)) : "memory");
        }
        f32x4 rq[4];
        if (RES) {
#pragma unroll
            for (int it = 0; it < 4; it++)
                asm volatile("ds_read_b128 %0, %1" : "=v"(rq[it]) : "v"(rs_lane + (unsigned)(rslot * R_SLOT + it * 8 * 256)) : "memory");
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
        for (int it = 0; it < 4; it++) asm volatile("" : "+v"(lo[it]), "+v"(hi[it]));
        if (RES) {
#pragma unroll
            for (int it = 0; it < 4; it++) asm volatile("" : "+v"(rq[it]));
        }
        __builtin_amdgcn_wave_barrier();             // the slab is free for the next tile
        unsigned short* __restrict__ yrow = yh + (size_t)(m_w + rl) * p.Cout + cw0 + cl;
#pragma unroll
        for (int it = 0; it < 4; it++) {
            brcnn_f32x2 v[4] = {{lo[it].x, lo[it].y}, {lo[it].z, lo[it].w}, {hi[it].x, hi[it].y}, {hi[it].z, hi[it].w}};
#pragma unroll
            for (int e = 0; e < 4; e++) v[e] = v[e] * sc8p[e] + sh8p[e];
            if (RES) {
                const unsigned rr[4] = {__float_as_uint(rq[it].x), __float_as_uint(rq[it].y), __float_as_uint(rq[it].z), __float_as_uint(rq[it].w)};
#pragma unroll
                for (int e = 0; e < 4; e++) v[e] += brcnn_unpk2<ET>(rr[e]);
            }
            uint4 o;
            o.x = brcnn_relu_pk(brcnn_pk2<ET>(v[0]), floor2);
            o.y = brcnn_relu_pk(brcnn_pk2<ET>(v[1]), floor2);
            o.z = brcnn_relu_pk(brcnn_pk2<ET>(v[2]), floor2);
            o.w = brcnn_relu_pk(brcnn_pk2<ET>(v[3]), floor2);
            if (whole || m_w + it * 8 + rl < p.M) *reinterpret_cast<uint4*>(yrow + (size_t)(it * 8) * p.Cout) = o;
        }
        slot = slot + 1 == D ? 0 : slot + 1;
        rslot = rslot + 1 == DR ? 0 : rslot + 1;
    }
}

template <int KT, bool RES, int ET>
int launch_stream(ConvParams& p, hipStream_t s) {
    constexpr size_t lds = stream_lds_bytes<KT, RES>();
    static bool attr_done = false;
    static int num_cus = 0;
    if (!attr_done) {
        BRCNN_HIP_CHECK(hipFuncSetAttribute((const void*)conv1x1_stream_kernel<KT, RES, ET>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        int dev = 0;
        hipDeviceProp_t prop;
        BRCNN_HIP_CHECK(hipGetDevice(&dev));
        BRCNN_HIP_CHECK(hipGetDeviceProperties(&prop, dev));
        num_cus = prop.multiProcessorCount;
        attr_done = true;
    }
    // one workgroup per CU (LDS), every column block of a strip on one XCD: strips = a multiple of 8 with
    // strips x column blocks <= CUs, at least 4 tiles per strip
    const int ncb = p.Cout / SBN;
    const int ntile = (p.M + SBM - 1) / SBM;
    const int per_cu = (int)(((size_t)160 << 10) / lds);        // workgroups per CU the LDS footprint allows (1 or 2)
    int strips = (num_cus * (per_cu > 2 ? 2 : per_cu) / ncb) / 8 * 8;
    if (strips < 8) strips = 8;
    while (strips > 8 && ntile / strips < 4) strips -= 8;
    p.st_strips = strips;
    hipLaunchKernelGGL((conv1x1_stream_kernel<KT, RES, ET>), dim3(strips * ncb), dim3(256), lds, s, p);
    BRCNN_LAUNCH_CHECK();
    return 0;
}

// tuning hook (brcnn_conv_set_tile_bf16(-15 / -16 / -17)): never / heuristic / wherever the shape allows.  Measured per layer
// (tools/experiments/stream1x1.py, M = 537600 / 134400, bf16): K = 64 no residual N = 256 116-121 -> 81-88 us, K = 128 + residual
// N = 512 84-88 -> 65-70 us, the others within 4 % (K = 64 + residual 138 -> 134, K = 128 no residual 54-57 -> 53-58, N = 128
// 46-47 -> 45-49); bf16 inference step, interleaved on one box: off 6.49, those two cases only 6.44, every eligible layer
// 6.39 ms -- the heuristic takes every eligible layer
int g_stream_mode = 1;

}  // namespace

namespace brcnn_conv {
// 1 launched, 0 not this kernel's shape, < 0 error
int conv1x1_stream_try(ConvParams& p, hipStream_t s, int f16) {
    if (g_stream_mode == 0 || p.no_fast) return 0;
    const bool plain = p.KH == 1 && p.KW == 1 && p.stride == 1 && p.pad == 0 && p.nseg == 1 && p.dilate <= 1;
    if (!plain || p.scatter || p.gstep || p.out_f32 || p.z_out || p.tail_z || p.tail_mask || p.sk_wgs) return 0;
    if ((p.K != 64 && p.K != 128) || (p.Cout % SBN) || p.pitch != p.Cin || p.M < 4096 || (long long)p.M * p.Cout * 2 >= 0x7fffffffLL) return 0;
    if (p.K == 64) {
        if (f16) return (p.residual ? launch_stream<1, true, 1>(p, s) : launch_stream<1, false, 1>(p, s)) ? BRCNN_EINVAL : 1;
        return (p.residual ? launch_stream<1, true, 0>(p, s) : launch_stream<1, false, 0>(p, s)) ? BRCNN_EINVAL : 1;
    }
    if (f16) return (p.residual ? launch_stream<2, true, 1>(p, s) : launch_stream<2, false, 1>(p, s)) ? BRCNN_EINVAL : 1;
    return (p.residual ? launch_stream<2, true, 0>(p, s) : launch_stream<2, false, 0>(p, s)) ? BRCNN_EINVAL : 1;
}
int conv1x1_stream_set(int mode) { g_stream_mode = mode; return 0; }
int tuning_get_persistent_1x1() { return g_stream_mode; }
}  // namespace brcnn_conv
