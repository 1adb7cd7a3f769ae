// bf16 weight gradient of the NHWC convolution / linear layer (BASELINE configs[2]: "bf16 MFMA
// backbone, full train step"):   dW[co, k] (fp32) += sum_m dY[m, co] * A[m, k],  dY / x in bf16.
//
// Same split-M decomposition as the fp32 kernel (conv_wgrad.hip): grid = (co tiles x k tiles) x
// slices of the M = N*Ho*Wo reduction, fp32 atomics into a zero-filled dW.  The difference is
// the operand fetch: v_mfma_f32_32x32x16_bf16 wants 8 CONSECUTIVE reduction elements per lane,
// but both operands are reduction-MAJOR in memory ((M,Cout) and (M,K) rows).  The tiles are
// therefore staged exactly as they lie ([m][64 columns] = 128-byte rows, LDS-DMA, XOR-swizzled
// 16-byte chunks) and read with the gfx950 transposing LDS load `ds_read_b64_tr_b16`: a
// 16-lane group presents the 8-byte pieces of a [4 m][16 columns] block (lane t: row t>>2,
// columns 4(t&3)..+3) and lane t receives column t of the 4 rows.  Two such reads give a lane the
// 8 reduction elements of its column -- the MFMA A (co) and B (k) fragments -- with no
// ds_write / shuffle pass.
#include <cstdlib>
#include <mutex>
#include <vector>
#include "common.h"
#include "conv_common.h"

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) void* lds_ptr_t;
typedef __attribute__((address_space(3))) s16x4* lds_s16x4_t;
constexpr int OOB = 0x7fffffff;
constexpr int TM = 64;          // reduction rows per LDS tile

struct WgradHParams {
    const unsigned short* dy;   // (M, Cout) bf16
    const unsigned short* x;    // (N, H, W, Cin) bf16 segments back to back
    float* dw;                  // (Cout, K) fp32
    int Cin, Cout, KH, KW, stride, pad, M, K;
    int tiles_co, tiles_k, slices, rows_per_slice;
    unsigned dy_bytes, x_bytes;
    int pitch, gstep;       // pixel pitch of x (== Cin unless grouped); grouped: co tile t reads channels [t*gstep, +Cin)
    // slices > 1: every workgroup stores its fp32 tile to slab[(slice * tiles + tile)] in register order (plain 16-byte
    // stores: 3.8 - 4.9 TB/s against 1.1 - 1.3 TB/s for the same volume of fp32 atomics, tools/experiments/ubench/atomic_scope.hip)
    // and wgrad_reduce_kernel adds the slices of a tile in slice order into dw -- a fixed order: the weight gradient is
    // reproducible bit for bit.  slab == nullptr: atomics straight into dw (one slice, or no workspace).
    float* slab;
    int nseg;
    int seg_m0[BRCNN_MAX_LEVELS + 1];
    int seg_H[BRCNN_MAX_LEVELS], seg_W[BRCNN_MAX_LEVELS], seg_Ho[BRCNN_MAX_LEVELS], seg_Wo[BRCNN_MAX_LEVELS];
    long long seg_xoff[BRCNN_MAX_LEVELS];
    unsigned seg_mhw[BRCNN_MAX_LEVELS], seg_shw[BRCNN_MAX_LEVELS], seg_mw[BRCNN_MAX_LEVELS], seg_sw[BRCNN_MAX_LEVELS];
};

__device__ __forceinline__ unsigned fastdiv(unsigned x, unsigned magic, unsigned shift) {
    return (unsigned)(((unsigned long long)__umulhi(x, magic) + x) >> shift);
}

// Output tile (64*WT) x (64*WT); 4 waves 2x2, each WT x WT MFMA tiles.  LDS per operand and
// buffer: WT column blocks of [64 m][64 columns] bf16 (128-byte rows, physical 16-byte chunk
// c' of row r holds logical chunk c' ^ ((r>>1)&7)).
// WG = waves per tile axis: 2 (4 waves, tile 64*WT square, two workgroups per CU) or 4 (16 waves, tile
// 128*WT square, one workgroup per CU -- per MFMA half the LDS-DMA bytes of the 4-wave tile, which is
// what bounds the kernel: a 128 x 128 tile needs the full 64 B/clk/CU of the vector-memory path to
// keep the MFMA pipe busy, a 256 x 256 tile half of it).
template <int WT, int ET = 0, int WG = 2>      // ET: 0 = bf16 operands, 1 = IEEE fp16 (v_mfma_f32_32x32x16_f16)
__global__ __launch_bounds__(64 * WG * WG, WG == 2 ? 2 : 1) void conv_wgrad_bf16_kernel(WgradHParams p) {
    extern __shared__ __attribute__((aligned(16))) unsigned short smem[];
    constexpr int BLK = TM * 64;                     // elements of one [64][64] block
    constexpr int CB = WG * WT / 2 > 0 ? WG * WT / 2 : 1;   // 64-column blocks per operand
    constexpr int NJ = WG == 2 ? 2 : 1;              // 8-row groups of a 64-row tile staged by one wave
    constexpr int CBW = WG == 2 ? CB : CB / 2;       // column blocks staged by one wave (per operand)
    constexpr int TILE = 32 * WT * WG;
    static_assert(WG == 2 || (WG == 4 && WT == 2), "4 waves (2x2) or 16 waves (4x4, 2x2 MFMA tiles each)");
    unsigned short* Ya = smem;                       // [2][CB][64][64]
    unsigned short* Xa = smem + 2 * CB * BLK;        // [2][CB][64][64]

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave / WG, wn = wave % WG;
    const int li = lane & 31, lh = lane >> 5;
    const int rg0 = WG == 2 ? wave * 2 : (wave & 7);           // first row group this wave stages
    const int cbw0 = WG == 2 ? 0 : (wave >> 3) * CBW;          // first column block this wave stages

    const int nwg = p.tiles_co * p.tiles_k * p.slices;
    int b;
    {
        const int q = nwg >> 3, r = nwg & 7, xcd = blockIdx.x & 7, loc = blockIdx.x >> 3;
        b = ((xcd < r) ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + loc;
    }
    const int tiles = p.tiles_co * p.tiles_k;
    const int slice = b / tiles;
    b -= slice * tiles;
    const int tk = b % p.tiles_k, tco = b / p.tiles_k;
    const int co0 = tco * TILE, k0 = tk * TILE;
    const int m_begin = slice * p.rows_per_slice;
    const int m_end = min(p.M, m_begin + p.rows_per_slice);
    if (m_begin >= m_end) return;

    const __amdgpu_buffer_rsrc_t rsrc_y = __builtin_amdgcn_make_buffer_rsrc((void*)p.dy, 0, (int)p.dy_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rsrc_x = __builtin_amdgcn_make_buffer_rsrc((void*)p.x, 0, (int)p.x_bytes, 0x00020000);

    // DMA: one wave instruction fills 8 rows x 128 B of one block; wave w owns row groups 2w, 2w+1
    const int rg_row = lane >> 3, pc = lane & 7;

    f32x16 acc[WT][WT];
#pragma unroll
    for (int a = 0; a < WT; a++)
#pragma unroll
        for (int c = 0; c < WT; c++)
#pragma unroll
            for (int r = 0; r < 16; r++) acc[a][c][r] = 0.f;

    // A lane stages the same two rows of every tile and, per column block, the same 8 columns:
    // the column -> (tap, channel) split is loop invariant, and the row -> input pixel decode
    // (two divisions, a segment search) advances incrementally from tile to tile -- one
    // conditional wrap per 64-row step on maps at least 64 wide; narrower maps and the first row
    // after a map boundary take the full decode.
    int c_yoff[NJ][CBW], c_xoff[NJ][CBW], c_kh[NJ][CBW], c_kw[NJ][CBW];     // -1: column out of range
#pragma unroll
    for (int j = 0; j < NJ; j++) {
        const int row = (rg0 + j) * 8 + rg_row;
        const int lc = (pc ^ ((row >> 1) & 7)) * 8;              // logical column of this lane's chunk
#pragma unroll
        for (int cb = 0; cb < CBW; cb++) {
            const int co = co0 + (cbw0 + cb) * 64 + lc;
            c_yoff[j][cb] = co < p.Cout ? co : -1;
            const int k = k0 + (cbw0 + cb) * 64 + lc;
            c_xoff[j][cb] = -1;
            c_kh[j][cb] = c_kw[j][cb] = 0;
            if (k < p.K) {
                const int tap = k / p.Cin;
                c_kh[j][cb] = tap / p.KW;
                c_kw[j][cb] = tap - c_kh[j][cb] * p.KW;
                c_xoff[j][cb] = k - tap * p.Cin + tco * p.gstep;
            }
        }
    }
    int s_ho[NJ], s_wo[NJ], s_rb[NJ], s_H[NJ], s_W[NJ], s_Ho[NJ], s_Wo[NJ], s_end[NJ] = {};
    auto decode = [&](int m, int j) {
        int sg = 0;
        if (p.nseg > 1) {       // single-map layers keep the geometry in scalar registers
#pragma unroll
            for (int t = 1; t < BRCNN_MAX_LEVELS; t++)
                if (t < p.nseg && m >= p.seg_m0[t]) sg = t;
        }
        const int ml = m - p.seg_m0[sg];
        const int Ho = p.seg_Ho[sg], Wo = p.seg_Wo[sg], H = p.seg_H[sg], W = p.seg_W[sg];
        const int n = (int)fastdiv((unsigned)ml, p.seg_mhw[sg], p.seg_shw[sg]);
        const int rem = ml - n * (Ho * Wo);
        const int ho = (int)fastdiv((unsigned)rem, p.seg_mw[sg], p.seg_sw[sg]);
        s_ho[j] = ho;
        s_wo[j] = rem - ho * Wo;
        s_H[j] = H; s_W[j] = W; s_Ho[j] = Ho; s_Wo[j] = Wo;
        s_rb[j] = (int)p.seg_xoff[sg] + (n * H + ho * p.stride) * W * p.pitch;   // input row ho*stride, column 0
        s_end[j] = p.seg_m0[sg + 1];
    };
    auto dma_tile = [&](int mt, int buf) {
#pragma unroll
        for (int j = 0; j < NJ; j++) {
            const int m = mt + (rg0 + j) * 8 + rg_row;
            const bool m_ok = m < m_end;
            if (m_ok) {
                if (m >= s_end[j] || s_Wo[j] < TM) {
                    decode(m, j);
                } else {                        // same map, 64 output pixels further
                    int wo = s_wo[j] + TM;
                    if (wo >= s_Wo[j]) {
                        wo -= s_Wo[j];
                        int ho = s_ho[j] + 1;
                        int rb = s_rb[j] + p.stride * s_W[j] * p.pitch;
                        if (ho == s_Ho[j]) {        // next image of the map
                            ho = 0;
                            rb += (s_H[j] - s_Ho[j] * p.stride) * s_W[j] * p.pitch;
                        }
                        s_ho[j] = ho;
                        s_rb[j] = rb;
                    }
                    s_wo[j] = wo;
                }
            }
            const int hi0 = s_ho[j] * p.stride - p.pad, wi0 = s_wo[j] * p.stride - p.pad;
#pragma unroll
            for (int cb = 0; cb < CBW; cb++) {
                const int offy = (m_ok && c_yoff[j][cb] >= 0) ? (m * p.Cout + c_yoff[j][cb]) * 2 : OOB;
                int offx = OOB;
                if (m_ok && c_xoff[j][cb] >= 0) {
                    const int hi = hi0 + c_kh[j][cb], wi = wi0 + c_kw[j][cb];
                    if ((unsigned)hi < (unsigned)s_H[j] && (unsigned)wi < (unsigned)s_W[j])
                        offx = (s_rb[j] + ((c_kh[j][cb] - p.pad) * s_W[j] + wi) * p.pitch + c_xoff[j][cb]) * 2;
                }
                unsigned short* dy_dst = Ya + (buf * CB + cbw0 + cb) * BLK + (rg0 + j) * 8 * 64;
                unsigned short* x_dst = Xa + (buf * CB + cbw0 + cb) * BLK + (rg0 + j) * 8 * 64;
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc_y, (lds_ptr_t)dy_dst, 16, offy, 0, 0, 0);
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc_x, (lds_ptr_t)x_dst, 16, offx, 0, 0, 0);
            }
        }
    };

    // Transposing read of one 8-element fragment: 32 columns starting at tile-local column
    // `col0`, reduction rows ks*16 + 8*lh + (0..7), as two `ds_read_b64_tr_b16` (rows +0..3 and
    // +4..7).  The reads are inline asm with explicit lgkmcnt waits: a compiler-visible LDS read
    // after `buffer_load ... lds` gets `s_waitcnt vmcnt(0)` put in front of it, which would
    // serialise the prefetch of the next 64 rows with this tile's MFMAs in every wave.
    // Per lane the address of (operand, MFMA tile, half) is loop invariant up to the buffer
    // offset; the ks step is an immediate (16 rows x 128 B).
    const int g = lane >> 4, t = lane & 15;
    auto frag_addr = [&](const unsigned short* base, int col0, int q) -> unsigned {
        const int col = col0 + 16 * (g & 1) + 4 * (t & 3);      // tile-local column of this lane's piece
        const int cc = col & 63;
        const int row = 8 * (g >> 1) + 4 * q + (t >> 2);
        const int off = (col >> 6) * BLK + row * 64 + (((cc >> 3) ^ ((row >> 1) & 7)) << 3) + (cc & 4);
        return (unsigned)(size_t)(lds_ptr_t)(base + off);
    };
    unsigned ya_addr[WT][2], xa_addr[WT][2];
#pragma unroll
    for (int a = 0; a < WT; a++)
#pragma unroll
        for (int q = 0; q < 2; q++) {
            ya_addr[a][q] = frag_addr(Ya, (wm * WT + a) * 32, q);
            xa_addr[a][q] = frag_addr(Xa, (wn * WT + a) * 32, q);
        }
    s16x4 fy[2][WT][2], fx[2][WT][2];       // [register slot][MFMA tile][half]
    auto tr_read = [&](s16x4& d, unsigned addr, int ks) {
        if (ks == 0) asm volatile("ds_read_b64_tr_b16 %0, %1" : "=v"(d) : "v"(addr) : "memory");
        else if (ks == 1) asm volatile("ds_read_b64_tr_b16 %0, %1 offset:2048" : "=v"(d) : "v"(addr) : "memory");
        else if (ks == 2) asm volatile("ds_read_b64_tr_b16 %0, %1 offset:4096" : "=v"(d) : "v"(addr) : "memory");
        else asm volatile("ds_read_b64_tr_b16 %0, %1 offset:6144" : "=v"(d) : "v"(addr) : "memory");
    };
    static_assert(TM == 64 && (WT == 1 || WT == 2), "fragment readers: 4 ks steps, 1 or 2 MFMA tiles per wave and axis");
    auto frag_read = [&](int slot, int ks, unsigned boff) {
#pragma unroll
        for (int a = 0; a < WT; a++)
#pragma unroll
            for (int q = 0; q < 2; q++) {
                tr_read(fy[slot][a][q], ya_addr[a][q] + boff, ks);
                tr_read(fx[slot][a][q], xa_addr[a][q] + boff, ks);
            }
    };
    auto frag_wait = [&](int slot) {
        if constexpr (WT == 2)
            asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(fy[slot][0][0]), "+v"(fy[slot][0][1]), "+v"(fy[slot][1][0]), "+v"(fy[slot][1][1]),
                         "+v"(fx[slot][0][0]), "+v"(fx[slot][0][1]), "+v"(fx[slot][1][0]), "+v"(fx[slot][1][1]) :: "memory");
        else
            asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(fy[slot][0][0]), "+v"(fy[slot][0][1]), "+v"(fx[slot][0][0]), "+v"(fx[slot][0][1]) :: "memory");
    };
    auto pack = [&](const s16x4& lo, const s16x4& hi) -> bf16x8 {
        union { s16x4 h[2]; bf16x8 f; } u;
        u.h[0] = lo;
        u.h[1] = hi;
        return u.f;
    };

    dma_tile(m_begin, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    int cur = 0;
    for (int mt = m_begin; mt < m_end; mt += TM) {
        if (mt + TM < m_end) dma_tile(mt + TM, cur ^ 1);
        const unsigned boff = cur * (CB * BLK * 2);
        frag_read(0, 0, boff);
        frag_wait(0);
#pragma unroll
        for (int ks = 0; ks < TM / 16; ks++) {
            const int sl = ks & 1;
            if (ks + 1 < TM / 16) frag_read(sl ^ 1, ks + 1, boff);
#pragma unroll
            for (int a = 0; a < WT; a++)
#pragma unroll
                for (int c = 0; c < WT; c++)
                    if constexpr (ET)
                        acc[a][c] = __builtin_amdgcn_mfma_f32_32x32x16_f16(
                            __builtin_bit_cast(f16x8, pack(fy[sl][a][0], fy[sl][a][1])),
                            __builtin_bit_cast(f16x8, pack(fx[sl][c][0], fx[sl][c][1])), acc[a][c], 0, 0, 0);
                    else
                        acc[a][c] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(pack(fy[sl][a][0], fy[sl][a][1]),
                                                                            pack(fx[sl][c][0], fx[sl][c][1]), acc[a][c], 0, 0, 0);
            if (ks + 1 < TM / 16) frag_wait(sl ^ 1);
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        cur ^= 1;
    }

    if (p.slab) {
        typedef float f32x4 __attribute__((ext_vector_type(4)));
        f32x4* dst = reinterpret_cast<f32x4*>(p.slab) + ((size_t)slice * tiles + b) * (size_t)(TILE * TILE / 4) + wave * 64 + lane;
#pragma unroll
        for (int a = 0; a < WT; a++)
#pragma unroll
            for (int c = 0; c < WT; c++)
#pragma unroll
                for (int g4 = 0; g4 < 4; g4++) {
                    f32x4 v;
                    v.x = acc[a][c][4 * g4 + 0]; v.y = acc[a][c][4 * g4 + 1];
                    v.z = acc[a][c][4 * g4 + 2]; v.w = acc[a][c][4 * g4 + 3];
                    dst[((a * WT + c) * 4 + g4) * (WG * WG * 64)] = v;
                }
        return;
    }
    // D[row = co][col = k]:  col = lane&31, row = (r&3) + 8*(r>>2) + 4*(lane>>5)
#pragma unroll
    for (int c = 0; c < WT; c++) {
        const int kk = k0 + (wn * WT + c) * 32 + li;
        if (kk >= p.K) continue;
#pragma unroll
        for (int a = 0; a < WT; a++)
#pragma unroll
            for (int r = 0; r < 16; r++) {
                const int co = co0 + (wm * WT + a) * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
                if (co < p.Cout) atomicAdd(p.dw + (size_t)co * p.K + kk, acc[a][c][r]);
            }
    }
}

// second stage of the sliced weight gradient: workgroup (tile, v) adds register group v = (a, c, g4) of the tile's
// slabs over `count` slices first, first + stride, ... in that order.  TO_DW: into dw -- the thread layout is the MFMA
// accumulator's: thread (wave, lane) of the first stage owns rows co = (wm WT + a) 32 + 8 g4 + 4 (lane >> 5) + (0..3),
// column k = (wn WT + c) 32 + (lane & 31).  !TO_DW (many slices: a first pass over groups of slices, blockIdx.y = group):
// the sum replaces the group's first slab.
template <int WT, int WG, bool TO_DW>
__global__ __launch_bounds__(64 * WG * WG) void wgrad_reduce_kernel(float* __restrict__ slab, float* __restrict__ dw,
                                                                   int tiles_k, int tiles, int slices, int stride, int group,
                                                                   int Cout, int K) {
    typedef float f32x4 __attribute__((ext_vector_type(4)));
    constexpr int TILE = 32 * WT * WG;
    const int tile = blockIdx.x / (WT * WT * 4), v = blockIdx.x % (WT * WT * 4);
    const int tid = threadIdx.x;
    const int first = blockIdx.y * group * stride;
    int count = (slices - first + stride - 1) / stride;
    if (count > group) count = group;
    const size_t step = (size_t)tiles * (TILE * TILE / 4);
    f32x4* src = reinterpret_cast<f32x4*>(slab) + (size_t)first * step + (size_t)tile * (TILE * TILE / 4) +
                 (size_t)v * (WG * WG * 64) + tid;
    f32x4 s = src[0];
    int i = 1;
    for (; i + 3 < count; i += 4) {          // four loads in flight, added in slice order
        const f32x4 t0 = src[(size_t)i * stride * step], t1 = src[(size_t)(i + 1) * stride * step];
        const f32x4 t2 = src[(size_t)(i + 2) * stride * step], t3 = src[(size_t)(i + 3) * stride * step];
        s.x += t0.x; s.y += t0.y; s.z += t0.z; s.w += t0.w;
        s.x += t1.x; s.y += t1.y; s.z += t1.z; s.w += t1.w;
        s.x += t2.x; s.y += t2.y; s.z += t2.z; s.w += t2.w;
        s.x += t3.x; s.y += t3.y; s.z += t3.z; s.w += t3.w;
    }
    for (; i < count; i++) {
        const f32x4 t = src[(size_t)i * stride * step];
        s.x += t.x; s.y += t.y; s.z += t.z; s.w += t.w;
    }
    if constexpr (!TO_DW) {
        src[0] = s;
    } else {
        const int a = v / (WT * 4), c = (v / 4) % WT, g4 = v & 3;
        const int wave = tid >> 6, lane = tid & 63;
        const int wm = wave / WG, wn = wave % WG;
        const int tk = tile % tiles_k, tco = tile / tiles_k;
        const int kk = tk * TILE + (wn * WT + c) * 32 + (lane & 31);
        const int co = tco * TILE + (wm * WT + a) * 32 + 8 * g4 + 4 * (lane >> 5);
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

// slab workspace of the sliced weight gradient: part of the stream's conv workspace (caller-provided through
// brcnn_conv_set_workspace, conv_igemm_bf16.hip; launches on one stream are serialised)
constexpr size_t WGRAD_WS_BYTES = (size_t)160 << 20;
int g_wgrad_slabs = 1;       // tuning hook (brcnn_conv_set_tile_wgrad_bf16(10 / 11)): 0 atomics, 1 slabs + second stage
int g_wgrad_slot_pct = 75;   // ... (2000 + n): n percent of a generation of workgroups per launch (75: the launches share the device with the main stream; same-box A/B 19.92 -> 19.75 ms per step, 50 % level, 35 % +0.9 ms)
int g_wgrad_slot_pct_big = 75;   // ... (3000 + n): the same for launches of more than 2^17 reduction rows on the 256 x 256 tile
int g_wgrad_two_pass = 24;   // ... (100 + n): more than n slices per tile -> the second stage runs as two passes

void magic_for(unsigned d, unsigned* magic, unsigned* shift) {
    unsigned l = 0;
    while ((1ull << l) < d) l++;
    *magic = (unsigned)(((1ull << 32) * ((1ull << l) - d)) / d + 1);
    *shift = l;
}

template <int WT, int ET = 0, int WG = 2>
int launch(WgradHParams& p, hipStream_t s) {
    const int T = 32 * WT * WG;
    constexpr int CB = WG * WT / 2 > 0 ? WG * WT / 2 : 1;
    p.tiles_co = (p.Cout + T - 1) / T;
    p.tiles_k = (p.K + T - 1) / T;
    const int tiles = p.tiles_co * p.tiles_k;
    // M slices: as many as fit ONE generation of resident workgroups (a count just above a
    // multiple of the slot number costs a whole extra generation; every extra slice adds a full
    // tile of fp32 atomics into dW), each at least `minrows` reduction rows deep
    static int slots_env = getenv("BRCNN_WG_SLOTS") ? atoi(getenv("BRCNN_WG_SLOTS")) : 0;
    static int minrows = getenv("BRCNN_WG_MINROWS") ? atoi(getenv("BRCNN_WG_MINROWS")) : 1024;
    // (g_wgrad_slot_pct: the weight-gradient launches share the device with the main stream's kernels -- a fraction of
    // a generation leaves them CUs and shrinks the slab traffic)
    const int slots = slots_env ? slots_env : (WG == 4 ? 256 : WT == 2 ? 512 : 1024) *
                                                  ((WG == 4 && p.M >= (1 << 17)) ? g_wgrad_slot_pct_big : g_wgrad_slot_pct) / 100;
    int slices = slots / tiles;
    const int max_slices = (p.M + minrows - 1) / minrows;
    if (slices > max_slices) slices = max_slices;
    if (slices < 1) slices = 1;
    int rps = (p.M + slices - 1) / slices;
    rps = (rps + TM - 1) / TM * TM;
    p.slices = (p.M + rps - 1) / rps;
    p.rows_per_slice = rps;
    const size_t lds = (size_t)2 * 2 * CB * TM * 64 * sizeof(unsigned short);
    static bool attr_done = false;
    if (!attr_done) {
        BRCNN_HIP_CHECK(hipFuncSetAttribute((const void*)conv_wgrad_bf16_kernel<WT, ET, WG>,
                                            hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        attr_done = true;
    }
    p.slab = nullptr;
    bool deferred = false;
    if (g_wgrad_slabs && p.slices > 1 && (size_t)tiles * p.slices * T * T * sizeof(float) <= WGRAD_WS_BYTES) {
        // second stage postponed to the stream's batched reduction (wgrad_defer.hip)?
        int derr = 0;
        p.slab = brcnn_conv::wgrad_defer_slabs(s, (size_t)tiles * p.slices * T * T * sizeof(float), &derr);
        if (derr) return derr;
        deferred = p.slab != nullptr;
        if (!p.slab) p.slab = brcnn_conv::conv_ws_wgrad_slabs(s);
    }
    hipLaunchKernelGGL((conv_wgrad_bf16_kernel<WT, ET, WG>), dim3(tiles * p.slices), dim3(64 * WG * WG), lds, s, p);
    BRCNN_LAUNCH_CHECK();
    if (deferred) {
        int group = p.slices;
        if (p.slices > g_wgrad_two_pass) {
            group = 4;
            while (group * group < p.slices) group++;
        }
        brcnn_conv::wgrad_defer_push(s, p.slab, p.dw, tiles, p.tiles_k, p.slices, group, p.Cout, p.K, (WT << 4) | WG);
    } else if (p.slab) {
        int stride = 1, count = p.slices;
        if (p.slices > g_wgrad_two_pass) {   // few output tiles, many slices: groups of ~sqrt(slices) first (more workgroups, shorter chains)
            int group = 4;
            while (group * group < p.slices) group++;
            const int ngroups = (p.slices + group - 1) / group;
            hipLaunchKernelGGL((wgrad_reduce_kernel<WT, WG, false>), dim3(tiles * WT * WT * 4, ngroups), dim3(64 * WG * WG), 0, s,
                               p.slab, p.dw, p.tiles_k, tiles, p.slices, 1, group, p.Cout, p.K);
            BRCNN_LAUNCH_CHECK();
            stride = group;
            count = ngroups;
        }
        hipLaunchKernelGGL((wgrad_reduce_kernel<WT, WG, true>), dim3(tiles * WT * WT * 4, 1), dim3(64 * WG * WG), 0, s,
                           p.slab, p.dw, p.tiles_k, tiles, p.slices, stride, count, p.Cout, p.K);
        BRCNN_LAUNCH_CHECK();
    }
    return 0;
}

int g_wgrad_bf16_tile = 0;      // tuning hook: 0 heuristic, 1 = 64x64, 2 = 128x128, 4 = 256x256 (16 waves)

}  // namespace

// entry used by brcnn_conv2d_wgrad_nhwc_multi for dtype == BRCNN_DT_BF16
int brcnn_wgrad_bf16_dispatch(const void* x, const void* dy, void* dw, int batch, int num_segments,
                              const int* heights_host, const int* widths_host, int cin, int cout,
                              int kh, int kw, int stride, int pad, hipStream_t stream, int f16) {
    if ((cin & 7) || (cout & 7)) return BRCNN_EINVAL;
    if (g_wgrad_bf16_tile == 0) {       // the eight-phase 256 x 256 kernel where whole tiles and enough rows exist
        const int rc = brcnn_conv::wgrad_pp_bf16_try(x, dy, dw, batch, num_segments, heights_host, widths_host, cin, cout, kh, kw,
                                                     stride, pad, stream, f16);
        if (rc != 0) return rc < 0 ? rc : 0;
    }
    WgradHParams p = {};
    p.dy = (const unsigned short*)dy; p.x = (const unsigned short*)x; p.dw = (float*)dw;
    p.Cin = cin; p.Cout = cout; p.KH = kh; p.KW = kw; p.stride = stride; p.pad = pad;
    p.pitch = cin; p.gstep = 0;
    p.nseg = num_segments;
    long long m_total = 0, x_off = 0;
    for (int s = 0; s < num_segments; s++) {
        const int H = heights_host[s], W = widths_host[s];
        const int Ho = (H + 2 * pad - kh) / stride + 1, Wo = (W + 2 * pad - kw) / stride + 1;
        if (H <= 0 || W <= 0 || Ho <= 0 || Wo <= 0) return BRCNN_EINVAL;
        p.seg_H[s] = H; p.seg_W[s] = W; p.seg_Ho[s] = Ho; p.seg_Wo[s] = Wo;
        p.seg_m0[s] = (int)m_total;
        p.seg_xoff[s] = x_off;
        magic_for((unsigned)(Ho * Wo), &p.seg_mhw[s], &p.seg_shw[s]);
        magic_for((unsigned)Wo, &p.seg_mw[s], &p.seg_sw[s]);
        m_total += (long long)batch * Ho * Wo;
        x_off += (long long)batch * H * W * cin;
    }
    for (int s = num_segments; s <= BRCNN_MAX_LEVELS; s++) p.seg_m0[s] = (int)m_total;
    if (m_total * cout * 2 >= 0x7fffffffLL || x_off * 2 >= 0x7fffffffLL) return BRCNN_EINVAL;
    p.M = (int)m_total;
    p.K = kh * kw * cin;
    p.dy_bytes = (unsigned)(m_total * cout * 2);
    p.x_bytes = (unsigned)(x_off * 2);
    int wt = g_wgrad_bf16_tile;
    if (wt == 0) {
        wt = (cout >= 128 && p.K >= 128) ? 2 : 1;      // (K = 128, stage-2 conv3: 128 x 128 tile 42 us, 64 x 64 51 us)
        // 256 x 256 on 16 waves where the 128 x 128 tile is bound by its LDS-DMA traffic and one generation
        // of 256 workgroups still has enough reduction rows each: the five-level tower layer (537 -> 912
        // TFLOP/s), the first FC (384 -> 676).  Every workgroup ends with a full tile of fp32 atomics (their
        // volume is workgroups x tile area whatever the layer: 67 MB here, 33 MB for the 128 x 128 tile, at
        // 2-3 TB/s), which is what keeps the medium layers (M = 33 600) on the smaller tile.
        // (with the slab reduction -- r03 -- the 256 x 256 tile's doubled reduction volume costs plain stores instead of
        // atomics, and it wins on every layer with >= 256 output channels and K >= 2048: stage-3 3x3 68 -> 60 us,
        // stage-4 3x3 89 -> 66 us, 3x3 on the 100x168 map 204 -> 159 us; profiles/r03_notes.md)
        if (cout >= 256 && p.K >= 2048) wt = 4;
    }
    if (wt == 4) return f16 ? launch<2, 1, 4>(p, stream) : launch<2, 0, 4>(p, stream);
    if (f16) return wt == 2 ? launch<2, 1>(p, stream) : launch<1, 1>(p, stream);
    return wt == 2 ? launch<2>(p, stream) : launch<1>(p, stream);
}

namespace brcnn_conv {
int tuning_get_wgrad_slabs() { return g_wgrad_slabs; }
int tuning_get_wgrad_generation_percent() { return g_wgrad_slot_pct; }
}  // namespace brcnn_conv

BRCNN_API int brcnn_conv_set_tile_wgrad_bf16(int wt) {
    if (wt == 10 || wt == 11) { g_wgrad_slabs = wt - 10; return 0; }      // reduction over the M slices: atomics / slabs
    if (wt >= 100 && wt < 1100) { g_wgrad_two_pass = wt - 100; return 0; }
    if (wt >= 2010 && wt <= 2400) { g_wgrad_slot_pct = wt - 2000; return 0; }
    if (wt >= 3010 && wt <= 3400) { g_wgrad_slot_pct_big = wt - 3000; return 0; }
    // eight-phase kernel (conv_wgrad_pp_bf16.hip): 20 never / 21 heuristic / 22 wherever the shape allows; 4000 + n: n
    // percent of the CUs per launch; 5000 + n: two reduce passes above n slices; 29: RETURNS the number of launches the
    // eight-phase kernel took since the last query (tests); 30 / 31: its slab reduction as separate launches / inside the
    // producing launch
    if ((wt >= 20 && wt <= 22) || wt == 29 || wt == 30 || wt == 31 || (wt >= 4010 && wt <= 4400) || (wt >= 5001 && wt <= 5999)) return brcnn_conv::wgrad_pp_set(wt);
    if (wt < 0 || wt == 3 || wt > 4) return BRCNN_EINVAL;
    g_wgrad_bf16_tile = wt;
    return 0;
}


// grouped variant (ResNeXt conv2): per 64-channel co tile a dense (64 x KH*KW*window) wgrad over the
// tile's input window; called by brcnn_conv2d_wgrad_nhwc_grouped for dtype == BRCNN_DT_BF16
int brcnn_wgrad_bf16_grouped_dispatch(const void* x, const void* dy, void* dw_tiles, int batch, int height, int width,
                                      int cin, int cout, int kh, int kw, int stride, int pad, int window,
                                      hipStream_t stream, int f16) {
    if ((window & 7) || (cout % 64) || (cout / 64) * window != cin) return BRCNN_EINVAL;
    const int Ho = (height + 2 * pad - kh) / stride + 1, Wo = (width + 2 * pad - kw) / stride + 1;
    if (Ho <= 0 || Wo <= 0) return BRCNN_EINVAL;
    WgradHParams p = {};
    p.dy = (const unsigned short*)dy; p.x = (const unsigned short*)x; p.dw = (float*)dw_tiles;
    p.Cin = window; p.Cout = cout; p.KH = kh; p.KW = kw; p.stride = stride; p.pad = pad;
    p.pitch = cin; p.gstep = window; p.nseg = 1;
    p.seg_H[0] = height; p.seg_W[0] = width; p.seg_Ho[0] = Ho; p.seg_Wo[0] = Wo;
    p.seg_m0[0] = 0; p.seg_xoff[0] = 0;
    magic_for((unsigned)(Ho * Wo), &p.seg_mhw[0], &p.seg_shw[0]);
    magic_for((unsigned)Wo, &p.seg_mw[0], &p.seg_sw[0]);
    const long long m_total = (long long)batch * Ho * Wo, x_elems = (long long)batch * height * width * cin;
    for (int s = 1; s <= BRCNN_MAX_LEVELS; s++) p.seg_m0[s] = (int)m_total;
    if (m_total * cout * 2 >= 0x7fffffffLL || x_elems * 2 >= 0x7fffffffLL) return BRCNN_EINVAL;
    p.M = (int)m_total;
    p.K = kh * kw * window;
    p.dy_bytes = (unsigned)(m_total * cout * 2);
    p.x_bytes = (unsigned)(x_elems * 2);
    if (f16) return launch<1, 1>(p, stream);
    return launch<1>(p, stream);          // 64 x 64 output tiles: the co tile is the group window
}
