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
// every workgroup reduces its slice into a 64x64 fp32 tile on v_mfma_f32_32x32x2_f32 and adds
// it to dW with fp32 atomics (dW zero-filled by the caller).  Both operands are
// "reduction-major" in memory ((M,Cout) and (M,K) rows), which is exactly the MFMA operand
// order lane -> column: the LDS tiles are stored [m][64 columns] unpadded (conflict-free
// ds_read_b32 of 32 consecutive columns), one read per operand per MFMA.
#include <cstdlib>
#include "common.h"

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
constexpr int OOB = 0x7fffffff;
constexpr int TM = 32;          // reduction (m) rows per LDS tile
constexpr int TC = 64;          // columns per operand tile (output tile TC x TC)

struct WgradParams {
    const float* dy;    // (M, Cout)
    const float* x;     // (N, H, W, Cin) segments back to back
    float* dw;          // (Cout, K) fp32, accumulated atomically
    int Cin, Cout, KH, KW, stride, pad, M, K;
    int tiles_co, tiles_k, slices, rows_per_slice;
    unsigned dy_bytes, x_bytes;
    int pitch;      // floats between adjacent input pixels (== Cin unless grouped)
    int gstep;      // grouped conv: co tile t reads input channels [t*gstep, t*gstep + Cin); 0 = dense
    int direct;     // 1x1 / stride 1 / pad 0: input pixel index == output row index, no decode at all
    int nseg;
    int seg_m0[BRCNN_MAX_LEVELS + 1];
    int seg_H[BRCNN_MAX_LEVELS], seg_W[BRCNN_MAX_LEVELS], seg_Ho[BRCNN_MAX_LEVELS], seg_Wo[BRCNN_MAX_LEVELS];
    long long seg_xoff[BRCNN_MAX_LEVELS];
    // division by the invariant Ho*Wo / Wo as multiply-high + shift (the row -> pixel decode runs
    // for every staged row of every tile; two hardware-emulated integer divisions per row made
    // the kernel VALU bound)
    unsigned seg_mhw[BRCNN_MAX_LEVELS], seg_shw[BRCNN_MAX_LEVELS], seg_mw[BRCNN_MAX_LEVELS], seg_sw[BRCNN_MAX_LEVELS];
};

__device__ __forceinline__ unsigned fastdiv(unsigned x, unsigned magic, unsigned shift) {
    return (unsigned)(((unsigned long long)__umulhi(x, magic) + x) >> shift);
}

// 64 x 64 output tile, 4 waves x one 32x32 MFMA tile; both operand tiles ([32 m][64 cols] fp32,
// 256-byte rows) are staged by LDS-DMA (`buffer_load_dwordx4 ... lds`: no VGPR round trip, no
// ds_write pass), double buffered, one `vmcnt(0)` + barrier per 32 reduction rows.
template <bool DIRECT>
__global__ __launch_bounds__(256, 2) void conv_wgrad_f32_kernel(WgradParams p) {
    __shared__ __attribute__((aligned(16))) float Ya[2][TM][TC];   // dY tile  [m][co]
    __shared__ __attribute__((aligned(16))) float Xa[2][TM][TC];   // im2col tile [m][k]
    typedef __attribute__((address_space(3))) void* lds_ptr_t;

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1;
    const int li = lane & 31, lh = lane >> 5;

    // Block order: all (co, k) tiles of one M slice are adjacent and each XCD owns a contiguous
    // run of blocks, so the dY rows (shared by the k tiles) and the x pixels (shared by the co
    // tiles and by neighbouring filter taps) of a slice are fetched into one L2 once.
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
    const int co0 = tco * TC, k0 = tk * TC;
    const int m_begin = slice * p.rows_per_slice;
    const int m_end = min(p.M, m_begin + p.rows_per_slice);
    if (m_begin >= m_end) return;

    const __amdgpu_buffer_rsrc_t rsrc_y = __builtin_amdgcn_make_buffer_rsrc((void*)p.dy, 0, (int)p.dy_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rsrc_x = __builtin_amdgcn_make_buffer_rsrc((void*)p.x, 0, (int)p.x_bytes, 0x00020000);

    // DMA assignment: a wave instruction moves 4 rows x 256 B; wave w owns rows (2w+j)*4 .. +4
    const int c4 = lane & 15, rr = lane >> 4;
    const int co = co0 + c4 * 4;
    const bool co_ok = co < p.Cout;                 // Cout % 4 == 0 enforced by the host
    const int k = k0 + c4 * 4;
    const bool k_ok = k < p.K;                      // K % 4 == 0 (Cin % 4 == 0)
    const int tap = k_ok ? k / p.Cin : 0;
    const int ci = k - tap * p.Cin;
    const int kh = tap / p.KW, kw = tap - kh * p.KW;

    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; r++) acc[r] = 0.f;

    // Row -> input pixel decode.  A lane stages the same two tile rows of every tile, i.e. output
    // pixels m, m+32, m+64, ...: the decode (two divisions, a segment search) runs once per lane
    // and map, afterwards (ho, wo) and the offset of the input row advance incrementally -- one
    // conditional wrap per step when the map is at least 32 wide; narrower maps and the first row
    // after a map boundary take the full decode again.
    int s_ho[2], s_wo[2], s_rb[2], s_H[2], s_W[2], s_Ho[2], s_Wo[2], s_end[2] = {0, 0};
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
        s_rb[j] = (int)p.seg_xoff[sg] + (n * H + ho * p.stride) * W * p.pitch;
        s_end[j] = p.seg_m0[sg + 1];
    };
    auto dma_tile = [&](int mt, int buf) {
#pragma unroll
        for (int j = 0; j < 2; j++) {
            const int row = (wave * 2 + j) * 4 + rr;
            const int m = mt + row;
            int offy = OOB, offx = OOB;
            if (m < m_end) {
                if (co_ok) offy = (m * p.Cout + co) * 4;
                if (k_ok && DIRECT) {
                    offx = (m * p.pitch + ci + tco * p.gstep) * 4;
                } else if (k_ok) {
                    if (m >= s_end[j] || s_Wo[j] < TM) {
                        decode(m, j);
                    } else {                    // same map, 32 output pixels further
                        int wo = s_wo[j] + TM;
                        if (wo >= s_Wo[j]) {
                            wo -= s_Wo[j];
                            int ho = s_ho[j] + 1;
                            int rb = s_rb[j] + p.stride * s_W[j] * p.pitch;
                            if (ho == s_Ho[j]) {    // next image of the map
                                ho = 0;
                                rb += (s_H[j] - s_Ho[j] * p.stride) * s_W[j] * p.pitch;
                            }
                            s_ho[j] = ho;
                            s_rb[j] = rb;
                        }
                        s_wo[j] = wo;
                    }
                    const int hi = s_ho[j] * p.stride - p.pad + kh, wi = s_wo[j] * p.stride - p.pad + kw;
                    if ((unsigned)hi < (unsigned)s_H[j] && (unsigned)wi < (unsigned)s_W[j])
                        offx = (s_rb[j] + ((kh - p.pad) * s_W[j] + wi) * p.pitch + ci + tco * p.gstep) * 4;
                }
            }
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc_y, (lds_ptr_t)&Ya[buf][(wave * 2 + j) * 4][0], 16, offy, 0, 0, 0);
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc_x, (lds_ptr_t)&Xa[buf][(wave * 2 + j) * 4][0], 16, offx, 0, 0, 0);
        }
    };

    dma_tile(m_begin, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    // Operand reads are inline asm (explicit lgkmcnt waits): a compiler-visible LDS read after a
    // `buffer_load ... lds` gets `s_waitcnt vmcnt(0)` put in front of it, which would serialise
    // the prefetch of the next 32 rows with this tile's MFMAs in every wave.
    // one MFMA step consumes reduction rows (2s, 2s+1): lane half lh takes row 2s + lh
    const unsigned y_lane = (unsigned)(size_t)(lds_ptr_t)&Ya[0][lh][wm * 32 + li];
    const unsigned x_lane = (unsigned)(size_t)(lds_ptr_t)&Xa[0][lh][wn * 32 + li];
    static_assert(TM == 32 && TC == 64, "read offsets below are written for 32 x 64 tiles");
    float ya[2][8], xa[2][8];
#define BRCNN_RD(H, S) asm volatile("ds_read_b32 %0, %2 offset:" #S "\n\tds_read_b32 %1, %3 offset:" #S \
                                    : "=v"(ya[H][(S / 512) & 7]), "=v"(xa[H][(S / 512) & 7]) : "v"(ya_addr), "v"(xa_addr) : "memory")
#define BRCNN_WAIT(H) asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(ya[H][0]), "+v"(ya[H][1]), "+v"(ya[H][2]), "+v"(ya[H][3]), \
                                   "+v"(ya[H][4]), "+v"(ya[H][5]), "+v"(ya[H][6]), "+v"(ya[H][7]), "+v"(xa[H][0]), "+v"(xa[H][1]), \
                                   "+v"(xa[H][2]), "+v"(xa[H][3]), "+v"(xa[H][4]), "+v"(xa[H][5]), "+v"(xa[H][6]), "+v"(xa[H][7]) :: "memory")
    int cur = 0;
    for (int mt = m_begin; mt < m_end; mt += TM) {
        if (mt + TM < m_end) dma_tile(mt + TM, cur ^ 1);
        const unsigned ya_addr = y_lane + cur * (TM * TC * 4), xa_addr = x_lane + cur * (TM * TC * 4);
        BRCNN_RD(0, 0); BRCNN_RD(0, 512); BRCNN_RD(0, 1024); BRCNN_RD(0, 1536);
        BRCNN_RD(0, 2048); BRCNN_RD(0, 2560); BRCNN_RD(0, 3072); BRCNN_RD(0, 3584);
        BRCNN_WAIT(0);
        BRCNN_RD(1, 4096); BRCNN_RD(1, 4608); BRCNN_RD(1, 5120); BRCNN_RD(1, 5632);
        BRCNN_RD(1, 6144); BRCNN_RD(1, 6656); BRCNN_RD(1, 7168); BRCNN_RD(1, 7680);
#pragma unroll
        for (int s = 0; s < 8; s++) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(ya[0][s], xa[0][s], acc, 0, 0, 0);
        BRCNN_WAIT(1);
#pragma unroll
        for (int s = 0; s < 8; s++) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(ya[1][s], xa[1][s], acc, 0, 0, 0);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        cur ^= 1;
    }
#undef BRCNN_RD
#undef BRCNN_WAIT

    // D[row = co][col = k]:  col = lane&31, row = (r&3) + 8*(r>>2) + 4*(lane>>5)
    const int kk = k0 + wn * 32 + li;
    if (kk < p.K) {
#pragma unroll
        for (int r = 0; r < 16; r++) {
            const int c = co0 + wm * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
            if (c < p.Cout) atomicAdd(p.dw + (size_t)c * p.K + kk, acc[r]);
        }
    }
}

// M slices: about two generations of resident workgroups (measured, tools/_sweep.sh: 2560 slots,
// >= 256 rows per slice; every extra slice adds a full tile of fp32 atomics into dW, too few leave
// CUs idle).  BRCNN_WG_SLOTS / BRCNN_WG_MINROWS override for tuning runs.
int wgrad_slices(int tiles, int M) {
    static int slots = getenv("BRCNN_WG_SLOTS") ? atoi(getenv("BRCNN_WG_SLOTS")) : 2560;
    static int minrows = getenv("BRCNN_WG_MINROWS") ? atoi(getenv("BRCNN_WG_MINROWS")) : 256;
    int slices = slots / tiles;
    const int max_slices = (M + minrows - 1) / minrows;
    if (slices > max_slices) slices = max_slices;
    return slices < 1 ? 1 : slices;
}

// q = (mulhi(x, magic) + x) >> shift == x / d for every x < 2^32 (Granlund-Montgomery, round-up
// variant with the implicit 2^32 term)
void magic_for(unsigned d, unsigned* magic, unsigned* shift) {
    unsigned l = 0;
    while ((1ull << l) < d) l++;
    *magic = (unsigned)(((1ull << 32) * ((1ull << l) - d)) / d + 1);
    *shift = l;
}

}  // namespace

int brcnn_wgrad_bf16_grouped_dispatch(const void* x, const void* dy, void* dw_tiles, int batch, int height, int width,
                                      int cin, int cout, int kh, int kw, int stride, int pad, int window,
                                      hipStream_t stream, int f16);    // conv_wgrad_bf16.hip
int brcnn_wgrad_bf16_dispatch(const void* x, const void* dy, void* dw, int batch, int num_segments,
                              const int* heights_host, const int* widths_host, int cin, int cout,
                              int kh, int kw, int stride, int pad, hipStream_t stream, int f16);   // conv_wgrad_bf16.hip

BRCNN_API int brcnn_conv2d_wgrad_nhwc_multi(const void* x, const void* dy, void* dw, int batch,
                                            int num_segments, const int* heights_host,
                                            const int* widths_host, int cin, int cout, int kh,
                                            int kw, int stride, int pad, int dtype, void* stream) {
    if (!x || !dy || !dw || batch <= 0 || cin <= 0 || cout <= 0 || kh <= 0 || kw <= 0 ||
        stride <= 0 || pad < 0 || num_segments <= 0 || num_segments > BRCNN_MAX_LEVELS ||
        !heights_host || !widths_host || !brcnn_elem_ok(dtype) || (cin & 3) ||
        (cout & 3))
        return BRCNN_EINVAL;
    if (dtype != BRCNN_DT_F32)
        return brcnn_wgrad_bf16_dispatch(x, dy, dw, batch, num_segments, heights_host, widths_host, cin, cout,
                                         kh, kw, stride, pad, (hipStream_t)stream, dtype == BRCNN_DT_F16);
    WgradParams p = {};
    p.dy = (const float*)dy; p.x = (const float*)x; p.dw = (float*)dw;
    p.Cin = cin; p.Cout = cout; p.KH = kh; p.KW = kw; p.stride = stride; p.pad = pad;
    p.pitch = cin; p.gstep = 0;
    p.nseg = num_segments;
    p.direct = (kh == 1 && kw == 1 && stride == 1 && pad == 0);
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
    if (m_total * cout * 4 >= 0x7fffffffLL || x_off * 4 >= 0x7fffffffLL) return BRCNN_EINVAL;
    p.M = (int)m_total;
    p.K = kh * kw * cin;
    p.dy_bytes = (unsigned)(m_total * cout * 4);
    p.x_bytes = (unsigned)(x_off * 4);
    p.tiles_co = (cout + TC - 1) / TC;
    p.tiles_k = (p.K + TC - 1) / TC;
    const int tiles = p.tiles_co * p.tiles_k;
    int slices = wgrad_slices(tiles, p.M);
    const int max_slices = p.M;
    if (slices > max_slices) slices = max_slices;
    if (slices < 1) slices = 1;
    int rps = (p.M + slices - 1) / slices;
    rps = (rps + TM - 1) / TM * TM;
    p.slices = (p.M + rps - 1) / rps;
    p.rows_per_slice = rps;
    if (p.direct) hipLaunchKernelGGL(conv_wgrad_f32_kernel<true>, dim3(tiles * p.slices), dim3(256), 0, (hipStream_t)stream, p);
    else hipLaunchKernelGGL(conv_wgrad_f32_kernel<false>, dim3(tiles * p.slices), dim3(256), 0, (hipStream_t)stream, p);
    BRCNN_LAUNCH_CHECK();
    return 0;
}


// weight gradient of the grouped conv: per 64-channel co tile a dense (64 x KH*KW*window) wgrad over
// the tile's input window; the caller keeps the block-diagonal entries (brcnn.autograd)
BRCNN_API int brcnn_conv2d_wgrad_nhwc_grouped(const void* x, const void* dy, void* dw_tiles, int batch,
                                              int height, int width, int cin, int cout, int kh, int kw,
                                              int stride, int pad, int window, int dtype, void* stream) {
    if (!x || !dy || !dw_tiles || batch <= 0 || cin <= 0 || cout <= 0 || kh <= 0 || kw <= 0 || stride <= 0 ||
        pad < 0 || !brcnn_elem_ok(dtype) || window <= 0 || (window & 3) || (cout % 64) ||
        (cout / 64) * window != cin)
        return BRCNN_EINVAL;
    if (dtype != BRCNN_DT_F32)
        return brcnn_wgrad_bf16_grouped_dispatch(x, dy, dw_tiles, batch, height, width, cin, cout, kh, kw, stride, pad,
                                                 window, (hipStream_t)stream, dtype == BRCNN_DT_F16);
    const int Ho = (height + 2 * pad - kh) / stride + 1, Wo = (width + 2 * pad - kw) / stride + 1;
    if (Ho <= 0 || Wo <= 0) return BRCNN_EINVAL;
    WgradParams p = {};
    p.dy = (const float*)dy; p.x = (const float*)x; p.dw = (float*)dw_tiles;
    p.Cin = window; p.Cout = cout; p.KH = kh; p.KW = kw; p.stride = stride; p.pad = pad;
    p.pitch = cin; p.gstep = window; p.nseg = 1;
    p.direct = (kh == 1 && kw == 1 && stride == 1 && pad == 0);
    p.seg_H[0] = height; p.seg_W[0] = width; p.seg_Ho[0] = Ho; p.seg_Wo[0] = Wo;
    p.seg_m0[0] = 0; p.seg_xoff[0] = 0;
    magic_for((unsigned)(Ho * Wo), &p.seg_mhw[0], &p.seg_shw[0]);
    magic_for((unsigned)Wo, &p.seg_mw[0], &p.seg_sw[0]);
    const long long m_total = (long long)batch * Ho * Wo, x_elems = (long long)batch * height * width * cin;
    for (int s = 1; s <= BRCNN_MAX_LEVELS; s++) p.seg_m0[s] = (int)m_total;
    if (m_total * cout * 4 >= 0x7fffffffLL || x_elems * 4 >= 0x7fffffffLL) return BRCNN_EINVAL;
    p.M = (int)m_total;
    p.K = kh * kw * window;
    p.dy_bytes = (unsigned)(m_total * cout * 4);
    p.x_bytes = (unsigned)(x_elems * 4);
    p.tiles_co = cout / TC;
    p.tiles_k = (p.K + TC - 1) / TC;
    const int tiles = p.tiles_co * p.tiles_k;
    int slices = wgrad_slices(tiles, p.M);
    const int max_slices = p.M;
    if (slices > max_slices) slices = max_slices;
    if (slices < 1) slices = 1;
    int rps = (p.M + slices - 1) / slices;
    rps = (rps + TM - 1) / TM * TM;
    p.slices = (p.M + rps - 1) / rps;
    p.rows_per_slice = rps;
    if (p.direct) hipLaunchKernelGGL(conv_wgrad_f32_kernel<true>, dim3(tiles * p.slices), dim3(256), 0, (hipStream_t)stream, p);
    else hipLaunchKernelGGL(conv_wgrad_f32_kernel<false>, dim3(tiles * p.slices), dim3(256), 0, (hipStream_t)stream, p);
    BRCNN_LAUNCH_CHECK();
    return 0;
}
