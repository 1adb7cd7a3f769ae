// Shared declarations of the implicit-GEMM convolution kernels (fp32: conv_igemm.hip,
// bf16: conv_igemm_bf16.hip).
#pragma once
#include "common.h"

namespace brcnn_conv {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

// any byte offset >= the buffer extent makes a raw buffer load return zeros: the zero padding
// of the convolution and the M / Cout tails cost no branch and no select
constexpr int OOB = 0x7fffffff;

constexpr int BK = 32, LDS_STRIDE = 36;

struct ConvParams {
    const float* x;
    const float* w;
    const float* scale;
    const float* shift;
    const float* residual;
    float* y;
    int batch, Cin, Cout, KH, KW, stride, pad;
    int M, K;
    int relu;
    int tiles_m, tiles_n;
    int pitch;                       // floats between adjacent input pixels (== Cin normally)
    int stem2;                       // 16-bit stem: a K tile = 8 pixels (32 elements) of filter row 2 kh and 8 of row 2 kh + 1
    unsigned x_bytes, w_bytes;       // extents for the bounds-checked buffer loads
    int dilate;                      // input dilation (data gradient of a strided conv), 1 otherwise
    int out_f32;                     // bf16 compute path: write the result as fp32 (head outputs)
    int f16;                         // 16-bit path: operands are IEEE fp16 instead of bf16
    int il;                          // tuning: 1 = LDS-DMA pieces interleaved with the MFMA groups (bf16 kernel)
    // output scatter (data gradient of a stride-2 conv by parity class): output pixel (a, b) of the
    // launch goes to pixel (2*(a-sc_o)+sc_ph, 2*(b-sc_o)+sc_pw) of a (batch, sc_H, sc_W, Cout) tensor and
    // is dropped when a-sc_o / b-sc_o fall outside [0, sc_na) / [0, sc_nb); 0 = dense output
    int scatter, sc_H, sc_W, sc_ph, sc_pw, sc_o, sc_na, sc_nb;
    int gstep;                       // grouped conv: N tile t reads input channels [t*gstep, t*gstep + Cin); 0 = dense
    // training-mode dual store (16-bit kernel): the raw conv output goes to z_out (what the BatchNorm
    // backward needs), y = [relu](z * scale + shift [+ residual]); with bn_mean / bn_var set, `scale` /
    // `shift` are gamma / beta of an eval-mode BatchNorm and the affine is formed in the epilogue
    void* z_out;
    // data-gradient launch that also runs the BatchNorm(+ReLU) backward of the layer that PRODUCED this conv's
    // input (16-bit kernel): tail_z = that layer's raw conv output (same shape as this launch's output),
    // scale / shift / bn_* its BatchNorm; the epilogue masks the data gradient with z * sc + sh > 0
    // (tail_relu), stores dz = d * sc as the output and writes per-row-tile sums of d and d * z to
    // tail_partials (tiles_m, 2, Cout) for dgamma / dbeta
    const void* tail_z;
    float* tail_partials;
    int tail_relu;
    // ... of a RESIDUAL producer (bn3 of the previous Bottleneck): `residual` is the identity branch's gradient
    // (added first, as in the plain data-gradient launch), tail_mask the producer's output (ReLU mask: > 0),
    // tail_dres receives the masked gradient itself (the identity gradient of the previous block)
    const void* tail_mask;
    void* tail_dres;
    const float* bn_mean;
    const float* bn_var;
    float bn_eps;
    // chained stream-K schedule (16-bit kernel): the launch's workgroups take their work item from sk_items[blockIdx.x]
    // = (tile, first K tile, end K tile, hand-over slot); tile < 0: padding.  The tiles x K-tiles iteration space is
    // cut into equal ranges, one per resident workgroup slot; a tile that straddles two ranges becomes two items: its K
    // head (dispatched in the launch's FIRST round; it stores the fp32 accumulators to sk_ws[slot] and sets
    // sk_flags[slot] = sk_epoch) and its K tail (dispatched in the LAST round; it starts from those accumulators, so the
    // MFMA chain over K is the unsplit one, bit for bit, and runs the epilogue).  sk_wgs = 0: one whole tile per block.
    int pp_rows;                     // eight-phase fp32 kernel: tuning hook, 0 heuristic / 128 / 256 row tiles
    int pp_cols;                     // ... 0 heuristic / 128: the 256 x 128 tile (Cout % 128 == 0)
    int sk_wgs;
    const int4* sk_items;
    float* sk_ws;
    unsigned* sk_flags;
    unsigned sk_epoch;
    // a K tail polls its head's flag at most sk_spin_limit times (~0.3 us each); on a time-out it writes the host-mapped
    // word sk_err (launch epoch | 1 << 31) before it falls through, and the host turns that into BRCNN_EHANDOVER
    unsigned* sk_err;
    int sk_spin_limit;
    int sk_drop_publish;             // test hook (brcnn_conv_set_tile_bf16(-11)): heads do not publish
    int st_strips;                   // conv1x1_stream_bf16.hip: row strips per column block
    int no_fast;                     // test hook (brcnn_conv_set_tile(-4, 1)): the general set-up and read-out everywhere
    // segment s covers output rows [seg_m0[s], seg_m0[s+1]) with its own geometry / input offset
    int nseg;
    int seg_m0[BRCNN_MAX_LEVELS + 1];
    int seg_H[BRCNN_MAX_LEVELS], seg_W[BRCNN_MAX_LEVELS], seg_Ho[BRCNN_MAX_LEVELS], seg_Wo[BRCNN_MAX_LEVELS];
    long long seg_xoff[BRCNN_MAX_LEVELS];   // element offset of the segment's input in x
};

__device__ __forceinline__ static int xcd_remap(int bid, int nwg) {
    // bijective: XCD x (= bid % 8) owns a contiguous chunk of logical tile ids
    const int q = nwg >> 3, r = nwg & 7;
    const int xcd = bid & 7, loc = bid >> 3;
    const int base = (xcd < r) ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
    return base + loc;
}


// element offset of output row m (-1: row dropped); dense launches: m * Cout
__device__ __forceinline__ static long long out_row_offset(const ConvParams& p, int m) {
    if (!p.scatter) return (long long)m * p.Cout;
    const int Ho = p.seg_Ho[0], Wo = p.seg_Wo[0];
    const int n = m / (Ho * Wo);
    const int r = m - n * (Ho * Wo);
    const int a = r / Wo - p.sc_o, b = r - (r / Wo) * Wo - p.sc_o;
    if ((unsigned)a >= (unsigned)p.sc_na || (unsigned)b >= (unsigned)p.sc_nb) return -1;
    return (((long long)n * p.sc_H + 2 * a + p.sc_ph) * p.sc_W + 2 * b + p.sc_pw) * p.Cout;
}

// zero-stuffed input (the data gradient of a strided conv reads dy with `dilate` - 1 zeros between its pixels): coordinate
// (hi, wi) of the stuffed map -> the dy pixel, `ok` cleared where a stuffed zero is addressed.  The model only strides by
// 2: a shift and a mask per piece and K tile; other strides pay the two integer divisions (~30 VALU instructions each --
// the stride-2 3x3 data gradients spent 44 % of their SIMD cycles on VALU with them).
__device__ __forceinline__ void brcnn_undilate(int dilate, int& hi, int& wi, bool& ok) {
    ok = ok & (hi >= 0) & (wi >= 0);
    if (dilate == 2) {
        ok = ok & (((hi | wi) & 1) == 0);
        hi >>= 1;
        wi >>= 1;
    } else {
        const int qh = hi / dilate, qw = wi / dilate;
        ok = ok & (qh * dilate == hi) & (qw * dilate == wi);
        hi = qh;
        wi = qw;
    }
}

// ---- straight-line read-out of a full output tile of the 16-bit kernels (conv_igemm_bf16.hip, conv_pp_bf16.hip) ----------
// The short-K layers of the backbone are bound by VALU instructions, not by bytes or MFMA (PMC: 83 % of the SIMD cycles of
// the 64 -> 64 layer were VALU), and most of those were the guards of the general read-out: a branch per scale / shift
// element, bounds and scatter tests per row, 64-bit multiplies per address.  A tile that lies inside the output
// (wave-uniform test) takes this form instead: vector loads of scale / shift, pointers advanced by a constant, packed fp32
// multiplies / adds, the hardware conversions, ReLU as a packed signed-16-bit max on the converted pair (bf16 / fp16 are
// sign-magnitude: max_i16 with 0 clears exactly the negative values; with -32768 it is the identity).  Same arithmetic per
// element as the general form: both give the same bits.
typedef short brcnn_i16x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ float brcnn_relu1(float v) { return __builtin_amdgcn_fmed3f(v, 0.f, __builtin_inff()); }
template <int ET> __device__ __forceinline__ brcnn_f32x2 brcnn_unpk2(unsigned w) {
    brcnn_f32x2 r;
    if constexpr (ET) {
        r.x = brcnn_h2f((unsigned short)(w & 0xffffu));
        r.y = brcnn_h2f((unsigned short)(w >> 16));
    } else {
        r.x = __uint_as_float(w << 16);
        r.y = __uint_as_float(w & 0xffff0000u);
    }
    return r;
}
template <int ET> __device__ __forceinline__ unsigned brcnn_pk2(brcnn_f32x2 v) { return ET ? brcnn_pk2h(v.x, v.y) : brcnn_pk2b(v.x, v.y); }
__device__ __forceinline__ unsigned brcnn_relu_pk(unsigned w, unsigned floor2) {
    const brcnn_i16x2 a = __builtin_bit_cast(brcnn_i16x2, w), f = __builtin_bit_cast(brcnn_i16x2, floor2);
    return __builtin_bit_cast(unsigned, __builtin_elementwise_max(a, f));
}

// persistent streaming kernel of the short-K plain 1x1 layers (conv1x1_stream_bf16.hip): 1 launched, 0 not taken, < 0 error
int conv1x1_stream_try(ConvParams& p, hipStream_t s, int f16);
int conv1x1_stream_set(int mode);
// bf16 dispatch (conv_igemm_bf16.hip)
int dispatch_conv_bf16(ConvParams& p, hipStream_t s);
// 256 x 256 tile on the eight-phase two-group schedule (conv_pp_bf16.hip)
int dispatch_conv_pp_bf16(ConvParams& p, hipStream_t s);
// 256 x 128 tile on the same two-group schedule, three slots per K tile (conv_pp128_bf16.hip)
int dispatch_conv_pp128_bf16(ConvParams& p, hipStream_t s);
// chained stream-K plan for the eight-phase kernel (conv_igemm_bf16.hip owns the per-stream slots and item tables):
// fills p.sk_*; p.sk_wgs = 0 when the plain one-tile-per-workgroup launch is the better one
int sk_plan_pp(ConvParams& p, int slots, int bm, int bn, hipStream_t s);
int sk_plan_pp_f32(ConvParams& p, int slots, int bm, int bn, hipStream_t s);
float* conv_ws_wgrad_slabs(hipStream_t s);          // the weight-gradient slab part of the stream's conv workspace (160 MiB)
// deferred second stage of the sliced weight gradients (wgrad_defer.hip): slabs from the stream's deferral arena (nullptr:
// deferral off / request larger than the arena; *err < 0: the flush it had to launch first failed), and the item of a
// producing launch just issued.  kind: 0 = eight-phase 256 x 256 register order, else (WT << 4) | WG of conv_wgrad_bf16_kernel
float* wgrad_defer_slabs(hipStream_t s, size_t bytes, int* err);
void wgrad_defer_push(hipStream_t s, float* slab, float* dw, int tiles, int tiles_k, int slices, int group, int Cout, int K,
                      int kind);
unsigned* conv_ws_wgrad_counters(hipStream_t s);   // 8192 zero-initialised, self-resetting arrival counters
// eight-phase 256 x 256 weight gradient (conv_wgrad_pp_bf16.hip): 1 launched, 0 shape not taken, < 0 error
int wgrad_pp_bf16_try(const void* x, const void* dy, void* dw, int batch, int num_segments, const int* heights_host,
                      const int* widths_host, int cin, int cout, int kh, int kw, int stride, int pad, hipStream_t stream,
                      int f16);
int wgrad_pp_set(int v);
bool sk_par_enabled();       // split-K of few-tile launches is on (conv_igemm_bf16.hip, sk_table_par)
// the same schedule on the exact-fp32 MFMA (conv_pp_f32.hip); plain epilogue
int dispatch_conv_pp_f32(ConvParams& p, hipStream_t s);

}  // namespace brcnn_conv
