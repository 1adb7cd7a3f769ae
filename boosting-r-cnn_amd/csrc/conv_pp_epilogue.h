// Read-out of the eight-phase 16-bit convolution kernels (conv_pp_bf16.hip: 256 x 256 tile; conv_pp128_bf16.hip: 256 x 128
// tile): a wave holds MT x NT MFMA tiles of 32 x 32 fp32 accumulators (D^T = W A^T: lane l holds pixel m = l & 31 and, per
// register group g, four consecutive channels co = 8 g + 4 (l >> 5) + (0..3)); 8 waves = WG row groups x 8 / WG column
// strips.  The arithmetic and its order are conv_igemm_bf16.hip's, so every tile shape gives the same bits.
#pragma once
#include "conv_common.h"

namespace brcnn_conv {

template <int ET> __device__ __forceinline__ float pp_e2f(unsigned short h) { return ET ? brcnn_h2f(h) : brcnn_b2f(h); }
template <int ET> __device__ __forceinline__ unsigned short pp_f2e(float v) { return ET ? brcnn_f2h(v) : brcnn_f2b(v); }
template <int ET> __device__ __forceinline__ unsigned pp_pk2e(float lo, float hi) { return ET ? brcnn_pk2h(lo, hi) : brcnn_pk2b(lo, hi); }

// smem: the workgroup's LDS (>= 8 waves x 32 x (32 NT + 4) floats, free of operand traffic: the caller has waited and
// passed a barrier); wave = wm * (8 / WG) + wn; (m0, n0) the tile's first row / column, tile_m its row-tile index.
template <bool RES, bool OUTF32, int ET, int MODE, int MT, int NT, int WG>
__device__ __forceinline__ void pp_epilogue(const ConvParams& p, float* smem, f32x16 (&acc)[MT][NT], int tid, int wave, int wm,
                                            int wn, int m0, int n0, int tile_m) {
    constexpr int WNW = 8 / WG;
    constexpr int BM = WG * 32 * MT, BN = WNW * 32 * NT;
    const int lane = tid & 63;
    const int li = lane & 31, lh = lane >> 5;
    // ---- epilogue (conv_igemm_bf16.hip's, the same arithmetic in the same order): lane l holds pixel m = l&31 and, per
    // register group g, four consecutive channels co = 8g + 4(l>>5) + (0..3); one 32-row slab at a time goes through the
    // wave's private LDS region so that a lane ends up with 8 consecutive channels of one pixel (16-byte accesses).
    // MODE 0: scale / shift in the accumulators, residual, ReLU.
    // MODE 1 (training forward, conv -> eval-BN [-> + residual] [-> ReLU]): the raw tile leaves as z_out, the affine
    //         (one channel per lane derived from gamma / beta / mean / var, handed out through the slab) is applied to
    //         the STORED z in the read-out layout.
    // MODE 2 (data gradient + the BatchNorm backward of the input's producer): tail_z is the producer's raw output;
    //         d = gradient masked by the producer's ReLU, out = d * scale, per-row-tile sums of d and d * z.
    constexpr int PITCH = 32 * NT + 4;            // floats
    float* cs = smem + wave * 32 * PITCH;
    constexpr int LPR = 4 * NT, RPI = 64 / LPR;
    const int cw0 = n0 + wn * 32 * NT;
    const bool vec_ok = (p.Cout & 7) == 0;
    const int rl = lane / LPR, cl = (lane % LPR) * 8;
    const unsigned short* __restrict__ res = reinterpret_cast<const unsigned short*>(MODE == 2 ? p.tail_z : (const void*)p.residual);
    constexpr bool HAS_RQ = RES || MODE == 2;
    float sc8[8], sh8[8];
    float sum_dz[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f}, sum_d[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    if constexpr (MODE != 0) {
        {
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
        if (!p.no_fast && vec_ok && !p.scatter && m0 + BM <= p.M && n0 + BN <= p.Cout) {
            const size_t row0 = (size_t)(m0 + wm * 32 * MT + rl) * p.Cout + cw0 + cl;
            unsigned short* __restrict__ yrow = reinterpret_cast<unsigned short*>(p.y) + row0;
            unsigned short* __restrict__ zrow = reinterpret_cast<unsigned short*>(p.z_out) + row0;
            const unsigned short* __restrict__ rrow = res + row0;
            const unsigned floor2 = p.relu ? 0u : 0x80008000u;
            brcnn_f32x2 scp[NT][4][2], shp[NT][4][2];
            if constexpr (MODE == 0) {
#pragma unroll
                for (int tn = 0; tn < NT; tn++)
#pragma unroll
                    for (int g = 0; g < 4; g++) {
                        const int co = cw0 + tn * 32 + 8 * g + 4 * lh;
                        const float4 a = p.scale ? *reinterpret_cast<const float4*>(p.scale + co) : make_float4(1.f, 1.f, 1.f, 1.f);
                        const float4 b = p.shift ? *reinterpret_cast<const float4*>(p.shift + co) : make_float4(0.f, 0.f, 0.f, 0.f);
                        scp[tn][g][0] = brcnn_f32x2{a.x, a.y}; scp[tn][g][1] = brcnn_f32x2{a.z, a.w};
                        shp[tn][g][0] = brcnn_f32x2{b.x, b.y}; shp[tn][g][1] = brcnn_f32x2{b.z, b.w};
                    }
            }
            brcnn_f32x2 sc8p[4], sh8p[4];
#pragma unroll
            for (int e = 0; e < 4; e++) { sc8p[e] = brcnn_f32x2{sc8[2 * e], sc8[2 * e + 1]}; sh8p[e] = brcnn_f32x2{sh8[2 * e], sh8[2 * e + 1]}; }
#pragma unroll
            for (int tm = 0; tm < MT; tm++) {
                uint4 rq[32 / RPI];
                if (RES) {
#pragma unroll
                    for (int it = 0; it < 32 / RPI; it++)
                        rq[it] = *reinterpret_cast<const uint4*>(rrow + (size_t)(tm * 32 + it * RPI) * p.Cout);
                }
#pragma unroll
                for (int tn = 0; tn < NT; tn++)
#pragma unroll
                    for (int g = 0; g < 4; g++) {
                        brcnn_f32x2 lo = {acc[tm][tn][4 * g + 0], acc[tm][tn][4 * g + 1]};
                        brcnn_f32x2 hi = {acc[tm][tn][4 * g + 2], acc[tm][tn][4 * g + 3]};
                        if constexpr (MODE == 0) {
                            lo = lo * scp[tn][g][0] + shp[tn][g][0];
                            hi = hi * scp[tn][g][1] + shp[tn][g][1];
                        } else {        // the general form's x * 1 + 0 (a -0 leaves as +0)
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
                    if constexpr (MODE == 1) {
                        const unsigned zw[4] = {brcnn_pk2<ET>(v[0]), brcnn_pk2<ET>(v[1]), brcnn_pk2<ET>(v[2]), brcnn_pk2<ET>(v[3])};
                        *reinterpret_cast<uint4*>(zrow + off) = make_uint4(zw[0], zw[1], zw[2], zw[3]);
#pragma unroll
                        for (int e = 0; e < 4; e++) v[e] = brcnn_unpk2<ET>(zw[e]) * sc8p[e] + sh8p[e];
                    }
                    if (RES) {
                        const unsigned rr[4] = {rq[it].x, rq[it].y, rq[it].z, rq[it].w};
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
    float4 scv[NT][4], shv[NT][4];
    if constexpr (MODE == 0) {
#pragma unroll
        for (int tn = 0; tn < NT; tn++)
#pragma unroll
            for (int g = 0; g < 4; g++) {
                const int co = cw0 + tn * 32 + 8 * g + 4 * lh;
                float sc4[4], sh4[4];
#pragma unroll
                for (int e = 0; e < 4; e++) {
                    const bool ok = co + e < p.Cout;
                    sc4[e] = (p.scale && ok) ? p.scale[co + e] : 1.f;
                    sh4[e] = (p.shift && ok) ? p.shift[co + e] : 0.f;
                }
                scv[tn][g] = make_float4(sc4[0], sc4[1], sc4[2], sc4[3]);
                shv[tn][g] = make_float4(sh4[0], sh4[1], sh4[2], sh4[3]);
            }
    }
    unsigned short* __restrict__ yh = reinterpret_cast<unsigned short*>(p.y);
    float* __restrict__ yf = p.y;
#pragma unroll
    for (int tm = 0; tm < MT; tm++) {
        const int mw = m0 + wm * 32 * MT + tm * 32;
        uint4 rq[32 / RPI];
        if (HAS_RQ) {
#pragma unroll
            for (int it = 0; it < 32 / RPI; it++) {
                const int m = mw + it * RPI + rl, co = cw0 + cl;
                rq[it] = make_uint4(0, 0, 0, 0);
                if (vec_ok && m < p.M && co < p.Cout) rq[it] = *reinterpret_cast<const uint4*>(res + (size_t)m * p.Cout + co);
            }
        }
#pragma unroll
        for (int tn = 0; tn < NT; tn++)
#pragma unroll
            for (int g = 0; g < 4; g++) {
                float4 v;
                if constexpr (MODE == 0) {
                    v.x = acc[tm][tn][4 * g + 0] * scv[tn][g].x + shv[tn][g].x;
                    v.y = acc[tm][tn][4 * g + 1] * scv[tn][g].y + shv[tn][g].y;
                    v.z = acc[tm][tn][4 * g + 2] * scv[tn][g].z + shv[tn][g].z;
                    v.w = acc[tm][tn][4 * g + 3] * scv[tn][g].w + shv[tn][g].w;
                } else {        // (the two-buffer kernel's shared code path: x * 1 + 0, which also turns -0 into +0)
                    v.x = acc[tm][tn][4 * g + 0] * 1.f + 0.f; v.y = acc[tm][tn][4 * g + 1] * 1.f + 0.f;
                    v.z = acc[tm][tn][4 * g + 2] * 1.f + 0.f; v.w = acc[tm][tn][4 * g + 3] * 1.f + 0.f;
                }
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
                if constexpr (MODE == 1) {
                    uint4 zq;
                    zq.x = pp_pk2e<ET>(v[0], v[1]);
                    zq.y = pp_pk2e<ET>(v[2], v[3]);
                    zq.z = pp_pk2e<ET>(v[4], v[5]);
                    zq.w = pp_pk2e<ET>(v[6], v[7]);
                    *reinterpret_cast<uint4*>(reinterpret_cast<unsigned short*>(p.z_out) + ro + co) = zq;
                    const unsigned zw[4] = {zq.x, zq.y, zq.z, zq.w};
#pragma unroll
                    for (int e = 0; e < 4; e++) {
                        v[2 * e] = pp_e2f<ET>((unsigned short)(zw[e] & 0xffffu)) * sc8[2 * e] + sh8[2 * e];
                        v[2 * e + 1] = pp_e2f<ET>((unsigned short)(zw[e] >> 16)) * sc8[2 * e + 1] + sh8[2 * e + 1];
                    }
                }
                if constexpr (MODE == 2) {
                    const unsigned zw[4] = {rq[it].x, rq[it].y, rq[it].z, rq[it].w};
#pragma unroll
                    for (int e = 0; e < 8; e++) {
                        const float zz = pp_e2f<ET>((unsigned short)((e & 1) ? (zw[e >> 1] >> 16) : (zw[e >> 1] & 0xffffu)));
                        const float g = pp_e2f<ET>(pp_f2e<ET>(v[e]));
                        const float pre = zz * sc8[e] + sh8[e];
                        const float d = (!p.tail_relu || pre > 0.f) ? g : 0.f;
                        sum_dz[e] += d * zz;
                        sum_d[e] += d;
                        v[e] = d * sc8[e];
                    }
                }
                if (RES && MODE != 2) {
                    const unsigned rr[4] = {rq[it].x, rq[it].y, rq[it].z, rq[it].w};
#pragma unroll
                    for (int e = 0; e < 4; e++) {
                        v[2 * e] += pp_e2f<ET>((unsigned short)(rr[e] & 0xffffu));
                        v[2 * e + 1] += pp_e2f<ET>((unsigned short)(rr[e] >> 16));
                    }
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
                    o.x = pp_pk2e<ET>(v[0], v[1]);
                    o.y = pp_pk2e<ET>(v[2], v[3]);
                    o.z = pp_pk2e<ET>(v[4], v[5]);
                    o.w = pp_pk2e<ET>(v[6], v[7]);
                    *reinterpret_cast<uint4*>(yh + ro + co) = o;
                }
            } else {       // ragged channel count: element-wise tail (MODE 0 only: the host checks)
#pragma unroll
                for (int e = 0; e < 8; e++) {
                    if (co + e >= p.Cout) break;
                    float t = v[e];
                    if (RES) t += pp_e2f<ET>(res[(size_t)m * p.Cout + co + e]);
                    if (p.relu) t = fmaxf(t, 0.f);
                    if (OUTF32) yf[ro + co + e] = t;
                    else yh[ro + co + e] = pp_f2e<ET>(t);
                }
            }
        }
        __builtin_amdgcn_wave_barrier();
    }
    if constexpr (MODE == 2) {
        // lanes that share the channel vector (equal lane % LPR) hold different rows: butterfly over the row bits, then
        // the two wave groups of one channel strip add up through their slabs in a fixed order
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
        if (wm == 0) {
            const int co = cw0 + lane;
            float a = 0.f, b = 0.f;
#pragma unroll
            for (int k = 0; k < WG; k++) {
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

}  // namespace brcnn_conv
