// 16-bit form of bottleneck_tail_f32.hip: the tail of a frozen stage-1 Bottleneck in ONE launch,
//     t = relu(bn2(conv2_3x3(x)))  (64 -> 64 channels, stride 1, pad 1; t rounded to the 16-bit type, as the two-launch
//                                   form stores it)
//     y = relu(bn3(conv3_1x1(t)) + identity)  (64 -> 256 channels)
// bf16 / fp16 operands, fp32 accumulation (v_mfma_f32_32x32x16).  In the 16-bit modes neither of the two launches it
// replaces is MFMA-bound: the 3x3 is bound by its LDS-DMA bytes (N = 64: 82 us at batch 8 x 200 x 336), the 1x1 by HBM
// (identity + output + t: 619 MB, 110 us on the persistent streaming kernel) -- different paths, used one after the other.
// Here a workgroup (4 waves, 128 rows) runs the nine K tiles of the 3x3 as conv_igemm_bf16_dma_kernel<2, 1> does, writes
// its 128 x 64 tile of t (bn2 + ReLU, rounded) into the free A buffer in the A operand's swizzled row layout, and
// multiplies it by the four 64-channel slices of conv3 (one K tile each); the identity rows of a slice are requested
// before its MFMAs, the read-out goes through per-wave fp32 slabs as in the conv kernels.  Same K order and MFMA operand
// order as those kernels, same epilogue arithmetic (x * scale + shift, + identity, one rounding, ReLU).
// Shapes: x (N,H,W,64), w2 (64,3,3,64), w3 (256,1,1,64), identity / y (N,H,W,256), all 16-bit; N*H*W a multiple of 128.
#include "common.h"

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef __attribute__((address_space(3))) void* lds_ptr_t;
constexpr int OOB = 0x7fffffff;
constexpr int BM = 128, CM = 64, CO = 256;
constexpr int A_BUF = BM * 128, B_BUF = 64 * 128;        // bytes of one operand buffer (rows of 64 elements = 128 bytes)

struct Tail16Params {
    const unsigned short* x;
    const unsigned short* w2;
    const float* s2;
    const float* b2;
    const unsigned short* w3;
    const float* s3;
    const float* b3;
    const unsigned short* res;
    unsigned short* y;
    int H, W, M, tiles_m;
    unsigned x_bytes, w2_bytes, w3_bytes;
};

template <int ET> __device__ __forceinline__ float e2f(unsigned short h) { return ET ? brcnn_h2f(h) : brcnn_b2f(h); }
template <int ET> __device__ __forceinline__ unsigned short f2e(float v) { return ET ? brcnn_f2h(v) : brcnn_f2b(v); }

template <int ET>       // 0 bf16, 1 fp16
__global__ __launch_bounds__(256, 2) void bottleneck_tail_16_kernel(Tail16Params p) {
    // LDS (bytes): A buffers [0, 32K) (two K tiles of x; later buffer 0 = t, buffer 1 = the four waves' read-out slabs),
    // B buffers [32K, 48K) (weight K tiles; second GEMM: conv3's slices alternate between them, slice nt + 1 on its way
    // while slice nt is multiplied and read out)
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    unsigned char* As = smem;
    unsigned char* Bs = smem + 2 * A_BUF;
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1;        // 2 x 2 waves: 64 rows (two 32-row MFMA tiles) x 32 channels each
    const int li = lane & 31, lh = lane >> 5;

    int tile_m;
    {   // consecutive tiles on one XCD
        const int nwg = p.tiles_m, q = nwg >> 3, r = nwg & 7, xcd = blockIdx.x & 7, loc = blockIdx.x >> 3;
        tile_m = ((xcd < r) ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + loc;
    }
    const int m0 = tile_m * BM;

    const __amdgpu_buffer_rsrc_t rsrc_x = __builtin_amdgcn_make_buffer_rsrc((void*)p.x, 0, (int)p.x_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rsrc_w2 = __builtin_amdgcn_make_buffer_rsrc((void*)p.w2, 0, (int)p.w2_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rsrc_w3 = __builtin_amdgcn_make_buffer_rsrc((void*)p.w3, 0, (int)p.w3_bytes, 0x00020000);

    // ---- DMA assignment: a wave moves 8-row groups (a lane: row in group, physical 16-byte chunk; it fetches the logical
    // chunk c ^ ((row >> 1) & 7)): four groups of the 128-row A tile, two of the 64-row weight tile
    const int rg = lane >> 3, pc = lane & 7;
    int a_base[4], a_hw[4], a_lc[4], b_row[2], b_lc[2];
#pragma unroll
    for (int j = 0; j < 4; j++) {
        const int r = (wave * 4 + j) * 8 + rg;
        a_lc[j] = (pc ^ ((r >> 1) & 7)) * 8;            // elements
        const int m = m0 + r;                           // < M: M is a multiple of 128
        const int n = m / (p.H * p.W);
        const int rem = m - n * (p.H * p.W);
        const int h = rem / p.W, w = rem - h * p.W;
        a_base[j] = n * p.H * p.W * CM;
        a_hw[j] = ((h - 1 + 4096) << 16) | (w - 1 + 4096);
    }
#pragma unroll
    for (int j = 0; j < 2; j++) {
        const int r = (wave * 2 + j) * 8 + rg;
        b_row[j] = r;
        b_lc[j] = (pc ^ ((r >> 1) & 7)) * 8;
    }

    f32x16 acc[2];
#pragma unroll
    for (int a = 0; a < 2; a++)
#pragma unroll
        for (int r = 0; r < 16; r++) acc[a][r] = 0.f;

    // first GEMM: one K tile per filter tap (64 channels = one 128-byte row), taps in (kh, kw) order
    int d_tap = 0;
    auto dma_tile1 = [&](int buf) {
        const int tap = d_tap++;
        const int kh = tap / 3, kw = tap - kh * 3;
#pragma unroll
        for (int j = 0; j < 4; j++) {
            const int hi = (a_hw[j] >> 16) - 4096 + kh, wi = (a_hw[j] & 0xffff) - 4096 + kw;
            const bool ok = ((unsigned)hi < (unsigned)p.H) & ((unsigned)wi < (unsigned)p.W);
            const int off = ok ? (a_base[j] + (hi * p.W + wi) * CM + a_lc[j]) * 2 : OOB;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc_x, (lds_ptr_t)(As + buf * A_BUF + (wave * 4 + j) * 1024), 16, off, 0, 0, 0);
        }
#pragma unroll
        for (int j = 0; j < 2; j++) {
            const int off = (b_row[j] * (9 * CM) + tap * CM + b_lc[j]) * 2;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc_w2, (lds_ptr_t)(Bs + buf * B_BUF + (wave * 2 + j) * 1024), 16, off, 0, 0, 0);
        }
    };
    // conv3's output-channel slice nt (64 x 64) -> weight buffer nt & 1
    auto dma_w3 = [&](int nt) {
#pragma unroll
        for (int j = 0; j < 2; j++) {
            const int off = ((nt * 64 + b_row[j]) * CM + b_lc[j]) * 2;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc_w3, (lds_ptr_t)(Bs + (nt & 1) * B_BUF + (wave * 2 + j) * 1024), 16, off, 0, 0, 0);
        }
    };

    dma_tile1(0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();

    // fragment reads as inline asm (conv_igemm_bf16.hip): row R = base + li, logical chunk 2 kk + lh at physical chunk
    // c ^ ((R >> 1) & 7); the second 32-row tile of a wave lies 4096 bytes further
    const int sw = (li >> 1) & 7;
    unsigned chb[4];
#pragma unroll
    for (int kk = 0; kk < 4; kk++) chb[kk] = (unsigned)(((2 * kk + lh) ^ sw) * 16);
    const unsigned a_lane = (unsigned)(size_t)(lds_ptr_t)(As + (wm * 64 + li) * 128);
    const unsigned b_lane = (unsigned)(size_t)(lds_ptr_t)(Bs + (wn * 32 + li) * 128);
    f32x4 av[2][2], bv[2];
    auto frag_read = [&](int slot, unsigned a_addr, unsigned b_addr) {
        asm volatile("ds_read_b128 %0, %1" : "=v"(av[slot][0]) : "v"(a_addr) : "memory");
        asm volatile("ds_read_b128 %0, %1 offset:4096" : "=v"(av[slot][1]) : "v"(a_addr) : "memory");
        asm volatile("ds_read_b128 %0, %1" : "=v"(bv[slot]) : "v"(b_addr) : "memory");
    };
    auto frag_wait = [&](int slot) {
        asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(av[slot][0]), "+v"(av[slot][1]), "+v"(bv[slot]) :: "memory");
    };
    auto mma = [&](f32x16& c, const f32x4& b, const f32x4& a) {
        if constexpr (ET) c = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, b), __builtin_bit_cast(f16x8, a), c, 0, 0, 0);
        else c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, b), __builtin_bit_cast(bf16x8, a), c, 0, 0, 0);
    };
    // one K tile: A from byte offset a_off of the A region, weights from b_off of the B region
    auto compute_tile = [&](f32x16 (&c)[2], unsigned a_off, unsigned b_off, int prefetch_buf) {
        const unsigned a_cur = a_lane + a_off, b_cur = b_lane + b_off;
        __builtin_amdgcn_s_setprio(1);
        frag_read(0, a_cur + chb[0], b_cur + chb[0]);
        if (prefetch_buf >= 0) dma_tile1(prefetch_buf);
        frag_wait(0);
        __builtin_amdgcn_s_setprio(0);
#pragma unroll
        for (int kk = 0; kk < 4; kk++) {
            const int sl = kk & 1;
            if (kk + 1 < 4) frag_read(sl ^ 1, a_cur + chb[kk + 1], b_cur + chb[kk + 1]);
            mma(c[0], bv[sl], av[sl][0]);
            mma(c[1], bv[sl], av[sl][1]);
            if (kk + 1 < 4) frag_wait(sl ^ 1);
        }
    };

    // ---- GEMM 1: nine K tiles
    int cur = 0;
    for (int kt = 0; kt + 1 < 9; kt++) {
        compute_tile(acc, cur * A_BUF, cur * B_BUF, cur ^ 1);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        cur ^= 1;
    }
    compute_tile(acc, cur * A_BUF, cur * B_BUF, -1);
    __syncthreads();            // every wave is done with the operand buffers (cur == 0 here: nine tiles)

    // ---- t = round(relu(acc * s2 + b2)) into A buffer 0 in the A operand's layout.  (D^T = W A^T: lane l holds pixel row
    // l & 31 of its tile and, per register group g, the four channels 8 g + 4 (l >> 5) + (0..3) of the wave's 32.)
    dma_w3(0);
    {
#pragma unroll
        for (int g = 0; g < 4; g++) {
            const int ch = wn * 32 + 8 * g + 4 * lh;
            float sc[4], sh[4];
#pragma unroll
            for (int e = 0; e < 4; e++) { sc[e] = p.s2 ? p.s2[ch + e] : 1.f; sh[e] = p.b2 ? p.b2[ch + e] : 0.f; }
#pragma unroll
            for (int tm = 0; tm < 2; tm++) {
                const int row = wm * 64 + tm * 32 + li;
                unsigned short t4[4];
#pragma unroll
                for (int e = 0; e < 4; e++) {
                    float v = acc[tm][4 * g + e];
                    v = v * sc[e] + sh[e];
                    t4[e] = f2e<ET>(fmaxf(v, 0.f));
                }
                uint2 q;
                q.x = (unsigned)t4[0] | ((unsigned)t4[1] << 16);
                q.y = (unsigned)t4[2] | ((unsigned)t4[3] << 16);
                const int c = wn * 4 + g;                   // logical 16-byte chunk of the row
                *reinterpret_cast<uint2*>(As + row * 128 + ((c ^ ((row >> 1) & 7)) * 16) + lh * 8) = q;
            }
        }
    }

    // ---- GEMM 2 + read-out, one 64-channel slice of conv3 at a time
    // this wave's slab: 32 rows x 32 channels, the 16-byte pieces of row r at piece ^ (r & 7) (unpadded rows, conflict-free
    // float4 writes of eight rows at a time)
    float* cs = reinterpret_cast<float*>(smem + A_BUF) + wave * 32 * 32;
    const int rl = lane >> 2, cl = (lane & 3) * 8;      // read-out: 4 lanes per row, 8 channels each, 16 rows per pass
    // identity rows of a slice, requested one slice ahead -- in front of the current slice's stores: vmcnt retires in issue
    // order, and the wait at a slice's start (its weight DMA, its identity rows) must not include store acknowledgements
    uint4 rq[2][2], rqn[2][2];
    auto load_identity = [&](int nt, uint4 (&q)[2][2]) {
#pragma unroll
        for (int tm = 0; tm < 2; tm++)
#pragma unroll
            for (int it = 0; it < 2; it++)
                q[tm][it] = *reinterpret_cast<const uint4*>(p.res + (size_t)(m0 + wm * 64 + tm * 32 + it * 16 + rl) * CO + nt * 64 + wn * 32 + cl);
    };
    load_identity(0, rq);
    for (int nt = 0; nt < CO / 64; nt++) {
        const int co0 = nt * 64 + wn * 32;
        // this slice's weight DMA, its identity rows and t's LDS stores are complete; only the previous slice's four stores
        // may still be in flight.  Behind the barrier every wave is done with the OTHER weight buffer: slice nt + 1 goes there
        asm volatile("s_waitcnt vmcnt(4) lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        if (nt + 1 < CO / 64) { dma_w3(nt + 1); load_identity(nt + 1, rqn); }
        f32x16 c2[2];
#pragma unroll
        for (int a = 0; a < 2; a++)
#pragma unroll
            for (int r = 0; r < 16; r++) c2[a][r] = 0.f;
        compute_tile(c2, 0, (nt & 1) * B_BUF, -1);
        // (slabs: A buffer 1, which this GEMM does not read -- no barrier needed in front of the slab writes)
        float sc8[8], sh8[8];
#pragma unroll
        for (int e = 0; e < 8; e++) { sc8[e] = p.s3 ? p.s3[co0 + cl + e] : 1.f; sh8[e] = p.b3 ? p.b3[co0 + cl + e] : 0.f; }
#pragma unroll
        for (int tm = 0; tm < 2; tm++) {
#pragma unroll
            for (int g = 0; g < 4; g++)
                *reinterpret_cast<float4*>(cs + li * 32 + ((8 * g + 4 * lh) ^ ((li & 7) << 2))) =
                    make_float4(c2[tm][4 * g + 0], c2[tm][4 * g + 1], c2[tm][4 * g + 2], c2[tm][4 * g + 3]);
            __builtin_amdgcn_s_waitcnt(0xc07f);
            __builtin_amdgcn_wave_barrier();
#pragma unroll
            for (int it = 0; it < 2; it++) {
                const int row = it * 16 + rl;
                const float4 lo = *reinterpret_cast<const float4*>(cs + row * 32 + (cl ^ ((row & 7) << 2)));
                const float4 hi = *reinterpret_cast<const float4*>(cs + row * 32 + ((cl + 4) ^ ((row & 7) << 2)));
                float v[8] = {lo.x, lo.y, lo.z, lo.w, hi.x, hi.y, hi.z, hi.w};
                const unsigned rr[4] = {rq[tm][it].x, rq[tm][it].y, rq[tm][it].z, rq[tm][it].w};
                unsigned short o8[8];
#pragma unroll
                for (int e = 0; e < 8; e++) {
                    float t = v[e] * sc8[e] + sh8[e];
                    t += e2f<ET>((unsigned short)((e & 1) ? (rr[e >> 1] >> 16) : (rr[e >> 1] & 0xffffu)));
                    const unsigned short h = f2e<ET>(t);
                    o8[e] = (h & 0x8000u) ? (unsigned short)0 : h;          // ReLU on the rounded value (sign bit set: <= -0)
                }
                uint4 o;
                o.x = (unsigned)o8[0] | ((unsigned)o8[1] << 16); o.y = (unsigned)o8[2] | ((unsigned)o8[3] << 16);
                o.z = (unsigned)o8[4] | ((unsigned)o8[5] << 16); o.w = (unsigned)o8[6] | ((unsigned)o8[7] << 16);
                *reinterpret_cast<uint4*>(p.y + (size_t)(m0 + wm * 64 + tm * 32 + row) * CO + co0 + cl) = o;
            }
            __builtin_amdgcn_wave_barrier();
        }
#pragma unroll
        for (int tm = 0; tm < 2; tm++)
#pragma unroll
            for (int it = 0; it < 2; it++) rq[tm][it] = rqn[tm][it];
    }
}

template <int ET>
int launch_tail16(const Tail16Params& p, hipStream_t s) {
    const size_t lds = (size_t)2 * A_BUF + 2 * B_BUF;
    hipLaunchKernelGGL((bottleneck_tail_16_kernel<ET>), dim3(p.tiles_m), dim3(256), lds, s, p);
    BRCNN_LAUNCH_CHECK();
    return 0;
}

}  // namespace

BRCNN_API int brcnn_bottleneck_tail_16(const void* x, const void* w2, const float* scale2, const float* shift2, const void* w3,
                                       const float* scale3, const float* shift3, const void* identity, void* y, int batch,
                                       int height, int width, int dtype, void* stream) {
    if (!x || !w2 || !w3 || !identity || !y || batch <= 0 || height <= 0 || width <= 0 || height >= 4096 || width >= 4096 ||
        (dtype != BRCNN_DT_BF16 && dtype != BRCNN_DT_F16))
        return BRCNN_EINVAL;
    const long long m = (long long)batch * height * width;
    if ((m & 127) || m * CO * 2 >= 0x7fffffffLL * 2 || m * CM * 2 >= 0x7fffffffLL) return BRCNN_EINVAL;
    static_assert(4 * 32 * 32 * 4 <= A_BUF, "the four slabs fit A buffer 1");
    Tail16Params p;
    p.x = (const unsigned short*)x; p.w2 = (const unsigned short*)w2; p.s2 = scale2; p.b2 = shift2;
    p.w3 = (const unsigned short*)w3; p.s3 = scale3; p.b3 = shift3; p.res = (const unsigned short*)identity; p.y = (unsigned short*)y;
    p.H = height; p.W = width; p.M = (int)m; p.tiles_m = (int)(m / BM);
    p.x_bytes = (unsigned)(m * CM * 2); p.w2_bytes = 64 * 9 * CM * 2; p.w3_bytes = CO * CM * 2;
    return dtype == BRCNN_DT_F16 ? launch_tail16<1>(p, (hipStream_t)stream) : launch_tail16<0>(p, (hipStream_t)stream);
}
