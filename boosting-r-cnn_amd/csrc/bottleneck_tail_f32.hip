// Tail of a frozen stage-1 Bottleneck in ONE fp32 launch (mmdet/models/backbones/resnet.py Bottleneck.forward:263-302):
//     t = relu(bn2(conv2_3x3(x)))  (64 -> 64 channels, stride 1, pad 1)
//     y = relu(bn3(conv3_1x1(t)) + identity)  (64 -> 256 channels)
// The two launches it replaces (conv_igemm_f32_dma_kernel, 64 x 64 tile) cost 343 + 238 us per block at batch 8 x 200 x 336:
// the 1x1 one is HBM-bound (it reads t, reads the identity, writes y: 1.24 GB) while the 3x3 one is MFMA-bound and moves
// hardly anything.  Here a workgroup keeps the 64 x 64 tile of t in LDS -- written in the layout of the A operand, so the
// second GEMM reads it exactly as the first reads its DMA-staged tiles -- and multiplies it by the four 64-channel slices of
// conv3 right away: t never reaches memory, and the identity / output traffic of one workgroup runs under the MFMAs of
// the others (5 workgroups per CU).
//
// Same K order, same MFMA sequence and the same epilogue arithmetic as conv_igemm_f32_dma_kernel<1, 1, .>, and t is an
// fp32 tensor either way: the result equals the two-launch form BIT FOR BIT (tests/test_ops_gpu.py).
// Shapes: x (N,H,W,64), w2 (64,3,3,64), w3 (256,1,1,64), identity / y (N,H,W,256); N*H*W a multiple of 64.
#include "common.h"

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) void* lds_ptr_t;
constexpr int OOB = 0x7fffffff;
constexpr int BM = 64, CM = 64, CO = 256, BK = 32;

struct TailParams {
    const float* x;
    const float* w2;
    const float* s2;
    const float* b2;
    const float* w3;
    const float* s3;
    const float* b3;
    const float* res;
    float* y;
    int H, W, M, tiles_m;
    unsigned x_bytes, w2_bytes, w3_bytes;
};

__global__ __launch_bounds__(256, 2) void bottleneck_tail_f32_kernel(TailParams p) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* As = smem;                   // [2][64][32]: the K tiles of x, then t (K tile 0 | 1 = channels 0-31 | 32-63)
    float* Bs = smem + 2 * BM * 32;     // [2][64][32]: weight K tiles; between the slices of conv3 the waves' read-out slabs
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1;
    const int li = lane & 31, lh = lane >> 5;

    int tile_m;
    {   // consecutive tiles on one XCD (the taps of neighbouring rows re-read the same pixels: one L2)
        const int nwg = p.tiles_m, q = nwg >> 3, r = nwg & 7, xcd = blockIdx.x & 7, loc = blockIdx.x >> 3;
        tile_m = ((xcd < r) ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + loc;
    }
    const int m0 = tile_m * BM;

    const __amdgpu_buffer_rsrc_t rsrc_x = __builtin_amdgcn_make_buffer_rsrc((void*)p.x, 0, (int)p.x_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rsrc_w2 = __builtin_amdgcn_make_buffer_rsrc((void*)p.w2, 0, (int)p.w2_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rsrc_w3 = __builtin_amdgcn_make_buffer_rsrc((void*)p.w3, 0, (int)p.w3_bytes, 0x00020000);

    // ---- DMA assignment (conv_igemm_f32_dma_kernel's): wave w moves the 8-row groups 2w, 2w + 1 of a 64-row tile; lane ->
    // (row in group, physical 16-byte chunk), the chunk it fetches is the logical chunk c ^ ((row >> 1) & 7)
    const int rg = lane >> 3, pc = lane & 7;
    int a_base[2], a_hw[2], a_lc[2], b_row[2];
#pragma unroll
    for (int j = 0; j < 2; j++) {
        const int r = (wave * 2 + j) * 8 + rg;
        a_lc[j] = (pc ^ ((r >> 1) & 7)) * 4;
        const int m = m0 + r;               // < M: M is a multiple of 64
        const int n = m / (p.H * p.W);
        const int rem = m - n * (p.H * p.W);
        const int h = rem / p.W, w = rem - h * p.W;
        a_base[j] = n * p.H * p.W * CM;
        a_hw[j] = ((h - 1 + 4096) << 16) | (w - 1 + 4096);
        b_row[j] = r;
    }

    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; r++) acc[r] = 0.f;

    // K order of the first GEMM: channel chunk by channel chunk, the nine taps inside a chunk (every fp32 kernel's order)
    int d_ci0 = 0, d_kh = 0, d_kw = 0;
    auto dma_tile1 = [&](int buf) {
        const int ci0 = d_ci0, kh = d_kh, kw = d_kw;
        const int k0 = (kh * 3 + kw) * CM + ci0;
        if (++d_kw == 3) {
            d_kw = 0;
            if (++d_kh == 3) { d_kh = 0; d_ci0 += BK; }
        }
#pragma unroll
        for (int j = 0; j < 2; j++) {
            const int hi = (a_hw[j] >> 16) - 4096 + kh, wi = (a_hw[j] & 0xffff) - 4096 + kw;
            const bool ok = ((unsigned)hi < (unsigned)p.H) & ((unsigned)wi < (unsigned)p.W);
            const int off = ok ? (a_base[j] + (hi * p.W + wi) * CM + ci0 + a_lc[j]) * 4 : OOB;
            float* dst = As + buf * BM * 32 + (wave * 2 + j) * 8 * 32;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc_x, (lds_ptr_t)dst, 16, off, 0, 0, 0);
        }
#pragma unroll
        for (int j = 0; j < 2; j++) {
            const int off = (b_row[j] * (9 * CM) + k0 + a_lc[j]) * 4;
            float* dst = Bs + buf * 64 * 32 + (wave * 2 + j) * 8 * 32;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc_w2, (lds_ptr_t)dst, 16, off, 0, 0, 0);
        }
    };
    // the two K tiles (channels 0-31 | 32-63) of conv3's output-channel slice nt -> the two weight buffers
    auto dma_w3 = [&](int nt) {
#pragma unroll
        for (int kt = 0; kt < 2; kt++)
#pragma unroll
            for (int j = 0; j < 2; j++) {
                const int off = ((nt * 64 + b_row[j]) * CM + kt * BK + a_lc[j]) * 4;
                float* dst = Bs + kt * 64 * 32 + (wave * 2 + j) * 8 * 32;
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc_w3, (lds_ptr_t)dst, 16, off, 0, 0, 0);
            }
    };

    dma_tile1(0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();

    // fragment reads as inline asm (a compiler-visible LDS read after `buffer_load ... lds` gets a vmcnt(0) in front of it)
    const int sw = (li >> 1) & 7;
    unsigned chb[BK / 8];
#pragma unroll
    for (int kk = 0; kk < BK / 8; kk++) chb[kk] = (unsigned)(((2 * kk + lh) ^ sw) * 16);
    const unsigned a_lane = (unsigned)(size_t)(lds_ptr_t)(As + (wm * 32 + li) * 32);
    const unsigned b_lane = (unsigned)(size_t)(lds_ptr_t)(Bs + (wn * 32 + li) * 32);
    f32x4 av[2], bv[2];
    auto frag_read = [&](int slot, unsigned a_addr, unsigned b_addr) {
        asm volatile("ds_read_b128 %0, %1" : "=v"(av[slot]) : "v"(a_addr) : "memory");
        asm volatile("ds_read_b128 %0, %1" : "=v"(bv[slot]) : "v"(b_addr) : "memory");
    };
    auto frag_wait = [&](int slot) { asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(av[slot]), "+v"(bv[slot]) :: "memory"); };
    auto compute_tile = [&](f32x16& c, int cur, bool prefetch) {
        const unsigned a_cur = a_lane + cur * (BM * 32 * 4);
        const unsigned b_cur = b_lane + cur * (64 * 32 * 4);
        __builtin_amdgcn_s_setprio(1);
        frag_read(0, a_cur + chb[0], b_cur + chb[0]);
        if (prefetch) dma_tile1(cur ^ 1);
        frag_wait(0);
        __builtin_amdgcn_s_setprio(0);
#pragma unroll
        for (int kk = 0; kk < BK / 8; kk++) {
            const int sl = kk & 1;
            if (kk + 1 < BK / 8) frag_read(sl ^ 1, a_cur + chb[kk + 1], b_cur + chb[kk + 1]);
#pragma unroll
            for (int e = 0; e < 4; e++) c = __builtin_amdgcn_mfma_f32_32x32x2f32(av[sl][e], bv[sl][e], c, 0, 0, 0);
            if (kk + 1 < BK / 8) frag_wait(sl ^ 1);
        }
    };

    // ---- GEMM 1: 18 K tiles (2 channel chunks x 9 taps)
    constexpr int NK1 = 9 * CM / BK;
    int cur = 0;
    for (int kt = 0; kt + 1 < NK1; kt++) {
        compute_tile(acc, cur, true);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        cur ^= 1;
    }
    compute_tile(acc, cur, false);
    __syncthreads();            // every wave is done with the operand buffers

    // ---- t = relu(acc * s2 + b2) into the A buffers, in the A operand's layout: element (row, ch) -> K tile ch / 32, row
    // `row`, logical chunk (ch % 32) / 4 at physical chunk c ^ ((row >> 1) & 7).  (The accumulator holds column ch = wn 32 +
    // li of rows wm 32 + (r & 3) + 8 (r >> 2) + 4 lh.)  Meanwhile the first slice of conv3 is on its way.
    dma_w3(0);
    {
        const int ch = wn * 32 + li;
        const float sc = p.s2 ? p.s2[ch] : 1.f;
        const float sh = p.b2 ? p.b2[ch] : 0.f;
        float* tbuf = As + wn * BM * 32;
        const int c = li >> 2, e = li & 3;
#pragma unroll
        for (int r = 0; r < 16; r++) {
            const int row = wm * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
            float v = acc[r];
            if (p.s2) v = v * sc;
            v = fmaxf(v + sh, 0.f);
            tbuf[row * 32 + ((c ^ ((row >> 1) & 7)) << 2) + e] = v;
        }
    }

    // ---- GEMM 2 + read-out, one 64-channel slice of conv3 at a time
    const int vrow = lane >> 3, vcol = (lane & 7) * 4;
    float* slab = Bs + wave * 1024;
    // identity tile of a slice: rows it*8 + vrow, 4 channels at vcol.  Requested one slice ahead, and the stores of a slice
    // are issued LAST -- behind the next slice's weight DMA and identity loads: vmcnt retires in issue order, so a wait for
    // the DMA would otherwise also wait for the acknowledgement of every store in front of it
    auto row_of = [&](int nt) { return (size_t)(m0 + wm * 32 + vrow) * CO + nt * 64 + wn * 32 + vcol; };
    float4 rv[4];
#pragma unroll
    for (int it = 0; it < 4; it++) rv[it] = *reinterpret_cast<const float4*>(p.res + row_of(0) + (size_t)(it * 8) * CO);
    for (int nt = 0; nt < CO / 64; nt++) {
        // the weight DMA and the identity loads of this slice (and t's LDS stores) are complete; only the previous slice's
        // four stores may still be in flight (__syncthreads() would wait for them too: it drains every counter)
        asm volatile("s_waitcnt vmcnt(4) lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        f32x16 c2;
#pragma unroll
        for (int r = 0; r < 16; r++) c2[r] = 0.f;
        compute_tile(c2, 0, false);
        compute_tile(c2, 1, false);
        asm volatile("" ::: "memory");
        __builtin_amdgcn_s_barrier();       // (the fragment reads were waited for) the weight buffers become the waves' slabs
        asm volatile("" ::: "memory");
        const int co = nt * 64 + wn * 32 + li;
        const float sc = p.s3 ? p.s3[co] : 1.f;
        const float sh = p.b3 ? p.b3[co] : 0.f;
#pragma unroll
        for (int r = 0; r < 16; r++) {
            float v = c2[r];
            if (p.s3) v = v * sc;
            slab[((r & 3) + 8 * (r >> 2) + 4 * lh) * 32 + li] = v + sh;
        }
        __builtin_amdgcn_s_waitcnt(0xc07f);
        __builtin_amdgcn_wave_barrier();
        float4 out[4];
#pragma unroll
        for (int it = 0; it < 4; it++) {
            float4 v = *reinterpret_cast<const float4*>(slab + (it * 8 + vrow) * 32 + vcol);
            v.x += rv[it].x; v.y += rv[it].y; v.z += rv[it].z; v.w += rv[it].w;
            v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f);
            out[it] = v;
        }
        if (nt + 1 < CO / 64) {
            // every wave has read its slab: the buffers take the next slice's weights, the identity loads follow
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            asm volatile("" ::: "memory");
            dma_w3(nt + 1);
#pragma unroll
            for (int it = 0; it < 4; it++) rv[it] = *reinterpret_cast<const float4*>(p.res + row_of(nt + 1) + (size_t)(it * 8) * CO);
        }
        asm volatile("" ::: "memory");
        float* __restrict__ yrow = p.y + row_of(nt);
#pragma unroll
        for (int it = 0; it < 4; it++) *reinterpret_cast<float4*>(yrow + (size_t)(it * 8) * CO) = out[it];
    }
}

}  // namespace

BRCNN_API int brcnn_bottleneck_tail_f32(const float* x, const float* w2, const float* scale2, const float* shift2,
                                        const float* w3, const float* scale3, const float* shift3, const float* identity,
                                        float* y, int batch, int height, int width, void* stream) {
    if (!x || !w2 || !w3 || !identity || !y || batch <= 0 || height <= 0 || width <= 0 || height >= 4096 || width >= 4096)
        return BRCNN_EINVAL;
    const long long m = (long long)batch * height * width;
    if ((m & 63) || m * CO * 4 >= 0x7fffffffLL * 4 || m * CM * 4 >= 0x7fffffffLL) return BRCNN_EINVAL;
    TailParams p;
    p.x = x; p.w2 = w2; p.s2 = scale2; p.b2 = shift2; p.w3 = w3; p.s3 = scale3; p.b3 = shift3; p.res = identity; p.y = y;
    p.H = height; p.W = width; p.M = (int)m; p.tiles_m = (int)(m / BM);
    p.x_bytes = (unsigned)(m * CM * 4); p.w2_bytes = 64 * 9 * CM * 4; p.w3_bytes = CO * CM * 4;
    const size_t lds = (size_t)2 * (BM + 64) * 32 * sizeof(float);
    hipLaunchKernelGGL(bottleneck_tail_f32_kernel, dim3(p.tiles_m), dim3(256), lds, (hipStream_t)stream, p);
    BRCNN_LAUNCH_CHECK();
    return 0;
}
