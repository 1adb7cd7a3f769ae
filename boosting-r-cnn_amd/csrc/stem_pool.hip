// Frozen ResNet stem in ONE launch: 7x7 / stride 2 / pad 3 convolution of the 3-channel NCHW image + folded
// BatchNorm + ReLU + 3x3 / stride 2 / pad 1 max-pool -> (N, Hp, Wp, 64) NHWC (mmdet/models/backbones/resnet.py:599-611
// conv1 / norm1 / relu / maxpool, forward:631-636).  Round 6: the two-launch form (`brcnn_stem7x7s2_nchw` +
// `brcnn_maxpool3x3s2_nhwc`) cost 665 + 152 us in fp32 and 245 + 85 us in bf16 at batch 8 x 800 x 1344 -- the generic
// implicit-GEMM kernel stages every output pixel's 7 x 8-pixel window separately (windows of neighbouring outputs overlap
// by 6 of 8 pixels; fp32: K = 7 x 32 for 147 real products) and the 2 150 400 x 64 conv output crosses HBM twice.
//
// Here a workgroup (4 waves) owns 7 x 8 POOLED outputs of one image:
//   * the 35 x 39 input pixels under them go from the NCHW planes into LDS once, interleaved (fp32: 3 floats per pixel,
//     16-bit: 4 elements per pixel), so the window of conv output (r, c) and filter row kh is the contiguous run that
//     starts at pixel (2r + kh, 2c): neighbouring outputs read overlapping LDS addresses, nothing is duplicated;
//   * the 15 x 17 conv outputs the pool windows need (255 of the tile's 256 GEMM rows; the one-row / one-column halo is
//     recomputed by the neighbour tile: 256 rows per 224 useful) x 64 channels are 2 x 2 MFMA tiles per wave:
//     fp32 v_mfma_f32_32x32x2_f32 over K = 7 x 22 (21 real + 1 zero per filter row, 77 steps),
//     16-bit v_mfma_f32_32x32x16 over K = 7 x 32 (14 steps), weights resident in LDS;
//   * epilogue per 32-channel half: scale / shift / ReLU -> LDS (aliasing the input tile) -> 3x3/s2 max over the valid
//     conv positions -> one store of the pooled row piece.  The conv output never reaches memory.
// HBM: the image once (+ halo re-reads that hit in L2) and the pooled map once.
#include "common.h"

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

constexpr int PH = 7, PW = 8;                   // pooled outputs per workgroup
constexpr int CR = 2 * PH + 1, CC = 2 * PW + 1; // conv outputs under them: 15 x 17
constexpr int NPOS = CR * CC;                   // 255 (GEMM row 255 repeats row 254 and is ignored)
constexpr int IR = 2 * CR + 5;                  // 35 input rows
constexpr int IC = 40;                          // 39 input columns + the column the dead eighth tap touches
constexpr int KROW32 = 22;                      // fp32: floats per filter row in K (7 pixels x 3 channels + one zero)
constexpr int K32 = 7 * KROW32;
constexpr int WROW16 = 232;                     // 16-bit: elements per output channel in LDS (7 x 32 + 8: 464-byte rows keep the b128 reads of 32 channels conflict-free)

struct StemPoolParams {
    const float* img;
    const void* w;
    const float* scale;
    const float* shift;
    void* y;
    int N, H, W, Ho, Wo, PHo, PWo, tiles_i, tiles_j;
};

template <int ET> struct StemPoolLds {
    static constexpr int W_BYTES = ET == 0 ? K32 * 64 * 4 : 64 * WROW16 * 2;
    static constexpr int IN_BYTES = ET == 0 ? IR * IC * 3 * 4 : IR * IC * 8;
    static constexpr int EP_BYTES = 256 * 32 * 4;
    static constexpr int BYTES = W_BYTES + (IN_BYTES > EP_BYTES ? IN_BYTES : EP_BYTES);
};

// workgroup barrier for LDS traffic only: __syncthreads() also waits for the global stores in flight (vmcnt(0)) -- the
// pooled rows of one channel half would have to reach memory before the other half's values may enter LDS
__device__ __forceinline__ void lds_barrier() {
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
}

template <int ET>       // 0 fp32, 1 bf16, 2 fp16
__global__ __launch_bounds__(256) void stem_pool_kernel(StemPoolParams p) {
    extern __shared__ __align__(16) unsigned char smem[];
    typedef StemPoolLds<ET> L;
    unsigned char* w_lds = smem;
    unsigned char* in_lds = smem + L::W_BYTES;
    float* ep = reinterpret_cast<float*>(smem + L::W_BYTES);        // aliases the input tile
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int li = lane & 31, lh = lane >> 5;
    const int total = p.N * p.tiles_i * p.tiles_j;

    // ---- weights (already in the LDS layout) -> LDS, once per (persistent) workgroup
    {
        const uint4* src = reinterpret_cast<const uint4*>(p.w);
        uint4* dst = reinterpret_cast<uint4*>(w_lds);
#pragma unroll
        for (int it = 0; it < (L::W_BYTES / 16 + 255) / 256; it++) {
            const int i = tid + it * 256;
            if (i < L::W_BYTES / 16) dst[i] = src[i];
        }
    }

    // the input tile of `tile` -> registers.  Every load is issued before anything waits on one (clamped address + select
    // instead of a branch: a conditional load per iteration ran the six iterations as six memory latencies back to back)
    constexpr int ITS = (IR * IC + 255) / 256;
    const size_t plane = (size_t)p.H * p.W;
    int st_r[ITS], st_x[ITS];
#pragma unroll
    for (int it = 0; it < ITS; it++) {
        int idx = tid + it * 256;
        if (idx > IR * IC - 1) idx = IR * IC - 1;
        st_r[it] = idx / IC;
        st_x[it] = idx - st_r[it] * IC;
    }
    auto tile_coords = [&](int tile, int& n, int& i0, int& j0) {
        const int tj = tile % p.tiles_j;
        const int q = tile / p.tiles_j;
        const int ti = q % p.tiles_i;
        n = q / p.tiles_i;
        i0 = ti * PH;
        j0 = tj * PW;
    };
    auto load_tile = [&](int tile, float (&v)[ITS][3]) {
        int n, i0, j0;
        tile_coords(tile, n, i0, j0);
        const int ir0 = 4 * i0 - 5, ic0 = 4 * j0 - 5;        // first input pixel: conv output 2 i0 - 1 (pool padding), minus 3 (conv padding)
        const float* img = p.img + (size_t)n * 3 * plane;
#pragma unroll
        for (int it = 0; it < ITS; it++) {
            const int gr = ir0 + st_r[it], gc = ic0 + st_x[it];
            const bool ok = (unsigned)gr < (unsigned)p.H && (unsigned)gc < (unsigned)p.W;
            const int grc = min(max(gr, 0), p.H - 1), gcc = min(max(gc, 0), p.W - 1);
            const float* q = img + (size_t)grc * p.W + gcc;
            const float t0 = q[0], t1 = q[plane], t2 = q[2 * plane];
            v[it][0] = ok ? t0 : 0.f; v[it][1] = ok ? t1 : 0.f; v[it][2] = ok ? t2 : 0.f;
        }
    };

    // per-thread constants of the K loop: the two conv positions of this lane's accumulator rows
    int pos_r[2], pos_c[2];
#pragma unroll
    for (int tm = 0; tm < 2; tm++) {
        int pp = (wave * 2 + tm) * 32 + li;
        if (pp > NPOS - 1) pp = NPOS - 1;
        pos_r[tm] = pp / CC;
        pos_c[tm] = pp - pos_r[tm] * CC;
    }
    // ... and of the epilogue: thread t pools column pj = t / 32 of every pooled row of the tile, channel t % 32 of the half
    const int pj = tid >> 5;
    const float* ep_rd = ep + (2 * pj) * 32 + li;
    float* ep_wr = ep + ((wave * 2) * 32 + 4 * lh) * 32 + li;
    float bn_sc[2], bn_sh[2];
#pragma unroll
    for (int tn = 0; tn < 2; tn++) {
        bn_sc[tn] = p.scale ? p.scale[tn * 32 + li] : 1.f;
        bn_sh[tn] = p.shift ? p.shift[tn * 32 + li] : 0.f;
    }

    float nxt[ITS][3];
    // tile order: workgroup b runs on XCD b % 8 (each with its own L2); XCD x takes the x-th eighth of the tile list and
    // its workgroups walk it side by side, so tiles that share halo pixels (35 x 39 input pixels per 28 x 32 net) meet in
    // ONE L2 instead of being fetched into eight (the plain b, b + grid, ... order: 357 MB of HBM reads per launch for a
    // 103 MB image batch, r06_conv_traffic.json).  A grid that is no multiple of 8 (fewer tiles than slots) keeps the plain order.
    int tile, tile_end = total, step = gridDim.x;
    if ((gridDim.x & 7) == 0) {
        const int per = (total + 7) >> 3, xcd = blockIdx.x & 7;
        tile = xcd * per + (int)(blockIdx.x >> 3);
        tile_end = min(total, (xcd + 1) * per);
        step = gridDim.x >> 3;
    } else {
        tile = blockIdx.x;
    }
    if (tile < tile_end) load_tile(tile, nxt);
    for (; tile < tile_end; tile += step) {
        int n, i0, j0;
        tile_coords(tile, n, i0, j0);
        const int cr0 = 2 * i0 - 1, cc0 = 2 * j0 - 1;   // first conv output of the tile (may be -1: pool padding)
        // ---- this tile's pixels: registers -> LDS (the previous tile's epilogue ended with a barrier); then the next
        // tile's loads go out and stay in flight under the K loop and the epilogue
#pragma unroll
        for (int it = 0; it < ITS; it++) {
            const int idx = tid + it * 256;
            if (idx < IR * IC) {
                if (ET == 0) {
                    float* d = reinterpret_cast<float*>(in_lds) + idx * 3;
                    d[0] = nxt[it][0]; d[1] = nxt[it][1]; d[2] = nxt[it][2];
                } else {
                    uint2 u;
                    u.x = ET == 1 ? brcnn_pk2b(nxt[it][0], nxt[it][1]) : brcnn_pk2h(nxt[it][0], nxt[it][1]);
                    u.y = ET == 1 ? (unsigned)brcnn_f2b(nxt[it][2]) : (unsigned)brcnn_f2h(nxt[it][2]);
                    reinterpret_cast<uint2*>(in_lds)[idx] = u;
                }
            }
        }
        lds_barrier();
        if (tile + step < tile_end) load_tile(tile + step, nxt);

        // ---- 256 x 64 outputs = 2 x 2 MFMA tiles per wave.  Operand A = conv positions (accumulator rows), B = channels
        // (accumulator column = lane & 31)
        f32x16 acc[2][2];
#pragma unroll
        for (int a = 0; a < 2; a++)
#pragma unroll
            for (int c = 0; c < 2; c++)
#pragma unroll
                for (int r = 0; r < 16; r++) acc[a][c][r] = 0.f;
        if (ET == 0) {
            const float* A = reinterpret_cast<const float*>(in_lds);
            const float* B = reinterpret_cast<const float*>(w_lds);
            // a step = K values 2s, 2s + 1 (lane half lh takes 2s + lh); 22 is even: a step never straddles two filter rows
            const int a0 = 2 * pos_r[0] * (IC * 3) + 6 * pos_c[0] + lh;
            const int a1 = 2 * pos_r[1] * (IC * 3) + 6 * pos_c[1] + lh;
            // weights: [step][channel half][K parity][32 channels] -- the 64 lanes of a read touch 64 consecutive floats
            const int b0 = lh * 32 + li;
#pragma unroll
            for (int kh = 0; kh < 7; kh++) {
#pragma unroll
                for (int js = 0; js < KROW32 / 2; js++) {
                    const int step = kh * (KROW32 / 2) + js;
                    const float x0 = A[a0 + kh * (IC * 3) + 2 * js];
                    const float x1 = A[a1 + kh * (IC * 3) + 2 * js];
                    const float w0 = B[b0 + step * 128];
                    const float w1 = B[b0 + step * 128 + 64];
                    acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(x0, w0, acc[0][0], 0, 0, 0);
                    acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(x0, w1, acc[0][1], 0, 0, 0);
                    acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(x1, w0, acc[1][0], 0, 0, 0);
                    acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(x1, w1, acc[1][1], 0, 0, 0);
                }
            }
        } else {
            // a step = 16 K values = 4 pixels of one filter row (lane half lh: two of them, 16 bytes)
            const unsigned char* A = in_lds;
            const unsigned char* B = w_lds;
            const int a0 = (2 * pos_r[0] * IC + 2 * pos_c[0]) * 8 + lh * 16;
            const int a1 = (2 * pos_r[1] * IC + 2 * pos_c[1]) * 8 + lh * 16;
            const int b0 = li * (WROW16 * 2) + lh * 16;
#pragma unroll
            for (int kh = 0; kh < 7; kh++) {
#pragma unroll
                for (int st = 0; st < 2; st++) {
                    const uint4 x0 = *reinterpret_cast<const uint4*>(A + a0 + kh * (IC * 8) + st * 32);
                    const uint4 x1 = *reinterpret_cast<const uint4*>(A + a1 + kh * (IC * 8) + st * 32);
                    const uint4 w0 = *reinterpret_cast<const uint4*>(B + b0 + (kh * 2 + st) * 32);
                    const uint4 w1 = *reinterpret_cast<const uint4*>(B + b0 + 32 * (WROW16 * 2) + (kh * 2 + st) * 32);
                    if (ET == 1) {
                        acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, x0), __builtin_bit_cast(bf16x8, w0), acc[0][0], 0, 0, 0);
                        acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, x0), __builtin_bit_cast(bf16x8, w1), acc[0][1], 0, 0, 0);
                        acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, x1), __builtin_bit_cast(bf16x8, w0), acc[1][0], 0, 0, 0);
                        acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, x1), __builtin_bit_cast(bf16x8, w1), acc[1][1], 0, 0, 0);
                    } else {
                        acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, x0), __builtin_bit_cast(f16x8, w0), acc[0][0], 0, 0, 0);
                        acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, x0), __builtin_bit_cast(f16x8, w1), acc[0][1], 0, 0, 0);
                        acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, x1), __builtin_bit_cast(f16x8, w0), acc[1][0], 0, 0, 0);
                        acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, x1), __builtin_bit_cast(f16x8, w1), acc[1][1], 0, 0, 0);
                    }
                }
            }
        }

        // ---- epilogue, one 32-channel half at a time: BN + ReLU -> LDS [conv position][32 channels] -> max over the
        // valid conv positions -> store.  ReLU makes every value >= 0 and every pool window holds a valid position:
        // starting the max at 0 equals the reference's -inf padding.  All LDS addresses are a per-thread base +
        // compile-time offsets (the half-waves of an access hit the same banks, 2-way: a swizzled layout cost more address
        // arithmetic than the conflicts do); separable: the row maxima of the 15 conv rows (45 reads, all issued before the
        // first max -- with a branch per window position the reads ran one LDS latency after the other), then 3 rows per
        // pooled row.  Interior tiles (every conv position inside the map) skip the masks.
        const int gj = j0 + pj;
        const bool interior = cr0 >= 0 && cr0 + CR <= p.Ho && cc0 >= 0 && cc0 + CC <= p.Wo;
        const unsigned yo = ((unsigned)(n * p.PHo + i0) * (unsigned)p.PWo + (unsigned)gj) * 64u + (unsigned)li;     // (elements; the pooled map is < 2^32 elements: checked by the host)
#pragma unroll
        for (int tn = 0; tn < 2; tn++) {
            lds_barrier();          // the K loop's / the previous half's LDS reads are done
#pragma unroll
            for (int tm = 0; tm < 2; tm++) {
#pragma unroll
                for (int r = 0; r < 16; r++) {
                    float v = acc[tm][tn][r];
                    if (p.scale) v = v * bn_sc[tn];
                    ep_wr[(tm * 32 + (r & 3) + 8 * (r >> 2)) * 32] = fmaxf(v + bn_sh[tn], 0.f);
                }
            }
            lds_barrier();
            float h[CR];
            if (interior) {
#pragma unroll
                for (int r = 0; r < CR; r++) {
                    const float* e = ep_rd + r * CC * 32;
                    h[r] = fmaxf(fmaxf(e[0], e[32]), e[64]);
                }
            } else {
                float v[CR][3];
#pragma unroll
                for (int r = 0; r < CR; r++)
#pragma unroll
                    for (int dc = 0; dc < 3; dc++) v[r][dc] = ep_rd[(r * CC + dc) * 32];
#pragma unroll
                for (int r = 0; r < CR; r++) {
                    const bool rok = (unsigned)(cr0 + r) < (unsigned)p.Ho;      // (uniform)
                    float m = 0.f;
#pragma unroll
                    for (int dc = 0; dc < 3; dc++) {
                        const bool ok = rok && (unsigned)(cc0 + 2 * pj + dc) < (unsigned)p.Wo;
                        m = fmaxf(m, ok ? v[r][dc] : 0.f);
                    }
                    h[r] = m;
                }
            }
            if (gj < p.PWo) {
#pragma unroll
                for (int pi = 0; pi < PH; pi++) {
                    const float m = fmaxf(fmaxf(h[2 * pi], h[2 * pi + 1]), h[2 * pi + 2]);
                    if (i0 + pi < p.PHo) {
                        const unsigned o = yo + (unsigned)(pi * p.PWo) * 64u + tn * 32;
                        if (ET == 0) reinterpret_cast<float*>(p.y)[o] = m;
                        else reinterpret_cast<unsigned short*>(p.y)[o] = ET == 1 ? brcnn_f2b(m) : brcnn_f2h(m);
                    }
                }
            }
        }
        lds_barrier();      // the pool's reads are done before the next tile's pixels overwrite them
    }
}

template <int ET>
int launch_stem_pool(const StemPoolParams& p, hipStream_t s) {
    static bool attr_done = false;
    static int num_cus = 0;
    if (!attr_done) {
        BRCNN_HIP_CHECK(hipFuncSetAttribute((const void*)stem_pool_kernel<ET>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                            StemPoolLds<ET>::BYTES));
        int dev = 0;
        hipDeviceProp_t prop;
        BRCNN_HIP_CHECK(hipGetDevice(&dev));
        BRCNN_HIP_CHECK(hipGetDeviceProperties(&prop, dev));
        num_cus = prop.multiProcessorCount;
        attr_done = true;
    }
    // persistent workgroups, two per CU (LDS): weights staged once, the next tile's pixels in flight under the current one
    const int total = p.N * p.tiles_i * p.tiles_j;
    hipLaunchKernelGGL((stem_pool_kernel<ET>), dim3(total < 2 * num_cus ? total : 2 * num_cus), dim3(256), StemPoolLds<ET>::BYTES, s, p);
    BRCNN_LAUNCH_CHECK();
    return 0;
}

}  // namespace

BRCNN_API int brcnn_stem7x7s2_pool_nchw(const float* img, const void* w_packed, const float* scale, const float* shift,
                                        void* y, int batch, int height, int width, int cout, int dtype, void* stream) {
    if (!img || !w_packed || !y || batch <= 0 || height < 7 || width < 7 || cout != 64 ||
        (dtype != BRCNN_DT_F32 && dtype != BRCNN_DT_BF16 && dtype != BRCNN_DT_F16))
        return BRCNN_EINVAL;
    StemPoolParams p;
    p.img = img; p.w = w_packed; p.scale = scale; p.shift = shift; p.y = y;
    p.N = batch; p.H = height; p.W = width;
    p.Ho = (height + 6 - 7) / 2 + 1; p.Wo = (width + 6 - 7) / 2 + 1;
    p.PHo = (p.Ho + 2 - 3) / 2 + 1; p.PWo = (p.Wo + 2 - 3) / 2 + 1;
    p.tiles_i = (p.PHo + PH - 1) / PH; p.tiles_j = (p.PWo + PW - 1) / PW;
    if ((long long)batch * p.tiles_i * p.tiles_j >= 0x7fffffffLL || (long long)batch * p.PHo * p.PWo * 64 >= 0xffffffffLL) return BRCNN_EINVAL;
    hipStream_t s = (hipStream_t)stream;
    if (dtype == BRCNN_DT_F32) return launch_stem_pool<0>(p, s);
    if (dtype == BRCNN_DT_BF16) return launch_stem_pool<1>(p, s);
    return launch_stem_pool<2>(p, s);
}
