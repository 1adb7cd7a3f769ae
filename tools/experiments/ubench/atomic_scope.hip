// Micro-benchmark: fp32 atomic-add tails of the weight-gradient kernels.
//  (a) every workgroup adds a 128x128 fp32 tile into ONE shared buffer (today's wgrad tail)
//  (b) the same, into a buffer private to the workgroup's XCD (HW_REG_XCC_ID): is the atomic then served by the
//      XCD's own L2, and how fast?
//  (c) plain 16-byte stores of the tile into a private slab (the workspace alternative)
// Build: hipcc --offload-arch=gfx950 -O3 -munsafe-fp-atomics atomic_scope.hip -o atomic_scope
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("err %s line %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

__device__ __forceinline__ int xcc_id() {
    int v;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(v));
    return v & 15;
}

template <int MODE, int SCOPE>
__global__ __launch_bounds__(256) void tail(float* __restrict__ out, int tiles, size_t tile_elems, size_t buf_elems) {
    const int t = blockIdx.x % tiles;
    float* base = out + (size_t)t * tile_elems;
    if (MODE == 1) base += (size_t)xcc_id() * buf_elems;
    if (MODE == 2) base = out + (size_t)blockIdx.x * tile_elems;
    const float v = 1.0f;
    if (MODE == 2) {
        for (size_t i = threadIdx.x * 4; i < tile_elems; i += 1024)
            *reinterpret_cast<float4*>(base + i) = make_float4(v, v, v, v);
    } else {
        // the accumulator layout of the MFMA epilogue: a lane owns one column, 4-row groups -> scalar atomics,
        // 32 consecutive floats per half-wave
        for (size_t i = threadIdx.x; i < tile_elems; i += 256) {
            if (SCOPE == 0) atomicAdd(base + i, v);
            else __hip_atomic_fetch_add(base + i, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        }
    }
}

int main() {
    const int tiles = 36, wgs = 504;          // stage-3 3x3 layer: 36 tiles x 14 slices
    for (int tdim : {128, 256}) {
        const size_t te = (size_t)tdim * tdim;
        const size_t be = te * tiles;
        float* buf;
        CK(hipMalloc(&buf, sizeof(float) * te * 1024));
        hipEvent_t a, b;
        CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
        auto run = [&](auto kern, const char* name, int n_wgs) {
            float best = 1e9;
            for (int it = 0; it < 6; it++) {
                CK(hipMemset(buf, 0, sizeof(float) * te * 1024));
                CK(hipDeviceSynchronize());
                CK(hipEventRecord(a));
                hipLaunchKernelGGL(kern, dim3(n_wgs), dim3(256), 0, 0, buf, tiles, te, be);
                CK(hipEventRecord(b));
                CK(hipEventSynchronize(b));
                float ms; CK(hipEventElapsedTime(&ms, a, b));
                if (ms < best) best = ms;
            }
            printf("tile %d  %-34s wgs %4d  %.1f us  %.2f TB/s (tile bytes x wgs)\n", tdim, name, n_wgs, best * 1e3,
                   (double)n_wgs * te * 4 / best / 1e9);
            return 0;
        };
        const int n = tdim == 128 ? wgs : 252;
        run(tail<0, 0>, "shared buffer, agent atomics", n);
        run(tail<0, 1>, "shared buffer, workgroup atomics", n);
        run(tail<1, 0>, "per-XCD buffer, agent atomics", n);
        run(tail<1, 1>, "per-XCD buffer, workgroup atomics", n);
        run(tail<2, 0>, "private slab, plain stores", n);
        // check (b): sum over the 8 buffers == wgs per element
        CK(hipMemset(buf, 0, sizeof(float) * te * 1024));
        hipLaunchKernelGGL((tail<1, 1>), dim3(n), dim3(256), 0, 0, buf, tiles, te, be);
        CK(hipDeviceSynchronize());
        std::vector<float> h(be * 8);
        CK(hipMemcpy(h.data(), buf, sizeof(float) * be * 8, hipMemcpyDeviceToHost));
        double tot = 0; int used = 0;
        for (int x = 0; x < 8; x++) { double s = 0; for (size_t i = 0; i < be; i++) s += h[x * be + i]; tot += s; used += s > 0; }
        printf("tile %d  per-XCD workgroup atomics: total %.0f expected %.0f, buffers used %d\n", tdim, tot, (double)n * te, used);
        CK(hipFree(buf));
    }
    return 0;
}
