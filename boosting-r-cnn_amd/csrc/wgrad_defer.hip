// Deferred second stage of the sliced 16-bit weight gradients.
//
// A sliced weight-gradient launch (conv_wgrad_bf16.hip, conv_wgrad_pp_bf16.hip) leaves one fp32 slab per M slice and a
// second launch adds the slabs of a tile in slice order into dW.  A bf16 train step of bench.py holds 61 such layers and
// 85 second-stage launches of 6-25 us each (two passes where a tile has more than 24 slices): every one of them costs
// the host a launch, and on a host that cannot keep up the step is as long as its launches (profiles/r05_notes.md).
//
// With deferral on for a stream (brcnn_wgrad_defer_begin) the producing launch takes its slabs from a caller-owned arena
// instead of the stream's 160 MiB scratch, and only an ITEM is recorded: (slabs, dW, tile geometry, slice grouping).
// brcnn_wgrad_defer_flush reduces every pending item in ONE table-driven launch -- the same additions in the same
// order as the per-layer second stage (two-level grouping included), so dW is bit for bit what the immediate form
// writes.  The arena is a bump allocator: when it (or the item table) is full the next producing launch flushes first;
// stream order makes the reuse safe.  dW of a deferred layer is complete only after the flush: whoever reads it earlier
// (a consumer on the same stream, a gradient reducer) asks for the flush first (autograd.py does).
// The per-stream state is guarded by a mutex, but the slab request and the item of ONE producing launch are two calls:
// launches and flushes for a given stream must come from one thread at a time (as any launch sequence on a stream does).
#include <hip/hip_runtime.h>

#include <mutex>
#include <vector>

#include "conv_common.h"

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int MAX_ITEMS = 64;

struct Item {                       // 48 bytes
    float* slab;
    float* dw;                      // (Cout, K) fp32, accumulated into
    int block0;                     // first workgroup of the item in the batched launch
    int tiles, tiles_k, slices;
    int group;                      // slices per first-level group (>= slices: one level)
    int Cout, K;
    int kind;                       // 0: eight-phase 256 x 256 register order; else (WT << 4) | WG of conv_wgrad_bf16_kernel
};

struct Table {                      // passed by value (3 KB of kernel arguments)
    int n, blocks;
    Item it[MAX_ITEMS];
};

// One workgroup = 256 consecutive float4 positions of one tile's register-order slab; a thread adds its position over
// the slices: first level groups of `group` consecutive slices, each summed in slice order starting from its first slab,
// then the group sums in group order -- exactly wgrad_pp_reduce_kernel<false> + <true> / wgrad_reduce_kernel<.., false> +
// <.., true>.  Eight loads in flight per thread; the additions stay in order.
__global__ __launch_bounds__(256) void wgrad_defer_reduce_kernel(const Table tb) {
    int i = 0;
    const int b = blockIdx.x;
    while (i + 1 < tb.n && b >= tb.it[i + 1].block0) i++;
    const Item& it = tb.it[i];
    const int kind = it.kind;
    const int WT = kind ? (kind >> 4) : 0, WG = kind ? (kind & 15) : 0;
    const int TILE = kind ? 32 * WT * WG : 256;
    const int tile_f4 = TILE * TILE / 4;
    const int bpt = tile_f4 / 256;                  // workgroups per tile
    const int lb = b - it.block0;
    const int tile = lb / bpt;
    const int e = (lb - tile * bpt) * 256 + threadIdx.x;
    const size_t step = (size_t)it.tiles * tile_f4;
    const f32x4* src = reinterpret_cast<const f32x4*>(it.slab) + (size_t)tile * tile_f4 + e;
    f32x4 total = {0.f, 0.f, 0.f, 0.f};
    for (int g0 = 0; g0 < it.slices; g0 += it.group) {
        int count = it.slices - g0;
        if (count > it.group) count = it.group;
        const f32x4* s0 = src + (size_t)g0 * step;
        f32x4 s = s0[0];
        int j = 1;
        for (; j + 7 < count; j += 8) {
            f32x4 t[8];
#pragma unroll
            for (int u = 0; u < 8; u++) t[u] = s0[(size_t)(j + u) * step];
#pragma unroll
            for (int u = 0; u < 8; u++) { s.x += t[u].x; s.y += t[u].y; s.z += t[u].z; s.w += t[u].w; }
        }
        for (; j < count; j++) {
            const f32x4 t = s0[(size_t)j * step];
            s.x += t.x; s.y += t.y; s.z += t.z; s.w += t.w;
        }
        if (g0 == 0) total = s;
        else { total.x += s.x; total.y += s.y; total.z += s.z; total.w += s.w; }
    }
    // position in the tile -> (co, k) of the four values (the producing kernel's accumulator layout)
    const int tk = tile % it.tiles_k, tco = tile / it.tiles_k;
    int kk, co;
    if (kind == 0) {
        const int v = e >> 9, t512 = e & 511;
        const int a = v >> 3, c = (v >> 2) & 1, g = v & 3;
        const int wave = t512 >> 6, lane = t512 & 63;
        const int wm = wave >> 2, wn = wave & 3;
        kk = tk * 256 + c * 128 + (wn >> 1) * 64 + (wn & 1) * 32 + (lane & 31);
        co = tco * 256 + (a >> 1) * 128 + wm * 64 + (a & 1) * 32 + 8 * g + 4 * (lane >> 5);
    } else {
        const int per = WG * WG * 64;
        const int v = e / per, tid = e - v * per;
        const int a = v / (WT * 4), c = (v >> 2) % WT, g4 = v & 3;
        const int wave = tid >> 6, lane = tid & 63;
        const int wm = wave / WG, wn = wave - wm * WG;
        kk = tk * TILE + (wn * WT + c) * 32 + (lane & 31);
        co = tco * TILE + (wm * WT + a) * 32 + 8 * g4 + 4 * (lane >> 5);
    }
    if (kk >= it.K) return;
    const float sv[4] = {total.x, total.y, total.z, total.w};
#pragma unroll
    for (int j = 0; j < 4; j++)
        if (co + j < it.Cout) it.dw[(size_t)(co + j) * it.K + kk] += sv[j];
}

struct DeferStream {
    hipStream_t stream;
    int device;
    char* arena;
    size_t bytes, off;
    int max_items;
    Table tb;
    long long flushes, items_total;       // statistics (brcnn_wgrad_defer_stats)
};

std::mutex g_mutex;
std::vector<DeferStream> g_streams;

DeferStream* find(hipStream_t s) {
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) return nullptr;
    for (auto& e : g_streams)
        if (e.stream == s && e.device == dev) return &e;
    return nullptr;
}

int flush_locked(DeferStream& d) {
    const int n = d.tb.n;
    if (n > 0) {
        hipLaunchKernelGGL(wgrad_defer_reduce_kernel, dim3(d.tb.blocks), dim3(256), 0, d.stream, d.tb);
        d.flushes++;
        d.items_total += n;
    }
    d.tb.n = 0;
    d.tb.blocks = 0;
    d.off = 0;
    if (n > 0) BRCNN_LAUNCH_CHECK();
    return n;
}

}  // namespace

namespace brcnn_conv {

// slabs for a producing launch on `s`, or nullptr: deferral is off for the stream / the request does not fit the arena
// at all.  May launch the batched reduction first (arena or table full).  *err < 0: that launch failed.
float* wgrad_defer_slabs(hipStream_t s, size_t bytes, int* err) {
    *err = 0;
    std::lock_guard<std::mutex> lock(g_mutex);
    DeferStream* d = find(s);
    if (!d || !d->arena) return nullptr;
    bytes = (bytes + 255) & ~(size_t)255;
    if (bytes > d->bytes) return nullptr;
    if (d->off + bytes > d->bytes || d->tb.n >= d->max_items) {
        const int rc = flush_locked(*d);
        if (rc < 0) { *err = rc; return nullptr; }
    }
    float* p = reinterpret_cast<float*>(d->arena + d->off);
    d->off += bytes;
    return p;
}

// the producing launch on `s` has been issued: its second stage joins the stream's pending items
void wgrad_defer_push(hipStream_t s, float* slab, float* dw, int tiles, int tiles_k, int slices, int group, int Cout, int K,
                      int kind) {
    std::lock_guard<std::mutex> lock(g_mutex);
    DeferStream* d = find(s);
    if (!d) return;
    Item& it = d->tb.it[d->tb.n++];
    const int tile = kind ? 32 * (kind >> 4) * (kind & 15) : 256;
    it.slab = slab; it.dw = dw; it.block0 = d->tb.blocks;
    it.tiles = tiles; it.tiles_k = tiles_k; it.slices = slices;
    it.group = group < 1 ? slices : group;
    it.Cout = Cout; it.K = K; it.kind = kind;
    d->tb.blocks += tiles * (tile * tile / 1024);
}

}  // namespace brcnn_conv

BRCNN_API int brcnn_wgrad_defer_begin(void* stream, void* arena, size_t bytes, int max_items) {
    hipStream_t s = (hipStream_t)stream;
    std::lock_guard<std::mutex> lock(g_mutex);
    DeferStream* d = find(s);
    if (d && d->tb.n) {
        const int rc = flush_locked(*d);
        if (rc < 0) return rc;
    }
    if (!arena || bytes < ((size_t)1 << 20)) {          // off
        if (d) { d->arena = nullptr; d->bytes = 0; }
        return 0;
    }
    if (reinterpret_cast<size_t>(arena) & 255) return BRCNN_EINVAL;
    if (!d) {
        int dev = 0;
        BRCNN_HIP_CHECK(hipGetDevice(&dev));
        if (g_streams.capacity() < 64) g_streams.reserve(64);
        if (g_streams.size() >= 64) return BRCNN_EINVAL;
        DeferStream e = {};
        e.stream = s;
        e.device = dev;
        g_streams.push_back(e);
        d = &g_streams.back();
    }
    d->arena = (char*)arena;
    d->bytes = bytes;
    d->off = 0;
    d->max_items = max_items < 1 ? MAX_ITEMS : (max_items > MAX_ITEMS ? MAX_ITEMS : max_items);
    return 0;
}

BRCNN_API int brcnn_wgrad_defer_flush(void* stream) {
    std::lock_guard<std::mutex> lock(g_mutex);
    DeferStream* d = find((hipStream_t)stream);
    if (!d) return 0;
    return flush_locked(*d);
}

BRCNN_API int brcnn_wgrad_defer_pending(void* stream) {
    std::lock_guard<std::mutex> lock(g_mutex);
    DeferStream* d = find((hipStream_t)stream);
    return d ? d->tb.n : 0;
}

BRCNN_API int brcnn_wgrad_defer_stats(void* stream, long long* flushes, long long* items) {
    std::lock_guard<std::mutex> lock(g_mutex);
    DeferStream* d = find((hipStream_t)stream);
    if (flushes) *flushes = d ? d->flushes : 0;
    if (items) *items = d ? d->items_total : 0;
    return 0;
}
