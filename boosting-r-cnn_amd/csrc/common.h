// Shared helpers for the gfx950 kernels of libbrcnn_hip.so.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "../../include/brcnn_hip.h"

#define BRCNN_EINVAL (-22)
#define BRCNN_API extern "C" __attribute__((visibility("default")))

#define BRCNN_HIP_CHECK(expr)                                   \
    do {                                                        \
        hipError_t _e = (expr);                                 \
        if (_e != hipSuccess) return -(1000 + (int)_e);         \
    } while (0)

#define BRCNN_LAUNCH_CHECK()                                    \
    do {                                                        \
        hipError_t _e = hipGetLastError();                      \
        if (_e != hipSuccess) return -(1000 + (int)_e);         \
    } while (0)

static inline int brcnn_cdiv(int64_t a, int64_t b) { return (int)((a + b - 1) / b); }

// wave width on gfx950
#define WAVE 64

// 16-bit element types of the NHWC activations in the reduced-precision modes: bf16 (bit pattern in an
// unsigned short) and IEEE fp16 (wrapped so that overloads can tell the two apart)
typedef unsigned short bf16_t;
struct f16_t { unsigned short v; };

static inline bool brcnn_elem_ok(int dt) { return dt == BRCNN_DT_F32 || dt == BRCNN_DT_BF16 || dt == BRCNN_DT_F16; }
static inline bool brcnn_is16(int dt) {
    return dt == BRCNN_DT_BF16 || dt == BRCNN_DT_BF16_OUT_F32 || dt == BRCNN_DT_F16 || dt == BRCNN_DT_F16_OUT_F32;
}
static inline bool brcnn_isf16(int dt) { return dt == BRCNN_DT_F16 || dt == BRCNN_DT_F16_OUT_F32; }
static inline bool brcnn_out_f32(int dt) { return dt == BRCNN_DT_BF16_OUT_F32 || dt == BRCNN_DT_F16_OUT_F32; }

#if defined(__HIPCC__)
__device__ __forceinline__ float brcnn_h2f(unsigned short h) { return (float)__builtin_bit_cast(_Float16, h); }
__device__ __forceinline__ unsigned short brcnn_f2h(float f) { return __builtin_bit_cast(unsigned short, (_Float16)f); }
__device__ __forceinline__ float brcnn_b2f(unsigned short h) { return __uint_as_float(((unsigned)h) << 16); }
// float -> 16 bit, round to nearest even: gfx950 has the conversions in hardware, two values per instruction
// (v_cvt_pk_bf16_f32 / v_cvt_pk_f16_f32) -- one VALU operation per pair instead of the five of the integer form
typedef __bf16 brcnn_bf16x2 __attribute__((ext_vector_type(2)));
typedef _Float16 brcnn_f16x2 __attribute__((ext_vector_type(2)));
typedef float brcnn_f32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ unsigned short brcnn_f2b(float v) { return __builtin_bit_cast(unsigned short, (__bf16)v); }
// the pair (lo, hi) as one 32-bit word, lo in bits 0..15
__device__ __forceinline__ unsigned brcnn_pk2b(float lo, float hi) {
    const brcnn_f32x2 v = {lo, hi};
    return __builtin_bit_cast(unsigned, __builtin_convertvector(v, brcnn_bf16x2));
}
__device__ __forceinline__ unsigned brcnn_pk2h(float lo, float hi) {
    const brcnn_f32x2 v = {lo, hi};
    return __builtin_bit_cast(unsigned, __builtin_convertvector(v, brcnn_f16x2));
}
#endif
