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
