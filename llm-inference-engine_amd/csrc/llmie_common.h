// Host+device common definitions for the gfx950 kernels behind include/llmie.h.
#pragma once
#include <hip/hip_runtime.h>
#include <hip/hip_fp16.h>
#include <cstdarg>
#include <cstdint>
#include <cstdio>

#include "../../include/llmie.h"

namespace llmie {

void set_error(const char *fmt, ...);

inline hipStream_t as_stream(llmie_stream s) { return reinterpret_cast<hipStream_t>(s); }

// Checks hipGetLastError after a launch; no synchronisation (graph-capture safe).
inline int launch_status(const char *what) {
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) {
        set_error("%s: launch failed: %s", what, hipGetErrorString(e));
        return LLMIE_ERR_LAUNCH;
    }
    return LLMIE_OK;
}

#define LLMIE_REQUIRE(cond, ...)                       \
    do {                                               \
        if (!(cond)) {                                 \
            ::llmie::set_error(__VA_ARGS__);           \
            return LLMIE_ERR_INVALID_ARG;              \
        }                                              \
    } while (0)

#define LLMIE_UNSUPPORTED(...)                         \
    do {                                               \
        ::llmie::set_error(__VA_ARGS__);               \
        return LLMIE_ERR_UNSUPPORTED;                  \
    } while (0)

constexpr int kWave = 64;  // gfx950 wavefront

}  // namespace llmie
