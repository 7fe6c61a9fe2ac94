// Shared helpers for the gfx950 kernels of libcoponerf_hip.so.
#pragma once
#include <hip/hip_runtime.h>
#include <hip/hip_fp16.h>
#include <stdint.h>
#include <stdio.h>
#include "../../include/coponerf_hip.h"

void cpn_set_error(const char* fmt, ...);
// CUs a launch on `stream` may occupy: the device's count, or the share of a CU-masked stream (streams.cpp)
int cpn_stream_cus(void* stream);

#define CPN_REQUIRE(cond, code, ...)                 \
    do {                                             \
        if (!(cond)) {                               \
            cpn_set_error(__VA_ARGS__);              \
            return (code);                           \
        }                                            \
    } while (0)

#define CPN_LAUNCH_CHECK(name)                                                   \
    do {                                                                         \
        hipError_t e_ = hipGetLastError();                                       \
        if (e_ != hipSuccess) {                                                  \
            cpn_set_error("%s: launch failed: %s", name, hipGetErrorString(e_)); \
            return (int)e_;                                                      \
        }                                                                        \
    } while (0)

static inline unsigned cpn_cdiv(long long a, long long b) { return (unsigned)((a + b - 1) / b); }

typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef _Float16 half4 __attribute__((ext_vector_type(4)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
