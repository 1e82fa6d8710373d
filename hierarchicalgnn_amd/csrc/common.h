// Shared host-side helpers for libhgnn_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <cstdarg>
#include <cstdio>
#include <cstdint>
#include "../../include/hgnn_hip.h"

namespace hgnn {

void set_error(const char* fmt, ...);

#define HGNN_CHECK_HIP(expr)                                                          \
    do {                                                                              \
        hipError_t e__ = (expr);                                                      \
        if (e__ != hipSuccess) {                                                      \
            ::hgnn::set_error("%s:%d: %s failed: %s", __FILE__, __LINE__, #expr,      \
                              hipGetErrorString(e__));                                \
            return HGNN_ERR_HIP;                                                      \
        }                                                                             \
    } while (0)

#define HGNN_REQUIRE(cond, ...)                                                       \
    do {                                                                              \
        if (!(cond)) {                                                                \
            ::hgnn::set_error(__VA_ARGS__);                                           \
            return HGNN_ERR_INVALID_ARG;                                              \
        }                                                                             \
    } while (0)

static inline int64_t ceil_div(int64_t a, int64_t b) { return (a + b - 1) / b; }
static inline size_t align_up(size_t x, size_t a) { return (x + a - 1) / a * a; }

constexpr int kWave = 64;          // CDNA wavefront
constexpr int kBlock = 256;        // 4 waves per workgroup
constexpr int kWavesPerBlock = kBlock / kWave;

}  // namespace hgnn
