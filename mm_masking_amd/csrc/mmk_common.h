// Shared host/device helpers for libmmk_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "mmk.h"

namespace mmk {

void set_error(const char *fmt, ...);

inline size_t align_up(size_t x, size_t a) { return (x + a - 1) / a * a; }

// Bump allocator over the caller's workspace.
struct Arena {
    char *base;
    size_t cap;
    size_t off = 0;
    Arena(void *p, size_t n) : base(static_cast<char *>(p)), cap(n) {}
    template <typename T>
    T *take(size_t count)
    {
        off = align_up(off, 256);
        T *p = reinterpret_cast<T *>(base + off);
        off += count * sizeof(T);
        return p;
    }
    bool ok() const { return off <= cap && (base != nullptr || off == 0); }
};

}  // namespace mmk

#define MMK_CHECK_HIP(expr)                                                              \
    do {                                                                                 \
        hipError_t e_ = (expr);                                                          \
        if (e_ != hipSuccess) {                                                          \
            mmk::set_error("%s:%d %s -> %s", __FILE__, __LINE__, #expr, hipGetErrorString(e_)); \
            return MMK_ERR_HIP;                                                          \
        }                                                                                \
    } while (0)

#define MMK_REQUIRE(cond, ...)            \
    do {                                  \
        if (!(cond)) {                    \
            mmk::set_error(__VA_ARGS__);  \
            return MMK_ERR_ARG;           \
        }                                 \
    } while (0)

#define MMK_LAUNCH_CHECK() MMK_CHECK_HIP(hipGetLastError())

// 64-lane wavefront reductions (gfx950: wave64).
__device__ __forceinline__ double wave_sum(double v)
{
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
    return v;
}

__device__ __forceinline__ int wave_sum_i(int v)
{
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
    return v;
}
