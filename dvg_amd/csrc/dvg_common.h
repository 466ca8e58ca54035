// Internal helpers shared by the kernel translation units of libdvg_hip.so.
#pragma once
#include <hip/hip_runtime.h>
#include <cstdarg>
#include <cstdio>
#include "../../include/dvg_hip.h"

namespace dvg {

// thread-local error string behind dvg_last_error()
char* err_buf();
int fail(int code, const char* fmt, ...);

inline bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15u) == 0; }

// Checks the launch that was just issued; no sync.
inline int check_launch(const char* what) {
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return fail(DVG_ERR_HIP, "%s: %s", what, hipGetErrorString(e));
    return DVG_OK;
}

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

__device__ __forceinline__ float apply_act(float v, int act, float slope) {
    switch (act) {
        case DVG_ACT_LRELU: return v > 0.f ? v : v * slope;
        case DVG_ACT_TANH: return tanhf(v);
        case DVG_ACT_SIGMOID: return 1.f / (1.f + expf(-v));
        default: return v;
    }
}

// Bijective XCD-aware remap of a linear workgroup id (guide §5 T1): workgroups
// that land on one XCD (id % 8 equal) get a contiguous range of logical tiles.
__device__ __forceinline__ unsigned xcd_remap(unsigned bid, unsigned nwg) {
    const unsigned q = nwg >> 3, r = nwg & 7u, xcd = bid & 7u;
    const unsigned base = xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
    return base + (bid >> 3);
}

}  // namespace dvg

#define DVG_REQUIRE(cond, code, ...) \
    do {                             \
        if (!(cond)) return dvg::fail(code, __VA_ARGS__); \
    } while (0)
