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
typedef float f32x2_t __attribute__((ext_vector_type(2)));
typedef unsigned u32x2_t __attribute__((ext_vector_type(2)));
typedef unsigned u32x4_t __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8_t __attribute__((ext_vector_type(8)));

// ---- DVG_BF16X3: fp32 products on the bf16 matrix pipe ------------------------------------------------------------------
// gfx950's f32-input MFMA runs at 1/16 of the bf16 rate (157 vs 2516 TFLOP/s).  With DVG_BF16X3 = 1 the implicit-GEMM
// kernels (conv_igemm2.hip, wgrad.hip, wino_wgrad.hip) split every fp32 operand EXACTLY into three bf16 terms, a = h + m + l
// with 8 + 8 + 8 significant bits (h = a rounded to nearest bf16, m = the exact remainder a - h rounded to nearest bf16, l the
// rest: |m| <= 2^-8 |a|, |l| <= 2^-17 |a|, and l is a bf16 exactly), and form a K = 16 slab of the product as six
// v_mfma_f32_32x32x16_bf16 with fp32 accumulation - (l,h) (m,m) (h,l) (m,h) (h,m) (h,h); bf16 x bf16 products are exact in
// fp32 and the dropped terms (m,l) (l,m) (l,l) are below 2^-24 |a||b| with no preferred sign, i.e. below the rounding of ONE
// fp32 product (tests/test_bf16x3_split.py restates this in numpy) - instead of eight v_mfma_f32_32x32x2_f32: 192 instead of
// 512 matrix-pipe cycles per slab.  Measured error against fp64: equal to or below the f32 MFMA's on every layer shape
// (tools/diag_mfma_precision.py, tests/test_gpu_parity.py).  The split is done ONCE per element: activations when a stage's
// tile is written to LDS, weights when they are packed (the packed row of 16 k-values is 3 x 16 bf16 = 24 floats).
// DVG_BF16X3 = 0 builds the native f32-MFMA library (same ABI; `dvg_mfma_mode()` tells which one is loaded).
#ifndef DVG_BF16X3
#define DVG_BF16X3 1
#endif
#define DVG_WROW (DVG_BF16X3 ? 24 : 16)   // floats per packed weight row (16 k-values of one output channel)

// ---- timing experiments (WRONG results) ---------------------------------------------------------------------------------
// conv_igemm2.hip carries three knobs that change what the kernels COMPUTE, for pricing experiments only: DVG_ABLATE (parts
// of the stage loop removed), DVG_X3_TERMS (< 6: fewer of the six bf16 MFMAs per product slab), DVG_FIRST_SELECTS (diagnostic
// forms of the fused first layer).  They exist only in builds that say -DDVG_TIMING_EXPERIMENTS=1: `make all` / `make
// f32mfma` never pass it (the Makefile hands DEFS to the `variant` target alone), anything else fails to compile here, and
// dvg_build_info() carries the values so that a loaded library can be told from the product (bench.py refuses such a build).
#ifndef DVG_TIMING_EXPERIMENTS
#define DVG_TIMING_EXPERIMENTS 0
#endif
#ifndef DVG_ABLATE
#define DVG_ABLATE 0
#endif
#ifndef DVG_X3_TERMS
#define DVG_X3_TERMS 6
#endif
#ifndef DVG_FIRST_SELECTS
#define DVG_FIRST_SELECTS 0
#endif
#if !DVG_TIMING_EXPERIMENTS && (DVG_ABLATE != 0 || DVG_X3_TERMS != 6 || DVG_FIRST_SELECTS != 0)
#error "DVG_ABLATE / DVG_X3_TERMS / DVG_FIRST_SELECTS give WRONG results: timing builds only (-DDVG_TIMING_EXPERIMENTS=1 via `make variant`)"
#endif

// two fp32 values -> their three bf16 terms, each pair packed into one dword (low half = the first value).  Nine VALU
// instructions: v_cvt_pk_bf16_f32 (round to nearest even) x 3, the two halves of a packed pair back to fp32 (shift / mask) x 2,
// v_pk_add_f32 x 2.  (A truncating split - mask instead of convert - costs the same and leaves dropped terms of up to
// 2^-20 |a||b|, all with the product's sign.)
typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));
__device__ __forceinline__ unsigned bf16_pack_rn(float a0, float a1) {
    const f32x2_t v = {a0, a1};
    return __builtin_bit_cast(unsigned, __builtin_convertvector(v, bf16x2_t));
}
__device__ __forceinline__ void bf16x3_split_pair(float a0, float a1, unsigned& ph, unsigned& pm, unsigned& pl) {
    const unsigned M = 0xffff0000u;
    const f32x2_t a = {a0, a1};
    ph = bf16_pack_rn(a0, a1);
    const f32x2_t h = {__uint_as_float(ph << 16), __uint_as_float(ph & M)};
    const f32x2_t r = a - h;
    pm = bf16_pack_rn(r[0], r[1]);
    const f32x2_t m = {__uint_as_float(pm << 16), __uint_as_float(pm & M)};
    const f32x2_t t = r - m;
    pl = bf16_pack_rn(t[0], t[1]);
}

// Packed weight rows.  A row = the 16 k-values (one 16-channel chunk) of ONE output channel; rows are stored in blocks of
// 64 output channels, [..][64][DVG_WROW], so that the weight tile of a workgroup's stage is one contiguous run.  With
// DVG_BF16X3 a row is [3 planes h, m, l][16 bf16], and the two 16-byte halves of every plane are swapped for rows with
// (co % 64) & 8 (the LDS image of the tile is this memory image: b128 fragment reads of 16 consecutive rows then hit 16
// distinct 16-byte slots).
#ifndef DVG_WROW_SHORT_STORES
#define DVG_WROW_SHORT_STORES 0
#endif
// Two adjacent k-values (k even) of a packed row as ONE 4-byte store per plane.
__device__ __forceinline__ void wrow_store_pair(float* __restrict__ rows, size_t row, int co_local, int k, float v0, float v1) {
#if DVG_BF16X3
    unsigned* d = reinterpret_cast<unsigned*>(rows + row * 24);
    unsigned ph, pm, pl;
    bf16x3_split_pair(v0, v1, ph, pm, pl);
    const int pos2 = ((((k >> 3) ^ ((co_local >> 3) & 1)) << 3) + (k & 7)) >> 1;
#if DVG_WROW_SHORT_STORES   // A/B knob (make variant): the same values as 2 x 2-byte stores per plane - which half of the fix matters
    unsigned short* d16 = reinterpret_cast<unsigned short*>(d);
    d16[2 * pos2] = (unsigned short)(ph & 0xffffu);
    d16[2 * pos2 + 1] = (unsigned short)(ph >> 16);
    d16[16 + 2 * pos2] = (unsigned short)(pm & 0xffffu);
    d16[16 + 2 * pos2 + 1] = (unsigned short)(pm >> 16);
    d16[32 + 2 * pos2] = (unsigned short)(pl & 0xffffu);
    d16[32 + 2 * pos2 + 1] = (unsigned short)(pl >> 16);
#else
    d[pos2] = ph;
    d[8 + pos2] = pm;
    d[16 + pos2] = pl;
#endif
#else
    (void)co_local;
    rows[row * 16 + k] = v0;
    rows[row * 16 + k + 1] = v1;
#endif
}

// The last statement of every kernel that writes packed rows: the stores above performed before the wave ends (tried against the
// zero rows of winograd.hip's wrow_owner_note when they still looked like lost stores; harmless, kept - once per weight version).
__device__ __forceinline__ void wrow_drain() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }

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
