// Winograd F(2x2, 3x3) for the deep 3x3 layers of the eval-mode rollout (vgg_64.py:5-15 at 16x16 and 8x8, 256-512
// channels: 60 % of the rollout's conv FLOPs).  fp32 data and transforms (the batched GEMM's products as in conv_igemm2.hip), 2.25x fewer multiply-adds than the direct form:
//
//   Y = A^T [ (G g G^T) .* (B^T d B) ] A      per 2x2 output tile, d = the 4x4 input patch around it (pad 1)
//
//   U[xi][ci][co] = (G g G^T)[xi]             dvg_winograd_weight   once per weight version, written in the k16 layout
//   V[xi][t][ci]  = (B^T d B)[xi]             dvg_winograd_input    one HBM pass: reads x (L2 absorbs the 4x tile overlap), writes 4x its size
//   M[xi][t][co]  = sum_ci V U                dvg_gemm_batched_k16  16 GEMMs on the igemm machinery (conv_igemm2.hip, M2_GEMM)
//   y             = act(scale * A^T M A + shift) (+ 2x2 max-pool: a Winograd tile IS a pool window)   dvg_winograd_output
//
// xi = 4*a + b over the 4x4 transform positions; t = (n * H/2 + ty) * W/2 + tx over the output tiles.
// The transforms are HBM-bound elementwise passes (thread = one tile x 4 channels, float4 everywhere).
#include <cstdlib>

#include "dvg_common.h"

// Nontemporal hints on the transforms' read-once M loads / written-once V stores.  r05 same-box A/B on the vgg_64 rollout, each
// twice: M loads of the chain kernels nontemporal 49.41 / 49.53 k frames/s in flight against 48.87 / 49.23 (+0.9 %), one chain
// 15.28 against 15.34 ms -> default; V stores nontemporal 49.32 / 49.24 (noise) -> not taken; the GEMM's M store nontemporal
// 48.86 / 49.07 (-0.3 %) -> not taken; the same load hint in the 8 x 8 hand-over and the plain output kernels 49.11 / 49.01
// against 49.13 / 48.94 (noise) -> not taken.
#ifndef DVG_WINO_NT_LOAD
#define DVG_WINO_NT_LOAD 1
#endif
#ifndef DVG_WINO_NT_STORE
#define DVG_WINO_NT_STORE 0
#endif

namespace dvg {

template <typename V> __device__ __forceinline__ V wino_ld(const V* p) {
#if DVG_WINO_NT_LOAD
    return __builtin_nontemporal_load(p);
#else
    return *p;
#endif
}
template <typename V> __device__ __forceinline__ void wino_st(V* p, const V& v) {
#if DVG_WINO_NT_STORE
    __builtin_nontemporal_store(v, p);
#else
    *p = v;
#endif
}

// wrow_owner_note - why the two weight transforms have the shape they have (r06, profiles/r06_dp_race_bisect.txt).
// Until r06 the F(4x4) kernel ran one k-value per thread (thread i = (co, ci), 2-byte stores).  Alone on the device it was right in
// every run ever compared.  With ANOTHER PROCESS busy on the same device (the one-GPU data-parallel rehearsal: two trainers) 1-30 %
// of its launches produced a few rows of U that were exactly ZERO: always rows of transform position (5, 5), always the 16 lanes of
// one VALU pass (tools/diag_pack_repeat.py, tools/diag_old_weight_kernel.py; a data gradient convolved with such a U is off by
// 1e-2 ... 4e-1 - the rehearsal's run-to-run different vgg_64 gradients).  What the stand-alone copy of that kernel established
// (tools/ubench/old_weight_kernel.hip, poisoned output buffers, another trainer beside it):
//   * the zeros are WRITTEN, and the fp32 value the thread computed for position (5, 5) is already 0.0 while the g[8] it read is
//     right: a wrong VALU result, not a lost or misdirected store (so neither one workgroup per cache line, nor s_waitcnt vmcnt(0)
//     after every store / before s_endpgm, nor a restored EXEC mask, nor 32-bit index arithmetic changed anything);
//   * the waves concerned were on the chip no longer than their peers (no context switch);
//   * position (5, 5) is where G's last row {0, 0, 1} meets itself: value = 0 * t50 + 0 * t51 + 1 * t52, which the compiler
//     emits as packed FMAs with an inline-constant 0 multiplier (v_pk_fma_f32 ..., 0, ... op_sel_hi; v_pk_add_f32 with op_sel).  With
//     that row handed in as a kernel ARGUMENT (zeros unknown at compile time): 0 of 720 against 98 of 720 in the same run.
// Why a packed FMA chain yields 0 for one pass only when another process loads the chip was not established.  The form below (two
// adjacent filters per thread, so the packed operations carry two real values; 4-byte stores) never failed: 0 of 17 100
// recomputations under the same contention, 0 of 39 + 20 + 3 training runs that differed in 25-60 % before; the F(2x2) kernel
// (hand-written transform, no literal zeros) and pack_k16_kernel never failed in any form.  What guards it is the test, not a
// theory: tests/test_gpu_multirank.py::test_packed_weights_with_a_second_process_on_the_device.
__global__ void winograd_weight_kernel(const float* __restrict__ w, float* __restrict__ u, int cout, int cin) {
    // thread = (chunk, co, pair of k); writes the 16 transform positions of two adjacent filters (wrow_owner_note)
    const unsigned total = (unsigned)cout * (unsigned)cin / 2;
    for (unsigned i = blockIdx.x * blockDim.x + threadIdx.x; i < total; i += gridDim.x * blockDim.x) {
        const unsigned rr = i >> 3;
        const int co = (int)(rr % (unsigned)cout), chunk = (int)(rr / (unsigned)cout), k = (int)(i & 7) * 2;
        const float* g = w + ((size_t)co * cin + chunk * 16 + k) * 9;       // the two filters are adjacent: 18 floats
        float t[2][4][3];   // G g
#pragma unroll
        for (int e = 0; e < 2; ++e)
#pragma unroll
            for (int s = 0; s < 3; ++s) {
                const float g0 = g[9 * e + s], g1 = g[9 * e + 3 + s], g2 = g[9 * e + 6 + s];
                t[e][0][s] = g0;
                t[e][1][s] = 0.5f * (g0 + g1 + g2);
                t[e][2][s] = 0.5f * (g0 - g1 + g2);
                t[e][3][s] = g2;
            }
        const unsigned row0 = ((unsigned)(co >> 6) * (unsigned)(cin / 16) + (unsigned)chunk) * 64u + (unsigned)(co & 63);
        const unsigned pos_rows = (unsigned)(cout >> 6) * (unsigned)(cin / 16) * 64u;
#pragma unroll
        for (int a = 0; a < 4; ++a) {
            float r[2][4];
#pragma unroll
            for (int e = 0; e < 2; ++e) {
                r[e][0] = t[e][a][0];
                r[e][1] = 0.5f * (t[e][a][0] + t[e][a][1] + t[e][a][2]);
                r[e][2] = 0.5f * (t[e][a][0] - t[e][a][1] + t[e][a][2]);
                r[e][3] = t[e][a][2];
            }
#pragma unroll
            for (int b = 0; b < 4; ++b)
                wrow_store_pair(u, (size_t)(a * 4 + b) * pos_rows + row0, co & 63, k, r[0][b], r[1][b]);
        }
    }
    wrow_drain();
}

__global__ __launch_bounds__(256) void winograd_input_kernel(const float* __restrict__ x, float* __restrict__ v, int N,
                                                             int H, int W, int C4) {
    const int Ht = H >> 1, Wt = W >> 1;
    const long T = (long)N * Ht * Wt, total = T * C4;
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int c4 = (int)(i % C4);
        const long t = i / C4;
        const int tx = (int)(t % Wt);
        const long r = t / Wt;
        const int ty = (int)(r % Ht), n = (int)(r / Ht);
        f32x4 d[4][4];
#pragma unroll
        for (int a = 0; a < 4; ++a) {
            const int yy = 2 * ty - 1 + a;
#pragma unroll
            for (int b = 0; b < 4; ++b) {
                const int xx = 2 * tx - 1 + b;
                const bool ok = (unsigned)yy < (unsigned)H && (unsigned)xx < (unsigned)W;
                d[a][b] = ok ? reinterpret_cast<const f32x4*>(x)[(((size_t)n * H + yy) * W + xx) * C4 + c4]
                             : f32x4{0.f, 0.f, 0.f, 0.f};
            }
        }
        // B^T d: rows (d0 - d2, d1 + d2, d2 - d1, d1 - d3), then the same on columns
        f32x4 e[4][4];
#pragma unroll
        for (int b = 0; b < 4; ++b) {
            e[0][b] = d[0][b] - d[2][b];
            e[1][b] = d[1][b] + d[2][b];
            e[2][b] = d[2][b] - d[1][b];
            e[3][b] = d[1][b] - d[3][b];
        }
#pragma unroll
        for (int a = 0; a < 4; ++a) {
            const f32x4 o0 = e[a][0] - e[a][2], o1 = e[a][1] + e[a][2], o2 = e[a][2] - e[a][1], o3 = e[a][1] - e[a][3];
            f32x4* dst = reinterpret_cast<f32x4*>(v) + ((size_t)(a * 4) * T + t) * C4 + c4;
            dst[0] = o0;
            dst[(size_t)T * C4] = o1;
            dst[(size_t)2 * T * C4] = o2;
            dst[(size_t)3 * T * C4] = o3;
        }
    }
}

template <bool POOL>
__global__ __launch_bounds__(256) void winograd_output_kernel(const float* __restrict__ m, const float* __restrict__ scale,
                                                              const float* __restrict__ shift, float* __restrict__ y,
                                                              float* __restrict__ y_pool, int N, int H, int W, int C4,
                                                              int act, float slope, int y_from) {
    // y_from: y holds the images [y_from, N) (POOL only; see dvg_winograd_output)
    const int Ht = H >> 1, Wt = W >> 1;
    const long T = (long)N * Ht * Wt, total = T * C4;
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int c4 = (int)(i % C4);
        const long t = i / C4;
        const int tx = (int)(t % Wt);
        const long r = t / Wt;
        const int ty = (int)(r % Ht), n = (int)(r / Ht);
        f32x4 q[4][4];
#pragma unroll
        for (int a = 0; a < 4; ++a)
#pragma unroll
            for (int b = 0; b < 4; ++b) q[a][b] = reinterpret_cast<const f32x4*>(m)[((size_t)(a * 4 + b) * T + t) * C4 + c4];
        // A^T q: rows (q0 + q1 + q2, q1 - q2 - q3), then columns
        f32x4 s0[4], s1[4];
#pragma unroll
        for (int b = 0; b < 4; ++b) {
            s0[b] = q[0][b] + q[1][b] + q[2][b];
            s1[b] = q[1][b] - q[2][b] - q[3][b];
        }
        f32x4 o[2][2];
        o[0][0] = s0[0] + s0[1] + s0[2];
        o[0][1] = s0[1] - s0[2] - s0[3];
        o[1][0] = s1[0] + s1[1] + s1[2];
        o[1][1] = s1[1] - s1[2] - s1[3];
        const f32x4 sc = scale ? reinterpret_cast<const f32x4*>(scale)[c4] : f32x4{1.f, 1.f, 1.f, 1.f};
        const f32x4 sf = shift ? reinterpret_cast<const f32x4*>(shift)[c4] : f32x4{0.f, 0.f, 0.f, 0.f};
        f32x4 mx;
#pragma unroll
        for (int pp = 0; pp < 2; ++pp)
#pragma unroll
            for (int qq = 0; qq < 2; ++qq) {
                f32x4 val;
#pragma unroll
                for (int k = 0; k < 4; ++k) val[k] = apply_act(o[pp][qq][k] * sc[k] + sf[k], act, slope);
                if (!POOL || n >= y_from)
                    reinterpret_cast<f32x4*>(y)[(((size_t)(n - y_from) * H + 2 * ty + pp) * W + 2 * tx + qq) * C4 + c4] = val;
                if (POOL) {
                    if (pp == 0 && qq == 0) mx = val;
                    else
#pragma unroll
                        for (int k = 0; k < 4; ++k) mx[k] = fmaxf(mx[k], val[k]);
                }
            }
        if (POOL) reinterpret_cast<f32x4*>(y_pool)[(((size_t)n * Ht + ty) * Wt + tx) * C4 + c4] = mx;
    }
}

// ---- F(4x4, 3x3): 36 transform positions per 4x4 output tile (2.25 multiplies per output instead of 4; transformed
// tensors 2.25x the activation instead of 4x).  Lavin's matrices; fp32 error ~1e-5 of the layer's max output (F(2x2): 2e-6).
//   B^T = [4 0 -5 0 1 0; 0 -4 -4 1 1 0; 0 4 -4 -1 1 0; 0 -2 -1 2 1 0; 0 2 -1 -2 1 0; 0 4 0 -5 0 1]
//   G   = [1/4 0 0; -1/6 -1/6 -1/6; -1/6 1/6 -1/6; 1/24 1/12 1/6; 1/24 -1/12 1/6; 0 0 1]
//   A^T = [1 1 1 1 1 0; 0 1 -1 2 -2 0; 0 1 1 4 4 0; 0 1 -1 8 -8 1]
typedef float f32x2 __attribute__((ext_vector_type(2)));
template <typename V> struct VecN { static constexpr int N = sizeof(V) / 4; };
template <typename V> __device__ __forceinline__ float vget(const V& v, int k) { return v[k]; }
template <> __device__ __forceinline__ float vget<float>(const float& v, int) { return v; }
template <typename V> __device__ __forceinline__ void vset(V& v, int k, float x) { v[k] = x; }
template <> __device__ __forceinline__ void vset<float>(float& v, int, float x) { v = x; }
template <typename V> __device__ __forceinline__ V vzero() { V z; for (int k = 0; k < VecN<V>::N; ++k) vset(z, k, 0.f); return z; }

template <typename V>
__device__ __forceinline__ void bt4(const V (&d)[6], V (&e)[6]) {
    e[0] = 4.f * d[0] - 5.f * d[2] + d[4];
    e[1] = -4.f * (d[1] + d[2]) + d[3] + d[4];
    e[2] = 4.f * (d[1] - d[2]) - d[3] + d[4];
    e[3] = 2.f * (d[3] - d[1]) - d[2] + d[4];
    e[4] = 2.f * (d[1] - d[3]) - d[2] + d[4];
    e[5] = 4.f * d[1] - 5.f * d[3] + d[5];
}

template <typename V>
__device__ __forceinline__ void at4(const V (&q)[6], V (&o)[4]) {
    const V a = q[1] + q[2], b = q[1] - q[2], c = q[3] + q[4], d = q[3] - q[4];
    o[0] = q[0] + a + c;
    o[1] = b + 2.f * d;
    o[2] = a + 4.f * c;
    o[3] = b + 8.f * d + q[5];
}

__global__ void winograd4_weight_kernel(const float* __restrict__ w, float* __restrict__ u, int cout, int cin) {
    const float G[6][3] = {{0.25f, 0.f, 0.f}, {-1.f / 6, -1.f / 6, -1.f / 6}, {-1.f / 6, 1.f / 6, -1.f / 6},
                           {1.f / 24, 1.f / 12, 1.f / 6}, {1.f / 24, -1.f / 12, 1.f / 6}, {0.f, 0.f, 1.f}};
    // thread = (chunk, co, pair of k): two adjacent input channels, one 4-byte store per plane and position
    const unsigned total = (unsigned)cout * (unsigned)cin / 2;
    for (unsigned i = blockIdx.x * blockDim.x + threadIdx.x; i < total; i += gridDim.x * blockDim.x) {
        const unsigned r = i >> 3;
        const int co = (int)(r % (unsigned)cout), chunk = (int)(r / (unsigned)cout), k = (int)(i & 7) * 2;
        const float* g = w + ((size_t)co * cin + chunk * 16 + k) * 9;       // the two filters are adjacent: 18 floats
        float t[2][6][3];
#pragma unroll
        for (int e = 0; e < 2; ++e)
#pragma unroll
            for (int a = 0; a < 6; ++a)
#pragma unroll
                for (int s2 = 0; s2 < 3; ++s2)
                    t[e][a][s2] = G[a][0] * g[9 * e + s2] + G[a][1] * g[9 * e + 3 + s2] + G[a][2] * g[9 * e + 6 + s2];
        const unsigned row0 = ((unsigned)(co >> 6) * (unsigned)(cin / 16) + (unsigned)chunk) * 64u + (unsigned)(co & 63);
        const unsigned pos_rows = (unsigned)(cout >> 6) * (unsigned)(cin / 16) * 64u;
#pragma unroll
        for (int a = 0; a < 6; ++a)
#pragma unroll
            for (int b = 0; b < 6; ++b)
                wrow_store_pair(u, (size_t)(a * 6 + b) * pos_rows + row0, co & 63, k,
                                t[0][a][0] * G[b][0] + t[0][a][1] * G[b][1] + t[0][a][2] * G[b][2],
                                t[1][a][0] * G[b][0] + t[1][a][1] * G[b][1] + t[1][a][2] * G[b][2]);
    }
    wrow_drain();
}

// V = f32x4 / f32x2 / float: channels per thread.  A thread owns a whole 6x6 patch (36 loads, 36 stores), so the grid is
// tiles x C / width threads: at 8x8 maps and B = 64 that is 128 workgroups with float4 - the narrower forms fill the chip
// (accesses stay coalesced: consecutive lanes take consecutive channels) and need a quarter of the registers.
template <typename V>
__global__ __launch_bounds__(256) void winograd4_input_kernel(const float* __restrict__ x, float* __restrict__ v, int N,
                                                              int H, int W, int C4, long Tt, long t_off, int up) {
    // Tt / t_off: tiles per position of the destination and this call's first tile (the weight-gradient path concatenates
    // the tiles of several uses of a layer; Tt = N * tiles per image, t_off = 0 otherwise)
    // up = 1: x is stored at (N, H/2, W/2, C) and read through nearest-x2 upsampling (the x half of a decoder block's first
    // conv, vgg_64.py:93,98-105: the upsampled tensor never exists)
    const int Ht = H >> 2, Wt = W >> 2, Hs = H >> up, Ws = W >> up;
    const long T = (long)N * Ht * Wt, total = T * C4;
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int c4 = (int)(i % C4);
        const long t = i / C4;
        const int tx = (int)(t % Wt);
        const long r = t / Wt;
        const int ty = (int)(r % Ht), n = (int)(r / Ht);
        V e[6][6];   // B^T d (rows transformed), column b
#pragma unroll
        for (int b = 0; b < 6; ++b) {
            const int xx = 4 * tx - 1 + b;
            V d[6];
#pragma unroll
            for (int a = 0; a < 6; ++a) {
                const int yy = 4 * ty - 1 + a;
                const bool ok = (unsigned)yy < (unsigned)H && (unsigned)xx < (unsigned)W;
                d[a] = ok ? reinterpret_cast<const V*>(x)[(((size_t)n * Hs + (yy >> up)) * Ws + (xx >> up)) * C4 + c4] : vzero<V>();
            }
            V col[6];
            bt4(d, col);
#pragma unroll
            for (int a = 0; a < 6; ++a) e[a][b] = col[a];
        }
#pragma unroll
        for (int a = 0; a < 6; ++a) {
            V o[6];
            bt4(e[a], o);
#pragma unroll
            for (int b = 0; b < 6; ++b) reinterpret_cast<V*>(v)[((size_t)(a * 6 + b) * Tt + t_off + t) * C4 + c4] = o[b];
        }
    }
}

// A d for a 4-vector d (A = (A^T)^T, 6x4): the adjoint of at4
template <typename V>
__device__ __forceinline__ void a4(const V (&d)[4], V (&o)[6]) {
    const V s02 = d[0] + d[2], s13 = d[1] + d[3], t = d[0] + 4.f * d[2], u = 2.f * d[1] + 8.f * d[3];
    o[0] = d[0];
    o[1] = s02 + s13;
    o[2] = s02 - s13;
    o[3] = t + u;
    o[4] = t - u;
    o[5] = d[3];
}

// dM = A dY A^T per 4x4 tile of d(out) (the weight gradient's second operand, wino_wgrad.hip): 16 loads, 36 stores per
// thread; same tile order and destination layout [position][tile][channel] as the input transform.
template <typename V>
__global__ __launch_bounds__(256) void winograd4_dy_kernel(const float* __restrict__ dy, float* __restrict__ dm, int N, int H,
                                                           int W, int C4, long Tt, long t_off) {
    const int Ht = H >> 2, Wt = W >> 2;
    const long T = (long)N * Ht * Wt, total = T * C4;
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int c4 = (int)(i % C4);
        const long t = i / C4;
        const int tx = (int)(t % Wt);
        const long r = t / Wt;
        const int ty = (int)(r % Ht), n = (int)(r / Ht);
        V e[6][4];   // A d over the rows, column b
#pragma unroll
        for (int b = 0; b < 4; ++b) {
            V d[4];
#pragma unroll
            for (int a = 0; a < 4; ++a)
                d[a] = reinterpret_cast<const V*>(dy)[(((size_t)n * H + 4 * ty + a) * W + 4 * tx + b) * C4 + c4];
            V col[6];
            a4(d, col);
#pragma unroll
            for (int a = 0; a < 6; ++a) e[a][b] = col[a];
        }
#pragma unroll
        for (int a = 0; a < 6; ++a) {
            V o[6];
            a4(e[a], o);
#pragma unroll
            for (int b = 0; b < 6; ++b) reinterpret_cast<V*>(dm)[((size_t)(a * 6 + b) * Tt + t_off + t) * C4 + c4] = o[b];
        }
    }
}

template <bool POOL, typename V>

__global__ __launch_bounds__(256) void winograd4_output_kernel(const float* __restrict__ m, const float* __restrict__ scale,
                                                               const float* __restrict__ shift, float* __restrict__ y,
                                                               float* __restrict__ y_pool, int N, int H, int W, int C4,
                                                               int act, float slope, const float* __restrict__ addend,
                                                               int y_from) {
    // y_from (POOL only, else 0): y holds the images [y_from, N); the images before store only their pooled output
    // addend (optional): raw partial sums in y's shape, y = act((A^T M A + addend) * scale + shift) - the hoisted skip half
    // of a decoder block's first conv (the same role as the igemm kernels' addend)
    constexpr int VN = VecN<V>::N;
    const int Ht = H >> 2, Wt = W >> 2;
    const long T = (long)N * Ht * Wt, total = T * C4;
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int c4 = (int)(i % C4);
        const long t = i / C4;
        const int tx = (int)(t % Wt);
        const long r = t / Wt;
        const int ty = (int)(r % Ht), n = (int)(r / Ht);
        V s[4][6];   // A^T q (rows), column b
#pragma unroll
        for (int b = 0; b < 6; ++b) {
            V q[6];
#pragma unroll
            for (int a = 0; a < 6; ++a) q[a] = reinterpret_cast<const V*>(m)[((size_t)(a * 6 + b) * T + t) * C4 + c4];
            V col[4];
            at4(q, col);
#pragma unroll
            for (int pp = 0; pp < 4; ++pp) s[pp][b] = col[pp];
        }
        V sc, sf;
#pragma unroll
        for (int k = 0; k < VN; ++k) {
            vset(sc, k, scale ? scale[c4 * VN + k] : 1.f);
            vset(sf, k, shift ? shift[c4 * VN + k] : 0.f);
        }
        V val[4][4];
#pragma unroll
        for (int pp = 0; pp < 4; ++pp) {
            V o[4];
            at4(s[pp], o);
#pragma unroll
            for (int qq = 0; qq < 4; ++qq) {
                const size_t px = (((size_t)n * H + 4 * ty + pp) * W + 4 * tx + qq) * C4 + c4;
                V ad = vzero<V>();
                if (addend) ad = reinterpret_cast<const V*>(addend)[px];
#pragma unroll
                for (int k = 0; k < VN; ++k)
                    vset(val[pp][qq], k, apply_act((vget(o[qq], k) + vget(ad, k)) * vget(sc, k) + vget(sf, k), act, slope));
                if (!POOL || n >= y_from) reinterpret_cast<V*>(y)[px - (size_t)y_from * H * W * C4] = val[pp][qq];
            }
        }
        if (POOL) {
#pragma unroll
            for (int pp = 0; pp < 2; ++pp)
#pragma unroll
                for (int qq = 0; qq < 2; ++qq) {
                    V mx;
#pragma unroll
                    for (int k = 0; k < VN; ++k)
                        vset(mx, k, fmaxf(fmaxf(vget(val[2 * pp][2 * qq], k), vget(val[2 * pp][2 * qq + 1], k)),
                                          fmaxf(vget(val[2 * pp + 1][2 * qq], k), vget(val[2 * pp + 1][2 * qq + 1], k))));
                    reinterpret_cast<V*>(y_pool)[(((size_t)n * (H >> 1) + 2 * ty + pp) * (W >> 1) + 2 * tx + qq) * C4 + c4] = mx;
                }
        }
    }
}

// Output transform of layer L + input transform of layer L + 1 in ONE pass, for two consecutive eval-mode vgg_layers at the
// same resolution whose intermediate activation has no other consumer (the inner layers of a vgg block, vgg_64.py:24-43,
// 70-87): y = act(scale * A^T M A + shift) is staged in LDS (one image x CS channels per workgroup: 16 KB) and leaves as
// V' = B^T y B - the activation never travels to HBM and back (M in, V' out: 4.5 units of traffic instead of 6.5) and one
// launch of two disappears.  HW = 8 or 16 (map side); thread = one 4x4 tile x one channel.
template <int HW>
__global__ __launch_bounds__(256) void winograd4_out_in_kernel(const float* __restrict__ m, const float* __restrict__ scale,
                                                               const float* __restrict__ shift, float* __restrict__ v, int N,
                                                               int C, int act, float slope, const float* __restrict__ addend) {
    constexpr int WT = HW / 4, TI = WT * WT, CS = 256 / TI;    // tiles per image side / per image, channels per workgroup
    __shared__ float ys[HW * HW * CS];
    const int tid = threadIdx.x, cl = tid % CS, t_img = tid / CS, ty = t_img / WT, tx = t_img % WT;
    const int cblocks = C / CS;
    const int n = blockIdx.x / cblocks, c = (blockIdx.x % cblocks) * CS + cl;
    const long T = (long)N * TI, t = (long)n * TI + t_img;
    {
        float sq[4][6];   // A^T q (rows), column b
#pragma unroll
        for (int b = 0; b < 6; ++b) {
            float q[6];
#pragma unroll
            for (int a = 0; a < 6; ++a) q[a] = m[((size_t)(a * 6 + b) * T + t) * C + c];
            float col[4];
            at4(q, col);
#pragma unroll
            for (int pp = 0; pp < 4; ++pp) sq[pp][b] = col[pp];
        }
        const float sc = scale ? scale[c] : 1.f, sf = shift ? shift[c] : 0.f;
#pragma unroll
        for (int pp = 0; pp < 4; ++pp) {
            float o[4];
            at4(sq[pp], o);
#pragma unroll
            for (int qq = 0; qq < 4; ++qq) {
                const float ad = addend ? addend[(((size_t)n * HW + 4 * ty + pp) * HW + 4 * tx + qq) * C + c] : 0.f;
                ys[((4 * ty + pp) * HW + 4 * tx + qq) * CS + cl] = apply_act((o[qq] + ad) * sc + sf, act, slope);
            }
        }
    }
    __syncthreads();
    float e[6][6];   // B^T d (rows transformed), column b
#pragma unroll
    for (int b = 0; b < 6; ++b) {
        const int xx = 4 * tx - 1 + b;
        float d[6];
#pragma unroll
        for (int a = 0; a < 6; ++a) {
            const int yy = 4 * ty - 1 + a;
            d[a] = ((unsigned)yy < (unsigned)HW && (unsigned)xx < (unsigned)HW) ? ys[(yy * HW + xx) * CS + cl] : 0.f;
        }
        float col[6];
        bt4(d, col);
#pragma unroll
        for (int a = 0; a < 6; ++a) e[a][b] = col[a];
    }
#pragma unroll
    for (int a = 0; a < 6; ++a) {
        float o[6];
        bt4(e[a], o);
#pragma unroll
        for (int b = 0; b < 6; ++b) v[((size_t)(a * 6 + b) * T + t) * C + c] = o[b];
    }
}

// Generalised hand-over (r03): output transform of layer L, optionally its 2x2 max-pool, and the input transform of the
// layer that consumes the result, in ONE pass per (image, CS-channel block):
//   POOL = false: two consecutive layers of a vgg block at the same resolution HW (8, 16 or 32) - as winograd4_out_in_kernel,
//                 but V channels per thread and CS channels per workgroup, so that a workgroup's global accesses are runs of
//                 CS * 4 bytes (the <16> instantiation above moves 64-byte runs: 3.2-3.7 TB/s; 256-byte runs: > 5);
//   POOL = true : the LAST layer of an encoder stage (vgg_64.py:51-56): the full-resolution activation is the skip tensor and
//                 goes to HBM (y), its 2x2 max-pool is staged in LDS and leaves as the next stage's input transform - the
//                 pooled tensor is never written or re-read and one launch disappears.
// Arithmetic per element is exactly that of the separate kernels (at4 / scale, shift, activation / max / bt4 in the same
// order): results are bit-identical to winograd4_output_kernel followed by winograd4_input_kernel.
//   UP = true   : the LAST layer of a decoder block hands over to the first conv of the NEXT block through nearest-x2
//                 upsampling (vgg_64.py:93,98-105, the x half of the concat conv in Winograd form): y (HW x HW) is staged in
//                 LDS and leaves as the input transform of up2(y) (2 HW x 2 HW, 4 x the tiles); neither y nor its upsampled
//                 form is written.  Bit-identical to winograd4_output_kernel followed by winograd4_input_kernel<up = 1>.
template <int HW, int CS, typename V, bool POOL, int NT, bool UP = false>
__global__ __launch_bounds__(NT) void winograd4_chain_kernel(const float* __restrict__ m, const float* __restrict__ scale,
                                                             const float* __restrict__ shift, float* __restrict__ y,
                                                             float* __restrict__ v, int N, int C, int act, float slope,
                                                             const float* __restrict__ addend, int y_from) {
    // y_from (POOL only): y holds the images [y_from, N) - the skip tensor of the images before is not stored
    constexpr int VN = VecN<V>::N;
    constexpr int WT = HW / 4, TI = WT * WT;         // tiles per image side / per image of layer L
    static_assert(!(POOL && UP), "pool and upsample exclude each other");
    constexpr int SS = POOL ? HW / 2 : HW;           // side of the map staged in LDS
    constexpr int S2 = UP ? 2 * HW : SS;             // side of the map layer L + 1 reads (UP: through the upsampling)
    constexpr int WT2 = S2 / 4, TI2 = WT2 * WT2;
    constexpr int CV = CS / VN;                      // channel vectors per workgroup
    extern __shared__ __attribute__((aligned(16))) float chain_ys[];   // [SS * SS][CS]
    float* const ys = chain_ys;
    const int cblocks = C / CS;
    const int n = blockIdx.x / cblocks, c0 = (blockIdx.x % cblocks) * CS;
    const int CVg = C / VN;
    const long T = (long)N * TI;
    for (int u = threadIdx.x; u < TI * CV; u += NT) {
        const int cv = u % CV, t_img = u / CV, ty = t_img / WT, tx = t_img % WT;
        const long t = (long)n * TI + t_img;
        const int cg = c0 / VN + cv;
        V s[4][6];   // A^T q (rows), column b
#pragma unroll
        for (int b = 0; b < 6; ++b) {
            V q[6];
#pragma unroll
            for (int a = 0; a < 6; ++a) q[a] = wino_ld(reinterpret_cast<const V*>(m) + ((size_t)(a * 6 + b) * T + t) * CVg + cg);
            V col[4];
            at4(q, col);
#pragma unroll
            for (int pp = 0; pp < 4; ++pp) s[pp][b] = col[pp];
        }
        V sc, sf;
#pragma unroll
        for (int k = 0; k < VN; ++k) {
            vset(sc, k, scale ? scale[c0 + cv * VN + k] : 1.f);
            vset(sf, k, shift ? shift[c0 + cv * VN + k] : 0.f);
        }
        V val[4][4];
#pragma unroll
        for (int pp = 0; pp < 4; ++pp) {
            V o[4];
            at4(s[pp], o);
#pragma unroll
            for (int qq = 0; qq < 4; ++qq) {
                V ad = vzero<V>();
                if (addend) ad = reinterpret_cast<const V*>(addend)[(((size_t)n * HW + 4 * ty + pp) * HW + 4 * tx + qq) * CVg + cg];
#pragma unroll
                for (int k = 0; k < VN; ++k)
                    vset(val[pp][qq], k, apply_act((vget(o[qq], k) + vget(ad, k)) * vget(sc, k) + vget(sf, k), act, slope));
                if (POOL) {
                    if (n >= y_from)       // workgroup-uniform
                        reinterpret_cast<V*>(y)[(((size_t)(n - y_from) * HW + 4 * ty + pp) * HW + 4 * tx + qq) * CVg + cg] = val[pp][qq];
                } else
                    *reinterpret_cast<V*>(&ys[((4 * ty + pp) * HW + 4 * tx + qq) * CS + cv * VN]) = val[pp][qq];
            }
        }
        if (POOL) {
#pragma unroll
            for (int pp = 0; pp < 2; ++pp)
#pragma unroll
                for (int qq = 0; qq < 2; ++qq) {
                    V mx;
#pragma unroll
                    for (int k = 0; k < VN; ++k)
                        vset(mx, k, fmaxf(fmaxf(vget(val[2 * pp][2 * qq], k), vget(val[2 * pp][2 * qq + 1], k)),
                                          fmaxf(vget(val[2 * pp + 1][2 * qq], k), vget(val[2 * pp + 1][2 * qq + 1], k))));
                    *reinterpret_cast<V*>(&ys[((2 * ty + pp) * SS + 2 * tx + qq) * CS + cv * VN]) = mx;
                }
        }
    }
    __syncthreads();
    const long T2 = (long)N * TI2;
    for (int u = threadIdx.x; u < TI2 * CV; u += NT) {
        const int cv = u % CV, t_img = u / CV, ty = t_img / WT2, tx = t_img % WT2;
        const long t2 = (long)n * TI2 + t_img;
        const int cg = c0 / VN + cv;
        V e[6][6];   // B^T d (rows transformed), column b
#pragma unroll
        for (int b = 0; b < 6; ++b) {
            const int xx = 4 * tx - 1 + b;
            V d[6];
#pragma unroll
            for (int a = 0; a < 6; ++a) {
                const int yy = 4 * ty - 1 + a;
                d[a] = ((unsigned)yy < (unsigned)S2 && (unsigned)xx < (unsigned)S2)
                           ? *reinterpret_cast<const V*>(&ys[((yy >> (UP ? 1 : 0)) * SS + (xx >> (UP ? 1 : 0))) * CS + cv * VN])
                           : vzero<V>();
            }
            V col[6];
            bt4(d, col);
#pragma unroll
            for (int a = 0; a < 6; ++a) e[a][b] = col[a];
        }
#pragma unroll
        for (int a = 0; a < 6; ++a) {
            V o[6];
            bt4(e[a], o);
#pragma unroll
            for (int b = 0; b < 6; ++b) wino_st(reinterpret_cast<V*>(v) + ((size_t)(a * 6 + b) * T2 + t2) * CVg + cg, o[b]);
        }
    }
}

template <int HW, int CS, typename V, bool POOL, int NT, bool UP = false>
static int launch_chain(const float* mm, const float* scale, const float* shift, float* y, float* v, int N, int C, int act,
                        float slope, hipStream_t st, const float* addend = nullptr, int y_from = 0) {
    constexpr int S2 = POOL ? HW / 2 : HW;
    constexpr size_t lds = (size_t)S2 * S2 * CS * 4;
    auto kern = winograd4_chain_kernel<HW, CS, V, POOL, NT, UP>;
    static bool attr = false;
    if (!attr && lds > 48 * 1024) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return fail(DVG_ERR_HIP, "hipFuncSetAttribute: %s", hipGetErrorString(e));
        attr = true;
    }
    hipLaunchKernelGGL(kern, dim3((unsigned)((long)N * (C / CS))), dim3(NT), lds, st, mm, scale, shift, y, v, N, C, act, slope,
                       addend, y_from);
    return check_launch("winograd4_chain");
}

// Decoder stem -> first conv of the first block (vgg_64.py:65-69 then :93,98-99): ConvTranspose2d(dim,C,4,1,0) on a 1x1 map +
// BN + LeakyReLU gives a 4 x 4 x C map per sample (a GEMM with K = dim: dense.hip's stem_kernel), which upc2 reads through `up`
// and the x half of its concat conv in Winograd form.  One kernel: a workgroup computes the 4 x 4 map of BR samples x 16
// channels into LDS (thread = one output column (pixel, channel), its K weights in registers, the latent vectors broadcast from
// LDS - the arithmetic of stem_kernel, bit for bit) and writes the input transform of the UPSAMPLED 8 x 8 map (the arithmetic
// of winograd4_input_kernel<up = 1>): the 4 x 4 map never travels.  wt: the weight transposed to [KP][16 C], rows K..KP-1 zero.
template <int KP>
__global__ __launch_bounds__(256) void stem_up_input_kernel(const float* __restrict__ vec, int ldv, const float* __restrict__ wt,
                                                            const float* __restrict__ scale, const float* __restrict__ shift,
                                                            float* __restrict__ v, int M, int C, int K, int act, float slope) {
    constexpr int BR = 8;
    __shared__ float vs[BR * KP];
    __shared__ float ys[BR * 16 * 16];     // [sample][pixel][channel]
    const int tid = threadIdx.x, cl = tid & 15, hw = tid >> 4;
    const int c = blockIdx.x * 16 + cl;
    const size_t N = (size_t)16 * C, n = (size_t)hw * C + c;
    const int m0 = blockIdx.y * BR, mrows = min(BR, M - m0);
    for (int i = tid; i < BR * KP; i += 256) {
        const int r = i / KP, k = i % KP;
        vs[i] = (r < mrows && k < K) ? vec[(size_t)(m0 + r) * ldv + k] : 0.f;
    }
    float w[KP];
#pragma unroll
    for (int k = 0; k < KP; ++k) w[k] = wt[(size_t)k * N + n];
    __syncthreads();
    const float sc = scale ? scale[c] : 1.f, sf = shift ? shift[c] : 0.f;
#pragma unroll
    for (int bb = 0; bb < BR; ++bb) {
        const f32x4* vr = reinterpret_cast<const f32x4*>(vs + bb * KP);
        float s0 = 0.f, s1 = 0.f;
#pragma unroll
        for (int k4 = 0; k4 < KP / 4; ++k4) {
            const f32x4 v4 = vr[k4];
            s0 = fmaf(v4[0], w[4 * k4], s0);
            s1 = fmaf(v4[1], w[4 * k4 + 1], s1);
            s0 = fmaf(v4[2], w[4 * k4 + 2], s0);
            s1 = fmaf(v4[3], w[4 * k4 + 3], s1);
        }
        ys[(bb * 16 + hw) * 16 + cl] = apply_act((s0 + s1) * sc + sf, act, slope);
    }
    __syncthreads();
    const long T = (long)M * 4;
    for (int u = tid; u < BR * 64; u += 256) {
        const int ucl = u & 15, tile = (u >> 4) & 3, bb = u >> 6;
        if (bb >= mrows) continue;
        const int ty = tile >> 1, tx = tile & 1;
        const float* src = ys + bb * 256 + ucl;
        float e[6][6];
#pragma unroll
        for (int b = 0; b < 6; ++b) {
            const int xx = 4 * tx - 1 + b;
            float d[6];
#pragma unroll
            for (int a = 0; a < 6; ++a) {
                const int yy = 4 * ty - 1 + a;
                d[a] = ((unsigned)yy < 8u && (unsigned)xx < 8u) ? src[((yy >> 1) * 4 + (xx >> 1)) * 16] : 0.f;
            }
            float col[6];
            bt4(d, col);
#pragma unroll
            for (int a = 0; a < 6; ++a) e[a][b] = col[a];
        }
        const long t = (long)(m0 + bb) * 4 + tile;
        float* dst = v + t * C + blockIdx.x * 16 + ucl;
#pragma unroll
        for (int a = 0; a < 6; ++a) {
            float o[6];
            bt4(e[a], o);
#pragma unroll
            for (int b = 0; b < 6; ++b) dst[(size_t)(a * 6 + b) * T * C] = o[b];
        }
    }
}

static int chain_variant() { return 0; }      // (the variants below were A/B'd per shape: numbers in the comments)

static inline unsigned wgrid(long n) {
    long g = (n + 255) / 256;
    return (unsigned)(g > 16384 ? 16384 : (g < 1 ? 1 : g));
}

}  // namespace dvg

using namespace dvg;

extern "C" int dvg_winograd_weight(const float* w_oihw, float* u_k16, int cout, int cin, int m, void* stream) {
    DVG_REQUIRE(w_oihw && u_k16, DVG_ERR_NULL, "dvg_winograd_weight: NULL pointer");
    DVG_REQUIRE(cout > 0 && cout % 64 == 0 && cin > 0 && cin % 16 == 0 && (m == 2 || m == 4), DVG_ERR_SHAPE,
                "dvg_winograd_weight: Cin must be a multiple of 16, Cout of 64, m 2 or 4");
    if (m == 2)
        hipLaunchKernelGGL(winograd_weight_kernel, dim3(wgrid((long)cout * cin / 2)), dim3(256), 0, (hipStream_t)stream, w_oihw,
                           u_k16, cout, cin);
    else
        hipLaunchKernelGGL(winograd4_weight_kernel, dim3(wgrid((long)cout * cin / 2)), dim3(256), 0, (hipStream_t)stream, w_oihw,
                           u_k16, cout, cin);
    return check_launch("dvg_winograd_weight");
}

extern "C" int dvg_winograd_input(const float* x, float* v, int N, int H, int W, int C, int m, int upsample, void* stream) {
    DVG_REQUIRE(x && v, DVG_ERR_NULL, "dvg_winograd_input: NULL pointer");
    DVG_REQUIRE((m == 2 || m == 4) && N > 0 && H > 0 && W > 0 && H % m == 0 && W % m == 0 && C > 0 && C % 4 == 0,
                DVG_ERR_SHAPE, "dvg_winograd_input: m 2 or 4, H and W multiples of m, C %% 4 == 0 needed");
    DVG_REQUIRE(upsample == 0 || (upsample == 1 && m == 4 && H % 2 == 0 && W % 2 == 0), DVG_ERR_SHAPE,
                "dvg_winograd_input: upsample needs m = 4 and even H, W");
    const int up = upsample;
    DVG_REQUIRE(aligned16(x) && aligned16(v), DVG_ERR_ALIGN, "dvg_winograd_input: alignment");
    const long tiles = (long)N * (H / m) * (W / m);
    const hipStream_t st = (hipStream_t)stream;
    if (m == 2) {
        hipLaunchKernelGGL(winograd_input_kernel, dim3(wgrid(tiles * (C / 4))), dim3(256), 0, st, x, v, N, H, W, C / 4);
    } else {
        // channels per thread: the widest form that still gives >= 1024 workgroups (see the kernel)
        if (tiles * (C / 4) >= 1024L * 256)
            hipLaunchKernelGGL(winograd4_input_kernel<f32x4>, dim3(wgrid(tiles * (C / 4))), dim3(256), 0, st, x, v, N, H, W, C / 4, tiles, 0L, up);
        else if (tiles * (C / 2) >= 1024L * 256)
            hipLaunchKernelGGL(winograd4_input_kernel<f32x2>, dim3(wgrid(tiles * (C / 2))), dim3(256), 0, st, x, v, N, H, W, C / 2, tiles, 0L, up);
        else
            hipLaunchKernelGGL(winograd4_input_kernel<float>, dim3(wgrid(tiles * C)), dim3(256), 0, st, x, v, N, H, W, C, tiles, 0L, up);
    }
    return check_launch("dvg_winograd_input");
}

// Operands of the Winograd-form weight gradient (wino_wgrad.hip) for ONE use of a layer: V = B^T x B of the layer input
// x (N,H,W,Cin) and dM = A dY A^T of d(out) (N,H,W,Cout), written at tile offset t_off of buffers that hold t_total tiles
// per position ([36][t_total][C]; several uses of a layer are concatenated along the tile axis).  x or dy may be NULL to
// skip that operand.  F(4x4,3x3) only.
extern "C" int dvg_winograd_wgrad_operands(const float* x, const float* dy, float* v, float* dm, int N, int H, int W, int Cin,
                                           int Cout, long t_total, long t_off, void* stream) {
    DVG_REQUIRE((x && v) || (dy && dm), DVG_ERR_NULL, "dvg_winograd_wgrad_operands: nothing to do");
    DVG_REQUIRE(N > 0 && H > 0 && W > 0 && H % 4 == 0 && W % 4 == 0 && Cin > 0 && Cin % 4 == 0 && Cout > 0 && Cout % 4 == 0,
                DVG_ERR_SHAPE, "dvg_winograd_wgrad_operands: H and W multiples of 4, channels multiples of 4 needed");
    const long tiles = (long)N * (H / 4) * (W / 4);
    DVG_REQUIRE(t_off >= 0 && t_off + tiles <= t_total, DVG_ERR_SHAPE,
                "dvg_winograd_wgrad_operands: tiles [%ld, %ld) outside the buffer of %ld", t_off, t_off + tiles, t_total);
    DVG_REQUIRE(aligned16(x) && aligned16(dy) && aligned16(v) && aligned16(dm), DVG_ERR_ALIGN,
                "dvg_winograd_wgrad_operands: alignment");
    const hipStream_t st = (hipStream_t)stream;
#define WOPS(KERNEL_, SRC_, DST_, C_, ...)                                                                                        \
    do {                                                                                                                       \
        if (tiles * ((C_) / 4) >= 1024L * 256)                                                                                 \
            hipLaunchKernelGGL(KERNEL_<f32x4>, dim3(wgrid(tiles * ((C_) / 4))), dim3(256), 0, st, SRC_, DST_, N, H, W, (C_) / 4, t_total, t_off __VA_ARGS__); \
        else if (tiles * ((C_) / 2) >= 1024L * 256)                                                                            \
            hipLaunchKernelGGL(KERNEL_<f32x2>, dim3(wgrid(tiles * ((C_) / 2))), dim3(256), 0, st, SRC_, DST_, N, H, W, (C_) / 2, t_total, t_off __VA_ARGS__); \
        else                                                                                                                   \
            hipLaunchKernelGGL(KERNEL_<float>, dim3(wgrid(tiles * (C_))), dim3(256), 0, st, SRC_, DST_, N, H, W, (C_), t_total, t_off __VA_ARGS__);           \
    } while (0)
    if (x && v) {
        WOPS(winograd4_input_kernel, x, v, Cin, , 0);
        if (int e = check_launch("dvg_winograd_wgrad_operands (input)")) return e;
    }
    if (dy && dm) {
        WOPS(winograd4_dy_kernel, dy, dm, Cout, );
        if (int e = check_launch("dvg_winograd_wgrad_operands (dy)")) return e;
    }
#undef WOPS
    return DVG_OK;
}

// M (36, T, C) of layer L -> V (36, T, C) of layer L + 1 (winograd4_out_in_kernel / winograd4_chain_kernel): H == W in
// {8, 16, 32}, F(4x4,3x3), C % 64 == 0.  The activation y = act(scale * A^T M A + shift) itself is not written.
extern "C" int dvg_winograd_output_input(const float* mm, const float* scale, const float* shift, float* v_next, int N, int H,
                                         int W, int C, int act, float slope, const float* addend, void* stream) {
    DVG_REQUIRE(mm && v_next, DVG_ERR_NULL, "dvg_winograd_output_input: NULL pointer");
    DVG_REQUIRE(aligned16(addend), DVG_ERR_ALIGN, "dvg_winograd_output_input: addend alignment");
    DVG_REQUIRE(N > 0 && H == W && (H == 8 || H == 16 || H == 32) && C > 0 && C % 64 == 0, DVG_ERR_SHAPE,
                "dvg_winograd_output_input: 8x8, 16x16 or 32x32 maps, C %% 64 == 0 needed (got %dx%d, C=%d)", H, W, C);
    DVG_REQUIRE(act >= 0 && act <= 3, DVG_ERR_SHAPE, "dvg_winograd_output_input: bad act");
    DVG_REQUIRE(aligned16(mm) && aligned16(v_next), DVG_ERR_ALIGN, "dvg_winograd_output_input: alignment");
    const hipStream_t st = (hipStream_t)stream;
    const int var = chain_variant();
    // Measured per shape (tools/bench_wino_parts.py, us at B = 64 / B = 576; the variants were A/B'd per shape in r03):
    //   16x16 256ch: r02 kernel (64-byte runs) 20.3 / 212; <16,64,f32x2> 15.1 / 140; <16,32,float> 13.8 / 160
    //   32x32 128ch: separate output + input passes 42.2 / 389; <32,16,f32x2> 36.5 / 388; <32,32,f32x4> (128 KB of LDS) 27.3 / 284
    //    8x8  512ch: r02 kernel 7.6 / 73 = <8,64,float> - kept
    if (H == 8) {
        hipLaunchKernelGGL(winograd4_out_in_kernel<8>, dim3((unsigned)((long)N * (C / 64))), dim3(256), 0, st, mm, scale, shift,
                           v_next, N, C, act, slope, addend);
    } else if (H == 16) {
        if (var == 9) {
            hipLaunchKernelGGL(winograd4_out_in_kernel<16>, dim3((unsigned)((long)N * (C / 16))), dim3(256), 0, st, mm, scale,
                               shift, v_next, N, C, act, slope, addend);
        } else if (var == 2 || (var == 0 && (long)N * (C / 64) < 1024)) {
            return launch_chain<16, 32, float, false, 512>(mm, scale, shift, nullptr, v_next, N, C, act, slope, st, addend);
        } else {
            return launch_chain<16, 64, f32x2, false, 512>(mm, scale, shift, nullptr, v_next, N, C, act, slope, st, addend);
        }
    } else {
        if (var == 1) return launch_chain<32, 16, f32x2, false, 512>(mm, scale, shift, nullptr, v_next, N, C, act, slope, st, addend);
        return launch_chain<32, 32, f32x4, false, 512>(mm, scale, shift, nullptr, v_next, N, C, act, slope, st, addend);
    }
    return check_launch("dvg_winograd_output_input");
}

// stem_up_input_kernel: vec (M, K) row stride ldv; w_kn the stem weight as [KP][16 C] (dvg_stem_gemm's operand: column
// (h * 4 + w) * C + c, rows K..KP-1 zero); v_next (36, 4 M, C) = the F(4x4,3x3) input transform of
// UpsamplingNearest2d(2)(act(scale * convT(vec) + shift)).  C % 16 == 0, KP in {96, 128}.
extern "C" int dvg_stem_up_winograd_input(const float* vec, int ldv, const float* w_kn, int KP, const float* scale,
                                          const float* shift, float* v_next, int M, int C, int K, int act, float slope,
                                          void* stream) {
    DVG_REQUIRE(vec && w_kn && v_next, DVG_ERR_NULL, "dvg_stem_up_winograd_input: NULL pointer");
    DVG_REQUIRE(M > 0 && C > 0 && C % 16 == 0 && K > 0 && K <= KP && (KP == 96 || KP == 128) && ldv >= K, DVG_ERR_SHAPE,
                "dvg_stem_up_winograd_input: bad shape M=%d C=%d K=%d KP=%d (C %% 16 == 0, KP 96 or 128)", M, C, K, KP);
    DVG_REQUIRE(act >= 0 && act <= 3, DVG_ERR_SHAPE, "dvg_stem_up_winograd_input: bad act");
    DVG_REQUIRE(aligned16(v_next), DVG_ERR_ALIGN, "dvg_stem_up_winograd_input: alignment");
    const dim3 grid(C / 16, (M + 7) / 8);
    if (KP == 96)
        hipLaunchKernelGGL(stem_up_input_kernel<96>, grid, dim3(256), 0, (hipStream_t)stream, vec, ldv, w_kn, scale, shift, v_next,
                           M, C, K, act, slope);
    else
        hipLaunchKernelGGL(stem_up_input_kernel<128>, grid, dim3(256), 0, (hipStream_t)stream, vec, ldv, w_kn, scale, shift, v_next,
                           M, C, K, act, slope);
    return check_launch("dvg_stem_up_winograd_input");
}

// Last layer of a decoder block -> first conv of the next block (vgg_64.py:98-105): M (36, T, C) of an H x H layer ->
// V' (36, 4 T, C), the F(4x4,3x3) input transform of UpsamplingNearest2d(2)(act(scale * A^T M A + shift)) (winograd4_chain_kernel
// <UP>); neither the activation nor its upsampled form is written.  H == W == 8 (the next block works at 16 x 16).
extern "C" int dvg_winograd_output_up_input(const float* mm, const float* scale, const float* shift, float* v_next, int N, int H,
                                            int W, int C, int act, float slope, void* stream) {
    DVG_REQUIRE(mm && v_next, DVG_ERR_NULL, "dvg_winograd_output_up_input: NULL pointer");
    DVG_REQUIRE(N > 0 && H == W && H == 8 && C > 0 && C % 64 == 0, DVG_ERR_SHAPE,
                "dvg_winograd_output_up_input: 8x8 maps, C %% 64 == 0 needed (got %dx%d, C=%d)", H, W, C);
    DVG_REQUIRE(act >= 0 && act <= 3, DVG_ERR_SHAPE, "dvg_winograd_output_up_input: bad act");
    DVG_REQUIRE(aligned16(mm) && aligned16(v_next), DVG_ERR_ALIGN, "dvg_winograd_output_up_input: alignment");
    const hipStream_t st = (hipStream_t)stream;
    return launch_chain<8, 32, float, false, 512, true>(mm, scale, shift, nullptr, v_next, N, C, act, slope, st);
}

// Last layer of an encoder stage (vgg_64.py:51-56): M (36, T, C) -> y = act(scale * A^T M A + shift) (N,H,W,C) NHWC, the
// skip tensor, AND V' (36, T / 4, C), the F(4x4,3x3) input transform of maxpool2x2(y) for the first layer of the next stage
// (winograd4_chain_kernel<POOL>).  H == W in {16, 32}, C % 64 == 0.  The pooled tensor itself is not written.
// y_from (ABI 8): y holds the images [y_from, N) only (NULL allowed when y_from == N): a rollout keeps the skip tensors of the
// last conditioning frame alone (generate_frames.py:154-157).
extern "C" int dvg_winograd_output_pool_input(const float* mm, const float* scale, const float* shift, float* y, float* v_next,
                                              int N, int H, int W, int C, int act, float slope, int y_from, void* stream) {
    DVG_REQUIRE(mm && v_next && (y || y_from == N), DVG_ERR_NULL, "dvg_winograd_output_pool_input: NULL pointer");
    DVG_REQUIRE(y_from >= 0 && y_from <= N, DVG_ERR_SHAPE, "dvg_winograd_output_pool_input: y_from=%d outside [0, N]", y_from);
    DVG_REQUIRE(N > 0 && H == W && (H == 16 || H == 32) && C > 0 && C % 64 == 0, DVG_ERR_SHAPE,
                "dvg_winograd_output_pool_input: 16x16 or 32x32 maps, C %% 64 == 0 needed (got %dx%d, C=%d)", H, W, C);
    DVG_REQUIRE(act >= 0 && act <= 3, DVG_ERR_SHAPE, "dvg_winograd_output_pool_input: bad act");
    DVG_REQUIRE(aligned16(mm) && aligned16(y) && aligned16(v_next), DVG_ERR_ALIGN, "dvg_winograd_output_pool_input: alignment");
    const hipStream_t st = (hipStream_t)stream;
    const int var = chain_variant();
    // us at B = 64 / 576 against output(+pool) followed by input: 16 -> 8 (256 ch) 12.2 / 120 vs 18.3 / 124; 32 -> 16 (128 ch)
    // 21.7 / 244 vs 30.0 / 247 (the 16-channel variant: 33.5 / 324)
    (void)var;
    if (H == 16) return launch_chain<16, 64, f32x2, true, 512>(mm, scale, shift, y, v_next, N, C, act, slope, st, nullptr, y_from);
    return launch_chain<32, 32, f32x2, true, 1024>(mm, scale, shift, y, v_next, N, C, act, slope, st, nullptr, y_from);
}

// y_from (ABI 8; 0 unless y_pool is given): y holds the images [y_from, N) only, the images before store only y_pool
// (NULL y allowed when y_from == N).
extern "C" int dvg_winograd_output(const float* m, const float* scale, const float* shift, float* y, float* y_pool, int N,
                                   int H, int W, int C, int act, float slope, int mt, const float* addend, int y_from,
                                   void* stream) {
    DVG_REQUIRE(m && (y || (y_pool && y_from == N)), DVG_ERR_NULL, "dvg_winograd_output: NULL pointer");
    DVG_REQUIRE(y_from >= 0 && y_from <= N && (y_from == 0 || y_pool != nullptr), DVG_ERR_SHAPE,
                "dvg_winograd_output: y_from=%d needs 0 <= y_from <= N and a pooled output", y_from);
    DVG_REQUIRE(addend == nullptr || (mt == 4 && y_pool == nullptr && aligned16(addend)), DVG_ERR_SHAPE,
                "dvg_winograd_output: addend needs m = 4, no pooled output, 16-byte alignment");
    DVG_REQUIRE((mt == 2 || mt == 4) && N > 0 && H > 0 && W > 0 && H % mt == 0 && W % mt == 0 && C > 0 && C % 4 == 0,
                DVG_ERR_SHAPE, "dvg_winograd_output: m 2 or 4, H and W multiples of m, C %% 4 == 0 needed");
    DVG_REQUIRE(act >= 0 && act <= 3, DVG_ERR_SHAPE, "dvg_winograd_output: bad act");
    DVG_REQUIRE(aligned16(m) && aligned16(y) && aligned16(y_pool) && aligned16(scale) && aligned16(shift), DVG_ERR_ALIGN,
                "dvg_winograd_output: alignment");
    const unsigned g = wgrid((long)N * (H / mt) * (W / mt) * (C / 4));
    const hipStream_t st = (hipStream_t)stream;
    if (mt == 2 && y_pool)
        hipLaunchKernelGGL(winograd_output_kernel<true>, dim3(g), dim3(256), 0, st, m, scale, shift, y, y_pool, N, H, W, C / 4, act, slope, y_from);
    else if (mt == 2)
        hipLaunchKernelGGL(winograd_output_kernel<false>, dim3(g), dim3(256), 0, st, m, scale, shift, y, y_pool, N, H, W, C / 4, act, slope, 0);
    else {
        const long tiles = (long)N * (H / 4) * (W / 4);
#define W4OUT(POOL_, V_, CW_)                                                                                             \
    hipLaunchKernelGGL((winograd4_output_kernel<POOL_, V_>), dim3(wgrid(tiles * (C / CW_))), dim3(256), 0, st, m, scale, shift, \
                       y, y_pool, N, H, W, C / CW_, act, slope, addend, y_from)
        if (tiles * (C / 4) >= 1024L * 256) { if (y_pool) W4OUT(true, f32x4, 4); else W4OUT(false, f32x4, 4); }
        else if (tiles * (C / 2) >= 1024L * 256) { if (y_pool) W4OUT(true, f32x2, 2); else W4OUT(false, f32x2, 2); }
        else { if (y_pool) W4OUT(true, float, 1); else W4OUT(false, float, 1); }
#undef W4OUT
    }
    return check_launch("dvg_winograd_output");
}
