// Batched sparse variational GP "trigger" (gp_models.py:10-24 + the gpytorch 0.3.x
// WhitenedVariationalStrategy / GaussianLikelihood / MultivariateNormal arithmetic
// reached from train.py:225-226,283-284 and generate_frames.py:131,170,229,273,291).
// Equations of record: DESIGN.md section "GP".
//
// One workgroup per latent dimension d (D = 90 independent 1-D GPs).  Everything —
// RBF covariance assembly over [Z ; x], the MxM and BxB Cholesky factorisations,
// the triangular solves, predictive mean / covariance, the KL term and the
// reparameterised sample — stays in LDS (<= ~120 KB at M = 40, B = 128; 33 KB at
// B = 64).  The factorisations are wave-synchronous (lane = matrix row, row
// broadcasts through LDS, no workgroup barriers inside the O(n^3) loops) while the
// other three waves assemble W = L_S^T K_Zx in parallel.
#include <cstdlib>

#include "dvg_common.h"

namespace dvg {

static unsigned long long* g_gp_clk = nullptr;
static unsigned g_gp_clk_cap = 0;  // debug only (dvg_debug_set_gp_clockbuf): phase stamps per workgroup

struct GpParams {
    const float* h;         // [B][D]
    const float* z;         // [D][M]
    const float* var_mean;  // [D][M]
    const float* chol_var;  // [D][M][M]
    const float* mean_const;
    const float* outputscale;
    const float* lengthscale;
    const float* noise;  // [D] or nullptr
    const float* eps;    // [D][B] or nullptr
    float* mean;         // [D][B]
    float* var;          // [D][B]
    float* sample;       // [D][B]
    float* cov;          // [D][B][B]
    float* kl;           // [D]
    int B, D, M, train_mode;
    float jitter;
    int raw_hypers;  // outputscale / lengthscale / noise point at the RAW parameters: soft-plus (+ noise floor) in-kernel
    int Dp;          // parameter period: column d of h uses the parameters of latent dim d % Dp (D = S x Dp: the S time steps of a
                     // training closure side by side, one parameter set - no tiled copies)
    int SG;          // steps per workgroup (r05; 1 = one workgroup per column).  The points of different time steps are just more
                     // points of the SAME GP: workgroup w = g Dp + dq takes the steps [g SG, g SG + SG) of latent dim dq as ONE
                     // problem of up to SG x B points, i.e. K_ZZ, its Cholesky factor and the KL term - the B-independent three
                     // quarters of a small-batch call - once per SG steps instead of once per step.  Train-mode outputs only.
    unsigned long long* clk;  // debug only: 12 x u64 per workgroup (latent dim), for the first clk_cap workgroups
    unsigned clk_cap;
};


// ---------------------------------------------------------------------------------------
// Arithmetic type.  Both kernels are templates on T: T = double is the product path (r03) - K_ZZ carries a 1e-3 jitter on
// an RBF Gram matrix of 40 points, cond(K_ZZ) ~ 1e4..1e5, so fp32 K entries (1 ulp of expf = 6e-8) alone cost
// cond * 6e-8 ~ 2e-3 of the predictive covariance; assembled, factored, solved and cancelled (k(x,x) - A^T A) in fp64 the
// kernel meets the 1e-4 bar of BASELINE.json with two decades to spare.  Inputs and outputs stay fp32.  T = float is kept
// for shapes whose fp64 working set exceeds the 160 KB of LDS (dvg_gp_precision() tells which one a shape gets) and for
// A/B runs (until r04).  v_fma_f64 issues at the fp32 vector rate on gfx950; the kernels are LDS-latency chains.
// ---------------------------------------------------------------------------------------
__device__ __forceinline__ float fma_t(float a, float b, float c) { return fmaf(a, b, c); }
__device__ __forceinline__ double fma_t(double a, double b, double c) { return fma(a, b, c); }
__device__ __forceinline__ float exp_t(float x) { return expf(x); }
__device__ __forceinline__ double exp_t(double x) { return exp(x); }
__device__ __forceinline__ float sqrt_t(float x) { return sqrtf(x); }
__device__ __forceinline__ double sqrt_t(double x) { return sqrt(x); }
__device__ __forceinline__ float log_t(float x) { return logf(x); }
__device__ __forceinline__ double log_t(double x) { return log(x); }
__device__ __forceinline__ float log1p_t(float x) { return log1pf(x); }
__device__ __forceinline__ double log1p_t(double x) { return log1p(x); }
__device__ __forceinline__ float abs_t(float x) { return fabsf(x); }
__device__ __forceinline__ double abs_t(double x) { return fabs(x); }
__device__ __forceinline__ float max_t(float a, float b) { return fmaxf(a, b); }
__device__ __forceinline__ double max_t(double a, double b) { return fmax(a, b); }

// 1 / sqrt(x) and 1 / x.  The fp64 forms are an fp32 seed (v_rsq_f32 / v_rcp_f32, ~1 ulp) refined by two Newton steps
// (relative error 1e-7 -> 1e-14 -> below the fp64 rounding): ten FMAs instead of the ~40-instruction v_sqrt_f64 / v_div
// fix-up sequences, and these sit on the SERIAL chains of the kernels (the pivot of every Cholesky column, the diagonal
// solve of every substitution row).
__device__ __forceinline__ float rsqrt_t(float x) { return 1.f / sqrtf(x); }
__device__ __forceinline__ double rsqrt_t(double x) {
    double r = (double)__frsqrt_rn((float)x);
    const double h = 0.5 * x;
    r = r * fma(-h * r, r, 1.5);
    r = r * fma(-h * r, r, 1.5);
    return r;
}
__device__ __forceinline__ float rcp_t(float x) { return 1.f / x; }
__device__ __forceinline__ double rcp_t(double x) {
    double r = (double)__frcp_rn((float)x);
    r = r * fma(-x, r, 2.0);
    r = r * fma(-x, r, 2.0);
    return r;
}

// value of `v` in lane `src` (compile-time constant after unrolling): v_readlane_b32, not a ds_bpermute round trip
__device__ __forceinline__ float read_lane(float v, int src) {
    return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), src));
}
__device__ __forceinline__ double read_lane(double v, int src) {
    const long long b = __builtin_bit_cast(long long, v);
    const unsigned lo = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)b, src);
    const unsigned hi = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)(b >> 32), src);
    return __builtin_bit_cast(double, ((unsigned long long)hi << 32) | lo);
}

extern __shared__ __attribute__((aligned(16))) unsigned char gp_lds_raw[];

// In-place lower Cholesky of the n x n matrix A (row stride ld, n <= 128) by the WHOLE 256-thread workgroup:
// right-looking in panels of 8 columns.  The panel (8 columns, all rows below) is factored by wave 0 with dot
// products of at most 7 terms; the trailing matrix gets its rank-8 update from all 256 threads (thread = one row x
// a strided set of columns, the row's panel entries in registers, the column's broadcast from LDS; the panel
// itself is factored in registers with uniform-lane shuffles).  The
// wave-serial left-looking version both GP kernels started with spent 242 K cycles on the 64x64 predictive covariance
// and 96 K on K_ZZ - 72 % of gp_predict.
// Every thread of the workgroup must call it (it contains __syncthreads).  Only the lower triangle is written.

// PACKED: A holds only the lower triangle, row i at offset i (i + 1) / 2 (the 128 x 128 fp64 predictive covariance does not
// fit beside the other operands as a full square).  Entries above the diagonal that the panel code touches inside a
// diagonal block then alias in-bounds entries of the next row; they are read into registers the code never uses.
template <int NT, typename T, bool PACKED = false>
__device__ void block_cholesky(T* A, int n, int ld, int tid) {
    auto at = [ld](int i, int j) { return PACKED ? i * (i + 1) / 2 + j : i * ld + j; };
    constexpr int NB = 8;
    const int lane = tid & 63, wave = tid >> 6;
    for (int k0 = 0; k0 < n; k0 += NB) {
        const int kb = min(NB, n - k0);
        if (wave == 0) {
            // The panel lives in registers: lane owns rows k0+lane and k0+lane+64, a0[q] / a1[q] = A[row][k0+q].
            // Column jj: pivot = lane jj's a0[jj]; the entries L[k0+q][jj] the remaining panel columns need are
            // lane q's freshly scaled value - uniform-lane shuffles (v_readlane), no LDS round trip per column.
            const int i0 = k0 + lane, i1 = k0 + lane + 64;
            const bool v0 = i0 < n, v1 = i1 < n;
            T a0[NB], a1[NB];
#pragma unroll
            for (int q = 0; q < NB; ++q) {
                a0[q] = (v0 && q < kb) ? A[at(i0, k0 + q)] : T(0.);
                a1[q] = (v1 && q < kb) ? A[at(i1, k0 + q)] : T(0.);
            }
#pragma unroll
            for (int jj = 0; jj < NB; ++jj) {
                if (jj < kb) {
                    const T piv = max_t(read_lane(a0[jj], jj), T(1e-12));
                    const T inv = rsqrt_t(piv);
                    const T dg = piv * inv;
                    const T l0 = (lane == jj) ? dg : a0[jj] * inv;   // lanes < jj: upper triangle, never read
                    const T l1 = a1[jj] * inv;
                    a0[jj] = l0;
                    a1[jj] = l1;
#pragma unroll
                    for (int q = jj + 1; q < NB; ++q) {
                        const T lq = read_lane(l0, q);               // L[k0+q][k0+jj]
                        a0[q] = fma_t(-l0, lq, a0[q]);
                        a1[q] = fma_t(-l1, lq, a1[q]);
                    }
                }
            }
#pragma unroll
            for (int q = 0; q < NB; ++q) {
                if (v0 && q < kb && lane >= q) A[at(i0, k0 + q)] = a0[q];
                if (v1 && q < kb) A[at(i1, k0 + q)] = a1[q];
            }
        }
        __syncthreads();
        const int t0 = k0 + kb;
        if (t0 < n) {
            const int RL = (n - t0 <= 64) ? 64 : 128, G = NT / RL;
            const int i = t0 + tid % RL, g = tid / RL;
            if (i < n) {
                T ai[NB];
#pragma unroll
                for (int q = 0; q < NB; ++q) ai[q] = q < kb ? A[at(i, k0 + q)] : T(0.);
                int j = t0 + g;
                for (; j + 3 * G <= i; j += 4 * G) {     // four columns per trip: their LDS reads issue together
                    T dot[4] = {T(0.), T(0.), T(0.), T(0.)}, cur[4];
#pragma unroll
                    for (int u = 0; u < 4; ++u) {
                        cur[u] = A[at(i, j + u * G)];
#pragma unroll
                        for (int q = 0; q < NB; ++q) dot[u] = fma_t(ai[q], q < kb ? A[at(j + u * G, k0 + q)] : T(0.), dot[u]);
                    }
#pragma unroll
                    for (int u = 0; u < 4; ++u) A[at(i, j + u * G)] = cur[u] - dot[u];
                }
                for (; j <= i; j += G) {
                    T dot = T(0.);
#pragma unroll
                    for (int q = 0; q < NB; ++q) dot = fma_t(ai[q], q < kb ? A[at(j, k0 + q)] : T(0.), dot);
                    A[at(i, j)] -= dot;
                }
            }
        }
        __syncthreads();
    }
}

template <int NT, typename T>
__device__ __forceinline__ T block_sum(T v, T* scratch, int tid) {
    // NT threads; scratch >= NT / 64 floats; fixed summation order
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) v += __shfl_xor(v, o);
    __syncthreads();
    if ((tid & 63) == 0) scratch[tid >> 6] = v;
    __syncthreads();
    T t = T(0.);
#pragma unroll
    for (int w = 0; w < NT / 64; ++w) t += scratch[w];
    return t;
}

// Blocked forward substitution L X = R in place (R = X [M][ncol], row stride LB; L lower, row stride LM): per block of 8
// rows one thread per column solves the 8x8 triangle, then all 256 threads subtract the block's contribution from the
// rows below.  One thread per column walking all M rows serially took 49 K cycles (M = 40, 65 columns).
// Every thread of the workgroup must call it (it contains __syncthreads); ends with a barrier.
template <int NT, typename T>
__device__ void block_forward_subst(const T* L, int LM, T* X, int LB, int M, int ncol, int tid) {
    constexpr int RB = 8;
    for (int r0 = 0; r0 < M; r0 += RB) {
        const int rb = min(RB, M - r0);
        for (int col = tid; col < ncol; col += NT) {
            T xv[RB];
#pragma unroll
            for (int a = 0; a < RB; ++a) {
                xv[a] = T(0.);
                if (a < rb) {
                    T acc = X[(r0 + a) * LB + col];
#pragma unroll
                    for (int q = 0; q < RB; ++q)
                        if (q < a) acc = fma_t(-L[(r0 + a) * LM + r0 + q], xv[q], acc);
                    xv[a] = acc * rcp_t(L[(r0 + a) * LM + r0 + a]);
                    X[(r0 + a) * LB + col] = xv[a];
                }
            }
        }
        __syncthreads();
        const int below = M - (r0 + rb);
        for (int e = tid; e < below * ncol; e += NT) {
            const int i = r0 + rb + e / ncol, col = e % ncol;
            T acc = X[i * LB + col];
#pragma unroll
            for (int q = 0; q < RB; ++q)
                if (q < rb) acc = fma_t(-L[i * LM + r0 + q], X[(r0 + q) * LB + col], acc);
            X[i * LB + col] = acc;
        }
        __syncthreads();
    }
}

// Blocked backward substitution L^T X = R in place, the mirror image: row blocks from the bottom, the 8x8 triangle of a
// block is the transpose of L's diagonal block, rows ABOVE the block get its contribution.
template <int NT, typename T>
__device__ void block_backward_subst(const T* L, int LM, T* X, int LB, int M, int ncol, int tid) {
    constexpr int RB = 8;
    for (int r1 = M; r1 > 0; r1 -= RB) {
        const int r0 = max(0, r1 - RB), rb = r1 - r0;
        for (int col = tid; col < ncol; col += NT) {
            T xv[RB];
#pragma unroll
            for (int a = RB - 1; a >= 0; --a) {
                xv[a] = T(0.);
                if (a < rb) {
                    T acc = X[(r0 + a) * LB + col];
#pragma unroll
                    for (int q = 0; q < RB; ++q)
                        if (q > a && q < rb) acc = fma_t(-L[(r0 + q) * LM + r0 + a], xv[q], acc);
                    xv[a] = acc * rcp_t(L[(r0 + a) * LM + r0 + a]);
                    X[(r0 + a) * LB + col] = xv[a];
                }
            }
        }
        __syncthreads();
        for (int e = tid; e < r0 * ncol; e += NT) {
            const int i = e / ncol, col = e % ncol;
            T acc = X[i * LB + col];
#pragma unroll
            for (int q = 0; q < RB; ++q)
                if (q < rb) acc = fma_t(-L[(r0 + q) * LM + i], X[(r0 + q) * LB + col], acc);
            X[i * LB + col] = acc;
        }
        __syncthreads();
    }
}

template <int NT, typename T, bool PACKED = false>
__global__ __launch_bounds__(NT) void gp_predict_kernel(const GpParams p) {
    T* sm = reinterpret_cast<T*>(gp_lds_raw);
    const int wg = blockIdx.x, tid = threadIdx.x;
    const int dq = wg % p.Dp;                       // the latent dim whose PARAMETERS this workgroup reads
    const int s0 = (wg / p.Dp) * p.SG;              // its first time step (SG = 1, Dp = D: step 0 of column wg)
    const int ns = min(p.SG, p.D / p.Dp - s0);      // its steps
    const int Bs = p.B, M = p.M, B = ns * Bs;       // B: the POINTS of this workgroup, point b = (step s0 + b / Bs, sample b % Bs)
    const int d = s0 * p.Dp + dq;                   // its first column of h (the only one when SG = 1)
    auto col_of = [&](int b) { return d + (b / Bs) * p.Dp; };
    const int LM = M + 1;   // row stride of the MxM matrices
    const int LB = B + 2;   // row stride of the M x (B+1) matrices
    const int LS = B + 1;   // row stride of Sigma
    T* L = sm;                 // [M][LM]   K_ZZ + jitter -> its Cholesky factor
    T* Ls = L + M * LM;        // [M][LM]   variational Cholesky factor (lower)
    // PACKED: the covariance triangle (B (B + 1) / 2 entries) will overlay L and L_S, so the operands it is built from start
    // behind whichever of the two regions is larger (host side: gp_predict_elems_packed)
    T* AK = PACKED ? sm + max(2 * M * LM, B * (B + 1) / 2) : Ls + M * LM;   // [M][LB]   [K_Zx | m-c] -> L^-1 [K_Zx | m-c]
    T* Wm = AK + M * LB;       // [M][LB]   L_S^T K_Zx
    T* zs = Wm + M * LB;       // [M]
    T* xs = zs + M;            // [B]
    T* mu = xs + B;            // [B]
    T* red = mu + B;           // [16]
    // predictive covariance (only when needed): [B][LS] behind the other operands, or - PACKED - its lower triangle on top
    // of L and L_S, which are dead once the mean / variance / KL phase has passed the barrier in front of the build
    T* Sg = PACKED ? sm : red + 16;
    auto sg = [LS](int r, int q) { return PACKED ? r * (r + 1) / 2 + q : r * LS + q; };

    // torch.nn.functional.softplus (beta 1, threshold 20) of the raw parameters when the caller passes them as they are
    // (gp_models.py hyper-parameters / GaussianLikelihood noise with its GreaterThan(1e-4) floor): saves three
    // 90-element launches per GP call
    auto softplus = [](T x) { return x > T(20.) ? x : log1p_t(exp_t(x)); };
    const T s = p.raw_hypers ? softplus(p.outputscale[dq]) : p.outputscale[dq];
    const T ell = p.raw_hypers ? softplus(p.lengthscale[dq]) : p.lengthscale[dq];
    const T ninv = -T(0.5) / (ell * ell);
    const T c0 = p.mean_const[dq];
    const T noise = p.noise ? (p.raw_hypers ? softplus(p.noise[dq]) + T(1e-4) : p.noise[dq]) : T(0.);
    const bool need_cov = (p.cov != nullptr) || (p.sample != nullptr);

    if (p.clk && tid == 0 && (unsigned)wg < p.clk_cap) p.clk[(size_t)wg * 12 + 0] = clock64();
    for (int i = tid; i < M; i += NT) zs[i] = p.z[(size_t)dq * M + i];
    for (int b = tid; b < B; b += NT) xs[b] = p.h[(size_t)(b % Bs) * p.D + col_of(b)];
    __syncthreads();
    for (int i = tid; i < M * M; i += NT) {
        const int r = i / M, q = i % M;
        const T dz = zs[r] - zs[q];
        L[r * LM + q] = s * exp_t(dz * dz * ninv) + (r == q ? p.jitter : T(0.));
        Ls[r * LM + q] = (q <= r) ? p.chol_var[((size_t)dq * M + r) * M + q] : T(0.);
    }
    for (int i = tid; i < M * (B + 1); i += NT) {
        const int r = i / (B + 1), b = i % (B + 1);
        T v;
        if (b < B) {
            const T dx = zs[r] - xs[b];
            v = s * exp_t(dx * dx * ninv);
        } else {
            v = p.var_mean[(size_t)dq * M + r] - c0;
        }
        AK[r * LB + b] = v;
    }
    __syncthreads();

    if (p.clk && tid == 0 && (unsigned)wg < p.clk_cap) p.clk[(size_t)wg * 12 + 1] = clock64();
    // W = L_S^T K_Zx (needs the un-solved K_Zx), then chol(K_ZZ) by the whole workgroup
    for (int i = tid; i < M * B; i += NT) {
        const int r = i / B, b = i % B;
        T acc = T(0.);
        for (int j = r; j < M; ++j) acc = fma_t(Ls[j * LM + r], AK[j * LB + b], acc);
        Wm[r * LB + b] = acc;
    }
    block_cholesky<NT, T>(L, M, LM, tid);   // ends with __syncthreads
    if (p.clk && tid == 0 && (unsigned)wg < p.clk_cap) p.clk[(size_t)wg * 12 + 2] = clock64();

    // L X = [K_Zx | m-c]
    block_forward_subst<NT, T>(L, LM, AK, LB, M, B + 1, tid);
    if (p.clk && tid == 0 && (unsigned)wg < p.clk_cap) p.clk[(size_t)wg * 12 + 3] = clock64();

    // predictive mean and marginal variance
    for (int b = tid; b < B; b += NT) {
        T m = T(0.), qa = T(0.), qw = T(0.);
        for (int i = 0; i < M; ++i) {
            const T a = AK[i * LB + b], w = Wm[i * LB + b];
            m = fma_t(a, AK[i * LB + B], m);
            qa = fma_t(a, a, qa);
            qw = fma_t(w, w, qw);
        }
        m += c0;
        mu[b] = m;
        const size_t ob = (size_t)col_of(b) * Bs + b % Bs;
        if (p.mean) p.mean[ob] = m;
        if (p.var) {
            T dd = s - qa;
            if (p.train_mode) dd = max_t(dd, T(0.));
            p.var[ob] = qw + dd + noise;
        }
    }

    if (p.kl != nullptr) {
        // KL(q(u)||p(u)) = 0.5 [ -log|K| - log|S'| + tr(S'K) + (m-c)^T K^-1 (m-c) - M ],
        // tr(S'K) = || L^T L_S ||_F^2,  (m-c)^T K^-1 (m-c) = || L^-1 (m-c) ||^2
        T part = T(0.);
        for (int i = tid; i < M * M; i += NT) {
            const int r = i / M, q = i % M;
            T acc = T(0.);
            for (int k = (r > q ? r : q); k < M; ++k) acc = fma_t(L[k * LM + r], Ls[k * LM + q], acc);
            part = fma_t(acc, acc, part);
        }
        for (int i = tid; i < M; i += NT) {
            const T v = AK[i * LB + B];
            part = fma_t(v, v, part);
            part -= T(2.) * log_t(L[i * LM + i]);
            part -= T(2.) * log_t(abs_t(Ls[i * LM + i]));
        }
        const T tot = block_sum<NT, T>(part, red, tid);
        if (tid < ns) p.kl[d + tid * p.Dp] = T(0.5) * (tot - (T)M);    // the same value for every step of the workgroup
    }

    if (need_cov) {
        __syncthreads();
        if (p.clk && tid == 0 && (unsigned)wg < p.clk_cap) p.clk[(size_t)wg * 12 + 4] = clock64();
        // Sigma = W^T W - A^T A + k(x,x) (+ noise I): 4x4 register tiles (16 LDS reads per 32 FMAs instead of 4 per 2),
        // lower-triangular tiles only, mirrored on store
        {
            const int nt = (B + 3) / 4;
            for (int t = tid; t < nt * nt; t += NT) {
                const int tr = t / nt, tq = t % nt;
                if (tq > tr) continue;
                T acc[4][4];
#pragma unroll
                for (int a = 0; a < 4; ++a)
#pragma unroll
                    for (int b2 = 0; b2 < 4; ++b2) acc[a][b2] = T(0.);
                int rr[4], qq[4];
#pragma unroll
                for (int a = 0; a < 4; ++a) {
                    rr[a] = min(tr * 4 + a, B - 1);
                    qq[a] = min(tq * 4 + a, B - 1);
                }
                for (int k = 0; k < M; ++k) {
                    T wr[4], wq[4], ar[4], aq[4];
#pragma unroll
                    for (int a = 0; a < 4; ++a) {
                        wr[a] = Wm[k * LB + rr[a]];
                        wq[a] = Wm[k * LB + qq[a]];
                        ar[a] = AK[k * LB + rr[a]];
                        aq[a] = AK[k * LB + qq[a]];
                    }
#pragma unroll
                    for (int a = 0; a < 4; ++a)
#pragma unroll
                        for (int b2 = 0; b2 < 4; ++b2) acc[a][b2] = fma_t(wr[a], wq[b2], fma_t(-ar[a], aq[b2], acc[a][b2]));
                }
#pragma unroll
                for (int a = 0; a < 4; ++a)
#pragma unroll
                    for (int b2 = 0; b2 < 4; ++b2) {
                        const int r = tr * 4 + a, q = tq * 4 + b2;
                        if (r < B && q < B) {
                            const T dx = xs[r] - xs[q];
                            T v = acc[a][b2] + s * exp_t(dx * dx * ninv);
                            if (r == q) v += noise;
                            if (PACKED) {
                                Sg[sg(max(r, q), min(r, q))] = v;
                            } else {
                                Sg[sg(r, q)] = v;
                                Sg[sg(q, r)] = v;
                            }
                            if (p.cov) {
                                p.cov[((size_t)d * B + r) * B + q] = v;
                                p.cov[((size_t)d * B + q) * B + r] = v;
                            }
                        }
                    }
            }
        }
        __syncthreads();
        if (p.clk && tid == 0 && (unsigned)wg < p.clk_cap) p.clk[(size_t)wg * 12 + 5] = clock64();
        if (p.sample != nullptr) {
            block_cholesky<NT, T, PACKED>(Sg, B, LS, tid);   // ends with __syncthreads
        if (p.clk && tid == 0 && (unsigned)wg < p.clk_cap) p.clk[(size_t)wg * 12 + 6] = clock64();
            for (int b = tid; b < B; b += NT) {
                T acc = mu[b];
#pragma unroll 4
                for (int j = 0; j <= b; ++j) acc = fma_t(Sg[sg(b, j)], (T)p.eps[(size_t)d * B + j], acc);
                p.sample[(size_t)d * B + b] = acc;
            }
        }
    }
    if (p.clk && tid == 0 && (unsigned)wg < p.clk_cap) p.clk[(size_t)wg * 12 + 7] = clock64();
}


// ---------------------------------------------------------------------------------------
// Train-mode backward.  Forward (see gp_predict_kernel, train_mode = 1):
//   K = Kzz + jitter I,  alpha = K^-1 (m - c),  P = K^-1 Kzx,  W = L_S^T Kzx
//   mean_b = c + Kzx[:,b].alpha ;  q_b = Kzx[:,b].P[:,b] ;  var_b = |W[:,b]|^2 + max(s - q_b, 0)
//   KL = 1/2 [ -log|K| - log|L_S L_S^T| + tr(L_S L_S^T K) + (m-c).alpha - M ]
// Given gm = dL/dmean, gv = dL/dvar, gk = dL/dKL it returns dL/d{h column, z, m, L_S, c, s, ell}.
// With tau = K^-1 Kzx gm, gq = -gv*[s-q>0], GW = 2 gv W:
//   G2 = dL/dKzx = alpha gm^T + 2 P diag(gq) + L_S GW
//   GK = dL/dK   = -tau alpha^T - P diag(gq) P^T + gk/2 (-K^-1 + L_S L_S^T - alpha alpha^T)
//   dL_S = tril( Kzx GW^T + gk (K L_S - diag(1/L_S_ii)) ),  dm = tau + gk alpha,  dc = sum gm - sum dm
// and the RBF chain rule maps (GK, G2) onto z, x, s, ell.  K^-1 is applied by blocked triangular solves with L; the
// explicit K^-1 of the KL trace term comes out of the same solves (identity columns appended to the right-hand sides).
// ---------------------------------------------------------------------------------------
struct GpBwdParams {
    const float* h; const float* z; const float* var_mean; const float* chol_var;
    const float* mean_const; const float* outputscale; const float* lengthscale;
    const float* gmean; const float* gvar; const float* gkl;   // upstream gradients (any may be nullptr)
    float* dh; float* dz; float* dm; float* dls; float* dc; float* ds; float* dell;
    int B, D, M;
    int Bc;                   // data points per chunk (gp_bwd_chunk): B when the whole fp64 working set fits the LDS
    int Dp;                   // parameter period (see GpParams)
    int SG;                   // steps per workgroup (see GpParams): parameter gradients are written per WORKGROUP (ceil(S / SG) x Dp
                              // rows), summed over the workgroup's steps
    float jitter;
    unsigned long long* clk;  // debug only (dvg_debug_set_gp_clockbuf): 12 x u64 phase stamps per workgroup
    unsigned clk_cap;
};

#define GP_STAMP(k) if (p.clk && tid == 0 && (unsigned)wg < p.clk_cap) p.clk[(size_t)wg * 12 + (k)] = clock64();

// Data points in CHUNKS of Bc (r04): everything that has a B-wide row - Kzx, W / GW, G2 and the solved columns P = K^-1 Kzx - is
// held for Bc points at a time, so that the fp64 working set fits the 160 KB of LDS up to B = 128 (M = 40: two chunks of 64,
// 155.6 KB; until r03 B > 71 fell back to fp32 arithmetic and a 5e-4 / 2e-3 bar).  What couples the points of a latent dim
// is small and is kept whole: tt = Kzx gm (a first pass over the chunks), alpha, tau, K^-1 (solved once from the columns
// [m - c | tt | I]), and the running sums GK (LDS), dL_S (accumulated in the output buffer by the thread that owns the entry),
// dz / ds / dell (registers).  With one chunk (B <= 71 at M = 40) the arithmetic is the r03 kernel's.
template <int NT, typename T>
__global__ __launch_bounds__(NT) void gp_train_bwd_kernel(const GpBwdParams p) {
    T* sm = reinterpret_cast<T*>(gp_lds_raw);
    const int wg = blockIdx.x, tid = threadIdx.x;
    const int dq = wg % p.Dp;                       // the latent dim whose PARAMETERS this workgroup reads
    const int s0 = (wg / p.Dp) * p.SG;              // first time step, steps and points of this workgroup (see gp_predict_kernel)
    const int ns = min(p.SG, p.D / p.Dp - s0);
    const int Bs = p.B, M = p.M, B = ns * Bs, Bc = p.Bc, LM = M + 1, LB = Bc + 2;
    const int d = s0 * p.Dp + dq;
    auto col_of = [&](int b) { return d + (b / Bs) * p.Dp; };
    // ONE chunk (Bc == B): P = K^-1 [Kzx | m-c | Kzx gm | I] comes out of one pair of triangular solves and K^-1 is read in
    // place from its last M columns (the r03 layout and cost).  Several chunks: the M + 2 columns that couple the points are
    // solved first, K^-1 is copied out to a buffer of its own, and P then holds K^-1 Kzx of one chunk at a time.
    const bool single = Bc >= p.SG * Bs;       // launch-uniform (a ragged last group keeps the launch's layout)
    const int LP = single ? B + M + 3 : (Bc > M + 2 ? Bc : M + 2) + 1;
    T* Kj = sm;               // [M][LM] K with jitter
    T* L = Kj + M * LM;       // chol(K)
    T* Ls = L + M * LM;       // variational factor
    T* GK = Ls + M * LM;      // dL/dK
    T* Kib = GK + M * LM;     // K^-1 (several chunks only)
    T* Kzx = Kib + (single ? 0 : M * LM);     // [M][LB]  (one chunk)
    T* Wm = Kzx + M * LB;     // W, then GW
    T* G2 = Wm + M * LB;      // dL/dKzx
    T* P = G2 + M * LB;       // [M][LP]
    T* zs = P + M * LP;       // [M]
    T* rr = zs + M;           // m - c
    T* al = rr + M;           // alpha
    T* tt = al + M;           // Kzx gm
    T* tau = tt + M;          // K^-1 tt
    T* xs = tau + M;          // [B]
    T* gm = xs + B;
    T* gv = gm + B;
    T* gq = gv + B;
    T* red = gq + B;          // [16]

    const T s = p.outputscale[dq], ell = p.lengthscale[dq];
    const T ninv = -T(0.5) / (ell * ell), c0 = p.mean_const[dq];
    T gk = T(0.);               // the KL value is shared by the workgroup's steps: its upstream gradients add up
    if (p.gkl)
        for (int st = 0; st < ns; ++st) gk += (T)p.gkl[d + st * p.Dp];
    const int nchunk = (B + Bc - 1) / Bc;

    GP_STAMP(0)
    for (int i = tid; i < M; i += NT) {
        zs[i] = p.z[(size_t)dq * M + i];
        rr[i] = p.var_mean[(size_t)dq * M + i] - c0;
        tt[i] = T(0.);
    }
    for (int b = tid; b < B; b += NT) {
        const int cb = col_of(b);
        xs[b] = p.h[(size_t)(b % Bs) * p.D + cb];
        gm[b] = p.gmean ? p.gmean[(size_t)cb * Bs + b % Bs] : T(0.);
        gv[b] = p.gvar ? p.gvar[(size_t)cb * Bs + b % Bs] : T(0.);
    }
    __syncthreads();
    for (int i = tid; i < M * M; i += NT) {
        const int r = i / M, q = i % M;
        const T dz = zs[r] - zs[q];
        const T v = s * exp_t(dz * dz * ninv) + (r == q ? p.jitter : T(0.));
        Kj[r * LM + q] = v;
        L[r * LM + q] = v;
        Ls[r * LM + q] = (q <= r) ? p.chol_var[((size_t)dq * M + r) * M + q] : T(0.);
    }
    auto build_kzx = [&](int b0, int bc) {       // Kzx of the chunk [b0, b0 + bc)
        for (int i = tid; i < M * bc; i += NT) {
            const int r = i / bc, b = i % bc;
            const T dx = zs[r] - xs[b0 + b];
            Kzx[r * LB + b] = s * exp_t(dx * dx * ninv);
        }
    };
    const int c_fix = single ? B : 0;               // first of the M + 2 coupling columns of P
    const T* Ki = single ? P + B + 2 : Kib;         // K^-1 and its row stride
    const int LKi = single ? LP : LM;
    // tt = Kzx gm over ALL points (the last chunk's Kzx stays in LDS: with one chunk nothing is built twice)
    for (int ch = 0; ch < nchunk; ++ch) {
        const int b0 = ch * Bc, bc = min(Bc, B - b0);
        __syncthreads();
        build_kzx(b0, bc);
        __syncthreads();
        for (int i = tid; i < M; i += NT) {
            T acc = tt[i];
            for (int b = 0; b < bc; ++b) acc = fma_t(Kzx[i * LB + b], gm[b0 + b], acc);
            tt[i] = acc;
        }
    }
    __syncthreads();
    GP_STAMP(1)
    GP_STAMP(2)
    block_cholesky<NT, T>(L, M, LM, tid);   // ends with __syncthreads
    GP_STAMP(3)
    // P <- [m-c | Kzx gm | I]; K^-1 applied column-wise by two blocked TRIANGULAR solves with L
    // (an explicit fp32 K^-1 from L^-1 loses ~cond(K)*eps = 1e-3 and the c / s gradients cancel to 1e-2 of it).  The
    // identity columns give K^-1 itself, which only the KL trace term -gk/2 K^-1 needs.
    for (int i = tid; i < M * (M + 2); i += NT) {
        const int r = i / (M + 2), b = i % (M + 2);
        P[r * LP + c_fix + b] = b == 0 ? rr[r] : (b == 1 ? tt[r] : (b - 2 == r ? T(1.) : T(0.)));
    }
    if (single)
        for (int i = tid; i < M * B; i += NT) P[(i / B) * LP + i % B] = Kzx[(i / B) * LB + i % B];
    __syncthreads();
    block_forward_subst<NT, T>(L, LM, P, LP, M, single ? B + M + 2 : M + 2, tid);
    block_backward_subst<NT, T>(L, LM, P, LP, M, single ? B + M + 2 : M + 2, tid);
    for (int i = tid; i < M; i += NT) {
        al[i] = P[i * LP + c_fix];
        tau[i] = P[i * LP + c_fix + 1];
    }
    if (!single)
        for (int i = tid; i < M * M; i += NT) Kib[(i / M) * LM + i % M] = P[(i / M) * LP + 2 + i % M];
    __syncthreads();
    GP_STAMP(4)
    // the parts of GK and dL_S that do not involve the data points
    for (int i = tid; i < M * M; i += NT) {
        const int r = i / M, q = i % M;
        T sp = T(0.);
        const int kmax = r < q ? r : q;
        for (int k = 0; k <= kmax; ++k) sp = fma_t(Ls[r * LM + k], Ls[q * LM + k], sp);
        GK[r * LM + q] = -tau[r] * al[q] + T(0.5) * gk * (-Ki[r * LKi + q] + sp - al[r] * al[q]);
    }
    // dL_S: entry (r, q) belongs to ONE thread (i = r M + q = tid + it NT), which keeps its running sum in a register
    constexpr int EPT = (64 * 64 + NT - 1) / NT;      // entries per thread at the largest M
    T dls_acc[EPT];
#pragma unroll
    for (int it = 0; it < EPT; ++it) {
        const int i = tid + it * NT;
        T g = T(0.);
        if (i < M * M) {
            const int r = i / M, q = i % M;
            if (q <= r) {
                T kl = (r == q) ? -T(1.) / Ls[r * LM + r] : T(0.);
                for (int k = q; k < M; ++k) kl = fma_t(Kj[r * LM + k], Ls[k * LM + q], kl);
                g = gk * kl;
            }
        }
        dls_acc[it] = g;
    }
    GP_STAMP(5)
    T ds_part = T(0.), ds_acc = T(0.), dl_acc = T(0.), dz_acc = T(0.);
    const T il2 = T(1.) / (ell * ell), il3 = il2 / ell;
    for (int ch = nchunk - 1; ch >= 0; --ch) {      // the last chunk first: its Kzx is still in LDS
        const int b0 = ch * Bc, bc = min(Bc, B - b0);
        if (ch != nchunk - 1) {
            __syncthreads();
            build_kzx(b0, bc);
        }
        __syncthreads();
        // W = L_S^T Kzx by all threads; P <- Kzx, then K^-1 Kzx by the two triangular solves
        for (int i = tid; i < M * bc; i += NT) {
            const int r = i / bc, b = i % bc;
            T acc = T(0.);
            for (int j = r; j < M; ++j) acc = fma_t(Ls[j * LM + r], Kzx[j * LB + b], acc);
            Wm[r * LB + b] = acc;
            if (!single) P[r * LP + b] = Kzx[r * LB + b];
        }
        __syncthreads();
        if (!single) {       // (one chunk: P's first B columns already are K^-1 Kzx)
            block_forward_subst<NT, T>(L, LM, P, LP, M, bc, tid);
            block_backward_subst<NT, T>(L, LM, P, LP, M, bc, tid);
        }
        for (int b = tid; b < bc; b += NT) {
            T q = T(0.);
            for (int i = 0; i < M; ++i) q = fma_t(Kzx[i * LB + b], P[i * LP + b], q);
            const T mask = (s - q > T(0.)) ? T(1.) : T(0.);
            gq[b0 + b] = -gv[b0 + b] * mask;
            ds_part += gv[b0 + b] * mask;
        }
        __syncthreads();
        GP_STAMP(6)
        for (int i = tid; i < M * bc; i += NT) {   // GW = 2 gv W (in place)
            const int r = i / bc, b = i % bc;
            Wm[r * LB + b] *= T(2.) * gv[b0 + b];
        }
        __syncthreads();
        for (int i = tid; i < M * bc; i += NT) {
            const int r = i / bc, b = i % bc;
            T acc = gm[b0 + b] * al[r] + T(2.) * gq[b0 + b] * P[r * LP + b];
            for (int k = 0; k <= r; ++k) acc = fma_t(Ls[r * LM + k], Wm[k * LB + b], acc);
            G2[r * LB + b] = acc;
        }
        for (int i = tid; i < M * M; i += NT) {     // the same thread owns entry (r, q) in every chunk
            const int r = i / M, q = i % M;
            T acc = GK[r * LM + q];
            for (int b = 0; b < bc; ++b) acc = fma_t(-gq[b0 + b] * P[r * LP + b], P[q * LP + b], acc);
            GK[r * LM + q] = acc;
        }
#pragma unroll
        for (int it = 0; it < EPT; ++it) {
            const int i = tid + it * NT;
            if (i < M * M && i % M <= i / M) {
                const int r = i / M, q = i % M;
                T g = dls_acc[it];
                for (int b = 0; b < bc; ++b) g = fma_t(Kzx[r * LB + b], Wm[q * LB + b], g);
                dls_acc[it] = g;
            }
        }
        __syncthreads();
        // RBF chain rule, the terms over this chunk's points
        for (int i = tid; i < M * bc; i += NT) {
            const int r = i / bc, b = i % bc;
            const T dx = zs[r] - xs[b0 + b];
            const T g = G2[r * LB + b] * Kzx[r * LB + b];
            ds_acc += g;
            dl_acc = fma_t(g, dx * dx, dl_acc);
        }
        for (int i = tid; i < M; i += NT) {
            T acc = T(0.);
            for (int b = 0; b < bc; ++b) acc = fma_t(G2[i * LB + b] * Kzx[i * LB + b], -(zs[i] - xs[b0 + b]), acc);
            dz_acc += acc;
        }
        for (int b = tid; b < bc; b += NT) {
            T acc = T(0.);
            for (int i = 0; i < M; ++i) acc = fma_t(G2[i * LB + b] * Kzx[i * LB + b], zs[i] - xs[b0 + b], acc);
            p.dh[(size_t)((b0 + b) % Bs) * p.D + col_of(b0 + b)] = acc * il2;
        }
    }
    __syncthreads();
#pragma unroll
    for (int it = 0; it < EPT; ++it) {
        const int i = tid + it * NT;
        if (i < M * M) p.dls[(size_t)wg * M * M + i] = dls_acc[it];
    }
    T dc_part = T(0.);
    for (int i = tid; i < M; i += NT) {
        const T dr = tau[i] + gk * al[i];
        p.dm[(size_t)wg * M + i] = dr;
        dc_part -= dr;
    }
    for (int b = tid; b < B; b += NT) dc_part += gm[b];
    GP_STAMP(7)
    // RBF chain rule, the terms over the inducing points (GK is complete)
    for (int i = tid; i < M * M; i += NT) {
        const int r = i / M, q = i % M;
        const T kp = Kj[r * LM + q] - (r == q ? p.jitter : T(0.));
        const T dz = zs[r] - zs[q];
        const T g = GK[r * LM + q] * kp;
        ds_acc += g;
        dl_acc = fma_t(g, dz * dz, dl_acc);
    }
    for (int i = tid; i < M; i += NT) {
        T acc = dz_acc;
        for (int j = 0; j < M; ++j) {
            const T kp = Kj[i * LM + j] - (i == j ? p.jitter : T(0.));
            acc = fma_t((GK[i * LM + j] + GK[j * LM + i]) * kp, -(zs[i] - zs[j]), acc);
        }
        p.dz[(size_t)wg * M + i] = acc * il2;
    }
    GP_STAMP(8)
    const T ds_tot = block_sum<NT, T>(ds_acc, red, tid);
    const T dl_tot = block_sum<NT, T>(dl_acc, red, tid);
    const T dc_tot = block_sum<NT, T>(dc_part, red, tid);
    const T dsdir_tot = block_sum<NT, T>(ds_part, red, tid);
    if (tid == 0) {
        p.ds[wg] = ds_tot / s + dsdir_tot;
        p.dell[wg] = dl_tot * il3;
        p.dc[wg] = dc_tot;
    }
    GP_STAMP(9)
}

// ---------------------------------------------------------------------------------------
// VariationalELBO(likelihood, gp, num_data, combine_terms=True)(pred, target) (train.py:112,164-169,225-226) as one launch
// forward and one backward, instead of ~25 torch launches on (D,B) tensors per GP call:
//   sig2_d = softplus(raw_noise_d) + 1e-4                       (GaussianLikelihood, noise floor GreaterThan(1e-4))
//   elbo_d = (1/B) sum_b [ -((y_db - mean_db)^2 + var_db) / (2 sig2_d) - log(sig2_d)/2 - log(2 pi)/2 ] - KL_d / num_data
// One workgroup per latent dim; target (D,B) with arbitrary strides (the reference hands over h_target.transpose(0,1)).
// ---------------------------------------------------------------------------------------
struct GpElboParams {
    const float* mean; const float* var; const float* kl; const float* target; long t_sd, t_sb;
    const float* raw_noise; const float* gelbo;
    float* elbo; float* gmean; float* gvar; float* gkl; float* gtarget; float* graw_noise;
    int B, D; float inv_num_data;
    int Dp;      // noise period: workgroup d reads raw_noise[d % Dp] (graw_noise is still written per d)
};

__device__ __forceinline__ float softplus_f(float x) { return x > 20.f ? x : log1pf(expf(x)); }

template <bool BWD>
__global__ __launch_bounds__(256) void gp_elbo_kernel(const GpElboParams p) {
    __shared__ float red[4];
    const int d = blockIdx.x, tid = threadIdx.x, B = p.B;
    const float raw = p.raw_noise[d % p.Dp];
    const float nz = softplus_f(raw) + 1e-4f;
    float acc = 0.f;
    for (int b = tid; b < B; b += 256) {
        const float r = p.target[d * p.t_sd + b * p.t_sb] - p.mean[(size_t)d * B + b];
        acc += fmaf(r, r, p.var[(size_t)d * B + b]);
    }
    const float tot = block_sum<256, float>(acc, red, tid);     // fixed order: deterministic
    if (!BWD) {
        if (tid == 0)
            p.elbo[d] = -0.5f * tot / (nz * (float)B) - 0.5f * logf(nz) - 0.9189385332046727f - p.kl[d] * p.inv_num_data;
        return;
    }
    const float g = p.gelbo[d];
    const float k = g / (nz * (float)B);
    for (int b = tid; b < B; b += 256) {
        const float r = p.target[d * p.t_sd + b * p.t_sb] - p.mean[(size_t)d * B + b];
        p.gmean[(size_t)d * B + b] = k * r;
        p.gvar[(size_t)d * B + b] = -0.5f * k;
        if (p.gtarget) p.gtarget[(size_t)d * B + b] = -k * r;
    }
    if (tid == 0) {
        p.gkl[d] = -g * p.inv_num_data;
        const float dnz = 0.5f * tot / (nz * nz * (float)B) - 0.5f / nz;
        const float sig = raw > 20.f ? 1.f : 1.f / (1.f + expf(-raw));     // d softplus / d raw
        p.graw_noise[d] = g * dnz * sig;
    }
}

}  // namespace dvg

using namespace dvg;

// Threads per workgroup (= per latent dim).  One workgroup per CU and every phase a chain of LDS round trips: more waves
// per SIMD are the only latency hiding there is (tools/diag_gp_bwd.py, tools/bench_gp.py; -DGP_PREDICT_THREADS / -DGP_BWD_THREADS variant builds
// overrides both kernels for A/B runs).
#ifndef GP_PREDICT_THREADS
#define GP_PREDICT_THREADS 1024
#endif
#ifndef GP_BWD_THREADS
#define GP_BWD_THREADS 1024
#endif
// Threads per workgroup by shape (r05, tools/bench_gp.py on lib variants built with -DGP_*_THREADS=...; us per launch):
//   one GP call (D = 90 workgroups < 256 CUs): 1024 threads is fastest or tied for every kernel (latency of one workgroup);
//   the S steps of a training closure side by side (D = S x 90 >= 990 workgroups, several rounds): train-mode predict
//   B=16,S=11 171 (1024) / 152 (512) / 105 (256);  B=4,S=15 245 / 219 / 140;  B=64,S=19 322 / 322 / 242 -> 256 threads: its 38-70 KB
//   of LDS let 2-4 small workgroups share a CU and overlap their serial Cholesky chains;
//   backward (75-154 KB of LDS: one workgroup per CU whatever its size) B=16 379 / 342 / 420;  B=4 505 / 458 / 538;  B=64
//   851 / 860 / 1134 -> 512 threads up to B = 32, else 1024.
static int gp_threads(int dflt, int D, int B, bool bwd) {
    if (D < 512) return dflt;
    if (!bwd) return 256;
    return B <= 32 ? 512 : dflt;
}
static bool gp_force_fp32() { return false; }        // fp32 variants only run where the fp64 working set does not fit the LDS
static constexpr size_t GP_LDS_MAX = 160 * 1024;

template <typename K, typename P>
static int gp_launch(K kernel, int NT, const P& p, int D, size_t lds, void* stream, const char* who) {
    // hipFuncSetAttribute is idempotent and cheap; it is legal during stream capture (not a stream operation)
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                                       (int)lds);
    if (e != hipSuccess) return fail(DVG_ERR_HIP, "hipFuncSetAttribute: %s", hipGetErrorString(e));
    hipLaunchKernelGGL(kernel, dim3(D), dim3(NT), lds, (hipStream_t)stream, p);
    return check_launch(who);
}

static size_t gp_predict_elems(int B, int M, int need_cov) {
    size_t f = (size_t)2 * M * (M + 1) + (size_t)2 * M * (B + 2) + M + 2 * (size_t)B + 16;
    if (need_cov) f += (size_t)B * (B + 1);
    return f;
}
// elements of gp_train_bwd_kernel's working set with the data points in chunks of Bc
static size_t gp_bwd_elems(int B, int M, int Bc) {
    if (Bc >= B)     // one chunk: K^-1 lives in P's last M columns
        return (size_t)4 * M * (M + 1) + (size_t)3 * M * (B + 2) + (size_t)M * (B + M + 3) + 5 * (size_t)M + 4 * (size_t)B + 16;
    const size_t lp = (size_t)(Bc > M + 2 ? Bc : M + 2) + 1;
    return (size_t)5 * M * (M + 1) + (size_t)3 * M * (Bc + 2) + (size_t)M * lp + 5 * (size_t)M + 4 * (size_t)B + 16;
}
// points per chunk: all of them when that fits `bytes_per_elem`-wide arithmetic into the LDS, else the fewest equal chunks
// that do (0: not even chunks of 8 points fit - M too large)
static int gp_bwd_chunk(int B, int M, int bytes_per_elem) {
    for (int n = 1; n <= 16; ++n) {
        const int bc = (B + n - 1) / n;
        if (gp_bwd_elems(B, M, bc) * bytes_per_elem <= 160 * 1024) return bc;
        if (bc <= 8) break;
    }
    return 0;
}

// fp64 with the covariance's lower triangle overlaid on L / L_S (gp_predict_kernel<.., PACKED = true>)
static size_t gp_predict_elems_packed(int B, int M) {
    const size_t tri = (size_t)B * (B + 1) / 2, ll = (size_t)2 * M * (M + 1);
    return (tri > ll ? tri : ll) + (size_t)2 * M * (B + 2) + M + 2 * (size_t)B + 16;
}
// 0: fp32, 1: fp64, 2: fp64 packed
static int gp_predict_variant(int B, int M, int need_cov) {
    if (gp_force_fp32()) return 0;
    if (gp_predict_elems(B, M, need_cov) * 8 <= GP_LDS_MAX) return 1;
    if (need_cov && gp_predict_elems_packed(B, M) * 8 <= GP_LDS_MAX) return 2;
    return 0;
}
extern "C" int dvg_gp_precision(int B, int M, int need_cov) { return gp_predict_variant(B, M, need_cov) ? 64 : 32; }
extern "C" int dvg_gp_bwd_precision(int B, int M) { return (!gp_force_fp32() && gp_bwd_chunk(B, M, 8) > 0) ? 64 : 32; }
extern "C" size_t dvg_gp_lds_bytes(int B, int M, int need_cov) {
    switch (gp_predict_variant(B, M, need_cov)) {
        case 1: return gp_predict_elems(B, M, need_cov) * 8;
        case 2: return gp_predict_elems_packed(B, M) * 8;
        default: return gp_predict_elems(B, M, need_cov) * 4;
    }
}
extern "C" size_t dvg_gp_bwd_lds_bytes(int B, int M) {
    const int bpe = dvg_gp_bwd_precision(B, M) / 8, bc = gp_bwd_chunk(B, M, bpe);
    return gp_bwd_elems(B, M, bc > 0 ? bc : B) * bpe;
}
// data points per chunk of dvg_gp_train_bwd (== B: one pass)
extern "C" int dvg_gp_bwd_chunk(int B, int M) {
    const int bc = gp_bwd_chunk(B, M, dvg_gp_bwd_precision(B, M) / 8);
    return bc > 0 ? bc : B;
}

// Steps per workgroup for a time-batched train-mode call (S steps x P latent dims x B points each; see GpParams::SG).  One
// workgroup per (step, dim) repeats K_ZZ, its factor and the KL term S times per dim and, with 40-155 KB of LDS each, runs the
// S x P workgroups in ceil(S P / 256 r) residency rounds.  The smallest group that brings the launch down to ONE workgroup per
// CU (P ceil(S / k) <= 256) wins when its points still fit: B=16,S=11: k=6, B=4,S=15: k=8 (tools/bench_gp.py, r05); at
// B = 64 no group fits and the call stays one workgroup per column.
extern "C" int dvg_gp_step_group(int B, int S, int P, int M) {
    if (B <= 0 || S <= 1 || P <= 0 || M <= 0 || M > 64 || (long)S * P <= 256) return 1;
    for (int k = 2; k <= S; ++k) {
        if ((long)P * ((S + k - 1) / k) > 256) continue;
        const long bw = (long)k * B;
        if (bw > 256 || gp_predict_variant((int)bw, M, 0) != 1 || gp_bwd_chunk((int)bw, M, 8) <= 0) return 1;
        return k;
    }
    return 1;
}

extern "C" int dvg_gp_predict(const float* h, const float* z, const float* var_mean, const float* chol_var,
                              const float* mean_const, const float* outputscale, const float* lengthscale,
                              const float* noise, const float* eps, float* mean, float* var, float* sample,
                              float* cov, float* kl, int B, int D, int M, int train_mode, float jitter,
                              int param_period, int step_group, void* stream) {
    DVG_REQUIRE(h && z && var_mean && chol_var && mean_const && outputscale && lengthscale, DVG_ERR_NULL,
                "dvg_gp_predict: NULL input");
    DVG_REQUIRE(B > 0 && D > 0 && M > 0 && M <= 64 && B <= 128, DVG_ERR_SHAPE,
                "dvg_gp_predict: need 1<=M<=64, 1<=B<=128 (got M=%d B=%d)", M, B);
    DVG_REQUIRE(sample == nullptr || eps != nullptr, DVG_ERR_NULL, "dvg_gp_predict: sample needs eps");
    const int need_cov = (cov != nullptr) || (sample != nullptr);
    DVG_REQUIRE(train_mode >= 0 && train_mode <= 3, DVG_ERR_SHAPE, "dvg_gp_predict: train_mode flags must be 0..3");
    DVG_REQUIRE(param_period >= 0 && (param_period == 0 || D % param_period == 0), DVG_ERR_SHAPE,
                "dvg_gp_predict: param_period=%d must divide D=%d", param_period, D);
    const int Dp = param_period ? param_period : D, S = D / Dp;
    const int SG = step_group > 1 ? step_group : 1;
    DVG_REQUIRE(SG == 1 || (SG <= S && !need_cov), DVG_ERR_SHAPE,
                "dvg_gp_predict: step_group=%d needs 1..S=%d steps and neither cov nor sample", step_group, S);
    const int Bw = SG * B, G = (S + SG - 1) / SG, nwg = G * Dp;      // points per workgroup, workgroups
    const size_t lds = dvg_gp_lds_bytes(Bw, M, need_cov);
    DVG_REQUIRE(lds <= GP_LDS_MAX, DVG_ERR_SHAPE, "dvg_gp_predict: %zu bytes of LDS needed (> 160 KiB)", lds);
    GpParams p{h, z, var_mean, chol_var, mean_const, outputscale, lengthscale, noise, eps, mean, var, sample, cov, kl,
               B, D, M, train_mode & 1, jitter, (train_mode >> 1) & 1, Dp, SG, g_gp_clk, g_gp_clk_cap};
    const int nt = gp_threads(GP_PREDICT_THREADS, nwg, Bw, false);
    const char* who = "dvg_gp_predict";
    const int variant = gp_predict_variant(Bw, M, need_cov);
    DVG_REQUIRE(SG == 1 || variant == 1, DVG_ERR_SHAPE,
                "dvg_gp_predict: step_group=%d x B=%d points do not fit the LDS in fp64", SG, B);   // never trade precision for it
    D = nwg;        // the grid
    if (variant == 2) {
        switch (nt) {
            case 256: return gp_launch(gp_predict_kernel<256, double, true>, 256, p, D, lds, stream, who);
            case 512: return gp_launch(gp_predict_kernel<512, double, true>, 512, p, D, lds, stream, who);
            default: return gp_launch(gp_predict_kernel<1024, double, true>, 1024, p, D, lds, stream, who);
        }
    }
    if (variant == 1) {
        switch (nt) {
            case 256: return gp_launch(gp_predict_kernel<256, double>, 256, p, D, lds, stream, who);
            case 512: return gp_launch(gp_predict_kernel<512, double>, 512, p, D, lds, stream, who);
            default: return gp_launch(gp_predict_kernel<1024, double>, 1024, p, D, lds, stream, who);
        }
    }
    switch (nt) {
        case 256: return gp_launch(gp_predict_kernel<256, float>, 256, p, D, lds, stream, who);
        case 512: return gp_launch(gp_predict_kernel<512, float>, 512, p, D, lds, stream, who);
        default: return gp_launch(gp_predict_kernel<1024, float>, 1024, p, D, lds, stream, who);
    }
}

extern "C" int dvg_gp_train_bwd(const float* h, const float* z, const float* var_mean, const float* chol_var,
                                const float* mean_const, const float* outputscale, const float* lengthscale,
                                const float* gmean, const float* gvar, const float* gkl, float* dh, float* dz,
                                float* dm, float* dls, float* dc, float* ds, float* dell, int B, int D, int M,
                                float jitter, int param_period, int step_group, void* stream) {
    DVG_REQUIRE(h && z && var_mean && chol_var && mean_const && outputscale && lengthscale, DVG_ERR_NULL,
                "dvg_gp_train_bwd: NULL input");
    DVG_REQUIRE(dh && dz && dm && dls && dc && ds && dell, DVG_ERR_NULL, "dvg_gp_train_bwd: NULL output");
    DVG_REQUIRE(B > 0 && D > 0 && M > 0 && M <= 64 && B <= 128, DVG_ERR_SHAPE,
                "dvg_gp_train_bwd: need 1<=M<=64, 1<=B<=128 (got M=%d B=%d)", M, B);
    DVG_REQUIRE(param_period >= 0 && (param_period == 0 || D % param_period == 0), DVG_ERR_SHAPE,
                "dvg_gp_train_bwd: param_period=%d must divide D=%d", param_period, D);
    const int Dp = param_period ? param_period : D, S = D / Dp;
    const int SG = step_group > 1 ? step_group : 1;
    DVG_REQUIRE(SG <= S, DVG_ERR_SHAPE, "dvg_gp_train_bwd: step_group=%d exceeds the %d steps", step_group, S);
    const int Bw = SG * B, G = (S + SG - 1) / SG, nwg = G * Dp;      // points per workgroup, workgroups (= gradient rows)
    const size_t lds = dvg_gp_bwd_lds_bytes(Bw, M);
    DVG_REQUIRE(lds <= GP_LDS_MAX, DVG_ERR_SHAPE, "dvg_gp_train_bwd: %zu bytes of LDS needed (> 160 KiB)", lds);
    GpBwdParams p{h, z, var_mean, chol_var, mean_const, outputscale, lengthscale, gmean, gvar, gkl,
                  dh, dz, dm, dls, dc, ds, dell, B, D, M, dvg_gp_bwd_chunk(Bw, M), Dp, SG, jitter,
                  g_gp_clk, g_gp_clk_cap};
    const int nt = gp_threads(GP_BWD_THREADS, nwg, Bw, true);
    const char* who = "dvg_gp_train_bwd";
    D = nwg;        // the grid
    DVG_REQUIRE(SG == 1 || dvg_gp_bwd_precision(Bw, M) == 64, DVG_ERR_SHAPE,
                "dvg_gp_train_bwd: step_group=%d x B=%d points do not fit the LDS in fp64", SG, B);
    if (dvg_gp_bwd_precision(Bw, M) == 64) {
        switch (nt) {
            case 256: return gp_launch(gp_train_bwd_kernel<256, double>, 256, p, D, lds, stream, who);
            case 512: return gp_launch(gp_train_bwd_kernel<512, double>, 512, p, D, lds, stream, who);
            default: return gp_launch(gp_train_bwd_kernel<1024, double>, 1024, p, D, lds, stream, who);
        }
    }
    switch (nt) {
        case 256: return gp_launch(gp_train_bwd_kernel<256, float>, 256, p, D, lds, stream, who);
        case 512: return gp_launch(gp_train_bwd_kernel<512, float>, 512, p, D, lds, stream, who);
        default: return gp_launch(gp_train_bwd_kernel<1024, float>, 1024, p, D, lds, stream, who);
    }
}

static int gp_elbo_checks(const float* mean, const float* var, const float* kl, const float* target, const float* raw_noise,
                          int B, int D, int num_data, const char* who) {
    DVG_REQUIRE(mean && var && kl && target && raw_noise, DVG_ERR_NULL, "%s: NULL input", who);
    DVG_REQUIRE(B > 0 && D > 0 && num_data > 0, DVG_ERR_SHAPE, "%s: need B, D, num_data > 0 (got %d, %d, %d)", who, B, D, num_data);
    return DVG_OK;
}

extern "C" int dvg_gp_elbo(const float* mean, const float* var, const float* kl, const float* target, long t_stride_d,
                           long t_stride_b, const float* raw_noise, float* elbo, int B, int D, int num_data, int noise_period,
                           void* stream) {
    if (int e = gp_elbo_checks(mean, var, kl, target, raw_noise, B, D, num_data, "dvg_gp_elbo")) return e;
    DVG_REQUIRE(elbo, DVG_ERR_NULL, "dvg_gp_elbo: NULL output");
    DVG_REQUIRE(noise_period >= 0 && (noise_period == 0 || D % noise_period == 0), DVG_ERR_SHAPE, "dvg_gp_elbo: noise_period must divide D");
    GpElboParams p{mean, var, kl, target, t_stride_d, t_stride_b, raw_noise, nullptr, elbo, nullptr, nullptr, nullptr,
                   nullptr, nullptr, B, D, 1.f / (float)num_data, noise_period ? noise_period : D};
    hipLaunchKernelGGL(gp_elbo_kernel<false>, dim3(D), dim3(256), 0, (hipStream_t)stream, p);
    return check_launch("dvg_gp_elbo");
}

extern "C" int dvg_gp_elbo_bwd(const float* mean, const float* var, const float* kl, const float* target, long t_stride_d,
                               long t_stride_b, const float* raw_noise, const float* gelbo, float* gmean, float* gvar,
                               float* gkl, float* gtarget, float* graw_noise, int B, int D, int num_data, int noise_period,
                               void* stream) {
    if (int e = gp_elbo_checks(mean, var, kl, target, raw_noise, B, D, num_data, "dvg_gp_elbo_bwd")) return e;
    DVG_REQUIRE(gelbo && gmean && gvar && gkl && graw_noise, DVG_ERR_NULL, "dvg_gp_elbo_bwd: NULL gradient buffer");
    DVG_REQUIRE(noise_period >= 0 && (noise_period == 0 || D % noise_period == 0), DVG_ERR_SHAPE, "dvg_gp_elbo_bwd: noise_period must divide D");
    GpElboParams p{mean, var, kl, target, t_stride_d, t_stride_b, raw_noise, gelbo, nullptr, gmean, gvar, gkl, gtarget,
                   graw_noise, B, D, 1.f / (float)num_data, noise_period ? noise_period : D};
    hipLaunchKernelGGL(gp_elbo_kernel<true>, dim3(D), dim3(256), 0, (hipStream_t)stream, p);
    return check_launch("dvg_gp_elbo_bwd");
}

// dst_k[i] = sum_s src_k[s * n_k + i] for up to 8 tensors in one launch: the per-(step, latent dim) parameter gradients of the S
// time steps a closure ran side by side (param_period above) back to one gradient per parameter (what the backward of torch's
// `repeat` did with a reshape + sum launch per tensor).  Fixed summation order.
struct SumStepsParams { const float* src[8]; float* dst[8]; long n[8]; int count, S; };
__global__ __launch_bounds__(256) void sum_steps_kernel(const SumStepsParams p) {
    const int k = blockIdx.y;
    if (k >= p.count) return;
    const long n = p.n[k];
    for (long i = blockIdx.x * 256L + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
        float a = p.src[k][i];
        for (int s_ = 1; s_ < p.S; ++s_) a += p.src[k][(size_t)s_ * n + i];
        p.dst[k][i] = a;
    }
}

extern "C" int dvg_sum_steps_multi(const float* const* src, float* const* dst, const long* n, int count, int S, void* stream) {
    DVG_REQUIRE(src && dst && n, DVG_ERR_NULL, "dvg_sum_steps_multi: NULL pointer");
    DVG_REQUIRE(count >= 1 && count <= 8 && S >= 1, DVG_ERR_SHAPE, "dvg_sum_steps_multi: 1..8 tensors, S >= 1");
    SumStepsParams p{};
    long nmax = 0;
    for (int k = 0; k < count; ++k) {
        DVG_REQUIRE(src[k] && dst[k] && n[k] > 0, DVG_ERR_NULL, "dvg_sum_steps_multi: NULL tensor %d", k);
        p.src[k] = src[k]; p.dst[k] = dst[k]; p.n[k] = n[k];
        nmax = n[k] > nmax ? n[k] : nmax;
    }
    p.count = count; p.S = S;
    long gx = (nmax + 255) / 256;
    if (gx > 1024) gx = 1024;
    hipLaunchKernelGGL(sum_steps_kernel, dim3((unsigned)gx, (unsigned)count), dim3(256), 0, (hipStream_t)stream, p);
    return check_launch("dvg_sum_steps_multi");
}

extern "C" void dvg_debug_set_gp_clockbuf(void* buf, unsigned records) {
    g_gp_clk = (unsigned long long*)buf;
    g_gp_clk_cap = buf ? records : 0;
}
