// Backward (training) kernels that are HBM-bound: BatchNorm + activation + max-pool backward,
// nearest-upsample backward, LSTM gate backward, column sums, and the weight gradient of the
// thin first / last layers.  The MFMA weight-gradient kernel lives in wgrad.hip; data gradients
// of the dense convs reuse conv_igemm2.hip with re-packed weights.
//
// Reference: these implement what `loss.backward()` (train.py:170,194,240) asks autograd to do for
// the modules of vgg_64.py:5-15,49,93, dcgan_64.py:4-26 and lstm.py:51,65-72.
#include "dvg_common.h"

namespace dvg {

__device__ __forceinline__ float act_grad_from_y(float y, int act, float slope) {
    switch (act) {
        case DVG_ACT_LRELU: return y > 0.f ? 1.f : slope;  // slope > 0: sign(y) == sign(pre-activation)
        case DVG_ACT_TANH: return 1.f - y * y;
        case DVG_ACT_SIGMOID: return y * (1.f - y);
        default: return 1.f;
    }
}

// ---------------------------------------------------------------------------------------
// dp = (dy + maxpool_scatter(dyp)) * act'(y);   partial[blk] = { sum dp, sum dp*u } per channel
// One thread = one 2x2 window (POOL) or one pixel, 4 channels.  blockDim = 256 = TC x TP with
// TC = min(C/4, 256) channel quads; rows of TP pixels reduce through LDS (deterministic).
// ---------------------------------------------------------------------------------------
template <bool POOL>
__global__ __launch_bounds__(256) void bn_act_bwd_reduce_kernel(const float* __restrict__ dy,
                                                                const float* __restrict__ dyp,
                                                                const float* __restrict__ y,
                                                                const float* __restrict__ u, float* __restrict__ dp,
                                                                float* __restrict__ partial, int N, int H, int W,
                                                                int C, int act, float slope, int units_per_block,
                                                                int bpg) {
    __shared__ float red[2 * 256 * 4];
    const int C4 = C >> 2;
    const int TC = C4 < 256 ? C4 : 256;
    const int TP = 256 / TC;
    const int tc = threadIdx.x % TC, tp = threadIdx.x / TC;
    const int Hu = POOL ? H >> 1 : H, Wu = POOL ? W >> 1 : W;
    // N = images PER GROUP; group g (time-batched training: an independent BatchNorm batch) owns the `bpg` consecutive
    // workgroups / partial rows [g * bpg, (g + 1) * bpg): no partial row mixes two groups
    const long units = (long)N * Hu * Wu;  // windows or pixels of one group
    const long g_ = blockIdx.x / bpg, lb = blockIdx.x % bpg;
    const long u0 = g_ * units + lb * units_per_block;
    const long u1 = min((g_ + 1) * units, u0 + units_per_block);
    for (int c4 = tc; c4 < C4; c4 += TC) {  // C4 > 256 only when C > 1024: not on this path, kept for safety
        f32x4 s1 = {0.f, 0.f, 0.f, 0.f}, s2 = {0.f, 0.f, 0.f, 0.f};
        for (long w_ = u0 + tp; w_ < u1; w_ += TP) {
            if (POOL) {
                const int xp = w_ % Wu;
                long r = w_ / Wu;
                const int yp_ = r % Hu;
                const int n = r / Hu;
                f32x4 g = {0.f, 0.f, 0.f, 0.f};
                if (dyp) g = reinterpret_cast<const f32x4*>(dyp)[w_ * C4 + c4];
                size_t off[4];
                f32x4 yv[4];
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    off[q] = (((size_t)n * H + 2 * yp_ + (q >> 1)) * W + 2 * xp + (q & 1)) * C4 + c4;
                    yv[q] = reinterpret_cast<const f32x4*>(y)[off[q]];
                }
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    f32x4 d = {0.f, 0.f, 0.f, 0.f};
                    if (dy) d = reinterpret_cast<const f32x4*>(dy)[off[q]];
                    const f32x4 uv = reinterpret_cast<const f32x4*>(u)[off[q]];
#pragma unroll
                    for (int k = 0; k < 4; ++k) {
                        // first maximum in scan order (0,0),(0,1),(1,0),(1,1) wins — nn.MaxPool2d semantics
                        const float m = fmaxf(fmaxf(yv[0][k], yv[1][k]), fmaxf(yv[2][k], yv[3][k]));
                        bool win = yv[q][k] == m;
#pragma unroll
                        for (int e = 0; e < 4; ++e)
                            if (e < q && yv[e][k] == m) win = false;
                        const float t = (d[k] + (win ? g[k] : 0.f)) * act_grad_from_y(yv[q][k], act, slope);
                        d[k] = t;
                        s1[k] += t;
                        s2[k] = fmaf(t, uv[k], s2[k]);
                    }
                    reinterpret_cast<f32x4*>(dp)[off[q]] = d;
                }
            } else {
                const size_t off = (size_t)w_ * C4 + c4;
                f32x4 d = reinterpret_cast<const f32x4*>(dy)[off];
                const f32x4 yv = reinterpret_cast<const f32x4*>(y)[off];
                const f32x4 uv = reinterpret_cast<const f32x4*>(u)[off];
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const float t = d[k] * act_grad_from_y(yv[k], act, slope);
                    d[k] = t;
                    s1[k] += t;
                    s2[k] = fmaf(t, uv[k], s2[k]);
                }
                reinterpret_cast<f32x4*>(dp)[off] = d;
            }
        }
        // reduce over the TP pixel rows
        __syncthreads();
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            red[(tp * TC + tc) * 4 + k] = s1[k];
            red[1024 + (tp * TC + tc) * 4 + k] = s2[k];
        }
        __syncthreads();
        if (tp == 0) {
            float* dst = partial + (size_t)blockIdx.x * 2 * C;
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                float a = 0.f, b = 0.f;
                for (int r = 0; r < TP; ++r) {
                    a += red[(r * TC + tc) * 4 + k];
                    b += red[1024 + (r * TC + tc) * 4 + k];
                }
                dst[c4 * 4 + k] = a;
                dst[C + c4 * 4 + k] = b;
            }
        }
    }
}

// coefficients of du = A*dp + B*u + Cc, plus dgamma, dbeta, dbias.  64 channels x 16 row lanes per workgroup
// (see bn_finalize_kernel in misc_kernels.hip for why).
__global__ __launch_bounds__(1024) void bn_bwd_finalize_kernel(const float* __restrict__ partial, int nrows,
                                                               const float* __restrict__ gamma,
                                                               const float* __restrict__ mean,
                                                               const float* __restrict__ invstd,
                                                               float* __restrict__ coefA, float* __restrict__ coefB,
                                                               float* __restrict__ coefC, float* __restrict__ dgamma,
                                                               float* __restrict__ dbeta, float* __restrict__ dbias,
                                                               int C, double count, int train, int accumulate) {
    __shared__ double red[64 * 16 * 2];
    // blockIdx.y = group (time-batched training): rows [g * nrows, (g + 1) * nrows) and row g of every [G][C] array -
    // dgamma / dbeta / dbias included (the host sums them over the groups: dvg_colsum)
    {
        const size_t go = (size_t)blockIdx.y * C;
        partial += (size_t)blockIdx.y * nrows * 2 * C;
        mean += go; invstd += go; coefA += go; coefB += go; coefC += go;
        if (dgamma) dgamma += go;
        if (dbeta) dbeta += go;
        if (dbias) dbias += go;
    }
    // 16 channels x 64 row lanes per workgroup, C/16 workgroups (misc_kernels.hip partial_colsums has the history)
    const int lc = threadIdx.x & 15, rl = threadIdx.x >> 4;
    const int c = blockIdx.x * 16 + lc;
    double a1[4] = {0.0, 0.0, 0.0, 0.0}, a2[4] = {0.0, 0.0, 0.0, 0.0};
    if (c < C) {
        int r = rl;
        for (; r + 192 < nrows; r += 256) {
            float v1[4], v2[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                v1[k] = partial[(size_t)(r + 64 * k) * 2 * C + c];
                v2[k] = partial[(size_t)(r + 64 * k) * 2 * C + C + c];
            }
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                a1[k] += (double)v1[k];
                a2[k] += (double)v2[k];
            }
        }
        for (; r < nrows; r += 64) {
            a1[0] += (double)partial[(size_t)r * 2 * C + c];
            a2[0] += (double)partial[(size_t)r * 2 * C + C + c];
        }
    }
    red[(rl * 16 + lc) * 2] = (a1[0] + a1[1]) + (a1[2] + a1[3]);
    red[(rl * 16 + lc) * 2 + 1] = (a2[0] + a2[1]) + (a2[2] + a2[3]);
    __syncthreads();
    if (rl != 0 || c >= C) return;
    double s1 = 0.0, s2 = 0.0;
#pragma unroll 8
    for (int k = 0; k < 64; ++k) {
        s1 += red[(k * 16 + lc) * 2];
        s2 += red[(k * 16 + lc) * 2 + 1];
    }
    const double mu = mean[c], is = invstd[c];
    const double g = gamma ? gamma[c] : 1.0;
    const double G = is * (s2 - mu * s1);  // sum dp * xhat
    const double A = g * is;
    // accumulate != 0: the three outputs ARE the parameters' .grad buffers (autograd.py writes gradients in place: no
    // per-use gradient tensor, no accumulation launch)
    if (dgamma) dgamma[c] = (accumulate ? dgamma[c] : 0.f) + (float)G;
    if (dbeta) dbeta[c] = (accumulate ? dbeta[c] : 0.f) + (float)s1;
    if (train) {
        const double B = -A * is * G / count;
        coefA[c] = (float)A;
        coefB[c] = (float)B;
        coefC[c] = (float)(-A * s1 / count - B * mu);
        if (dbias && !accumulate) dbias[c] = 0.f;  // sum_c du == 0 identically under batch statistics
    } else {
        coefA[c] = (float)A;
        coefB[c] = 0.f;
        coefC[c] = 0.f;
        if (dbias) dbias[c] = (accumulate ? dbias[c] : 0.f) + (float)(A * s1);
    }
}

// `sum` (optional): a second output shared by several calls - sum_mode 1: sum = du, 2: sum += du.  It collects d(addend)
// of the decoder calls that share one skip half (autograd._SkipHalf): their three du tensors used to be added by two
// separate full-tensor launches per block and time step.
__global__ void affine3_kernel(const float* __restrict__ dp, const float* __restrict__ u,
                               const float* __restrict__ A, const float* __restrict__ B,
                               const float* __restrict__ Cc, float* __restrict__ du, long n4, int C4,
                               float* __restrict__ sum, int sum_mode, long per_group4) {
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n4; i += (long)gridDim.x * blockDim.x) {
        const int c = (int)(i % C4) * 4 + (per_group4 ? (int)(i / per_group4) * C4 * 4 : 0);   // coefficients are [G][C]
        const f32x4 d = reinterpret_cast<const f32x4*>(dp)[i];
        const f32x4 uv = reinterpret_cast<const f32x4*>(u)[i];
        const f32x4 a = *reinterpret_cast<const f32x4*>(A + c), b = *reinterpret_cast<const f32x4*>(B + c),
                    cc = *reinterpret_cast<const f32x4*>(Cc + c);
        f32x4 o;
#pragma unroll
        for (int k = 0; k < 4; ++k) o[k] = fmaf(a[k], d[k], fmaf(b[k], uv[k], cc[k]));
        reinterpret_cast<f32x4*>(du)[i] = o;
        if (sum_mode == 1) {
            reinterpret_cast<f32x4*>(sum)[i] = o;
        } else if (sum_mode == 2) {
            const f32x4 t = reinterpret_cast<const f32x4*>(sum)[i];
#pragma unroll
            for (int k = 0; k < 4; ++k) o[k] += t[k];
            reinterpret_cast<f32x4*>(sum)[i] = o;
        }
    }
}

__global__ void affine3_scalar_kernel(const float* __restrict__ dp, const float* __restrict__ u,
                                      const float* __restrict__ A, const float* __restrict__ B,
                                      const float* __restrict__ Cc, float* __restrict__ du, long n, int C,
                                      long per_group) {
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
        const int c = (int)(i % C) + (per_group ? (int)(i / per_group) * C : 0);
        du[i] = fmaf(A[c], dp[i], fmaf(B[c], u[i], Cc[c]));
    }
}

// scalar (C % 4 != 0) version of the reduce: dp = dy*act'(y); partial sums per channel
__global__ __launch_bounds__(256) void act_bwd_reduce_scalar_kernel(const float* __restrict__ dy,
                                                                    const float* __restrict__ y,
                                                                    const float* __restrict__ u,
                                                                    float* __restrict__ dp,
                                                                    float* __restrict__ partial, long rows, int C,
                                                                    int act, float slope, int rows_per_block, int bpg) {
    // rows = rows PER GROUP, bpg workgroups per group (see bn_act_bwd_reduce_kernel)
    const long g_ = blockIdx.x / bpg, lb = blockIdx.x % bpg;
    const long r0 = g_ * rows + lb * rows_per_block, r1 = min((g_ + 1) * rows, r0 + rows_per_block);
    for (int c = threadIdx.x; c < C; c += blockDim.x) {
        float s1 = 0.f, s2 = 0.f;
        for (long r = r0; r < r1; ++r) {
            const float t = dy[r * C + c] * act_grad_from_y(y[r * C + c], act, slope);
            dp[r * C + c] = t;
            s1 += t;
            s2 = fmaf(t, u[r * C + c], s2);
        }
        partial[(size_t)blockIdx.x * 2 * C + c] = s1;
        partial[(size_t)blockIdx.x * 2 * C + C + c] = s2;
    }
}

// dst[b] = sum over the groups g (in ascending order: deterministic) with map[g] == b of src[g]; src [groups][m], dst
// [blocks][m], float4 elements.  The adjoint of the shared addend of the time-batched decoder calls (Igemm2Params::add_map):
// d(addend block) = the sum of d(pre-activation) of every group that read it.
__global__ void group_sum_kernel(const float* __restrict__ src, const int* __restrict__ map, float* __restrict__ dst,
                                 int groups, long m4) {
    const int b = blockIdx.y;
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < m4; i += (long)gridDim.x * blockDim.x) {
        f32x4 acc = {0.f, 0.f, 0.f, 0.f};
        for (int g = 0; g < groups; ++g) {
            if (map[g] != b) continue;
            const f32x4 v = reinterpret_cast<const f32x4*>(src)[(size_t)g * m4 + i];
#pragma unroll
            for (int k = 0; k < 4; ++k) acc[k] += v[k];
        }
        reinterpret_cast<f32x4*>(dst)[(size_t)b * m4 + i] = acc;
    }
}

// dpre = dy * act'(y) on a flat tensor (last layers: NCHW frames; nn.Linear+Tanh outputs)
__global__ void act_bwd_kernel(const float* __restrict__ dy, const float* __restrict__ y, float* __restrict__ dpre,
                               long n, int act, float slope) {
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x)
        dpre[i] = dy[i] * act_grad_from_y(y[i], act, slope);
}

// nearest x2 upsample backward: dx[n][y][x][c] = sum of the 2x2 block of dxu
__global__ void upsample2x_bwd_kernel(const float* __restrict__ dxu, float* __restrict__ dx, int N, int H, int W,
                                      int C4) {
    const long total = (long)N * H * W * C4;  // H,W = low resolution
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int c4 = i % C4;
        long r = i / C4;
        const int x = r % W; r /= W;
        const int y = r % H;
        const int n = r / H;
        f32x4 s = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const f32x4 v = reinterpret_cast<const f32x4*>(
                dxu)[(((size_t)n * 2 * H + 2 * y + (q >> 1)) * 2 * W + 2 * x + (q & 1)) * C4 + c4];
#pragma unroll
            for (int k = 0; k < 4; ++k) s[k] += v[k];
        }
        reinterpret_cast<f32x4*>(dx)[i] = s;
    }
}

// out[c] = sum_r a[r][c]   (bias gradients; rows = batch)
// Workgroup = 16 columns x 16 row lanes; a lane walks rows rl, rl + 16, ... with four independent partial sums (loads in
// flight), the lanes combine through LDS in a fixed order (deterministic).  One thread per column walking all rows serially
// took 226 us for the (1216, 1024) gate gradients of a whole BPTT pass (autograd._dense_wgrad).
__global__ __launch_bounds__(256) void colsum_kernel(const float* __restrict__ a, float* __restrict__ out, int rows, int C,
                                                     int accumulate) {
    __shared__ float red[16 * 16];
    const int lc = threadIdx.x & 15, rl = threadIdx.x >> 4;
    const int c = blockIdx.x * 16 + lc;
    float s[4] = {0.f, 0.f, 0.f, 0.f};
    if (c < C) {
        int r = rl;
        for (; r + 48 < rows; r += 64) {
            float v[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) v[k] = a[(size_t)(r + 16 * k) * C + c];
#pragma unroll
            for (int k = 0; k < 4; ++k) s[k] += v[k];
        }
        for (; r < rows; r += 16) s[0] += a[(size_t)r * C + c];
    }
    red[rl * 16 + lc] = (s[0] + s[1]) + (s[2] + s[3]);
    __syncthreads();
    if (rl != 0 || c >= C) return;
    float t = red[lc];
#pragma unroll
    for (int k = 1; k < 16; ++k) t += red[k * 16 + lc];
    out[c] = accumulate ? out[c] + t : t;
}

// Weight-gradient finish: dW (nn layout, possibly a channel slice of a wider weight) = beta * dW + sum_s partial[s], with
// partial[s] the packed [tap][Cout][Cin] slabs of dvg_conv_wgrad.  One launch replaces reduce_partials + unpack + the
// autograd accumulation add.  Same reduction scheme as reduce_partials_kernel (16 float4 columns x 16 slab lanes, fixed
// combination order); a float4 column is 4 consecutive ci of one (tap, co).
//   KIND 0: Conv2d weight (Cout, Ctot, KH, KW):          dst[co][c_lo + ci][a][b]           <- packed[a*KW + b][co][ci]
//   KIND 1: ConvTranspose2d weight (Ctot, Cout, KH, KW): dst[c_lo + ci][co][KH-1-a][KW-1-b] <- packed[a*KW + b][co][ci]
//   KIND 2: plain packed output (contiguous, same indexing as the slabs)
template <int KIND>
__global__ __launch_bounds__(256) void wgrad_finish_kernel(const float* __restrict__ partial, float* __restrict__ dst,
                                                           int S, long n4, int kh, int kw, int cout, int cin, int ctot,
                                                           int c_lo, float beta) {
    __shared__ f32x4 red[16 * 16];
    const int lc = threadIdx.x & 15, sl = threadIdx.x >> 4;
    const long i = (long)blockIdx.x * 16 + lc;
    f32x4 acc[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) acc[k] = f32x4{0.f, 0.f, 0.f, 0.f};
    if (i < n4) {
        int s = sl;
        for (; s + 48 < S; s += 64) {
            f32x4 v[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) v[k] = reinterpret_cast<const f32x4*>(partial)[(size_t)(s + 16 * k) * n4 + i];
#pragma unroll
            for (int k = 0; k < 4; ++k)
#pragma unroll
                for (int e = 0; e < 4; ++e) acc[k][e] += v[k][e];
        }
        for (; s < S; s += 16) {
            const f32x4 v = reinterpret_cast<const f32x4*>(partial)[(size_t)s * n4 + i];
#pragma unroll
            for (int e = 0; e < 4; ++e) acc[0][e] += v[e];
        }
    }
    f32x4 t;
#pragma unroll
    for (int e = 0; e < 4; ++e) t[e] = (acc[0][e] + acc[1][e]) + (acc[2][e] + acc[3][e]);
    red[sl * 16 + lc] = t;
    __syncthreads();
    if (sl != 0 || i >= n4) return;
    f32x4 o = red[lc];
#pragma unroll
    for (int k = 1; k < 16; ++k) {
        const f32x4 v = red[k * 16 + lc];
#pragma unroll
        for (int e = 0; e < 4; ++e) o[e] += v[e];
    }
    if (KIND == 2) {
        f32x4* d = reinterpret_cast<f32x4*>(dst) + i;
        if (beta != 0.f) {
            const f32x4 old = *d;
#pragma unroll
            for (int e = 0; e < 4; ++e) o[e] = fmaf(beta, old[e], o[e]);
        }
        *d = o;
        return;
    }
    const long e0 = i * 4;                 // packed element index of lane 0 of the column: (t, co, ci0)
    const int ci0 = (int)(e0 % cin);
    const long r = e0 / cin;
    const int co = (int)(r % cout), tp = (int)(r / cout);
    const int a = tp / kw, b = tp % kw;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        const int ci = c_lo + ci0 + e;
        const long j = KIND == 0 ? ((((long)co * ctot + ci) * kh + a) * kw + b)
                                 : ((((long)ci * cout + co) * kh + (kh - 1 - a)) * kw + (kw - 1 - b));
        dst[j] = beta != 0.f ? fmaf(beta, dst[j], o[e]) : o[e];
    }
}

// dW3 (Cout, Ctot, 3, 3)[:, c_lo : c_lo + C1] = beta * dW3 + the 2x2 window sums of dK4, the gradient w.r.t. the 4x4
// stride-2 transposed-conv kernel K4 = W (*) ones(2x2) that an upsample + conv3x3 runs as (fused._upconv_packed):
//   K4[co][ci][2 - ty + al][2 - tx + be] += W[co][ci][ty][tx]   =>   dW[co][ci][ty][tx] = sum_{al,be} dK4[co][ci][2-ty+al][2-tx+be]
// dk4p is dK4 in the packed transposed layout of dvg_conv_wgrad(CONVT4S2) after the slab reduction: [16][Cout][C1] with
// tap t = (3 - r) * 4 + (3 - s) holding dK4[co][ci][r][s].
__global__ void k4_to_w3_kernel(const float* __restrict__ dk4p, float* __restrict__ dw, int cout, int c1, int ctot,
                                int c_lo, float beta) {
    const long total = (long)9 * cout * c1;
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int ci = (int)(i % c1);
        long r = i / c1;
        const int co = (int)(r % cout);
        const int t3 = (int)(r / cout), ty = t3 / 3, tx = t3 % 3;
        float s = 0.f;
#pragma unroll
        for (int al = 0; al < 2; ++al)
#pragma unroll
            for (int be = 0; be < 2; ++be) {
                const int rr = 2 - ty + al, ss = 2 - tx + be;
                s += dk4p[((size_t)((3 - rr) * 4 + (3 - ss)) * cout + co) * c1 + ci];
            }
        const long j = (((long)co * ctot + c_lo + ci) * 3 + ty) * 3 + tx;
        dw[j] = beta != 0.f ? fmaf(beta, dw[j], s) : s;
    }
}

// out[i] = sum_s partial[s][i].  Workgroup = 64 float4 columns x 4 slab lanes; every lane walks its slabs with
// 4 independent accumulators (loads in flight), lanes combine through LDS in a fixed order (deterministic).
__global__ __launch_bounds__(256) void reduce_partials_kernel(const float* __restrict__ partial, float* __restrict__ out,
                                                              int S, long n4) {
    // 16 float4 columns x 16 split lanes per workgroup (n4/16 workgroups: a 64x64x9 weight gradient used to get only
    // 144 workgroups of serial loads); every lane keeps four independent partial sums; fixed combination order.
    __shared__ f32x4 red[16 * 16];
    const int lc = threadIdx.x & 15, sl = threadIdx.x >> 4;
    const long i = (long)blockIdx.x * 16 + lc;
    f32x4 acc[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) acc[k] = f32x4{0.f, 0.f, 0.f, 0.f};
    if (i < n4) {
        int s = sl;
        for (; s + 48 < S; s += 64) {
            f32x4 v[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) v[k] = reinterpret_cast<const f32x4*>(partial)[(size_t)(s + 16 * k) * n4 + i];
#pragma unroll
            for (int k = 0; k < 4; ++k)
#pragma unroll
                for (int e = 0; e < 4; ++e) acc[k][e] += v[k][e];
        }
        for (; s < S; s += 16) {
            const f32x4 v = reinterpret_cast<const f32x4*>(partial)[(size_t)s * n4 + i];
#pragma unroll
            for (int e = 0; e < 4; ++e) acc[0][e] += v[e];
        }
    }
    f32x4 t;
#pragma unroll
    for (int e = 0; e < 4; ++e) t[e] = (acc[0][e] + acc[1][e]) + (acc[2][e] + acc[3][e]);
    red[sl * 16 + lc] = t;
    __syncthreads();
    if (sl == 0 && i < n4) {
        f32x4 o = red[lc];
#pragma unroll
        for (int k = 1; k < 16; ++k) {
            const f32x4 v = red[k * 16 + lc];
#pragma unroll
            for (int e = 0; e < 4; ++e) o[e] += v[e];
        }
        reinterpret_cast<f32x4*>(out)[i] = o;
    }
}

// LSTM cell backward, elementwise part: from d(h'), d(c'), the saved activated gates, c and c'
// produce the gate pre-activation gradients dG [B][4H] and dc_prev [B][H].
__global__ void lstm_gates_bwd_kernel(const float* __restrict__ dh, const float* __restrict__ dc,
                                      const float* __restrict__ gates, const float* __restrict__ c_prev,
                                      const float* __restrict__ c_new, float* __restrict__ dG,
                                      float* __restrict__ dc_prev, int B, int H) {
    const long total = (long)B * H;
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int b = i / H, j = i % H;
        const float* g = gates + (size_t)b * 4 * H;
        const float gi = g[j], gf = g[H + j], gg = g[2 * H + j], go = g[3 * H + j];
        const float tc = tanhf(c_new[i]);
        const float dhv = dh ? dh[i] : 0.f;
        const float dcv = (dc ? dc[i] : 0.f) + dhv * go * (1.f - tc * tc);
        float* o = dG + (size_t)b * 4 * H;
        o[j] = dcv * gg * gi * (1.f - gi);
        o[H + j] = dcv * c_prev[i] * gf * (1.f - gf);
        o[2 * H + j] = dcv * gi * (1.f - gg * gg);
        o[3 * H + j] = dhv * tc * go * (1.f - go);
        dc_prev[i] = dcv * gf;
    }
}

// Weight gradient of the thin layers: inp NCHW (N,nc,Hi,Wi) [frame side], dout NHWC (N,Ho,Wo,C)
//   dW[c][ci][a][b] = sum_{n,oy,ox} dout[n][oy][ox][c] * inp[n][ci][S*oy + a - 1][S*ox + b - 1]
// This is the conv weight gradient of the first layers (dout = du) and, by adjointness, of the last
// transposed layers (dout := the layer input x, inp := dpre).  Same tiling as conv_first_kernel:
// lane = channel, one output row per wave, input taps are LDS broadcasts; per-block partial sums.
template <int KS, int S>
__global__ __launch_bounds__(256) void wgrad_thin_kernel(const float* __restrict__ inp,
                                                         const float* __restrict__ dout,
                                                         float* __restrict__ partial, int N, int Hi, int Wi,
                                                         int nc, int C) {
    constexpr int TH = 4, TW = 32;
    constexpr int HH = (TH - 1) * S + KS, HW = (TW - 1) * S + KS;
    constexpr int NT = KS * KS;
    __shared__ float tile[4 * HH * HW];
    __shared__ float red[4 * 64];
    const int Ho = (Hi + 2 - KS) / S + 1, Wo = (Wi + 2 - KS) / S + 1;
    const int tiles_x = (Wo + TW - 1) / TW, tiles_y = (Ho + TH - 1) / TH;
    const int cg = blockIdx.y;
    int t = blockIdx.x;
    const int tx_i = t % tiles_x; t /= tiles_x;
    const int ty_i = t % tiles_y;
    const int n = t / tiles_y;
    const int oy0 = ty_i * TH, ox0 = tx_i * TW;
    const int iy0 = oy0 * S - 1, ix0 = ox0 * S - 1;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int i = threadIdx.x; i < nc * HH * HW; i += 256) {
        const int ci = i / (HH * HW), r = i % (HH * HW);
        const int yy = iy0 + r / HW, xx = ix0 + r % HW;
        float v = 0.f;
        if ((unsigned)yy < (unsigned)Hi && (unsigned)xx < (unsigned)Wi) v = inp[(((size_t)n * nc + ci) * Hi + yy) * Wi + xx];
        tile[i] = v;
    }
    __syncthreads();
    const int c = cg * 64 + lane;
    float acc[4 * NT];
#pragma unroll
    for (int i = 0; i < 4 * NT; ++i) acc[i] = 0.f;
    const int oy = oy0 + wave;
    if (oy < Ho) {
        for (int px = 0; px < TW; ++px) {
            const int ox = ox0 + px;
            if (ox >= Wo) break;
            const float d = dout[(((size_t)n * Ho + oy) * Wo + ox) * C + c];
#pragma unroll
            for (int ci = 0; ci < 4; ++ci) {
                if (ci < nc) {
                    const float* tp = tile + ci * HH * HW + (wave * S) * HW + px * S;
#pragma unroll
                    for (int a = 0; a < KS; ++a)
#pragma unroll
                        for (int b = 0; b < KS; ++b) acc[ci * NT + a * KS + b] = fmaf(d, tp[a * HW + b], acc[ci * NT + a * KS + b]);
                }
            }
        }
    }
    // reduce the 4 waves (rows) through LDS, one tap at a time; write partial[block][c][ci*NT + tap]
    float* dst = partial + ((size_t)blockIdx.x * C + c) * (nc * NT);
#pragma unroll
    for (int i = 0; i < 4 * NT; ++i) {
        if (i < nc * NT) {
            __syncthreads();
            red[wave * 64 + lane] = acc[i];
            __syncthreads();
            if (wave == 0) dst[i] = red[lane] + red[64 + lane] + red[128 + lane] + red[192 + lane];
        }
    }
}

static inline unsigned grid_for(long n, int block = 256, int cap = 4096) {
    long g = (n + block - 1) / block;
    if (g < 1) g = 1;
    if (g > cap) g = cap;
    return (unsigned)g;
}


// ------------------------------------------------------------------------------------
// Fused Adam over a flat parameter group (train.py:95-106: torch.optim.Adam(lr=0.002), default betas / eps):
//   g' = g + wd p;  m = b1 m + (1-b1) g';  v = b2 v + (1-b2) g'^2;
//   p -= (lr / bc1) * m / (sqrt(v) / sqrt(bc2) + eps)        bc1 = 1 - b1^t, bc2 = 1 - b2^t
// (torch.optim.Adam's arithmetic, non-amsgrad).  One launch per group instead of the foreach chain.
// ------------------------------------------------------------------------------------
__global__ void adam_step_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m,
                                 float* __restrict__ v, long n, float lr, float b1, float b2, float eps, float wd,
                                 const int* __restrict__ step_dev, int step_host) {
    // the step count may live in device memory (a captured hipGraph replays with a fresh count every iteration)
    const int t = step_dev ? *step_dev : step_host;
    const double bc1 = 1.0 - pow((double)b1, (double)t), bc2 = 1.0 - pow((double)b2, (double)t);
    const float lr_over_bc1 = (float)((double)lr / bc1), inv_sqrt_bc2 = (float)(1.0 / sqrt(bc2));
    const long n4 = n >> 2;
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n4; i += (long)gridDim.x * blockDim.x) {
        f32x4 pv = reinterpret_cast<f32x4*>(p)[i], mv = reinterpret_cast<f32x4*>(m)[i], vv = reinterpret_cast<f32x4*>(v)[i];
        const f32x4 gv = reinterpret_cast<const f32x4*>(g)[i];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const float gg = gv[e] + wd * pv[e];
            mv[e] = b1 * mv[e] + (1.f - b1) * gg;
            vv[e] = b2 * vv[e] + (1.f - b2) * gg * gg;
            pv[e] -= lr_over_bc1 * mv[e] / (sqrtf(vv[e]) * inv_sqrt_bc2 + eps);
        }
        reinterpret_cast<f32x4*>(p)[i] = pv;
        reinterpret_cast<f32x4*>(m)[i] = mv;
        reinterpret_cast<f32x4*>(v)[i] = vv;
    }
    if (blockIdx.x == 0 && threadIdx.x < (n & 3)) {   // tail
        const long i = (n4 << 2) + threadIdx.x;
        const float gg = g[i] + wd * p[i];
        const float mm = b1 * m[i] + (1.f - b1) * gg, vv = b2 * v[i] + (1.f - b2) * gg * gg;
        m[i] = mm;
        v[i] = vv;
        p[i] -= lr_over_bc1 * mm / (sqrtf(vv) * inv_sqrt_bc2 + eps);
    }
}

// zero_grad of one or several ADJACENT parameter groups of the gradient arena as one fill, and the device-side step counts
// of up to four of them advanced by one in the same launch (a captured iteration replays with fresh bias corrections; the
// increment used to be a one-element torch add_ per optimiser step).
__global__ void zero_tick_kernel(float* __restrict__ g, long n, int* __restrict__ t0, int* __restrict__ t1, int* __restrict__ t2,
                                 int* __restrict__ t3) {
    const long n4 = n >> 2;
    const f32x4 z = {0.f, 0.f, 0.f, 0.f};
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n4; i += (long)gridDim.x * blockDim.x)
        reinterpret_cast<f32x4*>(g)[i] = z;
    if (blockIdx.x == 0) {
        if (threadIdx.x < (n & 3)) g[(n4 << 2) + threadIdx.x] = 0.f;
        if (threadIdx.x == 4 && t0) *t0 += 1;
        if (threadIdx.x == 5 && t1) *t1 += 1;
        if (threadIdx.x == 6 && t2) *t2 += 1;
        if (threadIdx.x == 7 && t3) *t3 += 1;
    }
}

}  // namespace dvg

using namespace dvg;

static int bwd_units_per_block(long units) {
    long upb = (units + 2047) / 2048;
    if (upb < 16) upb = 16;
    return (int)upb;
}

extern "C" int dvg_bn_act_bwd_rows(int N, int H, int W, int pool) {
    const long units = pool ? (long)N * (H / 2) * (W / 2) : (long)N * H * W;
    const int upb = bwd_units_per_block(units);
    return (int)((units + upb - 1) / upb);
}

extern "C" int dvg_bn_act_bwd_reduce(const float* dy, const float* dyp, const float* y, const float* u, float* dp,
                                     float* partial, int N, int H, int W, int C, int act, float slope, int groups,
                                     void* stream) {
    DVG_REQUIRE(y && u && dp && partial, DVG_ERR_NULL, "dvg_bn_act_bwd_reduce: NULL pointer");
    DVG_REQUIRE(dy || dyp, DVG_ERR_NULL, "dvg_bn_act_bwd_reduce: no incoming gradient");
    DVG_REQUIRE(N > 0 && H > 0 && W > 0 && C > 0 && groups > 0 && N % groups == 0, DVG_ERR_SHAPE,
                "dvg_bn_act_bwd_reduce: bad shape (groups must divide N)");
    N /= groups;   // images per group from here on
    DVG_REQUIRE(act >= 0 && act <= 3, DVG_ERR_SHAPE, "dvg_bn_act_bwd_reduce: bad act");
    const int pool = dyp != nullptr;
    if (C % 4 != 0) {
        DVG_REQUIRE(!pool, DVG_ERR_SHAPE, "dvg_bn_act_bwd_reduce: pool needs C %% 4 == 0");
        const long rows = (long)N * H * W;
        const int rpb = bwd_units_per_block(rows);
        const int bpg = (int)((rows + rpb - 1) / rpb);
        hipLaunchKernelGGL(act_bwd_reduce_scalar_kernel, dim3((unsigned)bpg * groups), dim3(C >= 256 ? 256 : 64), 0,
                           (hipStream_t)stream, dy, y, u, dp, partial, rows, C, act, slope, rpb, bpg);
        return check_launch("dvg_bn_act_bwd_reduce");
    }
    DVG_REQUIRE(C <= 1024 && (256 % (C / 4 < 256 ? C / 4 : 256)) == 0, DVG_ERR_SHAPE,
                "dvg_bn_act_bwd_reduce: C/4 must divide 256 (C=%d)", C);
    DVG_REQUIRE(aligned16(dy) && aligned16(dyp) && aligned16(y) && aligned16(u) && aligned16(dp), DVG_ERR_ALIGN,
                "dvg_bn_act_bwd_reduce: alignment");
    if (pool) DVG_REQUIRE(H % 2 == 0 && W % 2 == 0, DVG_ERR_SHAPE, "dvg_bn_act_bwd_reduce: odd H/W with pool");
    const long units = pool ? (long)N * (H / 2) * (W / 2) : (long)N * H * W;
    const int upb = bwd_units_per_block(units);
    const int bpg = (int)((units + upb - 1) / upb);
    const unsigned grid = (unsigned)bpg * groups;
    if (pool)
        hipLaunchKernelGGL((bn_act_bwd_reduce_kernel<true>), dim3(grid), dim3(256), 0, (hipStream_t)stream, dy, dyp, y,
                           u, dp, partial, N, H, W, C, act, slope, upb, bpg);
    else
        hipLaunchKernelGGL((bn_act_bwd_reduce_kernel<false>), dim3(grid), dim3(256), 0, (hipStream_t)stream, dy, dyp,
                           y, u, dp, partial, N, H, W, C, act, slope, upb, bpg);
    return check_launch("dvg_bn_act_bwd_reduce");
}

extern "C" int dvg_bn_bwd_finalize(const float* partial, int nrows, const float* gamma, const float* mean,
                                   const float* invstd, float* coefA, float* coefB, float* coefC, float* dgamma,
                                   float* dbeta, float* dbias, int C, double count, int train, int accumulate,
                                   int groups, void* stream) {
    DVG_REQUIRE(partial && mean && invstd && coefA && coefB && coefC, DVG_ERR_NULL, "dvg_bn_bwd_finalize: NULL");
    DVG_REQUIRE(C > 0 && nrows > 0 && count >= 1.0 && groups > 0 && groups < 65536, DVG_ERR_SHAPE, "dvg_bn_bwd_finalize: bad shape");
    DVG_REQUIRE(groups == 1 || !accumulate, DVG_ERR_SHAPE,
                "dvg_bn_bwd_finalize: with several groups dgamma / dbeta / dbias are [G][C] scratch rows (no accumulation)");
    hipLaunchKernelGGL(bn_bwd_finalize_kernel, dim3((C + 15) / 16, groups), dim3(1024), 0, (hipStream_t)stream, partial, nrows,
                       gamma, mean, invstd, coefA, coefB, coefC, dgamma, dbeta, dbias, C, count, train, accumulate);
    return check_launch("dvg_bn_bwd_finalize");
}

extern "C" int dvg_affine3_apply(const float* dp, const float* u, const float* A, const float* B, const float* Cc,
                                 float* du, long n, int C, float* sum, int sum_mode, int groups, void* stream) {
    DVG_REQUIRE(dp && u && A && B && Cc && du, DVG_ERR_NULL, "dvg_affine3_apply: NULL pointer");
    DVG_REQUIRE(n > 0 && C > 0 && n % C == 0 && groups > 0 && (n / C) % groups == 0, DVG_ERR_SHAPE, "dvg_affine3_apply: bad shape");
    const long per_group = groups > 1 ? n / groups : 0;   // coefficients [G][C], group = consecutive run of n / G elements
    DVG_REQUIRE(sum_mode >= 0 && sum_mode <= 2 && (sum_mode == 0 || (sum != nullptr && sum != du)), DVG_ERR_SHAPE,
                "dvg_affine3_apply: bad sum_mode / sum");
    const bool vec = C % 4 == 0 && aligned16(dp) && aligned16(u) && aligned16(du) && aligned16(A) && aligned16(B) &&
                     aligned16(Cc) && aligned16(sum);
    DVG_REQUIRE(sum_mode == 0 || vec, DVG_ERR_ALIGN, "dvg_affine3_apply: the sum output needs C %% 4 == 0 and 16-byte alignment");
    if (vec)
        hipLaunchKernelGGL(affine3_kernel, dim3(grid_for(n / 4)), dim3(256), 0, (hipStream_t)stream, dp, u, A, B, Cc,
                           du, n / 4, C / 4, sum, sum_mode, per_group / 4);
    else
        hipLaunchKernelGGL(affine3_scalar_kernel, dim3(grid_for(n)), dim3(256), 0, (hipStream_t)stream, dp, u, A, B,
                           Cc, du, n, C, per_group);
    return check_launch("dvg_affine3_apply");
}

extern "C" int dvg_group_sum(const float* src, const int* map, float* dst, int groups, int blocks, long block_elems,
                             void* stream) {
    DVG_REQUIRE(src && map && dst, DVG_ERR_NULL, "dvg_group_sum: NULL pointer");
    DVG_REQUIRE(groups > 0 && blocks > 0 && block_elems > 0 && block_elems % 4 == 0, DVG_ERR_SHAPE,
                "dvg_group_sum: bad shape (block_elems %% 4 == 0 needed)");
    DVG_REQUIRE(aligned16(src) && aligned16(dst), DVG_ERR_ALIGN, "dvg_group_sum: alignment");
    const long m4 = block_elems / 4;
    long gx = (m4 + 255) / 256;
    if (gx > 2048) gx = 2048;
    hipLaunchKernelGGL(group_sum_kernel, dim3((unsigned)gx, (unsigned)blocks), dim3(256), 0, (hipStream_t)stream, src, map, dst,
                       groups, m4);
    return check_launch("dvg_group_sum");
}

extern "C" int dvg_act_bwd(const float* dy, const float* y, float* dpre, long n, int act, float slope, void* stream) {
    DVG_REQUIRE(dy && y && dpre, DVG_ERR_NULL, "dvg_act_bwd: NULL pointer");
    DVG_REQUIRE(n > 0 && act >= 0 && act <= 3, DVG_ERR_SHAPE, "dvg_act_bwd: bad args");
    hipLaunchKernelGGL(act_bwd_kernel, dim3(grid_for(n)), dim3(256), 0, (hipStream_t)stream, dy, y, dpre, n, act,
                       slope);
    return check_launch("dvg_act_bwd");
}

extern "C" int dvg_upsample2x_bwd(const float* dxu, float* dx, int N, int H, int W, int C, void* stream) {
    DVG_REQUIRE(dxu && dx, DVG_ERR_NULL, "dvg_upsample2x_bwd: NULL pointer");
    DVG_REQUIRE(N > 0 && H > 0 && W > 0 && C > 0 && C % 4 == 0, DVG_ERR_SHAPE, "dvg_upsample2x_bwd: bad shape");
    DVG_REQUIRE(aligned16(dxu) && aligned16(dx), DVG_ERR_ALIGN, "dvg_upsample2x_bwd: alignment");
    hipLaunchKernelGGL(upsample2x_bwd_kernel, dim3(grid_for((long)N * H * W * (C / 4))), dim3(256), 0,
                       (hipStream_t)stream, dxu, dx, N, H, W, C / 4);
    return check_launch("dvg_upsample2x_bwd");
}

extern "C" int dvg_colsum(const float* a, float* out, int rows, int C, int accumulate, void* stream) {
    DVG_REQUIRE(a && out, DVG_ERR_NULL, "dvg_colsum: NULL pointer");
    DVG_REQUIRE(rows > 0 && C > 0, DVG_ERR_SHAPE, "dvg_colsum: bad shape");
    hipLaunchKernelGGL(colsum_kernel, dim3((C + 15) / 16), dim3(256), 0, (hipStream_t)stream, a, out, rows, C, accumulate);
    return check_launch("dvg_colsum");
}

extern "C" int dvg_wgrad_finish(const float* partial, int S, float* dst, int kind, int kh, int kw, int cout, int cin,
                                int ctot, int c_lo, float beta, void* stream) {
    DVG_REQUIRE(partial && dst, DVG_ERR_NULL, "dvg_wgrad_finish: NULL pointer");
    DVG_REQUIRE(S > 0 && kh > 0 && kw > 0 && cout > 0 && cin > 0 && cin % 4 == 0 && kind >= 0 && kind <= 2, DVG_ERR_SHAPE,
                "dvg_wgrad_finish: bad shape (Cin %% 4 == 0)");
    DVG_REQUIRE(kind == 2 || (c_lo >= 0 && c_lo + cin <= ctot), DVG_ERR_SHAPE, "dvg_wgrad_finish: channel slice out of range");
    DVG_REQUIRE(aligned16(partial) && (kind != 2 || aligned16(dst)), DVG_ERR_ALIGN, "dvg_wgrad_finish: alignment");
    const long n4 = (long)kh * kw * cout * cin / 4;
    const dim3 grid((unsigned)((n4 + 15) / 16));
    if (kind == 0)
        hipLaunchKernelGGL(wgrad_finish_kernel<0>, grid, dim3(256), 0, (hipStream_t)stream, partial, dst, S, n4, kh, kw, cout,
                           cin, ctot, c_lo, beta);
    else if (kind == 1)
        hipLaunchKernelGGL(wgrad_finish_kernel<1>, grid, dim3(256), 0, (hipStream_t)stream, partial, dst, S, n4, kh, kw, cout,
                           cin, ctot, c_lo, beta);
    else
        hipLaunchKernelGGL(wgrad_finish_kernel<2>, grid, dim3(256), 0, (hipStream_t)stream, partial, dst, S, n4, kh, kw, cout,
                           cin, ctot, c_lo, beta);
    return check_launch("dvg_wgrad_finish");
}

extern "C" int dvg_k4_to_w3(const float* dk4_packed, float* dw, int cout, int c1, int ctot, int c_lo, float beta,
                            void* stream) {
    DVG_REQUIRE(dk4_packed && dw, DVG_ERR_NULL, "dvg_k4_to_w3: NULL pointer");
    DVG_REQUIRE(cout > 0 && c1 > 0 && c_lo >= 0 && c_lo + c1 <= ctot, DVG_ERR_SHAPE, "dvg_k4_to_w3: bad shape");
    hipLaunchKernelGGL(k4_to_w3_kernel, dim3(grid_for((long)9 * cout * c1)), dim3(256), 0, (hipStream_t)stream, dk4_packed, dw,
                       cout, c1, ctot, c_lo, beta);
    return check_launch("dvg_k4_to_w3");
}

extern "C" int dvg_reduce_partials(const float* partial, float* out, int S, long n, void* stream) {
    DVG_REQUIRE(partial && out, DVG_ERR_NULL, "dvg_reduce_partials: NULL pointer");
    DVG_REQUIRE(S > 0 && n > 0 && n % 4 == 0, DVG_ERR_SHAPE, "dvg_reduce_partials: n %% 4 != 0");
    DVG_REQUIRE(aligned16(partial) && aligned16(out), DVG_ERR_ALIGN, "dvg_reduce_partials: alignment");
    hipLaunchKernelGGL(reduce_partials_kernel, dim3((unsigned)((n / 4 + 15) / 16)), dim3(256), 0, (hipStream_t)stream,
                       partial, out, S, n / 4);
    return check_launch("dvg_reduce_partials");
}

extern "C" int dvg_lstm_gates_bwd(const float* dh, const float* dc, const float* gates, const float* c_prev,
                                  const float* c_new, float* dG, float* dc_prev, int B, int H, void* stream) {
    DVG_REQUIRE(gates && c_prev && c_new && dG && dc_prev, DVG_ERR_NULL, "dvg_lstm_gates_bwd: NULL pointer");
    DVG_REQUIRE(dh || dc, DVG_ERR_NULL, "dvg_lstm_gates_bwd: no incoming gradient");
    DVG_REQUIRE(B > 0 && H > 0, DVG_ERR_SHAPE, "dvg_lstm_gates_bwd: bad shape");
    hipLaunchKernelGGL(lstm_gates_bwd_kernel, dim3(grid_for((long)B * H)), dim3(256), 0, (hipStream_t)stream, dh, dc,
                       gates, c_prev, c_new, dG, dc_prev, B, H);
    return check_launch("dvg_lstm_gates_bwd");
}

extern "C" int dvg_wgrad_thin_rows(int ks, int N, int Hi, int Wi) {
    const int S = ks == 4 ? 2 : 1;
    const int Ho = (Hi + 2 - ks) / S + 1, Wo = (Wi + 2 - ks) / S + 1;
    return N * ((Ho + 3) / 4) * ((Wo + 31) / 32);
}

extern "C" int dvg_wgrad_thin(const float* inp_nchw, const float* dout_nhwc, float* partial, int ks, int N, int Hi,
                              int Wi, int nc, int C, void* stream) {
    DVG_REQUIRE(inp_nchw && dout_nhwc && partial, DVG_ERR_NULL, "dvg_wgrad_thin: NULL pointer");
    DVG_REQUIRE(N > 0 && Hi > 0 && Wi > 0 && nc >= 1 && nc <= 4 && C > 0 && C % 64 == 0, DVG_ERR_SHAPE,
                "dvg_wgrad_thin: bad shape");
    DVG_REQUIRE(ks == 3 || ks == 4, DVG_ERR_SHAPE, "dvg_wgrad_thin: ks must be 3 or 4");
    const unsigned gx = (unsigned)dvg_wgrad_thin_rows(ks, N, Hi, Wi);
    if (ks == 3)
        hipLaunchKernelGGL((wgrad_thin_kernel<3, 1>), dim3(gx, C / 64), dim3(256), 0, (hipStream_t)stream, inp_nchw,
                           dout_nhwc, partial, N, Hi, Wi, nc, C);
    else
        hipLaunchKernelGGL((wgrad_thin_kernel<4, 2>), dim3(gx, C / 64), dim3(256), 0, (hipStream_t)stream, inp_nchw,
                           dout_nhwc, partial, N, Hi, Wi, nc, C);
    return check_launch("dvg_wgrad_thin");
}

extern "C" int dvg_adam_step(float* param, const float* grad, float* exp_avg, float* exp_avg_sq, long n, float lr,
                             float beta1, float beta2, float eps, float weight_decay, int step, const int* step_dev,
                             void* stream) {
    DVG_REQUIRE(param && grad && exp_avg && exp_avg_sq, DVG_ERR_NULL, "dvg_adam_step: NULL pointer");
    DVG_REQUIRE(n > 0 && (step >= 1 || step_dev != nullptr), DVG_ERR_SHAPE, "dvg_adam_step: n=%ld step=%d", n, step);
    DVG_REQUIRE(aligned16(param) && aligned16(grad) && aligned16(exp_avg) && aligned16(exp_avg_sq), DVG_ERR_ALIGN,
                "dvg_adam_step: buffers must be 16-byte aligned");
    hipLaunchKernelGGL(adam_step_kernel, dim3(grid_for((n + 3) / 4)), dim3(256), 0, (hipStream_t)stream, param, grad,
                       exp_avg, exp_avg_sq, n, lr, beta1, beta2, eps, weight_decay, step_dev, step);
    return check_launch("dvg_adam_step");
}

// g[0:n] = 0 and *t_k += 1 for the non-NULL step counters (see zero_tick_kernel): optimizer.zero_grad() of adjacent groups of
// the gradient arena (train.py:201-203) + the step-count increments of the optimiser steps that follow in a captured iteration.
extern "C" int dvg_zero_tick(float* g, long n, int* t0, int* t1, int* t2, int* t3, void* stream) {
    DVG_REQUIRE(g, DVG_ERR_NULL, "dvg_zero_tick: NULL pointer");
    DVG_REQUIRE(n > 0 && aligned16(g), DVG_ERR_SHAPE, "dvg_zero_tick: n > 0 and a 16-byte aligned buffer needed");
    hipLaunchKernelGGL(zero_tick_kernel, dim3(grid_for((n + 3) / 4)), dim3(256), 0, (hipStream_t)stream, g, n, t0, t1, t2, t3);
    return check_launch("dvg_zero_tick");
}
