// Weight re-layout, layout converters, train-mode BatchNorm helpers.
// All HBM-bound elementwise / reduction kernels: 16-byte accesses, grid-stride.
#include <algorithm>

#include "dvg_common.h"

namespace dvg {

static thread_local char g_err[512] = "";
char* err_buf() { return g_err; }
int fail(int code, const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
    return code;
}

// packed[t][co][ci]  <->  w[co][ci][kh][kw] (conv)  or  w[ci][co][KH-1-kh][KW-1-kw] (convT)
template <bool TRANSPOSED, bool UNPACK>
__global__ void pack_weight_kernel(const float* __restrict__ src, float* __restrict__ dst, int cout, int cin, int kh,
                                   int kw) {
    const long total = (long)cout * cin * kh * kw;
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        // i indexes the packed tensor [t][co][ci]
        const int ci = i % cin;
        long r = i / cin;
        const int co = r % cout;
        const int t = r / cout;
        const int a = t / kw, b = t % kw;
        long j;
        if (!TRANSPOSED) j = (((long)co * cin + ci) * kh + a) * kw + b;
        else j = (((long)ci * cout + co) * kh + (kh - 1 - a)) * kw + (kw - 1 - b);
        if (!UNPACK) dst[i] = src[j];
        else dst[j] = src[i];
    }
}

// [N][C][HW] <-> [N][HW][C] through a 32x33 LDS tile
template <bool TO_NHWC>
__global__ void layout_kernel(const float* __restrict__ x, float* __restrict__ y, int C, int HW) {
    __shared__ float tile[32][33];
    const int n = blockIdx.z;
    const int c0 = blockIdx.y * 32, p0 = blockIdx.x * 32;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;  // 32 x 8
    const float* xs = x + (size_t)n * C * HW;
    float* ys = y + (size_t)n * C * HW;
    if (TO_NHWC) {
        for (int i = ty; i < 32; i += 8) {
            const int c = c0 + i, p = p0 + tx;
            tile[i][tx] = (c < C && p < HW) ? xs[(size_t)c * HW + p] : 0.f;
        }
        __syncthreads();
        for (int i = ty; i < 32; i += 8) {
            const int p = p0 + i, c = c0 + tx;
            if (c < C && p < HW) ys[(size_t)p * C + c] = tile[tx][i];
        }
    } else {
        for (int i = ty; i < 32; i += 8) {
            const int p = p0 + i, c = c0 + tx;
            tile[i][tx] = (c < C && p < HW) ? xs[(size_t)p * C + c] : 0.f;
        }
        __syncthreads();
        for (int i = ty; i < 32; i += 8) {
            const int c = c0 + i, p = p0 + tx;
            if (c < C && p < HW) ys[(size_t)c * HW + p] = tile[tx][i];
        }
    }
}

// per-channel sum / sum of squares of an NHWC tensor viewed as [rows][C]:
// grid.x slabs of rows, each writes one partial row [2][C] (deterministic).
// Groups (time-batched training): the rows are `groups` consecutive runs of `rows` rows, every run gets its own `bpg`
// slabs (no slab straddles two groups) and the partial rows come out group-major.
__global__ void channel_stats_kernel(const float* __restrict__ u, float* __restrict__ partial, long rows, int C,
                                     int rows_per_block, int bpg) {
    const long g = blockIdx.x / bpg, lb = blockIdx.x % bpg;
    const long r0 = g * rows + lb * rows_per_block;
    const long r1 = min((g + 1) * rows, r0 + rows_per_block);
    for (int c = threadIdx.x; c < C; c += blockDim.x) {
        float s1 = 0.f, s2 = 0.f;
        for (long r = r0; r < r1; ++r) {
            const float v = u[r * C + c];
            s1 += v;
            s2 += v * v;
        }
        partial[(size_t)blockIdx.x * 2 * C + c] = s1;
        partial[(size_t)blockIdx.x * 2 * C + C + c] = s2;
    }
}

// Column sums of the partial rows in double: a workgroup = 16 channels x 64 row lanes (1024 threads) and the grid
// is C/16 workgroups; each lane walks its rows four at a time (eight independent loads in flight), lanes are
// combined through LDS in a fixed order (deterministic).  History: one thread per channel walking up to 2048
// rows serially took 100-230 us per call; 64 channels x 16 lanes in C/64 workgroups (ONE workgroup for a 64-channel
// layer) still 13-21 us.
#define DVG_COLSUM_CT 16
__device__ __forceinline__ void partial_colsums(const float* __restrict__ partial, int nrows, int C, int c, int rl,
                                                double* red, double& s1, double& s2) {
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
    const int lc = threadIdx.x & (DVG_COLSUM_CT - 1);
    red[(rl * DVG_COLSUM_CT + lc) * 2] = (a1[0] + a1[1]) + (a1[2] + a1[3]);
    red[(rl * DVG_COLSUM_CT + lc) * 2 + 1] = (a2[0] + a2[1]) + (a2[2] + a2[3]);
    __syncthreads();
    s1 = 0.0;
    s2 = 0.0;
    if (rl == 0) {
#pragma unroll 8
        for (int k = 0; k < 64; ++k) {
            s1 += red[(k * DVG_COLSUM_CT + lc) * 2];
            s2 += red[(k * DVG_COLSUM_CT + lc) * 2 + 1];
        }
    }
}

__global__ __launch_bounds__(1024) void bn_finalize_kernel(const float* __restrict__ partial, int nrows,
                                                           const float* __restrict__ gamma,
                                                           const float* __restrict__ beta, float* __restrict__ scale,
                                                           float* __restrict__ shift, float* __restrict__ running_mean,
                                                           float* __restrict__ running_var,
                                                           float* __restrict__ save_mean,
                                                           float* __restrict__ save_invstd, int C, double count,
                                                           float eps, float momentum, long long* __restrict__ nbt,
                                                           int nbt_inc, float* __restrict__ group_var) {
    __shared__ double red[64 * DVG_COLSUM_CT * 2];
    // Groups (blockIdx.y; time-batched training: one launch finalises the statistics of G independent BatchNorm batches):
    // group g owns partial rows [g * nrows, (g + 1) * nrows) and row g of every [G][C] output.  The running statistics
    // are a recurrence over the groups in call order: the host passes them only with one group, dvg_bn_running_update
    // applies it otherwise (from save_mean and group_var = the unbiased variance).
    const int g_ = blockIdx.y;
    partial += (size_t)g_ * nrows * 2 * C;
    scale += (size_t)g_ * C;
    shift += (size_t)g_ * C;
    if (save_mean) save_mean += (size_t)g_ * C;
    if (save_invstd) save_invstd += (size_t)g_ * C;
    if (group_var) group_var += (size_t)g_ * C;
    // num_batches_tracked += passes (nn.BatchNorm2d's int64 counter): one lane of the launch, instead of a torch add
    if (nbt != nullptr && blockIdx.x == 0 && blockIdx.y == 0 && threadIdx.x == 0) *nbt += nbt_inc;
    const int c = blockIdx.x * DVG_COLSUM_CT + (threadIdx.x & (DVG_COLSUM_CT - 1)), rl = threadIdx.x >> 4;
    double s1, s2;
    partial_colsums(partial, nrows, C, c, rl, red, s1, s2);
    if (rl != 0 || c >= C) return;
    const double mean = s1 / count;
    double var = s2 / count - mean * mean;
    if (var < 0.0) var = 0.0;
    const float invstd = (float)(1.0 / sqrt(var + (double)eps));
    const float g = gamma ? gamma[c] : 1.f, b = beta ? beta[c] : 0.f;
    const float sc = g * invstd;
    scale[c] = sc;
    shift[c] = b - (float)mean * sc;
    if (save_mean) save_mean[c] = (float)mean;
    if (save_invstd) save_invstd[c] = invstd;
    const double unbiased = count > 1.0 ? var * count / (count - 1.0) : var;
    if (group_var) group_var[c] = (float)unbiased;
    if (running_mean) running_mean[c] = (1.f - momentum) * running_mean[c] + momentum * (float)mean;
    if (running_var) running_var[c] = (1.f - momentum) * running_var[c] + momentum * (float)unbiased;
}

// Running statistics of G consecutive train-mode BatchNorm calls (groups of one time-batched launch) in call order:
//   running <- (1 - m_g) running + m_g stat_g,   g = 0 .. G-1,   m_0 = mom_first, m_{G-1} = mom_last, else mom_mid
// - the arithmetic bn_finalize_kernel applies per call, on the float values it stored (mean, unbiased variance).
// The first / last frame of a sequence is encoded once per closure and every middle frame twice (fused.bn_passes): hence
// three momenta.  One thread per channel.
__global__ void bn_running_update_kernel(const float* __restrict__ mean, const float* __restrict__ uvar, int G, int C,
                                         float mom_first, float mom_mid, float mom_last, float* __restrict__ running_mean,
                                         float* __restrict__ running_var, long long* __restrict__ nbt, int nbt_inc) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c == 0 && nbt != nullptr) *nbt += nbt_inc;
    if (c >= C) return;
    float rm = running_mean[c], rv = running_var[c];
    for (int g = 0; g < G; ++g) {
        const float momentum = g == 0 ? mom_first : (g == G - 1 ? mom_last : mom_mid);
        rm = (1.f - momentum) * rm + momentum * mean[(size_t)g * C + c];
        rv = (1.f - momentum) * rv + momentum * uvar[(size_t)g * C + c];
    }
    running_mean[c] = rm;
    running_var[c] = rv;
}

// y = act(u*scale+shift), float4 over channels
// per_group4: float4 elements per coefficient group (scale / shift are [G][C]); 0 = one group
__global__ void bn_act_kernel(const float* __restrict__ u, const float* __restrict__ scale,
                              const float* __restrict__ shift, float* __restrict__ y, long n4, int C4, int act,
                              float slope, long per_group4) {
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n4; i += (long)gridDim.x * blockDim.x) {
        const int c = (int)(i % C4) * 4 + (per_group4 ? (int)(i / per_group4) * C4 * 4 : 0);
        f32x4 v = reinterpret_cast<const f32x4*>(u)[i];
        const f32x4 sc = *reinterpret_cast<const f32x4*>(scale + c);
        const f32x4 sf = *reinterpret_cast<const f32x4*>(shift + c);
#pragma unroll
        for (int k = 0; k < 4; ++k) v[k] = apply_act(v[k] * sc[k] + sf[k], act, slope);
        reinterpret_cast<f32x4*>(y)[i] = v;
    }
}

// scalar variant for channel counts that are not a multiple of 4 (the (N,90) encoder head)
__global__ void bn_act_scalar_kernel(const float* __restrict__ u, const float* __restrict__ scale,
                                     const float* __restrict__ shift, float* __restrict__ y, long n, int C, int act,
                                     float slope, long per_group) {
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
        const int c = (int)(i % C) + (per_group ? (int)(i / per_group) * C : 0);
        y[i] = apply_act(u[i] * scale[c] + shift[c], act, slope);
    }
}

// same + fused 2x2 max-pool: one thread per (pooled pixel, 4 channels)
__global__ void bn_act_pool_kernel(const float* __restrict__ u, const float* __restrict__ scale,
                                   const float* __restrict__ shift, float* __restrict__ y,
                                   float* __restrict__ y_pool, int N, int H, int W, int C4, int act, float slope,
                                   int group_images) {
    const int Hp = H >> 1, Wp = W >> 1;
    const long total = (long)N * Hp * Wp * C4;
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int c4 = i % C4;
        long r = i / C4;
        const int xp = r % Wp; r /= Wp;
        const int yp = r % Hp;
        const int n = r / Hp;
        const int cg = (group_images ? (n / group_images) * C4 * 4 : 0) + c4 * 4;
        const f32x4 sc = *reinterpret_cast<const f32x4*>(scale + cg);
        const f32x4 sf = *reinterpret_cast<const f32x4*>(shift + cg);
        f32x4 mx;
#pragma unroll
        for (int dy = 0; dy < 2; ++dy)
#pragma unroll
            for (int dx = 0; dx < 2; ++dx) {
                const size_t off = ((((size_t)n * H + 2 * yp + dy) * W + 2 * xp + dx) * C4 + c4);
                f32x4 v = reinterpret_cast<const f32x4*>(u)[off];
#pragma unroll
                for (int k = 0; k < 4; ++k) v[k] = apply_act(v[k] * sc[k] + sf[k], act, slope);
                reinterpret_cast<f32x4*>(y)[off] = v;
                if (dy == 0 && dx == 0) mx = v;
                else
#pragma unroll
                    for (int k = 0; k < 4; ++k) mx[k] = fmaxf(mx[k], v[k]);
            }
        reinterpret_cast<f32x4*>(y_pool)[i] = mx;
    }
}

static inline unsigned grid_for(long n, int block = 256, int cap = 2048) {
    long g = (n + block - 1) / block;
    if (g < 1) g = 1;
    if (g > cap) g = cap;
    return (unsigned)g;
}


// ------------------------------------------------------------------------------------
// eval_frames: SSIM and PSNR of one predicted channel image against its ground truth, as utils.eval_seq
// (utils.py:220-234) computes them through skimage.measure.compare_ssim / compare_psnr (utils.py:13-14; skimage
// <= 0.15 defaults): 7x7 uniform window, sample covariance (x 49/48), K1 = 0.01, K2 = 0.03, data range 2 for float
// images, mean over the (H-6)x(W-6) window positions that lie inside the image; PSNR with data range 1 when the
// ground truth is non-negative (else 2).  One workgroup per (sample, channel) image, both images in LDS, window
// sums in double.
// ------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void eval_frames_kernel(const float* __restrict__ gt, const float* __restrict__ pred,
                                                          float* __restrict__ ssim, float* __restrict__ psnr, int H,
                                                          int W) {
    extern __shared__ __attribute__((aligned(16))) float img[];   // [2][H*W]
    float* X = img;
    float* Y = img + H * W;
    __shared__ double red[3 * 4];
    __shared__ float redmin[4];
    const size_t base = (size_t)blockIdx.x * H * W;
    double se = 0.0;
    float mn = 3.4e38f;
    for (int i = threadIdx.x; i < H * W; i += 256) {
        const float a = gt[base + i], b = pred[base + i];
        X[i] = a;
        Y[i] = b;
        const double d = (double)a - (double)b;
        se += d * d;
        mn = fminf(mn, a);
    }
    __syncthreads();
    constexpr int WIN = 7;
    const int Ho = H - WIN + 1, Wo = W - WIN + 1;
    const double NP = WIN * WIN, cov_norm = NP / (NP - 1.0);
    const double C1 = (0.01 * 2.0) * (0.01 * 2.0), C2 = (0.03 * 2.0) * (0.03 * 2.0);
    double ssum = 0.0;
    for (int o = threadIdx.x; o < Ho * Wo; o += 256) {
        const int oy = o / Wo, ox = o % Wo;
        double sx = 0, sy = 0, sxx = 0, syy = 0, sxy = 0;
        for (int dy = 0; dy < WIN; ++dy) {
            const float* xr = X + (oy + dy) * W + ox;
            const float* yr = Y + (oy + dy) * W + ox;
#pragma unroll
            for (int dx = 0; dx < WIN; ++dx) {
                const double a = xr[dx], b = yr[dx];
                sx += a; sy += b; sxx += a * a; syy += b * b; sxy += a * b;
            }
        }
        const double ux = sx / NP, uy = sy / NP;
        const double vx = cov_norm * (sxx / NP - ux * ux), vy = cov_norm * (syy / NP - uy * uy);
        const double vxy = cov_norm * (sxy / NP - ux * uy);
        ssum += ((2.0 * ux * uy + C1) * (2.0 * vxy + C2)) / ((ux * ux + uy * uy + C1) * (vx + vy + C2));
    }
    // workgroup reduction (fixed order)
    for (int off = 32; off > 0; off >>= 1) {
        ssum += __shfl_xor(ssum, off);
        se += __shfl_xor(se, off);
        mn = fminf(mn, __shfl_xor(mn, off));
    }
    const int wave = threadIdx.x >> 6;
    if ((threadIdx.x & 63) == 0) {
        red[wave] = ssum;
        red[4 + wave] = se;
        redmin[wave] = mn;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        const double s = (red[0] + red[1]) + (red[2] + red[3]);
        const double e = (red[4] + red[5]) + (red[6] + red[7]);
        const float m = fminf(fminf(redmin[0], redmin[1]), fminf(redmin[2], redmin[3]));
        ssim[blockIdx.x] = (float)(s / ((double)Ho * Wo));
        const double range = m >= 0.f ? 1.0 : 2.0;
        const double mse = e / ((double)H * W);
        psnr[blockIdx.x] = (float)(10.0 * log10(range * range / mse));   // +inf for identical images, like skimage
    }
}


// ------------------------------------------------------------------------------------
// moving_mnist_compose: the compositing step of data/moving_mnist.py:86-90 (x[t, sy:sy+32, sx:sx+32, 0] += digit
// per digit in order, then x[x > 1] = 1) written straight into the (T,B,1,S,S) layout utils.normalize_data
// produces (utils.py:86-95).  The integer trajectories (bounce rules, RNG order) stay on the host; this is the
// byte-moving part.  One thread per output pixel, digits summed in index order (bit-exact with the host loop).
// ------------------------------------------------------------------------------------
__global__ void moving_mnist_compose_kernel(const float* __restrict__ sprites, const int* __restrict__ ids,
                                            const int* __restrict__ pos, float* __restrict__ out, int T, int B,
                                            int ND, int S, int D, int n_sprites) {
    const long total = (long)T * B * S * S;
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int x = (int)(i % S);
        long r = i / S;
        const int y = (int)(r % S); r /= S;
        const int b = (int)(r % B);
        const int t = (int)(r / B);
        float v = 0.f;
        for (int d = 0; d < ND; ++d) {
            const int* pp = pos + (((size_t)b * ND + d) * T + t) * 2;
            const int yy = y - pp[0], xx = x - pp[1];
            if ((unsigned)yy < (unsigned)D && (unsigned)xx < (unsigned)D)
                v += sprites[((size_t)min(max(ids[b * ND + d], 0), n_sprites - 1) * D + yy) * D + xx];   // ids clamped: device data
        }
        out[i] = v > 1.f ? 1.f : v;
    }
}


// ---- GPtrigger_gen's bookkeeping on the device (generate_frames.py:227-232,283-296) ---------------------------------------------
// norms[b] = || var[:, b] ||_2 over the D latent dims (`np.linalg.norm(variance.numpy().transpose(), axis = 1)`, :230,275), the
// sum of squares in fp64, rounded to float32 like the reference's float32 array.  One wave per sample.
__global__ __launch_bounds__(64) void gp_var_norms_kernel(const float* __restrict__ var, float* __restrict__ norms, int D, int B) {
    const int b = blockIdx.x, lane = threadIdx.x;
    double s = 0.0;
    for (int d = lane; d < D; d += 64) {
        const double v = (double)var[(size_t)d * B + b];
        s += v * v;
    }
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) s += __shfl_xor(s, o);
    if (lane == 0) norms[b] = (float)sqrt(s);
}

// One step of the main loop's decision (:285-289): value = norm of sample `col`'s variance; the 12-long context window slides
// (`np.concatenate([ctx[1:], [value]])`); threshold = mean + coef * std (numpy population std) with float32 results at the
// points where the reference's float32 arrays round (mean, std, coef * std, the sum); flag = value > threshold.  ctx (W floats)
// is updated in place; value / threshold / flag are logged at `slot` for ONE read-back after the rollout.  One wave.
// TOLERANCE: the norm, the mean and the variance are accumulated in fp64 and rounded once to float32; the
// reference's np.linalg.norm / np.mean / np.std run on float32 arrays with float32 pairwise accumulation and can land 1 ulp
// (6e-8 relative) away, and whether `coef * std` is a float32 or a float64 product depends on the NumPy version (1.x value-based
// casting vs NumPy 2 scalar promotion; float32 here).  A decision can therefore differ from the reference's only where
// |value - threshold| is within a few ulp of the threshold; the parity tests (tests/test_gpu_rollouts.py,
// test_gpu_generate_config.py) compare decisions where the oracle's margin exceeds 1e-4 relative - ~1 000 ulp - and the logged
// values / thresholds everywhere to 1e-5.
// the slid window's statistics and the decision, one wave: lane i < W holds window element w; returns value > threshold
__device__ __forceinline__ bool trigger_window_decide(int lane, float w, int W, float coef, float value, float& thr_out) {
    double m = lane < W ? (double)w : 0.0;
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) m += __shfl_xor(m, o);
    const float mean = (float)(m / W);
    double q = 0.0;
    if (lane < W) {
        const float dlt = w - mean;                      // float32 like `x - x.mean()` on a float32 array
        q = (double)dlt * (double)dlt;
    }
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) q += __shfl_xor(q, o);
    const float sd = (float)sqrt((double)(float)(q / W));
    thr_out = mean + coef * sd;                          // float32 product, float32 sum (NumPy 2 scalar promotion)
    return value > thr_out;
}

__global__ __launch_bounds__(64) void gp_trigger_step_kernel(const float* __restrict__ var, int D, int B, int col, float* __restrict__ ctx,
                                                             int W, float coef, int* __restrict__ flag, float* __restrict__ values,
                                                             float* __restrict__ thresholds, int* __restrict__ flags, int slot) {
    const int lane = threadIdx.x;
    double s = 0.0;
    for (int d = lane; d < D; d += 64) {
        const double v = (double)var[(size_t)d * B + col];
        s += v * v;
    }
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) s += __shfl_xor(s, o);
    const float value = (float)sqrt(s);
    // the slid window: lane i holds element i
    float w = 0.f;
    if (lane < W) w = lane + 1 < W ? ctx[lane + 1] : value;
    float thr;
    const bool trig = trigger_window_decide(lane, w, W, coef, value, thr);
    __syncthreads();                                     // every lane has read its ctx[lane + 1]
    if (lane < W) ctx[lane] = w;
    if (lane == 0) {
        const int f = trig ? 1 : 0;
        *flag = f;
        values[slot] = value;
        thresholds[slot] = thr;
        flags[slot] = f;
    }
}

// The decisions another batch index WOULD take on a recorded sequence of values (the main loop's value is that of sample
// [3] whatever the index, :230; only the window's first entries - the warm-up norms of the index's own column, :275 - differ):
// the same arithmetic as gp_trigger_step_kernel, n steps in one launch.  Valid as long as the decisions it returns equal the
// recorded rollout's (then the index's rollout IS that rollout).
__global__ __launch_bounds__(64) void gp_trigger_replay_kernel(const float* __restrict__ values, int n, const float* __restrict__ ctx0,
                                                               int W, float coef, int* __restrict__ flags,
                                                               float* __restrict__ thresholds) {
    const int lane = threadIdx.x;
    float w = lane < W ? ctx0[lane] : 0.f;
    for (int i = 0; i < n; ++i) {
        const float value = values[i];
        const float up = __shfl_down(w, 1);
        if (lane < W) w = lane + 1 < W ? up : value;
        float thr;
        const bool trig = trigger_window_decide(lane, w, W, coef, value, thr);
        if (lane == 0) {
            flags[i] = trig ? 1 : 0;
            thresholds[i] = thr;
        }
    }
}

// What the step decodes and which recurrent state survives it (:289-296): a triggered step decodes the GP sample and does
// NOT step the LSTM, otherwise `generation` steps it and decodes its output.  vec[b][d] = flag ? sample[d][b] : h_pred[b][d];
// state_out[k] = flag ? state_old[k] : state_new[k] for the n_state tensors of `elems` floats each.
struct TriggerSelectParams {
    const int* flag;
    const float* sample;   // (D, B)
    const float* h_pred;   // (B, D)
    float* vec;            // (B, D)
    int D, B, n_state;
    long elems;
    const float* s_old[8];
    const float* s_new[8];
    float* s_out[8];
};

__global__ __launch_bounds__(256) void gp_trigger_select_kernel(const TriggerSelectParams p) {
    const bool trig = *p.flag != 0;
    const long i = blockIdx.x * 256L + threadIdx.x;
    const long nv = (long)p.B * p.D;
    if (i < nv) {
        const int b = (int)(i / p.D), d = (int)(i % p.D);
        p.vec[i] = trig ? p.sample[(size_t)d * p.B + b] : p.h_pred[i];
    }
    for (int k = 0; k < p.n_state; ++k)
        if (i < p.elems) p.s_out[k][i] = trig ? p.s_old[k][i] : p.s_new[k][i];
}


// ---- the frame losses of train_model and their gradient in one pass (train.py:227-239) ---------------------------------------------
// pred [S][K][n] (per time step the K decoder calls x_pred, x_target_pred, x_pred_gp in the reference's call order, :227-232),
// target [S][n] (the ground-truth frame of the step).  sums[k] = sum over steps and elements of (pred - target)^2 - nn.MSELoss
// per call, summed over the steps, times n - and dpred = 2 w[k] (pred - target): the gradient of sum_k w[k] sums[k], i.e. of
// the frame terms of `loss` (:239) with w[k] = weight_k / n.  One read of pred / target, one write of dpred; per-workgroup
// partial sums in a fixed order (deterministic), reduced by frame_losses_finish_kernel.
template <int K>
__global__ __launch_bounds__(256) void frame_losses_kernel(const float* __restrict__ pred, const float* __restrict__ target,
                                                           float* __restrict__ dpred, float* __restrict__ partial, long n4, int S,
                                                           const float* __restrict__ w) {
    __shared__ float red[4][K];
    const long total = (long)S * n4;
    float acc[K];
    float w2[K];
#pragma unroll
    for (int k = 0; k < K; ++k) { acc[k] = 0.f; w2[k] = 2.f * w[k]; }
    for (long i = blockIdx.x * 256L + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
        const long s = i / n4, e = i % n4;
        const f32x4 t = reinterpret_cast<const f32x4*>(target)[i];
#pragma unroll
        for (int k = 0; k < K; ++k) {
            const long j = (s * K + k) * n4 + e;
            const f32x4 p = reinterpret_cast<const f32x4*>(pred)[j];
            f32x4 d, g;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                d[q] = p[q] - t[q];
                acc[k] = fmaf(d[q], d[q], acc[k]);
                g[q] = w2[k] * d[q];
            }
            reinterpret_cast<f32x4*>(dpred)[j] = g;
        }
    }
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
    for (int k = 0; k < K; ++k) {
        float v = acc[k];
#pragma unroll
        for (int o = 32; o >= 1; o >>= 1) v += __shfl_xor(v, o);
        if (lane == 0) red[wave][k] = v;
    }
    __syncthreads();
    if (threadIdx.x < K) partial[(size_t)blockIdx.x * K + threadIdx.x] = (red[0][threadIdx.x] + red[1][threadIdx.x]) + (red[2][threadIdx.x] + red[3][threadIdx.x]);
}

__global__ __launch_bounds__(256) void frame_losses_finish_kernel(const float* __restrict__ partial, int blocks, int K,
                                                                  float* __restrict__ sums) {
    __shared__ double red[256];
    for (int k = 0; k < K; ++k) {
        double a = 0.0;
        for (int b = threadIdx.x; b < blocks; b += 256) a += (double)partial[(size_t)b * K + k];
        red[threadIdx.x] = a;
        __syncthreads();
        for (int o = 128; o >= 1; o >>= 1) {
            if ((int)threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o];
            __syncthreads();
        }
        if (threadIdx.x == 0) sums[k] = (float)red[0];
        __syncthreads();
    }
}

// sum (a - b)^2 and da = 2 scale (a - b) for a SMALL pair of tensors (the latent MSE of a closure: S x B x g_dim values,
// train.py:188,223): one workgroup, fixed summation order.
__global__ __launch_bounds__(1024) void mse_sum_grad_kernel(const float* __restrict__ a, const float* __restrict__ b,
                                                            float* __restrict__ sum, float* __restrict__ da, long n, float scale) {
    __shared__ double red[1024];
    double acc = 0.0;
    for (long i = threadIdx.x; i < n; i += 1024) {
        const float d = a[i] - b[i];
        acc += (double)d * (double)d;
        if (da) da[i] = 2.f * scale * d;
    }
    red[threadIdx.x] = acc;
    __syncthreads();
    for (int o = 512; o >= 1; o >>= 1) {
        if ((int)threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o];
        __syncthreads();
    }
    if (threadIdx.x == 0) *sum = (float)red[0];
}

}  // namespace dvg

using namespace dvg;

extern "C" int dvg_abi_version(void) { return 9; }  // 9: + dvg_build_info, sync-BN partial rows (r06); 8: y_from (skip tensors of part of a batch), r05 entry points; 3: first igemm schedule retired; 4: + dvg_winograd_wgrad_*; 5: + dvg_gp_elbo(_bwd); 6: GP kernels fp64-internal, + dvg_gp_(bwd_)precision; 7: blocked packed weights, dvg_mfma_mode / dvg_packed_row_floats
extern "C" const char* dvg_last_error(void) { return err_buf(); }

extern "C" long dvg_stream_capture_id(void* stream) {
    hipStreamCaptureStatus st = hipStreamCaptureStatusNone;
    unsigned long long id = 0;
    if (hipStreamGetCaptureInfo(static_cast<hipStream_t>(stream), &st, &id) != hipSuccess) {
        (void)hipGetLastError();
        return 0;
    }
    return st == hipStreamCaptureStatusActive ? (long)(id + 1) : 0;
}

#define PACK_ENTRY(NAME, TR, UN, A0, A1)                                                                        \
    extern "C" int NAME(const float* src, float* dst, int A0, int A1, int kh, int kw, void* stream) {          \
        DVG_REQUIRE(src && dst, DVG_ERR_NULL, #NAME ": NULL pointer");                                          \
        DVG_REQUIRE(cout > 0 && cin > 0 && kh > 0 && kw > 0, DVG_ERR_SHAPE, #NAME ": bad shape");               \
        const long total = (long)cout * cin * kh * kw;                                                          \
        hipLaunchKernelGGL((pack_weight_kernel<TR, UN>), dim3(grid_for(total)), dim3(256), 0, (hipStream_t)stream, \
                           src, dst, cout, cin, kh, kw);                                                        \
        return check_launch(#NAME);                                                                             \
    }
PACK_ENTRY(dvg_pack_conv_weight, false, false, cout, cin)
PACK_ENTRY(dvg_pack_convT_weight, true, false, cin, cout)
PACK_ENTRY(dvg_unpack_conv_weight, false, true, cout, cin)
PACK_ENTRY(dvg_unpack_convT_weight, true, true, cin, cout)

extern "C" int dvg_nchw_to_nhwc(const float* x, float* y, int N, int C, int H, int W, void* stream) {
    DVG_REQUIRE(x && y, DVG_ERR_NULL, "dvg_nchw_to_nhwc: NULL pointer");
    DVG_REQUIRE(N > 0 && C > 0 && H > 0 && W > 0 && N < 65536, DVG_ERR_SHAPE, "dvg_nchw_to_nhwc: bad shape");
    const int HW = H * W;
    hipLaunchKernelGGL((layout_kernel<true>), dim3((HW + 31) / 32, (C + 31) / 32, N), dim3(256), 0, (hipStream_t)stream,
                       x, y, C, HW);
    return check_launch("dvg_nchw_to_nhwc");
}
extern "C" int dvg_nhwc_to_nchw(const float* x, float* y, int N, int C, int H, int W, void* stream) {
    DVG_REQUIRE(x && y, DVG_ERR_NULL, "dvg_nhwc_to_nchw: NULL pointer");
    DVG_REQUIRE(N > 0 && C > 0 && H > 0 && W > 0 && N < 65536, DVG_ERR_SHAPE, "dvg_nhwc_to_nchw: bad shape");
    const int HW = H * W;
    hipLaunchKernelGGL((layout_kernel<false>), dim3((HW + 31) / 32, (C + 31) / 32, N), dim3(256), 0,
                       (hipStream_t)stream, x, y, C, HW);
    return check_launch("dvg_nhwc_to_nchw");
}

extern "C" int dvg_channel_stats_rows(long rows) {
    // number of partial rows dvg_channel_stats writes for a [rows][C] tensor
    long rpb = (rows + 1023) / 1024;
    if (rpb < 16) rpb = 16;
    return (int)((rows + rpb - 1) / rpb);
}

extern "C" int dvg_channel_stats(const float* u, float* stats_partial, long rows, int C, int groups, void* stream) {
    DVG_REQUIRE(u && stats_partial, DVG_ERR_NULL, "dvg_channel_stats: NULL pointer");
    DVG_REQUIRE(rows > 0 && C > 0 && groups > 0, DVG_ERR_SHAPE, "dvg_channel_stats: bad shape");
    long rpb = (rows + 1023) / 1024;
    if (rpb < 16) rpb = 16;
    const int nblk = (int)((rows + rpb - 1) / rpb);
    hipLaunchKernelGGL(channel_stats_kernel, dim3((unsigned)nblk * groups), dim3(C >= 256 ? 256 : (C > 64 ? 128 : 64)), 0,
                       (hipStream_t)stream, u, stats_partial, rows, C, (int)rpb, nblk);
    return check_launch("dvg_channel_stats");
}

extern "C" int dvg_bn_finalize(const float* stats_partial, int nrows, const float* gamma, const float* beta,
                               float* scale, float* shift, float* running_mean, float* running_var,
                               float* save_mean, float* save_invstd, int C, double count, float eps, float momentum,
                               int64_t* num_batches_tracked, int nbt_inc, int groups, float* group_var, void* stream) {
    DVG_REQUIRE(stats_partial && scale && shift, DVG_ERR_NULL, "dvg_bn_finalize: NULL pointer");
    DVG_REQUIRE(C > 0 && nrows > 0 && count >= 1.0 && groups > 0 && groups < 65536, DVG_ERR_SHAPE, "dvg_bn_finalize: bad shape");
    DVG_REQUIRE(groups == 1 || (running_mean == nullptr && running_var == nullptr && num_batches_tracked == nullptr),
                DVG_ERR_SHAPE, "dvg_bn_finalize: with several groups the running statistics are dvg_bn_running_update's job");
    hipLaunchKernelGGL(bn_finalize_kernel, dim3((C + DVG_COLSUM_CT - 1) / DVG_COLSUM_CT, groups), dim3(1024), 0,
                       (hipStream_t)stream, stats_partial, nrows, gamma, beta, scale, shift, running_mean, running_var,
                       save_mean, save_invstd, C, count, eps, momentum, (long long*)num_batches_tracked, nbt_inc, group_var);
    return check_launch("dvg_bn_finalize");
}

extern "C" int dvg_bn_running_update(const float* mean, const float* unbiased_var, int groups, int C, float mom_first,
                                     float mom_mid, float mom_last, float* running_mean, float* running_var,
                                     int64_t* num_batches_tracked, int nbt_inc, void* stream) {
    DVG_REQUIRE(mean && unbiased_var && running_mean && running_var, DVG_ERR_NULL, "dvg_bn_running_update: NULL pointer");
    DVG_REQUIRE(groups > 0 && C > 0, DVG_ERR_SHAPE, "dvg_bn_running_update: bad shape");
    hipLaunchKernelGGL(bn_running_update_kernel, dim3((C + 63) / 64), dim3(64), 0, (hipStream_t)stream, mean, unbiased_var,
                       groups, C, mom_first, mom_mid, mom_last, running_mean, running_var, (long long*)num_batches_tracked,
                       nbt_inc);
    return check_launch("dvg_bn_running_update");
}

extern "C" int dvg_bn_act_apply(const float* u, const float* scale, const float* shift, float* y, float* y_pool, int N,
                                int H, int W, int C, int act, float slope, int group_images, void* stream) {
    DVG_REQUIRE(u && scale && shift && y, DVG_ERR_NULL, "dvg_bn_act_apply: NULL pointer");
    DVG_REQUIRE(N > 0 && H > 0 && W > 0 && C > 0, DVG_ERR_SHAPE, "dvg_bn_act_apply: bad shape");
    DVG_REQUIRE(group_images >= 0 && (group_images == 0 || N % group_images == 0), DVG_ERR_SHAPE,
                "dvg_bn_act_apply: group_images must divide N");
    if (group_images == N) group_images = 0;   // one group
    if (C % 4 != 0) {
        DVG_REQUIRE(y_pool == nullptr, DVG_ERR_SHAPE, "dvg_bn_act_apply: pool needs C %% 4 == 0");
        const long n = (long)N * H * W * C;
        hipLaunchKernelGGL(bn_act_scalar_kernel, dim3(grid_for(n, 256, 4096)), dim3(256), 0, (hipStream_t)stream, u,
                           scale, shift, y, n, C, act, slope, (long)group_images * H * W * C);
        return check_launch("dvg_bn_act_apply");
    }
    DVG_REQUIRE(aligned16(u) && aligned16(y) && aligned16(scale) && aligned16(shift) && aligned16(y_pool),
                DVG_ERR_ALIGN, "dvg_bn_act_apply: alignment");
    if (y_pool) {
        DVG_REQUIRE(H % 2 == 0 && W % 2 == 0, DVG_ERR_SHAPE, "dvg_bn_act_apply: pool needs even H,W");
        const long total = (long)N * (H / 2) * (W / 2) * (C / 4);
        hipLaunchKernelGGL(bn_act_pool_kernel, dim3(grid_for(total, 256, 4096)), dim3(256), 0, (hipStream_t)stream, u,
                           scale, shift, y, y_pool, N, H, W, C / 4, act, slope, group_images);
    } else {
        const long n4 = (long)N * H * W * (C / 4);
        hipLaunchKernelGGL(bn_act_kernel, dim3(grid_for(n4, 256, 4096)), dim3(256), 0, (hipStream_t)stream, u, scale,
                           shift, y, n4, C / 4, act, slope, (long)group_images * H * W * (C / 4));
    }
    return check_launch("dvg_bn_act_apply");
}

extern "C" int dvg_eval_frames(const float* gt, const float* pred, float* ssim, float* psnr, int n_images, int H, int W,
                               void* stream) {
    DVG_REQUIRE(gt && pred && ssim && psnr, DVG_ERR_NULL, "dvg_eval_frames: NULL pointer");
    DVG_REQUIRE(n_images > 0 && H >= 7 && W >= 7, DVG_ERR_SHAPE, "dvg_eval_frames: images must be at least 7x7");
    const size_t lds = (size_t)2 * H * W * sizeof(float);
    DVG_REQUIRE(lds <= 150 * 1024, DVG_ERR_SHAPE, "dvg_eval_frames: %dx%d does not fit the LDS tile", H, W);
    static bool attr_set = false;
    if (!attr_set) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&eval_frames_kernel),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024);
        if (e != hipSuccess) return fail(DVG_ERR_HIP, "hipFuncSetAttribute: %s", hipGetErrorString(e));
        attr_set = true;
    }
    hipLaunchKernelGGL(eval_frames_kernel, dim3(n_images), dim3(256), lds, (hipStream_t)stream, gt, pred, ssim, psnr, H,
                       W);
    return check_launch("dvg_eval_frames");
}

extern "C" int dvg_moving_mnist_compose(const float* sprites, const int* ids, const int* pos, float* out, int n_sprites,
                                        int T, int B, int num_digits, int image_size, int digit_size, void* stream) {
    DVG_REQUIRE(sprites && ids && pos && out, DVG_ERR_NULL, "dvg_moving_mnist_compose: NULL pointer");
    DVG_REQUIRE(n_sprites > 0 && T > 0 && B > 0 && num_digits > 0 && digit_size > 0 && image_size >= digit_size,
                DVG_ERR_SHAPE, "dvg_moving_mnist_compose: bad shape");
    // ids / pos live in device memory: the kernel clamps ids and bounds-checks every sprite access against pos
    const long total = (long)T * B * image_size * image_size;
    hipLaunchKernelGGL(moving_mnist_compose_kernel, dim3(grid_for(total, 256, 8192)), dim3(256), 0, (hipStream_t)stream,
                       sprites, ids, pos, out, T, B, num_digits, image_size, digit_size, n_sprites);
    return check_launch("dvg_moving_mnist_compose");
}

// GPtrigger_gen's variance norms / threshold decision / branch select on the device (generate_frames.py:227-232,275,283-296):
// no host round trip per step, the 93-step loop is capturable as a hipGraph.
extern "C" int dvg_gp_var_norms(const float* var, float* norms, int D, int B, void* stream) {
    DVG_REQUIRE(var && norms, DVG_ERR_NULL, "dvg_gp_var_norms: NULL pointer");
    DVG_REQUIRE(D > 0 && B > 0, DVG_ERR_SHAPE, "dvg_gp_var_norms: empty shape");
    hipLaunchKernelGGL(gp_var_norms_kernel, dim3((unsigned)B), dim3(64), 0, (hipStream_t)stream, var, norms, D, B);
    return check_launch("dvg_gp_var_norms");
}

extern "C" int dvg_gp_trigger_step(const float* var, int D, int B, int col, float* ctx, int window, float coef, int* flag,
                                   float* values, float* thresholds, int* flags, int slot, void* stream) {
    DVG_REQUIRE(var && ctx && flag && values && thresholds && flags, DVG_ERR_NULL, "dvg_gp_trigger_step: NULL pointer");
    DVG_REQUIRE(D > 0 && B > 0 && col >= 0 && col < B && window > 0 && window <= 64 && slot >= 0, DVG_ERR_SHAPE,
                "dvg_gp_trigger_step: col=%d must index the batch of %d, window=%d in [1, 64]", col, B, window);
    hipLaunchKernelGGL(gp_trigger_step_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, var, D, B, col, ctx, window, coef, flag,
                       values, thresholds, flags, slot);
    return check_launch("dvg_gp_trigger_step");
}

extern "C" int dvg_gp_trigger_replay(const float* values, int n, const float* ctx0, int window, float coef, int* flags,
                                     float* thresholds, void* stream) {
    DVG_REQUIRE(values && ctx0 && flags && thresholds, DVG_ERR_NULL, "dvg_gp_trigger_replay: NULL pointer");
    DVG_REQUIRE(n > 0 && window > 0 && window <= 64, DVG_ERR_SHAPE, "dvg_gp_trigger_replay: n > 0, window in [1, 64]");
    hipLaunchKernelGGL(gp_trigger_replay_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, values, n, ctx0, window, coef, flags,
                       thresholds);
    return check_launch("dvg_gp_trigger_replay");
}

extern "C" int dvg_gp_trigger_select(const int* flag, const float* sample_db, const float* h_pred, float* vec, int D, int B,
                                     int n_state, long state_elems, const float* const* state_old,
                                     const float* const* state_new, float* const* state_out, void* stream) {
    DVG_REQUIRE(flag && sample_db && h_pred && vec, DVG_ERR_NULL, "dvg_gp_trigger_select: NULL pointer");
    DVG_REQUIRE(D > 0 && B > 0 && n_state >= 0 && n_state <= 8 && state_elems >= 0, DVG_ERR_SHAPE,
                "dvg_gp_trigger_select: at most 8 state tensors");
    TriggerSelectParams p{flag, sample_db, h_pred, vec, D, B, n_state, state_elems, {}, {}, {}};
    for (int k = 0; k < n_state; ++k) {
        DVG_REQUIRE(state_old[k] && state_new[k] && state_out[k], DVG_ERR_NULL, "dvg_gp_trigger_select: NULL state pointer");
        p.s_old[k] = state_old[k]; p.s_new[k] = state_new[k]; p.s_out[k] = state_out[k];
    }
    const long n = std::max((long)B * D, state_elems);
    hipLaunchKernelGGL(gp_trigger_select_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, p);
    return check_launch("dvg_gp_trigger_select");
}

// The frame terms of train_model's loss and their gradient (train.py:227-239): see frame_losses_kernel.  pred [S][K][n], target
// [S][n], n % 4 == 0, K in {1, 2, 3}; w[K] on the device (weight_k / n); partial: workspace of dvg_frame_losses_blocks(S * n) * K
// floats; sums[K] = raw sums of squares per call position.
extern "C" int dvg_frame_losses_blocks(long elems) {
    long b = (elems / 4 + 255) / 256;
    return (int)(b > 2048 ? 2048 : (b < 1 ? 1 : b));
}

extern "C" int dvg_frame_losses(const float* pred, const float* target, float* sums, float* dpred, long n, int S, int K,
                                const float* w, float* partial, void* stream) {
    DVG_REQUIRE(pred && target && sums && dpred && w && partial, DVG_ERR_NULL, "dvg_frame_losses: NULL pointer");
    DVG_REQUIRE(n > 0 && n % 4 == 0 && S > 0 && K >= 1 && K <= 3, DVG_ERR_SHAPE, "dvg_frame_losses: n %% 4 == 0, K in 1..3 needed");
    DVG_REQUIRE(aligned16(pred) && aligned16(target) && aligned16(dpred), DVG_ERR_ALIGN, "dvg_frame_losses: alignment");
    const int blocks = dvg_frame_losses_blocks((long)S * n);
    const hipStream_t st = (hipStream_t)stream;
    if (K == 1) hipLaunchKernelGGL(frame_losses_kernel<1>, dim3(blocks), dim3(256), 0, st, pred, target, dpred, partial, n / 4, S, w);
    else if (K == 2) hipLaunchKernelGGL(frame_losses_kernel<2>, dim3(blocks), dim3(256), 0, st, pred, target, dpred, partial, n / 4, S, w);
    else hipLaunchKernelGGL(frame_losses_kernel<3>, dim3(blocks), dim3(256), 0, st, pred, target, dpred, partial, n / 4, S, w);
    if (int e = check_launch("dvg_frame_losses")) return e;
    hipLaunchKernelGGL(frame_losses_finish_kernel, dim3(1), dim3(256), 0, st, partial, blocks, K, sums);
    return check_launch("dvg_frame_losses (finish)");
}

// *sum = sum (a - b)^2, da = 2 scale (a - b) (NULL: no gradient): the latent MSE of a closure and its gradient (train.py:188,
// 223,239); small tensors (one workgroup).
extern "C" int dvg_mse_sum_grad(const float* a, const float* b, float* sum, float* da, long n, float scale, void* stream) {
    DVG_REQUIRE(a && b && sum, DVG_ERR_NULL, "dvg_mse_sum_grad: NULL pointer");
    DVG_REQUIRE(n > 0 && n <= (1L << 24), DVG_ERR_SHAPE, "dvg_mse_sum_grad: 1 <= n <= 2^24 (one workgroup)");
    hipLaunchKernelGGL(mse_sum_grad_kernel, dim3(1), dim3(1024), 0, (hipStream_t)stream, a, b, sum, da, n, scale);
    return check_launch("dvg_mse_sum_grad");
}
