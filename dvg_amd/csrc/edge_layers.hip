// The thin layers at both ends of the encoder / decoder: 1..4 image channels on
// one side of the convolution.  They move a (N,nc,H,W) frame tile <-> a 64+ channel
// NHWC activation and are HBM-bound (arithmetic intensity 4-8 FLOP/B), so they are
// written as direct convolutions on the VALU with coalesced 256-B wave accesses:
//   first layers: lane = output channel (one pixel = one 256-B store per wave),
//                 input taps are LDS broadcasts;
//   last layers:  lane = pixel, channels staged through LDS in chunks of 16.
#include "dvg_common.h"

// Output stores of the first layer are nontemporal (the 67 MB activation is read back once, by a kernel that is not
// memory-bound): 4.14 vs 3.93 TB/s at B = 64 (same-box A/B, tools/ab_variants.sh); grid cap 512 / 1024 / 2048:
// 3.57 / 4.14 / 4.14 TB/s.
#ifndef DVG_FIRST_NT
#define DVG_FIRST_NT 1
#endif
#ifndef DVG_FIRST_GRID
#define DVG_FIRST_GRID 1024
#endif

namespace dvg {

// ---------------------------------------------------------------------------
// First layer: Conv2d(nc, Cout, KS, S, 1) + affine + act,  x NCHW -> y NHWC
//   KS=3,S=1: vgg_64.py:23      KS=4,S=2: dcgan_64.py:34
// ---------------------------------------------------------------------------
template <int KS, int S>
__global__ __launch_bounds__(256) void conv_first_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                         const float* __restrict__ scale,
                                                         const float* __restrict__ shift, float* __restrict__ y,
                                                         float* __restrict__ stats, int N, int H, int W, int nc,
                                                         int Cout, int act, float slope) {
    // Output tile per workgroup: 4 rows (one per wave) x 32 columns x 64 channels.
    // lane = (pixel sub-index 0..3) x (channel quad 0..15): a lane owns 4 consecutive output channels of one
    // pixel, so the store is 16 B per lane and a wave writes 4 pixels = 1 KiB contiguous NHWC bytes; the
    // KS*KS*nc input taps are LDS reads with 4 distinct addresses per wave (broadcast within a quad group).
    // Workgroups are persistent over tiles (grid-stride): the next tile's few input floats are prefetched into
    // registers while the current tile computes, so the global->LDS latency is paid once per workgroup.
    constexpr int TH = 4, TW = 32;
    constexpr int HH = (TH - 1) * S + KS, HW = (TW - 1) * S + KS;
    constexpr int NT = KS * KS;
    constexpr int NPRE = (4 * HH * HW + 255) / 256;
    __shared__ float tiles[2][4 * HH * HW];
    __shared__ __attribute__((aligned(16))) float wl[4 * NT * 64];  // weights [ci*NT+tap][64 channels of this group]
    __shared__ float red[2 * 4 * 64];
    const int Ho = (H + 2 - KS) / S + 1, Wo = (W + 2 - KS) / S + 1;
    const int tiles_x = (Wo + TW - 1) / TW, tiles_y = (Ho + TH - 1) / TH;
    const int ntiles = N * tiles_y * tiles_x;
    const int cg = blockIdx.y;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int sub = lane >> 4, cq = lane & 15;
    const int c0 = cg * 64 + cq * 4;
    const f32x4 sc = scale ? *reinterpret_cast<const f32x4*>(scale + c0) : f32x4{1.f, 1.f, 1.f, 1.f};
    const f32x4 sf = shift ? *reinterpret_cast<const f32x4*>(shift + c0) : f32x4{0.f, 0.f, 0.f, 0.f};

    // the 64 x (nc*NT) weights of this channel group stay in LDS for the whole workgroup lifetime
    for (int i = threadIdx.x; i < nc * NT * 64; i += 256) {
        const int c = i & 63, q = i >> 6;
        wl[q * 64 + c] = w[(size_t)(cg * 64 + c) * nc * NT + q];
    }

    float pre[NPRE];
    auto prefetch = [&](int tile) {
        int t = tile;
        const int tx_i = t % tiles_x; t /= tiles_x;
        const int ty_i = t % tiles_y;
        const int n = t / tiles_y;
        const int iy0 = ty_i * TH * S - 1, ix0 = tx_i * TW * S - 1;
#pragma unroll
        for (int j = 0; j < NPRE; ++j) {
            const int i = threadIdx.x + j * 256;
            const int ci = i / (HH * HW), r = i % (HH * HW);
            const int yy = iy0 + r / HW, xx = ix0 + r % HW;
            pre[j] = 0.f;
            if (i < nc * HH * HW && (unsigned)yy < (unsigned)H && (unsigned)xx < (unsigned)W)
                pre[j] = x[(((size_t)n * nc + ci) * H + yy) * W + xx];
        }
    };
    int buf = 0;
    if ((int)blockIdx.x < ntiles) prefetch(blockIdx.x);
    const bool lrelu = act == DVG_ACT_LRELU;  // the common case, kept off the generic (tanh / exp) switch
    for (int tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        float* tl = tiles[buf];
#pragma unroll
        for (int j = 0; j < NPRE; ++j) {
            const int i = threadIdx.x + j * 256;
            if (i < nc * HH * HW) tl[i] = pre[j];
        }
        __syncthreads();
        if (tile + (int)gridDim.x < ntiles) prefetch(tile + gridDim.x);

        int t = tile;
        const int tx_i = t % tiles_x; t /= tiles_x;
        const int ty_i = t % tiles_y;
        const int n = t / tiles_y;
        const int oy0 = ty_i * TH, ox0 = tx_i * TW;
        f32x4 s1 = {0.f, 0.f, 0.f, 0.f}, s2 = {0.f, 0.f, 0.f, 0.f};
        const int oy = oy0 + wave;
        if (oy < Ho) {
            // two passes of 4 pixels per lane: 16 accumulators live at a time keeps the kernel at <= 64 VGPRs
            // (8 waves/SIMD): this layer is HBM-write-bound and wants occupancy, not ILP
#pragma unroll 1
            for (int half = 0; half < 2; ++half) {
                f32x4 acc[TW / 8];
#pragma unroll
                for (int it = 0; it < TW / 8; ++it) acc[it] = f32x4{0.f, 0.f, 0.f, 0.f};
                for (int ci = 0; ci < nc; ++ci) {
#pragma unroll 1
                    for (int a = 0; a < KS; ++a) {
                        const float* row = tl + ci * HH * HW + (wave * S + a) * HW + (half * 16 + sub) * S;
                        const float* wrow = &wl[(ci * NT + a * KS) * 64 + cq * 4];
#pragma unroll
                        for (int b = 0; b < KS; ++b) {
                            const f32x4 wv = *reinterpret_cast<const f32x4*>(wrow + b * 64);
#pragma unroll
                            for (int it = 0; it < TW / 8; ++it) {
                                const float v = row[it * 4 * S + b];
#pragma unroll
                                for (int k = 0; k < 4; ++k) acc[it][k] = fmaf(v, wv[k], acc[it][k]);
                            }
                        }
                    }
                }
#pragma unroll
                for (int it = 0; it < TW / 8; ++it) {
                    const int ox = ox0 + half * 16 + it * 4 + sub;
                    if (ox < Wo) {
                        f32x4 o;
#pragma unroll
                        for (int k = 0; k < 4; ++k) {
                            const float u = acc[it][k] * sc[k] + sf[k];
                            s1[k] += u;
                            s2[k] += u * u;
                            o[k] = lrelu ? (u > 0.f ? u : u * slope) : apply_act(u, act, slope);
                        }
                        f32x4* dst = reinterpret_cast<f32x4*>(y + (((size_t)n * Ho + oy) * Wo + ox) * Cout + c0);
#if DVG_FIRST_NT
                        __builtin_nontemporal_store(o, dst);
#else
                        *dst = o;
#endif
                    }
                }
            }
        }
        if (stats != nullptr) {
#pragma unroll
            for (int k = 0; k < 4; ++k) {  // fold the 4 pixel sub-lanes that share a channel quad
                s1[k] += __shfl_xor(s1[k], 16);
                s1[k] += __shfl_xor(s1[k], 32);
                s2[k] += __shfl_xor(s2[k], 16);
                s2[k] += __shfl_xor(s2[k], 32);
            }
            __syncthreads();  // red[] of the previous tile has been consumed
            if (sub == 0) {
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    red[wave * 64 + cq * 4 + k] = s1[k];
                    red[256 + wave * 64 + cq * 4 + k] = s2[k];
                }
            }
            __syncthreads();
            if (wave == 0) {
                const int c = cg * 64 + lane;
                float* dst = stats + (size_t)tile * 2 * Cout;
                dst[c] = red[lane] + red[64 + lane] + red[128 + lane] + red[192 + lane];
                dst[Cout + c] = red[256 + lane] + red[320 + lane] + red[384 + lane] + red[448 + lane];
            }
        }
        buf ^= 1;
    }
}

// ---------------------------------------------------------------------------
// Last layer: ConvTranspose2d(C1+C2, nc, KS, S, 1) + bias + act on cat([x, skip]),
// x/skip NHWC -> y NCHW.
//   KS=3,S=1: vgg_64.py:88-92 (Sigmoid)     KS=4,S=2: dcgan_64.py:75-79 (Tanh)
// One thread per INPUT-grid pixel; for S=2 it produces the 2x2 output quad so
// that the weight of every (parity, tap) is wave-uniform (scalar loads).
// ---------------------------------------------------------------------------
template <int KS, int S>
__global__ __launch_bounds__(256) void convT_last_kernel(const float* __restrict__ x, const float* __restrict__ skip,
                                                         const float* __restrict__ w,
                                                         const float* __restrict__ bias, float* __restrict__ y, int N,
                                                         int H, int W, int C1, int C2, int nc, int act) {
    constexpr int TH = 8, TW = 32, HH = TH + 2, HW = TW + 2, CC = 16, LD = 20;  // 80-B rows: conflict-free b128
    constexpr int NOUT = S * S;
    __shared__ __attribute__((aligned(16))) float tile[HH * HW * LD];
    const int tiles_x = (W + TW - 1) / TW, tiles_y = (H + TH - 1) / TH;
    int t = blockIdx.x;
    const int tx_i = t % tiles_x; t /= tiles_x;
    const int ty_i = t % tiles_y;
    const int n = t / tiles_y;
    const int y0 = ty_i * TH, x0 = tx_i * TW;
    const int ly = threadIdx.x >> 5, lx = threadIdx.x & 31;
    const int Cin = C1 + C2;

    float acc[NOUT][4];
#pragma unroll
    for (int o = 0; o < NOUT; ++o)
#pragma unroll
        for (int co = 0; co < 4; ++co) acc[o][co] = 0.f;

    for (int c0 = 0; c0 < Cin; c0 += CC) {
        const float* src = c0 < C1 ? x : skip;
        const int Cs = c0 < C1 ? C1 : C2, cc = c0 < C1 ? c0 : c0 - C1;
        __syncthreads();
        for (int i = threadIdx.x; i < HH * HW * 4; i += 256) {
            const int hp = i >> 2, q = i & 3;
            const int yy = y0 - 1 + hp / HW, xx = x0 - 1 + hp % HW;
            f32x4 v = {0.f, 0.f, 0.f, 0.f};
            if ((unsigned)yy < (unsigned)H && (unsigned)xx < (unsigned)W)
                v = *reinterpret_cast<const f32x4*>(src + (((size_t)n * H + yy) * W + xx) * Cs + cc + q * 4);
            *reinterpret_cast<f32x4*>(&tile[hp * LD + q * 4]) = v;
        }
        __syncthreads();
        // weights w[ci][co][kh][kw] (ORIGINAL ConvTranspose2d layout)
        if (S == 1) {
            // out[y][x] = sum_{kh,kw} in[y+1-kh][x+1-kw] * w[kh][kw]
#pragma unroll
            for (int kh = 0; kh < 3; ++kh)
#pragma unroll
                for (int kw = 0; kw < 3; ++kw) {
                    const float* tp = &tile[((ly + 2 - kh) * HW + (lx + 2 - kw)) * LD];
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        const f32x4 v = *reinterpret_cast<const f32x4*>(tp + q * 4);
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            const float* wp = w + ((size_t)(c0 + q * 4 + e) * nc) * 9 + kh * 3 + kw;
#pragma unroll
                            for (int co = 0; co < 4; ++co)
                                if (co < nc) acc[0][co] = fmaf(v[e], wp[co * 9], acc[0][co]);
                        }
                    }
                }
        } else {
            // out[2q+py] uses input rows iy = q+py-a with kernel row kh = 1-py+2a  (a = 0,1)
#pragma unroll
            for (int py = 0; py < 2; ++py)
#pragma unroll
                for (int px = 0; px < 2; ++px)
#pragma unroll
                    for (int a = 0; a < 2; ++a)
#pragma unroll
                        for (int b = 0; b < 2; ++b) {
                            const int kh = 1 - py + 2 * a, kw = 1 - px + 2 * b;
                            const float* tp = &tile[((ly + 1 + py - a) * HW + (lx + 1 + px - b)) * LD];
#pragma unroll
                            for (int q = 0; q < 4; ++q) {
                                const f32x4 v = *reinterpret_cast<const f32x4*>(tp + q * 4);
#pragma unroll
                                for (int e = 0; e < 4; ++e) {
                                    const float* wp = w + ((size_t)(c0 + q * 4 + e) * nc) * 16 + kh * 4 + kw;
#pragma unroll
                                    for (int co = 0; co < 4; ++co)
                                        if (co < nc)
                                            acc[py * 2 + px][co] = fmaf(v[e], wp[co * 16], acc[py * 2 + px][co]);
                                }
                            }
                        }
        }
    }
    const int yy = y0 + ly, xx = x0 + lx;
    if (yy < H && xx < W) {
        const int Ho = H * S, Wo = W * S;
#pragma unroll
        for (int co = 0; co < 4; ++co) {
            if (co < nc) {
                const float b = bias ? bias[co] : 0.f;
#pragma unroll
                for (int o = 0; o < NOUT; ++o) {
                    const int oy = yy * S + (o >> 1) * (S - 1), ox = xx * S + (o & 1) * (S - 1);
                    y[(((size_t)n * nc + co) * Ho + oy) * Wo + ox] = apply_act(acc[o][co] + b, act, 0.f);
                }
            }
        }
    }
}


// ---------------------------------------------------------------------------
// Last layers, two-step form.  The transposed conv to nc <= 4 channels is a per-pixel projection followed by a
// shifted sum:   d[p][(kh,kw,co)] = sum_ci x[p][ci] * W[ci][co][kh][kw]      (small-M-style GEMM, reads x ONCE)
//                y[co][oy][ox]    = b[co] + sum_{(kh,kw) hitting (oy,ox)} d[source pixel][(kh,kw,co)]
// The GEMM runs on dvg_gemm_nt_bias_act (one call per concatenated input); this kernel is the shifted sum.
// It replaces the direct kernel above on the inference path: the direct form gives one wave per SIMD ~2 k serial
// FMAs + 2 k scalar weight loads (140 us for dcgan_64's last layer against a 7 us HBM floor).
// ---------------------------------------------------------------------------
template <int KS, int S>
__global__ void convT_gather_kernel(const float* __restrict__ d1, const float* __restrict__ d2,
                                    const float* __restrict__ bias, float* __restrict__ y, int N, int H, int W, int nc,
                                    int act, const int* __restrict__ d2_map, int d2_B) {
    const int T = KS * KS * nc;
    const long total = (long)N * H * W;
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int r = i % W;
        long t = i / W;
        const int q = t % H;
        const int n = t / H;
        // d2 (the skip tensor's projection) may hold shared blocks of d2_B images: image n reads block d2_map[n / d2_B]
        const long d2_shift = d2_map ? ((long)d2_map[n / d2_B] * d2_B + n % d2_B - n) * H * W * T : 0;
        const int Ho = H * S, Wo = W * S;
#pragma unroll
        for (int co = 0; co < 4; ++co) {
            if (co >= nc) break;
            const float b = bias ? bias[co] : 0.f;
            if (S == 1) {
                float acc = b;
#pragma unroll
                for (int kh = 0; kh < 3; ++kh)
#pragma unroll
                    for (int kw = 0; kw < 3; ++kw) {
                        const int yy = q + 1 - kh, xx = r + 1 - kw;
                        if ((unsigned)yy < (unsigned)H && (unsigned)xx < (unsigned)W) {
                            const size_t o = (((size_t)n * H + yy) * W + xx) * T + (kh * 3 + kw) * nc + co;
                            acc += d1[o] + (d2 ? d2[(long)o + d2_shift] : 0.f);
                        }
                    }
                y[(((size_t)n * nc + co) * Ho + q) * Wo + r] = apply_act(acc, act, 0.f);
            } else {
#pragma unroll
                for (int py = 0; py < 2; ++py)
#pragma unroll
                    for (int px = 0; px < 2; ++px) {
                        float acc = b;
#pragma unroll
                        for (int a = 0; a < 2; ++a)
#pragma unroll
                            for (int bb = 0; bb < 2; ++bb) {
                                const int yy = q + py - a, xx = r + px - bb;
                                const int kh = 1 - py + 2 * a, kw = 1 - px + 2 * bb;
                                if ((unsigned)yy < (unsigned)H && (unsigned)xx < (unsigned)W) {
                                    const size_t o = (((size_t)n * H + yy) * W + xx) * T + (kh * 4 + kw) * nc + co;
                                    acc += d1[o] + (d2 ? d2[(long)o + d2_shift] : 0.f);
                                }
                            }
                        y[(((size_t)n * nc + co) * Ho + 2 * q + py) * Wo + 2 * r + px] = apply_act(acc, act, 0.f);
                    }
            }
        }
    }
}


// ---------------------------------------------------------------------------
// pixel_proj: d[px][t] = sum_c in[px][c] * w[t][c]   (first step of the two-step last layer)
//
// The per-pixel projection of the decoder's last ConvTranspose (vgg_64.py:90-93 / dcgan_64.py:75-79) onto its
// T = ks*ks*nc tap outputs.  HBM-bound: the activation (B*H*W x C fp32, 67 MB for vgg_64 at B = 64) is read exactly
// once, 16 pixels x 16 channels per wave-instruction straight into MFMA A fragments (no LDS); the T x C weight
// lives in registers as B fragments.  v_mfma_f32_16x16x4_f32: lane l holds A[i = l%16][k = l/16]; a float4 load
// of channels 16j + 4*(l/16) .. +3 feeds the four MFMAs of channel block j (the K order inside a block is
// permuted identically for A and B).  D: lane l holds rows 4*(l/16) + r, column l%16.
// ---------------------------------------------------------------------------
typedef float f32x4_u __attribute__((ext_vector_type(4), aligned(4)));

template <int CB, int NT>
__global__ __launch_bounds__(256) void pixel_proj_kernel(const float* __restrict__ in, const float* __restrict__ w,
                                                         float* __restrict__ out, long P, int T) {
    constexpr int C = CB * 16, U = 4;
    const int lane = threadIdx.x & 63, i16 = lane & 15, kq = lane >> 4;
    float bf[NT][CB][4];
#pragma unroll
    for (int nt = 0; nt < NT; ++nt)
#pragma unroll
        for (int j = 0; j < CB; ++j) {
            const int t = nt * 16 + i16;
            f32x4 v = f32x4{0.f, 0.f, 0.f, 0.f};
            if (t < T) v = *reinterpret_cast<const f32x4_u*>(w + (size_t)t * C + 16 * j + 4 * kq);
#pragma unroll
            for (int e = 0; e < 4; ++e) bf[nt][j][e] = v[e];
        }
    const long ntiles = P >> 4;
    const long wave_id = ((long)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const long nwaves = ((long)gridDim.x * blockDim.x) >> 6;
    for (long t0 = wave_id * U; t0 < ntiles; t0 += nwaves * U) {
        f32x4 a[U][CB];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const long tile = (t0 + u < ntiles) ? t0 + u : ntiles - 1;   // clamped: the duplicate is not stored
            const float* src = in + ((tile << 4) + i16) * C + 4 * kq;
#pragma unroll
            for (int j = 0; j < CB; ++j) a[u][j] = *reinterpret_cast<const f32x4*>(src + 16 * j);
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            f32x4 acc[NT];
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) acc[nt] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int j = 0; j < CB; ++j)
#pragma unroll
                for (int e = 0; e < 4; ++e)
#pragma unroll
                    for (int nt = 0; nt < NT; ++nt)
                        acc[nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[u][j][e], bf[nt][j][e], acc[nt], 0, 0, 0);
            if (t0 + u < ntiles) {
                const long px0 = ((t0 + u) << 4) + 4 * kq;
#pragma unroll
                for (int nt = 0; nt < NT; ++nt) {
                    const int col = nt * 16 + i16;
                    if (col < T) {
#pragma unroll
                        for (int r = 0; r < 4; ++r) out[(px0 + r) * T + col] = acc[nt][r];
                    }
                }
            }
        }
    }
}

}  // namespace dvg

using namespace dvg;

static int first_checks(const char* who, const float* x, const float* w, float* y, int N, int H, int W, int nc,
                        int Cout, int act) {
    DVG_REQUIRE(x && w && y, DVG_ERR_NULL, "%s: NULL pointer", who);
    DVG_REQUIRE(N > 0 && H > 0 && W > 0, DVG_ERR_SHAPE, "%s: empty shape", who);
    DVG_REQUIRE(nc >= 1 && nc <= 4, DVG_ERR_SHAPE, "%s: nc=%d must be 1..4", who, nc);
    DVG_REQUIRE(Cout > 0 && Cout % 64 == 0, DVG_ERR_SHAPE, "%s: Cout=%d must be a multiple of 64", who, Cout);
    DVG_REQUIRE(act >= 0 && act <= 3, DVG_ERR_SHAPE, "%s: bad act", who);
    return DVG_OK;
}

extern "C" int dvg_conv_first_stats_rows(int ks, int N, int H, int W) {
    const int S = ks == 4 ? 2 : 1;
    const int Ho = (H + 2 - ks) / S + 1, Wo = (W + 2 - ks) / S + 1;
    return N * ((Ho + 3) / 4) * ((Wo + 31) / 32);
}

extern "C" int dvg_conv3x3_first(const float* x, const float* w, const float* scale, const float* shift, float* y,
                                 float* stats, int N, int H, int W, int nc, int Cout, int act, float slope,
                                 void* stream) {
    if (int e = first_checks("dvg_conv3x3_first", x, w, y, N, H, W, nc, Cout, act)) return e;
    unsigned gx = (unsigned)N * ((H + 3) / 4) * ((W + 31) / 32);
    if (gx > DVG_FIRST_GRID) gx = DVG_FIRST_GRID;  // persistent over tiles: 4 workgroups per CU
    hipLaunchKernelGGL((conv_first_kernel<3, 1>), dim3(gx, Cout / 64), dim3(256), 0, (hipStream_t)stream, x, w, scale,
                       shift, y, stats, N, H, W, nc, Cout, act, slope);
    return check_launch("dvg_conv3x3_first");
}

extern "C" int dvg_conv4x4s2_first(const float* x, const float* w, const float* scale, const float* shift, float* y,
                                   float* stats, int N, int H, int W, int nc, int Cout, int act, float slope,
                                   void* stream) {
    if (int e = first_checks("dvg_conv4x4s2_first", x, w, y, N, H, W, nc, Cout, act)) return e;
    DVG_REQUIRE(H % 2 == 0 && W % 2 == 0, DVG_ERR_SHAPE, "dvg_conv4x4s2_first: odd input");
    const int Ho = H / 2, Wo = W / 2;
    unsigned gx = (unsigned)N * ((Ho + 3) / 4) * ((Wo + 31) / 32);
    if (gx > 1024) gx = 1024;
    hipLaunchKernelGGL((conv_first_kernel<4, 2>), dim3(gx, Cout / 64), dim3(256), 0, (hipStream_t)stream, x, w, scale,
                       shift, y, stats, N, H, W, nc, Cout, act, slope);
    return check_launch("dvg_conv4x4s2_first");
}

static int last_checks(const char* who, const float* x, const float* skip, const float* w, float* y, int N, int H,
                       int W, int C1, int C2, int nc, int act) {
    DVG_REQUIRE(x && w && y, DVG_ERR_NULL, "%s: NULL pointer", who);
    DVG_REQUIRE((skip != nullptr) == (C2 > 0), DVG_ERR_SHAPE, "%s: skip pointer / C2 mismatch", who);
    DVG_REQUIRE(N > 0 && H > 0 && W > 0, DVG_ERR_SHAPE, "%s: empty shape", who);
    DVG_REQUIRE(nc >= 1 && nc <= 4, DVG_ERR_SHAPE, "%s: nc=%d must be 1..4", who, nc);
    DVG_REQUIRE(C1 > 0 && C1 % 16 == 0 && C2 % 16 == 0, DVG_ERR_SHAPE, "%s: C1=%d C2=%d must be multiples of 16", who,
                C1, C2);
    DVG_REQUIRE(aligned16(x) && aligned16(skip), DVG_ERR_ALIGN, "%s: alignment", who);
    DVG_REQUIRE(act >= 0 && act <= 3, DVG_ERR_SHAPE, "%s: bad act", who);
    return DVG_OK;
}

extern "C" int dvg_convT3x3_last(const float* x, const float* w, const float* bias, float* y, int N, int H, int W,
                                 int Cin, int nc, int act, void* stream) {
    if (int e = last_checks("dvg_convT3x3_last", x, nullptr, w, y, N, H, W, Cin, 0, nc, act)) return e;
    const unsigned gx = (unsigned)N * ((H + 7) / 8) * ((W + 31) / 32);
    hipLaunchKernelGGL((convT_last_kernel<3, 1>), dim3(gx), dim3(256), 0, (hipStream_t)stream, x, nullptr, w, bias, y,
                       N, H, W, Cin, 0, nc, act);
    return check_launch("dvg_convT3x3_last");
}

extern "C" int dvg_convT4x4s2_last(const float* x, const float* skip, const float* w, const float* bias, float* y,
                                   int N, int H, int W, int C1, int C2, int nc, int act, void* stream) {
    if (int e = last_checks("dvg_convT4x4s2_last", x, skip, w, y, N, H, W, C1, C2, nc, act)) return e;
    const unsigned gx = (unsigned)N * ((H + 7) / 8) * ((W + 31) / 32);
    hipLaunchKernelGGL((convT_last_kernel<4, 2>), dim3(gx), dim3(256), 0, (hipStream_t)stream, x, skip, w, bias, y, N,
                       H, W, C1, C2, nc, act);
    return check_launch("dvg_convT4x4s2_last");
}

extern "C" int dvg_convT_gather(const float* d1, const float* d2, const float* bias, float* y_nchw, int ks, int N, int H,
                                int W, int nc, int act, const int* d2_map, int d2_block, void* stream) {
    DVG_REQUIRE(d1 && y_nchw, DVG_ERR_NULL, "dvg_convT_gather: NULL pointer");
    DVG_REQUIRE(N > 0 && H > 0 && W > 0 && nc >= 1 && nc <= 4, DVG_ERR_SHAPE, "dvg_convT_gather: bad shape");
    if (d2 == nullptr) d2_map = nullptr;
    DVG_REQUIRE(d2_map == nullptr || (d2_block > 0 && N % d2_block == 0), DVG_ERR_SHAPE, "dvg_convT_gather: d2_block must divide N");
    DVG_REQUIRE(ks == 3 || ks == 4, DVG_ERR_SHAPE, "dvg_convT_gather: ks must be 3 or 4");
    DVG_REQUIRE(act >= 0 && act <= 3, DVG_ERR_SHAPE, "dvg_convT_gather: bad act");
    const long total = (long)N * H * W;
    long g = (total + 255) / 256;
    if (g > 4096) g = 4096;
    if (ks == 3)
        hipLaunchKernelGGL((convT_gather_kernel<3, 1>), dim3((unsigned)g), dim3(256), 0, (hipStream_t)stream, d1, d2, bias,
                           y_nchw, N, H, W, nc, act, d2_map, d2_block);
    else
        hipLaunchKernelGGL((convT_gather_kernel<4, 2>), dim3((unsigned)g), dim3(256), 0, (hipStream_t)stream, d1, d2, bias,
                           y_nchw, N, H, W, nc, act, d2_map, d2_block);
    return check_launch("dvg_convT_gather");
}

extern "C" int dvg_pixel_proj(const float* in, const float* w, float* out, long P, int C, int T, void* stream) {
    DVG_REQUIRE(in && w && out, DVG_ERR_NULL, "dvg_pixel_proj: NULL pointer");
    DVG_REQUIRE(P > 0 && P % 16 == 0, DVG_ERR_SHAPE, "dvg_pixel_proj: P=%ld must be a positive multiple of 16", P);
    DVG_REQUIRE(C == 64 || C == 128, DVG_ERR_SHAPE, "dvg_pixel_proj: C=%d must be 64 or 128", C);
    DVG_REQUIRE(T >= 1 && T <= 48, DVG_ERR_SHAPE, "dvg_pixel_proj: T=%d must be in 1..48", T);
    DVG_REQUIRE(aligned16(in), DVG_ERR_ALIGN, "dvg_pixel_proj: input must be 16-byte aligned");
    const int nt = (T + 15) / 16;
    const long tiles = P / 16;
    long g = (tiles + 15) / 16;   // 4 waves x 4 tiles per workgroup pass
    if (g > 1024) g = 1024;
    if (g < 1) g = 1;
    const dim3 grid((unsigned)g), block(256);
    hipStream_t s = (hipStream_t)stream;
#define DVG_PP(CB_, NT_) hipLaunchKernelGGL((pixel_proj_kernel<CB_, NT_>), grid, block, 0, s, in, w, out, P, T)
    if (C == 64) {
        if (nt == 1) DVG_PP(4, 1); else if (nt == 2) DVG_PP(4, 2); else DVG_PP(4, 3);
    } else {
        if (nt == 1) DVG_PP(8, 1); else if (nt == 2) DVG_PP(8, 2); else DVG_PP(8, 3);
    }
#undef DVG_PP
    return check_launch("dvg_pixel_proj");
}
