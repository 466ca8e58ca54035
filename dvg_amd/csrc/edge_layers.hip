// The thin layers at both ends of the encoder / decoder: 1..4 image channels on
// one side of the convolution.  They move a (N,nc,H,W) frame tile <-> a 64+ channel
// NHWC activation and are HBM-bound (arithmetic intensity 4-8 FLOP/B), so they are
// written as direct convolutions on the VALU with coalesced 256-B wave accesses:
//   first layers: lane = output channel (one pixel = one 256-B store per wave),
//                 input taps are LDS broadcasts;
//   last layers:  lane = pixel, channels staged through LDS in chunks of 16.
#include "dvg_common.h"

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
    constexpr int TH = 4, TW = 32;  // output tile per workgroup: 4 rows (one per wave) x 32 columns
    constexpr int HH = (TH - 1) * S + KS, HW = (TW - 1) * S + KS;
    __shared__ float tile[4 * HH * HW];
    __shared__ float red[2 * 4 * 64];
    const int Ho = (H + 2 - KS) / S + 1, Wo = (W + 2 - KS) / S + 1;
    const int tiles_x = (Wo + TW - 1) / TW, tiles_y = (Ho + TH - 1) / TH;
    const int cg = blockIdx.y;  // group of 64 output channels
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
        if ((unsigned)yy < (unsigned)H && (unsigned)xx < (unsigned)W) v = x[(((size_t)n * nc + ci) * H + yy) * W + xx];
        tile[i] = v;
    }
    const int c = cg * 64 + lane;
    float wr[4 * KS * KS];
#pragma unroll
    for (int i = 0; i < 4 * KS * KS; ++i) wr[i] = (i < nc * KS * KS) ? w[(size_t)c * nc * KS * KS + i] : 0.f;
    const float sc = scale ? scale[c] : 1.f, sf = shift ? shift[c] : 0.f;
    __syncthreads();

    float s1 = 0.f, s2 = 0.f;
    const int oy = oy0 + wave;
    if (oy < Ho) {
        for (int px = 0; px < TW; ++px) {
            const int ox = ox0 + px;
            if (ox >= Wo) break;
            float acc = 0.f;
#pragma unroll
            for (int ci = 0; ci < 4; ++ci) {  // static register indexing of wr[] (nc <= 4, uniform predicate)
                if (ci < nc) {
                    const float* tp = tile + ci * HH * HW + (wave * S) * HW + px * S;
#pragma unroll
                    for (int a = 0; a < KS; ++a)
#pragma unroll
                        for (int b = 0; b < KS; ++b) acc = fmaf(tp[a * HW + b], wr[ci * KS * KS + a * KS + b], acc);
                }
            }
            const float u = acc * sc + sf;
            s1 += u;
            s2 += u * u;
            y[(((size_t)n * Ho + oy) * Wo + ox) * Cout + c] = apply_act(u, act, slope);
        }
    }
    if (stats != nullptr) {
        red[wave * 64 + lane] = s1;
        red[256 + wave * 64 + lane] = s2;
        __syncthreads();
        if (wave == 0) {
            float* dst = stats + (size_t)blockIdx.x * 2 * Cout;
            dst[c] = red[lane] + red[64 + lane] + red[128 + lane] + red[192 + lane];
            dst[Cout + c] = red[256 + lane] + red[320 + lane] + red[384 + lane] + red[448 + lane];
        }
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
    const unsigned gx = (unsigned)N * ((H + 3) / 4) * ((W + 31) / 32);
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
    const unsigned gx = (unsigned)N * ((Ho + 3) / 4) * ((Wo + 31) / 32);
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
