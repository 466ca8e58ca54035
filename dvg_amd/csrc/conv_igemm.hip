// fp32-MFMA implicit-GEMM convolutions for gfx950 (CDNA4), NHWC activations.
//
// One kernel template covers the three dense conv shapes of the DVG encoder /
// decoder (reference call sites in include/dvg_hip.h):
//   MODE_CONV3    Conv2d(3,1,1)            vgg_64.py:8     9 taps
//   MODE_CONV4S2  Conv2d(4,2,1)            dcgan_64.py:8  16 taps, stride 2
//   MODE_CONVT4S2 ConvTranspose2d(4,2,1)   dcgan_64.py:20  4 output-parity classes
//                                                          x 4 taps each
// GEMM view: M = output pixels of the tile, N = Cout, K = taps * Cin.
//
// Workgroup = 256 threads = 4 waves in a 2(M) x 2(N) arrangement; each wave owns
// MT x NT accumulator tiles of 32x32 (v_mfma_f32_32x32x2_f32, exact fp32).
// Per 32-channel K chunk the input tile *with halo* is staged ONCE in LDS and
// re-read at shifted addresses for every tap, so each input element is fetched
// from L2/HBM once per chunk instead of once per tap; the weight tile of the
// next tap is prefetched into registers while the current tap's MFMAs issue.
// The epilogue fuses bias+BatchNorm(eval) scale/shift, the activation, the 2x2
// max-pool (in-register: all four pool partners live in one lane's accumulator)
// and the per-channel sum / sum-of-squares needed by train-mode BatchNorm.
#include "dvg_common.h"

namespace dvg {

enum { MODE_CONV3 = 0, MODE_CONV4S2 = 1, MODE_CONVT4S2 = 2 };

struct IgemmParams {
    const float* x;      // primary input  (N, H>>up, W>>up, C1)
    const float* skip;   // concat input   (N, H, W, C2) or nullptr
    const float* w;      // packed weights [taps][Cout][C1+C2]
    const float* scale;  // [Cout] or nullptr
    const float* shift;  // [Cout] or nullptr
    float* y;            // output NHWC
    float* y_pool;       // pooled output or nullptr (MODE_CONV3 only)
    float* stats;        // per-tile partial [rows][2][Cout] or nullptr
    int N, H, W;         // logical INPUT grid (for CONV3 = output grid)
    int C1, C2, Cout;
    int upsample;        // x is at half resolution (nearest x2)
    int act;
    float slope;
    int tiles_y, tiles_x, tiles_n, nblk_n;
};

template <int MODE, int TI, int TH, int TW, int BN>
struct Cfg {
    static constexpr int S = (MODE == MODE_CONV4S2) ? 2 : 1;
    static constexpr int SPAN = (MODE == MODE_CONV4S2) ? 4 : 3;
    static constexpr int HH = (TH - 1) * S + SPAN;
    static constexpr int HW = (TW - 1) * S + SPAN;
    static constexpr int HP = TI * HH * HW;  // halo pixels staged per chunk
    static constexpr int BM = TI * TH * TW;
    static constexpr int MT = BM / 64;
    static constexpr int NT = BN / 64;
    static constexpr int NTAPS = (MODE == MODE_CONV3) ? 9 : (MODE == MODE_CONV4S2 ? 16 : 4);
    static constexpr int KC = 32;   // channels per K chunk
    static constexpr int LD = 36;   // padded LDS row (floats): 144 B keeps b128 reads spread over banks
    static constexpr int A_FLOATS = HP * LD;
    static constexpr int B_FLOATS = BN * LD;
    static constexpr int NLA = (HP * 8 + 255) / 256;  // float4 loads per thread for the A halo
    static constexpr int NLB = (BN * 8) / 256;        // float4 loads per thread for a B tile
    static constexpr int LDS_BYTES = (A_FLOATS + 2 * B_FLOATS) * 4;
    static_assert(BM == 64 || BM == 128, "BM");
    static_assert(BN == 64 || BN == 128, "BN");
};

template <int MODE, int TI, int TH, int TW, int BN>
__global__ __launch_bounds__(256, 2) void conv_igemm_kernel(const IgemmParams p) {
    using C = Cfg<MODE, TI, TH, TW, BN>;
    constexpr int S = C::S, HH = C::HH, HW = C::HW, HP = C::HP, LD = C::LD;
    constexpr int MT = C::MT, NT = C::NT, NTAPS = C::NTAPS;

    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* As = smem;
    float* Bs = smem + C::A_FLOATS;

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int l31 = lane & 31, hh = lane >> 5;

    // ---- which tile am I ---------------------------------------------------
    unsigned lid = xcd_remap(blockIdx.x, gridDim.x);
    int par = 0;
    if (MODE == MODE_CONVT4S2) { par = lid & 3; lid >>= 2; }
    const int nb = lid % p.nblk_n;
    unsigned t = lid / p.nblk_n;
    const int tx_i = t % p.tiles_x; t /= p.tiles_x;
    const int ty_i = t % p.tiles_y; t /= p.tiles_y;
    const int n0 = (int)t * TI;
    const int y0 = ty_i * TH, x0 = tx_i * TW;  // tile origin in the OUTPUT-side grid of the mode
    const int yin0 = y0 * S - 1, xin0 = x0 * S - 1;
    const int nb0 = nb * BN;
    const int py = par >> 1, px = par & 1;
    const int Cin = p.C1 + p.C2;

    // ---- per-lane LDS bases --------------------------------------------------
    int a_base[MT];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
        const int m = wm * (C::BM / 2) + mt * 32 + l31;
        const int ti = m / (TH * TW), r = m % (TH * TW);
        const int ty = r / TW, tx = r % TW;
        a_base[mt] = ((ti * HH + ty * S) * HW + tx * S) * LD + hh * 16;
    }
    int b_base[NT];
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) b_base[nt] = (wn * (BN / 2) + nt * 32 + l31) * LD + hh * 16;

    f32x16 acc[MT][NT];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int nt = 0; nt < NT; ++nt)
#pragma unroll
            for (int i = 0; i < 16; ++i) acc[mt][nt][i] = 0.f;

    // ---- loaders ---------------------------------------------------------------
    auto load_a = [&](int c0) {
        const float* src;
        int Cs, cc, sh;
        if (c0 < p.C1) { src = p.x; Cs = p.C1; cc = c0; sh = p.upsample; }
        else { src = p.skip; Cs = p.C2; cc = c0 - p.C1; sh = 0; }
        const int Hs = p.H >> sh, Ws = p.W >> sh;
        f32x4 v[C::NLA];
#pragma unroll
        for (int i = 0; i < C::NLA; ++i) {
            const int idx = tid + i * 256;
            const int hp = idx >> 3, q = idx & 7;
            const int ti = hp / (HH * HW), r = hp % (HH * HW);
            const int hy = r / HW, hx = r % HW;
            const int n = n0 + ti, yy = yin0 + hy, xx = xin0 + hx;
            v[i] = f32x4{0.f, 0.f, 0.f, 0.f};
            if (idx < HP * 8 && n < p.N && (unsigned)yy < (unsigned)p.H && (unsigned)xx < (unsigned)p.W) {
                const size_t off = (((size_t)n * Hs + (yy >> sh)) * Ws + (xx >> sh)) * Cs + cc + q * 4;
                v[i] = *reinterpret_cast<const f32x4*>(src + off);
            }
        }
#pragma unroll
        for (int i = 0; i < C::NLA; ++i) {
            const int idx = tid + i * 256;
            if (idx < HP * 8) *reinterpret_cast<f32x4*>(&As[(idx >> 3) * LD + (idx & 7) * 4]) = v[i];
        }
    };
    auto tap_w = [&](int tap) -> int {
        if (MODE == MODE_CONVT4S2) {
            const int a = tap >> 1, b = tap & 1;
            return (2 + py - 2 * a) * 4 + (2 + px - 2 * b);
        }
        return tap;
    };
    auto tap_lds = [&](int tap) -> int {
        int th, tw;
        if (MODE == MODE_CONV3) { th = tap / 3; tw = tap % 3; }
        else if (MODE == MODE_CONV4S2) { th = tap >> 2; tw = tap & 3; }
        else { th = 1 + py - (tap >> 1); tw = 1 + px - (tap & 1); }
        return (th * HW + tw) * LD;
    };
    auto load_b = [&](int step, f32x4 (&r)[C::NLB]) {
        const int chunk = step / NTAPS, tap = step % NTAPS;
        const float* wp = p.w + ((size_t)tap_w(tap) * p.Cout + nb0) * Cin + chunk * C::KC;
#pragma unroll
        for (int i = 0; i < C::NLB; ++i) {
            const int idx = tid + i * 256;
            r[i] = *reinterpret_cast<const f32x4*>(wp + (size_t)(idx >> 3) * Cin + (idx & 7) * 4);
        }
    };
    auto store_b = [&](int buf, const f32x4 (&r)[C::NLB]) {
        float* Bb = Bs + buf * C::B_FLOATS;
#pragma unroll
        for (int i = 0; i < C::NLB; ++i) {
            const int idx = tid + i * 256;
            *reinterpret_cast<f32x4*>(&Bb[(idx >> 3) * LD + (idx & 7) * 4]) = r[i];
        }
    };

    // ---- main loop over (chunk, tap) -----------------------------------------------
    const int nsteps = (Cin / C::KC) * NTAPS;
    {
        f32x4 r[C::NLB];
        load_b(0, r);
        load_a(0);
        store_b(0, r);
    }
    __syncthreads();

    for (int step = 0; step < nsteps; ++step) {
        const bool has_next = step + 1 < nsteps;
        f32x4 rnext[C::NLB];
        if (has_next) load_b(step + 1, rnext);

        const int tap = step % NTAPS;
        const float* Bb = Bs + (step & 1) * C::B_FLOATS;
        const int aoff = tap_lds(tap);
#pragma unroll
        for (int half = 0; half < 2; ++half) {
            f32x4 a[MT][2], b[NT][2];
#pragma unroll
            for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                for (int j = 0; j < 2; ++j)
                    a[mt][j] = *reinterpret_cast<const f32x4*>(&As[a_base[mt] + aoff + half * 8 + j * 4]);
#pragma unroll
            for (int nt = 0; nt < NT; ++nt)
#pragma unroll
                for (int j = 0; j < 2; ++j)
                    b[nt][j] = *reinterpret_cast<const f32x4*>(&Bb[b_base[nt] + half * 8 + j * 4]);
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int e = 0; e < 4; ++e)
#pragma unroll
                    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                        for (int nt = 0; nt < NT; ++nt)
                            acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[mt][j][e], b[nt][j][e],
                                                                              acc[mt][nt], 0, 0, 0);
        }

        if (has_next) {
            if ((step + 1) % NTAPS == 0) {
                __syncthreads();  // every wave is done with this chunk's halo tile
                load_a(((step + 1) / NTAPS) * C::KC);
            }
            store_b((step + 1) & 1, rnext);
        }
        __syncthreads();
    }

    // ---- epilogue --------------------------------------------------------------------
    int Ho, Wo;
    if (MODE == MODE_CONV3) { Ho = p.H; Wo = p.W; }
    else if (MODE == MODE_CONV4S2) { Ho = p.H >> 1; Wo = p.W >> 1; }
    else { Ho = p.H * 2; Wo = p.W * 2; }

    float s1[NT], s2[NT];
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) { s1[nt] = 0.f; s2[nt] = 0.f; }

#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
        const int c = nb0 + wn * (BN / 2) + nt * 32 + l31;
        const float sc = p.scale ? p.scale[c] : 1.f;
        const float sf = p.shift ? p.shift[c] : 0.f;
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
            float v[16];
#pragma unroll
            for (int reg = 0; reg < 16; ++reg) {
                const float u = acc[mt][nt][reg] * sc + sf;
                v[reg] = u;
            }
            const int mbase = wm * (C::BM / 2) + mt * 32;
            const int ti = mbase / (TH * TW);  // a 32-row MFMA tile never straddles images when TH*TW >= 32;
                                               // for TH*TW == 16 (4x4 maps) it covers two images, handled below
#pragma unroll
            for (int reg = 0; reg < 16; ++reg) {
                const int row = (reg & 3) + 8 * (reg >> 2) + 4 * hh;
                const int m = mbase + row;
                const int tii = (TH * TW >= 32) ? ti : m / (TH * TW);
                const int r = m % (TH * TW);
                const int ty = r / TW, tx = r % TW;
                const int n = n0 + tii;
                if (n < p.N) {
                    s1[nt] += v[reg];
                    s2[nt] += v[reg] * v[reg];
                    const float o = apply_act(v[reg], p.act, p.slope);
                    v[reg] = o;
                    int oy, ox;
                    if (MODE == MODE_CONVT4S2) { oy = 2 * (y0 + ty) + py; ox = 2 * (x0 + tx) + px; }
                    else { oy = y0 + ty; ox = x0 + tx; }
                    p.y[(((size_t)n * Ho + oy) * Wo + ox) * p.Cout + c] = o;
                }
            }
            if (MODE == MODE_CONV3 && (TW == 16 || TW == 8)) {
                if (p.y_pool != nullptr) {
                    constexpr int RY = (TW == 16) ? 8 : 4;  // register distance of the +1-row pool partner
#pragma unroll
                    for (int reg = 0; reg < 16; ++reg) {
                        const bool ty_even = (TW == 16) ? ((reg >> 2) < 2) : (((reg >> 2) & 1) == 0);
                        if ((reg & 1) == 0 && ty_even) {
                            const int row = (reg & 3) + 8 * (reg >> 2) + 4 * hh;
                            const int m = mbase + row;
                            const int r = m % (TH * TW);
                            const int ty = r / TW, tx = r % TW;
                            const float mx = fmaxf(fmaxf(v[reg], v[reg + 1]), fmaxf(v[reg + RY], v[reg + RY + 1]));
                            const int n = n0 + ti;
                            if (n < p.N)
                                p.y_pool[(((size_t)n * (Ho >> 1) + ((y0 + ty) >> 1)) * (Wo >> 1) + ((x0 + tx) >> 1)) *
                                             p.Cout + c] = mx;
                        }
                    }
                }
            }
        }
    }

    if (p.stats != nullptr) {
        // column sums: fold the two lane halves, then the two M-waves through LDS
        // (deterministic: one partial row per tile, reduced later by dvg_bn_finalize).
        float* red = smem;  // main loop is over: LDS is free
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
            s1[nt] += __shfl_xor(s1[nt], 32);
            s2[nt] += __shfl_xor(s2[nt], 32);
        }
        if (wm == 1 && hh == 0) {
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) {
                red[(wn * NT + nt) * 64 + l31] = s1[nt];
                red[(wn * NT + nt) * 64 + 32 + l31] = s2[nt];
            }
        }
        __syncthreads();
        if (wm == 0 && hh == 0) {
            const unsigned rowid = (MODE == MODE_CONVT4S2 ? (lid / p.nblk_n) * 4 + par : lid / p.nblk_n);
            float* dst = p.stats + (size_t)rowid * 2 * p.Cout;
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) {
                const int c = nb0 + wn * (BN / 2) + nt * 32 + l31;
                dst[c] = s1[nt] + red[(wn * NT + nt) * 64 + l31];
                dst[p.Cout + c] = s2[nt] + red[(wn * NT + nt) * 64 + 32 + l31];
            }
        }
    }
}

// ---------------------------------------------------------------------------------
// host-side dispatch
// ---------------------------------------------------------------------------------
template <int MODE, int TI, int TH, int TW, int BN>
static int launch(IgemmParams p, int Hg, int Wg, hipStream_t stream) {
    using C = Cfg<MODE, TI, TH, TW, BN>;
    // Hg,Wg: the grid the tiles cover (output grid for CONV3/CONV4S2, input grid for CONVT)
    if (Hg % TH || Wg % TW || p.Cout % BN) return fail(DVG_ERR_SHAPE, "conv_igemm: tile does not divide shape");
    p.tiles_y = Hg / TH;
    p.tiles_x = Wg / TW;
    p.tiles_n = (p.N + TI - 1) / TI;
    p.nblk_n = p.Cout / BN;
    const unsigned grid = (unsigned)p.tiles_y * p.tiles_x * p.tiles_n * p.nblk_n * (MODE == MODE_CONVT4S2 ? 4 : 1);
    static bool attr_set = false;
    if (!attr_set) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_igemm_kernel<MODE, TI, TH, TW, BN>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, C::LDS_BYTES);
        if (e != hipSuccess) return fail(DVG_ERR_HIP, "hipFuncSetAttribute: %s", hipGetErrorString(e));
        attr_set = true;
    }
    hipLaunchKernelGGL((conv_igemm_kernel<MODE, TI, TH, TW, BN>), dim3(grid), dim3(256), C::LDS_BYTES, stream, p);
    return check_launch("conv_igemm");
}

// number of stats partial rows a given call produces (mirrors the dispatch below)
static int tile_choice_bm(int mode, int N, int Hg, int Wg, int Cout, int* ti, int* th, int* tw, int* bn) {
    // spatial tile
    if (Hg >= 8 && Wg >= 16 && Wg % 16 == 0 && mode != MODE_CONV4S2) { *ti = 1; *th = 8; *tw = 16; }
    else if (Hg >= 8 && Wg >= 8) { *ti = 1; *th = 8; *tw = 8; }
    else if (Hg == 4 && Wg == 4) { *ti = 4; *th = 4; *tw = 4; }
    else return -1;
    const int bm = (*ti) * (*th) * (*tw);
    const long mtiles = (long)((N + *ti - 1) / *ti) * (Hg / *th) * (Wg / *tw);
    // pick BN so that the launch has >= ~2 workgroups per CU when possible
    *bn = 128;
    if (Cout % 128 != 0 || mtiles * (Cout / 128) * (mode == MODE_CONVT4S2 ? 4 : 1) < 512) *bn = 64;
    // when the 8x16 tile leaves too few workgroups, fall back to 8x8 (BM=64)
    if (bm == 128 && mtiles * (Cout / *bn) * (mode == MODE_CONVT4S2 ? 4 : 1) < 512 && mode != MODE_CONV4S2) {
        *tw = 8;
    }
    return 0;
}

}  // namespace dvg

using namespace dvg;

extern "C" int dvg_conv_stats_rows(int mode, int N, int H, int W, int Cout) {
    // H,W = logical input grid, as in the conv entry points
    int Hg = H, Wg = W;
    if (mode == MODE_CONV4S2) { Hg = H / 2; Wg = W / 2; }
    int ti, th, tw, bn;
    if (tile_choice_bm(mode, N, Hg, Wg, Cout, &ti, &th, &tw, &bn)) return -1;
    return ((N + ti - 1) / ti) * (Hg / th) * (Wg / tw) * (mode == MODE_CONVT4S2 ? 4 : 1);
}

#define DVG_DISPATCH(MODE, TI_, TH_, TW_, BN_)                                   \
    if (ti == TI_ && th == TH_ && tw == TW_ && bn == BN_)                       \
        return launch<MODE, TI_, TH_, TW_, BN_>(p, Hg, Wg, (hipStream_t)stream);

static int common_checks(const IgemmParams& p, const char* who) {
    DVG_REQUIRE(p.x && p.w && p.y, DVG_ERR_NULL, "%s: x/w/y must not be NULL", who);
    DVG_REQUIRE((p.skip != nullptr) == (p.C2 > 0), DVG_ERR_SHAPE, "%s: skip pointer / C2 mismatch", who);
    DVG_REQUIRE(p.N > 0 && p.H > 0 && p.W > 0, DVG_ERR_SHAPE, "%s: empty shape", who);
    DVG_REQUIRE(p.C1 > 0 && p.C1 % 32 == 0 && p.C2 % 32 == 0, DVG_ERR_SHAPE,
                "%s: C1=%d C2=%d must be multiples of 32", who, p.C1, p.C2);
    DVG_REQUIRE(p.Cout > 0 && p.Cout % 64 == 0, DVG_ERR_SHAPE, "%s: Cout=%d must be a multiple of 64", who, p.Cout);
    DVG_REQUIRE(aligned16(p.x) && aligned16(p.w) && aligned16(p.y) && aligned16(p.skip), DVG_ERR_ALIGN,
                "%s: pointers must be 16-byte aligned", who);
    DVG_REQUIRE(p.act >= DVG_ACT_NONE && p.act <= DVG_ACT_SIGMOID, DVG_ERR_SHAPE, "%s: bad act", who);
    return DVG_OK;
}

extern "C" int dvg_conv3x3_bn_act(const float* x, const float* skip, const float* w_packed, const float* scale,
                                  const float* shift, float* y, float* y_pool, float* stats, int N, int H, int W,
                                  int C1, int C2, int Cout, int upsample_x, int act, float slope, void* stream) {
    IgemmParams p{x, skip, w_packed, scale, shift, y, y_pool, stats, N, H, W, C1, C2, Cout, upsample_x ? 1 : 0, act, slope,
                  0, 0, 0, 0};
    if (int e = common_checks(p, "dvg_conv3x3_bn_act")) return e;
    DVG_REQUIRE(H % 8 == 0 && W % 8 == 0, DVG_ERR_SHAPE, "dvg_conv3x3_bn_act: H=%d W=%d must be multiples of 8", H, W);
    int Hg = H, Wg = W, ti, th, tw, bn;
    DVG_REQUIRE(tile_choice_bm(MODE_CONV3, N, Hg, Wg, Cout, &ti, &th, &tw, &bn) == 0, DVG_ERR_SHAPE,
                "dvg_conv3x3_bn_act: no tile for %dx%d", H, W);
    DVG_DISPATCH(MODE_CONV3, 1, 8, 16, 128)
    DVG_DISPATCH(MODE_CONV3, 1, 8, 16, 64)
    DVG_DISPATCH(MODE_CONV3, 1, 8, 8, 128)
    DVG_DISPATCH(MODE_CONV3, 1, 8, 8, 64)
    return fail(DVG_ERR_SHAPE, "dvg_conv3x3_bn_act: no kernel for tile (%d,%d,%d,%d)", ti, th, tw, bn);
}

extern "C" int dvg_conv4x4s2_bn_act(const float* x, const float* w_packed, const float* scale, const float* shift,
                                    float* y, float* stats, int N, int H, int W, int Cin, int Cout, int act,
                                    float slope, void* stream) {
    IgemmParams p{x, nullptr, w_packed, scale, shift, y, nullptr, stats, N, H, W, Cin, 0, Cout, 0, act, slope, 0, 0, 0, 0};
    if (int e = common_checks(p, "dvg_conv4x4s2_bn_act")) return e;
    DVG_REQUIRE(H % 2 == 0 && W % 2 == 0, DVG_ERR_SHAPE, "dvg_conv4x4s2_bn_act: odd input");
    int Hg = H / 2, Wg = W / 2, ti, th, tw, bn;
    DVG_REQUIRE(tile_choice_bm(MODE_CONV4S2, N, Hg, Wg, Cout, &ti, &th, &tw, &bn) == 0 && Hg % th == 0 && Wg % tw == 0,
                DVG_ERR_SHAPE, "dvg_conv4x4s2_bn_act: unsupported map %dx%d", H, W);
    DVG_DISPATCH(MODE_CONV4S2, 1, 8, 8, 128)
    DVG_DISPATCH(MODE_CONV4S2, 1, 8, 8, 64)
    DVG_DISPATCH(MODE_CONV4S2, 4, 4, 4, 128)
    DVG_DISPATCH(MODE_CONV4S2, 4, 4, 4, 64)
    return fail(DVG_ERR_SHAPE, "dvg_conv4x4s2_bn_act: no kernel for tile (%d,%d,%d,%d)", ti, th, tw, bn);
}

extern "C" int dvg_convT4x4s2_bn_act(const float* x, const float* skip, const float* w_packed, const float* scale,
                                     const float* shift, float* y, float* stats, int N, int H, int W, int C1, int C2,
                                     int Cout, int act, float slope, void* stream) {
    IgemmParams p{x, skip, w_packed, scale, shift, y, nullptr, stats, N, H, W, C1, C2, Cout, 0, act, slope, 0, 0, 0, 0};
    if (int e = common_checks(p, "dvg_convT4x4s2_bn_act")) return e;
    int Hg = H, Wg = W, ti, th, tw, bn;
    DVG_REQUIRE(tile_choice_bm(MODE_CONVT4S2, N, Hg, Wg, Cout, &ti, &th, &tw, &bn) == 0 && Hg % th == 0 && Wg % tw == 0,
                DVG_ERR_SHAPE, "dvg_convT4x4s2_bn_act: unsupported map %dx%d", H, W);
    DVG_DISPATCH(MODE_CONVT4S2, 1, 8, 16, 128)
    DVG_DISPATCH(MODE_CONVT4S2, 1, 8, 16, 64)
    DVG_DISPATCH(MODE_CONVT4S2, 1, 8, 8, 128)
    DVG_DISPATCH(MODE_CONVT4S2, 1, 8, 8, 64)
    DVG_DISPATCH(MODE_CONVT4S2, 4, 4, 4, 128)
    DVG_DISPATCH(MODE_CONVT4S2, 4, 4, 4, 64)
    return fail(DVG_ERR_SHAPE, "dvg_convT4x4s2_bn_act: no kernel for tile (%d,%d,%d,%d)", ti, th, tw, bn);
}
