// fp32-MFMA weight gradient of the dense convolutions (train.py:170,194,240 `loss.backward()` for
// the Conv2d / ConvTranspose2d of vgg_64.py:8, dcgan_64.py:8,20).
//
//   dW[tap][co][ci] = sum over (n, pixel pairs)  dOut[n][out pixel][co] * In[n][in pixel(tap)][ci]
//
// GEMM view per tap: M = Cout tile (64), N = Cin tile (64), K = pixels.  In NHWC both operands are
// "k-major with the M/N index contiguous", which is exactly the f32 MFMA operand layout
// (lane l supplies A[i=l&31][k=l>>5]), so fragments are single ds_read_b32 per lane with unit stride
// across lanes: conflict-free without padding.  A workgroup stages one spatial tile of dOut and the
// matching input tile WITH HALO in LDS and accumulates all taps of its group from it (the dOut
// fragment is shared by every tap).  K is split across workgroups; partial sums go to
// partial[split][tap][Cout][Cin] and are reduced by dvg_reduce_partials (deterministic, no atomics).
// The input loader fuses nearest-upsample + concat exactly like the forward kernel.
//
// DVG_BF16X3 (dvg_common.h, default): the products run on the bf16 matrix pipe as in conv_igemm2.hip - both tiles are split
// into exact bf16 triples when they are written to LDS, [pixel][plane h, m, l][64 channels] (384 B per pixel) - and because K
// is the PIXEL index while the tiles are channel-contiguous, the MFMA operands (8 consecutive k per lane) are fetched with the
// transposing read `ds_read_b64_tr_b16`: a group of 16 lanes reads a block of 4 pixels x 16 channels and each lane receives
// the 4 pixels of one channel.  Two such reads per plane make one v_mfma_f32_32x32x16_bf16 operand; six MFMAs per tap and
// 16 pixels replace eight f32 MFMAs per tap and 2 pixels x 8.
#include "dvg_common.h"

namespace dvg {

enum { W_CONV3 = 0, W_CONV4S2 = 1, W_CONVT4S2 = 2 };

constexpr int WGRAD_MAX_ITEMS = 8;

struct WgradParams {
    // `items` (x, skip, dout) triples of IDENTICAL shape: dW = sum over items of dOut_i (x) In_i.  One launch over several
    // uses of the same layer (the time steps / decoder calls of train.py:213-232 share their weights) makes the GEMM K
    // dimension `items` times longer, i.e. the K-split partial slabs - 75 MB written and read back per launch whatever
    // the layer, 17 % of this kernel family's time at one item per launch - are amortised over `items` uses.
    const float* x[WGRAD_MAX_ITEMS];     // forward input, (N, H>>up, W>>up, C1)
    const float* skip[WGRAD_MAX_ITEMS];  // concat input (N,H,W,C2) or nullptr
    const float* dout[WGRAD_MAX_ITEMS];  // gradient w.r.t. the conv output, NHWC on the output grid
    int items, tiles_item;
    float* partial;     // [S][taps][Cout][Cin]
    int N, H, W;        // forward INPUT grid
    int C1, C2, Cout, upsample;
    int tiles_y, tiles_x, tiles_n, tiles_total, tiles_per_split, S, n_co, n_ci;
    unsigned long long* clk;  // debug only (dvg_debug_set_wgrad_clockbuf): per-workgroup phase cycle sums (4 x u64)
    unsigned clk_cap;
};

static unsigned long long* g_wclk = nullptr;
static unsigned g_wclk_cap = 0;

template <int MODE, int TI, int TH, int TW>
struct WCfg {
    static constexpr int S = (MODE == W_CONV4S2) ? 2 : 1;
    static constexpr int SPAN = (MODE == W_CONV4S2) ? 4 : 3;
    static constexpr int HH = (TH - 1) * S + SPAN, HW = (TW - 1) * S + SPAN;
    static constexpr int HP = TI * HH * HW;
    static constexpr int P = TI * TH * TW;  // pixel pairs per tile (GEMM K per tile)
    static constexpr int NTAPS_ALL = (MODE == W_CONV3) ? 9 : 16;
    static constexpr int GT = (MODE == W_CONV3) ? 9 : (MODE == W_CONV4S2 ? 8 : 4);  // taps per workgroup
    static constexpr int NG = NTAPS_ALL / GT + (MODE == W_CONV3 ? 0 : 0);           // groups: 1, 2, 4
    static constexpr bool X3 = DVG_BF16X3 != 0;
    static constexpr int PIX = X3 ? 96 : 64;                   // floats per pixel of an LDS tile (3 planes x 64 bf16 / 64 floats)
    static constexpr int LDS_BYTES = (P + HP) * PIX * 4;
    static_assert(P % 2 == 0 && (!X3 || (P % 16 == 0 && TW % 4 == 0)), "P");
};

typedef short s16x4_t __attribute__((ext_vector_type(4)));
typedef short s16x8_t __attribute__((ext_vector_type(8)));

template <int MODE, int TI, int TH, int TW>
__global__ __launch_bounds__(256, 2) void wgrad_igemm_kernel(const WgradParams p) {
    using C = WCfg<MODE, TI, TH, TW>;
    constexpr int S = C::S, HH = C::HH, HW = C::HW, HP = C::HP, P = C::P, GT = C::GT, PIX = C::PIX;
    constexpr bool X3 = C::X3;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* Ad = smem;             // [P][PIX]   dOut tile, this co tile
    float* Xh = smem + P * PIX;   // [HP][PIX]  input halo tile, this ci tile

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wr = wave >> 1, wc = wave & 1, l31 = lane & 31, kk = lane >> 5;

    unsigned b = blockIdx.x;
    const int cit = b % p.n_ci; b /= p.n_ci;
    const int cot = b % p.n_co; b /= p.n_co;
    const int grp = b % C::NG; b /= C::NG;
    const int split = b;
    const int ci0 = cit * 64, co0 = cot * 64;
    const int py = grp >> 1, px = grp & 1;  // CONVT: output parity of this group
    const int Cin = p.C1 + p.C2;

    int Ho, Wo;
    if (MODE == W_CONV3) { Ho = p.H; Wo = p.W; }
    else if (MODE == W_CONV4S2) { Ho = p.H >> 1; Wo = p.W >> 1; }
    else { Ho = p.H * 2; Wo = p.W * 2; }

    // which input tensor holds this ci tile
    const bool from_x = ci0 < p.C1;
    int Cs, cc, sh;
    if (from_x) { Cs = p.C1; cc = ci0; sh = p.upsample; }
    else { Cs = p.C2; cc = ci0 - p.C1; sh = 0; }
    const int Hs = p.H >> sh, Ws = p.W >> sh;

    f32x16 acc[GT];
#pragma unroll
    for (int t = 0; t < GT; ++t)
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[t][i] = 0.f;

    const int t_begin = split * p.tiles_per_split;
    const int t_end = min(p.tiles_total, t_begin + p.tiles_per_split);
    constexpr int NLD = (P * 16 + 255) / 256, NLH = (HP * 16 + 255) / 256;
    static_assert(P * 16 % 256 == 0, "dOut tile is a whole number of 256-thread passes");
    const int q4 = (tid & 15) * 4;
    const float* Ap = Ad + wr * 32 + l31;
    const float* Xp = Xh + wc * 32 + l31;
    auto tap_off = [&](int t2) -> int {
        int th, tw;
        if (MODE == W_CONV3) { th = t2 / 3; tw = t2 % 3; }
        else if (MODE == W_CONV4S2) { const int tap = grp * GT + t2; th = tap >> 2; tw = tap & 3; }
        else { th = 1 + py - (t2 >> 1); tw = 1 + px - (t2 & 1); }
        return (th * HW + tw) * (X3 ? 1 : 64);     // X3: in pixels
    };
    auto frag_base = [&](int pp) -> int {
        const int ti = pp / (TH * TW), r = pp % (TH * TW);
        return ((ti * HH + (r / TW) * S) * HW + (r % TW) * S) * (X3 ? 1 : 64);
    };
    // ---- bf16-triple image helpers ---------------------------------------------------------------------------------------
    // byte offset of (pixel r, plane pl, channel c) inside a tile: the two 32-channel halves of a plane row are swapped on
    // pixels with bit 1 set, so that the 4 pixel rows x 64 B a 32-lane half of a transposed read touches (4 consecutive
    // pixels, one 32-channel half) fall into the four 64-byte quarters of the 256-byte bank space
    auto img_off = [&](int r, int pl, int c) -> int {
        return r * 384 + pl * 128 + (((c >> 5) ^ ((r >> 1) & 1)) << 6) + (c & 31) * 2;
    };
    // one MFMA operand (8 consecutive pixels k0 + 8 (lane >> 5) ... of channel cb + (lane & 31)) of plane pl: two transposed
    // reads of 4 pixels x 16 channels per 16-lane group; lane 4 q + p of a group addresses pixel q, channels 4 p .. 4 p + 3
    const int trq = (lane & 15) >> 2, trc = ((lane >> 4) & 1) * 16 + (lane & 3) * 4, trk = (lane >> 5) * 8;
    // r0 / r1: the tile rows (pixels) this lane addresses in the two reads
    auto tr_operand = [&](const float* tile, int r0, int r1, int cb, int pl) -> s16x8_t {
        const char* base = reinterpret_cast<const char*>(tile);
        const s16x4_t lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
            (__attribute__((address_space(3))) s16x4_t*)(base + img_off(r0, pl, cb + trc)));
        const s16x4_t hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
            (__attribute__((address_space(3))) s16x4_t*)(base + img_off(r1, pl, cb + trc)));
        return __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
    };
    unsigned long long c_stage = 0, c_mfma = 0, c_t0 = p.clk ? clock64() : 0, c_mark = c_t0;
    for (int tile = t_begin; tile < t_end; ++tile) {
        const int item = tile / p.tiles_item;     // wave-uniform: the pointers below are scalar loads from the kernel arguments
        const float* const src = from_x ? p.x[item] : p.skip[item];
        const float* const dsrc = p.dout[item];
        int t = tile - item * p.tiles_item;
        const int tx_i = t % p.tiles_x; t /= p.tiles_x;
        const int ty_i = t % p.tiles_y; t /= p.tiles_y;
        const int n0 = t * TI;
        const int y0 = ty_i * TH, x0 = tx_i * TW;  // iteration grid: output grid (CONV3/CONV4S2), input grid (CONVT)
        const int yin0 = y0 * S - 1, xin0 = x0 * S - 1;
        // The tile's global loads are issued back to back, branch-free (out-of-range slots read a clamped, valid
        // address and are zeroed at the LDS write), the first batch BEFORE the barrier that waits for the
        // previous tile's MFMAs.  (As `for` loops with a guarded load each, hipcc emitted load -> vmcnt(0) ->
        // ds_write per iteration: twenty serialized memory round trips per tile.)  Two batches, so that at most
        // NB1 float4s are live next to the 16*GT accumulator registers.
        constexpr int NB1 = (NLD + NLH + 1) / 2;   // slots 0..NB1-1: dOut tile then the first halo slots
        auto load_slot = [&](int i, bool& ok) -> f32x4 {   // i: compile-time after unrolling
            if (i < NLD) {
                const int pp = (tid + i * 256) >> 4;
                const int d_ti = pp / (TH * TW), d_ty = (pp % (TH * TW)) / TW, d_tx = pp % TW;
                const int n = n0 + d_ti;
                ok = n < p.N;
                int oy, ox;
                if (MODE == W_CONVT4S2) { oy = 2 * (y0 + d_ty) + py; ox = 2 * (x0 + d_tx) + px; }
                else { oy = y0 + d_ty; ox = x0 + d_tx; }
                return *reinterpret_cast<const f32x4*>(dsrc + (((size_t)min(n, p.N - 1) * Ho + oy) * Wo + ox) * p.Cout + co0 + q4);
            }
            const int hp = min((tid + (i - NLD) * 256) >> 4, HP - 1);
            const int n = n0 + hp / (HH * HW), yy = yin0 + (hp % (HH * HW)) / HW, xx = xin0 + hp % HW;
            ok = n < p.N && (unsigned)yy < (unsigned)p.H && (unsigned)xx < (unsigned)p.W;
            const int yc = min(max(yy, 0), p.H - 1) >> sh, xc = min(max(xx, 0), p.W - 1) >> sh;
            return *reinterpret_cast<const f32x4*>(src + (((size_t)min(n, p.N - 1) * Hs + yc) * Ws + xc) * Cs + cc + q4);
        };
        auto store_slot = [&](int i, const f32x4& v, bool ok) {
            const f32x4 z = ok ? v : f32x4{0.f, 0.f, 0.f, 0.f};
            const int r = i < NLD ? (tid + i * 256) >> 4 : (tid + (i - NLD) * 256) >> 4;
            if (i >= NLD && !(NLH * 256 == HP * 16 || r < HP)) return;
            float* tile = i < NLD ? Ad : Xh;
            if constexpr (X3) {
                u32x2_t h, m, l;
                unsigned t0, t1, t2;
                bf16x3_split_pair(z[0], z[1], t0, t1, t2);
                h[0] = t0; m[0] = t1; l[0] = t2;
                bf16x3_split_pair(z[2], z[3], t0, t1, t2);
                h[1] = t0; m[1] = t1; l[1] = t2;
                char* d = reinterpret_cast<char*>(tile);
                *reinterpret_cast<u32x2_t*>(d + img_off(r, 0, q4)) = h;
                *reinterpret_cast<u32x2_t*>(d + img_off(r, 1, q4)) = m;
                *reinterpret_cast<u32x2_t*>(d + img_off(r, 2, q4)) = l;
            } else {
                *reinterpret_cast<f32x4*>(&tile[r * 64 + q4]) = z;
            }
        };
        {
            f32x4 r1[NB1];
            bool ok1[NB1];
#pragma unroll
            for (int i = 0; i < NB1; ++i) r1[i] = load_slot(i, ok1[i]);
            __builtin_amdgcn_sched_barrier(0);
            __syncthreads();   // every wave is done with the previous tile's LDS image
#pragma unroll
            for (int i = 0; i < NB1; ++i) store_slot(i, r1[i], ok1[i]);
        }
        {
            constexpr int NB2 = NLD + NLH - NB1;
            f32x4 r2[NB2];
            bool ok2[NB2];
#pragma unroll
            for (int i = 0; i < NB2; ++i) r2[i] = load_slot(NB1 + i, ok2[i]);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int i = 0; i < NB2; ++i) store_slot(NB1 + i, r2[i], ok2[i]);
        }
        __syncthreads();
        if (p.clk && tid == 0) { const unsigned long long now = clock64(); c_stage += now - c_mark; c_mark = now; }
        if constexpr (X3) {
            // k loop over the tile's pixels, 16 per step: three planes of the dOut operand, then per tap three planes of the
            // input operand and the six MFMAs (l,h) (m,m) (h,l) (m,h) (h,m) (h,h)
#pragma unroll 1
            for (int k0 = 0; k0 < P; k0 += 16) {
                // the two pixels this lane addresses (one per read: k0 + 8 (lane >> 5) + {0, 4} + q) and their halo positions
                const int pa0 = k0 + trk + trq, pa1 = pa0 + 4;
                const int hb0 = frag_base(pa0), hb1 = frag_base(pa1);
                s16x8_t a[3];
#pragma unroll
                for (int pl = 0; pl < 3; ++pl) a[pl] = tr_operand(Ad, pa0, pa1, wr * 32, pl);
                s16x8_t bcur[3], bnxt[3];
#pragma unroll
                for (int pl = 0; pl < 3; ++pl) bcur[pl] = tr_operand(Xh, hb0 + tap_off(0), hb1 + tap_off(0), wc * 32, pl);
#pragma unroll
                for (int t2 = 0; t2 < GT; ++t2) {
                    if (t2 + 1 < GT) {
#pragma unroll
                        for (int pl = 0; pl < 3; ++pl)
                            bnxt[pl] = tr_operand(Xh, hb0 + tap_off(t2 + 1), hb1 + tap_off(t2 + 1), wc * 32, pl);
                    }
                    auto mm = [&](int pa, int pb) {
                        acc[t2] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8_t, a[pa]),
                                                                          __builtin_bit_cast(bf16x8_t, bcur[pb]), acc[t2], 0, 0, 0);
                    };
                    mm(2, 0); mm(1, 1); mm(0, 2); mm(1, 0); mm(0, 1); mm(0, 0);
                    if (t2 + 1 < GT) {
#pragma unroll
                        for (int pl = 0; pl < 3; ++pl) bcur[pl] = bnxt[pl];
#pragma unroll
                        for (int r = 0; r < 6; ++r) {     // next tap's six transposed reads, one per MFMA
                            __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
                            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                        }
                    }
                }
            }
            if (p.clk && tid == 0) { const unsigned long long now = clock64(); c_mfma += now - c_mark; c_mark = now; }
            continue;
        }
        // k loop over the tile's pixels, two per MFMA; fragments of step k+1 are read while step k's MFMAs run
        float a_cur = Ap[kk * 64], b_cur[GT];
        {
            const int hb = frag_base(kk);
#pragma unroll
            for (int t2 = 0; t2 < GT; ++t2) b_cur[t2] = Xp[hb + tap_off(t2)];
        }
#pragma unroll 2
        for (int k0 = 0; k0 < P; k0 += 2) {
            const int pn = min(k0 + 2, P - 2) + kk;   // last step re-reads its own fragments (unused)
            const float a_nxt = Ap[pn * 64];
            float b_nxt[GT];
            const int hb = frag_base(pn);
#pragma unroll
            for (int t2 = 0; t2 < GT; ++t2) b_nxt[t2] = Xp[hb + tap_off(t2)];
#pragma unroll
            for (int t2 = 0; t2 < GT; ++t2) acc[t2] = __builtin_amdgcn_mfma_f32_32x32x2f32(a_cur, b_cur[t2], acc[t2], 0, 0, 0);
#pragma unroll
            for (int t2 = 0; t2 < GT; ++t2) {   // one ds_read per MFMA, a full step ahead of its consumer
                __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
            }
            __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
            a_cur = a_nxt;
#pragma unroll
            for (int t2 = 0; t2 < GT; ++t2) b_cur[t2] = b_nxt[t2];
        }
        if (p.clk && tid == 0) { const unsigned long long now = clock64(); c_mfma += now - c_mark; c_mark = now; }
    }
    const unsigned long long c_loop_end = p.clk ? clock64() : 0;
    // write partial[split][tapw][co][ci]
#pragma unroll
    for (int t2 = 0; t2 < GT; ++t2) {
        int tapw;
        if (MODE == W_CONV3) tapw = t2;
        else if (MODE == W_CONV4S2) tapw = grp * GT + t2;
        else tapw = (2 + py - 2 * (t2 >> 1)) * 4 + (2 + px - 2 * (t2 & 1));
        float* dst = p.partial + (((size_t)split * C::NTAPS_ALL + tapw) * p.Cout + co0 + wr * 32) * Cin + ci0 + wc * 32 + l31;
#pragma unroll
        for (int reg = 0; reg < 16; ++reg) {
            const int row = (reg & 3) + 8 * (reg >> 2) + 4 * kk;
            dst[(size_t)row * Cin] = acc[t2][reg];
        }
    }
    if (p.clk && tid == 0 && blockIdx.x < p.clk_cap) {
        unsigned long long* d = p.clk + (size_t)blockIdx.x * 4;
        d[0] = c_stage; d[1] = c_mfma; d[2] = clock64() - c_loop_end; d[3] = (unsigned long long)(t_end - t_begin);
    }
}

template <int MODE, int TI, int TH, int TW>
static int wlaunch(WgradParams p, int Hg, int Wg, hipStream_t stream) {
    using C = WCfg<MODE, TI, TH, TW>;
    if (Hg % TH || Wg % TW) return fail(DVG_ERR_SHAPE, "wgrad: tile does not divide shape");
    p.tiles_y = Hg / TH;
    p.tiles_x = Wg / TW;
    p.tiles_n = (p.N + TI - 1) / TI;
    p.tiles_item = p.tiles_y * p.tiles_x * p.tiles_n;
    p.tiles_total = p.tiles_item * p.items;
    p.tiles_per_split = (p.tiles_total + p.S - 1) / p.S;
    const unsigned grid = (unsigned)p.S * C::NG * p.n_co * p.n_ci;
    static bool attr_set = false;
    if (!attr_set) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&wgrad_igemm_kernel<MODE, TI, TH, TW>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, C::LDS_BYTES);
        if (e != hipSuccess) return fail(DVG_ERR_HIP, "hipFuncSetAttribute: %s", hipGetErrorString(e));
        attr_set = true;
    }
    hipLaunchKernelGGL((wgrad_igemm_kernel<MODE, TI, TH, TW>), dim3(grid), dim3(256), C::LDS_BYTES, stream, p);
    return check_launch("wgrad_igemm");
}

// tile choice on the iteration grid (Hg,Wg); returns tiles_total via out params
static int wgrad_tile(int mode, int Hg, int Wg, int* ti, int* th, int* tw) {
    if (mode == W_CONV4S2) {
        if (Hg >= 4 && Wg >= 8 && Hg % 4 == 0 && Wg % 8 == 0) { *ti = 1; *th = 4; *tw = 8; return 0; }
        if (Hg == 4 && Wg == 4) { *ti = 2; *th = 4; *tw = 4; return 0; }
        return -1;
    }
    // (bf16 triples: the 8 x 16 tile's LDS image is 118 KB, one workgroup per CU: the 8 x 8 tile, 63 KB, everywhere)
    if (!DVG_BF16X3 && Hg % 8 == 0 && Wg % 16 == 0) { *ti = 1; *th = 8; *tw = 16; return 0; }
    // 8x8 maps: one image per tile.  The two-image tile (2,8,8) needs 84 KB of LDS, i.e. ONE workgroup per CU and no
    // interleaving of staging and MFMA phases (94 TF against 112 TF for the other layers).
    if (Hg % 8 == 0 && Wg % 8 == 0) { *ti = 1; *th = 8; *tw = 8; return 0; }
    if (Hg == 4 && Wg == 4) { *ti = 4; *th = 4; *tw = 4; return 0; }
    return -1;
}

}  // namespace dvg

using namespace dvg;

// number of K splits (= leading dimension of `partial`) a wgrad call over `items` same-shape uses will make
extern "C" int dvg_conv_wgrad_splits_multi(int mode, int N, int H, int W, int Cin, int Cout, int items) {
    int Hg = H, Wg = W;
    if (mode == W_CONV4S2) { Hg = H / 2; Wg = W / 2; }
    int ti, th, tw;
    if (wgrad_tile(mode, Hg, Wg, &ti, &th, &tw)) return -1;
    if (Cin % 64 || Cout % 64 || items < 1 || items > WGRAD_MAX_ITEMS) return -1;
    const long tiles = (long)((N + ti - 1) / ti) * (Hg / th) * (Wg / tw) * items;
    const int groups = mode == W_CONV3 ? 1 : (mode == W_CONV4S2 ? 2 : 4);
    const long base = (long)(Cin / 64) * (Cout / 64) * groups;
    // two workgroups fit a CU (78 KB of LDS each): aim at <= 512 workgroups in ONE residency round — a grid of
    // e.g. 704 runs as 2 rounds for 1.4 rounds of work
    long S = 512 / base;
    if (S > tiles) S = tiles;
    if (S < 1) S = 1;
    // every split must own at least one tile
    const long tps = (tiles + S - 1) / S;
    S = (tiles + tps - 1) / tps;
    return (int)S;
}

extern "C" int dvg_conv_wgrad_splits(int mode, int N, int H, int W, int Cin, int Cout) {
    return dvg_conv_wgrad_splits_multi(mode, N, H, W, Cin, Cout, 1);
}

#define W_DISPATCH(MODE, TI_, TH_, TW_) \
    if (ti == TI_ && th == TH_ && tw == TW_) return wlaunch<MODE, TI_, TH_, TW_>(p, Hg, Wg, (hipStream_t)stream);

extern "C" int dvg_conv_wgrad_multi(int mode, int items, const float* const* x, const float* const* skip,
                                    const float* const* dout, float* partial, int N, int H, int W, int C1, int C2,
                                    int Cout, int upsample_x, void* stream);

extern "C" int dvg_conv_wgrad(int mode, const float* x, const float* skip, const float* dout, float* partial, int N,
                              int H, int W, int C1, int C2, int Cout, int upsample_x, void* stream) {
    return dvg_conv_wgrad_multi(mode, 1, &x, skip ? &skip : nullptr, &dout, partial, N, H, W, C1, C2, Cout, upsample_x,
                                stream);
}

extern "C" int dvg_conv_wgrad_multi(int mode, int items, const float* const* xs, const float* const* skips,
                                    const float* const* douts, float* partial, int N, int H, int W, int C1, int C2,
                                    int Cout, int upsample_x, void* stream) {
    DVG_REQUIRE(xs && douts && partial, DVG_ERR_NULL, "dvg_conv_wgrad: NULL pointer");
    DVG_REQUIRE(items >= 1 && items <= WGRAD_MAX_ITEMS, DVG_ERR_SHAPE, "dvg_conv_wgrad: 1 <= items <= %d", WGRAD_MAX_ITEMS);
    DVG_REQUIRE((skips != nullptr) == (C2 > 0), DVG_ERR_SHAPE, "dvg_conv_wgrad: skip pointer / C2 mismatch");
    for (int i = 0; i < items; ++i) {
        DVG_REQUIRE(xs[i] && douts[i] && (C2 == 0 || skips[i]), DVG_ERR_NULL, "dvg_conv_wgrad: NULL pointer in item %d", i);
        DVG_REQUIRE(aligned16(xs[i]) && aligned16(douts[i]) && (C2 == 0 || aligned16(skips[i])), DVG_ERR_ALIGN,
                    "dvg_conv_wgrad: alignment of item %d", i);
    }
    const float* x = xs[0];
    const float* skip = C2 ? skips[0] : nullptr;
    const float* dout = douts[0];
    DVG_REQUIRE(mode >= 0 && mode <= 2, DVG_ERR_SHAPE, "dvg_conv_wgrad: bad mode");
    DVG_REQUIRE(N > 0 && H > 0 && W > 0, DVG_ERR_SHAPE, "dvg_conv_wgrad: empty shape");
    DVG_REQUIRE(C1 > 0 && C1 % 64 == 0 && C2 % 64 == 0 && Cout % 64 == 0, DVG_ERR_SHAPE,
                "dvg_conv_wgrad: C1=%d C2=%d Cout=%d must be multiples of 64", C1, C2, Cout);
    DVG_REQUIRE(aligned16(x) && aligned16(skip) && aligned16(dout) && aligned16(partial), DVG_ERR_ALIGN,
                "dvg_conv_wgrad: alignment");
    DVG_REQUIRE(!upsample_x || mode == W_CONV3, DVG_ERR_SHAPE, "dvg_conv_wgrad: upsample only with CONV3");
    int Hg = H, Wg = W;
    if (mode == W_CONV4S2) { Hg = H / 2; Wg = W / 2; }
    int ti, th, tw;
    DVG_REQUIRE(wgrad_tile(mode, Hg, Wg, &ti, &th, &tw) == 0, DVG_ERR_SHAPE, "dvg_conv_wgrad: unsupported map %dx%d",
                H, W);
    WgradParams p{};
    for (int i = 0; i < items; ++i) {
        p.x[i] = xs[i];
        p.skip[i] = C2 ? skips[i] : nullptr;
        p.dout[i] = douts[i];
    }
    p.items = items;
    p.partial = partial;
    p.N = N; p.H = H; p.W = W; p.C1 = C1; p.C2 = C2; p.Cout = Cout; p.upsample = upsample_x ? 1 : 0;
    p.clk = g_wclk; p.clk_cap = g_wclk_cap;
    p.S = dvg_conv_wgrad_splits_multi(mode, N, H, W, C1 + C2, Cout, items);
    DVG_REQUIRE(p.S > 0, DVG_ERR_SHAPE, "dvg_conv_wgrad: bad split");
    p.n_co = Cout / 64;
    p.n_ci = (C1 + C2) / 64;
    if (mode == W_CONV3) {
        W_DISPATCH(W_CONV3, 1, 8, 16)
        W_DISPATCH(W_CONV3, 1, 8, 8)
    } else if (mode == W_CONV4S2) {
        W_DISPATCH(W_CONV4S2, 1, 4, 8)
        W_DISPATCH(W_CONV4S2, 2, 4, 4)
    } else {
        W_DISPATCH(W_CONVT4S2, 1, 8, 16)
        W_DISPATCH(W_CONVT4S2, 1, 8, 8)
        W_DISPATCH(W_CONVT4S2, 4, 4, 4)
    }
    return fail(DVG_ERR_SHAPE, "dvg_conv_wgrad: no kernel for tile (%d,%d,%d)", ti, th, tw);
}

extern "C" void dvg_debug_set_wgrad_clockbuf(void* buf, unsigned records) {
    g_wclk = (unsigned long long*)buf;
    g_wclk_cap = buf ? records : 0;
}
