// Dense ends of the encoder/decoder and the recurrent latent predictor.
//
//  * gemm_nt: out = act((a . w^T) * scale + shift) for small M (= batch).  Used for
//    the encoder head (4x4 valid conv on a 4x4 map = GEMM with K = 16*512,
//    vgg_64.py:44-48), the decoder stem (1x1 -> 4x4 transposed conv = GEMM with
//    N = 16*512, vgg_64.py:65-69) and the nn.Linear layers of lstm.py:50,53-55.
//    HBM/L2-bound on the weight matrix; split-K spreads the K = 8192 head over
//    the chip.
//  * lstm_cell: one nn.LSTMCell step (lstm.py:51,68-70).  A wave owns an
//    8 (batch) x 8 (2 hidden units x 4 gates) block of gate pre-activations with
//    K split across its 64 lanes; a 63-shuffle butterfly transpose-reduce leaves
//    exactly one finished pre-activation per lane, and three more shuffles bring
//    the i,f,g,o of one (batch, unit) together for the state update.
#include "dvg_common.h"

namespace dvg {

// ------------------------------------------------------------------------------------
// gemm_nt
// ------------------------------------------------------------------------------------
struct GemmParams {
    const float* a;
    const float* w;
    const float* scale;
    const float* shift;
    float* out;
    float* ws;
    int M, N, K, lda, ldo, period, splitk, kper, act;
    float slope;
    int vec;    // w rows are 16-B aligned and K % 4 == 0
    int vec_a;  // a rows are 16-B aligned (lda % 4 == 0)
    int accum;  // out += result (the output IS a parameter's .grad buffer: autograd.py accumulates gradients in place)
};

__global__ __launch_bounds__(256) void gemm_nt_kernel(const GemmParams p) {
    // 64x64 output tile, BK = 32, 4x4 micro-tile per thread.  The next K tile is prefetched into registers
    // while the current one is consumed from LDS (one barrier pair per 32-deep step).
    constexpr int BK = 32, LD = 68;
    __shared__ __attribute__((aligned(16))) float As[BK * LD];
    __shared__ __attribute__((aligned(16))) float Ws[BK * LD];
    const int tid = threadIdx.x;
    const int n0 = blockIdx.x * 64, m0 = blockIdx.y * 64, z = blockIdx.z;
    const int kbeg = z * p.kper, kend = min(p.K, kbeg + p.kper);
    const int tx = tid & 15, ty = tid >> 4;
    const int lr = tid >> 2, lq = tid & 3;  // loader: row 0..63, k-quad 0..3 (two quads per thread: lq and lq+4)

    float acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = 0.f;

    auto gload = [&](int k0, f32x4 (&av)[2], f32x4 (&wv)[2]) {
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const int kk = k0 + (lq + 4 * h) * 4;
            av[h] = f32x4{0.f, 0.f, 0.f, 0.f};
            wv[h] = f32x4{0.f, 0.f, 0.f, 0.f};
            if (m0 + lr < p.M && kk < kend) {
                const float* ap = p.a + (size_t)(m0 + lr) * p.lda + kk;
                if (p.vec_a && kk + 3 < kend) av[h] = *reinterpret_cast<const f32x4*>(ap);
                else
#pragma unroll
                    for (int e = 0; e < 4; ++e)
                        if (kk + e < kend) av[h][e] = ap[e];
            }
            if (n0 + lr < p.N && kk < kend) {
                const float* wp = p.w + (size_t)(n0 + lr) * p.K + kk;
                if (p.vec && kk + 3 < kend) wv[h] = *reinterpret_cast<const f32x4*>(wp);
                else
#pragma unroll
                    for (int e = 0; e < 4; ++e)
                        if (kk + e < kend) wv[h][e] = wp[e];
            }
        }
    };
    f32x4 av[2], wv[2];
    if (kbeg < kend) gload(kbeg, av, wv);
    for (int k0 = kbeg; k0 < kend; k0 += BK) {
        __syncthreads();
#pragma unroll
        for (int h = 0; h < 2; ++h)
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                As[((lq + 4 * h) * 4 + e) * LD + lr] = av[h][e];
                Ws[((lq + 4 * h) * 4 + e) * LD + lr] = wv[h][e];
            }
        __syncthreads();
        if (k0 + BK < kend) gload(k0 + BK, av, wv);
#pragma unroll
        for (int k = 0; k < BK; ++k) {
            const f32x4 a4 = *reinterpret_cast<const f32x4*>(&As[k * LD + ty * 4]);
            const f32x4 w4 = *reinterpret_cast<const f32x4*>(&Ws[k * LD + tx * 4]);
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) acc[i][j] = fmaf(a4[i], w4[j], acc[i][j]);
        }
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int m = m0 + ty * 4 + i;
        if (m >= p.M) continue;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int n = n0 + tx * 4 + j;
            if (n >= p.N) continue;
            if (p.splitk > 1) {
                p.ws[((size_t)z * p.M + m) * p.N + n] = acc[i][j];
            } else {
                const int c = n % p.period;
                const float sc = p.scale ? p.scale[c] : 1.f, sf = p.shift ? p.shift[c] : 0.f;
                float* o = p.out + (size_t)m * p.ldo + n;
                const float v = apply_act(acc[i][j] * sc + sf, p.act, p.slope);
                *o = p.accum ? *o + v : v;
            }
        }
    }
}

__global__ void gemm_splitk_reduce_kernel(const GemmParams p) {
    const long total = (long)p.M * p.N;
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int m = i / p.N, n = i % p.N;
        // eight independent loads per trip (one per split slab), then a fixed-order sum: a `s += ws[z]` loop is one
        // dependent memory round trip per split (9.8 us for the 16-way, 5760-element sum of the encoder head)
        float s = 0.f;
        int z = 0;
        for (; z + 8 <= p.splitk; z += 8) {
            float v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) v[u] = p.ws[(size_t)(z + u) * total + i];
#pragma unroll
            for (int u = 0; u < 8; ++u) s += v[u];
        }
        for (; z < p.splitk; ++z) s += p.ws[(size_t)z * total + i];
        const int c = n % p.period;
        const float sc = p.scale ? p.scale[c] : 1.f, sf = p.shift ? p.shift[c] : 0.f;
        float* o = p.out + (size_t)m * p.ldo + n;
        const float v = apply_act(s * sc + sf, p.act, p.slope);
        *o = p.accum ? *o + v : v;
    }
}

// ------------------------------------------------------------------------------------
// lstm_cell
// ------------------------------------------------------------------------------------
template <int NN>
__device__ __forceinline__ void butterfly_step(float (&v)[64], int lane) {
    // lanes with bit NN set keep the upper NN values, the others the lower NN; the
    // half that is not kept goes to the partner lane (lane ^ NN).
    const bool up = (lane & NN) != 0;
#pragma unroll
    for (int i = 0; i < NN; ++i) {
        const float keep = up ? v[i + NN] : v[i];
        const float send = up ? v[i] : v[i + NN];
        v[i] = keep + __shfl_xor(send, NN);
    }
}

__device__ __forceinline__ float sigmoidf_(float x) { return 1.f / (1.f + expf(-x)); }

// PRE (r04, teacher-forced training, train.py:213-222): the input half of the gates - W_ih x + b_ih + b_hh - has been computed
// for ALL time steps by one GEMM (the inputs of a teacher-forced sequence do not depend on the recurrence); `x` is then that
// pre-activation [B][4H] and only the recurrent half W_hh h runs per step (w_ih, b_ih, b_hh unused).
template <bool PRE>
__global__ __launch_bounds__(256) void lstm_cell_kernel(const float* __restrict__ x, const float* __restrict__ h,
                                                        const float* __restrict__ c,
                                                        const float* __restrict__ w_ih,
                                                        const float* __restrict__ w_hh,
                                                        const float* __restrict__ b_ih,
                                                        const float* __restrict__ b_hh, float* __restrict__ h_out,
                                                        float* __restrict__ c_out, float* __restrict__ gates_out,
                                                        int B, int H) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int j0 = blockIdx.x * 2;
    const int b0 = blockIdx.y * 32 + wave * 8;
    if (b0 >= B) return;  // whole wave out of range (wave-uniform; no barriers below)

    float v[64];  // v[b*8 + jj*4 + g]
#pragma unroll
    for (int i = 0; i < 64; ++i) v[i] = 0.f;

    // epilogue operands of this lane's (batch, unit, gate), fetched with the first K slice instead of after the
    // butterfly (three more dependent round trips otherwise)
    const int eb = min(b0 + (lane >> 3), B - 1), ej = j0 + ((lane >> 2) & 1), eg = lane & 3;
    const float e_bias = PRE ? x[(size_t)eb * 4 * H + eg * H + ej] : b_ih[eg * H + ej] + b_hh[eg * H + ej];
    const float e_c = c[(size_t)eb * H + ej];

    // All 32 float4 loads of a 256-deep K slice are issued before the first FMA and none sits behind a branch
    // (rows past B are clamped to B-1 and their results dropped at the end): with `if (b < B)` guards hipcc put
    // each x/h pair in its own block behind a full vmcnt(0) - eight serialized memory round trips per slice.
    for (int k0 = lane * 4; k0 < H; k0 += 256) {
        f32x4 xv[PRE ? 1 : 8], hv[8], wi[PRE ? 1 : 8], wh[8];
#pragma unroll
        for (int b = 0; b < 8; ++b) {
            const int bb = min(b0 + b, B - 1);
            if constexpr (!PRE) xv[b] = *reinterpret_cast<const f32x4*>(x + (size_t)bb * H + k0);
            hv[b] = *reinterpret_cast<const f32x4*>(h + (size_t)bb * H + k0);
        }
#pragma unroll
        for (int r = 0; r < 8; ++r) {
            const int row = (r & 3) * H + j0 + (r >> 2);
            if constexpr (!PRE) wi[r] = *reinterpret_cast<const f32x4*>(w_ih + (size_t)row * H + k0);
            wh[r] = *reinterpret_cast<const f32x4*>(w_hh + (size_t)row * H + k0);
        }
        __builtin_amdgcn_sched_barrier(0);  // hipcc otherwise re-interleaves load / vmcnt(0) / FMA one load at a time
#pragma unroll
        for (int r = 0; r < 8; ++r) {
#pragma unroll
            for (int b = 0; b < 8; ++b) {
                float s = v[b * 8 + r];
                if constexpr (!PRE) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) s = fmaf(xv[b][e], wi[r][e], s);
                }
#pragma unroll
                for (int e = 0; e < 4; ++e) s = fmaf(hv[b][e], wh[r][e], s);
                v[b * 8 + r] = s;
            }
        }
    }
    // 64 partial sums per lane -> one finished sum per lane (value index == lane)
    butterfly_step<32>(v, lane);
    butterfly_step<16>(v, lane);
    butterfly_step<8>(v, lane);
    butterfly_step<4>(v, lane);
    butterfly_step<2>(v, lane);
    butterfly_step<1>(v, lane);

    const int b = lane >> 3, jj = (lane >> 2) & 1, g = lane & 3;
    const int j = j0 + jj;
    const float pre = v[0] + e_bias;
    const float a = (g == 2) ? tanhf(pre) : sigmoidf_(pre);
    const int q = lane & ~3;
    const float gi = __shfl(a, q), gf = __shfl(a, q + 1), gg = __shfl(a, q + 2), go = __shfl(a, q + 3);
    if (b0 + b < B) {
        if (gates_out) gates_out[(size_t)(b0 + b) * 4 * H + g * H + j] = a;
        if (g == 0) {
            const float cn = gf * e_c + gi * gg;
            c_out[(size_t)(b0 + b) * H + j] = cn;
            h_out[(size_t)(b0 + b) * H + j] = go * tanhf(cn);
        }
    }
}


// ------------------------------------------------------------------------------------
// lstm_cell_x: the FIRST cell of lstm.py:65-70 with the embedding folded in.  `embed` is a plain nn.Linear (lstm.py:50,
// no activation), so  W_ih (W_e x + b_e) + b_ih + W_hh h + b_hh  =  (W_ih W_e) x + W_hh h + (W_ih b_e + b_ih + b_hh):
// the host folds W_x = W_ih W_e [4H][Kxp] (rows padded to Kxp floats, 16-B aligned) and the bias once per weight
// version, and a time step loses the embed launch and 3/4 of this cell's x-side FLOPs (Kx = 90 instead of 256).
// Same wave-level scheme as lstm_cell_kernel; the x part is one float2 per lane (x rows are Kx = 90 floats apart:
// 8-byte aligned only), lanes >= Kx/2 contribute zeros.
// ------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void lstm_cell_x_kernel(const float* __restrict__ x, int ldx, int Kx,
                                                          const float* __restrict__ h, const float* __restrict__ c,
                                                          const float* __restrict__ w_x, int Kxp,
                                                          const float* __restrict__ w_hh,
                                                          const float* __restrict__ bias, float* __restrict__ h_out,
                                                          float* __restrict__ c_out, int B, int H) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int j0 = blockIdx.x * 2;
    const int b0 = blockIdx.y * 32 + wave * 8;
    if (b0 >= B) return;  // wave-uniform; no barriers below

    float v[64];  // v[b*8 + jj*4 + g]
#pragma unroll
    for (int i = 0; i < 64; ++i) v[i] = 0.f;
    const int eb = min(b0 + (lane >> 3), B - 1), ej = j0 + ((lane >> 2) & 1), eg = lane & 3;
    const float e_bias = bias[eg * H + ej];
    const float e_c = c[(size_t)eb * H + ej];

    typedef float f32x2 __attribute__((ext_vector_type(2)));
    {   // x part: Kx <= 128, one float2 per lane
        const bool valid = 2 * lane < Kx;
        const int k0 = valid ? 2 * lane : 0;
        f32x2 xv[8], wx[8];
#pragma unroll
        for (int b = 0; b < 8; ++b) xv[b] = *reinterpret_cast<const f32x2*>(x + (size_t)min(b0 + b, B - 1) * ldx + k0);
#pragma unroll
        for (int r = 0; r < 8; ++r)
            wx[r] = *reinterpret_cast<const f32x2*>(w_x + (size_t)((r & 3) * H + j0 + (r >> 2)) * Kxp + k0);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int r = 0; r < 8; ++r) {
            const f32x2 wz = valid ? wx[r] : f32x2{0.f, 0.f};
#pragma unroll
            for (int b = 0; b < 8; ++b) v[b * 8 + r] = fmaf(xv[b][1], wz[1], fmaf(xv[b][0], wz[0], v[b * 8 + r]));
        }
    }
    for (int k0 = lane * 4; k0 < H; k0 += 256) {
        f32x4 hv[8], wh[8];
#pragma unroll
        for (int b = 0; b < 8; ++b) hv[b] = *reinterpret_cast<const f32x4*>(h + (size_t)min(b0 + b, B - 1) * H + k0);
#pragma unroll
        for (int r = 0; r < 8; ++r)
            wh[r] = *reinterpret_cast<const f32x4*>(w_hh + (size_t)((r & 3) * H + j0 + (r >> 2)) * H + k0);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int r = 0; r < 8; ++r)
#pragma unroll
            for (int b = 0; b < 8; ++b) {
                float s = v[b * 8 + r];
#pragma unroll
                for (int e = 0; e < 4; ++e) s = fmaf(hv[b][e], wh[r][e], s);
                v[b * 8 + r] = s;
            }
    }
    butterfly_step<32>(v, lane);
    butterfly_step<16>(v, lane);
    butterfly_step<8>(v, lane);
    butterfly_step<4>(v, lane);
    butterfly_step<2>(v, lane);
    butterfly_step<1>(v, lane);
    const int b = lane >> 3, jj = (lane >> 2) & 1, g = lane & 3;
    const int j = j0 + jj;
    const float pre = v[0] + e_bias;
    const float a = (g == 2) ? tanhf(pre) : sigmoidf_(pre);
    const int q = lane & ~3;
    const float gi = __shfl(a, q), gf = __shfl(a, q + 1), gg = __shfl(a, q + 2), go = __shfl(a, q + 3);
    if (b0 + b < B && g == 0) {
        const float cn = gf * e_c + gi * gg;
        c_out[(size_t)(b0 + b) * H + j] = cn;
        h_out[(size_t)(b0 + b) * H + j] = go * tanhf(cn);
    }
}

// ------------------------------------------------------------------------------------
// stem: the decoder's ConvTranspose2d(dim,512,4,1,0)+BN+LReLU on a 1x1 map (vgg_64.py:65-69, dcgan_64.py:62-67) for
// eval-mode rollouts: out[b][n] = act((sum_k vec[b][k] wt[k][n]) * scale[n % period] + shift[n % period]), K = dim = 90,
// N = 16*512.  wt is the weight TRANSPOSED to [K][N] (cached by the caller), so a thread that owns output column n loads
// its K weights with fully coalesced 4-byte loads, all issued up front; the latent vectors sit in LDS and are read as
// wave-wide broadcasts.  Workgroup = 32 columns x 8 batch groups; 256 workgroups at N = 8192.
// ------------------------------------------------------------------------------------
template <int KP>
__global__ __launch_bounds__(256) void stem_kernel(const float* __restrict__ vec, int ldv, const float* __restrict__ wt,
                                                   const float* __restrict__ scale, const float* __restrict__ shift,
                                                   float* __restrict__ out, int ldo, int M, int N, int K, int period,
                                                   int act, float slope) {
    __shared__ float vs[64 * KP];   // [mrows][KP], zero padded: no k predicate in the FMA loop
    const int tid = threadIdx.x, nl = tid & 31, bg = tid >> 5;
    const int n = blockIdx.x * 32 + nl;   // N % 32 == 0 (host check)
    const int m0 = blockIdx.y * 64;
    const int mrows = min(64, M - m0);
    for (int i = tid; i < mrows * KP; i += 256) {
        const int r = i / KP, k = i % KP;
        vs[i] = k < K ? vec[(size_t)(m0 + r) * ldv + k] : 0.f;
    }
    float w[KP];   // wt is [KP][N], rows K..KP-1 zero (padded by the caller): unconditional, fully coalesced loads
#pragma unroll
    for (int k = 0; k < KP; ++k) w[k] = wt[(size_t)k * N + n];
    __syncthreads();
    const int c = n % period;
    const float sc = scale ? scale[c] : 1.f, sf = shift ? shift[c] : 0.f;
    for (int bb = bg; bb < mrows; bb += 8) {
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
        out[(size_t)(m0 + bb) * ldo + n] = apply_act((s0 + s1) * sc + sf, act, slope);
    }
}

// ------------------------------------------------------------------------------------
// gemm_dot: the same product for SMALL M (the B = 64 latent path: lstm.py:50,55 embed / output, the encoder head
// and decoder stem of vgg_64.py:44-48,65-69 / dcgan_64.py:41-45,62-66).  A 64x64-tile GEMM leaves these shapes
// with a handful of workgroups crawling through K in serial, latency-bound steps (10-30 us for < 3 MB of
// operands).  Here a WAVE owns an 8 x 8 output block with K spread over its 64 lanes (the lstm_cell scheme):
// every operand load of a K slice is issued before the first FMA, the 64 partial sums per lane are combined by
// the 63-shuffle butterfly, and (M/8)(N/8)splitk waves cover the chip.  Rows / columns past M / N are clamped
// and dropped.
// ------------------------------------------------------------------------------------
template <bool VEC>
__global__ __launch_bounds__(256) void gemm_dot_kernel(const GemmParams p) {
    const int lane = threadIdx.x & 63;
    const long wid = ((long)blockIdx.x * 256 + threadIdx.x) >> 6;
    const int nblk_n = (p.N + 7) >> 3, nblk_m = (p.M + 7) >> 3;
    const long per_split = (long)nblk_n * nblk_m;
    if (wid >= per_split * p.splitk) return;  // wave-uniform
    const int z = (int)(wid / per_split);
    const int rem = (int)(wid % per_split);
    const int m0 = (rem / nblk_n) * 8, n0 = (rem % nblk_n) * 8;
    const int kbeg = z * p.kper, kend = min(p.K, kbeg + p.kper);

    float v[64];  // v[b * 8 + r]
#pragma unroll
    for (int i = 0; i < 64; ++i) v[i] = 0.f;
    const float* arow[8];
    const float* wrow[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        arow[i] = p.a + (size_t)min(m0 + i, p.M - 1) * p.lda;
        wrow[i] = p.w + (size_t)min(n0 + i, p.N - 1) * p.K;
    }
    if (VEC) {
        for (int kk = kbeg; kk < kend; kk += 256) {
            const int k0 = kk + lane * 4;
            const bool valid = k0 < kend;              // K % 4 == 0 and kper % 4 == 0: whole float4s
            const int kc = valid ? k0 : kbeg;
            f32x4 av[8], wv[8];
#pragma unroll
            for (int i = 0; i < 8; ++i) av[i] = *reinterpret_cast<const f32x4*>(arow[i] + kc);
#pragma unroll
            for (int i = 0; i < 8; ++i) wv[i] = *reinterpret_cast<const f32x4*>(wrow[i] + kc);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int r = 0; r < 8; ++r) {
                const f32x4 wz = valid ? wv[r] : f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int b = 0; b < 8; ++b) {
                    float s = v[b * 8 + r];
#pragma unroll
                    for (int e = 0; e < 4; ++e) s = fmaf(av[b][e], wz[e], s);
                    v[b * 8 + r] = s;
                }
            }
        }
    } else {
        for (int kk = kbeg; kk < kend; kk += 128) {
            float av[2][8], wv[2][8];
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const int k0 = kk + h * 64 + lane;
                const bool valid = k0 < kend;
                const int kc = valid ? k0 : kbeg;
#pragma unroll
                for (int i = 0; i < 8; ++i) av[h][i] = arow[i][kc];
#pragma unroll
                for (int i = 0; i < 8; ++i) wv[h][i] = valid ? wrow[i][kc] : 0.f;
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int h = 0; h < 2; ++h)
#pragma unroll
                for (int r = 0; r < 8; ++r)
#pragma unroll
                    for (int b = 0; b < 8; ++b) v[b * 8 + r] = fmaf(av[h][b], wv[h][r], v[b * 8 + r]);
        }
    }
    butterfly_step<32>(v, lane);
    butterfly_step<16>(v, lane);
    butterfly_step<8>(v, lane);
    butterfly_step<4>(v, lane);
    butterfly_step<2>(v, lane);
    butterfly_step<1>(v, lane);
    const int m = m0 + (lane >> 3), n = n0 + (lane & 7);
    if (m < p.M && n < p.N) {
        if (p.splitk > 1) {
            p.ws[((size_t)z * p.M + m) * p.N + n] = v[0];
        } else {
            const int c = n % p.period;
            const float sc = p.scale ? p.scale[c] : 1.f, sf = p.shift ? p.shift[c] : 0.f;
            float* o = p.out + (size_t)m * p.ldo + n;
            const float r = apply_act(v[0] * sc + sf, p.act, p.slope);
            *o = p.accum ? *o + r : r;
        }
    }
}

// ------------------------------------------------------------------------------------
// lstm_cell_bwd (r04): one BPTT step of one nn.LSTMCell in ONE launch - the gate pre-activation gradients dG_t [B][4H] and
// dc_{t-1} (what dvg_lstm_gates_bwd computes) AND the recurrent hand-over dh_{t-1} = dG_t W_hh, which used to be a GEMM launch
// of its own per step and layer.  A wave owns 8 batch rows x 2 columns of dh_{t-1}; K = 4H is spread over its lanes
// with H = 256, so that lane l holds units 4l .. 4l+3 and the four 256-wide K slices ARE the four gates: the lane
// derives the 16 gate gradients of each of its rows from the saved activations (recomputed by every wave of a row block:
// 8 x 1024 elementwise values, nothing against a launch) and multiplies them with its 2 x 4 float4 of W_hh^T rows; the
// forward cell's shuffle butterfly sums over the lanes.  The waves of column block 0 also write dG_t and dc_{t-1}.
// dh = dh_a + dh_b (from the layer above / the output head, and from step t+1; either may be NULL), w_hh_t = W_hh^T [H][4H].
// ------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void lstm_cell_bwd_kernel(const float* __restrict__ dh_a, const float* __restrict__ dh_b,
                                                            const float* __restrict__ dc, const float* __restrict__ gates,
                                                            const float* __restrict__ c_prev, const float* __restrict__ c_new,
                                                            const float* __restrict__ w_hh_t, float* __restrict__ dG,
                                                            float* __restrict__ dc_prev, float* __restrict__ dh_prev,
                                                            int B, int H, int nblk) {
    constexpr int NC = 2;         // columns of dh_prev per wave (x 8 batch rows)
    const int lane = threadIdx.x & 63;
    const long wid = ((long)blockIdx.x * 256 + threadIdx.x) >> 6;
    const long nwaves = (long)((B + 7) / 8) * nblk;
    if (wid >= nwaves) return;   // wave-uniform; no barriers below
    const int b0 = (int)(wid / nblk) * 8, n0 = (int)(wid % nblk) * NC;
    const int j = lane * 4;       // this lane's four hidden units
    // W_hh^T rows n0 .. n0 + NC - 1, the lane's four k-values of each gate slice
    f32x4 w[NC][4];
    if (dh_prev) {
#pragma unroll
        for (int n = 0; n < NC; ++n)
#pragma unroll
            for (int g = 0; g < 4; ++g) w[n][g] = *reinterpret_cast<const f32x4*>(w_hh_t + (size_t)(n0 + n) * 4 * H + g * H + j);
    }
    float v[8 * NC];   // v[b * NC + n]
#pragma unroll
    for (int i = 0; i < 8 * NC; ++i) v[i] = 0.f;
    const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};
    // rows in two batches of four: ALL loads of a batch are issued before its first FMA (left to hipcc, every row's loads sit
    // behind their own vmcnt(0): eight serialized round trips, the lstm_cell_kernel lesson); rows past B are clamped and
    // dropped at the stores
#pragma unroll
    for (int half = 0; half < 2; ++half) {
        f32x4 gi[4], gf[4], gg[4], go[4], cn[4], cp[4], dhv[4], dhw[4], dcin[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int bb = min(b0 + half * 4 + r, B - 1);
            const size_t o = (size_t)bb * H + j;
            const float* gp = gates + (size_t)bb * 4 * H + j;
            gi[r] = *reinterpret_cast<const f32x4*>(gp);
            gf[r] = *reinterpret_cast<const f32x4*>(gp + H);
            gg[r] = *reinterpret_cast<const f32x4*>(gp + 2 * H);
            go[r] = *reinterpret_cast<const f32x4*>(gp + 3 * H);
            cn[r] = *reinterpret_cast<const f32x4*>(c_new + o);
            cp[r] = c_prev ? *reinterpret_cast<const f32x4*>(c_prev + o) : zero4;      // NULL: the zero initial state
            dhv[r] = dh_a ? *reinterpret_cast<const f32x4*>(dh_a + o) : zero4;
            dhw[r] = dh_b ? *reinterpret_cast<const f32x4*>(dh_b + o) : zero4;
            dcin[r] = dc ? *reinterpret_cast<const f32x4*>(dc + o) : zero4;
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int b = half * 4 + r;
            const f32x4 dht = dhv[r] + dhw[r];
            f32x4 d[4], dcp;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const float tc = tanhf(cn[r][e]);
                const float dcv = dcin[r][e] + dht[e] * go[r][e] * (1.f - tc * tc);
                d[0][e] = dcv * gg[r][e] * gi[r][e] * (1.f - gi[r][e]);
                d[1][e] = dcv * cp[r][e] * gf[r][e] * (1.f - gf[r][e]);
                d[2][e] = dcv * gi[r][e] * (1.f - gg[r][e] * gg[r][e]);
                d[3][e] = dht[e] * tc * go[r][e] * (1.f - go[r][e]);
                dcp[e] = dcv * gf[r][e];
            }
            if (n0 == 0 && b0 + b < B) {
                const size_t o = (size_t)(b0 + b) * H + j;
                float* og = dG + (size_t)(b0 + b) * 4 * H + j;
#pragma unroll
                for (int g = 0; g < 4; ++g) *reinterpret_cast<f32x4*>(og + g * H) = d[g];
                *reinterpret_cast<f32x4*>(dc_prev + o) = dcp;
            }
            if (dh_prev) {
#pragma unroll
                for (int n = 0; n < NC; ++n) {
                    float s = 0.f;
#pragma unroll
                    for (int g = 0; g < 4; ++g)
#pragma unroll
                        for (int e = 0; e < 4; ++e) s = fmaf(d[g][e], w[n][g][e], s);
                    v[b * NC + n] = s;
                }
            }
        }
    }
    if (!dh_prev) return;
    // 16 partial sums per lane -> lanes hold one finished sum each (value index == lane & 15): two plain exchanges, then the
    // forward cell's halving butterfly on 16 values
#pragma unroll
    for (int i = 0; i < 16; ++i) v[i] += __shfl_xor(v[i], 32);
#pragma unroll
    for (int i = 0; i < 16; ++i) v[i] += __shfl_xor(v[i], 16);
    {
        const bool up8 = (lane & 8) != 0;
#pragma unroll
        for (int i = 0; i < 8; ++i) { const float keep = up8 ? v[i + 8] : v[i], send = up8 ? v[i] : v[i + 8]; v[i] = keep + __shfl_xor(send, 8); }
        const bool up4 = (lane & 4) != 0;
#pragma unroll
        for (int i = 0; i < 4; ++i) { const float keep = up4 ? v[i + 4] : v[i], send = up4 ? v[i] : v[i + 4]; v[i] = keep + __shfl_xor(send, 4); }
        const bool up2 = (lane & 2) != 0;
#pragma unroll
        for (int i = 0; i < 2; ++i) { const float keep = up2 ? v[i + 2] : v[i], send = up2 ? v[i] : v[i + 2]; v[i] = keep + __shfl_xor(send, 2); }
        const bool up1 = (lane & 1) != 0;
        { const float keep = up1 ? v[1] : v[0], send = up1 ? v[0] : v[1]; v[0] = keep + __shfl_xor(send, 1); }
    }
    // value index of lane l: bits (8, 4, 2, 1) of l select the upper halves in that order = l & 15 = b * NC + n
    const int idx = lane & 15, b = idx / NC, n = idx % NC;
    if (lane < 16 && b0 + b < B) dh_prev[(size_t)(b0 + b) * H + n0 + n] = v[0];
}


// ------------------------------------------------------------------------------------
// gemm_tn (r05): out[m][n] (+)= sum_r a[r][m] b[r][n] - the weight gradient dW = dY^T X of a dense layer (nn.Linear,
// nn.LSTMCell; lstm.py:50-55) over the R = steps x batch rows of a BPTT pass - together with the bias gradients (column sums
// of a = dY, into up to two sinks: an LSTMCell's b_ih and b_hh receive the same sums).  Until r05 this was two LDS-tiled
// transposes, an NT GEMM and a column-sum launch per weight: 71 of the ~520 launches of a 16-clip training iteration.
// Workgroup = 32 (m) x 64 (n) outputs, a thread owns 2 x 4 of them; the rows are walked in chunks of 32 through LDS with
// the next chunk's loads in flight (registers) during the FMAs; both operands are read along their contiguous dimension.
// fp32 FMAs, rows summed in order: deterministic.
// ------------------------------------------------------------------------------------
struct GemmTnParams {
    const float* a; const float* b; float* out; float* cs0; float* cs1;
    int R, M, N, lda, ldb, ldo, accumulate, cs_accumulate, vec_a, vec_b;
};

__global__ __launch_bounds__(256) void gemm_tn_kernel(const GemmTnParams p) {
    constexpr int TM = 32, TN = 64, KT = 32;
    __shared__ __attribute__((aligned(16))) float As[KT][TM + 4];
    __shared__ __attribute__((aligned(16))) float Bs[KT][TN + 4];
    const int tid = threadIdx.x, tn = tid & 15, tm = tid >> 4;
    const int m0 = blockIdx.y * TM, n0 = blockIdx.x * TN;
    // loaders: a chunk = 32 rows x 32 floats (thread: row tid / 8, floats (tid % 8) * 4 ..), b chunk = 32 rows x 64 floats (two
    // passes of 16 rows: row tid / 16 (+ 16), floats (tid % 16) * 4 ..)
    const int ar = tid >> 3, ac = m0 + (tid & 7) * 4, br = tid >> 4, bc = n0 + (tid & 15) * 4;
    auto ld4 = [](const float* base, int ld, int r, int c, int R, int C, int vec) {
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (r < R) {
            const float* q = base + (size_t)r * ld + c;
            if (vec) {
                if (c < C) v = *reinterpret_cast<const float4*>(q);      // C % 4 == 0: whole float4s
            } else {
                if (c < C) v.x = q[0];
                if (c + 1 < C) v.y = q[1];
                if (c + 2 < C) v.z = q[2];
                if (c + 3 < C) v.w = q[3];
            }
        }
        return v;
    };
    float4 ra, rb0, rb1;
    auto gload = [&](int r0) {
        ra = ld4(p.a, p.lda, r0 + ar, ac, p.R, p.M, p.vec_a);
        rb0 = ld4(p.b, p.ldb, r0 + br, bc, p.R, p.N, p.vec_b);
        rb1 = ld4(p.b, p.ldb, r0 + br + 16, bc, p.R, p.N, p.vec_b);
    };
    auto lstore = [&]() {
        *reinterpret_cast<float4*>(&As[ar][(tid & 7) * 4]) = ra;
        *reinterpret_cast<float4*>(&Bs[br][(tid & 15) * 4]) = rb0;
        *reinterpret_cast<float4*>(&Bs[br + 16][(tid & 15) * 4]) = rb1;
    };
    float acc[2][4] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
    float cs[2] = {0.f, 0.f};
    const bool do_cs = p.cs0 != nullptr && blockIdx.x == 0 && tn == 0;      // the bias gradient of this workgroup's 32 m
    gload(0);
    for (int r0 = 0; r0 < p.R; r0 += KT) {
        __syncthreads();            // the previous chunk has been read
        lstore();
        __syncthreads();
        if (r0 + KT < p.R) gload(r0 + KT);
#pragma unroll 8
        for (int k = 0; k < KT; ++k) {
            const float a0 = As[k][tm * 2], a1 = As[k][tm * 2 + 1];
            const float4 b4 = *reinterpret_cast<const float4*>(&Bs[k][tn * 4]);
            acc[0][0] = fmaf(a0, b4.x, acc[0][0]); acc[0][1] = fmaf(a0, b4.y, acc[0][1]);
            acc[0][2] = fmaf(a0, b4.z, acc[0][2]); acc[0][3] = fmaf(a0, b4.w, acc[0][3]);
            acc[1][0] = fmaf(a1, b4.x, acc[1][0]); acc[1][1] = fmaf(a1, b4.y, acc[1][1]);
            acc[1][2] = fmaf(a1, b4.z, acc[1][2]); acc[1][3] = fmaf(a1, b4.w, acc[1][3]);
            if (do_cs) { cs[0] += a0; cs[1] += a1; }
        }
    }
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int m = m0 + tm * 2 + i;
        if (m >= p.M) continue;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int n = n0 + tn * 4 + j;
            if (n < p.N) {
                float* o = p.out + (size_t)m * p.ldo + n;
                *o = p.accumulate ? *o + acc[i][j] : acc[i][j];
            }
        }
        if (do_cs) {
            p.cs0[m] = p.cs_accumulate ? p.cs0[m] + cs[i] : cs[i];
            if (p.cs1) p.cs1[m] = p.cs_accumulate ? p.cs1[m] + cs[i] : cs[i];
        }
    }
}
}  // namespace dvg

using namespace dvg;

extern "C" int dvg_gemm_nt_bias_act(const float* a, const float* w, const float* scale, const float* shift, float* out,
                                    float* workspace, int M, int N, int K, int lda, int ldo, int period, int splitk,
                                    int act, float slope, int accumulate, void* stream) {
    DVG_REQUIRE(a && w && out, DVG_ERR_NULL, "dvg_gemm_nt_bias_act: NULL pointer");
    DVG_REQUIRE(M > 0 && N > 0 && K > 0 && lda >= K && ldo >= N, DVG_ERR_SHAPE, "dvg_gemm_nt_bias_act: bad shape");
    DVG_REQUIRE(period > 0 && period <= N && N % period == 0, DVG_ERR_SHAPE, "dvg_gemm_nt_bias_act: bad period");
    DVG_REQUIRE(splitk >= 1 && splitk <= 256, DVG_ERR_SHAPE, "dvg_gemm_nt_bias_act: bad splitk");
    DVG_REQUIRE(splitk == 1 || workspace != nullptr, DVG_ERR_NULL, "dvg_gemm_nt_bias_act: workspace needed");
    DVG_REQUIRE(act >= 0 && act <= 3, DVG_ERR_SHAPE, "dvg_gemm_nt_bias_act: bad act");
    GemmParams p{a, w, scale, shift, out, workspace, M, N, K, lda, ldo, period, splitk, 0, act, slope, 0, 0,
                 accumulate ? 1 : 0};
    int kper = (K + splitk - 1) / splitk;
    kper = ((kper + 31) / 32) * 32;
    p.kper = kper;
    p.splitk = (K + kper - 1) / kper;  // drop empty splits
    p.vec = (K % 4 == 0 && aligned16(w)) ? 1 : 0;
    p.vec_a = (K % 4 == 0 && lda % 4 == 0 && aligned16(a)) ? 1 : 0;
    if (M <= 1024 && N <= 2048) {
        // small-M (latent path) shapes: wave-level dot products instead of the 64x64-tile kernel (the N = 8192
        // decoder stem re-reads each weight row from 8 waves here and stays on the tile kernel: 10.1 vs 11.3 us)
        if (p.vec && p.vec_a) {
            kper = (((K + splitk - 1) / splitk + 255) / 256) * 256;   // whole 256-wide lane slices per split
            p.kper = kper;
            p.splitk = (K + kper - 1) / kper;
        }
        const long waves = (long)((M + 7) / 8) * ((N + 7) / 8) * p.splitk;
        const dim3 grid((unsigned)((waves + 3) / 4));
        if (p.vec && p.vec_a) hipLaunchKernelGGL(gemm_dot_kernel<true>, grid, dim3(256), 0, (hipStream_t)stream, p);
        else hipLaunchKernelGGL(gemm_dot_kernel<false>, grid, dim3(256), 0, (hipStream_t)stream, p);
    } else {
        dim3 grid((N + 63) / 64, (M + 63) / 64, p.splitk);
        hipLaunchKernelGGL(gemm_nt_kernel, grid, dim3(256), 0, (hipStream_t)stream, p);
    }
    if (int e = check_launch("dvg_gemm_nt_bias_act")) return e;
    if (p.splitk > 1) {
        const long total = (long)M * N;
        const unsigned g = (unsigned)((total + 63) / 64 > 4096 ? 4096 : (total + 63) / 64);
        hipLaunchKernelGGL(gemm_splitk_reduce_kernel, dim3(g), dim3(64), 0, (hipStream_t)stream, p);
        return check_launch("dvg_gemm_nt_bias_act(reduce)");
    }
    return DVG_OK;
}

extern "C" int dvg_gemm_tn(const float* a, const float* b, float* out, float* colsum0, float* colsum1, int R, int M, int N,
                           int lda, int ldb, int ldo, int accumulate, int colsum_accumulate, void* stream) {
    DVG_REQUIRE(a && b && out, DVG_ERR_NULL, "dvg_gemm_tn: NULL pointer");
    DVG_REQUIRE(R > 0 && M > 0 && N > 0 && lda >= M && ldb >= N && ldo >= N, DVG_ERR_SHAPE, "dvg_gemm_tn: bad shape");
    DVG_REQUIRE(colsum0 != nullptr || colsum1 == nullptr, DVG_ERR_NULL, "dvg_gemm_tn: colsum1 without colsum0");
    GemmTnParams p{a, b, out, colsum0, colsum1, R, M, N, lda, ldb, ldo, accumulate ? 1 : 0, colsum_accumulate ? 1 : 0,
                   (M % 4 == 0 && lda % 4 == 0 && aligned16(a)) ? 1 : 0, (N % 4 == 0 && ldb % 4 == 0 && aligned16(b)) ? 1 : 0};
    hipLaunchKernelGGL(gemm_tn_kernel, dim3((N + 63) / 64, (M + 31) / 32), dim3(256), 0, (hipStream_t)stream, p);
    return check_launch("dvg_gemm_tn");
}

extern "C" int dvg_lstm_cell(const float* x, const float* h, const float* c, const float* w_ih, const float* w_hh,
                             const float* b_ih, const float* b_hh, float* h_out, float* c_out, float* gates_out,
                             int B, int H, void* stream) {
    DVG_REQUIRE(x && h && c && w_ih && w_hh && b_ih && b_hh && h_out && c_out, DVG_ERR_NULL,
                "dvg_lstm_cell: NULL pointer");
    DVG_REQUIRE(B > 0 && H > 0 && H % 64 == 0, DVG_ERR_SHAPE, "dvg_lstm_cell: H=%d must be a multiple of 64", H);
    DVG_REQUIRE(h_out != h && c_out != c && h_out != x, DVG_ERR_SHAPE, "dvg_lstm_cell: in-place state update");
    DVG_REQUIRE(aligned16(x) && aligned16(h) && aligned16(w_ih) && aligned16(w_hh), DVG_ERR_ALIGN,
                "dvg_lstm_cell: alignment");
    dim3 grid(H / 2, (B + 31) / 32);
    hipLaunchKernelGGL(lstm_cell_kernel<false>, grid, dim3(256), 0, (hipStream_t)stream, x, h, c, w_ih, w_hh, b_ih, b_hh,
                       h_out, c_out, gates_out, B, H);
    return check_launch("dvg_lstm_cell");
}

extern "C" int dvg_lstm_cell_pre(const float* pre, const float* h, const float* c, const float* w_hh, float* h_out,
                                 float* c_out, float* gates_out, int B, int H, void* stream) {
    DVG_REQUIRE(pre && h && c && w_hh && h_out && c_out, DVG_ERR_NULL, "dvg_lstm_cell_pre: NULL pointer");
    DVG_REQUIRE(B > 0 && H > 0 && H % 64 == 0, DVG_ERR_SHAPE, "dvg_lstm_cell_pre: H=%d must be a multiple of 64", H);
    DVG_REQUIRE(h_out != h && c_out != c, DVG_ERR_SHAPE, "dvg_lstm_cell_pre: in-place state update");
    DVG_REQUIRE(aligned16(h) && aligned16(w_hh), DVG_ERR_ALIGN, "dvg_lstm_cell_pre: alignment");
    dim3 grid(H / 2, (B + 31) / 32);
    hipLaunchKernelGGL(lstm_cell_kernel<true>, grid, dim3(256), 0, (hipStream_t)stream, pre, h, c, nullptr, w_hh, nullptr,
                       nullptr, h_out, c_out, gates_out, B, H);
    return check_launch("dvg_lstm_cell_pre");
}

extern "C" int dvg_lstm_cell_bwd(const float* dh_a, const float* dh_b, const float* dc, const float* gates,
                                 const float* c_prev, const float* c_new, const float* w_hh_t, float* dG, float* dc_prev,
                                 float* dh_prev, int B, int H, void* stream) {
    DVG_REQUIRE(gates && c_new && w_hh_t && dG && dc_prev, DVG_ERR_NULL, "dvg_lstm_cell_bwd: NULL pointer");
    DVG_REQUIRE(dh_a || dh_b || dc, DVG_ERR_NULL, "dvg_lstm_cell_bwd: no incoming gradient");
    DVG_REQUIRE(B > 0 && H == 256, DVG_ERR_SHAPE, "dvg_lstm_cell_bwd: H=%d (this kernel maps one gate to one 256-wide K slice)", H);
    DVG_REQUIRE(aligned16(dh_a) && aligned16(dh_b) && aligned16(dc) && aligned16(gates) && aligned16(c_prev) &&
                aligned16(c_new) && aligned16(w_hh_t) && aligned16(dG) && aligned16(dc_prev) && aligned16(dh_prev),
                DVG_ERR_ALIGN, "dvg_lstm_cell_bwd: alignment");
    // one wave = 8 batch rows x 2 columns of dh_prev; dh_prev == NULL (first time step): only dG and dc_prev
    const int nblk = dh_prev ? H / 2 : 1;
    const long waves = (long)((B + 7) / 8) * nblk;
    hipLaunchKernelGGL(lstm_cell_bwd_kernel, dim3((unsigned)((waves + 3) / 4)), dim3(256), 0, (hipStream_t)stream, dh_a, dh_b,
                       dc, gates, c_prev, c_new, w_hh_t, dG, dc_prev, dh_prev, B, H, nblk);
    return check_launch("dvg_lstm_cell_bwd");
}

extern "C" int dvg_lstm_cell_x(const float* x, int ldx, int Kx, const float* h, const float* c, const float* w_x,
                               int Kxp, const float* w_hh, const float* bias, float* h_out, float* c_out, int B, int H,
                               void* stream) {
    DVG_REQUIRE(x && h && c && w_x && w_hh && bias && h_out && c_out, DVG_ERR_NULL, "dvg_lstm_cell_x: NULL pointer");
    DVG_REQUIRE(B > 0 && H > 0 && H % 64 == 0, DVG_ERR_SHAPE, "dvg_lstm_cell_x: H=%d must be a multiple of 64", H);
    DVG_REQUIRE(Kx > 0 && Kx <= 128 && Kx % 2 == 0 && ldx >= Kx && ldx % 2 == 0 && Kxp >= Kx && Kxp % 4 == 0,
                DVG_ERR_SHAPE, "dvg_lstm_cell_x: Kx=%d (even, <= 128), ldx=%d (even), Kxp=%d (multiple of 4)", Kx, ldx, Kxp);
    DVG_REQUIRE(h_out != h && c_out != c, DVG_ERR_SHAPE, "dvg_lstm_cell_x: in-place state update");
    DVG_REQUIRE(aligned16(h) && aligned16(w_x) && aligned16(w_hh) && (reinterpret_cast<uintptr_t>(x) & 7u) == 0,
                DVG_ERR_ALIGN, "dvg_lstm_cell_x: alignment");
    dim3 grid(H / 2, (B + 31) / 32);
    hipLaunchKernelGGL(lstm_cell_x_kernel, grid, dim3(256), 0, (hipStream_t)stream, x, ldx, Kx, h, c, w_x, Kxp, w_hh, bias,
                       h_out, c_out, B, H);
    return check_launch("dvg_lstm_cell_x");
}

extern "C" int dvg_stem_gemm(const float* vec, int ldv, const float* w_kn, int KP, const float* scale, const float* shift,
                             float* out, int ldo, int M, int N, int K, int period, int act, float slope, void* stream) {
    DVG_REQUIRE(vec && w_kn && out, DVG_ERR_NULL, "dvg_stem_gemm: NULL pointer");
    DVG_REQUIRE(M > 0 && N > 0 && N % 32 == 0 && K > 0 && K <= KP && (KP == 96 || KP == 128) && ldv >= K && ldo >= N,
                DVG_ERR_SHAPE, "dvg_stem_gemm: bad shape M=%d N=%d K=%d KP=%d (N %% 32 == 0, KP 96 or 128)", M, N, K, KP);
    DVG_REQUIRE(period > 0 && period <= N && N % period == 0, DVG_ERR_SHAPE, "dvg_stem_gemm: bad period");
    DVG_REQUIRE(act >= 0 && act <= 3, DVG_ERR_SHAPE, "dvg_stem_gemm: bad act");
    const dim3 grid(N / 32, (M + 63) / 64);
    if (KP == 96)
        hipLaunchKernelGGL(stem_kernel<96>, grid, dim3(256), 0, (hipStream_t)stream, vec, ldv, w_kn, scale, shift, out, ldo,
                           M, N, K, period, act, slope);
    else
        hipLaunchKernelGGL(stem_kernel<128>, grid, dim3(256), 0, (hipStream_t)stream, vec, ldv, w_kn, scale, shift, out, ldo,
                           M, N, K, period, act, slope);
    return check_launch("dvg_stem_gemm");
}
