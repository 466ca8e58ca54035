// Weight gradient of a 3x3 / stride-1 convolution in Winograd F(4x4,3x3) form (training path, loss.backward() of
// train.py:170,194,240 for the vgg_layer convolutions of models/vgg_64.py:5-15 on maps up to 32x32 with >= 128 channels).
//
//   y = A^T [ (G g G^T) . (B^T d B) ] A          (forward, winograd.hip)
//   dg = G^T [ sum over tiles and images of (A dY A^T) . (B^T d B) ] G
//
// i.e. per transform position xi (36 of them) a GEMM over the TILES:  P[xi][co][ci] = sum_t dM[xi][t][co] * V[xi][t][ci]
// with V the forward's input transform of the layer input and dM = A dY A^T the 4x4 -> 6x6 transform of d(out).  2.25 x
// the operand bytes of the direct form, a quarter of its flops (36 products per 4x4 tile instead of 144).
//
// The GEMM is "TN": both operands are K-major ([t][channel], channel contiguous), which is exactly the fp32 MFMA operand
// layout (v_mfma_f32_32x32x2_f32: lane l holds A[row l % 32][k = l / 32]) - no transposition anywhere, LDS only shares the
// operand tiles between the four waves of a workgroup and regroups four consecutive k per column so that ONE ds_read_b128
// feeds four MFMA k-steps.
#include <hip/hip_runtime.h>

#include <cstdlib>
#include <type_traits>

#include "dvg_common.h"

namespace dvg {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int WW_BM = 128;                // Cout rows / Cin columns of a workgroup's output block
// Variants (template parameters; the host picks WW_DEFAULT_*):
//   KS  tiles (K rows) per stage: 32 (64 measured 2-4 % slower in the single-buffer form and is not instantiated)
//   DB  true: two LDS stages, one barrier per stage; false: one LDS stage, the next stage's ds_writes behind a barrier
//       under the last k group's MFMAs (as the forward igemm does)
constexpr bool WW_DEFAULT_DB = false;
constexpr int WW_DEFAULT_KS = 32;
constexpr int ww_opf(int ks) { return (ks / 4) * WW_BM * 4; }   // floats of one operand stage in LDS: [k quad][column][4 k]
constexpr int ww_lds_bytes(bool db, int ks) { return (db ? 2 : 1) * 2 * ww_opf(ks) * 4; }

struct WwParams {
    const float* dm;   // [36][Tp][Cout]
    const float* v[8]; // per use: [36][t_item][Cin] (the forward's saved input transforms), or ONE buffer [36][Tp][Cin]
    int t_item;        // tiles per V buffer (Tp when there is one); a multiple of the stage length
    float* part;       // [S][36][Cout][Cin]
    long Tp;           // tiles (padded to a multiple of 64; padding rows are zero)
    int Cin, Cout, S, nst, nbi, nbj;   // S slabs, nst = Tp / KS stages per output block
    int q, total;      // stages per workgroup, stages in all (= blocks * nst; the host checks that it fits)
};

// Work split ("stream-K"): the launch is ONE round of equally long workgroups.  The blocks x nst stages of all output blocks
// form one sequence, workgroup w takes stages [w q, (w + 1) q): the tail of one block's K range and the head of the next
// (or a slice of one block when q < nst).  A block cut into n segments sends segment s to slab s; the workgroup that writes a
// block's last segment zero-fills the slabs beyond it (S = the most segments any block has), so that the slab sum needs no
// per-block bookkeeping.  With whole K ranges per workgroup (576 blocks for a 512 -> 512 layer on 512 resident slots) the
// second round ran one eighth full: 107 TF; split 2 / 4 ways 108-115 TF plus the slabs' traffic.

// 256 threads = 4 waves as 2 x 2, each a 64 x 64 block of the 128 (Cout) x 128 (Cin) output: 4 accumulators, and per group
// of 8 k: 4 ds_read_b128 (2 dM + 2 V fragments) feed 16 MFMAs.  Global -> LDS: thread (k quad rq, column cg + 32 c) loads
// 4 rows x 4 columns as dwords (128 contiguous bytes per row per half wave) and stores one b128 per column - the 4 k of a
// column are the 4 rows it loaded, so the regrouping is free and the stores are conflict-free (consecutive lanes,
// consecutive 16-byte slots).
template <bool DB, int KS>
__global__ __launch_bounds__(256, 2) void wino_wgrad_gemm_kernel(const WwParams p) {
    constexpr int OPF = ww_opf(KS), NQ = KS / 32, NG = KS / 8;   // NQ: k quads per thread, NG: k groups per stage
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, wr = wave >> 1, wc = wave & 1, l31 = lane & 31, hh = lane >> 5;
    const int w = (int)xcd_remap(blockIdx.x, gridDim.x);
    const int rq = tid >> 5, cg = tid & 31;
    const float* A = nullptr;
    int xi = 0;
    long b_thread = 0;   // this thread's offset inside a V row block: (k quad rows) * Cin + Cin block + column

    float ra[NQ][16], rb[NQ][16];
    auto gload = [&](int st) {
#pragma unroll
        for (int q = 0; q < NQ; ++q) {
            const float* a = A + ((size_t)st * KS + q * 32) * p.Cout;
            const int row = st * KS + q * 32, item = row / p.t_item;     // wave-uniform: which use's V this stage reads
            const float* b = p.v[item] + ((size_t)xi * p.t_item + (row - item * p.t_item)) * p.Cin + b_thread;
#pragma unroll
            for (int c = 0; c < 4; ++c)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    ra[q][c * 4 + r] = a[(size_t)r * p.Cout + 32 * c];
                    rb[q][c * 4 + r] = b[(size_t)r * p.Cin + 32 * c];
                }
        }
    };
    auto lstore = [&](int buf) {
        float* As = smem + buf * 2 * OPF;
        float* Bs = As + OPF;
#pragma unroll
        for (int q = 0; q < NQ; ++q)
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                *reinterpret_cast<f32x4*>(&As[((q * 8 + rq) * WW_BM + cg + 32 * c) * 4]) =
                    f32x4{ra[q][c * 4], ra[q][c * 4 + 1], ra[q][c * 4 + 2], ra[q][c * 4 + 3]};
                *reinterpret_cast<f32x4*>(&Bs[((q * 8 + rq) * WW_BM + cg + 32 * c) * 4]) =
                    f32x4{rb[q][c * 4], rb[q][c * 4 + 1], rb[q][c * 4 + 2], rb[q][c * 4 + 3]};
            }
    };

    // fragment of k group j (8 k: lanes 0-31 take quad 2j, lanes 32-63 quad 2j + 1; component e of both = MFMA k-step e)
    const int fa_off = (hh * WW_BM + wr * 64 + l31) * 4, fb_off = (hh * WW_BM + wc * 64 + l31) * 4;
    const int u_end = min((w + 1) * p.q, p.total);
    for (int u = w * p.q; u < u_end;) {
    // ---- one segment: stages [st0, st1) of output block `blk` (Cin block fastest, then Cout block, then position: the
    // workgroups that stream the same K rows of one position are neighbours in one XCD's L2)
    const int blk = u / p.nst;
    const int st0 = u - blk * p.nst, st1 = min(p.nst, st0 + (u_end - u));
    const int seg = w - blk * p.nst / p.q;
    const int bj = blk % p.nbj, bi = (blk / p.nbj) % p.nbi;
    xi = blk / (p.nbj * p.nbi);
    A = p.dm + ((size_t)xi * p.Tp + rq * 4) * p.Cout + bi * WW_BM + cg;
    b_thread = (long)rq * 4 * p.Cin + bj * WW_BM + cg;
    u += st1 - st0;
    f32x16 acc[2][2];
#pragma unroll
    for (int m = 0; m < 2; ++m)
#pragma unroll
        for (int n = 0; n < 2; ++n)
#pragma unroll
            for (int i = 0; i < 16; ++i) acc[m][n][i] = 0.f;

    gload(st0);
    lstore(0);
    __syncthreads();
    // `more` (another stage follows) is compile-time - the loop is peeled: under a runtime branch the next stage's ds_writes
    // form their own basic block and cannot be interleaved with the last group's MFMAs
    auto stage = [&](const int st, auto more_c) {
        constexpr bool more = decltype(more_c)::value;
        const int buf = DB ? ((st - st0) & 1) : 0;
        if (more) gload(st + 1);
        const float* As = smem + buf * 2 * OPF;
        const float* Bs = As + OPF;
        f32x4 fa[2][2], fb[2][2];
#pragma unroll
        for (int m = 0; m < 2; ++m) {
            fa[0][m] = *reinterpret_cast<const f32x4*>(&As[fa_off + m * 128]);
            fb[0][m] = *reinterpret_cast<const f32x4*>(&Bs[fb_off + m * 128]);
        }
        // the first group's four reads as one scheduling group: without it they took the first slots of the read / MFMA
        // pattern below, every later read slipped to right before its use and hipcc waited lgkmcnt(0) every four MFMAs
        if (more) __builtin_amdgcn_sched_group_barrier(0x020, 32 * NQ, 0);   // the next stage's global loads first
        __builtin_amdgcn_sched_group_barrier(0x100, 4, 0);
#pragma unroll
        for (int j = 0; j < NG; ++j) {
            const int cur = j & 1, nxt = cur ^ 1;
            if (j + 1 < NG) {
#pragma unroll
                for (int m = 0; m < 2; ++m) {
                    fa[nxt][m] = *reinterpret_cast<const f32x4*>(&As[fa_off + (j + 1) * 2 * WW_BM * 4 + m * 128]);
                    fb[nxt][m] = *reinterpret_cast<const f32x4*>(&Bs[fb_off + (j + 1) * 2 * WW_BM * 4 + m * 128]);
                }
            }
            if (j + 1 == NG && more) {
                // the next stage's ds_writes between the last group's MFMAs: into the other buffer (DB; last read in stage
                // st - 1, which every wave left at that stage's barrier), or into the only one once every wave holds its last
                // fragments
                if (!DB) __syncthreads();
                lstore(DB ? buf ^ 1 : 0);
            }
#pragma unroll
            for (int e = 0; e < 4; ++e)
#pragma unroll
                for (int m = 0; m < 2; ++m)
#pragma unroll
                    for (int n = 0; n < 2; ++n)
                        acc[m][n] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[cur][m][e], fb[cur][n][e], acc[m][n], 0, 0, 0);
            // pin the pipeline: the next group's four ds_read_b128 one per four MFMAs, a whole group ahead of their use
            if (j + 1 < NG) {
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
                    __builtin_amdgcn_sched_group_barrier(0x008, 4, 0);
                }
            } else if (more) {
#pragma unroll
                for (int r = 0; r < 8 * NQ; ++r) {   // the ds_write_b128 between the MFMAs
                    if (r < 16) __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                    __builtin_amdgcn_sched_group_barrier(0x200, 1, 0);
                }
                if (8 * NQ < 16) __builtin_amdgcn_sched_group_barrier(0x008, 16 - 8 * NQ, 0);
            } else {
                __builtin_amdgcn_sched_group_barrier(0x008, 16, 0);
            }
        }
        __syncthreads();
    };
    for (int st = st0; st + 1 < st1; ++st) stage(st, std::integral_constant<bool, true>{});
    stage(st1 - 1, std::integral_constant<bool, false>{});

    const size_t slab = (size_t)36 * p.Cout * p.Cin;
    const int nzero = st1 == p.nst ? p.S - 1 - seg : 0;   // the block's last segment: zero the slabs it did not use
    float* out = p.part + (size_t)seg * slab + (((size_t)xi * p.Cout + bi * WW_BM + wr * 64) * p.Cin + bj * WW_BM + wc * 64 + l31);
#pragma unroll
    for (int m = 0; m < 2; ++m)
#pragma unroll
        for (int n = 0; n < 2; ++n)
#pragma unroll
            for (int reg = 0; reg < 16; ++reg) {
                const int row = m * 32 + (reg & 3) + 8 * (reg >> 2) + 4 * hh;
                out[(size_t)row * p.Cin + n * 32] = acc[m][n][reg];
            }
    // (hipcc hoists the 32 row offsets of these stores out of the segment loop - 64 VGPRs live through the kernel, 238 in
    // all; a running pointer with wave-uniform strides brought that to 174 and measured 4-6 % SLOWER at the same two waves
    // per SIMD, so the offsets stay)
    // rolled, cooperative and coalesced: with the accumulator stores' 64 row addresses reused for the zero stores hipcc kept
    // them all live across the kernel (256 VGPRs + spills)
    float* zbase = p.part + (size_t)(seg + 1) * slab + ((size_t)xi * p.Cout + bi * WW_BM) * p.Cin + bj * WW_BM;
#pragma unroll 1
    for (int z = 0; z < nzero; ++z, zbase += slab)
#pragma unroll 1
        for (int i = tid; i < WW_BM * (WW_BM / 4); i += 256)
            *reinterpret_cast<f32x4*>(zbase + (size_t)(i >> 5) * p.Cin + (i & 31) * 4) = f32x4{0.f, 0.f, 0.f, 0.f};
    }   // segments
}

// ---- the same product on the bf16 matrix pipe (DVG_BF16X3, dvg_common.h) -------------------------------------------------
// Operands as exact bf16 triples, six v_mfma_f32_32x32x16_bf16 per 32 x 32 tile and 16 tiles of K.  A thread loads 8
// consecutive K rows of two columns per operand (dwords: 256 contiguous bytes per row per wave), splits them and stores, per
// column and plane, the 8 k-values as ONE 16-byte run: the LDS image [k octet][plane][column][8 bf16] is the MFMA operand
// layout (lanes 0-31 take octet 2 s, lanes 32-63 octet 2 s + 1 of k-step s), a fragment is one conflict-free ds_read_b128.
// Same work split, slabs and epilogue as the f32 kernel above.
constexpr int ww3_opf() { return 4 * 3 * WW_BM * 4; }           // floats of one operand stage: 4 octets x 3 planes x 128 columns x 16 B
constexpr int ww3_lds_bytes() { return 2 * ww3_opf() * 4; }

__global__ __launch_bounds__(256, 2) void wino_wgrad_gemm_x3_kernel(const WwParams p) {
    constexpr int KS = 32, OPF = ww3_opf();
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* const As = smem;
    float* const Bs = smem + OPF;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, wr = wave >> 1, wc = wave & 1, l31 = lane & 31, hh = lane >> 5;
    const int w = (int)xcd_remap(blockIdx.x, gridDim.x);
    const int oct = tid >> 6, cp = tid & 63;     // loader: K rows 8 oct ... + 7 of columns cp and cp + 64
    const float* A = nullptr;
    int xi = 0;
    long b_thread = 0;

    float ra[16], rb[16];
    auto gload = [&](int st) {
        const float* a = A + (size_t)st * KS * p.Cout;
        const int row = st * KS, item = row / p.t_item;     // wave-uniform: which use's V this stage reads
        const float* b = p.v[item] + ((size_t)xi * p.t_item + (row - item * p.t_item)) * p.Cin + b_thread;
#pragma unroll
        for (int c = 0; c < 2; ++c)
#pragma unroll
            for (int r = 0; r < 8; ++r) {
                ra[c * 8 + r] = a[(size_t)r * p.Cout + 64 * c];
                rb[c * 8 + r] = b[(size_t)r * p.Cin + 64 * c];
            }
    };
    auto lstore = [&]() {
        auto put = [&](float* tile, const float (&v)[16], int c) {
            u32x4_t h, m, l;
            unsigned t0, t1, t2;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                bf16x3_split_pair(v[c * 8 + 2 * i], v[c * 8 + 2 * i + 1], t0, t1, t2);
                h[i] = t0; m[i] = t1; l[i] = t2;
            }
            float* d = &tile[((oct * 3) * WW_BM + cp + 64 * c) * 4];
            *reinterpret_cast<u32x4_t*>(d) = h;
            *reinterpret_cast<u32x4_t*>(d + WW_BM * 4) = m;
            *reinterpret_cast<u32x4_t*>(d + 2 * WW_BM * 4) = l;
        };
#pragma unroll
        for (int c = 0; c < 2; ++c) {
            put(As, ra, c);
            put(Bs, rb, c);
        }
    };
    // fragment of k-step s (16 k), plane pl, 32-column tile m: floats ((2 s + hh) * 3 + pl) * 128 * 4 + column * 4
    const int fa_off = (hh * 3 * WW_BM + wr * 64 + l31) * 4, fb_off = (hh * 3 * WW_BM + wc * 64 + l31) * 4;
    const int u_end = min((w + 1) * p.q, p.total);
    for (int u = w * p.q; u < u_end;) {
    const int blk = u / p.nst;
    const int st0 = u - blk * p.nst, st1 = min(p.nst, st0 + (u_end - u));
    const int seg = w - blk * p.nst / p.q;
    const int bj = blk % p.nbj, bi = (blk / p.nbj) % p.nbi;
    xi = blk / (p.nbj * p.nbi);
    A = p.dm + ((size_t)xi * p.Tp + oct * 8) * p.Cout + bi * WW_BM + cp;
    b_thread = (long)oct * 8 * p.Cin + bj * WW_BM + cp;
    u += st1 - st0;
    f32x16 acc[2][2];
#pragma unroll
    for (int m = 0; m < 2; ++m)
#pragma unroll
        for (int n = 0; n < 2; ++n)
#pragma unroll
            for (int i = 0; i < 16; ++i) acc[m][n][i] = 0.f;

    gload(st0);
    lstore();
    __syncthreads();
    auto stage = [&](const int st, auto more_c) {
        constexpr bool more = decltype(more_c)::value;
        if (more) gload(st + 1);
        f32x4 fa[2][2][3], fb[2][2][3];     // [buffer][tile][plane]
        auto read_step = [&](int buf, int s2) {
#pragma unroll
            for (int pl = 0; pl < 3; ++pl)
#pragma unroll
                for (int m = 0; m < 2; ++m) {
                    fa[buf][m][pl] = *reinterpret_cast<const f32x4*>(&As[fa_off + ((2 * s2) * 3 + pl) * WW_BM * 4 + m * 128]);
                    fb[buf][m][pl] = *reinterpret_cast<const f32x4*>(&Bs[fb_off + ((2 * s2) * 3 + pl) * WW_BM * 4 + m * 128]);
                }
        };
        read_step(0, 0);
        if (more) __builtin_amdgcn_sched_group_barrier(0x020, 32, 0);   // the next stage's global loads first
        __builtin_amdgcn_sched_group_barrier(0x100, 12, 0);
#pragma unroll
        for (int s2 = 0; s2 < 2; ++s2) {
            if (s2 == 0) read_step(1, 1);
            if (s2 == 1 && more) {
                __syncthreads();     // every wave holds its last fragments: the next stage's tiles may overwrite the image
                lstore();
            }
            auto mm = [&](int pa, int pb) {
#pragma unroll
                for (int m = 0; m < 2; ++m)
#pragma unroll
                    for (int n = 0; n < 2; ++n)
                        acc[m][n] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8_t, fa[s2][m][pa]),
                                                                            __builtin_bit_cast(bf16x8_t, fb[s2][n][pb]), acc[m][n], 0, 0, 0);
            };
            mm(2, 0); mm(1, 1); mm(0, 2); mm(1, 0); mm(0, 1); mm(0, 0);
            if (s2 == 0) {
#pragma unroll
                for (int r = 0; r < 12; ++r) {     // the second k-step's twelve reads, one per two MFMAs
                    __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
                    __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);
                }
            } else if (more) {
#pragma unroll
                for (int r = 0; r < 12; ++r) {     // the twelve ds_write_b128 of the next stage between the MFMAs
                    __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);
                    __builtin_amdgcn_sched_group_barrier(0x200, 1, 0);
                }
            }
        }
        __syncthreads();
    };
    for (int st = st0; st + 1 < st1; ++st) stage(st, std::integral_constant<bool, true>{});
    stage(st1 - 1, std::integral_constant<bool, false>{});

    const size_t slab = (size_t)36 * p.Cout * p.Cin;
    const int nzero = st1 == p.nst ? p.S - 1 - seg : 0;   // the block's last segment: zero the slabs it did not use
    float* out = p.part + (size_t)seg * slab + (((size_t)xi * p.Cout + bi * WW_BM + wr * 64) * p.Cin + bj * WW_BM + wc * 64 + l31);
#pragma unroll
    for (int m = 0; m < 2; ++m)
#pragma unroll
        for (int n = 0; n < 2; ++n)
#pragma unroll
            for (int reg = 0; reg < 16; ++reg) {
                const int row = m * 32 + (reg & 3) + 8 * (reg >> 2) + 4 * hh;
                out[(size_t)row * p.Cin + n * 32] = acc[m][n][reg];
            }
    float* zbase = p.part + (size_t)(seg + 1) * slab + ((size_t)xi * p.Cout + bi * WW_BM) * p.Cin + bj * WW_BM;
#pragma unroll 1
    for (int z = 0; z < nzero; ++z, zbase += slab)
#pragma unroll 1
        for (int i = tid; i < WW_BM * (WW_BM / 4); i += 256)
            *reinterpret_cast<f32x4*>(zbase + (size_t)(i >> 5) * p.Cin + (i & 31) * 4) = f32x4{0.f, 0.f, 0.f, 0.f};
    }   // segments
}

// packed[3a + b][co][ci] = (G^T (sum_s P[s]) G)[a][b]: the layout dvg_conv_wgrad's slabs have (S = 1), so that
// dvg_wgrad_finish places either form into the parameter's gradient.
__global__ __launch_bounds__(256) void wino_wgrad_reduce_kernel(const float* __restrict__ part, float* __restrict__ packed,
                                                                int S, long n) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    float t[3][6];   // G^T m over the first index
#pragma unroll
    for (int b = 0; b < 6; ++b) {
        float m[6];
#pragma unroll
        for (int a = 0; a < 6; ++a) {
            float s = 0.f;
            for (int k = 0; k < S; ++k) s += part[((size_t)k * 36 + a * 6 + b) * n + i];
            m[a] = s;
        }
        const float p12 = m[1] + m[2], p34 = m[3] + m[4];
        t[0][b] = 0.25f * m[0] - p12 * (1.f / 6) + p34 * (1.f / 24);
        t[1][b] = (m[2] - m[1]) * (1.f / 6) + (m[3] - m[4]) * (1.f / 12);
        t[2][b] = (p34 - p12) * (1.f / 6) + m[5];
    }
#pragma unroll
    for (int a = 0; a < 3; ++a) {
        const float* m = t[a];
        const float p12 = m[1] + m[2], p34 = m[3] + m[4];
        packed[(size_t)(a * 3 + 0) * n + i] = 0.25f * m[0] - p12 * (1.f / 6) + p34 * (1.f / 24);
        packed[(size_t)(a * 3 + 1) * n + i] = (m[2] - m[1]) * (1.f / 6) + (m[3] - m[4]) * (1.f / 12);
        packed[(size_t)(a * 3 + 2) * n + i] = (p34 - p12) * (1.f / 6) + m[5];
    }
}

static int ww_ks() { return WW_DEFAULT_KS; }   // (64-row stages in the single-buffer form: 237+ VGPRs, 2-4 % slower)

// q = stages per workgroup, W = workgroups, S = slabs (the most segments any output block is cut into)
static void ww_plan(long Tp, int Cin, int Cout, long* q_out, long* w_out, int* s_out) {
    const long nst = Tp / ww_ks();
    const long blocks = 36L * (Cin / WW_BM) * (Cout / WW_BM), total = blocks * nst;
    const long min_st = 256 / ww_ks();                 // >= 512 MFMAs per wave and workgroup
    // ONE round of resident workgroups, two per CU.  (Compiled for three per CU - 168 VGPRs, 9 of them spilled - and split 768
    // ways the single-buffer variant measured 107-112 TF against 115-123.)
    long wgs = 512;
    if (wgs * min_st > total) wgs = (total + min_st - 1) / min_st;
    const long q = (total + wgs - 1) / wgs;
    wgs = (total + q - 1) / q;
    int S = 1;
    for (long b = 0; b < blocks; ++b) {
        const int n = (int)((b * nst + nst - 1) / q - b * nst / q + 1);
        if (n > S) S = n;
    }
    *q_out = total < (1L << 30) ? q : 0;   // 0: does not fit the kernel's 32-bit stage arithmetic
    *w_out = wgs;
    *s_out = S;
}

template <bool DB, int KS>
static int ww_launch(WwParams p, hipStream_t stream) {
    static bool attr_set = false;
    if (!attr_set) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&wino_wgrad_gemm_kernel<DB, KS>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, ww_lds_bytes(DB, KS));
        if (e != hipSuccess) return fail(DVG_ERR_HIP, "hipFuncSetAttribute: %s", hipGetErrorString(e));
        attr_set = true;
    }
    const unsigned grid = (unsigned)((p.total + p.q - 1) / p.q);
    hipLaunchKernelGGL((wino_wgrad_gemm_kernel<DB, KS>), dim3(grid), dim3(256), ww_lds_bytes(DB, KS), stream, p);
    return check_launch("dvg_winograd_wgrad_gemm");
}

}  // namespace dvg

using namespace dvg;

extern "C" int dvg_winograd_wgrad_splits(long tiles_padded, int Cin, int Cout) {
    if (tiles_padded <= 0 || tiles_padded % 64 || Cin <= 0 || Cout <= 0 || Cin % WW_BM || Cout % WW_BM) return 0;
    long q, w;
    int S;
    ww_plan(tiles_padded, Cin, Cout, &q, &w, &S);
    return q > 0 ? S : 0;
}

static int ww_gemm(const float* dm, const float* const* v, int items, long t_item, float* partial, long tiles_padded, int Cin,
                   int Cout, void* stream);

extern "C" int dvg_winograd_wgrad_gemm(const float* dm, const float* v, float* partial, long tiles_padded, int Cin, int Cout,
                                       void* stream) {
    return ww_gemm(dm, &v, 1, tiles_padded, partial, tiles_padded, Cin, Cout, stream);
}

extern "C" int dvg_winograd_wgrad_gemm_items(const float* dm, const float* const* v_items, int items, long tiles_per_item,
                                             float* partial, int Cin, int Cout, void* stream) {
    DVG_REQUIRE(v_items && items >= 1 && items <= 8, DVG_ERR_SHAPE, "dvg_winograd_wgrad_gemm_items: 1..8 items");
    DVG_REQUIRE(tiles_per_item > 0 && tiles_per_item % 64 == 0, DVG_ERR_SHAPE,
                "dvg_winograd_wgrad_gemm_items: tiles per item (%ld) must be a multiple of 64", tiles_per_item);
    return ww_gemm(dm, v_items, items, tiles_per_item, partial, (long)items * tiles_per_item, Cin, Cout, stream);
}

static int ww_gemm(const float* dm, const float* const* v_items, int items, long t_item, float* partial, long tiles_padded,
                   int Cin, int Cout, void* stream) {
    DVG_REQUIRE(dm && v_items && partial, DVG_ERR_NULL, "dvg_winograd_wgrad_gemm: NULL pointer");
    for (int i = 0; i < items; ++i)
        DVG_REQUIRE(v_items[i] && aligned16(v_items[i]), DVG_ERR_NULL, "dvg_winograd_wgrad_gemm: V of item %d NULL / unaligned", i);
    const float* v = v_items[0];
    DVG_REQUIRE(tiles_padded > 0 && tiles_padded % 64 == 0 && Cin > 0 && Cout > 0 && Cin % WW_BM == 0 && Cout % WW_BM == 0,
                DVG_ERR_SHAPE, "dvg_winograd_wgrad_gemm: tiles=%ld must be a multiple of 64, Cin=%d / Cout=%d multiples of 128",
                tiles_padded, Cin, Cout);
    DVG_REQUIRE(aligned16(dm) && aligned16(v) && aligned16(partial), DVG_ERR_ALIGN, "dvg_winograd_wgrad_gemm: alignment");
    const int ks = ww_ks();
    WwParams p{dm, {}, (int)t_item, partial, tiles_padded, Cin, Cout, 1, (int)(tiles_padded / ks), Cout / WW_BM, Cin / WW_BM, 0, 0};
    for (int i = 0; i < 8; ++i) p.v[i] = v_items[i < items ? i : 0];
    long wgs, q;
    ww_plan(tiles_padded, Cin, Cout, &q, &wgs, &p.S);
    DVG_REQUIRE(q > 0, DVG_ERR_SHAPE, "dvg_winograd_wgrad_gemm: too many tiles (%ld)", tiles_padded);
    p.q = (int)q;
    p.total = 36 * p.nbi * p.nbj * p.nst;
    const bool db = WW_DEFAULT_DB;
    const hipStream_t st = (hipStream_t)stream;
#if DVG_BF16X3
    {
        static bool attr_set = false;
        if (!attr_set) {
            hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&wino_wgrad_gemm_x3_kernel),
                                               hipFuncAttributeMaxDynamicSharedMemorySize, ww3_lds_bytes());
            if (e != hipSuccess) return fail(DVG_ERR_HIP, "hipFuncSetAttribute: %s", hipGetErrorString(e));
            attr_set = true;
        }
        (void)db;
        const unsigned grid = (unsigned)((p.total + p.q - 1) / p.q);
        hipLaunchKernelGGL(wino_wgrad_gemm_x3_kernel, dim3(grid), dim3(256), ww3_lds_bytes(), st, p);
        return check_launch("dvg_winograd_wgrad_gemm");
    }
#endif
    return db ? ww_launch<true, 32>(p, st) : ww_launch<false, 32>(p, st);
}

extern "C" int dvg_winograd_wgrad_reduce(const float* partial, int S, float* packed, int Cin, int Cout, void* stream) {
    DVG_REQUIRE(partial && packed, DVG_ERR_NULL, "dvg_winograd_wgrad_reduce: NULL pointer");
    DVG_REQUIRE(S > 0 && Cin > 0 && Cout > 0, DVG_ERR_SHAPE, "dvg_winograd_wgrad_reduce: bad shape");
    const long n = (long)Cin * Cout;
    hipLaunchKernelGGL(wino_wgrad_reduce_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, partial,
                       packed, S, n);
    return check_launch("dvg_winograd_wgrad_reduce");
}
