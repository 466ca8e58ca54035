// fp32-MFMA implicit-GEMM convolutions ("v2" schedule; the first schedule, conv_igemm.hip, was retired in ABI 3).
//
// What changed, and why (profiles/r01_pmc_conv3x3.md: matrix pipe 2/3 busy, the rest were
// `ds_read -> s_waitcnt -> 8 MFMA` groups and one workgroup barrier per tap):
//   * a stage = ALL taps of a 16-channel K chunk: the halo tile (<=14 KB) and the weights of every
//     tap (9 x 64 x 16 floats) sit in LDS together, so there is ONE barrier pair per 72-144 MFMAs
//     per wave instead of one barrier per 16-32;
//   * the next stage's global loads are issued into registers before the stage's MFMAs and written
//     to LDS after them (T14 async-stage split of the CDNA guide): their latency hides under ~9 k
//     cycles of matrix work;
//   * fragments are double-buffered in registers across taps, so the LDS reads of tap t+1 are in
//     flight while the MFMAs of tap t issue;
//   * weights are re-packed [Cin/16][Cout/64][tap][64][row of 16 k-values]: a stage's weight tile is one
//     contiguous 36 KB (55 KB as bf16 triples) run (fully coalesced 16-B loads);
//   * DVG_BF16X3 (dvg_common.h, default): the products run on the bf16 matrix pipe - operands split exactly
//     into three bf16 terms, activations when the stage tile is written to LDS, weights when they are packed.
// BN is fixed at 64 (2 waves along N) so that two workgroups share a CU (<= 67 KB LDS each) and
// cover each other's stage hand-offs.
#include <cstdlib>
#include <type_traits>

#include "dvg_common.h"

// Stage schedule knobs, fixed by same-box A/B runs (tools/ab_variants.sh) over the 18 vgg_64 layers at B = 64:
//  DVG_WRITE_OVERLAP 1: barrier BEFORE the last tap's MFMAs, next stage's ds_writes interleaved with them
//                       (0: MFMA-less write pass between two barriers).                      84.7 % vs 80.8 % of peak
//  DVG_VMEM_POLICY: where the next stage's global loads issue: 0 = left to hipcc (sinks them behind the last taps:
//                       latency exposed at the ds_write), 1 = all at the stage top (delays the first MFMAs), 2 = two
//                       per tap from tap 0, 3 = three per tap, 4 = two per tap from tap 2.   4: 84.7, 2: 84.4, 3: 84.3
//                       (4 generalises to the 4- / 8-tap modes: from tap 0, enough per tap to place every load)
#ifndef DVG_VMEM_POLICY
#define DVG_VMEM_POLICY 4
#endif
#ifndef DVG_WRITE_OVERLAP
#define DVG_WRITE_OVERLAP 1
#endif
//  DVG_STAGE_PRIO 1: s_setprio(3 - (stage & 3)) at every stage start (see the stage loop)
#ifndef DVG_STAGE_PRIO
#define DVG_STAGE_PRIO 1
#endif

//  DVG_ABLATE / DVG_X3_TERMS / DVG_FIRST_SELECTS (timing experiments, WRONG results): declared in dvg_common.h, which refuses
//  to compile them unless the build says -DDVG_TIMING_EXPERIMENTS=1 (only `make variant` can pass that; dvg_build_info()
//  reports it and bench.py refuses to print a headline from such a library).
//  DVG_ABLATE: 1 = the stage loop issues no global loads (the next stage's LDS stores write stale registers), 2 = and no LDS
//  stores, 3 = and no workgroup barriers - what the staging costs the loop; 4 / 5 = no weight / activation LDS stores;
//  6 / 7 = neither loads nor stores of the weight / activation tile in the loop (6: what a weight-stationary workgroup would
//  save at most); 8 = 2 + the GEMM mode stores one product value per lane instead of 16; 9 = 8 without the workgroup
//  barriers: the bare fragment-read + MFMA loop
#ifndef DVG_GEMM_NT_STORE
#define DVG_GEMM_NT_STORE 0
#endif
//  DVG_GEMM_WGS_PER_CU: workgroups per CU the 64-row GEMM-mode tile is compiled for (register budget 512 / this per lane)
#ifndef DVG_GEMM_WGS_PER_CU
#define DVG_GEMM_WGS_PER_CU (DVG_BF16X3 ? 3 : 4)
#endif
//  DVG_GEMM128_WGS / _GT / _LEAN: the 128-row GEMM-mode tile.  bf16 triples: 3 workgroups per CU with K = 32 per stage (37 KB
//  of LDS) and the LEAN fragment schedule (158 VGPRs) - what the transposed-conv mode's best tile runs at; r04 same-box A/B at
//  the conditioning batch: 2077 us per pass against 2246 (2 per CU, K = 64) and 2267 (the 64-row tile).  f32 MFMA: as before.
#ifndef DVG_GEMM128_WGS
#define DVG_GEMM128_WGS (DVG_BF16X3 ? 3 : 2)
#endif
#ifndef DVG_GEMM128_GT
#define DVG_GEMM128_GT (DVG_BF16X3 ? 2 : 4)
#endif
#ifndef DVG_GEMM128_LEAN
#define DVG_GEMM128_LEAN 1
#endif
//  DVG_GEMM128_MIN_WGS: the batched GEMM takes the 128-row tile from this many (128-row) workgroups on - four residency rounds
//  (tile2 below; r06 A/B under the steady-state power cap: profiles/r06_ab_gemm128_threshold.txt)
//  DVG_GEMM_NT2: under the ENERGY tile policy (dvg_set_tile_policy) the batched GEMM takes the 128 x 128 tile (64 x 64 per wave,
//  NT = 2, two workgroups per CU) from this many workgroups of it on (0: never).  r06: under three rollouts in flight the board sits at its power cap (1 365 W, 1.98 GHz:
//  profiles/r06_power_trace_vgg.txt), so what a launch costs is its ENERGY, and the 128 x 128 tile moves a quarter of the LDS
//  bytes and half of the L2 -> LDS bytes per MFMA of the 64 x 64 one: vgg_64 rollouts in flight 49.2 -> 51.5 k frames/s (+4.6 %),
//  while ONE chain of launches - not power-bound, the finer tiles balance the CUs better - goes 15.7 -> 16.1 ms
//  (profiles/r06_ab_gemm_nt2.txt: thresholds 128 / 256 / 512 and 2 / 3 / 4 rollouts in flight).
#ifndef DVG_GEMM_NT2
#define DVG_GEMM_NT2 256
#endif
//  DVG_TILE16_MIN_WGS: under the ENERGY tile policy the 8 x 16 pixel tile (two accumulator tiles per wave: a third fewer fragment
//  reads per MFMA) from this many workgroups of it on, else 8 x 8 (LATENCY policy: 512, two workgroups per CU).  256 = one per CU, for the same reason as
//  DVG_GEMM_NT2: in flight dcgan_64 217.2 -> 219.3 k, vgg_64 50.1 -> 50.3 k frames/s, one chain unchanged (profiles/r06_ab_tile16.txt)
#ifndef DVG_TILE16_MIN_WGS
#define DVG_TILE16_MIN_WGS 256
#endif
#ifndef DVG_GEMM128_MIN_WGS
#define DVG_GEMM128_MIN_WGS (4 * 768)
#endif

namespace dvg {

// M2_GEMM: batched GEMM y[n][px][co] = sum_ci x[n][px][ci] * w[n][ci][co] on the same machinery (a "1x1 conv" whose weight
// depends on the image index n): the 16 Winograd-domain products of dvg_winograd_* (n = transform position).  A stage is
// GT consecutive 16-channel slabs (K = 64) instead of the taps of one chunk.
enum { M2_CONV3 = 0, M2_CONV4S2 = 1, M2_CONVT4S2 = 2, M2_GEMM = 3 };

struct Igemm2Params {
    const float* x;
    const float* skip;
    const float* w;      // packed [Cin/16][Cout/64][taps][64][DVG_WROW] (dvg_pack_conv_weight_k16); GEMM: [image][Cout/64][Cin/16][64][DVG_WROW]
    const float* scale;
    const float* shift;
    float* y;
    float* y_pool;
    float* stats;
    int N, H, W, C1, C2, Cout, upsample, act;
    float slope;
    int tiles_y, tiles_x, tiles_n, nblk_n;
    int splitk;  // K split across workgroups (v2 only): raw partial tiles go to `ws`, dvg finishes with splitk_finish
    int cps;     // K chunks (of 16 channels) per split
    float* ws;   // [splitk][N*Ho*Wo][Cout]
    const float* addend;  // optional raw (pre-scale) partial sums, NHWC like y: y = act((acc + addend) * scale + shift)
    unsigned long long* clk;  // debug only (dvg_debug_set_clockbuf): per-workgroup {clock64, wall_clock64} at entry/exit
    unsigned clk_cap;         // records the buffer holds (workgroups beyond it do not stamp)
    int nb_group;  // Cout blocks per XCD-contiguous group of the workgroup order (launch2 picks it; v2 only)
    int stage_prio;  // progress-based s_setprio in the stage loop (launch2 decides; see the stage loop)
    long w_image_stride;  // M2_GEMM: floats between the packed weights of consecutive images n
    int gemm_ni;          // M2_GEMM: consecutive images one workgroup runs back to back (one pipeline, one epilogue per image)
    // Time-batched decoder calls (ABI 6): the images form groups of add_B; group g = n / add_B reads its addend from block
    // add_map[g] of `addend` (an addend of add_B-image blocks) - the skip half of a concat conv is shared by the three
    // decoder calls of a time step and, once the skip is frozen, by all later steps (train.py:217-231).  NULL: image n.
    const int* add_map;
    int add_B;
    // FIRST (dvg_conv3x3_first_pair): the input of this 64-channel 3x3 layer is itself the first layer of the encoder,
    // vgg_layer(1, 64) on the raw frame (vgg_64.py:23-24): the workgroup computes the halo tile of that layer from the frame
    // patch under it instead of loading it, so the 64-channel activation between the two layers is never written or read
    const float* first_frame;   // (N, 1, H, W)
    const float* first_w;       // the first layer's weights transposed to [9 taps][64 channels]
    const float* first_scale;   // folded BatchNorm of the first layer, 64 each
    const float* first_shift;
    float first_slope;
    // FIRST: images n < y_from do not store the full-resolution output (only y_pool): a rollout discards the encoder's skip
    // tensors once the skip is frozen (generate_frames.py:154-157), and of the conditioning batch only the last frame's are
    // kept.  y is biased by -y_from images on the host, so image y_from lands at the start of the caller's buffer.
    int y_from;
};

// image of `addend` that output image n adds (see Igemm2Params::add_map)
__device__ __forceinline__ int addend_image(const int* __restrict__ add_map, int add_B, int n) {
    return add_map ? add_map[n / add_B] * add_B + n % add_B : n;
}

static unsigned long long* g_clk = nullptr;
static unsigned g_clk_cap = 0;

// What a halo / out-of-image slot of the A tile loads: 16 bytes of zeros.  The slot's ADDRESS is selected (before the load
// is issued), not its data afterwards: with `v = ok ? loaded : 0` hipcc placed the v_cndmask right behind the load and with
// it an `s_waitcnt vmcnt(0)` in the middle of the stage's MFMA stream - every stage of every mode waited out a full
// global-load latency for its first A loads (r03, found in the ISA of all nine instantiations).
__device__ __attribute__((aligned(16))) float dvg_zero_slot[4] = {0.f, 0.f, 0.f, 0.f};

// NT: 32-column tiles per wave (workgroup tile BM x 64 NT).  NT = 2 exists for the bf16-triple GEMM mode only: with the
// matrix pipe 2.7x faster the LDS, not the MFMA, bounds a 128 x 64 tile (per MFMA 384 B of tile stores at ~80 B/clk and
// 768 B of fragment reads at 256 B/clk: 97 % of the LDS cycles); 128 x 128 with 64 x 64 per wave needs 256 + 512 B (65 %).
template <int MODE, int TI, int TH, int TW, int NT_ = 1, bool FIRST_ = false>
struct Cfg2 {
    static constexpr int NT = NT_;
    static constexpr bool FIRST = FIRST_;
    static constexpr bool GEMM = MODE == M2_GEMM;
    static constexpr int S = (MODE == M2_CONV4S2) ? 2 : 1;                  // stride in the IMAGE
    // PAR4 (r05, the stride-2 conv): the 16 taps of a chunk run as FOUR stages, one per input parity (alpha, beta) = (tap row & 1,
    // tap column & 1).  The pixels (2 y - 1 + alpha + 2 a', 2 x - 1 + beta + 2 b') a parity's taps (2 a' + alpha, 2 b' + beta)
    // read form a stride-1 grid of (TH + 1) x (TW + 1) pixels - the space-to-depth view of the conv as a 2 x 2 stride-1 conv
    // on 4 C channels - so a stage's A tile is a quarter of the 18 x 18 halo (14 instead of 43 KB for 8 x 8 outputs: four
    // workgroups per CU instead of two, and the 8 x 16 tile with two accumulator tiles per wave fits in 46 KB) and its taps are
    // whole-row offsets in LDS like the other modes'.  Same global bytes: every halo pixel is still loaded once per chunk.
    static constexpr bool PAR4 = MODE == M2_CONV4S2;
    static constexpr int SPAN = GEMM ? 1 : (PAR4 ? 2 : 3);
    static constexpr int HH = (TH - 1) + SPAN, HW = (TW - 1) + SPAN;        // LDS tile (stride 1 in every mode)
    // Conflict-free A-fragment reads of the halo-tiled modes (r06).  `ds_read_b128` is served in four fixed 16-lane groups,
    // {0-3, 12-15, 20-27}, {4-11, 16-19, 28-31} and the same + 32 (MI355X_MICROARCH.md, LDS): a group's 16 rows must hit 16
    // distinct 16-byte slots of the 256-byte bank row.  Rows are an ODD number of slots long (112 B = 7, f32 build 80 B = 5),
    // so slot = pos x odd + const (mod 16) and what matters is that the 16 pixels of a group are distinct mod 16 in their
    // LDS position pos = image x IMG + row x HWP + column.  With lane l holding pixel l of the 32-row tile the halo pitches
    // (18, 10, 17, 9, 6, 5) put 2-3 rows on a slot: 30-42 % of all LDS cycles were conflict cycles in every one of these
    // instantiations (profiles/r05_pmc_by_kernel.json) while the LDS was busy 44-79 % of the time.  Fix = (a) WHICH pixel a
    // lane holds: lane-quad q = l >> 2 holds pixel-quad PERM[q], PERM[2 b + h] = P0[b] ^ (h PK) (so that the C-layout row of
    // accumulator register `reg` in lane half hh is (4 P0[reg >> 2] + (reg & 3)) ^ (4 PK hh): one XOR, no select), and (b) a
    // padded pixel pitch where no permutation suffices (8-wide tiles: 12; the 4 x 4 x 4-image tiles: rows of 6 pixels,
    // images 40 (3 x 3 halo) / 30 (parity-split stride-2 conv: the best that fits its 128 staged rows - 2 residual conflict
    // cycles per 8 reads instead of 8) pixels apart).  Found by exhaustive search (tools/lds_layout_search.py, which also
    // checks that vertical 2 x 2 pool partners stay in one lane: register blocks (0, 1) and (2, 3)).  The padding pixels are
    // never read and cost no LDS: the staged rows were already rounded up to whole 256-thread store passes.
    // DVG_HALO_LAYOUT=0: the r05 layout (lane l = pixel l, unpadded pitches).
#ifndef DVG_HALO_LAYOUT
#define DVG_HALO_LAYOUT 1
#endif
    static constexpr bool LAYOUT = !GEMM && DVG_HALO_LAYOUT != 0;
    static constexpr int HWP = !LAYOUT ? HW : (TW == 16 ? HW : (TW == 8 ? 12 : 6));         // pixel pitch of a halo row
    static constexpr int IMG = (LAYOUT && TI == 4) ? (PAR4 ? 30 : 40) : HH * HWP;           // pixel pitch of an image
    static_assert(IMG >= HH * HWP && HWP >= HW, "halo layout");
    static constexpr int HP = TI * IMG;
    // P0[b] as four 4-bit fields, and PK
    static constexpr unsigned P0PACK = !LAYOUT ? 0x6420u
        : (PAR4 ? (TW == 16 ? 0x3650u : (TI == 4 ? 0x7430u : 0x6530u))
                : (TW == 16 ? 0x2640u : (TI == 4 ? 0x6530u : 0x4620u)));
    static constexpr int PK = !LAYOUT ? 1 : (PAR4 ? (TW == 16 ? 4 : 1) : (TW == 16 ? 5 : (TI == 4 ? 1 : 3)));
    // ... and the A-tile STORES: four lanes write one pixel's 32 bytes of a plane (ds_write_b64, served in groups of 16
    // consecutive lanes = 4 pixels on a 128-byte bank window).  Consecutive pixels are 112 B apart = 7 x 16 B, so pixels p and
    // p + 1 overlap by 16 B (2-way, on every store); pixels 2 apart are 96 B = -32 B apart (mod 128): the 4 x 32 B of pixels
    // p, p + 2, p + 4, p + 6 tile the window exactly.  So thread slot k (= 4 lanes) of every 8-pixel block stages pixel
    // 2 (k & 3) + (k >> 2) of it.  (The global loads do not care: a pixel's 16-channel chunk is its own 64-byte segment.)
    // bf16-triple halo modes only: the f32 build's 80-byte rows and the GEMM mode's swizzled 96-byte rows store conflict-free
    // as they are.  Residual SQ_LDS_BANK_CONFLICT after the read fix alone: 6-14 % (profiles/r06_pmc_lds_after_layout.txt).
    static constexpr bool STORE_PERM = LAYOUT && DVG_BF16X3 != 0;
    static constexpr int hp_of(int slot) {     // staged pixel row of thread slot `slot` = idx >> 2
        return STORE_PERM ? ((slot & ~7) | (((slot & 3) << 1) | ((slot >> 2) & 1))) : slot;
    }
    // row (0..31) of the wave's 32-row tile that accumulator register `reg` holds in lane half 0; half 1: ^ (4 * PK)
    static constexpr int row_c(int reg) { return 4 * (int)((P0PACK >> (4 * (reg >> 2))) & 15u) + (reg & 3); }
    static constexpr int BM = TI * TH * TW, MT = BM / 64, BN = 64 * NT;
    static constexpr int NTAPS = GEMM ? 1 : ((MODE == M2_CONV3) ? 9 : 16);
    // DVG_GEMM_GT: 16-channel slabs per stage of the GEMM modes.  4 (K = 64, 60 KB of LDS with the 128-row tile); 8 was
    // measured 10-25 % slower on every Winograd shape (80 KB per workgroup: the second workgroup no longer fits the CU).
#ifndef DVG_GEMM_GT
#define DVG_GEMM_GT 4
#endif
    // (NT = 2: two slabs, K = 32, so that two 48 KB workgroups share a CU)
    static constexpr int GT = (MODE == M2_CONV3) ? 9 : (MODE == M2_CONV4S2 ? 4 : (GEMM ? (NT == 2 ? 2 : (BM == 128 ? DVG_GEMM128_GT : DVG_GEMM_GT)) : 4));  // taps (GEMM: 16-channel slabs) per stage
    static constexpr int NG = (MODE == M2_CONV4S2) ? 16 / GT : 1;                     // stages per K chunk
    static constexpr int CHUNKS_PER_STAGE = GEMM ? GT : 1;                            // 16-channel chunks one stage consumes
    static constexpr bool X3 = DVG_BF16X3 != 0;
    // LDS rows (floats).  f32: 16 k-values + 4 of padding (80 B: b128 lane groups land on distinct 16-B slots).  bf16 triples:
    // 3 planes x 16 bf16 = 96 B; the weight rows (and the GEMM modes' A rows, whose tap offsets are whole slabs) swap the two
    // 16-B halves of a plane on rows with bit 3 set instead of padding, the conv modes' halo rows are padded to 112 B (their
    // tap offsets move a lane to another row, so the swap cannot be folded into a per-lane base).
    static constexpr int KC = 16, LD = X3 ? (GEMM ? 24 : 28) : 20, LDB = X3 ? 24 : 20, WROW = DVG_WROW;
    static constexpr int NP = X3 ? 3 : 2, PSTEP = X3 ? 8 : 4;   // 16-B pieces of a lane's fragment (planes / k halves), floats between them
    static constexpr int BLK4 = GT * 64 * WROW / 4;             // float4s of a stage's packed weight tile, per 64-column block
    static constexpr int B_TILE4 = NT * BLK4;
    static constexpr int NLB = (B_TILE4 + 255) / 256;            // of them per thread
    static_assert(NT == 1 || (DVG_BF16X3 != 0 && MODE == M2_GEMM && BLK4 % 256 == 0), "NT = 2: bf16-triple GEMM mode only");
    static constexpr int NLA1 = (HP * 4 + 255) / 256;                                 // float4 loads per thread per slab
    static constexpr int NLA = NLA1 * (GEMM ? GT : 1);
    static constexpr int SLAB = NLA1 * 64 * LD;                                       // floats of one A slab (GEMM mode)
    // the A region is padded to whole 256-thread store passes (NLA * 64 rows): the halo store is branch-free
    static constexpr int A_FLOATS = NLA * 64 * LD, B_FLOATS = GT * BN * LDB;
    // FIRST: the frame patch under the halo tile, (HH + 2) x (HW + 2) floats, in LDS (the first layer's weights, scale and
    // shift of a stage's 16 channels travel in registers): 77.8 KB, two workgroups per CU.  (r03 dropped the A tile's 12 padding
    // rows here on the reading that "two 80.6 KB allocations are both admitted and then corrupt each other's last
    // kilobytes".  tools/ubench/lds_oversubscribe.hip, r04: the hardware never co-schedules two workgroups whose LDS sums
    // beyond 160 KiB - 2 x 81 912 B share a CU and keep their bytes, from 82 432 B on there is one workgroup per CU - so
    // whatever corrupted that build was not the allocation; with the padding rows back the halo store is branch-free
    // again, like every other instantiation's: no EXEC change inside the stage loop, tests/test_isa_invariants.py.)
    static constexpr int FP_W = HW + 2, FP_FLOATS = (HH + 2) * FP_W, FIRST_FLOATS = FIRST ? FP_FLOATS : 0;
    static_assert(!FIRST || (MODE == M2_CONV3 && TI == 1 && NT == 1), "FIRST: 3x3 mode, one image per tile");
    static constexpr int LDS_BYTES = (A_FLOATS + B_FLOATS + FIRST_FLOATS) * 4;
    // K is split only where a launch would leave CUs idle (< 384 workgroups): never on the 16-wide tiles (chosen from 512
    // workgroups on), the first layer or the batched GEMM - their kernels do not carry the split-K epilogue (its scalars cost
    // the 8x16 kernels 18 ... 35 SGPR spills)
    static constexpr bool CAN_SPLIT = !FIRST && !GEMM && TW != 16;
    static_assert(BM == 64 || BM == 128 || BM == 256, "BM");
};

template <int MODE, int TI, int TH, int TW, int NT = 1, bool FIRST = false>
__global__ __launch_bounds__(256, (MODE == M2_GEMM && TW == 8) ? DVG_GEMM_WGS_PER_CU : ((MODE == M2_GEMM && NT == 1) ? DVG_GEMM128_WGS : 2)) void conv_igemm2_kernel(const Igemm2Params p) {
    using C = Cfg2<MODE, TI, TH, TW, NT, FIRST>;
    constexpr int S = C::S, HH = C::HH, HW = C::HW, HP = C::HP, LD = C::LD, MT = C::MT, GT = C::GT, NG = C::NG,
                  BN = C::BN, NLA = C::NLA, NLA1 = C::NLA1, LDB = C::LDB, NP = C::NP, PSTEP = C::PSTEP, NLB = C::NLB;
    constexpr bool GEMM = C::GEMM, X3 = C::X3;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* As = smem;
    float* Bs = smem + C::A_FLOATS;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1, l31 = lane & 31, hh = lane >> 5;

    unsigned lid = xcd_remap(blockIdx.x, gridDim.x);
    int par = 0;
    if (MODE == M2_CONVT4S2) { par = lid & 3; lid >>= 2; }
    const int split = lid % p.splitk;
    lid /= p.splitk;
    // Order: Cout-block-within-group fastest, then pixel tile, then group.  xcd_remap hands each XCD a contiguous lid
    // range, i.e. nb_group Cout blocks x (range / nb_group) pixel tiles: the host picks nb_group so that the weight
    // slabs plus the input tiles an XCD's L2 has to fetch are smallest (with all Cout blocks per XCD every L2 pulled
    // the full 9.4 MB of an 8x8 512->512 layer's weights).
    const int tiles_total = p.tiles_x * p.tiles_y * p.tiles_n;
    const unsigned lr = lid / p.nb_group;
    const int nb = (int)(lr / tiles_total) * p.nb_group + (int)(lid % p.nb_group);
    const unsigned tile_id = lr % tiles_total;
    unsigned t = tile_id;
    const int tx_i = t % p.tiles_x; t /= p.tiles_x;
    const int ty_i = t % p.tiles_y; t /= p.tiles_y;
    const int n0 = (int)t * (MODE == M2_GEMM ? p.gemm_ni : TI), y0 = ty_i * TH, x0 = tx_i * TW;
    const int yin0 = y0 * S - (GEMM ? 0 : 1), xin0 = x0 * S - (GEMM ? 0 : 1), nb0 = nb * BN;
    const float* const wbase = p.w + (GEMM ? (size_t)n0 * p.w_image_stride : 0);
    const int py = par >> 1, px = par & 1;
    const int Cin = p.C1 + p.C2;

    int a_base[MT];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
        // the pixel this lane's A-operand row holds: lane-quad l31 >> 2 = 2 b + h -> pixel-quad P0[b] ^ (h PK) (Cfg2: LAYOUT)
        const int lp = 4 * ((int)((C::P0PACK >> (4 * (l31 >> 3))) & 15u) ^ (((l31 >> 2) & 1) * C::PK)) + (l31 & 3);
        const int m = wm * (C::BM / 2) + mt * 32 + lp;
        const int ti = m / (TH * TW), r = m % (TH * TW);
        const int pos = ti * C::IMG + (r / TW) * C::HWP + r % TW;
        a_base[mt] = pos * LD + (X3 ? (hh ^ (GEMM ? (pos >> 3) & 1 : 0)) * 4 : hh * 8);
    }
    // LDS image of the weight tile: [64-column block][tap][64 rows][LDB]; the wave's 32-column tile nt is rows
    // (wn * NT + nt) * 32 ... + 31 of the BN columns
    int b_base[NT];
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
        const int col = (wn * NT + nt) * 32 + l31;
        b_base[nt] = ((col >> 6) * GT * 64 + (col & 63)) * LDB + (X3 ? (hh ^ ((l31 >> 3) & 1)) * 4 : hh * 8);
    }

    f32x16 acc[MT * NT];     // [mt][nt]
#pragma unroll
    for (int mt = 0; mt < MT * NT; ++mt)
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[mt][i] = 0.f;

    // ---- loader geometry, computed once --------------------------------------------------------
    // element offsets of this thread's halo float4s in x / skip.  Halo / out-of-image slots read offset 0 (a
    // valid address) and are zeroed at the LDS write (okmask): no divergent branch, and no per-path wait
    // bookkeeping, around the loads.
    // GEMM mode: the GT slabs of a stage share their row offsets (slab s = + 16 s floats): NLA1 offsets, no skip operand
    constexpr int NOFF = GEMM ? NLA1 : NLA;
    // GEMM: offsets inside one image; PAR4: inside the whole activation (the host checks < 2^31 floats in both cases)
    using aoff_t = typename std::conditional<GEMM || C::PAR4, int, long>::type;
    // a 0 / 1 value the optimiser cannot see through: `bit * step` stays a multiply instead of becoming a select on a lane
    // mask (a dozen hoisted masks ran the kernel out of SGPRs, and the spilled ones were re-made inside the stage loop)
    auto opaque = [](unsigned v) { asm("" : "+v"(v)); return v; };
    aoff_t offx[NOFF], offs[GEMM ? 1 : NLA];
    unsigned okmask = 0;      // PAR4: bit par * NLA + i (the parity's pixel of slot i lies inside the image), 16 + i / 24 + i (steps)
    static_assert(!C::PAR4 || NLA <= 4, "PAR4: four validity bits and two step bits per halo slot in one word");
    if (GEMM) offs[0] = 0;
#pragma unroll
    for (int i = 0; i < NLA; ++i) {
        const int idx = tid + (i % NLA1) * 256;
        const int hp = C::hp_of(idx >> 2), q = idx & 3;
        // LDS pixel row hp -> (image, halo row, halo column); rows / columns of the layout's padding are never loaded
        const int ti = hp / C::IMG, r = hp % C::IMG;
        const int hy = r / C::HWP, hx = r % C::HWP;
        const bool slot = hp < HP && hy < HH && hx < HW;
        if constexpr (C::PAR4) {
            // slot (hy, hx) of parity (alpha, beta) is image pixel (yin0 + 2 hy + alpha, xin0 + 2 hx + beta).  ONE offset per slot:
            // parity (0, 0)'s pixel CLAMPED into the image, plus a row / column step per parity that is 0 where the step would
            // leave the image (rows -1 -> 0 and H - 1 -> H, likewise columns).  Every load reads inside the tensor; the conv's
            // zero padding is applied to the VALUE at the LDS store (an integer AND with 0 / ~0: no lane mask, no select, exact
            // zeros whatever was loaded).
            const int n = n0 + ti, yy = yin0 + 2 * hy, xx = xin0 + 2 * hx;
            const bool in = slot && n < p.N;
#pragma unroll
            for (int par_ = 0; par_ < 4; ++par_) {
                const bool ok = in && (unsigned)(yy + (par_ >> 1)) < (unsigned)p.H && (unsigned)(xx + (par_ & 1)) < (unsigned)p.W;
                okmask |= ok ? (1u << (par_ * NLA + i)) : 0u;
            }
            const int cy0 = min(max(yy, 0), p.H - 1), cy1 = min(max(yy + 1, 0), p.H - 1);
            const int cx0 = min(max(xx, 0), p.W - 1), cx1 = min(max(xx + 1, 0), p.W - 1);
            okmask |= (in && cy1 != cy0) ? (1u << (16 + i)) : 0u;
            okmask |= (in && cx1 != cx0) ? (1u << (24 + i)) : 0u;
            offx[i] = in ? (int)((((long)n * p.H + cy0) * p.W + cx0) * p.C1 + q * 4) : 0;
            offs[i] = 0;                                                    // (no skip operand in this mode)
        } else {
        const int n = n0 + ti, yy = yin0 + hy, xx = xin0 + hx;
        const bool ok = slot && n < p.N && (unsigned)yy < (unsigned)p.H && (unsigned)xx < (unsigned)p.W;
        const int sh = p.upsample;
        okmask |= ok ? (1u << i) : 0u;
        if (!GEMM || i < NLA1)
            offx[GEMM ? i % NLA1 : i] = (aoff_t)(ok ? ((((long)(GEMM ? 0 : n) * (p.H >> sh) + (yy >> sh)) * (p.W >> sh) + (xx >> sh)) * p.C1 + q * 4) : 0);
        if (!GEMM) offs[i] = (ok && p.C2) ? ((((long)n * p.H + yy) * p.W + xx) * p.C2 + q * 4) : 0;
        }
    }
    auto tap_lds = [&](int grp, int tt) -> int {
        if (GEMM) return tt * C::SLAB;
        int th, tw;
        if (MODE == M2_CONV3) { th = tt / 3; tw = tt % 3; }
        else if (MODE == M2_CONV4S2) { th = tt >> 1; tw = tt & 1; }      // PAR4: tap (2 th + alpha, 2 tw + beta) of the parity grp
        else { th = 1 + py - (tt >> 1); tw = 1 + px - (tt & 1); }
        return (th * C::HWP + tw) * LD;
    };
    // M2_GEMM: image (relative to n0) whose operands the NEXT loads fetch; the stage loop moves it on at image boundaries
    long ld_a_off = GEMM ? (long)n0 * p.H * p.W * p.C1 : 0;
    const float* ld_w = wbase;
    // FIRST: the frame patch behind the two tiles (origin (yin0 - 1, xin0 - 1)); the first layer's weights / scale / shift of
    // the NEXT stage's 16 channels in registers: this thread's four channels (all its halo slots share q = tid & 3)
    float* const Fp = smem + C::A_FLOATS + C::B_FLOATS;
    f32x4 fw[FIRST ? 11 : 1];      // [tap 0..8] weights of channels c0 + 4 q .. + 3, [9] scale, [10] shift
    auto first_tile = [&](f32x4 (&ra)[NLA]) {
        if constexpr (FIRST) {
            // the halo tile's channels of vgg_layer(1, 64) from the frame patch: 9 taps per value, folded BatchNorm, LeakyReLU;
            // slots outside the image are the SECOND layer's zero padding
#if DVG_FIRST_SELECTS == 2     // diagnostic: the select form behind a full drain of the memory counters
            asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
#elif DVG_FIRST_SELECTS == 3   // diagnostic: the select form with the scheduler fenced off (no MFMA interleaved with it)
            __builtin_amdgcn_sched_barrier(0);
#endif
#pragma unroll
            for (int i = 0; i < NLA; ++i) {
                const int hp = min(C::hp_of((tid + i * 256) >> 2), HP - 1);
                const float* f = Fp + (hp / C::HWP) * C::FP_W + hp % C::HWP;
                f32x4 v = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int t = 0; t < 9; ++t) {
                    const float px = f[(t / 3) * C::FP_W + t % 3];
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[e] = fmaf(fw[t][e], px, v[e]);
                }
                // LeakyReLU as max(a, slope a) (0 < slope < 1) and the padding as a multiply by 0 / 1: no lane masks in this block
#if DVG_FIRST_SELECTS      // diagnostic build only (tools/ubench/first_pair_selects.md): the form that gave run-to-run different tiles
                const bool okb = (okmask >> i) & 1u;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const float a = v[e] * fw[9][e] + fw[10][e];
                    v[e] = okb ? (a > 0.f ? a : a * p.first_slope) : 0.f;
                }
#else
                const float okf = (float)((okmask >> i) & 1u);
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const float a = v[e] * fw[9][e] + fw[10][e];
                    v[e] = fmaxf(a, a * p.first_slope) * okf;
                }
#endif
                ra[i] = v;
            }
        }
    };
    auto gload_a = [&](int c0, f32x4 (&ra)[NLA], const int par_ = 0) {
        if constexpr (FIRST) {      // no activation to load: the tile is computed where it is stored (first_tile, lds_store_a)
            const float* w = p.first_w + c0 + (tid & 3) * 4;                      // [9 taps][64 channels]
#pragma unroll
            for (int t = 0; t < 9; ++t) fw[t] = *reinterpret_cast<const f32x4*>(w + t * 64);
            fw[9] = *reinterpret_cast<const f32x4*>(p.first_scale + c0 + (tid & 3) * 4);
            fw[10] = *reinterpret_cast<const f32x4*>(p.first_shift + c0 + (tid & 3) * 4);
            return;
        }
        const bool from_x = GEMM || c0 < p.C1;
        const float* src = (from_x ? p.x + c0 : p.skip + (c0 - p.C1)) + (GEMM ? ld_a_off : 0);
#pragma unroll
        for (int i = 0; i < NLA; ++i) {
            if (GEMM) {    // whole tiles only (host checks): every slot is valid
                ra[i] = *reinterpret_cast<const f32x4*>(src + offx[i % NLA1] + (i / NLA1) * 16);
            } else if (C::PAR4) {
                int off = offx[i];
                if (par_ >> 1) off += (int)opaque((okmask >> (16 + i)) & 1u) * (p.W * p.C1);
                if (par_ & 1) off += (int)opaque((okmask >> (24 + i)) & 1u) * p.C1;
                ra[i] = *reinterpret_cast<const f32x4*>(src + off);
            } else {
                const float* a = src + (from_x ? offx[GEMM ? 0 : i] : offs[GEMM ? 0 : i]);
                ra[i] = *reinterpret_cast<const f32x4*>(((okmask >> i) & 1u) ? a : dvg_zero_slot);
            }
        }
    };
    // a stage's weight tile is ONE contiguous run of the packed tensor (GT slots x 64 rows): thread t takes float4 t + 256 j
    auto gload_b = [&](int chunk, int grp, f32x4 (&rb)[NLB]) {
        const float* tile;
        if (GEMM) tile = ld_w + ((size_t)nb * NT * (Cin / C::KC) + chunk) * (64 * C::WROW);
        else tile = wbase + (((size_t)chunk * (p.Cout >> 6) + nb) * C::NTAPS + (MODE == M2_CONVT4S2 ? par * 4 : grp * GT)) * (64 * C::WROW);
#pragma unroll
        for (int j = 0; j < NLB; ++j) {
            int i = tid + j * 256;
            if (C::B_TILE4 % 256) i = min(i, C::B_TILE4 - 1);   // the last pass of a tile that is not whole passes re-reads its last float4
            // NT = 2 (GEMM): the second 64-column block's run lies Cin/16 rows of 64 further on (whole passes per block)
            const size_t blk = (NT > 1 && j * 256 >= C::BLK4) ? (size_t)(Cin / C::KC) * (64 * C::WROW / 4) - C::BLK4 : 0;
            rb[j] = reinterpret_cast<const f32x4*>(tile)[i + blk];
        }
    };
    auto lds_store_a = [&](f32x4 (&ra)[NLA], const int par_ = 0) {
        if constexpr (FIRST) first_tile(ra);
#pragma unroll
        for (int i = 0; i < NLA; ++i) {
            const int idx = tid + (i % NLA1) * 256;
            const int hp = C::hp_of(idx >> 2), q = idx & 3;
            if constexpr (C::PAR4) {       // the zero padding (and the slots past the tile / the batch): value AND 0 / ~0
                const unsigned keep = 0u - opaque((okmask >> (par_ * NLA + i)) & 1u);
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const float f = ra[i][e];      // (a copy: __builtin_bit_cast on the vector-element lvalue reads element 0)
                    ra[i][e] = __builtin_bit_cast(float, __builtin_bit_cast(unsigned, f) & keep);
                }
            }
            // halo / out-of-image slots already hold zeros (gload_a read dvg_zero_slot for them); rows >= HP: padding
            if constexpr (X3) {
                // the thread's four k-values as three bf16 quadruples, 8 bytes into each plane of the row
                u32x2_t h, m, l;
                unsigned t0, t1, t2;
                bf16x3_split_pair(ra[i][0], ra[i][1], t0, t1, t2);
                h[0] = t0; m[0] = t1; l[0] = t2;
                bf16x3_split_pair(ra[i][2], ra[i][3], t0, t1, t2);
                h[1] = t0; m[1] = t1; l[1] = t2;
                float* d = &As[(i / NLA1) * C::SLAB + hp * LD + ((q >> 1) ^ (GEMM ? (hp >> 3) & 1 : 0)) * 4 + (q & 1) * 2];
                *reinterpret_cast<u32x2_t*>(d) = h;
                *reinterpret_cast<u32x2_t*>(d + 8) = m;
                *reinterpret_cast<u32x2_t*>(d + 16) = l;
            } else {
                *reinterpret_cast<f32x4*>(&As[(i / NLA1) * C::SLAB + hp * LD + q * 4]) = ra[i];
            }
        }
    };
    auto lds_store_b = [&](const f32x4 (&rb)[NLB]) {
#pragma unroll
        for (int j = 0; j < NLB; ++j) {
            int i = tid + j * 256;
            if (C::B_TILE4 % 256) i = min(i, C::B_TILE4 - 1);
            if constexpr (X3) reinterpret_cast<f32x4*>(Bs)[i] = rb[j];           // the packed tile IS the LDS image
            else *reinterpret_cast<f32x4*>(&Bs[(i >> 2) * LDB + (i & 3) * 4]) = rb[j];
        }
    };

    // chunk = index of a 16-channel chunk; a stage consumes CPS of them (1, or GT slabs in GEMM mode)
    constexpr int CPS = C::CHUNKS_PER_STAGE;
    const int chunk_begin = split * p.cps;
    const int chunk_end = min(Cin / C::KC, chunk_begin + p.cps);
    f32x4 ra[NLA], rb[NLB];
    unsigned long long clk0 = 0, wclk0 = 0;
    if (p.clk) {   // wave-uniform condition: the stamps live in SGPRs (under `threadIdx.x == 0` they cost 8 VGPRs kernel-wide)
        clk0 = clock64();
        wclk0 = wall_clock64();
    }
    if constexpr (FIRST) {
        const float* fr = p.first_frame + (size_t)n0 * p.H * p.W;
        for (int i = tid; i < C::FP_FLOATS; i += 256) {
            const int yy = yin0 - 1 + i / C::FP_W, xx = xin0 - 1 + i % C::FP_W;
            Fp[i] = ((unsigned)yy < (unsigned)p.H && (unsigned)xx < (unsigned)p.W) ? fr[(size_t)yy * p.W + xx] : 0.f;   // layer 1's padding
        }
        __syncthreads();
    }
    gload_a(chunk_begin * C::KC, ra);
    gload_b(chunk_begin, 0, rb);
    lds_store_a(ra);
    lds_store_b(rb);
    __syncthreads();
    unsigned long long clk1 = 0;
    if (p.clk) clk1 = clock64();

    // One stage = all resident taps of one 16-channel chunk from LDS.  GRP (stage within the chunk) and HAS_NEXT
    // are compile-time: the loop below is peeled so that inside it the next stage's global loads and their LDS
    // stores are UNCONDITIONAL.  With the loads under one runtime `if` and the stores under another, hipcc's
    // (path-insensitive) s_waitcnt insertion assumed the previous stage's loads could still be pending at the loop
    // head and emitted vmcnt(0) right after the new A loads were issued - a full memory latency exposed per stage.
    auto stage = [&](const int chunk, auto grp_c, auto has_next_c, const int nchunk_override = -1) __attribute__((always_inline)) {
        constexpr int grp = decltype(grp_c)::value;
        constexpr bool has_next = decltype(has_next_c)::value;
        constexpr int ngrp = (grp + 1) % NG;
        constexpr bool next_a = has_next && (ngrp == 0 || C::PAR4);        // PAR4: every stage has its own A tile
        const int nchunk = nchunk_override >= 0 ? nchunk_override : chunk + (ngrp == 0 ? CPS : 0);
        if (DVG_ABLATE < 1 || DVG_ABLATE == 6 || DVG_ABLATE == 7) {        // 6: the weight tile stays what the prologue loaded, 7: the A tile
            if constexpr (next_a) { if (DVG_ABLATE != 7) gload_a(nchunk * C::KC, ra, C::PAR4 ? ngrp : 0); }
            if constexpr (has_next) { if (DVG_ABLATE != 6) gload_b(nchunk, ngrp, rb); }
        }

        // ---- all taps of this stage from LDS; fragments double-buffered across taps ----
        // a lane's fragment of a tap: NP 16-byte pieces per 32-row tile - f32: its k-values 0..3 and 4..7 of the lane's
        // half of the chunk; bf16 triples: its eight k-values in each of the planes h, m, l
        f32x4 fa[2][MT][NP], fb[2][NT][NP];
        {
            const int ao = tap_lds(grp, 0);
#pragma unroll
            for (int j = 0; j < NP; ++j) {
#pragma unroll
                for (int mt = 0; mt < MT; ++mt)
                    fa[0][mt][j] = *reinterpret_cast<const f32x4*>(&As[a_base[mt] + ao + j * PSTEP]);
#pragma unroll
                for (int nt = 0; nt < NT; ++nt)
                    fb[0][nt][j] = *reinterpret_cast<const f32x4*>(&Bs[b_base[nt] + j * PSTEP]);
            }
        }
#pragma unroll
        for (int tt = 0; tt < GT; ++tt) {
            const int cur = tt & 1, nxt = cur ^ 1;
            // LEAN (the 64 x 64 wave tile): the next tap's planes are read when the current tap no longer needs the plane -
            // h up front into a second buffer, l after the two groups that use l, m after the three that use m - so that 16
            // instead of 48 VGPRs double-buffer the fragments (the whole set twice does not fit 256 registers)
            constexpr bool LEAN = X3 && (NT == 2 || (GEMM && MT == 2 && DVG_GEMM128_LEAN));
            auto read_plane = [&](int j) {
                const int ao = tap_lds(grp, tt + 1);
#pragma unroll
                for (int mt = 0; mt < MT; ++mt)
                    fa[nxt][mt][j] = *reinterpret_cast<const f32x4*>(&As[a_base[mt] + ao + j * PSTEP]);
#pragma unroll
                for (int nt = 0; nt < NT; ++nt)
                    fb[nxt][nt][j] = *reinterpret_cast<const f32x4*>(&Bs[b_base[nt] + (tt + 1) * 64 * LDB + j * PSTEP]);
            };
            if (LEAN && tt + 1 < GT) read_plane(0);
            if (!LEAN && tt + 1 < GT) {
                const int ao = tap_lds(grp, tt + 1);
#pragma unroll
                for (int j = 0; j < NP; ++j) {
#pragma unroll
                    for (int mt = 0; mt < MT; ++mt)
                        fa[nxt][mt][j] = *reinterpret_cast<const f32x4*>(&As[a_base[mt] + ao + j * PSTEP]);
#pragma unroll
                    for (int nt = 0; nt < NT; ++nt)
                        fb[nxt][nt][j] = *reinterpret_cast<const f32x4*>(&Bs[b_base[nt] + (tt + 1) * 64 * LDB + j * PSTEP]);
                }
            }
            constexpr bool overlap_writes = DVG_WRITE_OVERLAP && has_next;
            if (overlap_writes && tt == GT - 1) {
                // The last tap's fragments are in registers: every wave is done reading this stage's tiles after
                // this barrier, and the next stage's ds_writes interleave with the last tap's MFMAs instead of
                // forming an MFMA-less pass between two barriers.
                if (DVG_ABLATE < 3 || (DVG_ABLATE >= 4 && DVG_ABLATE != 9)) __syncthreads();
                if (DVG_ABLATE < 2 || (DVG_ABLATE >= 4 && DVG_ABLATE < 8)) {      // 4: no B stores, 5: no A stores (timing only)
                    if constexpr (next_a) { if (DVG_ABLATE != 5 && DVG_ABLATE != 7) lds_store_a(ra, C::PAR4 ? ngrp : 0); }
                    if (DVG_ABLATE != 4 && DVG_ABLATE != 6) lds_store_b(rb);
                }
            }
            if constexpr (X3) {
                // six bf16 MFMAs per 32 x 32 tile and K = 16 slab, small terms first: (l,h) (m,m) (h,l) (m,h) (h,m) (h,h)
                auto mm = [&](int pa, int pb) {
#pragma unroll
                    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                        for (int nt = 0; nt < NT; ++nt)
                            acc[mt * NT + nt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(
                                __builtin_bit_cast(bf16x8_t, fa[cur][mt][pa]), __builtin_bit_cast(bf16x8_t, fb[cur][nt][pb]),
                                acc[mt * NT + nt], 0, 0, 0);
                };
                // DVG_X3_TERMS (timing experiments only, WRONG results below 6): 3 keeps (m,h) (h,m) (h,h) - what a two-piece
                // operand split would issue - to measure how the rollout rate follows the MFMA count (notes r05 §8)
                if constexpr (LEAN) {
                    if (DVG_X3_TERMS >= 6) mm(2, 0);
                    if (DVG_X3_TERMS >= 6) mm(0, 2);
                    if (tt + 1 < GT) read_plane(2);
                    if (DVG_X3_TERMS >= 6) mm(1, 1);
                    mm(1, 0);
                    mm(0, 1);
                    if (tt + 1 < GT) read_plane(1);
                    mm(0, 0);
                } else {
                    if (DVG_X3_TERMS >= 6) mm(2, 0);
                    if (DVG_X3_TERMS >= 6) mm(1, 1);
                    if (DVG_X3_TERMS >= 6) mm(0, 2);
                    mm(1, 0);
                    mm(0, 1);
                    mm(0, 0);
                }
            } else {
#pragma unroll
                for (int j = 0; j < 2; ++j)
#pragma unroll
                    for (int e = 0; e < 4; ++e)
#pragma unroll
                        for (int mt = 0; mt < MT; ++mt)
                            acc[mt] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[cur][mt][j][e], fb[cur][0][j][e], acc[mt], 0, 0, 0);
            }
            // Pin the software pipeline: hipcc otherwise sinks the next tap's ds_reads down to their first use
            // (ds_read x3 -> s_waitcnt -> mfma x8), exposing the LDS latency every 8 MFMAs.  One ds_read_b128 per two
            // MFMAs, issued a full tap (16 / 8 MFMAs) ahead of its consumer.
            constexpr int NREAD = NP * (MT + NT), NMFMA = (X3 ? DVG_X3_TERMS : 8) * MT * NT, MPR = NMFMA >= 2 * NREAD ? 2 : 1;
            // next stage's global loads: two per tap behind the first taps' MFMAs.  Left free, hipcc sinks them to
            // the end of the stage (latency exposed at their ds_write); all at the top they delay the first MFMAs.
            constexpr int NVMEM = (next_a ? (FIRST ? 11 : NLA) : 0) + (has_next ? NLB : 0);
            constexpr int POLICY = DVG_VMEM_POLICY;
            // policy 4 (default): spread over the taps that are followed by another tap, starting at tap 2 in the 9-tap
            // mode and at tap 0 in the 4- / 8-tap modes, as many per tap as it takes to place ALL of them (with the
            // 9-tap constants, the 4-tap transposed mode pinned 2 of its 6-7 loads and the rest sank to the stage's end)
            constexpr int VT0 = (POLICY == 4) ? (GT >= 9 ? 2 : 0) : 0;
            constexpr int SLOTS = (GT - 1 - VT0) > 0 ? (GT - 1 - VT0) : 1;
            constexpr int VNEED = (NVMEM + SLOTS - 1) / SLOTS;
            constexpr int VPT = (POLICY == 3) ? 3 : (POLICY == 4 ? (VNEED > 2 ? VNEED : 2) : 2);
            constexpr int VTAPS = (NVMEM + VPT - 1) / VPT;
            if (tt == 0) {
                if (POLICY == 1 && NVMEM > 0) __builtin_amdgcn_sched_group_barrier(0x020, NVMEM, 0);
                __builtin_amdgcn_sched_group_barrier(0x100, NREAD, 0);  // tap 0's own fragments
            }
            if (LEAN && tt + 1 < GT) {
                constexpr int G = MT * NT, R = MT + NT;
#pragma unroll
                for (int r = 0; r < R; ++r) {      // next h planes behind the first MFMAs
                    __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
                    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                }
                __builtin_amdgcn_sched_group_barrier(0x008, 2 * G - R, 0);
#pragma unroll
                for (int r = 0; r < R; ++r) {      // l planes are free: next l
                    __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
                    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                }
                __builtin_amdgcn_sched_group_barrier(0x008, 3 * G - R, 0);
#pragma unroll
                for (int r = 0; r < R; ++r) {      // m planes are free: next m
                    __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
                    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                }
            } else if (tt + 1 < GT) {
#pragma unroll
                for (int r = 0; r < NREAD; ++r) {
                    __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
                    __builtin_amdgcn_sched_group_barrier(0x008, MPR, 0);
                }
                if (NMFMA > MPR * NREAD) __builtin_amdgcn_sched_group_barrier(0x008, NMFMA - MPR * NREAD, 0);
            } else if (overlap_writes) {
                constexpr int NW = (next_a ? NLA * (X3 ? 3 : 1) : 0) + NLB;
#pragma unroll
                for (int r = 0; r < NW; ++r) {
                    if (r < NMFMA) __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                    __builtin_amdgcn_sched_group_barrier(0x200, 1, 0);
                }
                if (NMFMA > NW) __builtin_amdgcn_sched_group_barrier(0x008, NMFMA - NW, 0);
            } else {
                __builtin_amdgcn_sched_group_barrier(0x008, NMFMA, 0);
            }
            if (POLICY >= 2 && tt >= VT0 && tt < VT0 + VTAPS && tt + 1 < GT) __builtin_amdgcn_sched_group_barrier(0x020, VPT, 0);
        }

        if constexpr (has_next) {
            if (!DVG_WRITE_OVERLAP) {
                __syncthreads();  // every wave has finished reading this stage's tiles
                if constexpr (next_a) lds_store_a(ra, C::PAR4 ? ngrp : 0);
                lds_store_b(rb);
            }
            if (DVG_ABLATE < 3 || (DVG_ABLATE >= 4 && DVG_ABLATE != 9)) __syncthreads();
        }
    };
    using std::integral_constant;
    // Progress-based wave priority.  The MFMA issue arbiter breaks ties by wave age, so of the two workgroups that share a
    // CU the one dispatched first won every contested slot: it finished ~7 us before its partner (27 vs 34 us on the B = 64
    // dcgan layers, tools/diag_clocks_dcgan.py DIAG_PAIRS=1), which then ran its tail alone - one MFMA-issuing wave per
    // SIMD, which cannot saturate the f32 pipe.  With priority 3 - (stage & 3) a workgroup that is one stage AHEAD of its
    // neighbour has the lower priority on 3 of every 4 stages: the pair stays within a stage of each other and both
    // finish together.  Needs no knowledge of who the neighbour is.
#if DVG_STAGE_PRIO
    const bool prio_on = p.stage_prio != 0;
    auto set_prio = [&](int st) {
        if (!prio_on) return;
        switch (st & 3) {
            case 0: __builtin_amdgcn_s_setprio(3); break;
            case 1: __builtin_amdgcn_s_setprio(2); break;
            case 2: __builtin_amdgcn_s_setprio(1); break;
            default: __builtin_amdgcn_s_setprio(0); break;
        }
    };
#else
    auto set_prio = [](int) {};
#endif
    int chunk = chunk_begin, st = 0;
    if constexpr (GEMM) {
        // gemm_ni images back to back in ONE software pipeline: the loads of image i+1's first stage are in flight while
        // image i's last stage computes, and there is one prologue per workgroup instead of one per image - the Winograd
        // GEMMs have K = 128 ... 512, i.e. only 2 ... 8 stages per image.
        const int spi = (chunk_end - chunk_begin) / CPS;
        const long a_img = (long)p.H * p.W * p.C1;
        // Store addressing of the products: ONE lane-dependent 32-bit offset (the lane's pixel part x Cout + its column) and a
        // wave-uniform base per accumulator register, so that the stores are `global_store saddr + voffset` and no 64-bit
        // per-register address stays live across the stage loop (with a divergent base pointer hipcc kept 2 x 16 MT VGPRs of
        // addresses alive through the kernel).  pixel(m) = (y0 + m / TW) W + x0 + m % TW with m = mbase + (reg & 3) + 8 (reg >> 2)
        // + 4 hh: the (reg, mt, wm) part is uniform, 4 hh is the lane's.
        const unsigned lane_off = (unsigned)(4 * hh * p.Cout + l31);
        float* const ybase = p.y + (size_t)n0 * p.H * p.W * p.Cout + nb0 + __builtin_amdgcn_readfirstlane(wn) * 32 * NT;
        const int wm_u = __builtin_amdgcn_readfirstlane(wm);
        for (int img = 0; img < p.gemm_ni; ++img) {
            for (int sg = 0; sg < spi; ++sg) {
                const bool wrap = sg == spi - 1, last = wrap && img == p.gemm_ni - 1;
                ld_a_off = (long)(n0 + img + (wrap ? 1 : 0)) * a_img;
                ld_w = wbase + (size_t)(img + (wrap ? 1 : 0)) * p.w_image_stride;
                const int nchunk = wrap ? chunk_begin : chunk_begin + (sg + 1) * CPS;
                set_prio(st++);
                if (last) stage(chunk_begin + sg * CPS, integral_constant<int, 0>{}, integral_constant<bool, false>{});
                else stage(chunk_begin + sg * CPS, integral_constant<int, 0>{}, integral_constant<bool, true>{}, nchunk);
            }
            float* const yb = ybase + (size_t)img * p.H * p.W * p.Cout;      // raw products: M[n0 + img][pixel][co]
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) {
                const int mbase = wm_u * (C::BM / 2) + mt * 32;
#pragma unroll
                for (int nt = 0; nt < NT; ++nt)
#pragma unroll
                    for (int reg = 0; reg < 16; ++reg) {
                        const int mu = mbase + (reg & 3) + 8 * (reg >> 2);       // the uniform part of m (4 hh is in lane_off)
                        // (DVG_ABLATE 8 / 9, timing only: one product value per lane instead of the tile's 16)
                        if (DVG_ABLATE < 8 || reg == 0) {
                            float* const dst = (yb + (size_t)((y0 + mu / TW) * p.W + x0 + mu % TW) * p.Cout + nt * 32) + lane_off;
#if DVG_GEMM_NT_STORE      // A/B (r05): the products are written once and read once, by the next kernel
                            __builtin_nontemporal_store(acc[mt * NT + nt][reg], dst);
#else
                            *dst = acc[mt * NT + nt][reg];
#endif
                        }
                        acc[mt * NT + nt][reg] = 0.f;
                    }
            }
        }
    } else {
    // the NG stages of a chunk, the last one of the last chunk without a successor
    auto chunk_stages = [&](const int ch, auto last_c) __attribute__((always_inline)) {
        constexpr bool last = decltype(last_c)::value;
        set_prio(st++);
        stage(ch, integral_constant<int, 0>{}, integral_constant<bool, !(last && NG == 1)>{});
        if constexpr (NG >= 2) {
            set_prio(st++);
            stage(ch, integral_constant<int, 1>{}, integral_constant<bool, !(last && NG == 2)>{});
        }
        if constexpr (NG >= 4) {
            set_prio(st++);
            stage(ch, integral_constant<int, 2>{}, integral_constant<bool, true>{});
            set_prio(st++);
            stage(ch, integral_constant<int, 3>{}, integral_constant<bool, !last>{});
        }
        static_assert(NG == 1 || NG == 2 || NG == 4, "stages per chunk");
    };
    for (; chunk + CPS < chunk_end; chunk += CPS) chunk_stages(chunk, integral_constant<bool, false>{});
    chunk_stages(chunk, integral_constant<bool, true>{});
    }
#if DVG_STAGE_PRIO
    __builtin_amdgcn_s_setprio(0);
#endif
    unsigned long long clk_loop = 0;
    if (p.clk) clk_loop = clock64();
    auto clk_exit = [&]() {
        if (p.clk && threadIdx.x == 0 && blockIdx.x < p.clk_cap) {
            unsigned long long* d = p.clk + (size_t)blockIdx.x * 8;
            d[0] = clk0; d[1] = clk1; d[2] = clk_loop; d[3] = clock64(); d[4] = wclk0; d[5] = wall_clock64();
            unsigned xcc;
            asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
            unsigned hwid;
            asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hwid));
            d[6] = xcc; d[7] = hwid;
        }
    };

    if constexpr (GEMM) {     // every image's products were stored inside the loop
        clk_exit();
        return;
    }
    // ---- epilogue (identical math to v1) --------------------------------------------------------------
    int Ho, Wo;
    if (MODE == M2_CONV3 || MODE == M2_GEMM) { Ho = p.H; Wo = p.W; }
    else if (MODE == M2_CONV4S2) { Ho = p.H >> 1; Wo = p.W >> 1; }
    else { Ho = p.H * 2; Wo = p.W * 2; }
    const int c = nb0 + wn * 32 + l31;
    if (C::CAN_SPLIT && p.splitk > 1) {
        // Split K: every split writes its raw partial tile to `ws`; splitk_finish_kernel (a second launch) sums them in split
        // order and does scale / activation / pool / statistics.  (r04 also built the combination INSIDE this kernel - tickets,
        // device-scope relaxed atomics on the partial tiles, the last starter sums: bit-equal, 11 % less traffic, 2.5 % slower on
        // the dcgan_64 rollout because the finisher's wait sits on every tile's critical path; removed in r05, numbers in
        // docs/DESIGN_NOTES_r04.md.)
        const size_t slab = (size_t)p.N * Ho * Wo * p.Cout;
        float* const wsp = p.ws + (size_t)split * slab;
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
            for (int reg = 0; reg < 16; ++reg) {
                const int m = wm * (C::BM / 2) + mt * 32 + (C::row_c(reg) ^ (hh * 4 * C::PK));
                const int tii = m / (TH * TW), r = m % (TH * TW);
                const int ty = r / TW, tx = r % TW;
                const int n = n0 + tii;
                int oy, ox;
                if (MODE == M2_CONVT4S2) { oy = 2 * (y0 + ty) + py; ox = 2 * (x0 + tx) + px; }
                else { oy = y0 + ty; ox = x0 + tx; }
                if (n < p.N) wsp[(((size_t)n * Ho + oy) * Wo + ox) * p.Cout + c] = acc[mt][reg];
            }
        clk_exit();
        return;
    }
    const float sc = p.scale ? p.scale[c] : 1.f, sf = p.shift ? p.shift[c] : 0.f;
    float s1 = 0.f, s2 = 0.f;
    // The activation is dispatched ONCE (it used to be a runtime switch per value, tanh/exp code inlined 32 times),
    // and addresses are a per-workgroup 64-bit base plus 32-bit per-value offsets.
    float* const yb = p.y + (size_t)n0 * Ho * Wo * p.Cout + c;
    const bool want_y = !FIRST || n0 >= p.y_from;      // workgroup-uniform (FIRST tiles hold one image)
    float* const pb = p.y_pool ? p.y_pool + (size_t)n0 * (Ho >> 1) * (Wo >> 1) * p.Cout + c : nullptr;
    const float* const ab = p.addend ? p.addend + (size_t)n0 * Ho * Wo * p.Cout + c : nullptr;
    // shared addend blocks (add_map): per image of the tile, the distance (floats) from "addend image n" to the image its
    // group really adds.  A 4-image tile (4x4 maps) may straddle two groups, hence per image.
    long add_shift[TI];
#pragma unroll
    for (int t_ = 0; t_ < TI; ++t_) {
        const int n_ = min(n0 + t_, p.N - 1);
        add_shift[t_] = (ab != nullptr && p.add_map != nullptr)
                            ? (long)(addend_image(p.add_map, p.add_B, n_) - n_) * Ho * Wo * p.Cout : 0;
    }
    auto shift_of = [&](int tii) -> long {
        if (TI == 1) return add_shift[0];
        long v = add_shift[0];
#pragma unroll
        for (int t_ = 1; t_ < TI; ++t_) v = tii == t_ ? add_shift[t_] : v;
        return v;
    };
    const int hrow = hh * 4 * C::PK;     // C-layout row of (reg, lane half): C::row_c(reg) ^ hrow  (Cfg2: LAYOUT)
    auto epilogue = [&](auto act_c) {
        constexpr int ACT = decltype(act_c)::value;  // -1: generic (runtime p.act)
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
            float v[16];
            const int mbase = wm * (C::BM / 2) + mt * 32;
            const int ti0 = mbase / (TH * TW);
            if (ab != nullptr) {
                // hoisted skip half of a decoder conv (fused.py): raw partial sums of the loop-invariant input
                float av[16];
#pragma unroll
                for (int reg = 0; reg < 16; ++reg) {
                    const int m = mbase + (C::row_c(reg) ^ hrow);
                    const int tii = (TH * TW >= 32) ? ti0 : m / (TH * TW);
                    const int r = m % (TH * TW);
                    const int ty = r / TW, tx = r % TW;
                    int oy, ox;
                    if (MODE == M2_CONVT4S2) { oy = 2 * (y0 + ty) + py; ox = 2 * (x0 + tx) + px; }
                    else { oy = y0 + ty; ox = x0 + tx; }
                    av[reg] = (TI == 1 || n0 + tii < p.N) ? ab[(long)((tii * Ho + oy) * Wo + ox) * p.Cout + shift_of(tii)] : 0.f;
                }
#pragma unroll
                for (int reg = 0; reg < 16; ++reg) v[reg] = (acc[mt][reg] + av[reg]) * sc + sf;
            } else {
#pragma unroll
                for (int reg = 0; reg < 16; ++reg) v[reg] = acc[mt][reg] * sc + sf;
            }
#pragma unroll
            for (int reg = 0; reg < 16; ++reg) {
                const int row = C::row_c(reg) ^ hrow;
                const int m = mbase + row;
                const int tii = (TH * TW >= 32) ? ti0 : m / (TH * TW);
                const int r = m % (TH * TW);
                const int ty = r / TW, tx = r % TW;
                if (TI == 1 || n0 + tii < p.N) {
                    s1 += v[reg];
                    s2 += v[reg] * v[reg];
                    float o;
                    if constexpr (ACT == DVG_ACT_LRELU) o = v[reg] > 0.f ? v[reg] : v[reg] * p.slope;
                    else if constexpr (ACT == DVG_ACT_NONE) o = v[reg];
                    else o = apply_act(v[reg], p.act, p.slope);
                    v[reg] = o;
                    int oy, ox;
                    if (MODE == M2_CONVT4S2) { oy = 2 * (y0 + ty) + py; ox = 2 * (x0 + tx) + px; }
                    else { oy = y0 + ty; ox = x0 + tx; }
                    if (!FIRST || want_y) yb[((tii * Ho + oy) * Wo + ox) * p.Cout] = o;
                }
            }
            if (MODE == M2_CONV3 && (TW == 16 || TW == 8)) {
                if (pb != nullptr) {
                    // a 2 x 2 window = two x-neighbours (reg, reg + 1) of two vertically adjacent rows in ONE lane.  r05
                    // layout: the 16-wide tile has tile row 0 in register blocks 0, 1 and row 1 in blocks 2, 3 (partner: + 8),
                    // the 8-wide one alternates (partner: + 4).  r06 layout (Cfg2::LAYOUT): the vertical partner of block
                    // 0 / 2 is block 1 / 3 in both tiles and both lane halves (which of the two is the upper row depends on
                    // the lane half; the window's maximum and its pooled coordinates do not).
                    constexpr int RY = (TW == 16 && !C::LAYOUT) ? 8 : 4;
#pragma unroll
                    for (int reg = 0; reg < 16; ++reg) {
                        const bool ty_even = (TW == 16 && !C::LAYOUT) ? ((reg >> 2) < 2) : (((reg >> 2) & 1) == 0);
                        if ((reg & 1) == 0 && ty_even) {
                            const int row = C::row_c(reg) ^ hrow;
                            const int r = (mbase + row) % (TH * TW);
                            const int ty = r / TW, tx = r % TW;
                            const float mx = fmaxf(fmaxf(v[reg], v[reg + 1]), fmaxf(v[reg + RY], v[reg + RY + 1]));
                            if (TI == 1 || n0 + ti0 < p.N)
                                pb[((ti0 * (Ho >> 1) + ((y0 + ty) >> 1)) * (Wo >> 1) + ((x0 + tx) >> 1)) * p.Cout] = mx;
                        }
                    }
                }
            }
        }
    };
    if (p.act == DVG_ACT_LRELU) epilogue(std::integral_constant<int, DVG_ACT_LRELU>{});
    else if (p.act == DVG_ACT_NONE) epilogue(std::integral_constant<int, DVG_ACT_NONE>{});
    else epilogue(std::integral_constant<int, -1>{});
    if (p.stats != nullptr) {
        float* red = smem;
        s1 += __shfl_xor(s1, 32);
        s2 += __shfl_xor(s2, 32);
        __syncthreads();
        if (wm == 1 && hh == 0) {
            red[wn * 64 + l31] = s1;
            red[wn * 64 + 32 + l31] = s2;
        }
        __syncthreads();
        if (wm == 0 && hh == 0) {
            const unsigned rowid = (MODE == M2_CONVT4S2 ? tile_id * 4 + par : tile_id);
            float* dst = p.stats + (size_t)rowid * 2 * p.Cout;
            dst[c] = s1 + red[wn * 64 + l31];
            dst[p.Cout + c] = s2 + red[wn * 64 + 32 + l31];
        }
    }
    clk_exit();
}

// out = act((sum_s ws[s]) * scale + shift) (+ 2x2 max-pool, + per-channel sum / sum of squares of the pre-activation)
// One thread = one pixel (or one 2x2 window when pooling) x 4 channels; blockDim = 256 = TC x TP as in the BN
// backward reduction; one partial statistics row per workgroup.
template <bool POOL>
__global__ __launch_bounds__(256) void splitk_finish_kernel(const float* __restrict__ ws, int S,
                                                            const float* __restrict__ scale,
                                                            const float* __restrict__ shift, float* __restrict__ y,
                                                            float* __restrict__ y_pool, float* __restrict__ stats, int N,
                                                            int H, int W, int C, int act, float slope,
                                                            int units_per_block, const float* __restrict__ addend,
                                                            const int* __restrict__ add_map, int add_B) {
    __shared__ float red[2 * 256 * 4];
    const int C4 = C >> 2;
    const int TC = C4 < 256 ? C4 : 256, TP = 256 / TC;
    const int tc = threadIdx.x % TC, tp = threadIdx.x / TC;
    const int Hu = POOL ? H >> 1 : H, Wu = POOL ? W >> 1 : W;
    const long units = (long)N * Hu * Wu, slab4 = (long)N * H * W * C4;
    const long u0 = (long)blockIdx.x * units_per_block, u1 = min(units, u0 + units_per_block);
    const int c4 = tc;
    const f32x4 sc = scale ? *reinterpret_cast<const f32x4*>(scale + c4 * 4) : f32x4{1.f, 1.f, 1.f, 1.f};
    const f32x4 sf = shift ? *reinterpret_cast<const f32x4*>(shift + c4 * 4) : f32x4{0.f, 0.f, 0.f, 0.f};
    f32x4 s1 = {0.f, 0.f, 0.f, 0.f}, s2 = {0.f, 0.f, 0.f, 0.f};
    for (long w_ = u0 + tp; w_ < u1; w_ += TP) {
        f32x4 mx;
#pragma unroll
        for (int q = 0; q < (POOL ? 4 : 1); ++q) {
            size_t off;
            if (POOL) {
                const int xp = w_ % Wu;
                long r = w_ / Wu;
                const int yp = r % Hu;
                const int n = r / Hu;
                off = (((size_t)n * H + 2 * yp + (q >> 1)) * W + 2 * xp + (q & 1)) * C4 + c4;
            } else {
                off = (size_t)w_ * C4 + c4;
            }
            f32x4 v = reinterpret_cast<const f32x4*>(ws)[off];
            {
                int s = 1;
                for (; s + 3 <= S; s += 3) {   // three slabs per trip: their loads issue together, fixed-order sum
                    f32x4 t[3];
#pragma unroll
                    for (int u = 0; u < 3; ++u) t[u] = reinterpret_cast<const f32x4*>(ws)[(size_t)(s + u) * slab4 + off];
#pragma unroll
                    for (int u = 0; u < 3; ++u)
#pragma unroll
                        for (int k = 0; k < 4; ++k) v[k] += t[u][k];
                }
                for (; s < S; ++s) {
                    const f32x4 t = reinterpret_cast<const f32x4*>(ws)[(size_t)s * slab4 + off];
#pragma unroll
                    for (int k = 0; k < 4; ++k) v[k] += t[k];
                }
            }
            if (addend != nullptr) {   // hoisted skip half: one more (loop-invariant) partial slab
                size_t aoff = off;
                if (add_map != nullptr) {      // time-batched decoder calls: the addend block this image's group shares
                    const long img4 = (long)H * W * C4;
                    const int n_ = (int)(off / img4);
                    aoff = off - (size_t)(n_ - addend_image(add_map, add_B, n_)) * img4;
                }
                const f32x4 t = reinterpret_cast<const f32x4*>(addend)[aoff];
#pragma unroll
                for (int k = 0; k < 4; ++k) v[k] += t[k];
            }
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const float u = v[k] * sc[k] + sf[k];
                s1[k] += u;
                s2[k] = fmaf(u, u, s2[k]);
                v[k] = apply_act(u, act, slope);
            }
            reinterpret_cast<f32x4*>(y)[off] = v;
            if (POOL) {
                if (q == 0) mx = v;
                else
#pragma unroll
                    for (int k = 0; k < 4; ++k) mx[k] = fmaxf(mx[k], v[k]);
            }
        }
        if (POOL) reinterpret_cast<f32x4*>(y_pool)[w_ * C4 + c4] = mx;
    }
    if (stats != nullptr) {
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            red[(tp * TC + tc) * 4 + k] = s1[k];
            red[1024 + (tp * TC + tc) * 4 + k] = s2[k];
        }
        __syncthreads();
        if (tp == 0) {
            float* dst = stats + (size_t)blockIdx.x * 2 * C;
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

static int finish_units_per_block(long units) {
    long upb = (units + 1023) / 1024;
    if (upb < 8) upb = 8;
    return (int)upb;
}

// Tile policy (r06).  0 = LATENCY: the tiles that make ONE launch alone on the chip fastest (finer tiles balance the CUs: the
// r05 chooser) - one chain of launches, GPtrigger_gen, training.  1 = ENERGY: several independent chains in flight keep the
// board at its power cap (1 365 W, 1.98 GHz), where a launch costs its energy, not its duration - the batched GEMM takes the
// 128 x 128 tile from DVG_GEMM_NT2 workgroups on, the halo modes the 8 x 16 tile from DVG_TILE16_MIN_WGS on (DESIGN.md 3.1e:
// +4.6 % vgg_64 / +1 % dcgan_64 rollouts in flight, one chain 3 % slower).  Process-global host state, read when a launch (or
// one of the dvg_conv_splitk_v2 / dvg_conv_stats_rows_v2 helpers) picks its tile; a captured hipGraph keeps what it captured.
static int g_tile_policy = 0;
extern "C" void dvg_set_tile_policy(int energy) { g_tile_policy = energy ? 1 : 0; }
extern "C" int dvg_tile_policy(void) { return g_tile_policy; }

// number of K splits for a v2 launch: only when the grid would leave CUs idle and K is deep enough
// (r05, stride-2 conv after the parity split: splitting further - 768 / 1 024 workgroup slots - or not at all measured -0.5 ... -3 % on
// the dcgan_64 rollout, in flight and on one chain: profiles/r05_ab_conv4s2.txt)
static int choose_splitk(long wgs, int nchunks) {
    if (wgs >= 384 || nchunks < 8) return 1;
    long s = 512 / wgs;
    if (s > 8) s = 8;
    if (s > nchunks / 4) s = nchunks / 4;   // at least 4 chunks (= 4 stages of 72-144 MFMAs) per split
    return s < 2 ? 1 : (int)s;
}

template <int MODE, int TI, int TH, int TW, int NT = 1, bool FIRST = false>
static int launch2(Igemm2Params p, int Hg, int Wg, float* ws, long ws_floats, hipStream_t stream) {
    using C = Cfg2<MODE, TI, TH, TW, NT, FIRST>;
    if (Hg % TH || Wg % TW || p.Cout % C::BN) return fail(DVG_ERR_SHAPE, "conv_igemm2: tile does not divide shape");
    p.tiles_y = Hg / TH;
    p.tiles_x = Wg / TW;
    p.tiles_n = (p.N + TI - 1) / TI;
    if (MODE == M2_GEMM) {
        // images per workgroup: ONE.  The kernel can run gemm_ni images back to back in one software pipeline (one prologue
        // per workgroup instead of one per image) and with 3 workgroups per CU (138 VGPRs) 2-4 images per workgroup measured
        // a few % faster on the K <= 256 shapes; at 4 per CU (127 VGPRs, __launch_bounds__ above) the extra wave hides the
        // prologues and the finer-grained workgroups balance the CUs better: 1 image is fastest or tied on every vgg_64
        // shape (tools/ab_gemm_ni.sh: e.g. 16x16 256->256 48.4 us vs 52.0 (2) / 59.0 (4); 8x8 512->256 31.7 vs 41.7 / 57.6).
        int ni = 1;
#if DVG_BF16X3
        // bf16 triples: the 64-row tile runs three workgroups per CU (768 resident), and with the matrix work 2.7 x shorter a
        // workgroup's prologue weighs more.  When the launch is a whole number r <= 6 of residency rounds, ONE round of
        // workgroups that run r images back to back in one pipeline is faster (16x16 256->256: 39.6 -> 36.1 us,
        // 32x32 128->128: 43.2 -> 40.4 us); otherwise the finer grain balances better (8x8 512->512, 1.5 rounds: 38.6 vs 39.5).
        if (TW == 8 && NT == 1) {
            const long w1 = (long)p.tiles_y * p.tiles_x * p.N * (p.Cout / C::BN), slots = 256L * DVG_GEMM_WGS_PER_CU;
            if (w1 % slots == 0 && w1 / slots >= 2 && w1 / slots <= 6 && p.N % (w1 / slots) == 0) ni = (int)(w1 / slots);
        }
#endif
        p.gemm_ni = ni;
        p.tiles_n = p.N / ni;
    }
    p.nblk_n = p.Cout / C::BN;
    p.clk = g_clk;
    p.clk_cap = g_clk_cap;
    const long wgs = (long)p.tiles_y * p.tiles_x * p.tiles_n * p.nblk_n * (MODE == M2_CONVT4S2 ? 4 : 1);
    {
        // Cout blocks per XCD group: minimise (weight slabs + input tiles) one XCD's L2 fetches for its share of the
        // launch; a must divide nblk_n, and a group must hold at least one XCD's share of pixel tiles
        const long tiles = (long)p.tiles_y * p.tiles_x * p.tiles_n;
        const double per_xcd = (double)tiles * p.nblk_n / 8.0;
        const double wb = (double)C::NTAPS * (p.C1 + p.C2) * C::BN * 4, ib = (double)C::BM * (p.C1 + p.C2) * 4;
        int best = p.nblk_n;
        double cost = 1e300;
        for (int a = 1; a <= p.nblk_n; ++a) {
            if (p.nblk_n % a) continue;
            const double share_tiles = per_xcd / a;
            if (share_tiles > (double)tiles) continue;   // more tiles than exist: the range would span several groups
            const double c = a * wb + (share_tiles < 1.0 ? 1.0 : share_tiles) * ib;
            if (c < cost) { cost = c; best = a; }
        }
        p.nb_group = best;
    }
    const int nchunks = (p.C1 + p.C2) / C::KC;
    int Ho, Wo;
    if (MODE == M2_CONV3 || MODE == M2_GEMM) { Ho = p.H; Wo = p.W; }
    else if (MODE == M2_CONV4S2) { Ho = p.H >> 1; Wo = p.W >> 1; }
    else { Ho = p.H * 2; Wo = p.W * 2; }
    const long out_floats = (long)p.N * Ho * Wo * p.Cout;
    int S = (ws != nullptr && C::CAN_SPLIT) ? choose_splitk(wgs, nchunks) : 1;
    if (S > 1 && (long)S * out_floats > ws_floats) return fail(DVG_ERR_SHAPE, "conv_igemm2: split-K workspace too small");
    p.cps = (nchunks + S - 1) / S;
    p.cps = (p.cps + C::CHUNKS_PER_STAGE - 1) / C::CHUNKS_PER_STAGE * C::CHUNKS_PER_STAGE;   // whole stages per split
    S = (nchunks + p.cps - 1) / p.cps;  // no empty split: the kernel's peeled stage loop needs >= 1 chunk per workgroup
    p.splitk = S;
    p.ws = ws;
    float* y_pool = p.y_pool;
    float* stats = p.stats;
    const unsigned grid = (unsigned)(wgs * S);
    // Stage priority (see the stage loop): measured +2..5 % wherever co-resident workgroups run in lockstep (single-round
    // launches of every mode) and on all 9-tap layers (144 MFMAs per stage); on multi-round launches of the short-stage
    // modes (4 / 8 taps, 64 MFMAs per stage) a starved workgroup's next-stage loads issue late: -3..-18 % -> off there.
    p.stage_prio = (MODE == M2_CONV3 || grid <= 3 * 256) ? 1 : 0;
    static bool attr_set = false;
    if (!attr_set) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_igemm2_kernel<MODE, TI, TH, TW, NT, FIRST>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, C::LDS_BYTES);
        if (e != hipSuccess) return fail(DVG_ERR_HIP, "hipFuncSetAttribute: %s", hipGetErrorString(e));
        attr_set = true;
    }
    hipLaunchKernelGGL((conv_igemm2_kernel<MODE, TI, TH, TW, NT, FIRST>), dim3(grid), dim3(256), C::LDS_BYTES, stream, p);
    if (int e = check_launch("conv_igemm2")) return e;
    if (S > 1) {
        const bool pool = y_pool != nullptr;
        const long units = pool ? (long)p.N * (Ho / 2) * (Wo / 2) : (long)p.N * Ho * Wo;
        const int upb = finish_units_per_block(units);
        const unsigned fgrid = (unsigned)((units + upb - 1) / upb);
        if (pool)
            hipLaunchKernelGGL((splitk_finish_kernel<true>), dim3(fgrid), dim3(256), 0, stream, ws, S, p.scale, p.shift, p.y,
                               y_pool, stats, p.N, Ho, Wo, p.Cout, p.act, p.slope, upb, p.addend, p.add_map, p.add_B);
        else
            hipLaunchKernelGGL((splitk_finish_kernel<false>), dim3(fgrid), dim3(256), 0, stream, ws, S, p.scale, p.shift, p.y,
                               y_pool, stats, p.N, Ho, Wo, p.Cout, p.act, p.slope, upb, p.addend, p.add_map, p.add_B);
        return check_launch("splitk_finish");
    }
    return DVG_OK;
}


// spatial tile on the grid the tiles cover; 8x16 unless the map is 8 wide / that leaves < 2 workgroups per CU
static int tile2(int mode, int N, int Hg, int Wg, int Cout, int* ti, int* th, int* tw) {
    const int par = mode == M2_CONVT4S2 ? 4 : 1;
    if (Hg == 4 && Wg == 4) { *ti = 4; *th = 4; *tw = 4; return 0; }
    if (Hg % 8 || Wg % 8) return -1;
    *ti = 1; *th = 8; *tw = 8;
    if (Wg % 16 == 0) {
        const long wgs = (long)N * (Hg / 8) * (Wg / 16) * (Cout / 64) * par;
        if (wgs >= (g_tile_policy ? DVG_TILE16_MIN_WGS : 512)) *tw = 16;
        if (mode == M2_GEMM && *tw == 16) {
            // batched GEMM.  The 64-row tile (K = 64 per stage, 3 workgroups per CU) is faster on every launch of a B = 64 step
            // (r04 same-box: 365 us per pass against 381), the 128-row tile (K = 32, LEAN, 3 per CU: twice the weight-fragment
            // reuse) from about four residency rounds on (the conditioning batch: 2077 us per pass against 2267)
            // (f32 MFMA build: the 64-row tile at 4 per CU throughout, as measured in r03)
            if (!DVG_BF16X3 || wgs < DVG_GEMM128_MIN_WGS) *tw = 8;
        }
    }
    return 0;
}

__global__ void pack_k16_kernel(const float* __restrict__ src, float* __restrict__ dst, int cout, int cin, int kh,
                                int kw, int transposed) {
    // dst[chunk][co / 64][slot][co % 64][row of 16 k-values]  <-  conv: w[co][ci][a][b]   convT: w[ci][co][KH-1-a][KW-1-b]
    // slot = the tap t = a * kw + b, except for 4 x 4 kernels (the stride-2 modes).  Transposed conv: the four taps an output
    // parity (py, px) uses are consecutive, slot = (py * 2 + px) * 4 + tt with tap row 2 + py - 2 (tt >> 1) and tap column
    // 2 + px - 2 (tt & 1).  Stride-2 conv: the four taps of an INPUT parity are consecutive, slot = ((a & 1) * 2 + (b & 1)) * 4
    // + (a >> 1) * 2 + (b >> 1).  So a stage's weight tile is contiguous in every mode
    // thread = (chunk, tap, co, pair of k): two adjacent k-values, one 4-byte store per plane (winograd.hip, wrow_owner_note)
    const long total = (long)cout * cin * kh * kw / 2;
    const int nblk = cout >> 6, taps = kh * kw;
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int k = (int)(i & 7) * 2;
        long r = i >> 3;
        const int co = r % cout; r /= cout;
        const int t = r % taps;
        const int chunk = r / taps;
        const int ci = chunk * 16 + k, a = t / kw, b = t % kw;
        const long j = transposed ? ((((long)ci * cout + co) * kh + (kh - 1 - a)) * kw + (kw - 1 - b))
                                  : ((((long)co * cin + ci) * kh + a) * kw + b);
        const long j1 = j + (transposed ? (long)cout * kh * kw : (long)kh * kw);        // the same tap of input channel ci + 1
        int slot = t;
        if (transposed && kh == 4 && kw == 4) {
            const int py = a & 1, px = b & 1;
            slot = (py * 2 + px) * 4 + ((2 + py - a) >> 1) * 2 + ((2 + px - b) >> 1);
        } else if (kh == 4 && kw == 4) {
            // the stride-2 conv (PAR4): the four taps of an input parity (a & 1, b & 1) are consecutive
            slot = ((a & 1) * 2 + (b & 1)) * 4 + (a >> 1) * 2 + (b >> 1);
        }
        wrow_store_pair(dst, (((size_t)chunk * nblk + (co >> 6)) * taps + slot) * 64 + (co & 63), co & 63, k, src[j], src[j1]);
    }
    wrow_drain();
}

}  // namespace dvg

using namespace dvg;

// 0: the native f32 MFMA; 1: fp32 operands as exact bf16 triples on the bf16 MFMA (dvg_common.h)
extern "C" int dvg_mfma_mode(void) { return DVG_BF16X3 ? 1 : 0; }
// floats per packed weight row (16 k-values of one output channel): what dvg_pack_conv_weight_k16 / dvg_winograd_weight write
extern "C" int dvg_packed_row_floats(void) { return DVG_WROW; }

extern "C" void dvg_debug_set_clockbuf(void* buf, unsigned records) {   // records of 8 x u64, one per workgroup
    g_clk = (unsigned long long*)buf;
    g_clk_cap = buf ? records : 0;
}

extern "C" int dvg_pack_conv_weight_k16(const float* w, float* w_packed, int cout, int cin, int kh, int kw,
                                        int transposed, void* stream) {
    DVG_REQUIRE(w && w_packed, DVG_ERR_NULL, "dvg_pack_conv_weight_k16: NULL pointer");
    DVG_REQUIRE(cout > 0 && cout % 64 == 0 && cin > 0 && cin % 16 == 0 && kh > 0 && kw > 0, DVG_ERR_SHAPE,
                "dvg_pack_conv_weight_k16: Cin must be a multiple of 16, Cout of 64");
    const long total = (long)cout * cin * kh * kw / 2;      // threads: one per pair of k-values
    long g = (total + 255) / 256;
    if (g > 4096) g = 4096;
    hipLaunchKernelGGL(pack_k16_kernel, dim3((unsigned)g), dim3(256), 0, (hipStream_t)stream, w, w_packed, cout, cin, kh,
                       kw, transposed);
    return check_launch("dvg_pack_conv_weight_k16");
}

static long v2_wgs(int mode, int N, int Hg, int Wg, int Cout, int ti, int th, int tw) {
    return (long)((N + ti - 1) / ti) * (Hg / th) * (Wg / tw) * (Cout / 64) * (mode == M2_CONVT4S2 ? 4 : 1);
}

// K splits of a launch with this tile: the 16-wide tiles never split (Cfg2::CAN_SPLIT: their kernels do not carry the split-K
// epilogue).  Until r05 that was implied by their threshold (512 workgroups > choose_splitk's 384); with DVG_TILE16_MIN_WGS = 256
// it has to be said: a host side that expected a split where the kernel made none read per-tile statistics rows as finish-block
// rows (train-mode BatchNorm statistics off by 16 % at B = 16: caught by tests/test_gpu_backward.py).
static int v2_splits(int mode, int N, int Hg, int Wg, int Cout, int Cin, int ti, int th, int tw) {
    if (tw == 16) return 1;
    return choose_splitk(v2_wgs(mode, N, Hg, Wg, Cout, ti, th, tw), Cin / 16);
}

// K splits the v2 launch of this shape will use when a workspace is supplied (1 = no split)
extern "C" int dvg_conv_splitk_v2(int mode, int N, int H, int W, int Cin, int Cout) {
    int Hg = H, Wg = W;
    if (mode == M2_CONV4S2) { Hg = H / 2; Wg = W / 2; }
    int ti, th, tw;
    if (Cin % 16 || Cout % 64 || tile2(mode, N, Hg, Wg, Cout, &ti, &th, &tw)) return -1;
    return v2_splits(mode, N, Hg, Wg, Cout, Cin, ti, th, tw);
}

extern "C" int dvg_conv_stats_rows_v2(int mode, int N, int H, int W, int Cin, int Cout, int pool, int with_workspace) {
    int Hg = H, Wg = W;
    if (mode == M2_CONV4S2) { Hg = H / 2; Wg = W / 2; }
    int ti, th, tw;
    if (tile2(mode, N, Hg, Wg, Cout, &ti, &th, &tw)) return -1;
    if (with_workspace && v2_splits(mode, N, Hg, Wg, Cout, Cin, ti, th, tw) > 1) {
        int Ho = H, Wo = W;
        if (mode == M2_CONV4S2) { Ho = H / 2; Wo = W / 2; }
        if (mode == M2_CONVT4S2) { Ho = 2 * H; Wo = 2 * W; }
        const long units = pool ? (long)N * (Ho / 2) * (Wo / 2) : (long)N * Ho * Wo;
        const int upb = finish_units_per_block(units);
        return (int)((units + upb - 1) / upb);
    }
    return ((N + ti - 1) / ti) * (Hg / th) * (Wg / tw) * (mode == M2_CONVT4S2 ? 4 : 1);
}

static int checks2(const Igemm2Params& p, const char* who) {
    DVG_REQUIRE(p.x && p.w && p.y, DVG_ERR_NULL, "%s: x/w/y must not be NULL", who);
    DVG_REQUIRE((p.skip != nullptr) == (p.C2 > 0), DVG_ERR_SHAPE, "%s: skip pointer / C2 mismatch", who);
    DVG_REQUIRE(p.N > 0 && p.H > 0 && p.W > 0, DVG_ERR_SHAPE, "%s: empty shape", who);
    DVG_REQUIRE(p.C1 > 0 && p.C1 % 16 == 0 && p.C2 % 16 == 0, DVG_ERR_SHAPE, "%s: C1=%d C2=%d must be multiples of 16",
                who, p.C1, p.C2);
    DVG_REQUIRE(p.Cout > 0 && p.Cout % 64 == 0, DVG_ERR_SHAPE, "%s: Cout=%d must be a multiple of 64", who, p.Cout);
    DVG_REQUIRE(aligned16(p.x) && aligned16(p.w) && aligned16(p.y) && aligned16(p.skip), DVG_ERR_ALIGN,
                "%s: pointers must be 16-byte aligned", who);
    DVG_REQUIRE(p.act >= DVG_ACT_NONE && p.act <= DVG_ACT_SIGMOID, DVG_ERR_SHAPE, "%s: bad act", who);
    return DVG_OK;
}

#define D2(MODE, TI_, TH_, TW_) \
    if (ti == TI_ && th == TH_ && tw == TW_)                                             \
        return launch2<MODE, TI_, TH_, TW_>(p, Hg, Wg, workspace, workspace_floats, (hipStream_t)stream);



extern "C" int dvg_conv3x3_bn_act_v2(const float* x, const float* skip, const float* w_k16, const float* scale,
                                     const float* shift, float* y, float* y_pool, float* stats, int N, int H, int W,
                                     int C1, int C2, int Cout, int upsample_x, int act, float slope,
                                     float* workspace, long workspace_floats, const float* addend,
                                     const int* addend_map, int addend_block, void* stream) {
    Igemm2Params p{x, skip, w_k16, scale, shift, y, y_pool, stats, N, H, W, C1, C2, Cout, upsample_x ? 1 : 0, act, slope,
                   0, 0, 0, 0, 0, 1, 0, nullptr};
    p.addend = addend;
    p.add_map = addend ? addend_map : nullptr;
    p.add_B = addend_block;
    DVG_REQUIRE(p.add_map == nullptr || (addend_block > 0 && N % addend_block == 0), DVG_ERR_SHAPE,
                "dvg_conv3x3_bn_act_v2: addend_block must divide N");
    if (int e = checks2(p, "dvg_conv3x3_bn_act_v2")) return e;
    DVG_REQUIRE(aligned16(addend) && (addend == nullptr || y_pool == nullptr), DVG_ERR_SHAPE,
                "dvg_conv3x3_bn_act_v2: addend must be 16-byte aligned and excludes the pooled output");
    DVG_REQUIRE(H % 8 == 0 && W % 8 == 0, DVG_ERR_SHAPE, "dvg_conv3x3_bn_act_v2: H=%d W=%d must be multiples of 8", H, W);
    int Hg = H, Wg = W, ti, th, tw;
    DVG_REQUIRE(tile2(M2_CONV3, N, Hg, Wg, Cout, &ti, &th, &tw) == 0 && ti == 1, DVG_ERR_SHAPE,
                "dvg_conv3x3_bn_act_v2: no tile for %dx%d", H, W);
    D2(M2_CONV3, 1, 8, 16)
    D2(M2_CONV3, 1, 8, 8)
    return fail(DVG_ERR_SHAPE, "dvg_conv3x3_bn_act_v2: no kernel");
}

// vgg_64's first stage c1 = vgg_layer(1, 64) -> vgg_layer(64, Cout) (vgg_64.py:23-26) in eval mode as ONE launch: the second
// layer's implicit GEMM computes its input tile from the frame (see Igemm2Params::first_*).  frame (N,1,H,W); w0 = the first
// layer's (64,1,3,3) weight TRANSPOSED to [9 taps][64 channels];
// scale0 / shift0 the first layer's folded BatchNorm (64 each); the rest as dvg_conv3x3_bn_act_v2 with C1 = 64, no skip,
// no split-K.  H % 8 == 0, W % 16 == 0, at least 512 workgroups (the 8 x 16 tile).
// y_from (ABI 8): y holds the images [y_from, N) only - the full-resolution output (the stage's skip tensor) of the images
// before is not stored (y may be NULL when y_from == N); y_pool always covers all N images.
extern "C" int dvg_conv3x3_first_pair(const float* frame, const float* w0, const float* scale0, const float* shift0,
                                      const float* w1_k16, const float* scale1, const float* shift1, float* y, float* y_pool,
                                      int N, int H, int W, int Cout, int act, float slope, int y_from, void* stream) {
    DVG_REQUIRE(frame && w0 && scale0 && shift0, DVG_ERR_NULL, "dvg_conv3x3_first_pair: NULL pointer");
    DVG_REQUIRE(y_from >= 0 && y_from <= N && (y_from == 0 || y_pool != nullptr) && (y != nullptr || y_from == N), DVG_ERR_SHAPE,
                "dvg_conv3x3_first_pair: y_from=%d needs 0 <= y_from <= N, a pooled output when > 0, y unless == N", y_from);
    // y holds the images [y_from, N): biased so that the kernel's image index applies (never dereferenced below y_from)
    float* const y_biased = y ? y - (size_t)y_from * H * W * Cout : y_pool;
    Igemm2Params p{frame /* never read as an activation */, nullptr, w1_k16, scale1, shift1, y_biased, y_pool, nullptr, N, H, W, 64, 0, Cout, 0,
                   act, slope, 0, 0, 0, 0, 0, 1, 0, nullptr};
    p.first_frame = frame; p.first_w = w0; p.first_scale = scale0; p.first_shift = shift0; p.first_slope = 0.2f;
    p.y_from = y_from;
    if (int e = checks2(p, "dvg_conv3x3_first_pair")) return e;
    DVG_REQUIRE(H % 8 == 0 && W % 16 == 0 && (long)N * (H / 8) * (W / 16) * (Cout / 64) >= 512, DVG_ERR_SHAPE,
                "dvg_conv3x3_first_pair: H %% 8, W %% 16 and >= 512 workgroups needed (N=%d H=%d W=%d Cout=%d)", N, H, W, Cout);
    return launch2<M2_CONV3, 1, 8, 16, 1, true>(p, H, W, nullptr, 0, (hipStream_t)stream);
}

extern "C" int dvg_conv4x4s2_bn_act_v2(const float* x, const float* w_k16, const float* scale, const float* shift,
                                       float* y, float* stats, int N, int H, int W, int Cin, int Cout, int act,
                                       float slope, float* workspace, long workspace_floats, void* stream) {
    Igemm2Params p{x, nullptr, w_k16, scale, shift, y, nullptr, stats, N, H, W, Cin, 0, Cout, 0, act, slope, 0, 0, 0, 0, 0,
                   1, 0, nullptr};
    if (int e = checks2(p, "dvg_conv4x4s2_bn_act_v2")) return e;
    DVG_REQUIRE(H % 2 == 0 && W % 2 == 0, DVG_ERR_SHAPE, "dvg_conv4x4s2_bn_act_v2: odd input");
    DVG_REQUIRE((long)N * H * W * Cin < (1L << 31), DVG_ERR_SHAPE,
                "dvg_conv4x4s2_bn_act_v2: input of %d x %d x %d x %d floats too large (32-bit offsets)", N, H, W, Cin);
    int Hg = H / 2, Wg = W / 2, ti, th, tw;
    DVG_REQUIRE(tile2(M2_CONV4S2, N, Hg, Wg, Cout, &ti, &th, &tw) == 0, DVG_ERR_SHAPE,
                "dvg_conv4x4s2_bn_act_v2: unsupported map %dx%d", H, W);
    D2(M2_CONV4S2, 1, 8, 16)
    D2(M2_CONV4S2, 1, 8, 8)
    D2(M2_CONV4S2, 4, 4, 4)
    return fail(DVG_ERR_SHAPE, "dvg_conv4x4s2_bn_act_v2: no kernel");
}

extern "C" int dvg_convT4x4s2_bn_act_v2(const float* x, const float* skip, const float* w_k16, const float* scale,
                                        const float* shift, float* y, float* stats, int N, int H, int W, int C1,
                                        int C2, int Cout, int act, float slope, float* workspace, long workspace_floats,
                                        const float* addend, const int* addend_map, int addend_block, void* stream) {
    Igemm2Params p{x, skip, w_k16, scale, shift, y, nullptr, stats, N, H, W, C1, C2, Cout, 0, act, slope, 0, 0, 0, 0, 0,
                   1, 0, nullptr};
    p.addend = addend;
    p.add_map = addend ? addend_map : nullptr;
    p.add_B = addend_block;
    DVG_REQUIRE(p.add_map == nullptr || (addend_block > 0 && N % addend_block == 0), DVG_ERR_SHAPE,
                "dvg_convT4x4s2_bn_act_v2: addend_block must divide N");
    if (int e = checks2(p, "dvg_convT4x4s2_bn_act_v2")) return e;
    DVG_REQUIRE(aligned16(addend), DVG_ERR_ALIGN, "dvg_convT4x4s2_bn_act_v2: addend must be 16-byte aligned");
    int Hg = H, Wg = W, ti, th, tw;
    DVG_REQUIRE(tile2(M2_CONVT4S2, N, Hg, Wg, Cout, &ti, &th, &tw) == 0, DVG_ERR_SHAPE,
                "dvg_convT4x4s2_bn_act_v2: unsupported map %dx%d", H, W);
    D2(M2_CONVT4S2, 1, 8, 16)
    D2(M2_CONVT4S2, 1, 8, 8)
    D2(M2_CONVT4S2, 4, 4, 4)
    return fail(DVG_ERR_SHAPE, "dvg_convT4x4s2_bn_act_v2: no kernel");
}

// Batched GEMM on the igemm machinery: y[b][px][co] = sum_ci x[b][px][ci] * w[b][ci][co] for b < NB "images" of H x W
// pixels each (x NHWC (NB,H,W,Cin), y NHWC (NB,H,W,Cout), w = NB packed slabs [Cin/16][1][Cout][16] as
// dvg_pack_conv_weight_k16 writes them for a 1x1 kernel).  The 16 Winograd-domain products of dvg_winograd_* below.
// Cin % 64 == 0, Cout % 64 == 0, H % 8 == 0, W % 8 == 0.
extern "C" int dvg_gemm_batched_k16(const float* x, const float* w_k16, float* y, int NB, int H, int W, int Cin, int Cout,
                                    void* stream) {
    Igemm2Params p{x, nullptr, w_k16, nullptr, nullptr, y, nullptr, nullptr, NB, H, W, Cin, 0, Cout, 0, DVG_ACT_NONE, 0.f,
                   0, 0, 0, 0, 0, 1, 0, nullptr};
    if (int e = checks2(p, "dvg_gemm_batched_k16")) return e;
    DVG_REQUIRE(Cin % (16 * DVG_GEMM_GT) == 0 && H % 8 == 0 && W % 8 == 0, DVG_ERR_SHAPE,
                "dvg_gemm_batched_k16: Cin=%d must be a multiple of %d, H=%d W=%d multiples of 8", Cin, 16 * DVG_GEMM_GT, H, W);
    DVG_REQUIRE((long)H * W * Cin < (1L << 31), DVG_ERR_SHAPE, "dvg_gemm_batched_k16: image of %d x %d x %d floats too large", H, W, Cin);
    p.w_image_stride = (long)Cin / 16 * Cout * DVG_WROW;
    float* workspace = nullptr;
    const long workspace_floats = 0;
    int Hg = H, Wg = W, ti, th, tw;
    DVG_REQUIRE(tile2(M2_GEMM, NB, Hg, Wg, Cout, &ti, &th, &tw) == 0 && ti == 1, DVG_ERR_SHAPE,
                "dvg_gemm_batched_k16: no tile for %dx%d", H, W);
#if DVG_BF16X3 && DVG_GEMM_NT2
    // 128 x 128 workgroup tile (64 x 64 per wave: half the fragment reads and half the L2 -> LDS bytes per MFMA of the 128 x 64
    // tile) from DVG_GEMM_NT2 workgroups on: the energy-lean tile (see the knob above)
    if (g_tile_policy && Cout % 128 == 0 && Wg % 16 == 0 && (long)NB * (Hg / 8) * (Wg / 16) * (Cout / 128) >= DVG_GEMM_NT2)
        return launch2<M2_GEMM, 1, 8, 16, 2>(p, Hg, Wg, workspace, workspace_floats, (hipStream_t)stream);
#endif
    D2(M2_GEMM, 1, 8, 16)
    D2(M2_GEMM, 1, 8, 8)
    return fail(DVG_ERR_SHAPE, "dvg_gemm_batched_k16: no kernel");
}
