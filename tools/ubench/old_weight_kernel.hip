// DIAGNOSTIC (r06, profiles/r06_dp_race_bisect.txt): the F(4x4) weight transform in its form until r06 (thread = (co, ci), 64-bit
// index arithmetic, 2-byte stores) - the kernel that lost its last stores for lanes 48-63 whenever another process was busy on the
// device - as a stand-alone library, in two variants:
//   mode 0: as it was (the wave reaches s_endpgm with EXEC = 0 right after its last three stores);
//   mode 1: the same kernel, but EXEC restored to all ones and s_waitcnt vmcnt(0) before the wave ends.
// tools/diag_old_weight_kernel.py recomputes U with both while another process trains on the device: does the EXEC = 0 tail lose
// the stores?  Not part of libdvg_hip.so.   hipcc -O3 -std=c++17 --offload-arch=gfx950 -shared -fPIC -o libold_weight.so old_weight_kernel.hip
#include "../../dvg_amd/csrc/dvg_common.h"

namespace dvg {

__device__ __forceinline__ void wrow_store_r05(float* __restrict__ rows, size_t row, int co_local, int k, float v) {
    unsigned short* d = reinterpret_cast<unsigned short*>(rows + row * 24);
    unsigned ph, pm, pl;
    bf16x3_split_pair(v, 0.f, ph, pm, pl);
    const int pos = (((k >> 3) ^ ((co_local >> 3) & 1)) << 3) + (k & 7);
    d[pos] = (unsigned short)(ph & 0xffffu);
    d[16 + pos] = (unsigned short)(pm & 0xffffu);
    d[32 + pos] = (unsigned short)(pl & 0xffffu);
}

//   mode 4: the kernel as it was, s_waitcnt vmcnt(0) after every position's three stores
//   mode 3: the kernel as it was, transform positions written in descending order (does the fault follow the LAST stores or position 35?)
//   mode 2: the kernel as it was, but with 32-bit loop / index arithmetic (IDX = unsigned) - no 64-bit division, no v_mad_u64_u32 chain
//           per store other than the final pointer add.
template <int MODE, typename IDX>
__global__ void winograd4_weight_kernel_r05(const float* __restrict__ w, float* __restrict__ u, int cout, int cin,
                                            unsigned long long* __restrict__ dbg, float z) {
    // mode 7: the last row of G, {0, 0, 1}, from a kernel argument (z = 0 at run time): the compiler can no longer fold the zero
    // multipliers into inline constants of packed FMAs
    // mode 5: per wave, the constant-rate wall clock (100 MHz) when it starts and when it ends -> dbg[2 wave], dbg[2 wave + 1]:
    // was a wave with a wrong row on the chip for much longer than its peers (context-switched out for the other process)?
    unsigned long long clk0 = 0;
    if (MODE == 5) clk0 = wall_clock64();
    const float G[6][3] = {{0.25f, 0.f, 0.f}, {-1.f / 6, -1.f / 6, -1.f / 6}, {-1.f / 6, 1.f / 6, -1.f / 6},
                           {1.f / 24, 1.f / 12, 1.f / 6}, {1.f / 24, -1.f / 12, 1.f / 6},
                           {MODE == 7 ? z : 0.f, MODE == 7 ? z : 0.f, MODE == 7 ? 1.f + z : 1.f}};
    const IDX total = (IDX)cout * (IDX)cin;
    for (IDX i = blockIdx.x * (IDX)blockDim.x + threadIdx.x; i < total; i += (IDX)gridDim.x * blockDim.x) {
        const int ci = (int)(i % (IDX)cin), co = (int)(i / (IDX)cin);
        const float* g = w + ((size_t)co * cin + ci) * 9;
        float t[6][3];
#pragma unroll
        for (int a = 0; a < 6; ++a)
#pragma unroll
            for (int s2 = 0; s2 < 3; ++s2) t[a][s2] = G[a][0] * g[s2] + G[a][1] * g[3 + s2] + G[a][2] * g[6 + s2];
#pragma unroll
        for (int aa = 0; aa < 6; ++aa)
#pragma unroll
            for (int bb = 0; bb < 6; ++bb) {
                const int a = MODE == 3 ? 5 - aa : aa, b = MODE == 3 ? 5 - bb : bb;      // mode 3: the positions in DESCENDING order
                const float val = t[a][0] * G[b][0] + t[a][1] * G[b][1] + t[a][2] * G[b][2];
                wrow_store_r05(u, (((IDX)(a * 6 + b) * (cout >> 6) + (co >> 6)) * (cin / 16) + ci / 16) * 64 + (co & 63), co & 63,
                               ci & 15, val);
                if (MODE == 6 && a == 5 && b == 5) {      // mode 6: the fp32 value of position 35 and the g[8] it comes from, as plain dwords
                    reinterpret_cast<float*>(dbg)[2 * i] = val;
                    reinterpret_cast<float*>(dbg)[2 * i + 1] = g[8];
                }
                if (MODE == 4) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     // mode 4: every store performed before anything else happens
            }
    }
    if (MODE == 1) asm volatile("s_mov_b64 exec, -1\n\ts_waitcnt vmcnt(0)" ::: "memory");
    if (MODE == 5) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        const unsigned long long clk1 = wall_clock64();
        const unsigned wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
        if ((threadIdx.x & 63) == 0) {
            dbg[2 * wave] = clk0;
            dbg[2 * wave + 1] = clk1;
        }
    }
}

}  // namespace dvg

extern "C" int old_weight_transform(const float* w, float* u, int cout, int cin, int mode, void* stream, void* dbg) {
    if (!w || !u || cout <= 0 || cout % 64 || cin <= 0 || cin % 16) return 1;
    long g = ((long)cout * cin + 255) / 256;
    if (g > 16384) g = 16384;
    if (mode == 0)
        hipLaunchKernelGGL((dvg::winograd4_weight_kernel_r05<0, long>), dim3((unsigned)g), dim3(256), 0, (hipStream_t)stream, w, u, cout, cin, (unsigned long long*)dbg, 0.f);
    else if (mode == 1)
        hipLaunchKernelGGL((dvg::winograd4_weight_kernel_r05<1, long>), dim3((unsigned)g), dim3(256), 0, (hipStream_t)stream, w, u, cout, cin, (unsigned long long*)dbg, 0.f);
    else if (mode == 7)
        hipLaunchKernelGGL((dvg::winograd4_weight_kernel_r05<7, long>), dim3((unsigned)g), dim3(256), 0, (hipStream_t)stream, w, u, cout, cin, (unsigned long long*)dbg, 0.f);
    else if (mode == 6)
        hipLaunchKernelGGL((dvg::winograd4_weight_kernel_r05<6, long>), dim3((unsigned)g), dim3(256), 0, (hipStream_t)stream, w, u, cout, cin, (unsigned long long*)dbg, 0.f);
    else if (mode == 5)
        hipLaunchKernelGGL((dvg::winograd4_weight_kernel_r05<5, long>), dim3((unsigned)g), dim3(256), 0, (hipStream_t)stream, w, u, cout, cin, (unsigned long long*)dbg, 0.f);
    else if (mode == 4)
        hipLaunchKernelGGL((dvg::winograd4_weight_kernel_r05<4, long>), dim3((unsigned)g), dim3(256), 0, (hipStream_t)stream, w, u, cout, cin, (unsigned long long*)dbg, 0.f);
    else if (mode == 3)
        hipLaunchKernelGGL((dvg::winograd4_weight_kernel_r05<3, long>), dim3((unsigned)g), dim3(256), 0, (hipStream_t)stream, w, u, cout, cin, (unsigned long long*)dbg, 0.f);
    else
        hipLaunchKernelGGL((dvg::winograd4_weight_kernel_r05<0, unsigned>), dim3((unsigned)g), dim3(256), 0, (hipStream_t)stream, w, u, cout, cin, (unsigned long long*)dbg, 0.f);
    return hipGetLastError() == hipSuccess ? 0 : 2;
}
