// Does an EXEC change right behind issued MFMAs disturb them on gfx950?  (the select-form of
// dvg_conv3x3_first_pair's store phase - EXEC-masked blocks inside the MFMA-interleaved stage loop - gave run-to-run
// different tiles with two workgroups per CU.)
//
// Every wave runs CHAIN dependent v_mfma_f32_32x32x16_bf16 on random operands and, in mode 1, changes EXEC to a partial
// mask immediately behind the last of them (no s_nop), executes VALU instructions under that mask that overwrite a
// scratch register, then restores EXEC; mode 2 issues two of the four MFMAs UNDER the partial mask (does an MFMA honour EXEC
// at all?).  Mode 0 is the same stream without the EXEC change.  Mode 1's accumulators must be bit-identical to mode 0's if
// MFMAs sample EXEC at issue.  Launched with enough waves per CU (2 workgroups x 8 waves) that the
// matrix pipe is contended: a queued MFMA is then still waiting when the EXEC change issues.
//
//   hipcc -O2 --offload-arch=gfx950 tools/ubench/mfma_exec_hazard.hip -o gpurun_out/mfma_exec_hazard && gpurun_out/mfma_exec_hazard
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(2); } } while (0)

template <int MODE>
__global__ __launch_bounds__(512, 2) void k(const f32x4* __restrict__ a, const f32x4* __restrict__ b, float* __restrict__ out,
                                             int reps, unsigned long long mask) {
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    f32x4 fa = a[t], fb = b[t];
    f32x16 acc;
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[i] = 0.f;
    float junk = (float)threadIdx.x;
    for (int r = 0; r < reps; ++r) {
        if (MODE == 0) {
            asm volatile(
                "v_mfma_f32_32x32x16_bf16 %0, %2, %3, %0\n\t"
                "v_mfma_f32_32x32x16_bf16 %0, %2, %3, %0\n\t"
                "v_mfma_f32_32x32x16_bf16 %0, %2, %3, %0\n\t"
                "v_mfma_f32_32x32x16_bf16 %0, %2, %3, %0\n\t"
                "v_add_f32 %1, %1, %1\n\t"
                "v_add_f32 %1, 1.0, %1\n\t"
                "s_nop 7\n\t"
                : "+v"(acc), "+v"(junk)
                : "v"(fa), "v"(fb));
        } else if (MODE == 1) {
            asm volatile(
                "v_mfma_f32_32x32x16_bf16 %0, %2, %3, %0\n\t"
                "v_mfma_f32_32x32x16_bf16 %0, %2, %3, %0\n\t"
                "v_mfma_f32_32x32x16_bf16 %0, %2, %3, %0\n\t"
                "v_mfma_f32_32x32x16_bf16 %0, %2, %3, %0\n\t"
                "s_mov_b64 exec, %4\n\t"
                "v_add_f32 %1, %1, %1\n\t"
                "v_add_f32 %1, 1.0, %1\n\t"
                "s_mov_b64 exec, -1\n\t"
                "s_nop 7\n\t"
                : "+v"(acc), "+v"(junk)
                : "v"(fa), "v"(fb), "s"(mask));
        } else {
            // two of the four MFMAs ISSUED under the partial mask: does an MFMA honour EXEC at all?
            asm volatile(
                "v_mfma_f32_32x32x16_bf16 %0, %2, %3, %0\n\t"
                "s_mov_b64 exec, %4\n\t"
                "v_mfma_f32_32x32x16_bf16 %0, %2, %3, %0\n\t"
                "v_mfma_f32_32x32x16_bf16 %0, %2, %3, %0\n\t"
                "v_add_f32 %1, %1, %1\n\t"
                "v_add_f32 %1, 1.0, %1\n\t"
                "s_mov_b64 exec, -1\n\t"
                "v_mfma_f32_32x32x16_bf16 %0, %2, %3, %0\n\t"
                "s_nop 7\n\t"
                : "+v"(acc), "+v"(junk)
                : "v"(fa), "v"(fb), "s"(mask));
        }
        // keep the accumulator bounded: scale down between repetitions (same in all modes)
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[i] *= 0.125f;
    }
    float* o = out + (size_t)t * 17;
#pragma unroll
    for (int i = 0; i < 16; ++i) o[i] = acc[i];
    o[16] = junk;
}

static uint16_t bf16_bits(float f) {
    uint32_t u;
    memcpy(&u, &f, 4);
    return (uint16_t)((u + 0x7FFF + ((u >> 16) & 1)) >> 16);
}

int main() {
    const int blocks = 1024, threads = 512, n = blocks * threads, reps = 64;
    std::vector<uint16_t> ha((size_t)n * 8), hb((size_t)n * 8);
    srand(7);
    for (auto& v : ha) v = bf16_bits((rand() / (float)RAND_MAX - 0.5f));
    for (auto& v : hb) v = bf16_bits((rand() / (float)RAND_MAX - 0.5f));
    f32x4 *da, *db;
    float* dout[3];
    CK(hipMalloc(&da, (size_t)n * 16));
    CK(hipMalloc(&db, (size_t)n * 16));
    CK(hipMemcpy(da, ha.data(), (size_t)n * 16, hipMemcpyHostToDevice));
    CK(hipMemcpy(db, hb.data(), (size_t)n * 16, hipMemcpyHostToDevice));
    for (auto& d : dout) CK(hipMalloc(&d, (size_t)n * 17 * 4));
    const unsigned long long masks[4] = {0x5555555555555555ull, 0x00000000FFFFFFFFull, 0x0ull, ~0ull};   // the last: same stream, nothing masked
    int bad_total = 0;
    for (int mi = 0; mi < 4; ++mi) {
        for (int trial = 0; trial < 3; ++trial) {
            hipLaunchKernelGGL(k<0>, dim3(blocks), dim3(threads), 0, 0, da, db, dout[0], reps, masks[mi]);
            hipLaunchKernelGGL(k<1>, dim3(blocks), dim3(threads), 0, 0, da, db, dout[1], reps, masks[mi]);
            hipLaunchKernelGGL(k<2>, dim3(blocks), dim3(threads), 0, 0, da, db, dout[2], reps, masks[mi]);
            CK(hipDeviceSynchronize());
            std::vector<float> h0((size_t)n * 17), h1((size_t)n * 17), h2((size_t)n * 17);
            CK(hipMemcpy(h0.data(), dout[0], h0.size() * 4, hipMemcpyDeviceToHost));
            CK(hipMemcpy(h1.data(), dout[1], h1.size() * 4, hipMemcpyDeviceToHost));
            CK(hipMemcpy(h2.data(), dout[2], h2.size() * 4, hipMemcpyDeviceToHost));
            long bad1 = 0, bad2 = 0, nz = 0;
            unsigned long long lanes2 = 0;
            unsigned regs2 = 0;
            for (size_t t = 0; t < (size_t)n; ++t)
                for (int i = 0; i < 16; ++i) {
                    nz += h0[t * 17 + i] != 0.f;
                    bad1 += memcmp(&h0[t * 17 + i], &h1[t * 17 + i], 4) != 0;
                    if (memcmp(&h0[t * 17 + i], &h2[t * 17 + i], 4) != 0) { ++bad2; lanes2 |= 1ull << (t & 63); regs2 |= 1u << i; }
                }
            if (trial == 0) printf("   mode 2: lanes that differ %016llx, accumulator registers that differ %04x\n", lanes2, regs2);
            printf("mask %016llx trial %d: accumulators differing from the unmasked stream: EXEC change behind the MFMAs %ld, MFMAs issued under the mask %ld (of %ld, %ld nonzero)\n",
                   masks[mi], trial, bad1, bad2, (long)n * 16, nz);
            bad_total += (bad1 != 0);
        }
    }
    printf("(mode 1 differing = an EXEC change behind issued MFMAs disturbs them; mode 2 differing = MFMAs honour EXEC)\n");
    return 0;
}
