// Microbenchmark: what does the f32 MFMA pipe sustain for the instruction mix of the conv inner loop?
//   variant 0: registers only            variant 1: operands re-read from LDS (6 ds_read_b128 per 16 MFMAs, one tap ahead)
//   waves per SIMD: 1 or 2 (block = 256 threads; 1 or 2 blocks per CU via LDS size), MT = accumulators per wave
// Build: hipcc -O3 --offload-arch=gfx950 mfma_f32_ceiling.hip -o mfma_f32_ceiling ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int MT, bool LDS>
__global__ __launch_bounds__(256, 2) void k(float* out, const float* in, int iters, int lds_pad_floats) {
    extern __shared__ __attribute__((aligned(16))) float sm[];
    const int tid = threadIdx.x, lane = tid & 63;
    for (int i = tid; i < 4096; i += 256) sm[i] = in[i];
    __syncthreads();
    f32x16 acc[MT];
    for (int m = 0; m < MT; ++m)
        for (int i = 0; i < 16; ++i) acc[m][i] = 0.f;
    const int base = (lane & 31) * 20 + (lane >> 5) * 8;
    f32x4 fa[2][MT][2], fb[2][2];
    for (int j = 0; j < 2; ++j) {
        for (int m = 0; m < MT; ++m) fa[0][m][j] = *(const f32x4*)&sm[base + m * 640 + j * 4];
        fb[0][j] = *(const f32x4*)&sm[base + 2048 + j * 4];
    }
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int tt = 0; tt < 8; ++tt) {
            const int cur = tt & 1, nxt = cur ^ 1;
            if (LDS) {
                for (int j = 0; j < 2; ++j) {
                    for (int m = 0; m < MT; ++m) fa[nxt][m][j] = *(const f32x4*)&sm[base + m * 640 + ((tt + 1) & 7) * 20 + j * 4];
                    fb[nxt][j] = *(const f32x4*)&sm[base + 2048 + ((tt + 1) & 7) * 20 + j * 4];
                }
            } else {
                for (int j = 0; j < 2; ++j) {
                    for (int m = 0; m < MT; ++m) fa[nxt][m][j] = fa[cur][m][j];
                    fb[nxt][j] = fb[cur][j];
                }
            }
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int e = 0; e < 4; ++e)
#pragma unroll
                    for (int m = 0; m < MT; ++m)
                        acc[m] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[cur][m][j][e], fb[cur][j][e], acc[m], 0, 0, 0);
            if (LDS) {
                constexpr int NREAD = 2 * MT + 2, NMFMA = 8 * MT;
                for (int r = 0; r < NREAD; ++r) {
                    __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
                    __builtin_amdgcn_sched_group_barrier(0x008, NMFMA / NREAD > 0 ? NMFMA / NREAD : 1, 0);
                }
            }
        }
    }
    float s = 0.f;
    for (int m = 0; m < MT; ++m)
        for (int i = 0; i < 16; ++i) s += acc[m][i];
    out[blockIdx.x * 256 + tid] = s;
}

template <int MT, bool LDS>
void run(const char* name, int blocks_per_cu, float* out, float* in) {
    const int iters = 2000;
    // force residency: 1 block/CU -> 100 KB of LDS, 2 blocks/CU -> 60 KB
    const int lds = blocks_per_cu == 1 ? 100 * 1024 : 60 * 1024;
    hipFuncSetAttribute((const void*)k<MT, LDS>, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    const int grid = 256 * blocks_per_cu;
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    hipLaunchKernelGGL((k<MT, LDS>), dim3(grid), dim3(256), lds, 0, out, in, 10, 0);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL((k<MT, LDS>), dim3(grid), dim3(256), lds, 0, out, in, iters, 0);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    const double flops = (double)grid * 4 /*waves*/ * iters * 8.0 * 8 * MT * 4096.0;
    printf("%-34s MT=%d waves/SIMD=%d : %7.1f TFLOP/s (%.1f%% of 157.3)\n", name, MT, blocks_per_cu, flops / ms / 1e9,
           flops / ms / 1e9 / 157.3 * 100);
}

int main() {
    float *out, *in;
    hipMalloc(&out, 512 * 256 * 4);
    hipMalloc(&in, 4096 * 4);
    std::vector<float> h(4096);
    for (int i = 0; i < 4096; ++i) h[i] = (float)((i * 2654435761u) % 1000) / 1000.f - 0.5f;
    hipMemcpy(in, h.data(), 4096 * 4, hipMemcpyHostToDevice);
    run<2, false>("registers only", 1, out, in);
    run<2, false>("registers only", 2, out, in);
    run<1, false>("registers only", 2, out, in);
    run<2, true>("LDS operands (conv loop mix)", 1, out, in);
    run<2, true>("LDS operands (conv loop mix)", 2, out, in);
    run<1, true>("LDS operands (conv loop mix)", 2, out, in);
    run<4, true>("LDS operands (conv loop mix)", 1, out, in);
    run<4, true>("LDS operands (conv loop mix)", 2, out, in);
    return 0;
}
