// Semantics check of global_load_lds_dwordx4 on gfx950: every lane supplies its own global address, the LDS destination is the
// wave-uniform base (M0) + lane * 16.  Prints "ok" when a 256-thread workgroup copies 4 KB through LDS-DMA correctly.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
__global__ void k(const float* in, float* out) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int tid = threadIdx.x;
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(in + (size_t)blockIdx.x * 1024 + tid * 4),
                                     (__attribute__((address_space(3))) void*)(lds + (tid >> 6) * 256), 16, 0, 0);
    __builtin_amdgcn_s_waitcnt(0);
    __syncthreads();
    for (int j = 0; j < 4; ++j) out[(size_t)blockIdx.x * 1024 + tid * 4 + j] = lds[tid * 4 + j] * 2.f;
}
int main() {
    const int nb = 64, n = nb * 1024;
    std::vector<float> h(n), r(n);
    for (int i = 0; i < n; ++i) h[i] = (float)(i % 9973);
    float *a, *b;
    hipMalloc(&a, n * 4); hipMalloc(&b, n * 4);
    hipMemcpy(a, h.data(), n * 4, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(nb), dim3(256), 4096, 0, a, b);
    hipMemcpy(r.data(), b, n * 4, hipMemcpyDeviceToHost);
    int bad = 0;
    for (int i = 0; i < n; ++i) bad += r[i] != 2.f * h[i];
    printf(bad ? "MISMATCH %d\n" : "ok\n", bad);
    return bad != 0;
}
