// Are two workgroups whose dynamic LDS allocations sum to MORE than 160 KiB ever co-resident on one CU of gfx950, and do they
// then see each other's bytes?  (conv_igemm2.hip, r03: "two 80.6 KB allocations are both admitted and then CORRUPT each
// other's last kilobytes" - VERDICT r03 item 4 asks for the cause.)
//
// Each workgroup fills ALL of its LDS with a pattern derived from its block id, spins ~20 us (long enough for a second
// workgroup to be placed on the same CU), and checks the pattern.  Per LDS size: workgroups with a corrupted pattern, the
// largest number of workgroups seen on one CU at the same time (by {XCC_ID, HW_ID CU fields} and overlapping [t0, t1]
// s_memrealtime intervals), and the first corrupted offset.
//   hipcc -O2 --offload-arch=gfx950 tools/ubench/lds_oversubscribe.hip -o /tmp/lds_oversubscribe && /tmp/lds_oversubscribe
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <map>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(2); } } while (0)

extern __shared__ unsigned lds[];

struct Rec { unsigned long long t0, t1; unsigned xcc, hwid, bad, first_bad; };

__global__ __launch_bounds__(256, 2) void k(Rec* rec, int words, int spin) {
    const unsigned tag = (blockIdx.x + 1) * 0x9E3779B1u;
    const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
    for (int i = threadIdx.x; i < words; i += 256) lds[i] = tag ^ (unsigned)i;
    __syncthreads();
    for (int s = 0; s < spin; ++s) __builtin_amdgcn_s_sleep(64);
    __syncthreads();
    unsigned bad = 0, first = 0xFFFFFFFFu;
    for (int i = threadIdx.x; i < words; i += 256)
        if (lds[i] != (tag ^ (unsigned)i)) { ++bad; first = min(first, (unsigned)i); }
    __shared__ unsigned sb, sf;
    if (threadIdx.x == 0) { sb = 0; sf = 0xFFFFFFFFu; }
    __syncthreads();
    atomicAdd(&sb, bad);
    atomicMin(&sf, first);
    __syncthreads();
    if (threadIdx.x == 0) {
        unsigned xcc, hwid;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hwid));
        rec[blockIdx.x] = Rec{t0, __builtin_amdgcn_s_memrealtime(), xcc, hwid, sb, sf};
    }
}

int main() {
    const int blocks = 1024;
    Rec* d;
    CK(hipMalloc(&d, blocks * sizeof(Rec)));
    const int sizes[] = {76 * 1024 + 768, 80 * 1024 - 32, 80 * 1024 - 8, 80 * 1024 + 512, 82534, 81 * 1024, 96 * 1024, 120 * 1024};
    for (int bytes : sizes) {
        const int dyn = bytes - 8;     // the kernel's two static words
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&k), hipFuncAttributeMaxDynamicSharedMemorySize, dyn);
        if (e != hipSuccess) { printf("%7d B: hipFuncSetAttribute: %s\n", bytes, hipGetErrorString(e)); continue; }
        int occ = -1;
        CK(hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, k, 256, dyn));
        hipLaunchKernelGGL(k, dim3(blocks), dim3(256), dyn, 0, d, dyn / 4, 40);
        CK(hipDeviceSynchronize());
        std::vector<Rec> h(blocks);
        CK(hipMemcpy(h.data(), d, blocks * sizeof(Rec), hipMemcpyDeviceToHost));
        long bad_wgs = 0, bad_words = 0;
        unsigned first = 0xFFFFFFFFu;
        std::map<unsigned long long, std::vector<std::pair<unsigned long long, int>>> ev;   // CU key -> (+1 at t0, -1 at t1)
        for (auto& r : h) {
            bad_wgs += r.bad != 0;
            bad_words += r.bad;
            first = std::min(first, r.first_bad);
            // HW_ID: [11:8] CU id, [15:13] SE id (gfx9 layout), + XCC id
            const unsigned long long key = ((unsigned long long)(r.xcc & 0xF) << 32) | (r.hwid & 0x0000FF00u);
            ev[key].push_back({r.t0, +1});
            ev[key].push_back({r.t1, -1});
        }
        int maxres = 0;
        for (auto& kv : ev) {
            std::sort(kv.second.begin(), kv.second.end());
            int cur = 0;
            for (auto& p : kv.second) { cur += p.second; maxres = std::max(maxres, cur); }
        }
        printf("%7d B per workgroup (2 x = %7d B, 160 KiB = 163840): occupancy API %d/CU; %zu CU keys; max co-resident per CU key %d; "
               "workgroups with corrupted LDS %ld (%ld words, first word %u)\n",
               bytes, 2 * bytes, occ, ev.size(), maxres, bad_wgs, bad_words, first);
    }
    return 0;
}
