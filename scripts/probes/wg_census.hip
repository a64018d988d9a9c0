// scripts/probes/wg_census.hip -- where do the workgroups of a 2-per-CU persistent launch land?  (speed question only: which
// blockIdx share a CU / a SIMD, in which wave slots.)  Same launch shape as mlp_pm_lds_persist_kernel: 256 threads, 73728 bytes of
// dynamic LDS, __launch_bounds__(256, 2), 512 workgroups.   hipcc --offload-arch=gfx950 -O2 wg_census.hip -o wg_census
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>

struct Rec { unsigned hw_id, xcc, wave, block; unsigned long long t0; };

__global__ void __launch_bounds__(256, 2) census(Rec* out, int spin)
{
    extern __shared__ unsigned char lds[];
    const int wave = threadIdx.x >> 6;
    if ((threadIdx.x & 63) == 0) {
        Rec r;
        r.hw_id = __builtin_amdgcn_s_getreg((31 << 11) | 4);      // HW_REG_HW_ID
        r.xcc = __builtin_amdgcn_s_getreg((31 << 11) | 20);       // HW_REG_XCC_ID
        r.wave = wave;
        r.block = blockIdx.x;
        r.t0 = __builtin_amdgcn_s_memtime();
        out[blockIdx.x * 4 + wave] = r;
    }
    // stay resident long enough for every workgroup to be placed
    float v = threadIdx.x;
    for (int i = 0; i < spin; ++i) v = v * 1.0001f + 0.5f;
    if (v == 12345.f) lds[threadIdx.x] = 1;
}

int main()
{
    const int grid = 512;
    Rec* d;
    hipMalloc(&d, grid * 4 * sizeof(Rec));
    hipFuncSetAttribute(reinterpret_cast<const void*>(&census), hipFuncAttributeMaxDynamicSharedMemorySize, 73728);
    hipLaunchKernelGGL(census, dim3(grid), dim3(256), 73728, 0, d, 200000);
    hipDeviceSynchronize();
    std::vector<Rec> h(grid * 4);
    hipMemcpy(h.data(), d, h.size() * sizeof(Rec), hipMemcpyDeviceToHost);
    unsigned long long tmin = ~0ull;
    for (auto& r : h) tmin = std::min(tmin, r.t0);
    printf("# block wave | xcc se sh cu simd wave_id tg_id | hw_id | t0-tmin\n");
    for (int b = 0; b < grid; ++b)
        for (int w = 0; w < 4; ++w) {
            const Rec& r = h[b * 4 + w];
            const unsigned id = r.hw_id;
            if (b < 80 || b % 37 == 0)
                printf("%4u %u | %u %u %u %2u %u %2u %2u | %08x | %llu\n", r.block, r.wave, r.xcc & 15, (id >> 13) & 7, (id >> 12) & 1, (id >> 8) & 15,
                       (id >> 4) & 3, id & 15, (id >> 16) & 15, id, r.t0 - tmin);
        }
    // which blocks share a CU?  key = (xcc, se, sh, cu)
    std::vector<std::pair<unsigned, unsigned>> key;
    for (int b = 0; b < grid; ++b) {
        const Rec& r = h[b * 4];
        key.push_back({((r.xcc & 15) << 16) | ((r.hw_id >> 8) & 0xff), (unsigned)b});
    }
    std::sort(key.begin(), key.end());
    printf("# blocks per CU (xcc, se/sh/cu bits): partner lists\n");
    int shown = 0;
    for (size_t i = 0; i < key.size();) {
        size_t j = i;
        while (j < key.size() && key[j].first == key[i].first) ++j;
        if (shown++ < 40) {
            printf("cu %06x:", key[i].first);
            for (size_t k = i; k < j; ++k) {
                const Rec& r = h[key[k].second * 4];
                printf(" b%u(simd%u slot%u tg%u)", key[k].second, (r.hw_id >> 4) & 3, r.hw_id & 15, (r.hw_id >> 16) & 15);
            }
            printf("\n");
        }
        i = j;
    }
    // histogram of workgroups per CU
    int hist[8] = {0};
    for (size_t i = 0; i < key.size();) {
        size_t j = i;
        while (j < key.size() && key[j].first == key[i].first) ++j;
        hist[std::min<size_t>(j - i, 7)]++;
        i = j;
    }
    printf("# CUs holding n workgroups: ");
    for (int n = 1; n < 8; ++n) printf("n=%d:%d ", n, hist[n]);
    printf("\n");
    return 0;
}
