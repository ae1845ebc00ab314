// Throughput of the per-tile sorts (csrc/tile_sort.h) as blend_fwd dispatches them: G lists of n uniformly spread keys each, one
// 256-thread workgroup per list (n <= 512: wave 0 alone; <= 2048: the workgroup's bucket sort in LDS; above: slabs through the
// global scratch array).  Prints microseconds per launch and nanoseconds per entry.
//   hipcc -O3 --offload-arch=gfx950 -I bundle-adjusting-gaussian-splatting_amd/csrc -I include tools/ubench/sort_rate.hip -o /tmp/sort_rate && /tmp/sort_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#include "bags_common.h"
#include "tile_sort.h"

__global__ void __launch_bounds__(256, 6) lists(const u32* words32, const u32* keys, u64* scratch, u32* point_list, u32 n)
{
    __shared__ u64 t_all[TSORT_BLOCK];
    __shared__ u32 cnt_all[TSORT_BLOCK / 2];
    __shared__ TileSortLds L;
    const u32 first = blockIdx.x * n;
    const uint4 desc = make_uint4(0u, first, n, 0u);
    const WordSrc src = tile_words(words32, first, keys);
    if (n <= TSORT_WAVE) { if ((threadIdx.x >> 6) == 0) sort_wave_role(desc, src, point_list, t_all, cnt_all); }
    else sort_list_block(desc, src, scratch, point_list, t_all, cnt_all, L);
}

int main()
{
    const u32 P = 500000;
    std::vector<u32> keys(P);
    srand(1);
    for (u32 i = 0; i < P; ++i) { const float z = 3.0f + 2.0f * (float)rand() / (float)RAND_MAX; memcpy(&keys[i], &z, 4); }
    u32* d_k; hipMalloc(&d_k, P * 4); hipMemcpy(d_k, keys.data(), P * 4, hipMemcpyHostToDevice);
    const u32 sizes[] = {128, 254, 512, 600, 1000, 1500, 2048, 2100, 3000, 5000, 10000, 16384, 20000};
    for (u32 n : sizes) {
        const u32 G = (8u << 20) / n;                          // ~8 M entries per launch
        const size_t I = (size_t)G * n;
        std::vector<u32> w(2 * I);
        for (u32 g = 0; g < G; ++g) for (u32 i = 0; i < n; ++i) w[2 * (size_t)g * n + i] = (u32)(((size_t)rand() * 7919u + i) % P);
        u32 *d_w, *d_pl; u64* d_s;
        hipMalloc(&d_w, 2 * I * 4); hipMalloc(&d_pl, I * 4); hipMalloc(&d_s, I * 8);
        hipMemcpy(d_w, w.data(), 2 * I * 4, hipMemcpyHostToDevice);
        hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
        for (int r = 0; r < 2; ++r) hipLaunchKernelGGL(lists, dim3(G), dim3(256), 0, 0, d_w, d_k, d_s, d_pl, n);
        hipEventRecord(a, 0);
        for (int r = 0; r < 5; ++r) hipLaunchKernelGGL(lists, dim3(G), dim3(256), 0, 0, d_w, d_k, d_s, d_pl, n);
        hipEventRecord(b, 0); hipEventSynchronize(b);
        float ms = 0; hipEventElapsedTime(&ms, a, b); ms /= 5;
        printf("n = %5u, %6u lists: %8.1f us per launch, %6.3f ns per entry\n", n, G, ms * 1e3f, ms * 1e6f / (float)I);
        hipFree(d_w); hipFree(d_pl); hipFree(d_s);
    }
    return 0;
}
