// Standalone check of sort_list_block (csrc/tile_sort.h) on ONE list with depth-clustered keys: K = ceil(n / 256) slabs of which
// some are empty and one holds a single entry.  Prints whether point_list is the sorted permutation; no rasterizer around it, so
// a wrong list cannot turn into an out-of-bounds gather.
//   hipcc -O3 --offload-arch=gfx950 -I bundle-adjusting-gaussian-splatting_amd/csrc -I include tools/ubench/sort_slabs.hip -o /tmp/sort_slabs && /tmp/sort_slabs
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#include <algorithm>
#include "bags_common.h"
#include "tile_sort.h"

__global__ void __launch_bounds__(256) one_list(const u32* words32, const u32* keys, u64* scratch, u32* point_list, u32 n, u32 first)
{
    __shared__ u64 t_all[TSORT_BLOCK];
    __shared__ u32 cnt_all[TSORT_BLOCK / 2];
    __shared__ TileSortLds L;
    const uint4 desc = make_uint4(0u, first, n, 0u);
    sort_list_block(desc, tile_words(words32, first, keys), scratch, point_list, t_all, cnt_all, L);
}

int main(int argc, char** argv)
{
    const u32 n = argc > 1 ? (u32)atoi(argv[1]) : 2304u, first = 7u;
    const u32 K = (n + 255) / 256;
    std::vector<u32> keys(n), ids(n);
    srand(3);
    // argv[2]: bit mask of the populated slabs (default 0x155 = 0, 2, 4, 6, 8); argv[3]: 1 = no first / last / lone special keys
    const u32 mask = argc > 2 ? (u32)strtoul(argv[2], nullptr, 0) : 0x155u;
    const bool plain = argc > 3 && atoi(argv[3]) == 3;
    std::vector<int> bands;
    for (int b = 0; b < (int)K; ++b) if (mask >> b & 1u) bands.push_back(b);
    for (u32 i = 0; i < n; ++i) {
        const int b = bands[rand() % bands.size()];
        const float z = 4.0f + 1.8f * ((float)b + 0.1f + 0.8f * (float)rand() / (float)RAND_MAX) / (float)K;
        memcpy(&keys[i], &z, 4);
        ids[i] = i;
    }
    const int skip = argc > 3 ? atoi(argv[3]) : 0;          // bit 0: no first / last key, bit 1: no lone entry, bit 2: one in every empty slab
    if (!(skip & 1)) { float z = 4.0f; memcpy(&keys[0], &z, 4); z = 4.0f + 1.8f * 0.999999f; memcpy(&keys[1], &z, 4); }
    if (!(skip & 2)) { float z = 4.0f + 1.8f * 1.5f / (float)K; memcpy(&keys[2], &z, 4); }
    if (skip & 4) {                                          // a single entry in EVERY slab the mask leaves empty
        u32 j = 3;
        for (u32 b = 0; b < K; ++b) if (!(mask >> b & 1u)) { float z = 4.0f + 1.8f * ((float)b + 0.5f) / (float)K; memcpy(&keys[j++], &z, 4); }
    }
    std::random_shuffle(ids.begin(), ids.end());
    std::vector<u32> words32(2 * (first + n), 0xDEADBEEFu);
    for (u32 i = 0; i < n; ++i) words32[2 * first + i] = ids[i];
    u32 *d_w, *d_k, *d_pl; u64* d_s;
    hipMalloc(&d_w, words32.size() * 4); hipMalloc(&d_k, n * 4); hipMalloc(&d_pl, (first + n) * 4); hipMalloc(&d_s, (first + n) * 8);
    hipMemcpy(d_w, words32.data(), words32.size() * 4, hipMemcpyHostToDevice);
    hipMemcpy(d_k, keys.data(), n * 4, hipMemcpyHostToDevice);
    hipMemset(d_pl, 0xFF, (first + n) * 4); hipMemset(d_s, 0xEE, (first + n) * 8);
    int bad_runs = 0;
    for (int rep = 0; rep < 20; ++rep) {
        hipMemset(d_pl, 0xFF, (first + n) * 4);
        hipLaunchKernelGGL(one_list, dim3(1), dim3(256), 0, 0, d_w, d_k, d_s, d_pl, n, first);
        if (hipDeviceSynchronize() != hipSuccess) { printf("launch failed\n"); return 1; }
        std::vector<u32> pl(first + n);
        hipMemcpy(pl.data(), d_pl, (first + n) * 4, hipMemcpyDeviceToHost);
        std::vector<u32> want(n);
        for (u32 i = 0; i < n; ++i) want[i] = i;
        std::sort(want.begin(), want.end(), [&](u32 a, u32 b) { return keys[a] != keys[b] ? keys[a] < keys[b] : a < b; });
        u32 mism = 0, unwritten = 0, firstbad = n;
        for (u32 i = 0; i < n; ++i) {
            if (pl[first + i] == 0xFFFFFFFFu) ++unwritten;
            if (pl[first + i] != want[i]) { ++mism; if (firstbad == n) firstbad = i; }
        }
        if (mism) {
            ++bad_runs;
            printf("rep %d: %u mismatches (%u never written), first at %u\n", rep, mism, unwritten, firstbad);
            if (bad_runs == 1) {                                 // which slabs are wrong, and is each a permutation of its own ids?
                u32 kmin = 0xFFFFFFFFu, kmax = 0;
                for (u32 i = 0; i < n; ++i) { kmin = std::min(kmin, keys[i]); kmax = std::max(kmax, keys[i]); }
                const float scale = (float)K / ((float)(kmax - kmin) + 1.0f);
                std::vector<u32> cnt(K + 1, 0), wrong(K, 0), foreign(K, 0);
                auto slab_of = [&](u32 id) { return std::min(K - 1, (u32)((float)(keys[id] - kmin) * scale)); };
                for (u32 i = 0; i < n; ++i) cnt[slab_of(want[i])]++;
                u32 pos = 0;
                for (u32 k = 0; k < K; ++k) {
                    for (u32 j = 0; j < cnt[k]; ++j, ++pos) {
                        const u32 got = pl[first + pos];
                        if (got != want[pos]) wrong[k]++;
                        if (got >= n || slab_of(got) != k) foreign[k]++;
                    }
                    printf("  slab %u: %u entries, %u wrong positions, %u entries not of this slab\n", k, cnt[k], wrong[k], foreign[k]);
                }
                {   // the level-1 scatter's output: does slab 0's region of scratch hold exactly slab 0's words?
                    std::vector<u64> sc(first + n);
                    hipMemcpy(sc.data(), d_s, (first + n) * 8, hipMemcpyDeviceToHost);
                    u32 miss = 0, dup = 0; std::vector<int> seen(n, 0);
                    for (u32 j = 0; j < cnt[0]; ++j) { const u32 id = (u32)sc[first + j]; if (id < n && slab_of(id) == 0) { if (seen[id]++) ++dup; } else ++miss; }
                    printf("  scratch, slab 0 region: %u foreign / garbage words, %u duplicates; word of id want[1]=%u present: %d\n", miss, dup, want[1], seen[want[1]]);
                }
                printf("  first 12 got :"); for (u32 i = 0; i < 12; ++i) printf(" %u", pl[first + i]); printf("\n  first 12 want:");
                for (u32 i = 0; i < 12; ++i) printf(" %u", want[i]); printf("\n");
            }
        }
    }
    printf("n = %u, K = %u, slabs 0x%x%s: %d of 20 runs wrong\n", n, K, mask, plain ? " (plain)" : "", bad_runs);
    return 0;
}
