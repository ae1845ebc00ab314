// tile_sort.h -- step 5 of the tile-binned lists: one tile's (depth key, id) words sorted into its slice of point_list.
// Used by blend.hip: since round 4 the tile's own blend_fwd workgroup sorts the list before it stages it -- the sort is a chain of
// LDS round trips, the blend is issue bound, so the two overlap across the workgroups of a CU, and the step has one launch fewer
// (until then binning.hip launched these functions as a kernel of their own).
#pragma once
#include "binning_common.h"

#ifndef TSORT_WAVE
#define TSORT_WAVE 512                                      // longest list the one-wave-per-tile sort takes (round 3: 1024 before -- half
#endif                                                      // the registers and unrolled passes per wave, 20 KB of LDS per workgroup instead of 40)
__device__ __forceinline__ void lds_wave_sync() { __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup"); __builtin_amdgcn_wave_barrier(); }

// Bitonic network in its all-ascending form (first sub-stage of a merge pairs i with its mirror image in the block, the
// others are plain half-cleaners): with every compare-exchange ascending, the list may be thought of as padded with
// +infinity up to the next power of two and a pair whose upper index is >= n is simply skipped.

// ONE WAVE sorts n <= 64 * PER (key, id) words.  One-pass bucket sort: depth keys of one tile are spread fairly evenly
// between the tile's nearest and farthest splat, so with as many buckets as entries (bucket = floor((key - min) * n /
// (max - min + 1)), monotone in the key) a bucket holds one or two entries; an entry's rank is its bucket's start + the
// number of entries of the same bucket that compare below it on the full word.  ~100 instructions per 64 entries against
// ~2600 per 512-entry list for a bitonic network, which remains the fallback for lists whose keys cluster (more than
// TS_BUCKET_MAX entries in one bucket), so the worst case stays O(n log^2 n).  The words are unique (ids are), hence the
// result does not depend on the order in which the LDS atomics of the counting pass retire.
// e[r] = word of entry r * 64 + lane (all ones beyond n).  t / cnt / bid: this wave's LDS (64 * PER entries each).  The
// caller's workgroup may hold several waves, each sorting its own list: only wave-level synchronisation is used.
#define TS_BUCKET_MAX 24

template <int PER>
__device__ __forceinline__ void wave_sort_words(const u64 (&e)[PER], u32 n, u32 kmin, u32 kmax, u32* __restrict__ out, u64* t, u32* cnt,
                                                u32* mirror = nullptr /* LDS copy of the sorted ids (outside t / cnt), or null */)
{
    // LDS per list: the words (8 B per entry) and the bucket counters, two 16-bit counters per word (a count or an offset
    // is at most n <= 64 * PER <= 65535); an entry's bucket is recomputed from its key where it is needed again instead
    // of being stored.  10 KB per 1024-entry list instead of 14: 16 instead of 11 one-wave workgroups per CU.
    const u32 lane = threadIdx.x & 63;
    const u32 rounds = (n + 63) >> 6;
    const u32 nb = n, nw = (nb + 1) >> 1;                       // buckets, packed counter words
    const float scale = (float)nb / ((float)(kmax - kmin) + 1.0f);
    auto bucket_of = [&](u32 key) -> u32 { return min(nb - 1, (u32)((float)(key - kmin) * scale)); };
    auto half = [](u32 w, u32 b) -> u32 { return (w >> ((b & 1u) * 16u)) & 0xFFFFu; };
    for (u32 i = lane; i < nw; i += 64) cnt[i] = 0u;
    lds_wave_sync();
    u32 bk[PER], rk[PER];
#pragma unroll
    for (u32 r = 0; r < PER; ++r) {
        bk[r] = 0; rk[r] = 0;
        if (r < rounds && r * 64 + lane < n) {
            bk[r] = bucket_of((u32)(e[r] >> 32));
            rk[r] = half(atomicAdd(&cnt[bk[r] >> 1], 1u << ((bk[r] & 1u) * 16u)), bk[r]);
        }
    }
    lds_wave_sync();
    // exclusive scan of the bucket counts (in place) and the fullest bucket
    u32 carry = 0, maxc = 0;
    for (u32 base = 0; base < nw; base += 64) {
        const u32 w = (base + lane < nw) ? cnt[base + lane] : 0u;
        const u32 c0 = w & 0xFFFFu, c1 = w >> 16;
        maxc = max(maxc, max(c0, c1));
        const u32 incl = wave_incl_scan(c0 + c1);
        const u32 ex = carry + incl - (c0 + c1);
        if (base + lane < nw) cnt[base + lane] = ex | ((ex + c0) << 16);
        carry += (u32)__builtin_amdgcn_readlane((int)incl, 63);
    }
    maxc = wave_max(maxc);
    lds_wave_sync();
    if (maxc > TS_BUCKET_MAX) {                               // clustered keys: bitonic network on the whole list
#pragma unroll
        for (u32 r = 0; r < PER; ++r)
            if (r < rounds && r * 64 + lane < n) t[r * 64 + lane] = e[r];
        u32 N = 2; while (N < n) N <<= 1;
        lds_wave_sync();
        for (u32 k = 2; k <= N; k <<= 1) {
            const u32 hk = k >> 1;
            for (u32 i = lane; i < (N >> 1); i += 64) {       // mirror stage
                const u32 blk = i / hk, r = i - blk * hk;
                const u32 a = blk * k + r, b = blk * k + (k - 1 - r);
                if (b < n) { const u64 x = t[a], y = t[b]; if (y < x) { t[a] = y; t[b] = x; } }
            }
            lds_wave_sync();
            for (u32 j = k >> 2; j >= 1; j >>= 1) {
                for (u32 i = lane; i < (N >> 1); i += 64) {
                    const u32 a = (i / j) * (2 * j) + (i & (j - 1)), b = a + j;
                    if (b < n) { const u64 x = t[a], y = t[b]; if (y < x) { t[a] = y; t[b] = x; } }
                }
                lds_wave_sync();
            }
        }
        for (u32 i = lane; i < n; i += 64) { out[i] = (u32)t[i]; if (mirror) mirror[i] = (u32)t[i]; }
        return;
    }
    // entries grouped by bucket (order inside a bucket = the order the atomics retired in: irrelevant, see above)
#pragma unroll
    for (u32 r = 0; r < PER; ++r)
        if (r < rounds && r * 64 + lane < n) t[half(cnt[bk[r] >> 1], bk[r]) + rk[r]] = e[r];
    lds_wave_sync();
    for (u32 p = lane; p < n; p += 64) {
        const u64 x = t[p];
        const u32 b = bucket_of((u32)(x >> 32));
        const u32 bs = half(cnt[b >> 1], b), be = (b + 1 < nb) ? half(cnt[(b + 1) >> 1], b + 1) : n;
        u32 rank = 0;
        for (u32 q = bs; q < be; ++q) rank += (t[q] < x) ? 1u : 0u;
        out[bs + rank] = (u32)x;
        if (mirror) mirror[bs + rank] = (u32)x;
    }
}

// One WAVE for a list of up to TSORT_WAVE entries: no workgroup barriers, only wave-level synchronisation (blend_fwd calls it from
// wave 0 of the tile's workgroup; the other three waves wait at the barrier behind the sort).
#define TS_PER (TSORT_WAVE / 64)
template <int PER>
__device__ __forceinline__ void sort_wave_list(u32 n, u32 start, const WordSrc words_in, u32* __restrict__ point_list, u64* t, u32* cnt,
                                               u32* mirror = nullptr)
{
    const u32 lane = threadIdx.x & 63;
    // the list's (depth key, id) words in one batch of coalesced loads (clamped indices, no branches)
    u64 e[PER];
    u32 kmin = 0xFFFFFFFFu, kmax = 0u;
#pragma unroll
    for (u32 r = 0; r < PER; ++r) {
        const u64 w = words_in[start + min(r * 64 + lane, n - 1)];
        const bool valid = r * 64 + lane < n;
        const u32 key = (u32)(w >> 32);
        e[r] = valid ? w : ~0ull;
        kmin = min(kmin, valid ? key : 0xFFFFFFFFu); kmax = max(kmax, valid ? key : 0u);
    }
    wave_sort_words<PER>(e, n, wave_min(kmin), wave_max(kmax), point_list + start, t, cnt, mirror);
}
__device__ __forceinline__ void sort_wave_role(const uint4 desc, const WordSrc words_in, u32* __restrict__ point_list, u64* t, u32* cnt,
                                               u32* mirror = nullptr)
{
    const u32 n = desc.z, start = desc.y;
    if (n == 0 || n > TSORT_WAVE) return;
    if (n == 1) { if ((threadIdx.x & 63) == 0) point_list[start] = words_in.ids[start]; return; }
    // every pass of the sort is unrolled over the entries a lane CAN hold: a list of half the capacity takes the half-size
    // instance (wave-uniform choice; the median tile of the bench scene holds 254 entries)
    if (n <= TSORT_WAVE / 2) sort_wave_list<TS_PER / 2>(n, start, words_in, point_list, t, cnt, mirror);
    else sort_wave_list<TS_PER>(n, start, words_in, point_list, t, cnt, mirror);
}

// The same bucket sort run by a whole 256-thread workgroup on n <= 256 * PER words (lists of 513..2048 entries: the bulk
// of the tiles of a dense scene, e.g. 718 instances per tile on average at scale multiplier 1.0).  e[r] = word of entry
// r * 256 + tid.  s_tmp: 8 words of scratch.
template <int PER>
__device__ __forceinline__ void block_sort_words(const u64 (&e)[PER], u32 n, u32 kmin, u32 kmax, u32* __restrict__ out, u64* t, u32* cnt,
                                                 u32* s_tmp)
{
    // same scheme as wave_sort_words on 256 threads: packed 16-bit bucket counters (n <= TSORT_BLOCK), buckets
    // recomputed from the key in the ranking pass
    const u32 tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const u32 nb = n, nw = (nb + 1) >> 1;
    const float scale = (float)nb / ((float)(kmax - kmin) + 1.0f);
    auto bucket_of = [&](u32 key) -> u32 { return min(nb - 1, (u32)((float)(key - kmin) * scale)); };
    auto half = [](u32 w, u32 b) -> u32 { return (w >> ((b & 1u) * 16u)) & 0xFFFFu; };
    for (u32 i = tid; i < nw; i += 256) cnt[i] = 0u;
    __syncthreads();
    u32 bk[PER], rk[PER];
#pragma unroll
    for (u32 r = 0; r < PER; ++r) {
        bk[r] = 0; rk[r] = 0;
        if (r * 256 + tid < n) {
            bk[r] = bucket_of((u32)(e[r] >> 32));
            rk[r] = half(atomicAdd(&cnt[bk[r] >> 1], 1u << ((bk[r] & 1u) * 16u)), bk[r]);
        }
    }
    __syncthreads();
    // exclusive scan of the bucket counts: thread tid owns the contiguous counter words [w0, w1)
    const u32 per = (nw + 255) >> 8, w0 = min(nw, tid * per), w1 = min(nw, w0 + per);
    u32 sum = 0, maxc = 0;
    for (u32 c = w0; c < w1; ++c) { const u32 v = cnt[c]; sum += (v & 0xFFFFu) + (v >> 16); maxc = max(maxc, max(v & 0xFFFFu, v >> 16)); }
    const u32 incl = wave_incl_scan(sum);
    maxc = wave_max(maxc);
    if (lane == 63) { s_tmp[wave] = incl; s_tmp[4 + wave] = maxc; }
    __syncthreads();
    u32 run = incl - sum;
    for (u32 w = 0; w < wave; ++w) run += s_tmp[w];
    maxc = max(max(s_tmp[4], s_tmp[5]), max(s_tmp[6], s_tmp[7]));
    for (u32 c = w0; c < w1; ++c) { const u32 v = cnt[c]; const u32 lo = v & 0xFFFFu; cnt[c] = run | ((run + lo) << 16); run += lo + (v >> 16); }
    __syncthreads();
    if (maxc > TS_BUCKET_MAX) {                               // clustered keys: bitonic network in LDS
#pragma unroll
        for (u32 r = 0; r < PER; ++r)
            if (r * 256 + tid < n) t[r * 256 + tid] = e[r];
        u32 N = 2; while (N < n) N <<= 1;
        __syncthreads();
        for (u32 k = 2; k <= N; k <<= 1) {
            const u32 hk = k >> 1;
            for (u32 i = tid; i < (N >> 1); i += 256) {
                const u32 blk = i / hk, r = i - blk * hk;
                const u32 a = blk * k + r, b = blk * k + (k - 1 - r);
                if (b < n) { const u64 x = t[a], y = t[b]; if (y < x) { t[a] = y; t[b] = x; } }
            }
            __syncthreads();
            for (u32 j = k >> 2; j >= 1; j >>= 1) {
                for (u32 i = tid; i < (N >> 1); i += 256) {
                    const u32 a = (i / j) * (2 * j) + (i & (j - 1)), b = a + j;
                    if (b < n) { const u64 x = t[a], y = t[b]; if (y < x) { t[a] = y; t[b] = x; } }
                }
                __syncthreads();
            }
        }
        for (u32 i = tid; i < n; i += 256) out[i] = (u32)t[i];
        return;
    }
#pragma unroll
    for (u32 r = 0; r < PER; ++r)
        if (r * 256 + tid < n) t[half(cnt[bk[r] >> 1], bk[r]) + rk[r]] = e[r];
    __syncthreads();
    for (u32 p = tid; p < n; p += 256) {
        const u64 x = t[p];
        const u32 b = bucket_of((u32)(x >> 32));
        const u32 bs = half(cnt[b >> 1], b), be = (b + 1 < nb) ? half(cnt[(b + 1) >> 1], b + 1) : n;
        u32 rank = 0;
        for (u32 q = bs; q < be; ++q) rank += (t[q] < x) ? 1u : 0u;
        out[bs + rank] = (u32)x;
    }
}

// Lists of more than TSORT_WAVE entries: a fixed grid of 256-thread workgroups walks the front of the heavy-first
// descriptor list (n_active[1] entries: every long list plus a few of the boundary level).
//   * up to TSORT_BLOCK entries: block_sort_words, the whole workgroup on one list.
//   * up to TSORT_LARGE entries: two levels.  The workgroup cuts the depth range of the list into slabs of ~512 entries
//     (coarse buckets, again monotone in the key), groups the (key, id) words by slab in the global scratch array `scratch`,
//     and its four waves then sort one slab each with wave_sort_words until none is left.  A slab that outgrows a wave's
//     capacity (very uneven depths) sends the whole list to the network below.
//   * beyond that (tens of thousands of splats over ONE tile: a camera far from the scene, adversarial inputs), or as that
//     fallback: the bitonic network in global memory, loads and stores at agent scope so that the waves of the workgroup
//     see each other's exchanges across the barriers.  Slow, correct, never on the path of an ordinary frame.
#ifndef TSORT_BLOCK
#define TSORT_BLOCK 2048
#endif
#define TSORT_LARGE 65536                                    // (16384 / 64 slabs until round 4: the global-memory network above that is 55 x
#define TS_SLABS_MAX 256                                     //  slower per entry than the LDS sorts, tools/ubench/sort_rate.hip)

// One list of at most 256 * BP entries by the whole 256-thread workgroup: its words in one batch of coalesced loads, then the
// block-wide bucket sort (every thread calls it; it starts and ends on barriers).
template <int BP>
__device__ __forceinline__ void sort_one_block(const u32 n, const u32 start, const WordSrc words_in, u32* __restrict__ point_list,
                                               u64* t_all, u32* cnt_all, u32* s_red)
{
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    u64 e[BP];
    u32 lo = 0xFFFFFFFFu, hi = 0u;
#pragma unroll
    for (u32 r = 0; r < BP; ++r) {
        const u64 w = words_in[start + min(r * 256 + (u32)tid, n - 1)];
        const bool valid = r * 256 + tid < n;
        const u32 key = (u32)(w >> 32);
        e[r] = valid ? w : ~0ull;
        lo = min(lo, valid ? key : 0xFFFFFFFFu); hi = max(hi, valid ? key : 0u);
    }
    lo = wave_min(lo); hi = wave_max(hi);
    __syncthreads();                                  // the previous list's LDS state is no longer in use
    if (lane == 0) { s_red[wave] = lo; s_red[4 + wave] = hi; }
    __syncthreads();
    lo = min(min(s_red[0], s_red[1]), min(s_red[2], s_red[3]));
    hi = max(max(s_red[4], s_red[5]), max(s_red[6], s_red[7]));
    __syncthreads();                                  // s_red is reused as scratch below
    block_sort_words<BP>(e, n, lo, hi, point_list + start, t_all, cnt_all, s_red);
}

#define TS_LDS_WORDS TSORT_BLOCK                             // u64 t_all[TSORT_BLOCK] + u32 cnt_all[TSORT_BLOCK / 2]: 20 KB
struct TileSortLds {                                         // small state of the long-list paths
    u32 s_red[8];
    u32 slab_cnt[TS_SLABS_MAX + 1], slab_start[TS_SLABS_MAX + 1], s_next, s_bad;
};
// f(word) for every entry of the list, 256 threads, EIGHT entries per thread in flight: the ids of a batch are requested
// together, then their keys (a thread's plain loop -- id, then key, then the use, entry after entry -- was two dependent memory
// latencies per entry: ~70 of them in series per pass for a 3000-entry list, which is what the long-list sort's time was).
template <typename F>
__device__ __forceinline__ void for_each_word_256(const WordSrc words_in, const u32 start, const u32 n, const u32 tid, F f)
{
    for (u32 base = tid; base < n; base += 256u * 8u) {
        u32 id[8], key[8];
#pragma unroll
        for (u32 j = 0; j < 8; ++j) id[j] = words_in.ids[start + min(base + 256u * j, n - 1)];
#pragma unroll
        for (u32 j = 0; j < 8; ++j) key[j] = words_in.keys[id[j]];
#pragma unroll
        for (u32 j = 0; j < 8; ++j)
            if (base + 256u * j < n) f(((u64)key[j] << 32) | (u64)id[j]);
    }
}
// The same over words whose keys were stashed by an earlier pass (slab_prepare): id and key of an entry are two INDEPENDENT coalesced
// loads -- one memory round trip per batch instead of two, and no gather.  The stash was written by other waves of this workgroup:
// read at agent scope (L2), like the scratch words of level 2.
template <typename F>
__device__ __forceinline__ void for_each_stashed_word_256(const WordSrc words_in, const u32* __restrict__ stash, const u32 start, const u32 n,
                                                          const u32 tid, F f)
{
    for (u32 base = tid; base < n; base += 256u * 8u) {
        u32 id[8], key[8];
#pragma unroll
        for (u32 j = 0; j < 8; ++j) {
            const u32 i = min(base + 256u * j, n - 1);
            id[j] = words_in.ids[start + i];
            key[j] = __hip_atomic_load(&stash[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
#pragma unroll
        for (u32 j = 0; j < 8; ++j)
            if (base + 256u * j < n) f(((u64)key[j] << 32) | (u64)id[j]);
    }
}
// ---- the two-level sort of a list of TSORT_BLOCK < n <= TSORT_LARGE entries, one function per level.  (Round 5 ran level 2 lazily
// from blend_fwd -- a dense tile's walk ends after a fraction of its list and nothing behind the tile's deepest contributor needs an
// ORDER -- and measured it slower: blend.hip, profiles/r05/ab_dense.txt.)
// Level 1 (every thread of the workgroup calls it): key range, slab of every entry, slab starts, the (key, id) words grouped by slab
// in the global scratch array (order inside a slab arbitrary).  Returns the number of slabs K, or 0 if the list has to take the
// network instead (a slab outgrows a wave: very uneven depths).  Ends on a barrier; L.slab_start[0..K] stays valid afterwards.
__device__ __forceinline__ u32 slab_prepare(const u32 n, const u32 start, const WordSrc words_in, u64* __restrict__ scratch, TileSortLds& L)
{
    u32* const s_red = L.s_red; u32* const slab_cnt = L.slab_cnt; u32* const slab_start = L.slab_start;
    u32& s_next = L.s_next; u32& s_bad = L.s_bad;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const u32 K = min((u32)TS_SLABS_MAX, (n + TSORT_WAVE / 2 - 1) / (TSORT_WAVE / 2));
    u32 kmin = 0xFFFFFFFFu, kmax = 0u;
    // The first pass gathers every entry's key by its id (two dependent round trips per batch, 64 lines per wave load) and leaves the keys
    // in the SECOND half of the tile's own slice of the words buffer -- its first half holds the unsorted ids, the second is free until
    // the compacted positions are written after the sort (blend.hip) -- so that the two passes below read id and key side by side.
    // (Round 5, timing build: level 1 was three quarters of what a dense scene's sort costs, and that 43 % / 61 % of blend_fwd at 1800 /
    // 4000 entries per tile: profiles/r05/ab_dense.txt 9.)
    u32* const stash = const_cast<u32*>(words_in.ids) + start + n;
    for (u32 base = tid; base < n; base += 256u * 8u) {
        u32 id[8], key[8];
#pragma unroll
        for (u32 j = 0; j < 8; ++j) id[j] = words_in.ids[start + min(base + 256u * j, n - 1)];
#pragma unroll
        for (u32 j = 0; j < 8; ++j) key[j] = words_in.keys[id[j]];
#pragma unroll
        for (u32 j = 0; j < 8; ++j)
            if (base + 256u * j < n) { kmin = min(kmin, key[j]); kmax = max(kmax, key[j]); stash[base + 256u * j] = key[j]; }
    }
    kmin = wave_min(kmin); kmax = wave_max(kmax);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // the stash is complete (read back by the other waves through the L2)
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    __syncthreads();                                  // the previous list's state is no longer in use
    if (lane == 0) { s_red[wave] = kmin; s_red[4 + wave] = kmax; }
    for (u32 i = tid; i <= TS_SLABS_MAX; i += 256) slab_cnt[i] = 0u;
    if (tid == 0) { s_next = 0u; s_bad = 0u; }
    __syncthreads();
    kmin = min(min(s_red[0], s_red[1]), min(s_red[2], s_red[3]));
    kmax = max(max(s_red[4], s_red[5]), max(s_red[6], s_red[7]));
    const float scale = (float)K / ((float)(kmax - kmin) + 1.0f);
    for_each_stashed_word_256(words_in, stash, start, n, (u32)tid, [&](const u64 w) {
        atomicAdd(&slab_cnt[min(K - 1, (u32)((float)((u32)(w >> 32) - kmin) * scale))], 1u);
    });
    __syncthreads();
    if (tid == 0) {
        u32 run = 0, bad = 0;
        for (u32 k = 0; k < K; ++k) { slab_start[k] = run; run += slab_cnt[k]; bad |= (slab_cnt[k] > TSORT_WAVE) ? 1u : 0u; slab_cnt[k] = 0u; }
        slab_start[K] = run; s_bad = bad;
    }
    __syncthreads();
    if (s_bad != 0u) return 0u;
    // ---- words grouped by slab in the scratch array (order inside a slab arbitrary)
    for_each_stashed_word_256(words_in, stash, start, n, (u32)tid, [&](const u64 w) {
        const u32 k = min(K - 1, (u32)((float)((u32)(w >> 32) - kmin) * scale));
        scratch[start + slab_start[k] + atomicAdd(&slab_cnt[k], 1u)] = w;
    });
    // The words are read back by OTHER WAVES OF THIS WORKGROUP only: same CU, same L1 (write-through), same L2, and the
    // reads of level 2 go to the L2 (agent-scope loads).  Completed stores + a workgroup-scope fence are enough.  Until the end
    // of round 4 this was __threadfence(): an agent-scope fence, i.e. `buffer_wbl2` -- a write-back of the whole L2's
    // dirty lines by every long-list workgroup (lists just above 2048 entries sorted 15 x slower per entry than lists
    // just below, tools/ubench/sort_rate.hip; blend_fwd 1.07 ms at sm 2.0).
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    __syncthreads();
    return K;
}
// Level 2, ONE slab by ONE wave (t / cnt: this wave's TSORT_WAVE words and TSORT_WAVE / 2 counter words of LDS).  An empty slab is
// skipped; a one-entry slab takes the general path.  Until round 4 both were early exits -- `return` for the wave, so that with
// depth-clustered lists every wave could leave on an empty slab before the last slabs were drawn and their part of point_list kept
// whatever the buffer held before (found by tools/soak.py: a camera drifted into the scene), and, once that was a `continue`,
// the one-entry special case `if (m == 1) { if (lane == 0) store; continue; }` came out of the compiler re-sorting slab 0 with an
// entry missing (tools/ubench/sort_slabs.hip).  No early exit, no special case:
__device__ __forceinline__ void slab_sort_wave(const u32 k, const u32 start, const u64* __restrict__ scratch, u32* __restrict__ point_list,
                                               u64* t, u32* cnt, const TileSortLds& L)
{
    const int lane = threadIdx.x & 63;
    const u32 s0 = L.slab_start[k], m = L.slab_start[k + 1] - s0;
    if (m != 0) {
        u64 e[TS_PER];
        u32 lo = 0xFFFFFFFFu, hi = 0u;
#pragma unroll
        for (u32 r = 0; r < TS_PER; ++r) {
            // written by other waves of this workgroup in level 1: read at agent scope (L2)
            const u64 raw = __hip_atomic_load(&scratch[start + s0 + min(r * 64 + (u32)lane, m - 1)], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            const u32 wkey = (u32)(raw >> 32);
            const bool valid = r * 64 + lane < m;
            e[r] = valid ? raw : ~0ull;
            lo = min(lo, valid ? wkey : 0xFFFFFFFFu); hi = max(hi, valid ? wkey : 0u);
        }
        wave_sort_words<TS_PER>(e, m, wave_min(lo), wave_max(hi), point_list + start + s0, t, cnt);
    }
}
// The bitonic network in global memory (more than TSORT_LARGE entries, or the fallback of the two-level sort): loads and stores at
// agent scope so that the waves of the workgroup see each other's exchanges across the barriers.
__device__ __forceinline__ void sort_list_network(const u32 n, const u32 start, const WordSrc words_in, u64* __restrict__ scratch,
                                                  u32* __restrict__ point_list)
{
    const int tid = threadIdx.x;
    u32 N = 2; while (N < n) N <<= 1;
    __syncthreads();
    for (u32 i = tid; i < n; i += 256) scratch[start + i] = words_in[start + i];
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");      // (read back by this workgroup's waves at agent scope: see above)
    __syncthreads();
    u64* gsm = scratch + start;
    auto ld = [&](u32 i) -> u64 { return __hip_atomic_load(&gsm[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); };
    auto st = [&](u32 i, u64 v) { __hip_atomic_store(&gsm[i], v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); };
    for (u32 k = 2; k <= N; k <<= 1) {
        const u32 hk = k >> 1;
        for (u32 i = tid; i < (N >> 1); i += 256) {
            const u32 blk = i / hk, r = i - blk * hk;
            const u32 a = blk * k + r, b = blk * k + (k - 1 - r);
            if (b < n) { const u64 x = ld(a), y = ld(b); if (y < x) { st(a, y); st(b, x); } }
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        for (u32 j = k >> 2; j >= 1; j >>= 1) {
            for (u32 i = tid; i < (N >> 1); i += 256) {
                const u32 a = (i / j) * (2 * j) + (i & (j - 1)), b = a + j;
                if (b < n) { const u64 x = ld(a), y = ld(b); if (y < x) { st(a, y); st(b, x); } }
            }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
        }
    }
    for (u32 i = tid; i < n; i += 256) point_list[start + i] = (u32)ld(i);
}
// One list of more than TSORT_WAVE entries, sorted completely, by a whole 256-thread workgroup (every thread calls it).
__device__ __forceinline__ void sort_list_block(const uint4 desc, const WordSrc words_in, u64* __restrict__ scratch,
                                                u32* __restrict__ point_list, u64* t_all, u32* cnt_all, TileSortLds& L)
{
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const u32 n = desc.z, start = desc.y;
    if (n <= TSORT_WAVE) return;                             // (the one-wave sort's sizes: sort_wave_role)
    if (n <= TSORT_BLOCK) { sort_one_block<TSORT_BLOCK / 256>(n, start, words_in, point_list, t_all, cnt_all, L.s_red); return; }
    const u32 K = (n > TSORT_LARGE) ? 0u : slab_prepare(n, start, words_in, scratch, L);
    if (K == 0u) { sort_list_network(n, start, words_in, scratch, point_list); return; }
    for (;;) {                                               // a wave per slab, drawn from a counter
        u32 k = 0;
        if (lane == 0) k = atomicAdd(&L.s_next, 1u);
        k = (u32)__builtin_amdgcn_readfirstlane((int)k);
        if (k >= K) break;
        slab_sort_wave(k, start, scratch, point_list, t_all + wave * TSORT_WAVE, cnt_all + wave * (TSORT_WAVE / 2), L);
    }
}
