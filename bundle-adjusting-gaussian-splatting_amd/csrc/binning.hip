// binning.hip -- K2..K5, tile-binned: per-tile depth-ordered instance lists without a global sort.
//
// The stock pipeline radix-sorts I (tile | depth) keys; the first version here sorted the P Gaussians by depth, emitted in
// that order and stable-sorted the I instances by tile (sort.hip: 20 launches of 5-22 us each, every one of them bounded by
// launch / dependency latency, 0.20 ms of a 1.0 ms step).  The same lists come out of FIVE launches that never order
// anything globally:
//   1. tile_count      one workgroup per block of ~P/256 consecutive Gaussians counts its instances per tile in LDS (packed
//                      16-bit counters, integer LDS atomics: counts do not depend on arrival order) and writes one row of
//                      the (block, tile) count matrix; also each Gaussian's instance offset inside its block
//   2. tile_prefix     per tile: exclusive prefix of the matrix column over the blocks, and the tile's total
//   3. ranges_order    ONE workgroup: exclusive scan of the tile totals = the tile ranges and the instance count; scan of
//                      the block totals; the heavy-first tile descriptor list of the blend launches
//   4. emit_binned     same workgroups as 1.: instance -> slot from a per-tile cursor in LDS (range start + column prefix,
//                      bumped by a returning LDS atomic: arbitrary order inside a (block, tile) group), writes the Gaussian id
//   5. tile_sort       one WAVE per tile sorts its (depth key, id) words in LDS (one-pass bucket sort; bitonic network for
//                      clustered keys and for lists of more than 512 entries)
// (depth key, id) is a strict total order, so the sorted list is unique: bit-identical to the stable sort of the 64-bit
// keys whatever order the atomics of step 4 retired in, and bitwise reproducible.  No global atomics anywhere.
// Limits: tiles * 2 bytes of LDS (<= 32768 tiles: up to 4K images) and <= 65535 Gaussians per block (P <= 16.7 M);
// beyond them, or on request (BagsSettings.binning), api.hip falls back to the radix path of sort.hip.
#include "bags_common.h"
#include "binning_common.h"
#include "tile_sort.h"

// Shared body of tile_count (EMIT = false) and emit_binned (EMIT = true): the block's Gaussians, thread by thread in
// contiguous runs; small rectangles by their own lane, large ones by the whole wave.
template <bool EMIT>
__device__ __forceinline__ u32 walk_block(u32* cnt, int P, int per_block, int grid_x, const uint2* __restrict__ rect,
                                          const u32* __restrict__ tiles_touched, const u64* __restrict__ keep,
                                          u32* __restrict__ ids, const int y_lo = 0, const int y_hi = 0x7FFF)
{
    const int lane = threadIdx.x & 63;
    const int per_thread = per_block / BIN_THREADS;
    const long long g0 = (long long)blockIdx.x * per_block + (long long)threadIdx.x * per_thread;
    u32 mine = 0;
    for (int k0 = 0; k0 < per_thread; k0 += 4) {            // uniform trip counts: the ballots below need every lane
        // four Gaussians' counts, rectangles and masks requested together, unconditionally (clamped index): one memory
        // round trip per batch instead of three per Gaussian
        u32 ntv[4]; uint2 rcv[4]; u64 kpv[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const long long g = g0 + k0 + u;
            const long long gc = g < P ? g : (long long)P - 1;
            ntv[u] = tiles_touched[gc]; rcv[u] = rect[gc]; kpv[u] = keep[gc];
            if (!(g < P) || k0 + u >= per_thread) ntv[u] = 0u;
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const long long g = g0 + k0 + u;
            const u32 nt = ntv[u]; const uint2 rc = rcv[u];
            const int w = (int)(rc.y & 0xFFFF) - (int)(rc.x & 0xFFFF), h = (int)(rc.y >> 16) - (int)(rc.x >> 16);
            mine += nt;                                     // a large rectangle emits every tile: nt is its area
            if (nt > 0) {                                   // (the id alone is emitted: WordSrc, binning_common.h)
                if (rect_small(w, h)) walk_mask<EMIT>(cnt, rc, kpv[u], grid_x, (u32)g, ids, y_lo, y_hi);
                else if (nt <= BIN_COOP) walk_rect<EMIT>(cnt, rc, grid_x, lane, false, (u32)g, ids, y_lo, y_hi);
            }
            u64 big = __ballot(nt > BIN_COOP);              // never a small rectangle (at most 64 tiles)
            while (big) {
                const int src = __ffsll((long long)big) - 1;
                big &= big - 1;
                const uint2 brc = make_uint2((u32)__shfl((int)rc.x, src), (u32)__shfl((int)rc.y, src));
                walk_rect<EMIT>(cnt, brc, grid_x, lane, true, (u32)__shfl((int)(u32)g, src), ids, y_lo, y_hi);
            }
        }
    }
    return mine;
}

// (tile_count_kernel, a launch of its own until round 3, is now part of K1: preprocess_fwd_count_kernel in preprocess_fwd.hip
// runs walk_mask / walk_rect<false> on every Gaussian it has just projected and leaves the same matrix row, block total and
// block-local offsets behind.)

// ------------------------------------------------------------------------------------------------ 2. tile_prefix
// thread = (packed word = two tiles, group of 8 blocks); the thread's 8 words stay in registers between the summing pass
// and the writing pass, so the matrix is read once and the prefix written once.  (First version: 64 words x 4 quarters of
// 64 blocks per 256-thread workgroup -- 64 workgroups in all, 9.0 us of load latency on a quarter of the CUs.)
// The blocks are taken in the order (b % 8, b / 8), not 0, 1, 2, ...: emit_binned's workgroup b runs on XCD b % 8, and the
// eight L2s do not merge partial lines with one another.  With the blocks in id order every 128-byte line of the list
// collected 8-byte words from all eight XCDs and went to memory up to eight times (PMC: 66 MB written for 18.6 MB of
// output); now a tile's list is eight stretches, each written through ONE L2.  The order inside a tile's list is free: the
// per-tile sort follows.
__device__ __forceinline__ int prefix_block(int seq) { return (seq & 31) * 8 + (seq >> 5); }
#define PFX_WORDS 32                                        // packed words (64 tiles) per workgroup
#define PFX_GROUPS 32                                       // groups of 8 blocks: 1024 threads = 32 words x 32 groups
__global__ void __launch_bounds__(1024)
tile_prefix_kernel(const u32* __restrict__ cnt_rows, int B, int T, int T2, u32* __restrict__ pre, u32* __restrict__ tile_total,
                   u32* __restrict__ tile_lstart, u32* __restrict__ group_total, u32* __restrict__ count_slot, u32* host_count)
{
    __shared__ u32 gsum[PFX_GROUPS][PFX_WORDS][2];
    const int wl = threadIdx.x & (PFX_WORDS - 1), grp = threadIdx.x / PFX_WORDS;
    const int w = blockIdx.x * PFX_WORDS + wl;               // packed word = tiles 2w, 2w + 1
    u32 v[8];
    u32 s0 = 0, s1 = 0;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const int b = prefix_block(grp * 8 + i);
        v[i] = (w < T2 && b < B) ? cnt_rows[(size_t)b * T2 + w] : 0u;
    }
#pragma unroll
    for (int i = 0; i < 8; ++i) { s0 += v[i] & 0xFFFFu; s1 += v[i] >> 16; }
    gsum[grp][wl][0] = s0; gsum[grp][wl][1] = s1;
    __syncthreads();
    u32 r0 = 0, r1 = 0;
    for (int p = 0; p < grp; ++p) { r0 += gsum[p][wl][0]; r1 += gsum[p][wl][1]; }
    const int t0 = 2 * w;
    if (w < T2) {
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int b = prefix_block(grp * 8 + i);
            if (b < B) {
                u32* dst = pre + (size_t)b * T + t0;
                dst[0] = r0;
                if (t0 + 1 < T) dst[1] = r1;
                r0 += v[i] & 0xFFFFu; r1 += v[i] >> 16;
            }
        }
        if (grp == PFX_GROUPS - 1) {                         // the last group ends on the column totals
            tile_total[t0] = r0;
            if (t0 + 1 < T) tile_total[t0 + 1] = r1;
        }
    }
    // Round 4: the tile ranges without a launch of their own.  The last group (threads 992..1023: lanes 32..63 of wave 15) holds
    // the totals of this workgroup's 64 tiles: their exclusive prefix INSIDE the workgroup and the workgroup's sum go out here;
    // whoever needs a range start adds the exclusive scan of the (at most 512) group sums (group_bases below) -- the emission
    // workgroups do that for themselves, and the descriptor workgroup beside them delivers the instance count.
    if (threadIdx.x >= 1024 - 64) {                          // wave 15, every lane (the scan needs them all)
        const bool lastg = grp == PFX_GROUPS - 1;
        const u32 c0 = lastg ? r0 : 0u, c1 = (lastg && t0 + 1 < T) ? r1 : 0u;   // (w >= T2: every count was read as zero)
        const u32 v2 = c0 + c1;
        const u32 incl = wave_incl_scan(v2);
        const u32 ex = incl - v2;
        if (lastg && w < T2) { tile_lstart[t0] = ex; if (t0 + 1 < T) tile_lstart[t0 + 1] = ex + c0; }
        if ((threadIdx.x & 63) == 63) group_total[blockIdx.x] = incl;
    }
    // where the asynchronous instance count goes (device-visible address of the caller's pinned word, or null): parked in the
    // caller's geometry buffer for the kernel that computes the count
    if (blockIdx.x == 0 && threadIdx.x == 0) *reinterpret_cast<unsigned long long*>(count_slot) = (unsigned long long)(size_t)host_count;
}

// Exclusive scan of the group sums of tile_prefix_kernel into LDS: s_gb[g] = first instance of tile group g (64 tiles),
// s_gb[512] = instance count.  G <= 512.  Every thread of the workgroup calls it (it ends on a barrier).
__device__ __forceinline__ void group_bases(const u32* __restrict__ group_total, int G, u32* s_gb)
{
    if (threadIdx.x < 64) {
        const int lane = threadIdx.x;
        u32 v[8], sum = 0;
#pragma unroll
        for (int k = 0; k < 8; ++k) { const int idx = lane * 8 + k; v[k] = idx < G ? group_total[idx] : 0u; sum += v[k]; }
        const u32 incl = wave_incl_scan(sum);
        u32 run = incl - sum;
#pragma unroll
        for (int k = 0; k < 8; ++k) { const int idx = lane * 8 + k; if (idx < G) s_gb[idx] = run; run += v[k]; }
        if (lane == 63) s_gb[512] = incl;
    }
    __syncthreads();
}

// ------------------------------------------------------------------------------------------------ 3. ranges_order
// One workgroup: tile ranges (exclusive scan of the tile totals), instance count, block bases, and the heavy-first tile
// descriptor list (same ordering rule as tile_order_kernel of sort.hip; see there and blend.hip for why).
#define ORD_LEVELS 64
#define ORD_SUB 32
__device__ __forceinline__ int ord_level(u32 n)
{
    if (n == 0) return ORD_LEVELS - 1;
    const int e = 31 - __clz((int)n);
    const int frac = (e >= 2) ? (int)((n >> (e - 2)) & 3u) : (int)((n << (2 - e)) & 3u);
    return ORD_LEVELS - 2 - min(ORD_LEVELS - 2, e * 4 + frac);
}
__device__ __forceinline__ u32 block_excl_scan_1024(u32 v, u32* s_wave /*[17]*/)
{
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const u32 incl = wave_incl_scan(v);
    __syncthreads();                                         // s_wave free again
    if (lane == 63) s_wave[wave] = incl;
    __syncthreads();
    u32 before = 0;
#pragma unroll
    for (int w = 0; w < 16; ++w) before += (w < wave) ? s_wave[w] : 0u;
    if (threadIdx.x == 1023) s_wave[16] = before + incl;      // grand total
    return before + incl - v;
}
#define ORD_PER_MAX 32                                      // tiles per thread: 1024 x 32 = 32768 tiles at most
__device__ __forceinline__ int ord_pad(int i) { return i + (i >> 5); }
// thread tid owns the contiguous tiles [ta, tb): their counts stay in registers for every pass.  One CU issues this
// kernel's every memory request, and a thread reading or writing its own run puts 64 requests per instruction on that CU's
// path (32 400 tiles: 99 us).  So the wave moves its span of 64 x per tiles in tile order -- 64 consecutive words per
// instruction -- and the lanes pick their runs out of a padded LDS buffer (stride per + per / 32 words: conflict free).
template <int ORD_PER>
__device__ __forceinline__ u32 ord_load_counts(const u32* __restrict__ tile_total, int T, int per, u32* tr, u32 (&cntv)[ORD_PER])
{
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int span0 = wave * 64 * per;
    // (all loads of a batch of eight before the first LDS store: `tr` reaches this function as a plain pointer, the compiler keeps every
    // load behind the store before it, and the batch was eight dependent round trips in the one workgroup the emission launch waits for)
#pragma unroll
    for (int j0 = 0; j0 < ORD_PER; j0 += 8) {                // load order: tile span0 + j * 64 + lane
        u32 v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int j = j0 + u, t = span0 + j * 64 + lane;
            v[u] = (j < per && t < T) ? tile_total[t] : 0u;
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int j = j0 + u;
            if (j < per) tr[ord_pad(j * 64 + lane)] = v[u];
        }
    }
    lds_wave_sync();
    u32 sum = 0;
#pragma unroll
    for (int i = 0; i < ORD_PER; ++i) { cntv[i] = (i < per) ? tr[ord_pad(lane * per + i)] : 0u; sum += cntv[i]; }
    __builtin_amdgcn_sched_barrier(0);
    return sum;
}

// Tile ranges, instance count, block bases.  (The heavy-first descriptor list is built by one extra workgroup of the
// emission launch, build_tile_desc below: nothing needs it before the per-tile sorts, and its scattered 16-byte stores
// through ONE CU were half of this kernel's 13.8 us on the path to the instance count.)
template <int ORD_PER>
__global__ void __launch_bounds__(1024)
ranges_order_kernel(const u32* __restrict__ tile_total, int T, uint2* __restrict__ ranges,
                    const u32* __restrict__ block_total, int B, u32* __restrict__ block_base,
                    u32* __restrict__ num_rendered, u32* host_count)
{
    __shared__ u32 s_wave[17];
    extern __shared__ u32 tr_all[];                          // 16 waves x ORD_TRW(ORD_PER) words: wave-local transposes
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int per = (T + 1023) / 1024;
    u32* tr = tr_all + wave * (64 * ORD_PER + 2 * ORD_PER);
    const int span0 = wave * 64 * per;
    u32 cntv[ORD_PER];
    const u32 sum = ord_load_counts<ORD_PER>(tile_total, T, per, tr, cntv);
    const u32 first = block_excl_scan_1024(sum, s_wave);     // (its barriers also order the two uses of `tr`)
    {
        u32 run = first;
#pragma unroll
        for (int i = 0; i < ORD_PER; ++i)
            if (i < per) { tr[ord_pad(lane * per + i)] = run; run += cntv[i]; }
        __builtin_amdgcn_sched_barrier(0);
        lds_wave_sync();
#pragma unroll 8
        for (int j = 0; j < ORD_PER; ++j) {                  // the range starts go back the way the counts came
            const int t = span0 + j * 64 + lane;
            // (the count is read again, coalesced, rather than kept: 32 more live registers spill at 1024 threads)
            if (j < per && t < T) { const u32 s0 = tr[ord_pad(j * 64 + lane)]; ranges[t] = make_uint2(s0, s0 + tile_total[t]); }
        }
    }
    __syncthreads();
    if (tid == 0) {
        num_rendered[0] = s_wave[16];
        // the caller's pinned host word, written from here: a device-to-host copy of four bytes is a 5 us copy kernel on the
        // stream between this launch and the emission
        if (host_count) __hip_atomic_store(host_count, s_wave[16], __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    }
    // ---- block bases (B <= 256)
    {
        const u32 v = (tid < B) ? block_total[tid] : 0u;
        const u32 ex = block_excl_scan_1024(v, s_wave);
        if (tid < B) block_base[tid] = ex;
    }
}

// The heavy-first descriptor list of the blend launches {tile, first instance, instance count, -} (same ordering rule as
// tile_order_kernel of sort.hip; see there and blend.hip for why): counting sort of the tiles by instance count, by ONE
// workgroup of 1024 threads.  ORD_LEVELS levels, four per octave, ORD_SUB sub-counters per level picked by the THREAD
// (tid & 31): the lanes of a wave hit 32 different counters of a level and a thread's run of neighbouring tiles stays
// together in the list.  (Picked by tile, (t >> 3) & 31, the 64 lanes of a wave shared 8 counters once a thread held 32
// tiles: eight-deep same-address LDS atomics, 99 us at 32 400 tiles.)  The unrolled loops are cut every 8 tiles so that
// the compiler keeps 8, not 32, atomics and their results in flight (124 spilled registers at 32 tiles per thread otherwise).
template <int ORD_PER>
__device__ __forceinline__ void build_tile_desc(const u32* __restrict__ tile_total, const uint2* __restrict__ ranges, int T,
                                                uint4* __restrict__ tile_desc, u32* __restrict__ n_active, u32* tr_all,
                                                u32* s_cur /*[ORD_LEVELS * ORD_SUB]*/, u32* s_wave /*[17]*/,
                                                const u32* __restrict__ lstart = nullptr, const u32* s_gb = nullptr,
                                                uint2* __restrict__ ranges_out = nullptr)
{
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int per = (T + 1023) / 1024;
    const int ta = min(T, tid * per), tb = min(T, ta + per);
    u32* tr = tr_all + wave * (64 * ORD_PER + 2 * ORD_PER);
    u32 cntv[ORD_PER];
    (void)ord_load_counts<ORD_PER>(tile_total, T, per, tr, cntv);
    // range start of the thread's first tile: from the ranges array (ranges_order ran) or group base + start inside the group
    const u32 first = (ta < T) ? (ranges ? ranges[ta].x : s_gb[ta >> 6] + lstart[ta]) : 0u;
    s_cur[tid] = 0; s_cur[tid + 1024] = 0;
    __syncthreads();
    auto counter_of = [&](u32 n) -> int { return ord_level(n) * ORD_SUB + (tid & (ORD_SUB - 1)); };
#pragma unroll
    for (int i = 0; i < ORD_PER; ++i) {
        if (i < per && ta + i < tb) atomicAdd(&s_cur[counter_of(cntv[i])], 1u);
        if ((i & 7) == 7) __builtin_amdgcn_sched_barrier(0);
    }
    __syncthreads();
    {
        const u32 c0 = s_cur[2 * tid], c1 = s_cur[2 * tid + 1];
        const u32 inc = wave_incl_scan(c0 + c1);
        if (lane == 63) s_wave[wave] = inc;
        __syncthreads();
        u32 before = 0;
#pragma unroll
        for (int w = 0; w < 16; ++w) before += (w < wave) ? s_wave[w] : 0u;
        const u32 excl = before + inc - (c0 + c1);
        s_cur[2 * tid] = excl; s_cur[2 * tid + 1] = excl + c0;
        if (2 * tid == (ORD_LEVELS - 1) * ORD_SUB) n_active[0] = excl;                       // everything in front of the empty tiles
        // descriptors [0, n_active[1]) hold every tile of more than TSORT_WAVE instances (whole levels: a few shorter
        // tiles of the boundary level come along and are skipped by the large-list sort)
        if (2 * tid == (ord_level(TSORT_WAVE) + 1) * ORD_SUB) n_active[1] = excl;
    }
    __syncthreads();
    {
        u32 run = first;
#pragma unroll
        for (int i = 0; i < ORD_PER; ++i) {
            if (i < per && ta + i < tb) {
                u32 c = cntv[i];
                asm volatile("" : "+v"(c));                 // recompute the level here: carried over from the counting pass it
                                                            // would be 32 more live registers
                tile_desc[atomicAdd(&s_cur[counter_of(c)], 1u)] = make_uint4((u32)(ta + i), run, c, 0u);
                if (ranges_out) ranges_out[ta + i] = make_uint2(run, run + c);
                run += c;
            }
            if ((i & 7) == 7) __builtin_amdgcn_sched_barrier(0);
        }
    }
}
// the list on its own: when no emission runs (no Gaussians, or none visible)
template <int ORD_PER>
__global__ void __launch_bounds__(1024)
tile_desc_kernel(const u32* __restrict__ tile_total, const uint2* __restrict__ ranges, int T, uint4* __restrict__ tile_desc,
                 u32* __restrict__ n_active)
{
    __shared__ u32 s_cur[ORD_LEVELS * ORD_SUB];
    __shared__ u32 s_wave[17];
    extern __shared__ u32 tr_all[];
    build_tile_desc<ORD_PER>(tile_total, ranges, T, tile_desc, n_active, tr_all, s_cur, s_wave);
}

#ifndef EMIT_BANDS_MAX
#define EMIT_BANDS_MAX 8
#endif
#ifndef EMIT_OPEN_BYTES
#define EMIT_OPEN_BYTES (3u << 20)                          // bytes of partially written lines per XCD that a band of the emission may keep open
#endif
// ------------------------------------------------------------------------------------------------ 4. emit_binned
// The workgroup behind the last block of Gaussians builds the tile descriptor list (build_tile_desc) beside the emission.
template <int ORD_PER>
__global__ void __launch_bounds__(BIN_THREADS)
emit_binned_kernel(int P, int per_block, int grid_x, int T, const uint2* __restrict__ rect,
                   const u32* __restrict__ tiles_touched, const u64* __restrict__ keep, const u32* __restrict__ pre,
                   uint2* __restrict__ ranges, const u32* __restrict__ tile_lstart, const u32* __restrict__ group_total,
                   const u32* __restrict__ depth_key, u64* __restrict__ words, u32 capacity,
                   const u32* __restrict__ tile_total, uint4* __restrict__ tile_desc, u32* __restrict__ n_active,
                   u32* __restrict__ num_rendered, const u32* __restrict__ block_total, int B, u32* __restrict__ block_base,
                   const int deliver_count)
{
    static_assert(BIN_THREADS == 1024, "build_tile_desc is written for 1024 threads");
    extern __shared__ u32 cur[];                             // T slot cursors (the descriptor workgroup: its transposes)
    __shared__ u32 s_gb[513];                                // first instance of every group of 64 tiles; [512] = instance count
    // the first eight tiles' cursor inputs of every thread (all of them up to 8192 tiles) are requested before the group bases -- one
    // memory round trip for both -- and in one batch: as a loop of `load, load, store` with a run-time trip count the cursor
    // initialisation was eight dependent round trips in a 21 us kernel
    const bool emitter = blockIdx.x != gridDim.x - 1;
    const u32* prow = pre + (size_t)blockIdx.x * T;
    u32 ls0[8], pr0[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) {
        const int t = threadIdx.x + u * BIN_THREADS;
        ls0[u] = (emitter && t < T) ? tile_lstart[t] : 0u;
        pr0[u] = (emitter && t < T) ? prow[t] : 0u;
    }
    group_bases(group_total, (T + 63) >> 6, s_gb);
    if (blockIdx.x == gridDim.x - 1) {
        // the workgroup behind the last block of Gaussians: heavy-first descriptor list, the ranges array, the instance count
        // (device word + the caller's pinned host word), block bases of the record slots
        __shared__ u32 s_cur[ORD_LEVELS * ORD_SUB];
        __shared__ u32 s_wave[17];
        if (threadIdx.x == 0) {
            num_rendered[0] = s_gb[512];
            u32* host_count = reinterpret_cast<u32*>((size_t)*reinterpret_cast<const unsigned long long*>(num_rendered + 2));
            // (only the speculative finish delivers: by the time an exact re-run of this phase executes, the caller has read the
            // word and may have handed it to another forward)
            if (host_count && deliver_count) __hip_atomic_store(host_count, s_gb[512], __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
        }
        build_tile_desc<ORD_PER>(tile_total, nullptr, T, tile_desc, n_active, cur, s_cur, s_wave, tile_lstart, s_gb, ranges);
        const u32 v = ((int)threadIdx.x < B) ? block_total[threadIdx.x] : 0u;
        const u32 ex = block_excl_scan_1024(v, s_wave);
        if ((int)threadIdx.x < B) block_base[threadIdx.x] = ex;
        return;
    }
    if (s_gb[512] > capacity) return;                        // the lists do not fit the buffer: nothing is written (the caller reruns)
    // u32 cursors: a tile's ids go to the first half of its own slice of `words` (u32 index 2 * first instance + position)
#pragma unroll
    for (int u = 0; u < 8; ++u) {
        const int t = threadIdx.x + u * BIN_THREADS;
        if (t < T) cur[t] = 2u * (s_gb[t >> 6] + ls0[u]) + pr0[u];
    }
    for (int t0 = threadIdx.x + 8 * BIN_THREADS; t0 < T; t0 += 8 * BIN_THREADS) {       // (more than 8192 tiles: 4K images)
        u32 ls[8], pr[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int t = t0 + u * BIN_THREADS;
            ls[u] = t < T ? tile_lstart[t] : 0u; pr[u] = t < T ? prow[t] : 0u;
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int t = t0 + u * BIN_THREADS;
            if (t < T) cur[t] = 2u * (s_gb[t >> 6] + ls[u]) + pr[u];
        }
    }
    __syncthreads();
    (void)depth_key;
    // Dense scenes: the emission in BANDS of tile rows, one pass over the block's Gaussians per band.  A tile's list is written 4 bytes at
    // a time by 245 workgroups through eight L2s; with thousands of instances per tile the partial lines of all 8160 lists do not stay in
    // the L2s until their neighbours arrive -- at sm 2.0 the launch wrote 525 MB to memory for 58 MB of ids (PMC, profiles/r05/ab_dense.txt).
    // A band's lists do.  The order inside a (block, tile) group was arbitrary before and still is; the per-tile sort follows.
    // How many: the lines one XCD's workgroups have open -- half a byte per instance (4 bytes, eight XCDs) plus a line per tile -- should fit
    // EMIT_OPEN_BYTES of its L2; but every band is another pass over the block's Gaussians, which only pays while a Gaussian has more than a
    // tile or so per pass: 5 M Gaussians @4K (3.8 tiles each, 32 400 tiles) would need six bands and is slower with any number of them.
    const int grid_y = T / grid_x;
    const u32 total = s_gb[512];
    const u32 open_bytes = total / 2u + (u32)T * 128u;
    int bands = min(EMIT_BANDS_MAX, max(1, (int)((open_bytes + EMIT_OPEN_BYTES - 1u) / EMIT_OPEN_BYTES)));
    if ((unsigned long long)bands * (unsigned long long)P * 4ull > (unsigned long long)total * 3ull) bands = 1;     // < 4/3 tiles per Gaussian and pass
    for (int bd = 0; bd < bands; ++bd) {
        const int y_lo = bd * grid_y / bands, y_hi = (bd + 1 == bands) ? 0x7FFF : (bd + 1) * grid_y / bands;
        (void)walk_block<true>(cur, P, per_block, grid_x, rect, tiles_touched, keep, reinterpret_cast<u32*>(words), y_lo, y_hi);
        if (bd + 1 < bands) __syncthreads();                 // (keeps the workgroup's waves in one band; nothing depends on it)
    }
}

// (5. the per-tile sorts: tile_sort.h, run by each tile's own blend_fwd workgroup since round 4 -- the kernel that used to launch
//  them here, one launch for every list, is in the git history.)

// ------------------------------------------------------------------------------------------------ launchers
static size_t ord_tr_bytes(int per) { return (size_t)16 * (64 * per + 2 * per) * 4; }   // 34 / 68 / 135 KB
static int ord_per_for(int T) { return T <= 1024 * 8 ? 8 : T <= 1024 * 16 ? 16 : 32; }
template <typename K>
static hipError_t big_lds(K kernel, size_t bytes)             // more than 64 KB of dynamic LDS has to be asked for
{
    if (bytes <= 65536 - 9 * 1024) return hipSuccess;        // (leaves room for the kernels' static arrays)
    return hipFuncSetAttribute(reinterpret_cast<const void*>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
}
static hipError_t launch_ranges_order(const ImgView& im, const GeomView& g, int T, int B, hipStream_t st, u32* host_count = nullptr)
{
    hipError_t e = hipSuccess;
#define RO_LAUNCH(PER) { e = big_lds(ranges_order_kernel<PER>, ord_tr_bytes(PER));                                               \
        if (e == hipSuccess) hipLaunchKernelGGL(ranges_order_kernel<PER>, dim3(1), dim3(1024), ord_tr_bytes(PER), st, im.tile_total, T, \
                                                im.ranges, g.block_total, B, g.block_base, g.num_rendered, host_count); }
    switch (ord_per_for(T)) { case 8: RO_LAUNCH(8) break; case 16: RO_LAUNCH(16) break; default: RO_LAUNCH(32) }
#undef RO_LAUNCH
    return e;
}
hipError_t launch_binned_desc_only(const ImgView& im, int T, hipStream_t st)
{
    hipError_t e = hipSuccess;
#define TD_LAUNCH(PER) { e = big_lds(tile_desc_kernel<PER>, ord_tr_bytes(PER));                                                  \
        if (e == hipSuccess) hipLaunchKernelGGL(tile_desc_kernel<PER>, dim3(1), dim3(1024), ord_tr_bytes(PER), st, im.tile_total,     \
                                                im.ranges, T, im.tile_desc, im.n_active); }
    switch (ord_per_for(T)) { case 8: TD_LAUNCH(8) break; case 16: TD_LAUNCH(16) break; default: TD_LAUNCH(32) }
#undef TD_LAUNCH
    return e != hipSuccess ? e : hipGetLastError();
}
int binned_per_block(int P)
{
    int per = (P + 255) / 256;                                // <= 256 blocks
    per = (per + BIN_THREADS - 1) / BIN_THREADS * BIN_THREADS;   // a multiple of the workgroup size
    return per < BIN_THREADS ? BIN_THREADS : per;
}
bool binned_supported(int P, int T)
{
    return T <= 1024 * ORD_PER_MAX && ((T + 1) / 2) * 4 <= 65536 && binned_per_block(P) <= 65535;
}

hipError_t launch_binned_prepare(const GeomView& g, const ImgView& im, int P, int grid_x, int T, hipStream_t st, u32* host_count, bool count_now)
{
    const int per = binned_per_block(P), B = cdiv(P, per), T2 = (T + 1) / 2;
    (void)grid_x;                                             // (the counts came with K1: launch_preprocess_fwd(count_into))
    LAUNCH_K(tile_prefix_kernel, dim3(cdiv(T2, PFX_WORDS)), dim3(1024), 0, st, im.cnt_rows, B, T, T2, im.pre, im.tile_total,
                       im.tile_lstart, im.group_total, g.num_rendered + 2, count_now ? nullptr : host_count);
    // count_now: the caller needs the instance count before the second phase is enqueued (bags_forward_prepare hands it to the
    // host): tile ranges, count and block bases by ranges_order, one more launch.  Otherwise (speculative forward) the emission
    // launch computes them itself and its descriptor workgroup writes the count into the caller's pinned word.
    if (!count_now) return hipGetLastError();
    hipError_t e = launch_ranges_order(im, g, T, B, st, host_count);
    return e != hipSuccess ? e : hipGetLastError();
}

// P == 0: no Gaussian block exists; the tile list is all empty tiles
hipError_t launch_binned_empty(const GeomView& g, const ImgView& im, int T, hipStream_t st)
{
    hipError_t e = hipMemsetAsync(im.tile_total, 0, sizeof(u32) * (size_t)T, st);
    if (e != hipSuccess) return e;
    e = launch_ranges_order(im, g, T, 0, st);
    if (e != hipSuccess) return e;
    return launch_binned_desc_only(im, T, st);
}

hipError_t launch_binned_finish(const GeomView& g, const ImgView& im, int P, int grid_x, int T, u64* words, u32 capacity, hipStream_t st,
                                bool deliver_count)
{
    const int per = binned_per_block(P), B = cdiv(P, per);
    // T slot cursors (up to 128 KB at 32768 tiles: one workgroup per CU); the descriptor workgroup's transposes fit beside
    hipError_t e = hipSuccess;
#define EM_LAUNCH(PER) { const size_t lds = (size_t)T * 4 > ord_tr_bytes(PER) ? (size_t)T * 4 : ord_tr_bytes(PER);              \
        e = big_lds(emit_binned_kernel<PER>, lds);                                                                                \
        if (e == hipSuccess) LAUNCH_K(emit_binned_kernel<PER>, dim3(B + 1), dim3(BIN_THREADS), lds, st, P, per, grid_x, T, g.rect, \
                g.tiles_touched, g.keep, im.pre, im.ranges, im.tile_lstart, im.group_total, g.depth_key, words, capacity,   \
                im.tile_total, im.tile_desc, im.n_active, g.num_rendered, g.block_total, B, g.block_base, deliver_count ? 1 : 0); }
    switch (ord_per_for(T)) { case 8: EM_LAUNCH(8) break; case 16: EM_LAUNCH(16) break; default: EM_LAUNCH(32) }
#undef EM_LAUNCH
    // (blend_fwd sorts every tile's list itself: tile_sort.h)
    return e != hipSuccess ? e : hipGetLastError();
}

// 64-bit (tile | depth) keys of the sorted list, for the parity tests
__global__ void debug_keys_ranges_kernel(const uint2* ranges, const u32* point_list, const u32* depth_key, int T, u64* out)
{
    const int t = blockIdx.x;
    if (t >= T) return;
    const uint2 r = ranges[t];
    for (u32 i = r.x + threadIdx.x; i < r.y; i += blockDim.x) out[i] = ((u64)t << 32) | (u64)depth_key[point_list[i]];
}
hipError_t launch_debug_keys_ranges(const uint2* ranges, const u32* point_list, const u32* depth_key, int T, u64* out, hipStream_t st)
{
    if (T <= 0) return hipSuccess;
    hipLaunchKernelGGL(debug_keys_ranges_kernel, dim3(T), dim3(128), 0, st, ranges, point_list, depth_key, T, out);
    return hipGetLastError();
}
