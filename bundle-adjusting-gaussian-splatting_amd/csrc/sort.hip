// sort.hip -- K2..K5: depth ordering of Gaussians, instance offsets, instance emission, stable per-tile ordering,
// tile ranges.
//
// The stock pipeline sorts I (tile<<32 | depth) 64-bit keys in 6+ radix passes.  Here the same total order
// (tile ascending, depth ascending, Gaussian id ascending among equal depths) is produced with far less traffic:
//   1. sort the P Gaussians once by their 32-bit depth key (stable, ids ascending among ties)        [P items]
//   2. exclusive-scan tiles_touched in that depth order -> emission offsets, instance count I
//   3. emit (tile id, Gaussian id) instances in depth order
//   4. stable LSD radix sort of the instances by tile id only (ceil(bit_length(T-1)/8) passes)         [I items]
// Since step 4 is stable and its input is depth-ordered, each tile's run is depth-ordered: bit-identical to the
// stable sort of the 64-bit keys (checked against oracle/raster_oracle.py: bin_and_sort).
//
// All radix passes are deterministic: ranks come from wave64 ballots and per-wave LDS counters, never atomics
// that race for order.
#include "bags_common.h"

// ------------------------------------------------------------------------------------------------ radix: histogram
template <int ITEMS>
__global__ void __launch_bounds__(SORT_BLOCK)
radix_hist_kernel(const u32* __restrict__ keys, long long n_cap, const u32* __restrict__ n_dev, int shift, int nblocks,
                  u32* __restrict__ hist)
{
    // speculative forward: the count lives on the device; a count above the capacity turns the pass into a no-op
    const long long n = n_dev ? ((long long)*n_dev <= n_cap ? (long long)*n_dev : 0ll) : n_cap;
    __shared__ u32 h[RADIX_BINS];
    h[threadIdx.x] = 0;
    __syncthreads();
    const long long base = (long long)blockIdx.x * (SORT_BLOCK * ITEMS);
#pragma unroll 4
    for (int r = 0; r < ITEMS; ++r) {
        const long long idx = base + (long long)r * SORT_BLOCK + threadIdx.x;
        if (idx < n) atomicAdd(&h[(keys[idx] >> shift) & (RADIX_BINS - 1)], 1u);
    }
    __syncthreads();
    hist[(size_t)threadIdx.x * nblocks + blockIdx.x] = h[threadIdx.x];
}

// one workgroup per digit: exclusive scan of that digit's per-block counts (in place) + digit total
__global__ void __launch_bounds__(256)
radix_scan_kernel(u32* __restrict__ hist, int nblocks, u32* __restrict__ totals)
{
    __shared__ u32 wsum[4];
    __shared__ u32 carry_s;
    u32* row = hist + (size_t)blockIdx.x * nblocks;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (threadIdx.x == 0) carry_s = 0;
    __syncthreads();
    for (int base = 0; base < nblocks; base += 256) {
        const int i = base + threadIdx.x;
        const u32 v = (i < nblocks) ? row[i] : 0u;
        u32 incl = v;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
            const u32 t = __shfl_up(incl, d);
            if (lane >= d) incl += t;
        }
        if (lane == 63) wsum[wave] = incl;
        __syncthreads();
        u32 wbase = 0;
        for (int w = 0; w < wave; ++w) wbase += wsum[w];
        const u32 carry = carry_s;
        if (i < nblocks) row[i] = carry + wbase + incl - v;
        __syncthreads();
        if (threadIdx.x == 255) carry_s = carry + wbase + incl;
        __syncthreads();
    }
    if (threadIdx.x == 0) totals[blockIdx.x] = carry_s;
}

// ------------------------------------------------------------------------------------------------ radix: scatter
// Item order inside a workgroup is (wave, round, lane) == ascending index, so ranks computed as
//   [digits of earlier waves] + [same-digit items of earlier rounds of this wave] + [same-digit lower lanes]
// make the pass stable.
template <bool IOTA, int ITEMS>
__global__ void __launch_bounds__(SORT_BLOCK)
radix_scatter_kernel(const u32* __restrict__ keys_in, const u32* __restrict__ vals_in, u32* __restrict__ keys_out,
                     u32* __restrict__ vals_out, long long n_cap, const u32* __restrict__ n_dev, int shift, int nblocks,
                     const u32* __restrict__ hist, const u32* __restrict__ totals)
{
    const long long n = n_dev ? ((long long)*n_dev <= n_cap ? (long long)*n_dev : 0ll) : n_cap;
    constexpr int WAVES = SORT_BLOCK / 64;
    constexpr int ROUNDS = ITEMS;                            // rounds of 64 keys per wave
    constexpr int TILE = SORT_BLOCK * ITEMS;
    __shared__ u32 whist[WAVES][RADIX_BINS];                 // per-wave digit counts -> per-wave local bases
    __shared__ u32 gbase[RADIX_BINS];                        // global position of this workgroup's first key of digit d
    __shared__ u32 lbase[RADIX_BINS];                        // local (in-workgroup) position of the first key of digit d
    __shared__ u32 skey[TILE], sval[TILE];                   // the workgroup's keys in digit order
    __shared__ u32 ws[WAVES];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int w = 0; w < WAVES; ++w) whist[w][threadIdx.x] = 0;
    // exclusive scan of the 256 digit totals (every workgroup repeats this 1 KB scan) + this workgroup's offset
    {
        const u32 v = totals[threadIdx.x];
        u32 incl = v;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
            const u32 t = __shfl_up(incl, d);
            if (lane >= d) incl += t;
        }
        if (lane == 63) ws[wave] = incl;
        __syncthreads();
        u32 wb = 0;
        for (int w = 0; w < wave; ++w) wb += ws[w];
        gbase[threadIdx.x] = wb + incl - v + hist[(size_t)threadIdx.x * nblocks + blockIdx.x];
    }
    __syncthreads();

    const long long seg = (long long)blockIdx.x * TILE + (long long)wave * (ROUNDS * 64);
    u32 key[ROUNDS];
    u32 rank[ROUNDS];
    const u64 lt_mask = (lane == 0) ? 0ull : (~0ull >> (64 - lane));
#pragma unroll
    for (int r = 0; r < ROUNDS; ++r) {
        const long long idx = seg + r * 64 + lane;
        const bool valid = idx < n;
        const u32 k = valid ? keys_in[idx] : 0xFFFFFFFFu;
        const u32 d = (k >> shift) & (RADIX_BINS - 1);
        u64 m = __ballot(valid);
#pragma unroll
        for (int b = 0; b < RADIX_BITS; ++b) {
            const u64 bb = __ballot((d >> b) & 1u);
            m &= ((d >> b) & 1u) ? bb : ~bb;
        }
        const u32 below = (u32)__popcll(m & lt_mask);
        const u32 pre = whist[wave][d];
        if (valid && below == 0) whist[wave][d] = pre + (u32)__popcll(m);
        key[r] = k;
        rank[r] = pre + below;
    }
    __syncthreads();
    // digit t: workgroup total -> exclusive scan over digits (lbase); per-wave counts -> per-wave local bases
    {
        u32 c[WAVES], tot = 0;
#pragma unroll
        for (int w = 0; w < WAVES; ++w) { c[w] = whist[w][threadIdx.x]; tot += c[w]; }
        u32 incl = tot;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
            const u32 t = __shfl_up(incl, d);
            if (lane >= d) incl += t;
        }
        __syncthreads();                                     // ws reuse
        if (lane == 63) ws[wave] = incl;
        __syncthreads();
        u32 wb = 0;
        for (int w = 0; w < wave; ++w) wb += ws[w];
        u32 run = wb + incl - tot;
        lbase[threadIdx.x] = run;
#pragma unroll
        for (int w = 0; w < WAVES; ++w) { whist[w][threadIdx.x] = run; run += c[w]; }
    }
    __syncthreads();
    // stage in digit order (stable: waves, rounds, lanes in ascending index order inside every digit)
#pragma unroll
    for (int r = 0; r < ROUNDS; ++r) {
        const long long idx = seg + r * 64 + lane;
        if (idx < n) {
            const u32 d = (key[r] >> shift) & (RADIX_BINS - 1);
            const u32 lp = whist[wave][d] + rank[r];
            skey[lp] = key[r];
            sval[lp] = IOTA ? (u32)idx : vals_in[idx];
        }
    }
    __syncthreads();
    // coalesced write-out: consecutive threads hold consecutive keys of the same digit run
    const long long left = n - (long long)blockIdx.x * TILE;
    const int cnt = left < TILE ? (int)left : TILE;
#pragma unroll
    for (int r = 0; r < ROUNDS; ++r) {
        const int i = r * SORT_BLOCK + threadIdx.x;
        if (i < cnt) {
            const u32 k = skey[i];
            const u32 d = (k >> shift) & (RADIX_BINS - 1);
            const u32 pos = gbase[d] + ((u32)i - lbase[d]);
            keys_out[pos] = k;
            vals_out[pos] = sval[i];
        }
    }
}

template <int ITEMS>
static hipError_t radix_sort_impl(const u32* src_k, const u32* src_v, u32* a_k, u32* a_v, u32* b_k, u32* b_v, long long n,
                                  int bits, bool iota_vals, u32* hist, u32* totals, hipStream_t st, const u32* n_dev)
{
    const int nblocks = cdiv(n, SORT_BLOCK * ITEMS);
    const int passes = (bits + RADIX_BITS - 1) / RADIX_BITS;
    const u32* ik = src_k; const u32* iv = src_v;
    for (int p = 0; p < passes; ++p) {
        const int shift = p * RADIX_BITS;
        u32* ok = (p & 1) ? b_k : a_k; u32* ov = (p & 1) ? b_v : a_v;
        hipLaunchKernelGGL(radix_hist_kernel<ITEMS>, dim3(nblocks), dim3(SORT_BLOCK), 0, st, ik, n, n_dev, shift, nblocks, hist);
        hipLaunchKernelGGL(radix_scan_kernel, dim3(RADIX_BINS), dim3(256), 0, st, hist, nblocks, totals);
        if (p == 0 && iota_vals)
            hipLaunchKernelGGL((radix_scatter_kernel<true, ITEMS>), dim3(nblocks), dim3(SORT_BLOCK), 0, st, ik, iv, ok, ov, n,
                               n_dev, shift, nblocks, hist, totals);
        else
            hipLaunchKernelGGL((radix_scatter_kernel<false, ITEMS>), dim3(nblocks), dim3(SORT_BLOCK), 0, st, ik, iv, ok, ov, n,
                               n_dev, shift, nblocks, hist, totals);
        ik = ok; iv = ov;
    }
    return hipGetLastError();
}

// keys per workgroup: small inputs get small tiles (more workgroups, shorter dependent chains), large inputs big ones
int radix_items_for(long long n) { return n <= (1ll << 21) ? SORT_ITEMS_SMALL : SORT_ITEMS; }
int radix_blocks_for(long long n) { return cdiv(n > 0 ? n : 1, (long long)SORT_BLOCK * radix_items_for(n)); }

hipError_t launch_radix_sort(const u32* src_k, const u32* src_v, u32* a_k, u32* a_v, u32* b_k, u32* b_v, long long n,
                             int bits, bool iota_vals, u32* hist, u32* totals, int /*nblocks*/, hipStream_t st,
                             const u32* n_dev)
{
    if (n <= 0) return hipSuccess;
    if (radix_items_for(n) == SORT_ITEMS_SMALL)
        return radix_sort_impl<SORT_ITEMS_SMALL>(src_k, src_v, a_k, a_v, b_k, b_v, n, bits, iota_vals, hist, totals, st, n_dev);
    return radix_sort_impl<SORT_ITEMS>(src_k, src_v, a_k, a_v, b_k, b_v, n, bits, iota_vals, hist, totals, st, n_dev);
}

// ------------------------------------------------------------------------------------------------ offsets scan
// Exclusive scan of tiles_touched in depth order.  Two small kernels: workgroup sums, then a rescan in which every
// workgroup first adds up the sums of the workgroups in front of it.
__global__ void __launch_bounds__(SCAN_BLOCK)
offsets_partial_kernel(const u32* __restrict__ sorted_ids, const u32* __restrict__ tiles_touched, int P,
                       u32* __restrict__ partials)
{
    __shared__ u32 ws[SCAN_BLOCK / 64];
    const int base = blockIdx.x * SCAN_TILE;
    u32 s = 0;
#pragma unroll
    for (int r = 0; r < SCAN_ITEMS; ++r) {
        const int j = base + r * SCAN_BLOCK + threadIdx.x;
        if (j < P) s += tiles_touched[sorted_ids[j]];
    }
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) s += __shfl_xor(s, d);
    if ((threadIdx.x & 63) == 0) ws[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) {
        u32 t = 0;
        for (int w = 0; w < SCAN_BLOCK / 64; ++w) t += ws[w];
        partials[blockIdx.x] = t;
    }
}

// item order inside a workgroup here is (thread, item) with SCAN_ITEMS consecutive ranks per thread
__global__ void __launch_bounds__(SCAN_BLOCK)
offsets_final_kernel(const u32* __restrict__ sorted_ids, const u32* __restrict__ tiles_touched, int P,
                     const u32* __restrict__ partials, u32* __restrict__ rank_offset, u32* __restrict__ inst_off,
                     u32* __restrict__ total)
{
    __shared__ u32 wsum[SCAN_BLOCK / 64], wpre[SCAN_BLOCK / 64];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int j0 = blockIdx.x * SCAN_TILE + threadIdx.x * SCAN_ITEMS;
    u32 id[SCAN_ITEMS], t[SCAN_ITEMS];
    u32 s = 0;
#pragma unroll
    for (int r = 0; r < SCAN_ITEMS; ++r) {
        const int j = j0 + r;
        id[r] = (j < P) ? sorted_ids[j] : 0u;
        t[r] = (j < P) ? tiles_touched[id[r]] : 0u;
        s += t[r];
    }
    u32 incl = s;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const u32 u = __shfl_up(incl, d);
        if (lane >= d) incl += u;
    }
    if (lane == 63) wsum[wave] = incl;
    // what the earlier workgroups hold: every workgroup adds their sums up itself (a few hundred values; a kernel of its
    // own for this scan cost 4.6 us plus a launch gap)
    u32 before = 0;
    for (int b = threadIdx.x; b < (int)blockIdx.x; b += SCAN_BLOCK) before += partials[b];
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) before += __shfl_xor(before, d);
    if (lane == 0) wpre[wave] = before;
    __syncthreads();
    u32 run = incl - s;
#pragma unroll
    for (int w = 0; w < SCAN_BLOCK / 64; ++w) run += wpre[w] + ((w < wave) ? wsum[w] : 0u);
#pragma unroll
    for (int r = 0; r < SCAN_ITEMS; ++r) {
        const int j = j0 + r;
        if (j < P) { rank_offset[j] = run; inst_off[id[r]] = run; }
        run += t[r];
    }
    if (blockIdx.x == gridDim.x - 1 && threadIdx.x == SCAN_BLOCK - 1) total[0] = run;       // the instance count
}

hipError_t launch_offsets_scan(const GeomView& g, const u32* sorted_ids, int P, hipStream_t st)
{
    if (P == 0) { return hipMemsetAsync(g.num_rendered, 0, sizeof(u32), st); }
    const int nb = g.nblocks_scan;
    hipLaunchKernelGGL(offsets_partial_kernel, dim3(nb), dim3(SCAN_BLOCK), 0, st, sorted_ids, g.tiles_touched, P, g.scan_partials);
    hipLaunchKernelGGL(offsets_final_kernel, dim3(nb), dim3(SCAN_BLOCK), 0, st, sorted_ids, g.tiles_touched, P,
                       g.scan_partials, g.rank_offset, g.inst_off, g.num_rendered);
    return hipGetLastError();
}

// ------------------------------------------------------------------------------------------------ emission
// One thread per depth rank writes the Gaussian's rectangle of tiles (y outer, x inner) at its offset.  Rectangles of
// more than EMIT_COOP tiles (a few huge splats can cover the whole image) are written by the whole wave instead, 64
// consecutive instances per step, so no lane serialises thousands of stores.
#define EMIT_COOP 32
__global__ void __launch_bounds__(256)
emit_kernel(const u32* __restrict__ sorted_ids, const u32* __restrict__ rank_offset, const uint2* __restrict__ rect,
            const u32* __restrict__ tiles_touched, const u64* __restrict__ keep, int P, int grid_x, u32* __restrict__ keys, u32* __restrict__ vals,
            u32 capacity, const u32* __restrict__ n_dev, uint2* __restrict__ ranges, int T)
{
    // tile_ranges only writes the tiles that hold instances: the others must read (0, 0).  Cleared here, two kernels ahead
    // of their use (a memset node of its own cost 4.6 us of stream time), and before the early exit below: a failed
    // speculative pass still runs the blend over these ranges.
    for (int t = blockIdx.x * blockDim.x + threadIdx.x; t < T; t += gridDim.x * blockDim.x) ranges[t] = make_uint2(0u, 0u);
    if (n_dev && *n_dev > capacity) return;                 // speculative capacity exceeded: nothing is emitted, the caller reruns
    const int j = blockIdx.x * blockDim.x + threadIdx.x;
    const int lane = threadIdx.x & 63;
    u32 g = 0, nt = 0, off = 0;
    uint2 rc = make_uint2(0u, 0u);
    u64 kp = 0ull;                                          // tile mask of a small rectangle (GeomView::keep)
    if (j < P) {
        g = sorted_ids[j];
        nt = tiles_touched[g];
        if (nt) { rc = rect[g]; off = rank_offset[j]; kp = keep[g]; }
        if ((unsigned long long)off + nt > (unsigned long long)capacity) nt = 0;
    }
    // Rectangles of up to EMIT_COOP tiles: the wave expands its 64 rectangles TOGETHER, lane = output element, so the
    // stores are coalesced.  (One lane writing its own rectangle is a 4-byte store per element at 64 unrelated addresses
    // per instruction: 4.9 M write requests per frame, and the request rate of the L2 channels -- not the 19.5 MB --
    // was what the kernel took 18.7 us for.)  Elements are numbered over the wave's small rectangles (exclusive wave scan
    // `lp`), a byte map in LDS names each element's owner lane, and the owner's rectangle comes through shuffles.
    __shared__ unsigned char owner[4][64 * EMIT_COOP];
    const int wave = threadIdx.x >> 6;
    const u32 nt_s = (nt > 0 && nt <= EMIT_COOP) ? nt : 0u;
    u32 incl = nt_s;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) { const u32 o_ = (u32)__shfl_up((int)incl, d); if (lane >= d) incl += o_; }
    const u32 lp = incl - nt_s;
    const u32 total = (u32)__shfl((int)incl, 63);
    for (u32 q = 0; q < nt_s; ++q) owner[wave][lp + q] = (unsigned char)lane;
    __builtin_amdgcn_wave_barrier();
    for (u32 eb = 0; eb < total; eb += 64) {                      // uniform trip count: a shuffle cannot read a lane that left the loop
        const u32 e = eb + (u32)lane;
        const bool have = e < total;
        const int ol = have ? (int)owner[wave][e] : 0;
        const u32 g_o = (u32)__shfl((int)g, ol), off_o = (u32)__shfl((int)off, ol), lp_o = (u32)__shfl((int)lp, ol);
        const u32 rx = (u32)__shfl((int)rc.x, ol), ry = (u32)__shfl((int)rc.y, ol);
        const u64 kp_o = (u64)(u32)__shfl((int)(u32)kp, ol) | ((u64)(u32)__shfl((int)(u32)(kp >> 32), ol) << 32);
        const u32 kk = e - lp_o;                                   // element of the owner's rectangle, y outer, x inner
        const int minx = rx & 0xFFFF, miny = rx >> 16, w = (int)(ry & 0xFFFF) - minx, h = (int)(ry >> 16) - miny;
        int dy = (int)(((float)kk + 0.5f) / (float)w);             // exact: kk < 32, w <= 32
        int dx = (int)kk - dy * w;
        if (have && rect_small(w, h)) { const int bit = rect_nth_tile(kp_o, kk); dy = bit >> 3; dx = bit & 7; }
        if (have) {
            keys[off_o + kk] = (u32)((miny + dy) * grid_x + minx + dx);
            vals[off_o + kk] = g_o;
        }
    }
    u64 big = __ballot(nt > EMIT_COOP);
    while (big) {
        const int src = __ffsll((long long)big) - 1;
        big &= big - 1;
        const u32 bg_ = __shfl(g, src), bn = __shfl(nt, src), bo = __shfl(off, src);
        const u32 bx = __shfl(rc.x, src), by = __shfl(rc.y, src);
        const u64 bk = (u64)(u32)__shfl((int)(u32)kp, src) | ((u64)(u32)__shfl((int)(u32)(kp >> 32), src) << 32);
        const int minx = bx & 0xFFFF, miny = bx >> 16, w = (int)(by & 0xFFFF) - minx, h = (int)(by >> 16) - miny;
        for (u32 k = lane; k < bn; k += 64) {
            int y = miny + (int)(k / (u32)w), x = minx + (int)(k % (u32)w);
            if (rect_small(w, h)) { const int bit = rect_nth_tile(bk, k); y = miny + (bit >> 3); x = minx + (bit & 7); }
            keys[bo + k] = (u32)(y * grid_x + x);
            vals[bo + k] = bg_;
        }
    }
}

hipError_t launch_emit(const GeomView& g, const u32* sorted_ids, int P, int grid_x, u32* keys, u32* vals, u32 capacity,
                       hipStream_t st, const u32* n_dev, uint2* ranges, int T)
{
    if (P == 0) return hipMemsetAsync(ranges, 0, sizeof(uint2) * (size_t)T, st);
    hipLaunchKernelGGL(emit_kernel, dim3(cdiv(P, 256)), dim3(256), 0, st, sorted_ids, g.rank_offset, g.rect,
                       g.tiles_touched, g.keep, P, grid_x, keys, vals, capacity, n_dev, ranges, T);
    return hipGetLastError();
}

// ------------------------------------------------------------------------------------------------ tile ranges
__global__ void __launch_bounds__(256)
tile_ranges_kernel(const u32* __restrict__ tile_sorted, long long I_cap, const u32* __restrict__ n_dev, uint2* __restrict__ ranges)
{
    const long long I = n_dev ? ((long long)*n_dev <= I_cap ? (long long)*n_dev : 0ll) : I_cap;
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= I) return;
    const u32 t = tile_sorted[i];
    if (i == 0) ranges[t].x = 0;
    else {
        const u32 p = tile_sorted[i - 1];
        if (p != t) { ranges[p].y = (u32)i; ranges[t].x = (u32)i; }
    }
    if (i == I - 1) ranges[t].y = (u32)I;
}

// `cleared`: launch_emit has already zeroed the ranges on this stream
hipError_t launch_tile_ranges(const u32* tile_sorted, long long I, uint2* ranges, int T, hipStream_t st, const u32* n_dev,
                              bool cleared)
{
    hipError_t e = cleared ? hipSuccess : hipMemsetAsync(ranges, 0, sizeof(uint2) * (size_t)T, st);
    if (e != hipSuccess || I == 0) return e;
    hipLaunchKernelGGL(tile_ranges_kernel, dim3(cdiv(I, 256)), dim3(256), 0, st, tile_sorted, I, n_dev, ranges);
    return hipGetLastError();
}

// ------------------------------------------------------------------------------------------------ tile order
// Heavy-first list of the tiles for the blend launches (blend.hip): a counting sort of the tiles by instance count into
// ORDER_LEVELS levels, four per octave of the count (scale-free: no pass for the maximum), fullest level first, empty
// tiles last.  Output: one descriptor {tile, first instance, instance count, 0} per tile in that order, and the number of
// non-empty tiles.  One workgroup (T is a few thousand to a few ten thousand).  Where a wave's tiles land inside their
// level depends on the order the waves' LDS atomics arrive in, which is harmless: the list only decides which workgroup
// computes which tile and when, never what is computed.
#define ORDER_LEVELS 64
__device__ __forceinline__ int order_level(u32 n)
{
    if (n == 0) return ORDER_LEVELS - 1;
    const int e = 31 - __clz((int)n);                                    // floor(log2 n)
    const int frac = (e >= 2) ? (int)((n >> (e - 2)) & 3u) : (int)((n << (2 - e)) & 3u);
    const int q = e * 4 + frac;                                          // 4 levels per octave
    return ORDER_LEVELS - 2 - min(ORDER_LEVELS - 2, q);                  // n >= 2^15.5 share level 0
}
#define ORDER_SUB 32                       // sub-lists per level
__global__ void __launch_bounds__(1024)
tile_order_kernel(const uint2* __restrict__ ranges, int T, uint4* __restrict__ tile_desc, u32* __restrict__ n_active)
{
    // Neighbouring tiles mostly share a level, and same-address LDS atomics retire one lane per clock: a single counter
    // per level made the two passes ~16 k serialised atomics (most of the kernel's 11 us; peeling a wave's distinct
    // levels with ballots instead was 4x slower still).  Every level therefore has ORDER_SUB counters, picked by
    // (tile >> 3) & 31: the 64 tiles one wave handles per round spread over 8 of them, and runs of 8 neighbouring tiles
    // stay together in the list (they share splats, hence L2 lines, when their workgroups run side by side).
    __shared__ u32 s_cur[ORDER_LEVELS * ORDER_SUB];                      // 8 KB
    __shared__ u32 s_wave[16];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    s_cur[tid] = 0; s_cur[tid + 1024] = 0;
    __syncthreads();
    auto counter_of = [&](int t, u32 n) -> int { return order_level(n) * ORDER_SUB + ((t >> 3) & (ORDER_SUB - 1)); };
#pragma unroll 4
    for (int t = tid; t < T; t += 1024) { const uint2 r = ranges[t]; atomicAdd(&s_cur[counter_of(t, r.y - r.x)], 1u); }
    __syncthreads();
    {   // exclusive scan over the 2048 counters: two per thread, wave scan, 16 wave totals
        const u32 c0 = s_cur[2 * tid], c1 = s_cur[2 * tid + 1];
        u32 inc = c0 + c1;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) { const u32 o = (u32)__shfl_up((int)inc, d); if (lane >= d) inc += o; }
        if (lane == 63) s_wave[wave] = inc;
        __syncthreads();
        u32 before = 0;
#pragma unroll
        for (int w = 0; w < 16; ++w) before += (w < wave) ? s_wave[w] : 0u;
        const u32 excl = before + inc - (c0 + c1);
        s_cur[2 * tid] = excl; s_cur[2 * tid + 1] = excl + c0;
        if (2 * tid == (ORDER_LEVELS - 1) * ORDER_SUB) *n_active = excl; // everything in front of the empty tiles
    }
    __syncthreads();
#pragma unroll 4
    for (int t = tid; t < T; t += 1024) {
        const uint2 r = ranges[t];
        const u32 n = r.y - r.x;
        tile_desc[atomicAdd(&s_cur[counter_of(t, n)], 1u)] = make_uint4((u32)t, r.x, n, 0u);
    }
}

hipError_t launch_tile_order(const uint2* ranges, int T, uint4* tile_desc, u32* n_active, hipStream_t st)
{
    if (T <= 0) return hipSuccess;
    hipLaunchKernelGGL(tile_order_kernel, dim3(1), dim3(1024), 0, st, ranges, T, tile_desc, n_active);
    return hipGetLastError();
}

// ------------------------------------------------------------------------------------------------ debug views
__global__ void debug_keys_kernel(const u32* tile_sorted, const u32* point_list, const u32* depth_key, long long I, u64* out)
{
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < I) out[i] = ((u64)tile_sorted[i] << 32) | (u64)depth_key[point_list[i]];
}
hipError_t launch_debug_keys(const u32* tile_sorted, const u32* point_list, const u32* depth_key, long long I, u64* out, hipStream_t st)
{
    if (I == 0) return hipSuccess;
    hipLaunchKernelGGL(debug_keys_kernel, dim3(cdiv(I, 256)), dim3(256), 0, st, tile_sorted, point_list, depth_key, I, out);
    return hipGetLastError();
}
__global__ void unpack_rect_kernel(const uint2* rect, int P, u32* out)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < P) {
        const uint2 r = rect[i];
        out[4 * i] = r.x & 0xFFFF; out[4 * i + 1] = r.x >> 16; out[4 * i + 2] = r.y & 0xFFFF; out[4 * i + 3] = r.y >> 16;
    }
}
hipError_t launch_unpack_rect(const uint2* rect, int P, u32* out, hipStream_t st)
{
    if (P == 0) return hipSuccess;
    hipLaunchKernelGGL(unpack_rect_kernel, dim3(cdiv(P, 256)), dim3(256), 0, st, rect, P, out);
    return hipGetLastError();
}
