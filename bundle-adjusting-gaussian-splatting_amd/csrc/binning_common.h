// binning_common.h -- pieces of the tile-binned list construction shared by binning.hip (emit, prefix, sorts) and
// preprocess_fwd.hip (K1 counts the (block of Gaussians, tile) matrix itself since round 4: one launch fewer on the path to the
// instance count, and every Gaussian's offset inside its block lands in its 64-byte geometry line).
#pragma once
#include "bags_common.h"

// Wave-wide inclusive add scan on DPP (row_shr 1/2/4/8 inside the 16-lane rows, then row_bcast:15 / row_bcast:31 across
// them): six VALU instructions.  (__shfl_up goes through ds_bpermute: six dependent LDS round trips per scan, which is
// what a one-wave sort spent most of its time waiting for.)
__device__ __forceinline__ u32 wave_incl_scan(u32 x)
{
    x += (u32)__builtin_amdgcn_update_dpp(0, (int)x, 0x111, 0xf, 0xf, false);
    x += (u32)__builtin_amdgcn_update_dpp(0, (int)x, 0x112, 0xf, 0xf, false);
    x += (u32)__builtin_amdgcn_update_dpp(0, (int)x, 0x114, 0xf, 0xf, false);
    x += (u32)__builtin_amdgcn_update_dpp(0, (int)x, 0x118, 0xf, 0xf, false);
    x += (u32)__builtin_amdgcn_update_dpp(0, (int)x, 0x142, 0xa, 0xf, false);      // row_bcast:15 into rows 1 and 3
    x += (u32)__builtin_amdgcn_update_dpp(0, (int)x, 0x143, 0xc, 0xf, false);      // row_bcast:31 into rows 2 and 3
    return x;
}
// wave-wide max / min on the same DPP pattern (the value of lane 63 of the inclusive scan), broadcast with readlane
__device__ __forceinline__ u32 wave_max(u32 x)
{
    x = max(x, (u32)__builtin_amdgcn_update_dpp(0, (int)x, 0x111, 0xf, 0xf, false));
    x = max(x, (u32)__builtin_amdgcn_update_dpp(0, (int)x, 0x112, 0xf, 0xf, false));
    x = max(x, (u32)__builtin_amdgcn_update_dpp(0, (int)x, 0x114, 0xf, 0xf, false));
    x = max(x, (u32)__builtin_amdgcn_update_dpp(0, (int)x, 0x118, 0xf, 0xf, false));
    x = max(x, (u32)__builtin_amdgcn_update_dpp(0, (int)x, 0x142, 0xa, 0xf, false));
    x = max(x, (u32)__builtin_amdgcn_update_dpp(0, (int)x, 0x143, 0xc, 0xf, false));
    return (u32)__builtin_amdgcn_readlane((int)x, 63);
}
__device__ __forceinline__ u32 wave_min(u32 x) { return ~wave_max(~x); }

#define BIN_COOP 64          // rectangles of more tiles than this are walked by the whole wave
#define BIN_THREADS 1024     // count / emit workgroup: one block of Gaussians = one workgroup = one row of the count matrix;
                             // 256 threads left every thread eight Gaussians to walk one after the other (latency bound)

// ------------------------------------------------------------------------------------------------ 1. tile_count
// Packed counters: tile t lives in the (t & 1) half of word t >> 1.  A half never overflows: a Gaussian covers a tile at
// most once, so a (block, tile) count is at most the block size (<= 65535 by construction).
__device__ __forceinline__ u32 lds_count_tile(u32* cnt, u32 t) { return atomicAdd(&cnt[t >> 1], 1u << ((t & 1u) * 16u)); }

// COUNT: packed 16-bit counters.  EMIT: `cnt` holds one 32-bit slot cursor per tile (range start + column prefix of this
// block, loaded as two coalesced rows); the returning LDS atomic hands the instance its final slot, and the Gaussian id is
// the only thing written (the per-tile sort fetches the depth key by id: a 4-byte scattered store per instance instead of
// two scattered loads and an 8-byte store -- the request rate of the L2 channels, not the bytes, bounded this kernel).
// What the per-tile sort orders: (depth key << 32) | Gaussian id.  The emission writes only the 4-byte id (round 4: the 8-byte
// word was 55 MB of partial-line HBM writes per step for 16.6 MB of words -- every word of a tile's list comes from another
// workgroup, on another XCD); the sort forms the word when it loads the list, the key gathered from the 2 MB depth_key array,
// which the L2 holds.  A tile's unsorted ids occupy the FIRST HALF of its own slice of the `words` buffer (u32 index
// 2 * first + k): the slice stays the tile's private scratch afterwards (reach words, compacted positions: blend.hip).
struct WordSrc {
    const u32* __restrict__ ids;       // the tile's ids, offset so that ids[first + k] is its k-th entry (= words32 + first)
    const u32* __restrict__ keys;      // GeomView::depth_key
    __device__ __forceinline__ u64 operator[](const u32 i) const { const u32 id = ids[i]; return ((u64)keys[id] << 32) | (u64)id; }
};
__device__ __forceinline__ WordSrc tile_words(const void* words, const u32 first, const u32* __restrict__ keys)
{
    WordSrc w; w.ids = reinterpret_cast<const u32*>(words) + first; w.keys = keys; return w;
}

// [y_lo, y_hi): the band of tile rows this pass of the emission serves (the count takes all rows)
template <bool EMIT>
__device__ __forceinline__ void walk_rect(u32* cnt, uint2 rc, int grid_x, int lane, bool coop, u32 id, u32* __restrict__ ids,
                                          const int y_lo = 0, const int y_hi = 0x7FFF)
{
    const int minx = rc.x & 0xFFFF, w = (int)(rc.y & 0xFFFF) - minx;
    const int miny = max((int)(rc.x >> 16), y_lo), h = min((int)(rc.y >> 16), y_hi) - miny;
    const int nt = w * h;                                    // (<= 0: the rectangle has no row in the band)
    for (int k = coop ? lane : 0; k < nt; k += coop ? 64 : 1) {
        const int dy = k / w, dx = k - dy * w;
        const u32 t = (u32)((miny + dy) * grid_x + minx + dx);
        if (EMIT) ids[atomicAdd(&cnt[t], 1u)] = id;
        else (void)lds_count_tile(cnt, t);
    }
}
// A rectangle of at most 8 x 8 tiles: the set bits of its tile mask (GeomView::keep), no division.
template <bool EMIT>
__device__ __forceinline__ void walk_mask(u32* cnt, uint2 rc, u64 m, int grid_x, u32 id, u32* __restrict__ ids,
                                          const int y_lo = 0, const int y_hi = 0x7FFF)
{
    const int minx = rc.x & 0xFFFF, miny = rc.x >> 16;
    const u32 t0 = (u32)(miny * grid_x + minx);
    if (EMIT) {                                              // mask rows (8 bits each) outside the band
        const int lo = min(8, max(0, y_lo - miny)), hi = min(8, max(0, y_hi - miny));
        const u64 below_hi = hi >= 8 ? ~0ull : ((1ull << (8 * hi)) - 1ull);
        const u64 below_lo = lo >= 8 ? ~0ull : ((1ull << (8 * lo)) - 1ull);
        m &= below_hi & ~below_lo;
    }
    while (m) {
        const int bit = __ffsll((long long)m) - 1;
        m &= m - 1ull;
        const u32 t = t0 + (u32)((bit >> 3) * grid_x + (bit & 7));
        if (EMIT) ids[atomicAdd(&cnt[t], 1u)] = id;
        else (void)lds_count_tile(cnt, t);
    }
}

