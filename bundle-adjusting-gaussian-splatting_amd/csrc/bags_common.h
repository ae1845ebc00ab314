// Shared host/device declarations of libbags_raster.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stddef.h>
#include "../../include/bags_raster.h"

typedef uint32_t u32;
typedef uint64_t u64;

#define BAGS_WAVE 64
#define RADIX_BITS 8
#define RADIX_BINS 256
#define SORT_BLOCK 256
#ifndef SORT_ITEMS
#define SORT_ITEMS 8                      // keys per thread per radix pass (large inputs)
#endif
#ifndef SORT_ITEMS_SMALL
#define SORT_ITEMS_SMALL 4                // ... for inputs up to 2M keys
#endif
#define SCAN_BLOCK 256
#define SCAN_ITEMS 8
#define SCAN_TILE (SCAN_BLOCK * SCAN_ITEMS)   // 2048 offsets per workgroup
#ifndef PART_FLOATS
#define PART_FLOATS 12
#endif                                    // one 48-byte partial-gradient record per sorted instance (11 sums), densely packed:
                                          // every byte of every line is written, and K8a streams 25 % less than with 64-B slots
#define POSE_VALS 40                      // pose-gradient slab row (35 used)
#define KEY_CULLED 0xFFFFFFFFu

static inline size_t align_up(size_t v, size_t a) { return (v + a - 1) / a * a; }
static inline int cdiv(long long a, long long b) { return (int)((a + b - 1) / b); }
static inline int bit_length(u32 v) { int b = 0; while (v) { ++b; v >>= 1; } return b; }

// theta = atan2(rho, tz) for rho >= 0, tz > 0 from operations every IEEE implementation rounds identically (min / max / div /
// mul / add, no libm call), so that the depth keys and tile rectangles stay bit-comparable with the fp32 oracle when the
// shift_factors polynomial is non-zero (oracle/raster_oracle.py: _atan2_pos repeats the sequence in torch).  Both callers are
// compiled with -ffp-contract=off.  u = min/max in [0,1]; above tan(pi/8) it is folded with atan(u) = pi/4 + atan((u-1)/(u+1));
// |w| <= tan(pi/8): atan(w) = w + w (s P(s)), s = w^2, P fitted to 6e-8 relative (fp32 rounding level).
__device__ __forceinline__ float det_atan2_pos(float rho, float tz)
{
    const float lo = fminf(rho, tz), hi = fmaxf(rho, tz);
    const float u = lo / hi;
    const bool red = u > 0.414213568f;
    const float w = red ? (u - 1.0f) / (u + 1.0f) : u;
    const float s = w * w;
    float p = -0.0607120693f;
    p = p * s + 0.105907366f;
    p = p * s + -0.142430589f;
    p = p * s + 0.199984416f;
    p = p * s + -0.333333135f;
    float a = w + w * (s * p);
    if (red) a = 0.785398185f + a;
    return (rho > tz) ? 1.57079637f - a : a;
}

// ---- tile masks of small rectangles (GeomView::keep) ------------------------------------------------------
// A rectangle of at most 8 x 8 tiles carries a 64-bit mask of the tiles it emits, bit ry * 8 + rx; a larger one emits all.
__device__ __forceinline__ bool rect_small(int w, int h) { return w <= 8 && h <= 8; }
__device__ __forceinline__ u64 rect_full_mask(int w, int h)
{
    if (w <= 0 || h <= 0) return 0ull;
    if (!rect_small(w, h)) return ~0ull;
    return ((1ull << w) - 1ull) * (0x0101010101010101ull >> (8 * (8 - h)));
}
// rank of tile (rx, ry) among the emitted tiles of its Gaussian = index of its partial-gradient record
__device__ __forceinline__ u32 rect_tile_rank(u64 keep, int w, int h, int rx, int ry)
{
    return rect_small(w, h) ? (u32)__popcll(keep & ((1ull << (ry * 8 + rx)) - 1ull)) : (u32)(ry * w + rx);
}
// position (ry * 8 + rx) of the n-th emitted tile of a small rectangle
__device__ __forceinline__ int rect_nth_tile(u64 keep, u32 n)
{
    for (u32 i = 0; i < n; ++i) keep &= keep - 1ull;
    return __ffsll((long long)keep) - 1;
}

// ---- carved views of the three caller-owned state buffers -------------------------------------------------
struct GeomView {            // per Gaussian, indexed by Gaussian id unless stated
    u32*    depth_key;       // float bits of the sort depth, KEY_CULLED when not rendered
    // One 64-byte line per Gaussian with everything the blend kernels gather per (tile, Gaussian) instance, so an
    // instance costs ONE line fetch instead of five partial ones from id-ordered (spatially random) SoA arrays:
    //   q0 = conic a, b, c, opacity      q1 = pixel x, y, colour r, g
    //   q2 = colour b, view depth z, rect.x, rect.y (bits)      q3 = keep lo, block of Gaussians, offset inside the block, keep hi (bits)
    float4* g2d;             // [4 * P]
    u32*    inst_off;        // [P] first emission slot of the Gaussian's partial-gradient records -- RADIX PATH ONLY since round 4 (an array
                             // of its own: a 4-byte write into every 64-byte line of g2d after the scan cost 6.6 us per frame).  On the
                             // tile-binned path the record base is block_base[block] + local_off, both known to K1 (line.q3.y / q3.z)
    uint2*  rect;            // (minx | miny<<16, maxx | maxy<<16), max exclusive   (compact copy for emit)
    u32*    tiles_touched;   // instances the Gaussian emits (compact copy for the offsets scan / emit)
    u32*    rec_count;       // partial-gradient records of the Gaussian (= tiles_touched, except on the tile-binned path with the stock tile
                             // rule: there only the tiles the opacity rule keeps have a record, round 4)
    // [10 * P] d(colour)/d(view direction) of the SH colour path, written by K1 for the visible Gaussians (round 3):
    //   M[axis][c] = sum_t (d basis_t / d dir_axis) * sh[t][c]   as   Mxr Mxg Mxb  Myr Myg Myb  Mzr Mzg Mzb, then the clamp bits
    // so that preprocess_bwd reads 40 bytes per Gaussian instead of the 192-byte SH row a second time (and nothing of g2d)
    float*  shjac;
    u64*    keep;            // tile mask of a rectangle of at most 8 x 8 tiles: bit ry * 8 + rx set = tile (minx + rx, miny + ry)
                             // is emitted (D7: the alpha >= 1/255 ellipse reaches it); larger rectangles emit every tile
    // scratch (dead after forward)
    u32 *keys_a, *keys_b, *vals_a, *vals_b;   // depth ordering of Gaussians (ping-pong)
    u32*    rank_offset;     // exclusive instance offset per depth rank
    u32*    scan_partials;
    u32*    radix_hist;      // [256][nblocks]
    u32*    digit_totals;    // [256]
    u32*    num_rendered;    // [0] instance count; [2..3] device-visible address of the caller's pinned count word (or 0); [8..] debug counters
    // tile-binned path (binning.hip)
    u32*    local_off;       // [P] instance offset of a Gaussian inside its block of Gaussians (id order)
    u32*    block_total;     // [256] instances per block
    u32*    block_base;      // [256] exclusive scan of block_total
    int     nblocks_sort;    // radix workgroups for P keys
    int     nblocks_scan;
};
struct BinView {
    u32 *keys_a, *keys_b, *vals_a, *vals_b;   // tile id / Gaussian id per instance (ping-pong)
    uint2*  ranges;          // [T]
    u32*    radix_hist;
    u32*    digit_totals;
    int     nblocks_sort;
    int     passes;          // radix passes over the tile id
    u32*    point_list;      // vals after the last pass (radix path) / the per-tile sorted ids (tile-binned path)
    u32*    tile_sorted;     // keys after the last pass (radix path only)
    u64*    words;           // tile-binned path: (depth key << 32 | Gaussian id) per instance, grouped by tile, unsorted inside a tile
    u64*    scratch;         // tile-binned path: scratch of the two-level / global-memory sorts (lists of > 4096 entries only)
    // A 32-bit word per sorted instance -- bits 0..15 the reach mask of the tile's 4x4-pixel blocks, bit 16 "has a gradient record" --
    // written by blend_fwd when it stages the instance, read back by blend_bwd instead of evaluating block_mask16 again.  Lives in
    // memory that is dead once the lists are sorted: the radix path's spare key buffer (instance i: word i), or -- tile-binned path --
    // the tile's OWN slice of `words` (instance j of a tile that starts at instance s: byte 8 s + 4 j; blend_fwd sorts its tile's
    // list itself, so other tiles' words may still be unsorted when it writes; the second half of the slice holds the compacted
    // positions of the record holders, stock tile rule).
    unsigned short* reach_mask;
};
struct ImgView {
    float* final_T;          // [H*W]
    u32*   n_contrib;        // [H*W]
    // [tiles] the tiles heavy-first by instance count: {tile, first instance, instance count, deepest contributor}.
    // x,y,z by tile_order_kernel, w by blend_fwd; decides which workgroup of a blend launch takes which tile.
    uint4* tile_desc;
    u32*   n_active;         // [1] tiles that hold at least one instance (they come first in tile_desc)
    uint4* tile_aux;         // [tiles] by descriptor slot, stock tile rule on the tile-binned path only (blend_fwd -> blend_bwd):
                             // {record-holding instances in front of the deepest contributor, record-holding instances staged,
                             //  list positions staged, -}
    // tile-binned path (binning.hip): (block of Gaussians, tile) instance counts and their prefix over the blocks
    u32*   cnt_rows;         // [256][(T+1)/2] packed 16-bit counters
    u32*   pre;              // [256][T]
    u32*   tile_total;       // [T]
    uint2* ranges;           // [T]
    u32*   tile_lstart;      // [T] first instance of a tile relative to its group of 64 tiles (tile_prefix_kernel)
    u32*   group_total;      // [512] instances per group of 64 tiles
};

// ---- kernel launches that can carry the stage profiler's events ON THEIR OWN DISPATCH (hipExtLaunchKernelGGL: the timestamps of the
// kernel's completion signal) instead of between two hipEventRecord packets.  The profiler (api.hip: ProfScope) parks a start / stop pair
// in these two thread-local words right before it calls a launcher whose stage is ONE kernel; the first LAUNCH_K of that launcher takes
// them.  Without a pair parked this is hipLaunchKernelGGL.
#include <hip/hip_ext.h>
extern thread_local hipEvent_t g_attach_start, g_attach_stop;
template <typename... KArgs, typename... Args>
static inline void launch_k(void (*kernel)(KArgs...), dim3 grid, dim3 block, size_t lds, hipStream_t st, Args... args)
{
    if (g_attach_start || g_attach_stop) {
        hipEvent_t ea = g_attach_start, eb = g_attach_stop;
        g_attach_start = nullptr; g_attach_stop = nullptr;
        hipExtLaunchKernelGGL(kernel, grid, block, (std::uint32_t)lds, st, ea, eb, 0, static_cast<KArgs>(args)...);
    } else {
        hipLaunchKernelGGL(kernel, grid, block, lds, st, static_cast<KArgs>(args)...);
    }
}
#define LAUNCH_K(kernel, grid, block, lds, st, ...) launch_k(kernel, grid, block, lds, st, __VA_ARGS__)

int radix_items_for(long long n);
int radix_blocks_for(long long n);
size_t carve_geom(void* base, int P, GeomView* v);
size_t carve_binning(void* base, long long I, int W, int H, BinView* v, bool binned = false);
size_t carve_image(void* base, int W, int H, ImgView* v);

// ---- launchers (each enqueues on `st`; returns hipError_t) --------------------------------------------------
// count_into != nullptr (tile-binned path): K1 also counts the (block of Gaussians, tile) matrix and the block-local instance
// offsets (step 1 of binning.hip), grid_x = tiles per image row
hipError_t launch_preprocess_fwd(const BagsSettings& s, const BagsInputs& in, const GeomView& g, int32_t* radii,
                                 float* mean2D, hipStream_t st, const ImgView* count_into = nullptr, int grid_x = 0);
hipError_t launch_radix_sort(const u32* src_k, const u32* src_v, u32* a_k, u32* a_v, u32* b_k, u32* b_v, long long n,
                             int bits, bool iota_vals, u32* hist, u32* totals, int nblocks, hipStream_t st,
                             const u32* n_dev = nullptr);
hipError_t launch_offsets_scan(const GeomView& g, const u32* sorted_ids, int P, hipStream_t st);
hipError_t launch_emit(const GeomView& g, const u32* sorted_ids, int P, int grid_x, u32* keys, u32* vals, u32 capacity,
                       hipStream_t st, const u32* n_dev, uint2* ranges, int T);
hipError_t launch_tile_ranges(const u32* tile_sorted, long long I, uint2* ranges, int T, hipStream_t st,
                              const u32* n_dev, bool cleared);
hipError_t launch_tile_order(const uint2* ranges, int T, uint4* tile_desc, u32* n_active, hipStream_t st);
hipError_t launch_blend_fwd(const BagsSettings& s, const GeomView& g, const BinView& b, const ImgView& im,
                            const BagsForwardOut& out, hipStream_t st, const u32* n_dev = nullptr, u32 capacity = 0, bool sort_here = false);
// binned: the record base of a Gaussian is block_base[line.q3.y] + line.q3.z (K1 wrote both into the geometry line); otherwise
// (radix path) it is gathered from g.inst_off
hipError_t launch_blend_bwd(const BagsSettings& s, const GeomView& g, const BinView& b, const ImgView& im,
                            const float* grad_color, float* partials, bool want_abs, bool binned, hipStream_t st,
                            long long n_records = 0,                      // instance count
                            unsigned char* live_map = nullptr,            // one byte per record (dense-scene mode: the caller's decision), or null
                            hipEvent_t ev_start = nullptr, hipEvent_t ev_stop = nullptr);   // attached to the kernel's dispatch when given
bool bwd_dense_mode(long long n_records, int T, int dense_per_tile);      // does a backward of this size run in dense-scene mode?
hipError_t launch_preprocess_bwd(const BagsSettings& s, const BagsInputs& in, const GeomView& g, const int32_t* radii_or_null,
                                 const float* partials, float* pose_slab, int* nblocks_out, const BagsBackwardArgs& a, hipStream_t st,
                                 bool binned, const unsigned char* live_map = nullptr);
hipError_t launch_pose_reduce(const float* pose_slab, int nblocks, const BagsBackwardArgs& a, hipStream_t st);
hipError_t launch_sh_grad_from_views(int P, int M, int deg, const float* means3D, const BagsShViews& views, float* g_shs, float* g_shs_rest,
                                     int accumulate, hipStream_t st);
// loss.hip: fused L1 + SSIM terms and their image gradient
size_t loss_workspace_bytes(int C, int H, int W);
hipError_t launch_loss_fwd(const float* img, const float* gt, int C, int H, int W, void* ws, float* out_terms, hipStream_t st,
                           bool combined = false, float lambda_dssim = 0.f);
hipError_t launch_loss_bwd(const float* img, const float* gt, int C, int H, int W, const void* ws, const float* grad_terms,
                           float* grad_img, hipStream_t st, bool combined = false, float lambda_dssim = 0.f);
// camera.hip: pose leaves -> viewmatrix / projmatrix / intrinsic / campos, and the adjoint
hipError_t launch_camera_fwd(const float* q0, const float* dq, const float* t0, const float* dt, const float* fovx, const float* fovy,
                             const float* grot, const float* gscale, float znear, float zfar,
                             float* V, float* M, float* K, float* C, hipStream_t st);
hipError_t launch_camera_bwd(const float* q0, const float* dq, const float* t0, const float* dt, const float* fovx, const float* fovy,
                             const float* grot, const float* gscale, float znear, float zfar,
                             const float* gV, const float* gM, const float* gK, const float* gC,
                             float* g_dq, float* g_dt, float* g_fovx, float* g_fovy, float* g_grot, float* g_gscale, hipStream_t st);
// resample.hip: flow upsample + grid_sample + centre crop + mask in one pass, and the adjoint
hipError_t launch_resample_fwd(const float* image, int C, int H, int W, const float* ctrl, int h, int w, int Hf, int Wf, int Hc, int Wc,
                               float* out, float* mask, float* flow_out, hipStream_t st);
hipError_t launch_resample_bwd(const float* image, int C, int H, int W, const float* ctrl, int h, int w, int Hf, int Wf, int Hc, int Wc,
                               const float* grad_out, void* workspace, float* grad_image, float* grad_ctrl, hipStream_t st);
size_t resample_workspace_bytes(int H, int W, int Hc, int Wc);
// activations.hip: cat / sigmoid / exp / normalize of the raw Gaussian parameters, and the adjoint
hipError_t launch_activations_fwd(int P, int K, const float* dc, const float* rest, const float* opacity, const float* scaling,
                                  const float* rotation, float* shs, float* o_opacity, float* o_scales, float* o_rot, hipStream_t st);
hipError_t launch_activations_bwd(int P, int K, const float* dc, const float* rest, const float* opacity, const float* scaling,
                                  const float* rotation, const float* g_shs, const float* g_opacity, const float* g_scales,
                                  const float* g_rot, float* g_dc, float* g_rest, float* g_opacity_raw, float* g_scaling,
                                  float* g_rotation, hipStream_t st);
// knn.hip: mean squared distance to the three nearest neighbours (distCUDA2)
size_t knn_workspace_bytes(int P);
hipError_t launch_knn(const float* pts, int P, void* ws, float* out, hipStream_t st);
// binning.hip: tile-binned instance lists (no global sort)
int binned_per_block(int P);
bool binned_supported(int P, int T);
hipError_t launch_binned_empty(const GeomView& g, const ImgView& im, int T, hipStream_t st);
// the tile descriptor list alone (launch_binned_finish builds it beside the emission): for forwards that emit nothing
hipError_t launch_binned_desc_only(const ImgView& im, int T, hipStream_t st);
// host_count: device-visible address of a pinned host word that also receives the instance count (may be null)
// (the (block, tile) counts themselves come from launch_preprocess_fwd(count_into = &im))
hipError_t launch_binned_prepare(const GeomView& g, const ImgView& im, int P, int grid_x, int T, hipStream_t st, u32* host_count = nullptr,
                                 bool count_now = true);
hipError_t launch_binned_finish(const GeomView& g, const ImgView& im, int P, int grid_x, int T, u64* words, u32 capacity, hipStream_t st,
                                bool deliver_count = false);
hipError_t launch_debug_keys_ranges(const uint2* ranges, const u32* point_list, const u32* depth_key, int T, u64* out, hipStream_t st);
hipError_t launch_debug_keys(const u32* tile_sorted, const u32* point_list, const u32* depth_key, long long I, u64* out, hipStream_t st);
hipError_t launch_unpack_rect(const uint2* rect, int P, u32* out, hipStream_t st);
