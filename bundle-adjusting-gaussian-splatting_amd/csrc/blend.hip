// blend.hip -- K6 / K7: front-to-back alpha compositing and its backward.
//
// MI355X mapping (none of this is the CUDA "256 threads = 256 pixels, atomicAdd per (pixel, splat)" layout):
//   * a 256-thread workgroup (4 wave64) owns a 16x16 tile and consumes its depth-ordered splat list in chunks of 256:
//     every thread gathers one splat (id -> xy, conic, opacity, colour) and computes a 16-bit mask of the 4x4-pixel
//     blocks whose pixels can reach alpha >= 1/255 (conservative: Mahalanobis triangle inequality), so the inner
//     loops only ever touch (block, splat) pairs that can contribute.  Skipped pairs are exactly those the per-pixel
//     test would reject => images and gradients are unchanged by the culling.
//   * FORWARD  (blend_fwd_rows_kernel): lane = pixel.  Each 16-lane DPP row owns one 4x4 block and walks its own
//     ballot-compacted list; no scalar branches inside the walk.
//   * BACKWARD (blend_bwd_scan_kernel): lane = splat.  The per-pixel recurrences of compositing (transmittance in
//     front of a splat, colour behind it) are wave64 DPP prefix scans over a block's list, so every lane owns its
//     splat's 11 gradient sums outright: no cross-lane reduction and NO atomics.  One 48-byte record per sorted
//     instance is written at the instance's emission slot; preprocess_bwd sums each Gaussian's consecutive records.
//     Gradients are bitwise reproducible run to run.
// PMC history that led here is in profiles/r01 and DESIGN.md section 5 (the first lane = pixel kernels saturated VALU
// issue at 96 % with a third of the lanes contributing; they are in the git history).
#include "bags_common.h"
#include "tile_sort.h"
#include <type_traits>

#define LOG2E 1.4426950408889634f
#define ALPHA_MIN (1.0f / 255.0f)
#define T_EPS 0.0001f

// Sizes that tools/sweep_blend.sh sweeps with -D...; every other switch this file once had lost its A/B and is gone (the
// measurements are in profiles/r0*/ab_*.txt and DESIGN.md section 3, the code in the git history).
#define CHUNK 256             // forward: splats staged per chunk (255 + the all-zero sentinel record in slot 255)
// The backward's chunk.  A wave's LDS copy of the per-splat sums holds only the staged splats that REACH its quadrant (56 % of them on the
// bench scene), indexed by their rank among those: BSLOTS slots per copy for BCHUNK staged splats.  A chunk in which more than BSLOTS splats
// reach some quadrant runs its group phase twice, once per half (blend_bwd_scan_kernel).  Until round 5 every copy held a slot for every
// staged splat: 176-splat chunks in the same LDS (4 x 48 B x 176 = 33 of 51 KB); chunks of 128 / 144 / 160 / 176 measured 0.385 / 0.385 /
// 0.376 / 0.371 ms then, and the longer ones now: profiles/r05/ab_blend_bwd.txt section 11 (lane fill 0.826 -> 0.85, a fifth fewer chunks).
// Two geometries, chosen per launch by the scene's instances per tile (BWD_SPARSE_PER_TILE): 224 / 176 overflows only where 79 % of a
// full chunk reach one quadrant and is never slower than 176 / 176 was (sm 0.5 .. 2.0); 240 / 168 is another 1 % faster on scenes of small
// splats and up to 6 % slower on dense ones.  LDS: 53.8 KB is the most a workgroup may hold at three per CU (54.2 KB measured: two per CU).
#ifndef BCHUNK
#define BCHUNK 224
#endif
#ifndef BSLOTS
#define BSLOTS 176
#endif
#ifndef WCHUNK
#define WCHUNK 240            // ... on sparse scenes
#endif
#ifndef WSLOTS
#define WSLOTS 168
#endif
#ifndef BWD_SPARSE_PER_TILE
#define BWD_SPARSE_PER_TILE 300   // instances per tile (scene average) up to which WCHUNK / WSLOTS are used (bench scene: 254; sm 0.625: 325)
#endif
#ifndef SCAN_WG_PER_CU
#define SCAN_WG_PER_CU 3      // backward workgroups per CU: 3 x 53 KB of LDS, 160 VGPRs (measured: 128/3 beats 256/2 by 3 %)
#endif
#define FWD_WG_PER_CU 6       // forward workgroups per CU (= waves per SIMD): 80 VGPRs (8 / 7 / 6: 148.3 / 146.9 / 144.3 us)
#define PRIO_SERIAL 3         // backward: wave priority in the serial section between a chunk's two barriers ...
#define PRIO_GROUPS 0         // ... and in its list-building / scan phase (0/0, 1/0, 3/1 measured: flat)
#define PQ 3                  // float4s per pixel pair in the backward's LDS image (below)
#define BWD_DENSE_PER_TILE 480          // instances per tile (scene average) above which the backward runs in dense-scene mode: a byte per
                                        // record says whether blend_bwd wrote it, nobody writes or reads a zero record
                                        // (BagsBackwardArgs.dense_per_tile overrides it; profiles/r05/ab_dense.txt)
struct __attribute__((aligned(16))) SplatRec {
    float x, y, ap, cp;       // centre, pre-scaled conic: exp2(ap dx^2 + bp dx dy + cp dy^2) == exp(power)
    float bp, o, r, g;
    float b, z; u32 pos; u32 mask;   // pos: 1-based position in the tile list; mask: 4x4 blocks reachable (bit by*4+bx)
};

// Workgroup -> tile.  Workgroups are dispatched in blockIdx order, round-robin over the 8 XCDs (b and b+8 share one), and
// a launch ends when its busiest XCD ends.  tile_order_kernel (sort.hip) lists the tiles heavy-first by instance count
// as descriptors {tile, first instance, instance count, deepest contributor (written by the forward)}; virtual block v
// takes descriptor slot_of_vblock(v): runs of TILE_ILV consecutive descriptors go round-robin over the XCDs, so
//   * every XCD gets the same mix of heavy and light tiles (one contiguous band of tiles per XCD, the first layout, left
//     the two XCDs holding the top and bottom of the bench image with 0.62-0.67x the mean load and the other six with
//     1.12x: the launch ran 12-20 % longer than a balanced one, tests/analysis_lane_fill.py),
//   * the long tiles start first and the launch does not end on a few late heavy ones,
//   * descriptors of a run are mostly neighbouring tiles, which share splats, and stay on one L2.
// Placement only affects speed: every tile is computed independently of where and when it runs.
#ifndef TILE_ILV
#define TILE_ILV 16
#endif
#define TILE_RUN (8 * TILE_ILV)        // virtual blocks [k TILE_RUN, (k+1) TILE_RUN) permute descriptors of the same interval
__device__ __forceinline__ int slot_of_vblock(int v)
{
    const int k = v >> 3;
    return ((k / TILE_ILV) * 8 + (v & 7)) * TILE_ILV + (k % TILE_ILV);
}

// exp(power) for one (pixel, splat) pair: the SAME instruction sequence in forward and backward so both make the
// same contribute / skip decision.
__device__ __forceinline__ float pair_power2(float dx, float dy, float ap, float bp, float cp)
{
    const float t = __fmaf_rn(bp, dy, __fmul_rn(ap, dx));
    const float u = __fmul_rn(__fmul_rn(cp, dy), dy);
    return __fmaf_rn(dx, t, u);
}

typedef float f2 __attribute__((ext_vector_type(2)));
// number of set bits of a wave ballot below this lane: v_mbcnt_lo + v_mbcnt_hi, no per-lane mask register
__device__ __forceinline__ u32 ballot_rank(u64 bal) { return __builtin_amdgcn_mbcnt_hi((u32)(bal >> 32), __builtin_amdgcn_mbcnt_lo((u32)bal, 0u)); }

// Workgroup barrier that orders LDS traffic only.  __syncthreads() also drains the vector-memory counter
// (s_waitcnt vmcnt(0)), which would stall every wave on the gathers it has just put in flight for the NEXT chunk
// (measured: 23 % of the wave time parked at the first barrier).  Nothing exchanged between waves here lives in
// global memory, so lgkmcnt(0) + s_barrier is sufficient.
__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

// Inclusive prefix composition of affine maps x -> a x + b over each 16-lane DPP row, four independent chains
// interleaved (a dependent DPP op needs two wait states after the VALU write it reads; seven other instructions sit
// between here).  Lane l ends with (A_l, b_l) such that F_l(F_{l-1}(...F_0(x))) = A_l x + b_l, lane 0 innermost.
// A step with shift d combines lane l with lane l-d:  b_l += a_l * b_{l-d};  a_l *= a_{l-d}  (b first: it needs the old a_l).
// Lanes whose source falls outside the row are not written (bound_ctrl off), which is the identity they need.
// The scan works on register PAIRS pinned to v[152:159]: the values on either side of it are packed (v_pk_*_f32 needs even-aligned
// pairs) while DPP instructions name single registers, and an asm operand cannot name half of a pair -- with free operands the
// compiler splits and re-joins the pairs by copies (134 -> 132 vector instructions per row step, 355.9 -> 352.7 us).
#define AFF4P_STEP(PAT)                                                                                  \
        "v_fmac_f32_dpp v156, v156, v152 " PAT "\n\t" "v_fmac_f32_dpp v157, v157, v153 " PAT "\n\t"     \
        "v_fmac_f32_dpp v158, v158, v154 " PAT "\n\t" "v_fmac_f32_dpp v159, v159, v155 " PAT "\n\t"     \
        "v_mul_f32_dpp v152, v152, v152 " PAT "\n\t"  "v_mul_f32_dpp v153, v153, v153 " PAT "\n\t"      \
        "v_mul_f32_dpp v154, v154, v154 " PAT "\n\t"  "v_mul_f32_dpp v155, v155, v155 " PAT "\n\t"
__device__ __forceinline__ void scan_affine16x4_pinned(f2& A01, f2& A23, f2& O01, f2& O23)
{
    asm volatile(
        "s_nop 1\n\t"
        AFF4P_STEP("row_shr:1 row_mask:0xf bank_mask:0xf")
        AFF4P_STEP("row_shr:2 row_mask:0xf bank_mask:0xf")
        AFF4P_STEP("row_shr:4 row_mask:0xf bank_mask:0xf")
        AFF4P_STEP("row_shr:8 row_mask:0xf bank_mask:0xf")
        "s_nop 1"
        : "+{v[152:153]}"(A01), "+{v[154:155]}"(A23), "+{v[156:157]}"(O01), "+{v[158:159]}"(O23));
}
// r_k <- v_k of the next lower lane of the row; the row's lane 0 keeps the value r_k came in with (the carry)
__device__ __forceinline__ void shift_up16x4(float& r0, float& r1, float& r2, float& r3, float v0, float v1, float v2, float v3)
{
    asm volatile(
        "s_nop 1\n\t"
        "v_mov_b32_dpp %0, %4 row_shr:1 row_mask:0xf bank_mask:0xf\n\t"
        "v_mov_b32_dpp %1, %5 row_shr:1 row_mask:0xf bank_mask:0xf\n\t"
        "v_mov_b32_dpp %2, %6 row_shr:1 row_mask:0xf bank_mask:0xf\n\t"
        "v_mov_b32_dpp %3, %7 row_shr:1 row_mask:0xf bank_mask:0xf"
        : "+v"(r0), "+v"(r1), "+v"(r2), "+v"(r3) : "v"(v0), "v"(v1), "v"(v2), "v"(v3));
}

// ================================================================================================================
// backward
// Scan-based backward ("lane = splat").  PMC showed a lane = pixel backward saturates VALU issue (96 % of SIMD cycles)
// and spends most of it on 64-wide work where a third of the lanes contribute, plus an 11-value cross-lane reduction
// per (tile, splat).  Here the roles are swapped:
//   * a 256-thread workgroup owns a tile; the tile list is consumed back to front in chunks of BCHUNK (224 / 240) splats, staged
//     once in LDS together with a 16-bit mask of the 4x4-pixel blocks each splat's alpha >= 1/255 ellipse can reach
//     (Mahalanobis triangle-inequality test, conservative; evaluated by the forward, which leaves it next to the sorted list);
//   * wave w owns quadrant w and ballot-compacts, per block, the chunk's splats that reach it;
//   * each 16-lane DPP row of the wave owns one block and takes 16 entries of its list per step (deepest in the row's
//     lane 0); the block's pixels are visited two at a time with packed fp32 math.  With x_i = c_i . dL/dC, the per-pixel
//     recurrences of alpha compositing are ONE prefix scan of affine maps over the row's lanes:
//         F_i(x) = alpha_i x_i + (1 - alpha_i) x          (what is seen at the front face of splat i, given x behind it)
//         A_i = prod_{j at or behind i} (1 - alpha_j)      T_i = T_final / (A_i * carry)   (transmittance in front of i)
//         R_i = (F_{i+1} o F_{i+2} o ...)(bg . dL/dC)      dL/dalpha_i = T_i (x_i - R_i)
//     so no division by (1 - alpha) is needed.  (A, R) behind the group is carried through one pair per pixel in LDS;
//   * every lane then owns its splat's 11 sums outright: no cross-lane reduction, no atomics.  Each wave adds into its
//     own LDS copy of the sums -- a slot per staged splat that reaches the wave's quadrant, all four rows at once, each on
//     another quarter of its slot's record (a splat can sit in several rows: "rotating quarters"); the four copies are
//     added in fixed order => bitwise reproducible.
//   * the gathers of chunk k+1 and the ids of chunk k+2 are in flight under the list building and the step loop of chunk k;
//     nothing in the loop may copy or use a loaded register before the point of consumption (profiles/r05/ab_blend_bwd.txt 12).
// ================================================================================================================
struct StagedSplat { float x, y, ap, bp, cp, o, r, g, b; u32 pos; };   // what the row loop reads of a staged splat (pos: 1-based list position)
struct __attribute__((aligned(16))) ChunkRec {
    float x, y, ap, bp;
    float cp, o, r, g;
    float b; u32 pos; u32 mask; u32 e;         // mask: 4x4 blocks reachable (bit by*4+bx); e: emission slot
};
// per pixel PAIR (two horizontally adjacent pixels A,B of one block row), PQ float4s:
//   TF_FOLD: [g0A g0B g1A g1B] [g2A g2B ncA ncB] [AcA AcB RcA RcB]                 (Ac starts at 1 / T_final)
//   + PIX_NARROW: [g0A g0B g1A g1B] [g2A g2B AcA AcB] [RcA RcB ncA ncB]
//   else:    [g0A g0B g1A g1B] [g2A g2B TfA TfB] [ -    -   ncA ncB] [AcA AcB RcA RcB]     (Ac, Rc: carries behind the group)

// blocks of the tile a splat can reach with alpha >= 1/255 (conservative; exactness comes from the per-pixel test).
// Branch-free: one thread evaluates all 16 blocks (the lanes of a wave hold unrelated splats, so early-outs would only
// serialise).  Block (bx,by) is reachable iff the splat's bounding box overlaps it AND the Q-norm distance from the
// splat centre to the block centre is within sqrt(2 ln(255 o)) + r_block (triangle inequality in the Q-norm).
__device__ __forceinline__ u32 block_mask16(float x, float y, float a, float b, float c, float o, float X0, float Y0)
{
    const float vis = 255.0f * o;
    const float det = a * c - b * b;
    const bool degenerate = !(det > 0.f) || !(a > 0.f) || !(c > 0.f);
    const float tau2 = 2.0f * (fmaxf(__logf(fmaxf(vis, 1e-30f)), 0.f) + 1e-3f);
    const float idet = __builtin_amdgcn_rcpf(det);
    const float hx = sqrtf(tau2 * c * idet) * 1.001f + 0.05f;
    const float hy = sqrtf(tau2 * a * idet) * 1.001f + 0.05f;
    // Q-norm radius of a 4x4 block around its centre: corners at (+-1.5, +-1.5)
    const float qd = 2.25f * (a + c), qo = 4.5f * b;
    const float rb = sqrtf(fmaxf(qd + qo, qd - qo));
    const float lim = sqrtf(tau2) * 1.001f + rb + 1e-3f;
    const float lim2 = lim * lim;
    const float xl = x - hx - X0, xh = x + hx - X0, yl = y - hy - Y0, yh = y + hy - Y0;   // box relative to the tile
    const float ex = (X0 + 1.5f) - x, ey = (Y0 + 1.5f) - y;                               // centre of block (0,0) - splat
    const float b2 = 2.f * b;
    u32 m = 0;
#pragma unroll
    for (int by = 0; by < 4; ++by) {
        const float dy = ey + 4.f * by;
        const float cdy = c * dy * dy, bdy = b2 * dy;
        const bool rowok = (yh >= 4.f * by) && (yl <= 4.f * by + 3.f);
#pragma unroll
        for (int bx = 0; bx < 4; ++bx) {
            const float dx = ex + 4.f * bx;
            const float Q = __fmaf_rn(dx, __fmaf_rn(a, dx, bdy), cdy);                    // squared Q-norm of centre offset
            const bool ok = rowok && (xh >= 4.f * bx) && (xl <= 4.f * bx + 3.f) && (Q <= lim2);
            m |= ok ? (1u << (by * 4 + bx)) : 0u;
        }
    }
    if (degenerate) m = 0xFFFFu;
    if (!(vis >= 0.99f)) m = 0u;
    return m;
}

// Two diagnostic builds (tools/diag_phases.py, tools/diag_pairs.sh; tests/test_abi_cpu.py compiles both so that they keep building):
// PH(...) / PH_MARK(i) exist only with -DDIAG_PHASES, DG(...) only with -DDIAG_PAIRS.
#ifdef DIAG_PHASES
__device__ unsigned long long g_phase_cycles[8];
#define PH(...) __VA_ARGS__
#define PH_MARK(i) do { const unsigned long long t_ = __builtin_amdgcn_s_memtime(); ph[i] += t_ - tlast; tlast = t_; } while (0)
#else
#define PH(...)
#define PH_MARK(i) do {} while (0)
#endif
#ifndef DIAG_PAIRS
#define DG(...)
#else
#define DG(...) __VA_ARGS__
// Diagnostic build (tools/diag_pairs.sh): (pixel, splat) pairs the blend kernels EVALUATE (every pixel of every 4x4 block a
// staged splat's reach mask admits) against the pairs that CONTRIBUTE (pass the alpha / power / n_contrib tests), summed
// over all launches since the last read: [0] backward evaluated, [1] backward contributing, [2] / [3] the same, forward.
__device__ unsigned long long g_pair_counts[8];   // [4] backward (splat, block) entries, [5] entries without a contributing pixel, [6] wave steps, [7] chunks x waves
extern "C" void bags_diag_pairs(unsigned long long* out, int reset)
{
    hipMemcpyFromSymbol(out, HIP_SYMBOL(g_pair_counts), sizeof(unsigned long long) * 8);
    if (reset) { const unsigned long long z[8] = {0, 0, 0, 0, 0, 0, 0, 0}; hipMemcpyToSymbol(HIP_SYMBOL(g_pair_counts), z, sizeof(z)); }
}
__device__ __forceinline__ void diag_pairs_flush(int slot, u32 ev, u32 co)
{
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) { ev += (u32)__shfl_xor((int)ev, d); co += (u32)__shfl_xor((int)co, d); }
    if ((threadIdx.x & 63) == 0) { atomicAdd(&g_pair_counts[slot], (unsigned long long)ev); atomicAdd(&g_pair_counts[slot + 1], (unsigned long long)co); }
}
#endif
// `power <= 0` cannot fail for a splat whose conic is well conditioned.  With Q = [[a, b], [b, c]] the fp32 numbers in the
// geometry line, the evaluated power is p^ = fma(dx, fma(bp, dy, ap dx), (cp dy) dy) on coefficients that are -K Q rounded
// (K = log2(e) / 2): |p^ - p| <= ~4 eps K (a dx^2 + 2 |b| |dx dy| + c dy^2) <= 8 eps K (a + c) |d|^2, while -p >= K lambda_min
// |d|^2 with lambda_min >= det / (a + c).  So det >= 1e-4 (a + c)^2 leaves a factor 200 between the rounding error and the
// value: p^ < 0 for d != 0, and p^ = 0 for d = 0.  The forward ORs "some staged splat fails that bound" (conic eigenvalue
// ratio above ~1e4: a needle more than 100 x longer than wide after the 0.3 px dilation) into bit 30 of the tile descriptor.
__device__ __forceinline__ bool conic_ill_conditioned(float a, float b, float c)
{
    const float tr = a + c;
    return !(a * c - b * b >= 1.0e-4f * (tr * tr));         // also catches NaN and non-positive determinants
}
// Is tile (tx, ty) among the tiles the Gaussian has a partial-gradient record for?  rc / kp: rectangle and tile mask of the
// geometry line (the RECORDS' tiles: with the stock tile rule on the tile-binned path a subset of the tiles the lists cover).
__device__ __forceinline__ bool tile_has_record(const u32 rc_x, const u32 rc_y, const u64 kp, const int tx, const int ty)
{
    const int minx = (int)(rc_x & 0xFFFF), miny = (int)(rc_x >> 16), rw = (int)(rc_y & 0xFFFF) - minx, rh = (int)(rc_y >> 16) - miny;
    const int dx = tx - minx, dy = ty - miny;
    if (!((dx >= 0) && (dy >= 0) && (dx < rw) && (dy < rh))) return false;
    return rect_small(rw, rh) ? (((kp >> (dy * 8 + dx)) & 1ull) != 0ull) : true;
}
struct TileRef { int tx, ty; u32 rx, n, maxc; bool early, needle; };   // early: some pixel of the tile stopped before its list ended (or lies outside the image)   // wave-uniform: tile coordinates, first instance, instances, deepest contributor

// COMPACT (stock tile rule on the tile-binned path): the chunks are staged from the forward's compacted list of record-holding
// positions (tile_aux, cpos) instead of from consecutive list positions; everything that needs a list POSITION (the pos <=
// n_contrib test, the per-block "behind the last contributor" filter) takes it from the compacted entry.
template <bool ABS, bool COMPACT, bool SPARSE>
__global__ void __launch_bounds__(256, SCAN_WG_PER_CU)
blend_bwd_scan_kernel(int W, int H, int grid_x, int T, const uint4* __restrict__ tile_desc,
                      const u32* __restrict__ point_list, const unsigned char* __restrict__ reach_mask, const u32 rm_stride,
                      const float4* __restrict__ g2d, const u32* __restrict__ inst_off, const u32* __restrict__ block_base,
                      const float* __restrict__ bg, const float* __restrict__ final_T, const u32* __restrict__ n_contrib,
                      const float* __restrict__ grad_color, float* __restrict__ partials,
                      const int test_keep, const uint4* __restrict__ tile_aux, unsigned char* __restrict__ live_map)
{
    // One workgroup per tile that holds at least one instance, heavy tiles first (slot_of_vblock).  Tried and dropped:
    // persistent workgroups that run the chunk pipeline over the flattened (tile, chunk) sequence, with the next tile's
    // ids and gathers in flight during the current tile's last chunk (static b, b+G, ... walk or per-XCD ticket counters).
    // Every variant was 2-8 % SLOWER than one workgroup per tile although the per-tile dependent chain disappears from
    // the wave timeline: the hardware dispatcher refills a CU as soon as any workgroup leaves, while a persistent
    // workgroup keeps its four waves coupled at two barriers per chunk for the whole launch.
    constexpr int CH = SPARSE ? WCHUNK : BCHUNK;                              // splats staged per chunk
    constexpr int SL = (SPARSE ? WSLOTS : BSLOTS) - (COMPACT ? 8 : 0);        // slots per wave copy of the sums (COMPACT: chunk_pos needs the LDS of two slots x 4)
    static_assert(CH > 128 && CH <= 256 && 128 <= SL && SL <= 255, "a chunk is four 64-slot segments; each half of it must fit a wave copy; indices are bytes");
    constexpr int NSEG = (CH + 63) / 64;                     // 64-slot segments of a chunk: thread tid stages slot tid of segment tid >> 6
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    // The chunk loop never reads `tid`: a thread index that is live across the row loop (which uses every register) is spilled, and its
    // reload at the loop's back edge puts an s_waitcnt vmcnt(0) -- a drain of the chunk's record stores and of the id prefetch -- at the
    // top of every chunk.  Each section of the loop forms the index anew from the (scalar) wave number and v_mbcnt.
    auto tid_now = [&]() -> int {
        int l;
        asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\n\tv_mbcnt_hi_u32_b32 %0, -1, %0" : "=v"(l));
        return (wave << 6) | l;
    };
    const int dslot = slot_of_vblock((int)blockIdx.x);
    if (dslot >= T) return;
    const uint4 desc = tile_desc[dslot];                     // {tile, first instance, instances, deepest contributor}
    if (desc.z == 0) return;                                 // empty tile: no records to write
    TileRef A;
    A.tx = (int)(desc.x % (u32)grid_x); A.ty = (int)(desc.x / (u32)grid_x); A.rx = desc.y; A.n = desc.z;
    A.maxc = min(desc.w & 0x3FFFFFFFu, desc.z); A.early = (desc.w >> 31) != 0u; A.needle = ((desc.w >> 30) & 1u) != 0u;
    // where the chunks start (back to front): the deepest contributor's list position, or -- COMPACT -- the number of
    // record-holding instances in front of it
    u32 hi0 = A.maxc, n_live = 0, n_staged = A.n;
    if (COMPACT) { const uint4 ax = tile_aux[dslot]; hi0 = min(ax.x, A.n); n_live = min(ax.y, A.n); n_staged = min(ax.z, A.n); }
    const u32* const cposp = reinterpret_cast<const u32*>(reach_mask + (size_t)A.rx * 8u) + A.n;   // COMPACT: positions of the record holders

    // the staged records as the row loop reads them: (x y ap bp) (cp o r g) (b) -- 36 bytes per slot; the emission slot and the reach mask
    // stay with the thread that staged the slot
    __shared__ float4 recA[CH], recB[CH];
    __shared__ float recC[CH];
    // 8 pixel pairs (PixPair = 4 x float4) per block + one float4 of padding: the four rows of a wave read four different
    // blocks in one ds_read_b128, and a 512-B block stride would put all four on the same banks
    __shared__ float4 pixq[16][8 * PQ + 1];          //  6.25 KB (8.25 without TF_FOLD)
    __shared__ unsigned char lists[16][CH];
    __shared__ u32 masks[CH];                        // block reach masks
    __shared__ float acc[4][SL][12];                 // 33 KB, one copy per wave (LDS float atomics: 2.7x slower, profiles/r05/ab_blend_bwd.txt)
    __shared__ u32 blk_maxc[16];                     // last contributor over the 16 pixels of each block
    __shared__ u32 chunk_pos[COMPACT ? CH : 1];      // COMPACT: list position of every staged slot (the slots are no longer consecutive positions)
    // a slot's index in wave w's copy = its rank among the chunk's slots that reach quadrant w: rk4[slot] holds the four ranks INSIDE the slot's
    // 64-slot segment (a byte each, from a ballot of the staging wave), segtot[segment] the segment's four totals
    __shared__ u32 rk4[CH];
    __shared__ __attribute__((aligned(16))) u32 segtot[4];

    PH(unsigned long long ph[8] = {0, 0, 0, 0, 0, 0, 0, 0};)
    PH(unsigned long long tlast = __builtin_amdgcn_s_memtime();)
    struct Raw { float4 q0, q1, q2; u32 io, blk, kpl, kph; };   // conic+opacity | x y r g | b z rect | record offset (in its block) | block | tile mask (two halves)
    // id (x) and 4x4-block reach mask (y, as the forward staged it) of this thread's slot in the chunk of a tile's list that ends at hi_
    // COMPACT: entry (hi_ - c_) + tid of the compacted list -> its list position (fetch_pos) -> id, reach word (fetch_id with that position):
    // two dependent loads.
    auto fetch_pos = [&](u32 hi_, const int tid) -> u32 {
        const u32 c_ = min(hi_, (u32)CH);
        return ((u32)tid < c_) ? cposp[(hi_ - c_) + tid] : 0u;
    };
    auto fetch_id = [&](u32 rx_, u32 hi_, u32& pos_, const int tid) -> uint2 {      // (COMPACT: pos_ comes IN, from fetch_pos)
        const u32 c_ = min(hi_, (u32)CH);
        if (!COMPACT) pos_ = 0u;
        if ((u32)tid >= c_) return make_uint2(0xFFFFFFFFu, 0u);
        if (COMPACT) {
            const u32 info = reinterpret_cast<const u32*>(reach_mask + (size_t)rx_ * 8u)[pos_];
            return make_uint2(point_list[rx_ + pos_], info);
        }
        const u32 at = rx_ + (hi_ - c_) + tid;
        // The forward left a 32-bit word per staged instance -- bits 0..15 the reach mask, bit 16 "this instance has a record" (with the
        // stock tile rule the lists hold instances whose tile the opacity rule drops: no record, nothing to gather, nothing to write;
        // they come back as an empty slot) -- at byte rm_stride * (the tile's first instance) of a buffer that is dead by then: the tile's own
        // slice of the unsorted words (tile-binned path: 8 bytes per instance), the key half the last radix pass read (radix path: 4).
        // Word and id are loaded side by side, no branch between them: behind a branch on the word the id load waits for it -- an
        // s_waitcnt vmcnt(0) that also drains the gathers just issued for the chunk before, once per chunk (so it was until round 5,
        // when the radix path still kept 16-bit masks and the two layouts were told apart here).
        const u32 info = reinterpret_cast<const u32*>(reach_mask + (size_t)rx_ * rm_stride)[at - rx_];
        return make_uint2(point_list[at], info);             // raw: decode_id() where the pair is consumed
    };
    // (id, word) as fetched -> (id or "empty slot", reach mask).  Arithmetic, not a select: the compiler turns `word says record ? id : none`
    // back into a branch around the id load.  Applied where the pair is CONSUMED, a chunk after the loads were issued: any arithmetic on
    // a loaded value sits right behind its load in the instruction stream, with the wait
    // `one`: the number 1 through an empty asm at the point of consumption, so that none of this can be scheduled above that point
    auto decode_id = [&](const uint2 raw, const u32 one) -> uint2 {
        const u32 lo16 = (one << 16) - one;
        if (COMPACT) return make_uint2(raw.x, raw.y & lo16);
        const u32 has = (raw.y >> (15u + one)) & one;
        return make_uint2(raw.x | (has - one), raw.y & lo16 & (0u - has));
    };
    // The geometry line of a staged instance, straight into the registers it stays in.  Lessons of the ISA, all of the same kind -- a
    // register COPY of a loaded value waits for the load, i.e. drains the gathers where they were issued instead of under the list building:
    //   * no branch around the loads: an empty slot (g = ~0) reads line 0 -- nobody looks at it: its reach mask is 0 and its emission slot
    //     "none" (with zeroed registers on the other path the two meet in copies);
    //   * the fourth float4 word by word, the tile mask as two halves (joining them is a copy), and not merged into a wide load whose dead
    //     components the register allocator would lend out as scratch while the load is in flight;
    //   * the record offset -- tile-binned path: in the line since K1 counts the (block, tile) matrix itself (round 4; the gather of
    //     inst_off[g] was a second line miss per instance); radix path: the inst_off array -- by ONE load through a selected address.
    auto fetch = [&](u32 g) {
        Raw r;
        const u32 gc = (g != 0xFFFFFFFFu) ? g : 0u;
        const float4* rec = g2d + 4 * (size_t)gc;              // one 64-byte line
        r.q0 = rec[0]; r.q1 = rec[1]; r.q2 = rec[2];
        const u32* q3 = reinterpret_cast<const u32*>(rec + 3);
        u32 o1 = 1u, o3 = 3u;
        asm volatile("" : "+v"(o1), "+v"(o3));               // (opaque offsets: the four words are not to be merged into wider loads)
        r.kpl = q3[0]; r.kph = q3[o3];
        r.blk = q3[o1];
        r.io = *(inst_off ? inst_off + gc : q3 + 2);
        return r;
    };
    // emission slot of a Gaussian's record for tile (tx, ty): the tile's rank among the tiles the Gaussian emits (its
    // rectangle y outer, x inner, minus what the tile mask of a small rectangle drops)
    auto emission_slot = [&](u32 io, uint2 rc, u64 kp, int tx, int ty) -> u32 {
        const int minx = (int)(rc.x & 0xFFFF), miny = (int)(rc.x >> 16);
        return io + rect_tile_rank(kp, (int)(rc.y & 0xFFFF) - minx, (int)(rc.y >> 16) - miny, tx - minx, ty - miny);
    };
    // emission slot of a fetched line for this tile.  Tile-binned path: the block base comes from a 1 KB table (L1-resident), a
    // dependent load -- so this is evaluated once the line has landed and long before the slot is needed (after the list
    // building of the chunk before), and from then on ONE register stands for six (offset, block, rectangle, tile mask).
    auto slot_of = [&](const Raw& rw, const bool have, const TileRef& t) -> u32 {
        if (!have) return 0xFFFFFFFFu;                        // an empty slot of the chunk: no record is written for it
        const u32 base_ = inst_off ? 0u : block_base[rw.blk];
        return emission_slot(base_ + rw.io, make_uint2(__float_as_uint(rw.q2.z), __float_as_uint(rw.q2.w)), (u64)rw.kpl | ((u64)rw.kph << 32), t.tx, t.ty);
    };
    // (c_half = -0.5 log2(e), c_one = -log2(e): through an empty asm at the point of consumption, for the same reason as decode_id's `one`)
    auto make_rec = [&](const Raw& rw, const u32 e_, const u32 mask_, const TileRef& t, u32 lo_, u32 cnt_, const u32 pos_, const int tid,
                        const float c_half, const float c_one) {
        ChunkRec rec; rec.mask = 0; rec.e = 0;
        rec.x = rec.y = rec.ap = rec.bp = rec.cp = rec.o = rec.r = rec.g = rec.b = 0.f; rec.pos = 0;
        if ((u32)tid < cnt_) {
            const float2 c2 = make_float2(rw.q1.x, rw.q1.y); const float4 co = rw.q0;
            const float4 cz = make_float4(rw.q1.z, rw.q1.w, rw.q2.x, rw.q2.y);
            rec.e = e_;
            rec.x = c2.x; rec.y = c2.y;
            rec.ap = c_half * co.x; rec.bp = c_one * co.y; rec.cp = c_half * co.z; rec.o = co.w;
            rec.r = cz.x; rec.g = cz.y; rec.b = cz.z;
            rec.pos = (COMPACT ? pos_ : lo_ + tid) + 1;
            rec.mask = mask_;                 // block_mask16 of this (tile, splat), evaluated once: by the forward
        }
        return rec;
    };
    // Per-pixel constants and scan carries into LDS, per-block deepest contributor, and zero records for the instances
    // behind every pixel's last contributor (they are never visited).  The first chunk's first barrier publishes it.
    auto tile_begin = [&](const TileRef& t) {
        {   // thread tid <-> block tid>>4, pixel tid&15 (ix = &3, iy = >>2)
            const int b = tid >> 4, i = tid & 15;
            const int px = t.tx * BAGS_TILE + (b & 3) * 4 + (i & 3);
            const int py = t.ty * BAGS_TILE + (b >> 2) * 4 + (i >> 2);
            const bool in = (px < W) && (py < H);
            const size_t HW = (size_t)W * H, pixi = (size_t)min(py, H - 1) * W + min(px, W - 1);   // always a valid address
            const float l0 = grad_color[pixi], l1 = grad_color[HW + pixi], l2 = grad_color[2 * HW + pixi];
            const float lT = final_T[pixi];
            const u32 lnc = n_contrib[pixi];
            const float g0 = in ? l0 : 0.f, g1 = in ? l1 : 0.f, g2 = in ? l2 : 0.f;
            const float Tf = in ? lT : 1.f;
            const u32 nc = in ? lnc : 0u;
            const float bgg = bg[0] * g0 + bg[1] * g1 + bg[2] * g2;          // what lies behind the deepest splat
            float* pp = reinterpret_cast<float*>(&pixq[tid >> 4][((tid >> 1) & 7) * PQ]);
            const int h = tid & 1;                        // A or B of the pair
            pp[0 + h] = g0; pp[2 + h] = g1; pp[4 + h] = g2; pp[6 + h] = 1.0f / Tf;   // T_final >= 1e-6: a pixel stops before T falls below 1e-4 and alpha <= 0.99
            pp[8 + h] = bgg; pp[10 + h] = __uint_as_float(nc);
            u32 m = nc;
#pragma unroll
            for (int d = 8; d >= 1; d >>= 1) m = max(m, (u32)__shfl_xor((int)m, d));   // 16 consecutive threads = one block
            if ((tid & 15) == 0) blk_maxc[b] = m;
        }
        // zero records for the record-holding instances no chunk will visit: those at or behind the deepest contributor.
        // COMPACT: first the staged ones (compacted entries hi0 .. n_live), then, below, the positions the forward never staged
        // Dense scenes (thousands of entries per tile, the walk ends after a fraction of them): no zero records at all.  live_map
        // holds one byte per record, cleared by the launcher (I bytes instead of 48 I); the write-out below marks the records
        // it writes and preprocess_bwd sums only those.  (Round 4 cleared the whole record array with one memset instead:
        // 700 MB written here and read back there at 1800 entries per tile.)
        if (live_map) return;
        if (COMPACT) {
            for (u32 k = hi0 + tid; k < n_live; k += 256) {
                const u32 g = point_list[t.rx + cposp[k]];
                const float4 t2 = g2d[4 * (size_t)g + 2], t3 = g2d[4 * (size_t)g + 3];
                const u32 io = block_base[__float_as_uint(t3.y)] + __float_as_uint(t3.z);
                const u32 e = emission_slot(io, make_uint2(__float_as_uint(t2.z), __float_as_uint(t2.w)),
                                            (u64)__float_as_uint(t3.x) | ((u64)__float_as_uint(t3.w) << 32), t.tx, t.ty);
                float4* dst = reinterpret_cast<float4*>(partials + (size_t)e * PART_FLOATS);
                const float4 z4 = make_float4(0.f, 0.f, 0.f, 0.f);
                dst[0] = z4; dst[1] = z4; dst[2] = z4;
            }
        }
        for (u32 p = (COMPACT ? n_staged : t.maxc) + tid; p < t.n; p += 256) {
            // (the forward stops staging once every pixel of the tile has finished: these instances may have no reach word)
            const u32 g = point_list[t.rx + p];
            const float4 t2 = g2d[4 * (size_t)g + 2], t3 = g2d[4 * (size_t)g + 3];
            if (test_keep && !tile_has_record(__float_as_uint(t2.z), __float_as_uint(t2.w),
                                              (u64)__float_as_uint(t3.x) | ((u64)__float_as_uint(t3.w) << 32), t.tx, t.ty)) continue;   // no record
            const u32 io = inst_off ? inst_off[g] : block_base[__float_as_uint(t3.y)] + __float_as_uint(t3.z);
            const u32 e = emission_slot(io, make_uint2(__float_as_uint(t2.z), __float_as_uint(t2.w)),
                                        (u64)__float_as_uint(t3.x) | ((u64)__float_as_uint(t3.w) << 32), t.tx, t.ty);
            float4* dst = reinterpret_cast<float4*>(partials + (size_t)e * PART_FLOATS);
            const float4 z4 = make_float4(0.f, 0.f, 0.f, 0.f);
            dst[0] = z4; dst[1] = z4; dst[2] = z4;
        }
    };

    // The tile's descriptor carries its deepest contributor (from the forward), so the ids of the first two chunks are
    // requested before anything else and the pixel loads of tile_begin overlap them: the per-tile dependent chain is
    // descriptor -> ids -> gathers.
    u32 gid0p = 0u, gid1p = 0u;                              // (COMPACT: list positions of the slots of chunks 0, 1)
    if (COMPACT) {
        gid0p = fetch_pos(hi0, tid);
        if (hi0 > CH) gid1p = fetch_pos(hi0 - CH, tid);
    }
    const uint2 gid0r = fetch_id(A.rx, hi0, gid0p, tid);
    uint2 gid1 = (hi0 > CH) ? fetch_id(A.rx, hi0 - CH, gid1p, tid) : make_uint2(0xFFFFFFFFu, 0u);      // ids of chunk 1, in flight with chunk 0's
    tile_begin(A);
    if (hi0 == 0) return;                                    // nothing contributed anywhere in the tile: all records are zero
    if (tid < SL) {
        const float4 z4 = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
        for (int w = 0; w < 4; ++w) {
            float4* a4 = reinterpret_cast<float4*>(&acc[w][tid][0]);
            a4[0] = z4; a4[1] = z4; a4[2] = z4;
        }
    }
    PH_MARK(0);        // prologue
    // Chunk pipeline.  Every value carried around the loop is COMPLETE (the compiler copies loop-carried registers at
    // the loop head, and copying the destination of an in-flight load stalls on it -- measured: 23 % of the wave time):
    //   top      publish rec(k) to LDS, barrier
    //   then     issue the gathers of chunk k+1 (ids known) and the id load of chunk k+2
    //   ...      lists + groups of chunk k            <- the loads land underneath
    //   barrier  consume gathers -> rec(k+1), ids(k+2); only then store the records of chunk k
    ChunkRec rec;
    {
        const uint2 gid0 = decode_id(gid0r, 1u);
        const Raw raw0 = fetch(gid0.x);
        rec = make_rec(raw0, slot_of(raw0, gid0.x != 0xFFFFFFFFu, A), gid0.y, A, hi0 - min(hi0, (u32)CH), min(hi0, (u32)CH), gid0p, tid,
                       -0.5f * LOG2E, -LOG2E);
    }
    asm volatile("" :: "v"(gid1.x), "v"(gid1.y), "v"(gid1p));    // complete before the loop: a pending load on the entry edge costs a vmcnt(0) at every loop top
    gid1 = decode_id(gid1, 1u);

    const int row = lane >> 4, li = lane & 15;
    const int qx = (wave & 1) * 2, qy = (wave >> 1) * 2;
    const int myblk = (qy + (row >> 1)) * 4 + qx + (row & 1);
    float4* const pixb = &pixq[myblk][0];
    DG(u32 dg_eval = 0, dg_con = 0, dg_ent = 0, dg_ent0 = 0, dg_steps = 0, dg_chunks = 0;)

    for (u32 hi = hi0;;) {
        const u32 cnt = min(hi, (u32)CH);
        const u32 lo = hi - cnt;
        // ---- publish the staged chunk [lo, hi) of tile A: slot s <-> list position lo + s (front to back)
        // (the slot index formed here, see tid_now: from a loop-invariant index the compiler also hoists the six LDS addresses of this section
        // out of the chunk loop, spills them around the row loop and reloads each behind an s_waitcnt vmcnt(0))
        const int ptid = tid_now();
        if (ptid < CH) {                                      // safe without a barrier: after the previous chunk's second barrier nobody reads them
            recA[ptid] = make_float4(rec.x, rec.y, rec.ap, rec.bp); recB[ptid] = make_float4(rec.cp, rec.o, rec.r, rec.g); recC[ptid] = rec.b;
            masks[ptid] = rec.mask;
            if (COMPACT) chunk_pos[ptid] = rec.pos - 1u;
        }
        // ranks of this slot among its segment's slots that reach quadrant 0..3 (an empty slot has mask 0), and the segment's totals
        u32 myidx;
        {
            const u64 b0 = __ballot((rec.mask & 0x0033u) != 0u), b1 = __ballot((rec.mask & 0x00CCu) != 0u);
            const u64 b2 = __ballot((rec.mask & 0x3300u) != 0u), b3 = __ballot((rec.mask & 0xCC00u) != 0u);
            myidx = ballot_rank(b0) | (ballot_rank(b1) << 8) | (ballot_rank(b2) << 16) | (ballot_rank(b3) << 24);
            if (ptid < CH) rk4[ptid] = myidx;
            // (a byte per quadrant: a segment's totals are at most 64, a chunk's at most CH <= 255, so the prefix sums below are plain adds)
            if ((ptid & 63) == 0) segtot[wave] = (u32)__popcll(b0) | ((u32)__popcll(b1) << 8) | ((u32)__popcll(b2) << 16) | ((u32)__popcll(b3) << 24);
        }
        lds_barrier();
        PH_MARK(2);    // barrier 1
        __builtin_amdgcn_s_setprio(PRIO_GROUPS);
        const u32 nx_cnt = min(lo, (u32)CH), nx_lo = lo - nx_cnt;                      // chunk k+1 = [nx_lo, lo)
        // ids of chunk k+2 (as fetched: decode_id), then the gathers of chunk k+1.  COMPACT: the position comes first and the id load waits
        // for it -- in front of the gathers, so that the wait does not drain them (fetching the position a chunk earlier still, in a
        // third stage, measured 1.8 % slower at sm 1.0 and no faster at sm 0.5: profiles/r05/ab_blend_bwd.txt section 12)
        u32 gid2p = (COMPACT && nx_lo > 0) ? fetch_pos(nx_lo, ptid) : 0u;
        const uint2 gid2 = (nx_lo > 0) ? fetch_id(A.rx, nx_lo, gid2p, ptid) : make_uint2(0xFFFFFFFFu, 0u);
        Raw raw_n = fetch(lo > 0 ? gid1.x : 0xFFFFFFFFu);

        // The per-(splat, block) work: one block row (4 pixels = two packed pairs) per step; four independent scan chains
        // interleave without pipeline bubbles.  `s` is the lane's splat, (bx0, by0) its block origin, pixb its block's
        // pixel pairs, `carry` marks the lane that holds the scan totals (the shallowest of its segment).
        f2 a0, a1, a2, a6, a9, a10;
        float sa3, sa4, sa5, sa7, sa8;
        auto block_rows = [&](auto skip_nc_tag, auto skip_p_tag, const StagedSplat s, const bool live, const float bx0, const float by0, float4* pixb, const bool carry) {
            constexpr bool SKIP_NC = decltype(skip_nc_tag)::value;
            constexpr bool SKIP_P = decltype(skip_p_tag)::value;
            float pyf = by0;                                // the row's pixel y: integers, so the += 1 below is exact and dy is
                                                            // the forward's s.y - (float)py bit for bit
            a0 = (f2){0.f, 0.f}; a1 = a0; a2 = a0; a6 = a0; a9 = a0; a10 = a0;
            sa3 = sa4 = sa5 = sa7 = sa8 = 0.f;
            // (Round 5, measured and not kept, profiles/r05/ab_blend_bwd.txt: peeling the first pixel row so that it STARTS the sums
            // instead of adding to 17 zeroed registers, the sums folded into register pairs once per step instead of inside every
            // read-modify-write phase, 24-bit instead of 32- / 64-bit multiplies for the slot addresses -- ~30 fewer vector
            // instructions per 16-splat step, 0.3462 -> 0.3474 / 0.3480 ms: plain instructions are not what this kernel waits for.)
#pragma unroll 1
            for (int iy = 0; iy < 4; ++iy) {
                float4* P0 = pixb + iy * 2 * PQ;
                float4* P1 = P0 + PQ;
                const float4 q00 = P0[0], q01 = P0[1], q02 = P0[2];       // q*1.zw: carry of prod (1 - alpha); q*2 = Rc Rc nc nc
                const float4 q10 = P1[0], q11 = P1[1], q12 = P1[2];
                const u32 nc0 = __float_as_uint(q02.z), nc1 = __float_as_uint(q02.w), nc2 = __float_as_uint(q12.z), nc3 = __float_as_uint(q12.w);
                const float4 q03 = make_float4(q01.z, q01.w, q02.x, q02.y), q13 = make_float4(q11.z, q11.w, q12.x, q12.y);   // (Ac Ac Rc Rc)
                const float dy = s.y - pyf;
                pyf += 1.0f;
                const float u = __fmul_rn(__fmul_rn(s.cp, dy), dy);
                const f2 dyy = {dy, dy};
                // ---- part 1: alpha of the four pixels (same arithmetic as pair_power2 on d = centre - pixel)
                const f2 dxa = {s.x - bx0, s.x - (bx0 + 1.f)}, dxb = {s.x - (bx0 + 2.f), s.x - (bx0 + 3.f)};
                const f2 apdxa = s.ap * dxa, apdxb = s.ap * dxb;
                const f2 ta = __builtin_elementwise_fma((f2){s.bp, s.bp}, dyy, apdxa);
                const f2 tb = __builtin_elementwise_fma((f2){s.bp, s.bp}, dyy, apdxb);
                const f2 pa = __builtin_elementwise_fma(dxa, ta, (f2){u, u});
                const f2 pb = __builtin_elementwise_fma(dxb, tb, (f2){u, u});
                f2 Ga = {__builtin_amdgcn_exp2f(pa.x), __builtin_amdgcn_exp2f(pa.y)};
                f2 Gb = {__builtin_amdgcn_exp2f(pb.x), __builtin_amdgcn_exp2f(pb.y)};
                // unclamped o G, zeroed for non-contributing pairs; min(0.99, x) >= 1/255 <=> x >= 1/255, so the contribute /
                // skip decision is the forward's.  The clamp is applied after the select (0 stays 0).
                f2 aua = s.o * Ga, aub = s.o * Gb;
                // bitwise &, not &&: a short-circuit makes the compiler branch around one pixel's n_contrib read and wait for
                // every outstanding LDS read (lgkmcnt(0)) in the middle of the row
                // SKIP_NC: no pixel of the tile stopped early, so a splat behind a pixel's last contributor fails the alpha test
                // at that pixel anyway (it failed it in the forward): the position test is redundant
                const bool v0 = live & (SKIP_P | (pa.x <= 0.f)) & (aua.x >= ALPHA_MIN) & (SKIP_NC | (s.pos <= nc0));
                const bool v1 = live & (SKIP_P | (pa.y <= 0.f)) & (aua.y >= ALPHA_MIN) & (SKIP_NC | (s.pos <= nc1));
                const bool v2 = live & (SKIP_P | (pb.x <= 0.f)) & (aub.x >= ALPHA_MIN) & (SKIP_NC | (s.pos <= nc2));
                const bool v3 = live & (SKIP_P | (pb.y <= 0.f)) & (aub.y >= ALPHA_MIN) & (SKIP_NC | (s.pos <= nc3));
                aua.x = v0 ? aua.x : 0.f; aua.y = v1 ? aua.y : 0.f; aub.x = v2 ? aub.x : 0.f; aub.y = v3 ? aub.y : 0.f;
                DG(dg_eval += live ? 4u : 0u; dg_con += (u32)v0 + (u32)v1 + (u32)v2 + (u32)v3;)
                const f2 ala = {fminf(0.99f, aua.x), fminf(0.99f, aua.y)}, alb = {fminf(0.99f, aub.x), fminf(0.99f, aub.y)};
                const f2 oma = 1.f - ala, omb = 1.f - alb;
                const f2 g0a = {q00.x, q00.y}, g1a = {q00.z, q00.w}, g2a = {q01.x, q01.y};
                const f2 g0b = {q10.x, q10.y}, g1b = {q10.z, q10.w}, g2b = {q11.x, q11.y};
                const f2 sda = __builtin_elementwise_fma((f2){s.b, s.b}, g2a, __builtin_elementwise_fma((f2){s.g, s.g}, g1a, s.r * g0a));
                const f2 sdb = __builtin_elementwise_fma((f2){s.b, s.b}, g2b, __builtin_elementwise_fma((f2){s.g, s.g}, g1b, s.r * g0b));
                // ---- one affine scan per pixel: F_i(x) = alpha_i (c_i . g) + (1 - alpha_i) x, lane 0 (deepest) innermost
                const f2 ofa = ala * sda, ofb = alb * sdb;
                f2 A01 = oma, A23 = omb, O01 = ofa, O23 = ofb;
                scan_affine16x4_pinned(A01, A23, O01, O23);
                const float A0 = A01.x, A1 = A01.y, A2 = A23.x, A3 = A23.y, o0 = O01.x, o1 = O01.y, o2 = O23.x, o3 = O23.y;
                // carries in q*3: .xy = prod (1 - alpha) behind the group, .zw = colour . g seen behind the group
                const f2 Ba = (f2){A0, A1} * (f2){q03.x, q03.y}, Bb = (f2){A2, A3} * (f2){q13.x, q13.y};
                const f2 Tna = {__builtin_amdgcn_rcpf(Ba.x), __builtin_amdgcn_rcpf(Ba.y)};                   // T in front of i
                const f2 Tnb = {__builtin_amdgcn_rcpf(Bb.x), __builtin_amdgcn_rcpf(Bb.y)};
                const f2 wa = ala * Tna, wb = alb * Tnb;
                // what is seen at the FRONT face of i; the next deeper lane's value is what lies BEHIND i
                const f2 Va = __builtin_elementwise_fma((f2){A0, A1}, (f2){q03.z, q03.w}, (f2){o0, o1});
                const f2 Vb = __builtin_elementwise_fma((f2){A2, A3}, (f2){q13.z, q13.w}, (f2){o2, o3});
                float R0 = q03.z, R1 = q03.w, R2 = q13.z, R3 = q13.w;
                shift_up16x4(R0, R1, R2, R3, Va.x, Va.y, Vb.x, Vb.y);     // (pinned like the scan: same instruction count, same time)
                if (carry) {                                            // carries for the next (shallower) group
                    reinterpret_cast<float2*>(P0 + 1)[1] = make_float2(Ba.x, Ba.y); reinterpret_cast<float2*>(P0 + 2)[0] = make_float2(Va.x, Va.y);
                    reinterpret_cast<float2*>(P1 + 1)[1] = make_float2(Bb.x, Bb.y); reinterpret_cast<float2*>(P1 + 2)[0] = make_float2(Vb.x, Vb.y);
                }
                const f2 dLa = Tna * (sda - (f2){R0, R1});
                const f2 dLb = Tnb * (sdb - (f2){R2, R3});
                a0 = __builtin_elementwise_fma(wa, g0a, a0); a1 = __builtin_elementwise_fma(wa, g1a, a1); a2 = __builtin_elementwise_fma(wa, g2a, a2);
                a0 = __builtin_elementwise_fma(wb, g0b, a0); a1 = __builtin_elementwise_fma(wb, g1b, a1); a2 = __builtin_elementwise_fma(wb, g2b, a2);
                // q = dL/dpower = (o G) dL/dalpha (straight through the clamp, decision D3); 0 for non-contributing pairs.
                // dy is common to the row: the moments in dy are taken on the row sums.  a3 collects sum q = o * sum G dL/dalpha;
                // the division by o happens once per record.
                const f2 qva = aua * dLa, qvb = aub * dLb;
                const f2 qdxa = qva * dxa, qdxb = qvb * dxb;
                const f2 rq = qva + qvb, rqdx = qdxa + qdxb;
                a6 = __builtin_elementwise_fma(qdxa, dxa, a6); a6 = __builtin_elementwise_fma(qdxb, dxb, a6);
                const float rqs = rq.x + rq.y, rqdxs = rqdx.x + rqdx.y;
                const float rqdys = rqs * dy;
                sa3 += rqs; sa4 += rqdxs; sa5 += rqdys;
                sa7 = __fmaf_rn(rqdxs, dy, sa7);
                sa8 = __fmaf_rn(rqdys, dy, sa8);
                if (ABS) {                                              // sum |q d power / d centre| per pixel
                    const f2 bp2 = {s.bp, s.bp};
                    const f2 hxa = ta + apdxa, hxb = tb + apdxb;                              // 2 ap dx + bp dy
                    const float cdy2 = 2.f * s.cp * dy;
                    const f2 hya = __builtin_elementwise_fma(bp2, dxa, (f2){cdy2, cdy2});       // 2 cp dy + bp dx
                    const f2 hyb = __builtin_elementwise_fma(bp2, dxb, (f2){cdy2, cdy2});
                    // acc += |q| |h| as ONE v_fma_f32 with abs source modifiers (8 per row step; packed fp32 has no modifiers:
                    // four v_pk_mul + eight v_add with |.| before)
#define ACC_ABS2(acc, q, h) asm("v_fma_f32 %0, |%1|, |%2|, %0" : "+v"(acc) : "v"(q), "v"(h))
                    ACC_ABS2(a9.x, qva.x, hxa.x); ACC_ABS2(a9.y, qva.y, hxa.y); ACC_ABS2(a9.x, qvb.x, hxb.x); ACC_ABS2(a9.y, qvb.y, hxb.y);
                    ACC_ABS2(a10.x, qva.x, hya.x); ACC_ABS2(a10.y, qva.y, hya.y); ACC_ABS2(a10.x, qvb.x, hyb.x); ACC_ABS2(a10.y, qvb.y, hyb.y);
#undef ACC_ABS2
                }
            }
        };
        // the two halves of every packed accumulator are added ONCE per step (h*), outside the four row phases below: each
        // phase is issued for the whole wave although only one 16-lane row takes part in it
        // ---- a step's eleven sums into the wave's LDS copy (round 5: "rotating quarters").  One splat sits in several rows of a wave in the
        // same step (it reaches ~2 blocks of the quadrant), so the rows cannot all read-modify-write their slots at once; until round 5
        // they took turns -- four phases of 16 lanes, 3 + 3 wide LDS instructions each, and timing builds showed the twelve exec-masked
        // stores alone at 9 % of the kernel (a ds_write_b128 holds its SIMD's operand path ~13 cycles whatever its exec mask).  Now a slot's
        // record is four QUARTERS of three sums, and in phase p a lane of row r updates quarter (r + p) mod 4 of its slot: all four rows
        // are active in every phase, two lanes that share a slot are never on the same quarter, and over the four phases every lane has
        // added all of its sums.  Per step 4 x (ds_read_b64 + ds_read_b32) and 4 x (ds_write_b64 + ds_write_b32) with full lanes instead
        // of 12 + 12 wide ones with a quarter of them.  A (slot, quarter) receives its rows' terms in phase order -- fixed, so the sums
        // stay bitwise reproducible.  LDS record (12 floats): quarter q = {s[3q], s[3q+1]} at floats 2q, 2q+1 and s[3q+2] at float 8 + q
        // (s = the eleven sums in the order of the gradient record, s[11] = 0): every piece aligned for its access, the write-out below
        // still reads three float4s per copy and puts the sums back in record order.
        float t0[4], t1[4], t2[4];                          // t_j[k] = s[3k + j]
        auto fold_sums = [&]() {
            t0[0] = a0.x + a0.y; t1[0] = a1.x + a1.y; t2[0] = a2.x + a2.y;
            t0[1] = sa3;         t1[1] = sa4;         t2[1] = sa5;
            t0[2] = a6.x + a6.y; t1[2] = sa7;         t2[2] = sa8;
            t0[3] = a9.x + a9.y; t1[3] = a10.x + a10.y; t2[3] = 0.f;
            // rotate by the lane's row: afterwards t_j[p] belongs to quarter (row + p) & 3 (two stages of selects per component)
            const bool b0 = (row & 1) != 0, b1 = (row & 2) != 0;
#define ROT4(t) { const float x0 = b0 ? t[1] : t[0], x1 = b0 ? t[2] : t[1], x2 = b0 ? t[3] : t[2], x3 = b0 ? t[0] : t[3];            \
                  t[0] = b1 ? x2 : x0; t[1] = b1 ? x3 : x1; t[2] = b1 ? x0 : x2; t[3] = b1 ? x1 : x3; }
            ROT4(t0) ROT4(t1) ROT4(t2)
#undef ROT4
        };
        char* const acc_wave = reinterpret_cast<char*>(&acc[wave][0][0]);
        auto add_to_copy = [&](const int slot) {      // (every live lane calls it: no row takes turns any more)
            char* const base = acc_wave + __umul24((u32)slot, 48u);
#pragma unroll
            for (int p = 0; p < 4; ++p) {
                const u32 q = (u32)(row + p) & 3u;
                float2* const pq = reinterpret_cast<float2*>(base + 8u * q);
                float* const ps = reinterpret_cast<float*>(base + 32u + 4u * q);
                float2 v = *pq; float w = *ps;
                v.x += t0[p]; v.y += t1[p]; w += t2[p];
                *pq = v; *ps = w;
                // another LANE's next phase may read what this phase wrote (same slot in another row): the hardware keeps a wave's LDS
                // accesses in program order, the compiler must too (to it the two addresses belong to different quarters of one record)
                asm volatile("" ::: "memory");
            }
        };
        // ---- where this chunk's slots sit in the wave copies.  Normally a slot's index in copy w is its rank among ALL the chunk's slots that
        // reach quadrant w; if some quadrant is reached by more than SL of them, the group phase runs twice -- first for
        // the deeper half of the slots (segments 2, 3), then for the other -- and the ranks count from the start of a slot's own half (at
        // most CH / 2 <= SL).  Everything here is uniform over the workgroup.
        const uint4 stq = *reinterpret_cast<const uint4*>(&segtot[0]);
        const u32 g0 = (u32)__builtin_amdgcn_readfirstlane((int)stq.x), g1 = (u32)__builtin_amdgcn_readfirstlane((int)stq.y);
        const u32 g2 = (u32)__builtin_amdgcn_readfirstlane((int)stq.z), g3 = (u32)__builtin_amdgcn_readfirstlane((int)stq.w);
        const u32 p2 = g0 + g1, tot = p2 + g2 + g3;        // (scalars, four byte lanes each)
        const bool split = max(max(tot & 0xFFu, (tot >> 8) & 0xFFu), max((tot >> 16) & 0xFFu, tot >> 24)) > (u32)SL;
        // index base of each segment, all four quadrants at once: b1 = g0, b2 = g0 + g1 (0 in a split chunk), b3 = b2 + g2
        const u32 pb2 = split ? 0u : p2, pb3 = pb2 + g2;
        myidx += wave == 0 ? 0u : wave == 1 ? g0 : wave == 2 ? pb2 : pb3;       // the staging thread's own four indices
        // ... and, for the row loop of THIS wave's quadrant, the bases as plain numbers
        const u32 cb1 = (g0 >> (8 * wave)) & 0xFFu, cb2 = (pb2 >> (8 * wave)) & 0xFFu, cb3 = (pb3 >> (8 * wave)) & 0xFFu;
        const int rk_shift = 8 * wave;
        u32 mreg[NSEG];
        int llane = 0;                                   // (lane id, formed anew per round: see tid_now)
        // ballot-compact the slots of segments [r_lo, r_hi) that reach block `blk` (list order = depth order); returns the list length
        auto build_list = [&](const int blk, const int r_lo, const int r_hi) -> int {
            int L = 0;
            const u32 bmax = blk_maxc[blk];              // splats behind every pixel's last contributor cannot matter here
#pragma unroll
            for (int r = 0; r < NSEG; ++r) {
                if (r < r_lo || r >= r_hi) continue;    // (uniform: the other half of a split chunk, or segments past the chunk's last slot)
                const int slot = r * 64 + llane;
                // pos = lo + slot + 1, or the compacted entry's own position
                const u32 posv = COMPACT ? (slot < CH ? chunk_pos[slot] : 0xFFFFFFFFu) : lo + (u32)slot;
                const bool hit = ((mreg[r] >> blk) & 1u) && (posv < bmax);
                const u64 bal = __ballot(hit);
                if (hit) lists[blk][L + ballot_rank(bal)] = (unsigned char)slot;
                L += __popcll(bal);
            }
            return L;
        };
        const float bx0 = (float)(A.tx * BAGS_TILE) + 4.f * (float)(myblk & 3), by0 = (float)(A.ty * BAGS_TILE) + 4.f * (float)(myblk >> 2);
        u32 e_n = 0xFFFFFFFFu;
        const u32 cur_e = rec.e, cur_mask = rec.mask;    // (this chunk's: `rec` becomes the next chunk's record inside the rounds)
        const int nround = split ? 2 : 1;
#pragma unroll 1
        for (int rd = 0; rd < nround; ++rd) {
        const int r_lo = (split && rd == 0) ? 2 : 0, r_hi = (split && rd == 1) ? 2 : min(NSEG, (int)((cnt + 63u) >> 6));
        // ---- the wave owns quadrant `wave`; DPP row r of the wave owns one 4x4 block of it and walks that block's
        // list 16 splats per step (deepest in the row's lane 0).  Against "64 lanes = 64 splats of one block" (in the git
        // history) this quantises the lists at 16 instead of 64 entries and shortens the scans from six DPP steps to four.
        llane = tid_now() & 63;
#pragma unroll
        for (int r = 0; r < NSEG; ++r) mreg[r] = (r * 64 + llane < CH) ? masks[r * 64 + llane] : 0u;
        int Lr[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) Lr[j] = build_list((qy + (j >> 1)) * 4 + qx + (j & 1), r_lo, r_hi);
        PH_MARK(3);    // list building
        // the lines of chunk k+1 have landed by now (a list-building phase is several memory latencies long): their emission
        // slots are formed here and used after the second barrier
        if (rd == 0) {
            // (everything the slot is formed from becomes visible to the compiler HERE: otherwise it schedules the rectangle arithmetic right
            // behind the gathers -- a wait for them in front of the list building -- and spills the table address across it)
            asm volatile("" : "+v"(raw_n.blk), "+v"(raw_n.io), "+v"(raw_n.q2.z), "+v"(raw_n.q2.w), "+v"(raw_n.kpl), "+v"(raw_n.kph));
            e_n = slot_of(raw_n, lo > 0 && gid1.x != 0xFFFFFFFFu, A);
        }
        const int myL = (row == 0) ? Lr[0] : (row == 1) ? Lr[1] : (row == 2) ? Lr[2] : Lr[3];
        const int nIter = (max(max(Lr[0], Lr[1]), max(Lr[2], Lr[3])) + 15) >> 4;
        int slot_next = (li < myL) ? (int)lists[myblk][myL - 1 - li] : 0;
        for (int it = 0; it < nIter; ++it) {
            const int gend = myL - 16 * it;                 // <= 0: this row's list is exhausted
            const bool live = li < gend;
            const int slot = slot_next;
            const float4 ra = recA[slot], rb = recB[slot];
            const StagedSplat s = {ra.x, ra.y, ra.z, ra.w, rb.x, rb.y, rb.z, rb.w, recC[slot], (COMPACT ? chunk_pos[slot] : lo + (u32)slot) + 1u};
            const u32 rkw = rk4[slot];                      // (consumed at the end of the step)
            // the next step's list entry is read now: list byte -> record is a chain of two LDS latencies otherwise
            slot_next = (li < gend - 16) ? (int)lists[myblk][gend - 17 - li] : 0;
            DG(const u32 dg_before = dg_con;)
            // three instances of the row loop (wave-uniform choice per tile): the fast one for tiles where no pixel stopped early
            // and every staged conic is well conditioned; without the position test only; with both tests
            const bool cry = (li == 15) && (gend > 0);
            if (!A.early && !A.needle) block_rows(std::true_type{}, std::true_type{}, s, live, bx0, by0, pixb, cry);
            else if (!A.needle) block_rows(std::false_type{}, std::true_type{}, s, live, bx0, by0, pixb, cry);
            else if (!A.early) block_rows(std::true_type{}, std::false_type{}, s, live, bx0, by0, pixb, cry);
            else block_rows(std::false_type{}, std::false_type{}, s, live, bx0, by0, pixb, cry);
            DG(dg_ent += live ? 1u : 0u; dg_ent0 += (live && dg_con == dg_before) ? 1u : 0u; dg_steps += (lane == 0) ? 1u : 0u;)
            // the same splat can sit in several rows (it reaches several blocks of the quadrant): rotating quarters
            fold_sums();
            const int sg = slot >> 6;
            const u32 cidx = ((rkw >> rk_shift) & 0xFFu) + (sg == 0 ? 0u : sg == 1 ? cb1 : sg == 2 ? cb2 : cb3);
            if (live) add_to_copy((int)cidx);
            PH_MARK(4);    // groups
        }
        PH_MARK(3);
        DG(dg_chunks += (lane == 0) ? 1u : 0u;)
        lds_barrier();
        PH_MARK(5);    // barrier 2
        // The short serial section between the two barriers shares its SIMDs with another workgroup that is usually in
        // its VALU-saturated group phase; without priority the four waves crawl through it at different speeds and
        // the skew is paid at the next barrier.
        __builtin_amdgcn_s_setprio(PRIO_SERIAL);
        const int wtid = tid_now();
        // ---- next chunk: its gathers were issued before the groups; turn them into the staged record.  Every loaded register is redefined
        // an input of this asm, also the components nobody reads (q2.y, the view depth):
        //   * the register allocator recycles a dead component of an in-flight load as scratch, which needs s_waitcnt vmcnt(0) right after
        //     the loads were issued and exposes the whole gather latency once per chunk (seen in the ISA; 20 % of the wave time);
        //   * arithmetic on a loaded value (make_rec's scaling of the conic, decode_id) is otherwise scheduled right behind the load: its
        //     constants come out of this asm;
        //   * the loads are retired before this chunk's record stores are issued, so the top of the next chunk does not have to drain
        //     the stores to be sure they have arrived.
        // In EVERY round, not only the last (the result is the same): with a path around it, the loads count as pending on the loop's
        // back edge and the next chunk's first write to one of their registers waits for everything, the record stores included.
        float c_half = -0.5f * LOG2E, c_one = -LOG2E;
        asm volatile("" : "+v"(c_half), "+v"(c_one)
                        : "v"(gid2.x), "v"(gid2.y), "v"(raw_n.q0.x), "v"(raw_n.q0.y), "v"(raw_n.q0.z), "v"(raw_n.q0.w), "v"(raw_n.q1.x), "v"(raw_n.q1.y),
                          "v"(raw_n.q1.z), "v"(raw_n.q1.w), "v"(raw_n.q2.x), "v"(raw_n.q2.y), "v"(e_n), "v"(gid2p));
        if (lo > 0) rec = make_rec(raw_n, e_n, gid1.y, A, nx_lo, nx_cnt, gid1p, wtid, c_half, c_one);
        PH_MARK(1);
        // ---- one record per staged instance of the round's segments: the wave copies it sits in added in fixed order.  The record holds
        // the raw sums (sum q rather than sum q / o, the abs sums on the scaled conic): preprocess_bwd applies the per-Gaussian factors once.
        if ((u32)wtid < cnt && wave >= r_lo && wave < r_hi) {
            float4 r0 = make_float4(0.f, 0.f, 0.f, 0.f), r1 = r0, r2 = r0;
            const float4 z4 = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
            for (int w = 0; w < 4; ++w) {
                if (cur_mask & (w == 0 ? 0x0033u : w == 1 ? 0x00CCu : w == 2 ? 0x3300u : 0xCC00u)) {
                    float4* a4 = reinterpret_cast<float4*>(&acc[w][(myidx >> (8 * w)) & 0xFFu][0]);
                    const float4 x0 = a4[0], x1 = a4[1], x2 = a4[2];
                    a4[0] = z4; a4[1] = z4; a4[2] = z4;               // owner re-zeroes its slot for the next chunk
                    // LDS layout (s0 s1 s3 s4 | s6 s7 s9 s10 | s2 s5 s8 -) -> record order (s0 s1 s2 s3 | s4 s5 s6 s7 | s8 s9 s10 -)
                    r0.x += x0.x; r0.y += x0.y; r0.z += x2.x; r0.w += x0.z;
                    r1.x += x0.w; r1.y += x2.y; r1.z += x1.x; r1.w += x1.y;
                    r2.x += x2.z; r2.y += x1.z; r2.z += x1.w;
                }
            }
            if (cur_e != 0xFFFFFFFFu) {
                float4* dst = reinterpret_cast<float4*>(partials + (size_t)cur_e * PART_FLOATS);
                dst[0] = r0; dst[1] = r1; dst[2] = r2;
                if (live_map) live_map[cur_e] = 1;
            }
        }
        PH_MARK(6);    // record sums + stores
        // between two rounds of one chunk: the second round's adds must not meet the first round's write-out (the slots it re-zeroes).
        // After the last round no barrier: the next chunk's first barrier orders these LDS accesses before any reuse
        if (rd + 1 < nround) { lds_barrier(); __builtin_amdgcn_s_setprio(PRIO_GROUPS); }
        }
        {
            u32 one = 1u;
            asm volatile("" : "+v"(one) : "v"(gid2.x), "v"(gid2.y));
            gid1 = decode_id(gid2, one); gid1p = gid2p;
        }
        if (lo == 0) break;
        hi = lo;
    }
    PH(if (lane == 0) for (int i = 0; i < 8; ++i) atomicAdd(&g_phase_cycles[i], ph[i]);)
    DG(diag_pairs_flush(0, dg_eval, dg_con);)
    DG(diag_pairs_flush(4, dg_ent, dg_ent0);)
    DG(diag_pairs_flush(6, dg_steps, dg_chunks);)
}
PH(extern "C" void bags_diag_phases(unsigned long long* out) { hipMemcpyFromSymbol(out, HIP_SYMBOL(g_phase_cycles), sizeof(unsigned long long) * 8); })

// BagsBackwardArgs.dense_per_tile: 0 = the default above, < 0 = never, > 0 = that threshold (the robustness tests force both paths)
bool bwd_dense_mode(long long n_records, int T, int dense_per_tile_arg)
{
    const long long thr = dense_per_tile_arg == 0 ? BWD_DENSE_PER_TILE : dense_per_tile_arg;
    return thr > 0 && n_records > thr * (long long)T;
}
// live_map: the caller's ONE decision about the dense-scene mode (bags_backward: bwd_dense_mode) -- the byte map, or null.  The same
// pointer goes to launch_preprocess_bwd, so the kernel that marks records and the kernel that reads the marks cannot disagree.
hipError_t launch_blend_bwd(const BagsSettings& s, const GeomView& g, const BinView& b, const ImgView& im,
                            const float* grad_color, float* partials, bool want_abs, bool binned, hipStream_t st,
                            long long n_records, unsigned char* live_map, hipEvent_t ev_start, hipEvent_t ev_stop)
{
    const int gx = cdiv(s.image_width, BAGS_TILE), gy = cdiv(s.image_height, BAGS_TILE);
    const int T = gx * gy;
    if (T == 0) return hipSuccess;
    const int grid = cdiv(T, TILE_RUN) * TILE_RUN;
    const bool compact = binned && s.tile_bounds != BAGS_TILES_OPACITY;
    // Dense scenes (long tile lists, most of each list behind the deepest contributor): clearing the record array with one
    // streaming memset is cheaper than the per-tile zero loops, which gather an id and two geometry lines per dead instance.
    // (the map is carved 256-byte aligned, in 256-byte units, with room for the 64 bytes preprocess_bwd reads from a mark on --
    // bags_backward_workspace_size: a fill of whole units is ONE launch of the runtime's fill kernel; with the odd tail it was two, ~5.5 us
    // each -- and the bytes behind the last mark are zero rather than arbitrary)
    if (live_map) { hipError_t e = hipMemsetAsync(live_map, 0, ((size_t)n_records + 64 + 255) / 256 * 256, st); if (e != hipSuccess) return e; }
    // (ev_start / ev_stop: the stage profiler's events ride on this dispatch -- bags_backward says why)
#define BWD_ARGS_ s.image_width, s.image_height, gx, T, im.tile_desc, (const u32*)b.point_list,                                          \
                  reinterpret_cast<const unsigned char*>(b.reach_mask), (u32)(binned ? 8u : 4u), (const float4*)g.g2d,                  \
                  (const u32*)(binned ? nullptr : g.inst_off), (const u32*)g.block_base, (const float*)s.bg, (const float*)im.final_T,  \
                  (const u32*)im.n_contrib, grad_color, partials, (int)(compact ? 1 : 0), (const uint4*)im.tile_aux, live_map
#define BWD_LAUNCH_(ABS_, CMP_, SPARSE_)                                                                                             \
    do { if (ev_start || ev_stop) hipExtLaunchKernelGGL((blend_bwd_scan_kernel<ABS_, CMP_, SPARSE_>), dim3(grid), dim3(256), 0, st,     \
                                                        ev_start, ev_stop, 0, BWD_ARGS_);                                             \
         else LAUNCH_K((blend_bwd_scan_kernel<ABS_, CMP_, SPARSE_>), dim3(grid), dim3(256), 0, st, BWD_ARGS_); } while (0)
    // The chunk geometry follows the SCENE, not the dense-scene decision: forcing the dense-scene mode on or off leaves the arithmetic
    // untouched, so the two modes stay bit-identical (tests, tools/fuzz_paths.py --cross-dense)
    // (stock tile rule on the tile-binned path: the chunks are staged from the compacted list of record holders -- ~60 % of the list's
    // instances on the bench scene, the instances the opacity rule keeps -- so it is THEIR density that picks the geometry)
#ifndef COMPACT_DENSITY_NUM
#define COMPACT_DENSITY_NUM 3
#endif
    const long long per_tile_x5 = compact ? n_records * COMPACT_DENSITY_NUM : n_records * 5;
    const bool sparse = per_tile_x5 <= 5ll * BWD_SPARSE_PER_TILE * T;
#define BWD_LAUNCH(ABS_, CMP_) do { if (sparse) BWD_LAUNCH_(ABS_, CMP_, true); else BWD_LAUNCH_(ABS_, CMP_, false); } while (0)
    if (want_abs) { if (compact) BWD_LAUNCH(true, true); else BWD_LAUNCH(true, false); }
    else          { if (compact) BWD_LAUNCH(false, true); else BWD_LAUNCH(false, false); }
#undef BWD_LAUNCH
#undef BWD_LAUNCH_
#undef BWD_ARGS_
    return hipGetLastError();
}

// ================================================================================================================
// Forward, "one DPP row per 4x4 block".  (The first forward -- one wave per tile, 4 pixels per lane, 8x8 quadrant
// culling -- was latency-bound: ~3.7 busy waves per SIMD on a scene whose splats cluster in half of the tiles, each
// serialising on LDS-read -> exp -> compare -> scalar-branch chains.)  Here a 256-thread workgroup owns the tile:
//   * the list is staged 256 splats at a time with the same 16-bit reach mask of 4x4 blocks as the backward;
//   * wave w owns quadrant w; each of its four 16-lane rows owns one 4x4 block and walks ITS OWN ballot-compacted list
//     of the chunk (one splat per row per step, 16 pixels each): ~1.8x fewer (pixel, splat) evaluations than 8x8
//     quadrants, four times as many waves to hide latency, no scalar branch inside the walk.
// Compositing arithmetic per (pixel, splat) pair is pair_power2 / exp2 / the reference thresholds, shared with backward.
// Tried and dropped: two vertically adjacent pixels per lane (128-thread workgroups, 8 lanes per block, power / alpha /
// compositing on v_pk_*_f32): 16 % fewer vector and 23 % fewer scalar instructions per (pixel, splat) pair, but 105 VGPRs
// hold it at 4-5 waves per SIMD against 8 here and a wave waits on the longest of 8 block lists instead of 4: 0.169-0.171 ms
// against 0.167 ms.
// ================================================================================================================
#define FWD_STAGE (CHUNK - 1)  // splats staged per chunk: slot CHUNK-1 always holds an all-zero record and pads every list (no `i < Lrow` test per step)
template <int DUMMY>
__global__ void __launch_bounds__(256, FWD_WG_PER_CU)
blend_fwd_rows_kernel(int W, int H, int grid_x, int T, uint4* __restrict__ tile_desc, u32* __restrict__ point_list,
                      const u64* __restrict__ words_in, const u32* __restrict__ depth_key, u64* __restrict__ sort_scratch,
                      unsigned char* __restrict__ reach_mask, const u32 rm_stride, const float4* __restrict__ g2d, const float* __restrict__ bg, float* __restrict__ out_color,
                      float* __restrict__ out_depth, float* __restrict__ out_weights, float* __restrict__ final_T,
                      u32* __restrict__ n_contrib, const u32* __restrict__ n_dev, u32 capacity, const int test_keep,
                      uint4* __restrict__ tile_aux)
{
    const int dslot = slot_of_vblock(blockIdx.x);            // heavy tiles first, balanced over the XCDs
    if (dslot >= T) return;
    uint4 desc = tile_desc[dslot];                           // {tile, first instance, instances, -}
    // speculative forward on the tile-binned path: the ranges are known before the lists exist; a list that did not fit
    // its buffer was never written, so the tile is rendered empty (the caller reruns the phase on an exact buffer)
    if (n_dev && *n_dev > capacity) desc.z = 0u;
    const int tile = (int)desc.x;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, row = lane >> 4, li = lane & 15;
    const int tile_x = tile % grid_x, tile_y = tile / grid_x;
    const float X0 = (float)(tile_x * BAGS_TILE), Y0 = (float)(tile_y * BAGS_TILE);
    const uint2 range = make_uint2(desc.y, desc.y + desc.z);
    const u32 n = desc.z;

    typedef unsigned char list_t;
#define LIST_ENTRY(slot) ((list_t)(slot))
    // One raw LDS block, used twice: first by the sort of the tile's list (u64 t_all[2048] | u32 cnt_all[1024] | small state:
    // tile_sort.h), then by the blend (recs x y ap cp | bp o r g | b z pos mask, [16][CHUNK] list bytes + 16 bytes of padding for
    // the walk's read-ahead, compact masks).  20.6 KB: six workgroups per CU as before.
    constexpr int LDS_RECS = 0, LDS_LISTS = CHUNK * (int)sizeof(SplatRec), LDS_MASKS = LDS_LISTS + 16 * CHUNK * (int)sizeof(list_t) + 16;
    constexpr int LDS_BLEND = LDS_MASKS + CHUNK * 4;
    constexpr int LDS_TS_CNT = TS_LDS_WORDS * 8, LDS_TS_L = LDS_TS_CNT + TS_LDS_WORDS * 2, LDS_SORT = LDS_TS_L + (int)sizeof(TileSortLds);
    constexpr int LDS_IDS = (LDS_SORT > LDS_BLEND ? LDS_SORT : LDS_BLEND);                 // u32 ids[TSORT_WAVE]
    constexpr int LDS_BYTES = LDS_IDS;
    static_assert(LDS_MASKS % 16 == 0 && LDS_TS_L % 8 == 0 && LDS_IDS % 4 == 0, "alignment of the carved arrays");
    __shared__ __attribute__((aligned(16))) unsigned char lds_raw[LDS_BYTES];
    SplatRec* const recs = reinterpret_cast<SplatRec*>(lds_raw + LDS_RECS);
    list_t* const lists = reinterpret_cast<list_t*>(lds_raw + LDS_LISTS);
    u32* const masks = reinterpret_cast<u32*>(lds_raw + LDS_MASKS);
    __shared__ int s_live[4];
    __shared__ int s_ill[4];                                 // this chunk: did wave w stage a splat with an ill-conditioned conic
    __shared__ u32 s_cnt[4];                                 // stock tile rule: record-holding instances staged by each wave (this chunk)
    // Stock tile rule on the tile-binned path (test_keep): 40 % of the list positions hold instances without a gradient record.
    // The backward stages its chunks from a COMPACTED list of the record-holding positions, written here as they are staged
    // (u32 positions in the upper half of the tile's own slice of the words, behind the n reach words), so that its
    // chunks hold live splats only (62 k instead of 93 k wave-chunks on config 3 with 176-slot chunks; 50 k with 224 / 240).  tile_aux[slot] = {compact entries in front of
    // the deepest contributor, compact entries, list positions staged, -}.
    u32 n_live_run = 0, n_staged = 0;
    u32* const cpos = reinterpret_cast<u32*>(reach_mask + (size_t)range.x * 8u) + n;
    if (words_in != nullptr && n > 0) {
        // ---- the tile's list: (depth key, id) words grouped by the emission, unsorted -> ids in depth order in point_list
        u64* const t_all = reinterpret_cast<u64*>(lds_raw);
        u32* const cnt_all = reinterpret_cast<u32*>(lds_raw + LDS_TS_CNT);
        // the emission left the tile's unsorted ids in the first half of its slice; the words are formed on load (WordSrc)
        const WordSrc src = tile_words(words_in, desc.y, depth_key);
        if (n == 1) { if (tid == 0) point_list[desc.y] = src.ids[desc.y]; }
        else if (n <= TSORT_WAVE) { if (wave == 0) sort_wave_role(desc, src, point_list, t_all, cnt_all); }
        // (Round 5 ran the second level of the long-list sort LAZILY -- a slab sorted only when the walk was about to stage it, four
        // at a time, the unsorted rest of a dense tile's list copied out as it lay -- and dropped it: the pixel state is live across
        // that sort, the kernel went to 84 bytes of scratch per lane at its 80 VGPRs, and blend_fwd got 14 % slower on EVERY scene
        // (0.1505 -> 0.1715 ms on the headline) for ~0.05 ms saved at 1800+ entries per tile: profiles/r05/ab_dense.txt.)
        else sort_list_block(desc, src, sort_scratch, point_list, t_all, cnt_all, *reinterpret_cast<TileSortLds*>(lds_raw + LDS_TS_L));
        // the ids were stored by this workgroup and are loaded by it below (at agent scope: a line of the neighbouring tile's
        // slice may sit in this CU's L1 with our first ids still unsorted in it)
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
    }

    const int bx = (wave & 1) * 2 + (row & 1), by = (wave >> 1) * 2 + (row >> 1), blk = by * 4 + bx;
    const int px = tile_x * BAGS_TILE + bx * 4 + (li & 3), py = tile_y * BAGS_TILE + by * 4 + (li >> 2);
    float pxf = (float)px, pyf = (float)py;
    asm volatile("" : "+v"(pxf), "+v"(pyf));                 // opaque: otherwise the walk re-converts them from px, py every step
    const bool inside = (px < W) && (py < H);
    // A finished pixel (T would fall below 1e-4, or outside the image) carries its transmittance NEGATED: one VGPR sign
    // instead of a lane mask kept in scalar registers (four scalar instructions per step of a kernel that is as busy on
    // its scalar unit as on its vector units).
    float Tq = inside ? 1.f : -1.f, Cr = 0.f, Cg = 0.f, Cb = 0.f, Dq = 0.f;
    u32 last = 0;
    bool needle = false;                                     // wave-uniform: a thread of this wave staged a splat with an ill-conditioned conic
    const u64 lt_mask = (lane == 0) ? 0ull : (~0ull >> (64 - lane));
    DG(u32 dg_eval = 0, dg_con = 0;)
    // blocks of this wave's rows 0..3
    const int qb = (wave >> 1) * 8 + (wave & 1) * 2;         // block index of row 0; rows: +0, +1, +4, +5

    for (u32 base = 0; base < n; base += FWD_STAGE) {
        const u64 live_b = __ballot(Tq > 0.f);
        if (lane == 0) s_live[wave] = (live_b != 0ull);
        // (lds_barrier, here and below: what the waves exchange is in LDS.  __syncthreads() is a workgroup-scope fence too -- s_waitcnt
        // vmcnt(0) -- and CDNA4's vmcnt counts stores: every barrier of the chunk loop waited for the chunk's reach-word stores to be
        // acknowledged, and the two of the epilogue for the tile's pixel stores)
        lds_barrier();                                       // previous chunk consumed by every wave
        if (!(s_live[0] | s_live[1] | s_live[2] | s_live[3])) break;
        const u32 cnt = min((u32)FWD_STAGE, n - base);
        SplatRec rec; rec.mask = 0; rec.x = rec.y = rec.ap = rec.bp = rec.cp = rec.o = rec.r = rec.g = rec.b = rec.z = 0.f; rec.pos = 0;
        bool ill = false, live_rec = false;
        if ((u32)tid < cnt) {
            const u32 g = words_in ? __hip_atomic_load(&point_list[range.x + base + tid], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)
                                                 : point_list[range.x + base + tid];
            const float4* grec = g2d + 4 * (size_t)g;             // one 64-byte line per instance
            const float4 co = grec[0], g1 = grec[1], g2v = grec[2];
            const float2 c2 = make_float2(g1.x, g1.y);
            const float4 cz = make_float4(g1.z, g1.w, g2v.x, g2v.y);
            rec.x = c2.x; rec.y = c2.y;
            rec.ap = -0.5f * LOG2E * co.x; rec.bp = -LOG2E * co.y; rec.cp = -0.5f * LOG2E * co.z; rec.o = co.w;
            rec.r = cz.x; rec.g = cz.y; rec.b = cz.z; rec.z = cz.w;
            rec.pos = base + tid + 1;
            // Tile-binned path with the stock tile rule (test_keep): the list holds every tile of the 3-sigma square, the geometry
            // line the rectangle and tile mask of the opacity rule; an instance whose tile is not among those can reach no pixel
            // of this tile with alpha >= 1/255 (decision D7) -- it has no gradient record, and neither blend kernel looks at it.
            bool has_rec = true;
            if (test_keep) {
                const float4 q3 = grec[3];
                has_rec = tile_has_record(__float_as_uint(g2v.z), __float_as_uint(g2v.w),
                                          (u64)__float_as_uint(q3.x) | ((u64)__float_as_uint(q3.w) << 32), tile_x, tile_y);
            }
            rec.mask = has_rec ? block_mask16(c2.x, c2.y, co.x, co.y, co.z, co.w, X0, Y0) : 0u;
            // the backward stages the same instance: it reads this word (mask | has-record << 16; tile-binned path: inside the tile's own
            // slice of the words, which this workgroup has just finished with; radix path: in the dead half of the key buffers)
            reinterpret_cast<u32*>(reach_mask + (size_t)range.x * rm_stride)[base + tid] = rec.mask | (has_rec ? 0x10000u : 0u);
            ill = conic_ill_conditioned(co.x, co.y, co.z);
            live_rec = has_rec;
        }
        const bool wave_ill = __ballot(ill) != 0ull;
        needle |= wave_ill;                                  // (a scalar register, not a lane's)
        if (lane == 0) s_ill[wave] = wave_ill ? 1 : 0;
        const u64 rec_b = __ballot(live_rec);
        if (test_keep && lane == 0) s_cnt[wave] = (u32)__popcll(rec_b);
        n_staged = base + cnt;
        if (tid < CHUNK) { recs[tid] = rec; masks[tid] = rec.mask; }
        // this wave's four lists (rows qb, qb+1 | qb+4, qb+5: two runs of 2 x CHUNK bytes) start out as all-sentinel: a row
        // past the end of its own list composites the zero record of slot CHUNK-1 (opacity 0: alpha = 0 fails the 1/255 test)
        {
            static_assert(CHUNK == 256, "sentinel fill assumes 2 x 256 list bytes = 32 lanes x 16 bytes");
            const int run = (lane < 32) ? qb : qb + 4;
            const u32 f = (sizeof(list_t) == 1 ? 0x01010101u : 0x00010001u) * (u32)LIST_ENTRY(CHUNK - 1);
            uint4* dst = reinterpret_cast<uint4*>(&lists[run * CHUNK]) + (lane & 31) * (int)sizeof(list_t);
            dst[0] = make_uint4(f, f, f, f);
            if (sizeof(list_t) == 2) dst[1] = make_uint4(f, f, f, f);
        }
        lds_barrier();
        if (test_keep) {                                     // the chunk's record-holding positions, compacted in list order
            const u32 c0 = s_cnt[0], c1 = s_cnt[1], c2_ = s_cnt[2], c3 = s_cnt[3];
            const u32 before = (wave > 0 ? c0 : 0u) + (wave > 1 ? c1 : 0u) + (wave > 2 ? c2_ : 0u);
            if (live_rec) cpos[n_live_run + before + (u32)__popcll(rec_b & lt_mask)] = base + (u32)tid;
            n_live_run += c0 + c1 + c2_ + c3;
        }
        if (live_b == 0ull) continue;                        // this quadrant is finished; keep pace at the barriers
        // ---- per-row lists (rows whose 16 pixels are all done take nothing)
        int L0 = 0, L1 = 0, L2 = 0, L3 = 0;
        const bool r0 = (live_b & 0xFFFFull) != 0, r1 = (live_b & 0xFFFF0000ull) != 0,
                   r2 = (live_b & 0xFFFF00000000ull) != 0, r3 = (live_b & 0xFFFF000000000000ull) != 0;
#pragma unroll
        for (int rnd = 0; rnd < CHUNK / 64; ++rnd) {
            const int slot = rnd * 64 + lane;
            const u32 m = masks[slot] >> qb;                 // bits 0,1,4,5 = this wave's rows 0..3
            const bool h0 = r0 && (m & 1u), h1 = r1 && (m & 2u), h2 = r2 && (m & 16u), h3 = r3 && (m & 32u);
            const u64 b0 = __ballot(h0), b1 = __ballot(h1), b2 = __ballot(h2), b3 = __ballot(h3);
            if (h0) lists[qb * CHUNK + L0 + __popcll(b0 & lt_mask)] = LIST_ENTRY(slot);
            if (h1) lists[(qb + 1) * CHUNK + L1 + __popcll(b1 & lt_mask)] = LIST_ENTRY(slot);
            if (h2) lists[(qb + 4) * CHUNK + L2 + __popcll(b2 & lt_mask)] = LIST_ENTRY(slot);
            if (h3) lists[(qb + 5) * CHUNK + L3 + __popcll(b3 & lt_mask)] = LIST_ENTRY(slot);
            L0 += __popcll(b0); L1 += __popcll(b1); L2 += __popcll(b2); L3 += __popcll(b3);
        }
        const int Lrow = (row == 0) ? L0 : (row == 1) ? L1 : (row == 2) ? L2 : L3;         // (only the DIAG_PAIRS step reads it)
        (void)Lrow;
        const int Lmax = __builtin_amdgcn_readfirstlane(max(max(L0, L1), max(L2, L3)));
        __builtin_amdgcn_wave_barrier();
        // ---- every row walks its own list; the next list entry is fetched while the current splat is composited
        // This kernel is issue bound, so the walk's bookkeeping counts: the next entry is read unconditionally (the array is
        // padded so that entry Lmax of the last list exists), the record address is one 24-bit multiply by an inline constant,
        // and with the sentinel a row past the end of its list needs no test (it composites the zero record).
        const list_t* mylist = &lists[blk * CHUNK];
        static_assert(sizeof(SplatRec) == 48, "record stride is spelled out in the instruction below");
        struct SplatW { float x, y, ap, cp, bp, o, r, g, b, z; u32 pos; };      // pos: the record's byte offset in `recs`
        auto load = [&](u32 roff) {
            const char* p = reinterpret_cast<const char*>(recs) + roff;
            const float4 a = *reinterpret_cast<const float4*>(p), b4 = *reinterpret_cast<const float4*>(p + 16);
            const float2 c = *reinterpret_cast<const float2*>(p + 32);
            SplatW w; w.x = a.x; w.y = a.y; w.ap = a.z; w.cp = a.w; w.bp = b4.x; w.o = b4.y; w.r = b4.z; w.g = b4.w; w.b = c.x; w.z = c.y; w.pos = roff;
            return w;
        };
        u32 last_off = 0xFFFFFFFFu;                          // byte offset of the record of this chunk's last contributor (per pixel)
        // PTEST: with the `power <= 0` test.  It cannot fail for a well-conditioned conic (the bound above conic_ill_conditioned), so a chunk
        // without an ill-conditioned splat walks without it: one vector instruction of 23 per step (the backward has skipped the test on
        // such TILES since round 4, on this kernel's flag)
        auto step = [&](auto ptest_tag, int i, const SplatW& s) {
            constexpr bool PTEST = decltype(ptest_tag)::value;
            const float dx = s.x - pxf, dy = s.y - pyf;
            const float p2 = pair_power2(dx, dy, s.ap, s.bp, s.cp);
            const float G = __builtin_amdgcn_exp2f(p2);
            const float alpha = fminf(0.99f, s.o * G);
#ifndef DIAG_PAIRS
            // contribute (power <= 0, alpha >= 1/255), then stop (T would fall below 1e-4, or the pixel has finished: sign set) or
            // composite: each v_cmpx narrows EXEC, so there is no mask arithmetic on the scalar unit (the compiler's form of the
            // same logic is 3 s_and + 3 saveexec + 2 s_or + 2 branches per step, and this kernel is as busy on its scalar unit
            // as on its vector units).  EXEC is all ones here: 256-thread workgroups, only wave-uniform control flow above.
            {
                float t0, t1;
#define FWD_STEP_TAIL                                                                                                  \
                    "v_sub_f32 %[t0], 1.0, %[al]\n\t"                                                                  \
                    "v_mul_f32 %[t1], %[al], %[T]\n\t"                                                                 \
                    "v_mul_f32 %[t0], %[T], %[t0]\n\t"                                                                 \
                    "v_cmp_ngt_f32 vcc, 0x38d1b717, %[t0]\n\t"           /* !(1e-4 > T (1 - alpha)): composite; else stop */   \
                    "v_cndmask_b32_e64 %[T], -|%[T]|, %[t0], vcc\n\t"     /* (a finished pixel, T < 0, lands here too and stays as it is) */ \
                    "s_mov_b64 exec, vcc\n\t"                            /* (vcc is 0 in inactive lanes; s_and would clobber SCC, which holds the compiler's loop test) */ \
                    "v_fmac_f32 %[cr], %[t1], %[r]\n\t"                                                                \
                    "v_fmac_f32 %[cg], %[t1], %[g]\n\t"                                                                \
                    "v_fmac_f32 %[cb], %[t1], %[b]\n\t"                                                                \
                    "v_fmac_f32 %[dq], %[t1], %[z]\n\t"                                                                \
                    "v_mov_b32 %[last], %[pos]\n\t"                                                                    \
                    "s_mov_b64 exec, -1"
#define FWD_STEP_OPERANDS                                                                                              \
                    : [T] "+v"(Tq), [cr] "+v"(Cr), [cg] "+v"(Cg), [cb] "+v"(Cb), [dq] "+v"(Dq), [last] "+v"(last_off),   \
                      [t0] "=&v"(t0), [t1] "=&v"(t1)                                                                     \
                    : [p2] "v"(p2), [al] "v"(alpha), [r] "v"(s.r), [g] "v"(s.g), [b] "v"(s.b), [z] "v"(s.z), [pos] "v"(s.pos) \
                    : "vcc"
                if (PTEST) asm volatile(
                    "v_cmpx_ge_f32 vcc, 0, %[p2]\n\t"
                    "v_cmpx_le_f32 vcc, 0x3b808081, %[al]\n\t"          // 1 / 255
                    FWD_STEP_TAIL FWD_STEP_OPERANDS);
                else asm volatile(
                    "v_cmpx_le_f32 vcc, 0x3b808081, %[al]\n\t"
                    FWD_STEP_TAIL FWD_STEP_OPERANDS);
#undef FWD_STEP_TAIL
#undef FWD_STEP_OPERANDS
                (void)i;
            }
#else           // the pair counters live in the plain C++ form of the step (a row past the end of its list composites the sentinel: not counted)
            const bool contrib = (i < Lrow) && (p2 <= 0.f) && (alpha >= ALPHA_MIN) && (Tq > 0.f);
            dg_eval += (i < Lrow) ? 1u : 0u; dg_con += contrib ? 1u : 0u;
            if (contrib) {
                const float test_T = Tq * (1.f - alpha);
                if (test_T < T_EPS) {
                    Tq = -Tq;
                } else {
                    const float w = alpha * Tq;
                    Cr = __fmaf_rn(w, s.r, Cr); Cg = __fmaf_rn(w, s.g, Cg); Cb = __fmaf_rn(w, s.b, Cb);
                    Dq = __fmaf_rn(w, s.z, Dq);
                    Tq = test_T;
                    last_off = s.pos;
                }
            }
#endif
        };
        // FOUR steps per trip, no remainder: entry Lmax of every list is the sentinel (at most CHUNK - 1 real entries, the lists are
        // pre-filled with it), and inline asm is `convergent`, which keeps the compiler from unrolling a loop with a run-time trip
        // count by itself.  The two list bytes of a step pair are one 16-bit read; record address = byte * 48 (SDWA byte select, the
        // factor in a scalar register).
        const unsigned short* mypairs = reinterpret_cast<const unsigned short*>(mylist);
        auto addr2 = [&](u32 two, u32& ra, u32& rb) {
            asm("v_mul_u32_u24_sdwa %0, %2, %3 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_0 src1_sel:DWORD\n\t"
                "v_mul_u32_u24_sdwa %1, %2, %3 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_1 src1_sel:DWORD"
                : "=&v"(ra), "=v"(rb) : "v"(two), "s"(48u));
        };
        // Software pipeline over two register sets, no copies: a record is requested one step before it is composited (its
        // three LDS reads land behind the other record's arithmetic), its address one step pair before that.  Left to itself the
        // compiler issued a step's third read after its exp and waited for it: two exposed LDS latencies per step, and
        // dropping that read (timing experiment) bought 9 % -- latency, not bandwidth.  sched_barrier pins the issue points.
        // Round 5: two step pairs per trip.  A record's byte offset doubles as its "list position" (`pos`), so an address register
        // stays live until its record has been composited; with one pair per trip the next pair's addresses therefore needed
        // registers of their own and two v_mov per trip to rotate them (2 of 49 vector instructions).  Over two pairs the roles
        // alternate and the rotation is a renaming.  (Reads run up to 9 entries past Lmax: the next list or the 16 bytes of padding.)
        auto walk = [&](auto ptest_tag) {
            u32 ra, rb;
            addr2((u32)mypairs[0], ra, rb);
            SplatW S0 = load(ra);
            u32 two = (u32)mypairs[1];                           // entries 2, 3
            for (int i = 0; i < Lmax; i += 4) {
                const SplatW S1 = load(rb);
                u32 ra2, rb2;
                addr2(two, ra2, rb2);
                u32 nxt = (u32)mypairs[(i >> 1) + 2];            // entries i + 4, i + 5
                __builtin_amdgcn_sched_barrier(0);
                step(ptest_tag, i, S0);
                __builtin_amdgcn_sched_barrier(0);
                const SplatW S2 = load(ra2);
                __builtin_amdgcn_sched_barrier(0);
                step(ptest_tag, i + 1, S1);
                asm("" : "+v"(nxt));                             // (keeps the zero extension with the load, not behind the loop's phi)
                __builtin_amdgcn_sched_barrier(0);
                const SplatW S3 = load(rb2);
                u32 ra3, rb3;
                addr2(nxt, ra3, rb3);
                u32 nxt2 = (u32)mypairs[(i >> 1) + 3];           // entries i + 6, i + 7
                __builtin_amdgcn_sched_barrier(0);
                step(ptest_tag, i + 2, S2);
                __builtin_amdgcn_sched_barrier(0);
                S0 = load(ra3);
                __builtin_amdgcn_sched_barrier(0);
                step(ptest_tag, i + 3, S3);
                asm("" : "+v"(nxt2));
                two = nxt2; rb = rb3;
            }
        };
        // (wave-uniform: the four flags were written before the barrier that published the chunk)
        if (s_ill[0] | s_ill[1] | s_ill[2] | s_ill[3]) walk(std::true_type{}); else walk(std::false_type{});
        // record offset -> slot (/ 48: x 43691 >> 21, exact below 2^17 slots) -> 1-based list position
        if (last_off != 0xFFFFFFFFu) last = base + ((last_off * 43691u) >> 21) + 1u;
    }
    const bool stopped = Tq < 0.f;                           // stopped early (T would fall below 1e-4) or outside the image
    Tq = fabsf(Tq);
    if (inside) {
        const size_t HW = (size_t)W * H, pix = (size_t)py * W + px;
        out_color[pix] = Cr + Tq * bg[0];
        out_color[HW + pix] = Cg + Tq * bg[1];
        out_color[2 * HW + pix] = Cb + Tq * bg[2];
        if (out_depth) out_depth[pix] = Dq;
        if (out_weights) out_weights[pix] = 1.f - Tq;
        final_T[pix] = Tq;
        n_contrib[pix] = last;
    }
    // the tile's deepest contributor goes into the tile's descriptor: with it the backward can fetch a tile's first chunk of
    // ids before it has seen a pixel of it
    u32 m = last;
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) m = max(m, (u32)__shfl_xor((int)m, d));
    const u32 early = ((__ballot(stopped) != 0ull) ? 0x80000000u : 0u) | (needle ? 0x40000000u : 0u);
    lds_barrier();                                           // s_live is free again
    if (lane == 0) s_live[wave] = (int)(m | early);
    lds_barrier();
    // bit 31: some pixel of the tile did not walk its whole list (the backward then needs its `pos <= n_contrib` test);
    // bit 30: some staged splat has an ill-conditioned conic (the backward then keeps its `power <= 0` test)
    const u32 w0 = (u32)s_live[0], w1 = (u32)s_live[1], w2 = (u32)s_live[2], w3 = (u32)s_live[3];
    const u32 maxc_all = max(max(w0 & 0x3FFFFFFFu, w1 & 0x3FFFFFFFu), max(w2 & 0x3FFFFFFFu, w3 & 0x3FFFFFFFu));
    if (tid == 0) tile_desc[dslot].w = maxc_all | ((w0 | w1 | w2 | w3) & 0xC0000000u);
    if (test_keep) {
        // compact entries in front of the deepest contributor (positions < maxc: the backward's chunks start there); the
        // positions were stored by this workgroup (waited for below) and are read back at agent scope
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        u32 c = 0;
        for (u32 k = (u32)tid; k < n_live_run; k += 256u)
            c += (__hip_atomic_load(&cpos[k], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < maxc_all) ? 1u : 0u;
#pragma unroll
        for (int d = 32; d >= 1; d >>= 1) c += (u32)__shfl_xor((int)c, d);
        if (lane == 0) s_cnt[wave] = c;
        __syncthreads();
        if (tid == 0) tile_aux[dslot] = make_uint4(s_cnt[0] + s_cnt[1] + s_cnt[2] + s_cnt[3], n_live_run, n_staged, 0u);
    }
    DG(diag_pairs_flush(2, dg_eval, dg_con);)
}

hipError_t launch_blend_fwd(const BagsSettings& s, const GeomView& g, const BinView& b, const ImgView& im,
                            const BagsForwardOut& out, hipStream_t st, const u32* n_dev, u32 capacity, bool sort_here)
{
    const int gx = cdiv(s.image_width, BAGS_TILE), gy = cdiv(s.image_height, BAGS_TILE);
    const int T = gx * gy;
    if (T == 0) return hipSuccess;
    const int grid = cdiv(T, TILE_RUN) * TILE_RUN;
    LAUNCH_K(blend_fwd_rows_kernel<0>, dim3(grid), dim3(256), 0, st, s.image_width, s.image_height, gx, T,
                       im.tile_desc, b.point_list, sort_here ? b.words : nullptr, g.depth_key, b.scratch,
                       reinterpret_cast<unsigned char*>(b.reach_mask), b.words ? 8u : 4u, g.g2d, s.bg, out.color, out.depth, out.weights,
                       im.final_T, im.n_contrib, n_dev, capacity, (b.words && s.tile_bounds != BAGS_TILES_OPACITY) ? 1 : 0, im.tile_aux);
    return hipGetLastError();
}
