// api.hip -- the extern "C" surface of libbags_raster.so (include/bags_raster.h) and the state-buffer layout.
#include "bags_common.h"
#include <stdio.h>
#include <string.h>
#include <stdarg.h>

static thread_local char g_err[512] = "";

static int fail(int code, const char* fmt, ...)
{
    va_list ap; va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
    return code;
}

#define HIP_TRY(expr)                                                                                   \
    do { hipError_t e_ = (expr);                                                                        \
         if (e_ != hipSuccess) return fail(BAGS_ERR_HIP, "%s failed: %s (%s:%d)", #expr, hipGetErrorString(e_), __FILE__, __LINE__); \
    } while (0)

// after a launcher when settings.debug is on: drain the stream so a faulting kernel is attributed correctly
#define DEBUG_SYNC(s, st, what)                                                                         \
    do { if ((s)->debug) { hipError_t e_ = hipStreamSynchronize(st);                                    \
         if (e_ != hipSuccess) return fail(BAGS_ERR_HIP, "debug: %s failed at iteration %d: %s", what, (s)->debug_iter, hipGetErrorString(e_)); } \
    } while (0)

// settings.debug also scans what a call produced for NaN / Inf (the reference's debug flag dumps a snapshot
// when the CUDA op throws; here the op names the first tensor that went non-finite and the iteration it happened in).
__global__ void __launch_bounds__(256) count_nonfinite_kernel(const float* __restrict__ x, size_t n, u32* __restrict__ counter)
{
    u32 bad = 0;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
        const u32 b = __float_as_uint(x[i]);
        bad += ((b & 0x7F800000u) == 0x7F800000u) ? 1u : 0u;          // exponent all ones: Inf or NaN
    }
    for (int d = 32; d >= 1; d >>= 1) bad += (u32)__shfl_xor((int)bad, d);
    if ((threadIdx.x & 63) == 0 && bad) atomicAdd(counter, bad);
}
struct ScanItem { const char* name; const float* ptr; size_t n; };
// counters: device words nobody else uses (GeomView::num_rendered[8 ...], 64 words are carved for it); one per item
static int debug_scan(const BagsSettings* s, hipStream_t st, u32* counters, const ScanItem* items, int n_items, const char* phase,
                      int (*failfn)(int, const char*, ...))
{
    if (!s->debug) return BAGS_OK;
    if (n_items > 32) n_items = 32;
    if (hipMemsetAsync(counters, 0, sizeof(u32) * 32, st) != hipSuccess) return failfn(BAGS_ERR_HIP, "debug scan: memset failed");
    for (int i = 0; i < n_items; ++i) {
        if (!items[i].ptr || items[i].n == 0) continue;
        const int grid = (int)((items[i].n + 256 * 16 - 1) / (256 * 16));
        hipLaunchKernelGGL(count_nonfinite_kernel, dim3(grid < 1 ? 1 : (grid > 4096 ? 4096 : grid)), dim3(256), 0, st, items[i].ptr, items[i].n, counters + i);
    }
    u32 host[32];
    if (hipMemcpyAsync(host, counters, sizeof(u32) * 32, hipMemcpyDeviceToHost, st) != hipSuccess || hipStreamSynchronize(st) != hipSuccess)
        return failfn(BAGS_ERR_HIP, "debug scan: %s", hipGetErrorString(hipGetLastError()));
    for (int i = 0; i < n_items; ++i)
        if (items[i].ptr && host[i])
            return failfn(BAGS_ERR_DEVICE, "debug: %s produced %u non-finite values in %s at iteration %d", phase, host[i], items[i].name, s->debug_iter);
    return BAGS_OK;
}

// ---------------------------------------------------------------------------------------------- stage profiler
// Opt-in (bags_profile_enable): a start/stop hipEvent pair per stage per call, resolved in bags_profile_read.
#include <atomic>
#include <mutex>
#include <vector>
enum Stage { ST_PRE_FWD, ST_DEPTH_SORT, ST_OFFSETS, ST_EMIT, ST_TILE_SORT, ST_RANGES, ST_BLEND_FWD, ST_BLEND_BWD,
             ST_PRE_BWD, ST_POSE_REDUCE, ST_COUNT };
static const char* kStageNames[ST_COUNT] = {"preprocess_fwd", "depth_sort", "offsets_scan", "emit", "tile_sort",
                                            "tile_ranges", "blend_fwd", "blend_bwd", "preprocess_bwd", "pose_reduce"};
struct ProfInterval { int stage; hipEvent_t a, b; };
// The profiler is the library's only mutable global state (everything a call needs lives in the caller's buffers); calls
// from several threads / streams may profile at once, so the lists are guarded.
static std::mutex g_prof_mutex;
static int g_prof_mode = 0;          // 0 off, 1 dominant kernel only (blend_bwd), 2 every stage
static int g_prof_stride = 1;        // mode 1: every n-th launch of the dominant kernel carries events (bags_profile_stride)
static std::atomic<unsigned long long> g_prof_seq{0};
static std::vector<ProfInterval> g_prof_pending;
static std::vector<hipEvent_t> g_prof_free;
static double g_prof_ms[ST_COUNT];
static long long g_prof_calls[ST_COUNT];

static hipEvent_t prof_event()
{
    {
        std::lock_guard<std::mutex> lk(g_prof_mutex);
        if (!g_prof_free.empty()) { hipEvent_t e = g_prof_free.back(); g_prof_free.pop_back(); return e; }
    }
    hipEvent_t e = nullptr; (void)hipEventCreate(&e); return e;
}
thread_local hipEvent_t g_attach_start = nullptr, g_attach_stop = nullptr;     // (bags_common.h: launch_k)
struct ProfScope {
    hipStream_t st; int stage; hipEvent_t a = nullptr, b = nullptr; bool attached = false;
    // mode 2 (every stage, bench.py's separate untimed pass).  attach: the stage is ONE kernel -- the events ride on its dispatch (no
    // packet of their own: what they report is the kernel's time, as rocprofv3 does); otherwise the stage is bracketed by two recorded
    // events.  Mode 1 (the dominant kernel alone, inside bench.py's timed region) is handled in bags_backward.
    ProfScope(int stage_, hipStream_t st_, bool attach = false) : st(st_), stage(stage_) {
        if (g_prof_mode != 2) return;
        a = prof_event();
        if (attach) { b = prof_event(); attached = true; g_attach_start = a; g_attach_stop = b; }
        else (void)hipEventRecord(a, st);
    }
    ~ProfScope() {
        if (!a) return;
        if (attached) {
            const bool taken = (g_attach_start == nullptr && g_attach_stop == nullptr);
            g_attach_start = nullptr; g_attach_stop = nullptr;
            std::lock_guard<std::mutex> lk(g_prof_mutex);
            if (taken) g_prof_pending.push_back({stage, a, b});
            else { g_prof_free.push_back(a); g_prof_free.push_back(b); }       // the launcher launched nothing
            return;
        }
        b = prof_event(); (void)hipEventRecord(b, st);
        std::lock_guard<std::mutex> lk(g_prof_mutex);
        g_prof_pending.push_back({stage, a, b});
    }
};

// ---------------------------------------------------------------------------------------------- buffer carving
template <typename Tp>
static inline void take(char*& p, Tp*& out, size_t count)
{
    out = reinterpret_cast<Tp*>(p);
    p += align_up(count * sizeof(Tp), 256);
}

size_t carve_geom(void* base, int P, GeomView* v)
{
    char* p = reinterpret_cast<char*>(base);
    GeomView g;
    const size_t n = (size_t)(P > 0 ? P : 1);
    g.nblocks_sort = radix_blocks_for((long long)n);
    g.nblocks_scan = cdiv((long long)n, SCAN_TILE);
    take(p, g.depth_key, n); take(p, g.g2d, 4 * n); take(p, g.rect, n); take(p, g.tiles_touched, n); take(p, g.inst_off, n);
    take(p, g.keep, n); take(p, g.shjac, 10 * n); take(p, g.rec_count, n);
    take(p, g.keys_a, n); take(p, g.keys_b, n); take(p, g.vals_a, n); take(p, g.vals_b, n);
    take(p, g.rank_offset, n);
    take(p, g.scan_partials, (size_t)g.nblocks_scan + 1);
    take(p, g.radix_hist, (size_t)RADIX_BINS * g.nblocks_sort);
    take(p, g.digit_totals, RADIX_BINS);
    take(p, g.num_rendered, 64);
    take(p, g.local_off, n); take(p, g.block_total, 256); take(p, g.block_base, 256);
    if (v) *v = g;
    return (size_t)(p - reinterpret_cast<char*>(base));
}

static int tile_passes(int T) { return (bit_length((u32)(T > 1 ? T - 1 : 1)) + RADIX_BITS - 1) / RADIX_BITS; }

size_t carve_binning(void* base, long long I, int W, int H, BinView* v, bool binned)
{
    char* p = reinterpret_cast<char*>(base);
    BinView b;
    const size_t n = (size_t)(I > 0 ? I : 1);
    const int T = cdiv(W, BAGS_TILE) * cdiv(H, BAGS_TILE);
    b.nblocks_sort = radix_blocks_for((long long)n);
    b.passes = tile_passes(T);
    b.words = nullptr; b.scratch = nullptr;
    if (binned) {                                            // tile-binned path: 20 bytes per instance
        take(p, b.words, n); take(p, b.scratch, n); take(p, b.point_list, n);
        b.keys_a = b.keys_b = b.vals_a = b.vals_b = nullptr; b.ranges = nullptr; b.radix_hist = b.digit_totals = nullptr;
        b.tile_sorted = nullptr;
        b.reach_mask = reinterpret_cast<unsigned short*>(b.words);       // the unsorted words are dead after the per-tile sorts
    } else {
        take(p, b.keys_a, n); take(p, b.vals_a, n); take(p, b.keys_b, n); take(p, b.vals_b, n);
        take(p, b.ranges, (size_t)(T > 0 ? T : 1));
        take(p, b.radix_hist, (size_t)RADIX_BINS * b.nblocks_sort);
        take(p, b.digit_totals, RADIX_BINS);
        // emission writes the *_b half; pass 0: b -> a, pass 1: a -> b, ...
        b.point_list = (b.passes & 1) ? b.vals_a : b.vals_b;
        b.tile_sorted = (b.passes & 1) ? b.keys_a : b.keys_b;
        b.reach_mask = reinterpret_cast<unsigned short*>((b.passes & 1) ? b.keys_b : b.keys_a);   // the ping-pong half the last pass read
    }
    if (v) *v = b;
    return (size_t)(p - reinterpret_cast<char*>(base));
}

size_t carve_image(void* base, int W, int H, ImgView* v)
{
    char* p = reinterpret_cast<char*>(base);
    ImgView im;
    const size_t n = (size_t)W * H > 0 ? (size_t)W * H : 1;
    take(p, im.final_T, n); take(p, im.n_contrib, n);
    const size_t T = (size_t)cdiv(W > 0 ? W : 1, BAGS_TILE) * cdiv(H > 0 ? H : 1, BAGS_TILE);
    take(p, im.tile_desc, T); take(p, im.n_active, 64); take(p, im.tile_aux, T);
    im.cnt_rows = im.pre = im.tile_total = nullptr; im.ranges = nullptr; im.tile_lstart = im.group_total = nullptr;
    if (binned_supported(1, (int)T)) {                        // images the tile-binned path can take (<= 32768 tiles)
        take(p, im.cnt_rows, 256 * ((T + 1) / 2)); take(p, im.pre, 256 * T); take(p, im.tile_total, T); take(p, im.ranges, T);
        take(p, im.tile_lstart, T); take(p, im.group_total, 512);
    }
    if (v) *v = im;
    return (size_t)(p - reinterpret_cast<char*>(base));
}

// ---------------------------------------------------------------------------------------------- validation
static int check_common(const BagsSettings* s, const BagsInputs* in, const BagsState* stt)
{
    if (!s || !in || !stt) return fail(BAGS_ERR_ARG, "null argument struct");
    if (in->P < 0) return fail(BAGS_ERR_ARG, "P < 0");
    if (s->image_width <= 0 || s->image_height <= 0) return fail(BAGS_ERR_ARG, "empty image %dx%d", s->image_width, s->image_height);
    if (cdiv(s->image_width, BAGS_TILE) > 65535 || cdiv(s->image_height, BAGS_TILE) > 65535)
        return fail(BAGS_ERR_ARG, "image too large for 16-bit tile coordinates");
    if (s->sh_degree < 0 || s->sh_degree > 3) return fail(BAGS_ERR_ARG, "sh_degree %d not in 0..3", s->sh_degree);
    if (!s->bg || !s->viewmatrix || !s->projmatrix || !s->intrinsic || !s->campos)
        return fail(BAGS_ERR_ARG, "bg/viewmatrix/projmatrix/intrinsic/campos must be given");
    if (in->P > 0) {
        if (!in->means3D || !in->opacities) return fail(BAGS_ERR_ARG, "means3D/opacities must be given");
        if ((in->shs != nullptr) == (in->colors_precomp != nullptr))
            return fail(BAGS_ERR_ARG, "Please provide excatly one of either SHs or precomputed colors!");
        const bool sr = in->scales && in->rotations;
        if ((in->scales != nullptr) != (in->rotations != nullptr) || sr == (in->cov3D_precomp != nullptr))
            return fail(BAGS_ERR_ARG, "Please provide exactly one of either scale/rotation pair or precomputed 3D covariance!");
        if (in->shs_rest && (!in->shs || s->sh_coeffs < 2))
            return fail(BAGS_ERR_ARG, "shs_rest (features_rest) needs shs (features_dc) and sh_coeffs >= 2 (got %d)", s->sh_coeffs);
        if (in->shs && (s->sh_degree + 1) * (s->sh_degree + 1) > s->sh_coeffs)
            return fail(BAGS_ERR_ARG, "sh_degree %d needs %d coefficients, shs holds %d", s->sh_degree,
                        (s->sh_degree + 1) * (s->sh_degree + 1), s->sh_coeffs);
    }
    if (!stt->geom || stt->geom_bytes < bags_geom_size(in->P)) return fail(BAGS_ERR_SIZE, "geom buffer too small");
    if (!stt->image || stt->image_bytes < bags_image_size(s->image_width, s->image_height))
        return fail(BAGS_ERR_SIZE, "image buffer too small");
    return BAGS_OK;
}


// ---------------------------------------------------------------------------------------------- C ABI
extern "C" {

int bags_abi_version(void) { return BAGS_ABI_VERSION; }
#ifndef BAGS_SRC_HASH
#define BAGS_SRC_HASH "unknown"
#define BAGS_KERNEL_COMMIT "unknown"
#endif
const char* bags_build_info(void) { return "src=" BAGS_SRC_HASH " commit=" BAGS_KERNEL_COMMIT; }
const char* bags_last_error(void) { return g_err; }

size_t bags_geom_size(int32_t P) { return carve_geom(nullptr, P, nullptr) + 256; }
size_t bags_binning_size(int64_t I, int32_t W, int32_t H)
{   // the caller does not know which path a call takes: the larger of the two layouts
    const size_t a = carve_binning(nullptr, I, W, H, nullptr, false), b = carve_binning(nullptr, I, W, H, nullptr, true);
    return (a > b ? a : b) + 256;
}
size_t bags_image_size(int32_t W, int32_t H) { return carve_image(nullptr, W, H, nullptr) + 256; }
size_t bags_backward_workspace_size(int32_t P, int64_t I)
{
    const size_t part = align_up((size_t)(I > 0 ? I : 1) * PART_FLOATS * sizeof(float), 256);
    const size_t slab = align_up((size_t)(cdiv(P > 0 ? P : 1, 256)) * POSE_VALS * sizeof(float), 256);
    // dense-scene mode: one byte per record, + the 64 bytes preprocess_bwd reads from a Gaussian's first mark on (launch_blend_bwd clears
    // the same number of bytes)
    const size_t live = align_up((size_t)(I > 0 ? I : 1) + 64, 256);
    return part + slab + live + 256;
}

// tile-binned lists (binning.hip) unless the caller asked for the radix path or the problem is outside their limits
static bool use_binned(const BagsSettings* s, int P)
{
    const int T = cdiv(s->image_width, BAGS_TILE) * cdiv(s->image_height, BAGS_TILE);
    return s->binning != BAGS_BINNING_RADIX && binned_supported(P, T);
}

static inline void* align256(void* p) { return reinterpret_cast<void*>(align_up(reinterpret_cast<size_t>(p), 256)); }

// K1 + everything the instance count needs; leaves it in g.num_rendered (device)
// host_count (optional): device-visible address of the caller's pinned host word; when the tile-binned path takes it, the
// count is written there by the kernel that computes it and *host_written is set (no copy needed)
static int enqueue_prepare(const BagsSettings* s, const BagsInputs* in, const GeomView& g, const ImgView& im, const BagsForwardOut* out,
                           hipStream_t st, u32* host_count = nullptr, bool* host_written = nullptr)
{
    const bool binned = use_binned(s, in->P);
    { ProfScope ps(ST_PRE_FWD, st, true);
      HIP_TRY(launch_preprocess_fwd(*s, *in, g, out->radii, out->mean2D, st, binned ? &im : nullptr, cdiv(s->image_width, BAGS_TILE))); }
    DEBUG_SYNC(s, st, "preprocess_fwd");
    if (binned) {
        // (block of Gaussians, tile) count matrix -> column prefixes -> tile ranges, instance count, heavy-first tile list
        const int gx = cdiv(s->image_width, BAGS_TILE), gy = cdiv(s->image_height, BAGS_TILE);
        // host_count given (device-visible pinned word, speculative forward): no ranges_order launch -- the emission launch of
        // the second phase computes the ranges itself and delivers the count there
        { ProfScope ps(ST_OFFSETS, st, host_count != nullptr);          // (one kernel unless the count is wanted now: then ranges_order follows)
          HIP_TRY(launch_binned_prepare(g, im, in->P, gx, gx * gy, st, host_count, host_count == nullptr)); }
        if (host_count && host_written) *host_written = true;
        DEBUG_SYNC(s, st, "tile count / prefix / ranges");
        return BAGS_OK;
    }
    // depth order of the Gaussians: 4 stable 8-bit passes over the float bits (positive floats order like u32)
    { ProfScope ps(ST_DEPTH_SORT, st);
      HIP_TRY(launch_radix_sort(g.depth_key, nullptr, g.keys_a, g.vals_a, g.keys_b, g.vals_b, in->P, 32, true,
                                g.radix_hist, g.digit_totals, g.nblocks_sort, st)); }
    DEBUG_SYNC(s, st, "depth sort");
    const u32* sorted_ids = g.vals_b;                     // 4 passes: src->a->b->a->b
    { ProfScope ps(ST_OFFSETS, st); HIP_TRY(launch_offsets_scan(g, sorted_ids, in->P, st)); }
    DEBUG_SYNC(s, st, "offsets scan");
    return BAGS_OK;
}

static int scan_forward(const BagsSettings* s, const BagsInputs* in, const GeomView& g, const BagsForwardOut* out, hipStream_t st)
{
    if (!s->debug) return BAGS_OK;
    const size_t HW = (size_t)s->image_width * s->image_height;
    const ScanItem items[] = {{"rendered_image", out->color, 3 * HW}, {"depth", out->depth, HW}, {"weights", out->weights, HW},
                              {"mean2D", out->mean2D, 2 * (size_t)in->P}};
    return debug_scan(s, st, g.num_rendered + 8, items, 4, "the forward", fail);
}

// emission, per-tile ordering, blend.  n_dev != nullptr: the instance count is read on the device and checked against
// `I` (the capacity the binning buffer was sized for)
static int enqueue_finish(const BagsSettings* s, const BagsInputs* in, const GeomView& g, const BinView& b, const ImgView& im,
                          const BagsForwardOut* out, int64_t I, const u32* n_dev, hipStream_t st, bool speculative = false)
{
    const int W = s->image_width, H = s->image_height;
    const int gx = cdiv(W, BAGS_TILE), gy = cdiv(H, BAGS_TILE);
    if (use_binned(s, in->P)) {
        if (in->P == 0) HIP_TRY(launch_binned_empty(g, im, gx * gy, st));      // no prepare phase ran: an all-empty tile list
        if (I > 0 && in->P > 0) {
            ProfScope ps(ST_TILE_SORT, st, true);
            HIP_TRY(launch_binned_finish(g, im, in->P, gx, gx * gy, b.words, (u32)I, st, speculative));
        } else if (in->P > 0) {
            HIP_TRY(launch_binned_desc_only(im, gx * gy, st));                   // nothing to emit: only the (all-empty) tile list
        }
        DEBUG_SYNC(s, st, "emit / tile sort");
        { ProfScope ps(ST_BLEND_FWD, st, true); HIP_TRY(launch_blend_fwd(*s, g, b, im, *out, st, n_dev, (u32)I, I > 0 && in->P > 0)); }
        DEBUG_SYNC(s, st, "blend_fwd");
        return scan_forward(s, in, g, out, st);
    }
    if (I > 0) {
        { ProfScope ps(ST_EMIT, st); HIP_TRY(launch_emit(g, g.vals_b, in->P, gx, b.keys_b, b.vals_b, (u32)I, st, n_dev, b.ranges, gx * gy)); }
        DEBUG_SYNC(s, st, "emit");
        { ProfScope ps(ST_TILE_SORT, st);
          HIP_TRY(launch_radix_sort(b.keys_b, b.vals_b, b.keys_a, b.vals_a, b.keys_b, b.vals_b, I, b.passes * RADIX_BITS,
                                    false, b.radix_hist, b.digit_totals, b.nblocks_sort, st, n_dev)); }
        DEBUG_SYNC(s, st, "tile sort");
    }
    { ProfScope ps(ST_RANGES, st);
      HIP_TRY(launch_tile_ranges(b.tile_sorted, I, b.ranges, gx * gy, st, n_dev, I > 0));
      HIP_TRY(launch_tile_order(b.ranges, gx * gy, im.tile_desc, im.n_active, st)); }
    DEBUG_SYNC(s, st, "tile ranges");
    { ProfScope ps(ST_BLEND_FWD, st, true); HIP_TRY(launch_blend_fwd(*s, g, b, im, *out, st)); }
    DEBUG_SYNC(s, st, "blend_fwd");
    return scan_forward(s, in, g, out, st);
}

int bags_forward_prepare(const BagsSettings* s, const BagsInputs* in, const BagsState* stt, const BagsForwardOut* out,
                         int64_t* host_num_rendered, void* stream)
{
    int rc = check_common(s, in, stt);
    if (rc) return rc;
    if (!out || (in->P > 0 && !out->radii) || !host_num_rendered) return fail(BAGS_ERR_ARG, "radii / host_num_rendered must be given");
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    GeomView g; carve_geom(align256(stt->geom), in->P, &g);
    ImgView im; carve_image(align256(stt->image), s->image_width, s->image_height, &im);
    *host_num_rendered = 0;
    if (in->P == 0) return BAGS_OK;
    rc = enqueue_prepare(s, in, g, im, out, st);
    if (rc) return rc;
    u32 host_I = 0;
    HIP_TRY(hipMemcpyAsync(&host_I, g.num_rendered, sizeof(u32), hipMemcpyDeviceToHost, st));
    HIP_TRY(hipStreamSynchronize(st));
    *host_num_rendered = (int64_t)host_I;
    return BAGS_OK;
}

int bags_forward_finish(const BagsSettings* s, const BagsInputs* in, const BagsState* stt, const BagsForwardOut* out,
                        int64_t I, void* stream)
{
    int rc = check_common(s, in, stt);
    if (rc) return rc;
    if (!out || !out->color) return fail(BAGS_ERR_ARG, "color output must be given");
    if (I < 0 || I > 0xFFFFFFF0ll) return fail(BAGS_ERR_ARG, "num_rendered out of range");
    const int W = s->image_width, H = s->image_height;
    if (!stt->binning || stt->binning_bytes < bags_binning_size(I, W, H)) return fail(BAGS_ERR_SIZE, "binning buffer too small");
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    GeomView g; carve_geom(align256(stt->geom), in->P, &g);
    BinView b; carve_binning(align256(stt->binning), I, W, H, &b, use_binned(s, in->P));
    ImgView im; carve_image(align256(stt->image), W, H, &im);
    // the count is compared with I on the device as well: a caller-supplied I below the true count renders the overflowing
    // lists empty instead of writing past the binning buffer (same guard as the speculative finish)
    return enqueue_finish(s, in, g, b, im, out, I, in->P > 0 ? g.num_rendered : nullptr, st);
}

int bags_forward_prepare_async(const BagsSettings* s, const BagsInputs* in, const BagsState* stt, const BagsForwardOut* out,
                               uint32_t* host_num_rendered, void* stream)
{
    int rc = check_common(s, in, stt);
    if (rc) return rc;
    if (!out || (in->P > 0 && !out->radii) || !host_num_rendered) return fail(BAGS_ERR_ARG, "radii / host_num_rendered must be given");
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    GeomView g; carve_geom(align256(stt->geom), in->P, &g);
    ImgView im; carve_image(align256(stt->image), s->image_width, s->image_height, &im);
    if (in->P == 0) {
        HIP_TRY(hipMemsetAsync(g.num_rendered, 0, sizeof(u32), st));
    } else {
        // a pinned (page-locked, device-mapped) host word can be written by the kernel that computes the count
        u32* dev_alias = nullptr;
        hipPointerAttribute_t attr;
        if (hipPointerGetAttributes(&attr, host_num_rendered) == hipSuccess && attr.type == hipMemoryTypeHost && attr.devicePointer)
            dev_alias = static_cast<u32*>(attr.devicePointer);
        else (void)hipGetLastError();                         // not a registered pointer: fall back to the copy
        bool written = false;
        rc = enqueue_prepare(s, in, g, im, out, st, dev_alias, &written);
        if (rc) return rc;
        if (written) return BAGS_OK;
    }
    HIP_TRY(hipMemcpyAsync(host_num_rendered, g.num_rendered, sizeof(u32), hipMemcpyDeviceToHost, st));
    return BAGS_OK;
}

int bags_forward_finish_speculative(const BagsSettings* s, const BagsInputs* in, const BagsState* stt,
                                    const BagsForwardOut* out, int64_t capacity, void* stream)
{
    int rc = check_common(s, in, stt);
    if (rc) return rc;
    if (!out || !out->color) return fail(BAGS_ERR_ARG, "color output must be given");
    if (capacity < 1 || capacity > 0xFFFFFFF0ll) return fail(BAGS_ERR_ARG, "capacity out of range");
    const int W = s->image_width, H = s->image_height;
    if (!stt->binning || stt->binning_bytes < bags_binning_size(capacity, W, H)) return fail(BAGS_ERR_SIZE, "binning buffer too small");
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    GeomView g; carve_geom(align256(stt->geom), in->P, &g);
    BinView b; carve_binning(align256(stt->binning), capacity, W, H, &b, use_binned(s, in->P));
    ImgView im; carve_image(align256(stt->image), W, H, &im);
    return enqueue_finish(s, in, g, b, im, out, capacity, g.num_rendered, st, true);
}

int bags_backward(const BagsSettings* s, const BagsInputs* in, const BagsState* stt, const BagsBackwardArgs* a, void* stream)
{
    int rc = check_common(s, in, stt);
    if (rc) return rc;
    if (!a || !a->grad_color) return fail(BAGS_ERR_ARG, "grad_color must be given");
    const int W = s->image_width, H = s->image_height;
    const int64_t I = a->num_rendered;
    if (I < 0) return fail(BAGS_ERR_ARG, "num_rendered < 0");
    const int64_t cap = a->binning_capacity > 0 ? a->binning_capacity : I;     // what the forward carved the binning buffer for
    if (cap < I) return fail(BAGS_ERR_ARG, "binning_capacity %lld < num_rendered %lld", (long long)cap, (long long)I);
    if (!stt->binning || stt->binning_bytes < bags_binning_size(cap, W, H)) return fail(BAGS_ERR_SIZE, "binning buffer too small");
    if (!a->workspace || a->workspace_bytes < bags_backward_workspace_size(in->P, I)) return fail(BAGS_ERR_SIZE, "backward workspace too small");
    if (a->phase < BAGS_BWD_ALL || a->phase > BAGS_BWD_PREPROCESS) return fail(BAGS_ERR_ARG, "phase %d is not BAGS_BWD_ALL / _BLEND / _PREPROCESS", a->phase);
    if (a->grad_dldc && (!in->shs || in->colors_precomp || a->grad_shs || a->grad_shs_rest))
        return fail(BAGS_ERR_ARG, "grad_dldc (factored SH gradient) goes with inputs.shs and WITHOUT grad_shs / grad_shs_rest");
    if (in->shs_rest ? ((a->grad_shs != nullptr) != (a->grad_shs_rest != nullptr)) : (a->grad_shs_rest != nullptr))
        return fail(BAGS_ERR_ARG, "grad_shs_rest goes with inputs.shs_rest, and then grad_shs (features_dc) and grad_shs_rest are given together");
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    GeomView g; carve_geom(align256(stt->geom), in->P, &g);
    BinView b; carve_binning(align256(stt->binning), cap, W, H, &b, use_binned(s, in->P));
    ImgView im; carve_image(align256(stt->image), W, H, &im);
    char* ws = reinterpret_cast<char*>(align256(a->workspace));
    float* partials = reinterpret_cast<float*>(ws);
    float* slab = reinterpret_cast<float*>(ws + align_up((size_t)(I > 0 ? I : 1) * PART_FLOATS * sizeof(float), 256));
    unsigned char* live_map = reinterpret_cast<unsigned char*>(slab) + align_up((size_t)(cdiv(in->P > 0 ? in->P : 1, 256)) * POSE_VALS * sizeof(float), 256);
    // the dense-scene mode (a byte per gradient record instead of zero records) is decided HERE, once: both launchers get the map or null
    const bool dense = I > 0 && bwd_dense_mode(I, cdiv(W, BAGS_TILE) * cdiv(H, BAGS_TILE), a->dense_per_tile);
    if (I > 0 && a->phase != BAGS_BWD_PREPROCESS) {
        // Profile mode 1 (bench.py's timed region: the dominant kernel's launch time for `roofline`): the start / stop events are attached
        // to the kernel's OWN dispatch (hipExtLaunchKernelGGL: the timestamps of its completion signal), not recorded around it.  Two
        // hipEventRecord calls are two more packets in the queue of a step whose seven launches otherwise follow each other without a
        // gap: they cost the timed region 10-25 us per step (0.605 against 0.594 ms on one device, 0.637 against 0.611 on the driver's
        // box of round 5), i.e. the measurement slowed down what it measured.
        hipEvent_t ea = nullptr, eb = nullptr;
        if (g_prof_mode == 1 && (g_prof_seq.fetch_add(1, std::memory_order_relaxed) % (unsigned long long)g_prof_stride) == 0ull) { ea = prof_event(); eb = prof_event(); }
        { ProfScope ps(ST_BLEND_BWD, st, true); HIP_TRY(launch_blend_bwd(*s, g, b, im, a->grad_color, partials, a->grad_means2D_densify != nullptr, use_binned(s, in->P), st,
                                                                  I, dense ? live_map : nullptr, ea, eb)); }
        if (ea) { std::lock_guard<std::mutex> lk(g_prof_mutex); g_prof_pending.push_back({ST_BLEND_BWD, ea, eb}); }
        DEBUG_SYNC(s, st, "blend_bwd");
    }
    if (a->phase == BAGS_BWD_BLEND) return BAGS_OK;          // the per-Gaussian half comes with a second call (BAGS_BWD_PREPROCESS)
    int nblocks = 0;
    { ProfScope ps(ST_PRE_BWD, st, true); HIP_TRY(launch_preprocess_bwd(*s, *in, g, nullptr, partials, slab, &nblocks, *a, st, use_binned(s, in->P), dense ? live_map : nullptr)); }
    DEBUG_SYNC(s, st, "preprocess_bwd");
    { ProfScope ps(ST_POSE_REDUCE, st, true); HIP_TRY(launch_pose_reduce(slab, nblocks, *a, st)); }
    DEBUG_SYNC(s, st, "pose_reduce");
    if (s->debug) {
        const size_t P = (size_t)in->P;
        const ScanItem items[] = {{"grad_means3D", a->grad_means3D, 3 * P}, {"grad_means2D", a->grad_means2D, 3 * P},
                                  {"grad_means2D_densify", a->grad_means2D_densify, 3 * P}, {"grad_shs", a->grad_shs, 3 * P * (size_t)(in->shs_rest ? 1 : s->sh_coeffs)},
                                  {"grad_shs_rest", in->shs_rest ? a->grad_shs_rest : nullptr, 3 * P * (size_t)(s->sh_coeffs - 1)},
                                  {"grad_colors_precomp", a->grad_colors_precomp, 3 * P}, {"grad_opacities", a->grad_opacities, P},
                                  {"grad_scales", a->grad_scales, 3 * P}, {"grad_rotations", a->grad_rotations, 4 * P},
                                  {"grad_cov3D_precomp", a->grad_cov3D_precomp, 6 * P}, {"grad_viewmatrix", a->grad_viewmatrix, 16},
                                  {"grad_projmatrix", a->grad_projmatrix, 16}, {"grad_intrinsic", a->grad_intrinsic, 16},
                                  {"grad_campos", a->grad_campos, 3}, {"grad_shift_factors", a->grad_shift_factors, 3}};
        return debug_scan(s, st, g.num_rendered + 8, items, 15, "the backward", fail);
    }
    return BAGS_OK;
}

int bags_sh_gradient_from_views(int32_t P, int32_t M, int32_t sh_degree, const float* means3D, const BagsShViews* views,
                                float* grad_shs, float* grad_shs_rest, int32_t accumulate, void* stream)
{
    if (P < 0) return fail(BAGS_ERR_ARG, "sh_gradient_from_views: P < 0");
    if (!views || views->n_views < 0 || views->n_views > BAGS_MAX_SH_VIEWS)
        return fail(BAGS_ERR_ARG, "sh_gradient_from_views: n_views must be 0..%d (more views: a second call with accumulate = 1)", BAGS_MAX_SH_VIEWS);
    if (sh_degree < 0 || sh_degree > 3 || M < 1 || (sh_degree + 1) * (sh_degree + 1) > M)
        return fail(BAGS_ERR_ARG, "sh_gradient_from_views: sh_degree %d needs %d coefficients, M = %d", sh_degree, (sh_degree + 1) * (sh_degree + 1), M);
    if (grad_shs_rest && M < 2) return fail(BAGS_ERR_ARG, "sh_gradient_from_views: grad_shs_rest needs M >= 2");
    if (P == 0 || views->n_views == 0) return BAGS_OK;
    if (!means3D || !grad_shs) return fail(BAGS_ERR_ARG, "sh_gradient_from_views: means3D / grad_shs must be given");
    for (int v = 0; v < views->n_views; ++v)
        if (!views->campos[v] || !views->dldc[v]) return fail(BAGS_ERR_ARG, "sh_gradient_from_views: view %d has a NULL campos / dldc", v);
    HIP_TRY(launch_sh_grad_from_views(P, M, sh_degree, means3D, *views, grad_shs, grad_shs_rest, accumulate, reinterpret_cast<hipStream_t>(stream)));
    return BAGS_OK;
}

int bags_debug_views(const BagsSettings* s, const BagsInputs* in, const BagsState* stt, int64_t I,
                     const BagsDebugViews* d, void* stream)
{
    int rc = check_common(s, in, stt);
    if (rc) return rc;
    if (!d) return fail(BAGS_ERR_ARG, "null views");
    const int W = s->image_width, H = s->image_height;
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    GeomView g; carve_geom(align256(stt->geom), in->P, &g);
    ImgView im; carve_image(align256(stt->image), W, H, &im);
    const size_t P = (size_t)in->P, T = (size_t)cdiv(W, BAGS_TILE) * cdiv(H, BAGS_TILE);
    if (d->tiles_touched && P) HIP_TRY(hipMemcpyAsync(d->tiles_touched, g.tiles_touched, P * 4, hipMemcpyDeviceToDevice, st));
    if (d->depth_bits && P) HIP_TRY(hipMemcpyAsync(d->depth_bits, g.depth_key, P * 4, hipMemcpyDeviceToDevice, st));
    if (d->rect && P) HIP_TRY(launch_unpack_rect(g.rect, in->P, d->rect, st));
    if (d->n_contrib) HIP_TRY(hipMemcpyAsync(d->n_contrib, im.n_contrib, (size_t)W * H * 4, hipMemcpyDeviceToDevice, st));
    if (d->final_T) HIP_TRY(hipMemcpyAsync(d->final_T, im.final_T, (size_t)W * H * 4, hipMemcpyDeviceToDevice, st));
    if (d->point_list || d->keys_sorted || d->ranges) {
        if (!stt->binning || stt->binning_bytes < bags_binning_size(I, W, H)) return fail(BAGS_ERR_SIZE, "binning buffer too small");
        BinView b; carve_binning(align256(stt->binning), I, W, H, &b, use_binned(s, in->P));
        const bool binned = use_binned(s, in->P);
        const uint2* ranges = binned ? im.ranges : b.ranges;
        if (d->point_list && I) HIP_TRY(hipMemcpyAsync(d->point_list, b.point_list, (size_t)I * 4, hipMemcpyDeviceToDevice, st));
        if (d->keys_sorted && I) {
            if (binned) HIP_TRY(launch_debug_keys_ranges(ranges, b.point_list, g.depth_key, (int)T, d->keys_sorted, st));
            else HIP_TRY(launch_debug_keys(b.tile_sorted, b.point_list, g.depth_key, I, d->keys_sorted, st));
        }
        if (d->ranges) {
            if (in->P == 0 || (binned && I == 0 && in->P == 0)) HIP_TRY(hipMemsetAsync(d->ranges, 0, T * 8, st));
            else HIP_TRY(hipMemcpyAsync(d->ranges, ranges, T * 8, hipMemcpyDeviceToDevice, st));
        }
    }
    return BAGS_OK;
}

int bags_profile_enable(int mode)
{
    g_prof_mode = mode < 0 ? 0 : (mode > 2 ? 2 : mode);
    return BAGS_OK;
}

int bags_profile_stride(int n)
{
    g_prof_stride = n < 1 ? 1 : n;
    g_prof_seq.store(0, std::memory_order_relaxed);
    return BAGS_OK;
}

int bags_profile_read(int max_stages, const char** names, double* total_ms, int64_t* calls)
{
    std::lock_guard<std::mutex> lk(g_prof_mutex);
    for (const ProfInterval& iv : g_prof_pending) {
        float ms = 0.f;
        if (hipEventSynchronize(iv.b) == hipSuccess && hipEventElapsedTime(&ms, iv.a, iv.b) == hipSuccess) {
            g_prof_ms[iv.stage] += ms; g_prof_calls[iv.stage] += 1;
        }
        g_prof_free.push_back(iv.a); g_prof_free.push_back(iv.b);
    }
    g_prof_pending.clear();
    const int n = max_stages < ST_COUNT ? max_stages : (int)ST_COUNT;
    for (int i = 0; i < n; ++i) {
        if (names) names[i] = kStageNames[i];
        if (total_ms) total_ms[i] = g_prof_ms[i];
        if (calls) calls[i] = g_prof_calls[i];
    }
    for (int i = 0; i < ST_COUNT; ++i) { g_prof_ms[i] = 0.0; g_prof_calls[i] = 0; }
    return (int)ST_COUNT;
}

// ---------------------------------------------------------------------------------------------- photometric loss
size_t bags_loss_workspace_size(int32_t C, int32_t H, int32_t W)
{
    if (C <= 0 || H <= 0 || W <= 0) return 256;
    return loss_workspace_bytes(C, H, W);
}

int bags_loss_forward(const float* image, const float* gt, int32_t C, int32_t H, int32_t W, void* workspace,
                      size_t workspace_bytes, float* out_terms, void* stream)
{
    if (C <= 0 || H <= 0 || W <= 0) return fail(BAGS_ERR_ARG, "loss_forward: C, H, W must be positive (got %d, %d, %d)", C, H, W);
    if (!image || !gt || !workspace || !out_terms) return fail(BAGS_ERR_ARG, "loss_forward: null pointer");
    if (workspace_bytes < loss_workspace_bytes(C, H, W))
        return fail(BAGS_ERR_SIZE, "loss_forward: workspace %zu bytes < %zu", workspace_bytes, loss_workspace_bytes(C, H, W));
    HIP_TRY(launch_loss_fwd(image, gt, C, H, W, workspace, out_terms, (hipStream_t)stream));
    return BAGS_OK;
}

int bags_loss_backward(const float* image, const float* gt, int32_t C, int32_t H, int32_t W, const void* workspace,
                       size_t workspace_bytes, const float* grad_terms, float* grad_image, void* stream)
{
    if (C <= 0 || H <= 0 || W <= 0) return fail(BAGS_ERR_ARG, "loss_backward: C, H, W must be positive (got %d, %d, %d)", C, H, W);
    if (!image || !gt || !workspace || !grad_terms || !grad_image) return fail(BAGS_ERR_ARG, "loss_backward: null pointer");
    if (workspace_bytes < loss_workspace_bytes(C, H, W))
        return fail(BAGS_ERR_SIZE, "loss_backward: workspace %zu bytes < %zu", workspace_bytes, loss_workspace_bytes(C, H, W));
    HIP_TRY(launch_loss_bwd(image, gt, C, H, W, workspace, grad_terms, grad_image, (hipStream_t)stream));
    return BAGS_OK;
}

int bags_photometric_loss_forward(const float* image, const float* gt, int32_t C, int32_t H, int32_t W, void* workspace,
                                  size_t workspace_bytes, float lambda_dssim, float* out_loss_terms, void* stream)
{
    if (C <= 0 || H <= 0 || W <= 0) return fail(BAGS_ERR_ARG, "photometric_loss_forward: C, H, W must be positive (got %d, %d, %d)", C, H, W);
    if (!image || !gt || !workspace || !out_loss_terms) return fail(BAGS_ERR_ARG, "photometric_loss_forward: null pointer");
    if (!(lambda_dssim >= 0.f && lambda_dssim <= 1.f)) return fail(BAGS_ERR_ARG, "photometric_loss_forward: lambda_dssim %g outside [0, 1]", (double)lambda_dssim);
    if (workspace_bytes < loss_workspace_bytes(C, H, W))
        return fail(BAGS_ERR_SIZE, "photometric_loss_forward: workspace %zu bytes < %zu", workspace_bytes, loss_workspace_bytes(C, H, W));
    HIP_TRY(launch_loss_fwd(image, gt, C, H, W, workspace, out_loss_terms, (hipStream_t)stream, true, lambda_dssim));
    return BAGS_OK;
}

int bags_photometric_loss_backward(const float* image, const float* gt, int32_t C, int32_t H, int32_t W, const void* workspace,
                                   size_t workspace_bytes, float lambda_dssim, const float* grad_loss, float* grad_image, void* stream)
{
    if (C <= 0 || H <= 0 || W <= 0) return fail(BAGS_ERR_ARG, "photometric_loss_backward: C, H, W must be positive (got %d, %d, %d)", C, H, W);
    if (!image || !gt || !workspace || !grad_loss || !grad_image) return fail(BAGS_ERR_ARG, "photometric_loss_backward: null pointer");
    if (!(lambda_dssim >= 0.f && lambda_dssim <= 1.f)) return fail(BAGS_ERR_ARG, "photometric_loss_backward: lambda_dssim %g outside [0, 1]", (double)lambda_dssim);
    if (workspace_bytes < loss_workspace_bytes(C, H, W))
        return fail(BAGS_ERR_SIZE, "photometric_loss_backward: workspace %zu bytes < %zu", workspace_bytes, loss_workspace_bytes(C, H, W));
    HIP_TRY(launch_loss_bwd(image, gt, C, H, W, workspace, grad_loss, grad_image, (hipStream_t)stream, true, lambda_dssim));
    return BAGS_OK;
}

// ---------------------------------------------------------------------------------------------- camera chain
static int check_camera(const BagsCamera* c)
{
    if (!c) return fail(BAGS_ERR_ARG, "camera: null struct");
    if (!c->init_quaternion || !c->delta_quaternion || !c->init_translation || !c->delta_translation || !c->fovx || !c->fovy)
        return fail(BAGS_ERR_ARG, "camera: quaternion / translation / fov pointers must be given");
    if (!(c->zfar > c->znear) || !(c->znear > 0.f)) return fail(BAGS_ERR_ARG, "camera: need 0 < znear < zfar (got %g, %g)", c->znear, c->zfar);
    return BAGS_OK;
}

int bags_camera_forward(const BagsCamera* c, float* V, float* M, float* K, float* C, void* stream)
{
    int rc = check_camera(c);
    if (rc) return rc;
    if (!V || !M || !K || !C) return fail(BAGS_ERR_ARG, "camera_forward: null output");
    HIP_TRY(launch_camera_fwd(c->init_quaternion, c->delta_quaternion, c->init_translation, c->delta_translation, c->fovx, c->fovy,
                              c->global_rotation, c->global_translation_scale, c->znear, c->zfar, V, M, K, C, (hipStream_t)stream));
    return BAGS_OK;
}

int bags_camera_backward(const BagsCamera* c, const float* gV, const float* gM, const float* gK, const float* gC,
                         float* g_dq, float* g_dt, float* g_fovx, float* g_fovy, float* g_grot, float* g_gscale, void* stream)
{
    int rc = check_camera(c);
    if (rc) return rc;
    HIP_TRY(launch_camera_bwd(c->init_quaternion, c->delta_quaternion, c->init_translation, c->delta_translation, c->fovx, c->fovy,
                              c->global_rotation, c->global_translation_scale, c->znear, c->zfar, gV, gM, gK, gC,
                              g_dq, g_dt, g_fovx, g_fovy, g_grot, g_gscale, (hipStream_t)stream));
    return BAGS_OK;
}

// ---------------------------------------------------------------------------------------------- distortion resampling
static int check_resample(int C, int H, int W, int h, int w, int Hf, int Wf, int Hc, int Wc)
{
    if (C <= 0 || H <= 0 || W <= 0 || h <= 0 || w <= 0 || Hf <= 0 || Wf <= 0 || Hc <= 0 || Wc <= 0)
        return fail(BAGS_ERR_ARG, "resample: every extent must be positive");
    if (Hc > Hf || Wc > Wf) return fail(BAGS_ERR_ARG, "resample: crop %dx%d exceeds the flow size %dx%d", Wc, Hc, Wf, Hf);
    // the backward keeps a 16x16xC accumulator of 64-bit words in LDS (2 KB per channel)
    if (C > 24) return fail(BAGS_ERR_ARG, "resample: at most 24 channels per call (got %d): split the image along C", C);
    if (H > 32000 || W > 32000) return fail(BAGS_ERR_ARG, "resample: image %dx%d too large (tap coordinates are kept as int16)", W, H);
    return BAGS_OK;
}

int bags_resample_forward(const float* image, int32_t C, int32_t H, int32_t W, const float* ctrl, int32_t h, int32_t w,
                          int32_t Hf, int32_t Wf, int32_t Hc, int32_t Wc, float* out, float* mask, float* flow_out, void* stream)
{
    int rc = check_resample(C, H, W, h, w, Hf, Wf, Hc, Wc);
    if (rc) return rc;
    if (!image || !ctrl || !out) return fail(BAGS_ERR_ARG, "resample_forward: null pointer");
    HIP_TRY(launch_resample_fwd(image, C, H, W, ctrl, h, w, Hf, Wf, Hc, Wc, out, mask, flow_out, (hipStream_t)stream));
    return BAGS_OK;
}

size_t bags_resample_workspace_size(int32_t H, int32_t W, int32_t Hc, int32_t Wc)
{
    return resample_workspace_bytes(H > 0 ? H : 1, W > 0 ? W : 1, Hc > 0 ? Hc : 1, Wc > 0 ? Wc : 1);
}

int bags_resample_backward(const float* image, int32_t C, int32_t H, int32_t W, const float* ctrl, int32_t h, int32_t w,
                           int32_t Hf, int32_t Wf, int32_t Hc, int32_t Wc, const float* grad_out, void* workspace,
                           size_t workspace_bytes, float* grad_image, float* grad_ctrl, void* stream)
{
    int rc = check_resample(C, H, W, h, w, Hf, Wf, Hc, Wc);
    if (rc) return rc;
    if (!image || !ctrl || !grad_out) return fail(BAGS_ERR_ARG, "resample_backward: null pointer");
    if (!grad_image && !grad_ctrl) return BAGS_OK;
    if (!workspace || workspace_bytes < resample_workspace_bytes(H, W, Hc, Wc))
        return fail(BAGS_ERR_SIZE, "resample_backward: needs a workspace of %zu bytes", resample_workspace_bytes(H, W, Hc, Wc));
    HIP_TRY(launch_resample_bwd(image, C, H, W, ctrl, h, w, Hf, Wf, Hc, Wc, grad_out, workspace, grad_image, grad_ctrl, (hipStream_t)stream));
    return BAGS_OK;
}

// ---------------------------------------------------------------------------------------------- activations
static int check_raw(const BagsRawGaussians* r, bool features)      // features: the SH concatenation (or its gradient) is asked for
{
    if (!r) return fail(BAGS_ERR_ARG, "activations: null struct");
    if (r->P < 0 || r->K < 1) return fail(BAGS_ERR_ARG, "activations: need P >= 0 and K >= 1 (got %d, %d)", r->P, r->K);
    if (r->P > 0 && (!r->opacity || !r->scaling || !r->rotation)) return fail(BAGS_ERR_ARG, "activations: null parameter pointer");
    if (r->P > 0 && features && (!r->features_dc || (r->K > 1 && !r->features_rest)))
        return fail(BAGS_ERR_ARG, "activations: null feature pointer (features_dc / features_rest may only be NULL when shs / g_shs is)");
    return BAGS_OK;
}

int bags_activations_forward(const BagsRawGaussians* r, float* shs, float* opacity, float* scales, float* rotations, void* stream)
{
    int rc = check_raw(r, shs != nullptr);
    if (rc) return rc;
    HIP_TRY(launch_activations_fwd(r->P, r->K, r->features_dc, r->features_rest, r->opacity, r->scaling, r->rotation, shs, opacity,
                                   scales, rotations, (hipStream_t)stream));
    return BAGS_OK;
}

int bags_activations_backward(const BagsRawGaussians* r, const float* g_shs, const float* g_opacity, const float* g_scales,
                              const float* g_rotations, float* g_dc, float* g_rest, float* g_opacity_raw, float* g_scaling,
                              float* g_rotation, void* stream)
{
    int rc = check_raw(r, g_shs != nullptr && (g_dc != nullptr || g_rest != nullptr));
    if (rc) return rc;
    HIP_TRY(launch_activations_bwd(r->P, r->K, r->features_dc, r->features_rest, r->opacity, r->scaling, r->rotation, g_shs, g_opacity,
                                   g_scales, g_rotations, g_dc, g_rest, g_opacity_raw, g_scaling, g_rotation, (hipStream_t)stream));
    return BAGS_OK;
}

// ---------------------------------------------------------------------------------------------- kNN scale initialiser
size_t bags_knn_workspace_size(int32_t P) { return knn_workspace_bytes(P > 0 ? P : 1); }

int bags_knn_mean_dist2(const float* points, int32_t P, void* workspace, size_t workspace_bytes, float* out, void* stream)
{
    if (P < 0) return fail(BAGS_ERR_ARG, "knn: P < 0");
    if (P == 0) return BAGS_OK;
    if (!points || !workspace || !out) return fail(BAGS_ERR_ARG, "knn: null pointer");
    if (workspace_bytes < knn_workspace_bytes(P))
        return fail(BAGS_ERR_SIZE, "knn: workspace %zu bytes < %zu", workspace_bytes, knn_workspace_bytes(P));
    HIP_TRY(launch_knn(points, P, align256(workspace), out, (hipStream_t)stream));
    return BAGS_OK;
}

int bags_compute_relocation(const float*, const float*, const int32_t*, const float*, int32_t, int32_t, float*, float*, void*)
{
    return fail(BAGS_ERR_ARG, "compute_relocation: the reference's only caller is commented out (scene/gaussian_model.py:23,494-504); not implemented");
}

}  // extern "C"
