// knn.hip -- distCUDA2: per point, the mean of the squared distances to its three nearest neighbours
// (the reference's second native dependency, simple_knn._C.distCUDA2, imported at
// scene/gaussian_model.py:20 and called once per scene at scene/gaussian_model.py:177 to initialise the scales:
// scales = log(sqrt(clamp_min(distCUDA2(points), 1e-7))).  The submodule's source is absent from the reference tree;
// the contract restated here is the published one: exact 3-NN, self excluded by index, duplicates count with distance
// 0, missing neighbours (P < 4) contribute FLT_MAX.)
//
// MI355X design: a uniform grid instead of simple-knn's Morton-sorted boxes.  Everything is sized and decided on the
// device (no host round trip):
//   bbox -> grid of about P/4 cells (<= 1024 per axis, <= P cells) -> cell histogram -> exclusive scan -> counting sort
//   of the points by cell -> one thread per point walks cubic shells of cells around its own cell; a row of cells
//   along x is one contiguous run of the sorted array.  After shell r every unvisited point is farther than r cell
//   widths, so the walk stops as soon as the third-best squared distance is <= (r * cell)^2: the result is exact.
#include "bags_common.h"
#include <float.h>

struct KnnGrid {
    float ox, oy, oz;          // grid origin (bbox min)
    float inv_cell, cell;
    int nx, ny, nz, ncells;
};

__device__ __forceinline__ u32 f2ord(float f) { const u32 u = __float_as_uint(f); return (u & 0x80000000u) ? ~u : (u | 0x80000000u); }
__device__ __forceinline__ float ord2f(u32 o) { return __uint_as_float((o & 0x80000000u) ? (o & 0x7FFFFFFFu) : ~o); }

__global__ void __launch_bounds__(256)
knn_bbox_kernel(const float* __restrict__ pts, int P, u32* __restrict__ box /* min xyz, max xyz (order-preserving u32) */)
{
    float lo[3] = {FLT_MAX, FLT_MAX, FLT_MAX}, hi[3] = {-FLT_MAX, -FLT_MAX, -FLT_MAX};
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < P; i += gridDim.x * blockDim.x) {
#pragma unroll
        for (int a = 0; a < 3; ++a) { const float v = pts[3 * (size_t)i + a]; lo[a] = fminf(lo[a], v); hi[a] = fmaxf(hi[a], v); }
    }
#pragma unroll
    for (int a = 0; a < 3; ++a) {
#pragma unroll
        for (int d = 32; d >= 1; d >>= 1) { lo[a] = fminf(lo[a], __shfl_xor(lo[a], d)); hi[a] = fmaxf(hi[a], __shfl_xor(hi[a], d)); }
    }
    if ((threadIdx.x & 63) == 0) {
#pragma unroll
        for (int a = 0; a < 3; ++a) { atomicMin(&box[a], f2ord(lo[a])); atomicMax(&box[3 + a], f2ord(hi[a])); }
    }
}

__global__ void knn_setup_kernel(const u32* __restrict__ box, int P, KnnGrid* __restrict__ grid)
{
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    KnnGrid g;
    const float lx = ord2f(box[0]), ly = ord2f(box[1]), lz = ord2f(box[2]);
    const float ex = fmaxf(ord2f(box[3]) - lx, 0.f), ey = fmaxf(ord2f(box[4]) - ly, 0.f), ez = fmaxf(ord2f(box[5]) - lz, 0.f);
    const float emax = fmaxf(ex, fmaxf(ey, ez));
    const float target = fmaxf(1.f, 0.25f * (float)P);                 // ~4 points per cell
    // cell edge from the occupied volume; flat / degenerate clouds fall back to the longest extent
    float cell = cbrtf(fmaxf(ex, 1e-30f) * fmaxf(ey, 1e-30f) * fmaxf(ez, 1e-30f) / target);
    cell = fmaxf(cell, emax / 1024.f);
    if (!(cell > 0.f) || !isfinite(cell)) cell = 1.f;
    const long long cap = (long long)(P > 0 ? P : 1);
    for (int it = 0; it < 64; ++it) {
        g.nx = min(1024, (int)(ex / cell) + 1); g.ny = min(1024, (int)(ey / cell) + 1); g.nz = min(1024, (int)(ez / cell) + 1);
        if ((long long)g.nx * g.ny * g.nz <= cap) break;
        cell *= 1.26f;                                                  // ~2x fewer cells per step
    }
    if ((long long)g.nx * g.ny * g.nz > cap) { g.nx = g.ny = g.nz = 1; cell = fmaxf(emax, 1.f) * 2.f; }
    g.ox = lx; g.oy = ly; g.oz = lz; g.cell = cell; g.inv_cell = 1.0f / cell; g.ncells = g.nx * g.ny * g.nz;
    *grid = g;
}

__device__ __forceinline__ int3 cell_of(const KnnGrid& g, float x, float y, float z)
{
    int3 c;
    c.x = min(g.nx - 1, max(0, (int)((x - g.ox) * g.inv_cell)));
    c.y = min(g.ny - 1, max(0, (int)((y - g.oy) * g.inv_cell)));
    c.z = min(g.nz - 1, max(0, (int)((z - g.oz) * g.inv_cell)));
    return c;
}

__global__ void __launch_bounds__(256)
knn_count_kernel(const float* __restrict__ pts, int P, const KnnGrid* __restrict__ grid, u32* __restrict__ counts, u32* __restrict__ cell_id)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= P) return;
    const KnnGrid g = *grid;
    const int3 c = cell_of(g, pts[3 * (size_t)i], pts[3 * (size_t)i + 1], pts[3 * (size_t)i + 2]);
    const u32 id = (u32)((c.z * g.ny + c.y) * g.nx + c.x);
    cell_id[i] = id;
    atomicAdd(&counts[id], 1u);
}

// exclusive scan of n u32 values, three small kernels (block sums, scan of the sums, rescan)
#define KSCAN_BLOCK 256
#define KSCAN_ITEMS 8
#define KSCAN_TILE (KSCAN_BLOCK * KSCAN_ITEMS)
__global__ void __launch_bounds__(KSCAN_BLOCK)
knn_scan_partial_kernel(const u32* __restrict__ v, int n, u32* __restrict__ partials)
{
    __shared__ u32 ws[KSCAN_BLOCK / 64];
    const int base = blockIdx.x * KSCAN_TILE + threadIdx.x * KSCAN_ITEMS;
    u32 s = 0;
#pragma unroll
    for (int r = 0; r < KSCAN_ITEMS; ++r) if (base + r < n) s += v[base + r];
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) s += __shfl_xor(s, d);
    if ((threadIdx.x & 63) == 0) ws[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) partials[blockIdx.x] = ws[0] + ws[1] + ws[2] + ws[3];
}
__global__ void __launch_bounds__(256)
knn_scan_top_kernel(u32* __restrict__ partials, int nparts)
{
    __shared__ u32 wsum[4];
    __shared__ u32 carry_s;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (threadIdx.x == 0) carry_s = 0;
    __syncthreads();
    for (int base = 0; base < nparts; base += 256) {
        const int i = base + threadIdx.x;
        const u32 v = (i < nparts) ? partials[i] : 0u;
        u32 incl = v;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) { const u32 t = __shfl_up(incl, d); if (lane >= d) incl += t; }
        if (lane == 63) wsum[wave] = incl;
        __syncthreads();
        u32 wbase = 0;
        for (int w = 0; w < wave; ++w) wbase += wsum[w];
        const u32 carry = carry_s;
        if (i < nparts) partials[i] = carry + wbase + incl - v;
        __syncthreads();
        if (threadIdx.x == 255) carry_s = carry + wbase + incl;
        __syncthreads();
    }
}
__global__ void __launch_bounds__(KSCAN_BLOCK)
knn_scan_final_kernel(const u32* __restrict__ v, int n, const u32* __restrict__ partials, u32* __restrict__ out /* n + 1 */)
{
    __shared__ u32 wsum[KSCAN_BLOCK / 64];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int base = blockIdx.x * KSCAN_TILE + threadIdx.x * KSCAN_ITEMS;
    u32 t[KSCAN_ITEMS], s = 0;
#pragma unroll
    for (int r = 0; r < KSCAN_ITEMS; ++r) { t[r] = (base + r < n) ? v[base + r] : 0u; s += t[r]; }
    u32 incl = s;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) { const u32 u = __shfl_up(incl, d); if (lane >= d) incl += u; }
    if (lane == 63) wsum[wave] = incl;
    __syncthreads();
    u32 run = partials[blockIdx.x] + incl - s;
    for (int w = 0; w < wave; ++w) run += wsum[w];
#pragma unroll
    for (int r = 0; r < KSCAN_ITEMS; ++r) {
        if (base + r < n) out[base + r] = run;
        run += t[r];
        if (base + r == n - 1) out[n] = run;
    }
}

__global__ void __launch_bounds__(256)
knn_scatter_kernel(const float* __restrict__ pts, int P, const u32* __restrict__ cell_id, const u32* __restrict__ cell_start,
                   u32* __restrict__ fill, float4* __restrict__ sorted)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= P) return;
    const u32 c = cell_id[i];
    const u32 pos = cell_start[c] + atomicAdd(&fill[c], 1u);
    sorted[pos] = make_float4(pts[3 * (size_t)i], pts[3 * (size_t)i + 1], pts[3 * (size_t)i + 2], __uint_as_float((u32)i));
}

__device__ __forceinline__ void knn_push(float d, float& b0, float& b1, float& b2)
{
    if (d < b2) {
        if (d < b1) { b2 = b1; if (d < b0) { b1 = b0; b0 = d; } else b1 = d; }
        else b2 = d;
    }
}

__global__ void __launch_bounds__(256)
knn_search_kernel(const float4* __restrict__ sorted, int P, const KnnGrid* __restrict__ grid, const u32* __restrict__ cell_start,
                  float* __restrict__ out)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= P) return;
    const KnnGrid g = *grid;
    const float4 p = sorted[i];
    const int3 c = cell_of(g, p.x, p.y, p.z);
    float b0 = FLT_MAX, b1 = FLT_MAX, b2 = FLT_MAX;
    const int rmax = max(g.nx, max(g.ny, g.nz));
    for (int r = 0; r <= rmax; ++r) {
        const int z0 = max(0, c.z - r), z1 = min(g.nz - 1, c.z + r);
        const int y0 = max(0, c.y - r), y1 = min(g.ny - 1, c.y + r);
        const int x0 = max(0, c.x - r), x1 = min(g.nx - 1, c.x + r);
        for (int z = z0; z <= z1; ++z) {
            const bool zshell = (z == c.z - r) || (z == c.z + r);
            for (int y = y0; y <= y1; ++y) {
                const bool full = zshell || (y == c.y - r) || (y == c.y + r);     // whole x run belongs to shell r
                const u32 row = (u32)((z * g.ny + y) * g.nx);
                // shell cells of this row: all of [x0, x1], or only its two ends (each end only if it is really at +-r)
                for (int part = 0; part < (full ? 1 : 2); ++part) {
                    int xa, xb;
                    if (full) { xa = x0; xb = x1; }
                    else if (part == 0) { if (c.x - r < 0) continue; xa = xb = c.x - r; }
                    else { if (c.x + r > g.nx - 1 || r == 0) continue; xa = xb = c.x + r; }
                    const u32 s = cell_start[row + xa], e = cell_start[row + xb + 1];
                    for (u32 j = s; j < e; ++j) {
                        if ((int)j == i) continue;
                        const float4 q = sorted[j];
                        const float dx = p.x - q.x, dy = p.y - q.y, dz = p.z - q.z;
                        knn_push(dx * dx + dy * dy + dz * dz, b0, b1, b2);
                    }
                }
            }
        }
        const float reach = (float)r * g.cell;               // every point not visited yet is farther than this
        if (b2 <= reach * reach) break;
        if (x0 == 0 && y0 == 0 && z0 == 0 && x1 == g.nx - 1 && y1 == g.ny - 1 && z1 == g.nz - 1) break;   // whole grid seen
    }
    out[__float_as_uint(p.w)] = (b0 + b1 + b2) / 3.0f;
}

// workspace: [box 8 u32 | grid | counts (P+1) | cell_start (P+2) | fill (P+1) | cell_id P | partials | sorted P float4]
static size_t knn_carve(void* base, int P, u32** box, KnnGrid** grid, u32** counts, u32** start, u32** fill, u32** cid,
                        u32** partials, float4** sorted)
{
    char* p = reinterpret_cast<char*>(base);
    const size_t n = (size_t)(P > 0 ? P : 1);
    auto take = [&](size_t bytes) { char* q = p; p += align_up(bytes, 256); return q; };
    u32* b = reinterpret_cast<u32*>(take(8 * sizeof(u32)));
    KnnGrid* g = reinterpret_cast<KnnGrid*>(take(sizeof(KnnGrid)));
    u32* c = reinterpret_cast<u32*>(take((n + 1) * sizeof(u32)));
    u32* f = reinterpret_cast<u32*>(take((n + 1) * sizeof(u32)));          // counts and fill are adjacent: one memset
    u32* s = reinterpret_cast<u32*>(take((n + 2) * sizeof(u32)));
    u32* ci = reinterpret_cast<u32*>(take(n * sizeof(u32)));
    u32* pa = reinterpret_cast<u32*>(take(((n + 1) / KSCAN_TILE + 2) * sizeof(u32)));
    float4* so = reinterpret_cast<float4*>(take(n * sizeof(float4)));
    if (box) { *box = b; *grid = g; *counts = c; *fill = f; *start = s; *cid = ci; *partials = pa; *sorted = so; }
    return (size_t)(p - reinterpret_cast<char*>(base));
}

size_t knn_workspace_bytes(int P) { return knn_carve(nullptr, P, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr) + 256; }

hipError_t launch_knn(const float* pts, int P, void* ws, float* out, hipStream_t st)
{
    if (P <= 0) return hipSuccess;
    u32 *box, *counts, *start, *fill, *cid, *partials; KnnGrid* grid; float4* sorted;
    knn_carve(ws, P, &box, &grid, &counts, &start, &fill, &cid, &partials, &sorted);
    hipError_t e;
    if ((e = hipMemsetAsync(box, 0xFF, 3 * sizeof(u32), st)) != hipSuccess) return e;               // min slots
    if ((e = hipMemsetAsync(box + 3, 0x00, 3 * sizeof(u32), st)) != hipSuccess) return e;           // max slots
    const size_t zero_bytes = (size_t)(reinterpret_cast<char*>(start) - reinterpret_cast<char*>(counts));
    if ((e = hipMemsetAsync(counts, 0, zero_bytes, st)) != hipSuccess) return e;                     // counts + fill
    const int nb = cdiv(P, 256);
    hipLaunchKernelGGL(knn_bbox_kernel, dim3(nb < 1024 ? nb : 1024), dim3(256), 0, st, pts, P, box);
    hipLaunchKernelGGL(knn_setup_kernel, dim3(1), dim3(64), 0, st, (const u32*)box, P, grid);
    hipLaunchKernelGGL(knn_count_kernel, dim3(nb), dim3(256), 0, st, pts, P, (const KnnGrid*)grid, counts, cid);
    const int n = P + 1;                                                 // scan over the cell capacity (>= ncells)
    const int nparts = cdiv(n, KSCAN_TILE);
    hipLaunchKernelGGL(knn_scan_partial_kernel, dim3(nparts), dim3(KSCAN_BLOCK), 0, st, (const u32*)counts, n, partials);
    hipLaunchKernelGGL(knn_scan_top_kernel, dim3(1), dim3(256), 0, st, partials, nparts);
    hipLaunchKernelGGL(knn_scan_final_kernel, dim3(nparts), dim3(KSCAN_BLOCK), 0, st, (const u32*)counts, n, (const u32*)partials, start);
    hipLaunchKernelGGL(knn_scatter_kernel, dim3(nb), dim3(256), 0, st, pts, P, (const u32*)cid, (const u32*)start, fill, sorted);
    hipLaunchKernelGGL(knn_search_kernel, dim3(nb), dim3(256), 0, st, (const float4*)sorted, P, (const KnnGrid*)grid, (const u32*)start, out);
    return hipGetLastError();
}
