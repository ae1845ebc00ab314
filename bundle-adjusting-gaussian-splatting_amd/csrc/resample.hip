// resample.hip -- image-space distortion resampling of the rendered image.
//
// The reference warps the rendered image with a dense flow that a lens network predicts on a coarse control grid
// (utils/util_distortion.py:271-311 apply_distortion, apply2gt == False branch; call site train.py:255-263):
//     flow = F.interpolate(control flow (h,w,2) -> (Hf,Wf), bilinear, align_corners=False)
//     img  = F.grid_sample(image (C,H,W), flow, bilinear, zeros padding, align_corners=True)      -> (C,Hf,Wf)
//     img  = center_crop(img, Hc, Wc)          (utils/util_distortion.py:58-77: a second grid_sample on an integer grid)
//     mask = ~((img[0] == 0) & (img[1] == 0))
// i.e. three full-resolution passes and their autograd backward.  Here one kernel each way touches only the cropped
// pixels: the control flow is interpolated at the pixel, the image is sampled once; the backward accumulates dL/dimage per
// 16x16 source tile in LDS as 64-bit fixed-point integers (no global float atomics, bitwise reproducible) and GATHERS
// dL/d(control flow) per control node.
// The crop is taken as exact integer indexing (the reference's second grid_sample reproduces integer positions only to
// ~1e-4 px after normalising and un-normalising the grid; the difference is below 2e-4 of the image range).
#include "bags_common.h"

struct ResampleGeom {
    int C, H, W;          // image
    int h, w;             // control grid of the flow
    int Hf, Wf;           // size the flow is upsampled to (= size of the warped image before the crop)
    int Hc, Wc;           // centre crop
    int y0, x0;           // crop origin inside (Hf, Wf)
};

struct FlowTap { int i00, i01, i10, i11; float w00, w01, w10, w11; };

// bilinear upsampling source taps of F.interpolate(align_corners=False) at destination (Y, X)
__device__ __forceinline__ FlowTap flow_taps(const ResampleGeom& g, int Y, int X)
{
    const float sy = fmaxf(((float)Y + 0.5f) * ((float)g.h / (float)g.Hf) - 0.5f, 0.f);
    const float sx = fmaxf(((float)X + 0.5f) * ((float)g.w / (float)g.Wf) - 0.5f, 0.f);
    const int y0 = min((int)sy, g.h - 1), x0 = min((int)sx, g.w - 1);
    const int y1 = y0 + (y0 < g.h - 1), x1 = x0 + (x0 < g.w - 1);
    const float ly = sy - (float)y0, lx = sx - (float)x0;
    FlowTap t;
    t.i00 = y0 * g.w + x0; t.i01 = y0 * g.w + x1; t.i10 = y1 * g.w + x0; t.i11 = y1 * g.w + x1;
    t.w00 = (1.f - ly) * (1.f - lx); t.w01 = (1.f - ly) * lx; t.w10 = ly * (1.f - lx); t.w11 = ly * lx;
    return t;
}

struct ImgTap { int x0, y0; float fx, fy; bool in00, in01, in10, in11; };

// grid_sample(align_corners=True, zeros padding) taps for the normalised coordinate (gx, gy)
__device__ __forceinline__ ImgTap img_taps(const ResampleGeom& g, float gx, float gy)
{
    const float ix = (gx + 1.f) * 0.5f * (float)(g.W - 1), iy = (gy + 1.f) * 0.5f * (float)(g.H - 1);
    const float fx0 = floorf(ix), fy0 = floorf(iy);
    ImgTap t;
    // far-away coordinates (|ix| beyond int range) are simply out of bounds
    t.x0 = (fx0 > -2.f && fx0 < (float)g.W + 1.f) ? (int)fx0 : -2;
    t.y0 = (fy0 > -2.f && fy0 < (float)g.H + 1.f) ? (int)fy0 : -2;
    t.fx = ix - fx0; t.fy = iy - fy0;
    const bool xa = t.x0 >= 0 && t.x0 < g.W, xb = t.x0 + 1 >= 0 && t.x0 + 1 < g.W;
    const bool ya = t.y0 >= 0 && t.y0 < g.H, yb = t.y0 + 1 >= 0 && t.y0 + 1 < g.H;
    t.in00 = xa && ya; t.in01 = xb && ya; t.in10 = xa && yb; t.in11 = xb && yb;
    if (!isfinite(ix) || !isfinite(iy)) { t.in00 = t.in01 = t.in10 = t.in11 = false; t.fx = t.fy = 0.f; }
    return t;
}

__global__ void __launch_bounds__(256)
resample_fwd_kernel(ResampleGeom g, const float* __restrict__ image, const float* __restrict__ ctrl,
                    float* __restrict__ out, float* __restrict__ mask, float* __restrict__ flow_out)
{
    const int xc = blockIdx.x * 64 + (threadIdx.x & 63), yc = blockIdx.y * 4 + (threadIdx.x >> 6);
    if (xc >= g.Wc || yc >= g.Hc) return;
    const FlowTap f = flow_taps(g, g.y0 + yc, g.x0 + xc);
    const float2* c2 = reinterpret_cast<const float2*>(ctrl);
    const float2 a = c2[f.i00], b = c2[f.i01], c = c2[f.i10], d = c2[f.i11];
    const float gx = f.w00 * a.x + f.w01 * b.x + f.w10 * c.x + f.w11 * d.x;
    const float gy = f.w00 * a.y + f.w01 * b.y + f.w10 * c.y + f.w11 * d.y;
    if (flow_out) reinterpret_cast<float2*>(flow_out)[(size_t)yc * g.Wc + xc] = make_float2(gx, gy);
    const ImgTap t = img_taps(g, gx, gy);
    const float w00 = (1.f - t.fx) * (1.f - t.fy), w01 = t.fx * (1.f - t.fy), w10 = (1.f - t.fx) * t.fy, w11 = t.fx * t.fy;
    const size_t plane = (size_t)g.H * g.W, oplane = (size_t)g.Hc * g.Wc;
    const size_t base = (size_t)t.y0 * g.W + t.x0;
    float v01 = 0.f, first = 0.f, second = 0.f;
    for (int ch = 0; ch < g.C; ++ch) {
        const float* im = image + ch * plane;
        float v = 0.f;
        if (t.in00) v += w00 * im[base];
        if (t.in01) v += w01 * im[base + 1];
        if (t.in10) v += w10 * im[base + g.W];
        if (t.in11) v += w11 * im[base + g.W + 1];
        out[ch * oplane + (size_t)yc * g.Wc + xc] = v;
        if (ch == 0) first = v;
        if (ch == 1) second = v;
        v01 = v;
    }
    (void)v01;
    if (mask) mask[(size_t)yc * g.Wc + xc] = (first == 0.f && (g.C < 2 || second == 0.f)) ? 0.f : 1.f;
}

// ---------------------------------------------------------------------------------------------------- backward
// dL/dimage without global float atomics and bitwise reproducible.  (The first version scattered 4 x C float atomics per output
// pixel: 0.6 ms at 1080p, 17x the forward, the one place of the library that used the ~1.3 TB/s atomic path, and the
// summation order changed from run to run.)  Two kernels:
//   1. resample_bwd_pixels_kernel, one workgroup per 16x16 OUTPUT tile: dL/dflow of every pixel (for the control-flow
//      gather), the bounding box of the tile's in-bounds taps in source pixels, max |dL/dout| of the tile, and the tile's
//      id appended to the list of every 16x16 SOURCE tile the box overlaps (integer atomics on a counter: which slot a
//      tile gets is irrelevant, see 2.).  The flow is smooth, so a box overlaps about four source tiles.
//   2. resample_gather_kernel, one workgroup per SOURCE tile: for every output tile on its list, thread = output pixel
//      reads its taps (left behind by kernel 1) and adds those that land in this source tile into a 16x16xC accumulator in LDS -- as 64-bit
//      FIXED-POINT integers (LDS integer atomics): integer addition is associative, so the sum does not depend on the
//      order in which lanes, waves or list entries arrive.  The quantum is a power of two derived from the largest
//      |dL/dout| on the list and the list length (both order-independent): 2^-47 of that maximum for lists of up to 32
//      tiles, i.e. far below fp32 resolution of the result.  Every source tile is written (zeros where nothing lands): no
//      memset of dL/dimage.  A source tile whose list overflows RS_CAP (extreme minification: dozens of output tiles
//      sampling one source tile) ignores the list and tests the boxes of all output tiles instead: slow, exact.
#define RS_TILE 16
#define RS_CAP 32
struct RsWork { float2* gflow; int4* bbox; float* tmax; u32* count; u32* list; float4* taps; };
// What the pixels kernel leaves for the gather per OUTPUT pixel: the four bilinear taps of its sample, (x0 | y0 << 16 as int16s,
// fx, fy, 0).  The gather used to re-derive them per listed tile (upsampling taps of the control flow -> four control-node loads
// -> interpolation -> image taps: two dependent memory round trips per list entry, in a serial loop: 105 us, latency bound --
// a build without its LDS atomics was no faster); now it is one coalesced 16-byte load, requested four list entries at a time.

__global__ void __launch_bounds__(256)
resample_bwd_pixels_kernel(ResampleGeom g, const float* __restrict__ image, const float* __restrict__ ctrl, const float* __restrict__ grad_out,
                           float2* __restrict__ gflow /* (Hc,Wc) or NULL */, int4* __restrict__ bbox, float* __restrict__ tmax,
                           u32* __restrict__ count, u32* __restrict__ list, float4* __restrict__ taps, int src_tiles_x, int src_tiles_y)
{
    __shared__ int s_box[4][4];
    __shared__ float s_max[4];
    const int otx = blockIdx.x, oty = blockIdx.y, ot = oty * gridDim.x + otx;
    const int xc = otx * RS_TILE + (threadIdx.x & 15), yc = oty * RS_TILE + (threadIdx.x >> 4);
    int bx0 = 0x7FFFFFFF, by0 = 0x7FFFFFFF, bx1 = -1, by1 = -1;
    float amax = 0.f;
    if (xc < g.Wc && yc < g.Hc) {
        const FlowTap f = flow_taps(g, g.y0 + yc, g.x0 + xc);
        const float2* c2 = reinterpret_cast<const float2*>(ctrl);
        const float2 a = c2[f.i00], b = c2[f.i01], c = c2[f.i10], d = c2[f.i11];
        const float gx = f.w00 * a.x + f.w01 * b.x + f.w10 * c.x + f.w11 * d.x;
        const float gy = f.w00 * a.y + f.w01 * b.y + f.w10 * c.y + f.w11 * d.y;
        const ImgTap t = img_taps(g, gx, gy);
        if (bbox) {                                          // dL/dimage requested: the gather reads this instead of recomputing it
            const u32 xy = ((u32)t.x0 & 0xFFFFu) | ((u32)t.y0 << 16);          // x0, y0 in [-2, 65535): int16 pairs
            taps[(size_t)yc * g.Wc + xc] = make_float4(__uint_as_float(xy), (t.in00 | t.in01 | t.in10 | t.in11) ? t.fx : 0.f, t.fy, 0.f);
        }
        const size_t plane = (size_t)g.H * g.W, oplane = (size_t)g.Hc * g.Wc;
        const size_t base = (size_t)t.y0 * g.W + t.x0;
        float dix = 0.f, diy = 0.f;
        for (int ch = 0; ch < g.C; ++ch) {
            const float go = grad_out[ch * oplane + (size_t)yc * g.Wc + xc];
            amax = fmaxf(amax, fabsf(go));
            if (gflow) {
                const float* im = image + ch * plane;
                const float v00 = t.in00 ? im[base] : 0.f, v01 = t.in01 ? im[base + 1] : 0.f;
                const float v10 = t.in10 ? im[base + g.W] : 0.f, v11 = t.in11 ? im[base + g.W + 1] : 0.f;
                dix += go * ((v01 - v00) * (1.f - t.fy) + (v11 - v10) * t.fy);
                diy += go * ((v10 - v00) * (1.f - t.fx) + (v11 - v01) * t.fx);
            }
        }
        if (gflow) gflow[(size_t)yc * g.Wc + xc] = make_float2(dix * 0.5f * (float)(g.W - 1), diy * 0.5f * (float)(g.H - 1));
        if (t.in00 || t.in01 || t.in10 || t.in11) {
            bx0 = (t.in00 || t.in10) ? t.x0 : t.x0 + 1; bx1 = (t.in01 || t.in11) ? t.x0 + 1 : t.x0;
            by0 = (t.in00 || t.in01) ? t.y0 : t.y0 + 1; by1 = (t.in10 || t.in11) ? t.y0 + 1 : t.y0;
        }
    }
    if (!bbox) return;                                       // dL/dimage not requested
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) {
        bx0 = min(bx0, __shfl_xor(bx0, d)); by0 = min(by0, __shfl_xor(by0, d));
        bx1 = max(bx1, __shfl_xor(bx1, d)); by1 = max(by1, __shfl_xor(by1, d));
        amax = fmaxf(amax, __shfl_xor(amax, d));
    }
    const int wave = threadIdx.x >> 6;
    if ((threadIdx.x & 63) == 0) { s_box[wave][0] = bx0; s_box[wave][1] = by0; s_box[wave][2] = bx1; s_box[wave][3] = by1; s_max[wave] = amax; }
    __syncthreads();
    bx0 = min(min(s_box[0][0], s_box[1][0]), min(s_box[2][0], s_box[3][0]));
    by0 = min(min(s_box[0][1], s_box[1][1]), min(s_box[2][1], s_box[3][1]));
    bx1 = max(max(s_box[0][2], s_box[1][2]), max(s_box[2][2], s_box[3][2]));
    by1 = max(max(s_box[0][3], s_box[1][3]), max(s_box[2][3], s_box[3][3]));
    amax = fmaxf(fmaxf(s_max[0], s_max[1]), fmaxf(s_max[2], s_max[3]));
    if (threadIdx.x == 0) { bbox[ot] = make_int4(bx0, by0, bx1, by1); tmax[ot] = amax; }
    if (bx1 < bx0 || by1 < by0 || !(amax > 0.f)) return;      // nothing lands in the image, or a zero cotangent: on no list
    const int sx0 = bx0 / RS_TILE, sx1 = bx1 / RS_TILE, sy0 = by0 / RS_TILE, sy1 = by1 / RS_TILE;
    const int nsx = sx1 - sx0 + 1, n = nsx * (sy1 - sy0 + 1);
    for (int k = threadIdx.x; k < n; k += 256) {
        const int st = (sy0 + k / nsx) * src_tiles_x + sx0 + k % nsx;
        const u32 slot = atomicAdd(&count[st], 1u);
        if (slot < RS_CAP) list[(size_t)st * RS_CAP + slot] = (u32)ot;
    }
}

__global__ void __launch_bounds__(256)
resample_gather_kernel(ResampleGeom g, const float* __restrict__ ctrl, const float* __restrict__ grad_out, const int4* __restrict__ bbox,
                       const float* __restrict__ tmax, const u32* __restrict__ count, const u32* __restrict__ list,
                       const float4* __restrict__ taps, float* __restrict__ grad_image, int out_tiles_x, int out_tiles)
{
    extern __shared__ unsigned long long acc[];              // [C][256] fixed-point sums
    __shared__ u32 s_list[RS_CAP];
    __shared__ float s_red[4];
    const int stx = blockIdx.x, sty = blockIdx.y, st = sty * gridDim.x + stx;
    const int X0 = stx * RS_TILE, Y0 = sty * RS_TILE;
    const u32 n_all = count[st];
    const bool listed = n_all <= RS_CAP;
    const u32 n = listed ? n_all : (u32)out_tiles;
    for (int i = threadIdx.x; i < g.C * 256; i += 256) acc[i] = 0ull;
    if (listed && threadIdx.x < n_all) s_list[threadIdx.x] = list[(size_t)st * RS_CAP + threadIdx.x];
    // largest |dL/dout| among the contributing tiles (order-independent) -> the power-of-two quantum
    float m = 0.f;
    if (listed) { if (threadIdx.x < n_all) m = tmax[list[(size_t)st * RS_CAP + threadIdx.x]]; }
    else for (int ot = threadIdx.x; ot < out_tiles; ot += 256) {
        const int4 bb = bbox[ot];
        if (bb.x <= X0 + RS_TILE - 1 && bb.z >= X0 && bb.y <= Y0 + RS_TILE - 1 && bb.w >= Y0) m = fmaxf(m, tmax[ot]);
    }
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) m = fmaxf(m, __shfl_xor(m, d));
    if ((threadIdx.x & 63) == 0) s_red[threadIdx.x >> 6] = m;
    __syncthreads();
    m = fmaxf(fmaxf(s_red[0], s_red[1]), fmaxf(s_red[2], s_red[3]));
    // a term is at most m (bilinear weights <= 1) and at most n * 1024 terms meet in one accumulator:
    // quantum = 2^(e - bits), m < 2^e, bits = 62 - ceil(log2(n * 1024))  (>= 47 for listed tiles)
    // below 2^-100 the scale 2^(24-e) would overflow to +inf (and (int)(v * inf) is undefined): such a tile's terms are
    // accumulated on the 2^-100 scale instead, where they round to ~0, which is what they are
    int e = 0; (void)frexpf(m, &e);
    if (e < -100) e = -100;
    int lg = 10; { u32 v = (n > 1 ? n : 1u) - 1u; while (v) { ++lg; v >>= 1; } }
    const int bits = 62 - lg;
    const float up = ldexpf(1.0f, 24 - e);                   // term * up is below 2^24 in magnitude: its integer part fits an int32
    const int shift = bits - 24;                             // fixed point = (term * 2^(24-e)) * 2^shift: exact power-of-two scaling
    const size_t oplane = (size_t)g.Hc * g.Wc;
    if (m > 0.f && isfinite(m)) {
        // one list entry = one output tile; thread = one of its pixels.  Four entries per batch: their tap records and
        // cotangents are requested together (one round trip per batch instead of two per entry)
        auto add_taps = [&](const float4 rec, const float* go) {
            const u32 xy = __float_as_uint(rec.x);
            const int tx0 = (int)(short)(xy & 0xFFFFu), ty0 = (int)(short)(xy >> 16);
            const float fxw = rec.y, fyw = rec.z;
            const float w[4] = {(1.f - fxw) * (1.f - fyw), fxw * (1.f - fyw), (1.f - fxw) * fyw, fxw * fyw};
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int sx = tx0 + (q & 1), sy = ty0 + (q >> 1);
                const int lx = sx - X0, ly = sy - Y0;
                // in this source tile (and hence inside the image: every source tile is clipped by the write-out below)?
                if (lx < 0 || lx >= RS_TILE || ly < 0 || ly >= RS_TILE || sx >= g.W || sy >= g.H) continue;
                for (int ch = 0; ch < g.C; ++ch) {
                    const float v = w[q] * go[ch];                                                  // the term, as the scatter form had it
                    const float x = v * up;                                                         // exact (power of two), |x| < 2^24
                    const int hi = (int)x;                                                          // integer part
                    const int lo = (int)((x - (float)hi) * 16777216.0f);                            // 24 more fractional bits, exact
                    const long long fx = shift >= 24 ? (((long long)hi << 24) + (long long)lo) << (shift - 24)
                                                     : (((long long)hi << 24) + (long long)lo) >> (24 - shift);
                    atomicAdd(&acc[ch * 256 + ly * RS_TILE + lx], (unsigned long long)fx);
                }
            }
        };
        constexpr int RS_BATCH = 4, RS_CMAX = 4;             // channels held in registers per entry (more: loaded inside)
        if (listed && g.C <= RS_CMAX) {
            for (u32 k0 = 0; k0 < n; k0 += RS_BATCH) {
                float4 rec[RS_BATCH]; float go[RS_BATCH][RS_CMAX]; bool on[RS_BATCH];
#pragma unroll
                for (int j = 0; j < RS_BATCH; ++j) {
                    on[j] = false; rec[j] = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
                    for (int ch = 0; ch < RS_CMAX; ++ch) go[j][ch] = 0.f;
                    if (k0 + j < n) {
                        const int ot = (int)s_list[k0 + j];
                        const int xc = (ot % out_tiles_x) * RS_TILE + (threadIdx.x & 15), yc = (ot / out_tiles_x) * RS_TILE + (threadIdx.x >> 4);
                        if (xc < g.Wc && yc < g.Hc) {
                            on[j] = true;
                            const size_t px = (size_t)yc * g.Wc + xc;
                            rec[j] = taps[px];
#pragma unroll
                            for (int ch = 0; ch < RS_CMAX; ++ch) if (ch < g.C) go[j][ch] = grad_out[ch * oplane + px];
                        }
                    }
                }
#pragma unroll
                for (int j = 0; j < RS_BATCH; ++j) if (on[j]) add_taps(rec[j], go[j]);
            }
        } else {
            for (u32 k = 0; k < n; ++k) {
                int ot;
                if (listed) ot = (int)s_list[k];
                else {
                    ot = (int)k;
                    const int4 bb = bbox[ot];
                    if (!(bb.x <= X0 + RS_TILE - 1 && bb.z >= X0 && bb.y <= Y0 + RS_TILE - 1 && bb.w >= Y0)) continue;   // uniform
                }
                const int xc = (ot % out_tiles_x) * RS_TILE + (threadIdx.x & 15), yc = (ot / out_tiles_x) * RS_TILE + (threadIdx.x >> 4);
                if (xc >= g.Wc || yc >= g.Hc) continue;
                const size_t px = (size_t)yc * g.Wc + xc;
                const float4 rec = taps[px];
                float go[24];                                 // check_resample caps C at 24
                for (int ch = 0; ch < g.C; ++ch) go[ch] = grad_out[ch * oplane + px];
                const u32 xy = __float_as_uint(rec.x);
                const int tx0 = (int)(short)(xy & 0xFFFFu), ty0 = (int)(short)(xy >> 16);
                const float fxw = rec.y, fyw = rec.z;
                const float w[4] = {(1.f - fxw) * (1.f - fyw), fxw * (1.f - fyw), (1.f - fxw) * fyw, fxw * fyw};
                for (int q = 0; q < 4; ++q) {
                    const int sx = tx0 + (q & 1), sy = ty0 + (q >> 1);
                    const int lx = sx - X0, ly = sy - Y0;
                    if (lx < 0 || lx >= RS_TILE || ly < 0 || ly >= RS_TILE || sx >= g.W || sy >= g.H) continue;
                    for (int ch = 0; ch < g.C; ++ch) {
                        const float v = w[q] * go[ch];
                        const float x = v * up;
                        const int hi = (int)x;
                        const int lo = (int)((x - (float)hi) * 16777216.0f);
                        const long long fx = shift >= 24 ? (((long long)hi << 24) + (long long)lo) << (shift - 24)
                                                         : (((long long)hi << 24) + (long long)lo) >> (24 - shift);
                        atomicAdd(&acc[ch * 256 + ly * RS_TILE + lx], (unsigned long long)fx);
                    }
                }
            }
        }
    }
    __syncthreads();
    const int x = X0 + (threadIdx.x & 15), y = Y0 + (threadIdx.x >> 4);
    if (x < g.W && y < g.H) {
        const double q = ldexp(1.0, e - 24 - 24 - (shift - 24));            // value of one fixed-point unit
        // a non-finite cotangent anywhere on this source tile's list poisons the tile, as PyTorch's scatter would the taps it hits
        const bool bad = !isfinite(m);
        for (int ch = 0; ch < g.C; ++ch)
            grad_image[(size_t)ch * g.H * g.W + (size_t)y * g.W + x] =
                bad ? __builtin_nanf("") : (float)((double)(long long)acc[ch * 256 + threadIdx.x] * q);
    }
}

// dL/d(control flow): the adjoint of the bilinear upsample as a GATHER -- one wave per control node walks the cropped
// pixels whose upsampling taps can touch the node, recomputes their taps with the forward's own function and sums
// weight x dL/dflow in a fixed order.  No atomics (the scatter form put ~1200 float atomics on every node: 1.7 ms).
__global__ void __launch_bounds__(64)
resample_ctrl_gather_kernel(ResampleGeom g, const float2* __restrict__ gflow, float* __restrict__ grad_ctrl)
{
    const int node = blockIdx.x, ny = node / g.w, nx = node - ny * g.w, lane = threadIdx.x;
    const float sy = (float)g.Hf / (float)g.h, sx = (float)g.Wf / (float)g.w;     // destination pixels per control cell
    // destination rows whose source coordinate can fall in [ny - 1, ny + 1); the first / last node also collect the clamped ends
    int Ya = (ny == 0) ? 0 : (int)floorf(((float)ny - 0.5f) * sy - 0.5f) - 2, Yb = (ny == g.h - 1) ? g.Hf - 1 : (int)ceilf(((float)ny + 1.5f) * sy - 0.5f) + 2;
    int Xa = (nx == 0) ? 0 : (int)floorf(((float)nx - 0.5f) * sx - 0.5f) - 2, Xb = (nx == g.w - 1) ? g.Wf - 1 : (int)ceilf(((float)nx + 1.5f) * sx - 0.5f) + 2;
    Ya = max(Ya, g.y0); Yb = min(Yb, g.y0 + g.Hc - 1); Xa = max(Xa, g.x0); Xb = min(Xb, g.x0 + g.Wc - 1);
    float ax = 0.f, ay = 0.f;
    if (Ya <= Yb && Xa <= Xb) {
        const int nxr = Xb - Xa + 1, total = nxr * (Yb - Ya + 1);
        for (int i = lane; i < total; i += 64) {
            const int Y = Ya + i / nxr, X = Xa + i % nxr;
            const FlowTap f = flow_taps(g, Y, X);
            const float wgt = (f.i00 == node ? f.w00 : 0.f) + (f.i01 == node ? f.w01 : 0.f) + (f.i10 == node ? f.w10 : 0.f) +
                              (f.i11 == node ? f.w11 : 0.f);
            if (wgt != 0.f) {
                const float2 gf = gflow[(size_t)(Y - g.y0) * g.Wc + (X - g.x0)];
                ax = fmaf(wgt, gf.x, ax); ay = fmaf(wgt, gf.y, ay);
            }
        }
    }
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) { ax += __shfl_xor(ax, d); ay += __shfl_xor(ay, d); }
    if (lane == 0) { grad_ctrl[2 * node] = ax; grad_ctrl[2 * node + 1] = ay; }
}

static ResampleGeom make_geom(int C, int H, int W, int h, int w, int Hf, int Wf, int Hc, int Wc)
{
    ResampleGeom g{C, H, W, h, w, Hf, Wf, Hc, Wc, (Hf - Hc) / 2, (Wf - Wc) / 2};
    return g;
}

hipError_t launch_resample_fwd(const float* image, int C, int H, int W, const float* ctrl, int h, int w, int Hf, int Wf, int Hc, int Wc,
                               float* out, float* mask, float* flow_out, hipStream_t st)
{
    const ResampleGeom g = make_geom(C, H, W, h, w, Hf, Wf, Hc, Wc);
    hipLaunchKernelGGL(resample_fwd_kernel, dim3(cdiv(Wc, 64), cdiv(Hc, 4)), dim3(256), 0, st, g, image, ctrl, out, mask, flow_out);
    return hipGetLastError();
}

static size_t rs_carve(void* base, int H, int W, int Hc, int Wc, RsWork* w)
{
    char* p = reinterpret_cast<char*>(align_up(reinterpret_cast<size_t>(base), 256));
    const size_t To = (size_t)cdiv(Wc, RS_TILE) * cdiv(Hc, RS_TILE), Ts = (size_t)cdiv(W, RS_TILE) * cdiv(H, RS_TILE);
    RsWork r;
    r.gflow = reinterpret_cast<float2*>(p); p += align_up((size_t)Hc * Wc * sizeof(float2), 256);
    r.bbox = reinterpret_cast<int4*>(p); p += align_up(To * sizeof(int4), 256);
    r.tmax = reinterpret_cast<float*>(p); p += align_up(To * sizeof(float), 256);
    r.count = reinterpret_cast<u32*>(p); p += align_up(Ts * sizeof(u32), 256);
    r.list = reinterpret_cast<u32*>(p); p += align_up(Ts * RS_CAP * sizeof(u32), 256);
    r.taps = reinterpret_cast<float4*>(p); p += align_up((size_t)Hc * Wc * sizeof(float4), 256);
    if (w) *w = r;
    return (size_t)(p - reinterpret_cast<char*>(base));
}
size_t resample_workspace_bytes(int H, int W, int Hc, int Wc) { return rs_carve(nullptr, H, W, Hc, Wc, nullptr) + 512; }

hipError_t launch_resample_bwd(const float* image, int C, int H, int W, const float* ctrl, int h, int w, int Hf, int Wf, int Hc, int Wc,
                               const float* grad_out, void* workspace, float* grad_image, float* grad_ctrl, hipStream_t st)
{
    const ResampleGeom g = make_geom(C, H, W, h, w, Hf, Wf, Hc, Wc);
    RsWork wk; rs_carve(workspace, H, W, Hc, Wc, &wk);
    const int otx = cdiv(Wc, RS_TILE), oty = cdiv(Hc, RS_TILE), stx = cdiv(W, RS_TILE), sty = cdiv(H, RS_TILE);
    hipError_t e;
    if (grad_image && (e = hipMemsetAsync(wk.count, 0, (size_t)stx * sty * sizeof(u32), st)) != hipSuccess) return e;
    hipLaunchKernelGGL(resample_bwd_pixels_kernel, dim3(otx, oty), dim3(256), 0, st, g, image, ctrl, grad_out,
                       grad_ctrl ? wk.gflow : (float2*)nullptr, grad_image ? wk.bbox : (int4*)nullptr, wk.tmax, wk.count, wk.list, wk.taps, stx, sty);
    if (grad_image)
        hipLaunchKernelGGL(resample_gather_kernel, dim3(stx, sty), dim3(256), (size_t)C * 256 * sizeof(unsigned long long), st, g, ctrl, grad_out,
                           (const int4*)wk.bbox, (const float*)wk.tmax, (const u32*)wk.count, (const u32*)wk.list, (const float4*)wk.taps, grad_image, otx, otx * oty);
    if (grad_ctrl)
        hipLaunchKernelGGL(resample_ctrl_gather_kernel, dim3(h * w), dim3(64), 0, st, g, (const float2*)wk.gflow, grad_ctrl);
    return hipGetLastError();
}
