// resample.hip -- image-space distortion resampling of the rendered image.
//
// The reference warps the rendered image with a dense flow that a lens network predicts on a coarse control grid
// (utils/util_distortion.py:271-311 apply_distortion, apply2gt == False branch; call site train.py:255-263):
//     flow = F.interpolate(control flow (h,w,2) -> (Hf,Wf), bilinear, align_corners=False)
//     img  = F.grid_sample(image (C,H,W), flow, bilinear, zeros padding, align_corners=True)      -> (C,Hf,Wf)
//     img  = center_crop(img, Hc, Wc)          (utils/util_distortion.py:58-77: a second grid_sample on an integer grid)
//     mask = ~((img[0] == 0) & (img[1] == 0))
// i.e. three full-resolution passes and their autograd backward.  Here one kernel each way touches only the cropped
// pixels: the control flow is interpolated at the pixel, the image is sampled once; the backward scatters dL/dimage
// (float atomics, as PyTorch's grid_sample backward does) and GATHERS dL/d(control flow) per control node (no atomics).
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

__global__ void __launch_bounds__(256)
resample_bwd_kernel(ResampleGeom g, const float* __restrict__ image, const float* __restrict__ ctrl, const float* __restrict__ grad_out,
                    float* __restrict__ grad_image, float2* __restrict__ gflow /* (Hc,Wc): dL/d(upsampled flow), or NULL */)
{
    const int xc = blockIdx.x * 64 + (threadIdx.x & 63), yc = blockIdx.y * 4 + (threadIdx.x >> 6);
    if (xc >= g.Wc || yc >= g.Hc) return;
    const FlowTap f = flow_taps(g, g.y0 + yc, g.x0 + xc);
    const float2* c2 = reinterpret_cast<const float2*>(ctrl);
    const float2 a = c2[f.i00], b = c2[f.i01], c = c2[f.i10], d = c2[f.i11];
    const float gx = f.w00 * a.x + f.w01 * b.x + f.w10 * c.x + f.w11 * d.x;
    const float gy = f.w00 * a.y + f.w01 * b.y + f.w10 * c.y + f.w11 * d.y;
    const ImgTap t = img_taps(g, gx, gy);
    const float w00 = (1.f - t.fx) * (1.f - t.fy), w01 = t.fx * (1.f - t.fy), w10 = (1.f - t.fx) * t.fy, w11 = t.fx * t.fy;
    const size_t plane = (size_t)g.H * g.W, oplane = (size_t)g.Hc * g.Wc;
    const size_t base = (size_t)t.y0 * g.W + t.x0;
    float dix = 0.f, diy = 0.f;
    for (int ch = 0; ch < g.C; ++ch) {
        const float go = grad_out[ch * oplane + (size_t)yc * g.Wc + xc];
        const float* im = image + ch * plane;
        const float v00 = t.in00 ? im[base] : 0.f, v01 = t.in01 ? im[base + 1] : 0.f;
        const float v10 = t.in10 ? im[base + g.W] : 0.f, v11 = t.in11 ? im[base + g.W + 1] : 0.f;
        dix += go * ((v01 - v00) * (1.f - t.fy) + (v11 - v10) * t.fy);
        diy += go * ((v10 - v00) * (1.f - t.fx) + (v11 - v01) * t.fx);
        if (grad_image) {
            float* gi = grad_image + ch * plane;
            if (t.in00) unsafeAtomicAdd(gi + base, w00 * go);
            if (t.in01) unsafeAtomicAdd(gi + base + 1, w01 * go);
            if (t.in10) unsafeAtomicAdd(gi + base + g.W, w10 * go);
            if (t.in11) unsafeAtomicAdd(gi + base + g.W + 1, w11 * go);
        }
    }
    if (gflow) gflow[(size_t)yc * g.Wc + xc] = make_float2(dix * 0.5f * (float)(g.W - 1), diy * 0.5f * (float)(g.H - 1));
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

size_t resample_workspace_bytes(int Hc, int Wc) { return align_up((size_t)Hc * Wc * sizeof(float2), 256) + 256; }

hipError_t launch_resample_bwd(const float* image, int C, int H, int W, const float* ctrl, int h, int w, int Hf, int Wf, int Hc, int Wc,
                               const float* grad_out, void* workspace, float* grad_image, float* grad_ctrl, hipStream_t st)
{
    const ResampleGeom g = make_geom(C, H, W, h, w, Hf, Wf, Hc, Wc);
    hipError_t e;
    if (grad_image && (e = hipMemsetAsync(grad_image, 0, (size_t)C * H * W * sizeof(float), st)) != hipSuccess) return e;
    float2* gflow = grad_ctrl ? reinterpret_cast<float2*>(align_up(reinterpret_cast<size_t>(workspace), 256)) : nullptr;
    hipLaunchKernelGGL(resample_bwd_kernel, dim3(cdiv(Wc, 64), cdiv(Hc, 4)), dim3(256), 0, st, g, image, ctrl, grad_out, grad_image, gflow);
    if (grad_ctrl)
        hipLaunchKernelGGL(resample_ctrl_gather_kernel, dim3(h * w), dim3(64), 0, st, g, (const float2*)gflow, grad_ctrl);
    return hipGetLastError();
}
