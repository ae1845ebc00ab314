// loss.hip -- fused photometric loss terms, L1 and SSIM (11x11 Gaussian window, sigma 1.5, zero padding), and their
// gradient with respect to the rendered image  (utils/loss_utils.py:18-19,35-76;
// consumer train.py:311-331).
//
// The reference evaluates SSIM as five dense 11x11 depthwise conv2d calls plus ~20 elementwise kernels and lets
// autograd run the same number backwards.  Here:
//   forward  (loss_fwd_kernel)   one 32x32 tile of one channel per workgroup.  The tile of both images is staged in LDS
//            with its 5-pixel halo, the five window sums (a, b, a^2, b^2, ab) are taken separably (rows, then columns),
//            the SSIM value and its three partial derivatives (w.r.t. mu1, E[a^2], E[ab]) are formed in registers, the
//            derivative maps go to the workspace and the tile's sums of |a-b| and SSIM to one slot per workgroup;
//            loss_reduce_kernel adds the slots in fixed order (deterministic) and divides by the element count.
//   backward (loss_bwd_kernel)   dSSIM/da(p) = sum_q G(q-p) [ dm/dmu1(q) + 2 a(p) dm/dE11(q) + b(p) dm/dE12(q) ]:
//            the same separable filter over the three maps, plus sign(a-b) for L1, scaled by the two upstream scalars
//            read from device memory (no host round trip).
// Both kernels are HBM-bound: forward reads 2 and writes 3 floats per element, backward reads 5 and writes 1.
#include "bags_common.h"

#define LT 32                 // tile edge
#define LHALO 5
#define LEXT (LT + 2 * LHALO) // 42
#define LROW 48               // staged row: [x0 - 8, x0 + 40): twelve 16-byte groups, so rows load as aligned dwordx4 when W % 4 == 0
#define LOFF 3                // column of x0 - 5 (first tap of output 0) inside the staged row
#define LSTR 49               // LDS row stride (floats): odd, so the row pass (8 rows x 8 column groups per wave) is conflict-free
#define SSIM_C1 0.0001f       // 0.01^2
#define SSIM_C2 0.0009f       // 0.03^2

struct LossWindow { float w[11]; };

__device__ __forceinline__ float block_sum_256(float v, float* red)
{
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) v += __shfl_xor(v, d);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (lane == 0) red[wave] = v;
    __syncthreads();
    const float s = (red[0] + red[1]) + (red[2] + red[3]);
    __syncthreads();
    return s;
}

#ifndef LOSS_FWD_WAVES
#define LOSS_FWD_WAVES 1
#endif
#ifndef LOSS_BWD_WAVES
#define LOSS_BWD_WAVES 1
#endif
__global__ void __launch_bounds__(256, LOSS_FWD_WAVES)
loss_fwd_kernel(const float* __restrict__ img, const float* __restrict__ gt, int H, int W, int vec, LossWindow win,
                float* __restrict__ dmu, float* __restrict__ de11, float* __restrict__ de12, float* __restrict__ partials)
{
    // LDS: the two staged images (2 x 42 x 49 floats) are dead once the row pass has read them, so three of the five
    // row-filtered maps (a^2, b^2, ab) are written over them; the row pass keeps its sums in registers across the barrier that
    // makes this safe.  27.6 KB instead of 44.2 KB: five workgroups per CU instead of three.
    //   pool[0, 1386)      h3 (b^2), row stride 33      | over sa
    //   pool[1386, 2730)   h2 (a^2), row stride 32      | over the end of sa and the start of sb
    //   pool[2730, 4116)   h4 (ab),  row stride 33      | over sb
    //   pool[4116, 6888)   h0 (a), h1 (b), row stride 33
    constexpr int HS = LT + 1, HN = LEXT * HS;            // 33, 1386
    __shared__ float pool[2 * LEXT * LSTR + 2 * HN];
    static_assert(2 * LEXT * LSTR == 2 * HN + LEXT * LT, "the three aliased maps must fill the two staged images exactly");
    float (*sa)[LSTR] = reinterpret_cast<float (*)[LSTR]>(pool);
    float (*sb)[LSTR] = reinterpret_cast<float (*)[LSTR]>(pool + LEXT * LSTR);
    float* const h3 = pool; float* const h2 = pool + HN; float* const h4 = pool + HN + LEXT * LT;
    float* const h0 = pool + 2 * LEXT * LSTR; float* const h1 = h0 + HN;
    __shared__ float red[4];
    const int tid = threadIdx.x;
    const int x0 = blockIdx.x * LT, y0 = blockIdx.y * LT, c = blockIdx.z;
    const size_t plane = (size_t)H * W;
    const float* A = img + c * plane;
    const float* B = gt + c * plane;

    if (vec) {                     // W % 4 == 0 and 16-byte aligned planes: a 16-byte group is entirely inside or outside the image
        for (int i = tid; i < LEXT * (LROW / 4); i += 256) {
            const int ly = i / (LROW / 4), q = i - ly * (LROW / 4);
            const int gx = x0 - 8 + 4 * q, gy = y0 + ly - LHALO;
            const bool in = (gx >= 0) && (gx < W) && (gy >= 0) && (gy < H);
            const float4 z4 = make_float4(0.f, 0.f, 0.f, 0.f);
            const float4 va = in ? *reinterpret_cast<const float4*>(A + (size_t)gy * W + gx) : z4;
            const float4 vb = in ? *reinterpret_cast<const float4*>(B + (size_t)gy * W + gx) : z4;
            float* da = &sa[ly][4 * q]; float* db = &sb[ly][4 * q];
            da[0] = va.x; da[1] = va.y; da[2] = va.z; da[3] = va.w;
            db[0] = vb.x; db[1] = vb.y; db[2] = vb.z; db[3] = vb.w;
        }
    } else {
        for (int i = tid; i < LEXT * LEXT; i += 256) {
            const int ly = i / LEXT, lx = i - ly * LEXT;
            const int gx = x0 + lx - LHALO, gy = y0 + ly - LHALO;
            const bool in = (gx >= 0) && (gx < W) && (gy >= 0) && (gy < H);
            sa[ly][lx + LOFF] = in ? A[(size_t)gy * W + gx] : 0.f;
            sb[ly][lx + LOFF] = in ? B[(size_t)gy * W + gx] : 0.f;
        }
    }
    __syncthreads();
    // rows: LEXT x LT outputs, four consecutive columns per work item (14 LDS reads per image feed 4 x 11 taps); a thread's
    // one or two work items stay in registers until every thread has finished reading the staged images
    float S[2][5][4];
    static_assert(LEXT * (LT / 4) <= 2 * 256, "two work items per thread cover the row pass");
#pragma unroll
    for (int u = 0; u < 2; ++u) {
        const int i = tid + 256 * u;
#pragma unroll
        for (int q = 0; q < 5; ++q)
#pragma unroll
            for (int j = 0; j < 4; ++j) S[u][q][j] = 0.f;
        if (i < LEXT * (LT / 4)) {
            const int ly = i / (LT / 4), lx = (i - ly * (LT / 4)) * 4;
#pragma unroll
            for (int k = 0; k < 14; ++k) {
                const float a = sa[ly][lx + k + LOFF], b = sb[ly][lx + k + LOFF];
                const float aa = a * a, bb = b * b, ab = a * b;
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const int t = k - j;                       // tap index of output j
                    if (t >= 0 && t < 11) {
                        const float w = win.w[t];
                        S[u][0][j] = fmaf(w, a, S[u][0][j]); S[u][1][j] = fmaf(w, b, S[u][1][j]); S[u][2][j] = fmaf(w, aa, S[u][2][j]);
                        S[u][3][j] = fmaf(w, bb, S[u][3][j]); S[u][4][j] = fmaf(w, ab, S[u][4][j]);
                    }
                }
            }
        }
    }
    const int lx = tid & 31, ty = (tid >> 5) * 4;
    float l1v[4];                                            // |a - b| of the thread's four output pixels, while the images are there
#pragma unroll
    for (int r = 0; r < 4; ++r) l1v[r] = fabsf(sa[ty + r + LHALO][lx + LHALO + LOFF] - sb[ty + r + LHALO][lx + LHALO + LOFF]);
    __syncthreads();
#pragma unroll
    for (int u = 0; u < 2; ++u) {
        const int i = tid + 256 * u;
        if (i < LEXT * (LT / 4)) {
            const int ly = i / (LT / 4), lx4 = (i - ly * (LT / 4)) * 4;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                h0[ly * HS + lx4 + j] = S[u][0][j]; h1[ly * HS + lx4 + j] = S[u][1][j]; h2[ly * LT + lx4 + j] = S[u][2][j];
                h3[ly * HS + lx4 + j] = S[u][3][j]; h4[ly * HS + lx4 + j] = S[u][4][j];
            }
        }
    }
    __syncthreads();
    // columns + SSIM: one column, four consecutive rows per thread (14 reads per map feed 4 x 11 taps)
    float sum_l1 = 0.f, sum_ssim = 0.f;
    float cm[5][4];
#pragma unroll
    for (int q = 0; q < 5; ++q)
#pragma unroll
        for (int j = 0; j < 4; ++j) cm[q][j] = 0.f;
#pragma unroll
    for (int k = 0; k < 14; ++k) {
        const float v[5] = {h0[(ty + k) * HS + lx], h1[(ty + k) * HS + lx], h2[(ty + k) * LT + lx], h3[(ty + k) * HS + lx],
                            h4[(ty + k) * HS + lx]};
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int t = k - j;
            if (t >= 0 && t < 11) {
                const float w = win.w[t];
#pragma unroll
                for (int q = 0; q < 5; ++q) cm[q][j] = fmaf(w, v[q], cm[q][j]);
            }
        }
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int ly = ty + r;
        const int gx = x0 + lx, gy = y0 + ly;
        const float mu1 = cm[0][r], mu2 = cm[1][r], e11 = cm[2][r], e22 = cm[3][r], e12 = cm[4][r];
        if (gx < W && gy < H) {
            const float mu1s = mu1 * mu1, mu2s = mu2 * mu2, mu12 = mu1 * mu2;
            const float s11 = e11 - mu1s, s22 = e22 - mu2s, s12 = e12 - mu12;
            const float A1 = 2.f * mu12 + SSIM_C1, A2 = 2.f * s12 + SSIM_C2;
            const float B1 = mu1s + mu2s + SSIM_C1, B2 = s11 + s22 + SSIM_C2;
            const float iB1 = 1.0f / B1, iB2 = 1.0f / B2;
            const float m = (A1 * A2) * (iB1 * iB2);
            // partial derivatives of m w.r.t. (mu1 | E[a^2] | E[ab]), with sigma's dependence on mu1 folded in
            const float d_mu = 2.f * mu2 * (A2 - A1) * (iB1 * iB2) + 2.f * mu1 * m * (iB2 - iB1);
            const float d_e11 = -m * iB2;
            const float d_e12 = 2.f * A1 * (iB1 * iB2);
            const size_t o = c * plane + (size_t)gy * W + gx;
            dmu[o] = d_mu; de11[o] = d_e11; de12[o] = d_e12;
            sum_ssim += m;
            sum_l1 += l1v[r];
        }
    }
    const float t_l1 = block_sum_256(sum_l1, red);
    const float t_ss = block_sum_256(sum_ssim, red);
    if (tid == 0) {
        const size_t slot = ((size_t)blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x;
        partials[2 * slot] = t_l1; partials[2 * slot + 1] = t_ss;
    }
}

// one workgroup: fixed-order sum of the per-tile slots in fp64, then the means
__global__ void __launch_bounds__(256)
loss_reduce_kernel(const float* __restrict__ partials, int nslots, double inv_count, float* __restrict__ out_terms,
                   const int combined, const float lambda_dssim)
{
    __shared__ double r0[256], r1[256];
    double a = 0.0, b = 0.0;
    for (int i = threadIdx.x; i < nslots; i += 256) { a += (double)partials[2 * i]; b += (double)partials[2 * i + 1]; }
    r0[threadIdx.x] = a; r1[threadIdx.x] = b;
    __syncthreads();
    for (int d = 128; d >= 1; d >>= 1) {
        if ((int)threadIdx.x < d) { r0[threadIdx.x] += r0[threadIdx.x + d]; r1[threadIdx.x] += r1[threadIdx.x + d]; }
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        const float l1 = (float)(r0[0] * inv_count), ss = (float)(r1[0] * inv_count);
        if (combined) {     // train.py:325 with the reference's own roundings: (1 - lambda) * L1 + lambda * (1 - SSIM), fp32, left to right
            out_terms[0] = __fadd_rn(__fmul_rn(1.0f - lambda_dssim, l1), __fmul_rn(lambda_dssim, __fsub_rn(1.0f, ss))); out_terms[1] = l1; out_terms[2] = ss;
        } else { out_terms[0] = l1; out_terms[1] = ss; }
    }
}

__global__ void __launch_bounds__(256, LOSS_BWD_WAVES)
loss_bwd_kernel(const float* __restrict__ img, const float* __restrict__ gt, int H, int W, int vec, LossWindow win,
                const float* __restrict__ dmu, const float* __restrict__ de11, const float* __restrict__ de12,
                const float* __restrict__ grad_terms, float inv_count, float* __restrict__ grad_img, const int combined,
                const float lambda_dssim)
{
    // 42-wide rows, scalar staging.  The three row-filtered maps are written over the staged ones (the row pass keeps its sums
    // in registers across the barrier in between): 21.7 KB of LDS instead of 38.3 KB, seven workgroups per CU instead of four.
    __shared__ float sm[3][LEXT][LEXT + 1];
    constexpr int HS = LT + 1, HN = LEXT * HS;
    static_assert(3 * HN <= 3 * LEXT * (LEXT + 1), "the row-filtered maps fit over the staged ones");
    float* const hp = &sm[0][0][0];
    const int tid = threadIdx.x;
    const int x0 = blockIdx.x * LT, y0 = blockIdx.y * LT, c = blockIdx.z;
    const size_t plane = (size_t)H * W;
    const float* M0 = dmu + c * plane; const float* M1 = de11 + c * plane; const float* M2 = de12 + c * plane;
    (void)vec;
    for (int i = tid; i < LEXT * LEXT; i += 256) {
        const int ly = i / LEXT, lx = i - ly * LEXT;
        const int gx = x0 + lx - LHALO, gy = y0 + ly - LHALO;
        const bool in = (gx >= 0) && (gx < W) && (gy >= 0) && (gy < H);
        const size_t o = (size_t)gy * W + gx;
        sm[0][ly][lx] = in ? M0[o] : 0.f; sm[1][ly][lx] = in ? M1[o] : 0.f; sm[2][ly][lx] = in ? M2[o] : 0.f;
    }
    __syncthreads();
    float S[2][3][4];
#pragma unroll
    for (int u = 0; u < 2; ++u) {
        const int i = tid + 256 * u;
#pragma unroll
        for (int q = 0; q < 3; ++q)
#pragma unroll
            for (int j = 0; j < 4; ++j) S[u][q][j] = 0.f;
        if (i < LEXT * (LT / 4)) {
            const int ly = i / (LT / 4), lx = (i - ly * (LT / 4)) * 4;
#pragma unroll
            for (int k = 0; k < 14; ++k) {
                const float m0 = sm[0][ly][lx + k], m1 = sm[1][ly][lx + k], m2 = sm[2][ly][lx + k];
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const int t = k - j;
                    if (t >= 0 && t < 11) {
                        const float w = win.w[t];
                        S[u][0][j] = fmaf(w, m0, S[u][0][j]); S[u][1][j] = fmaf(w, m1, S[u][1][j]); S[u][2][j] = fmaf(w, m2, S[u][2][j]);
                    }
                }
            }
        }
    }
    __syncthreads();                                         // every thread has read the staged maps
#pragma unroll
    for (int u = 0; u < 2; ++u) {
        const int i = tid + 256 * u;
        if (i < LEXT * (LT / 4)) {
            const int ly = i / (LT / 4), lx = (i - ly * (LT / 4)) * 4;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                hp[ly * HS + lx + j] = S[u][0][j]; hp[HN + ly * HS + lx + j] = S[u][1][j]; hp[2 * HN + ly * HS + lx + j] = S[u][2][j];
            }
        }
    }
    __syncthreads();
    // combined: grad_terms[0] is dL/d(loss) of loss = (1 - lambda) L1 + lambda (1 - SSIM)
    const float g_l1 = (combined ? grad_terms[0] * (1.0f - lambda_dssim) : grad_terms[0]) * inv_count;
    const float g_ss = (combined ? -(grad_terms[0] * lambda_dssim) : grad_terms[1]) * inv_count;
    const int lx = tid & 31, ty = (tid >> 5) * 4;
    float cf[3][4];
#pragma unroll
    for (int q = 0; q < 3; ++q)
#pragma unroll
        for (int j = 0; j < 4; ++j) cf[q][j] = 0.f;
#pragma unroll
    for (int k = 0; k < 14; ++k) {
        const float v0 = hp[(ty + k) * HS + lx], v1 = hp[HN + (ty + k) * HS + lx], v2 = hp[2 * HN + (ty + k) * HS + lx];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int t = k - j;
            if (t >= 0 && t < 11) {
                const float w = win.w[t];
                cf[0][j] = fmaf(w, v0, cf[0][j]); cf[1][j] = fmaf(w, v1, cf[1][j]); cf[2][j] = fmaf(w, v2, cf[2][j]);
            }
        }
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int ly = ty + r;
        const int gx = x0 + lx, gy = y0 + ly;
        const float f0 = cf[0][r], f1 = cf[1][r], f2 = cf[2][r];
        if (gx < W && gy < H) {
            const size_t o = c * plane + (size_t)gy * W + gx;
            const float a = img[o], b = gt[o];
            const float d = a - b;
            const float sgn = (d > 0.f) ? 1.f : ((d < 0.f) ? -1.f : 0.f);
            grad_img[o] = g_ss * (f0 + 2.f * a * f1 + b * f2) + g_l1 * sgn;
        }
    }
}

static LossWindow make_window()
{   // utils/loss_utils.py:35-37: exp(-(x-5)^2 / (2 sigma^2)) evaluated in double, stored as float, normalised in float
    LossWindow w; float s = 0.f;
    for (int i = 0; i < 11; ++i) { w.w[i] = (float)exp(-(double)((i - 5) * (i - 5)) / (2.0 * 1.5 * 1.5)); s += w.w[i]; }
    for (int i = 0; i < 11; ++i) w.w[i] /= s;
    return w;
}

size_t loss_workspace_bytes(int C, int H, int W)
{
    const size_t n = (size_t)C * H * W;
    const size_t slots = (size_t)C * cdiv(H, LT) * cdiv(W, LT);
    return 3 * align_up(n * sizeof(float), 256) + align_up(2 * slots * sizeof(float), 256);
}

static void carve_loss(void* ws, int C, int H, int W, float** dmu, float** de11, float** de12, float** partials)
{
    char* p = reinterpret_cast<char*>(ws);
    const size_t n = align_up((size_t)C * H * W * sizeof(float), 256);
    *dmu = reinterpret_cast<float*>(p); *de11 = reinterpret_cast<float*>(p + n); *de12 = reinterpret_cast<float*>(p + 2 * n);
    *partials = reinterpret_cast<float*>(p + 3 * n);
}

hipError_t launch_loss_fwd(const float* img, const float* gt, int C, int H, int W, void* ws, float* out_terms, hipStream_t st,
                           bool combined, float lambda_dssim)
{
    float *dmu, *de11, *de12, *partials;
    carve_loss(ws, C, H, W, &dmu, &de11, &de12, &partials);
    const dim3 grid(cdiv(W, LT), cdiv(H, LT), C);
    const LossWindow win = make_window();
    const int vec = ((W & 3) == 0) && (((size_t)img | (size_t)gt) & 15) == 0;
    hipLaunchKernelGGL(loss_fwd_kernel, grid, dim3(256), 0, st, img, gt, H, W, vec, win, dmu, de11, de12, partials);
    const int nslots = (int)(grid.x * grid.y * grid.z);
    hipLaunchKernelGGL(loss_reduce_kernel, dim3(1), dim3(256), 0, st, (const float*)partials, nslots,
                       1.0 / ((double)C * H * W), out_terms, combined ? 1 : 0, lambda_dssim);
    return hipGetLastError();
}

hipError_t launch_loss_bwd(const float* img, const float* gt, int C, int H, int W, const void* ws, const float* grad_terms,
                           float* grad_img, hipStream_t st, bool combined, float lambda_dssim)
{
    float *dmu, *de11, *de12, *partials;
    carve_loss(const_cast<void*>(ws), C, H, W, &dmu, &de11, &de12, &partials);
    const dim3 grid(cdiv(W, LT), cdiv(H, LT), C);
    const LossWindow win = make_window();
    const int vec = ((W & 3) == 0) && (((size_t)dmu | (size_t)de11 | (size_t)de12) & 15) == 0;
    hipLaunchKernelGGL(loss_bwd_kernel, grid, dim3(256), 0, st, img, gt, H, W, vec, win, (const float*)dmu, (const float*)de11,
                       (const float*)de12, grad_terms, (float)(1.0 / ((double)C * H * W)), grad_img, combined ? 1 : 0, lambda_dssim);
    return hipGetLastError();
}
