// activations.hip -- the parameter activations that feed the rasterizer, one launch each way.
//
// GaussianModel.get_features / get_opacity / get_scaling / get_rotation (scene/gaussian_model.py:118-141, set up at :26-43):
//   shs       = cat(features_dc (P,1,3), features_rest (P,K-1,3)) along dim 1
//   opacity   = sigmoid(_opacity)
//   scales    = exp(_scaling)
//   rotations = normalize(_rotation)            (x / max(|x|, 1e-12), torch.nn.functional.normalize)
// In PyTorch that is ~10 kernels forward and as many backward (0.28 ms per iteration at 500 k Gaussians, a quarter of the
// rasterizer itself); here one elementwise kernel each way: thread i copies element i of the (P,K,3) SH tensor and, for
// i < P, activates Gaussian i.  get_xyz is the identity and needs nothing.
#include "bags_common.h"

struct RawView { const float* dc; const float* rest; const float* opacity; const float* scaling; const float* rotation; };

__global__ void __launch_bounds__(256)
activations_fwd_kernel(int P, int K3, RawView raw, float* __restrict__ shs, float* __restrict__ opacity, float* __restrict__ scales,
                       float* __restrict__ rotations)
{
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    const size_t n = (size_t)P * K3;
    if (i < n && shs) {
        const size_t g = i / K3; const int r = (int)(i - g * K3);
        shs[i] = (r < 3) ? raw.dc[3 * g + r] : raw.rest[g * (K3 - 3) + (r - 3)];
    }
    if (i < (size_t)P) {
        if (opacity) opacity[i] = 1.0f / (1.0f + expf(-raw.opacity[i]));
        if (scales) {
#pragma unroll
            for (int a = 0; a < 3; ++a) scales[3 * i + a] = expf(raw.scaling[3 * i + a]);
        }
        if (rotations) {
            const float4 q = reinterpret_cast<const float4*>(raw.rotation)[i];
            const float d = fmaxf(sqrtf(q.x * q.x + q.y * q.y + q.z * q.z + q.w * q.w), 1e-12f);
            reinterpret_cast<float4*>(rotations)[i] = make_float4(q.x / d, q.y / d, q.z / d, q.w / d);
        }
    }
}

__global__ void __launch_bounds__(256)
activations_bwd_kernel(int P, int K3, RawView raw, const float* __restrict__ g_shs, const float* __restrict__ g_opacity,
                       const float* __restrict__ g_scales, const float* __restrict__ g_rotations, float* __restrict__ g_dc,
                       float* __restrict__ g_rest, float* __restrict__ g_opacity_raw, float* __restrict__ g_scaling,
                       float* __restrict__ g_rotation)
{
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    const size_t n = (size_t)P * K3;
    if (i < n && g_shs) {
        const size_t g = i / K3; const int r = (int)(i - g * K3);
        const float v = g_shs[i];
        if (r < 3) { if (g_dc) g_dc[3 * g + r] = v; }
        else if (g_rest) g_rest[g * (K3 - 3) + (r - 3)] = v;
    }
    if (i < (size_t)P) {
        if (g_opacity && g_opacity_raw) {
            const float s = 1.0f / (1.0f + expf(-raw.opacity[i]));
            g_opacity_raw[i] = g_opacity[i] * s * (1.0f - s);
        }
        if (g_scales && g_scaling) {
#pragma unroll
            for (int a = 0; a < 3; ++a) g_scaling[3 * i + a] = g_scales[3 * i + a] * expf(raw.scaling[3 * i + a]);
        }
        if (g_rotations && g_rotation) {
            const float4 q = reinterpret_cast<const float4*>(raw.rotation)[i];
            const float4 g = reinterpret_cast<const float4*>(g_rotations)[i];
            const float nrm = sqrtf(q.x * q.x + q.y * q.y + q.z * q.z + q.w * q.w);
            const float d = fmaxf(nrm, 1e-12f);
            float4 o;
            if (nrm > 1e-12f) {
                const float hx = q.x / d, hy = q.y / d, hz = q.z / d, hw = q.w / d;
                const float dot = hx * g.x + hy * g.y + hz * g.z + hw * g.w;
                o = make_float4((g.x - hx * dot) / d, (g.y - hy * dot) / d, (g.z - hz * dot) / d, (g.w - hw * dot) / d);
            } else {
                o = make_float4(g.x / d, g.y / d, g.z / d, g.w / d);      // clamped denominator: constant scale
            }
            reinterpret_cast<float4*>(g_rotation)[i] = o;
        }
    }
}

hipError_t launch_activations_fwd(int P, int K, const float* dc, const float* rest, const float* opacity, const float* scaling,
                                  const float* rotation, float* shs, float* o_opacity, float* o_scales, float* o_rot, hipStream_t st)
{
    if (P <= 0) return hipSuccess;
    const RawView raw{dc, rest, opacity, scaling, rotation};
    const size_t n = shs ? (size_t)P * K * 3 : (size_t)P;      // without the SH concatenation (packed features): one thread per Gaussian
    hipLaunchKernelGGL(activations_fwd_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, P, K * 3, raw, shs, o_opacity, o_scales, o_rot);
    return hipGetLastError();
}

hipError_t launch_activations_bwd(int P, int K, const float* dc, const float* rest, const float* opacity, const float* scaling,
                                  const float* rotation, const float* g_shs, const float* g_opacity, const float* g_scales,
                                  const float* g_rot, float* g_dc, float* g_rest, float* g_opacity_raw, float* g_scaling,
                                  float* g_rotation, hipStream_t st)
{
    if (P <= 0) return hipSuccess;
    const RawView raw{dc, rest, opacity, scaling, rotation};
    const size_t n = (g_shs && (g_dc || g_rest)) ? (size_t)P * K * 3 : (size_t)P;
    hipLaunchKernelGGL(activations_bwd_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, P, K * 3, raw, g_shs, g_opacity, g_scales,
                       g_rot, g_dc, g_rest, g_opacity_raw, g_scaling, g_rotation);
    return hipGetLastError();
}
