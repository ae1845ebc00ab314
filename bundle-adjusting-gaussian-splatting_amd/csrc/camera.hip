// camera.hip -- the pose -> matrix chain in one launch each way.
//
// The reference builds the four camera tensors of GaussianRasterizationSettings from the pose leaves with ~40 small
// PyTorch kernels and six 4x4 inversions per render() call, and autograd replays as many backwards
// (scene/cameras.py:356-381: get_intrinsic, get_world_view_transform, get_full_proj_transform, get_camera_center;
// quaternion_to_rotation_matrix :399-416; utils/graphics_utils.py:83-107 getProjectionMatrix).  This is 4x4 arithmetic:
// one thread does all of it.
//   forward : q = normalize(q0 + dq); R(q); R <- G R (optional global rotation); t = s (t0 + dt) (optional global
//             translation scale: scaling row 3 of inverse(W2C^T) by s and inverting back is t -> s t for a rigid W2C);
//             viewmatrix = W2C^T; intrinsic = P^T(fovx, fovy, znear, zfar); projmatrix = viewmatrix intrinsic;
//             campos = -R^-1 t (= inverse(viewmatrix)[3,:3])
//   backward: the adjoint of exactly those steps, to dq, dt, fovx, fovy (and G, s when given).
#include "bags_common.h"

struct CamIn {
    const float* q0; const float* dq;          // (4) w,x,y,z
    const float* t0; const float* dt;          // (3)
    const float* fovx; const float* fovy;      // scalars (learnable)
    const float* grot;                         // (3,3) row-major or NULL
    const float* gscale;                       // scalar or NULL
    float znear, zfar;
};

struct CamMid { float qn[4], n, R0[9], R[9], Ri[9], t[3], tu[3], s, c[3]; };   // tu: t before the global scale; Ri = R^-1; c = campos

__device__ static void cam_forward_mid(const CamIn& in, CamMid& m)
{
    float q[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) q[i] = in.q0[i] + in.dq[i];
    m.n = sqrtf(q[0] * q[0] + q[1] * q[1] + q[2] * q[2] + q[3] * q[3]);
#pragma unroll
    for (int i = 0; i < 4; ++i) m.qn[i] = q[i] / m.n;
    const float w = m.qn[0], x = m.qn[1], y = m.qn[2], z = m.qn[3];
    m.R0[0] = 1.f - 2.f * y * y - 2.f * z * z; m.R0[1] = 2.f * x * y - 2.f * w * z; m.R0[2] = 2.f * x * z + 2.f * w * y;
    m.R0[3] = 2.f * x * y + 2.f * w * z; m.R0[4] = 1.f - 2.f * x * x - 2.f * z * z; m.R0[5] = 2.f * y * z - 2.f * w * x;
    m.R0[6] = 2.f * x * z - 2.f * w * y; m.R0[7] = 2.f * y * z + 2.f * w * x; m.R0[8] = 1.f - 2.f * x * x - 2.f * y * y;
    if (in.grot) {
#pragma unroll
        for (int i = 0; i < 3; ++i)
#pragma unroll
            for (int j = 0; j < 3; ++j)
                m.R[3 * i + j] = in.grot[3 * i] * m.R0[j] + in.grot[3 * i + 1] * m.R0[3 + j] + in.grot[3 * i + 2] * m.R0[6 + j];
    } else {
#pragma unroll
        for (int i = 0; i < 9; ++i) m.R[i] = m.R0[i];
    }
    m.s = in.gscale ? in.gscale[0] : 1.f;
#pragma unroll
    for (int i = 0; i < 3; ++i) { m.tu[i] = in.t0[i] + in.dt[i]; m.t[i] = m.s * m.tu[i]; }
    // campos = inverse(viewmatrix)[3,:3] = -R^-1 t, with a genuine 3x3 inverse: R is orthonormal only while the optional
    // global rotation is, and the reference differentiates the inverse as a general matrix function
    const float* R = m.R;
    const float c00 = R[4] * R[8] - R[5] * R[7], c01 = R[5] * R[6] - R[3] * R[8], c02 = R[3] * R[7] - R[4] * R[6];
    const float idet = 1.0f / (R[0] * c00 + R[1] * c01 + R[2] * c02);
    m.Ri[0] = c00 * idet; m.Ri[1] = (R[2] * R[7] - R[1] * R[8]) * idet; m.Ri[2] = (R[1] * R[5] - R[2] * R[4]) * idet;
    m.Ri[3] = c01 * idet; m.Ri[4] = (R[0] * R[8] - R[2] * R[6]) * idet; m.Ri[5] = (R[2] * R[3] - R[0] * R[5]) * idet;
    m.Ri[6] = c02 * idet; m.Ri[7] = (R[1] * R[6] - R[0] * R[7]) * idet; m.Ri[8] = (R[0] * R[4] - R[1] * R[3]) * idet;
#pragma unroll
    for (int i = 0; i < 3; ++i) m.c[i] = -(m.Ri[3 * i] * m.t[0] + m.Ri[3 * i + 1] * m.t[1] + m.Ri[3 * i + 2] * m.t[2]);
}

__global__ void camera_fwd_kernel(CamIn in, float* __restrict__ V, float* __restrict__ M, float* __restrict__ K, float* __restrict__ C)
{
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    CamMid m; cam_forward_mid(in, m);
    float v[16], k[16];
#pragma unroll
    for (int i = 0; i < 3; ++i) {
#pragma unroll
        for (int j = 0; j < 3; ++j) v[4 * i + j] = m.R[3 * j + i];           // W2C^T
        v[4 * i + 3] = 0.f; v[12 + i] = m.t[i];
    }
    v[15] = 1.f;
#pragma unroll
    for (int i = 0; i < 16; ++i) k[i] = 0.f;
    const float tx = tanf(0.5f * in.fovx[0]), ty = tanf(0.5f * in.fovy[0]);
    // P (column-vector form): P00 = 2 zn / (r - l), P11 = 2 zn / (t - b), P22 = zf/(zf-zn), P23 = -zf zn/(zf-zn), P32 = 1
    const float right = tx * in.znear, top = ty * in.znear;
    k[0] = 2.f * in.znear / (right + right);                                   // K = P^T
    k[5] = 2.f * in.znear / (top + top);
    k[10] = in.zfar / (in.zfar - in.znear);
    k[14] = -(in.zfar * in.znear) / (in.zfar - in.znear);
    k[11] = 1.f;
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j)
            M[4 * i + j] = v[4 * i] * k[j] + v[4 * i + 1] * k[4 + j] + v[4 * i + 2] * k[8 + j] + v[4 * i + 3] * k[12 + j];
#pragma unroll
    for (int i = 0; i < 16; ++i) { V[i] = v[i]; K[i] = k[i]; }
#pragma unroll
    for (int i = 0; i < 3; ++i) C[i] = m.c[i];
}

__global__ void camera_bwd_kernel(CamIn in, const float* __restrict__ gV_, const float* __restrict__ gM, const float* __restrict__ gK_,
                                  const float* __restrict__ gC, float* __restrict__ g_dq, float* __restrict__ g_dt,
                                  float* __restrict__ g_fovx, float* __restrict__ g_fovy, float* __restrict__ g_grot, float* __restrict__ g_gscale)
{
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    CamMid m; cam_forward_mid(in, m);
    float v[16], k[16], gV[16], gK[16];
#pragma unroll
    for (int i = 0; i < 3; ++i) {
#pragma unroll
        for (int j = 0; j < 3; ++j) v[4 * i + j] = m.R[3 * j + i];
        v[4 * i + 3] = 0.f; v[12 + i] = m.t[i];
    }
    v[15] = 1.f;
#pragma unroll
    for (int i = 0; i < 16; ++i) k[i] = 0.f;
    const float tx = tanf(0.5f * in.fovx[0]), ty = tanf(0.5f * in.fovy[0]);
    k[0] = 1.f / tx; k[5] = 1.f / ty; k[10] = in.zfar / (in.zfar - in.znear); k[14] = -(in.zfar * in.znear) / (in.zfar - in.znear); k[11] = 1.f;
#pragma unroll
    for (int i = 0; i < 16; ++i) { gV[i] = gV_ ? gV_[i] : 0.f; gK[i] = gK_ ? gK_[i] : 0.f; }
    if (gM) {                                                  // M = V K
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                float a = 0.f, b = 0.f;
#pragma unroll
                for (int l = 0; l < 4; ++l) { a += gM[4 * i + l] * k[4 * j + l]; b += v[4 * l + i] * gM[4 * l + j]; }
                gV[4 * i + j] += a; gK[4 * i + j] += b;
            }
    }
    // K00 = 1 / tan(fx/2):  d/dfx = -(1 + tan^2) / (2 tan^2)
    if (g_fovx) g_fovx[0] = gK[0] * (-(1.f + tx * tx) / (2.f * tx * tx));
    if (g_fovy) g_fovy[0] = gK[5] * (-(1.f + ty * ty) / (2.f * ty * ty));
    float gR[9], gt[3];
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        gt[i] = gV[12 + i];
#pragma unroll
        for (int j = 0; j < 3; ++j) gR[3 * j + i] = gV[4 * i + j];          // V[i][j] = R[j][i]
    }
    if (gC) {                                                  // C = -R^-1 t:  dC = -R^-1 dR C - R^-1 dt
#pragma unroll
        for (int a = 0; a < 3; ++a) {
            const float u = m.Ri[a] * gC[0] + m.Ri[3 + a] * gC[1] + m.Ri[6 + a] * gC[2];      // (R^-T gC)_a
            gt[a] -= u;
#pragma unroll
            for (int b = 0; b < 3; ++b) gR[3 * a + b] -= u * m.c[b];
        }
    }
    // t = s tu
    if (g_gscale) g_gscale[0] = gt[0] * m.tu[0] + gt[1] * m.tu[1] + gt[2] * m.tu[2];
    if (g_dt) { g_dt[0] = m.s * gt[0]; g_dt[1] = m.s * gt[1]; g_dt[2] = m.s * gt[2]; }
    // R = G R0
    float gR0[9];
    if (in.grot) {
#pragma unroll
        for (int i = 0; i < 3; ++i)
#pragma unroll
            for (int j = 0; j < 3; ++j) {
                gR0[3 * i + j] = in.grot[i] * gR[j] + in.grot[3 + i] * gR[3 + j] + in.grot[6 + i] * gR[6 + j];       // G^T gR
                if (g_grot) g_grot[3 * i + j] = gR[3 * i] * m.R0[3 * j] + gR[3 * i + 1] * m.R0[3 * j + 1] + gR[3 * i + 2] * m.R0[3 * j + 2];   // gR R0^T
            }
    } else {
#pragma unroll
        for (int i = 0; i < 9; ++i) gR0[i] = gR[i];
    }
    // R0(qn) -> qn
    const float w = m.qn[0], x = m.qn[1], y = m.qn[2], z = m.qn[3];
    float gq[4];
    gq[0] = 2.f * (-z * gR0[1] + y * gR0[2] + z * gR0[3] - x * gR0[5] - y * gR0[6] + x * gR0[7]);
    gq[1] = 2.f * (y * gR0[1] + z * gR0[2] + y * gR0[3] - 2.f * x * gR0[4] - w * gR0[5] + z * gR0[6] + w * gR0[7] - 2.f * x * gR0[8]);
    gq[2] = 2.f * (-2.f * y * gR0[0] + x * gR0[1] + w * gR0[2] + x * gR0[3] + z * gR0[5] - w * gR0[6] + z * gR0[7] - 2.f * y * gR0[8]);
    gq[3] = 2.f * (-2.f * z * gR0[0] - w * gR0[1] + x * gR0[2] + w * gR0[3] - 2.f * z * gR0[4] + y * gR0[5] + x * gR0[6] + y * gR0[7]);
    // qn = q / |q|
    const float dot = m.qn[0] * gq[0] + m.qn[1] * gq[1] + m.qn[2] * gq[2] + m.qn[3] * gq[3];
    if (g_dq) {
#pragma unroll
        for (int i = 0; i < 4; ++i) g_dq[i] = (gq[i] - m.qn[i] * dot) / m.n;
    }
}

hipError_t launch_camera_fwd(const float* q0, const float* dq, const float* t0, const float* dt, const float* fovx, const float* fovy,
                             const float* grot, const float* gscale, float znear, float zfar,
                             float* V, float* M, float* K, float* C, hipStream_t st)
{
    CamIn in{q0, dq, t0, dt, fovx, fovy, grot, gscale, znear, zfar};
    hipLaunchKernelGGL(camera_fwd_kernel, dim3(1), dim3(64), 0, st, in, V, M, K, C);
    return hipGetLastError();
}

hipError_t launch_camera_bwd(const float* q0, const float* dq, const float* t0, const float* dt, const float* fovx, const float* fovy,
                             const float* grot, const float* gscale, float znear, float zfar,
                             const float* gV, const float* gM, const float* gK, const float* gC,
                             float* g_dq, float* g_dt, float* g_fovx, float* g_fovy, float* g_grot, float* g_gscale, hipStream_t st)
{
    CamIn in{q0, dq, t0, dt, fovx, fovy, grot, gscale, znear, zfar};
    hipLaunchKernelGGL(camera_bwd_kernel, dim3(1), dim3(64), 0, st, in, gV, gM, gK, gC, g_dq, g_dt, g_fovx, g_fovy, g_grot, g_gscale);
    return hipGetLastError();
}
