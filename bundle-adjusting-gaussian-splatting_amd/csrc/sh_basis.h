// Real spherical-harmonics basis, degrees 0..3, with the signs of utils/sh_utils.py:57-112 folded in:
// rgb_c = sum_t basis[t] * sh[t][c].  Pinned against the reference's eval_sh by tests/golden/sh_basis.npz.
#pragma once
#include <hip/hip_runtime.h>

#define SH_C0 0.28209479177387814f
#define SH_C1 0.4886025119029199f
#define SH_C2_0 1.0925484305920792f
#define SH_C2_1 -1.0925484305920792f
#define SH_C2_2 0.31539156525252005f
#define SH_C2_3 -1.0925484305920792f
#define SH_C2_4 0.5462742152960396f
#define SH_C3_0 -0.5900435899266435f
#define SH_C3_1 2.890611442640554f
#define SH_C3_2 -0.4570457994644658f
#define SH_C3_3 0.3731763325901154f
#define SH_C3_4 -0.4570457994644658f
#define SH_C3_5 1.445305721320277f
#define SH_C3_6 -0.5900435899266435f

__device__ __forceinline__ void sh_basis(int deg, float x, float y, float z, float* __restrict__ b)
{
    b[0] = SH_C0;
    if (deg > 0) {
        b[1] = -SH_C1 * y; b[2] = SH_C1 * z; b[3] = -SH_C1 * x;
        if (deg > 1) {
            const float xx = x * x, yy = y * y, zz = z * z, xy = x * y, yz = y * z, xz = x * z;
            b[4] = SH_C2_0 * xy; b[5] = SH_C2_1 * yz; b[6] = SH_C2_2 * (2.0f * zz - xx - yy);
            b[7] = SH_C2_3 * xz; b[8] = SH_C2_4 * (xx - yy);
            if (deg > 2) {
                b[9] = SH_C3_0 * y * (3.0f * xx - yy);
                b[10] = SH_C3_1 * xy * z;
                b[11] = SH_C3_2 * y * (4.0f * zz - xx - yy);
                b[12] = SH_C3_3 * z * (2.0f * zz - 3.0f * xx - 3.0f * yy);
                b[13] = SH_C3_4 * x * (4.0f * zz - xx - yy);
                b[14] = SH_C3_5 * z * (xx - yy);
                b[15] = SH_C3_6 * x * (xx - 3.0f * yy);
            }
        }
    }
}

// d basis[t] / d(x,y,z)
__device__ __forceinline__ void sh_basis_grad(int deg, float x, float y, float z, float* __restrict__ bx,
                                              float* __restrict__ by, float* __restrict__ bz)
{
    bx[0] = by[0] = bz[0] = 0.f;
    if (deg > 0) {
        bx[1] = 0.f; by[1] = -SH_C1; bz[1] = 0.f;
        bx[2] = 0.f; by[2] = 0.f; bz[2] = SH_C1;
        bx[3] = -SH_C1; by[3] = 0.f; bz[3] = 0.f;
        if (deg > 1) {
            const float xx = x * x, yy = y * y, zz = z * z, xy = x * y, yz = y * z, xz = x * z;
            bx[4] = SH_C2_0 * y; by[4] = SH_C2_0 * x; bz[4] = 0.f;
            bx[5] = 0.f; by[5] = SH_C2_1 * z; bz[5] = SH_C2_1 * y;
            bx[6] = SH_C2_2 * -2.0f * x; by[6] = SH_C2_2 * -2.0f * y; bz[6] = SH_C2_2 * 4.0f * z;
            bx[7] = SH_C2_3 * z; by[7] = 0.f; bz[7] = SH_C2_3 * x;
            bx[8] = SH_C2_4 * 2.0f * x; by[8] = SH_C2_4 * -2.0f * y; bz[8] = 0.f;
            if (deg > 2) {
                bx[9] = SH_C3_0 * 6.0f * xy;            by[9] = SH_C3_0 * (3.0f * xx - 3.0f * yy);       bz[9] = 0.f;
                bx[10] = SH_C3_1 * yz;                  by[10] = SH_C3_1 * xz;                            bz[10] = SH_C3_1 * xy;
                bx[11] = SH_C3_2 * -2.0f * xy;          by[11] = SH_C3_2 * (4.0f * zz - xx - 3.0f * yy);  bz[11] = SH_C3_2 * 8.0f * yz;
                bx[12] = SH_C3_3 * -6.0f * xz;          by[12] = SH_C3_3 * -6.0f * yz;                    bz[12] = SH_C3_3 * (6.0f * zz - 3.0f * xx - 3.0f * yy);
                bx[13] = SH_C3_4 * (4.0f * zz - 3.0f * xx - yy); by[13] = SH_C3_4 * -2.0f * xy;           bz[13] = SH_C3_4 * 8.0f * xz;
                bx[14] = SH_C3_5 * 2.0f * xz;           by[14] = SH_C3_5 * -2.0f * yz;                    bz[14] = SH_C3_5 * (xx - yy);
                bx[15] = SH_C3_6 * (3.0f * xx - 3.0f * yy); by[15] = SH_C3_6 * -6.0f * xy;                bz[15] = 0.f;
            }
        }
    }
}
