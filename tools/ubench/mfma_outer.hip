// Microbenchmark (diagnostic, not product): can the per-splat moment sums of blend_bwd's row loop ride on the MFMA pipe?
//
// v_mfma_f32_16x16x1_4b_f32 does four independent 16x16x1 outer products per instruction, one per 16-lane DPP row --
// exactly the "row = block, lane = splat" layout of blend_bwd_scan_kernel: with A = the lane's per-(pixel, splat) weight
// and B = the pixel's feature held by lane n of the row, D[row][splat][n] accumulates sum_pixel A * B, i.e. the nine
// sums sum q [1, dx, dy, dx^2, dx dy, dy^2] and sum w [g0, g1, g2] of a splat -- but they need TWO weights (q and w), hence two
// MFMAs per pixel, eight per row step of four pixels, and they would replace ~18 packed VALU instructions of ~133.
//
// Kernels (one "row step" per loop iteration, instruction mix of the real loop: see csrc/blend.hip):
//   0  the row step as it is:        51 v_pk_*  + 36 DPP + 8 transcendental + 38 plain
//   1  sums on the MFMA pipe:        33 v_pk_*  + 36 DPP + 8 transcendental + 38 plain + 8 v_mfma_f32_16x16x1_4b_f32
//   2  the same with ONE MFMA/pixel: 33 v_pk_*  + ...                                  + 4 MFMA  (if q and w could share one)
//   3  only the VALU part of 1/2:    33 v_pk_*  + 36 DPP + 8 transcendental + 38 plain            (what the MFMAs cost on top)
// each at 1, 2 and 3 waves per SIMD (the product kernel runs 3).
// Build + run on the GPU box: hipcc --offload-arch=gfx950 -O3 mfma_outer.hip -o /tmp/mfma_outer && /tmp/mfma_outer
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>

typedef float f2 __attribute__((ext_vector_type(2)));
typedef float f16v __attribute__((ext_vector_type(16)));

#define PK8()                                                                                                                 \
    asm volatile("v_pk_fma_f32 %0, %0, %8, %8\n v_pk_fma_f32 %1, %1, %8, %8\n v_pk_fma_f32 %2, %2, %8, %8\n v_pk_fma_f32 %3, %3, %8, %8\n" \
                 "v_pk_fma_f32 %4, %4, %8, %8\n v_pk_fma_f32 %5, %5, %8, %8\n v_pk_fma_f32 %6, %6, %8, %8\n v_pk_fma_f32 %7, %7, %8, %8"  \
                 : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3), "+v"(p4), "+v"(p5), "+v"(p6), "+v"(p7) : "v"(pb))
#define PK1(P) asm volatile("v_pk_fma_f32 %0, %0, %1, %1" : "+v"(P) : "v"(pb))
#define FMA8()                                                                                                                \
    asm volatile("v_fma_f32 %0, %0, %8, %9\n v_fma_f32 %1, %1, %8, %9\n v_fma_f32 %2, %2, %8, %9\n v_fma_f32 %3, %3, %8, %9\n"  \
                 "v_fma_f32 %4, %4, %8, %9\n v_fma_f32 %5, %5, %8, %9\n v_fma_f32 %6, %6, %8, %9\n v_fma_f32 %7, %7, %8, %9"     \
                 : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b0), "v"(b1))
#define FMA1(A) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(A) : "v"(b0), "v"(b1))
#define SCAN_STEP(PAT)                                                                                                        \
    asm volatile("s_nop 1\n v_fmac_f32_dpp %4, %4, %0 " PAT "\n v_fmac_f32_dpp %5, %5, %1 " PAT "\n v_fmac_f32_dpp %6, %6, %2 " PAT "\n" \
                 "v_fmac_f32_dpp %7, %7, %3 " PAT "\n v_mul_f32_dpp %0, %0, %0 " PAT "\n v_mul_f32_dpp %1, %1, %1 " PAT "\n"   \
                 "v_mul_f32_dpp %2, %2, %2 " PAT "\n v_mul_f32_dpp %3, %3, %3 " PAT                                            \
                 : "+v"(c0), "+v"(c1), "+v"(c2), "+v"(c3), "+v"(c4), "+v"(c5), "+v"(c6), "+v"(c7))
#define TRANS8()                                                                                                              \
    asm volatile("v_exp_f32 %0, %0\n v_exp_f32 %1, %1\n v_exp_f32 %2, %2\n v_exp_f32 %3, %3\n v_rcp_f32 %4, %4\n v_rcp_f32 %5, %5\n" \
                 "v_rcp_f32 %6, %6\n v_rcp_f32 %7, %7" : "+v"(t0), "+v"(t1), "+v"(t2), "+v"(t3), "+v"(t4), "+v"(t5), "+v"(t6), "+v"(t7))

template <int KIND, int WPS>
__global__ void __launch_bounds__(256, WPS) k_row(float* out, int iters)
{
    float a0 = threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
    float c0 = 0.99f, c1 = 0.98f, c2 = 0.97f, c3 = 0.96f, c4 = 0.1f, c5 = 0.2f, c6 = 0.3f, c7 = 0.4f;
    float t0 = -1.f, t1 = -2.f, t2 = -3.f, t3 = -.5f, t4 = 1.5f, t5 = 2.5f, t6 = 3.5f, t7 = 4.5f;
    const float b0 = 1.0001f, b1 = 0.9999f;
    f2 p0 = {a0, a1}, p1 = {a2, a3}, p2 = {a4, a5}, p3 = {a6, a7}, p4 = {a1, a2}, p5 = {a3, a4}, p6 = {a5, a6}, p7 = {a7, a0};
    const f2 pb = {b0, b1};
    f16v acc0, acc1;
#pragma unroll
    for (int i = 0; i < 16; ++i) { acc0[i] = 0.f; acc1[i] = 0.f; }
    for (int i = 0; i < iters; ++i) {
        // ---- alpha of the four pixels, validity, x = c . g: 13 packed, 22 plain, 8 transcendental
        PK8(); PK1(p0); PK1(p1); PK1(p2); PK1(p3); PK1(p4);
        FMA8(); TRANS8(); FMA8(); FMA1(a0); FMA1(a1); FMA1(a2); FMA1(a3); FMA1(a4); FMA1(a5);
        // ---- the four affine scans + the shift: 36 DPP
        SCAN_STEP("row_shr:1 row_mask:0xf bank_mask:0xf"); SCAN_STEP("row_shr:2 row_mask:0xf bank_mask:0xf");
        SCAN_STEP("row_shr:4 row_mask:0xf bank_mask:0xf"); SCAN_STEP("row_shr:8 row_mask:0xf bank_mask:0xf");
        asm volatile("s_nop 1\n v_mov_b32_dpp %0, %4 row_shr:1 row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %1, %5 row_shr:1 row_mask:0xf bank_mask:0xf\n"
                     "v_mov_b32_dpp %2, %6 row_shr:1 row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %3, %7 row_shr:1 row_mask:0xf bank_mask:0xf"
                     : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(c4), "v"(c5), "v"(c6), "v"(c7));
        // ---- T, dL/dalpha, q, w: 20 packed, 16 plain
        PK8(); PK8(); PK1(p0); PK1(p1); PK1(p2); PK1(p3);
        FMA8(); FMA8();
        // ---- the per-splat sums
        if (KIND == 0) {                         // 18 packed (a0..a2: 6, a6: 2, the abs part: 10 counted with the plain ones above)
            PK8(); PK8(); PK1(p0); PK1(p1);
        } else if (KIND == 1) {                  // two outer products per pixel: A = q resp. w of the lane's splat, B = the pixel's features
#pragma unroll
            for (int px = 0; px < 4; ++px) {
                const float q = (px & 1) ? p4.x : p4.y, w = (px & 1) ? p5.x : p5.y, fq = (px & 2) ? p6.x : p6.y, fw = (px & 2) ? p7.x : p7.y;
                acc0 = __builtin_amdgcn_mfma_f32_16x16x1f32(q, fq, acc0, 0, 0, 0);
                acc1 = __builtin_amdgcn_mfma_f32_16x16x1f32(w, fw, acc1, 0, 0, 0);
            }
        } else if (KIND == 2) {                  // one outer product per pixel
#pragma unroll
            for (int px = 0; px < 4; ++px) {
                const float q = (px & 1) ? p4.x : p4.y, fq = (px & 2) ? p6.x : p6.y;
                if (px & 1) acc0 = __builtin_amdgcn_mfma_f32_16x16x1f32(q, fq, acc0, 0, 0, 0);
                else acc1 = __builtin_amdgcn_mfma_f32_16x16x1f32(q, fq, acc1, 0, 0, 0);
            }
        }
    }
    float s = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 + c0 + c1 + c2 + c3 + c4 + c5 + c6 + c7 + t0 + t1 + t2 + t3 + t4 + t5 + t6 + t7 +
              p0.x + p1.y + p2.x + p3.y + p4.x + p5.y + p6.x + p7.y;
#pragma unroll
    for (int i = 0; i < 16; ++i) s += acc0[i] + acc1[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int KIND, int WPS>
static double run(float* out, int iters)
{
    int dev = 0, cus = 0;
    hipGetDevice(&dev);
    hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
    const int grid = cus * WPS;                  // WPS workgroups of 4 waves per CU = WPS waves per SIMD
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL((k_row<KIND, WPS>), dim3(grid), dim3(256), 0, 0, out, iters / 10);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL((k_row<KIND, WPS>), dim3(grid), dim3(256), 0, 0, out, iters);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms = 0.f;
    hipEventElapsedTime(&ms, e0, e1);
    return (double)ms * 1e6 / iters / WPS;       // ns of SIMD time per row step
}

int main()
{
    float* out = nullptr;
    hipMalloc(&out, sizeof(float) * 256 * 256 * 8);
    const int iters = 20000;
    printf("ns of SIMD time per row step (lower = faster); columns: waves per SIMD 1 2 3\n");
#define ROW(K, NAME) printf("%-52s %8.1f %8.1f %8.1f\n", NAME, run<K, 1>(out, iters), run<K, 2>(out, iters), run<K, 3>(out, iters));
    ROW(0, "row step as it is (133 VALU)");
    ROW(3, "VALU part without the 18 packed sums (115 VALU)");
    ROW(1, "115 VALU + 8 v_mfma_f32_16x16x1_4b_f32");
    ROW(2, "115 VALU + 4 v_mfma_f32_16x16x1_4b_f32");
    hipFree(out);
    return 0;
}
