// Microbenchmark (diagnostic, not product): VALU / DPP / transcendental / LDS issue cost on gfx950 as a function of the waves
// resident per SIMD.  Build + run on the GPU box: hipcc --offload-arch=gfx950 -O3 issue_rates.hip -o /tmp/issue_rates && /tmp/issue_rates
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>
#include <string>

#define REP8(X) X X X X X X X X
#define ITER 20000

template <int KIND>
__global__ void __launch_bounds__(256) k_issue(float* out, int iters, unsigned long long* cyc)
{
    __shared__ float4 lds[1024];
    float a0 = threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
    float b0 = 1.0001f, b1 = 0.9999f;
    typedef float f2 __attribute__((ext_vector_type(2)));
    f2 p0 = {a0, a1}, p1 = {a2, a3}, p2 = {a4, a5}, p3 = {a6, a7}, p4 = {a1, a2}, p5 = {a3, a4}, p6 = {a5, a6}, p7 = {a7, a0};
    f2 pb = {b0, b1};
    lds[threadIdx.x] = make_float4(a0, a1, a2, a3); lds[threadIdx.x + 256] = lds[threadIdx.x]; lds[threadIdx.x + 512] = lds[threadIdx.x]; lds[threadIdx.x + 768] = lds[threadIdx.x];
    __syncthreads();
    const float4* lp = &lds[(threadIdx.x >> 4) * 33 & 1023];      // one address per 16-lane row (broadcast inside the row)
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int i = 0; i < iters; ++i) {
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        if (KIND == 0) {        // 8 independent v_fma_f32
            asm volatile("v_fma_f32 %0, %0, %8, %9\n v_fma_f32 %1, %1, %8, %9\n v_fma_f32 %2, %2, %8, %9\n v_fma_f32 %3, %3, %8, %9\n"
                         "v_fma_f32 %4, %4, %8, %9\n v_fma_f32 %5, %5, %8, %9\n v_fma_f32 %6, %6, %8, %9\n v_fma_f32 %7, %7, %8, %9"
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b0), "v"(b1));
        } else if (KIND == 1) { // 8 independent v_pk_fma_f32
            asm volatile("v_pk_fma_f32 %0, %0, %8, %8\n v_pk_fma_f32 %1, %1, %8, %8\n v_pk_fma_f32 %2, %2, %8, %8\n v_pk_fma_f32 %3, %3, %8, %8\n"
                         "v_pk_fma_f32 %4, %4, %8, %8\n v_pk_fma_f32 %5, %5, %8, %8\n v_pk_fma_f32 %6, %6, %8, %8\n v_pk_fma_f32 %7, %7, %8, %8"
                         : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3), "+v"(p4), "+v"(p5), "+v"(p6), "+v"(p7) : "v"(pb));
        } else if (KIND == 2) { // 8 independent DPP fmac (row_shr:1)
            asm volatile("s_nop 1\n v_fmac_f32_dpp %0, %8, %9 row_shr:1 row_mask:0xf bank_mask:0xf\n v_fmac_f32_dpp %1, %8, %9 row_shr:1 row_mask:0xf bank_mask:0xf\n"
                         "v_fmac_f32_dpp %2, %8, %9 row_shr:1 row_mask:0xf bank_mask:0xf\n v_fmac_f32_dpp %3, %8, %9 row_shr:1 row_mask:0xf bank_mask:0xf\n"
                         "v_fmac_f32_dpp %4, %8, %9 row_shr:1 row_mask:0xf bank_mask:0xf\n v_fmac_f32_dpp %5, %8, %9 row_shr:1 row_mask:0xf bank_mask:0xf\n"
                         "v_fmac_f32_dpp %6, %8, %9 row_shr:1 row_mask:0xf bank_mask:0xf\n v_fmac_f32_dpp %7, %8, %9 row_shr:1 row_mask:0xf bank_mask:0xf"
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b0), "v"(b1));
        } else if (KIND == 3) { // 8 independent v_exp_f32
            asm volatile("v_exp_f32 %0, %0\n v_exp_f32 %1, %1\n v_exp_f32 %2, %2\n v_exp_f32 %3, %3\n v_exp_f32 %4, %4\n v_exp_f32 %5, %5\n v_exp_f32 %6, %6\n v_exp_f32 %7, %7"
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));
        } else if (KIND == 4) { // dependent chain of 8 v_fma_f32
            asm volatile("v_fma_f32 %0, %0, %1, %2\n v_fma_f32 %0, %0, %1, %2\n v_fma_f32 %0, %0, %1, %2\n v_fma_f32 %0, %0, %1, %2\n"
                         "v_fma_f32 %0, %0, %1, %2\n v_fma_f32 %0, %0, %1, %2\n v_fma_f32 %0, %0, %1, %2\n v_fma_f32 %0, %0, %1, %2"
                         : "+v"(a0) : "v"(b0), "v"(b1));
        } else if (KIND == 5) { // 8 ds_read_b128, one address per row, then wait
            float4 r0, r1, r2, r3, r4, r5, r6, r7;
            asm volatile("ds_read_b128 %0, %8\n ds_read_b128 %1, %8 offset:16\n ds_read_b128 %2, %8 offset:32\n ds_read_b128 %3, %8 offset:48\n"
                         "ds_read_b128 %4, %8 offset:64\n ds_read_b128 %5, %8 offset:80\n ds_read_b128 %6, %8 offset:96\n ds_read_b128 %7, %8 offset:112\n s_waitcnt lgkmcnt(0)"
                         : "=v"(r0), "=v"(r1), "=v"(r2), "=v"(r3), "=v"(r4), "=v"(r5), "=v"(r6), "=v"(r7) : "v"((unsigned)(size_t)lp));
            a0 += r0.x + r1.x + r2.x + r3.x + r4.x + r5.x + r6.x + r7.x;
        } else if (KIND == 6) { // dependent DPP scan step pairs as in the affine scan (4 chains x 2 ops), one step
            asm volatile("s_nop 1\n v_fmac_f32_dpp %4, %4, %0 row_shr:1 row_mask:0xf bank_mask:0xf\n v_fmac_f32_dpp %5, %5, %1 row_shr:1 row_mask:0xf bank_mask:0xf\n"
                         "v_fmac_f32_dpp %6, %6, %2 row_shr:1 row_mask:0xf bank_mask:0xf\n v_fmac_f32_dpp %7, %7, %3 row_shr:1 row_mask:0xf bank_mask:0xf\n"
                         "v_mul_f32_dpp %0, %0, %0 row_shr:1 row_mask:0xf bank_mask:0xf\n v_mul_f32_dpp %1, %1, %1 row_shr:1 row_mask:0xf bank_mask:0xf\n"
                         "v_mul_f32_dpp %2, %2, %2 row_shr:1 row_mask:0xf bank_mask:0xf\n v_mul_f32_dpp %3, %3, %3 row_shr:1 row_mask:0xf bank_mask:0xf"
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));
        } else if (KIND == 7) { // 8 independent v_pk_mul_f32
            asm volatile("v_pk_mul_f32 %0, %0, %8\n v_pk_mul_f32 %1, %1, %8\n v_pk_mul_f32 %2, %2, %8\n v_pk_mul_f32 %3, %3, %8\n"
                         "v_pk_mul_f32 %4, %4, %8\n v_pk_mul_f32 %5, %5, %8\n v_pk_mul_f32 %6, %6, %8\n v_pk_mul_f32 %7, %7, %8"
                         : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3), "+v"(p4), "+v"(p5), "+v"(p6), "+v"(p7) : "v"(pb));
        } else if (KIND == 8) { // 8 independent v_rcp_f32
            asm volatile("v_rcp_f32 %0, %0\n v_rcp_f32 %1, %1\n v_rcp_f32 %2, %2\n v_rcp_f32 %3, %3\n v_rcp_f32 %4, %4\n v_rcp_f32 %5, %5\n v_rcp_f32 %6, %6\n v_rcp_f32 %7, %7"
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));
        } else if (KIND == 9) { // 4 x (ds_read_b128 x3 + ds_write_b128 x3) on a per-lane slot: the accumulator read-modify-write
            float4* sl = &lds[(threadIdx.x * 3) & 1020];
            float4 r0 = sl[0], r1 = sl[1], r2 = sl[2];
            r0.x += a0; r1.y += a1; r2.z += a2;
            sl[0] = r0; sl[1] = r1; sl[2] = r2;
            asm volatile("" ::: "memory");
        }
      }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    if (blockIdx.x == 0 && threadIdx.x == 0) cyc[0] = t1 - t0;
    out[blockIdx.x * blockDim.x + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 + p0.x + p1.y + p2.x + p3.y + p4.x + p5.y + p6.x + p7.y;
}

static unsigned long long* g_cyc;
static double g_clock;
template <int KIND>
static double run(int blocks_per_cu, float* out)
{
    const int cus = 256;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(k_issue<KIND>, dim3(cus * blocks_per_cu), dim3(256), 0, 0, out, 10, g_cyc);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL(k_issue<KIND>, dim3(cus * blocks_per_cu), dim3(256), 0, 0, out, ITER, g_cyc);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms = 0; hipEventElapsedTime(&ms, e0, e1);
    // each SIMD runs blocks_per_cu waves (a 256-thread block puts one wave on each of the CU's 4 SIMDs); 8 instructions per iteration
    unsigned long long c = 0; hipMemcpy(&c, g_cyc, 8, hipMemcpyDeviceToHost);
    g_clock = (double)c / ((double)ms * 1e-3) / 1e9;          // shader clock seen by wave 0 of block 0 (GHz), if s_memtime ticks at it
    (void)c;
    return (double)ms * 1e-3 * 1e9 / ((double)ITER * 64 * blocks_per_cu);   // ns per wave-instruction per SIMD (hipEvent time)   // cycles (at 2.4 GHz) per wave-instruction per SIMD
}

int main()
{
    float* out; hipMalloc(&out, 256 * 8 * 256 * sizeof(float)); hipMalloc(&g_cyc, 64);
    const char* names[] = {"v_fma_f32 x8 indep", "v_pk_fma_f32 x8 indep", "v_fmac_f32_dpp x8 indep", "v_exp_f32 x8", "v_fma_f32 x8 dependent",
                           "ds_read_b128 x8 (row bcast)+wait", "affine scan step (8 dpp)", "v_pk_mul_f32 x8", "v_rcp_f32 x8", "lds rmw 3xb128 (per 8)"};
    printf("ns per wave-instruction per SIMD, hipEvent time, 64 instructions per loop iteration (lower = faster); columns: waves per SIMD 1 2 3 4 8\n");
    for (int k = 0; k < 10; ++k) {
        printf("%-36s", names[k]);
        for (int w : {1, 2, 3, 4, 8}) {
            double c = 0;
            switch (k) {
                case 0: c = run<0>(w, out); break; case 1: c = run<1>(w, out); break; case 2: c = run<2>(w, out); break;
                case 3: c = run<3>(w, out); break; case 4: c = run<4>(w, out); break; case 5: c = run<5>(w, out); break;
                case 6: c = run<6>(w, out); break; case 7: c = run<7>(w, out); break; case 8: c = run<8>(w, out); break;
                case 9: c = run<9>(w, out); break;
            }
            printf(" %7.3f", c);
        }
        printf("   (memtime GHz %.2f)\n", g_clock);
    }
    return 0;
}
