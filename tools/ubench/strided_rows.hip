// Microbenchmark (diagnostic, not product): how fast can a kernel read 192-byte rows (the SH rows of preprocess_fwd) when
//   0  every lane reads ITS OWN row with twelve 16-byte loads at a 192-byte stride (what preprocess_fwd does),
//   1  the wave reads its 64 rows (12 KB, contiguous) lane-linearly, twelve fully coalesced 1-KB instructions,
//   2  like 0, and every lane also writes 104 bytes (a 64-byte line + 40 bytes, preprocess_fwd's output per Gaussian),
//   3  like 1 with the same writes,
// at 5 waves per SIMD, P = 500 000 rows (96 MB)?  Prints GB/s of bytes moved.
// Build + run on the GPU box: hipcc --offload-arch=gfx950 -O3 strided_rows.hip -o /tmp/strided_rows && /tmp/strided_rows
#include <hip/hip_runtime.h>
#include <stdio.h>

template <int KIND>
__global__ void __launch_bounds__(256, 5) k_rows(int P, const float4* __restrict__ rows, float* __restrict__ out, float4* __restrict__ lines,
                                                 float* __restrict__ jac)
{
    const int i = blockIdx.x * 256 + threadIdx.x;
    const int lane = threadIdx.x & 63;
    const int ic = i < P ? i : P - 1;
    float4 v[12];
    if (KIND == 0 || KIND == 2) {
#pragma unroll
        for (int k = 0; k < 12; ++k) v[k] = rows[(size_t)ic * 12 + k];
    } else {
        const size_t wave_base = (size_t)(ic - lane) * 12;       // the wave's 64 rows = 768 float4
#pragma unroll
        for (int k = 0; k < 12; ++k) v[k] = rows[min(wave_base + (size_t)k * 64 + lane, (size_t)P * 12 - 1)];
    }
    float s = 0.f;
#pragma unroll
    for (int k = 0; k < 12; ++k) s += v[k].x * 1.0001f + v[k].y + v[k].z + v[k].w;
    if (i < P) {
        out[i] = s;
        if (KIND >= 2) {
#pragma unroll
            for (int k = 0; k < 4; ++k) lines[(size_t)i * 4 + k] = make_float4(s, v[k].x, v[k].y, v[k].z);
#pragma unroll
            for (int k = 0; k < 10; ++k) jac[(size_t)i * 10 + k] = v[k].w + s;
        }
    }
}

template <int KIND>
static void run(int P, const float4* rows, float* out, float4* lines, float* jac, const char* name)
{
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    const int nb = (P + 255) / 256;
    for (int w = 0; w < 5; ++w) hipLaunchKernelGGL((k_rows<KIND>), dim3(nb), dim3(256), 0, 0, P, rows, out, lines, jac);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    const int reps = 50;
    for (int r = 0; r < reps; ++r) hipLaunchKernelGGL((k_rows<KIND>), dim3(nb), dim3(256), 0, 0, P, rows, out, lines, jac);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms = 0.f;
    hipEventElapsedTime(&ms, e0, e1);
    const double us = ms * 1e3 / reps;
    const double bytes = (double)P * (192 + 4 + (KIND >= 2 ? 104 : 0));
    printf("%-58s %7.1f us  %6.0f GB/s\n", name, us, bytes / us * 1e-3);
}

int main()
{
    const int P = 500000;
    float4* rows; float* out; float4* lines; float* jac;
    hipMalloc(&rows, (size_t)P * 192); hipMalloc(&out, (size_t)P * 4); hipMalloc(&lines, (size_t)P * 64); hipMalloc(&jac, (size_t)P * 40);
    hipMemset(rows, 0, (size_t)P * 192);
    // something large in between launches so that the rows do not stay in the 256 MB Infinity Cache: the bench step touches ~700 MB
    float* big; hipMalloc(&big, (size_t)1024 << 20);
    run<0>(P, rows, out, lines, jac, "own row per lane, 12 x 16 B at a 192-B stride (cached run)");
    run<1>(P, rows, out, lines, jac, "wave reads its 12 KB lane-linearly (cached run)");
    run<2>(P, rows, out, lines, jac, "own row per lane + 104 B written per lane (cached run)");
    run<3>(P, rows, out, lines, jac, "lane-linear + 104 B written per lane (cached run)");
    // From cold caches.  The 256 MB Infinity Cache is flushed with a READ of 600 MB (clean lines); optionally a predecessor
    // then leaves `dirty_mb` of freshly written lines behind (what preprocess_bwd's gradient rows are to the next step's
    // preprocess_fwd): their write-back is paid by whoever evicts them.
    float4* flush; hipMalloc(&flush, (size_t)600 << 20);
    hipMemset(flush, 0, (size_t)600 << 20);
    const int PF = (600 << 20) / 192;
    for (int dirty_mb = 0; dirty_mb <= 260; dirty_mb += 130)
        for (int kind = 0; kind < 4; kind += 2) {
            double tot = 0;
            hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
            const int nb = (P + 255) / 256;
            for (int r = 0; r < 8; ++r) {
                hipLaunchKernelGGL((k_rows<1>), dim3((PF + 255) / 256), dim3(256), 0, 0, PF, flush, (float*)big, lines, jac);
                if (dirty_mb) hipMemsetAsync(big + (64 << 20), r, (size_t)dirty_mb << 20, 0);
                hipEventRecord(e0);
                if (kind == 0) hipLaunchKernelGGL((k_rows<0>), dim3(nb), dim3(256), 0, 0, P, rows, out, lines, jac);
                if (kind == 2) hipLaunchKernelGGL((k_rows<2>), dim3(nb), dim3(256), 0, 0, P, rows, out, lines, jac);
                hipEventRecord(e1); hipEventSynchronize(e1);
                float ms; hipEventElapsedTime(&ms, e0, e1);
                if (r >= 2) tot += ms * 1e3;
            }
            const double us = tot / 6;
            const double bytes = (double)P * (192 + 4 + (kind >= 2 ? 104 : 0));
            printf("cold, %3d MB of dirty lines left by the predecessor, %-28s %7.1f us  %6.0f GB/s of its own bytes\n", dirty_mb,
                   kind == 0 ? "own row per lane, read only:" : "own row per lane + 104 B out:", us, bytes / us * 1e-3);
        }
    return 0;
}
