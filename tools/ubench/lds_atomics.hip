// Microbenchmark (diagnostic): LDS integer atomics with and without a returned value, random addresses, 1024-thread workgroups,
// followed by a scattered 4-byte global store (the access pattern of emit_binned_kernel).  Self-contained: every access in bounds.
#include <hip/hip_runtime.h>
#include <stdio.h>
#define T 8160
template <int MODE>   // 0: non-returning atomic, store to own slot; 1: returning atomic, store at returned slot; 2: as 1 without the global store
__global__ void __launch_bounds__(1024) k(unsigned* out, int per_thread, unsigned n_out)
{
    __shared__ unsigned cur[T];
    for (int t = threadIdx.x; t < T; t += 1024) cur[t] = (unsigned)(((size_t)t * n_out) / T);     // spread bases over the buffer
    __syncthreads();
    unsigned x = blockIdx.x * 1024u + threadIdx.x + 1u;
    unsigned acc = 0;
    for (int k2 = 0; k2 < per_thread; ++k2) {
        x = x * 1664525u + 1013904223u;
        const unsigned t = (x >> 8) % T;
        if (MODE == 0) { atomicAdd(&cur[t], 0u); out[(blockIdx.x * 1024u + threadIdx.x) * 16u % n_out] = x; }
        else {
            const unsigned s = atomicAdd(&cur[t], 1u) % n_out;
            if (MODE == 1) out[s] = x; else acc += s;
        }
    }
    if (acc == 0xFFFFFFFFu) out[0] = acc;
}
template <int MODE> static void run(unsigned* out, unsigned n_out, const char* name)
{
    hipEvent_t a, b; (void)hipEventCreate(&a); (void)hipEventCreate(&b);
    hipLaunchKernelGGL(k<MODE>, dim3(245), dim3(1024), 0, 0, out, 10, n_out); (void)hipDeviceSynchronize();
    (void)hipEventRecord(a);
    for (int i = 0; i < 20; ++i) hipLaunchKernelGGL(k<MODE>, dim3(245), dim3(1024), 0, 0, out, 10, n_out);
    (void)hipEventRecord(b); (void)hipEventSynchronize(b);
    float ms = 0; (void)hipEventElapsedTime(&ms, a, b);
    printf("%-48s %.1f us per launch (245 x 1024 threads x 10 atomics)\n", name, ms * 1000.f / 20.f);
}
int main()
{
    const unsigned n_out = 2500000; unsigned* out; (void)hipMalloc(&out, n_out * 4);
    run<0>(out, n_out, "non-returning atomic + strided store");
    run<1>(out, n_out, "returning atomic + store at the returned slot");
    run<2>(out, n_out, "returning atomic, no global store");
    return 0;
}
