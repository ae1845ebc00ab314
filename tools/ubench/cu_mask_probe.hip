// Diagnostic, not product: which physical CUs does a stream created with hipExtStreamCreateWithCUMask run on?  For each mask the
// kernel's workgroups record (XCC id, SE id, CU id) from the hardware registers; the host prints how many distinct CUs of every XCD
// were used.  Answers how mask bit i maps to (XCD, CU) on this part, which decides how a "one partition = k XCDs" mask is spelled.
// Build + run on the GPU box: hipcc --offload-arch=gfx950 -O2 tools/ubench/cu_mask_probe.hip -o /tmp/cu_mask_probe && /tmp/cu_mask_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <string.h>
#include <vector>

__global__ void __launch_bounds__(64) probe(unsigned* hist, int spin)
{
    unsigned xcc, hw;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
    const unsigned x = xcc & 0xF, cu = (hw >> 8) & 0xF, sh = (hw >> 12) & 1, se = (hw >> 13) & 7;
    // keep the workgroup alive a little so that the dispatcher has to spread the grid over every CU it may use
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
    while (__builtin_amdgcn_s_memtime() - t0 < (unsigned long long)spin) {}
    if (threadIdx.x == 0) atomicAdd(&hist[((x * 8 + se) * 2 + sh) * 16 + cu], 1u);
}

static void run(const char* name, const std::vector<unsigned>& mask, unsigned* d_hist)
{
    hipStream_t st;
    hipError_t e = hipExtStreamCreateWithCUMask(&st, (unsigned)mask.size(), mask.data());
    if (e != hipSuccess) { printf("%s: hipExtStreamCreateWithCUMask failed: %s\n", name, hipGetErrorString(e)); return; }
    const int N = 16 * 8 * 2 * 16;
    hipMemsetAsync(d_hist, 0, N * 4, st);
    hipLaunchKernelGGL(probe, dim3(8192), dim3(64), 0, st, d_hist, 20000);
    std::vector<unsigned> h(N);
    hipMemcpyAsync(h.data(), d_hist, N * 4, hipMemcpyDeviceToHost, st);
    hipStreamSynchronize(st);
    printf("%-28s", name);
    int total = 0;
    for (int x = 0; x < 16; ++x) {
        int cus = 0;
        for (int i = 0; i < 8 * 2 * 16; ++i) cus += h[x * 256 + i] ? 1 : 0;
        if (x < 8 || cus) printf(" xcc%d:%2d", x, cus);
        total += cus;
    }
    printf("  | CUs used %d\n", total);
    hipStreamDestroy(st);
}

int main()
{
    hipDeviceProp_t p; hipGetDeviceProperties(&p, 0);
    printf("%s: %d CUs\n", p.name, p.multiProcessorCount);
    unsigned* d_hist; hipMalloc(&d_hist, 16 * 8 * 2 * 16 * 4);
    const int W = 8;                                       // 256 bits
    auto bits = [&](auto pred) { std::vector<unsigned> m(W, 0u); for (int i = 0; i < 256; ++i) if (pred(i)) m[i >> 5] |= 1u << (i & 31); return m; };
    run("all 256", bits([](int) { return true; }), d_hist);
    run("bits 0..31", bits([](int i) { return i < 32; }), d_hist);
    run("bits 0..127", bits([](int i) { return i < 128; }), d_hist);
    run("bits 128..255", bits([](int i) { return i >= 128; }), d_hist);
    run("bits i%8==0", bits([](int i) { return i % 8 == 0; }), d_hist);
    run("bits i%8<4", bits([](int i) { return i % 8 < 4; }), d_hist);
    run("bits i%8>=4", bits([](int i) { return i % 8 >= 4; }), d_hist);
    run("bits i%2==0", bits([](int i) { return i % 2 == 0; }), d_hist);
    run("bits (i/8)%2==0", bits([](int i) { return (i / 8) % 2 == 0; }), d_hist);
    run("bits 0..63", bits([](int i) { return i < 64; }), d_hist);
    run("bits i%8<2", bits([](int i) { return i % 8 < 2; }), d_hist);
    return 0;
}
