// How fast does the chip START workgroups?  An almost empty kernel records each workgroup's start time (100 MHz wall clock); the
// spread between the first and the last start of a one-round grid is the dispatch skew every launch of the conv kernels pays.
// Variants: threads per workgroup, dynamic LDS per workgroup, registers per thread (forced by launch bounds + a live array).
//   hipcc --offload-arch=gfx950 -O3 -o tools/dispatch_rate.bin tools/dispatch_rate.hip && tools/dispatch_rate.bin
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <vector>

template <int REGS>
__global__ void probe(unsigned long long *out, int spin) {
    extern __shared__ float lds[];
    float keep[REGS];
#pragma unroll
    for (int i = 0; i < REGS; ++i) keep[i] = (float)(threadIdx.x + i);
    unsigned long long t0 = wall_clock64();
    for (int k = 0; k < spin; ++k)
#pragma unroll
        for (int i = 0; i < REGS; ++i) keep[i] = keep[i] * 1.0001f + 0.5f;
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < REGS; ++i) s += keep[i];
    if (threadIdx.x == 0) {
        out[blockIdx.x * 2] = t0;
        out[blockIdx.x * 2 + 1] = wall_clock64();
        if (s == 12345.678f) lds[0] = s;
    }
}

template <int REGS>
void run(const char *name, int wgs, int threads, size_t lds, int spin) {
    unsigned long long *d;
    hipMalloc(&d, wgs * 16);
    hipFuncSetAttribute(reinterpret_cast<const void *>(&probe<REGS>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    std::vector<unsigned long long> h(wgs * 2);
    double best = 1e9, bestlife = 0;
    for (int it = 0; it < 5; ++it) {
        hipLaunchKernelGGL(probe<REGS>, dim3(wgs), dim3(threads), lds, 0, d, spin);
        hipDeviceSynchronize();
        hipMemcpy(h.data(), d, wgs * 16, hipMemcpyDeviceToHost);
        unsigned long long lo = ~0ull, hi = 0, endmax = 0;
        for (int i = 0; i < wgs; ++i) { lo = std::min(lo, h[2 * i]); hi = std::max(hi, h[2 * i]); endmax = std::max(endmax, h[2 * i + 1]); }
        const double skew = (hi - lo) / 100.0;
        if (skew < best) { best = skew; bestlife = (endmax - lo) / 100.0; }
    }
    printf("%-44s %4d workgroups x %4d threads, %6zu B LDS: last start %6.2f us after the first (best of 5), kernel %6.2f us\n", name,
           wgs, threads, lds, best, bestlife);
    hipFree(d);
}

int main() {
    run<4>("few registers, no LDS", 480, 256, 0, 2000);
    run<4>("few registers, 69 KB LDS", 480, 256, 69 * 1024, 2000);
    run<96>("~128 registers, no LDS", 480, 256, 0, 100);
    run<96>("~128 registers, 69 KB LDS", 480, 256, 69 * 1024, 100);
    run<200>("~256 registers, 69 KB LDS", 480, 256, 69 * 1024, 50);
    run<4>("few registers, 512 threads, 117 KB LDS", 252, 512, 117 * 1024, 2000);
    run<96>("~128 registers, 512 threads, 117 KB LDS", 252, 512, 117 * 1024, 100);
    run<4>("few registers, 64 threads, no LDS", 1920, 64, 0, 2000);
    run<4>("few registers, 1024 threads, no LDS", 256, 1024, 0, 2000);
    return 0;
}
