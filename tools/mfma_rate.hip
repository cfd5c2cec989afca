// Micro-benchmark: sustained issue rate of the f32-input MFMAs on gfx950 (one wave per SIMD, operands in registers,
// optionally with LDS operand reads interleaved).  Build: hipcc --offload-arch=gfx950 -O3 tools/mfma_rate.hip -o /tmp/mfma_rate
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int NACC, bool LDS>
__global__ __launch_bounds__(256) void k16(float *out, int iters) {
    __shared__ float lds[4096];
    for (int i = threadIdx.x; i < 4096; i += 256) lds[i] = 0.001f * i;
    __syncthreads();
    f32x4 acc[NACC];
    for (int i = 0; i < NACC; ++i) acc[i] = f32x4{0, 0, 0, 0};
    float a = threadIdx.x * 0.01f, b = 1.0f + threadIdx.x * 0.001f;
    const int lane = threadIdx.x & 63;
    for (int it = 0; it < iters; ++it) {
        float av[2], bv[5];
        if (LDS) {
#pragma unroll
            for (int r = 0; r < 2; ++r) av[r] = lds[((it * 7 + r) * 64 + lane) & 4095];
#pragma unroll
            for (int c = 0; c < 5; ++c) bv[c] = lds[((it * 7 + 2 + c) * 64 + lane) & 4095];
        } else {
#pragma unroll
            for (int r = 0; r < 2; ++r) av[r] = a + r;
#pragma unroll
            for (int c = 0; c < 5; ++c) bv[c] = b + c;
        }
#pragma unroll
        for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[i % 2], bv[i % 5], acc[i], 0, 0, 0);
    }
    float s = 0;
    for (int i = 0; i < NACC; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}

// LDS operands prefetched ONE iteration ahead with b128 reads (4 k-steps per read), schedule pinned by sched_barrier
template <int MODE>
__global__ __launch_bounds__(256) void k16p(float *out, int iters) {
    __shared__ __align__(16) float lds[8192];
    for (int i = threadIdx.x; i < 8192; i += 256) lds[i] = 0.001f * i;
    __syncthreads();
    f32x4 acc[10];
    for (int i = 0; i < 10; ++i) acc[i] = f32x4{0, 0, 0, 0};
    const int lane = threadIdx.x & 63;
    const float4 *l4 = reinterpret_cast<const float4 *>(lds);
    float4 a[2][2], b[2][5];
    const int lrow = lane & 15, lk = lane >> 4;
    auto fetch = [&](int it, int slot) {
        if (MODE == 3) {       // the conv kernel's image: rows of 16 k (pitch 20 floats), lane -> (row lrow, slot lk)
#pragma unroll
            for (int r = 0; r < 2; ++r) a[slot][r] = *reinterpret_cast<const float4 *>(lds + (((it & 7) * 32 + r * 16 + lrow) * 20 + lk * 4));
#pragma unroll
            for (int c = 0; c < 5; ++c) b[slot][c] = *reinterpret_cast<const float4 *>(lds + ((256 + (it & 1) + c * 16 + lrow) * 20 + lk * 4));
        } else {
#pragma unroll
            for (int r = 0; r < 2; ++r) a[slot][r] = l4[((it * 7 + r) * 64 + lane) & 2047];
#pragma unroll
            for (int c = 0; c < 5; ++c) b[slot][c] = l4[((it * 7 + 2 + c) * 64 + lane) & 2047];
        }
    };
    fetch(0, 0);
    for (int it = 0; it < iters; it += 2) {
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            fetch(it + h + 1, (h + 1) & 1);
            if (MODE >= 1) __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int i = 0; i < 10; ++i) {
                    if (MODE == 2)      // same A operand for 5 consecutive MFMAs (r-major tile order)
                        acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32((&a[h][i / 5].x)[j], (&b[h][i % 5].x)[j], acc[i], 0, 0, 0);
                    else
                        acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32((&a[h][i % 2].x)[j], (&b[h][i % 5].x)[j], acc[i], 0, 0, 0);
                }
            if (MODE >= 1) __builtin_amdgcn_sched_barrier(0);
        }
    }
    float s = 0;
    for (int i = 0; i < 10; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}

template <int NACC>
__global__ __launch_bounds__(256) void k32(float *out, int iters) {
    f32x16 acc[NACC];
    for (int i = 0; i < NACC; ++i)
        for (int j = 0; j < 16; ++j) acc[i][j] = 0;
    float a = threadIdx.x * 0.01f, b = 1.0f + threadIdx.x * 0.001f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(a + i, b, acc[i], 0, 0, 0);
    }
    float s = 0;
    for (int i = 0; i < NACC; ++i)
        for (int j = 0; j < 16; ++j) s += acc[i][j];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}

template <typename F>
void run(const char *name, F launch, double flops_per_block_iter, int blocks, int iters) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    launch(); hipDeviceSynchronize();
    hipEventRecord(e0);
    for (int r = 0; r < 5; ++r) launch();
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1); ms /= 5;
    double tf = flops_per_block_iter * blocks * iters / (ms * 1e-3) / 1e12;
    printf("%-34s %8.3f ms  %7.1f TFLOP/s\n", name, ms, tf);
}

int main() {
    float *out; hipMalloc(&out, 1024 * 256 * 4);
    const int iters = 20000;
    for (int blocks : {256, 512}) {
        printf("blocks=%d (x4 waves)\n", blocks);
        run("16x16x4 10 acc regs-only", [&] { hipLaunchKernelGGL((k16<10, false>), dim3(blocks), dim3(256), 0, 0, out, iters); }, 4.0 * 10 * 2048, blocks, iters);
        run("16x16x4 10 acc + 7 ds_read/iter", [&] { hipLaunchKernelGGL((k16<10, true>), dim3(blocks), dim3(256), 0, 0, out, iters); }, 4.0 * 10 * 2048, blocks, iters);
        run("16x16x4 b128 prefetch (compiler)", [&] { hipLaunchKernelGGL((k16p<0>), dim3(blocks), dim3(256), 0, 0, out, iters / 4); }, 4.0 * 40 * 2048, blocks, iters / 4);
        run("16x16x4 b128 prefetch (pinned)", [&] { hipLaunchKernelGGL((k16p<1>), dim3(blocks), dim3(256), 0, 0, out, iters / 4); }, 4.0 * 40 * 2048, blocks, iters / 4);
        run("16x16x4 b128 pinned, A reused x5", [&] { hipLaunchKernelGGL((k16p<2>), dim3(blocks), dim3(256), 0, 0, out, iters / 4); }, 4.0 * 40 * 2048, blocks, iters / 4);
        run("16x16x4 b128 pinned, pitch-20 image", [&] { hipLaunchKernelGGL((k16p<3>), dim3(blocks), dim3(256), 0, 0, out, iters / 4); }, 4.0 * 40 * 2048, blocks, iters / 4);
        run("16x16x4 4 acc regs-only", [&] { hipLaunchKernelGGL((k16<4, false>), dim3(blocks), dim3(256), 0, 0, out, iters); }, 4.0 * 4 * 2048, blocks, iters);
        run("16x16x4 20 acc regs-only", [&] { hipLaunchKernelGGL((k16<20, false>), dim3(blocks), dim3(256), 0, 0, out, iters); }, 4.0 * 20 * 2048, blocks, iters);
        run("32x32x2 4 acc regs-only", [&] { hipLaunchKernelGGL((k32<4>), dim3(blocks), dim3(256), 0, 0, out, iters); }, 4.0 * 4 * 4096, blocks, iters);
    }
    return 0;
}
