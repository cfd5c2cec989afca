// Micro-benchmark for DESIGN.md §8-0: the MFMA loop of a Winograd F(4, 5) form of the gated in-conv (M = 384 rows, 192 channels,
// 32 x 400 frames = 3 200 Winograd tiles) WITHOUT its staging — can ONE wave per SIMD with a 2 row-tile x CB column-block x 8 point
// register tile keep the matrix pipe busy while it streams the transformed weights from L2 (register ring) and reads the
// transformed input from LDS?  Operands are random bf16 planes; results are summed into a checksum only.
//   hipcc --offload-arch=gfx950 -O3 tools/wino_skeleton.hip -o /tmp/wino_skeleton && /tmp/wino_skeleton
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef int i32x4 __attribute__((ext_vector_type(4)));
typedef int i32x2 __attribute__((ext_vector_type(2)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

__host__ __device__ constexpr int product_a(int k) { return k == 0 ? 0 : k == 1 ? 2 : k == 2 ? 1 : k == 3 ? 0 : k == 4 ? 1 : 0; }
__host__ __device__ constexpr int product_b(int k) { return k == 0 ? 2 : k == 1 ? 0 : k == 2 ? 1 : k == 3 ? 1 : k == 4 ? 0 : 0; }

constexpr int RP = 40;      // bf16 per LDS row (32 channels + 8 pad): the conflict-free 80-byte pitch of convgemm_split.hip

template <int CB, int RING, int RTW, int NV = 0, int KIND = 0>
__global__ __launch_bounds__(256, 1) void wino_skel(const unsigned short *__restrict__ U, long plane_stride, int M, int G, int nks,
                                                    float *out, int sync_per_ks, int waves_rows) {
    extern __shared__ __align__(16) float smem[];
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, lrow = lane & 15, lk = lane >> 4;
    constexpr int COLS = 16 * CB;
    constexpr int IMG = 3 * 8 * COLS * RP / 2;          // dwords
    for (int i = tid; i < IMG; i += 256) smem[i] = __uint_as_float(0x3c003c00u + ((unsigned)(i * 2654435761u) >> 20) * 0x00010001u);
    __syncthreads();
    const int tile_m = blockIdx.y;
    f32x4 acc[RTW][8][CB];
#pragma unroll
    for (int r = 0; r < RTW; ++r)
#pragma unroll
        for (int p = 0; p < 8; ++p)
#pragma unroll
            for (int c = 0; c < CB; ++c) acc[r][p][c] = f32x4{0.f, 0.f, 0.f, 0.f};
    const int wbytes = 8 * G * M * 32;
    __amdgpu_buffer_rsrc_t wrs[3];
#pragma unroll
    for (int pl = 0; pl < 3; ++pl)
        wrs[pl] = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned short *>(U + pl * plane_stride), 0, wbytes, 0x00020000);
    int wvo[RTW];
#pragma unroll
    for (int r = 0; r < RTW; ++r) wvo[r] = ((tile_m * (64 * RTW) + (wave * RTW + r) * 16 + lrow) * 16 + lk * 4) * 2;
    const int wtap = G * M * 32, wgrp = M * 32;
    i32x4 a[RING][RTW][3];
    auto wload = [&](int q, int slot) {
        const int ks = (q >> 3) % (G / 2), p = q & 7, g = 2 * ks;
        if (waves_rows == 0) {                           // the packing of convgemm_split.hip: two 8-byte loads (groups g, g + 1)
            const int so0 = g < G ? p * wtap + g * wgrp : wbytes;
            const int so1 = g + 1 < G ? p * wtap + (g + 1) * wgrp : wbytes;
#pragma unroll
            for (int r = 0; r < RTW; ++r)
#pragma unroll
                for (int pl = 0; pl < 3; ++pl) {
                    const i32x2 lo = __builtin_bit_cast(i32x2, __builtin_amdgcn_raw_buffer_load_b64(wrs[pl], wvo[r], so0, 0));
                    const i32x2 hi = __builtin_bit_cast(i32x2, __builtin_amdgcn_raw_buffer_load_b64(wrs[pl], wvo[r], so1, 0));
                    a[slot][r][pl] = i32x4{lo[0], lo[1], hi[0], hi[1]};
                }
        } else {                                         // [point][k-step][row][32 channels]: ONE 16-byte load per fragment
            const int so = g < G ? (p * (G / 2) + ks) * M * 64 : wbytes;
#pragma unroll
            for (int r = 0; r < RTW; ++r)
#pragma unroll
                for (int pl = 0; pl < 3; ++pl)
                    a[slot][r][pl] = __builtin_bit_cast(i32x4, __builtin_amdgcn_raw_buffer_load_b128(wrs[pl], wvo[r] * 2, so, 0));
        }
    };
    const float *xd = smem + lrow * (RP / 2) + lk * 4;
    float dummy[8] = {1.f, 2.f, 3.f, 4.f, 5.f, 6.f, 7.f, 8.f}, dscale = 0.999f + 1e-6f * lane;
    i32x4 bv[3][3];
    auto bfetch = [&](int p, int cb, int slot) {
#pragma unroll
        for (int pl = 0; pl < 3; ++pl)
            bv[slot][pl] = *reinterpret_cast<const i32x4 *>(xd + ((pl * 8 + p) * COLS + cb * 16) * (RP / 2));
    };
#pragma unroll
    for (int i = 0; i < RING - 1; ++i) wload(i, i);
    for (int ks = 0; ks < nks; ++ks) {
        bfetch(0, 0, 0);
        bfetch(0, 1 % CB, 1);
#pragma unroll
        for (int p = 0; p < 8; ++p) {
            wload(ks * 8 + p + RING - 1, (p + RING - 1) % RING);
#pragma unroll
            for (int cb = 0; cb < CB; ++cb) {
                const int n = p * CB + cb;
                if (n + 2 < 8 * CB) bfetch((n + 2) / CB, (n + 2) % CB, (n + 2) % 3);     // TWO groups ahead
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int k = 0; k < 6; ++k)
#pragma unroll
                    for (int r = 0; r < RTW; ++r)
                        acc[r][p][cb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(
                            __builtin_bit_cast(bf16x8, a[p % RING][r][product_a(k)]), __builtin_bit_cast(bf16x8, bv[n % 3][product_b(k)]),
                            acc[r][p][cb], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        if (sync_per_ks) __syncthreads();
    }
    f32x4 s = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int r = 0; r < RTW; ++r)
#pragma unroll
        for (int p = 0; p < 8; ++p)
#pragma unroll
            for (int c = 0; c < CB; ++c) s += acc[r][p][c];
    out[(blockIdx.y * gridDim.x + blockIdx.x) * 256 + tid] = s[0] + s[1] + s[2] + s[3] + dummy[0] + dummy[1] + dummy[2] + dummy[3] + dummy[4] + dummy[5] + dummy[6] + dummy[7];
}

template <int CB, int RING, int RTW, int NV = 0, int KIND = 0>
static void run(const unsigned short *U, long stride, int M, int G, float *out, int sync, const char *what, int nks_mul = 1) {
    const int nks = G / 2 * nks_mul, tiles = 3200;
    dim3 grid((tiles + 16 * CB - 1) / (16 * CB), M / (64 * RTW));
    const size_t lds = (size_t)3 * 8 * 16 * CB * RP * 2;
    CHECK(hipFuncSetAttribute(reinterpret_cast<const void *>(&wino_skel<CB, RING, RTW, NV, KIND>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    for (int i = 0; i < 5; ++i) hipLaunchKernelGGL((wino_skel<CB, RING, RTW, NV, KIND>), grid, dim3(256), lds, 0, U, stride, M, G, nks, out, sync & 1, sync >> 1);
    CHECK(hipDeviceSynchronize());
    const int n = 200;
    CHECK(hipEventRecord(e0));
    for (int i = 0; i < n; ++i) hipLaunchKernelGGL((wino_skel<CB, RING, RTW, NV, KIND>), grid, dim3(256), lds, 0, U, stride, M, G, nks, out, sync & 1, sync >> 1);
    CHECK(hipEventRecord(e1));
    CHECK(hipEventSynchronize(e1));
    float ms = 0.f;
    CHECK(hipEventElapsedTime(&ms, e0, e1));
    const double us = ms * 1e3 / n;
    const double mfma_per_wave = (double)RTW * CB * 8 * 6 * nks;
    const double cyc = mfma_per_wave * 16;
    printf("%-34s grid %3d x %d = %3d workgroups, %5.0f MFMAs per wave = %5.1f us at 1.85 GHz; LDS %3zu KB: %6.2f us per launch (back to back)\n", what,
           grid.x, grid.y, grid.x * grid.y, mfma_per_wave, cyc / 1850.0, lds / 1024, us);
}

int main() {
    const int M = 384, G = 12;
    const long stride = (long)8 * G * M * 16;            // bf16 elements per plane
    std::vector<unsigned short> h(3 * stride);
    unsigned s = 12345u;
    for (auto &v : h) { s = s * 1664525u + 1013904223u; v = (unsigned short)(0x3c00u + ((s >> 16) & 0x1ffu) + ((s >> 31) << 15)); }
    unsigned short *U; float *out;
    CHECK(hipMalloc(&U, h.size() * 2)); CHECK(hipMalloc(&out, 1 << 22));
    CHECK(hipMemcpy(U, h.data(), h.size() * 2, hipMemcpyHostToDevice));
    run<3, 4, 2>(U, stride, M, G, out, 3, "6 k-steps");
    run<3, 4, 2>(U, stride, M, G, out, 3, "12 k-steps", 2);
    run<3, 4, 2>(U, stride, M, G, out, 3, "24 k-steps", 4);
    run<3, 4, 2>(U, stride, M, G, out, 2, "24 k-steps, no barrier", 4);
    run<3, 8, 2>(U, stride, M, G, out, 3, "24 k-steps, ring 8", 4);
    run<4, 4, 2>(U, stride, M, G, out, 3, "CB 4: 6 k-steps");
    run<4, 4, 2>(U, stride, M, G, out, 3, "CB 4: 24 k-steps", 4);
    return 0;
}
