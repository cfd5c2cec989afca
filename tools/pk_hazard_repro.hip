// tools/pk_hazard_repro.hip — minimal two-kernel reproducer for DESIGN.md lesson 12.
//
// Question: were actnorm_invconv_bwd's wrong accumulators (lanes 48-63 of the low register of a v_pk_add_f32 pair, only next
// to the bf16-MFMA weight-gradient kernel) caused by the AGGRESSOR touching memory / LDS / registers it does not own, or by
// the hardware executing packed-fp32 VALU math wrongly beside another wave's MFMAs?
//
// This program has NO shared state between its two kernels:
//   victim    : the loop shape of actnorm_invconv_bwd — a divergent `while (have)` with two items in flight per thread, 24
//               accumulators per thread — on INTEGER-valued floats, so every sum is exact in fp32 and the expected value of
//               every accumulator of every thread is known bit for bit (no atomics, no rounding, no ordering).
//               PK = true keeps the accumulators as <2 x float> (v_pk_mul_f32 / v_pk_add_f32); PK = false as scalars.
//   aggressors: loops that live in registers only — no LDS allocation, no global store except one guarded word at the end:
//               bf16 MFMA (v_mfma_f32_16x16x32_bf16), fp32 MFMA (v_mfma_f32_16x16x4_f32), packed-fp32 VALU only, and
//               a bf16-MFMA loop that also reads LDS.
// The victim runs N times on one stream while an aggressor loops on another; a checker kernel counts, per launch, the
// accumulators that differ from the exact expectation, with histograms by lane / wave / accumulator index.
//
//   hipcc -O3 --offload-arch=gfx950 -ffp-contract=off [-fno-slp-vectorize] tools/pk_hazard_repro.hip -o tools/pk_hazard_repro.bin
//   tools/pk_hazard_repro.bin [launches=600] [aggressor_blocks=288]
// As a shared library for tools/pk_hazard_repro.py (a LIBRARY bf16 GEMM as the aggressor, CU-masked streams, float data
// compared bit for bit with the victim's own result when it runs alone):
//   hipcc -O3 --offload-arch=gfx950 -ffp-contract=off -fno-slp-vectorize -shared -fPIC -DPK_AS_LIBRARY tools/pk_hazard_repro.hip \
//         -o tools/libpk_hazard_repro.bin
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

#define CK(x)                                                                                   \
    do {                                                                                        \
        hipError_t e_ = (x);                                                                    \
        if (e_ != hipSuccess) {                                                                 \
            fprintf(stderr, "%s:%d %s: %s\n", __FILE__, __LINE__, #x, hipGetErrorString(e_));  \
            exit(2);                                                                            \
        }                                                                                       \
    } while (0)

constexpr int NACC = 24;     // 16 matrix sums + 4 + 4 channel sums, as actnorm_invconv_bwd<4, 4>

// ---------------------------------------------------------------------------------------------------------- victim
// x, g: [items][4] float4 (4 channels x 4 frames), m: [items] float4; all values small integers
template <bool PK>
__global__ __launch_bounds__(256) void victim_kernel(const float4 *__restrict__ x, const float4 *__restrict__ g,
                                                     const float4 *__restrict__ m, float *__restrict__ out, int n_items,
                                                     int nb) {
    const int i0 = blockIdx.x * nb, i1 = min(n_items, i0 + nb);
    float aw[16], al[4], ab[4];
    f32x2 pw[8], pl[2], pb[2];
#pragma unroll
    for (int q = 0; q < 16; ++q) aw[q] = 0.f;
#pragma unroll
    for (int q = 0; q < 8; ++q) pw[q] = f32x2{0.f, 0.f};
#pragma unroll
    for (int k = 0; k < 4; ++k) { al[k] = 0.f; ab[k] = 0.f; }
    pl[0] = pl[1] = pb[0] = pb[1] = f32x2{0.f, 0.f};
    float4 mv, xv[4], gz[4];
    auto fetch = [&](int it, float4 &m_, float4 *x_, float4 *g_) {
        m_ = m[it];
#pragma unroll
        for (int k = 0; k < 4; ++k) { x_[k] = x[(long)it * 4 + k]; g_[k] = g[(long)it * 4 + k]; }
    };
    int it = i0 + threadIdx.x;
    bool have = it < i1;
    if (have) fetch(it, mv, xv, gz);
    while (have) {
        const int nx = it + 256;
        const bool have_nx = nx < i1;
        float4 mv2, xv2[4], gz2[4];
        if (have_nx) fetch(nx, mv2, xv2, gz2);
        const float *mf = reinterpret_cast<const float *>(&mv);
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const float *xf = reinterpret_cast<const float *>(&xv[k]);
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const float yk = (1.f + 2.f * xf[j]) * mf[j];
                if (PK) {
#pragma unroll
                    for (int oo = 0; oo < 4; oo += 2) {
                        const f32x2 gg = {reinterpret_cast<const float *>(&gz[oo])[j], reinterpret_cast<const float *>(&gz[oo + 1])[j]};
                        pw[(oo >> 1) * 4 + k] += gg * f32x2{yk, yk};
                    }
                    const float dym = reinterpret_cast<const float *>(&gz[k])[j] * mf[j];
                    if (k & 1) { pl[k >> 1][1] += dym * xf[j]; pb[k >> 1][1] += dym; }
                    else       { pl[k >> 1][0] += dym * xf[j]; pb[k >> 1][0] += dym; }
                } else {
#pragma unroll
                    for (int oo = 0; oo < 4; ++oo) aw[oo * 4 + k] += reinterpret_cast<const float *>(&gz[oo])[j] * yk;
                    const float dym = reinterpret_cast<const float *>(&gz[k])[j] * mf[j];
                    al[k] += dym * xf[j];
                    ab[k] += dym;
                }
            }
        }
        if (have_nx) {
            mv = mv2;
#pragma unroll
            for (int k = 0; k < 4; ++k) { xv[k] = xv2[k]; gz[k] = gz2[k]; }
        }
        it = nx;
        have = have_nx;
    }
    float *o = out + ((long)blockIdx.x * 256 + threadIdx.x) * NACC;
    if (PK) {
#pragma unroll
        for (int h = 0; h < 2; ++h)
#pragma unroll
            for (int k = 0; k < 4; ++k) { o[(2 * h) * 4 + k] = pw[h * 4 + k][0]; o[(2 * h + 1) * 4 + k] = pw[h * 4 + k][1]; }
#pragma unroll
        for (int k = 0; k < 4; ++k) { o[16 + k] = pl[k >> 1][k & 1]; o[20 + k] = pb[k >> 1][k & 1]; }
    } else {
#pragma unroll
        for (int q = 0; q < 16; ++q) o[q] = aw[q];
#pragma unroll
        for (int k = 0; k < 4; ++k) { o[16 + k] = al[k]; o[20 + k] = ab[k]; }
    }
}

// The loop of actnorm_invconv_bwd_kernel<4, 4> itself (scalar source; flows.hip): per-workgroup-uniform multipliers (the 4x4
// matrix, exp(logs), bias) come from memory through scalar loads, so a build WITH the SLP vectorizer turns the arithmetic
// into v_pk_mul_f32 / v_pk_add_f32 with SGPR-PAIR operands — what the minimal victims above do not have.  dx is stored as in
// the real kernel; the 24 accumulators go straight to memory (no LDS, no atomics).
template <bool UNIFORM_IN_VGPR>
__global__ __launch_bounds__(256) void victim_real_kernel(const float4 *__restrict__ x, const float4 *__restrict__ g,
                                                          const float4 *__restrict__ m, const float *__restrict__ w,
                                                          const float *__restrict__ logs, const float *__restrict__ bias,
                                                          float4 *__restrict__ dx, float *__restrict__ out, int n_items, int nb) {
    constexpr int N = 4, V = 4;
    float wr[N * N], e[N], bi[N];
    int zero = 0;
    if (UNIFORM_IN_VGPR) asm volatile("" : "+v"(zero));    // an opaque per-lane 0: the loads below become vector loads and
                                                           // the multipliers live in VGPRs instead of SGPR pairs
#pragma unroll
    for (int q = 0; q < N * N; ++q) wr[q] = w[q + zero];
#pragma unroll
    for (int k = 0; k < N; ++k) { e[k] = expf(logs[blockIdx.x * N + k + zero]); bi[k] = bias[blockIdx.x * N + k + zero]; }
    float aw[N * N], al[N], ab[N];
#pragma unroll
    for (int q = 0; q < N * N; ++q) aw[q] = 0.f;
#pragma unroll
    for (int k = 0; k < N; ++k) { al[k] = 0.f; ab[k] = 0.f; }
    const int i0 = blockIdx.x * nb, i1 = min(n_items, i0 + nb);
    float4 mv4, xv4[N], gz4[N];
    auto fetch = [&](int it, float4 &m_, float4 *x_, float4 *g_) {
        m_ = m[it];
#pragma unroll
        for (int k = 0; k < N; ++k) { x_[k] = x[(long)it * 4 + k]; g_[k] = g[(long)it * 4 + k]; }
    };
    int it = i0 + threadIdx.x;
    bool have = it < i1;
    if (have) fetch(it, mv4, xv4, gz4);
    while (have) {
        const int nx = it + 256;
        const bool have_nx = nx < i1;
        float4 mv2, xv2[N], gz2[N];
        if (have_nx) fetch(nx, mv2, xv2, gz2);
        float mv[V], xv[N][V], gz[N][V];
#pragma unroll
        for (int j = 0; j < V; ++j) mv[j] = reinterpret_cast<const float *>(&mv4)[j];
#pragma unroll
        for (int k = 0; k < N; ++k)
#pragma unroll
            for (int j = 0; j < V; ++j) {
                xv[k][j] = reinterpret_cast<const float *>(&xv4[k])[j];
                gz[k][j] = reinterpret_cast<const float *>(&gz4[k])[j] * mv[j];
            }
#pragma unroll
        for (int k = 0; k < N; ++k) {
            float o[V];
#pragma unroll
            for (int j = 0; j < V; ++j) {
                const float yk = (bi[k] + e[k] * xv[k][j]) * mv[j];
                float dy = 0.f;
#pragma unroll
                for (int oo = 0; oo < N; ++oo) {
                    dy += wr[oo * N + k] * gz[oo][j];
                    aw[oo * N + k] += gz[oo][j] * yk;
                }
                const float dym = dy * mv[j];
                o[j] = dym * e[k];
                al[k] += dym * e[k] * xv[k][j];
                ab[k] += dym;
            }
            dx[(long)it * 4 + k] = float4{o[0], o[1], o[2], o[3]};
        }
        if (have_nx) {
            mv4 = mv2;
#pragma unroll
            for (int k = 0; k < N; ++k) { xv4[k] = xv2[k]; gz4[k] = gz2[k]; }
        }
        it = nx;
        have = have_nx;
    }
    float *op = out + ((long)blockIdx.x * 256 + threadIdx.x) * NACC;
#pragma unroll
    for (int q = 0; q < 16; ++q) op[q] = aw[q];
#pragma unroll
    for (int k = 0; k < 4; ++k) { op[16 + k] = al[k]; op[20 + k] = ab[k]; }
}

// hist: [0] mismatching accumulators in total, [1 .. 64] by lane, [65 .. 68] by wave, [69 .. 92] by accumulator,
// per_launch[launch] = mismatches of that launch
__global__ void check_kernel(const float *__restrict__ out, const float *__restrict__ want, long n, unsigned *hist,
                             unsigned *per_launch, int launch, float *first_bad) {
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
        const float a = out[i], e = want[i];
        if (__float_as_uint(a) != __float_as_uint(e)) {
            const long thread = i / NACC;
            const int q = (int)(i % NACC), tid = (int)(thread & 255);
            const unsigned k = atomicAdd(hist, 1u);
            atomicAdd(hist + 1 + (tid & 63), 1u);
            atomicAdd(hist + 65 + (tid >> 6), 1u);
            atomicAdd(hist + 69 + q, 1u);
            atomicAdd(per_launch + launch, 1u);
            if (k < 8) { first_bad[4 * k] = (float)launch; first_bad[4 * k + 1] = (float)i; first_bad[4 * k + 2] = a; first_bad[4 * k + 3] = e; }
        }
    }
}

// ------------------------------------------------------------------------------------------------------ aggressors
// All of them keep ~200 VGPRs live (at most 256 allocated, like convwrw_split_kernel<3,5,5,2>: victim waves fit beside them) and
// touch no memory until the guarded store at the end.
constexpr int NA = 48;
enum { AG_NONE = 0, AG_MFMA_BF16 = 1, AG_MFMA_F32 = 2, AG_PKVALU = 3, AG_MFMA_BF16_LDS = 4, AG_COUNT = 5,
       // library-only kinds (tools/pk_hazard_repro.py): what else do the real bf16 GEMMs have that the loops above lack?
       AG_MFMA_BF16_V256 = 5,      // kind 1 with all 256 VGPRs allocated (v255 touched), as convwrw_split_kernel / hipBLASLt kernels
       AG_MFMA_BF16_A512 = 6,      // ... and the accumulator file too (a255 touched): 512 registers, one wave per SIMD
       AG_MFMA_BF16_GLOBAL = 7,    // kind 1 with a streaming 16-byte global load per MFMA group
       AG_VALU_V256 = 8,           // no MFMA at all: plain fp32 VALU with 256 VGPRs allocated
       AG_MFMA_BF16_LDSW = 9,      // kind 4 that also WRITES its LDS every iteration (ds_write_b64)
       AG_LDSW64_NOMFMA = 10,      // the LDS reads + ds_write_b64 of kind 9 around plain fp32 VALU work: no MFMA
       AG_MFMA_LDSW32 = 11,        // kind 9 with a ds_write_b32
       AG_MFMA_LDSW128 = 12,       // kind 9 with a ds_write_b128
       AG_LDSW64_ONLY = 13,        // nothing but ds_write_b64 (and a little address arithmetic)
       AG_MFMA_LDSW2X32 = 14,      // kind 9 with a ds_write2_b32 (two dwords, two addresses)
       AG_MFMA_LDSW16 = 15,        // kind 9 with a ds_write_b16
       AG_MFMA_LDSW96 = 16,        // kind 9 with a ds_write_b96
       AG_F32MFMA_LDSW64 = 17,     // the ds_write_b64 beside FP32 MFMAs
       AG_MFMA_LDSW2X32_ADJ = 18,  // ds_write2_b32 of two ADJACENT dwords (the same 8 bytes as a ds_write_b64)
       AG_MFMA_LDSW2ST64_B64 = 19 };   // ds_write2st64_b64 (two 8-byte pieces)
constexpr bool ag_has_lds(int k) { return k == 4 || (k >= 9 && k <= 19); }
constexpr bool ag_has_mfma(int k) { return k == 1 || k == 4 || k == 5 || k == 6 || k == 7 || k == 9 || k == 11 || k == 12 || k == 14 || k == 15 || k == 16 || k == 18 || k == 19; }
static const char *kAgName[AG_COUNT] = {"none", "bf16 MFMA (registers only)", "fp32 MFMA (registers only)",
                                        "packed-fp32 VALU only", "bf16 MFMA + ds_read_b128 from its own LDS"};

template <int KIND>
__global__ __launch_bounds__(256, KIND == 6 ? 1 : 2) void aggressor_kernel(float *sink, int iters, unsigned seed) {
    __shared__ __attribute__((aligned(16))) float lds[ag_has_lds(KIND) ? 4096 : 4];
    const int tid = threadIdx.x;
    if (KIND == AG_MFMA_BF16_V256 || KIND == AG_VALU_V256) asm volatile("v_mov_b32 v255, 0" ::: "v255");
    if (KIND == AG_MFMA_BF16_A512) asm volatile("v_accvgpr_write_b32 a255, 0" ::: "a255");
    if (ag_has_lds(KIND)) {
        for (int i = tid; i < 4096; i += 256) lds[i] = (float)((i * 37 + seed) & 7) - 3.f;
        __syncthreads();
    }
    unsigned r = seed * 2654435761u + tid * 40503u;
    union { bf16x8 v; unsigned u[4]; } a, b;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        r = r * 1664525u + 1013904223u;
        a.u[i] = (r & 0x007f007fu) | 0x3f003f00u;        // two bf16 in [0.5, 1)
        r = r * 1664525u + 1013904223u;
        b.u[i] = (r & 0x007f007fu) | 0xbf003f00u;        // mixed signs
    }
    f32x4 acc[NA];
#pragma unroll
    for (int i = 0; i < NA; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    for (int it = 0; it < iters; ++it) {
        if (ag_has_lds(KIND)) {
            f32x4 l = f32x4{1.f, 2.f, 3.f, 4.f};
            if (KIND != AG_LDSW64_ONLY) {
                l = *reinterpret_cast<const f32x4 *>(lds + ((tid * 4 + it * 16) & 4092));
                a.u[0] ^= __float_as_uint(l[0]) & 0x00010001u;
            }
            if (KIND == AG_MFMA_BF16_LDSW || KIND == AG_LDSW64_NOMFMA || KIND == AG_LDSW64_ONLY || KIND == AG_F32MFMA_LDSW64)
                *reinterpret_cast<f32x2 *>(lds + ((tid * 2 + it * 8) & 4094)) = f32x2{l[1], l[2]};
            if (KIND == AG_MFMA_LDSW32) lds[(tid + it * 8) & 4095] = l[1];
            if (KIND == AG_MFMA_LDSW2X32) {
                float *q = lds + ((tid + it * 8) & 2047);
                asm volatile("ds_write2_b32 %0, %1, %2 offset1:64" :: "v"((unsigned)(size_t)(__attribute__((address_space(3))) float *)q), "v"(l[1]), "v"(l[2]) : "memory");
            }
            if (KIND == AG_MFMA_LDSW2X32_ADJ) {
                float *q = lds + ((tid * 2 + it * 8) & 4094);
                asm volatile("ds_write2_b32 %0, %1, %2 offset1:1" :: "v"((unsigned)(size_t)(__attribute__((address_space(3))) float *)q), "v"(l[1]), "v"(l[2]) : "memory");
            }
            if (KIND == AG_MFMA_LDSW2ST64_B64) {
                float *q = lds + ((tid * 2 + it * 8) & 1022);
                asm volatile("ds_write2st64_b64 %0, %1, %2 offset1:1" :: "v"((unsigned)(size_t)(__attribute__((address_space(3))) float *)q), "v"(f32x2{l[1], l[2]}), "v"(f32x2{l[3], l[0]}) : "memory");
            }
            if (KIND == AG_MFMA_LDSW16) reinterpret_cast<unsigned short *>(lds)[(tid + it * 8) & 8191] = (unsigned short)__float_as_uint(l[1]);
            if (KIND == AG_MFMA_LDSW96) {
                float *q = lds + ((tid * 4 + it * 8) & 4092);
                asm volatile("ds_write_b96 %0, %1" :: "v"((unsigned)(size_t)(__attribute__((address_space(3))) float *)q), "v"(__builtin_shufflevector(l, l, 0, 1, 2)) : "memory");
            }
            if (KIND == AG_MFMA_LDSW128) *reinterpret_cast<f32x4 *>(lds + ((tid * 4 + it * 8) & 4092)) = f32x4{l[1], l[2], l[3], l[0]};
        }
        if (KIND == AG_MFMA_BF16_GLOBAL) {
            const f32x4 l = reinterpret_cast<const f32x4 *>(sink)[64 + ((blockIdx.x * 256 + tid + it * 4096) & 0xfffff)];
            a.u[0] ^= __float_as_uint(l[0]) & 0x00010001u;
        }
#pragma unroll
        for (int i = 0; i < NA; ++i) {
            if (KIND == AG_LDSW64_ONLY) {
                if (i == 0) acc[0][0] += 1.f;
            } else if (ag_has_mfma(KIND)) {
                acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a.v, b.v, acc[i], 0, 0, 0);
            } else if (KIND == AG_MFMA_F32 || KIND == AG_F32MFMA_LDSW64) {
                acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(__uint_as_float(a.u[i & 3]), __uint_as_float(b.u[i & 3]), acc[i], 0, 0, 0);
            } else if (KIND == AG_VALU_V256 || KIND == AG_LDSW64_NOMFMA) {
                const float sa = __uint_as_float(a.u[i & 3]), sb = __uint_as_float(b.u[(i + 1) & 3]);
#pragma unroll
                for (int e = 0; e < 4; ++e) acc[i][e] = acc[i][e] * sa + sb;
            } else {
                const f32x2 s = {__uint_as_float(a.u[i & 3]), __uint_as_float(b.u[(i + 1) & 3])};
                f32x2 lo = {acc[i][0], acc[i][1]}, hi = {acc[i][2], acc[i][3]};
                lo = lo * s + s;
                hi = hi * s - s;
                acc[i] = f32x4{lo[0], lo[1], hi[0], hi[1]};
            }
        }
        a.u[1] ^= (unsigned)it & 0x00010001u;
    }
    float t = 0.f;
#pragma unroll
    for (int i = 0; i < NA; ++i) t += (acc[i][0] + acc[i][1]) + (acc[i][2] + acc[i][3]);
    if (t == 123456.789f) sink[tid] = t;                   // never true in practice: keeps the loop alive
}

static void launch_aggressor(int kind, int blocks, float *sink, int iters, unsigned seed, hipStream_t s) {
    switch (kind) {
    case AG_MFMA_BF16: hipLaunchKernelGGL((aggressor_kernel<AG_MFMA_BF16>), dim3(blocks), dim3(256), 0, s, sink, iters, seed); break;
    case AG_MFMA_F32: hipLaunchKernelGGL((aggressor_kernel<AG_MFMA_F32>), dim3(blocks), dim3(256), 0, s, sink, iters / 2, seed); break;
    case AG_PKVALU: hipLaunchKernelGGL((aggressor_kernel<AG_PKVALU>), dim3(blocks), dim3(256), 0, s, sink, iters, seed); break;
    case AG_MFMA_BF16_LDS: hipLaunchKernelGGL((aggressor_kernel<AG_MFMA_BF16_LDS>), dim3(blocks), dim3(256), 0, s, sink, iters, seed); break;
    case AG_MFMA_BF16_V256: hipLaunchKernelGGL((aggressor_kernel<AG_MFMA_BF16_V256>), dim3(blocks), dim3(256), 0, s, sink, iters, seed); break;
    case AG_MFMA_BF16_A512: hipLaunchKernelGGL((aggressor_kernel<AG_MFMA_BF16_A512>), dim3(blocks), dim3(256), 0, s, sink, iters, seed); break;
    case AG_MFMA_BF16_GLOBAL: hipLaunchKernelGGL((aggressor_kernel<AG_MFMA_BF16_GLOBAL>), dim3(blocks), dim3(256), 0, s, sink, iters, seed); break;
    case AG_VALU_V256: hipLaunchKernelGGL((aggressor_kernel<AG_VALU_V256>), dim3(blocks), dim3(256), 0, s, sink, iters, seed); break;
    case AG_MFMA_BF16_LDSW: hipLaunchKernelGGL((aggressor_kernel<AG_MFMA_BF16_LDSW>), dim3(blocks), dim3(256), 0, s, sink, iters, seed); break;
    case AG_LDSW64_NOMFMA: hipLaunchKernelGGL((aggressor_kernel<AG_LDSW64_NOMFMA>), dim3(blocks), dim3(256), 0, s, sink, iters, seed); break;
    case AG_MFMA_LDSW32: hipLaunchKernelGGL((aggressor_kernel<AG_MFMA_LDSW32>), dim3(blocks), dim3(256), 0, s, sink, iters, seed); break;
    case AG_MFMA_LDSW128: hipLaunchKernelGGL((aggressor_kernel<AG_MFMA_LDSW128>), dim3(blocks), dim3(256), 0, s, sink, iters, seed); break;
    case AG_LDSW64_ONLY: hipLaunchKernelGGL((aggressor_kernel<AG_LDSW64_ONLY>), dim3(blocks), dim3(256), 0, s, sink, iters * 40, seed); break;
    case AG_MFMA_LDSW2X32: hipLaunchKernelGGL((aggressor_kernel<AG_MFMA_LDSW2X32>), dim3(blocks), dim3(256), 0, s, sink, iters, seed); break;
    case AG_MFMA_LDSW16: hipLaunchKernelGGL((aggressor_kernel<AG_MFMA_LDSW16>), dim3(blocks), dim3(256), 0, s, sink, iters, seed); break;
    case AG_MFMA_LDSW96: hipLaunchKernelGGL((aggressor_kernel<AG_MFMA_LDSW96>), dim3(blocks), dim3(256), 0, s, sink, iters, seed); break;
    case AG_MFMA_LDSW2X32_ADJ: hipLaunchKernelGGL((aggressor_kernel<AG_MFMA_LDSW2X32_ADJ>), dim3(blocks), dim3(256), 0, s, sink, iters, seed); break;
    case AG_MFMA_LDSW2ST64_B64: hipLaunchKernelGGL((aggressor_kernel<AG_MFMA_LDSW2ST64_B64>), dim3(blocks), dim3(256), 0, s, sink, iters, seed); break;
    case AG_F32MFMA_LDSW64: hipLaunchKernelGGL((aggressor_kernel<AG_F32MFMA_LDSW64>), dim3(blocks), dim3(256), 0, s, sink, iters / 2, seed); break;
    default: break;
    }
}

extern "C" int pk_victim(int pk, const void *x, const void *g, const void *m, float *out, int groups, int n_items, int nb,
                         void *stream) {
    hipStream_t s = (hipStream_t)stream;
    if (pk) hipLaunchKernelGGL((victim_kernel<true>), dim3(groups), dim3(256), 0, s, (const float4 *)x, (const float4 *)g, (const float4 *)m, out, n_items, nb);
    else    hipLaunchKernelGGL((victim_kernel<false>), dim3(groups), dim3(256), 0, s, (const float4 *)x, (const float4 *)g, (const float4 *)m, out, n_items, nb);
    return (int)hipGetLastError();
}

extern "C" int pk_victim_real(int uniform_in_vgpr, const void *x, const void *g, const void *m, const float *w, const float *logs,
                              const float *bias, void *dx, float *out, int groups, int n_items, int nb, void *stream) {
    if (uniform_in_vgpr)
        hipLaunchKernelGGL(victim_real_kernel<true>, dim3(groups), dim3(256), 0, (hipStream_t)stream, (const float4 *)x,
                           (const float4 *)g, (const float4 *)m, w, logs, bias, (float4 *)dx, out, n_items, nb);
    else
        hipLaunchKernelGGL(victim_real_kernel<false>, dim3(groups), dim3(256), 0, (hipStream_t)stream, (const float4 *)x,
                           (const float4 *)g, (const float4 *)m, w, logs, bias, (float4 *)dx, out, n_items, nb);
    return (int)hipGetLastError();
}

extern "C" int pk_aggressor(int kind, int blocks, float *sink, int iters, unsigned seed, void *stream) {
    launch_aggressor(kind, blocks, sink, iters, seed, (hipStream_t)stream);
    return (int)hipGetLastError();
}

// a stream whose kernels run only on the compute units whose bit is set (n_words x 32 bits, CU 0 = bit 0 of word 0)
extern "C" void *pk_masked_stream(const unsigned *mask, int n_words) {
    hipStream_t s = nullptr;
    if (hipExtStreamCreateWithCUMask(&s, (unsigned)n_words, mask) != hipSuccess) return nullptr;
    return (void *)s;
}

#ifndef PK_AS_LIBRARY
int main(int argc, char **argv) {
    const int launches = argc > 1 ? atoi(argv[1]) : 600;
    const int ag_blocks = argc > 2 ? atoi(argv[2]) : 288;
    // victim geometry: the two failing shapes of lesson 12 — B = 8, T' = 64 (items per group 128: waves 2, 3 never enter the
    // loop) and T' = 124 (items 248: the last wave has 56 active lanes) — as 40 "groups" = workgroups with nb items each
    const int shapes[2] = {128, 248};
    hipStream_t sv, sa;
    CK(hipStreamCreateWithFlags(&sv, hipStreamNonBlocking));
    CK(hipStreamCreateWithFlags(&sa, hipStreamNonBlocking));
    float *sink;
    CK(hipMalloc(&sink, 1024));
    int total_bad = 0;
    for (int shp = 0; shp < 2; ++shp) {
        const int nb = shapes[shp], groups = 40, n_items = nb * groups;
        std::vector<float> hx((size_t)n_items * 16), hg((size_t)n_items * 16), hm((size_t)n_items * 4);
        unsigned r = 12345u + shp;
        auto rnd = [&](int lo, int hi) { r = r * 1664525u + 1013904223u; return (float)(lo + (int)((r >> 8) % (unsigned)(hi - lo + 1))); };
        for (auto &v : hx) v = rnd(-2, 2);
        for (auto &v : hg) v = rnd(-3, 3);
        for (auto &v : hm) v = rnd(0, 1);
        // exact expectation (integers: every partial sum < 2^24)
        std::vector<float> want((size_t)groups * 256 * NACC, 0.f);
        for (int gidx = 0; gidx < groups; ++gidx)
            for (int tid = 0; tid < 256; ++tid) {
                long acc[NACC] = {0};
                for (int it = gidx * nb + tid; it < (gidx + 1) * nb && it < n_items; it += 256)
                    for (int k = 0; k < 4; ++k)
                        for (int j = 0; j < 4; ++j) {
                            const long xv = (long)hx[(size_t)it * 16 + k * 4 + j], mv = (long)hm[(size_t)it * 4 + j];
                            const long yk = (1 + 2 * xv) * mv;
                            for (int oo = 0; oo < 4; ++oo) acc[oo * 4 + k] += (long)hg[(size_t)it * 16 + oo * 4 + j] * yk;
                            const long dym = (long)hg[(size_t)it * 16 + k * 4 + j] * mv;
                            acc[16 + k] += dym * xv;
                            acc[20 + k] += dym;
                        }
                for (int q = 0; q < NACC; ++q) want[((size_t)gidx * 256 + tid) * NACC + q] = (float)acc[q];
            }
        float4 *dx, *dg, *dm;
        float *dout, *dwant, *dbad;
        unsigned *dhist, *dper;
        const size_t nout = want.size();
        CK(hipMalloc(&dx, hx.size() * 4)); CK(hipMalloc(&dg, hg.size() * 4)); CK(hipMalloc(&dm, hm.size() * 4));
        CK(hipMalloc(&dout, nout * 4)); CK(hipMalloc(&dwant, nout * 4)); CK(hipMalloc(&dbad, 32 * 4));
        CK(hipMalloc(&dhist, 93 * 4)); CK(hipMalloc(&dper, (size_t)launches * 4));
        CK(hipMemcpy(dx, hx.data(), hx.size() * 4, hipMemcpyHostToDevice));
        CK(hipMemcpy(dg, hg.data(), hg.size() * 4, hipMemcpyHostToDevice));
        CK(hipMemcpy(dm, hm.data(), hm.size() * 4, hipMemcpyHostToDevice));
        CK(hipMemcpy(dwant, want.data(), nout * 4, hipMemcpyHostToDevice));
        for (int pk = 1; pk >= 0; --pk)
            for (int ag = 0; ag < AG_COUNT; ++ag) {
                CK(hipMemset(dhist, 0, 93 * 4)); CK(hipMemset(dper, 0, (size_t)launches * 4)); CK(hipMemset(dbad, 0, 32 * 4));
                CK(hipDeviceSynchronize());
                // the aggressor stream gets ~3x the victim stream's work so that it covers every victim launch
                for (int i = 0; i < launches; ++i) {
                    if (ag != AG_NONE && (i % 4) == 0)
                        for (int k = 0; k < 3; ++k) launch_aggressor(ag, ag_blocks, sink, 160, (unsigned)(i + k), sa);
                    CK(hipMemsetAsync(dout, 0xff, nout * 4, sv));
                    if (pk) hipLaunchKernelGGL((victim_kernel<true>), dim3(groups), dim3(256), 0, sv, dx, dg, dm, dout, n_items, nb);
                    else    hipLaunchKernelGGL((victim_kernel<false>), dim3(groups), dim3(256), 0, sv, dx, dg, dm, dout, n_items, nb);
                    hipLaunchKernelGGL(check_kernel, dim3(64), dim3(256), 0, sv, dout, dwant, (long)nout, dhist, dper, i, dbad);
                }
                CK(hipDeviceSynchronize());
                unsigned hist[93];
                std::vector<unsigned> per(launches);
                float bad[32];
                CK(hipMemcpy(hist, dhist, sizeof(hist), hipMemcpyDeviceToHost));
                CK(hipMemcpy(per.data(), dper, (size_t)launches * 4, hipMemcpyDeviceToHost));
                CK(hipMemcpy(bad, dbad, sizeof(bad), hipMemcpyDeviceToHost));
                int bad_launches = 0;
                for (unsigned v : per) bad_launches += v != 0;
                printf("items/group %3d  victim %-22s  aggressor %-42s : %4d / %d launches wrong, %u accumulators\n", nb,
                       pk ? "packed <2 x float>" : "scalar float", kAgName[ag], bad_launches, launches, hist[0]);
                if (hist[0]) {
                    printf("    by lane :");
                    for (int l = 0; l < 64; ++l) printf(" %u", hist[1 + l]);
                    printf("\n    by wave : %u %u %u %u\n    by acc  :", hist[65], hist[66], hist[67], hist[68]);
                    for (int q = 0; q < NACC; ++q) printf(" %u", hist[69 + q]);
                    printf("\n    first   : launch %.0f index %.0f got %g want %g\n", bad[0], bad[1], bad[2], bad[3]);
                }
                total_bad += bad_launches;
                fflush(stdout);
            }
        for (void *q : {(void *)dx, (void *)dg, (void *)dm, (void *)dout, (void *)dwant, (void *)dbad, (void *)dhist, (void *)dper}) CK(hipFree(q));
    }
    printf("total wrong launches: %d\n", total_bad);
    return 0;
}
#endif
