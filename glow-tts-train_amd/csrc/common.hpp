// common.hpp — shared device/host helpers for libglowtts_hip.so (gfx950 only, wave = 64).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

#include <atomic>
#include <initializer_list>

#include "glowtts_hip.h"

namespace glowtts {

constexpr int kWave = 64;

void set_error(const char *fmt, ...);

// Every entry point funnels launch errors through here (no sync: hipGetLastError only sees launch-time errors).
#define GLOWTTS_CHECK_ARG(cond, ...)            \
    do {                                        \
        if (!(cond)) {                          \
            ::glowtts::set_error(__VA_ARGS__);  \
            return 1;                           \
        }                                       \
    } while (0)

#define GLOWTTS_LAUNCH_CHECK(name)                                                   \
    do {                                                                             \
        hipError_t e_ = hipGetLastError();                                           \
        if (e_ != hipSuccess) {                                                      \
            ::glowtts::set_error("%s: launch failed: %s", name, hipGetErrorString(e_)); \
            return (int)e_;                                                          \
        }                                                                            \
        return 0;                                                                    \
    } while (0)

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
    return v;
}

// Sum over a 256-thread workgroup; result valid in thread 0.  `red` = 4 floats of LDS per call site.
__device__ __forceinline__ float block_sum_256(float v, float *red) {
    v = wave_sum(v);
    const int w = threadIdx.x >> 6;
    __syncthreads();
    if ((threadIdx.x & 63) == 0) red[w] = v;
    __syncthreads();
    return red[0] + red[1] + red[2] + red[3];
}

// Call after the LAST MFMA of a group whenever control flow (a uniform `if`, a loop exit) stands between it and the first read
// of its accumulators.  The matrix pipe has no interlock for a VALU / LDS / memory instruction that reads a register an MFMA
// is still writing; the compiler pads that distance — but it counts it in LAYOUT order, so a conditional branch that jumps
// forward over a block lands on the read with the skipped block's instructions counted and not executed (ROCm 7.2;
// tools/mfma_hazard_scan.py finds such paths in the built library, tests/test_host_cpu.py runs it).  Round 4, attn_qblock_kernel:
// `if (jt + 1 < njt) { commit; barrier }` after the last key tile's MFMAs, the accumulator read at the join: register [3] of the
// last 16-key tile (the last pass to land) kept the sum of the previous k step in ~25 % of launches at T = 128 / 160 / 192.
// The scheduling barriers keep the MFMAs above the wait (they were otherwise free to sink below it — they touch no memory);
// 16 wait states cover every 16x16 opcode (the compiler keeps 8-10); a 32x32 one (19) would need two calls.
__device__ __forceinline__ void mfma_settle() {
    __builtin_amdgcn_sched_barrier(0);
    asm volatile("s_nop 15");
    __builtin_amdgcn_sched_barrier(0);
}

__device__ __forceinline__ float sigmoidf_(float x) { return 1.0f / (1.0f + expf(-x)); }

// The gate of the fused conv epilogues: hardware exp2 / rcp (v_exp_f32, v_rcp_f32: ~1 ulp each) instead of the libm
// routines — absolute error <= 3e-7 on values in [-1, 1], a third of the epilogue's instructions.  (Round 4: the reciprocal
// IS v_rcp_f32 now — `__frcp_rn` compiled to the correctly-rounded division sequence, v_div_scale / v_div_fmas / v_div_fixup
// around the v_rcp_f32: ~10 instructions per reciprocal, 40 % of the gate phase of csrc/wn_fused.hip.)
__device__ __forceinline__ float fast_sigmoid(float x) { return __builtin_amdgcn_rcpf(1.0f + __expf(-x)); }
__device__ __forceinline__ float fast_tanh(float x) { return 2.0f * __builtin_amdgcn_rcpf(1.0f + __expf(-2.0f * x)) - 1.0f; }


// 16-byte (V = 4) or scalar (V = 1) access along the contiguous T axis.
template <int V> struct Vec;
template <> struct Vec<4> {
    float4 d;
    __device__ __forceinline__ static Vec load(const float *p) { Vec r; r.d = *reinterpret_cast<const float4 *>(p); return r; }
    __device__ __forceinline__ static Vec zero() { Vec r; r.d = make_float4(0.f, 0.f, 0.f, 0.f); return r; }
    __device__ __forceinline__ void store(float *p) const { *reinterpret_cast<float4 *>(p) = d; }
    __device__ __forceinline__ float &operator[](int i) { return (&d.x)[i]; }
    __device__ __forceinline__ float operator[](int i) const { return (&d.x)[i]; }
};
template <> struct Vec<1> {
    float d;
    __device__ __forceinline__ static Vec load(const float *p) { Vec r; r.d = *p; return r; }
    __device__ __forceinline__ static Vec zero() { Vec r; r.d = 0.f; return r; }
    __device__ __forceinline__ void store(float *p) const { *p = d; }
    __device__ __forceinline__ float &operator[](int) { return d; }
    __device__ __forceinline__ float operator[](int) const { return d; }
};

// Typed access for tensors that are fp32 or bf16 in HBM (B16): element offset `off`, V elements; arithmetic stays fp32.
// bf16 -> fp32 is a shift; fp32 -> bf16 rounds to nearest even (v_cvt_pk_bf16_f32).
typedef __bf16 bf16x2_io __attribute__((ext_vector_type(2)));
typedef float f32x2_io __attribute__((ext_vector_type(2)));
__device__ __forceinline__ unsigned io_pack_bf16x2(float a, float b) {
    const f32x2_io v = {a, b};
    return __builtin_bit_cast(unsigned, __builtin_convertvector(v, bf16x2_io));
}
template <int V, bool B16> struct VecIO;
template <int V> struct VecIO<V, false> {
    static __device__ __forceinline__ Vec<V> load(const void *base, long off) { return Vec<V>::load(static_cast<const float *>(base) + off); }
    static __device__ __forceinline__ void store(void *base, long off, const Vec<V> &v) { v.store(static_cast<float *>(base) + off); }
};
template <> struct VecIO<4, true> {
    static __device__ __forceinline__ Vec<4> load(const void *base, long off) {
        const uint2 w = *reinterpret_cast<const uint2 *>(static_cast<const unsigned short *>(base) + off);
        Vec<4> r;
        r.d = make_float4(__uint_as_float(w.x << 16), __uint_as_float(w.x & 0xffff0000u), __uint_as_float(w.y << 16),
                          __uint_as_float(w.y & 0xffff0000u));
        return r;
    }
    static __device__ __forceinline__ void store(void *base, long off, const Vec<4> &v) {
        *reinterpret_cast<uint2 *>(static_cast<unsigned short *>(base) + off) =
            make_uint2(io_pack_bf16x2(v.d.x, v.d.y), io_pack_bf16x2(v.d.z, v.d.w));
    }
};
template <> struct VecIO<1, true> {
    static __device__ __forceinline__ Vec<1> load(const void *base, long off) {
        Vec<1> r;
        r.d = __uint_as_float((unsigned)static_cast<const unsigned short *>(base)[off] << 16);
        return r;
    }
    static __device__ __forceinline__ void store(void *base, long off, const Vec<1> &v) {
        static_cast<unsigned short *>(base)[off] = (unsigned short)(io_pack_bf16x2(v.d, 0.f) & 0xffffu);
    }
};

static inline bool aligned16(const void *p) { return (reinterpret_cast<uintptr_t>(p) & 15u) == 0; }

// 16-byte path is legal when every row start is 16-B aligned: T % 4 == 0 and all bases aligned.
static inline bool can_vec4(int T, std::initializer_list<const void *> ptrs) {
    if (T & 3) return false;
    for (const void *p : ptrs)
        if (p && !aligned16(p)) return false;
    return true;
}

// Dynamic-LDS limit of one kernel (`static LdsLimit x;` at the launch site): hipFuncSetAttribute is per DEVICE, so the
// high-water mark is kept per device — a process that drives a second GPU sets the attribute there too — and is monotone
// under concurrent callers (autograd runs one backward thread per device).
constexpr int kMaxDevices = 16;
struct LdsLimit {
    std::atomic<size_t> seen[kMaxDevices];
    int ensure(const void *kernel, size_t bytes, const char *who) {
        int dev = 0;
        (void)hipGetDevice(&dev);
        const bool tracked = dev >= 0 && dev < kMaxDevices;
        if (tracked && bytes <= seen[dev].load(std::memory_order_acquire)) return 0;
        hipError_t e = hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
        if (e != hipSuccess) {
            set_error("%s: cannot reserve %zu B of dynamic LDS: %s", who, bytes, hipGetErrorString(e));
            return (int)e;
        }
        if (tracked) {
            size_t cur = seen[dev].load(std::memory_order_relaxed);
            while (cur < bytes && !seen[dev].compare_exchange_weak(cur, bytes, std::memory_order_release)) {}
        }
        return 0;
    }
};

static inline int cdiv(long a, long b) { return (int)((a + b - 1) / b); }

// Tuning switches (include/glowtts_hip.h lists every one with its default).  ALL of them are read from the environment ONCE,
// at the library's first use of any of them, into a table of atomics; afterwards a launch costs one relaxed load per switch it
// consults (no getenv / atoi on the launch path, nothing for autograd's backward threads to race on), and the A/B tools flip a
// switch between blocks of steps with glowtts_set_knob(name, value).  They select between kernels / launch geometries with
// identical results.  The two timing-experiment switches that make kernels SKIP work (wrong results) exist in the tuning
// build only (-DGLOWTTS_TRACE: `make trace`), as does every `exp` test inside the kernels (GLOWTTS_EXP_BITS).
enum Knob {
    K_CONV_ROW_ADJ, K_CONV32_1X1, K_WRW1_PIPE, K_WRW1_MULTI, K_WRW1_CUS, K_WRW1_XCD, K_WN_FUSED, K_WRW_BATCH, K_WRW5_BSPLIT,
    K_WRW_TR, K_WRW_TR_MT, K_WRW_TR_NG, K_WRW_TR_NG_SPLITS, K_WRW_TR_PRIO, K_WRW_TR3, K_WRW_TR3_MT, K_MAS_WAVES, K_WRW5_CUS, K_WINO,
#ifdef GLOWTTS_TRACE
    K_BND_EXP, K_WRW1_EXP, K_CONV_EXP,
#endif
    K_COUNT
};
int knob(Knob k);
#ifdef GLOWTTS_TRACE
#define GLOWTTS_EXP_BITS(v) (v)
#else
#define GLOWTTS_EXP_BITS(v) 0
#endif

}  // namespace glowtts
