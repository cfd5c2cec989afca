// split_planes.hpp — an fp32 value as the exact sum of NS bf16 planes (shared by convgemm_split.hip and convwrw_tr.hip)
#pragma once
#include <hip/hip_runtime.h>

namespace glowtts {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef int i32x4 __attribute__((ext_vector_type(4)));
typedef int i32x2 __attribute__((ext_vector_type(2)));

typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

// Two values at a time: plane pl of (a, b) as one packed word (a in the low half) — v_cvt_pk_bf16_f32 (round to nearest
// even) for the plane, then the remainder v - plane, which is exact: each plane takes the top 8 significand bits of what
// is left, so three planes hold all 24 and h + m + l == v (checked bit for bit in tests/test_conv_math.py).
template <int NS>
__device__ __forceinline__ void split_planes2(float a, float b, unsigned (&o)[NS]) {
#pragma unroll
    for (int pl = 0; pl < NS; ++pl) {
        const f32x2 v = {a, b};
        const unsigned w = __builtin_bit_cast(unsigned, __builtin_convertvector(v, bf16x2));
        o[pl] = w;
        if (pl + 1 < NS) {
            a = a - __uint_as_float(w << 16);
            b = b - __uint_as_float(w & 0xffff0000u);
        }
    }
}

template <int NS>
__device__ __forceinline__ void split_planes(float v, unsigned (&o)[NS]) {      // one value: bf16 pattern in the low half
    split_planes2<NS>(v, 0.f, o);
#pragma unroll
    for (int pl = 0; pl < NS; ++pl) o[pl] &= 0xffffu;
}

// (a plane, b plane) pairs of the products kept, small magnitudes first
__host__ __device__ constexpr int n_products(int ns) { return ns == 3 ? 6 : (ns == 2 ? 3 : 1); }
__host__ __device__ constexpr int product_a(int ns, int k) {
    return ns == 3 ? (k == 0 ? 0 : k == 1 ? 2 : k == 2 ? 1 : k == 3 ? 0 : k == 4 ? 1 : 0) : (ns == 2 ? (k == 1 ? 1 : 0) : 0);
}
__host__ __device__ constexpr int product_b(int ns, int k) {
    return ns == 3 ? (k == 0 ? 2 : k == 1 ? 0 : k == 2 ? 1 : k == 3 ? 1 : k == 4 ? 0 : 0) : (ns == 2 ? (k == 0 ? 1 : 0) : 0);
}

}  // namespace glowtts
