// convwino.hip — the WN stack's gated 5-tap in-conv (reference layers.py:146 + utils.py:31-38) in a one-dimensional
// Winograd / Cook-Toom form F(4, 5) on the bf16 matrix pipe with fp32-equivalent results.
//
//   out[m][4j + i] = sum_p AT[i][p] * ( sum_c U[m][c][p] * V[c][p][j] ),   U = G w (8 points from 5 taps),
//                                                                          V = BT d (8 points from the 8 input frames 4j-2 .. 4j+5)
//
// 8 products per (row, channel, 4 frames) where the direct form needs 20: 2.5 x fewer MFMAs.  U and V are fp32 values, each split
// into three bf16 planes exactly as in convgemm_split.hip (six products per fp32 product, fp32 accumulation), so what the
// form adds to the direct kernel's error are the fp32 roundings of the three transforms (tools/winograd_study.py: 2.4e-6 /
// 2.4e-7 of the output's largest element against 9.5e-7 / 1.1e-7 — the level of the native fp32 MFMA kernel).
//
// Operand traffic is what shapes the kernel: U is 8 / 5 and V 2 x the size of what they come from, so per MFMA the
// kernel takes in ~4 x the operand bytes of the direct form at equal tile shape.  Hence ONE wave per SIMD with a register tile
// of 2 row tiles x 3 column blocks x 8 points (192 accumulator registers): a workgroup = 4 waves = 128 rows (64 tanh + the
// 64 sigmoid rows of the same channels) x 48 Winograd tiles (192 frames, tiles numbered across utterances).  U streams from
// L2 through a register ring (one 16-byte load per fragment), V is made in the kernel: per 32-channel step every thread loads
// 3 x (2 channels x 4 frames; the halo frames come from the neighbouring lanes by DPP), transforms, splits and stores into the
// OTHER of two LDS images while the MFMAs read this one.  One wave per SIMD also means that nothing hides a stall: the staging and
// the epilogue are written as PINNED instructions in stages of independent operations (lesson 43).
#include "convgemm_common.hpp"
#include "split_planes.hpp"
#include <atomic>
#include <type_traits>

namespace glowtts {

constexpr int WINO_P = 8;          // points
constexpr int WINO_CB = 3;         // column blocks (16 Winograd tiles = 64 frames each) per workgroup
constexpr int WINO_COLS = 16 * WINO_CB;

// offset (bf16 elements, a multiple of 8) of the U planes of the convolution whose packed fp32 weights start `o` floats into
// the bound buffer: sizes are multiples of 5 x 16 floats, so consecutive convolutions keep their order and never overlap
__host__ __device__ inline long wino_u_offset(long o) { return ((8 * o + 4) / 5 + 7) / 8 * 8; }

// ---- U = G w for every listed convolution of a packed-weight buffer, split into three bf16 planes -----------------------
// table rows: (offset of the packed forward weights in floats, G = Cin / 16, M).  Packed layout [tap][G][M][16]; U layout
// [point][k-step = group pair][M][32] with the 32 channels of a k-step in MFMA slot order: chunk lk = channels 4lk..4lk+3 of
// group 2ks, then of group 2ks + 1 (the relabelling of convgemm_split.hip: a lane's 8 k values are ONE 16-byte load).
__global__ __launch_bounds__(256) void wino_weights_kernel(const float *__restrict__ base, const long *__restrict__ table, int n_conv,
                                                           unsigned short *__restrict__ planes, long plane_stride) {
    const int conv = blockIdx.y;
    const long off = table[conv * 3];
    const int G = (int)table[conv * 3 + 1], M = (int)table[conv * 3 + 2];
    const int nks = G / 2;
    const float *wp = base + off;
    unsigned short *U = planes + wino_u_offset(off);
    const long per_point = (long)nks * M * 32;
    for (long idx = blockIdx.x * 256L + threadIdx.x; idx < (long)nks * M * 4; idx += (long)gridDim.x * 256) {
        const int lk = (int)(idx & 3);
        const long rest = idx >> 2;
        const int m = (int)(rest % M), ks = (int)(rest / M);
        float w[5][8];
#pragma unroll
        for (int tap = 0; tap < 5; ++tap)
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const float4 v = *reinterpret_cast<const float4 *>(wp + (((long)tap * G + 2 * ks + h) * M + m) * 16 + lk * 4);
                w[tap][4 * h + 0] = v.x; w[tap][4 * h + 1] = v.y; w[tap][4 * h + 2] = v.z; w[tap][4 * h + 3] = v.w;
            }
        unsigned o[WINO_P][3][4];
#pragma unroll
        for (int j = 0; j < 8; j += 2) {
            float u[2][WINO_P];
#pragma unroll
            for (int e = 0; e < 2; ++e) {
                const float w0 = w[0][j + e], w1 = w[1][j + e], w2 = w[2][j + e], w3 = w[3][j + e], w4 = w[4][j + e];
                // rows of G for the points {0, 1, -1, 2, -2, 1/2, -1/2, inf}
                const float e1 = (w0 + w2) + w4, o1 = w1 + w3;
                const float e2 = (w0 * (1.f / 90.f) + w2 * (2.f / 45.f)) + w4 * (8.f / 45.f), o2 = w1 * (1.f / 45.f) + w3 * (4.f / 45.f);
                const float e3 = (w0 * (32.f / 45.f) + w2 * (8.f / 45.f)) + w4 * (2.f / 45.f), o3 = w1 * (16.f / 45.f) + w3 * (4.f / 45.f);
                u[e][0] = -w0;
                u[e][1] = (e1 + o1) * (-2.f / 9.f);
                u[e][2] = (e1 - o1) * (-2.f / 9.f);
                u[e][3] = e2 + o2;
                u[e][4] = e2 - o2;
                u[e][5] = e3 + o3;
                u[e][6] = e3 - o3;
                u[e][7] = w4;
            }
#pragma unroll
            for (int p = 0; p < WINO_P; ++p) {
                unsigned pl[3];
                split_planes2<3>(u[0][p], u[1][p], pl);
#pragma unroll
                for (int q = 0; q < 3; ++q) o[p][q][j >> 1] = pl[q];
            }
        }
#pragma unroll
        for (int p = 0; p < WINO_P; ++p)
#pragma unroll
            for (int q = 0; q < 3; ++q)
                *reinterpret_cast<uint4 *>(U + q * plane_stride + p * per_point + ((long)ks * M + m) * 32 + lk * 8) =
                    make_uint4(o[p][q][0], o[p][q][1], o[p][q][2], o[p][q][3]);
    }
}


// Vector instructions whose ORDER matters (the staging arithmetic beside the MFMAs, one wave per SIMD): `asm volatile` keeps them
// where they are written, between the scheduling barriers around each MFMA pair — plain C++ arithmetic is emitted wherever
// instruction selection likes, which put every consumer right behind its producer (8 cycles of latency each with nobody to hide it).
__device__ __forceinline__ float vadd(float a, float b) { float r; asm volatile("v_add_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b)); return r; }
__device__ __forceinline__ float vsub(float a, float b) { float r; asm volatile("v_sub_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b)); return r; }
__device__ __forceinline__ float vmulk(float k, float a) { float r; asm volatile("v_mul_f32 %0, %1, %2" : "=v"(r) : "s"(k), "v"(a)); return r; }
__device__ __forceinline__ unsigned vcvtpk(float a, float b) { unsigned r; asm volatile("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b)); return r; }
__device__ __forceinline__ float vlo(unsigned w) { float r; asm volatile("v_lshlrev_b32 %0, 16, %1" : "=v"(r) : "v"(w)); return r; }
__device__ __forceinline__ float vmul(float a, float b) { float r; asm volatile("v_mul_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b)); return r; }
__device__ __forceinline__ float vaddk(float k, float a) { float r; asm volatile("v_add_f32 %0, %1, %2" : "=v"(r) : "s"(k), "v"(a)); return r; }
__device__ __forceinline__ float vexp2(float a) { float r; asm volatile("v_exp_f32 %0, %1" : "=v"(r) : "v"(a)); return r; }
__device__ __forceinline__ float vrcp(float a) { float r; asm volatile("v_rcp_f32 %0, %1" : "=v"(r) : "v"(a)); return r; }
__device__ __forceinline__ float vtanh_from_rcp(float a) { float r; asm volatile("v_fma_f32 %0, %1, 2.0, -1.0" : "=v"(r) : "v"(a)); return r; }
template <int J> __device__ __forceinline__ float vubyte(unsigned w) {
    float r;
    if (J == 0) asm volatile("v_cvt_f32_ubyte0 %0, %1" : "=v"(r) : "v"(w));
    if (J == 1) asm volatile("v_cvt_f32_ubyte1 %0, %1" : "=v"(r) : "v"(w));
    if (J == 2) asm volatile("v_cvt_f32_ubyte2 %0, %1" : "=v"(r) : "v"(w));
    if (J == 3) asm volatile("v_cvt_f32_ubyte3 %0, %1" : "=v"(r) : "v"(w));
    return r;
}
__device__ __forceinline__ float vhi(unsigned w) { float r; asm volatile("v_and_b32 %0, 0xffff0000, %1" : "=v"(r) : "v"(w)); return r; }

// ---- the kernel ----------------------------------------------------------------------------------------------------------
// LDS image of one 32-channel step: [plane 3][point 8][column 48][32 bf16 = 64 B], the four 16-byte chunks of a row XOR-ed
// with (column >> 1) & 3 (a 64-byte pitch alone would put the 16 columns of a fragment read on 2 of the 8 chunk positions).
constexpr int WINO_IMG_DW = 3 * WINO_P * WINO_COLS * 16;        // dwords per image (73 728 B)

template <int EXP>
__global__ __launch_bounds__(256, 1) void wino_gate_fwd_kernel(ConvGemmParams p, const unsigned short *__restrict__ U, long plane_stride) {
    extern __shared__ __align__(16) float smem[];
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, lrow = lane & 15, lk = lane >> 4;
    const int n_rt = p.H / 64;                          // row groups: 64 tanh + 64 sigmoid rows
    const int tile_m = blockIdx.x % n_rt;
    const int colg = blockIdx.x / n_rt;
    const int tpu = p.T >> 2;                           // Winograd tiles per utterance
    const int n_tiles = p.B * tpu;
    const int nks = p.Cin >> 5;

    f32x4 acc[2][WINO_P][WINO_CB];
#pragma unroll
    for (int r = 0; r < 2; ++r)
#pragma unroll
        for (int q = 0; q < WINO_P; ++q)
#pragma unroll
            for (int c = 0; c < WINO_CB; ++c) acc[r][q][c] = f32x4{0.f, 0.f, 0.f, 0.f};

    // ---- A: U planes, [point][k-step][M][32]; a lane's fragment = 16 bytes at (row, chunk lk) ------------------------------
    const long per_point = (long)nks * p.M * 32;        // bf16 elements
    const int ubytes = (int)(WINO_P * per_point * 2);
    __amdgpu_buffer_rsrc_t urs[3];
#pragma unroll
    for (int pl = 0; pl < 3; ++pl)
        urs[pl] = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned short *>(U + pl * plane_stride), 0, ubytes, 0x00020000);
    int uvo[2];
    uvo[0] = ((tile_m * 64 + wave * 16 + lrow) * 32 + lk * 8) * 2;                   // tanh rows
    uvo[1] = ((p.H + tile_m * 64 + wave * 16 + lrow) * 32 + lk * 8) * 2;             // sigmoid rows of the same channels
    constexpr int RING = 4;
    i32x4 a[RING][2][3];
    auto uload = [&](int q, int slot) {                 // q = k-step * 8 + point; past the end: an out-of-range offset (zeros)
        const int ks = q >> 3, pt = q & 7;
        const int so = ks < nks ? (int)((pt * per_point + (long)ks * p.M * 32) * 2) : ubytes;
#pragma unroll
        for (int r = 0; r < 2; ++r)
#pragma unroll
            for (int pl = 0; pl < 3; ++pl)
                a[slot][r][pl] = __builtin_bit_cast(i32x4, __builtin_amdgcn_raw_buffer_load_b128(urs[pl], uvo[r], so, 0));
    };

    // ---- V: three items per thread and k-step; item = (column, channel pair) ----------------------------------------------
    const __amdgpu_buffer_rsrc_t xrs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(p.x), 0, (int)((long)p.B * p.x_bs * 4), 0x00020000);
    constexpr int OOR = 0x7ffffff0;
    int voL[3], voM[3], voR[3], sdw[3];
    bool hasL[3], hasR[3];
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        // lanes: 16 consecutive columns x 4 channel pairs — a 16-byte load instruction covers 4 (x 2) channel rows of 256
        // contiguous bytes (with the channel pair fastest it was 16 rows of 64 bytes, and the input loads cost the kernel 10 us:
        // the texture path works per cache line touched)
        const int idx = tid + 256 * i;
        const int r12 = idx >> 6;
        const int col = (r12 % 3) * 16 + (idx & 15), cp = (r12 / 3) * 4 + ((idx >> 4) & 3);
        const int n = colg * WINO_COLS + col;
        const int b = n / tpu, t0 = (n - b * tpu) * 4;
        const int gsel = cp >> 3, kk = 2 * (cp & 7);
        const bool ok = n < n_tiles;
        const int o = (int)(((long)b * p.x_bs + (long)(gsel * 16 + kk) * p.T + t0) * 4);
        voM[i] = ok ? o : OOR;
        // the two halo frames on either side are the neighbouring tile's frames — the neighbouring LANE's (16 consecutive columns per
        // row of lanes): only the lanes at the ends of a row load theirs, the others take them with a DPP row shift (tstage)
        hasL[i] = ok && t0 > 0;
        hasR[i] = ok && t0 + 4 < p.T;
        voL[i] = hasL[i] && (lane & 15) == 0 ? o - 8 : OOR;
        voR[i] = hasR[i] && (lane & 15) == 15 ? o + 16 : OOR;
        sdw[i] = col * 16 + (((kk >> 2) ^ ((col >> 1) & 3)) * 4) + gsel * 2 + (cp & 1);
    }
    const int rowb = p.T * 4;
    float xr[3][2][8];
    float dmy[6] = {1.f, 2.f, 3.f, 4.f, 5.f, 6.f}, dsc = 0.999f + 1e-6f * lane;
    const int xbytes = (int)((long)p.B * p.x_bs * 4);
    auto xload1 = [&](int ks, int i) {
        const int so = ks < nks ? ks * 32 * rowb : xbytes;        // past the last step: out of range (zeros, no memory access)
#pragma unroll
        for (int c = 0; c < 2; ++c) {
            const f32x2 l = __builtin_bit_cast(f32x2, __builtin_amdgcn_raw_buffer_load_b64(xrs, voL[i] == OOR ? OOR : voL[i] + c * rowb, so, 0));
            const f32x4 m = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(xrs, voM[i] == OOR ? OOR : voM[i] + c * rowb, so, 0));
            const f32x2 r = __builtin_bit_cast(f32x2, __builtin_amdgcn_raw_buffer_load_b64(xrs, voR[i] == OOR ? OOR : voR[i] + c * rowb, so, 0));
            xr[i][c][0] = l[0]; xr[i][c][1] = l[1];
            xr[i][c][2] = m[0]; xr[i][c][3] = m[1]; xr[i][c][4] = m[2]; xr[i][c][5] = m[3];
            xr[i][c][6] = r[0]; xr[i][c][7] = r[1];
        }
    };
    auto xhalo = [&](int i, int c) {                    // frames t0 - 2, t0 - 1 / t0 + 4, t0 + 5 from the neighbouring lanes' loads
        const float l0 = __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(__builtin_bit_cast(int, xr[i][c][0]), __builtin_bit_cast(int, xr[i][c][4]), 0x111, 0xf, 0xf, false));
        const float l1 = __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(__builtin_bit_cast(int, xr[i][c][1]), __builtin_bit_cast(int, xr[i][c][5]), 0x111, 0xf, 0xf, false));
        const float r0 = __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(__builtin_bit_cast(int, xr[i][c][6]), __builtin_bit_cast(int, xr[i][c][2]), 0x101, 0xf, 0xf, false));
        const float r1 = __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(__builtin_bit_cast(int, xr[i][c][7]), __builtin_bit_cast(int, xr[i][c][3]), 0x101, 0xf, 0xf, false));
        xr[i][c][0] = hasL[i] ? l0 : 0.f; xr[i][c][1] = hasL[i] ? l1 : 0.f;
        xr[i][c][6] = hasR[i] ? r0 : 0.f; xr[i][c][7] = hasR[i] ? r1 : 0.f;
    };
    auto lds_barrier = [&]() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); };

    // ---- B fragments: 16 bytes at (column cb * 16 + lrow, chunk lk ^ swizzle) -------------------------------------------------
    const int bdw = lrow * 16 + ((lk ^ ((lrow >> 1) & 3)) * 4);
    i32x4 bv[2][3];
    auto bfetch = [&](const float *img, int pt, int cb, int slot) {
#pragma unroll
        for (int pl = 0; pl < 3; ++pl)
            bv[slot][pl] = *reinterpret_cast<const i32x4 *>(img + bdw + ((pl * WINO_P + pt) * WINO_COLS + cb * 16) * 16);
    };

    GLOWTTS_TRACE_POINT(0);
#pragma unroll
    for (int i = 0; i < RING - 1; ++i) uload(i, i);
    xload1(0, 0); xload1(0, 1); xload1(0, 2);
    {   // the first image: the thread's six channels (3 items x 2) transformed in LOCKSTEP and the 24 (item, point) pairs split in
        // lockstep, as pinned instructions (the compiler's order for this — one value chain after the other — took 2.4 us)
        float pv[3][2][WINO_P];
#pragma unroll
        for (int i = 0; i < 3; ++i) { xhalo(i, 0); xhalo(i, 1); }
        float a0[6], a1[6], a2[6], a3[6], a4[6], a5[6], m0[6], m1[6], m2[6], m3[6], m4[6], m5[6], u0[6], u1[6], u2[6], u3[6];
#define WD(j) xr[w >> 1][w & 1][j]
#define WV(j) pv[w >> 1][w & 1][j]
#pragma unroll
        for (int w = 0; w < 6; ++w) { a0[w] = vsub(WD(6), WD(0)); a1[w] = vsub(WD(2), WD(4)); a2[w] = vsub(WD(7), WD(1)); a3[w] = vsub(WD(3), WD(5)); a4[w] = vadd(WD(2), WD(6)); a5[w] = vadd(WD(1), WD(5)); }
#pragma unroll
        for (int w = 0; w < 6; ++w) { m0[w] = vmulk(5.25f, a1[w]); m1[w] = vmulk(5.25f, a3[w]); m2[w] = vmulk(4.25f, WD(4)); m3[w] = vmulk(4.25f, WD(3)); m4[w] = vmulk(0.25f, WD(2)); m5[w] = vmulk(1.25f, WD(4)); }
#pragma unroll
        for (int w = 0; w < 6; ++w) { WV(0) = vadd(a0[w], m0[w]); WV(7) = vadd(a2[w], m1[w]); u0[w] = vsub(a4[w], m2[w]); u1[w] = vsub(a5[w], m3[w]); u2[w] = vadd(WD(6), m4[w]); m0[w] = vmulk(0.5f, WD(1)); }
#pragma unroll
        for (int w = 0; w < 6; ++w) { WV(1) = vadd(u0[w], u1[w]); WV(2) = vsub(u0[w], u1[w]); u3[w] = vsub(u2[w], m5[w]); m1[w] = vmulk(2.5f, WD(3)); m2[w] = vmulk(2.f, WD(5)); m3[w] = vmulk(4.f, WD(2)); }
#pragma unroll
        for (int w = 0; w < 6; ++w) { a0[w] = vsub(m0[w], m1[w]); a1[w] = vadd(WD(6), m3[w]); m4[w] = vmulk(5.f, WD(4)); m5[w] = vmulk(2.f, WD(1)); m0[w] = vmulk(0.5f, WD(5)); }
#pragma unroll
        for (int w = 0; w < 6; ++w) { u0[w] = vadd(a0[w], m2[w]); u1[w] = vsub(a1[w], m4[w]); a2[w] = vsub(m5[w], m1[w]); }
#pragma unroll
        for (int w = 0; w < 6; ++w) { WV(3) = vadd(u3[w], u0[w]); WV(4) = vsub(u3[w], u0[w]); a3[w] = vadd(a2[w], m0[w]); }
#pragma unroll
        for (int w = 0; w < 6; ++w) { WV(5) = vadd(u1[w], a3[w]); WV(6) = vsub(u1[w], a3[w]); }
#undef WD
#undef WV
#pragma unroll
        for (int i = 0; i < 3; ++i) {                   // eight points of an item in lockstep
            float sa[WINO_P], sb[WINO_P], lo[WINO_P], hi[WINO_P];
            unsigned sw[3][WINO_P];
#pragma unroll
            for (int q = 0; q < WINO_P; ++q) { sa[q] = pv[i][0][q]; sb[q] = pv[i][1][q]; }
#pragma unroll
            for (int k = 0; k < 3; ++k) {
#pragma unroll
                for (int q = 0; q < WINO_P; ++q) sw[k][q] = vcvtpk(sa[q], sb[q]);
                if (k < 2) {
#pragma unroll
                    for (int q = 0; q < WINO_P; ++q) { lo[q] = vlo(sw[k][q]); hi[q] = vhi(sw[k][q]); }
#pragma unroll
                    for (int q = 0; q < WINO_P; ++q) { sa[q] = vsub(sa[q], lo[q]); sb[q] = vsub(sb[q], hi[q]); }
                }
            }
            unsigned *dst = reinterpret_cast<unsigned *>(smem) + sdw[i];
#pragma unroll
            for (int q = 0; q < WINO_P; ++q)
#pragma unroll
                for (int k = 0; k < 3; ++k) dst[(k * WINO_P + q) * (WINO_COLS * 16)] = sw[k][q];
        }
    }
    xload1(1, 0); xload1(1, 1); xload1(1, 2);
    lds_barrier();
    GLOWTTS_TRACE_POINT(1);
    // One k-step.  MORE (compile time): the next step's image is made beside the MFMAs; loads past the last step use an
    // out-of-range offset instead of a branch.
    //
    // One wave per SIMD means nobody else fills a stall: a vector instruction that needs the result of the one before it waits
    // ~8 cycles, and the compiler orders the transform / split arithmetic for few registers, i.e. as dependent chains (measured:
    // the staging arithmetic cost the same 10 us per launch wherever the scheduler put it).  So the order is pinned here: a region
    // = one (point, column block) = six MFMA PAIRS (one product each for the tanh and the sigmoid row tile), and behind every
    // pair ONE STAGE of the staging work — up to six vector instructions that are independent of each other and whose inputs
    // were produced a whole pair (32 cycles) earlier.  24 regions per step; item i of the thread's three takes regions
    // 8 i .. 8 i + 7: its two channels' transforms (regions 0, 1), its eight points split three at a time in lockstep and stored
    // (regions 2, 4, 6), then its registers take the step after next.
    auto kstep = [&](int ks, auto more_tag) {
        constexpr bool MORE = decltype(more_tag)::value && EXP != 3;
        const float *cur = smem + (ks & 1) * WINO_IMG_DW;
        float *nxt = smem + ((ks + 1) & 1) * WINO_IMG_DW;
        float v[2][WINO_P];
        float ta[6], tm[13], tt[6];                      // transform temporaries
        float sa[3], sb[3];                              // split: what is left of (channel, channel + 1) of three points
        unsigned sw[3][3];
        float flo[3], fhi[3];
        auto tstage = [&](int i, int c, int st) {        // BT d of channel c of item i, eight stages of independent instructions
            if (st == 0) xhalo(i, c);
            const float d0 = xr[i][c][0], d1 = xr[i][c][1], d2 = xr[i][c][2], d3 = xr[i][c][3], d4 = xr[i][c][4], d5 = xr[i][c][5],
                        d6 = xr[i][c][6], d7 = xr[i][c][7];
            if (EXP == 8) {                              // (timing experiment: the loads are waited for, nothing is computed)
                if (st == 0) asm volatile("" :: "v"(d0), "v"(d1), "v"(d2), "v"(d3), "v"(d4), "v"(d5), "v"(d6), "v"(d7));
                return;
            }
            if (st == 0) { ta[0] = vsub(d6, d0); ta[1] = vsub(d2, d4); ta[2] = vsub(d7, d1); ta[3] = vsub(d3, d5); ta[4] = vadd(d2, d6); ta[5] = vadd(d1, d5); }
            if (st == 1) { tm[0] = vmulk(5.25f, ta[1]); tm[1] = vmulk(5.25f, ta[3]); tm[2] = vmulk(4.25f, d4); tm[3] = vmulk(4.25f, d3); tm[4] = vmulk(0.25f, d2); tm[5] = vmulk(1.25f, d4); }
            if (st == 2) { v[c][0] = vadd(ta[0], tm[0]); v[c][7] = vadd(ta[2], tm[1]); tt[0] = vsub(ta[4], tm[2]); tt[1] = vsub(ta[5], tm[3]); tt[2] = vadd(d6, tm[4]); tm[6] = vmulk(0.5f, d1); }
            if (st == 3) { v[c][1] = vadd(tt[0], tt[1]); v[c][2] = vsub(tt[0], tt[1]); tt[3] = vsub(tt[2], tm[5]); tm[7] = vmulk(2.5f, d3); tm[8] = vmulk(2.f, d5); tm[9] = vmulk(4.f, d2); }
            if (st == 4) { ta[0] = vsub(tm[6], tm[7]); ta[1] = vadd(d6, tm[9]); tm[10] = vmulk(5.f, d4); tm[11] = vmulk(2.f, d1); tm[12] = vmulk(0.5f, d5); }
            if (st == 5) { tt[4] = vadd(ta[0], tm[8]); tt[5] = vsub(ta[1], tm[10]); ta[2] = vsub(tm[11], tm[7]); }
            if (st == 6) { v[c][3] = vadd(tt[3], tt[4]); v[c][4] = vsub(tt[3], tt[4]); ta[3] = vadd(ta[2], tm[12]); }
            if (st == 7) { v[c][5] = vadd(tt[5], ta[3]); v[c][6] = vsub(tt[5], ta[3]); }
        };
        auto sstage = [&](int i, int q0, int nq, int st) {   // points q0 .. q0 + nq - 1 split in lockstep, seven stages
            if (st == 0) {
#pragma unroll
                for (int n = 0; n < 3; ++n) if (n < nq) { sa[n] = v[0][q0 + n]; sb[n] = v[1][q0 + n]; }
            }
            if (st == 0 || st == 3 || st == 6) {
                const int k = st / 3;
#pragma unroll
                for (int n = 0; n < 3; ++n) if (n < nq) sw[k][n] = vcvtpk(sa[n], sb[n]);
            }
            if (st == 1 || st == 4) {
                const int k = st / 3;
#pragma unroll
                for (int n = 0; n < 3; ++n) if (n < nq) { flo[n] = vlo(sw[k][n]); fhi[n] = vhi(sw[k][n]); }
            }
            if (st == 2 || st == 5) {
#pragma unroll
                for (int n = 0; n < 3; ++n) if (n < nq) { sa[n] = vsub(sa[n], flo[n]); sb[n] = vsub(sb[n], fhi[n]); }
            }
            if (st == 6) {
                unsigned *dst = reinterpret_cast<unsigned *>(nxt) + sdw[i];
#pragma unroll
                for (int n = 0; n < 3; ++n) if (n < nq) {
#pragma unroll
                    for (int k = 0; k < 3; ++k) dst[(k * WINO_P + q0 + n) * (WINO_COLS * 16)] = sw[k][n];
                }
            }
        };
        bfetch(cur, 0, 0, 0);
#pragma unroll
        for (int pt = 0; pt < WINO_P; ++pt) {
            uload(ks * 8 + pt + RING - 1, (pt + RING - 1) % RING);
#pragma unroll
            for (int cb = 0; cb < WINO_CB; ++cb) {
                const int n = pt * WINO_CB + cb;
                const int i = n >> 3, j = n & 7;                   // item, region of the item
                if (n + 1 < WINO_P * WINO_CB) bfetch(cur, (n + 1) / WINO_CB, (n + 1) % WINO_CB, (n + 1) & 1);
#pragma unroll
                for (int k = 0; k < 6; ++k) {
                    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                    for (int r = 0; r < 2; ++r)
                        acc[r][pt][cb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(
                            __builtin_bit_cast(bf16x8, a[pt % RING][r][product_a(3, k)]),
                            __builtin_bit_cast(bf16x8, bv[n & 1][product_b(3, k)]), acc[r][pt][cb], 0, 0, 0);
                    __builtin_amdgcn_sched_barrier(0);
                    if constexpr (EXP == 7 && decltype(more_tag)::value) {       // (timing experiment: six independent FMAs behind every pair)
#pragma unroll
                        for (int e = 0; e < 6; ++e) asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(dmy[e]) : "v"(dsc));
                    }
                    if constexpr (MORE && EXP != 7) {
                        if (j < 2) tstage(i, j, k);
                        else if (EXP != 1 && EXP != 8 && j == 2) sstage(i, 0, 3, k);
                        else if (EXP != 1 && EXP != 8 && j == 4) sstage(i, 3, 3, k);
                        else if (EXP != 1 && EXP != 8 && j == 6) sstage(i, 6, 2, k);
                    }
                }
                __builtin_amdgcn_sched_barrier(0);
                if constexpr (MORE && EXP != 7) {
                    if (j < 2) { tstage(i, j, 6); __builtin_amdgcn_sched_barrier(0); tstage(i, j, 7); }
                    else if (EXP != 1 && EXP != 8 && j == 2) sstage(i, 0, 3, 6);
                    else if (EXP != 1 && EXP != 8 && j == 4) sstage(i, 3, 3, 6);
                    else if (EXP != 1 && EXP != 8 && j == 6) sstage(i, 6, 2, 6);
                    if (j == 7) xload1(ks + 2, i);
                }
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        lds_barrier();
        if (ks < 6) GLOWTTS_TRACE_POINT(2 + ks);
    };
    for (int ks = 0; ks + 1 < nks; ++ks) kstep(ks, std::true_type{});
    kstep(nks - 1, std::false_type{});
    GLOWTTS_TRACE_POINT(8);

    // ---- output transform (AT) in registers, then the gate (utils.py:31-38) and the stores ------------------------------------
    // acc[r][pt][cb][reg]: row lk * 4 + reg of the wave's tanh (r = 0) / sigmoid (r = 1) tile, Winograd tile cb * 16 + lrow.
    // Every global read of the epilogue (keep bytes, biases, conditioning rows) is issued FIRST: one wave per SIMD would otherwise
    // sit through twelve memory round trips one after the other.
    unsigned kt[WINO_CB][4], ksg[WINO_CB][4];
    float bt[4], bs[4], ct[WINO_CB][4], cs[WINO_CB][4];
    int eb[WINO_CB], et0[WINO_CB];
#pragma unroll
    for (int reg = 0; reg < 4; ++reg) {
        const int ch = tile_m * 64 + wave * 16 + lk * 4 + reg;
        bt[reg] = p.bias ? p.bias[ch] : 0.f;
        bs[reg] = p.bias ? p.bias[p.H + ch] : 0.f;
    }
#pragma unroll
    for (int cb = 0; cb < WINO_CB; ++cb) {
        const int n = colg * WINO_COLS + cb * 16 + lrow;
        const bool ok = n < n_tiles;
        const int b = ok ? n / tpu : 0, t0 = ok ? (n - b * tpu) * 4 : 0;     // (clamped: always a valid address)
        eb[cb] = ok ? b : -1; et0[cb] = t0;
#pragma unroll
        for (int reg = 0; reg < 4; ++reg) {
            const int ch = tile_m * 64 + wave * 16 + lk * 4 + reg;
            const long ot = ((long)b * 2 * p.H + ch) * p.T + t0, os = ot + (long)p.H * p.T;
            kt[cb][reg] = ksg[cb][reg] = 0x01010101u;
            if (p.drop) {
                kt[cb][reg] = *reinterpret_cast<const unsigned *>(p.drop + ot);
                ksg[cb][reg] = *reinterpret_cast<const unsigned *>(p.drop + os);
            }
            ct[cb][reg] = cs[cb][reg] = 0.f;
            if (p.cond) { ct[cb][reg] = p.cond[(long)b * 2 * p.H + ch]; cs[cb][reg] = p.cond[(long)b * 2 * p.H + p.H + ch]; }
        }
    }
    // Per column block the 8 accumulator rows of a lane (4 registers x tanh / sigmoid) go through the output transform and the
    // gate IN LOCKSTEP, stage by stage, as pinned instructions (see vadd above): 8 to 32 independent instructions per stage where
    // the compiler's order was one row at a time — a chain of ~40 dependent instructions, two transcendentals deep, twelve times.
    const float kT = -2.f * 1.4426950408889634f, kS = -1.4426950408889634f;
#pragma unroll
    for (int cb = 0; cb < WINO_CB; ++cb) {
        if (eb[cb] < 0) continue;
        const int b = eb[cb], t0 = et0[cb];
        float y[8][4];                                   // row = reg * 2 + r
        {
            float m[8][WINO_P], s12[8], d12[8], s34[8], d34[8], s56[8], d56[8], ta_[8], tb_[8], tc_[8], td_[8], te_[8], tf_[8], tg_[8];
#pragma unroll
            for (int w = 0; w < 8; ++w)
#pragma unroll
                for (int q = 0; q < WINO_P; ++q) m[w][q] = acc[w & 1][q][cb][w >> 1];
#pragma unroll
            for (int w = 0; w < 8; ++w) { s12[w] = vadd(m[w][1], m[w][2]); d12[w] = vsub(m[w][1], m[w][2]); }
#pragma unroll
            for (int w = 0; w < 8; ++w) { s34[w] = vadd(m[w][3], m[w][4]); d34[w] = vsub(m[w][3], m[w][4]); }
#pragma unroll
            for (int w = 0; w < 8; ++w) { s56[w] = vadd(m[w][5], m[w][6]); d56[w] = vsub(m[w][5], m[w][6]); }
#pragma unroll
            for (int w = 0; w < 8; ++w) { ta_[w] = vadd(m[w][0], s12[w]); tb_[w] = vadd(s34[w], s56[w]); tc_[w] = vmulk(2.f, d34[w]); td_[w] = vmulk(4.f, s34[w]); }
#pragma unroll
            for (int w = 0; w < 8; ++w) { te_[w] = vmulk(8.f, d34[w]); tf_[w] = vmulk(0.5f, d56[w]); tg_[w] = vmulk(0.25f, s56[w]); d56[w] = vmulk(0.125f, d56[w]); }
#pragma unroll
            for (int w = 0; w < 8; ++w) { y[w][0] = vadd(ta_[w], tb_[w]); tc_[w] = vadd(d12[w], tc_[w]); td_[w] = vadd(s12[w], td_[w]); te_[w] = vadd(d12[w], te_[w]); }
#pragma unroll
            for (int w = 0; w < 8; ++w) { y[w][1] = vadd(tc_[w], tf_[w]); y[w][2] = vadd(td_[w], tg_[w]); te_[w] = vadd(te_[w], d56[w]); }
#pragma unroll
            for (int w = 0; w < 8; ++w) y[w][3] = vadd(te_[w], m[w][7]);
        }
#pragma unroll
        for (int w = 0; w < 8; ++w)
#pragma unroll
            for (int j = 0; j < 4; ++j) y[w][j] = vadd(y[w][j], (w & 1) ? bs[w >> 1] : bt[w >> 1]);
        if (p.drop) {                                    // dropout on the pre-activation (layers.py:147): keep byte -> 0 / 1 -> x scale
            float f[8][4];
#pragma unroll
            for (int w = 0; w < 8; ++w) {
                const unsigned kk = (w & 1) ? ksg[cb][w >> 1] : kt[cb][w >> 1];
                f[w][0] = vubyte<0>(kk); f[w][1] = vubyte<1>(kk); f[w][2] = vubyte<2>(kk); f[w][3] = vubyte<3>(kk);
            }
#pragma unroll
            for (int w = 0; w < 8; ++w)
#pragma unroll
                for (int j = 0; j < 4; ++j) f[w][j] = vmulk(p.drop_scale, f[w][j]);
#pragma unroll
            for (int w = 0; w < 8; ++w)
#pragma unroll
                for (int j = 0; j < 4; ++j) y[w][j] = vmul(y[w][j], f[w][j]);
        }
        if (p.cond) {
#pragma unroll
            for (int w = 0; w < 8; ++w)
#pragma unroll
                for (int j = 0; j < 4; ++j) y[w][j] = vadd(y[w][j], (w & 1) ? cs[cb][w >> 1] : ct[cb][w >> 1]);
        }
#pragma unroll
        for (int w = 0; w < 8; ++w)
#pragma unroll
            for (int j = 0; j < 4; ++j) y[w][j] = vmulk((w & 1) ? kS : kT, y[w][j]);
#pragma unroll
        for (int w = 0; w < 8; ++w)
#pragma unroll
            for (int j = 0; j < 4; ++j) y[w][j] = vexp2(y[w][j]);
#pragma unroll
        for (int w = 0; w < 8; ++w)
#pragma unroll
            for (int j = 0; j < 4; ++j) y[w][j] = vaddk(1.f, y[w][j]);
#pragma unroll
        for (int w = 0; w < 8; ++w)
#pragma unroll
            for (int j = 0; j < 4; ++j) y[w][j] = vrcp(y[w][j]);
#pragma unroll
        for (int reg = 0; reg < 4; ++reg)
#pragma unroll
            for (int j = 0; j < 4; ++j) y[reg * 2][j] = vtanh_from_rcp(y[reg * 2][j]);      // tanh = 2 / (1 + e^-2x) - 1
#pragma unroll
        for (int reg = 0; reg < 4; ++reg) {
            const int ch = tile_m * 64 + wave * 16 + lk * 4 + reg;
            const long ot = ((long)b * 2 * p.H + ch) * p.T + t0, os = ot + (long)p.H * p.T;
            const float *th = y[reg * 2], *sg = y[reg * 2 + 1];
            *reinterpret_cast<float4 *>(p.y0 + ((long)b * p.H + ch) * p.T + t0) = make_float4(th[0] * sg[0], th[1] * sg[1], th[2] * sg[2], th[3] * sg[3]);
            if (p.y1) {
                *reinterpret_cast<float4 *>(p.y1 + ot) = make_float4(th[0], th[1], th[2], th[3]);
                *reinterpret_cast<float4 *>(p.y1 + os) = make_float4(sg[0], sg[1], sg[2], sg[3]);
            }
        }
    }
    if (EXP == 7 && dmy[0] + dmy[1] + dmy[2] + dmy[3] + dmy[4] + dmy[5] == 1234.5f) p.y0[0] = 0.f;
    GLOWTTS_TRACE_POINT(10);
}

// ---- host side --------------------------------------------------------------------------------------------------------------
struct WinoBinding {
    const float *wp = nullptr;     // the packed-weight buffer the U planes were made from
    long n = 0;
    const unsigned short *planes = nullptr;
    long stride = 0;
};
static thread_local WinoBinding t_wino;
static std::atomic<long> g_wino_launches{0};      // launches of the Winograd kernel so far (tests / bench.py ask: did it run?)

// -1 = not handled (switch off, no U planes bound for these weights, shape or alignment outside the kernel's)
int conv_wino_gate_dispatch(ConvGemmParams &p, hipStream_t s) {
    if (!knob(K_WINO) || conv_math_forward() != 3) return -1;
    const WinoBinding &w = t_wino;
    if (w.wp == nullptr || p.wp < w.wp || p.wp >= w.wp + w.n) return -1;
    if (p.taps != 5 || p.dil != 1 || p.pad != 2 || p.xb || p.yb || p.x2 || p.mask_in) return -1;
    if (p.H % 64 != 0 || p.Cin % 32 != 0 || p.M != 2 * p.H || p.T % 4 != 0) return -1;
    if (!aligned16(p.x) || !aligned16(p.y0) || !aligned16(p.y1) || !aligned16(p.drop) || (long)p.B * p.x_bs * 4 >= 0x7ffffff0L) return -1;
    const unsigned short *U = w.planes + wino_u_offset(p.wp - w.wp);
    const size_t lds = (size_t)2 * WINO_IMG_DW * sizeof(float);
    static LdsLimit attr_max_e;
    if (int rc_ = attr_max_e.ensure(reinterpret_cast<const void *>(&wino_gate_fwd_kernel<0>), lds, "glowtts_conv_gate_fwd (Winograd)")) return rc_;
    const int n_tiles = p.B * (p.T / 4);
    dim3 grid((unsigned)(((n_tiles + WINO_COLS - 1) / WINO_COLS) * (p.H / 64)));
#ifdef GLOWTTS_TRACE
    // timing experiments of the tuning build (tools/wino_bench.py; results WRONG): GLOWTTS_WINO = 1 + 2 x {1: no split / store of the
    // next image, 3: no staging at all, 7: six independent FMAs behind every MFMA pair instead of the staging, 8: the input loads
    // are waited for and nothing is computed}
    const int exp = knob(K_WINO) >> 1;
    if (exp == 1 || exp == 3 || exp == 7 || exp == 8) {
        static LdsLimit attr_x[4];
        const int slot = exp == 1 ? 0 : exp == 3 ? 1 : exp == 7 ? 2 : 3;
        const void *fn = exp == 1 ? reinterpret_cast<const void *>(&wino_gate_fwd_kernel<1>) : exp == 3 ? reinterpret_cast<const void *>(&wino_gate_fwd_kernel<3>)
                       : exp == 7 ? reinterpret_cast<const void *>(&wino_gate_fwd_kernel<7>) : reinterpret_cast<const void *>(&wino_gate_fwd_kernel<8>);
        if (int rc_ = attr_x[slot].ensure(fn, lds, "glowtts_conv_gate_fwd (Winograd)")) return rc_;
        if (exp == 1) hipLaunchKernelGGL(wino_gate_fwd_kernel<1>, grid, dim3(256), lds, s, p, U, w.stride);
        else if (exp == 3) hipLaunchKernelGGL(wino_gate_fwd_kernel<3>, grid, dim3(256), lds, s, p, U, w.stride);
        else if (exp == 7) hipLaunchKernelGGL(wino_gate_fwd_kernel<7>, grid, dim3(256), lds, s, p, U, w.stride);
        else hipLaunchKernelGGL(wino_gate_fwd_kernel<8>, grid, dim3(256), lds, s, p, U, w.stride);
        GLOWTTS_LAUNCH_CHECK("glowtts_conv_gate_fwd (Winograd)");
    }
#endif
    g_wino_launches.fetch_add(1, std::memory_order_relaxed);
    hipLaunchKernelGGL(wino_gate_fwd_kernel<0>, grid, dim3(256), lds, s, p, U, w.stride);
    GLOWTTS_LAUNCH_CHECK("glowtts_conv_gate_fwd (Winograd)");
}

}  // namespace glowtts

using namespace glowtts;

extern "C" long glowtts_wino_launches(void) { return g_wino_launches.load(std::memory_order_relaxed); }

extern "C" long glowtts_wino_plane_elems(long n) { return wino_u_offset(n) + 8; }

extern "C" int glowtts_wino_weights(const float *wp, long n, const long *table, int n_conv, unsigned short *planes, long plane_stride,
                                    glowtts_stream_t stream) {
    GLOWTTS_CHECK_ARG(wp && table && planes && n > 0 && n_conv > 0 && plane_stride >= glowtts_wino_plane_elems(n),
                      "glowtts_wino_weights: bad arguments (plane stride %ld for %ld packed floats needs %ld)", plane_stride, n,
                      glowtts_wino_plane_elems(n));
    hipLaunchKernelGGL(wino_weights_kernel, dim3(36, (unsigned)n_conv), dim3(256), 0, (hipStream_t)stream, wp, table, n_conv, planes, plane_stride);
    GLOWTTS_LAUNCH_CHECK("glowtts_wino_weights");
}

extern "C" int glowtts_conv_bind_wino(const float *wp, long n, const unsigned short *planes, long plane_stride) {
    GLOWTTS_CHECK_ARG(wp == nullptr || (planes && n > 0 && plane_stride >= glowtts_wino_plane_elems(n)), "glowtts_conv_bind_wino: bad arguments");
    t_wino.wp = wp;
    t_wino.n = wp ? n : 0;
    t_wino.planes = wp ? planes : nullptr;
    t_wino.stride = wp ? plane_stride : 0;
    return 0;
}

#ifdef GLOWTTS_TRACE
extern "C" int glowtts_debug_trace_read_wino(unsigned long long *host, int n_words, int clear) {
    hipError_t e = hipMemcpyFromSymbol(host, HIP_SYMBOL(glowtts::g_trace), (size_t)n_words * 8);
    if (e != hipSuccess) return (int)e;
    if (clear) {
        static unsigned long long zeros[8192 * 16];
        e = hipMemcpyToSymbol(HIP_SYMBOL(glowtts::g_trace), zeros, sizeof(zeros));
    }
    return (int)e;
}
#endif
