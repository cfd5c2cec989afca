// convgemm_split.hip — the forward-type implicit-GEMM convolution of convgemm.hip with each fp32 operand split into
// NS bf16 planes and the products formed on the bf16 matrix pipe (v_mfma_f32_16x16x32_bf16, fp32 accumulate).
//
// Why: the native fp32 MFMA (16x16x4, 32 cycles for 2 048 FLOP) caps the WN convolutions at 157 TFLOP/s and the
// kernels of convgemm.hip sit at 66-73 % of that.  The bf16 instruction does 16 384 FLOP in 16 cycles.  An fp32 value
// is the exact sum of three bf16 values (8 + 8 + 8 mantissa bits: h = top bits, m = top bits of v - h, l = v - h - m),
// so   x * w = (xh + xm + xl) * (wh + wm + wl)   and the six products  hh, hm, mh, hl, lh, mm  carry everything down
// to 2^-24 |x w| — the size of ONE fp32 rounding; every product of two bf16 is exact in fp32 and the sums are fp32, as
// in the native kernel.  Six bf16 MFMAs per 32-deep step cost 96 cycles where the fp32 path needs 256.
//   NS = 3 ("bf16x6"): fp32-equivalent (measured: same error against an fp64 reference as the native kernel);
//   NS = 2 ("bf16x3"): hh + hl + lh, products good to 2^-16;      NS = 1: plain bf16 operands (config 3's arithmetic).
// Selected with glowtts_conv_math (the Python host's default is NS = 3 for forward-type and weight-gradient kernels alike;
// the library itself starts in native fp32, which stays the reference for parity).
//
// Data flow (differences from convgemm_wd_kernel, whose tiling, staging map, ring and epilogues are kept):
//   weights   : the packed fp32 buffer [tap][g][M][16] is split ONCE per step by split_weights_kernel into bf16 planes
//               with the same element order (glowtts_conv_split_weights into a caller-owned buffer, bound to the
//               calling thread by glowtts_conv_bind_planes around the launches that use it); a lane
//               takes 4 channels of group g and 4 of group g+1 (two 8-byte buffer loads per plane) = one bf16x8 A
//               operand: k is relabelled so that lane slot lk consumes channels 4 lk..4 lk+3 of both groups.
//   activations: split while they are stored to LDS; plane image [g/2][frame][40 bf16]: a row holds the 32 channels of
//               a group pair in the A operand's order (pitch 80 B, the conflict-free pitch of the fp32 image), so one
//               ds_read_b128 per plane is a B operand.
#include <atomic>
#include "convgemm_common.hpp"
#include "split_planes.hpp"

namespace glowtts {

__global__ __launch_bounds__(256) void split_weights_kernel(const float *__restrict__ w, unsigned short *__restrict__ planes,
                                                            long n, long plane_stride, int ns) {
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
        const float v = w[i];
        if (ns == 3) {
            unsigned o[3];
            split_planes<3>(v, o);
            planes[i] = (unsigned short)o[0]; planes[plane_stride + i] = (unsigned short)o[1];
            planes[2 * plane_stride + i] = (unsigned short)o[2];
        } else if (ns == 2) {
            unsigned o[2];
            split_planes<2>(v, o);
            planes[i] = (unsigned short)o[0]; planes[plane_stride + i] = (unsigned short)o[1];
        } else {
            unsigned o[1];
            split_planes<1>(v, o);
            planes[i] = (unsigned short)o[0];
        }
    }
}

// An 8-byte LDS store as ds_write2_b32 of two adjacent dwords.  Not ds_write_b64 / ds_write2st64_b64: on gfx950 a kernel that
// issues 64-bit LDS stores next to bf16 MFMAs makes OTHER waves' packed-fp32 VALU results wrong (DESIGN.md lesson 12; this library
// has no packed-fp32 code, but a collective or a framework kernel on the same CU may).  hipcc does not count an asm LDS store:
// the staging code waits for lgkmcnt(0) itself before its barriers.
__device__ __forceinline__ void lds_store8(void *p, int lo, int hi) {
    asm volatile("ds_write2_b32 %0, %1, %2 offset1:1"
                 :: "v"((unsigned)(size_t)(__attribute__((address_space(3))) void *)p), "v"(lo), "v"(hi) : "memory");
}
__device__ __forceinline__ void lds_stores_done() { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); }

// IOB: 0 = fp32 tensors (operands split / rounded while they are staged); 1 = x and the epilogue's tensors are bf16 in HBM
// (NS = 1 only: nothing to split — 8-byte loads go to LDS as they are, results leave as bf16); 2 = x bf16, epilogue fp32
// (the coupling's end conv: its output (m, logs) feeds the log-determinant and stays fp32).
// NSA (bf16 tensors only): planes of the WEIGHTS.  1 = weights rounded to bf16; 3 = the exact fp32 weights as h + m + l
// against the one activation plane (three MFMAs per step, small terms first) — used by the convolutions with fp32 results:
// the coupling's end conv, whose `logs` rows are summed over every frame into the log-determinant (a weight rounding is
// the only error there that repeats in every frame and so adds up coherently instead of averaging out), and the start
// conv's input gradient that is added into an fp32 flow gradient.  Both are 1x1 convolutions: the two extra MFMAs per
// step are noise in their run time.  (At configs[2] sizes the log-det error is dominated by the round-off of the stored
// hidden tensors; this removes the one term that would grow with T.)
template <int NS, int RTW, int NCT, int EPI, int TAPS, int IOB = 0, int NSA = NS>
__global__ __launch_bounds__(256, 2) void convgemm_split_kernel(ConvGemmParams p, const unsigned short *__restrict__ wpl,
                                                                long plane_stride) {
    static_assert(IOB == 0 || NS == 1, "bf16 tensors carry one plane");
    static_assert(IOB != 0 || NSA == NS, "operand planes pair up in the split arithmetic");
    constexpr bool XB = IOB != 0, YB = IOB == 1;
    constexpr int ES = XB ? 2 : 4;                   // bytes per activation element in HBM
    constexpr int WGR = 64 * RTW, NT = 16 * NCT, XC = NT + 16, KG = 6, G2C = KG / 2;
    constexpr int RP = 40;                           // bf16 per LDS row (32 used + 8 pad): 80 B = 20 dwords
    constexpr int PLANE16 = G2C * XC * RP;           // bf16 elements per plane image
    constexpr int X4 = KG * 16 * (XC / 4);
    constexpr int S = G2C * TAPS;                    // 32-deep MFMA steps per chunk (a multiple of 3: static ring)
    static_assert(S % 3 == 0, "ring of 3");
    extern __shared__ __align__(16) float smem[];
    unsigned short *Xh = reinterpret_cast<unsigned short *>(smem);      // [NS][G2C][XC][RP]

    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int lrow = lane & 15, lk = lane >> 4;
    const int ntile_t = (p.T + NT - 1) / NT;
    int bx = blockIdx.x, by = blockIdx.y;
    if (p.wg_order) {
        // workgroup ids are dealt round-robin over the 8 XCDs in id order = (x, then y): the R row tiles of a frame tile — which read
        // the same activation tile — land on one XCD (gridDim.x % 8 == 0) but a whole grid row apart in dispatch order.  Re-numbered:
        // ids i, i + 8, .., i + 8 (R - 1) are the R row tiles of one frame tile (same XCD, dispatched back to back)
        const int id = blockIdx.x + blockIdx.y * gridDim.x, R = gridDim.y;
        by = (id >> 3) % R;
        bx = (id / (8 * R)) * 8 + (id & 7);
    }
    const int b = bx / ntile_t;
    const int t0 = (bx - b * ntile_t) * NT;
    const int tile_m = by;
    const int off = (4 - (p.pad & 3)) & 3;
    const int ts = t0 - p.pad - off;
    const int G = (p.Cin + 15) / 16;
    const int nchunks = (G + KG - 1) / KG;

    auto grow = [&](int lr) -> int {
        if (EPI == EPI_GATE) return lr < 64 ? tile_m * 64 + lr : p.H + tile_m * 64 + (lr - 64);
        return tile_m * WGR + lr;
    };
    auto row_ok = [&](int lr) -> bool {
        if (EPI == EPI_GATE) return tile_m * 64 + (lr & 63) < p.H;
        return tile_m * WGR + lr < p.M;
    };
    auto ltile = [&](int r) -> int { return (EPI == EPI_GATE) ? (r == 0 ? wave : 4 + wave) : wave * RTW + r; };

    f32x4 acc[RTW][NCT];
#pragma unroll
    for (int r = 0; r < RTW; ++r)
#pragma unroll
        for (int c = 0; c < NCT; ++c) acc[r][c] = f32x4{0.f, 0.f, 0.f, 0.f};

    // ---- weights: bf16 planes, element order of the fp32 packing; range-checked buffer loads (rows beyond M and groups
    // beyond G return zeros through an out-of-range offset)
    const int wbytes = TAPS * G * p.M * 32;          // one plane
    __amdgpu_buffer_rsrc_t wrs[NSA];
#pragma unroll
    for (int pl = 0; pl < NSA; ++pl)
        wrs[pl] = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned short *>(wpl + pl * plane_stride), 0, wbytes, 0x00020000);
    int wvo[RTW];
#pragma unroll
    for (int r = 0; r < RTW; ++r) {
        const int lr = ltile(r) * 16 + lrow;
        wvo[r] = row_ok(lr) ? (grow(lr) * 16 + lk * 4) * 2 : wbytes;
    }
    const int wtap = G * p.M * 32, wgrp = p.M * 32;  // bytes
    i32x4 a[3][RTW][NSA] = {};
    auto wload = [&](int c, int s, int slot) {       // weights of step s of chunk c (s may run past the chunk: next chunk)
        if ((GLOWTTS_EXP_BITS(p.exp) & 4) && (c > 0 || s > 2)) return;       // (timing experiment: no weight loads after the first three)
        if (s >= S) { s -= S; c += 1; }
        const int g = c * KG + 2 * (s / TAPS), tap = s % TAPS;
        const int so0 = g < G ? tap * wtap + g * wgrp : wbytes;
        const int so1 = g + 1 < G ? tap * wtap + (g + 1) * wgrp : wbytes;
#pragma unroll
        for (int r = 0; r < RTW; ++r)
#pragma unroll
            for (int pl = 0; pl < NSA; ++pl) {
                const i32x2 lo = __builtin_bit_cast(i32x2, __builtin_amdgcn_raw_buffer_load_b64(wrs[pl], wvo[r], so0, 0));
                const i32x2 hi = __builtin_bit_cast(i32x2, __builtin_amdgcn_raw_buffer_load_b64(wrs[pl], wvo[r], so1, 0));
                a[slot][r][pl] = i32x4{lo[0], lo[1], hi[0], hi[1]};
            }
    };

    // ---- activations: the staging map of convgemm_wd_kernel (thread = channel kk of a group, frame quad qq, group
    // parity gsel; pieces (group gsel + 2 gi, quad qq + 8 jq)); the group pair of piece gi is gi, its half is gsel
    const float *mk = p.mask ? p.mask + (long)b * p.T : nullptr;
    float *Ms = smem + NS * PLANE16 / 2;             // [XC]
    constexpr int NQ = (XC / 4 + 7) / 8, NG = KG / 2;
    static_assert(NQ * NG * 256 >= X4, "piece map covers the chunk");
    const int kk = tid & 15, qq = (tid >> 4) & 7, gsel = tid >> 7;
    const int c_first = p.x2 ? p.x_split : p.Cin;
    const int xbytes = c_first * p.T * ES;
    const char *xb8 = reinterpret_cast<const char *>(p.x) + (long)b * p.x_bs * ES;
    const __amdgpu_buffer_rsrc_t xrs = __builtin_amdgcn_make_buffer_rsrc(const_cast<char *>(xb8), 0, xbytes, 0x00020000);
    const int x2bytes = p.x2 ? (p.Cin - p.x_split) * p.T * ES : 0;
    const __amdgpu_buffer_rsrc_t xrs2 = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<char *>(p.x2 ? reinterpret_cast<const char *>(p.x2) + (long)b * p.x2_bs * ES : xb8), 0, x2bytes, 0x00020000);
    int xvo[NQ];
#pragma unroll
    for (int jq = 0; jq < NQ; ++jq) {
        const int t = ts + (qq + 8 * jq) * 4;
        const bool ok = (qq + 8 * jq < XC / 4) && t >= 0 && t < p.T;
        xvo[jq] = ok ? ((gsel * 16 + kk) * p.T + t) * ES : 0x7fffffff;
    }
    const int dbase = (qq * 4) * RP + (kk >> 2) * 8 + gsel * 4 + (kk & 3);     // bf16 index inside a plane image
    i32x2 xregb[XB ? NG : 1][XB ? NQ : 1];           // bf16 tensors: 4 frames of one channel = 8 bytes
    // fp32 tensors (round 3): a piece is 4 channels x 4 frames — four 16-byte loads; v_cvt_pk_bf16_f32 of two neighbouring
    // channels of one frame IS a dword of the plane image, so a frame of the piece is ONE 8-byte LDS store per plane (12 per 16
    // values; the channel-per-thread map above needs 12 two-byte stores per 4 values, which made the staging phase of a 1x1
    // convolution as long as its MFMA loop).  Thread bits: c4l (3: channel quad of a group pair), fql (2), hl (3); a round covers
    // hl + 8 r of the HI = (group pairs) x (frame quads / 4) values; 16 lanes with equal hl store two rows 4 frames apart x 8
    // even dword offsets: all 32 banks once.  What does not fill a round goes as quarter pieces (1 channel x 4 frames).
    constexpr int NFQ = (TAPS > 1 ? XC : NT) / 4;    // frame quads staged per channel (a 1x1 convolution has no halo)
    constexpr int NFQH = NFQ / 4, HI = G2C * NFQH;
    constexpr int F0 = HI / 8, LH0 = HI - 8 * F0;
    constexpr int FR = XB ? 0 : F0 + (LH0 >= 5 ? 1 : 0), LH = (XB || LH0 >= 5) ? 0 : LH0, QR = (LH + 1) / 2;
    static_assert(NFQ % 4 == 0, "frame quads in fours");
    f32x4 xf[FR > 0 ? FR : 1][4] = {}, xq[QR > 0 ? QR : 1] = {};
    unsigned xfo[FR > 0 ? FR : 1], xqo[QR > 0 ? QR : 1];
    int xfd[FR > 0 ? FR : 1], xqd[QR > 0 ? QR : 1], xfq[FR > 0 ? FR : 1], xqq[QR > 0 ? QR : 1];
    const unsigned rowb = (unsigned)p.T * 4u;
    if constexpr (!XB) {
        const int c4l = tid & 7, c4 = c4l & 3, gs = c4l >> 2, fql = (tid >> 3) & 3;
#pragma unroll
        for (int r = 0; r < FR; ++r) {
            const int hi = (tid >> 5) + 8 * r;
            const int gp = hi / NFQH, fq = (hi - gp * NFQH) * 4 + fql;
            const int t = ts + fq * 4;
            const bool ok = hi < HI && t >= 0 && t < p.T;
            xfo[r] = ok ? (unsigned)((gp * 32 + gs * 16 + c4 * 4) * p.T + t) * 4u : 0x7fffffffu;
            xfd[r] = hi < HI ? (gp * XC + fq * 4) * RP + c4 * 8 + gs * 4 : -1;
            xfq[r] = fq;
        }
#pragma unroll
        for (int r = 0; r < QR; ++r) {
            const int ch = tid & 3, c4q = (tid >> 2) & 3, gsq = (tid >> 4) & 1, fqlq = (tid >> 5) & 3, hl = (tid >> 7) + 2 * r;
            const int hi = 8 * F0 + hl;
            const int gp = hi / NFQH, fq = (hi - gp * NFQH) * 4 + fqlq;
            const int t = ts + fq * 4;
            const bool ok = hl < LH && t >= 0 && t < p.T;
            xqo[r] = ok ? (unsigned)((gp * 32 + gsq * 16 + c4q * 4 + ch) * p.T + t) * 4u : 0x7fffffffu;
            xqd[r] = hl < LH ? (gp * XC + fq * 4) * RP + c4q * 8 + gsq * 4 + ch : -1;
            xqq[r] = fq;
        }
    }
    if (p.mask_in && tid < XC) {
        const int t = ts + tid;
        Ms[tid] = (t >= 0 && t < p.T) ? mk[t] : 0.f;
    }
    auto xload = [&](int c, int part = -1) {           // part: -1 = everything, r < FR = round r, FR = the quarter pieces
        if (GLOWTTS_EXP_BITS(p.exp) & 1) return;         // (timing experiment: no activation loads)
        const bool second = p.x2 != nullptr && c * KG * 16 >= p.x_split;
        const int cbase = c * KG * 16 - (second ? p.x_split : 0);
        if constexpr (XB) {
#pragma unroll
            for (int gi = 0; gi < NG; ++gi)
#pragma unroll
                for (int jq = 0; jq < NQ; ++jq)
                    xregb[gi][jq] = __builtin_bit_cast(i32x2, __builtin_amdgcn_raw_buffer_load_b64(
                        second ? xrs2 : xrs, xvo[jq], (cbase + 2 * gi * 16) * p.T * ES, 0));
        } else {
#pragma unroll
            for (int r = 0; r < FR; ++r)
                if (part < 0 || part == r) {
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        int vo = (int)(xfo[r] + i * rowb);   /* an out-of-range offset stays out of range */
                        int so = cbase * p.T * ES;
                        if (GLOWTTS_EXP_BITS(p.exp) & 128) { vo &= 0xff0; so = 0; }     // (timing experiment: every load hits the same 4 KB)
                        xf[r][i] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(second ? xrs2 : xrs, vo, so, 0));
                    }
                }
            if (part < 0 || part == FR) {
#pragma unroll
                for (int r = 0; r < QR; ++r)
                    xq[r] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(second ? xrs2 : xrs, (int)xqo[r], cbase * p.T * ES, 0));
            }
        }
    };
    auto xstore = [&]() {
        if (GLOWTTS_EXP_BITS(p.exp) & 2) return;         // (timing experiment: no split / LDS stores)
        if constexpr (XB) {
#pragma unroll
            for (int gi = 0; gi < NG; ++gi)
#pragma unroll
                for (int jq = 0; jq < NQ; ++jq)
                    if (qq + 8 * jq < XC / 4) {
                        unsigned short *d = Xh + dbase + (gi * XC + 32 * jq) * RP;
                        // already bf16: the 0 / 1 mask selects, nothing is rounded
                        unsigned w0 = (unsigned)xregb[gi][jq][0], w1 = (unsigned)xregb[gi][jq][1];
                        if (p.mask_in) {
                            const f32x4 m = *reinterpret_cast<const f32x4 *>(Ms + (qq + 8 * jq) * 4);
                            w0 = (m[0] != 0.f ? (w0 & 0xffffu) : 0u) | (m[1] != 0.f ? (w0 & 0xffff0000u) : 0u);
                            w1 = (m[2] != 0.f ? (w1 & 0xffffu) : 0u) | (m[3] != 0.f ? (w1 & 0xffff0000u) : 0u);
                        }
                        d[0] = (unsigned short)w0; d[RP] = (unsigned short)(w0 >> 16);
                        d[2 * RP] = (unsigned short)w1; d[3 * RP] = (unsigned short)(w1 >> 16);
                    }
        } else {
#pragma unroll
            for (int r = 0; r < FR; ++r)
                if (xfd[r] >= 0) {
                    f32x4 m = {1.f, 1.f, 1.f, 1.f};
                    if (p.mask_in) m = *reinterpret_cast<const f32x4 *>(Ms + xfq[r] * 4);
#pragma unroll
                    for (int f = 0; f < 4; ++f) {
                        unsigned oa[NS], ob[NS];
                        float a0 = xf[r][0][f], a1 = xf[r][1][f], a2 = xf[r][2][f], a3 = xf[r][3][f];
                        if (p.mask_in) { a0 *= m[f]; a1 *= m[f]; a2 *= m[f]; a3 *= m[f]; }
                        split_planes2<NS>(a0, a1, oa);
                        split_planes2<NS>(a2, a3, ob);
#pragma unroll
                        for (int pl = 0; pl < NS; ++pl) lds_store8(Xh + pl * PLANE16 + xfd[r] + f * RP, (int)oa[pl], (int)ob[pl]);
                    }
                }
#pragma unroll
            for (int r = 0; r < QR; ++r)
                if (xqd[r] >= 0) {
                    unsigned short *d = Xh + xqd[r];
                    f32x4 v = xq[r];
                    if (p.mask_in) v *= *reinterpret_cast<const f32x4 *>(Ms + xqq[r] * 4);
#pragma unroll
                    for (int f = 0; f < 4; f += 2) {
                        unsigned o[NS];
                        split_planes2<NS>(v[f], v[f + 1], o);
#pragma unroll
                        for (int pl = 0; pl < NS; ++pl) {
                            d[pl * PLANE16 + f * RP] = (unsigned short)o[pl];
                            d[pl * PLANE16 + (f + 1) * RP] = (unsigned short)(o[pl] >> 16);
                        }
                    }
                }
            lds_stores_done();                           // the asm stores are not counted by the compiler's own waits
        }
    };
    const float *xd = smem + (off + lrow) * (RP / 2) + lk * 4;      // dword view of a plane image
    i32x4 bv[2][NS];
    auto bfetch = [&](int q, int slot) {             // q = step * NCT + column tile
        const int s = q / NCT, cc = q - s * NCT;
        const int g2 = s / TAPS, tap = s % TAPS;
#pragma unroll
        for (int pl = 0; pl < NS; ++pl)
            bv[slot][pl] = *reinterpret_cast<const i32x4 *>(xd + pl * (PLANE16 / 2) + (g2 * XC + cc * 16 + tap * p.dil) * (RP / 2));
    };

    GLOWTTS_TRACE_POINT(0);
    wload(0, 0, 0);
    wload(0, 1, 1);
    wload(0, 2, 2);
    xload(0);
    if (p.mask_in) __syncthreads();
    xstore();
    __syncthreads();
    GLOWTTS_TRACE_POINT(1);
    for (int c = 0; c < nchunks; ++c) {
        const bool more = c + 1 < nchunks;
        bfetch(0, 0);
#pragma unroll
        for (int s = 0; s < S; ++s) {
#pragma unroll
            for (int cc = 0; cc < NCT; ++cc) {
                const int q = s * NCT + cc;
                if (q + 1 < S * NCT) bfetch(q + 1, (q + 1) & 1);
                __builtin_amdgcn_sched_barrier(0);  // the next tile's LDS reads stay ahead of this tile's MFMAs
#pragma unroll
                for (int r = 0; r < RTW; ++r) {
                    if constexpr (IOB != 0) {
#pragma unroll
                        for (int k = NSA - 1; k >= 0; --k)          // weight planes l, m, h against the one activation plane
                            acc[r][cc] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(
                                __builtin_bit_cast(bf16x8, a[s % 3][r][k]), __builtin_bit_cast(bf16x8, bv[q & 1][0]), acc[r][cc], 0, 0, 0);
                    } else {
#pragma unroll
                        for (int k = 0; k < n_products(NS); ++k)
                            acc[r][cc] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(
                                __builtin_bit_cast(bf16x8, a[s % 3][r][product_a(NS, k)]),
                                __builtin_bit_cast(bf16x8, bv[q & 1][product_b(NS, k)]), acc[r][cc], 0, 0, 0);
                    }
                }
                __builtin_amdgcn_sched_barrier(0);
            }
            wload(c, s + 3, s % 3);                 // refill the slot just consumed: three steps of lead
            if (more) {
#ifdef GLOWTTS_TRACE
                const int e = p.exp;
                if (e & 16) { if (s == S / 2) xload(c + 1); }                       // (experiments on WHEN the next chunk's loads are issued)
                else if (e & 32) { if (s == S - 4) xload(c + 1); }
                else if (e & 64) { if (s % 2 == 0 && s / 2 <= FR) xload(c + 1, s / 2); }
                else if (s == 0) xload(c + 1);
#else
                if (s == 0) xload(c + 1);
#endif
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        GLOWTTS_TRACE_POINT(2 + 2 * (c & 3));
        __syncthreads();      // (an LDS-only barrier here — the weight ring's loads stay in flight across it — measured: no gain, round 5)
        if (more) {
            xstore();
            __syncthreads();
        }
        GLOWTTS_TRACE_POINT(3 + 2 * (c & 3));
    }
    if (GLOWTTS_EXP_BITS(p.exp) & 8) {                   // (timing experiment: no epilogue: one store keeps the accumulators alive)
        if (acc[0][0][0] == 1234.5f) p.y0[0] = acc[RTW - 1][NCT - 1][1];
    } else if (p.vec_epilogue) {
        conv_epilogue_lds<RTW, NCT, EPI, YB>(p, acc, smem, b, t0, tile_m, wave, lane);
    } else {
        conv_epilogue<RTW, NCT, EPI, YB>(p, acc, b, t0, tile_m, wave, lane);
    }
    GLOWTTS_TRACE_POINT(10);
}

// ---------------------------------------------------------------------------------------------------------------
// weight gradient on bf16 planes (the frame-packed kernel of convgemm.hip: contraction over frames, dilation 1, 'same'
// padding).  Both operands are activations, split while they are stored to LDS (4 frames -> one ds_write_b64 per plane).
// A 32-frame MFMA step: lane slot lk takes frames 8 lk .. 8 lk + 7 of the step, for x rows (A) and d rows (B) alike.
// The tap shift of an x row is a shift by whole bf16 elements inside a lane's window of 16 frames (two aligned
// ds_read_b128): even shifts are a choice of registers, odd shifts one v_alignbit per register.
// Chunks stay 80 frames (T' = 400 = 5 chunks): the third step of a chunk is half zeros (d rows are zero past frame 80).
// ---------------------------------------------------------------------------------------------------------------
// PRE: x and d arrive as bf16 planes already (p.xpl / p.dpl, written by glowtts_split_planes or a producer's epilogue):
// staging is then 8-byte loads straight into 8-byte LDS stores, with no vector work at all.
template <int NS, int TAPS, int NGRP, int MT, bool PRE = false>
__global__ __launch_bounds__(256, 2) void convwrw_split_kernel(ConvWrwParams p) {
    constexpr int MR = 16 * MT;
    constexpr int CT = 16 * NGRP, NSTEP = (CT + 31) / 32, CTP = NSTEP * 32;
    constexpr int PAD = (TAPS - 1) / 2, OFF = (4 - (PAD & 3)) & 3;
    constexpr int XWL = CT + 8;                       // frames loaded per x row: window [tc - PAD - OFF, + XWL)
    constexpr int XP16 = 104, DP16 = 104;             // bf16 pitches (== 8 mod 16: conflict-free ds_read_b128), >= CTP + 8
    static_assert(CTP + 8 <= XP16 && XWL % 4 == 0 && TAPS - 1 + OFF < 8, "window fits");
    constexpr int XPLANE = 64 * XP16, DPLANE = MR * DP16;
    constexpr int X4 = 64 * (XWL / 4), D4 = MR * (CT / 4);
    constexpr int NX = (X4 + 255) / 256, ND = (D4 + 255) / 256;
    extern __shared__ __align__(16) float smem[];
    unsigned short *Xh = reinterpret_cast<unsigned short *>(smem);          // [NS][64][XP16]
    unsigned short *Dh = Xh + NS * XPLANE;                                  // [NS][MR][DP16]
    float *Mx = smem + (NS * (XPLANE + DPLANE)) / 2;                        // [XWL]
    float *Md = Mx + XWL;                                                   // [CT]
    float *rowacc = Md + CT;                                                // [64]
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int lrow = lane & 15, lk = lane >> 4;
    const int nkt = (p.Cin + 63) / 64;
    const int ntiles = gridDim.x, nwg = gridDim.x * gridDim.z;
    const int id = blockIdx.x + blockIdx.z * gridDim.x;
    const int xcd = id & 7, slot = id >> 3;
    int item = xcd * (nwg >> 3) + min(xcd, nwg & 7) + slot;
    if (p.nbatch > 0) {                               // several problems of one shape (ConvWrwParams::nbatch)
        const int per = nwg / p.nbatch, q = item / per;
        item -= q * per;
#pragma unroll
        for (int j = 0; j < ConvWrwParams::kMaxBatch; ++j)         // static indices: a run-time index would put the tables in scratch
            if (j == q) { p.x = p.bx[j]; p.d = p.bd[j]; p.d2 = p.bd2[j]; p.dwp = p.bdwp[j]; p.dbias = p.bdbias[j]; }
    }
    const int tile = item % ntiles, split = item / ntiles;
    const int kt = tile % nkt, mt = tile / nkt;
    const int k0 = kt * 64, m0 = mt * MR;
    const int nct = (p.T + CT - 1) / CT;
    const int c0 = split * p.nb;
    const int nchunks = min(p.B * nct, c0 + p.nb) - c0;

    f32x4 acc[TAPS][MT];
#pragma unroll
    for (int tp = 0; tp < TAPS; ++tp)
#pragma unroll
        for (int i = 0; i < MT; ++i) acc[tp][i] = f32x4{0.f, 0.f, 0.f, 0.f};
    // AG (1x1 convolutions): a wave's 16 x rows are used by that wave alone, so they skip LDS — each lane loads the 8 frames of
    // its MFMA slot straight from global memory (two 16-byte loads per 32-frame step, the next chunk's while this one is
    // multiplied) and splits them in registers.  LDS then carries the d rows only: a third fewer ds_read_b128 per MFMA in a loop
    // that was bound by them (4 waves x 18 reads per step against 24 MFMAs), and half the ds_write traffic of the staging.
    constexpr bool AG = (TAPS == 1) && !PRE;
    f32x4 xreg[(PRE || AG) ? 1 : NX], dreg[PRE ? 1 : ND], mreg;
    f32x4 araw[AG ? NSTEP : 1][2];                     // next chunk's x values of this lane's slots (fp32, as loaded)
    int apl[AG ? NSTEP : 1][NS][4];                    // this chunk's A operands: [step][plane] = 8 bf16
    i32x2 xpr[PRE ? NX : 1][NS], dpr[PRE ? ND : 1][NS];
    float bsum[ND];
#pragma unroll
    for (int i = 0; i < ND; ++i) bsum[i] = 0.f;
    const bool do_bias = (p.dbias != nullptr) && (kt == 0);
    const bool masked = (p.mask != nullptr) || (p.mask_x != nullptr);

    const int xbytes = (int)(((long)(p.B - 1) * p.x_bs + (long)p.Cin * p.T) * 4);
    const bool d_second = p.d2 != nullptr && m0 >= p.d_split;
    const int m_rows = d_second ? p.M - p.d_split : (p.d2 ? p.d_split : p.M);
    const int m_base = d_second ? m0 - p.d_split : m0;
    const long d_bs = d_second ? p.d2_bs : p.d_bs;
    const int dbytes = (int)(((long)(p.B - 1) * d_bs + (long)m_rows * p.T) * 4);
    const __amdgpu_buffer_rsrc_t xrs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(p.x), 0, xbytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t drs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(d_second ? p.d2 : p.d), 0, dbytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t mdrs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(p.mask), 0, p.mask ? p.B * p.T * 4 : 0, 0x00020000);
    const __amdgpu_buffer_rsrc_t mxrs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(p.mask_x), 0, p.mask_x ? p.B * p.T * 4 : 0, 0x00020000);
    constexpr int kOOB = 0x7fffffff;
    auto ld16 = [&](const __amdgpu_buffer_rsrc_t &rs, int byte_off) -> f32x4 {
        return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs, byte_off, 0, 0));
    };
    __amdgpu_buffer_rsrc_t xprs[NS], dprs[NS];
    if (PRE) {
#pragma unroll
        for (int pl = 0; pl < NS; ++pl) {
            xprs[pl] = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned short *>(p.xpl + pl * p.xpl_stride), 0, xbytes / 2, 0x00020000);
            dprs[pl] = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned short *>(p.dpl + pl * p.dpl_stride), 0, dbytes / 2, 0x00020000);
        }
    }
    int xrow[NX], xq[NX], drow[ND], dq[ND];
#pragma unroll
    for (int i = 0; i < NX; ++i) {
        const int idx = tid + i * 256;
        const int q = idx % (XWL / 4), r = idx / (XWL / 4);
        xq[i] = q * 4;
        xrow[i] = (idx < X4 && k0 + r < p.Cin) ? (k0 + r) * p.T : -1;
    }
#pragma unroll
    for (int i = 0; i < ND; ++i) {
        const int idx = tid + i * 256;
        const int q = idx % (CT / 4), r = idx / (CT / 4);
        dq[i] = q * 4;
        drow[i] = (idx < D4 && m_base + r < m_rows) ? (m_base + r) * p.T : -1;
    }
    const bool mx_thread = tid < XWL / 4, md_thread = tid >= 64 && tid < 64 + CT / 4;

    // the images' pad columns (x beyond the loaded window, d beyond the chunk) stay zero for the whole kernel
    for (int i = tid; i < NS * (XPLANE + DPLANE) / 2; i += 256) smem[i] = 0.f;
    __syncthreads();

    auto load_chunk = [&](int c) {
        const int b = (c0 + c) / nct;
        const int tc = ((c0 + c) % nct) * CT;
        const int ts = tc - PAD - OFF;
        const int xb = b * (int)p.x_bs, db = b * (int)d_bs;
        if (PRE) {
#pragma unroll
            for (int i = 0; i < NX; ++i) {
                const int t = ts + xq[i];
                const bool ok = xrow[i] >= 0 && t >= 0 && t < p.T;
#pragma unroll
                for (int pl = 0; pl < NS; ++pl)
                    xpr[i][pl] = __builtin_bit_cast(i32x2, __builtin_amdgcn_raw_buffer_load_b64(xprs[pl], ok ? (xb + xrow[i] + t) * 2 : kOOB, 0, 0));
            }
#pragma unroll
            for (int i = 0; i < ND; ++i) {
                const int t = tc + dq[i];
                const bool ok = drow[i] >= 0 && t < p.T;
#pragma unroll
                for (int pl = 0; pl < NS; ++pl)
                    dpr[i][pl] = __builtin_bit_cast(i32x2, __builtin_amdgcn_raw_buffer_load_b64(dprs[pl], ok ? (db + drow[i] + t) * 2 : kOOB, 0, 0));
            }
            return;
        }
        if constexpr (AG) {
            const int row = k0 + wave * 16 + lrow;
#pragma unroll
            for (int g = 0; g < NSTEP; ++g) {
                const int t = tc + g * 32 + lk * 8;      // (frames past the chunk meet zero d rows; past T the loads return zeros)
                const bool ok = row < p.Cin;
                araw[g][0] = ld16(xrs, (ok && t < p.T) ? (xb + row * p.T + t) * 4 : kOOB);
                araw[g][1] = ld16(xrs, (ok && t + 4 < p.T) ? (xb + row * p.T + t + 4) * 4 : kOOB);
            }
        } else {
#pragma unroll
            for (int i = 0; i < NX; ++i) {
                const int t = ts + xq[i];
                const bool ok = xrow[i] >= 0 && t >= 0 && t < p.T;
                xreg[i] = ld16(xrs, ok ? (xb + xrow[i] + t) * 4 : kOOB);
            }
        }
#pragma unroll
        for (int i = 0; i < ND; ++i) {
            const int t = tc + dq[i];
            const bool ok = drow[i] >= 0 && t < p.T;
            dreg[i] = ld16(drs, ok ? (db + drow[i] + t) * 4 : kOOB);
        }
        if (masked) {
            const int tx = ts + tid * 4, td = tc + (tid - 64) * 4;
            if (mx_thread) mreg = ld16(mxrs, (tx >= 0 && tx < p.T) ? (b * p.T + tx) * 4 : kOOB);
            if (md_thread) mreg = ld16(mdrs, (td < p.T) ? (b * p.T + td) * 4 : kOOB);
        }
    };
    auto store_chunk = [&]() {
        if (PRE) {
#pragma unroll
            for (int i = 0; i < NX; ++i) {
                const int idx = tid + i * 256;
                const int q = idx % (XWL / 4), r = idx / (XWL / 4);
                if (idx < X4)
#pragma unroll
                    for (int pl = 0; pl < NS; ++pl) lds_store8(Xh + pl * XPLANE + r * XP16 + q * 4, xpr[i][pl][0], xpr[i][pl][1]);
            }
#pragma unroll
            for (int i = 0; i < ND; ++i) {
                const int idx = tid + i * 256;
                const int q = idx % (CT / 4), r = idx / (CT / 4);
                if (idx < D4) {
                    if (do_bias) {                           // the planes sum back to the fp32 value exactly
                        float sum = 0.f;
#pragma unroll
                        for (int pl = 0; pl < NS; ++pl) {
                            const unsigned a = (unsigned)dpr[i][pl][0], b2 = (unsigned)dpr[i][pl][1];
                            sum += (__uint_as_float(a << 16) + __uint_as_float(a & 0xffff0000u)) +
                                   (__uint_as_float(b2 << 16) + __uint_as_float(b2 & 0xffff0000u));
                        }
                        bsum[i] += sum;
                    }
#pragma unroll
                    for (int pl = 0; pl < NS; ++pl) lds_store8(Dh + pl * DPLANE + r * DP16 + q * 4, dpr[i][pl][0], dpr[i][pl][1]);
                }
            }
            return;
        }
        if (masked) {
            if (mx_thread) *reinterpret_cast<f32x4 *>(Mx + tid * 4) = mreg;
            if (md_thread) *reinterpret_cast<f32x4 *>(Md + (tid - 64) * 4) = mreg;
            __syncthreads();
        }
        if constexpr (AG) {
#pragma unroll
            for (int g = 0; g < NSTEP; ++g)
#pragma unroll
                for (int h = 0; h < 2; ++h) {
                    f32x4 v = araw[g][h];
                    const int f0 = g * 32 + lk * 8 + h * 4;                    // first of the four frames, inside the chunk's window
                    if (p.mask_x && f0 + 4 <= XWL) v *= *reinterpret_cast<const f32x4 *>(Mx + f0);
                    unsigned o01[NS], o23[NS];
                    split_planes2<NS>(v[0], v[1], o01);
                    split_planes2<NS>(v[2], v[3], o23);
#pragma unroll
                    for (int pl = 0; pl < NS; ++pl) { apl[g][pl][2 * h] = (int)o01[pl]; apl[g][pl][2 * h + 1] = (int)o23[pl]; }
                }
        } else {
#pragma unroll
            for (int i = 0; i < NX; ++i) {
                const int idx = tid + i * 256;
                const int q = idx % (XWL / 4), r = idx / (XWL / 4);
                if (idx < X4) {
                    f32x4 v = xreg[i];
                    if (p.mask_x) v *= *reinterpret_cast<const f32x4 *>(Mx + q * 4);
                    unsigned o01[NS], o23[NS];
                    split_planes2<NS>(v[0], v[1], o01);
                    split_planes2<NS>(v[2], v[3], o23);
#pragma unroll
                    for (int pl = 0; pl < NS; ++pl)
                        lds_store8(Xh + pl * XPLANE + r * XP16 + q * 4, (int)o01[pl], (int)o23[pl]);
                }
            }
        }
#pragma unroll
        for (int i = 0; i < ND; ++i) {
            const int idx = tid + i * 256;
            const int q = idx % (CT / 4), r = idx / (CT / 4);
            if (idx < D4) {
                f32x4 v = dreg[i];
                if (p.mask) v *= *reinterpret_cast<const f32x4 *>(Md + q * 4);
                if (do_bias) bsum[i] += (v[0] + v[1]) + (v[2] + v[3]);
                unsigned o01[NS], o23[NS];
                split_planes2<NS>(v[0], v[1], o01);
                split_planes2<NS>(v[2], v[3], o23);
#pragma unroll
                for (int pl = 0; pl < NS; ++pl)
                    lds_store8(Dh + pl * DPLANE + r * DP16 + q * 4, (int)o01[pl], (int)o23[pl]);
            }
        }
    };
    const unsigned short *xa = Xh + (wave * 16 + lrow) * XP16 + lk * 8;
    const unsigned short *db_ = Dh + lrow * DP16 + lk * 8;
    auto compute = [&]() {
        int aw[AG ? 1 : 2][NS][8];
        i32x4 bv[2][MT][NS];
        auto fetch = [&](int g, int sl) {
            if constexpr (!AG) {
#pragma unroll
                for (int pl = 0; pl < NS; ++pl) {
                    const i32x4 lo = *reinterpret_cast<const i32x4 *>(xa + pl * XPLANE + g * 32);
                    const i32x4 hi = *reinterpret_cast<const i32x4 *>(xa + pl * XPLANE + g * 32 + 8);
#pragma unroll
                    for (int e = 0; e < 4; ++e) { aw[sl][pl][e] = lo[e]; aw[sl][pl][4 + e] = hi[e]; }
                }
            }
#pragma unroll
            for (int i = 0; i < MT; ++i)
#pragma unroll
                for (int pl = 0; pl < NS; ++pl)
                    bv[sl][i][pl] = *reinterpret_cast<const i32x4 *>(db_ + pl * DPLANE + i * 16 * DP16 + g * 32);
        };
        fetch(0, 0);
#pragma unroll
        for (int g = 0; g < NSTEP; ++g) {
            if (g + 1 < NSTEP) fetch(g + 1, (g + 1) & 1);
            __builtin_amdgcn_sched_barrier(0);
            const int sl = g & 1;
#pragma unroll
            for (int tp = 0; tp < TAPS; ++tp) {
                const int sh = tp + OFF;                 // shift in bf16 elements inside the 16-frame window
                i32x4 av[NS];
#pragma unroll
                for (int pl = 0; pl < NS; ++pl)
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        if constexpr (AG)
                            av[pl][e] = apl[g][pl][e];
                        else
                            av[pl][e] = (sh & 1) ? (int)__builtin_amdgcn_alignbit((unsigned)aw[sl][pl][(sh >> 1) + e + 1],
                                                                                 (unsigned)aw[sl][pl][(sh >> 1) + e], 16)
                                                 : aw[sl][pl][(sh >> 1) + e];
                    }
#pragma unroll
                for (int i = 0; i < MT; ++i)
#pragma unroll
                    for (int k = 0; k < n_products(NS); ++k)
                        acc[tp][i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(
                            __builtin_bit_cast(bf16x8, av[product_a(NS, k)]),
                            __builtin_bit_cast(bf16x8, bv[sl][i][product_b(NS, k)]), acc[tp][i], 0, 0, 0);
            }
            __builtin_amdgcn_sched_barrier(0);
        }
    };

    GLOWTTS_TRACE_POINT_Z(0);
    // ---- round 4: the 1x1 weight gradient, software-pipelined -------------------------------------------------------------
    // What bounded this kernel (tools/trace_conv.py wrw1: 26.6 us per launch, MFMA pipe 28 % busy): per 80-frame chunk a
    // workgroup spent 1.9 us in its 72 MFMAs per wave and then 1.2 us — behind a barrier, pipe idle — splitting the NEXT chunk's
    // fp32 values into bf16 planes (~235 vector instructions per wave).  Here the split of chunk c + 1 is cut into 11 units
    // (6 of the x rows a lane owns, 5 of the d rows it stages) that are placed BETWEEN the MFMA groups of chunk c: a group of 6
    // MFMAs occupies the matrix pipe for 96 cycles, one unit issues in ~90.  The planes wait in registers; behind the barrier only
    // the 15 LDS stores per thread remain.  The loads of chunk c + 2 are issued once the units have consumed chunk c + 1's
    // registers.  B operands are fetched per 16-row tile (24 registers instead of 96) to make room.  Same arithmetic, same order.
    if constexpr (AG) {
        if (!masked && p.ds_pitch == 0) {                    // (ds_pitch: unused by this kernel otherwise — 1 = GLOWTTS_WRW1_PIPE=0, A/B switch)
            constexpr int NU = 2 * NSTEP + ND;               // split units per chunk
            static_assert(NU <= NSTEP * MT, "one unit per MFMA group");
            int apl_n[NSTEP][NS][4];
            unsigned dpl[ND][NS][2];
            auto unit = [&](int u) {
                if (u < 2 * NSTEP) {
                    const int g = u >> 1, h = u & 1;
                    const f32x4 v = araw[g][h];
                    unsigned o01[NS], o23[NS];
                    split_planes2<NS>(v[0], v[1], o01);
                    split_planes2<NS>(v[2], v[3], o23);
#pragma unroll
                    for (int pl = 0; pl < NS; ++pl) { apl_n[g][pl][2 * h] = (int)o01[pl]; apl_n[g][pl][2 * h + 1] = (int)o23[pl]; }
                } else if (u - 2 * NSTEP < ND) {
                    const int i = u - 2 * NSTEP;
                    const f32x4 v = dreg[i];
                    if (do_bias && tid + i * 256 < D4) bsum[i] += (v[0] + v[1]) + (v[2] + v[3]);
                    unsigned o01[NS], o23[NS];
                    split_planes2<NS>(v[0], v[1], o01);
                    split_planes2<NS>(v[2], v[3], o23);
#pragma unroll
                    for (int pl = 0; pl < NS; ++pl) { dpl[i][pl][0] = o01[pl]; dpl[i][pl][1] = o23[pl]; }
                }
            };
            auto commit_planes = [&]() {
#pragma unroll
                for (int i = 0; i < ND; ++i) {
                    const int idx = tid + i * 256;
                    const int q = idx % (CT / 4), r = idx / (CT / 4);
                    if (idx < D4)
#pragma unroll
                        for (int pl = 0; pl < NS; ++pl) lds_store8(Dh + pl * DPLANE + r * DP16 + q * 4, (int)dpl[i][pl][0], (int)dpl[i][pl][1]);
                }
#pragma unroll
                for (int g = 0; g < NSTEP; ++g)
#pragma unroll
                    for (int pl = 0; pl < NS; ++pl)
#pragma unroll
                        for (int e = 0; e < 4; ++e) apl[g][pl][e] = apl_n[g][pl][e];
            };
            if (nchunks > 0) {
                load_chunk(0);
#pragma unroll
                for (int u = 0; u < NU; ++u) unit(u);
                commit_planes();
                if (nchunks > 1) load_chunk(1);
            }
            lds_stores_done();
            __syncthreads();
            GLOWTTS_TRACE_POINT_Z(1);
            for (int c = 0; c < nchunks; ++c) {
                const bool more = c + 1 < nchunks;
                i32x4 bt[2][NS];
                auto fetch_t = [&](int q, int sl) {          // q = g * MT + i
                    const int g = q / MT, i = q - g * MT;
#pragma unroll
                    for (int pl = 0; pl < NS; ++pl)
                        bt[sl][pl] = *reinterpret_cast<const i32x4 *>(db_ + pl * DPLANE + i * 16 * DP16 + g * 32);
                };
                fetch_t(0, 0);
#pragma unroll
                for (int q = 0; q < NSTEP * MT; ++q) {
                    const int g = q / MT, i = q - g * MT;
                    if (q + 1 < NSTEP * MT) fetch_t(q + 1, (q + 1) & 1);
                    __builtin_amdgcn_sched_barrier(0);
                    i32x4 av[NS];
#pragma unroll
                    for (int pl = 0; pl < NS; ++pl)
#pragma unroll
                        for (int e = 0; e < 4; ++e) av[pl][e] = apl[g][pl][e];
#pragma unroll
                    for (int k = 0; k < n_products(NS); ++k)
                        acc[0][i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(
                            __builtin_bit_cast(bf16x8, av[product_a(NS, k)]),
                            __builtin_bit_cast(bf16x8, bt[q & 1][product_b(NS, k)]), acc[0][i], 0, 0, 0);
                    __builtin_amdgcn_sched_barrier(0);
                    // one split unit of the NEXT chunk behind each MFMA group but the first (its loads were issued a barrier ago)
                    if (more && q >= 1 && q - 1 < NU) unit(q - 1);
                    __builtin_amdgcn_sched_barrier(0);
                }
                // the groups above gave slots to units 0 .. NSTEP * MT - 2: the 64-frame form <., 1, 4, 4> has NU = NSTEP * MT = 8,
                // so its last unit (d item 3: rows 48..63 of the d tile, bsum[3]) runs here, behind the last group
                if (more) {
#pragma unroll
                    for (int u = NSTEP * MT - 1; u < NU; ++u) unit(u);
                }
                if (c + 2 < nchunks) load_chunk(c + 2);      // the units have consumed chunk c + 1's registers
                if (c == 0) GLOWTTS_TRACE_POINT_Z(2);
                __syncthreads();                             // every wave is through with this chunk's d image
                if (more) {
                    commit_planes();
                    lds_stores_done();
                    __syncthreads();
                }
                if (c == 0) GLOWTTS_TRACE_POINT_Z(3);
            }
            goto wrw_epilogue;
        }
    }
    if (nchunks > 0) {
        load_chunk(0);
        store_chunk();
    }
    lds_stores_done();
    __syncthreads();
    GLOWTTS_TRACE_POINT_Z(1);
    for (int c = 0; c < nchunks; ++c) {
        const bool more = c + 1 < nchunks;
        if (more) load_chunk(c + 1);
        compute();
        if (c == 0) GLOWTTS_TRACE_POINT_Z(2);
        __syncthreads();
        if (c == 0) GLOWTTS_TRACE_POINT_Z(5);
        if (more) {
            store_chunk();
            lds_stores_done();
            if (c == 0) GLOWTTS_TRACE_POINT_Z(6);
            __syncthreads();
        }
        if (c == 0) GLOWTTS_TRACE_POINT_Z(3);
    }
wrw_epilogue:
    GLOWTTS_TRACE_POINT_Z(4);
    if (k0 + 64 <= p.Cin && m0 + MR <= p.M) {
        float *base = p.dwp + (long)(k0 + wave * 16 + lk * 4) * p.M + m0 + lrow;
#pragma unroll
        for (int tp = 0; tp < TAPS; ++tp)
#pragma unroll
            for (int i = 0; i < MT; ++i)
#pragma unroll
                for (int reg = 0; reg < 4; ++reg)
                    atomicAdd(base + ((long)tp * p.Cin + reg) * p.M + i * 16, acc[tp][i][reg]);
    } else {
#pragma unroll
        for (int tp = 0; tp < TAPS; ++tp)
#pragma unroll
            for (int i = 0; i < MT; ++i)
#pragma unroll
                for (int reg = 0; reg < 4; ++reg) {
                    const int k = k0 + wave * 16 + lk * 4 + reg;
                    const int m = m0 + i * 16 + lrow;
                    if (k < p.Cin && m < p.M) atomicAdd(p.dwp + ((long)tp * p.Cin + k) * p.M + m, acc[tp][i][reg]);
                }
    }
    GLOWTTS_TRACE_POINT_Z(10);
    if (do_bias) {
        __syncthreads();
        if (tid < 64) rowacc[tid] = 0.f;
        __syncthreads();
#pragma unroll
        for (int i = 0; i < ND; ++i) {
            const int idx = tid + i * 256;
            if (idx < D4) atomicAdd(rowacc + idx / (CT / 4), bsum[i]);
        }
        __syncthreads();
        if (tid < MR && m0 + tid < p.M) atomicAdd(p.dbias + m0 + tid, rowacc[tid]);
    }
}

// ---------------------------------------------------------------------------------------------------------------
// host side: arithmetic mode, plane registry, launch
// ---------------------------------------------------------------------------------------------------------------
// The planes of a packed-weight buffer belong to the caller (a device buffer of 3 x n bf16 next to its n fp32 values).
// A thread binds ONE such pair for the launches it is about to make (the host side binds a WN stack's buffer around its
// forward / backward and unbinds after): no hidden storage here, nothing that can go stale.
struct PlaneBinding {
    const float *wp = nullptr;
    long n = 0;
    const unsigned short *planes = nullptr;
    int ns = 0;
};
static thread_local PlaneBinding t_bound;
// The library's ONE piece of mutable process-wide state (include/glowtts_hip.h, conventions): read by launches made from any
// thread (autograd's backward thread included) while another may call the setter, hence atomic; a launch takes the value
// in force when it is queued.
static std::atomic<int> g_conv_math{0};       // planes for the forward-type kernels (0 = native fp32)
static std::atomic<int> g_conv_math_wrw{0};   // planes for the weight-gradient kernel

static bool find_planes(const float *wp, int ns, const unsigned short **out, long *stride) {
    const PlaneBinding &b = t_bound;
    if (b.wp == nullptr || b.ns != ns || wp < b.wp || wp >= b.wp + b.n) return false;
    *out = b.planes + (wp - b.wp);
    *stride = b.n;
    return true;
}

// wn_fused.hip: the planes bound to the calling thread for packed weights at `wp`, and the forward-type arithmetic code
bool conv_find_planes(const float *wp, int ns, const unsigned short **out, long *stride) { return find_planes(wp, ns, out, stride); }
int conv_math_forward() { return g_conv_math.load(std::memory_order_relaxed); }
int conv_math_wrw() { return g_conv_math_wrw.load(std::memory_order_relaxed); }       // convwrw1.hip

template <int NS, int RTW, int NCT, int EPI, int TAPS, int IOB = 0, int NSA = NS>
static int launch_split(ConvGemmParams &p, const unsigned short *planes, long stride, hipStream_t s) {
    constexpr int WGR = 64 * RTW, NT = 16 * NCT;
    constexpr size_t lds_pipe = (size_t)NS * 3 * (NT + 16) * 80 + (size_t)(NT + 16) * sizeof(float);
    constexpr size_t lds_epi = ((size_t)WGR * (NT + 4) + (EPI == EPI_GATEBWD ? 2 * WGR : 0)) * sizeof(float);
    constexpr size_t lds = lds_pipe > lds_epi ? lds_pipe : lds_epi;
    static_assert(lds <= 80 * 1024, "two workgroups per CU");
    p.vec_epilogue = aligned16(p.y0) && aligned16(p.y1) && aligned16(p.r0) && aligned16(p.r1) && aligned16(p.mask) &&
                     aligned16(p.drop) && (p.y_bs % 4 == 0) && (p.r_bs % 4 == 0) && (EPI != EPI_GATE || p.H % 4 == 0);
    static LdsLimit attr_max_e;   // per device: raised only when a launch needs more than any earlier one
    if (int rc_ = attr_max_e.ensure(reinterpret_cast<const void *>(&convgemm_split_kernel<NS, RTW, NCT, EPI, TAPS, IOB, NSA>), lds, "glowtts_conv (split)")) return rc_;
    p.exp = GLOWTTS_EXP_BITS(knob(K_CONV_EXP));
    const int ntile_t = (p.T + NT - 1) / NT;
    const int rows = (EPI == EPI_GATE) ? p.H : p.M;
    const int per = (EPI == EPI_GATE) ? 64 : WGR;
    dim3 grid(ntile_t * p.B, (rows + per - 1) / per);
    p.wg_order = (grid.y > 1 && grid.x % 8 == 0 && knob(K_CONV_ROW_ADJ) == 1) ? 1 : 0;
    hipLaunchKernelGGL((convgemm_split_kernel<NS, RTW, NCT, EPI, TAPS, IOB, NSA>), grid, dim3(256), lds, s, p, planes, stride);
    GLOWTTS_LAUNCH_CHECK("glowtts_conv (split)");
}

// bf16 tensors: the instantiations a flow block needs (csrc/wn_stack.hip, io = 1)
template <int IOB>
static int dispatch_bf16_io(ConvGemmParams &p, int epi, bool big, bool n5, const unsigned short *pl, long st, hipStream_t s) {
#define GLOWTTS_BF16_CASE(E, R, TP)                                                    \
    if (epi == E && (R == 2) == big && p.taps == TP)                                   \
        return n5 ? launch_split<1, R, 5, E, TP, IOB, (IOB == 2 ? 3 : 1)>(p, pl, st, s)  \
                  : launch_split<1, R, 4, E, TP, IOB, (IOB == 2 ? 3 : 1)>(p, pl, st, s);
    // the 1x1 PLAIN / ADD convolutions (start, end, start's input gradient) exist with 64-row workgroups only, which serve
    // any M: dispatch_convgemm's 128-row preference (M % 128 == 0 or M > 192: squeezed C = 128 / 256) must not leave them
    // without a kernel
    if (p.taps == 1 && (epi == EPI_PLAIN || epi == EPI_ADD)) big = false;
    GLOWTTS_BF16_CASE(EPI_PLAIN, 1, 1)       // fp32 results (IOB == 2): exact weights, three planes
    GLOWTTS_BF16_CASE(EPI_ADD, 1, 1)         // ... and the start conv's input gradient added into an fp32 flow gradient
    if constexpr (IOB == 1) {
        GLOWTTS_BF16_CASE(EPI_GATE, 2, 5)
        GLOWTTS_BF16_CASE(EPI_RESSKIP, 2, 1)
        GLOWTTS_BF16_CASE(EPI_RESSKIP_LAST, 1, 1)
        GLOWTTS_BF16_CASE(EPI_GATEBWD, 1, 1)
        GLOWTTS_BF16_CASE(EPI_ADD, 1, 5)
        GLOWTTS_BF16_CASE(EPI_PLAIN, 1, 5)
    }
#undef GLOWTTS_BF16_CASE
    return -1;
}

template <int NS>
static int dispatch_split_ns(ConvGemmParams &p, int epi, bool big, int nct, const unsigned short *pl, long st, hipStream_t s) {
    const bool n5 = nct == 5;
#define GLOWTTS_SPLIT_CASE(E, R, TP)                                                   \
    if (epi == E && (R == 2) == big && p.taps == TP && nct != 2)                       \
        return n5 ? launch_split<NS, R, 5, E, TP>(p, pl, st, s) : launch_split<NS, R, 4, E, TP>(p, pl, st, s);
    GLOWTTS_SPLIT_CASE(EPI_GATE, 2, 5)
    GLOWTTS_SPLIT_CASE(EPI_RESSKIP, 2, 1)
    GLOWTTS_SPLIT_CASE(EPI_RESSKIP_LAST, 1, 1)
    GLOWTTS_SPLIT_CASE(EPI_GATEBWD, 1, 1)
    GLOWTTS_SPLIT_CASE(EPI_ADD, 1, 5)
    GLOWTTS_SPLIT_CASE(EPI_PLAIN, 1, 5)
    // round 3: the text encoder's 3-tap FFN convolutions (attentions.py:347-381; forward, and backward-data as a forward-type
    // convolution) when their ConvGroup's planes are bound — fp32-equivalent form only; with 32-frame tiles where the native
    // dispatch would pick them (T_text = 160: 80-frame tiles leave a quarter of the CUs without a workgroup)
    if constexpr (NS == 3) {
        // (round 4: the coupling's 1x1 start / end convolutions and their backward-data forms — 80 / 160 / 192 channels, 48 launches
        // per step on the decoder's chain — were instantiated here as EPI_PLAIN / EPI_ADD with one tap: under rocprofv3 the
        // bf16-plane forms take what the fp32-MFMA ones do (17.2 / 14.5 us against 16.4 / 15.3 in the step, 8.7 against 8.0 us at
        // best) and the step does not move (14.89 against 14.92 ms): these launches are their fixed costs.  Not kept.)
        GLOWTTS_SPLIT_CASE(EPI_PLAIN, 1, 3)
        GLOWTTS_SPLIT_CASE(EPI_PLAIN, 2, 3)
        GLOWTTS_SPLIT_CASE(EPI_ADD, 1, 3)
        GLOWTTS_SPLIT_CASE(EPI_ADD, 2, 3)
        // 32-frame tiles: only the 64-row form pays (rocprofv3, T = 160: 192 <- 768 channels 50.0 -> 45.4 us; the 128-row form
        // of 768 <- 192 channels measured 47.3 us against 45.3 native: it stays on the fp32 MFMA kernel)
        if (epi == EPI_PLAIN && p.taps == 3 && nct == 2 && !big) return launch_split<NS, 1, 2, EPI_PLAIN, 3>(p, pl, st, s);
    }
#undef GLOWTTS_SPLIT_CASE
    return -1;
}

template <int NS, int TAPS, int NGRP, int MT>
static int launch_wrw_split(ConvWrwParams &p, hipStream_t s) {
    constexpr int CT = 16 * NGRP, MR = 16 * MT;
    constexpr size_t lds = (size_t)NS * (64 + MR) * 104 * 2 + (size_t)(CT + 8 + CT + 64) * sizeof(float);
    static_assert(lds <= 80 * 1024, "two workgroups per CU");
    static LdsLimit attr_max_e;   // per device: raised only when a launch needs more than any earlier one
    if (int rc_ = attr_max_e.ensure(reinterpret_cast<const void *>(&convwrw_split_kernel<NS, TAPS, NGRP, MT>), lds, "glowtts_conv_wrw (split)")) return rc_;
    const int tiles = ((p.Cin + 63) / 64) * ((p.M + MR - 1) / MR);
    const int total = p.B * ((p.T + CT - 1) / CT);
    int splits = 512 / tiles;                       // all workgroups resident at once (2 per CU; 256 / 128 slots for the single
                                                    // launches measured 15.08 / 15.17 ms per step against 15.11: no gain)
    if (splits > total) splits = total;
    if (splits < 1) splits = 1;
    p.nb = (total + splits - 1) / splits;
    p.ds_pitch = knob(K_WRW1_PIPE) == 0 ? 1 : 0;                   // 1x1 kernel: 1 = the unpipelined loop (tuning / A-B switch)
    dim3 grid(tiles, 1, ((total + p.nb - 1) / p.nb) * (p.nbatch > 0 ? p.nbatch : 1));
    hipLaunchKernelGGL((convwrw_split_kernel<NS, TAPS, NGRP, MT>), grid, dim3(256), lds, s, p);
    GLOWTTS_LAUNCH_CHECK("glowtts_conv_wrw (split)");
}

template <int NS, int TAPS, int NGRP, int MT>
static int launch_wrw_planes(ConvWrwParams &p, hipStream_t s) {
    constexpr int CT = 16 * NGRP, MR = 16 * MT;
    constexpr size_t lds = (size_t)NS * (64 + MR) * 104 * 2 + (size_t)(CT + 8 + CT + 64) * sizeof(float);
    static LdsLimit attr_max_e;   // per device: raised only when a launch needs more than any earlier one
    if (int rc_ = attr_max_e.ensure(reinterpret_cast<const void *>(&convwrw_split_kernel<NS, TAPS, NGRP, MT, true>), lds, "glowtts_conv_wrw_planes")) return rc_;
    const int tiles = ((p.Cin + 63) / 64) * ((p.M + MR - 1) / MR);
    const int total = p.B * ((p.T + CT - 1) / CT);
    int splits = 512 / tiles;
    if (splits > total) splits = total;
    if (splits < 1) splits = 1;
    p.nb = (total + splits - 1) / splits;
    dim3 grid(tiles, 1, (total + p.nb - 1) / p.nb);
    hipLaunchKernelGGL((convwrw_split_kernel<NS, TAPS, NGRP, MT, true>), grid, dim3(256), lds, s, p);
    GLOWTTS_LAUNCH_CHECK("glowtts_conv_wrw_planes");
}

int conv_wrw_planes_dispatch(ConvWrwParams &p, int ns, hipStream_t s) {
    const bool n5 = (p.T % 80 == 0) || ((p.T + 79) / 80) * 80 <= ((p.T + 63) / 64) * 64;
    if (ns == 3 && p.taps == 5 && p.M % 32 == 0) return n5 ? launch_wrw_planes<3, 5, 5, 2>(p, s) : launch_wrw_planes<3, 5, 4, 2>(p, s);
    if (ns == 3 && p.taps == 1) return n5 ? launch_wrw_planes<3, 1, 5, 4>(p, s) : launch_wrw_planes<3, 1, 4, 4>(p, s);
    // one plane = the tensors ARE bf16 (flow blocks with bf16 activations in HBM)
    if (ns == 1 && p.taps == 5 && p.M % 32 == 0) return n5 ? launch_wrw_planes<1, 5, 5, 2>(p, s) : launch_wrw_planes<1, 5, 4, 2>(p, s);
    if (ns == 1 && p.taps == 1) return n5 ? launch_wrw_planes<1, 1, 5, 4>(p, s) : launch_wrw_planes<1, 1, 4, 4>(p, s);
    return -1;
}

template <int NS>
static int dispatch_wrw_split_ns(ConvWrwParams &p, hipStream_t s) {
    const bool n5 = (p.T % 80 == 0) || ((p.T + 79) / 80) * 80 <= ((p.T + 63) / 64) * 64;
    if (p.taps == 5 && p.M % 32 == 0 && (!p.d2 || p.d_split % 32 == 0))
        return n5 ? launch_wrw_split<NS, 5, 5, 2>(p, s) : launch_wrw_split<NS, 5, 4, 2>(p, s);
    if (p.taps == 1) return n5 ? launch_wrw_split<NS, 1, 5, 4>(p, s) : launch_wrw_split<NS, 1, 4, 4>(p, s);
    return -1;
}

// called by the frame-packed weight-gradient entries (dilation 1, 'same' padding, 16-byte rows already checked)
int conv_wrw_split_dispatch(ConvWrwParams &p, hipStream_t s) {
    const int ns = g_conv_math_wrw.load(std::memory_order_relaxed);
    if (ns != 0)                                   // 5-tap convolutions: the frame-major / transposed-read kernel (convwrw_tr.hip)
        if (int rc = conv_wrw_tr_dispatch(p, ns, s); rc >= 0) return rc;
    if (ns == 3) return dispatch_wrw_split_ns<3>(p, s);
    if (ns == 2) return dispatch_wrw_split_ns<2>(p, s);
    if (ns == 1) return dispatch_wrw_split_ns<1>(p, s);
    return -1;
}

int conv_bf16_dispatch(ConvGemmParams &p, int epi, bool big, bool n5, bool pipe_ok, hipStream_t s) {
    GLOWTTS_CHECK_ARG(pipe_ok && (p.T % 4) == 0, "glowtts_conv (bf16 tensors): needs T %% 4 == 0 and 16-byte aligned rows");
    const unsigned short *pl = nullptr;
    long st = 0;
    GLOWTTS_CHECK_ARG(find_planes(p.wp, 3, &pl, &st), "glowtts_conv (bf16 tensors): the packed weights have no bf16 planes bound "
                      "(glowtts_split_planes(wp, n, planes, 3) + glowtts_conv_bind_planes_ns(wp, n, planes, 3))");
    const int rc = p.yb ? dispatch_bf16_io<1>(p, epi, big, n5, pl, st, s) : dispatch_bf16_io<2>(p, epi, big, n5, pl, st, s);
    GLOWTTS_CHECK_ARG(rc >= 0, "glowtts_conv (bf16 tensors): no kernel for epilogue %d, taps %d, M %d, output %s", epi, p.taps, p.M,
                      p.yb ? "bf16" : "fp32");
    return rc;
}

// called first by dispatch_convgemm: -1 = not handled here (mode off, weights not registered, shape not instantiated)
int conv_split_dispatch(ConvGemmParams &p, int epi, bool big, int nct, bool pipe_ok, hipStream_t s) {
    const int ns = g_conv_math.load(std::memory_order_relaxed);
    if (ns == 0 || !pipe_ok) return -1;
    const unsigned short *pl = nullptr;
    long st = 0;
    if (!find_planes(p.wp, ns, &pl, &st)) return -1;
    if (ns == 3) return dispatch_split_ns<3>(p, epi, big, nct, pl, st, s);
    if (ns == 2) return dispatch_split_ns<2>(p, epi, big, nct, pl, st, s);
    return dispatch_split_ns<1>(p, epi, big, nct, pl, st, s);
}

}  // namespace glowtts

using namespace glowtts;

extern "C" int glowtts_conv_math(int nsplit) {
    if (nsplit < 0) return g_conv_math.load() | (g_conv_math_wrw.load() << 2);
    GLOWTTS_CHECK_ARG(nsplit <= 15, "glowtts_conv_math: mode %d (0 = native fp32, 1 = bf16, 2 = bf16x3, 3 = bf16x6; "
                      "+ 4 x the same code for the weight-gradient kernel)", nsplit);
    g_conv_math.store(nsplit & 3);
    g_conv_math_wrw.store((nsplit >> 2) & 3);
    return 0;
}

extern "C" int glowtts_conv_split_weights(const float *wp, long n, unsigned short *planes, glowtts_stream_t stream) {
    GLOWTTS_CHECK_ARG(wp && planes && n > 0, "glowtts_conv_split_weights: bad arguments");
    const int ns = g_conv_math.load(std::memory_order_relaxed);
    if (ns == 0) return 0;
    long grid = (n + 255) / 256;
    if (grid > 2048) grid = 2048;
    hipLaunchKernelGGL(split_weights_kernel, dim3((unsigned)grid), dim3(256), 0, (hipStream_t)stream, wp, planes, n, n, ns);
    GLOWTTS_LAUNCH_CHECK("glowtts_conv_split_weights");
}

extern "C" int glowtts_conv_bind_planes(const float *wp, long n, const unsigned short *planes) {
    GLOWTTS_CHECK_ARG(wp == nullptr || (planes && n > 0), "glowtts_conv_bind_planes: bad arguments");
    t_bound.wp = wp;
    t_bound.n = wp ? n : 0;
    t_bound.planes = wp ? planes : nullptr;
    t_bound.ns = wp ? g_conv_math.load(std::memory_order_relaxed) : 0;           // the planes were written for the mode in force now
    return 0;
}

extern "C" int glowtts_conv_bind_planes_ns(const float *wp, long n, const unsigned short *planes, int n_planes) {
    GLOWTTS_CHECK_ARG(wp == nullptr || (planes && n > 0 && n_planes >= 1 && n_planes <= 3), "glowtts_conv_bind_planes_ns: bad arguments");
    t_bound.wp = wp;
    t_bound.n = wp ? n : 0;
    t_bound.planes = wp ? planes : nullptr;
    t_bound.ns = wp ? n_planes : 0;
    return 0;
}

extern "C" int glowtts_split_planes(const float *x, long n, unsigned short *planes, int n_planes, glowtts_stream_t stream) {
    GLOWTTS_CHECK_ARG(x && planes && n > 0 && n_planes >= 1 && n_planes <= 3, "glowtts_split_planes: bad arguments");
    long grid = (n + 255) / 256;
    if (grid > 4096) grid = 4096;
    hipLaunchKernelGGL(split_weights_kernel, dim3((unsigned)grid), dim3(256), 0, (hipStream_t)stream, x, planes, n, n, n_planes);
    GLOWTTS_LAUNCH_CHECK("glowtts_split_planes");
}

extern "C" int glowtts_conv_wrw_planes(const unsigned short *x_planes, long x_plane_stride, long x_bs,
                                       const unsigned short *d_planes, long d_plane_stride, long d_bs, float *dwp, float *dbias,
                                       int B, int Cin, int M, int T, int taps, int n_planes, glowtts_stream_t stream) {
    GLOWTTS_CHECK_ARG(x_planes && d_planes && dwp, "glowtts_conv_wrw_planes: null pointer");
    GLOWTTS_CHECK_ARG(B >= 0 && Cin > 0 && M > 0 && T >= 0 && T % 4 == 0 && x_bs % 4 == 0 && d_bs % 4 == 0,
                      "glowtts_conv_wrw_planes: bad shape (T and the batch strides must be multiples of 4)");
    GLOWTTS_CHECK_ARG((reinterpret_cast<uintptr_t>(x_planes) & 7u) == 0 && (reinterpret_cast<uintptr_t>(d_planes) & 7u) == 0 &&
                          x_plane_stride % 4 == 0 && d_plane_stride % 4 == 0,
                      "glowtts_conv_wrw_planes: planes must be 8-byte aligned");
    if ((long)B * T == 0) return 0;
    ConvWrwParams p{};
    p.xpl = x_planes; p.dpl = d_planes; p.xpl_stride = x_plane_stride; p.dpl_stride = d_plane_stride;
    p.dwp = dwp; p.dbias = dbias; p.x_bs = x_bs; p.d_bs = d_bs;
    p.B = B; p.Cin = Cin; p.M = M; p.T = T; p.taps = taps; p.dil = 1; p.pad = (taps - 1) / 2;
    const int rc = conv_wrw_planes_dispatch(p, n_planes, (hipStream_t)stream);
    GLOWTTS_CHECK_ARG(rc >= 0, "glowtts_conv_wrw_planes: no kernel for taps=%d, M=%d, planes=%d (3 planes; 5 taps with M %% 32 == 0, or 1 tap)",
                      taps, M, n_planes);
    return rc;
}

#ifdef GLOWTTS_TRACE
extern "C" int glowtts_debug_trace_read_split(unsigned long long *host, int n_words, int clear) {
    hipError_t e = hipMemcpyFromSymbol(host, HIP_SYMBOL(glowtts::g_trace), (size_t)n_words * 8);
    if (e != hipSuccess) return (int)e;
    if (clear) {
        static unsigned long long zeros[8192 * 16];
        e = hipMemcpyToSymbol(HIP_SYMBOL(glowtts::g_trace), zeros, sizeof(zeros));
    }
    return (int)e;
}
#endif
