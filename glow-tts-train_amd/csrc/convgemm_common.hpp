// convgemm_common.hpp — parameter block, epilogues and small helpers shared by the implicit-GEMM convolution kernels
// (convgemm.hip: native fp32 MFMA; convgemm_split.hip: fp32 operands split into bf16 planes).
#pragma once
#include "common.hpp"

namespace glowtts {

typedef float f32x4 __attribute__((ext_vector_type(4)));

// 16 bytes of zeros in device memory: the zero source of the generic kernel's out-of-range staging loads
__device__ __attribute__((aligned(16))) const float g_zero16[4] = {0.f, 0.f, 0.f, 0.f};

#ifdef GLOWTTS_TRACE   // tuning builds only (tools/trace_conv.py): per-workgroup phase timestamps, 100 MHz wall clock
static __device__ unsigned long long g_trace[8192 * 16];   // one copy per translation unit
#define GLOWTTS_TRACE_POINT(i) do { if (threadIdx.x == 0) { g_trace[((blockIdx.y * gridDim.x + blockIdx.x) & 8191) * 16 + (i)] = wall_clock64(); if ((i) == 3 || (i) == 4) g_trace[((blockIdx.y * gridDim.x + blockIdx.x) & 8191) * 16 + 8 + (i)] = __builtin_readcyclecounter(); if ((i) == 0) { g_trace[((blockIdx.y * gridDim.x + blockIdx.x) & 8191) * 16 + 15] = __builtin_amdgcn_s_getreg((4 << 0) | (0 << 6) | (31 << 11)); g_trace[((blockIdx.y * gridDim.x + blockIdx.x) & 8191) * 16 + 14] = 0x100 | __builtin_amdgcn_s_getreg(20 | (0 << 6) | (3 << 11)); } } } while (0)
#define GLOWTTS_TRACE_POINT_Z(i) do { if (threadIdx.x == 0) g_trace[((blockIdx.z * gridDim.x + blockIdx.x) & 8191) * 16 + (i)] = wall_clock64(); } while (0)
#else
#define GLOWTTS_TRACE_POINT(i) do { } while (0)
#define GLOWTTS_TRACE_POINT_Z(i) do { } while (0)
#endif

enum { EPI_PLAIN = 0, EPI_GATE = 1, EPI_RESSKIP = 2, EPI_RESSKIP_LAST = 3, EPI_ADD = 4, EPI_GATEBWD = 5 };

struct ConvGemmParams {
    const float *x;          // (B, Cin, T) activations, batch stride x_bs elements
    const float *x2;         // optional second source: channels [x_split, Cin) come from x2 (B, Cin - x_split, T), batch
    long x2_bs;              //   stride x2_bs — a "virtual concat" of two tensors (x_split a multiple of 96); else null
    int x_split;
    const float *wp;         // packed weights [taps][G = ceil(Cin/16)][M][16]: 16 consecutive input channels per row
    const float *bias;       // [M] or null
    const float *mask;       // (B, T) or null (applied where the epilogue says so)
    const float *cond;       // EPI_GATE: (B, 2H) conditioning added before the gate, or null
    const float *r0;         // EPI_RESSKIP: x_in (B,H,T) ; EPI_ADD: addend (B,M,T) ; EPI_GATEBWD: stored tanh/sigmoid (B,2H,T)
    const float *r1;         // EPI_RESSKIP / _LAST: skip_in (B,H,T) or null
    const unsigned char *drop;  // EPI_GATE: dropout keep-mask (B,2H,T) bytes or null
    float *y0;               // PLAIN/ADD: y (B,M,T), batch stride y_bs ; GATE: acts (B,H,T) ; RESSKIP: x_out (B,H,T)
    float *y1;               // GATE: ts (B,2H,T) tanh / sigmoid values for the backward, or null ; RESSKIP/_LAST: skip_out
    long x_bs, y_bs, r_bs;   // r_bs: batch stride of the EPI_ADD addend
    int B, Cin, M, T, taps, dil, pad, H;
    int mask_in;             // multiply the staged activations by mask (backward-data of a masked output)
    int xp_pitch;            // LDS pitch of the activation slab (floats)
    int mask_out;            // PLAIN: multiply the result by mask
    int mask_add;            // ADD: multiply the addend by mask
    int vec_epilogue;        // pipelined kernel: 16-byte epilogue through LDS (all epilogue tensors 16-byte aligned)
    float drop_scale;        // 1 / (1 - p)
    float *dcond;            // EPI_GATEBWD: (B, 2H) accumulated row sums over t of the UN-dropped pre-activation gradient — the
                             //   gradient of the speaker conditioning row added before the gate (layers.py:150-153) — or null
    int relu;                // PLAIN / ADD: y = max(y, 0) after bias (and mask_out), before the dropout below
    const float *gate_pos;   // PLAIN / ADD (backward-data of a conv that follows ReLU [+ dropout]): (B, M, T) tensor g; the
    float gate_scale;        //   conv result is multiplied by gate_scale where g > 0 and zeroed elsewhere, before the addend.
                             //   PLAIN / ADD also honour `drop` / `drop_scale` (keep bytes (B, M, T)) as the last step
    int xb;                  // 1: x / x2 are bf16 tensors (same element indexing; convgemm_split.hip NS = 1 only)
    int yb;                  // 1: the epilogue's tensors y0, y1, r0, r1 are bf16 (bias, mask, cond stay fp32)
    int wg_order;            // convgemm_split.hip: 1 = the row tiles of a frame tile take consecutive slots of ONE XCD (set by the launcher)
    int exp;                 // tuning build only (GLOWTTS_CONV_EXP, tools/microbench_conv.py): bits that make the kernel SKIP pieces
};

// ---- element access of the epilogues: fp32 tensors, or bf16 tensors behind the same float* fields (p.yb) ------------
typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));
typedef float f32x2_t __attribute__((ext_vector_type(2)));
__device__ __forceinline__ unsigned pack_bf16x2(float a, float b) {      // v_cvt_pk_bf16_f32, round to nearest even
    const f32x2_t v = {a, b};
    return __builtin_bit_cast(unsigned, __builtin_convertvector(v, bf16x2_t));
}
template <bool B16> struct EpiIO;
template <> struct EpiIO<false> {
    static __device__ __forceinline__ float4 ld4(const float *base, long off) { return *reinterpret_cast<const float4 *>(base + off); }
    static __device__ __forceinline__ void st4(float *base, long off, float4 v) { *reinterpret_cast<float4 *>(base + off) = v; }
    static __device__ __forceinline__ float ld1(const float *base, long off) { return base[off]; }
    static __device__ __forceinline__ void st1(float *base, long off, float v) { base[off] = v; }
};
template <> struct EpiIO<true> {
    static __device__ __forceinline__ float4 ld4(const float *base, long off) {
        const uint2 w = *reinterpret_cast<const uint2 *>(reinterpret_cast<const unsigned short *>(base) + off);
        return make_float4(__uint_as_float(w.x << 16), __uint_as_float(w.x & 0xffff0000u), __uint_as_float(w.y << 16),
                           __uint_as_float(w.y & 0xffff0000u));
    }
    static __device__ __forceinline__ void st4(float *base, long off, float4 v) {
        *reinterpret_cast<uint2 *>(reinterpret_cast<unsigned short *>(base) + off) = make_uint2(pack_bf16x2(v.x, v.y), pack_bf16x2(v.z, v.w));
    }
    static __device__ __forceinline__ float ld1(const float *base, long off) {
        return __uint_as_float((unsigned)reinterpret_cast<const unsigned short *>(base)[off] << 16);
    }
    static __device__ __forceinline__ void st1(float *base, long off, float v) {
        reinterpret_cast<unsigned short *>(base)[off] = (unsigned short)(pack_bf16x2(v, 0.f) & 0xffffu);
    }
};

// ---- shared epilogue: lane holds rows lk*4 + reg, column lrow of every 16x16 accumulator tile -------------------
template <int RTW, int NCT, int EPI, bool YB = false>
__device__ __forceinline__ void conv_epilogue(const ConvGemmParams &p, f32x4 (&acc)[RTW][NCT], int b, int t0, int tile_m,
                                              int wave, int lane) {
    using E = EpiIO<YB>;
    constexpr int WGR = 64 * RTW;
    const int lrow = lane & 15, lk = lane >> 4;
    auto ltile = [&](int r) -> int { return (EPI == EPI_GATE) ? (r == 0 ? wave : 4 + wave) : wave * RTW + r; };
    const float *mk = p.mask ? p.mask + (long)b * p.T : nullptr;
    if (EPI == EPI_GATE) {
#pragma unroll
        for (int c = 0; c < NCT; ++c) {
            const int t = t0 + c * 16 + lrow;
#pragma unroll
            for (int reg = 0; reg < 4; ++reg) {
                const int ch = tile_m * 64 + wave * 16 + lk * 4 + reg;       // channel in [0, H)
                if (ch < p.H && t < p.T) {
                    float vt = acc[0][c][reg], vs = acc[RTW - 1][c][reg];
                    if (p.bias) { vt += p.bias[ch]; vs += p.bias[p.H + ch]; }
                    const long ot = ((long)b * 2 * p.H + ch) * p.T + t;
                    const long os = ot + (long)p.H * p.T;
                    if (p.drop) {   // dropout on the pre-activation (layers.py:147), keep-mask generated by the host RNG
                        vt = p.drop[ot] ? vt * p.drop_scale : 0.f;
                        vs = p.drop[os] ? vs * p.drop_scale : 0.f;
                    }
                    if (p.cond) { vt += p.cond[(long)b * 2 * p.H + ch]; vs += p.cond[(long)b * 2 * p.H + p.H + ch]; }
                    const float th = fast_tanh(vt), sg = fast_sigmoid(vs);
                    E::st1(p.y0, ((long)b * p.H + ch) * p.T + t, th * sg);
                    if (p.y1) { E::st1(p.y1, ot, th); E::st1(p.y1, os, sg); }
                }
            }
        }
        return;
    }
#pragma unroll
    for (int r = 0; r < RTW; ++r) {
#pragma unroll
        for (int c = 0; c < NCT; ++c) {
            const int t = t0 + c * 16 + lrow;
#pragma unroll
            for (int reg = 0; reg < 4; ++reg) {
                const int row = tile_m * WGR + ltile(r) * 16 + lk * 4 + reg;
                if (row >= p.M || t >= p.T) continue;
                float v = acc[r][c][reg];
                if (p.bias) v += p.bias[row];
                const float m = mk ? mk[t] : 1.f;
                if (EPI == EPI_PLAIN || EPI == EPI_ADD) {
                    const long od = ((long)b * p.M + row) * p.T + t;         // dense (B, M, T) side tensors
                    if (p.gate_pos) v = p.gate_pos[od] > 0.f ? v * p.gate_scale : 0.f;
                    if (EPI == EPI_ADD) {
                        const float add = E::ld1(p.r0, (long)b * p.r_bs + (long)row * p.T + t);
                        v += p.mask_add ? add * m : add;
                    }
                    if (p.mask_out) v *= m;
                    if (p.relu) v = fmaxf(v, 0.f);
                    if (p.drop) v = p.drop[od] ? v * p.drop_scale : 0.f;
                    E::st1(p.y0, (long)b * p.y_bs + (long)row * p.T + t, v);
                } else if (EPI == EPI_RESSKIP) {
                    // rows [0,H): residual -> next layer input ; rows [H,2H): skip accumulation  (layers.py:157-159)
                    if (row < p.H) {
                        const long o = ((long)b * p.H + row) * p.T + t;
                        E::st1(p.y0, o, (E::ld1(p.r0, o) + v) * m);
                    } else {
                        const long o = ((long)b * p.H + (row - p.H)) * p.T + t;
                        E::st1(p.y1, o, (p.r1 ? E::ld1(p.r1, o) : 0.f) + v);
                    }
                } else if (EPI == EPI_RESSKIP_LAST) {
                    // last layer: all H rows go to the skip sum, and WN's final `output * x_mask` is folded in (:161-162)
                    const long o = ((long)b * p.H + row) * p.T + t;
                    E::st1(p.y1, o, ((p.r1 ? E::ld1(p.r1, o) : 0.f) + v) * m);
                } else if (EPI == EPI_GATEBWD) {
                    // v = d(acts): chain through acts = tanh * sigmoid with the STORED values, then through the
                    // forward's dropout (utils.py:31-38, layers.py:147) -> d(pre-activation) rows ch and H + ch
                    const long ot = ((long)b * 2 * p.H + row) * p.T + t, os = ot + (long)p.H * p.T;
                    const float th = E::ld1(p.r0, ot), sg = E::ld1(p.r0, os);
                    float dt = v * sg * (1.0f - th * th), ds = v * th * sg * (1.0f - sg);
                    if (p.dcond) {           // (rare path: rows not 16-byte aligned) one atomic per element
                        atomicAdd(p.dcond + (long)b * 2 * p.H + row, dt);
                        atomicAdd(p.dcond + (long)b * 2 * p.H + p.H + row, ds);
                    }
                    if (p.drop) {
                        dt = p.drop[ot] ? dt * p.drop_scale : 0.f;
                        ds = p.drop[os] ? ds * p.drop_scale : 0.f;
                    }
                    E::st1(p.y0, ot, dt);
                    E::st1(p.y0, os, ds);
                }
            }
        }
    }
}

template <int RTW, int NCT, int EPI>
__global__ __launch_bounds__(256) void convgemm_kernel(ConvGemmParams p) {
    constexpr int WGR = 64 * RTW;      // rows (output channels) per workgroup
    constexpr int WP = WGR + 16;       // LDS pitch of a packed-weight k-row: == 16 (mod 32)
    constexpr int NT = 16 * NCT;       // columns (frames) per workgroup
    constexpr int KT = 16;             // input channels per K chunk
    extern __shared__ __align__(16) float smem[];
    float *Ws = smem;                              // [taps][KT][WP]
    float *Xs = smem + p.taps * KT * WP;           // [KT][xp_pitch]
    const int XP = p.xp_pitch;
    const int ncols = NT + (p.taps - 1) * p.dil;   // staged columns incl. halo

    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int lrow = lane & 15, lk = lane >> 4;
    const int ntile_t = (p.T + NT - 1) / NT;
    const int b = blockIdx.x / ntile_t;
    const int t0 = (blockIdx.x - b * ntile_t) * NT;
    const int tile_m = blockIdx.y;

    // local row (0..WGR) -> global output row.  The gate pairs channel c (tanh) with H + c (sigmoid): a workgroup
    // takes 64 channels from each half so both land in the same lane/register of two accumulator tiles.
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

    const float *xb = p.x + (long)b * p.x_bs;
    for (int kc0 = 0; kc0 < p.Cin; kc0 += KT) {
        // ---- stage packed weights: [tap][k][rows] as 16-byte pieces (rows contiguous in global and LDS) ----------
        {
            const int G = (p.Cin + 15) / 16;
            const int nw = p.taps * KT * WGR;
            for (int idx = tid; idx < nw; idx += 256) {
                const int lr = idx % WGR;
                const int rest = idx / WGR;
                const int k = rest % KT, tap = rest / KT;
                const int kk = kc0 + k;
                float v = 0.f;
                if (kk < p.Cin && row_ok(lr)) v = p.wp[(((long)tap * G + (kk >> 4)) * p.M + grow(lr)) * 16 + (kk & 15)];
                Ws[(tap * KT + k) * WP + lr] = v;
            }
        }
        // ---- stage the activation slab with halo: [k][t0 - pad .. t0 + NT + halo) --------------------------------------
        for (int idx = tid; idx < KT * ncols; idx += 256) {
            const int k = idx / ncols;
            const int j = idx - k * ncols;
            const int t = t0 - p.pad + j;
            float v = 0.f;
            if (kc0 + k < p.Cin && t >= 0 && t < p.T) {
                v = xb[(long)(kc0 + k) * p.T + t];
                if (p.mask_in) v *= p.mask[(long)b * p.T + t];
            }
            Xs[k * XP + j] = v;
        }
        __syncthreads();
        for (int tap = 0; tap < p.taps; ++tap) {
            const float *wt = Ws + tap * KT * WP;
            const int shift = tap * p.dil;
#pragma unroll
            for (int k4 = 0; k4 < KT / 4; ++k4) {
                float a[RTW], bv[NCT];
#pragma unroll
                for (int r = 0; r < RTW; ++r) a[r] = wt[(k4 * 4 + lk) * WP + ltile(r) * 16 + lrow];
#pragma unroll
                for (int c = 0; c < NCT; ++c) bv[c] = Xs[(k4 * 4 + lk) * XP + c * 16 + lrow + shift];
#pragma unroll
                for (int r = 0; r < RTW; ++r)
#pragma unroll
                    for (int c = 0; c < NCT; ++c)
                        acc[r][c] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[r], bv[c], acc[r][c], 0, 0, 0);
            }
        }
        __syncthreads();
    }

    conv_epilogue<RTW, NCT, EPI>(p, acc, b, t0, tile_m, wave, lane);
}

// ---- vectorised epilogue for the pipelined kernel: accumulator tiles -> LDS [row][frame] -> every lane handles 4
// consecutive frames of one row with 16-byte global loads / stores (the direct epilogue writes 64-byte row segments).
// Needs T % 4 == 0 (a float4 is then entirely inside or outside the utterance).
template <int RTW, int NCT, int EPI, bool YB = false>
__device__ __forceinline__ void conv_epilogue_lds(const ConvGemmParams &p, f32x4 (&acc)[RTW][NCT], float *Ls, int b, int t0,
                                                  int tile_m, int wave, int lane) {
    using E = EpiIO<YB>;
    constexpr int WGR = 64 * RTW, NT = 16 * NCT, LP = NT + 4, Q = NT / 4;
    const int lrow = lane & 15, lk = lane >> 4;
    auto ltile = [&](int r) -> int { return (EPI == EPI_GATE) ? (r == 0 ? wave : 4 + wave) : wave * RTW + r; };
#pragma unroll
    for (int r = 0; r < RTW; ++r)
#pragma unroll
        for (int c = 0; c < NCT; ++c)
#pragma unroll
            for (int reg = 0; reg < 4; ++reg) Ls[(ltile(r) * 16 + lk * 4 + reg) * LP + c * 16 + lrow] = acc[r][c][reg];
    float *rsum = Ls + WGR * LP;                         // EPI_GATEBWD with dcond: [2][WGR] row sums (the launch reserves them)
    if (EPI == EPI_GATEBWD && p.dcond != nullptr)
        for (int i = threadIdx.x; i < 2 * WGR; i += 256) rsum[i] = 0.f;
    __syncthreads();
    const int tid = threadIdx.x;
    const float *mk = p.mask ? p.mask + (long)b * p.T : nullptr;
    auto ld4 = [](const float *q) { return *reinterpret_cast<const float4 *>(q); };      // fp32 only: LDS tile, mask
    if (EPI == EPI_GATE) {
        constexpr int NG = (64 * Q) / 256;               // = NCT items per thread; keep-mask bytes are fetched up front
        unsigned int kt[NG], ks[NG];
#pragma unroll
        for (int i = 0; i < NG; ++i) {
            const int idx = tid + i * 256;
            const int lr = idx / Q, q = idx - lr * Q;
            const int ch = tile_m * 64 + lr, t = t0 + q * 4;
            const bool ok = ch < p.H && t < p.T;
            const long ot = ((long)b * 2 * p.H + (ok ? ch : 0)) * p.T + (ok ? t : 0);
            kt[i] = ks[i] = 0x01010101u;
            if (p.drop) {
                kt[i] = *reinterpret_cast<const unsigned int *>(p.drop + ot);
                ks[i] = *reinterpret_cast<const unsigned int *>(p.drop + ot + (long)p.H * p.T);
            }
        }
#pragma unroll
        for (int i = 0; i < NG; ++i) {
            const int idx = tid + i * 256;
            const int lr = idx / Q, q = idx - lr * Q;
            const int ch = tile_m * 64 + lr;
            const int t = t0 + q * 4;
            if (ch >= p.H || t >= p.T) continue;
            float4 vt = ld4(Ls + lr * LP + q * 4), vs = ld4(Ls + (64 + lr) * LP + q * 4);
            const float bt = p.bias ? p.bias[ch] : 0.f, bs = p.bias ? p.bias[p.H + ch] : 0.f;
            const long ot = ((long)b * 2 * p.H + ch) * p.T + t;
            const long os = ot + (long)p.H * p.T;
            float pt[4] = {vt.x + bt, vt.y + bt, vt.z + bt, vt.w + bt};
            float ps[4] = {vs.x + bs, vs.y + bs, vs.z + bs, vs.w + bs};
            if (p.drop) {   // dropout on the pre-activation (layers.py:147), keep-mask generated by the host RNG
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    pt[j] = ((kt[i] >> (8 * j)) & 0xffu) ? pt[j] * p.drop_scale : 0.f;
                    ps[j] = ((ks[i] >> (8 * j)) & 0xffu) ? ps[j] * p.drop_scale : 0.f;
                }
            }
            if (p.cond) {
                const float ct = p.cond[(long)b * 2 * p.H + ch], cs = p.cond[(long)b * 2 * p.H + p.H + ch];
#pragma unroll
                for (int j = 0; j < 4; ++j) { pt[j] += ct; ps[j] += cs; }
            }
            float th[4], sg[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) { th[j] = fast_tanh(pt[j]); sg[j] = fast_sigmoid(ps[j]); }
            E::st4(p.y0, ((long)b * p.H + ch) * p.T + t, make_float4(th[0] * sg[0], th[1] * sg[1], th[2] * sg[2], th[3] * sg[3]));
            if (p.y1) {
                E::st4(p.y1, ot, make_float4(th[0], th[1], th[2], th[3]));
                E::st4(p.y1, os, make_float4(sg[0], sg[1], sg[2], sg[3]));
            }
        }
        return;
    }
    // WGR * Q = 256 * RTW * NCT float4 items, NI per thread.  Phase 1 issues EVERY global read of the epilogue (residual
    // inputs, stored tanh / sigmoid, masks) back to back; phase 2 combines them with the LDS tile and stores.  (As one
    // run-time loop the compiler kept a single load in flight per iteration: 10 serial HBM round trips = 10 us on the
    // res/skip conv, more than its MFMA time.)
    constexpr int NI = RTW * NCT;
    float4 ra[NI], rb[NI], rm[NI];
    unsigned int ka[NI], kb[NI];
    const float4 zero4 = make_float4(0.f, 0.f, 0.f, 0.f), one4 = make_float4(1.f, 1.f, 1.f, 1.f);
#pragma unroll
    for (int i = 0; i < NI; ++i) {
        const int idx = tid + i * 256;
        const int lr = idx / Q, q = idx - lr * Q;
        const int row = tile_m * WGR + lr, t = t0 + q * 4;
        const bool ok = row < p.M && t < p.T;
        const int rc = ok ? row : 0, tc = ok ? t : 0;               // clamped: always a valid address
        ra[i] = zero4; rb[i] = zero4; ka[i] = 0x01010101u; kb[i] = 0x01010101u;
        rm[i] = mk ? ld4(mk + tc) : one4;
        if (EPI == EPI_PLAIN || EPI == EPI_ADD) {
            const long od = ((long)b * p.M + rc) * p.T + tc;
            if (EPI == EPI_ADD) ra[i] = E::ld4(p.r0, (long)b * p.r_bs + (long)rc * p.T + tc);
            if (p.gate_pos) rb[i] = ld4(p.gate_pos + od);
            if (p.drop) ka[i] = *reinterpret_cast<const unsigned int *>(p.drop + od);
        } else if (EPI == EPI_RESSKIP) {
            const bool res = rc < p.H;
            const long o = ((long)b * p.H + (res ? rc : rc - p.H)) * p.T + tc;
            const float *src = res ? p.r0 : (p.r1 ? p.r1 : p.r0);
            ra[i] = E::ld4(src, o);
            if (!res && !p.r1) ra[i] = zero4;
        } else if (EPI == EPI_RESSKIP_LAST) {
            if (p.r1) ra[i] = E::ld4(p.r1, ((long)b * p.H + rc) * p.T + tc);
        } else if (EPI == EPI_GATEBWD) {
            const long ot = ((long)b * 2 * p.H + rc) * p.T + tc, os = ot + (long)p.H * p.T;
            ra[i] = E::ld4(p.r0, ot);
            rb[i] = E::ld4(p.r0, os);
            if (p.drop) {
                ka[i] = *reinterpret_cast<const unsigned int *>(p.drop + ot);
                kb[i] = *reinterpret_cast<const unsigned int *>(p.drop + os);
            }
        }
    }
#pragma unroll
    for (int i = 0; i < NI; ++i) {
        const int idx = tid + i * 256;
        const int lr = idx / Q, q = idx - lr * Q;
        const int row = tile_m * WGR + lr, t = t0 + q * 4;
        if (row >= p.M || t >= p.T) continue;
        float4 v = ld4(Ls + lr * LP + q * 4);
        if (p.bias) { const float bb = p.bias[row]; v.x += bb; v.y += bb; v.z += bb; v.w += bb; }
        const float4 m = rm[i], a = ra[i];
        if (EPI == EPI_PLAIN || EPI == EPI_ADD) {
            float o[4] = {v.x, v.y, v.z, v.w};
            const float mm[4] = {m.x, m.y, m.z, m.w};
            if (p.gate_pos) {
                const float g[4] = {rb[i].x, rb[i].y, rb[i].z, rb[i].w};
#pragma unroll
                for (int jj = 0; jj < 4; ++jj) o[jj] = g[jj] > 0.f ? o[jj] * p.gate_scale : 0.f;
            }
            if (EPI == EPI_ADD) {
                const float ad[4] = {a.x, a.y, a.z, a.w};
#pragma unroll
                for (int jj = 0; jj < 4; ++jj) o[jj] += p.mask_add ? ad[jj] * mm[jj] : ad[jj];
            }
#pragma unroll
            for (int jj = 0; jj < 4; ++jj) {
                if (p.mask_out) o[jj] *= mm[jj];
                if (p.relu) o[jj] = fmaxf(o[jj], 0.f);
                if (p.drop) o[jj] = ((ka[i] >> (8 * jj)) & 0xffu) ? o[jj] * p.drop_scale : 0.f;
            }
            E::st4(p.y0, (long)b * p.y_bs + (long)row * p.T + t, make_float4(o[0], o[1], o[2], o[3]));
        } else if (EPI == EPI_RESSKIP) {
            if (row < p.H) {
                const long o = ((long)b * p.H + row) * p.T + t;
                E::st4(p.y0, o, make_float4((a.x + v.x) * m.x, (a.y + v.y) * m.y, (a.z + v.z) * m.z, (a.w + v.w) * m.w));
            } else {
                const long o = ((long)b * p.H + (row - p.H)) * p.T + t;
                E::st4(p.y1, o, make_float4(a.x + v.x, a.y + v.y, a.z + v.z, a.w + v.w));
            }
        } else if (EPI == EPI_RESSKIP_LAST) {
            const long o = ((long)b * p.H + row) * p.T + t;
            E::st4(p.y1, o, make_float4((a.x + v.x) * m.x, (a.y + v.y) * m.y, (a.z + v.z) * m.z, (a.w + v.w) * m.w));
        } else if (EPI == EPI_GATEBWD) {
            // v = d(acts): chain through acts = tanh * sigmoid with the STORED values, then through the forward's dropout
            const long ot = ((long)b * 2 * p.H + row) * p.T + t, os = ot + (long)p.H * p.T;
            const float go[4] = {v.x, v.y, v.z, v.w}, th[4] = {a.x, a.y, a.z, a.w}, sg[4] = {rb[i].x, rb[i].y, rb[i].z, rb[i].w};
            float dt[4], ds[4];
#pragma unroll
            for (int jj = 0; jj < 4; ++jj) {
                dt[jj] = go[jj] * sg[jj] * (1.0f - th[jj] * th[jj]);
                ds[jj] = go[jj] * th[jj] * sg[jj] * (1.0f - sg[jj]);
            }
            if (p.dcond != nullptr) {                     // conditioning is added AFTER the dropout: its gradient is un-dropped
                atomicAdd(rsum + lr, (dt[0] + dt[1]) + (dt[2] + dt[3]));
                atomicAdd(rsum + WGR + lr, (ds[0] + ds[1]) + (ds[2] + ds[3]));
            }
#pragma unroll
            for (int jj = 0; jj < 4; ++jj) {
                if (p.drop) {
                    dt[jj] = ((ka[i] >> (8 * jj)) & 0xffu) ? dt[jj] * p.drop_scale : 0.f;
                    ds[jj] = ((kb[i] >> (8 * jj)) & 0xffu) ? ds[jj] * p.drop_scale : 0.f;
                }
            }
            E::st4(p.y0, ot, make_float4(dt[0], dt[1], dt[2], dt[3]));
            E::st4(p.y0, os, make_float4(ds[0], ds[1], ds[2], ds[3]));
        }
    }
    if (EPI == EPI_GATEBWD && p.dcond != nullptr) {       // one global atomic per (row, half) and workgroup
        __syncthreads();
        for (int i = tid; i < 2 * WGR; i += 256) {
            const int half = i / WGR, lr = i - half * WGR, row = tile_m * WGR + lr;
            if (row < p.M) atomicAdd(p.dcond + (long)b * 2 * p.H + half * p.H + row, rsum[i]);
        }
    }
}

struct ConvWrwParams {
    const float *x;      // (B, Cin, T) forward input, batch stride x_bs
    const float *d;      // (B, M, T) output gradient, batch stride d_bs
    const float *d2;     // optional: rows [d_split, M) come from d2 (B, M - d_split, T), batch stride d2_bs (d_split % 64 == 0)
    long d2_bs;
    int d_split;
    float *dwp;          // [taps][Cin][M] accumulated (atomics)
    float *dbias;        // [M] accumulated row sums of (masked) d, or null
    const float *mask;   // (B, T): multiply d by it while staging, or null
    const float *mask_x; // (B, T): multiply x by it while staging (the forward conv consumed x * mask), or null
    long x_bs, d_bs;
    int B, Cin, M, T, taps, dil, pad;
    const unsigned short *xpl, *dpl;   // convgemm_split.hip, pre-split form: bf16 planes of x / d (same element order as
    long xpl_stride, dpl_stride;       //   the fp32 tensors, plane pl at + pl * stride elements); x / d are then unused
    int nb;              // utterances per workgroup (split of the contraction)
    int xs_pitch, ds_pitch;
    // Several problems of ONE shape in one launch (the weight gradients of a WN stack's layers: their operands all exist when the
    // stack's dx chain has run): nbatch > 0 -> problem q takes bx[q], bd[q], bd2[q] (or null), bdwp[q], bdbias[q] (or null); the
    // grid's z extent is nbatch x splits.  A launch per problem ends with its split-K atomics draining before the next one may
    // start; in one launch the next problem's workgroups take the compute units as they free up.
    static constexpr int kMaxBatch = 4;
    const float *bx[kMaxBatch], *bd[kMaxBatch], *bd2[kMaxBatch];
    float *bdwp[kMaxBatch], *bdbias[kMaxBatch];
    int nbatch;
};

// convgemm_split.hip: the frame-packed weight-gradient kernel on bf16 planes; -1 = not handled
int conv_wrw_split_dispatch(ConvWrwParams &p, hipStream_t s);
int conv_wrw_planes_dispatch(ConvWrwParams &p, int ns, hipStream_t s);
// convwrw_tr.hip: the 5-tap weight gradient from frame-major LDS images (transposing LDS reads, two half-period groups per
// workgroup); ns = planes per fp32 operand; -1 = not handled (other taps, M % 32, GLOWTTS_WRW_TR=0)
int conv_wrw_tr_dispatch(ConvWrwParams &p, int ns, hipStream_t s);

// convgemm_split.hip: bf16 TENSORS (p.xb): one bf16 plane of weights (bound with ns = 1), x loaded as bf16, the epilogue's
// tensors bf16 or fp32 by p.yb; an error when the shape has no instantiation or no planes are bound (there is no
// fp32 kernel that could read these tensors)
int conv_bf16_dispatch(ConvGemmParams &p, int epi, bool big, bool n5, bool pipe_ok, hipStream_t s);

// convwino.hip: the gated 5-tap in-conv in Winograd F(4, 5) form from U planes bound to the calling thread; -1 = not handled
int conv_wino_gate_dispatch(ConvGemmParams &p, hipStream_t s);
int conv_math_forward();

// convgemm_split.hip: bf16-plane arithmetic for the forward-type kernels; -1 = not handled (mode off / weights not
// registered / shape not instantiated), otherwise the launch's return code
int conv_split_dispatch(ConvGemmParams &p, int epi, bool big, int nct, bool pipe_ok, hipStream_t s);   // nct: 16-frame column tiles (5 / 4 / 2)

}  // namespace glowtts
