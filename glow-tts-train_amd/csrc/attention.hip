// attention.hip — windowed relative-position multi-head self-attention of the text encoder on fp32 MFMA
// (reference attentions.py:214-333: QK^T + relative-key logits -> mask -> softmax -> dropout -> PV + relative values).
//
// The reference materialises (B,h,T,2T-1) relative-logit tensors and moves them between "relative" and "absolute"
// indexing with pad / reshape tricks (attentions.py:284-333), ~15 launches per layer forward.  Here one workgroup owns a
// 64-query block of one (utterance, head):
//   phase 1   S = A^T B1 on v_mfma_f32_16x16x4_f32 (A = Q block, B1 = K streamed in 64-key tiles through LDS); each
//             wave keeps its 16 x T score strip in accumulator registers (T <= 256);
//   phase 1b  R = A^T E1^T (16 x (2w+1)) once per wave; the band |j-i| <= w of S takes R[i][j-i+w]  (relative keys);
//   phase 2   scale, mask (-1e4 fill, optional block band), row softmax by 16-lane shuffles; P is written once to HBM
//             for the backward, and (after dropout) to LDS as the A operand of
//   phase 4   O = P B2^T (B2 = V streamed in 64-key tiles) + PW E2 with PW[i][r] = P[i][i+r-w]  (relative values);
//   phase 5   O transposed through LDS and stored in the reference layout (B, C, T).
// The SAME kernel with MODE = 1 is the first half of the backward: A = dO, B1 = V, E1 = E_v give dP (incl. the
// relative-value term); phase 2 becomes the softmax backward dS = P (dP - sum_j P dP) * keep/scale; B2 = K, E2 = E_k
// give dQ; dS is written to HBM for the second half (attn_dkv_kernel: dV = dO Pd, dK = Q dS, contraction over
// queries) and for the two small embedding-gradient reductions (attn_relgrad_kernel).
// Limits: d_k % 16 == 0, d_k <= 128, window <= 7.  T <= 256: strip in LDS; 256 < T <= 512: the LONG form below (strip in registers);
// T > 512: attention_long.hip (plain tiled kernels through the (B, h, T, T) matrices; the backward's dK / dV kernel here has no limit).
#include "common.hpp"

namespace glowtts {

typedef float f32x4 __attribute__((ext_vector_type(4)));

struct AttnParams {
    const float *a;        // MODE 0: q        MODE 1: dO          (B, C, T)
    const float *b1;       // MODE 0: k        MODE 1: v
    const float *b2;       // MODE 0: v        MODE 1: k
    const float *e1;       // MODE 0: emb_rel_k MODE 1: emb_rel_v   (n_rel, 2w+1, dk) or null
    const float *e2;       // MODE 0: emb_rel_v MODE 1: emb_rel_k
    const float *mask;     // (B, T)
    const unsigned char *drop;   // (B, h, T, T) keep bytes or null
    float *p;              // (B, h, T, T): MODE 0 writes softmax(P) ; MODE 1 reads it
    float *ds;             // MODE 1: (B, h, T, T) scaled score gradient out
    float *out;            // MODE 0: O (B, C, T) ; MODE 1: dQ (B, C, T)
    int B, H, T, dk, w, block_len, e_hs;   // e_hs: head stride of the embeddings (0 when shared across heads)
    float scale, drop_scale;
    int TP;                // LDS pitch of the probability strip
};

typedef short bf16x4_s __attribute__((ext_vector_type(4)));

// BF = true: the contractions (Q K^T, Q E_k^T, P V, P_w E_v and their backward counterparts) run on the bf16 matrix pipe —
// v_mfma_f32_16x16x16_bf16, fp32 accumulate; operands are rounded to bf16 (nearest even) in registers on their way from
// the SAME fp32 LDS images the fp32 path uses (lane slot lk takes k = 4 lk .. 4 lk + 3 of a 16-deep step), scores, softmax
// and every tensor in HBM stay fp32.  BASELINE configs[2]: "encoder MultiHeadAttention on MFMA" in the bf16 configuration.
__device__ __forceinline__ bf16x4_s bf4(float a, float b, float c, float d) {
    const unsigned lo = io_pack_bf16x2(a, b), hi = io_pack_bf16x2(c, d);
    return bf16x4_s{(short)(lo & 0xffffu), (short)(lo >> 16), (short)(hi & 0xffffu), (short)(hi >> 16)};
}
__device__ __forceinline__ bf16x4_s bf4_strided(const float *p, int stride) {
    return bf4(p[0], p[stride], p[2 * stride], p[3 * stride]);
}
__device__ __forceinline__ bf16x4_s bf4_row(const float *p) { return bf4(p[0], p[1], p[2], p[3]); }
__device__ __forceinline__ f32x4 mma_bf16(bf16x4_s a, bf16x4_s b, f32x4 c) {
    return __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(a, b, c, 0, 0, 0);
}

constexpr int kAP = 80;    // LDS pitch of the [d][64 queries] A block          (== 16 mod 32)
constexpr int kBP = 68;    // LDS pitch of the [d][64 keys] B tile              (==  4 mod 32)

__device__ __forceinline__ float group16_max(float v) {
#pragma unroll
    for (int o = 8; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
    return v;
}
__device__ __forceinline__ float group16_sum(float v) {
#pragma unroll
    for (int o = 8; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

// Stage a [rows][64] tile (row r = src + r*T, columns col0 .. col0+63, zero beyond T) into LDS with pitch `pitch`.
// The loads of one batch are independent (8 in flight per thread) — a plain `for idx` loop serialises load->store and made
// the seven staging phases of the q-block kernel cost more than all its MFMAs.
__device__ __forceinline__ void stage_rows64(float *dst, int pitch, const float *src, int rows, int T, int col0, int tid) {
    if (((T | pitch | col0) & 3) == 0 && (reinterpret_cast<uintptr_t>(src) & 15u) == 0) {
        // 16-byte pieces: a [96][64] tile is 6 loads per thread, ALL in flight at once — one L2 / HBM round trip per staging
        // phase instead of three (the q-block kernel has seven such phases: they, not its MFMAs, set its duration)
        const int total4 = rows * 16;
        for (int base = 0; base < total4; base += 256 * 8) {
            float4 v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int idx = base + u * 256 + tid;
                const int r = idx >> 4, c = (idx & 15) << 2;
                v[u] = make_float4(0.f, 0.f, 0.f, 0.f);
                if (idx < total4 && col0 + c < T) v[u] = *reinterpret_cast<const float4 *>(src + (long)r * T + col0 + c);
            }
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int idx = base + u * 256 + tid;
                if (idx < total4) *reinterpret_cast<float4 *>(dst + (idx >> 4) * pitch + ((idx & 15) << 2)) = v[u];
            }
        }
        return;
    }
    const int total = rows * 64;
    for (int base = 0; base < total; base += 256 * 8) {
        float v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int idx = base + u * 256 + tid;
            const int r = idx >> 6, c = idx & 63;
            v[u] = (idx < total && col0 + c < T) ? src[(long)r * T + col0 + c] : 0.f;
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int idx = base + u * 256 + tid;
            if (idx < total) dst[(idx >> 6) * pitch + (idx & 63)] = v[u];
        }
    }
}

// The same tile in two halves — its loads into registers, its LDS stores later — so that the NEXT tile's trip to L2 / HBM runs
// under the MFMAs of the tile being multiplied (16-byte pieces only: T % 4 == 0 and 16-byte aligned rows; rows <= 128).
template <int N> struct TileRegsN { float4 v[N]; };       // N * 16 rows at most
typedef TileRegsN<8> TileRegs;
template <int N>
__device__ __forceinline__ void tile_load(TileRegsN<N> &r, const float *src, int rows, int T, int col0, int tid) {
    const int total4 = rows * 16;
#pragma unroll
    for (int u = 0; u < N; ++u) {
        const int idx = u * 256 + tid;
        const int row = idx >> 4, c = (idx & 15) << 2;
        r.v[u] = make_float4(0.f, 0.f, 0.f, 0.f);
        if (idx < total4 && col0 + c < T) r.v[u] = *reinterpret_cast<const float4 *>(src + (long)row * T + col0 + c);
    }
}
template <int N>
__device__ __forceinline__ void tile_store(const TileRegsN<N> &r, float *dst, int pitch, int rows, int tid) {
    const int total4 = rows * 16;
#pragma unroll
    for (int u = 0; u < N; ++u) {
        const int idx = u * 256 + tid;
        if (idx < total4) *reinterpret_cast<float4 *>(dst + (idx >> 4) * pitch + ((idx & 15) << 2)) = r.v[u];
    }
}
__device__ __forceinline__ bool rows16(const void *p) { return (reinterpret_cast<uintptr_t>(p) & 15u) == 0; }

#ifdef GLOWTTS_TRACE   // tuning builds only (tools/trace_attn.py)
__device__ unsigned long long g_attn_trace[1024 * 8];
#define ATTN_TRACE(i) do { if (threadIdx.x == 0) g_attn_trace[((blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x) % 1024 * 8 + (i)] = wall_clock64(); } while (0)
#else
#define ATTN_TRACE(i) do { } while (0)
#endif

// NT: 16-column tiles of the score strip the kernel is compiled for (T <= 16 NT): 10 covers config 2's T_text = 160 with 40
// accumulator registers per lane instead of 64, and every loop over the strip shorter by 3/8.
// DTC: head width / 16 when it is compiled in (6: the model's d_k = 96 — MFMA loops without a uniform branch per d tile, which
// is what lets the compiler put a k-step's LDS reads ahead of the previous step's MFMAs), 0 = any width <= 128.
// LONG (round 4; 256 < T <= 512, NT = 32): the score strip still lives in registers (128 per lane; one wave per SIMD has 512), but
// the probability strip no longer fits LDS (64 x 516 floats).  Phase 2 leaves its results IN the strip registers; phase 4 moves
// them through a per-wave [16 queries][64 keys] LDS chunk, one key tile at a time (same wave writes and reads: no barrier); the
// band P[i][i + r - w] the relative-value term needs is collected in a per-wave [16][16] table while phase 2 passes the
// diagonal tiles; the backward (MODE 1) reads its probabilities from HBM into registers instead of an LDS strip.
constexpr int kCP = 68;    // LDS pitch of a wave's [16][64] probability chunk (LONG)

template <int MODE, bool BF, int NT, int DTC, bool LONG = false>
__global__ __launch_bounds__(256) void attn_qblock_kernel(AttnParams p) {
    extern __shared__ __align__(16) float smem[];
    const int dk = DTC ? DTC * 16 : p.dk, T = p.T, w = p.w, TP = p.TP;
    constexpr int DA = DTC ? DTC : 8;
    const int EP = dk + 4, E2P = dk + 16;
    float *As = smem;                         // [dk][kAP]
    float *Bs = As + dk * kAP;                // [dk][kBP]
    float *Ps = Bs + dk * kBP;                // [64][TP]   (LONG: [4 waves][16][kCP] chunks, then [4][16][16] band tables)
    float *E1s = Ps + (LONG ? 4 * 16 * kCP + 4 * 16 * 16 : 64 * TP);   // [16][EP]     E1[r][d]
    float *E2s = E1s + 16 * EP;               // [16][E2P]    E2[r][d]
    float *Rs = E2s + 16 * E2P;               // [4 waves][16][17]
    unsigned char *Ks = reinterpret_cast<unsigned char *>(Rs + 4 * 16 * 17);   // [64][TP] dropout keep bytes of the strip
    constexpr int NJT = (NT + 3) / 4;

    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int waveu = __builtin_amdgcn_readfirstlane(wave);
    const int lcol = lane & 15, lk = lane >> 4;
    const int q0 = blockIdx.x * 64, h = blockIdx.y, b = blockIdx.z;
    const int DT = DTC ? DTC : (dk >> 4);                 // 16-wide tiles along d
    const int njt = (T + 63) >> 6;                        // 64-key tiles
    const int ntile = (T + 15) >> 4;                      // 16-key tiles (<= NT: checked by the launcher)
    const long cbase = ((long)b * p.H + h) * dk;          // first channel row of this head in a (B, C, T) tensor
    const float *Ag = p.a + cbase * T;
    const float *B1g = p.b1 + cbase * T;
    const float *B2g = p.b2 + cbase * T;
    const float *mk = p.mask + (long)b * T;
    const long pbase = ((long)b * p.H + h) * T * T;
    const bool rel = (p.e1 != nullptr) && (w >= 0);
    const bool has_drop = p.drop != nullptr;

    ATTN_TRACE(0);
    // B tiles are double-buffered through registers when their rows can be moved in 16-byte pieces: the loads of tile n + 1 are
    // issued before the MFMAs of tile n (round 2's kernel made seven trips to memory one after the other: 62 us at T = 160)
    const bool fast = (T & 3) == 0 && rows16(B1g) && rows16(B2g);
    TileRegs nx;
    auto prefetch = [&](const float *src, int col0) { if (fast) tile_load(nx, src, dk, T, col0, tid); };
    auto commit = [&](const float *src, int col0) {
        if (fast) tile_store(nx, Bs, kBP, dk, tid);
        else stage_rows64(Bs, kBP, src, dk, T, col0, tid);
    };
    prefetch(B1g, 0);
    // ---- stage the A block [d][64 queries] and both embedding tables --------------------------------------------------
    stage_rows64(As, kAP, Ag, dk, T, q0, tid);
    for (int base = 0; base < 16 * dk; base += 256 * 8) {       // both tables: loads first (independent), stores after
        float v1[8], v2[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int idx = base + u * 256 + tid;
            const int r = idx / dk, d = idx - r * dk;
            const bool ok = idx < 16 * dk && rel && r <= 2 * w;
            v1[u] = ok ? p.e1[(long)h * p.e_hs + (long)r * dk + d] : 0.f;
            v2[u] = ok ? p.e2[(long)h * p.e_hs + (long)r * dk + d] : 0.f;
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int idx = base + u * 256 + tid;
            const int r = idx / dk, d = idx - r * dk;
            if (idx < 16 * dk) {
                E1s[r * EP + d] = v1[u];
                E2s[r * E2P + d] = v2[u];
            }
        }
    }
    // ---- the strip's keep bytes, and (backward) its probabilities, as whole rows into LDS: phase 2 then reads them with LDS
    // latency, per element, without a global load between its stores
    const int nrow = min(64, T - q0);                     // query rows of this block that exist
    if (has_drop) {
        const unsigned char *dg = p.drop + pbase + (long)q0 * T;
        if ((T & 3) == 0 && (reinterpret_cast<uintptr_t>(dg) & 3u) == 0) {
            const int total = nrow * (T >> 2);            // rows are contiguous: word idx of the strip = word idx in memory
            const unsigned *dgw = reinterpret_cast<const unsigned *>(dg);
            const int W = T >> 2;
            for (int base = 0; base < total; base += 256 * 8) {
                unsigned v[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    const int idx = base + u * 256 + tid;
                    v[u] = idx < total ? dgw[idx] : 0u;
                }
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    const int idx = base + u * 256 + tid;
                    const int r = idx / W, c = idx - r * W;
                    if (idx < total) reinterpret_cast<unsigned *>(Ks + r * TP)[c] = v[u];
                }
            }
        } else {
            for (int idx = tid; idx < nrow * T; idx += 256) {
                const int r = idx / T, c = idx - r * T;
                Ks[r * TP + c] = dg[idx];
            }
        }
    }
    if (MODE == 1 && !LONG) {
        const float *pg = p.p + pbase + (long)q0 * T;
        const int ncol = ntile * 16;                       // columns the strip uses: zeros beyond T and below row nrow
        if ((T & 3) == 0 && rows16(pg)) {
            const int W = ncol >> 2, total = 64 * W;
            for (int base = 0; base < total; base += 256 * 8) {
                float4 v[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    const int idx = base + u * 256 + tid;
                    const int r = idx / W, c = (idx - r * W) << 2;
                    v[u] = make_float4(0.f, 0.f, 0.f, 0.f);
                    if (idx < total && r < nrow && c < T) v[u] = *reinterpret_cast<const float4 *>(pg + (long)r * T + c);
                }
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    const int idx = base + u * 256 + tid;
                    const int r = idx / W, c = (idx - r * W) << 2;
                    if (idx < total) *reinterpret_cast<float4 *>(Ps + r * TP + c) = v[u];
                }
            }
        } else {
            for (int idx = tid; idx < 64 * ncol; idx += 256) {
                const int r = idx / ncol, c = idx - r * ncol;
                Ps[r * TP + c] = (r < nrow && c < T) ? pg[(long)r * T + c] : 0.f;
            }
        }
    }

    // ---- phase 1: S strip (16 queries x T keys per wave) ------------------------------------------------------------------
    f32x4 S[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t) S[t] = f32x4{0.f, 0.f, 0.f, 0.f};
    commit(B1g, 0);
    __syncthreads();
#pragma unroll
    for (int jt = 0; jt < NJT; ++jt) {
        if (jt < njt) {
            if (jt + 1 < njt) prefetch(B1g, (jt + 1) * 64);
            else prefetch(B2g, 0);
            if constexpr (BF) {
                for (int kk = 0; kk < dk; kk += 16) {
                    const bf16x4_s av = bf4_strided(As + (kk + 4 * lk) * kAP + wave * 16 + lcol, kAP);
#pragma unroll
                    for (int ct = 0; ct < 4; ++ct)
                        if (jt * 4 + ct < NT)
                            S[jt * 4 + ct] = mma_bf16(av, bf4_strided(Bs + (kk + 4 * lk) * kBP + ct * 16 + lcol, kBP), S[jt * 4 + ct]);
                }
            } else {
#pragma unroll 4
                for (int kk = 0; kk < dk; kk += 4) {
                    const float av = As[(kk + lk) * kAP + wave * 16 + lcol];
#pragma unroll
                    for (int ct = 0; ct < 4; ++ct) {
                        if (jt * 4 + ct < NT) {
                            const float bv = Bs[(kk + lk) * kBP + ct * 16 + lcol];
                            S[jt * 4 + ct] = __builtin_amdgcn_mfma_f32_16x16x4f32(av, bv, S[jt * 4 + ct], 0, 0, 0);
                        }
                    }
                }
            }
            mfma_settle();                             // (a uniform branch stands between these MFMAs and the strip's first read)
            __syncthreads();                           // every wave is done with this tile
            if (jt + 1 < njt) {
                commit(B1g, (jt + 1) * 64);
                __syncthreads();
            }
        }
    }
    commit(B2g, 0);                                    // first tile of phase 4: visible after the barrier that opens it
    ATTN_TRACE(1);
    // ---- phase 1b: relative term R[i][r] = sum_d A[d][i] E1[r][d], added on the band j - i + w = r ----------------------
    if (rel) {
        f32x4 R = f32x4{0.f, 0.f, 0.f, 0.f};
        if constexpr (BF) {
            for (int kk = 0; kk < dk; kk += 16)
                R = mma_bf16(bf4_strided(As + (kk + 4 * lk) * kAP + wave * 16 + lcol, kAP), bf4_row(E1s + lcol * EP + kk + 4 * lk), R);
        } else {
            for (int kk = 0; kk < dk; kk += 4) {
                const float av = As[(kk + lk) * kAP + wave * 16 + lcol];
                const float bv = E1s[lcol * EP + kk + lk];
                R = __builtin_amdgcn_mfma_f32_16x16x4f32(av, bv, R, 0, 0, 0);
            }
        }
        mfma_settle();
        float *rw = Rs + wave * 16 * 17;
#pragma unroll
        for (int reg = 0; reg < 4; ++reg) rw[(lk * 4 + reg) * 17 + lcol] = R[reg];
        if (lcol == 0) {
#pragma unroll
            for (int reg = 0; reg < 4; ++reg) rw[(lk * 4 + reg) * 17 + 16] = 0.f;       // column 16: "outside the band"
        }
        // (same wave wrote and reads: LDS is in order within a wave; the barrier below is for the compiler)
        __builtin_amdgcn_s_waitcnt(0xc07f);
        __builtin_amdgcn_wave_barrier();
        // the band |j - i| <= w <= 7 of this wave's 16 rows touches the column tiles tq - 1 .. tq + 1 only (a uniform test)
        const int tq = (q0 >> 4) + waveu;
#pragma unroll
        for (int t = 0; t < NT; ++t) {
            if (t >= tq - 1 && t <= tq + 1) {
#pragma unroll
                for (int reg = 0; reg < 4; ++reg) {
                    const int il = lk * 4 + reg;
                    const int r = t * 16 + lcol - (q0 + wave * 16 + il) + w;
                    const int rc = (unsigned)r <= (unsigned)(2 * w) ? r : 16;
                    S[t][reg] += rw[il * 17 + rc];
                }
            }
        }
    }

    ATTN_TRACE(2);
    // ---- phase 2: softmax (MODE 0) / softmax backward (MODE 1) in registers -----------------------------------------------
    // Branch-free per element: every condition that is not uniform is a select, the global stores are range-checked buffer
    // stores (a row beyond T lies beyond the buffer; a column beyond T — last tile only — gets an out-of-range offset), keep
    // bytes and probabilities come from LDS.  (Round 2's form — a guarded load / store per element — compiled to ~90
    // instructions per element, most of them exec-mask bookkeeping: 19 us of the kernel's 58.)
    float mi[4];
    int ig[4], voff[4];
#pragma unroll
    for (int reg = 0; reg < 4; ++reg) {
        ig[reg] = q0 + wave * 16 + lk * 4 + reg;
        mi[reg] = mk[min(ig[reg], T - 1)];
        mi[reg] = ig[reg] < T ? mi[reg] : 0.f;
        voff[reg] = (ig[reg] * T + lcol) * 4;             // byte offset of (row, column lcol) in this head's (T, T) matrix
    }
    float mkc[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t) {
        const int j = t * 16 + lcol;
        mkc[t] = mk[min(j, T - 1)];
        mkc[t] = j < T ? mkc[t] : 0.f;
    }
    const bool lastok = (ntile - 1) * 16 + lcol < T;      // the only tile with columns beyond T is the last one
    const float cb_last = lastok ? 0.f : -3.0e38f;
    const int co_last = lastok ? 0 : 0x40000000;
    const int bl = p.block_len < 0 ? (1 << 20) : p.block_len;
    float *pw = LONG ? Ps + wave * 16 * kCP : Ps + wave * 16 * TP;        // LONG: this wave's [16][kCP] chunk
    float *bandw = Ps + 4 * 16 * kCP + wave * 256;                        // LONG: this wave's [16][16] band table
    const int tqd = (q0 >> 4) + waveu;                                    // the column tile of this wave's diagonal
    const unsigned char *kw = Ks + wave * 16 * TP;
    const __amdgpu_buffer_rsrc_t prs = __builtin_amdgcn_make_buffer_rsrc(
        (MODE == 0 ? p.p : p.ds) + pbase, 0, T * T * 4, 0x00020000);
    f32x4 Pr[(LONG && MODE == 1) ? NT : 1];                                // LONG backward: the probabilities of the strip
    if constexpr (LONG && MODE == 1) {
        const __amdgpu_buffer_rsrc_t pin = __builtin_amdgcn_make_buffer_rsrc(p.p + pbase, 0, T * T * 4, 0x00020000);
#pragma unroll
        for (int t = 0; t < NT; ++t)
#pragma unroll
            for (int reg = 0; reg < 4; ++reg)
                Pr[t][reg] = t < ntile ? __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(
                                             pin, voff[reg] + t * 64 + (t == ntile - 1 ? co_last : 0), 0, 0)) : 0.f;
    }
    if (MODE == 0) {
#pragma unroll
        for (int reg = 0; reg < 4; ++reg) {
            const int il = lk * 4 + reg;
            float mx = -3.0e38f;
#pragma unroll
            for (int t = 0; t < NT; ++t) {
                if (t < ntile) {
                    const bool keep = (mi[reg] * mkc[t] != 0.f) && (abs(t * 16 + lcol - ig[reg]) <= bl);
                    float sv = keep ? S[t][reg] * p.scale : -1e4f;
                    if (t == ntile - 1) sv += cb_last;
                    S[t][reg] = sv;
                    mx = fmaxf(mx, sv);
                }
            }
            mx = group16_max(mx);
            float sum = 0.f;
#pragma unroll
            for (int t = 0; t < NT; ++t) {
                if (t < ntile) {
                    const float e = __expf(S[t][reg] - mx);
                    S[t][reg] = e;
                    sum += e;
                }
            }
            sum = group16_sum(sum);
            const float inv = ig[reg] < T ? 1.0f / sum : 0.f;       // rows beyond T: zeros into the strip, stores dropped
#pragma unroll
            for (int t = 0; t < NT; ++t) {
                if (t < ntile) {
                    float pv = S[t][reg] * inv;
                    __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(pv), prs, voff[reg] + t * 64 + (t == ntile - 1 ? co_last : 0), 0, 0);
                    if (has_drop) pv = kw[il * TP + t * 16 + lcol] ? pv * p.drop_scale : 0.f;
                    if constexpr (LONG) {
                        S[t][reg] = pv;
                        if (t >= tqd - 1 && t <= tqd + 1) {               // (uniform) the band of the relative-value term
                            const int r = t * 16 + lcol - ig[reg] + w;
                            if ((unsigned)r <= (unsigned)(2 * w)) bandw[il * 16 + r] = pv;
                        }
                    } else {
                        pw[il * TP + t * 16 + lcol] = pv;
                    }
                }
            }
        }
    } else {
#pragma unroll
        for (int reg = 0; reg < 4; ++reg) {
            const int il = lk * 4 + reg;
            float dot = 0.f;
#pragma unroll
            for (int t = 0; t < NT; ++t) {
                if (t < ntile) {
                    float dp = S[t][reg];
                    if (has_drop) dp = kw[il * TP + t * 16 + lcol] ? dp * p.drop_scale : 0.f;
                    S[t][reg] = dp;
                    const float pr = LONG ? Pr[LONG ? t : 0][reg] : pw[il * TP + t * 16 + lcol];
                    dot += pr * dp;                                    // (staged / loaded as zero beyond T in either direction)
                }
            }
            dot = group16_sum(dot);
#pragma unroll
            for (int t = 0; t < NT; ++t) {
                if (t < ntile) {
                    const bool keep = (mi[reg] * mkc[t] != 0.f) && (abs(t * 16 + lcol - ig[reg]) <= bl);
                    const float pr = LONG ? Pr[LONG ? t : 0][reg] : pw[il * TP + t * 16 + lcol];
                    const float dsv = keep ? pr * (S[t][reg] - dot) * p.scale : 0.f;
                    __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(dsv), prs, voff[reg] + t * 64 + (t == ntile - 1 ? co_last : 0), 0, 0);
                    if constexpr (LONG) {
                        S[t][reg] = dsv;
                        if (t >= tqd - 1 && t <= tqd + 1) {
                            const int r = t * 16 + lcol - ig[reg] + w;
                            if ((unsigned)r <= (unsigned)(2 * w)) bandw[il * 16 + r] = dsv;
                        }
                    } else {
                        pw[il * TP + t * 16 + lcol] = dsv;
                    }
                }
            }
        }
    }

    ATTN_TRACE(3);
    // ---- phase 4: O (16 queries x dk per wave) = P B2^T + PW E2 -----------------------------------------------------------
    f32x4 O[DA];
#pragma unroll
    for (int dt = 0; dt < DA; ++dt) O[dt] = f32x4{0.f, 0.f, 0.f, 0.f};
    __syncthreads();
    auto pv_tile = [&](int jt, const float *arow) {              // O += P[:, 64-key tile jt] V[tile jt]^T; arow: this lane's A row
        const int jmax = min(64, ((T - jt * 64 + 15) >> 4) << 4);     // keys of this tile that exist in the P strip
        if constexpr (BF) {
            for (int kk = 0; kk < jmax; kk += 16) {
                const bf16x4_s av = bf4_row(arow + kk + 4 * lk);
#pragma unroll
                for (int dt = 0; dt < DA; ++dt)
                    if (DTC || dt < DT) O[dt] = mma_bf16(av, bf4_row(Bs + (dt * 16 + lcol) * kBP + kk + 4 * lk), O[dt]);
                if constexpr (!DTC) mfma_settle();     // (run-time head width: a branch per d tile, accumulators read at the loop head)
            }
        } else {
#pragma unroll 4
            for (int kk = 0; kk < jmax; kk += 4) {
                const float av = arow[kk + lk];
#pragma unroll
                for (int dt = 0; dt < DA; ++dt) {
                    if (DTC || dt < DT) {
                        const float bv = Bs[(dt * 16 + lcol) * kBP + kk + lk];
                        O[dt] = __builtin_amdgcn_mfma_f32_16x16x4f32(av, bv, O[dt], 0, 0, 0);
                    }
                }
                if constexpr (!DTC) mfma_settle();
            }
        }
        mfma_settle();                                 // (callers branch on `jt + 1 < njt` / `rel` before O is read)
    };
    if constexpr (LONG) {
#pragma unroll
        for (int jt = 0; jt < NJT; ++jt) {
            if (jt < njt) {
                if (jt + 1 < njt) prefetch(B2g, (jt + 1) * 64);
                // this wave's [16 queries][64 keys] piece of the strip, from its registers (the reads of the previous piece are
                // in order before these writes: one wave, one LDS queue)
#pragma unroll
                for (int ct = 0; ct < 4; ++ct)
                    if (jt * 4 + ct < NT)
#pragma unroll
                        for (int reg = 0; reg < 4; ++reg) pw[(lk * 4 + reg) * kCP + ct * 16 + lcol] = S[jt * 4 + ct][reg];
                __builtin_amdgcn_s_waitcnt(0xc07f);
                __builtin_amdgcn_wave_barrier();
                pv_tile(jt, pw + lcol * kCP);
                if (jt + 1 < njt) {
                    __syncthreads();
                    commit(B2g, (jt + 1) * 64);
                    __syncthreads();
                }
            }
        }
    } else {
        for (int jt = 0; jt < njt; ++jt) {
            if (jt + 1 < njt) prefetch(B2g, (jt + 1) * 64);
            pv_tile(jt, pw + lcol * TP + jt * 64);
            if (jt + 1 < njt) {
                __syncthreads();
                commit(B2g, (jt + 1) * 64);
                __syncthreads();
            }
        }
    }
    if (rel) {
        const int iq = q0 + wave * 16 + lcol;                  // A operand row = query lcol of this wave
        if constexpr (BF) {                                    // one 16-deep step covers the 2w + 1 <= 15 offsets
            float pv[4];
#pragma unroll
            for (int jj = 0; jj < 4; ++jj) {
                const int r = 4 * lk + jj, j = iq + r - w;
                pv[jj] = (r <= 2 * w && j >= 0 && j < T) ? (LONG ? bandw[lcol * 16 + (r & 15)] : pw[lcol * TP + j]) : 0.f;
            }
            const bf16x4_s av = bf4(pv[0], pv[1], pv[2], pv[3]);
#pragma unroll
            for (int dt = 0; dt < DA; ++dt)
                if (DTC || dt < DT) O[dt] = mma_bf16(av, bf4_strided(E2s + (4 * lk) * E2P + dt * 16 + lcol, E2P), O[dt]);
        } else {
            for (int kk = 0; kk < 2 * w + 1; kk += 4) {
                const int r = kk + lk;
                const int j = iq + r - w;
                const float av = (r <= 2 * w && j >= 0 && j < T) ? (LONG ? bandw[lcol * 16 + (r & 15)] : pw[lcol * TP + j]) : 0.f;
#pragma unroll
                for (int dt = 0; dt < DA; ++dt) {
                    if (DTC || dt < DT) {
                        const float bv = E2s[r * E2P + dt * 16 + lcol];
                        O[dt] = __builtin_amdgcn_mfma_f32_16x16x4f32(av, bv, O[dt], 0, 0, 0);
                    }
                }
            }
        }
        mfma_settle();
    }

    ATTN_TRACE(4);
    // ---- phase 5: transpose O through LDS (reuse the A block) and store rows of 64 queries -----------------------------
    __syncthreads();
#pragma unroll
    for (int dt = 0; dt < DA; ++dt)
        if (DTC || dt < DT)
#pragma unroll
            for (int reg = 0; reg < 4; ++reg) As[(dt * 16 + lcol) * kAP + wave * 16 + lk * 4 + reg] = O[dt][reg];
    __syncthreads();
    float *og = p.out + cbase * T;
    for (int idx = tid; idx < dk * 64; idx += 256) {
        const int d = idx >> 6, i = idx & 63;
        if (q0 + i < T) og[(long)d * T + q0 + i] = As[d * kAP + i];
    }
    ATTN_TRACE(5);
}

// ------------------------------------------------------------------------------------------------------------
// second half of the backward: dV[d][j] = sum_i dO[d][i] Pd[i][j] ;  dK[d][j] = sum_i Q[d][i] dS[i][j]
// workgroup = one 64-key block of one (utterance, head); wave = 16 keys x all d; contraction over queries in 64-chunks
// ------------------------------------------------------------------------------------------------------------
struct AttnDkvParams {
    const float *dout, *q;             // (B, C, T)
    const float *p, *ds;               // (B, h, T, T)
    const unsigned char *drop;
    float *dv, *dkk;                   // (B, C, T)
    int B, H, T, dk;
    float drop_scale;
};

template <bool BF, int DTC>
__global__ __launch_bounds__(256) void attn_dkv_kernel(AttnDkvParams p) {
    extern __shared__ __align__(16) float smem[];
    const int dk = DTC ? DTC * 16 : p.dk, T = p.T;
    constexpr int DA = DTC ? DTC : 8;
    float *Dos = smem;                  // [dk][kBP]  dO chunk   [d][i]
    float *Qs = Dos + dk * kBP;         // [dk][kBP]  Q chunk
    float *Pds = Qs + dk * kBP;         // [64][kAP]  dropped P chunk  [i][j]
    float *Dss = Pds + 64 * kAP;        // [64][kAP]  dS chunk
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int lcol = lane & 15, lk = lane >> 4;
    const int j0 = blockIdx.x * 64, h = blockIdx.y, b = blockIdx.z;
    const int DT = DTC ? DTC : (dk >> 4);
    const long cbase = ((long)b * p.H + h) * dk;
    const long pbase = ((long)b * p.H + h) * T * T;
    const float *dog = p.dout + cbase * T, *qg = p.q + cbase * T;
    f32x4 aV[DA], aK[DA];
#pragma unroll
    for (int dt = 0; dt < DA; ++dt) { aV[dt] = f32x4{0.f, 0.f, 0.f, 0.f}; aK[dt] = f32x4{0.f, 0.f, 0.f, 0.f}; }
    // the four tiles of the next 64-query chunk travel while this chunk is multiplied (16-byte pieces; else staged in place)
    const bool fast = (T & 3) == 0 && rows16(dog) && rows16(qg) && rows16(p.p + pbase) && rows16(p.ds + pbase) &&
                      (!p.drop || (reinterpret_cast<uintptr_t>(p.drop + pbase) & 3u) == 0);
    TileRegs rDo, rQ;
    TileRegsN<4> rP, rDs;                       // 64 rows
    unsigned rK[4];
    auto prefetch = [&](int i0) {
        const int rows = min(64, T - i0);
        tile_load(rDo, dog, dk, T, i0, tid);
        tile_load(rQ, qg, dk, T, i0, tid);
        tile_load(rP, p.p + pbase + (long)i0 * T, rows, T, j0, tid);
        tile_load(rDs, p.ds + pbase + (long)i0 * T, rows, T, j0, tid);
        if (p.drop) {
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int idx = u * 256 + tid;
                const int row = idx >> 4, c = (idx & 15) << 2;
                rK[u] = (row < rows && j0 + c < T)
                            ? *reinterpret_cast<const unsigned *>(p.drop + pbase + (long)(i0 + row) * T + j0 + c) : 0u;
            }
        }
    };
    auto commit = [&]() {
        tile_store(rDo, Dos, kBP, dk, tid);
        tile_store(rQ, Qs, kBP, dk, tid);
        if (p.drop) {
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                rP.v[u].x = (rK[u] & 0xffu) ? rP.v[u].x * p.drop_scale : 0.f;
                rP.v[u].y = (rK[u] & 0xff00u) ? rP.v[u].y * p.drop_scale : 0.f;
                rP.v[u].z = (rK[u] & 0xff0000u) ? rP.v[u].z * p.drop_scale : 0.f;
                rP.v[u].w = (rK[u] & 0xff000000u) ? rP.v[u].w * p.drop_scale : 0.f;
            }
        }
        tile_store(rP, Pds, kAP, 64, tid);            // all 64 rows: rows beyond the chunk were loaded as zeros
        tile_store(rDs, Dss, kAP, 64, tid);
    };
    if (fast) prefetch(0);
    for (int i0 = 0; i0 < T; i0 += 64) {
        __syncthreads();
        if (fast) {
            commit();
        } else {
            stage_rows64(Dos, kBP, dog, dk, T, i0, tid);
            stage_rows64(Qs, kBP, qg, dk, T, i0, tid);
            const int rows = min(64, T - i0);
            stage_rows64(Dss, kAP, p.ds + pbase + (long)i0 * T, rows, T, j0, tid);
            // dropped probabilities: keep byte applied while staging, 8 independent loads per thread as above
            for (int base = 0; base < 64 * 64; base += 256 * 8) {
                float v[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    const int idx = base + u * 256 + tid;
                    const int i = idx >> 6, j = idx & 63;
                    float pv = 0.f;
                    if (i < rows && j0 + j < T) {
                        const long o = pbase + (long)(i0 + i) * T + j0 + j;
                        pv = p.p[o];
                        if (p.drop) pv = p.drop[o] ? pv * p.drop_scale : 0.f;
                    }
                    v[u] = pv;
                }
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    const int idx = base + u * 256 + tid;
                    Pds[(idx >> 6) * kAP + (idx & 63)] = v[u];
                }
            }
            for (int idx = rows * 64 + tid; idx < 64 * 64; idx += 256) Dss[(idx >> 6) * kAP + (idx & 63)] = 0.f;
        }
        __syncthreads();
        if (fast && i0 + 64 < T) prefetch(i0 + 64);
        if constexpr (BF) {
            const int imax = min(64, ((T - i0 + 15) >> 4) << 4);               // (rows / columns beyond T are staged as zeros)
            for (int kk = 0; kk < imax; kk += 16) {
                const bf16x4_s bp = bf4_strided(Pds + (kk + 4 * lk) * kAP + wave * 16 + lcol, kAP);
                const bf16x4_s bd = bf4_strided(Dss + (kk + 4 * lk) * kAP + wave * 16 + lcol, kAP);
#pragma unroll
                for (int dt = 0; dt < DA; ++dt) {
                    if (DTC || dt < DT) {
                        aV[dt] = mma_bf16(bf4_row(Dos + (dt * 16 + lcol) * kBP + kk + 4 * lk), bp, aV[dt]);
                        aK[dt] = mma_bf16(bf4_row(Qs + (dt * 16 + lcol) * kBP + kk + 4 * lk), bd, aK[dt]);
                    }
                }
            }
        } else {
            const int imax = min(64, ((T - i0 + 3) >> 2) << 2);
#pragma unroll 2
            for (int kk = 0; kk < imax; kk += 4) {
                const float bp = Pds[(kk + lk) * kAP + wave * 16 + lcol];      // B[k = query][col = key]
                const float bd = Dss[(kk + lk) * kAP + wave * 16 + lcol];
#pragma unroll
                for (int dt = 0; dt < DA; ++dt) {
                    if (DTC || dt < DT) {
                        const float ao = Dos[(dt * 16 + lcol) * kBP + kk + lk];   // A[row = d][k = query]
                        const float aq = Qs[(dt * 16 + lcol) * kBP + kk + lk];
                        aV[dt] = __builtin_amdgcn_mfma_f32_16x16x4f32(ao, bp, aV[dt], 0, 0, 0);
                        aK[dt] = __builtin_amdgcn_mfma_f32_16x16x4f32(aq, bd, aK[dt], 0, 0, 0);
                    }
                }
            }
        }
    }
    const int j = j0 + wave * 16 + lcol;
#pragma unroll
    for (int dt = 0; dt < DA; ++dt) {
        if ((DTC || dt < DT) && j < T) {
#pragma unroll
            for (int reg = 0; reg < 4; ++reg) {
                const long o = (cbase + dt * 16 + lk * 4 + reg) * T + j;
                p.dv[o] = aV[dt][reg];
                p.dkk[o] = aK[dt][reg];
            }
        }
    }
}

// embedding gradients: dE[r][d] += sum_i M[i][i + r - w] * A[d][i]   (M, A) = (Pd, dO) -> dE_v ; (dS, Q) -> dE_k
// workgroup = one (utterance, head, slab of DC channels): the 2w+1 diagonals of Pd / dS and the slab's dO / Q rows are
// staged in LDS once (coalesced, 8 loads in flight per thread); then thread = one (r, d) output walks the queries out
// of LDS.  DC is chosen by the host so that (2w+1) * DC <= 256 outputs — one per thread — and the slab fits LDS at
// T = 256 whatever the head width.
__global__ __launch_bounds__(256) void attn_relgrad_kernel(const float *__restrict__ p, const float *__restrict__ ds,
                                                           const unsigned char *__restrict__ drop, float drop_scale,
                                                           const float *__restrict__ dout, const float *__restrict__ q,
                                                           float *__restrict__ dek, float *__restrict__ dev, int H, int T,
                                                           int dk, int w, int e_hs, int DC) {
    extern __shared__ __align__(16) float smem[];
    const int nr = 2 * w + 1;
    const int TP = T + 1;               // odd pitch: threads of one wave read different rows at the same column
    float *dp_ = smem;                  // [nr][T]   Pd[i][i+r-w]
    float *dd_ = dp_ + nr * T;          // [nr][T]   dS[i][i+r-w]
    float *os_ = dd_ + nr * T;          // [DC][TP]  dO rows of this slab
    float *qs_ = os_ + DC * TP;         // [DC][TP]  Q rows of this slab
    const int h = blockIdx.x % H, b = blockIdx.x / H;
    const int d0 = blockIdx.y * DC;
    const int nd = min(DC, dk - d0);
    const long cbase = ((long)b * H + h) * dk + d0;
    const long pbase = ((long)b * H + h) * T * T;
    for (int idx = threadIdx.x; idx < nr * T; idx += 256) {
        const int r = idx / T, i = idx - r * T;
        const int j = i + r - w;
        float pv = 0.f, dv = 0.f;
        if (j >= 0 && j < T) {
            const long o = pbase + (long)i * T + j;
            pv = p[o];
            if (drop) pv = drop[o] ? pv * drop_scale : 0.f;
            dv = ds[o];
        }
        dp_[idx] = pv;
        dd_[idx] = dv;
    }
    const int total = nd * T;
    for (int base = 0; base < total; base += 256 * 8) {
        float vo[8], vq[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int idx = base + u * 256 + threadIdx.x;
            vo[u] = idx < total ? dout[cbase * T + idx] : 0.f;
            vq[u] = idx < total ? q[cbase * T + idx] : 0.f;
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int idx = base + u * 256 + threadIdx.x;
            if (idx < total) {
                const int d = idx / T, i = idx - d * T;
                os_[d * TP + i] = vo[u];
                qs_[d * TP + i] = vq[u];
            }
        }
    }
    __syncthreads();
    for (int idx = threadIdx.x; idx < nr * nd; idx += 256) {
        const int r = idx / nd, d = idx - r * nd;
        float sv = 0.f, sk = 0.f;
#pragma unroll 4
        for (int i = 0; i < T; ++i) {
            sv += dp_[r * T + i] * os_[d * TP + i];
            sk += dd_[r * T + i] * qs_[d * TP + i];
        }
        atomicAdd(dev + (long)h * e_hs + r * dk + d0 + d, sv);
        atomicAdd(dek + (long)h * e_hs + r * dk + d0 + d, sk);
    }
}

// attention_long.hip: the same arithmetic beyond the strip kernels' 512 tokens (plain tiled kernels through p_attn / ds)
int attn_long_forward(const float *q, const float *k, const float *v, const float *emb_k, const float *emb_v, const float *mask,
                      const unsigned char *drop, float drop_scale, float *p_attn, float *out, int B, int H, int T, int dk, int w,
                      int e_hs, int block_len, float scale, hipStream_t s);
int attn_long_backward(const float *dout, const float *q, const float *k, const float *v, const float *emb_k, const float *emb_v,
                       const float *mask, const unsigned char *drop, float drop_scale, const float *p_attn, float *ds, float *dq,
                       float *demb_k, float *demb_v, int B, int H, int T, int dk, int w, int e_hs, int block_len, float scale,
                       bool relgrad, hipStream_t s);
constexpr int kAttnStripMaxT = 512;

static int attn_check(const char *name, int B, int H, int T, int dk, int w) {
    GLOWTTS_CHECK_ARG(B >= 0 && H > 0 && T >= 0 && dk > 0, "%s: bad shape", name);
    GLOWTTS_CHECK_ARG(dk % 16 == 0 && dk <= 128, "%s: head width %d must be a multiple of 16 and <= 128", name, dk);
    GLOWTTS_CHECK_ARG(w <= 7, "%s: window %d > 7", name, w);
    return 0;
}

static size_t attn_lds(int dk, int TP, bool lng) {
    const size_t strip = lng ? (size_t)4 * 16 * kCP + 4 * 16 * 16 : (size_t)64 * TP;
    return ((size_t)dk * kAP + (size_t)dk * kBP + strip + (size_t)16 * (dk + 4) + (size_t)16 * (dk + 16) + 4 * 16 * 17) *
               sizeof(float) + (size_t)64 * TP;        // + the keep bytes of the strip
}

template <int MODE, bool BF, int NT, int DTC, bool LONG = false>
static int attn_launch_nt(AttnParams &p, hipStream_t s) {
    p.TP = ((p.T + 15) / 16) * 16 + 4;
    const size_t lds = attn_lds(p.dk, p.TP, LONG);
    GLOWTTS_CHECK_ARG(lds <= 160 * 1024, "glowtts_rel_attn: needs %zu B of LDS", lds);
    static LdsLimit attr_max_e;   // per device: raised only when a launch needs more than any earlier one
    if (int rc_ = attr_max_e.ensure(reinterpret_cast<const void *>(&attn_qblock_kernel<MODE, BF, NT, DTC, LONG>), lds, "glowtts_rel_attn")) return rc_;
    dim3 grid((p.T + 63) / 64, p.H, p.B);
    hipLaunchKernelGGL((attn_qblock_kernel<MODE, BF, NT, DTC, LONG>), grid, dim3(256), lds, s, p);
    GLOWTTS_LAUNCH_CHECK("glowtts_rel_attn");
}

template <int MODE, bool BF>
static int attn_launch(AttnParams &p, hipStream_t s) {
    if (p.T > 256)           // the strip in registers only (round 4): 256 < T <= 512
        return p.dk == 96 ? attn_launch_nt<MODE, BF, 32, 6, true>(p, s) : attn_launch_nt<MODE, BF, 32, 0, true>(p, s);
    if (p.dk == 96) return p.T <= 160 ? attn_launch_nt<MODE, BF, 10, 6>(p, s) : attn_launch_nt<MODE, BF, 16, 6>(p, s);
    return p.T <= 160 ? attn_launch_nt<MODE, BF, 10, 0>(p, s) : attn_launch_nt<MODE, BF, 16, 0>(p, s);
}

}  // namespace glowtts

using namespace glowtts;

extern "C" int glowtts_rel_attn_fwd_ex(const float *q, const float *k, const float *v, const float *emb_k, const float *emb_v,
                                       const float *mask, const unsigned char *drop, float drop_scale, float *p_attn,
                                       float *out, int B, int H, int T, int dk, int window, int heads_share, int block_len,
                                       int bf16_mma, glowtts_stream_t stream) {
    GLOWTTS_CHECK_ARG(q && k && v && mask && p_attn && out, "glowtts_rel_attn_fwd: null pointer");
    if (int rc = attn_check("glowtts_rel_attn_fwd", B, H, T, dk, window)) return rc;
    if ((long)B * T == 0) return 0;
    AttnParams p{};
    p.a = q; p.b1 = k; p.b2 = v; p.e1 = emb_k; p.e2 = emb_v; p.mask = mask; p.drop = drop; p.p = p_attn; p.out = out;
    p.B = B; p.H = H; p.T = T; p.dk = dk; p.w = emb_k ? window : -1; p.block_len = block_len;
    p.e_hs = heads_share ? 0 : (2 * window + 1) * dk;
    p.scale = 1.0f / sqrtf((float)dk); p.drop_scale = drop_scale;
    if (T > kAttnStripMaxT)
        return attn_long_forward(q, k, v, emb_k, emb_v, mask, drop, drop_scale, p_attn, out, B, H, T, dk, window, p.e_hs, block_len,
                                 p.scale, (hipStream_t)stream);
    return bf16_mma ? attn_launch<0, true>(p, (hipStream_t)stream) : attn_launch<0, false>(p, (hipStream_t)stream);
}

extern "C" int glowtts_rel_attn_fwd(const float *q, const float *k, const float *v, const float *emb_k, const float *emb_v,
                                    const float *mask, const unsigned char *drop, float drop_scale, float *p_attn,
                                    float *out, int B, int H, int T, int dk, int window, int heads_share, int block_len,
                                    glowtts_stream_t stream) {
    return glowtts_rel_attn_fwd_ex(q, k, v, emb_k, emb_v, mask, drop, drop_scale, p_attn, out, B, H, T, dk, window, heads_share,
                                   block_len, 0, stream);
}

extern "C" int glowtts_rel_attn_bwd_ex(const float *dout, const float *q, const float *k, const float *v, const float *emb_k,
                                       const float *emb_v, const float *mask, const unsigned char *drop, float drop_scale,
                                       const float *p_attn, float *ds, float *dq, float *dk_out, float *dv, float *demb_k,
                                       float *demb_v, int B, int H, int T, int dk, int window, int heads_share, int block_len,
                                       int bf16_mma, glowtts_stream_t stream) {
    GLOWTTS_CHECK_ARG(dout && q && k && v && mask && p_attn && ds && dq && dk_out && dv, "glowtts_rel_attn_bwd: null pointer");
    GLOWTTS_CHECK_ARG(!emb_k || (emb_v && demb_k && demb_v), "glowtts_rel_attn_bwd: relative embeddings need their gradients");
    if (int rc = attn_check("glowtts_rel_attn_bwd", B, H, T, dk, window)) return rc;
    if ((long)B * T == 0) return 0;
    hipStream_t s = (hipStream_t)stream;
    AttnParams p{};
    p.a = dout; p.b1 = v; p.b2 = k; p.e1 = emb_v; p.e2 = emb_k; p.mask = mask; p.drop = drop;
    p.p = const_cast<float *>(p_attn); p.ds = ds; p.out = dq;
    p.B = B; p.H = H; p.T = T; p.dk = dk; p.w = emb_k ? window : -1; p.block_len = block_len;
    p.e_hs = heads_share ? 0 : (2 * window + 1) * dk;
    p.scale = 1.0f / sqrtf((float)dk); p.drop_scale = drop_scale;
    const bool long_form = T > kAttnStripMaxT;
    if (long_form) {
        if (int rc = attn_long_backward(dout, q, k, v, emb_k, emb_v, mask, drop, drop_scale, p_attn, ds, dq, demb_k, demb_v, B, H, T, dk,
                                        window, p.e_hs, block_len, p.scale, true, s))
            return rc;
    } else if (int rc = bf16_mma ? attn_launch<1, true>(p, s) : attn_launch<1, false>(p, s)) return rc;
    AttnDkvParams d{};
    d.dout = dout; d.q = q; d.p = p_attn; d.ds = ds; d.drop = drop; d.dv = dv; d.dkk = dk_out;
    d.B = B; d.H = H; d.T = T; d.dk = dk; d.drop_scale = drop_scale;
    const size_t lds = ((size_t)2 * dk * kBP + (size_t)2 * 64 * kAP) * sizeof(float);
    {
        const void *fn = bf16_mma ? (dk == 96 ? reinterpret_cast<const void *>(&attn_dkv_kernel<true, 6>) : reinterpret_cast<const void *>(&attn_dkv_kernel<true, 0>))
                                  : (dk == 96 ? reinterpret_cast<const void *>(&attn_dkv_kernel<false, 6>) : reinterpret_cast<const void *>(&attn_dkv_kernel<false, 0>));
        static LdsLimit lim[4];   // per device and instantiation: raised only when a launch needs more than any earlier one
        if (int rc_ = lim[(bf16_mma ? 2 : 0) + (dk == 96 ? 1 : 0)].ensure(fn, lds, "glowtts_rel_attn_bwd")) return rc_;
        const dim3 grid((T + 63) / 64, H, B);
        if (bf16_mma) {
            if (dk == 96) hipLaunchKernelGGL((attn_dkv_kernel<true, 6>), grid, dim3(256), lds, s, d);
            else hipLaunchKernelGGL((attn_dkv_kernel<true, 0>), grid, dim3(256), lds, s, d);
        } else {
            if (dk == 96) hipLaunchKernelGGL((attn_dkv_kernel<false, 6>), grid, dim3(256), lds, s, d);
            else hipLaunchKernelGGL((attn_dkv_kernel<false, 0>), grid, dim3(256), lds, s, d);
        }
    }
    if (emb_k && !long_form) {
        const int nr = 2 * window + 1;
        int DC = (256 / nr) & ~7;                      // one (r, d) output per thread, slab rows a multiple of 8
        if (DC > dk) DC = dk;
        const size_t lds_r = ((size_t)2 * nr * T + (size_t)2 * DC * (T + 1)) * sizeof(float);
        static LdsLimit attr_max_r;   // per device: raised only when a launch needs more than any earlier one
    if (int rc_ = attr_max_r.ensure(reinterpret_cast<const void *>(&attn_relgrad_kernel), lds_r, "glowtts_rel_attn_bwd")) return rc_;
        hipLaunchKernelGGL(attn_relgrad_kernel, dim3(B * H, (dk + DC - 1) / DC), dim3(256), lds_r, s, p_attn, ds, drop,
                           drop_scale, dout, q, demb_k, demb_v, H, T, dk, window, p.e_hs, DC);
    }
    GLOWTTS_LAUNCH_CHECK("glowtts_rel_attn_bwd");
}

extern "C" int glowtts_rel_attn_bwd(const float *dout, const float *q, const float *k, const float *v, const float *emb_k,
                                    const float *emb_v, const float *mask, const unsigned char *drop, float drop_scale,
                                    const float *p_attn, float *ds, float *dq, float *dk_out, float *dv, float *demb_k,
                                    float *demb_v, int B, int H, int T, int dk, int window, int heads_share, int block_len,
                                    glowtts_stream_t stream) {
    return glowtts_rel_attn_bwd_ex(dout, q, k, v, emb_k, emb_v, mask, drop, drop_scale, p_attn, ds, dq, dk_out, dv, demb_k, demb_v,
                                   B, H, T, dk, window, heads_share, block_len, 0, stream);
}

#ifdef GLOWTTS_TRACE
extern "C" int glowtts_debug_attn_trace_read(unsigned long long *host, int n_words) {
    return (int)hipMemcpyFromSymbol(host, HIP_SYMBOL(glowtts::g_attn_trace), (size_t)n_words * 8);
}
#endif
