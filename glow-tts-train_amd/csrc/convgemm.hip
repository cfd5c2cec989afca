// convgemm.hip — fp32 MFMA implicit-GEMM 1-D convolution family for gfx950 (the dense contractions of the WN
// coupling network, layers.py:138-162, and of the 1x1 start/end convs, attentions.py:124-126).
//
// Why hand-written: the first profile (profiles/r01_*_baseline.csv) shows PyTorch-ROCm's path spends, per
// training step, ~18 ms in MIOpen igemm kernels PLUS ~5 ms transposing NCHW<->NHWC around every call, ~10 ms in
// ~1 700 tiny bias / mask / add / reduce launches, and falls back to "naive_conv" kernels for strided inputs.
// These kernels consume the reference layout (B, C, T) directly, keep T contiguous for coalesced 64-B row
// segments, and fuse bias, mask, the WN gate (tanh * sigmoid, optional dropout + conditioning) and the
// residual/skip update into the GEMM epilogue.
//
// Arithmetic: v_mfma_f32_16x16x4_f32 — exact fp32 FMA chains (MI355X_MICROARCH.md: 157 TFLOP/s dense peak, same
// as the vector rate, no reduced-precision path), so results match an fp32 reference to rounding-order noise.
//
// Forward-type kernel (conv fwd, conv bwd-data with transposed/flipped packed weights, 1x1 GEMMs):
//     out[m, t] = sum_tap sum_k Wp[tap][k][m] * X[k][t + tap*dil - pad]            per utterance b
//   workgroup = 4 waves = (64*RTW rows) x (16*NCT columns) of one utterance; wave = RTW x NCT tiles of 16x16;
//   K loop over 16-channel chunks: packed weights [tap][16][rows] and the activation slab [16][cols + halo] are
//   staged in LDS (row pitches chosen == 16 mod 32 so the two k-rows a 32-lane group touches fall on disjoint
//   banks); every tap re-reads the same activation slab at a shifted column, which is what makes the k-tap
//   convolution an "implicit" GEMM with no im2col traffic.
// Weight-gradient kernel:
//     dWp[tap][k][m] += sum_{b,t} X[b][k][t + tap*dil - pad] * D[b][m][t]
//   A = X rows (k), B = D rows (m), contraction over time; split over utterance groups, one float atomic per
//   element and workgroup (64-B contiguous segments along m).
#include "common.hpp"

namespace glowtts {

typedef float f32x4 __attribute__((ext_vector_type(4)));

enum { EPI_PLAIN = 0, EPI_GATE = 1, EPI_RESSKIP = 2, EPI_RESSKIP_LAST = 3, EPI_ADD = 4 };

struct ConvGemmParams {
    const float *x;          // (B, Cin, T) activations, batch stride x_bs elements
    const float *wp;         // packed weights [taps][Cin][M], M contiguous
    const float *bias;       // [M] or null
    const float *mask;       // (B, T) or null (applied where the epilogue says so)
    const float *cond;       // EPI_GATE: (B, 2H) conditioning added before the gate, or null
    const float *r0;         // EPI_RESSKIP: x_in (B,H,T) ; EPI_ADD: addend (B,M,T)
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
    float drop_scale;        // 1 / (1 - p)
};

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
        if ((p.M & 3) == 0) {
            const int nw4 = p.taps * KT * (WGR / 4);
            for (int idx = tid; idx < nw4; idx += 256) {
                const int m4 = idx % (WGR / 4);
                const int rest = idx / (WGR / 4);
                const int k = rest % KT, tap = rest / KT;
                const int lr = m4 * 4;
                float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
                if (kc0 + k < p.Cin && row_ok(lr))
                    v = *reinterpret_cast<const float4 *>(p.wp + ((long)tap * p.Cin + kc0 + k) * p.M + grow(lr));
                *reinterpret_cast<float4 *>(Ws + (tap * KT + k) * WP + lr) = v;
            }
        } else {   // row count not a multiple of 4: rows of the packed matrix are not 16-byte aligned
            const int nw = p.taps * KT * WGR;
            for (int idx = tid; idx < nw; idx += 256) {
                const int lr = idx % WGR;
                const int rest = idx / WGR;
                const int k = rest % KT, tap = rest / KT;
                float v = 0.f;
                if (kc0 + k < p.Cin && row_ok(lr)) v = p.wp[((long)tap * p.Cin + kc0 + k) * p.M + grow(lr)];
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

    // ---- epilogue: lane holds rows lk*4 + reg, column lrow of every 16x16 tile ----------------------------------------
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
                    const float th = tanhf(vt), sg = sigmoidf_(vs);
                    p.y0[((long)b * p.H + ch) * p.T + t] = th * sg;
                    if (p.y1) { p.y1[ot] = th; p.y1[os] = sg; }
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
                if (EPI == EPI_PLAIN) {
                    if (p.mask_out) v *= m;
                    p.y0[(long)b * p.y_bs + (long)row * p.T + t] = v;
                } else if (EPI == EPI_ADD) {
                    const float add = p.r0[(long)b * p.r_bs + (long)row * p.T + t];
                    p.y0[(long)b * p.y_bs + (long)row * p.T + t] = v + (p.mask_add ? add * m : add);
                } else if (EPI == EPI_RESSKIP) {
                    // rows [0,H): residual -> next layer input ; rows [H,2H): skip accumulation  (layers.py:157-159)
                    if (row < p.H) {
                        const long o = ((long)b * p.H + row) * p.T + t;
                        p.y0[o] = (p.r0[o] + v) * m;
                    } else {
                        const long o = ((long)b * p.H + (row - p.H)) * p.T + t;
                        p.y1[o] = (p.r1 ? p.r1[o] : 0.f) + v;
                    }
                } else if (EPI == EPI_RESSKIP_LAST) {
                    // last layer: all H rows go to the skip sum, and WN's final `output * x_mask` is folded in (:161-162)
                    const long o = ((long)b * p.H + row) * p.T + t;
                    p.y1[o] = ((p.r1 ? p.r1[o] : 0.f) + v) * m;
                }
            }
        }
    }
}

// ------------------------------------------------------------------------------------------------------------
// weight gradient
// ------------------------------------------------------------------------------------------------------------
struct ConvWrwParams {
    const float *x;      // (B, Cin, T) forward input, batch stride x_bs
    const float *d;      // (B, M, T) output gradient, batch stride d_bs
    float *dwp;          // [taps][Cin][M] accumulated (atomics)
    const float *mask;   // (B, T): multiply d by it while staging, or null
    long x_bs, d_bs;
    int B, Cin, M, T, taps, dil, pad;
    int nb;              // utterances per workgroup (split of the contraction)
    int xs_pitch, ds_pitch;
};

// workgroup: 64 input channels (4 k-tiles, one per wave) x 128 output channels (8 m-tiles per wave), one tap,
// nb utterances.  Time is consumed in chunks of 64 frames.
__global__ __launch_bounds__(256) void convwrw_kernel(ConvWrwParams p) {
    constexpr int CT = 64;
    extern __shared__ __align__(16) float smem[];
    const int XP = p.xs_pitch, DP = p.ds_pitch;
    float *Xs = smem;                 // [64][XP]   rows = input channel, cols = frames (shifted by the tap)
    float *Ds = smem + 64 * XP;       // [128][DP]  rows = output channel
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int lrow = lane & 15, lk = lane >> 4;
    const int nkt = (p.Cin + 63) / 64;
    const int kt = blockIdx.x % nkt, mt = blockIdx.x / nkt;
    const int tap = blockIdx.y;
    const int b0 = blockIdx.z * p.nb;
    const int b1 = min(p.B, b0 + p.nb);
    const int k0 = kt * 64, m0 = mt * 128;
    const int shift = tap * p.dil - p.pad;

    f32x4 acc[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};

    for (int b = b0; b < b1; ++b) {
        const float *xb = p.x + (long)b * p.x_bs;
        const float *db = p.d + (long)b * p.d_bs;
        for (int tc = 0; tc < p.T; tc += CT) {
            for (int idx = tid; idx < 64 * CT; idx += 256) {
                const int r = idx / CT, j = idx - r * CT;
                const int t = tc + j + shift;
                float v = 0.f;
                if (k0 + r < p.Cin && tc + j < p.T && t >= 0 && t < p.T) v = xb[(long)(k0 + r) * p.T + t];
                Xs[r * XP + j] = v;
            }
            for (int idx = tid; idx < 128 * CT; idx += 256) {
                const int r = idx / CT, j = idx - r * CT;
                float v = 0.f;
                if (m0 + r < p.M && tc + j < p.T) {
                    v = db[(long)(m0 + r) * p.T + tc + j];
                    if (p.mask) v *= p.mask[(long)b * p.T + tc + j];
                }
                Ds[r * DP + j] = v;
            }
            __syncthreads();
#pragma unroll 4
            for (int c4 = 0; c4 < CT / 4; ++c4) {
                const float a = Xs[(wave * 16 + lrow) * XP + c4 * 4 + lk];          // A[row = channel][k = frame]
#pragma unroll
                for (int i = 0; i < 8; ++i) {
                    const float bv = Ds[(i * 16 + lrow) * DP + c4 * 4 + lk];         // B[k = frame][col = out channel]
                    acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, bv, acc[i], 0, 0, 0);
                }
            }
            __syncthreads();
        }
    }
#pragma unroll
    for (int i = 0; i < 8; ++i) {
#pragma unroll
        for (int reg = 0; reg < 4; ++reg) {
            const int k = k0 + wave * 16 + lk * 4 + reg;
            const int m = m0 + i * 16 + lrow;
            if (k < p.Cin && m < p.M) atomicAdd(p.dwp + ((long)tap * p.Cin + k) * p.M + m, acc[i][reg]);
        }
    }
}

// ------------------------------------------------------------------------------------------------------------
// weight packing (+ weight norm) and its backward; row sums for bias gradients
// ------------------------------------------------------------------------------------------------------------
// one workgroup per output channel o:  w[o] = v[o] * g[o] / ||v[o]||  (torch.nn.utils.weight_norm, dim 0) or w = v
//   wp_f[tap][c][o]            = w[o][c][tap]          forward packing   (rows = o, K = c)
//   wp_b[taps-1-tap][o][c]     = w[o][c][tap]          backward-data packing (rows = c, K = o, taps flipped)
__global__ __launch_bounds__(256) void pack_weight_kernel(const float *__restrict__ v, const float *__restrict__ g,
                                                          float *__restrict__ wp_f, float *__restrict__ wp_b,
                                                          float *__restrict__ inv_norm, int Cout, int Cin, int taps) {
    __shared__ float red[4];
    const int o = blockIdx.x;
    const int n = Cin * taps;
    const float *vo = v + (long)o * n;
    float scale = 1.f;
    if (g != nullptr) {
        float s = 0.f;
        for (int i = threadIdx.x; i < n; i += 256) s += vo[i] * vo[i];
        s = block_sum_256(s, red);
        const float inv = 1.0f / sqrtf(s);
        scale = g[o] * inv;
        if (threadIdx.x == 0 && inv_norm) inv_norm[o] = inv;
    }
    for (int i = threadIdx.x; i < n; i += 256) {
        const int c = i / taps, tap = i - c * taps;
        const float w = vo[i] * scale;
        if (wp_f) wp_f[((long)tap * Cin + c) * Cout + o] = w;
        if (wp_b) wp_b[((long)(taps - 1 - tap) * Cout + o) * Cin + c] = w;
    }
}

// backward of the packing: dw[o][c][tap] = dwp[tap][c][o];  plain conv: dweight += dw
// weight norm: dg[o] += sum(dw * v) / n ;  dv += (g / n) * (dw - v * sum(dw * v) / n^2)      (n = ||v[o]||)
__global__ __launch_bounds__(256) void unpack_weight_grad_kernel(const float *__restrict__ dwp, const float *__restrict__ v,
                                                                 const float *__restrict__ g, const float *__restrict__ inv_norm,
                                                                 float *__restrict__ dv, float *__restrict__ dg, int Cout,
                                                                 int Cin, int taps) {
    __shared__ float red[4];
    const int o = blockIdx.x;
    const int n = Cin * taps;
    const float *vo = v + (long)o * n;
    if (g == nullptr) {
        for (int i = threadIdx.x; i < n; i += 256) {
            const int c = i / taps, tap = i - c * taps;
            dv[(long)o * n + i] += dwp[((long)tap * Cin + c) * Cout + o];
        }
        return;
    }
    float dot = 0.f;
    for (int i = threadIdx.x; i < n; i += 256) {
        const int c = i / taps, tap = i - c * taps;
        dot += dwp[((long)tap * Cin + c) * Cout + o] * vo[i];
    }
    dot = block_sum_256(dot, red);
    const float inv = inv_norm[o];
    const float gn = g[o] * inv;
    if (threadIdx.x == 0) dg[o] += dot * inv;
    const float proj = dot * inv * inv;
    for (int i = threadIdx.x; i < n; i += 256) {
        const int c = i / taps, tap = i - c * taps;
        dv[(long)o * n + i] += gn * (dwp[((long)tap * Cin + c) * Cout + o] - vo[i] * proj);
    }
}

// out[m] += sum_{b,t} d[b][m][t]   (bias gradients) — grid (M, slabs of utterances)
__global__ __launch_bounds__(256) void rowsum_kernel(const float *__restrict__ d, const float *__restrict__ mask,
                                                     float *__restrict__ out, long d_bs, int B, int M, int T, int nb) {
    __shared__ float red[4];
    const int m = blockIdx.x;
    const int b0 = blockIdx.y * nb, b1 = min(B, b0 + nb);
    float s = 0.f;
    for (int b = b0; b < b1; ++b) {
        const float *row = d + (long)b * d_bs + (long)m * T;
        for (int t = threadIdx.x; t < T; t += 256) s += mask ? row[t] * mask[(long)b * T + t] : row[t];
    }
    s = block_sum_256(s, red);
    if (threadIdx.x == 0) atomicAdd(out + m, s);
}

static inline int pitch16(int n) {          // smallest pitch >= n with pitch % 32 == 16
    int p = (n + 15) / 32 * 32 + 16;
    if (p - 32 >= n) p -= 32;
    return p;
}
static inline int pitch2(int n) {           // smallest pitch >= n with pitch % 32 == 2
    int p = (n - 2 + 31) / 32 * 32 + 2;
    return p < n ? p + 32 : p;
}

template <int RTW, int NCT, int EPI>
static int launch_convgemm(ConvGemmParams &p, hipStream_t s) {
    constexpr int WGR = 64 * RTW, NT = 16 * NCT;
    p.xp_pitch = pitch16(NT + (p.taps - 1) * p.dil);
    const size_t lds = ((size_t)p.taps * 16 * (WGR + 16) + (size_t)16 * p.xp_pitch) * sizeof(float);
    GLOWTTS_CHECK_ARG(lds <= 160 * 1024, "glowtts_conv: %d taps x dilation %d needs %zu B of LDS", p.taps, p.dil, lds);
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(&convgemm_kernel<RTW, NCT, EPI>),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) { set_error("glowtts_conv: LDS attribute: %s", hipGetErrorString(e)); return (int)e; }
    const int ntile_t = (p.T + NT - 1) / NT;
    const int rows = (EPI == EPI_GATE) ? p.H : p.M;
    const int per = (EPI == EPI_GATE) ? 64 : WGR;
    dim3 grid(ntile_t * p.B, (rows + per - 1) / per);
    hipLaunchKernelGGL((convgemm_kernel<RTW, NCT, EPI>), grid, dim3(256), lds, s, p);
    GLOWTTS_LAUNCH_CHECK("glowtts_conv");
}

template <int EPI>
static int dispatch_convgemm(ConvGemmParams &p, hipStream_t s) {
    // 80-frame tiles when T divides evenly (e.g. T' = 400), else 64; 128-row workgroups unless M is small
    const bool n5 = (p.T % 80 == 0) || (p.T % 64 != 0 && ((p.T + 79) / 80) * 80 < ((p.T + 63) / 64) * 64);
    const bool big = (EPI == EPI_GATE) || (p.M % 128 == 0) || p.M > 192;
    if (big) return n5 ? launch_convgemm<2, 5, EPI>(p, s) : launch_convgemm<2, 4, EPI>(p, s);
    return n5 ? launch_convgemm<1, 5, EPI>(p, s) : launch_convgemm<1, 4, EPI>(p, s);
}

}  // namespace glowtts

using namespace glowtts;

static int check_conv_common(const char *name, const void *x, const void *wp, int B, int Cin, int M, int T, int taps,
                             int dil, int pad) {
    GLOWTTS_CHECK_ARG(x && wp, "%s: null pointer", name);
    GLOWTTS_CHECK_ARG(B >= 0 && Cin > 0 && M > 0 && T >= 0 && taps >= 1 && dil >= 1 && pad >= 0,
                      "%s: bad shape B=%d Cin=%d M=%d T=%d taps=%d dil=%d pad=%d", name, B, Cin, M, T, taps, dil, pad);
    GLOWTTS_CHECK_ARG(aligned16(wp), "%s: packed weights must be 16-byte aligned", name);
    return 0;
}

extern "C" int glowtts_conv_fwd(const float *x, long x_bs, const float *wp, const float *bias, const float *mask,
                                const float *addend, long addend_bs, float *y, long y_bs, int B, int Cin, int M, int T,
                                int taps, int dil, int pad, int mask_in, int mask_out, int mask_add,
                                glowtts_stream_t stream) {
    if (int rc = check_conv_common("glowtts_conv_fwd", x, wp, B, Cin, M, T, taps, dil, pad)) return rc;
    GLOWTTS_CHECK_ARG(y, "glowtts_conv_fwd: null output");
    GLOWTTS_CHECK_ARG(!(mask_in || mask_out || mask_add) || mask, "glowtts_conv_fwd: mask flag without mask");
    if ((long)B * T == 0) return 0;
    ConvGemmParams p{};
    p.x = x; p.wp = wp; p.bias = bias; p.mask = mask; p.r0 = addend; p.y0 = y;
    p.x_bs = x_bs; p.y_bs = y_bs; p.B = B; p.Cin = Cin; p.M = M; p.T = T; p.taps = taps; p.dil = dil; p.pad = pad;
    p.mask_in = mask_in; p.mask_out = mask_out; p.mask_add = mask_add; p.r_bs = addend_bs;
    return addend ? dispatch_convgemm<EPI_ADD>(p, (hipStream_t)stream) : dispatch_convgemm<EPI_PLAIN>(p, (hipStream_t)stream);
}

extern "C" int glowtts_conv_gate_fwd(const float *x, const float *wp, const float *bias, const float *cond,
                                     const unsigned char *drop, float drop_scale, float *acts, float *ts, int B, int H,
                                     int T, int taps, int dil, int pad, glowtts_stream_t stream) {
    if (int rc = check_conv_common("glowtts_conv_gate_fwd", x, wp, B, H, 2 * H, T, taps, dil, pad)) return rc;
    GLOWTTS_CHECK_ARG(acts, "glowtts_conv_gate_fwd: null output");
    GLOWTTS_CHECK_ARG(H % 4 == 0, "glowtts_conv_gate_fwd: hidden width %d must be a multiple of 4", H);
    if ((long)B * T == 0) return 0;
    ConvGemmParams p{};
    p.x = x; p.wp = wp; p.bias = bias; p.cond = cond; p.drop = drop; p.drop_scale = drop_scale; p.y0 = acts; p.y1 = ts;
    p.x_bs = (long)H * T; p.B = B; p.Cin = H; p.M = 2 * H; p.H = H; p.T = T; p.taps = taps; p.dil = dil; p.pad = pad;
    return dispatch_convgemm<EPI_GATE>(p, (hipStream_t)stream);
}

extern "C" int glowtts_conv_res_skip_fwd(const float *acts, const float *wp, const float *bias, const float *mask,
                                         const float *x_in, const float *skip_in, float *x_out, float *skip_out, int B,
                                         int H, int T, int last, glowtts_stream_t stream) {
    const int M = last ? H : 2 * H;
    if (int rc = check_conv_common("glowtts_conv_res_skip_fwd", acts, wp, B, H, M, T, 1, 1, 0)) return rc;
    GLOWTTS_CHECK_ARG(mask && skip_out && (last || (x_in && x_out)), "glowtts_conv_res_skip_fwd: null pointer");
    if ((long)B * T == 0) return 0;
    ConvGemmParams p{};
    p.x = acts; p.wp = wp; p.bias = bias; p.mask = mask; p.r0 = x_in; p.r1 = skip_in; p.y0 = x_out; p.y1 = skip_out;
    p.x_bs = (long)H * T; p.B = B; p.Cin = H; p.M = M; p.H = H; p.T = T; p.taps = 1; p.dil = 1; p.pad = 0;
    return last ? dispatch_convgemm<EPI_RESSKIP_LAST>(p, (hipStream_t)stream)
                : dispatch_convgemm<EPI_RESSKIP>(p, (hipStream_t)stream);
}

extern "C" int glowtts_conv_wrw(const float *x, long x_bs, const float *d, long d_bs, const float *mask, float *dwp, int B,
                                int Cin, int M, int T, int taps, int dil, int pad, glowtts_stream_t stream) {
    GLOWTTS_CHECK_ARG(x && d && dwp, "glowtts_conv_wrw: null pointer");
    GLOWTTS_CHECK_ARG(B >= 0 && Cin > 0 && M > 0 && T >= 0 && taps >= 1 && dil >= 1 && pad >= 0, "glowtts_conv_wrw: bad shape");
    if ((long)B * T == 0) return 0;
    ConvWrwParams p{};
    p.x = x; p.d = d; p.dwp = dwp; p.mask = mask; p.x_bs = x_bs; p.d_bs = d_bs;
    p.B = B; p.Cin = Cin; p.M = M; p.T = T; p.taps = taps; p.dil = dil; p.pad = pad;
    const int tiles = ((Cin + 63) / 64) * ((M + 127) / 128) * taps;
    int splits = (768 + tiles - 1) / tiles;          // aim at ~3 workgroups per CU
    if (splits > B) splits = B;
    if (splits < 1) splits = 1;
    p.nb = (B + splits - 1) / splits;
    p.xs_pitch = pitch2(64);
    p.ds_pitch = pitch2(64);
    const size_t lds = ((size_t)64 * p.xs_pitch + (size_t)128 * p.ds_pitch) * sizeof(float);
    dim3 grid(((Cin + 63) / 64) * ((M + 127) / 128), taps, (B + p.nb - 1) / p.nb);
    hipLaunchKernelGGL(convwrw_kernel, grid, dim3(256), lds, (hipStream_t)stream, p);
    GLOWTTS_LAUNCH_CHECK("glowtts_conv_wrw");
}

extern "C" int glowtts_pack_weight(const float *v, const float *g, float *wp_f, float *wp_b, float *inv_norm, int Cout,
                                   int Cin, int taps, glowtts_stream_t stream) {
    GLOWTTS_CHECK_ARG(v && (wp_f || wp_b), "glowtts_pack_weight: null pointer");
    GLOWTTS_CHECK_ARG(!g || inv_norm, "glowtts_pack_weight: weight norm needs inv_norm");
    GLOWTTS_CHECK_ARG(Cout > 0 && Cin > 0 && taps > 0, "glowtts_pack_weight: bad shape");
    hipLaunchKernelGGL(pack_weight_kernel, dim3(Cout), dim3(256), 0, (hipStream_t)stream, v, g, wp_f, wp_b, inv_norm, Cout, Cin, taps);
    GLOWTTS_LAUNCH_CHECK("glowtts_pack_weight");
}

extern "C" int glowtts_unpack_weight_grad(const float *dwp, const float *v, const float *g, const float *inv_norm,
                                          float *dv, float *dg, int Cout, int Cin, int taps, glowtts_stream_t stream) {
    GLOWTTS_CHECK_ARG(dwp && v && dv, "glowtts_unpack_weight_grad: null pointer");
    GLOWTTS_CHECK_ARG(!g || (inv_norm && dg), "glowtts_unpack_weight_grad: weight norm needs inv_norm and dg");
    GLOWTTS_CHECK_ARG(Cout > 0 && Cin > 0 && taps > 0, "glowtts_unpack_weight_grad: bad shape");
    hipLaunchKernelGGL(unpack_weight_grad_kernel, dim3(Cout), dim3(256), 0, (hipStream_t)stream, dwp, v, g, inv_norm, dv, dg, Cout, Cin, taps);
    GLOWTTS_LAUNCH_CHECK("glowtts_unpack_weight_grad");
}

extern "C" int glowtts_rowsum(const float *d, long d_bs, const float *mask, float *out, int B, int M, int T,
                              glowtts_stream_t stream) {
    GLOWTTS_CHECK_ARG(d && out, "glowtts_rowsum: null pointer");
    GLOWTTS_CHECK_ARG(B >= 0 && M > 0 && T >= 0, "glowtts_rowsum: bad shape");
    if ((long)B * T == 0) return 0;
    int slabs = (1024 + M - 1) / M;
    if (slabs > B) slabs = B;
    if (slabs < 1) slabs = 1;
    const int nb = (B + slabs - 1) / slabs;
    hipLaunchKernelGGL(rowsum_kernel, dim3(M, (B + nb - 1) / nb), dim3(256), 0, (hipStream_t)stream, d, mask, out, d_bs, B, M, T, nb);
    GLOWTTS_LAUNCH_CHECK("glowtts_rowsum");
}
