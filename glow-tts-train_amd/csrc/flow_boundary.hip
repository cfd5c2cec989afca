// flow_boundary.hip — what lies between the WN stacks of two consecutive flow blocks, in ONE launch (round 4).
//
//   block k    : out = end(skip)                                   attentions.py:131-133  (1x1 conv H -> C, fp32 weights)
//                z   = [y0 ; (m + e^logs' y1) mask], logdet_k += sum logs' mask          attentions.py:135-142
//   block k + 1: y'  = W ((bias + e^logs z) mask) mask, logdet_{k+1} = (sum logs + log det W C/n) x_len
//                                                                  layers.py:182-199 (ActNorm), 238-272 (InvConvNear)
//                h0  = (start(y'[:, :C/2]) + b) mask               attentions.py:122-123  (1x1 conv C/2 -> H, weight norm)
//
// Before: three launches on the decoder's chain per block boundary — end conv (16 us), the fused coupling / ActNorm / InvConv
// kernel (7-10 us), start conv (11 us) — for 1.5 us of matrix work: each is its fixed costs (DESIGN.md lessons 31, 35).  Here a
// workgroup owns a 32-frame tile of one utterance and walks the three steps with the tile in LDS; every tensor the backward
// needs (out, y', h0) is written once, z is never written.
//   phase 0  skip tile (H x 32 fp32) -> LDS, k-packed [g][frame][16 channels] (convgemm.hip's activation image)
//   phase 1  out = W_end skip + b_end on v_mfma_f32_16x16x4_f32, weights straight from L2 in the packed layout [g][M][16]
//            (the same products in the same order as convgemm_wd_kernel: the per-operator path's numbers bit for bit)
//   phase 2  out tile -> LDS [C][33]; element-wise flows per (group of n_split channels, frame quad): thread-local, as in
//            coupling_ai_fwd_kernel (flows.hip); out and y' leave with 16-byte stores, y'[:, :C/2] goes to LDS k-packed
//   phase 3  h0 = (W_start y'0 + b_start) mask, same MFMA loop, through LDS to 16-byte stores
// Limits (host-checked; anything else takes the three launches): fp32 tensors, T % 4 == 0, C <= 192, H <= 192, n_split 2 or 4.
#include "common.hpp"

namespace glowtts {

typedef float f32x4 __attribute__((ext_vector_type(4)));

struct BoundaryParams {
    const float *skip, *wp_end, *b_end, *y_prev, *mask, *logs, *bias, *w, *logdet_w, *x_len, *wp_start, *b_start;
    float *out, *y, *h0, *logdet_prev, *logdet;
    int B, C, H, T, sig;
    int exp;      // timing experiments: bit 0 = no end-conv MFMAs, bit 1 = no start-conv MFMAs, bit 2 = no element-wise phase
};

__device__ __forceinline__ float bnd_coupling_logs(float raw, bool sig) { return sig ? logf(1e-6f + sigmoidf_(raw + 2.0f)) : raw; }
template <int N>
__device__ __forceinline__ int bnd_channel(int k, int g, int C) { return (k / (N / 2)) * (C / 2) + g * (N / 2) + (k % (N / 2)); }

constexpr int kBndKP = 20, kBndMaxG = 12, kBndRT = 3;

// D[rows of this wave's tiles][32 frames] = W X: X k-packed in LDS ([g][32][kBndKP]), W packed [g][M][16] in global memory.
// wave w owns row tiles w, w + 4, w + 8 (rows beyond M: zero weights through the buffer descriptor's range check)
// the weights of k group `g` for this lane's row tiles (issued early by the caller for g = 0: they fly under the phase before)
__device__ __forceinline__ void bnd_wload(const float *__restrict__ wp, int M, int G, int g, int wave, int lrow, int lk,
                                          f32x4 (&a)[kBndRT]) {
    const int wbytes = G * M * 64;
    const __amdgpu_buffer_rsrc_t wrs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(wp), 0, wbytes, 0x00020000);
    const int so = g < G ? g * M * 64 : wbytes;
#pragma unroll
    for (int r = 0; r < kBndRT; ++r) {
        const int row = (wave + 4 * r) * 16 + lrow;
        a[r] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(wrs, row < M ? (row * 16 + lk * 4) * 4 : wbytes, so, 0));
    }
}

template <int NT>
__device__ __forceinline__ void bnd_gemm(const float *__restrict__ wp, int M, int G, const float *Xs, int wave, int lrow, int lk,
                                         f32x4 (&a)[2][kBndRT], f32x4 (&acc)[kBndRT][NT / 16]) {      // a[0]: group 0, loaded by the caller
    constexpr int NC = NT / 16;
#pragma unroll
    for (int r = 0; r < kBndRT; ++r)
#pragma unroll
        for (int c = 0; c < NC; ++c) acc[r][c] = f32x4{0.f, 0.f, 0.f, 0.f};
    auto wload = [&](int g, int slot) { bnd_wload(wp, M, G, g, wave, lrow, lk, a[slot]); };
    const int waveu = __builtin_amdgcn_readfirstlane(wave);
    bool t_ok[kBndRT];                                   // (scalar conditions: row tiles beyond M cost no MFMAs)
#pragma unroll
    for (int r = 0; r < kBndRT; ++r) t_ok[r] = (waveu + 4 * r) * 16 < M;
    const float *xd = Xs + lrow * kBndKP + lk * 4;
    auto step = [&](int g, const f32x4 (&aw)[kBndRT]) {
        f32x4 bv[NC];
#pragma unroll
        for (int c = 0; c < NC; ++c) bv[c] = *reinterpret_cast<const f32x4 *>(xd + (g * NT + c * 16) * kBndKP);
#pragma unroll
        for (int r = 0; r < kBndRT; ++r)
            if (t_ok[r]) {
#pragma unroll
                for (int j = 0; j < 4; ++j)
#pragma unroll
                    for (int c = 0; c < NC; ++c)
                        acc[r][c] = __builtin_amdgcn_mfma_f32_16x16x4f32(aw[r][j], bv[c][j], acc[r][c], 0, 0, 0);
            }
    };
    for (int g = 0; g < G; g += 2) {                     // (two steps per trip: the weight ring's slots stay compile-time)
        wload(g + 1, 1);
        step(g, a[0]);
        wload(g + 2, 0);
        if (g + 1 < G) step(g + 1, a[1]);
    }
    mfma_settle();
}

template <int N, int NT>
__global__ __launch_bounds__(256, NT == 16 ? 3 : 2) void flow_boundary_fwd_kernel(BoundaryParams p) {
    constexpr int kBndNT = NT, kBndOP = NT + 1, NC = NT / 16, NQ = NT / 4;      // frames, LDS pitch of the row-major tiles, 16-frame column tiles, frame quads
    constexpr int NSEL = 256 / (16 * NQ), NGI = kBndMaxG / NSEL;                // staging: group = gsel + NSEL gi
    static_assert(NT == 16 || NT == 32, "16 or 32 frames per workgroup");
    extern __shared__ __align__(16) float smem[];
    const int C = p.C, H = p.H, T = p.T, half = C / 2;
    const int GH = (H + 15) / 16, GS = (half + 15) / 16;
    float *Xs = smem;                                    // [GH][32][kBndKP]   skip tile, k-packed
    float *Os = Xs + kBndMaxG * kBndNT * kBndKP;         // [192][NT + 1]      out tile, later the h0 tile
    float *Ys = Os + 192 * kBndOP;                       // [6][32][kBndKP]    y'[:, :C/2] tile, k-packed
    float *Ms = Ys + 6 * kBndNT * kBndKP;                // [32]               mask
    float *red = Ms + kBndNT;                            // [4]
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, lrow = lane & 15, lk = lane >> 4;
    const int ntile = (T + kBndNT - 1) / kBndNT;
    const int b = blockIdx.x / ntile, t0 = (blockIdx.x - b * ntile) * kBndNT;

    // ---- loads that have no producer inside the kernel go out first: the end conv's first weights, this thread's pieces of y_k
    f32x4 aw[2][kBndRT];
    bnd_wload(p.wp_end, C, GH, 0, wave, lrow, lk, aw[0]);
    const int G = C / N;
    float4 yin[2][N];
#pragma unroll
    for (int u = 0; u < 2; ++u) {
        const int i = tid + 256 * u, g = i / NQ, q = i % NQ;
        const bool ok = i < G * NQ && t0 + q * 4 < T;
#pragma unroll
        for (int k = 0; k < N; ++k) {
            yin[u][k] = make_float4(0.f, 0.f, 0.f, 0.f);
            if (ok) yin[u][k] = *reinterpret_cast<const float4 *>(p.y_prev + ((long)b * C + bnd_channel<N>(k, g, C)) * T + t0 + q * 4);
        }
    }

    // ---- phase 0: skip tile and mask into LDS ---------------------------------------------------------------------------
    {
        const int kk = tid & 15, qq = (tid >> 4) % NQ, gsel = tid / (16 * NQ);
        const __amdgpu_buffer_rsrc_t xrs = __builtin_amdgcn_make_buffer_rsrc(
            const_cast<float *>(p.skip + (long)b * H * T), 0, H * T * 4, 0x00020000);
        const bool tok = t0 + qq * 4 < T;
        f32x4 v[NGI];
#pragma unroll
        for (int gi = 0; gi < NGI; ++gi) {
            const int ch = (gsel + NSEL * gi) * 16 + kk;
            v[gi] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(
                                                  xrs, (tok && ch < H) ? (ch * T + t0 + qq * 4) * 4 : 0x7fffffff, 0, 0));
        }
        if (tid < kBndNT) Ms[tid] = t0 + tid < T ? p.mask[(long)b * T + t0 + tid] : 0.f;
        for (int i = tid; i < 6 * kBndNT * kBndKP; i += 256) Ys[i] = 0.f;      // (channels beyond C/2 of the last group: zeros)
#pragma unroll
        for (int gi = 0; gi < NGI; ++gi) {
            float *d = Xs + ((gsel + NSEL * gi) * kBndNT + qq * 4) * kBndKP + kk;
            d[0] = v[gi][0]; d[kBndKP] = v[gi][1]; d[2 * kBndKP] = v[gi][2]; d[3 * kBndKP] = v[gi][3];
        }
    }
    __syncthreads();

    // ---- phase 1: out = W_end skip + b_end -> LDS [C][33] -------------------------------------------------------------
    f32x4 acc[kBndRT][NC];
    bnd_gemm<NT>(p.wp_end, C, (GLOWTTS_EXP_BITS(p.exp) & 1) ? 0 : GH, Xs, wave, lrow, lk, aw, acc);
    bnd_wload(p.wp_start, H, GS, 0, wave, lrow, lk, aw[0]);          // (the start conv's first weights: under phase 2)
#pragma unroll
    for (int r = 0; r < kBndRT; ++r) {
        const int row0 = (wave + 4 * r) * 16 + lk * 4;
#pragma unroll
        for (int reg = 0; reg < 4; ++reg) {
            const int row = row0 + reg;
            if (row < C) {
                const float bb = p.b_end[row];
#pragma unroll
                for (int c = 0; c < NC; ++c) Os[row * kBndOP + c * 16 + lrow] = acc[r][c][reg] + bb;
            }
        }
    }
    __syncthreads();

    // ---- phase 2: out leaves; coupling of block k, ActNorm + InvConv of block k + 1 per (group, frame quad) -----------------
    for (int i = tid; i < C * NQ; i += 256) {              // out tile -> global, 16-byte stores along t
        const int row = i / NQ, q = i % NQ;
        if (t0 + q * 4 < T) {
            const float *s = Os + row * kBndOP + q * 4;
            *reinterpret_cast<float4 *>(p.out + ((long)b * C + row) * T + t0 + q * 4) = make_float4(s[0], s[1], s[2], s[3]);
        }
    }
    float wr[N * N];
#pragma unroll
    for (int q = 0; q < N * N; ++q) wr[q] = p.w[q];
    float ld = 0.f;
#pragma unroll
    for (int u = 0; u < 2; ++u) {                          // (G * 8 <= 512 items: C <= 192, n_split >= 2 ... checked on the host)
        const int i = tid + 256 * u;
        const int g = i / NQ, q = i % NQ;
        if (i >= G * NQ || t0 + q * 4 >= T || (GLOWTTS_EXP_BITS(p.exp) & 4)) continue;
        const float *mq = Ms + q * 4;
        float yv[N][4];
#pragma unroll
        for (int k = 0; k < N; ++k) {
            const int ch = bnd_channel<N>(k, g, C);
            const float4 zin = yin[u][k];
            float z[4] = {zin.x, zin.y, zin.z, zin.w};
            if (k >= N / 2) {                              // second half: the affine apply of block k
                const float *m = Os + (ch - half) * kBndOP + q * 4, *lr = Os + ch * kBndOP + q * 4;
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const float l = bnd_coupling_logs(lr[j], p.sig != 0);
                    z[j] = (m[j] + expf(l) * z[j]) * mq[j];
                    ld += l * mq[j];
                }
            }
            const float e = expf(p.logs[ch]), bi = p.bias[ch];
#pragma unroll
            for (int j = 0; j < 4; ++j) yv[k][j] = (bi + e * z[j]) * mq[j];
        }
#pragma unroll
        for (int o = 0; o < N; ++o) {
            float r4[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                float s2 = 0.f;
#pragma unroll
                for (int k = 0; k < N; ++k) s2 += wr[o * N + k] * yv[k][j];
                r4[j] = s2 * mq[j];
            }
            const int ch = bnd_channel<N>(o, g, C);
            *reinterpret_cast<float4 *>(p.y + ((long)b * C + ch) * T + t0 + q * 4) = make_float4(r4[0], r4[1], r4[2], r4[3]);
            if (o < N / 2) {                               // first half: the start conv's input, k-packed
                float *d = Ys + ((ch >> 4) * kBndNT + q * 4) * kBndKP + (ch & 15);
                d[0] = r4[0]; d[kBndKP] = r4[1]; d[2 * kBndKP] = r4[2]; d[3 * kBndKP] = r4[3];
            }
        }
    }
    ld = block_sum_256(ld, red);                           // (two barriers inside: Ys is complete and Os is free after it)
    if (tid == 0) atomicAdd(p.logdet_prev + b, ld);
    if (blockIdx.x == 0) {
        __syncthreads();
        float s2 = 0.f;
        for (int c = tid; c < C; c += 256) s2 += p.logs[c];
        s2 = block_sum_256(s2, red);
        const float l0 = s2 + p.logdet_w[0] * (float)(C / N);
        for (int bb = tid; bb < p.B; bb += 256) p.logdet[bb] = l0 * p.x_len[bb];
    }

    // ---- phase 3: h0 = (W_start y'0 + b_start) mask ------------------------------------------------------------------------
    bnd_gemm<NT>(p.wp_start, H, (GLOWTTS_EXP_BITS(p.exp) & 2) ? 0 : GS, Ys, wave, lrow, lk, aw, acc);
#pragma unroll
    for (int r = 0; r < kBndRT; ++r) {
        const int row0 = (wave + 4 * r) * 16 + lk * 4;
#pragma unroll
        for (int reg = 0; reg < 4; ++reg) {
            const int row = row0 + reg;
            if (row < H) {
                const float bb = p.b_start[row];
#pragma unroll
                for (int c = 0; c < NC; ++c) Os[row * kBndOP + c * 16 + lrow] = (acc[r][c][reg] + bb) * Ms[c * 16 + lrow];
            }
        }
    }
    __syncthreads();
    for (int i = tid; i < H * NQ; i += 256) {
        const int row = i / NQ, q = i % NQ;
        if (t0 + q * 4 < T) {
            const float *s = Os + row * kBndOP + q * 4;
            *reinterpret_cast<float4 *>(p.h0 + ((long)b * H + row) * T + t0 + q * 4) = make_float4(s[0], s[1], s[2], s[3]);
        }
    }
}


// ------------------------------------------------------------------------------------------------------------------------------
// The same boundary backwards (block k + 1's backward ends, block k's begins):
//   dyf  = dy_{k+1} + [ W_start^T (dx_wn mask) ; 0 ]                       (start conv's backward-data, added into the first half)
//   ActNorm + InvConv backward of block k + 1 on z recomputed from (y_k, out_k), then block k's coupling backward: dy_k, dout_k
//        (the arithmetic of coupling_ai_bwd_kernel, flows.hip, item by item)
//   dskip_k = (W_end^T dout_k) [mask]                                      (end conv's backward-data)
// The parameter gradients of block k + 1's ActNorm / InvConv (dlogs, dbias, dW: sums over every frame) leave the kernel as one
// partial per (workgroup, group): flow_boundary_bwd_reduce_kernel adds them up — off the chain, on the weight-gradient stream.
struct BoundaryBwdParams {
    const float *dxw, *wb_start, *dyn, *y_prev, *out_prev, *mask, *logs, *bias, *w, *dlogdet, *wb_end;
    float *dy_prev, *dout_prev, *dskip, *partial;
    int B, C, H, T, sig, mask_dskip;
    int exp;      // timing experiments: bit 0 = no start-conv MFMAs, bit 1 = no end-conv MFMAs, bit 2 = no element-wise phase, bit 3 = no group reduction
};

template <int N, int NT>
__global__ __launch_bounds__(256, 2) void flow_boundary_bwd_kernel(BoundaryBwdParams p) {
    constexpr int OP = NT + 1, NC = NT / 16, NQ = NT / 4, NV = 2 * N + N * N;
    constexpr int NSEL = 256 / (16 * NQ), NGI = kBndMaxG / NSEL;
    extern __shared__ __align__(16) float smem[];
    const int C = p.C, H = p.H, T = p.T, half = C / 2;
    const int GH = (H + 15) / 16, GC = (C + 15) / 16, G = C / N;
    float *Xs = smem;                                    // [12][NT][kBndKP]  dx_wn tile, later the dout tile (k-packed)
    float *Os = Xs + kBndMaxG * NT * kBndKP;             // [192][NT + 1]     the start conv's input gradient, later the dskip tile
    float *Ms = Os + 192 * OP;                           // [NT]
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, lrow = lane & 15, lk = lane >> 4;
    const int ntile = (T + NT - 1) / NT;
    const int b = blockIdx.x / ntile, t0 = (blockIdx.x - b * ntile) * NT;

    f32x4 aw[2][kBndRT];
    bnd_wload(p.wb_start, half, GH, 0, wave, lrow, lk, aw[0]);
    float4 yin[2][N], gin[2][N], oin[2][N];              // this thread's pieces of y_k, dy_{k+1}, out_k (m for k < N/2 ... by channel)
#pragma unroll
    for (int u = 0; u < 2; ++u) {
        const int i = tid + 256 * u, g = i / NQ, q = i % NQ;
        const bool ok = i < G * NQ && t0 + q * 4 < T;
#pragma unroll
        for (int k = 0; k < N; ++k) {
            const long o = ((long)b * C + bnd_channel<N>(k, g, C)) * T + t0 + q * 4;
            yin[u][k] = gin[u][k] = oin[u][k] = make_float4(0.f, 0.f, 0.f, 0.f);
            if (ok) {
                yin[u][k] = *reinterpret_cast<const float4 *>(p.y_prev + o);
                gin[u][k] = *reinterpret_cast<const float4 *>(p.dyn + o);
                oin[u][k] = *reinterpret_cast<const float4 *>(p.out_prev + o);
            }
        }
    }
    // ---- phase 0: dx_wn tile (times the mask) into LDS, k-packed ----------------------------------------------------------
    {
        const int kk = tid & 15, qq = (tid >> 4) % NQ, gsel = tid / (16 * NQ);
        const __amdgpu_buffer_rsrc_t xrs = __builtin_amdgcn_make_buffer_rsrc(
            const_cast<float *>(p.dxw + (long)b * H * T), 0, H * T * 4, 0x00020000);
        const bool tok = t0 + qq * 4 < T;
        f32x4 v[NGI];
#pragma unroll
        for (int gi = 0; gi < NGI; ++gi) {
            const int ch = (gsel + NSEL * gi) * 16 + kk;
            v[gi] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(
                                                  xrs, (tok && ch < H) ? (ch * T + t0 + qq * 4) * 4 : 0x7fffffff, 0, 0));
        }
        if (tid < NT) Ms[tid] = t0 + tid < T ? p.mask[(long)b * T + t0 + tid] : 0.f;
        float4 mv = make_float4(0.f, 0.f, 0.f, 0.f);
        if (tok) mv = *reinterpret_cast<const float4 *>(p.mask + (long)b * T + t0 + qq * 4);
#pragma unroll
        for (int gi = 0; gi < NGI; ++gi) {
            float *d = Xs + ((gsel + NSEL * gi) * NT + qq * 4) * kBndKP + kk;
            d[0] = v[gi][0] * mv.x; d[kBndKP] = v[gi][1] * mv.y; d[2 * kBndKP] = v[gi][2] * mv.z; d[3 * kBndKP] = v[gi][3] * mv.w;
        }
    }
    __syncthreads();

    // ---- phase 1: the start conv's input gradient (C/2 rows) -> LDS ---------------------------------------------------------
    f32x4 acc[kBndRT][NC];
    bnd_gemm<NT>(p.wb_start, half, (GLOWTTS_EXP_BITS(p.exp) & 1) ? 0 : GH, Xs, wave, lrow, lk, aw, acc);
    bnd_wload(p.wb_end, H, GC, 0, wave, lrow, lk, aw[0]);            // (the end conv's first weights: under phase 2)
#pragma unroll
    for (int r = 0; r < kBndRT; ++r) {
        const int row0 = (wave + 4 * r) * 16 + lk * 4;
#pragma unroll
        for (int reg = 0; reg < 4; ++reg)
            if (row0 + reg < half)
#pragma unroll
                for (int c = 0; c < NC; ++c) Os[(row0 + reg) * OP + c * 16 + lrow] = acc[r][c][reg];
    }
    __syncthreads();                                     // Os complete; every wave is done reading Xs

    // ---- phase 2: per (group, frame quad): ActNorm / InvConv backward of block k + 1, coupling backward of block k ----------
    float wr[N * N];
#pragma unroll
    for (int q = 0; q < N * N; ++q) wr[q] = p.w[q];
#pragma unroll
    for (int u = 0; u < 2; ++u) {
        const int i = tid + 256 * u;
        const int g = i / NQ, q = i % NQ;
        const bool item = i < G * NQ;                    // (uniform over the 8 lanes of a group: they reduce together below)
        const bool live = item && t0 + q * 4 < T && !(GLOWTTS_EXP_BITS(p.exp) & 4);
        float aw_[N * N], al[N], ab[N];
#pragma unroll
        for (int v = 0; v < N * N; ++v) aw_[v] = 0.f;
#pragma unroll
        for (int k = 0; k < N; ++k) { al[k] = 0.f; ab[k] = 0.f; }
        if (live) {
            const float *mq = Ms + q * 4;
            const float dld = p.dlogdet ? p.dlogdet[b] : 0.f;
            float xv[N][4], gz[N][4], y1[N / 2][4], el[N / 2][4], lr[N / 2][4], e[N], bi[N];
            int chn[N];
#pragma unroll
            for (int k = 0; k < N; ++k) {
                chn[k] = bnd_channel<N>(k, g, C);
                e[k] = expf(p.logs[chn[k]]);
                bi[k] = p.bias[chn[k]];
                const float4 a = yin[u][k], d = gin[u][k];
                xv[k][0] = a.x; xv[k][1] = a.y; xv[k][2] = a.z; xv[k][3] = a.w;
                gz[k][0] = d.x; gz[k][1] = d.y; gz[k][2] = d.z; gz[k][3] = d.w;
                if (k < N / 2) {                         // first half: plus the start conv's input gradient
                    const float *s = Os + chn[k] * OP + q * 4;
#pragma unroll
                    for (int j = 0; j < 4; ++j) gz[k][j] += s[j];
                }
            }
#pragma unroll
            for (int k = N / 2; k < N; ++k) {            // z1 = (m + e^logs' y1) mask, recomputed
                const float4 m4 = oin[u][k - N / 2], l4 = oin[u][k];
                const float m[4] = {m4.x, m4.y, m4.z, m4.w};
                lr[k - N / 2][0] = l4.x; lr[k - N / 2][1] = l4.y; lr[k - N / 2][2] = l4.z; lr[k - N / 2][3] = l4.w;
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    y1[k - N / 2][j] = xv[k][j];
                    const float ee = expf(bnd_coupling_logs(lr[k - N / 2][j], p.sig != 0));
                    el[k - N / 2][j] = ee;
                    xv[k][j] = (m[j] + ee * xv[k][j]) * mq[j];
                }
            }
#pragma unroll
            for (int k = 0; k < N; ++k)
#pragma unroll
                for (int j = 0; j < 4; ++j) gz[k][j] *= mq[j];
#pragma unroll
            for (int k = 0; k < N; ++k) {
                float o[4];                              // gradient of z (channel k of the group)
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const float yk = (bi[k] + e[k] * xv[k][j]) * mq[j];
                    float dy = 0.f;
#pragma unroll
                    for (int oo = 0; oo < N; ++oo) {
                        dy += wr[oo * N + k] * gz[oo][j];
                        aw_[oo * N + k] += gz[oo][j] * yk;
                    }
                    const float dym = dy * mq[j];
                    o[j] = dym * e[k];
                    al[k] += dym * e[k] * xv[k][j];
                    ab[k] += dym;
                }
                const long go = ((long)b * C + chn[k]) * T + t0 + q * 4;
                if (k < N / 2) {                         // first half passes through the coupling
                    *reinterpret_cast<float4 *>(p.dy_prev + go) = make_float4(o[0], o[1], o[2], o[3]);
                } else {
                    float dx1[4], dm[4], dl[4];
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        const float gm = o[j] * mq[j];
                        const float ee = el[k - N / 2][j];
                        dx1[j] = gm * ee;
                        dm[j] = gm;
                        float dlp = gm * ee * y1[k - N / 2][j] + dld * mq[j];
                        if (p.sig) {
                            const float sg = sigmoidf_(lr[k - N / 2][j] + 2.0f);
                            dlp *= sg * (1.0f - sg) / (1e-6f + sg);
                        }
                        dl[j] = dlp;
                    }
                    *reinterpret_cast<float4 *>(p.dy_prev + go) = make_float4(dx1[0], dx1[1], dx1[2], dx1[3]);
                    *reinterpret_cast<float4 *>(p.dout_prev + go - (long)half * T) = make_float4(dm[0], dm[1], dm[2], dm[3]);
                    *reinterpret_cast<float4 *>(p.dout_prev + go) = make_float4(dl[0], dl[1], dl[2], dl[3]);
                    const int cm = chn[k] - half, cl = chn[k];                 // dout rows: dm, dlogs' -> the end conv's operand
                    float *d0 = Xs + ((cm >> 4) * NT + q * 4) * kBndKP + (cm & 15);
                    float *d1 = Xs + ((cl >> 4) * NT + q * 4) * kBndKP + (cl & 15);
#pragma unroll
                    for (int j = 0; j < 4; ++j) { d0[j * kBndKP] = dm[j]; d1[j * kBndKP] = dl[j]; }
                }
            }
        } else if (item) {                               // a quad beyond T: zeros into the operand tile
#pragma unroll
            for (int k = 0; k < N; ++k) {
                const int ch = bnd_channel<N>(k, g, C);
                float *d = Xs + ((ch >> 4) * NT + q * 4) * kBndKP + (ch & 15);
#pragma unroll
                for (int j = 0; j < 4; ++j) d[j * kBndKP] = 0.f;
            }
        }
        // the group's NQ lanes add up (NQ = 8 consecutive lanes); lane q == 0 hands the partial over
        if (item && !(GLOWTTS_EXP_BITS(p.exp) & 8)) {                      // (G * NQ is a multiple of NQ: the NQ lanes of a group are all in or all out)
#pragma unroll
            for (int off = 1; off < NQ; off <<= 1) {
#pragma unroll
                for (int k = 0; k < N; ++k) { al[k] += __shfl_xor(al[k], off, 64); ab[k] += __shfl_xor(ab[k], off, 64); }
#pragma unroll
                for (int v = 0; v < N * N; ++v) aw_[v] += __shfl_xor(aw_[v], off, 64);
            }
            if (item && q == 0) {
                float *d = p.partial + ((long)blockIdx.x * G + g) * NV;
#pragma unroll
                for (int k = 0; k < N; ++k) { d[k] = al[k]; d[N + k] = ab[k]; }
#pragma unroll
                for (int v = 0; v < N * N; ++v) d[2 * N + v] = aw_[v];
            }
        }
    }
    // channels beyond C of the last k group: zeros (their weights are zero, but LDS garbage may be NaN)
    for (int i = tid; i < (GC * 16 - C) * NT; i += 256) {
        const int ch = C + i / NT, f = i % NT;
        Xs[((ch >> 4) * NT + f) * kBndKP + (ch & 15)] = 0.f;
    }
    __syncthreads();

    // ---- phase 3: dskip = (W_end^T dout) [mask] ------------------------------------------------------------------------------
    bnd_gemm<NT>(p.wb_end, H, (GLOWTTS_EXP_BITS(p.exp) & 2) ? 0 : GC, Xs, wave, lrow, lk, aw, acc);
#pragma unroll
    for (int r = 0; r < kBndRT; ++r) {
        const int row0 = (wave + 4 * r) * 16 + lk * 4;
#pragma unroll
        for (int reg = 0; reg < 4; ++reg)
            if (row0 + reg < H)
#pragma unroll
                for (int c = 0; c < NC; ++c)
                    Os[(row0 + reg) * OP + c * 16 + lrow] = p.mask_dskip ? acc[r][c][reg] * Ms[c * 16 + lrow] : acc[r][c][reg];
    }
    __syncthreads();
    for (int i = tid; i < H * NQ; i += 256) {
        const int row = i / NQ, q = i % NQ;
        if (t0 + q * 4 < T) {
            const float *s = Os + row * OP + q * 4;
            *reinterpret_cast<float4 *>(p.dskip + ((long)b * H + row) * T + t0 + q * 4) = make_float4(s[0], s[1], s[2], s[3]);
        }
    }
}

// dlogs / dbias / dW of one ActNorm + InvConv from the per-workgroup partials: workgroup = group g, thread = (value v, slice of the
// workgroups); the log-determinant terms (d sum(logs) x_len, d log det W) ride in as in actnorm_invconv_bwd_kernel
template <int N>
__global__ __launch_bounds__(256) void flow_boundary_bwd_reduce_kernel(const float *__restrict__ partial, int n_wg, const float *__restrict__ w_inv,
                                                                       const float *__restrict__ dlogdet, const float *__restrict__ x_len,
                                                                       float *__restrict__ dlogs, float *__restrict__ dbias,
                                                                       float *__restrict__ dw, int B, int C) {
    constexpr int NV = 2 * N + N * N, NS = 256 / NV;
    __shared__ float part[NS][NV];
    __shared__ float tt_s;
    const int g = blockIdx.x, G = C / N, tid = threadIdx.x;
    const int v = tid % NV, sl = tid / NV;
    if (sl < NS) {
        float s = 0.f;
        for (int wg = sl; wg < n_wg; wg += NS) s += partial[((long)wg * G + g) * NV + v];
        part[sl][v] = s;
    }
    if (tid >= 192) {                                    // (wave 3: the log-determinant's factor)
        float t = 0.f;
        if (dlogdet != nullptr)
            for (int bb = tid - 192; bb < B; bb += 64) t += dlogdet[bb] * x_len[bb];
        t = wave_sum(t);
        if (tid == 192) tt_s = t;
    }
    __syncthreads();
    if (tid < NV) {
        float s = 0.f;
#pragma unroll
        for (int k = 0; k < NS; ++k) s += part[k][tid];
        const float tt = tt_s;
        if (tid < N) atomicAdd(dlogs + bnd_channel<N>(tid, g, C), s + tt);
        else if (tid < 2 * N) atomicAdd(dbias + bnd_channel<N>(tid - N, g, C), s);
        else {
            const int r = tid - 2 * N;
            if (g == 0 && dlogdet != nullptr) s += w_inv[(r % N) * N + r / N] * (float)(C / N) * tt;
            atomicAdd(dw + r, s);
        }
    }
}

}  // namespace glowtts

using namespace glowtts;

extern "C" int glowtts_flow_boundary_fwd(const float *skip, const float *wp_end, const float *b_end, const float *y_prev,
                                         const float *mask, const float *logs, const float *bias, const float *w,
                                         const float *logdet_w, const float *x_len, const float *wp_start, const float *b_start,
                                         float *out, float *y, float *h0, float *logdet_prev, float *logdet, int B, int C, int H,
                                         int T, int n_split, int sigmoid_scale, glowtts_stream_t stream) {
    GLOWTTS_CHECK_ARG(skip && wp_end && b_end && y_prev && mask && logs && bias && w && logdet_w && x_len && wp_start && b_start &&
                          out && y && h0 && logdet_prev && logdet,
                      "glowtts_flow_boundary_fwd: null pointer");
    GLOWTTS_CHECK_ARG(n_split == 2 || n_split == 4, "glowtts_flow_boundary_fwd: n_split=%d (2 or 4)", n_split);
    GLOWTTS_CHECK_ARG(B >= 0 && T >= 0 && C > 0 && C <= 192 && C / n_split <= 64 && H > 0 && H <= 192 && C % n_split == 0 && C % 2 == 0 &&
                          T % 4 == 0,
                      "glowtts_flow_boundary_fwd: shape (B=%d, C=%d, H=%d, T=%d): C, H <= 192, T %% 4 == 0", B, C, H, T);
    GLOWTTS_CHECK_ARG(aligned16(skip) && aligned16(y_prev) && aligned16(out) && aligned16(y) && aligned16(h0) && aligned16(wp_end) &&
                          aligned16(wp_start),
                      "glowtts_flow_boundary_fwd: tensors must be 16-byte aligned");
    if ((long)B * T == 0) return 0;
    BoundaryParams p{skip, wp_end, b_end, y_prev, mask, logs, bias, w, logdet_w, x_len, wp_start, b_start,
                     out, y, h0, logdet_prev, logdet, B, C, H, T, sigmoid_scale, GLOWTTS_EXP_BITS(knob(K_BND_EXP))};
    // 32 frames per workgroup (400 workgroups at config 2).  16-frame tiles (800 workgroups, three per CU) were measured: 36 us against
    // 28 — every workgroup streams both weight matrices from L2, and halving the tile halves the use of each weight register
    constexpr int nt = 32;
    constexpr size_t lds = ((size_t)kBndMaxG * nt * kBndKP + 192 * (nt + 1) + 6 * nt * kBndKP + nt + 4) * sizeof(float);
    const dim3 grid(B * ((T + nt - 1) / nt));
    static LdsLimit lim[2];
    if (n_split == 4) {
        if (int rc_ = lim[0].ensure(reinterpret_cast<const void *>(&flow_boundary_fwd_kernel<4, nt>), lds, "glowtts_flow_boundary_fwd")) return rc_;
        hipLaunchKernelGGL((flow_boundary_fwd_kernel<4, nt>), grid, dim3(256), lds, (hipStream_t)stream, p);
    } else {
        if (int rc_ = lim[1].ensure(reinterpret_cast<const void *>(&flow_boundary_fwd_kernel<2, nt>), lds, "glowtts_flow_boundary_fwd")) return rc_;
        hipLaunchKernelGGL((flow_boundary_fwd_kernel<2, nt>), grid, dim3(256), lds, (hipStream_t)stream, p);
    }
    GLOWTTS_LAUNCH_CHECK("glowtts_flow_boundary_fwd");
}

extern "C" int glowtts_flow_boundary_bwd(const float *dx_wn, const float *wb_start, const float *dy_next, const float *y_prev,
                                         const float *out_prev, const float *mask, const float *logs, const float *bias,
                                         const float *w, const float *dlogdet, const float *wb_end, float *dy_prev, float *dout_prev,
                                         float *dskip, float *partial, int B, int C, int H, int T, int n_split, int sigmoid_scale,
                                         int mask_dskip, glowtts_stream_t stream) {
    GLOWTTS_CHECK_ARG(dx_wn && wb_start && dy_next && y_prev && out_prev && mask && logs && bias && w && wb_end && dy_prev && dout_prev &&
                          dskip && partial,
                      "glowtts_flow_boundary_bwd: null pointer");
    GLOWTTS_CHECK_ARG(n_split == 2 || n_split == 4, "glowtts_flow_boundary_bwd: n_split=%d (2 or 4)", n_split);
    GLOWTTS_CHECK_ARG(B >= 0 && T >= 0 && C > 0 && C <= 192 && C / n_split <= 64 && H > 0 && H <= 192 && C % n_split == 0 && C % 2 == 0 &&
                          T % 4 == 0,
                      "glowtts_flow_boundary_bwd: shape (B=%d, C=%d, H=%d, T=%d): C, H <= 192, T %% 4 == 0", B, C, H, T);
    GLOWTTS_CHECK_ARG(aligned16(dx_wn) && aligned16(dy_next) && aligned16(y_prev) && aligned16(out_prev) && aligned16(dy_prev) &&
                          aligned16(dout_prev) && aligned16(dskip) && aligned16(wb_start) && aligned16(wb_end) && aligned16(mask),
                      "glowtts_flow_boundary_bwd: tensors must be 16-byte aligned");
    if ((long)B * T == 0) return 0;
    BoundaryBwdParams p{dx_wn, wb_start, dy_next, y_prev, out_prev, mask, logs, bias, w, dlogdet, wb_end,
                        dy_prev, dout_prev, dskip, partial, B, C, H, T, sigmoid_scale, mask_dskip, GLOWTTS_EXP_BITS(knob(K_BND_EXP))};
    constexpr int nt = 32;
    constexpr size_t lds = ((size_t)kBndMaxG * nt * kBndKP + 192 * (nt + 1) + nt) * sizeof(float);
    const dim3 grid(B * ((T + nt - 1) / nt));
    static LdsLimit lim[2];
    if (n_split == 4) {
        if (int rc_ = lim[0].ensure(reinterpret_cast<const void *>(&flow_boundary_bwd_kernel<4, nt>), lds, "glowtts_flow_boundary_bwd")) return rc_;
        hipLaunchKernelGGL((flow_boundary_bwd_kernel<4, nt>), grid, dim3(256), lds, (hipStream_t)stream, p);
    } else {
        if (int rc_ = lim[1].ensure(reinterpret_cast<const void *>(&flow_boundary_bwd_kernel<2, nt>), lds, "glowtts_flow_boundary_bwd")) return rc_;
        hipLaunchKernelGGL((flow_boundary_bwd_kernel<2, nt>), grid, dim3(256), lds, (hipStream_t)stream, p);
    }
    GLOWTTS_LAUNCH_CHECK("glowtts_flow_boundary_bwd");
}

extern "C" int glowtts_flow_boundary_bwd_reduce(const float *partial, const float *w_inv, const float *dlogdet, const float *x_len,
                                                float *dlogs, float *dbias, float *dw, int B, int C, int T, int n_split,
                                                glowtts_stream_t stream) {
    GLOWTTS_CHECK_ARG(partial && dlogs && dbias && dw, "glowtts_flow_boundary_bwd_reduce: null pointer");
    GLOWTTS_CHECK_ARG(n_split == 2 || n_split == 4, "glowtts_flow_boundary_bwd_reduce: n_split=%d (2 or 4)", n_split);
    GLOWTTS_CHECK_ARG(!dlogdet || (w_inv && x_len), "glowtts_flow_boundary_bwd_reduce: dlogdet given without w_inv / x_len");
    GLOWTTS_CHECK_ARG(B >= 0 && T >= 0 && C > 0 && C % n_split == 0, "glowtts_flow_boundary_bwd_reduce: bad shape");
    if ((long)B * T == 0) return 0;
    const int n_wg = B * ((T + 31) / 32);
    if (n_split == 4)
        hipLaunchKernelGGL(flow_boundary_bwd_reduce_kernel<4>, dim3(C / 4), dim3(256), 0, (hipStream_t)stream, partial, n_wg, w_inv, dlogdet,
                           x_len, dlogs, dbias, dw, B, C);
    else
        hipLaunchKernelGGL(flow_boundary_bwd_reduce_kernel<2>, dim3(C / 2), dim3(256), 0, (hipStream_t)stream, partial, n_wg, w_inv, dlogdet,
                           x_len, dlogs, dbias, dw, B, C);
    GLOWTTS_LAUNCH_CHECK("glowtts_flow_boundary_bwd_reduce");
}
