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
    const float *xd = Xs + lrow * kBndKP + lk * 4;
    auto step = [&](int g, const f32x4 (&aw)[kBndRT]) {
        f32x4 bv[NC];
#pragma unroll
        for (int c = 0; c < NC; ++c) bv[c] = *reinterpret_cast<const f32x4 *>(xd + (g * NT + c * 16) * kBndKP);
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int r = 0; r < kBndRT; ++r)
#pragma unroll
                for (int c = 0; c < NC; ++c)
                    acc[r][c] = __builtin_amdgcn_mfma_f32_16x16x4f32(aw[r][j], bv[c][j], acc[r][c], 0, 0, 0);
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
    bnd_gemm<NT>(p.wp_end, C, (p.exp & 1) ? 0 : GH, Xs, wave, lrow, lk, aw, acc);
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
        if (i >= G * NQ || t0 + q * 4 >= T || (p.exp & 4)) continue;
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
    bnd_gemm<NT>(p.wp_start, H, (p.exp & 2) ? 0 : GS, Ys, wave, lrow, lk, aw, acc);
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
                     out, y, h0, logdet_prev, logdet, B, C, H, T, sigmoid_scale, env_knob("GLOWTTS_BND_EXP", 0)};
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
