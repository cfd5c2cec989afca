// align.hip — the alignment-side glue of FlowGenerator.forward (reference models.py:361-393) as three small kernels:
//   align_logp    : log N(z_t'; x_m_t, exp(x_logs_t)) for every (token, frame) pair — the lattice the alignment search runs
//                   on.  The reference forms it from two bmm's, two channel sums and ~10 elementwise ops (models.py:362-376);
//                   here it is ONE contraction over k = 2C per utterance on the fp32 MFMA:
//                       logp[x, y] = rowconst[x] + sum_c s[c,x] (-z[c,y]^2 / 2) + sum_c (x_m s)[c,x] z[c,y],   s = exp(-2 x_logs)
//                       rowconst[x] = sum_c (-log(2 pi)/2 - x_logs[c,x] - x_m[c,x]^2 s[c,x] / 2)
//   align_expand  : z_m = attn^T x_m (models.py:383-392).  attn is a hard monotonic 0/1 path, so the bmm with its one-hot
//                   rows is a GATHER: every frame copies the statistics of its token (tok[b, y], written by the search
//                   kernel); the backward is a segment sum over each token's span [first[x], first[x+1]) — no atomics.
#include "common.hpp"

namespace glowtts {

typedef float f32x4_a __attribute__((ext_vector_type(4)));

constexpr int kLogpKC = 40;                  // channels per LDS chunk (k = 2 * 40 per chunk)
constexpr int kLogpP = 68;                   // LDS pitch of a k-row: 64 columns + 4

// grid (ceil(Ty/64), ceil(Tx/64), B); workgroup = 64 tokens x 64 frames, wave w owns tokens 16 w .. 16 w + 15
__global__ __launch_bounds__(256) void align_logp_kernel(const float *__restrict__ x_m, const float *__restrict__ x_logs,
                                                         const float *__restrict__ z, float *__restrict__ logp, int C,
                                                         int Tx, int Ty) {
    __shared__ float As[2 * kLogpKC * kLogpP];      // [k][token]: k < KC: s ; k >= KC: x_m s
    __shared__ float Bs[2 * kLogpKC * kLogpP];      // [k][frame]: k < KC: -z^2/2 ; k >= KC: z
    __shared__ float rowc[64];
    const int b = blockIdx.z, x0 = blockIdx.y * 64, y0 = blockIdx.x * 64;
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, lrow = lane & 15, lk = lane >> 4;
    const float *xm = x_m + (long)b * C * Tx;
    const float *xl = x_logs ? x_logs + (long)b * C * Tx : nullptr;
    const float *zb = z + (long)b * C * Ty;
    if (tid < 64) rowc[tid] = 0.f;
    f32x4_a acc[4];
#pragma unroll
    for (int c = 0; c < 4; ++c) acc[c] = f32x4_a{0.f, 0.f, 0.f, 0.f};
    const int col = tid & 63, crow = tid >> 6;      // staging: thread = (column, channel crow + 4 i)
    float rc = 0.f;
    for (int c0 = 0; c0 < C; c0 += kLogpKC) {
        __syncthreads();
#pragma unroll 2
        for (int i = 0; i < kLogpKC / 4; ++i) {
            const int cl = crow + 4 * i, c = c0 + cl;
            float s = 0.f, ms = 0.f, q = 0.f, zz = 0.f;
            if (c < C) {
                if (x0 + col < Tx) {
                    const float m = xm[(long)c * Tx + x0 + col];
                    const float l = xl ? xl[(long)c * Tx + x0 + col] : 0.f;
                    s = xl ? expf(-2.0f * l) : 1.0f;
                    ms = m * s;
                    rc += -0.91893853320467274178f - l - 0.5f * m * ms;        // -log(2 pi)/2 - logs - m^2 s / 2
                }
                if (y0 + col < Ty) {
                    zz = zb[(long)c * Ty + y0 + col];
                    q = -0.5f * zz * zz;
                }
            }
            As[cl * kLogpP + col] = s;
            As[(kLogpKC + cl) * kLogpP + col] = ms;
            Bs[cl * kLogpP + col] = q;
            Bs[(kLogpKC + cl) * kLogpP + col] = zz;
        }
        __syncthreads();
#pragma unroll 4
        for (int k4 = 0; k4 < 2 * kLogpKC / 4; ++k4) {
            const float a = As[(k4 * 4 + lk) * kLogpP + wave * 16 + lrow];
#pragma unroll
            for (int c = 0; c < 4; ++c)
                acc[c] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, Bs[(k4 * 4 + lk) * kLogpP + c * 16 + lrow], acc[c], 0, 0, 0);
        }
    }
    atomicAdd(rowc + col, rc);                       // four threads per token column (LDS atomic)
    __syncthreads();
    float *out = logp + (long)b * Tx * Ty;
#pragma unroll
    for (int c = 0; c < 4; ++c) {
        const int y = y0 + c * 16 + lrow;
#pragma unroll
        for (int reg = 0; reg < 4; ++reg) {
            const int xr = wave * 16 + lk * 4 + reg;
            if (x0 + xr < Tx && y < Ty) out[(long)(x0 + xr) * Ty + y] = acc[c][reg] + rowc[xr];
        }
    }
}

// out[b, d, y] = stats[b, d, tok[b, y]]  (0 where tok < 0: frames past the utterance)
__global__ __launch_bounds__(256) void align_expand_fwd_kernel(const float *__restrict__ stats, const int *__restrict__ tok,
                                                               float *__restrict__ out, int D, int Tx, int Ty, long n) {
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const int y = (int)(i % Ty);
    const long bd = i / Ty;
    const int b = (int)(bd / D);
    const int t = tok[(long)b * Ty + y];
    out[i] = t >= 0 ? stats[bd * Tx + t] : 0.f;
}

// dstats[b, d, x] = sum of dout[b, d, y] over the token's span y in [first[b, x], first[b, x + 1])
__global__ __launch_bounds__(256) void align_expand_bwd_kernel(const float *__restrict__ dout, const int *__restrict__ first,
                                                               float *__restrict__ dstats, int D, int Tx, int Ty, long n) {
    // A thread sums the first kShort frames of its token's span itself; what a LONG span has beyond that (a pause, or a degenerate
    // alignment early in training: one token holding hundreds of frames) is summed by the whole wave, 64 frames at a time — a
    // single thread walking 600 frames made this kernel 60 us at config 2 with random weights, against 9 us for its forward.
    constexpr int kShort = 16;
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    const bool live = i < n;
    const long ii = live ? i : n - 1;
    const int x = (int)(ii % Tx);
    const long bd = ii / Tx;
    const int b = (int)(bd / D);
    const int lo = first[(long)b * (Tx + 1) + x], hi = live ? first[(long)b * (Tx + 1) + x + 1] : lo;
    const float *src = dout + bd * Ty;
    float s = 0.f;
    const int mid = min(hi, lo + kShort);
    for (int y = lo; y < mid; ++y) s += src[y];
    const int lane = threadIdx.x & 63;
    unsigned long long todo = __ballot(hi > mid);
    while (todo != 0ull) {                              // uniform: every lane of the wave walks the same list of long spans
        const int owner = __ffsll((long long)todo) - 1;
        todo &= todo - 1;
        const long obd = __shfl(bd, owner, 64);
        const int olo = __shfl(mid, owner, 64), ohi = __shfl(hi, owner, 64);
        const float *osrc = dout + obd * Ty;
        float part = 0.f;
        for (int y = olo + lane; y < ohi; y += 64) part += osrc[y];
        part = wave_sum(part);
        if (lane == owner) s += part;
    }
    if (live) dstats[i] = s;
}

}  // namespace glowtts

using namespace glowtts;

extern "C" int glowtts_align_logp(const float *x_m, const float *x_logs, const float *z, float *logp, int B, int C, int Tx,
                                  int Ty, glowtts_stream_t stream) {
    GLOWTTS_CHECK_ARG(x_m && z && logp, "glowtts_align_logp: null pointer");
    GLOWTTS_CHECK_ARG(B >= 0 && C > 0 && Tx >= 0 && Ty >= 0, "glowtts_align_logp: bad shape");
    if ((long)B * Tx * Ty == 0) return 0;
    GLOWTTS_CHECK_ARG(B <= 65535 && (Tx + 63) / 64 <= 65535, "glowtts_align_logp: batch / text length beyond the grid limits");
    dim3 grid((Ty + 63) / 64, (Tx + 63) / 64, B);
    hipLaunchKernelGGL(align_logp_kernel, grid, dim3(256), 0, (hipStream_t)stream, x_m, x_logs, z, logp, C, Tx, Ty);
    GLOWTTS_LAUNCH_CHECK("glowtts_align_logp");
}

extern "C" int glowtts_align_expand_fwd(const float *stats, const int32_t *tok, float *out, int B, int D, int Tx, int Ty,
                                        glowtts_stream_t stream) {
    GLOWTTS_CHECK_ARG(stats && tok && out, "glowtts_align_expand_fwd: null pointer");
    GLOWTTS_CHECK_ARG(B >= 0 && D > 0 && Tx > 0 && Ty >= 0, "glowtts_align_expand_fwd: bad shape");
    const long n = (long)B * D * Ty;
    if (n == 0) return 0;
    hipLaunchKernelGGL(align_expand_fwd_kernel, dim3(cdiv(n, 256)), dim3(256), 0, (hipStream_t)stream, stats, tok, out, D, Tx, Ty, n);
    GLOWTTS_LAUNCH_CHECK("glowtts_align_expand_fwd");
}

extern "C" int glowtts_align_expand_bwd(const float *dout, const int32_t *first, float *dstats, int B, int D, int Tx, int Ty,
                                        glowtts_stream_t stream) {
    GLOWTTS_CHECK_ARG(dout && first && dstats, "glowtts_align_expand_bwd: null pointer");
    GLOWTTS_CHECK_ARG(B >= 0 && D > 0 && Tx >= 0 && Ty >= 0, "glowtts_align_expand_bwd: bad shape");
    const long n = (long)B * D * Tx;
    if (n == 0) return 0;
    hipLaunchKernelGGL(align_expand_bwd_kernel, dim3(cdiv(n, 256)), dim3(256), 0, (hipStream_t)stream, dout, first, dstats, D, Tx, Ty, n);
    GLOWTTS_LAUNCH_CHECK("glowtts_align_expand_bwd");
}
