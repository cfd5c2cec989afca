// norm.hip — channel LayerNorm of the text encoder on the reference layout (B, C, T) (reference layers.py:10-28:
// mean / biased variance over the CHANNEL axis per (b, t) column, eps 1e-4, per-channel gamma / beta), with the
// residual add of the post-LN transformer layer fused in:  y = LN(x + res).
//
// PyTorch's layer_norm normalises trailing dims, so the eager path needs transpose -> contiguous copy -> layer_norm ->
// transpose (and the same again in backward: ~8 launches and 4 full copies per norm).  Here a workgroup owns 64
// consecutive frames of one utterance: lanes walk t (coalesced row segments), the 4 waves split the channels and meet
// in LDS once for the column statistics.  HBM-bound, tiny tensors (3.9 MB at config 2): what matters is launch count.
#include "common.hpp"

namespace glowtts {

// A workgroup owns 16 consecutive frames of one utterance: thread = (frame = tid & 15, channel slice = tid >> 4);
// each thread walks C/16 channels (short, unrolled loops: a first version with 64 frames x 4 slices per workgroup ran
// 48-iteration dependent-load loops on only 96 workgroups and took 115 us for a 3.9 MB tensor).
constexpr int kLnCols = 16, kLnSlices = 16;

__device__ __forceinline__ float ln_column_sum(float v, float (*sh)[kLnCols], int col, int slice) {
    sh[slice][col] = v;
    __syncthreads();
    float s = 0.f;
#pragma unroll
    for (int k = 0; k < kLnSlices; ++k) s += sh[k][col];
    __syncthreads();
    return s;
}

// The value that is normalised: v = x * mask_x[b, t] (+ res * keep * drop_scale).  mask_x folds the `x * x_mask` that opens
// every transformer layer (attentions.py:64), (drop, drop_scale) the dropout on the branch output that is added back
// (`self.drop(y)`, attentions.py:67,71): neither product is ever written to HBM.
struct LnIn {
    const float *x, *res, *mask_x;
    const unsigned char *drop;
    float drop_scale;
    __device__ __forceinline__ float at(long o, float m) const {
        float v = x[o] * m;
        if (res) {
            float r = res[o];
            if (drop) r = drop[o] ? r * drop_scale : 0.f;
            v += r;
        }
        return v;
    }
};

// forward: y = gamma * (v - mean) * rstd + beta;  stats[b][0][t] = mean, stats[b][1][t] = rstd
__global__ __launch_bounds__(256) void chan_layernorm_fwd_kernel(LnIn in, const float *__restrict__ gamma,
                                                                 const float *__restrict__ beta, float *__restrict__ y,
                                                                 float *__restrict__ stats, int C, int T, float eps) {
    __shared__ float sh[kLnSlices][kLnCols];
    const int col = threadIdx.x & 15, slice = threadIdx.x >> 4;
    const int b = blockIdx.y, t = blockIdx.x * kLnCols + col;
    const bool ok = t < T;
    const long base = (long)b * C * T + (ok ? t : 0);
    const float mx = in.mask_x ? in.mask_x[(long)b * T + (ok ? t : 0)] : 1.f;
    float a = 0.f;
#pragma unroll 4
    for (int c = slice; c < C; c += kLnSlices) a += in.at(base + (long)c * T, mx);
    const float mean = ln_column_sum(a, sh, col, slice) / C;
    float q = 0.f;                               // two-pass variance, as the reference: mean((v - mean)^2)
#pragma unroll 4
    for (int c = slice; c < C; c += kLnSlices) {
        const float v = in.at(base + (long)c * T, mx);
        q += (v - mean) * (v - mean);
    }
    const float var = ln_column_sum(q, sh, col, slice) / C;
    const float rstd = rsqrtf(var + eps);
    if (ok && slice == 0 && stats) {
        stats[((long)b * 2 + 0) * T + t] = mean;
        stats[((long)b * 2 + 1) * T + t] = rstd;
    }
    if (!ok) return;
#pragma unroll 4
    for (int c = slice; c < C; c += kLnSlices) {
        const float v = in.at(base + (long)c * T, mx);
        y[base + (long)c * T] = (v - mean) * rstd * gamma[c] + beta[c];
    }
}

// backward, input part: with xhat = (v - mean) rstd, g = dy * gamma:  dv = rstd * (g - mean_c(g) - xhat * mean_c(g * xhat))
// dx = dv * mask_x ; dres = dv * keep * drop_scale (written only when `dres` is given: without dropout dres == dx)
__global__ __launch_bounds__(256) void chan_layernorm_bwd_kernel(LnIn in, const float *__restrict__ gamma,
                                                                 const float *__restrict__ stats, const float *__restrict__ dy,
                                                                 float *__restrict__ dx, float *__restrict__ dres, int C, int T) {
    __shared__ float sh[kLnSlices][kLnCols];
    const int col = threadIdx.x & 15, slice = threadIdx.x >> 4;
    const int b = blockIdx.y, t = blockIdx.x * kLnCols + col;
    const bool ok = t < T;
    const long base = (long)b * C * T + (ok ? t : 0);
    const float mean = stats[((long)b * 2 + 0) * T + (ok ? t : 0)];
    const float rstd = stats[((long)b * 2 + 1) * T + (ok ? t : 0)];
    const float mx = in.mask_x ? in.mask_x[(long)b * T + (ok ? t : 0)] : 1.f;
    float a = 0.f, q = 0.f;
#pragma unroll 4
    for (int c = slice; c < C; c += kLnSlices) {
        const float v = in.at(base + (long)c * T, mx);
        const float g = dy[base + (long)c * T] * gamma[c];
        a += g;
        q += g * (v - mean) * rstd;
    }
    const float mg = ln_column_sum(a, sh, col, slice) / C;
    const float mgx = ln_column_sum(q, sh, col, slice) / C;
    if (!ok) return;
#pragma unroll 4
    for (int c = slice; c < C; c += kLnSlices) {
        const long o = base + (long)c * T;
        const float xh = (in.at(o, mx) - mean) * rstd;
        const float dv = rstd * (dy[o] * gamma[c] - mg - xh * mgx);
        dx[o] = dv * mx;
        if (dres) dres[o] = in.drop ? (in.drop[o] ? dv * in.drop_scale : 0.f) : dv;
    }
}

// backward, parameter part: dgamma[c] += sum_{b,t} dy * xhat ;  dbeta[c] += sum_{b,t} dy
// grid (C, slabs of utterances): rows (b, c, :) are contiguous in t, one block sum and one atomic pair per workgroup
__global__ __launch_bounds__(256) void chan_layernorm_bwd_param_kernel(LnIn in, const float *__restrict__ stats,
                                                                       const float *__restrict__ dy, float *__restrict__ dgamma,
                                                                       float *__restrict__ dbeta, int B, int C, int T, int nb) {
    __shared__ float red[4];
    const int c = blockIdx.x;
    const int b0 = blockIdx.y * nb, b1 = min(B, b0 + nb);
    float dg = 0.f, db = 0.f;
    const int items = (b1 - b0) * T;
#pragma unroll 2
    for (int i = threadIdx.x; i < items; i += 256) {
        const int b = b0 + i / T, t = i % T;
        const long o = ((long)b * C + c) * T + t;
        const float v = in.at(o, in.mask_x ? in.mask_x[(long)b * T + t] : 1.f);
        const float d = dy[o];
        dg += d * (v - stats[((long)b * 2 + 0) * T + t]) * stats[((long)b * 2 + 1) * T + t];
        db += d;
    }
    dg = block_sum_256(dg, red);
    db = block_sum_256(db, red);
    if (threadIdx.x == 0) {
        atomicAdd(dgamma + c, dg);
        atomicAdd(dbeta + c, db);
    }
}

}  // namespace glowtts

using namespace glowtts;

// `_ex` forms: mask_x (B, T) multiplies x, (drop (B, C, T) keep bytes, drop_scale) apply dropout to res — each may be NULL;
// the backward writes dx = dv * mask_x and, when dres != NULL, dres = dv * keep * drop_scale (else the gradient of res is dx)
extern "C" int glowtts_chan_layernorm_fwd_ex(const float *x, const float *res, const float *mask_x, const unsigned char *drop,
                                             float drop_scale, const float *gamma, const float *beta, float *y, float *stats,
                                             int B, int C, int T, float eps, glowtts_stream_t stream) {
    GLOWTTS_CHECK_ARG(x && gamma && beta && y, "glowtts_chan_layernorm_fwd: null pointer");
    GLOWTTS_CHECK_ARG(B >= 0 && C > 0 && T >= 0 && (!drop || res), "glowtts_chan_layernorm_fwd: bad shape");
    if ((long)B * T == 0) return 0;
    LnIn in{x, res, mask_x, drop, drop_scale};
    hipLaunchKernelGGL(chan_layernorm_fwd_kernel, dim3((T + kLnCols - 1) / kLnCols, B), dim3(256), 0, (hipStream_t)stream, in,
                       gamma, beta, y, stats, C, T, eps);
    GLOWTTS_LAUNCH_CHECK("glowtts_chan_layernorm_fwd");
}

extern "C" int glowtts_chan_layernorm_fwd(const float *x, const float *res, const float *gamma, const float *beta, float *y,
                                          float *stats, int B, int C, int T, float eps, glowtts_stream_t stream) {
    return glowtts_chan_layernorm_fwd_ex(x, res, nullptr, nullptr, 1.f, gamma, beta, y, stats, B, C, T, eps, stream);
}

extern "C" int glowtts_chan_layernorm_bwd_ex(const float *x, const float *res, const float *mask_x, const unsigned char *drop,
                                             float drop_scale, const float *gamma, const float *stats, const float *dy,
                                             float *dx, float *dres, float *dgamma, float *dbeta, int B, int C, int T,
                                             glowtts_stream_t stream) {
    GLOWTTS_CHECK_ARG(x && gamma && stats && dy && dx && dgamma && dbeta, "glowtts_chan_layernorm_bwd: null pointer");
    GLOWTTS_CHECK_ARG(B >= 0 && C > 0 && T >= 0 && (!drop || (res && dres)), "glowtts_chan_layernorm_bwd: bad shape");
    if ((long)B * T == 0) return 0;
    LnIn in{x, res, mask_x, drop, drop_scale};
    hipLaunchKernelGGL(chan_layernorm_bwd_kernel, dim3((T + kLnCols - 1) / kLnCols, B), dim3(256), 0, (hipStream_t)stream, in,
                       gamma, stats, dy, dx, dres, C, T);
    int slabs = (1024 + C - 1) / C;
    if (slabs > B) slabs = B;
    const int nb = (B + slabs - 1) / slabs;
    hipLaunchKernelGGL(chan_layernorm_bwd_param_kernel, dim3(C, (B + nb - 1) / nb), dim3(256), 0, (hipStream_t)stream, in, stats,
                       dy, dgamma, dbeta, B, C, T, nb);
    GLOWTTS_LAUNCH_CHECK("glowtts_chan_layernorm_bwd");
}

extern "C" int glowtts_chan_layernorm_bwd(const float *x, const float *res, const float *gamma, const float *stats,
                                          const float *dy, float *dx, float *dgamma, float *dbeta, int B, int C, int T,
                                          glowtts_stream_t stream) {
    return glowtts_chan_layernorm_bwd_ex(x, res, nullptr, nullptr, 1.f, gamma, stats, dy, dx, nullptr, dgamma, dbeta, B, C, T,
                                         stream);
}
