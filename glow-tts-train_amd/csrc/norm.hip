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

// forward: y = gamma * (v - mean) * rstd + beta, v = x (+ res);  stats[b][0][t] = mean, stats[b][1][t] = rstd
__global__ __launch_bounds__(256) void chan_layernorm_fwd_kernel(const float *__restrict__ x, const float *__restrict__ res,
                                                                 const float *__restrict__ gamma, const float *__restrict__ beta,
                                                                 float *__restrict__ y, float *__restrict__ stats, int C,
                                                                 int T, float eps) {
    __shared__ float s1[4][64], s2[4][64];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int b = blockIdx.y, t = blockIdx.x * 64 + lane;
    const bool ok = t < T;
    const long base = (long)b * C * T + t;
    float a = 0.f;
#pragma unroll 8
    for (int c = wave; c < C; c += 4) {
        float v = 0.f;
        if (ok) { v = x[base + (long)c * T]; if (res) v += res[base + (long)c * T]; }
        a += v;
    }
    s1[wave][lane] = a;
    __syncthreads();
    const float mean = (s1[0][lane] + s1[1][lane] + s1[2][lane] + s1[3][lane]) / C;
    float q = 0.f;                               // two-pass variance, as the reference: mean((v - mean)^2)
#pragma unroll 8
    for (int c = wave; c < C; c += 4) {
        float v = 0.f;
        if (ok) { v = x[base + (long)c * T]; if (res) v += res[base + (long)c * T]; }
        q += (v - mean) * (v - mean);
    }
    s2[wave][lane] = q;
    __syncthreads();
    const float var = (s2[0][lane] + s2[1][lane] + s2[2][lane] + s2[3][lane]) / C;
    const float rstd = rsqrtf(var + eps);
    if (ok && wave == 0 && stats) {
        stats[((long)b * 2 + 0) * T + t] = mean;
        stats[((long)b * 2 + 1) * T + t] = rstd;
    }
#pragma unroll 8
    for (int c = wave; c < C; c += 4) {
        if (ok) {
            float v = x[base + (long)c * T];
            if (res) v += res[base + (long)c * T];
            y[base + (long)c * T] = (v - mean) * rstd * gamma[c] + beta[c];
        }
    }
}

// backward: with xhat = (v - mean) rstd, g = dy * gamma:  dv = rstd * (g - mean_c(g) - xhat * mean_c(g * xhat))
//           dgamma[c] += sum_{b,t} dy * xhat ;  dbeta[c] += sum_{b,t} dy          (dv is the gradient of x AND of res)
__global__ __launch_bounds__(256) void chan_layernorm_bwd_kernel(const float *__restrict__ x, const float *__restrict__ res,
                                                                 const float *__restrict__ gamma, const float *__restrict__ stats,
                                                                 const float *__restrict__ dy, float *__restrict__ dx,
                                                                 float *__restrict__ dgamma, float *__restrict__ dbeta, int C,
                                                                 int T) {
    __shared__ float s1[4][64], s2[4][64];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int b = blockIdx.y, t = blockIdx.x * 64 + lane;
    const bool ok = t < T;
    const long base = (long)b * C * T + t;
    const float mean = ok ? stats[((long)b * 2 + 0) * T + t] : 0.f;
    const float rstd = ok ? stats[((long)b * 2 + 1) * T + t] : 0.f;
    float a = 0.f, q = 0.f;
#pragma unroll 8
    for (int c = wave; c < C; c += 4) {
        if (ok) {
            float v = x[base + (long)c * T];
            if (res) v += res[base + (long)c * T];
            const float g = dy[base + (long)c * T] * gamma[c];
            a += g;
            q += g * (v - mean) * rstd;
        }
    }
    s1[wave][lane] = a;
    s2[wave][lane] = q;
    __syncthreads();
    const float mg = (s1[0][lane] + s1[1][lane] + s1[2][lane] + s1[3][lane]) / C;
    const float mgx = (s2[0][lane] + s2[1][lane] + s2[2][lane] + s2[3][lane]) / C;
#pragma unroll 4
    for (int c = wave; c < C; c += 4) {
        float dg = 0.f, db = 0.f;
        if (ok) {
            float v = x[base + (long)c * T];
            if (res) v += res[base + (long)c * T];
            const float xh = (v - mean) * rstd;
            const float d = dy[base + (long)c * T];
            dx[base + (long)c * T] = rstd * (d * gamma[c] - mg - xh * mgx);
            dg = d * xh;
            db = d;
        }
        dg = wave_sum(dg);
        db = wave_sum(db);
        if (lane == 0) {
            atomicAdd(dgamma + c, dg);
            atomicAdd(dbeta + c, db);
        }
    }
}

}  // namespace glowtts

using namespace glowtts;

extern "C" int glowtts_chan_layernorm_fwd(const float *x, const float *res, const float *gamma, const float *beta, float *y,
                                          float *stats, int B, int C, int T, float eps, glowtts_stream_t stream) {
    GLOWTTS_CHECK_ARG(x && gamma && beta && y, "glowtts_chan_layernorm_fwd: null pointer");
    GLOWTTS_CHECK_ARG(B >= 0 && C > 0 && T >= 0, "glowtts_chan_layernorm_fwd: bad shape");
    if ((long)B * T == 0) return 0;
    hipLaunchKernelGGL(chan_layernorm_fwd_kernel, dim3((T + 63) / 64, B), dim3(256), 0, (hipStream_t)stream, x, res, gamma, beta, y,
                       stats, C, T, eps);
    GLOWTTS_LAUNCH_CHECK("glowtts_chan_layernorm_fwd");
}

extern "C" int glowtts_chan_layernorm_bwd(const float *x, const float *res, const float *gamma, const float *stats,
                                          const float *dy, float *dx, float *dgamma, float *dbeta, int B, int C, int T,
                                          glowtts_stream_t stream) {
    GLOWTTS_CHECK_ARG(x && gamma && stats && dy && dx && dgamma && dbeta, "glowtts_chan_layernorm_bwd: null pointer");
    GLOWTTS_CHECK_ARG(B >= 0 && C > 0 && T >= 0, "glowtts_chan_layernorm_bwd: bad shape");
    if ((long)B * T == 0) return 0;
    hipLaunchKernelGGL(chan_layernorm_bwd_kernel, dim3((T + 63) / 64, B), dim3(256), 0, (hipStream_t)stream, x, res, gamma, stats,
                       dy, dx, dgamma, dbeta, C, T);
    GLOWTTS_LAUNCH_CHECK("glowtts_chan_layernorm_bwd");
}
