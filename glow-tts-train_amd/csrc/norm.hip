// norm.hip — channel LayerNorm of the text encoder on the reference layout (B, C, T) (reference layers.py:10-28:
// mean / biased variance over the CHANNEL axis per (b, t) column, eps 1e-4, per-channel gamma / beta), with the
// residual add of the post-LN transformer layer fused in:  y = LN(x + res).
//
// PyTorch's layer_norm normalises trailing dims, so the eager path needs transpose -> contiguous copy -> layer_norm ->
// transpose (and the same again in backward: ~8 launches and 4 full copies per norm).  Here a workgroup owns 32
// consecutive frames of one utterance: lanes walk t (coalesced row segments), 8 channel slices keep their part of the
// column in registers and meet in LDS once per statistic.  HBM-bound, tiny tensors (3.9 MB at config 2).
#include "common.hpp"

namespace glowtts {

// A workgroup owns 32 consecutive frames of one utterance: thread = (frame = tid & 31, channel slice = tid >> 5), each thread
// keeps its CPT = ceil(C / 8) channels of its frame IN REGISTERS, so the tensor is read ONCE per pass (round 2's kernels walked
// the column three times forward and twice backward: 18.5 / 31.5 us for a 3.9 MB tensor = 0.5 TB/s) and the mean / variance /
// backward sums are exact two-pass quantities over registers.  Lanes 0-31 of a wave walk 32 consecutive frames (128-byte row
// segments), the two halves of a wave are neighbouring channels.  The slices meet in LDS once per statistic.
constexpr int kLnCols = 32, kLnSlices = 8;

__device__ __forceinline__ float ln_column_sum(float v, float (*sh)[kLnCols], int col, int slice) {
    sh[slice][col] = v;
    __syncthreads();
    float s = 0.f;
#pragma unroll
    for (int k = 0; k < kLnSlices; ++k) s += sh[k][col];
    __syncthreads();
    return s;
}

// The value that is normalised: v = f(x) * mask_x[b, t] (+ res * keep * drop_scale), f = ReLU when relu_in (the duration
// predictor's conv -> ReLU -> LayerNorm, models.py:45-46: the ReLU never gets a launch or a tensor of its own) else identity.
// mask_x folds the `x * x_mask` that opens every transformer layer (attentions.py:64), (drop, drop_scale) the dropout on the
// branch output that is added back (`self.drop(y)`, attentions.py:67,71): none of these products is ever written to HBM.
struct LnIn {
    const float *x, *res, *mask_x;
    const unsigned char *drop;
    float drop_scale;
    int relu_in;
    __device__ __forceinline__ float at(long o, float m) const {
        float xv = x[o];
        if (relu_in) xv = fmaxf(xv, 0.f);
        float v = xv * m;
        if (res) {
            float r = res[o];
            if (drop) r = drop[o] ? r * drop_scale : 0.f;
            v += r;
        }
        return v;
    }
};

// What follows the normalisation in the pre-net (layers.py:73-80: LayerNorm -> ReLU -> Dropout) and the duration predictor
// (LayerNorm -> Dropout): y = drop(relu(LN(v))), applied as the value is stored; the backward gates dy the same way
// (ReLU through the sign of the stored y: y > 0 <=> the ReLU passed and the dropout kept).
struct LnOut {
    const unsigned char *odrop;   // (B, C, T) keep bytes or NULL
    float oscale;
    int relu_out;
    __device__ __forceinline__ float fwd(float v, long o) const {
        if (relu_out) v = fmaxf(v, 0.f);
        if (odrop) v = odrop[o] ? v * oscale : 0.f;
        return v;
    }
    __device__ __forceinline__ float bwd(float dy, float y, long o) const {
        if (odrop) dy = odrop[o] ? dy * oscale : 0.f;
        if (relu_out) dy = y > 0.f ? dy : 0.f;
        return dy;
    }
};

// All of a thread's global loads are issued back to back, BEFORE the first dependent instruction (clamped channel index: no
// exec masking; the optional operands behind ONE uniform branch each): written as `v += res[o]` inside the channel loop the
// compiler waited for every element before it issued the next load — 24 serial round trips per thread (~20 us per launch).
template <int CPT>
struct LnCols {
    float xr[CPT], rr[CPT];
    unsigned char dk[CPT];
    __device__ __forceinline__ void load(const LnIn &in, long base, int slice, int C, int T) {
#pragma unroll
        for (int i = 0; i < CPT; ++i) xr[i] = in.x[base + (long)min(slice + kLnSlices * i, C - 1) * T];
        if (in.res) {
#pragma unroll
            for (int i = 0; i < CPT; ++i) rr[i] = in.res[base + (long)min(slice + kLnSlices * i, C - 1) * T];
            if (in.drop) {
#pragma unroll
                for (int i = 0; i < CPT; ++i) dk[i] = in.drop[base + (long)min(slice + kLnSlices * i, C - 1) * T];
            }
        }
    }
    __device__ __forceinline__ float value(const LnIn &in, int i, float m) const {      // what LnIn::at computes, from registers
        float xv = xr[i];
        if (in.relu_in) xv = fmaxf(xv, 0.f);
        float v = xv * m;
        if (in.res) v += in.drop ? (dk[i] ? rr[i] * in.drop_scale : 0.f) : rr[i];
        return v;
    }
};

// forward: y = post(gamma * (v - mean) * rstd + beta);  stats[b][0][t] = mean, stats[b][1][t] = rstd
template <int CPT>
__global__ __launch_bounds__(256) void chan_layernorm_fwd_kernel(LnIn in, LnOut post, const float *__restrict__ gamma,
                                                                 const float *__restrict__ beta, float *__restrict__ y,
                                                                 float *__restrict__ stats, int C, int T, float eps) {
    __shared__ float sh[kLnSlices][kLnCols];
    const int col = threadIdx.x & 31, slice = threadIdx.x >> 5;
    const int b = blockIdx.y, t = blockIdx.x * kLnCols + col;
    const bool ok = t < T;
    const long base = (long)b * C * T + (ok ? t : 0);
    const float mx = in.mask_x ? in.mask_x[(long)b * T + (ok ? t : 0)] : 1.f;
    LnCols<CPT> cols;
    cols.load(in, base, slice, C, T);
    unsigned char ok_[CPT];
    float gm[CPT], bt[CPT];
#pragma unroll
    for (int i = 0; i < CPT; ++i) {
        const int c = min(slice + kLnSlices * i, C - 1);
        gm[i] = gamma[c];
        bt[i] = beta[c];
    }
    if (post.odrop) {
#pragma unroll
        for (int i = 0; i < CPT; ++i) ok_[i] = post.odrop[base + (long)min(slice + kLnSlices * i, C - 1) * T];
    }
    float v[CPT];
    float a = 0.f;
#pragma unroll
    for (int i = 0; i < CPT; ++i) {
        v[i] = (slice + kLnSlices * i < C) ? cols.value(in, i, mx) : 0.f;
        a += v[i];
    }
    const float mean = ln_column_sum(a, sh, col, slice) / C;
    float q = 0.f;                               // two-pass variance, as the reference: mean((v - mean)^2)
#pragma unroll
    for (int i = 0; i < CPT; ++i)
        if (slice + kLnSlices * i < C) q += (v[i] - mean) * (v[i] - mean);
    const float var = ln_column_sum(q, sh, col, slice) / C;
    const float rstd = rsqrtf(var + eps);
    if (ok && slice == 0 && stats) {
        stats[((long)b * 2 + 0) * T + t] = mean;
        stats[((long)b * 2 + 1) * T + t] = rstd;
    }
    if (!ok) return;
#pragma unroll
    for (int i = 0; i < CPT; ++i) {
        const int c = slice + kLnSlices * i;
        if (c < C) {
            float val = (v[i] - mean) * rstd * gm[i] + bt[i];
            if (post.relu_out) val = fmaxf(val, 0.f);
            if (post.odrop) val = ok_[i] ? val * post.oscale : 0.f;
            y[base + (long)c * T] = val;
        }
    }
}

// backward, input part: with xhat = (v - mean) rstd, g = dy' * gamma (dy' = dy through the output gates):
//   dv = rstd * (g - mean_c(g) - xhat * mean_c(g * xhat));  dx = dv * mask_x (* [x > 0] when relu_in);
//   dres = dv * keep * drop_scale (written only when `dres` is given: without dropout dres == dx)
template <int CPT>
__global__ __launch_bounds__(256) void chan_layernorm_bwd_kernel(LnIn in, LnOut post, const float *__restrict__ gamma,
                                                                 const float *__restrict__ stats, const float *__restrict__ yout,
                                                                 const float *__restrict__ dy, float *__restrict__ dx,
                                                                 float *__restrict__ dres, int C, int T) {
    __shared__ float sh[kLnSlices][kLnCols];
    const int col = threadIdx.x & 31, slice = threadIdx.x >> 5;
    const int b = blockIdx.y, t = blockIdx.x * kLnCols + col;
    const bool ok = t < T;
    const long base = (long)b * C * T + (ok ? t : 0);
    const float mean = stats[((long)b * 2 + 0) * T + (ok ? t : 0)];
    const float rstd = stats[((long)b * 2 + 1) * T + (ok ? t : 0)];
    const float mx = in.mask_x ? in.mask_x[(long)b * T + (ok ? t : 0)] : 1.f;
    LnCols<CPT> cols;
    cols.load(in, base, slice, C, T);
    float dyr[CPT], yr[CPT], gm[CPT];
    unsigned char ok_[CPT];
#pragma unroll
    for (int i = 0; i < CPT; ++i) {
        const int c = min(slice + kLnSlices * i, C - 1);
        dyr[i] = dy[base + (long)c * T];
        gm[i] = gamma[c];
    }
    if (post.relu_out) {
#pragma unroll
        for (int i = 0; i < CPT; ++i) yr[i] = yout[base + (long)min(slice + kLnSlices * i, C - 1) * T];
    }
    if (post.odrop) {
#pragma unroll
        for (int i = 0; i < CPT; ++i) ok_[i] = post.odrop[base + (long)min(slice + kLnSlices * i, C - 1) * T];
    }
    float xh[CPT], g[CPT];
    float a = 0.f, q = 0.f;
#pragma unroll
    for (int i = 0; i < CPT; ++i) {
        xh[i] = 0.f;
        g[i] = 0.f;
        if (slice + kLnSlices * i < C) {
            xh[i] = (cols.value(in, i, mx) - mean) * rstd;
            float d = dyr[i];
            if (post.odrop) d = ok_[i] ? d * post.oscale : 0.f;
            if (post.relu_out) d = yr[i] > 0.f ? d : 0.f;
            g[i] = d * gm[i];
            a += g[i];
            q += g[i] * xh[i];
        }
    }
    const float mg = ln_column_sum(a, sh, col, slice) / C;
    const float mgx = ln_column_sum(q, sh, col, slice) / C;
    if (!ok) return;
#pragma unroll
    for (int i = 0; i < CPT; ++i) {
        const int c = slice + kLnSlices * i;
        if (c < C) {
            const long o = base + (long)c * T;
            const float dv = rstd * (g[i] - mg - xh[i] * mgx);
            float d = dv * mx;
            if (in.relu_in) d = cols.xr[i] > 0.f ? d : 0.f;
            dx[o] = d;
            if (dres) dres[o] = in.drop ? (cols.dk[i] ? dv * in.drop_scale : 0.f) : dv;
        }
    }
}

// backward, parameter part: dgamma[c] += sum_{b,t} dy * xhat ;  dbeta[c] += sum_{b,t} dy
// grid (C, slabs of utterances): rows (b, c, :) are contiguous in t, one block sum and one atomic pair per workgroup
__global__ __launch_bounds__(256) void chan_layernorm_bwd_param_kernel(LnIn in, LnOut post, const float *__restrict__ stats,
                                                                       const float *__restrict__ yout,
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
        const float d = post.bwd(dy[o], post.relu_out ? yout[o] : 0.f, o);
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

// Channel counts the register-resident kernels are instantiated for (CPT = ceil(C / 8) channels per thread)
#define GLOWTTS_LN_DISPATCH(KERNEL, GRID, ...)                                                              \
    do {                                                                                                    \
        if (C <= 8 * 8)       hipLaunchKernelGGL((KERNEL<8>), GRID, dim3(256), 0, s, __VA_ARGS__);          \
        else if (C <= 8 * 24) hipLaunchKernelGGL((KERNEL<24>), GRID, dim3(256), 0, s, __VA_ARGS__);         \
        else if (C <= 8 * 32) hipLaunchKernelGGL((KERNEL<32>), GRID, dim3(256), 0, s, __VA_ARGS__);         \
        else                  hipLaunchKernelGGL((KERNEL<96>), GRID, dim3(256), 0, s, __VA_ARGS__);         \
    } while (0)

// `_act` forms (everything the `_ex` forms do, plus): relu_in — the value normalised is relu(x) * mask_x (+ res ...), dx is gated by
// x > 0; relu_out / (odrop, oscale) — y = dropout(relu(LN(v))), dy gated the same way in the backward (which then needs y).
// Reference: layers.py:73-80 (pre-net: LayerNorm -> ReLU -> Dropout), models.py:44-50 (duration predictor: ReLU -> LayerNorm -> Dropout).
extern "C" int glowtts_chan_layernorm_fwd_act(const float *x, const float *res, const float *mask_x, const unsigned char *drop,
                                              float drop_scale, const float *gamma, const float *beta, float *y, float *stats,
                                              int relu_in, int relu_out, const unsigned char *odrop, float oscale, int B, int C,
                                              int T, float eps, glowtts_stream_t stream) {
    GLOWTTS_CHECK_ARG(x && gamma && beta && y, "glowtts_chan_layernorm_fwd: null pointer");
    GLOWTTS_CHECK_ARG(B >= 0 && C > 0 && C <= 8 * 96 && T >= 0 && (!drop || res), "glowtts_chan_layernorm_fwd: bad shape (C <= 768)");
    if ((long)B * T == 0) return 0;
    LnIn in{x, res, mask_x, drop, drop_scale, relu_in};
    LnOut post{odrop, oscale, relu_out};
    hipStream_t s = (hipStream_t)stream;
    const dim3 grid((T + kLnCols - 1) / kLnCols, B);
    GLOWTTS_LN_DISPATCH(chan_layernorm_fwd_kernel, grid, in, post, gamma, beta, y, stats, C, T, eps);
    GLOWTTS_LAUNCH_CHECK("glowtts_chan_layernorm_fwd");
}

extern "C" int glowtts_chan_layernorm_bwd_act(const float *x, const float *res, const float *mask_x, const unsigned char *drop,
                                              float drop_scale, const float *gamma, const float *stats, const float *y,
                                              const float *dy, int relu_in, int relu_out, const unsigned char *odrop, float oscale,
                                              float *dx, float *dres, float *dgamma, float *dbeta, int B, int C, int T,
                                              glowtts_stream_t stream) {
    GLOWTTS_CHECK_ARG(x && gamma && stats && dy && dx && dgamma && dbeta, "glowtts_chan_layernorm_bwd: null pointer");
    GLOWTTS_CHECK_ARG(B >= 0 && C > 0 && C <= 8 * 96 && T >= 0 && (!drop || (res && dres)) && (!relu_out || y),
                      "glowtts_chan_layernorm_bwd: bad shape (C <= 768; relu_out needs y)");
    if ((long)B * T == 0) return 0;
    LnIn in{x, res, mask_x, drop, drop_scale, relu_in};
    LnOut post{odrop, oscale, relu_out};
    hipStream_t s = (hipStream_t)stream;
    const dim3 grid((T + kLnCols - 1) / kLnCols, B);
    GLOWTTS_LN_DISPATCH(chan_layernorm_bwd_kernel, grid, in, post, gamma, stats, y, dy, dx, dres, C, T);
    int slabs = (1024 + C - 1) / C;
    if (slabs > B) slabs = B;
    const int nb = (B + slabs - 1) / slabs;
    hipLaunchKernelGGL(chan_layernorm_bwd_param_kernel, dim3(C, (B + nb - 1) / nb), dim3(256), 0, s, in, post, stats, y, dy, dgamma,
                       dbeta, B, C, T, nb);
    GLOWTTS_LAUNCH_CHECK("glowtts_chan_layernorm_bwd");
}

// `_ex` forms: mask_x (B, T) multiplies x, (drop (B, C, T) keep bytes, drop_scale) apply dropout to res — each may be NULL;
// the backward writes dx = dv * mask_x and, when dres != NULL, dres = dv * keep * drop_scale (else the gradient of res is dx)
extern "C" int glowtts_chan_layernorm_fwd_ex(const float *x, const float *res, const float *mask_x, const unsigned char *drop,
                                             float drop_scale, const float *gamma, const float *beta, float *y, float *stats,
                                             int B, int C, int T, float eps, glowtts_stream_t stream) {
    return glowtts_chan_layernorm_fwd_act(x, res, mask_x, drop, drop_scale, gamma, beta, y, stats, 0, 0, nullptr, 1.f, B, C, T, eps,
                                          stream);
}

extern "C" int glowtts_chan_layernorm_fwd(const float *x, const float *res, const float *gamma, const float *beta, float *y,
                                          float *stats, int B, int C, int T, float eps, glowtts_stream_t stream) {
    return glowtts_chan_layernorm_fwd_ex(x, res, nullptr, nullptr, 1.f, gamma, beta, y, stats, B, C, T, eps, stream);
}

extern "C" int glowtts_chan_layernorm_bwd_ex(const float *x, const float *res, const float *mask_x, const unsigned char *drop,
                                             float drop_scale, const float *gamma, const float *stats, const float *dy,
                                             float *dx, float *dres, float *dgamma, float *dbeta, int B, int C, int T,
                                             glowtts_stream_t stream) {
    return glowtts_chan_layernorm_bwd_act(x, res, mask_x, drop, drop_scale, gamma, stats, nullptr, dy, 0, 0, nullptr, 1.f, dx, dres,
                                          dgamma, dbeta, B, C, T, stream);
}

extern "C" int glowtts_chan_layernorm_bwd(const float *x, const float *res, const float *gamma, const float *stats,
                                          const float *dy, float *dx, float *dgamma, float *dbeta, int B, int C, int T,
                                          glowtts_stream_t stream) {
    return glowtts_chan_layernorm_bwd_ex(x, res, nullptr, nullptr, 1.f, gamma, stats, dy, dx, nullptr, dgamma, dbeta, B, C, T,
                                         stream);
}
