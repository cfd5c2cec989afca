// train_ops.hip — the remaining HBM-bound pieces of the training step for gfx950:
//   squeeze / unsqueeze                    (utils.py:135-160)
//   mle_loss forward reduction + backward  (utils.py:14-23)
//   clip_grad_value_ over a flat buffer    (utils.py:118-132)   — replaces one .item() host sync PER PARAMETER TENSOR
//   Adam + Noam schedule over flat buffers (optimize.py:8-64)   — learning rate derived on device from a step
//                                                                  counter, so the whole step is hipGraph-capturable
#include "common.hpp"

namespace glowtts {

// ------------------------------------------------------------------------------------------------------------
// squeeze: thread -> (b, c, t'), reads n consecutive frames (8 B for n = 2), writes n rows (each coalesced in t').
// ------------------------------------------------------------------------------------------------------------
template <bool B16>     // B16: the SQUEEZED tensor is bf16 in HBM (the un-squeezed one — mel frames, latent — is always fp32)
__global__ __launch_bounds__(256) void squeeze_kernel(const float *__restrict__ x, const float *__restrict__ mask,
                                                      void *__restrict__ xs, float *__restrict__ ms, int B, int C, int T,
                                                      int n) {
    const int Ts = T / n;
    const long total = (long)B * C * Ts;
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i >= total) return;
    const int ts = (int)(i % Ts);
    const long row = i / Ts;
    const int c = (int)(row % C);
    const int b = (int)(row / C);
    const float m = mask ? mask[(long)b * T + (long)ts * n + (n - 1)] : 1.0f;
    const float *src = x + row * T + (long)ts * n;
    for (int s = 0; s < n; ++s) {
        Vec<1> v;
        v.d = src[s] * m;
        VecIO<1, B16>::store(xs, ((long)b * n * C + (long)s * C + c) * Ts + ts, v);
    }
    if (c == 0 && ms) ms[(long)b * Ts + ts] = m;
}

template <bool B16>
__global__ __launch_bounds__(256) void unsqueeze_kernel(const void *__restrict__ xs, const float *__restrict__ ms,
                                                        float *__restrict__ x, float *__restrict__ mask_out, int B, int C,
                                                        int Ts, int n) {
    const long total = (long)B * C * Ts;
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i >= total) return;
    const int ts = (int)(i % Ts);
    const long row = i / Ts;
    const int c = (int)(row % C);
    const int b = (int)(row / C);
    const float m = ms ? ms[(long)b * Ts + ts] : 1.0f;
    float *dst = x + row * (long)Ts * n + (long)ts * n;
    for (int s = 0; s < n; ++s) dst[s] = VecIO<1, B16>::load(xs, ((long)b * n * C + (long)s * C + c) * Ts + ts).d * m;
    if (c == 0 && mask_out)
        for (int s = 0; s < n; ++s) mask_out[(long)b * Ts * n + (long)ts * n + s] = m;
}

// ------------------------------------------------------------------------------------------------------------
// mle_loss
// ------------------------------------------------------------------------------------------------------------
template <int V>
__global__ __launch_bounds__(256) void mle_fwd_kernel(const float *__restrict__ z, const float *__restrict__ m,
                                                      const float *__restrict__ logs, const float *__restrict__ mask,
                                                      float *__restrict__ acc, long nv, long nmask) {
    __shared__ float red[4], red2[4];
    float s = 0.f;
    const long stride = (long)gridDim.x * 256;
    for (long i0 = (long)blockIdx.x * 256 + threadIdx.x; i0 < nv; i0 += 2 * stride) {
        Vec<V> zv[2], mv[2], lv[2];
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const bool ok = i0 + u * stride < nv;
            const long i = ok ? i0 + u * stride : i0;
            zv[u] = Vec<V>::load(z + i * V);
            mv[u] = Vec<V>::load(m + i * V);
            lv[u] = Vec<V>::load(logs + i * V);
        }
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            if (i0 + u * stride >= nv) continue;
#pragma unroll
            for (int j = 0; j < V; ++j) {
                const float d = zv[u][j] - mv[u][j];
                s += lv[u][j] + 0.5f * expf(-2.0f * lv[u][j]) * d * d;
            }
        }
    }
    // sum(mask) rides along: every workgroup adds its grid-stride share (one workgroup walking all B*T' mask values was
    // 40 of this kernel's 49 us), and both sums leave in ONE atomic instruction (lanes 0 and 1, adjacent addresses)
    float t = 0.f;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < nmask; i += stride) t += mask[i];
    s = wave_sum(s);
    t = wave_sum(t);
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    if (lane == 0) { red[wave] = s; red2[wave] = t; }
    __syncthreads();
    if (threadIdx.x < 2) {
        const float *r = threadIdx.x == 0 ? red : red2;
        atomicAdd(acc + threadIdx.x, (r[0] + r[1]) + (r[2] + r[3]));
    }
}

template <int V>
__global__ __launch_bounds__(256) void mle_bwd_kernel(const float *__restrict__ z, const float *__restrict__ m,
                                                      const float *__restrict__ logs, const float *__restrict__ scale,
                                                      float *__restrict__ dz, float *__restrict__ dm,
                                                      float *__restrict__ dlogs, long nv,
                                                      const float *__restrict__ denom = nullptr,
                                                      float *__restrict__ dlogdet = nullptr, int B = 0) {
    const float sc = denom ? scale[0] / denom[0] : scale[0];
    if (dlogdet != nullptr && blockIdx.x == 0)
        for (int b = threadIdx.x; b < B; b += 256) dlogdet[b] = -sc;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < nv; i += (long)gridDim.x * 256) {
        Vec<V> zv = Vec<V>::load(z + i * V);
        Vec<V> mv = Vec<V>::load(m + i * V);
        Vec<V> lv = Vec<V>::load(logs + i * V);
        Vec<V> gz, gm, gl;
#pragma unroll
        for (int j = 0; j < V; ++j) {
            const float d = zv[j] - mv[j];
            const float e = expf(-2.0f * lv[j]);
            gz[j] = sc * e * d;
            gm[j] = -gz[j];
            gl[j] = sc * (1.0f - e * d * d);
        }
        gz.store(dz + i * V);
        gm.store(dm + i * V);
        gl.store(dlogs + i * V);
    }
}

// ------------------------------------------------------------------------------------------------------------
// clip + Adam/Noam on flat buffers
// The scalar tails of the losses as kernels of their own (each was 5-8 one-element torch launches on the critical path
// between the decoder's forward and its backward).
//   mle_finish : out[0] = (acc[0] - sum_b logdet[b]) / (acc[1] * C) + 0.5 log(2 pi) ; out[1] = acc[1] * C     (utils.py:17-22)
//   dur_fwd    : out[0] = sum (logw - logw_)^2 / sum_b lengths[b] ; out[1] = sum_b lengths[b]                  (utils.py:26-28)
//   dur_bwd    : dlogw = 2 (logw - logw_) dloss / out[1]
//   span_logw  : logw_[b][x] = log(1e-8 + first[b][x+1] - first[b][x]) for x < t_x[b], else 0                 (models.py:392)
__global__ __launch_bounds__(256) void mle_finish_kernel(const float *__restrict__ acc, const float *__restrict__ logdet, int B,
                                                         int C, float *__restrict__ out) {
    __shared__ float red[4];
    float s = 0.f;
    for (int b = threadIdx.x; b < B; b += 256) s += logdet[b];
    s = block_sum_256(s, red);
    if (threadIdx.x == 0) {
        const float denom = acc[1] * (float)C;
        out[0] = (acc[0] - s) / denom + 0.91893853320467274178f;
        out[1] = denom;
    }
}

__global__ __launch_bounds__(256) void dur_fwd_kernel(const float *__restrict__ logw, const float *__restrict__ logw_,
                                                      const long long *__restrict__ lengths, int B, long n,
                                                      float *__restrict__ out) {
    __shared__ float red[4];
    float s = 0.f, l = 0.f;
    for (long i = threadIdx.x; i < n; i += 256) {
        const float d = logw[i] - logw_[i];
        s += d * d;
    }
    for (int b = threadIdx.x; b < B; b += 256) l += (float)lengths[b];
    s = block_sum_256(s, red);
    __syncthreads();
    l = block_sum_256(l, red);
    if (threadIdx.x == 0) {
        out[0] = s / l;
        out[1] = l;
    }
}

__global__ __launch_bounds__(256) void dur_bwd_kernel(const float *__restrict__ logw, const float *__restrict__ logw_,
                                                      const float *__restrict__ dloss, const float *__restrict__ denom,
                                                      float *__restrict__ dlogw, long n) {
    const float sc = 2.0f * dloss[0] / denom[0];
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) dlogw[i] = sc * (logw[i] - logw_[i]);
}

__global__ __launch_bounds__(256) void span_logw_kernel(const int *__restrict__ first, const int *__restrict__ t_x, int B, int Tx,
                                                        float *__restrict__ out) {
    const long n = (long)B * Tx;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
        const int b = (int)(i / Tx), x = (int)(i - (long)b * Tx);
        const int *f = first + (long)b * (Tx + 1);
        out[i] = x < t_x[b] ? logf(1e-8f + (float)(f[x + 1] - f[x])) : 0.0f;
    }
}

// ------------------------------------------------------------------------------------------------------------
// Reducing kernels end in one same-address atomic per workgroup, and those retire serially in L2 (~13 ns each): they
// run on a small grid (reduce_grid) and get their memory-level parallelism from 4 independent vector loads per thread.
template <int V>
__global__ __launch_bounds__(256) void clip_kernel(float *__restrict__ g, long nv, float clip, float *__restrict__ sumsq) {
    __shared__ float red[4];
    float s = 0.f;
    const long stride = (long)gridDim.x * 256;
    for (long i0 = (long)blockIdx.x * 256 + threadIdx.x; i0 < nv; i0 += 4 * stride) {
        Vec<V> gv[4];
#pragma unroll
        for (int u = 0; u < 4; ++u)
            gv[u] = (i0 + u * stride < nv) ? Vec<V>::load(g + (i0 + u * stride) * V) : Vec<V>::zero();
#pragma unroll
        for (int u = 0; u < 4; ++u) {
#pragma unroll
            for (int j = 0; j < V; ++j) {
                s += gv[u][j] * gv[u][j];
                gv[u][j] = fminf(fmaxf(gv[u][j], -clip), clip);
            }
            if (i0 + u * stride < nv) gv[u].store(g + (i0 + u * stride) * V);
        }
    }
    s = block_sum_256(s, red);
    if (threadIdx.x == 0 && sumsq) atomicAdd(sumsq, s);
}

__device__ __forceinline__ float noam_rate(float step, float lr, float dim_model, float warmup) {
    if (warmup <= 0.f) return lr;
    // optimize.py:32-41 — lr * d^-0.5 * min(s^-0.5, s * w^-1.5); fp64 like the reference's numpy arithmetic
    const double s = (double)step;
    const double a = 1.0 / sqrt(s);
    const double b = s * pow((double)warmup, -1.5);
    return (float)((double)lr * (1.0 / sqrt((double)dim_model)) * (a < b ? a : b));
}

template <int V>
__global__ __launch_bounds__(256) void adam_kernel(float *__restrict__ p, const float *__restrict__ g, float *__restrict__ m,
                                                   float *__restrict__ v, long nv, const float *__restrict__ state, float lr,
                                                   float b1, float b2, float eps, float dim_model, float warmup) {
    // torch.optim.Adam (no amsgrad, no weight decay): step_size = lr_t / (1 - b1^t); denom = sqrt(v)/sqrt(1 - b2^t) + eps
    const float t = state[0];
    // state[3] > 0: a learning rate imposed for this one update (a resumed optimizer applies the rate stored in its
    // checkpoint before its schedule takes over again, as torch's load_state_dict leaves it in the reference)
    const float lr_t = state[3] > 0.f ? state[3] : noam_rate(state[1], lr, dim_model, warmup);
    const double bc1 = 1.0 - pow((double)b1, (double)t);
    const double bc2 = 1.0 - pow((double)b2, (double)t);
    const float step_size = (float)((double)lr_t / bc1);
    const float inv_sqrt_bc2 = (float)(1.0 / sqrt(bc2));
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < nv; i += (long)gridDim.x * 256) {
        Vec<V> pv = Vec<V>::load(p + i * V);
        Vec<V> gv = Vec<V>::load(g + i * V);
        Vec<V> mv = Vec<V>::load(m + i * V);
        Vec<V> vv = Vec<V>::load(v + i * V);
#pragma unroll
        for (int j = 0; j < V; ++j) {
            mv[j] = b1 * mv[j] + (1.0f - b1) * gv[j];
            vv[j] = b2 * vv[j] + (1.0f - b2) * gv[j] * gv[j];
            const float denom = sqrtf(vv[j]) * inv_sqrt_bc2 + eps;
            pv[j] -= step_size * (mv[j] / denom);
        }
        pv.store(p + i * V);
        mv.store(m + i * V);
        vv.store(v + i * V);
    }
}

__global__ void adam_advance_kernel(float *state, float lr, float dim_model, float warmup) {
    if (threadIdx.x == 0 && blockIdx.x == 0) {
        state[0] += 1.0f;
        state[1] += 1.0f;
        state[2] = noam_rate(state[1], lr, dim_model, warmup);
        state[3] = 0.f;
    }
}

static inline int stream_grid(long nv) {
    long g = (nv + 255) / 256;
    if (g > 2048) g = 2048;   // 256 CUs x 8 workgroups, grid-stride the rest (cdna_hip_programming.md G11)
    if (g < 1) g = 1;
    return (int)g;
}

static inline int reduce_grid(long nv) {
    long g = (nv + 255) / 256;
    if (g > 512) g = 512;
    if (g < 1) g = 1;
    return (int)g;
}

}  // namespace glowtts

using namespace glowtts;

// `_io` forms: io = 1 -> the squeezed tensor (xs) is bf16 in HBM, the un-squeezed one stays fp32
extern "C" int glowtts_squeeze_io(const float *x, const float *mask, void *xs, float *ms, int B, int C, int T, int n, int io,
                                  glowtts_stream_t stream) {
    GLOWTTS_CHECK_ARG(x && xs, "glowtts_squeeze: null pointer");
    GLOWTTS_CHECK_ARG(B >= 0 && C > 0 && T >= 0 && n >= 1, "glowtts_squeeze: bad shape");
    const long total = (long)B * C * (T / n);
    if (total == 0) return 0;
    if (io) hipLaunchKernelGGL(squeeze_kernel<true>, dim3(cdiv(total, 256)), dim3(256), 0, (hipStream_t)stream, x, mask, xs, ms, B, C, T, n);
    else    hipLaunchKernelGGL(squeeze_kernel<false>, dim3(cdiv(total, 256)), dim3(256), 0, (hipStream_t)stream, x, mask, xs, ms, B, C, T, n);
    GLOWTTS_LAUNCH_CHECK("glowtts_squeeze");
}

extern "C" int glowtts_squeeze(const float *x, const float *mask, float *xs, float *ms, int B, int C, int T, int n,
                               glowtts_stream_t stream) {
    return glowtts_squeeze_io(x, mask, xs, ms, B, C, T, n, 0, stream);
}

extern "C" int glowtts_unsqueeze_io(const void *xs, const float *ms, float *x, float *mask_out, int B, int C, int Tsq,
                                    int n, int io, glowtts_stream_t stream) {
    GLOWTTS_CHECK_ARG(xs && x, "glowtts_unsqueeze: null pointer");
    GLOWTTS_CHECK_ARG(B >= 0 && C > 0 && Tsq >= 0 && n >= 1, "glowtts_unsqueeze: bad shape");
    const long total = (long)B * C * Tsq;
    if (total == 0) return 0;
    if (io) hipLaunchKernelGGL(unsqueeze_kernel<true>, dim3(cdiv(total, 256)), dim3(256), 0, (hipStream_t)stream, xs, ms, x, mask_out, B, C, Tsq, n);
    else    hipLaunchKernelGGL(unsqueeze_kernel<false>, dim3(cdiv(total, 256)), dim3(256), 0, (hipStream_t)stream, xs, ms, x, mask_out, B, C, Tsq, n);
    GLOWTTS_LAUNCH_CHECK("glowtts_unsqueeze");
}

extern "C" int glowtts_unsqueeze(const float *xs, const float *ms, float *x, float *mask_out, int B, int C, int Tsq,
                                 int n, glowtts_stream_t stream) {
    return glowtts_unsqueeze_io(xs, ms, x, mask_out, B, C, Tsq, n, 0, stream);
}

extern "C" int glowtts_mle_fwd(const float *z, const float *m, const float *logs, const float *mask, float *acc,
                               int B, int C, int T, glowtts_stream_t stream) {
    GLOWTTS_CHECK_ARG(z && m && logs && mask && acc, "glowtts_mle_fwd: null pointer");
    GLOWTTS_CHECK_ARG(B >= 0 && C > 0 && T >= 0, "glowtts_mle_fwd: bad shape");
    const long n = (long)B * C * T;
    if (n == 0) return 0;
    hipStream_t s = (hipStream_t)stream;
    if ((n & 3) == 0 && aligned16(z) && aligned16(m) && aligned16(logs))
        hipLaunchKernelGGL((mle_fwd_kernel<4>), dim3(reduce_grid(n / 4)), dim3(256), 0, s, z, m, logs, mask, acc, n / 4, (long)B * T);
    else
        hipLaunchKernelGGL((mle_fwd_kernel<1>), dim3(reduce_grid(n)), dim3(256), 0, s, z, m, logs, mask, acc, n, (long)B * T);
    GLOWTTS_LAUNCH_CHECK("glowtts_mle_fwd");
}

extern "C" int glowtts_mle_bwd(const float *z, const float *m, const float *logs, const float *scale, float *dz,
                               float *dm, float *dlogs, int64_t n, glowtts_stream_t stream) {
    GLOWTTS_CHECK_ARG(z && m && logs && scale && dz && dm && dlogs, "glowtts_mle_bwd: null pointer");
    GLOWTTS_CHECK_ARG(n >= 0, "glowtts_mle_bwd: negative size");
    if (n == 0) return 0;
    hipStream_t s = (hipStream_t)stream;
    if ((n & 3) == 0 && aligned16(z) && aligned16(m) && aligned16(logs) && aligned16(dz) && aligned16(dm) && aligned16(dlogs))
        hipLaunchKernelGGL((mle_bwd_kernel<4>), dim3(stream_grid(n / 4)), dim3(256), 0, s, z, m, logs, scale, dz, dm, dlogs, (long)(n / 4));
    else
        hipLaunchKernelGGL((mle_bwd_kernel<1>), dim3(stream_grid(n)), dim3(256), 0, s, z, m, logs, scale, dz, dm, dlogs, (long)n);
    GLOWTTS_LAUNCH_CHECK("glowtts_mle_bwd");
}

extern "C" int glowtts_mle_loss_fwd(const float *z, const float *m, const float *logs, const float *mask, const float *logdet,
                                    float *acc, float *out, int B, int C, int T, glowtts_stream_t stream) {
    GLOWTTS_CHECK_ARG(logdet && out, "glowtts_mle_loss_fwd: null pointer");
    if (int rc = glowtts_mle_fwd(z, m, logs, mask, acc, B, C, T, stream)) return rc;
    if ((long)B * C * T == 0) return 0;
    hipLaunchKernelGGL(mle_finish_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, acc, logdet, B, C, out);
    GLOWTTS_LAUNCH_CHECK("glowtts_mle_loss_fwd");
}

extern "C" int glowtts_mle_loss_bwd(const float *z, const float *m, const float *logs, const float *dloss, const float *denom,
                                    float *dz, float *dm, float *dlogs, float *dlogdet, int B, int64_t n,
                                    glowtts_stream_t stream) {
    GLOWTTS_CHECK_ARG(z && m && logs && dloss && denom && dz && dm && dlogs && dlogdet, "glowtts_mle_loss_bwd: null pointer");
    GLOWTTS_CHECK_ARG(n >= 0 && B >= 0, "glowtts_mle_loss_bwd: negative size");
    if (n == 0) return 0;
    hipStream_t s = (hipStream_t)stream;
    if ((n & 3) == 0 && aligned16(z) && aligned16(m) && aligned16(logs) && aligned16(dz) && aligned16(dm) && aligned16(dlogs))
        hipLaunchKernelGGL((mle_bwd_kernel<4>), dim3(stream_grid(n / 4)), dim3(256), 0, s, z, m, logs, dloss, dz, dm, dlogs, (long)(n / 4), denom, dlogdet, B);
    else
        hipLaunchKernelGGL((mle_bwd_kernel<1>), dim3(stream_grid(n)), dim3(256), 0, s, z, m, logs, dloss, dz, dm, dlogs, (long)n, denom, dlogdet, B);
    GLOWTTS_LAUNCH_CHECK("glowtts_mle_loss_bwd");
}

extern "C" int glowtts_duration_loss_fwd(const float *logw, const float *logw_, const long long *lengths, float *out, int B,
                                         int64_t n, glowtts_stream_t stream) {
    GLOWTTS_CHECK_ARG(logw && logw_ && lengths && out, "glowtts_duration_loss_fwd: null pointer");
    GLOWTTS_CHECK_ARG(B >= 0 && n >= 0, "glowtts_duration_loss_fwd: negative size");
    hipLaunchKernelGGL(dur_fwd_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, logw, logw_, lengths, B, (long)n, out);
    GLOWTTS_LAUNCH_CHECK("glowtts_duration_loss_fwd");
}

extern "C" int glowtts_duration_loss_bwd(const float *logw, const float *logw_, const float *dloss, const float *denom,
                                         float *dlogw, int64_t n, glowtts_stream_t stream) {
    GLOWTTS_CHECK_ARG(logw && logw_ && dloss && denom && dlogw, "glowtts_duration_loss_bwd: null pointer");
    GLOWTTS_CHECK_ARG(n >= 0, "glowtts_duration_loss_bwd: negative size");
    if (n == 0) return 0;
    hipLaunchKernelGGL(dur_bwd_kernel, dim3(stream_grid(n)), dim3(256), 0, (hipStream_t)stream, logw, logw_, dloss, denom, dlogw, (long)n);
    GLOWTTS_LAUNCH_CHECK("glowtts_duration_loss_bwd");
}

extern "C" int glowtts_span_logw(const int32_t *first, const int32_t *t_x, float *logw_, int B, int Tx, glowtts_stream_t stream) {
    GLOWTTS_CHECK_ARG(first && t_x && logw_, "glowtts_span_logw: null pointer");
    GLOWTTS_CHECK_ARG(B >= 0 && Tx >= 0, "glowtts_span_logw: negative size");
    if ((long)B * Tx == 0) return 0;
    hipLaunchKernelGGL(span_logw_kernel, dim3(stream_grid((long)B * Tx)), dim3(256), 0, (hipStream_t)stream, first, t_x, B, Tx, logw_);
    GLOWTTS_LAUNCH_CHECK("glowtts_span_logw");
}

extern "C" int glowtts_clip_grad_value(float *g, int64_t n, float clip, float *sumsq, glowtts_stream_t stream) {
    GLOWTTS_CHECK_ARG(g, "glowtts_clip_grad_value: null pointer");
    GLOWTTS_CHECK_ARG(n >= 0 && clip >= 0.f, "glowtts_clip_grad_value: bad argument");
    if (n == 0) return 0;
    hipStream_t s = (hipStream_t)stream;
    if ((n & 3) == 0 && aligned16(g))
        hipLaunchKernelGGL((clip_kernel<4>), dim3(reduce_grid(n / 4)), dim3(256), 0, s, g, (long)(n / 4), clip, sumsq);
    else
        hipLaunchKernelGGL((clip_kernel<1>), dim3(reduce_grid(n)), dim3(256), 0, s, g, (long)n, clip, sumsq);
    GLOWTTS_LAUNCH_CHECK("glowtts_clip_grad_value");
}

extern "C" int glowtts_adam_noam(float *p, const float *g, float *m, float *v, int64_t n, const float *state,
                                 float lr, float beta1, float beta2, float eps, float dim_model, float warmup,
                                 glowtts_stream_t stream) {
    GLOWTTS_CHECK_ARG(p && g && m && v && state, "glowtts_adam_noam: null pointer");
    GLOWTTS_CHECK_ARG(n >= 0, "glowtts_adam_noam: negative size");
    if (n == 0) return 0;
    hipStream_t s = (hipStream_t)stream;
    if ((n & 3) == 0 && aligned16(p) && aligned16(g) && aligned16(m) && aligned16(v))
        hipLaunchKernelGGL((adam_kernel<4>), dim3(stream_grid(n / 4)), dim3(256), 0, s, p, g, m, v, (long)(n / 4), state, lr, beta1, beta2, eps, dim_model, warmup);
    else
        hipLaunchKernelGGL((adam_kernel<1>), dim3(stream_grid(n)), dim3(256), 0, s, p, g, m, v, (long)n, state, lr, beta1, beta2, eps, dim_model, warmup);
    GLOWTTS_LAUNCH_CHECK("glowtts_adam_noam");
}

extern "C" int glowtts_adam_advance(float *state, float lr, float dim_model, float warmup, glowtts_stream_t stream) {
    GLOWTTS_CHECK_ARG(state, "glowtts_adam_advance: null pointer");
    hipLaunchKernelGGL(adam_advance_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, state, lr, dim_model, warmup);
    GLOWTTS_LAUNCH_CHECK("glowtts_adam_advance");
}

// ------------------------------------------------------------------------------------------------------------
// phoneme embedding (reference models.py:90,121: self.emb(x) * sqrt(H), transposed to (B, H, T)) and its backward
//   fwd : out[b][h][t] = weight[ids[b][t]][h] * scale          (the (B, T, H) tensor and its transpose never exist)
//   bwd : dweight[v][h] += scale * sum_{(b,t): ids == v} dout[b][h][t]
// One workgroup per vocabulary entry in the backward: it collects its positions through LDS in chunks and sums them row by
// row — a segment sum without atomics (torch's embedding_dense_backward sorts the ids with a device-wide partition first,
// which also made the step un-capturable in a one-stream hipGraph).
// ------------------------------------------------------------------------------------------------------------
namespace glowtts {

__global__ __launch_bounds__(256) void embed_fwd_kernel(const long *__restrict__ ids, const float *__restrict__ w, float scale,
                                                        float *__restrict__ out, int T, int H, int V) {
    const int b = blockIdx.y, t = blockIdx.x * 64 + (threadIdx.x & 63), wave = threadIdx.x >> 6;
    if (t >= T) return;
    const long id = ids[(long)b * T + t];
    // an id outside the vocabulary (a bad phoneme map) is a device assert in the reference's nn.Embedding; here its column is
    // NaN, so the step's loss is NaN instead of a model that trains silently on a clamped id (the backward skips such ids)
    const bool ok = id >= 0 && id < V;
    const float *row = w + (ok ? id : 0) * H;
    for (int h = wave; h < H; h += 4) out[((long)b * H + h) * T + t] = ok ? row[h] * scale : __builtin_nanf("");
}

__global__ __launch_bounds__(256) void embed_bwd_kernel(const long *__restrict__ ids, const float *__restrict__ dout, float scale,
                                                        float *__restrict__ dw, int B, int T, int H) {
    constexpr int CH = 2048;
    __shared__ int pos[CH];
    __shared__ int wcount[4];
    const int v = blockIdx.x, n = B * T;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    float acc[4] = {0.f, 0.f, 0.f, 0.f};                     // h = tid, tid + 256, ... (H <= 1024)
    for (int c0 = 0; c0 < n; c0 += CH) {
        // order-preserving compaction of the positions that hold token v (an LDS atomicAdd here made the summation
        // order — and with it the last bits of emb.weight.grad — vary from run to run): ballot + prefix inside a wave, wave
        // counts through LDS, 256 positions per round
        int m = 0;
        for (int r0 = c0; r0 < min(n, c0 + CH); r0 += 256) {
            const int i = r0 + threadIdx.x;
            const bool hit = i < n && ids[i] == v;
            const unsigned long long bal = __ballot(hit);
            if (lane == 0) wcount[wave] = __popcll(bal);
            __syncthreads();
            int base = m;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                if (q < wave) base += wcount[q];
                m += wcount[q];
            }
            if (hit) pos[base + __popcll(bal & ((1ull << lane) - 1ull))] = i;
            __syncthreads();
        }
        for (int j = 0; j < m; ++j) {
            const int i = pos[j], b = i / T, t = i - b * T;
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const int h = threadIdx.x + 256 * k;
                if (h < H) acc[k] += dout[((long)b * H + h) * T + t];
            }
        }
        __syncthreads();
    }
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const int h = threadIdx.x + 256 * k;
        if (h < H && acc[k] != 0.f) dw[(long)v * H + h] += acc[k] * scale;
    }
}

// Dropout keep-masks (1 = keep) for a whole step's worth of activations from one launch: Philox4x32-7 (Salmon et al., SC'11: passes
// BigCrush from 7 rounds), counter = the index of an 8-byte group, key = the seed; each 32-bit output gives two 16-bit draws, a
// byte is 1 when its draw is >= round(p * 65536) (p to 1.5e-5).  Reference: torch.nn.functional.dropout inside layers.py:147 —
// a Bernoulli(1 - p) keep decision per element; which generator makes it is not part of the model.  torch's bernoulli_ spends a
// Philox-10 call on four BYTES: 203 us for the 236 MB of a config-2 step, on the decoder's forward chain.
__device__ __forceinline__ void philox_round(unsigned (&c)[4], unsigned k0, unsigned k1) {
    const unsigned long long p0 = (unsigned long long)0xD2511F53u * c[0], p1 = (unsigned long long)0xCD9E8D57u * c[2];
    const unsigned n0 = (unsigned)(p1 >> 32) ^ c[1] ^ k0, n2 = (unsigned)(p0 >> 32) ^ c[3] ^ k1;
    c[1] = (unsigned)p1; c[3] = (unsigned)p0; c[0] = n0; c[2] = n2;
}

__global__ __launch_bounds__(256) void keep_mask_kernel(unsigned char *__restrict__ out, long n, unsigned seed_lo, unsigned seed_hi,
                                                        unsigned thr) {
    const long groups = (n + 7) >> 3;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < groups; i += (long)gridDim.x * 256) {
        unsigned c[4] = {(unsigned)i, (unsigned)(i >> 32), 0x6b656570u, 0x6d61736bu};
        unsigned k0 = seed_lo, k1 = seed_hi;
#pragma unroll
        for (int r = 0; r < 7; ++r) {
            philox_round(c, k0, k1);
            k0 += 0x9E3779B9u;
            k1 += 0xBB67AE85u;
        }
        unsigned w[2];
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const unsigned a = c[2 * h], b = c[2 * h + 1];
            w[h] = ((a & 0xffffu) >= thr ? 1u : 0u) | ((a >> 16) >= thr ? 0x100u : 0u) | ((b & 0xffffu) >= thr ? 0x10000u : 0u) |
                   ((b >> 16) >= thr ? 0x1000000u : 0u);
        }
        if (i * 8 + 8 <= n) {
            *reinterpret_cast<uint2 *>(out + i * 8) = make_uint2(w[0], w[1]);
        } else {
            for (long j = i * 8; j < n; ++j) out[j] = (unsigned char)((w[(j >> 2) & 1] >> (8 * (j & 3))) & 1u);
        }
    }
}

}  // namespace glowtts

extern "C" int glowtts_embed_fwd(const long long *ids, const float *weight, float scale, float *out, int B, int T, int H, int V,
                                 glowtts_stream_t stream) {
    GLOWTTS_CHECK_ARG(ids && weight && out, "glowtts_embed_fwd: null pointer");
    GLOWTTS_CHECK_ARG(B >= 0 && T >= 0 && H > 0 && V > 0, "glowtts_embed_fwd: bad shape");
    if ((long)B * T == 0) return 0;
    hipLaunchKernelGGL(glowtts::embed_fwd_kernel, dim3((T + 63) / 64, B), dim3(256), 0, (hipStream_t)stream,
                       reinterpret_cast<const long *>(ids), weight, scale, out, T, H, V);
    GLOWTTS_LAUNCH_CHECK("glowtts_embed_fwd");
}

extern "C" int glowtts_embed_bwd(const long long *ids, const float *dout, float scale, float *dweight, int B, int T, int H, int V,
                                 glowtts_stream_t stream) {
    GLOWTTS_CHECK_ARG(ids && dout && dweight, "glowtts_embed_bwd: null pointer");
    GLOWTTS_CHECK_ARG(B >= 0 && T >= 0 && H > 0 && H <= 1024 && V > 0, "glowtts_embed_bwd: bad shape (H <= 1024)");
    if ((long)B * T == 0) return 0;
    hipLaunchKernelGGL(glowtts::embed_bwd_kernel, dim3(V), dim3(256), 0, (hipStream_t)stream, reinterpret_cast<const long *>(ids),
                       dout, scale, dweight, B, T, H);
    GLOWTTS_LAUNCH_CHECK("glowtts_embed_bwd");
}

extern "C" int glowtts_keep_mask(unsigned char *out, long n, unsigned long long seed, float p_drop, glowtts_stream_t stream) {
    GLOWTTS_CHECK_ARG(out && n >= 0 && p_drop >= 0.f && p_drop < 1.f, "glowtts_keep_mask: bad argument");
    GLOWTTS_CHECK_ARG((reinterpret_cast<uintptr_t>(out) & 7u) == 0, "glowtts_keep_mask: the mask buffer must be 8-byte aligned");
    if (n == 0) return 0;
    const unsigned thr = (unsigned)(p_drop * 65536.0f + 0.5f);
    long grid = (((n + 7) >> 3) + 255) / 256;
    if (grid > 8192) grid = 8192;
    hipLaunchKernelGGL(glowtts::keep_mask_kernel, dim3((unsigned)grid), dim3(256), 0, (hipStream_t)stream, out, n, (unsigned)seed,
                       (unsigned)(seed >> 32), thr);
    GLOWTTS_LAUNCH_CHECK("glowtts_keep_mask");
}
