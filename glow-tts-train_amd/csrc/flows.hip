// flows.hip — the invertible, HBM-bound operators of FlowSpecDecoder for gfx950:
//   ActNorm        (layers.py:182-221)      z = (bias + exp(logs) x) mask
//   InvConvNear    (layers.py:238-275)      grouped n x n channel mix + log|det W|
//   affine coupling apply (attentions.py:128-142)
//
// All tensors are (B, C, T) fp32 with T contiguous; every kernel reads each input once and writes each output
// once with 16-byte accesses along T (scalar fall-back when T % 4 != 0 or a pointer is not 16-B aligned).
// Algorithmic bytes per squeezed column (e = 4): ActNorm 2C e, InvConvNear 2C e, coupling 3C e
// (2.5C e once z_0 aliases x_0), backward = read grad + input, write grad.
// Per-channel / per-matrix reductions are wave-shuffle + LDS partial sums followed by ONE float atomic per
// workgroup and output element (global_atomic_add_f32, executed at the memory side: MI355X_MICROARCH.md).
#include "common.hpp"

namespace glowtts {

// ------------------------------------------------------------------------------------------------------------
// mask_len: x_len[b] = sum_t mask[b,t]      one wave per utterance
// ------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(64) void mask_len_kernel(const float *__restrict__ mask, float *__restrict__ x_len, int T) {
    const int b = blockIdx.x;
    float s = 0.f;
    for (int t = threadIdx.x; t < T; t += 64) s += mask[(size_t)b * T + t];
    s = wave_sum(s);
    if (threadIdx.x == 0) x_len[b] = s;
}

// ------------------------------------------------------------------------------------------------------------
// ActNorm
// ------------------------------------------------------------------------------------------------------------
template <int V, bool REV>
__global__ __launch_bounds__(256) void actnorm_fwd_kernel(const float *__restrict__ x, const float *__restrict__ mask,
                                                          const float *__restrict__ logs, const float *__restrict__ bias,
                                                          const float *__restrict__ x_len, float *__restrict__ z,
                                                          float *__restrict__ logdet, int B, int C, int T) {
    __shared__ float red[4];
    const int TV = T / V;
    const long n = (long)B * C * TV;
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i < n) {
        const int tv = (int)(i % TV);
        const long row = i / TV;
        const int c = (int)(row % C);
        const int b = (int)(row / C);
        const float l = logs[c], bi = bias[c];
        const float e = expf(REV ? -l : l);
        Vec<V> xv = Vec<V>::load(x + row * T + (long)tv * V);
        Vec<V> mv = Vec<V>::load(mask + (long)b * T + (long)tv * V);
        Vec<V> o;
#pragma unroll
        for (int j = 0; j < V; ++j) o[j] = REV ? (xv[j] - bi) * e * mv[j] : (bi + e * xv[j]) * mv[j];
        o.store(z + row * T + (long)tv * V);
    }
    if (!REV && logdet != nullptr && blockIdx.x == 0) {
        float s = 0.f;
        for (int c = threadIdx.x; c < C; c += 256) s += logs[c];
        s = block_sum_256(s, red);
        for (int b = threadIdx.x; b < B; b += 256) logdet[b] = s * x_len[b];
    }
}

// grid (C, NB): workgroup = one channel x a slab of utterances.  dx elementwise + the two per-channel sums.
template <int V>
__global__ __launch_bounds__(256) void actnorm_bwd_kernel(const float *__restrict__ x, const float *__restrict__ mask,
                                                          const float *__restrict__ logs, const float *__restrict__ dz,
                                                          const float *__restrict__ dlogdet, const float *__restrict__ x_len,
                                                          float *__restrict__ dx, float *__restrict__ dlogs,
                                                          float *__restrict__ dbias, int B, int C, int T, int nb) {
    __shared__ float red[4];
    const int c = blockIdx.x;
    const int b0 = blockIdx.y * nb;
    const int b1 = min(B, b0 + nb);
    const int TV = T / V;
    const float e = expf(logs[c]);
    float s_logs = 0.f, s_bias = 0.f;
    const int items = (b1 - b0) * TV;
    for (int i = threadIdx.x; i < items; i += 256) {
        const int b = b0 + i / TV;
        const int tv = i % TV;
        const long off = ((long)b * C + c) * T + (long)tv * V;
        Vec<V> xv = Vec<V>::load(x + off);
        Vec<V> gv = Vec<V>::load(dz + off);
        Vec<V> mv = Vec<V>::load(mask + (long)b * T + (long)tv * V);
        Vec<V> o;
#pragma unroll
        for (int j = 0; j < V; ++j) {
            const float gm = gv[j] * mv[j];
            o[j] = gm * e;
            s_logs += gm * e * xv[j];
            s_bias += gm;
        }
        o.store(dx + off);
    }
    if (blockIdx.y == 0 && dlogdet != nullptr) {
        for (int b = threadIdx.x; b < B; b += 256) s_logs += dlogdet[b] * x_len[b];
    }
    s_logs = block_sum_256(s_logs, red);
    s_bias = block_sum_256(s_bias, red);
    if (threadIdx.x == 0) {
        atomicAdd(dlogs + c, s_logs);
        atomicAdd(dbias + c, s_bias);
    }
}

template <int V>
__global__ __launch_bounds__(256) void actnorm_stats_kernel(const float *__restrict__ x, const float *__restrict__ mask,
                                                            float *__restrict__ sum_x, float *__restrict__ sum_x2, int B,
                                                            int C, int T, int nb) {
    __shared__ float red[4];
    const int c = blockIdx.x;
    const int b0 = blockIdx.y * nb;
    const int b1 = min(B, b0 + nb);
    const int TV = T / V;
    float s1 = 0.f, s2 = 0.f;
    const int items = (b1 - b0) * TV;
    for (int i = threadIdx.x; i < items; i += 256) {
        const int b = b0 + i / TV;
        const int tv = i % TV;
        Vec<V> xv = Vec<V>::load(x + ((long)b * C + c) * T + (long)tv * V);
        Vec<V> mv = Vec<V>::load(mask + (long)b * T + (long)tv * V);
#pragma unroll
        for (int j = 0; j < V; ++j) {
            s1 += xv[j] * mv[j];
            s2 += xv[j] * xv[j] * mv[j];
        }
    }
    s1 = block_sum_256(s1, red);
    s2 = block_sum_256(s2, red);
    if (threadIdx.x == 0) {
        atomicAdd(sum_x + c, s1);
        atomicAdd(sum_x2 + c, s2);
    }
}

// ------------------------------------------------------------------------------------------------------------
// InvConvNear
// ------------------------------------------------------------------------------------------------------------
// One wavefront: lane (r, c) = (lane >> 3, lane & 7) holds A[r][c] and Inv[r][c] of an 8x8 frame whose top-left
// n x n block is W (identity elsewhere).  Gauss-Jordan with partial pivoting, rows exchanged by lane shuffles;
// fp64 internally (a few hundred flops), so log|det| and W^-1 are at least as accurate as torch's fp32 LU.
// (multi form: blockIdx.x = problem; w from `w_table`, results at w_inv + blockIdx.x * out_stride, log|det| right behind W^-1)
__global__ __launch_bounds__(64) void invconv_prepare_kernel(const float *__restrict__ w, float *__restrict__ w_inv,
                                                             float *__restrict__ logdet_w, int n,
                                                             const long long *__restrict__ w_table, long out_stride) {
    if (w_table) {
        w = reinterpret_cast<const float *>(w_table[blockIdx.x]);
        w_inv += (long)blockIdx.x * out_stride;
        logdet_w = w_inv + n * n;
    }
    const int lane = threadIdx.x;
    const int r = lane >> 3, c = lane & 7;
    double a = (r < n && c < n) ? (double)w[r * n + c] : (r == c ? 1.0 : 0.0);
    double inv = (r == c) ? 1.0 : 0.0;
    double logabs = 0.0;
    int neg = 0;
    for (int k = 0; k < n; ++k) {
        // pivot: row p >= k maximising |A[p][k]|
        double colv = __shfl(a, (r << 3) + k, 64);
        double best = (r >= k) ? fabs(colv) : -1.0;
        int brow = r;
#pragma unroll
        for (int off = 8; off < 64; off <<= 1) {
            double ob = __shfl_xor(best, off, 64);
            int orow = __shfl_xor(brow, off, 64);
            if (ob > best || (ob == best && orow < brow)) { best = ob; brow = orow; }
        }
        const int p = brow;  // uniform
        const int src_row = (r == k) ? p : ((r == p) ? k : r);
        a = __shfl(a, (src_row << 3) + c, 64);
        inv = __shfl(inv, (src_row << 3) + c, 64);
        if (p != k) neg ^= 1;
        const double piv = __shfl(a, (k << 3) + k, 64);
        logabs += log(fabs(piv));
        if (piv < 0.0) neg ^= 1;
        const double rk_a = __shfl(a, (k << 3) + c, 64) / piv;
        const double rk_i = __shfl(inv, (k << 3) + c, 64) / piv;
        const double f = __shfl(a, (r << 3) + k, 64);
        if (r == k) { a = rk_a; inv = rk_i; }
        else        { a -= f * rk_a; inv -= f * rk_i; }
    }
    if (r < n && c < n) w_inv[r * n + c] = (float)inv;
    if (lane == 0) logdet_w[0] = neg ? __builtin_nanf("") : (float)logabs;
}

// channel of row k (k = h*(N/2)+s) in group g:  h*(C/2) + g*(N/2) + s      (layers.py:247-252)
template <int N>
__device__ __forceinline__ int invconv_channel(int k, int g, int C) {
    return (k / (N / 2)) * (C / 2) + g * (N / 2) + (k % (N / 2));
}

template <int N, int V>
__global__ __launch_bounds__(256) void invconv_fwd_kernel(const float *__restrict__ x, const float *__restrict__ mask,
                                                          const float *__restrict__ w, const float *__restrict__ logdet_w,
                                                          const float *__restrict__ x_len, float *__restrict__ z,
                                                          float *__restrict__ logdet, int B, int C, int T) {
    const int TV = T / V;
    const int G = C / N;
    const long n = (long)B * G * TV;
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i < n) {
        const int tv = (int)(i % TV);
        const long bg = i / TV;
        const int g = (int)(bg % G);
        const int b = (int)(bg / G);
        float wr[N * N];
#pragma unroll
        for (int q = 0; q < N * N; ++q) wr[q] = w[q];
        Vec<V> xin[N];
#pragma unroll
        for (int k = 0; k < N; ++k)
            xin[k] = Vec<V>::load(x + ((long)b * C + invconv_channel<N>(k, g, C)) * T + (long)tv * V);
        Vec<V> mv = Vec<V>::load(mask + (long)b * T + (long)tv * V);
#pragma unroll
        for (int o = 0; o < N; ++o) {
            Vec<V> acc;
#pragma unroll
            for (int j = 0; j < V; ++j) {
                float s = 0.f;
#pragma unroll
                for (int k = 0; k < N; ++k) s += wr[o * N + k] * xin[k][j];
                acc[j] = s * mv[j];
            }
            acc.store(z + ((long)b * C + invconv_channel<N>(o, g, C)) * T + (long)tv * V);
        }
    }
    if (logdet != nullptr && blockIdx.x == 0) {
        const float ld = logdet_w[0] * (float)(C / N);
        for (int b = threadIdx.x; b < B; b += 256) logdet[b] = ld * x_len[b];
    }
}

template <int N, int V>
__global__ __launch_bounds__(256) void invconv_bwd_kernel(const float *__restrict__ x, const float *__restrict__ mask,
                                                          const float *__restrict__ w, const float *__restrict__ w_inv,
                                                          const float *__restrict__ dz, const float *__restrict__ dlogdet,
                                                          const float *__restrict__ x_len, float *__restrict__ dx,
                                                          float *__restrict__ dw, int B, int C, int T) {
    __shared__ float red[4][N * N];
    const int TV = T / V;
    const int G = C / N;
    const long n = (long)B * G * TV;
    float wr[N * N];
#pragma unroll
    for (int q = 0; q < N * N; ++q) wr[q] = w[q];
    float acc[N * N];
#pragma unroll
    for (int q = 0; q < N * N; ++q) acc[q] = 0.f;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
        const int tv = (int)(i % TV);
        const long bg = i / TV;
        const int g = (int)(bg % G);
        const int b = (int)(bg / G);
        Vec<V> xin[N], gz[N];
        Vec<V> mv = Vec<V>::load(mask + (long)b * T + (long)tv * V);
#pragma unroll
        for (int k = 0; k < N; ++k) {
            const long off = ((long)b * C + invconv_channel<N>(k, g, C)) * T + (long)tv * V;
            xin[k] = Vec<V>::load(x + off);
            gz[k] = Vec<V>::load(dz + off);
#pragma unroll
            for (int j = 0; j < V; ++j) gz[k][j] *= mv[j];
        }
#pragma unroll
        for (int k = 0; k < N; ++k) {
            Vec<V> o;
#pragma unroll
            for (int j = 0; j < V; ++j) {
                float s = 0.f;
#pragma unroll
                for (int oo = 0; oo < N; ++oo) s += wr[oo * N + k] * gz[oo][j];
                o[j] = s;
            }
            o.store(dx + ((long)b * C + invconv_channel<N>(k, g, C)) * T + (long)tv * V);
        }
#pragma unroll
        for (int oo = 0; oo < N; ++oo)
#pragma unroll
            for (int k = 0; k < N; ++k)
#pragma unroll
                for (int j = 0; j < V; ++j) acc[oo * N + k] += gz[oo][j] * xin[k][j];
    }
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
#pragma unroll
    for (int q = 0; q < N * N; ++q) {
        const float s = wave_sum(acc[q]);
        if (lane == 0) red[wave][q] = s;
    }
    __syncthreads();
    if (threadIdx.x < N * N) {
        const int q = threadIdx.x;
        float s = red[0][q] + red[1][q] + red[2][q] + red[3][q];
        if (blockIdx.x == 0 && dlogdet != nullptr) {
            float t = 0.f;
            for (int b = 0; b < B; ++b) t += dlogdet[b] * x_len[b];
            const int o = q / N, k = q % N;
            s += w_inv[k * N + o] * (float)(C / N) * t;  // d logdet(W) / dW = W^-T
        }
        atomicAdd(dw + q, s);
    }
}

// ---- any even n_split (reference layers.py:227 asserts nothing more): run-time N forms for N other than 2 / 4 / 8 ----------
// The register-resident kernels above are instantiated for the group sizes Glow-TTS configurations use; an InvConvNear with
// n_split = 6, 10, 16, ... (C % n_split == 0, N <= 32) runs here: one thread per OUTPUT row of a group (its N inputs are read
// through L1 by the N threads of the group), the matrix in LDS; the matrix gradient by one workgroup per (o, k) entry.
constexpr int kInvConvMaxN = 32;

__device__ __forceinline__ int invconv_channel_rt(int k, int g, int C, int N) {
    return (k / (N / 2)) * (C / 2) + g * (N / 2) + (k % (N / 2));
}

// Gauss-Jordan with partial pivoting in LDS, fp64; thread r owns row r
__global__ __launch_bounds__(64) void invconv_prepare_generic_kernel(const float *__restrict__ w, float *__restrict__ w_inv,
                                                                     float *__restrict__ logdet_w, int n,
                                                                     const long long *__restrict__ w_table, long out_stride) {
    if (w_table) {
        w = reinterpret_cast<const float *>(w_table[blockIdx.x]);
        w_inv += (long)blockIdx.x * out_stride;
        logdet_w = w_inv + n * n;
    }
    __shared__ double a[kInvConvMaxN][kInvConvMaxN + 1], inv[kInvConvMaxN][kInvConvMaxN + 1];
    __shared__ int piv_row;
    __shared__ double logabs;
    __shared__ int neg;
    const int r = threadIdx.x;
    if (r < n)
        for (int c = 0; c < n; ++c) { a[r][c] = (double)w[r * n + c]; inv[r][c] = r == c ? 1.0 : 0.0; }
    if (r == 0) { logabs = 0.0; neg = 0; }
    __syncthreads();
    for (int k = 0; k < n; ++k) {
        if (r == 0) {
            int p = k;
            double best = fabs(a[k][k]);
            for (int q = k + 1; q < n; ++q)
                if (fabs(a[q][k]) > best) { best = fabs(a[q][k]); p = q; }
            piv_row = p;
        }
        __syncthreads();
        const int p = piv_row;
        if (p != k && r < n) {                       // thread r swaps COLUMN r of rows k and p
            const double t0 = a[k][r]; a[k][r] = a[p][r]; a[p][r] = t0;
            const double t1 = inv[k][r]; inv[k][r] = inv[p][r]; inv[p][r] = t1;
        }
        __syncthreads();
        const double piv = a[k][k];
        if (r == 0) {
            logabs += log(fabs(piv));
            if (piv < 0.0) neg ^= 1;
            if (p != k) neg ^= 1;
        }
        __syncthreads();
        if (r < n) { a[k][r] /= piv; inv[k][r] /= piv; }      // thread r: column r of the pivot row (a[k][k] read above)
        __syncthreads();
        if (r < n && r != k) {
            const double f = a[r][k];
            for (int c = 0; c < n; ++c) { a[r][c] -= f * a[k][c]; inv[r][c] -= f * inv[k][c]; }
        }
        __syncthreads();
    }
    if (r < n)
        for (int c = 0; c < n; ++c) w_inv[r * n + c] = (float)inv[r][c];
    if (r == 0) logdet_w[0] = neg ? __builtin_nanf("") : (float)logabs;
}

// TRANSPOSED = false: z[o] = mask * sum_k W[o][k] x[k] ; TRANSPOSED = true (the input gradient): dx[k] = sum_o W[o][k] (dz[o] mask)
template <int V, bool TRANSPOSED>
__global__ __launch_bounds__(256) void invconv_mix_generic_kernel(const float *__restrict__ x, const float *__restrict__ mask,
                                                                  const float *__restrict__ w, const float *__restrict__ logdet_w,
                                                                  const float *__restrict__ x_len, float *__restrict__ z,
                                                                  float *__restrict__ logdet, int B, int C, int T, int N) {
    __shared__ float ws[kInvConvMaxN * kInvConvMaxN];
    for (int q = threadIdx.x; q < N * N; q += 256) ws[q] = w[q];
    __syncthreads();
    const int TV = T / V, G = C / N;
    const long n = (long)B * G * N * TV;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
        const int tv = (int)(i % TV);
        long rest = i / TV;
        const int o = (int)(rest % N);
        rest /= N;
        const int g = (int)(rest % G), b = (int)(rest / G);
        const Vec<V> mv = Vec<V>::load(mask + (long)b * T + (long)tv * V);
        Vec<V> acc = Vec<V>::zero();
        for (int k = 0; k < N; ++k) {
            const Vec<V> xv = Vec<V>::load(x + ((long)b * C + invconv_channel_rt(k, g, C, N)) * T + (long)tv * V);
            const float wk = TRANSPOSED ? ws[k * N + o] : ws[o * N + k];
#pragma unroll
            for (int j = 0; j < V; ++j) acc[j] += wk * (TRANSPOSED ? xv[j] * mv[j] : xv[j]);
        }
        if (!TRANSPOSED)
#pragma unroll
            for (int j = 0; j < V; ++j) acc[j] *= mv[j];
        acc.store(z + ((long)b * C + invconv_channel_rt(o, g, C, N)) * T + (long)tv * V);
    }
    if (logdet != nullptr && blockIdx.x == 0) {
        const float ld = logdet_w[0] * (float)(C / N);
        for (int b = threadIdx.x; b < B; b += 256) logdet[b] = ld * x_len[b];
    }
}

// dW[o][k] += sum_{b, g, t} (dz[o] mask) x[k]  (+ the log-det term): one workgroup per entry
template <int V>
__global__ __launch_bounds__(256) void invconv_dw_generic_kernel(const float *__restrict__ x, const float *__restrict__ mask,
                                                                 const float *__restrict__ w_inv, const float *__restrict__ dz,
                                                                 const float *__restrict__ dlogdet, const float *__restrict__ x_len,
                                                                 float *__restrict__ dw, int B, int C, int T, int N) {
    __shared__ float red[4];
    const int o = blockIdx.x / N, k = blockIdx.x % N;
    const int TV = T / V, G = C / N;
    const long n = (long)B * G * TV;
    float acc = 0.f;
    for (long i = threadIdx.x; i < n; i += 256) {
        const int tv = (int)(i % TV);
        const long bg = i / TV;
        const int g = (int)(bg % G), b = (int)(bg / G);
        const Vec<V> mv = Vec<V>::load(mask + (long)b * T + (long)tv * V);
        const Vec<V> gz = Vec<V>::load(dz + ((long)b * C + invconv_channel_rt(o, g, C, N)) * T + (long)tv * V);
        const Vec<V> xv = Vec<V>::load(x + ((long)b * C + invconv_channel_rt(k, g, C, N)) * T + (long)tv * V);
#pragma unroll
        for (int j = 0; j < V; ++j) acc += gz[j] * mv[j] * xv[j];
    }
    float s = block_sum_256(acc, red);
    if (threadIdx.x == 0) {
        if (dlogdet != nullptr) {
            float t = 0.f;
            for (int b = 0; b < B; ++b) t += dlogdet[b] * x_len[b];
            s += w_inv[k * N + o] * (float)(C / N) * t;  // d logdet(W) / dW = W^-T
        }
        atomicAdd(dw + o * N + k, s);
    }
}

// ------------------------------------------------------------------------------------------------------------
// ActNorm + InvConvNear fused (consecutive flows 3i, 3i+1 of every block, models.py:176-179): one pass over the tensor
//   fwd : y = (bias + exp(logs) x) mask ; z = (W y) mask ; logdet[b] = (sum(logs) + logdet_w * C/n) * x_len[b]
//   bwd : gz = dz mask ; dy = W^T gz ; dW += gz y^T ; dym = dy mask ; dx = dym exp(logs) ; dlogs += dym exp(logs) x ;
//         dbias += dym   (+ the log-det terms)                                      traffic: fwd 2X, bwd 3X  (was 4X / 6X)
// ------------------------------------------------------------------------------------------------------------
template <int N, int V, bool B16 = false>
__global__ __launch_bounds__(256) void actnorm_invconv_fwd_kernel(const void *__restrict__ x, const float *__restrict__ mask,
                                                                  const float *__restrict__ logs, const float *__restrict__ bias,
                                                                  const float *__restrict__ w, const float *__restrict__ logdet_w,
                                                                  const float *__restrict__ x_len, void *__restrict__ z,
                                                                  float *__restrict__ logdet, int B, int C, int T,
                                                                  void *__restrict__ z0h) {
    // z0h (optional): a bf16 copy (B, C/2, T) of the FIRST half of z — what the coupling's start conv reads when the
    // hidden tensors are bf16 but the flow tensor itself stays fp32
    using IO = VecIO<V, B16>;
    __shared__ float red[4];
    const int TV = T / V;
    const int G = C / N;
    const long n = (long)B * G * TV;
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i < n) {
        const int tv = (int)(i % TV);
        const long bg = i / TV;
        const int g = (int)(bg % G);
        const int b = (int)(bg / G);
        float wr[N * N];
#pragma unroll
        for (int q = 0; q < N * N; ++q) wr[q] = w[q];
        Vec<V> mv = Vec<V>::load(mask + (long)b * T + (long)tv * V);
        Vec<V> y[N];
#pragma unroll
        for (int k = 0; k < N; ++k) {
            const int ch = invconv_channel<N>(k, g, C);
            const float e = expf(logs[ch]), bi = bias[ch];
            Vec<V> xv = IO::load(x, ((long)b * C + ch) * T + (long)tv * V);
#pragma unroll
            for (int j = 0; j < V; ++j) y[k][j] = (bi + e * xv[j]) * mv[j];
        }
#pragma unroll
        for (int o = 0; o < N; ++o) {
            Vec<V> acc;
#pragma unroll
            for (int j = 0; j < V; ++j) {
                float s = 0.f;
#pragma unroll
                for (int k = 0; k < N; ++k) s += wr[o * N + k] * y[k][j];
                acc[j] = s * mv[j];
            }
            const int cho = invconv_channel<N>(o, g, C);
            IO::store(z, ((long)b * C + cho) * T + (long)tv * V, acc);
            if (z0h != nullptr && cho < C / 2) VecIO<V, true>::store(z0h, ((long)b * (C / 2) + cho) * T + (long)tv * V, acc);
        }
    }
    if (logdet != nullptr && blockIdx.x == 0) {
        float s = 0.f;
        for (int c = threadIdx.x; c < C; c += 256) s += logs[c];
        s = block_sum_256(s, red);
        const float ld = s + logdet_w[0] * (float)(C / N);
        for (int b = threadIdx.x; b < B; b += 256) logdet[b] = ld * x_len[b];
    }
}

// grid (G, slabs): every thread of a workgroup works on the same channel group, so the 2N per-channel sums and the N*N
// matrix sums reduce inside the workgroup and leave as one atomic each.  A workgroup takes `nb` consecutive items of the
// flattened (b, t/V) axis (every thread busy whatever T is); the host sizes nb for ~200 workgroups: the N*N matrix sums of
// ALL workgroups land on one cache line, where float atomics retire one instruction at a time (~7.5 ns: 640 workgroups were
// 5 us of a 17 us kernel), while fewer, longer workgroups stream worse (no atomics at all: 6.0 us with 1 040 workgroups,
// 8.4 us with 140).
//
// The NS = 2N + N*N per-thread partial sums are reduced through LDS: every thread deposits its NS values as one column of
// part[NS][256]; then a lane owns (value v, segment seg), adds its share of row v with 16-byte reads, and the segments meet
// in log2(nseg) shuffles.  (Butterflies on every value — NS * 6 ds_bpermute per wave — were 3.5 us of the kernel.)
template <int NVAL>
__device__ __forceinline__ float row_sum_256(const float *part, int pitch, int row0, int lane, int &v_out, bool &owner) {
    constexpr int NSEG = 64 / NVAL;                 // lanes per value; each adds 256 / NSEG elements
    const int v = lane % NVAL, seg = lane / NVAL;
    const float *row = part + (row0 + v) * pitch;
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < 64 / NSEG; ++i) {
        const float4 p = *reinterpret_cast<const float4 *>(row + 4 * (seg + NSEG * i));
        s += (p.x + p.y) + (p.z + p.w);
    }
#pragma unroll
    for (int off = NVAL; off < 64; off <<= 1) s += __shfl_xor(s, off, 64);
    v_out = v;
    owner = seg == 0;
    return s;
}

template <int N, int V, bool B16 = false>
__global__ __launch_bounds__(256) void actnorm_invconv_bwd_kernel(const void *__restrict__ x, const float *__restrict__ mask,
                                                                  const float *__restrict__ logs, const float *__restrict__ bias,
                                                                  const float *__restrict__ w, const float *__restrict__ w_inv,
                                                                  const void *__restrict__ dz, const float *__restrict__ dlogdet,
                                                                  const float *__restrict__ x_len, void *__restrict__ dx,
                                                                  float *__restrict__ dlogs, float *__restrict__ dbias,
                                                                  float *__restrict__ dw, int B, int C, int T, int nb) {
    using IO = VecIO<V, B16>;
    const int TV = T / V;
    const int g = blockIdx.x;
    float wr[N * N], e[N], bi[N];
#pragma unroll
    for (int q = 0; q < N * N; ++q) wr[q] = w[q];
#pragma unroll
    for (int k = 0; k < N; ++k) {
        const int ch = invconv_channel<N>(k, g, C);
        e[k] = expf(logs[ch]);
        bi[k] = bias[ch];
    }
    float aw[N * N], al[N], ab[N];
#pragma unroll
    for (int q = 0; q < N * N; ++q) aw[q] = 0.f;
#pragma unroll
    for (int k = 0; k < N; ++k) { al[k] = 0.f; ab[k] = 0.f; }
    // two items in flight per thread: the loads of item i + 1 are issued before item i is worked on (with ~1 workgroup per
    // CU a wave has nobody to hide its memory latency behind)
    const int i0 = blockIdx.y * nb, i1 = min(B * TV, i0 + nb);
    Vec<V> mv, xv[N], gz[N];
    long off_cur[N];
    auto fetch = [&](int it, Vec<V> &m_, Vec<V> *x_, Vec<V> *g_, long *off_) {
        const int b = it / TV, tv = it - b * TV;
        m_ = Vec<V>::load(mask + (long)b * T + (long)tv * V);
#pragma unroll
        for (int k = 0; k < N; ++k) {
            off_[k] = ((long)b * C + invconv_channel<N>(k, g, C)) * T + (long)tv * V;
            x_[k] = IO::load(x, off_[k]);
            g_[k] = IO::load(dz, off_[k]);
        }
    };
    int it = i0 + threadIdx.x;
    bool have = it < i1;
    if (have) fetch(it, mv, xv, gz, off_cur);
    while (have) {
        const int nx = it + 256;
        const bool have_nx = nx < i1;
        Vec<V> mv2, xv2[N], gz2[N];
        long off_nx[N];
        if (have_nx) fetch(nx, mv2, xv2, gz2, off_nx);
#pragma unroll
        for (int k = 0; k < N; ++k)
#pragma unroll
            for (int j = 0; j < V; ++j) gz[k][j] *= mv[j];
#pragma unroll
        for (int k = 0; k < N; ++k) {
            Vec<V> o;
#pragma unroll
            for (int j = 0; j < V; ++j) {
                const float yk = (bi[k] + e[k] * xv[k][j]) * mv[j];
                float dy = 0.f;
#pragma unroll
                for (int oo = 0; oo < N; ++oo) {
                    dy += wr[oo * N + k] * gz[oo][j];
                    aw[oo * N + k] += gz[oo][j] * yk;
                }
                const float dym = dy * mv[j];
                o[j] = dym * e[k];
                al[k] += dym * e[k] * xv[k][j];
                ab[k] += dym;
            }
            IO::store(dx, off_cur[k], o);
        }
        if (have_nx) {
            mv = mv2;
#pragma unroll
            for (int k = 0; k < N; ++k) { xv[k] = xv2[k]; gz[k] = gz2[k]; off_cur[k] = off_nx[k]; }
        }
        it = nx;
        have = have_nx;
    }
    // one LDS round for all 2N + N*N sums, then ONE atomic instruction per output cache line: same-line float atomics
    // retire serially in L2, so a workgroup must not send them one by one
    constexpr int NS = 2 * N + N * N;
    constexpr int RP = 260;                         // row pitch (floats): 16-byte aligned rows, 4 banks apart
    __shared__ __attribute__((aligned(16))) float part[NS * RP + 4];
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
#pragma unroll
    for (int k = 0; k < N; ++k) { part[k * RP + tid] = al[k]; part[(N + k) * RP + tid] = ab[k]; }
#pragma unroll
    for (int q = 0; q < N * N; ++q) part[(2 * N + q) * RP + tid] = aw[q];
    if (wave == 3) {                                // sum_b dlogdet[b] * x_len[b]: feeds dlogs (every channel) and dW
        float t = 0.f;
        if (dlogdet != nullptr)
            for (int b = lane; b < B; b += 64) t += dlogdet[b] * x_len[b];
        t = wave_sum(t);
        if (lane == 0) part[NS * RP] = t;
    }
    __syncthreads();
    const float tt = part[NS * RP];
    if (wave == 0) {                                // rows [0, 2N): dlogs, dbias
        int q; bool owner;
        float v = row_sum_256<2 * N>(part, RP, 0, lane, q, owner);
        if (owner) {
            if (q < N) {
                if (blockIdx.y == 0) v += tt;
                atomicAdd(dlogs + invconv_channel<N>(q, g, C), v);
            } else {
                atomicAdd(dbias + invconv_channel<N>(q - N, g, C), v);
            }
        }
    } else if (wave == 1) {                         // rows [2N, 2N + N*N): dW, one instruction for the whole matrix
        int r; bool owner;
        float v = row_sum_256<N * N>(part, RP, 2 * N, lane, r, owner);
        if (owner) {
            if (g == 0 && blockIdx.y == 0 && dlogdet != nullptr) v += w_inv[(r % N) * N + r / N] * (float)(C / N) * tt;
            atomicAdd(dw + r, v);
        }
    }
}

// ------------------------------------------------------------------------------------------------------------
// affine coupling apply
// ------------------------------------------------------------------------------------------------------------
__device__ __forceinline__ float coupling_logs(float raw, bool sig) {
    return sig ? logf(1e-6f + sigmoidf_(raw + 2.0f)) : raw;
}

// grid (chunks, B): each workgroup stays inside one utterance so logdet[b] takes one atomic per workgroup.
template <int V, bool REV, bool B16 = false>
__global__ __launch_bounds__(256) void coupling_fwd_kernel(const void *__restrict__ x, const float *__restrict__ out,
                                                           const float *__restrict__ mask, void *__restrict__ z,
                                                           float *__restrict__ logdet, int C, int T, int sig) {
    using IO = VecIO<V, B16>;
    __shared__ float red[4];
    const int b = blockIdx.y;
    const int TV = T / V;
    const int half = C / 2;
    const int items = half * TV;
    float ld = 0.f;
    for (int i = blockIdx.x * 256 + threadIdx.x; i < items; i += gridDim.x * 256) {
        const int c = i / TV, tv = i % TV;
        const long o0 = ((long)b * C + c) * T + (long)tv * V;
        const long o1 = o0 + (long)half * T;
        Vec<V> x0 = IO::load(x, o0);
        Vec<V> x1 = IO::load(x, o1);
        Vec<V> m = Vec<V>::load(out + o0);
        Vec<V> lr = Vec<V>::load(out + o1);
        Vec<V> mv = Vec<V>::load(mask + (long)b * T + (long)tv * V);
        Vec<V> z1;
#pragma unroll
        for (int j = 0; j < V; ++j) {
            const float l = coupling_logs(lr[j], sig != 0);
            if (REV) {
                z1[j] = (x1[j] - m[j]) * expf(-l) * mv[j];
            } else {
                z1[j] = (m[j] + expf(l) * x1[j]) * mv[j];
                ld += l * mv[j];
            }
        }
        IO::store(z, o0, x0);
        IO::store(z, o1, z1);
    }
    if (!REV) {
        ld = block_sum_256(ld, red);
        if (threadIdx.x == 0) atomicAdd(logdet + b, ld);
    }
}

template <int V, bool B16 = false, bool D16 = B16>      // B16: flow tensors (x, dz, dx) bf16 ; D16: dout bf16
__global__ __launch_bounds__(256) void coupling_bwd_kernel(const void *__restrict__ x, const float *__restrict__ out,
                                                           const float *__restrict__ mask, const void *__restrict__ dz,
                                                           const float *__restrict__ dlogdet, void *__restrict__ dx,
                                                           void *__restrict__ dout, int C, int T, int sig) {
    using IO = VecIO<V, B16>;
    const int b = blockIdx.y;
    const int TV = T / V;
    const int half = C / 2;
    const int items = half * TV;
    const float dld = dlogdet ? dlogdet[b] : 0.f;
    for (int i = blockIdx.x * 256 + threadIdx.x; i < items; i += gridDim.x * 256) {
        const int c = i / TV, tv = i % TV;
        const long o0 = ((long)b * C + c) * T + (long)tv * V;
        const long o1 = o0 + (long)half * T;
        Vec<V> x1 = IO::load(x, o1);
        Vec<V> lr = Vec<V>::load(out + o1);
        Vec<V> g0 = IO::load(dz, o0);
        Vec<V> g1 = IO::load(dz, o1);
        Vec<V> mv = Vec<V>::load(mask + (long)b * T + (long)tv * V);
        Vec<V> dx1, dm, dl;
#pragma unroll
        for (int j = 0; j < V; ++j) {
            const float l = coupling_logs(lr[j], sig != 0);
            const float e = expf(l);
            const float gm = g1[j] * mv[j];
            dx1[j] = gm * e;
            dm[j] = gm;
            float dlp = (gm * e * x1[j]) + dld * mv[j];      // d / d logs'
            if (sig) {
                // logs' = log(1e-6 + s), s = sigmoid(raw + 2):  dlogs'/draw = s (1 - s) / (1e-6 + s)
                const float s = sigmoidf_(lr[j] + 2.0f);
                dlp *= s * (1.0f - s) / (1e-6f + s);
            }
            dl[j] = dlp;
        }
        IO::store(dx, o0, g0);
        IO::store(dx, o1, dx1);
        VecIO<V, D16>::store(dout, o0, dm);
        VecIO<V, D16>::store(dout, o1, dl);
    }
}

// ------------------------------------------------------------------------------------------------------------
// Round 4: the affine apply of block k fused with ActNorm + InvConvNear of block k + 1 — adjacent
// element-wise passes over the same flow tensor (reference attentions.py:128-142 followed by layers.py:182-199, 238-272).
//   fwd : z = [y0 ; (m + e^logs' y1) mask]  (never written) ; y' = W ((bias + e^logs z) mask) mask
//         logdet_prev[b] += sum logs' mask ; logdet[b] = (sum logs + logdet_w C/n) x_len[b]          traffic 3X (was 5X)
//   bwd : z recomputed from (y, out) ; the ActNorm / InvConv backward of actnorm_invconv_bwd_kernel on it (dz_ai -> d z,
//         dlogs, dbias, dW) ; then the coupling's backward on d z: dy = [dz0 ; dz1 e^logs' mask], dout = [dz1 mask ; dlogs']
//                                                                                                    traffic 5X (was 7.5X)
// A group of N channels holds N/2 channels of each half (invconv_channel), so both steps are local to a thread.  fp32 tensors.
// ------------------------------------------------------------------------------------------------------------
template <int N, int V>
__global__ __launch_bounds__(256) void coupling_ai_fwd_kernel(const float *__restrict__ yp, const float *__restrict__ outp,
                                                              const float *__restrict__ mask, const float *__restrict__ logs,
                                                              const float *__restrict__ bias, const float *__restrict__ w,
                                                              const float *__restrict__ logdet_w, const float *__restrict__ x_len,
                                                              float *__restrict__ y, float *__restrict__ logdet_prev,
                                                              float *__restrict__ logdet, int B, int C, int T, int sig) {
    __shared__ float red[4];
    const int TV = T / V, G = C / N, half = C / 2;
    const int b = blockIdx.y;
    const int items = G * TV;
    float wr[N * N];
#pragma unroll
    for (int q = 0; q < N * N; ++q) wr[q] = w[q];
    float ld = 0.f;
    for (int i = blockIdx.x * 256 + threadIdx.x; i < items; i += gridDim.x * 256) {
        const int g = i / TV, tv = i - g * TV;
        const Vec<V> mv = Vec<V>::load(mask + (long)b * T + (long)tv * V);
        Vec<V> yv[N];
#pragma unroll
        for (int k = 0; k < N; ++k) {
            const int ch = invconv_channel<N>(k, g, C);
            const long o = ((long)b * C + ch) * T + (long)tv * V;
            Vec<V> z = Vec<V>::load(yp + o);
            if (k >= N / 2) {                               // second half: the affine apply of the previous block
                const Vec<V> m = Vec<V>::load(outp + o - (long)half * T);
                const Vec<V> lr = Vec<V>::load(outp + o);
#pragma unroll
                for (int j = 0; j < V; ++j) {
                    const float l = coupling_logs(lr[j], sig != 0);
                    z[j] = (m[j] + expf(l) * z[j]) * mv[j];
                    ld += l * mv[j];
                }
            }
            const float e = expf(logs[ch]), bi = bias[ch];
#pragma unroll
            for (int j = 0; j < V; ++j) yv[k][j] = (bi + e * z[j]) * mv[j];
        }
#pragma unroll
        for (int o = 0; o < N; ++o) {
            Vec<V> acc;
#pragma unroll
            for (int j = 0; j < V; ++j) {
                float s2 = 0.f;
#pragma unroll
                for (int k = 0; k < N; ++k) s2 += wr[o * N + k] * yv[k][j];
                acc[j] = s2 * mv[j];
            }
            acc.store(y + ((long)b * C + invconv_channel<N>(o, g, C)) * T + (long)tv * V);
        }
    }
    ld = block_sum_256(ld, red);
    if (threadIdx.x == 0) atomicAdd(logdet_prev + b, ld);
    if (blockIdx.x == 0 && blockIdx.y == 0) {
        __syncthreads();
        float s2 = 0.f;
        for (int c = threadIdx.x; c < C; c += 256) s2 += logs[c];
        s2 = block_sum_256(s2, red);
        const float l0 = s2 + logdet_w[0] * (float)(C / N);
        for (int bb = threadIdx.x; bb < B; bb += 256) logdet[bb] = l0 * x_len[bb];
    }
}

template <int N, int V>
__global__ __launch_bounds__(256) void coupling_ai_bwd_kernel(const float *__restrict__ yp, const float *__restrict__ outp,
                                                              const float *__restrict__ mask, const float *__restrict__ logs,
                                                              const float *__restrict__ bias, const float *__restrict__ w,
                                                              const float *__restrict__ w_inv, const float *__restrict__ dz,
                                                              const float *__restrict__ dlogdet, const float *__restrict__ x_len,
                                                              float *__restrict__ dyp, float *__restrict__ doutp,
                                                              float *__restrict__ dlogs, float *__restrict__ dbias,
                                                              float *__restrict__ dw, int B, int C, int T, int nb, int sig) {
    const int TV = T / V, half = C / 2;
    const int g = blockIdx.x;
    float wr[N * N], e[N], bi[N];
#pragma unroll
    for (int q = 0; q < N * N; ++q) wr[q] = w[q];
#pragma unroll
    for (int k = 0; k < N; ++k) {
        const int ch = invconv_channel<N>(k, g, C);
        e[k] = expf(logs[ch]);
        bi[k] = bias[ch];
    }
    float aw[N * N], al[N], ab[N];
#pragma unroll
    for (int q = 0; q < N * N; ++q) aw[q] = 0.f;
#pragma unroll
    for (int k = 0; k < N; ++k) { al[k] = 0.f; ab[k] = 0.f; }
    const int i0 = blockIdx.y * nb, i1 = min(B * TV, i0 + nb);
    for (int it = i0 + threadIdx.x; it < i1; it += 256) {
        const int b = it / TV, tv = it - b * TV;
        const Vec<V> mv = Vec<V>::load(mask + (long)b * T + (long)tv * V);
        const float dld = dlogdet ? dlogdet[b] : 0.f;
        Vec<V> xv[N], gz[N], y1[N / 2], el[N / 2], lr[N / 2];
        long off[N];
#pragma unroll
        for (int k = 0; k < N; ++k) {
            off[k] = ((long)b * C + invconv_channel<N>(k, g, C)) * T + (long)tv * V;
            xv[k] = Vec<V>::load(yp + off[k]);
            gz[k] = Vec<V>::load(dz + off[k]);
        }
#pragma unroll
        for (int k = N / 2; k < N; ++k) {                   // z1 = (m + e^logs' y1) mask, recomputed
            const Vec<V> m = Vec<V>::load(outp + off[k] - (long)half * T);
            lr[k - N / 2] = Vec<V>::load(outp + off[k]);
            y1[k - N / 2] = xv[k];
#pragma unroll
            for (int j = 0; j < V; ++j) {
                const float ee = expf(coupling_logs(lr[k - N / 2][j], sig != 0));
                el[k - N / 2][j] = ee;
                xv[k][j] = (m[j] + ee * xv[k][j]) * mv[j];
            }
        }
#pragma unroll
        for (int k = 0; k < N; ++k)
#pragma unroll
            for (int j = 0; j < V; ++j) gz[k][j] *= mv[j];
#pragma unroll
        for (int k = 0; k < N; ++k) {
            Vec<V> o;                                        // gradient of z (channel k of the group)
#pragma unroll
            for (int j = 0; j < V; ++j) {
                const float yk = (bi[k] + e[k] * xv[k][j]) * mv[j];
                float dy = 0.f;
#pragma unroll
                for (int oo = 0; oo < N; ++oo) {
                    dy += wr[oo * N + k] * gz[oo][j];
                    aw[oo * N + k] += gz[oo][j] * yk;
                }
                const float dym = dy * mv[j];
                o[j] = dym * e[k];
                al[k] += dym * e[k] * xv[k][j];
                ab[k] += dym;
            }
            if (k < N / 2) {
                o.store(dyp + off[k]);                       // first half passes through the coupling
            } else {
                Vec<V> dx1, dm, dl;
#pragma unroll
                for (int j = 0; j < V; ++j) {
                    const float gm = o[j] * mv[j];
                    const float ee = el[k - N / 2][j];
                    dx1[j] = gm * ee;
                    dm[j] = gm;
                    float dlp = gm * ee * y1[k - N / 2][j] + dld * mv[j];
                    if (sig) {
                        const float sg = sigmoidf_(lr[k - N / 2][j] + 2.0f);
                        dlp *= sg * (1.0f - sg) / (1e-6f + sg);
                    }
                    dl[j] = dlp;
                }
                dx1.store(dyp + off[k]);
                dm.store(doutp + off[k] - (long)half * T);
                dl.store(doutp + off[k]);
            }
        }
    }
    constexpr int NS = 2 * N + N * N;
    constexpr int RP = 260;
    __shared__ __attribute__((aligned(16))) float part[NS * RP + 4];
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
#pragma unroll
    for (int k = 0; k < N; ++k) { part[k * RP + tid] = al[k]; part[(N + k) * RP + tid] = ab[k]; }
#pragma unroll
    for (int q = 0; q < N * N; ++q) part[(2 * N + q) * RP + tid] = aw[q];
    if (wave == 3) {
        float t = 0.f;
        if (dlogdet != nullptr)
            for (int b = lane; b < B; b += 64) t += dlogdet[b] * x_len[b];
        t = wave_sum(t);
        if (lane == 0) part[NS * RP] = t;
    }
    __syncthreads();
    const float tt = part[NS * RP];
    if (wave == 0) {
        int q; bool owner;
        float v = row_sum_256<2 * N>(part, RP, 0, lane, q, owner);
        if (owner) {
            if (q < N) {
                if (blockIdx.y == 0) v += tt;
                atomicAdd(dlogs + invconv_channel<N>(q, g, C), v);
            } else {
                atomicAdd(dbias + invconv_channel<N>(q - N, g, C), v);
            }
        }
    } else if (wave == 1) {
        int r; bool owner;
        float v = row_sum_256<N * N>(part, RP, 2 * N, lane, r, owner);
        if (owner) {
            if (g == 0 && blockIdx.y == 0 && dlogdet != nullptr) v += w_inv[(r % N) * N + r / N] * (float)(C / N) * tt;
            atomicAdd(dw + r, v);
        }
    }
}

}  // namespace glowtts

// ================================================================================================================
// C ABI
// ================================================================================================================
using namespace glowtts;

extern "C" int glowtts_mask_len(const float *mask, float *x_len, int B, int T, glowtts_stream_t stream) {
    GLOWTTS_CHECK_ARG(mask && x_len && B >= 0 && T >= 0, "glowtts_mask_len: bad argument");
    if (B == 0) return 0;
    hipLaunchKernelGGL(mask_len_kernel, dim3(B), dim3(64), 0, (hipStream_t)stream, mask, x_len, T);
    GLOWTTS_LAUNCH_CHECK("glowtts_mask_len");
}

extern "C" int glowtts_actnorm_fwd(const float *x, const float *mask, const float *logs, const float *bias,
                                   const float *x_len, float *z, float *logdet, int B, int C, int T, int reverse,
                                   glowtts_stream_t stream) {
    GLOWTTS_CHECK_ARG(x && mask && logs && bias && z, "glowtts_actnorm_fwd: null pointer");
    GLOWTTS_CHECK_ARG(B >= 0 && C > 0 && T >= 0, "glowtts_actnorm_fwd: bad shape (%d,%d,%d)", B, C, T);
    GLOWTTS_CHECK_ARG(reverse || !logdet || x_len, "glowtts_actnorm_fwd: logdet requested without x_len");
    if ((long)B * C * T == 0) return 0;
    hipStream_t s = (hipStream_t)stream;
    const bool v4 = can_vec4(T, {x, mask, z});
    const long n = (long)B * C * (v4 ? T / 4 : T);
    const int grid = cdiv(n, 256);
    if (reverse) {
        if (v4) hipLaunchKernelGGL((actnorm_fwd_kernel<4, true>), dim3(grid), dim3(256), 0, s, x, mask, logs, bias, x_len, z, logdet, B, C, T);
        else    hipLaunchKernelGGL((actnorm_fwd_kernel<1, true>), dim3(grid), dim3(256), 0, s, x, mask, logs, bias, x_len, z, logdet, B, C, T);
    } else {
        if (v4) hipLaunchKernelGGL((actnorm_fwd_kernel<4, false>), dim3(grid), dim3(256), 0, s, x, mask, logs, bias, x_len, z, logdet, B, C, T);
        else    hipLaunchKernelGGL((actnorm_fwd_kernel<1, false>), dim3(grid), dim3(256), 0, s, x, mask, logs, bias, x_len, z, logdet, B, C, T);
    }
    GLOWTTS_LAUNCH_CHECK("glowtts_actnorm_fwd");
}

static inline int slab_size(int B, int C) {
    // enough workgroups to fill 256 CUs several times over, at least one utterance per workgroup
    int nbk = (2048 + C - 1) / C;
    if (nbk > B) nbk = B;
    if (nbk < 1) nbk = 1;
    return (B + nbk - 1) / nbk;
}

extern "C" int glowtts_actnorm_bwd(const float *x, const float *mask, const float *logs, const float *dz,
                                   const float *dlogdet, const float *x_len, float *dx, float *dlogs, float *dbias,
                                   int B, int C, int T, glowtts_stream_t stream) {
    GLOWTTS_CHECK_ARG(x && mask && logs && dz && dx && dlogs && dbias, "glowtts_actnorm_bwd: null pointer");
    GLOWTTS_CHECK_ARG(B >= 0 && C > 0 && T >= 0, "glowtts_actnorm_bwd: bad shape");
    GLOWTTS_CHECK_ARG(!dlogdet || x_len, "glowtts_actnorm_bwd: dlogdet given without x_len");
    if ((long)B * C * T == 0) return 0;
    hipStream_t s = (hipStream_t)stream;
    const int nb = slab_size(B, C);
    dim3 grid(C, (B + nb - 1) / nb);
    if (can_vec4(T, {x, mask, dz, dx}))
        hipLaunchKernelGGL((actnorm_bwd_kernel<4>), grid, dim3(256), 0, s, x, mask, logs, dz, dlogdet, x_len, dx, dlogs, dbias, B, C, T, nb);
    else
        hipLaunchKernelGGL((actnorm_bwd_kernel<1>), grid, dim3(256), 0, s, x, mask, logs, dz, dlogdet, x_len, dx, dlogs, dbias, B, C, T, nb);
    GLOWTTS_LAUNCH_CHECK("glowtts_actnorm_bwd");
}

extern "C" int glowtts_actnorm_stats(const float *x, const float *mask, float *sum_x, float *sum_x2, int B, int C,
                                     int T, glowtts_stream_t stream) {
    GLOWTTS_CHECK_ARG(x && mask && sum_x && sum_x2, "glowtts_actnorm_stats: null pointer");
    GLOWTTS_CHECK_ARG(B >= 0 && C > 0 && T >= 0, "glowtts_actnorm_stats: bad shape");
    if ((long)B * C * T == 0) return 0;
    const int nb = slab_size(B, C);
    dim3 grid(C, (B + nb - 1) / nb);
    if (can_vec4(T, {x, mask}))
        hipLaunchKernelGGL((actnorm_stats_kernel<4>), grid, dim3(256), 0, (hipStream_t)stream, x, mask, sum_x, sum_x2, B, C, T, nb);
    else
        hipLaunchKernelGGL((actnorm_stats_kernel<1>), grid, dim3(256), 0, (hipStream_t)stream, x, mask, sum_x, sum_x2, B, C, T, nb);
    GLOWTTS_LAUNCH_CHECK("glowtts_actnorm_stats");
}

extern "C" int glowtts_invconv_prepare(const float *w, float *w_inv, float *logdet_w, int n, glowtts_stream_t stream) {
    GLOWTTS_CHECK_ARG(w && w_inv && logdet_w, "glowtts_invconv_prepare: null pointer");
    GLOWTTS_CHECK_ARG(n >= 1 && n <= kInvConvMaxN, "glowtts_invconv_prepare: n_split=%d not in [1,%d]", n, kInvConvMaxN);
    if (n <= 8)
        hipLaunchKernelGGL(invconv_prepare_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, w, w_inv, logdet_w, n, nullptr, 0L);
    else
        hipLaunchKernelGGL(invconv_prepare_generic_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, w, w_inv, logdet_w, n, nullptr, 0L);
    GLOWTTS_LAUNCH_CHECK("glowtts_invconv_prepare");
}

extern "C" int glowtts_invconv_prepare_multi(const long long *w_table, float *w_inv, long out_stride, int n_problems, int n,
                                             glowtts_stream_t stream) {
    GLOWTTS_CHECK_ARG(w_table && w_inv && n_problems >= 0, "glowtts_invconv_prepare_multi: bad argument");
    GLOWTTS_CHECK_ARG(n >= 1 && n <= kInvConvMaxN, "glowtts_invconv_prepare_multi: n_split=%d not in [1,%d]", n, kInvConvMaxN);
    GLOWTTS_CHECK_ARG(out_stride >= (long)n * n + 1, "glowtts_invconv_prepare_multi: out_stride %ld < n * n + 1", out_stride);
    if (n_problems == 0) return 0;
    if (n <= 8)
        hipLaunchKernelGGL(invconv_prepare_kernel, dim3(n_problems), dim3(64), 0, (hipStream_t)stream, nullptr, w_inv, nullptr, n,
                           w_table, out_stride);
    else
        hipLaunchKernelGGL(invconv_prepare_generic_kernel, dim3(n_problems), dim3(64), 0, (hipStream_t)stream, nullptr, w_inv, nullptr,
                           n, w_table, out_stride);
    GLOWTTS_LAUNCH_CHECK("glowtts_invconv_prepare_multi");
}

#define INVCONV_DISPATCH(KERNEL, GRID, ...)                                                              \
    do {                                                                                                 \
        if (n_split == 4) {                                                                              \
            if (v4) hipLaunchKernelGGL((KERNEL<4, 4>), GRID, dim3(256), 0, s, __VA_ARGS__);              \
            else    hipLaunchKernelGGL((KERNEL<4, 1>), GRID, dim3(256), 0, s, __VA_ARGS__);              \
        } else if (n_split == 2) {                                                                       \
            if (v4) hipLaunchKernelGGL((KERNEL<2, 4>), GRID, dim3(256), 0, s, __VA_ARGS__);              \
            else    hipLaunchKernelGGL((KERNEL<2, 1>), GRID, dim3(256), 0, s, __VA_ARGS__);              \
        } else {                                                                                         \
            if (v4) hipLaunchKernelGGL((KERNEL<8, 4>), GRID, dim3(256), 0, s, __VA_ARGS__);              \
            else    hipLaunchKernelGGL((KERNEL<8, 1>), GRID, dim3(256), 0, s, __VA_ARGS__);              \
        }                                                                                                \
    } while (0)

extern "C" int glowtts_invconv_fwd(const float *x, const float *mask, const float *w, const float *logdet_w,
                                   const float *x_len, float *z, float *logdet, int B, int C, int T, int n_split,
                                   glowtts_stream_t stream) {
    GLOWTTS_CHECK_ARG(x && mask && w && z, "glowtts_invconv_fwd: null pointer");
    GLOWTTS_CHECK_ARG(n_split >= 2 && n_split % 2 == 0 && n_split <= kInvConvMaxN,
                      "glowtts_invconv_fwd: n_split=%d (supported: even values up to %d)", n_split, kInvConvMaxN);
    GLOWTTS_CHECK_ARG(B >= 0 && C > 0 && T >= 0 && C % n_split == 0, "glowtts_invconv_fwd: C=%d not divisible by n_split=%d", C, n_split);
    GLOWTTS_CHECK_ARG(!logdet || (logdet_w && x_len), "glowtts_invconv_fwd: logdet requested without logdet_w/x_len");
    if ((long)B * C * T == 0) return 0;
    hipStream_t s = (hipStream_t)stream;
    const bool v4 = can_vec4(T, {x, mask, z});
    if (n_split != 2 && n_split != 4 && n_split != 8) {          // run-time N form
        const long items = (long)B * C * (v4 ? T / 4 : T);
        int gg = cdiv(items, 256);
        if (gg > 4096) gg = 4096;
        if (v4) hipLaunchKernelGGL((invconv_mix_generic_kernel<4, false>), dim3(gg), dim3(256), 0, s, x, mask, w, logdet_w, x_len, z, logdet, B, C, T, n_split);
        else    hipLaunchKernelGGL((invconv_mix_generic_kernel<1, false>), dim3(gg), dim3(256), 0, s, x, mask, w, logdet_w, x_len, z, logdet, B, C, T, n_split);
        GLOWTTS_LAUNCH_CHECK("glowtts_invconv_fwd");
    }
    const long n = (long)B * (C / n_split) * (v4 ? T / 4 : T);
    dim3 grid(cdiv(n, 256));
    INVCONV_DISPATCH(invconv_fwd_kernel, grid, x, mask, w, logdet_w, x_len, z, logdet, B, C, T);
    GLOWTTS_LAUNCH_CHECK("glowtts_invconv_fwd");
}

extern "C" int glowtts_invconv_bwd(const float *x, const float *mask, const float *w, const float *w_inv,
                                   const float *dz, const float *dlogdet, const float *x_len, float *dx, float *dw,
                                   int B, int C, int T, int n_split, glowtts_stream_t stream) {
    GLOWTTS_CHECK_ARG(x && mask && w && dz && dx && dw, "glowtts_invconv_bwd: null pointer");
    GLOWTTS_CHECK_ARG(n_split >= 2 && n_split % 2 == 0 && n_split <= kInvConvMaxN,
                      "glowtts_invconv_bwd: n_split=%d (supported: even values up to %d)", n_split, kInvConvMaxN);
    GLOWTTS_CHECK_ARG(B >= 0 && C > 0 && T >= 0 && C % n_split == 0, "glowtts_invconv_bwd: bad shape");
    GLOWTTS_CHECK_ARG(!dlogdet || (w_inv && x_len), "glowtts_invconv_bwd: dlogdet given without w_inv/x_len");
    if ((long)B * C * T == 0) return 0;
    hipStream_t s = (hipStream_t)stream;
    const bool v4 = can_vec4(T, {x, mask, dz, dx});
    if (n_split != 2 && n_split != 4 && n_split != 8) {          // run-time N form: dx = W^T (dz mask), then dW entry by entry
        const long items = (long)B * C * (v4 ? T / 4 : T);
        int gg = cdiv(items, 256);
        if (gg > 4096) gg = 4096;
        const dim3 gw(n_split * n_split);
        if (v4) {
            hipLaunchKernelGGL((invconv_mix_generic_kernel<4, true>), dim3(gg), dim3(256), 0, s, dz, mask, w, nullptr, nullptr, dx, nullptr, B, C, T, n_split);
            hipLaunchKernelGGL((invconv_dw_generic_kernel<4>), gw, dim3(256), 0, s, x, mask, w_inv, dz, dlogdet, x_len, dw, B, C, T, n_split);
        } else {
            hipLaunchKernelGGL((invconv_mix_generic_kernel<1, true>), dim3(gg), dim3(256), 0, s, dz, mask, w, nullptr, nullptr, dx, nullptr, B, C, T, n_split);
            hipLaunchKernelGGL((invconv_dw_generic_kernel<1>), gw, dim3(256), 0, s, x, mask, w_inv, dz, dlogdet, x_len, dw, B, C, T, n_split);
        }
        GLOWTTS_LAUNCH_CHECK("glowtts_invconv_bwd");
    }
    const long n = (long)B * (C / n_split) * (v4 ? T / 4 : T);
    int g = cdiv(n, 256);
    if (g > 1024) g = 1024;
    dim3 grid(g);
    INVCONV_DISPATCH(invconv_bwd_kernel, grid, x, mask, w, w_inv, dz, dlogdet, x_len, dx, dw, B, C, T);
    GLOWTTS_LAUNCH_CHECK("glowtts_invconv_bwd");
}

// `_io` forms: io = 1 -> the flow tensors (x, z / dz, dx, and dout) are bf16 in HBM; (m, logs) = `out`, masks, log-dets
// and all arithmetic stay fp32
extern "C" int glowtts_coupling_fwd_io(const void *x, const float *out, const float *mask, void *z, float *logdet,
                                       int B, int C, int T, int sigmoid_scale, int reverse, int io, glowtts_stream_t stream) {
    GLOWTTS_CHECK_ARG(x && out && mask && z, "glowtts_coupling_fwd: null pointer");
    GLOWTTS_CHECK_ARG(reverse || logdet, "glowtts_coupling_fwd: forward needs logdet");
    GLOWTTS_CHECK_ARG(B >= 0 && C > 0 && (C % 2) == 0 && T >= 0, "glowtts_coupling_fwd: bad shape");
    if ((long)B * C * T == 0) return 0;
    hipStream_t s = (hipStream_t)stream;
    const bool v4 = can_vec4(T, {x, out, mask, z});
    const long items = (long)(C / 2) * (v4 ? T / 4 : T);
    int gx = cdiv(items, 256);
    const int gmax = reverse ? 64 : 16;          // forward: one logdet[b] atomic per workgroup, same cache line for 16 b
    if (gx > gmax) gx = gmax;
    dim3 grid(gx, B);
#define GLOWTTS_CPL(REV, B16)                                                                                                      \
    do {                                                                                                                            \
        if (v4) hipLaunchKernelGGL((coupling_fwd_kernel<4, REV, B16>), grid, dim3(256), 0, s, x, out, mask, z, logdet, C, T, sigmoid_scale); \
        else    hipLaunchKernelGGL((coupling_fwd_kernel<1, REV, B16>), grid, dim3(256), 0, s, x, out, mask, z, logdet, C, T, sigmoid_scale); \
    } while (0)
    if (io) { if (reverse) GLOWTTS_CPL(true, true); else GLOWTTS_CPL(false, true); }
    else    { if (reverse) GLOWTTS_CPL(true, false); else GLOWTTS_CPL(false, false); }
#undef GLOWTTS_CPL
    GLOWTTS_LAUNCH_CHECK("glowtts_coupling_fwd");
}

extern "C" int glowtts_coupling_fwd(const float *x, const float *out, const float *mask, float *z, float *logdet,
                                    int B, int C, int T, int sigmoid_scale, int reverse, glowtts_stream_t stream) {
    return glowtts_coupling_fwd_io(x, out, mask, z, logdet, B, C, T, sigmoid_scale, reverse, 0, stream);
}

extern "C" int glowtts_coupling_bwd_io(const void *x, const float *out, const float *mask, const void *dz,
                                       const float *dlogdet, void *dx, void *dout, int B, int C, int T,
                                       int sigmoid_scale, int io, int io_dout, glowtts_stream_t stream) {
    GLOWTTS_CHECK_ARG(x && out && mask && dz && dx && dout, "glowtts_coupling_bwd: null pointer");
    GLOWTTS_CHECK_ARG(B >= 0 && C > 0 && (C % 2) == 0 && T >= 0, "glowtts_coupling_bwd: bad shape");
    if ((long)B * C * T == 0) return 0;
    hipStream_t s = (hipStream_t)stream;
    const bool v4 = can_vec4(T, {x, out, mask, dz, dx, dout});
    const long items = (long)(C / 2) * (v4 ? T / 4 : T);
    int gx = cdiv(items, 256);
    if (gx > 64) gx = 64;
    dim3 grid(gx, B);
    GLOWTTS_CHECK_ARG(!io || io_dout, "glowtts_coupling_bwd: bf16 flow tensors go with a bf16 dout");
#define GLOWTTS_CPB(B16, D16)                                                                                                       \
    do {                                                                                                                             \
        if (v4) hipLaunchKernelGGL((coupling_bwd_kernel<4, B16, D16>), grid, dim3(256), 0, s, x, out, mask, dz, dlogdet, dx, dout, C, T, sigmoid_scale); \
        else    hipLaunchKernelGGL((coupling_bwd_kernel<1, B16, D16>), grid, dim3(256), 0, s, x, out, mask, dz, dlogdet, dx, dout, C, T, sigmoid_scale); \
    } while (0)
    if (io) GLOWTTS_CPB(true, true);
    else if (io_dout) GLOWTTS_CPB(false, true);
    else GLOWTTS_CPB(false, false);
#undef GLOWTTS_CPB
    GLOWTTS_LAUNCH_CHECK("glowtts_coupling_bwd");
}

extern "C" int glowtts_coupling_bwd(const float *x, const float *out, const float *mask, const float *dz,
                                    const float *dlogdet, float *dx, float *dout, int B, int C, int T,
                                    int sigmoid_scale, glowtts_stream_t stream) {
    return glowtts_coupling_bwd_io(x, out, mask, dz, dlogdet, dx, dout, B, C, T, sigmoid_scale, 0, 0, stream);
}

#define FUSED_DISPATCH_T(KERNEL, B16, GRID, ...)                                                        \
    do {                                                                                                \
        if (n_split == 4) {                                                                             \
            if (v4) hipLaunchKernelGGL((KERNEL<4, 4, B16>), GRID, dim3(256), 0, s, __VA_ARGS__);        \
            else    hipLaunchKernelGGL((KERNEL<4, 1, B16>), GRID, dim3(256), 0, s, __VA_ARGS__);        \
        } else {                                                                                        \
            if (v4) hipLaunchKernelGGL((KERNEL<2, 4, B16>), GRID, dim3(256), 0, s, __VA_ARGS__);        \
            else    hipLaunchKernelGGL((KERNEL<2, 1, B16>), GRID, dim3(256), 0, s, __VA_ARGS__);        \
        }                                                                                               \
    } while (0)
#define FUSED_DISPATCH(KERNEL, GRID, ...)                                                               \
    do {                                                                                                \
        if (io) FUSED_DISPATCH_T(KERNEL, true, GRID, __VA_ARGS__);                                      \
        else    FUSED_DISPATCH_T(KERNEL, false, GRID, __VA_ARGS__);                                     \
    } while (0)

extern "C" int glowtts_actnorm_invconv_fwd_io(const void *x, const float *mask, const float *logs, const float *bias,
                                              const float *w, const float *logdet_w, const float *x_len, void *z,
                                              float *logdet, void *z0h, int B, int C, int T, int n_split, int io,
                                              glowtts_stream_t stream) {
    GLOWTTS_CHECK_ARG(x && mask && logs && bias && w && z, "glowtts_actnorm_invconv_fwd: null pointer");
    GLOWTTS_CHECK_ARG(n_split == 2 || n_split == 4, "glowtts_actnorm_invconv_fwd: n_split=%d (fused path: 2 or 4)", n_split);
    GLOWTTS_CHECK_ARG(B >= 0 && C > 0 && T >= 0 && C % n_split == 0, "glowtts_actnorm_invconv_fwd: bad shape");
    GLOWTTS_CHECK_ARG(!logdet || (logdet_w && x_len), "glowtts_actnorm_invconv_fwd: logdet requested without logdet_w/x_len");
    if ((long)B * C * T == 0) return 0;
    hipStream_t s = (hipStream_t)stream;
    const bool v4 = can_vec4(T, {x, mask, z});
    const long n = (long)B * (C / n_split) * (v4 ? T / 4 : T);
    dim3 grid(cdiv(n, 256));
    FUSED_DISPATCH(actnorm_invconv_fwd_kernel, grid, x, mask, logs, bias, w, logdet_w, x_len, z, logdet, B, C, T, z0h);
    GLOWTTS_LAUNCH_CHECK("glowtts_actnorm_invconv_fwd");
}

extern "C" int glowtts_actnorm_invconv_fwd(const float *x, const float *mask, const float *logs, const float *bias,
                                           const float *w, const float *logdet_w, const float *x_len, float *z,
                                           float *logdet, int B, int C, int T, int n_split, glowtts_stream_t stream) {
    return glowtts_actnorm_invconv_fwd_io(x, mask, logs, bias, w, logdet_w, x_len, z, logdet, nullptr, B, C, T, n_split, 0, stream);
}

extern "C" int glowtts_actnorm_invconv_bwd_io(const void *x, const float *mask, const float *logs, const float *bias,
                                              const float *w, const float *w_inv, const void *dz, const float *dlogdet,
                                              const float *x_len, void *dx, float *dlogs, float *dbias, float *dw, int B,
                                              int C, int T, int n_split, int io, glowtts_stream_t stream) {
    GLOWTTS_CHECK_ARG(x && mask && logs && bias && w && dz && dx && dlogs && dbias && dw, "glowtts_actnorm_invconv_bwd: null pointer");
    GLOWTTS_CHECK_ARG(n_split == 2 || n_split == 4, "glowtts_actnorm_invconv_bwd: n_split=%d (fused path: 2 or 4)", n_split);
    GLOWTTS_CHECK_ARG(B >= 0 && C > 0 && T >= 0 && C % n_split == 0, "glowtts_actnorm_invconv_bwd: bad shape");
    GLOWTTS_CHECK_ARG(!dlogdet || (w_inv && x_len), "glowtts_actnorm_invconv_bwd: dlogdet given without w_inv/x_len");
    if ((long)B * C * T == 0) return 0;
    hipStream_t s = (hipStream_t)stream;
    const bool v4 = can_vec4(T, {x, mask, dz, dx});
    const int G = C / n_split;
    const long items = (long)B * (v4 ? T / 4 : T);                       // per channel group
    long slabs = (200 + G - 1) / G;     // ~200 workgroups (see the kernel's comment; 80 / 120 / 160 / 200 / 288 / 400 / 520 at
                                        // B=32, C=160, T=400: 10.9 / 9.5 / 9.1 / 8.9 / 9.6 / 10.6 / 11.7 us)
    if (slabs > (items + 255) / 256) slabs = (items + 255) / 256;
    if (slabs < 1) slabs = 1;
    const int nb = (int)((items + slabs - 1) / slabs);
    dim3 grid(G, (unsigned)((items + nb - 1) / nb));
    FUSED_DISPATCH(actnorm_invconv_bwd_kernel, grid, x, mask, logs, bias, w, w_inv, dz, dlogdet, x_len, dx, dlogs, dbias, dw, B, C, T, nb);
    GLOWTTS_LAUNCH_CHECK("glowtts_actnorm_invconv_bwd");
}

extern "C" int glowtts_actnorm_invconv_bwd(const float *x, const float *mask, const float *logs, const float *bias,
                                           const float *w, const float *w_inv, const float *dz, const float *dlogdet,
                                           const float *x_len, float *dx, float *dlogs, float *dbias, float *dw, int B,
                                           int C, int T, int n_split, glowtts_stream_t stream) {
    return glowtts_actnorm_invconv_bwd_io(x, mask, logs, bias, w, w_inv, dz, dlogdet, x_len, dx, dlogs, dbias, dw, B, C, T,
                                          n_split, 0, stream);
}

// ---- coupling(k) fused with ActNorm + InvConvNear (k + 1): see coupling_ai_fwd_kernel ---------------------------------------
extern "C" int glowtts_coupling_actnorm_invconv_fwd(const float *y_prev, const float *out_prev, const float *mask, const float *logs,
                                                    const float *bias, const float *w, const float *logdet_w, const float *x_len,
                                                    float *y, float *logdet_prev, float *logdet, int B, int C, int T, int n_split,
                                                    int sigmoid_scale, glowtts_stream_t stream) {
    GLOWTTS_CHECK_ARG(y_prev && out_prev && mask && logs && bias && w && logdet_w && x_len && y && logdet_prev && logdet,
                      "glowtts_coupling_actnorm_invconv_fwd: null pointer");
    GLOWTTS_CHECK_ARG(n_split == 2 || n_split == 4, "glowtts_coupling_actnorm_invconv_fwd: n_split=%d (fused path: 2 or 4)", n_split);
    GLOWTTS_CHECK_ARG(B >= 0 && C > 0 && T >= 0 && C % n_split == 0 && C % 2 == 0, "glowtts_coupling_actnorm_invconv_fwd: bad shape");
    if ((long)B * C * T == 0) return 0;
    hipStream_t s = (hipStream_t)stream;
    const bool v4 = can_vec4(T, {y_prev, out_prev, mask, y});
    const long items = (long)(C / n_split) * (v4 ? T / 4 : T);
    int gx = cdiv(items, 256);
    if (gx > 8) gx = 8;                                 // one logdet_prev[b] atomic per workgroup; 8 x B workgroups of ~2 items per
                                                        // thread: 7.3 us at config 2 where 16 x B (one item per thread) took 8.9
    dim3 grid(gx, B);
#define GLOWTTS_CAF(NN)                                                                                                          \
    do {                                                                                                                          \
        if (v4) hipLaunchKernelGGL((coupling_ai_fwd_kernel<NN, 4>), grid, dim3(256), 0, s, y_prev, out_prev, mask, logs, bias, w, \
                                   logdet_w, x_len, y, logdet_prev, logdet, B, C, T, sigmoid_scale);                              \
        else    hipLaunchKernelGGL((coupling_ai_fwd_kernel<NN, 1>), grid, dim3(256), 0, s, y_prev, out_prev, mask, logs, bias, w, \
                                   logdet_w, x_len, y, logdet_prev, logdet, B, C, T, sigmoid_scale);                              \
    } while (0)
    if (n_split == 4) GLOWTTS_CAF(4); else GLOWTTS_CAF(2);
#undef GLOWTTS_CAF
    GLOWTTS_LAUNCH_CHECK("glowtts_coupling_actnorm_invconv_fwd");
}

extern "C" int glowtts_coupling_actnorm_invconv_bwd(const float *y_prev, const float *out_prev, const float *mask, const float *logs,
                                                    const float *bias, const float *w, const float *w_inv, const float *dz,
                                                    const float *dlogdet, const float *x_len, float *dy_prev, float *dout_prev,
                                                    float *dlogs, float *dbias, float *dw, int B, int C, int T, int n_split,
                                                    int sigmoid_scale, glowtts_stream_t stream) {
    GLOWTTS_CHECK_ARG(y_prev && out_prev && mask && logs && bias && w && dz && dy_prev && dout_prev && dlogs && dbias && dw,
                      "glowtts_coupling_actnorm_invconv_bwd: null pointer");
    GLOWTTS_CHECK_ARG(n_split == 2 || n_split == 4, "glowtts_coupling_actnorm_invconv_bwd: n_split=%d (fused path: 2 or 4)", n_split);
    GLOWTTS_CHECK_ARG(B >= 0 && C > 0 && T >= 0 && C % n_split == 0 && C % 2 == 0, "glowtts_coupling_actnorm_invconv_bwd: bad shape");
    GLOWTTS_CHECK_ARG(!dlogdet || (w_inv && x_len), "glowtts_coupling_actnorm_invconv_bwd: dlogdet given without w_inv/x_len");
    if ((long)B * C * T == 0) return 0;
    hipStream_t s = (hipStream_t)stream;
    const bool v4 = can_vec4(T, {y_prev, out_prev, mask, dz, dy_prev, dout_prev});
    const int G = C / n_split;
    const int items = B * (v4 ? T / 4 : T);
    int slabs = (200 + G - 1) / G;                      // (120 .. 800 workgroups measured: 200 is the fastest, 13.2 us at config 2)
    const int max_slabs = (items + 255) / 256;
    if (slabs > max_slabs) slabs = max_slabs;
    if (slabs < 1) slabs = 1;
    const int nb = (items + slabs - 1) / slabs;
    dim3 grid(G, (items + nb - 1) / nb);
#define GLOWTTS_CAB(NN)                                                                                                            \
    do {                                                                                                                            \
        if (v4) hipLaunchKernelGGL((coupling_ai_bwd_kernel<NN, 4>), grid, dim3(256), 0, s, y_prev, out_prev, mask, logs, bias, w, w_inv, \
                                   dz, dlogdet, x_len, dy_prev, dout_prev, dlogs, dbias, dw, B, C, T, nb, sigmoid_scale);             \
        else    hipLaunchKernelGGL((coupling_ai_bwd_kernel<NN, 1>), grid, dim3(256), 0, s, y_prev, out_prev, mask, logs, bias, w, w_inv, \
                                   dz, dlogdet, x_len, dy_prev, dout_prev, dlogs, dbias, dw, B, C, T, nb, sigmoid_scale);             \
    } while (0)
    if (n_split == 4) GLOWTTS_CAB(4); else GLOWTTS_CAB(2);
#undef GLOWTTS_CAB
    GLOWTTS_LAUNCH_CHECK("glowtts_coupling_actnorm_invconv_bwd");
}
