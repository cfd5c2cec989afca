// attention_long.hip — relative-position self-attention beyond the register-strip kernels' 512 tokens (round 4).
//
// The q-block kernels of attention.hip keep a 64-query x T score strip on the chip (registers, LDS up to T = 256); the reference
// (attentions.py:204-264) has no length limit.  Past 512 tokens the same arithmetic runs as a few plain tiled kernels through the
// (B, h, T, T) matrices the backward needs anyway (p_attn is an output of the module, ds a workspace of the backward) — no
// MFMA, no strip: this path exists so that a long utterance is served by hand-written HIP and not by a framework composition; the
// model's lengths (T_text <= 240 in every BASELINE configuration) never take it.
//   forward : scores (64 x 64 tiles, fp32 FMA from LDS) -> p ; row softmax in place ; O = Pd V + band(Pd) E_v
//   backward: dPd = (dO^T V + band) keep/scale -> ds ; rows: dS = P (dPd - sum P dPd) scale in place ; dQ = dS K + band(dS) E_k ;
//             dK, dV by attn_dkv_kernel (attention.hip, no length limit) ; dE_k, dE_v by a diagonal-walking kernel.
// Same definitions as attention.hip: keep(i, j) = mask_i mask_j != 0 and |j - i| <= block_len ; masked scores are -1e4 ; the
// stored p is the softmax BEFORE dropout ; Pd = p * keep_byte * drop_scale.
#include "common.hpp"

namespace glowtts {

struct AttnLongParams {
    const float *a, *b1, *e1;      // scores: rows A (B, C, T), columns B1 (B, C, T), band table E1 (n_rel, 2w+1, dk) or null
    const float *mask;             // (B, T)
    const unsigned char *drop;     // (B, h, T, T) keep bytes or null
    float *mat;                    // (B, h, T, T): MODE 0 scores -> p ; MODE 1 dPd -> ds
    int H, T, dk, w, block_len, e_hs, mode;
    float scale, drop_scale;
};

// mat[i][j] = sum_d A[d][i] B1[d][j] (+ sum_d A[d][i] E1[j - i + w][d] on the band), then the mode's element-wise tail
__global__ __launch_bounds__(256) void attn_long_scores_kernel(AttnLongParams p) {
    __shared__ float As[16][64 + 1], Bs[16][64 + 1];
    const int T = p.T, dk = p.dk;
    const int j0 = blockIdx.x * 64, i0 = blockIdx.y * 64, bh = blockIdx.z, b = bh / p.H, h = bh - b * p.H;
    const long cbase = ((long)b * p.H + h) * dk;
    const float *Ag = p.a + cbase * T, *Bg = p.b1 + cbase * T;
    const int tx = threadIdx.x & 15, ty = threadIdx.x >> 4;
    float acc[4][4] = {};
    for (int d0 = 0; d0 < dk; d0 += 16) {
        for (int idx = threadIdx.x; idx < 16 * 64; idx += 256) {
            const int d = idx >> 6, c = idx & 63;
            As[d][c] = (d0 + d < dk && i0 + c < T) ? Ag[(long)(d0 + d) * T + i0 + c] : 0.f;
            Bs[d][c] = (d0 + d < dk && j0 + c < T) ? Bg[(long)(d0 + d) * T + j0 + c] : 0.f;
        }
        __syncthreads();
#pragma unroll
        for (int d = 0; d < 16; ++d) {
            float av[4], bv[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) { av[u] = As[d][ty * 4 + u]; bv[u] = Bs[d][tx * 4 + u]; }
#pragma unroll
            for (int u = 0; u < 4; ++u)
#pragma unroll
                for (int v = 0; v < 4; ++v) acc[u][v] += av[u] * bv[v];
        }
        __syncthreads();
    }
    const float *mk = p.mask + (long)b * T;
    const long pbase = (long)bh * T * T;
    const int bl = p.block_len < 0 ? (1 << 30) : p.block_len;
#pragma unroll
    for (int u = 0; u < 4; ++u) {
        const int i = i0 + ty * 4 + u;
        if (i >= T) continue;
#pragma unroll
        for (int v = 0; v < 4; ++v) {
            const int j = j0 + tx * 4 + v;
            if (j >= T) continue;
            float s = acc[u][v];
            const int r = j - i + p.w;
            if (p.e1 != nullptr && p.w >= 0 && r >= 0 && r <= 2 * p.w) {       // the band: a dk-long dot product per element
                const float *e = p.e1 + (long)h * p.e_hs + (long)r * dk;
                float t = 0.f;
                for (int d = 0; d < dk; ++d) t += Ag[(long)d * T + i] * e[d];
                s += t;
            }
            const long o = pbase + (long)i * T + j;
            if (p.mode == 0) {
                const bool keep = (mk[i] * mk[j] != 0.f) && (abs(j - i) <= bl);
                p.mat[o] = keep ? s * p.scale : -1e4f;
            } else {
                if (p.drop) s = p.drop[o] ? s * p.drop_scale : 0.f;
                p.mat[o] = s;
            }
        }
    }
}

// one wave per row.  MODE 0: softmax in place.  MODE 1: ds = keep ? P (dPd - sum_j P dPd) scale : 0 in place (P from `prob`).
__global__ __launch_bounds__(256) void attn_long_rows_kernel(float *__restrict__ mat, const float *__restrict__ prob,
                                                             const float *__restrict__ mask, int H, int T, int block_len,
                                                             float scale, int mode, long n_rows) {
    const long row = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= n_rows) return;
    const int lane = threadIdx.x & 63;
    const int i = (int)(row % T);
    const long bh = row / T;
    const int b = (int)(bh / H);
    float *m = mat + row * T;
    if (mode == 0) {
        float mx = -3.0e38f;
        for (int j = lane; j < T; j += 64) mx = fmaxf(mx, m[j]);
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o, 64));
        float sum = 0.f;
        for (int j = lane; j < T; j += 64) {
            const float e = __expf(m[j] - mx);
            m[j] = e;
            sum += e;
        }
        sum = wave_sum(sum);
        const float inv = 1.0f / sum;
        for (int j = lane; j < T; j += 64) m[j] *= inv;
    } else {
        const float *pr = prob + row * T;
        const float *mk = mask + (long)b * T;
        const int bl = block_len < 0 ? (1 << 30) : block_len;
        float dot = 0.f;
        for (int j = lane; j < T; j += 64) dot += pr[j] * m[j];
        dot = wave_sum(dot);
        const float mi = mk[i];
        for (int j = lane; j < T; j += 64) {
            const bool keep = (mi * mk[j] != 0.f) && (abs(j - i) <= bl);
            m[j] = keep ? pr[j] * (m[j] - dot) * scale : 0.f;
        }
    }
}

// out[d][i] = sum_j M[i][j] B2[d][j] + sum_r M[i][i + r - w] E2[r][d] ; M = mat (x keep bytes x drop_scale when drop != null)
__global__ __launch_bounds__(256) void attn_long_apply_kernel(const float *__restrict__ mat, const unsigned char *__restrict__ drop,
                                                              float drop_scale, const float *__restrict__ b2,
                                                              const float *__restrict__ e2, float *__restrict__ out, int H, int T,
                                                              int dk, int w, int e_hs) {
    __shared__ float Ms[64][16 + 1], Vs[64][16 + 1];
    const int i0 = blockIdx.x * 64, d0 = blockIdx.y * 64, bh = blockIdx.z, b = bh / H, h = bh - b * H;
    const long cbase = ((long)b * H + h) * dk;
    const long pbase = (long)bh * T * T;
    const int tx = threadIdx.x & 15, ty = threadIdx.x >> 4;       // tx: 4 queries, ty: 4 channels
    float acc[4][4] = {};                                          // [channel][query]
    for (int j0 = 0; j0 < T; j0 += 16) {
        for (int idx = threadIdx.x; idx < 64 * 16; idx += 256) {
            const int r = idx >> 4, c = idx & 15;                  // row r (query / channel), column c (key)
            float mv = 0.f, vv = 0.f;
            if (i0 + r < T && j0 + c < T) {
                const long o = pbase + (long)(i0 + r) * T + j0 + c;
                mv = mat[o];
                if (drop) mv = drop[o] ? mv * drop_scale : 0.f;
            }
            if (d0 + r < dk && j0 + c < T) vv = b2[(cbase + d0 + r) * T + j0 + c];
            Ms[r][c] = mv;
            Vs[r][c] = vv;
        }
        __syncthreads();
#pragma unroll
        for (int c = 0; c < 16; ++c) {
            float mv[4], vv[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) { mv[u] = Ms[tx * 4 + u][c]; vv[u] = Vs[ty * 4 + u][c]; }
#pragma unroll
            for (int u = 0; u < 4; ++u)
#pragma unroll
                for (int v = 0; v < 4; ++v) acc[u][v] += vv[u] * mv[v];
        }
        __syncthreads();
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) {
        const int d = d0 + ty * 4 + u;
        if (d >= dk) continue;
#pragma unroll
        for (int v = 0; v < 4; ++v) {
            const int i = i0 + tx * 4 + v;
            if (i >= T) continue;
            float s = acc[u][v];
            if (e2 != nullptr && w >= 0) {
                for (int r = 0; r <= 2 * w; ++r) {
                    const int j = i + r - w;
                    if (j < 0 || j >= T) continue;
                    const long o = pbase + (long)i * T + j;
                    float mv = mat[o];
                    if (drop) mv = drop[o] ? mv * drop_scale : 0.f;
                    s += mv * e2[(long)h * e_hs + (long)r * dk + d];
                }
            }
            out[(cbase + d) * T + i] = s;
        }
    }
}

// dE[r][d] += sum_i M[i][i + r - w] A[d][i]: workgroup = (utterance x head, r); the diagonal in LDS, one block sum per channel
__global__ __launch_bounds__(256) void attn_long_relgrad_kernel(const float *__restrict__ mat, const unsigned char *__restrict__ drop,
                                                                float drop_scale, const float *__restrict__ a, float *__restrict__ de,
                                                                int H, int T, int dk, int w, int e_hs) {
    extern __shared__ __align__(16) float diag[];                 // [T]
    __shared__ float red[4];
    const int bh = blockIdx.x, r = blockIdx.y, b = bh / H, h = bh - b * H;
    const long pbase = (long)bh * T * T, cbase = ((long)b * H + h) * dk;
    for (int i = threadIdx.x; i < T; i += 256) {
        const int j = i + r - w;
        float v = 0.f;
        if (j >= 0 && j < T) {
            const long o = pbase + (long)i * T + j;
            v = mat[o];
            if (drop) v = drop[o] ? v * drop_scale : 0.f;
        }
        diag[i] = v;
    }
    __syncthreads();
    for (int d = 0; d < dk; ++d) {
        float s = 0.f;
        for (int i = threadIdx.x; i < T; i += 256) s += diag[i] * a[(cbase + d) * T + i];
        s = block_sum_256(s, red);
        if (threadIdx.x == 0) atomicAdd(de + (long)h * e_hs + (long)r * dk + d, s);
        __syncthreads();
    }
}

// host side, called by attention.hip's entry points when T exceeds the strip kernels' limit
int attn_long_forward(const float *q, const float *k, const float *v, const float *emb_k, const float *emb_v, const float *mask,
                      const unsigned char *drop, float drop_scale, float *p_attn, float *out, int B, int H, int T, int dk, int w,
                      int e_hs, int block_len, float scale, hipStream_t s) {
    AttnLongParams p{};
    p.a = q; p.b1 = k; p.e1 = emb_k; p.mask = mask; p.drop = nullptr; p.mat = p_attn; p.H = H; p.T = T; p.dk = dk; p.w = emb_k ? w : -1;
    p.block_len = block_len; p.e_hs = e_hs; p.mode = 0; p.scale = scale; p.drop_scale = drop_scale;
    const int nt = (T + 63) / 64;
    hipLaunchKernelGGL(attn_long_scores_kernel, dim3(nt, nt, B * H), dim3(256), 0, s, p);
    const long rows = (long)B * H * T;
    hipLaunchKernelGGL(attn_long_rows_kernel, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, s, p_attn, nullptr, mask, H, T, block_len,
                       scale, 0, rows);
    hipLaunchKernelGGL(attn_long_apply_kernel, dim3(nt, (dk + 63) / 64, B * H), dim3(256), 0, s, p_attn, drop, drop_scale, v,
                       emb_k ? emb_v : nullptr, out, H, T, dk, emb_k ? w : -1, e_hs);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) { set_error("glowtts_rel_attn_fwd (long form): launch failed: %s", hipGetErrorString(e)); return (int)e; }
    return 0;
}

// first half of the backward for T beyond the strip kernels: ds and dq (dK, dV and, for T <= its LDS limit, the embedding gradients
// stay with attention.hip's kernels); the embedding gradients here when `relgrad` is set
int attn_long_backward(const float *dout, const float *q, const float *k, const float *v, const float *emb_k, const float *emb_v,
                       const float *mask, const unsigned char *drop, float drop_scale, const float *p_attn, float *ds, float *dq,
                       float *demb_k, float *demb_v, int B, int H, int T, int dk, int w, int e_hs, int block_len, float scale,
                       bool relgrad, hipStream_t s) {
    AttnLongParams p{};
    p.a = dout; p.b1 = v; p.e1 = emb_k ? emb_v : nullptr; p.mask = mask; p.drop = drop; p.mat = ds; p.H = H; p.T = T; p.dk = dk;
    p.w = emb_k ? w : -1; p.block_len = block_len; p.e_hs = e_hs; p.mode = 1; p.scale = scale; p.drop_scale = drop_scale;
    const int nt = (T + 63) / 64;
    const long rows = (long)B * H * T;
    if (emb_k && relgrad)        // dE_v from Pd and dO: before ds is needed (p_attn and the keep bytes are inputs)
        hipLaunchKernelGGL(attn_long_relgrad_kernel, dim3(B * H, 2 * w + 1), dim3(256), (size_t)T * sizeof(float), s, p_attn, drop,
                           drop_scale, dout, demb_v, H, T, dk, w, e_hs);
    hipLaunchKernelGGL(attn_long_scores_kernel, dim3(nt, nt, B * H), dim3(256), 0, s, p);
    hipLaunchKernelGGL(attn_long_rows_kernel, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, s, ds, p_attn, mask, H, T, block_len, scale,
                       1, rows);
    hipLaunchKernelGGL(attn_long_apply_kernel, dim3(nt, (dk + 63) / 64, B * H), dim3(256), 0, s, ds, nullptr, 1.f, k,
                       emb_k ? emb_k : nullptr, dq, H, T, dk, emb_k ? w : -1, e_hs);
    if (emb_k && relgrad)        // dE_k from dS and Q
        hipLaunchKernelGGL(attn_long_relgrad_kernel, dim3(B * H, 2 * w + 1), dim3(256), (size_t)T * sizeof(float), s, ds, nullptr, 1.f, q,
                           demb_k, H, T, dk, w, e_hs);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) { set_error("glowtts_rel_attn_bwd (long form): launch failed: %s", hipGetErrorString(e)); return (int)e; }
    return 0;
}

}  // namespace glowtts
