// wn.hip — the elementwise pieces of the WaveNet-style coupling network (WN, layers.py:138-162) for gfx950:
//   gate      acts = tanh(a[:, :H] + g) * sigmoid(a[:, H:] + g)            (utils.py:31-38), forward + backward
//   res_skip  x <- (x + rs[:, :H]) * mask ; skip <- skip + rs[:, H:]        (layers.py:157-162), forward + backward
// HBM-bound: gate fwd 3H e, gate bwd 5H e, res_skip fwd 5H e bytes per squeezed column (e = 4).  The dense
// contractions of WN (dilated k-tap conv, 1x1 res/skip conv) are not here.
#include "common.hpp"

namespace glowtts {

// thread -> (b, c < H, t-vector)
template <int V>
__global__ __launch_bounds__(256) void gate_fwd_kernel(const float *__restrict__ a, const float *__restrict__ g,
                                                       float *__restrict__ acts, int B, int H, int T) {
    const int TV = T / V;
    const long n = (long)B * H * TV;
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const int tv = (int)(i % TV);
    const long row = i / TV;
    const int c = (int)(row % H);
    const int b = (int)(row / H);
    const long oa = ((long)b * 2 * H + c) * T + (long)tv * V;
    Vec<V> at = Vec<V>::load(a + oa);
    Vec<V> as = Vec<V>::load(a + oa + (long)H * T);
    const float gt = g ? g[(long)b * 2 * H + c] : 0.f;
    const float gs = g ? g[(long)b * 2 * H + H + c] : 0.f;
    Vec<V> o;
#pragma unroll
    for (int j = 0; j < V; ++j) o[j] = tanhf(at[j] + gt) * sigmoidf_(as[j] + gs);
    o.store(acts + row * T + (long)tv * V);
}

template <int V>
__global__ __launch_bounds__(256) void gate_bwd_kernel(const float *__restrict__ a, const float *__restrict__ g,
                                                       const float *__restrict__ dacts, float *__restrict__ da, int B,
                                                       int H, int T) {
    const int TV = T / V;
    const long n = (long)B * H * TV;
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const int tv = (int)(i % TV);
    const long row = i / TV;
    const int c = (int)(row % H);
    const int b = (int)(row / H);
    const long oa = ((long)b * 2 * H + c) * T + (long)tv * V;
    Vec<V> at = Vec<V>::load(a + oa);
    Vec<V> as = Vec<V>::load(a + oa + (long)H * T);
    Vec<V> go = Vec<V>::load(dacts + row * T + (long)tv * V);
    const float gt = g ? g[(long)b * 2 * H + c] : 0.f;
    const float gs = g ? g[(long)b * 2 * H + H + c] : 0.f;
    Vec<V> dt, ds;
#pragma unroll
    for (int j = 0; j < V; ++j) {
        const float th = tanhf(at[j] + gt);
        const float sg = sigmoidf_(as[j] + gs);
        dt[j] = go[j] * sg * (1.0f - th * th);
        ds[j] = go[j] * th * sg * (1.0f - sg);
    }
    dt.store(da + oa);
    ds.store(da + oa + (long)H * T);
}

// gate backward from the SAVED tanh / sigmoid values (the fused conv+gate forward stores them): da = d(pre-activation),
// including the dropout keep-mask / scale of the forward when one was applied (layers.py:147).
template <int V>
__global__ __launch_bounds__(256) void gate_bwd_ts_kernel(const float *__restrict__ ts, const float *__restrict__ dacts,
                                                          const unsigned char *__restrict__ drop, float drop_scale,
                                                          float *__restrict__ da, int B, int H, int T) {
    const int TV = T / V;
    const long n = (long)B * H * TV;
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const int tv = (int)(i % TV);
    const long row = i / TV;
    const int c = (int)(row % H);
    const int b = (int)(row / H);
    const long ot = ((long)b * 2 * H + c) * T + (long)tv * V;
    const long os = ot + (long)H * T;
    Vec<V> th = Vec<V>::load(ts + ot);
    Vec<V> sg = Vec<V>::load(ts + os);
    Vec<V> go = Vec<V>::load(dacts + row * T + (long)tv * V);
    Vec<V> dt, ds;
#pragma unroll
    for (int j = 0; j < V; ++j) {
        dt[j] = go[j] * sg[j] * (1.0f - th[j] * th[j]);
        ds[j] = go[j] * th[j] * sg[j] * (1.0f - sg[j]);
        if (drop) {
            dt[j] = drop[ot + j] ? dt[j] * drop_scale : 0.f;
            ds[j] = drop[os + j] ? ds[j] * drop_scale : 0.f;
        }
    }
    dt.store(da + ot);
    ds.store(da + os);
}

template <int V, bool LAST>
__global__ __launch_bounds__(256) void res_skip_fwd_kernel(const float *__restrict__ x, const float *__restrict__ rs,
                                                           const float *__restrict__ mask, const float *__restrict__ skip_in,
                                                           float *__restrict__ x_out, float *__restrict__ skip_out, int B,
                                                           int H, int T) {
    const int TV = T / V;
    const long n = (long)B * H * TV;
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const int tv = (int)(i % TV);
    const long row = i / TV;
    const int c = (int)(row % H);
    const int b = (int)(row / H);
    const long oh = row * T + (long)tv * V;                                  // offset in a (B,H,T) tensor
    Vec<V> mv = Vec<V>::load(mask + (long)b * T + (long)tv * V);
    Vec<V> sk = skip_in ? Vec<V>::load(skip_in + oh) : Vec<V>::zero();
    if (LAST) {
        Vec<V> r = Vec<V>::load(rs + oh);
#pragma unroll
        for (int j = 0; j < V; ++j) sk[j] = (sk[j] + r[j]) * mv[j];
        sk.store(skip_out + oh);
    } else {
        const long o2 = ((long)b * 2 * H + c) * T + (long)tv * V;            // offset in the (B,2H,T) rs
        Vec<V> r0 = Vec<V>::load(rs + o2);
        Vec<V> r1 = Vec<V>::load(rs + o2 + (long)H * T);
        Vec<V> xv = Vec<V>::load(x + oh);
#pragma unroll
        for (int j = 0; j < V; ++j) {
            xv[j] = (xv[j] + r0[j]) * mv[j];
            sk[j] += r1[j];
        }
        xv.store(x_out + oh);
        sk.store(skip_out + oh);
    }
}

template <int V, bool LAST, bool B16 = false>
__global__ __launch_bounds__(256) void res_skip_bwd_kernel(const void *__restrict__ dx_out, const void *__restrict__ dskip,
                                                           const float *__restrict__ mask, void *__restrict__ dx,
                                                           void *__restrict__ drs, int B, int H, int T) {
    using IO = VecIO<V, B16>;
    const int TV = T / V;
    const long n = (long)B * H * TV;
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const int tv = (int)(i % TV);
    const long row = i / TV;
    const int c = (int)(row % H);
    const int b = (int)(row / H);
    const long oh = row * T + (long)tv * V;
    Vec<V> mv = Vec<V>::load(mask + (long)b * T + (long)tv * V);
    Vec<V> ds = IO::load(dskip, oh);
    if (LAST) {
#pragma unroll
        for (int j = 0; j < V; ++j) ds[j] *= mv[j];
        IO::store(drs, oh, ds);
    } else {
        const long o2 = ((long)b * 2 * H + c) * T + (long)tv * V;
        Vec<V> gx = IO::load(dx_out, oh);
#pragma unroll
        for (int j = 0; j < V; ++j) gx[j] *= mv[j];
        IO::store(drs, o2, gx);
        IO::store(drs, o2 + (long)H * T, ds);
        if (dx) IO::store(dx, oh, gx);
    }
}

}  // namespace glowtts

using namespace glowtts;

extern "C" int glowtts_gate_fwd(const float *a, const float *g, float *acts, int B, int H, int T,
                                glowtts_stream_t stream) {
    GLOWTTS_CHECK_ARG(a && acts, "glowtts_gate_fwd: null pointer");
    GLOWTTS_CHECK_ARG(B >= 0 && H > 0 && T >= 0, "glowtts_gate_fwd: bad shape");
    if ((long)B * H * T == 0) return 0;
    const bool v4 = can_vec4(T, {a, acts});
    const long n = (long)B * H * (v4 ? T / 4 : T);
    if (v4) hipLaunchKernelGGL((gate_fwd_kernel<4>), dim3(cdiv(n, 256)), dim3(256), 0, (hipStream_t)stream, a, g, acts, B, H, T);
    else    hipLaunchKernelGGL((gate_fwd_kernel<1>), dim3(cdiv(n, 256)), dim3(256), 0, (hipStream_t)stream, a, g, acts, B, H, T);
    GLOWTTS_LAUNCH_CHECK("glowtts_gate_fwd");
}

extern "C" int glowtts_gate_bwd(const float *a, const float *g, const float *dacts, float *da, int B, int H, int T,
                                glowtts_stream_t stream) {
    GLOWTTS_CHECK_ARG(a && dacts && da, "glowtts_gate_bwd: null pointer");
    GLOWTTS_CHECK_ARG(B >= 0 && H > 0 && T >= 0, "glowtts_gate_bwd: bad shape");
    if ((long)B * H * T == 0) return 0;
    const bool v4 = can_vec4(T, {a, dacts, da});
    const long n = (long)B * H * (v4 ? T / 4 : T);
    if (v4) hipLaunchKernelGGL((gate_bwd_kernel<4>), dim3(cdiv(n, 256)), dim3(256), 0, (hipStream_t)stream, a, g, dacts, da, B, H, T);
    else    hipLaunchKernelGGL((gate_bwd_kernel<1>), dim3(cdiv(n, 256)), dim3(256), 0, (hipStream_t)stream, a, g, dacts, da, B, H, T);
    GLOWTTS_LAUNCH_CHECK("glowtts_gate_bwd");
}

extern "C" int glowtts_gate_bwd_ts(const float *ts, const float *dacts, const unsigned char *drop, float drop_scale,
                                   float *da, int B, int H, int T, glowtts_stream_t stream) {
    GLOWTTS_CHECK_ARG(ts && dacts && da, "glowtts_gate_bwd_ts: null pointer");
    GLOWTTS_CHECK_ARG(B >= 0 && H > 0 && T >= 0, "glowtts_gate_bwd_ts: bad shape");
    if ((long)B * H * T == 0) return 0;
    const bool v4 = can_vec4(T, {ts, dacts, da});
    const long n = (long)B * H * (v4 ? T / 4 : T);
    if (v4) hipLaunchKernelGGL((gate_bwd_ts_kernel<4>), dim3(cdiv(n, 256)), dim3(256), 0, (hipStream_t)stream, ts, dacts, drop, drop_scale, da, B, H, T);
    else    hipLaunchKernelGGL((gate_bwd_ts_kernel<1>), dim3(cdiv(n, 256)), dim3(256), 0, (hipStream_t)stream, ts, dacts, drop, drop_scale, da, B, H, T);
    GLOWTTS_LAUNCH_CHECK("glowtts_gate_bwd_ts");
}

extern "C" int glowtts_res_skip_fwd(const float *x, const float *rs, const float *mask, const float *skip_in,
                                    float *x_out, float *skip_out, int B, int H, int T, int last,
                                    glowtts_stream_t stream) {
    GLOWTTS_CHECK_ARG(rs && mask && skip_out, "glowtts_res_skip_fwd: null pointer");
    GLOWTTS_CHECK_ARG(last || (x && x_out), "glowtts_res_skip_fwd: x / x_out required unless last");
    GLOWTTS_CHECK_ARG(B >= 0 && H > 0 && T >= 0, "glowtts_res_skip_fwd: bad shape");
    if ((long)B * H * T == 0) return 0;
    hipStream_t s = (hipStream_t)stream;
    const bool v4 = can_vec4(T, {x, rs, mask, skip_in, x_out, skip_out});
    const long n = (long)B * H * (v4 ? T / 4 : T);
    dim3 grid(cdiv(n, 256));
    if (last) {
        if (v4) hipLaunchKernelGGL((res_skip_fwd_kernel<4, true>), grid, dim3(256), 0, s, x, rs, mask, skip_in, x_out, skip_out, B, H, T);
        else    hipLaunchKernelGGL((res_skip_fwd_kernel<1, true>), grid, dim3(256), 0, s, x, rs, mask, skip_in, x_out, skip_out, B, H, T);
    } else {
        if (v4) hipLaunchKernelGGL((res_skip_fwd_kernel<4, false>), grid, dim3(256), 0, s, x, rs, mask, skip_in, x_out, skip_out, B, H, T);
        else    hipLaunchKernelGGL((res_skip_fwd_kernel<1, false>), grid, dim3(256), 0, s, x, rs, mask, skip_in, x_out, skip_out, B, H, T);
    }
    GLOWTTS_LAUNCH_CHECK("glowtts_res_skip_fwd");
}

extern "C" int glowtts_res_skip_bwd_io(const void *dx_out, const void *dskip, const float *mask, void *dx, void *drs,
                                       int B, int H, int T, int last, int io, glowtts_stream_t stream) {
    GLOWTTS_CHECK_ARG(dskip && mask && drs, "glowtts_res_skip_bwd: null pointer");
    GLOWTTS_CHECK_ARG(last || dx_out, "glowtts_res_skip_bwd: dx_out required unless last");
    GLOWTTS_CHECK_ARG(B >= 0 && H > 0 && T >= 0, "glowtts_res_skip_bwd: bad shape");
    if ((long)B * H * T == 0) return 0;
    hipStream_t s = (hipStream_t)stream;
    const bool v4 = can_vec4(T, {dx_out, dskip, mask, dx, drs});
    const long n = (long)B * H * (v4 ? T / 4 : T);
    dim3 grid(cdiv(n, 256));
#define GLOWTTS_RSB(LAST, B16)                                                                                                     \
    do {                                                                                                                            \
        if (v4) hipLaunchKernelGGL((res_skip_bwd_kernel<4, LAST, B16>), grid, dim3(256), 0, s, dx_out, dskip, mask, dx, drs, B, H, T); \
        else    hipLaunchKernelGGL((res_skip_bwd_kernel<1, LAST, B16>), grid, dim3(256), 0, s, dx_out, dskip, mask, dx, drs, B, H, T); \
    } while (0)
    if (io) { if (last) GLOWTTS_RSB(true, true); else GLOWTTS_RSB(false, true); }
    else    { if (last) GLOWTTS_RSB(true, false); else GLOWTTS_RSB(false, false); }
#undef GLOWTTS_RSB
    GLOWTTS_LAUNCH_CHECK("glowtts_res_skip_bwd");
}

extern "C" int glowtts_res_skip_bwd(const float *dx_out, const float *dskip, const float *mask, float *dx, float *drs,
                                    int B, int H, int T, int last, glowtts_stream_t stream) {
    return glowtts_res_skip_bwd_io(dx_out, dskip, mask, dx, drs, B, H, T, last, 0, stream);
}
