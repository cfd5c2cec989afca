// wn_stack.hip — host-side executor of a whole WN stack (reference layers.py:134-162 and its backward): the launch
// sequence of one coupling block's gated conv stack in ONE C call.
//
// The per-layer sequence (k-tap conv + gate, 1x1 res/skip conv; backwards: res/skip gradient assembly, 1x1 backward-data
// + gate derivative, k-tap backward-data, two weight-gradient convs) was driven from Python: ~36 launches per block and
// direction at 12-18 us of interpreter / ctypes / allocator time each — 19 ms of host time per training step against
// 22 ms of GPU time.  Here the sequence is native: the caller allocates the activation slabs once, passes the packed
// weights as a small host-side table, and gets every launch queued from C.  Weight-gradient kernels (and the final
// un-packing) go to a second stream behind events, exactly as convops._WgradStream did.
#include "common.hpp"

namespace glowtts {

// Ordering events: a ring per (thread, device).  An event belongs to the device that was current when it was created,
// so a process that drives a second GPU gets a second ring; the cursor is thread-local because autograd issues the
// backward of each device from its own thread (a shared cursor could hand two threads the same event between its
// record and its wait).  64 events per ring: an event is re-used long after the wait that consumed it was queued.
constexpr int kEventPool = 64;
struct EventRing {
    hipEvent_t ev[kEventPool];
    bool ready = false;
    int next = 0;
};

static hipEvent_t next_event() {
    static thread_local EventRing rings[kMaxDevices];
    int dev = 0;
    (void)hipGetDevice(&dev);
    EventRing &r = rings[dev >= 0 && dev < kMaxDevices ? dev : 0];
    if (!r.ready) {
        for (int i = 0; i < kEventPool; ++i) hipEventCreateWithFlags(&r.ev[i], hipEventDisableTiming);
        r.ready = true;
    }
    hipEvent_t e = r.ev[r.next];
    r.next = (r.next + 1) % kEventPool;
    return e;
}

// wn_fused.hip: the whole stack's forward as one kernel; -1 = not applicable
int wn_fused_dispatch(const glowtts_wn_layer *layers, int n_layers, const float *x, const float *mask, const unsigned char *drop,
                      float drop_scale, float *xs, float *acts, float *ts, float *skip, int B, int H, int T, int taps, int dil_rate,
                      hipStream_t s);

// everything queued on `from` so far happens before whatever is queued on `to` next
static int order_after(hipStream_t from, hipStream_t to) {
    if (from == to) return 0;
    hipEvent_t e = next_event();
    hipError_t r = hipEventRecord(e, from);
    if (r == hipSuccess) r = hipStreamWaitEvent(to, e, 0);
    if (r != hipSuccess) { set_error("glowtts_wn: stream ordering: %s", hipGetErrorString(r)); return (int)r; }
    return 0;
}

}  // namespace glowtts

using namespace glowtts;

#define WN_TRY(expr) do { int rc_ = (expr); if (rc_ != 0) return rc_; } while (0)

// element-typed slab arithmetic: io = 0 -> fp32 tensors, io = 1 -> bf16 tensors (same shapes, same element offsets)
static inline const void *at(const void *base, long elems, int io) {
    return base ? static_cast<const char *>(base) + elems * (io ? 2 : 4) : nullptr;
}
static inline void *at(void *base, long elems, int io) {
    return base ? static_cast<char *>(base) + elems * (io ? 2 : 4) : nullptr;
}

// weight gradient of y = conv(x) for either tensor type: the fp32 kernels, or (bf16 tensors) the plane kernel fed with ONE
// plane — the tensor itself.  The plane kernel takes no masks: its callers hand it gradients that are masked already.
static int wrw_any(const void *x, long x_bs, const void *d, long d_bs, const float *mask_d, float *dwp, float *dbias, int B,
                   int Cin, int M, int T, int taps, int dil, int pad, int io, glowtts_stream_t stream) {
    if (!io)
        return glowtts_conv_wrw(static_cast<const float *>(x), x_bs, static_cast<const float *>(d), d_bs, mask_d, nullptr, dwp,
                                dbias, B, Cin, M, T, taps, dil, pad, stream);
    GLOWTTS_CHECK_ARG(dil == 1 && pad == (taps - 1) / 2 && !mask_d, "glowtts (bf16 tensors): weight gradient needs dilation 1, 'same' padding, pre-masked gradients");
    return glowtts_conv_wrw_planes(static_cast<const uint16_t *>(x), 0, x_bs, static_cast<const uint16_t *>(d), 0, d_bs, dwp, dbias,
                                   B, Cin, M, T, taps, 1, stream);
}

// slab_B: utterances per LAYER slab of xs / acts / ts / drop (>= B).  The stack node's forward runs two half-batch chains on two
// streams into the same (n_layers, slab_B, ..) slabs: each call then covers B = slab_B / 2 utterances of every layer's slab.
static int wn_fwd_impl(const glowtts_wn_layer *layers, int n_layers, const void *x, const float *mask, const float *cond,
                       const unsigned char *drop, float drop_scale, void *xs, void *acts, void *ts, void *skip, int B, int slab_B, int H,
                       int T, int taps, int dil_rate, int io, glowtts_stream_t stream) {
    GLOWTTS_CHECK_ARG(layers && x && mask && acts && ts && skip, "glowtts_wn_fwd: null pointer");
    GLOWTTS_CHECK_ARG(n_layers >= 1 && (n_layers == 1 || xs), "glowtts_wn_fwd: bad layer count / missing xs");
    GLOWTTS_CHECK_ARG(B >= 0 && H > 0 && T >= 0 && taps >= 1 && (taps & 1) && dil_rate >= 1, "glowtts_wn_fwd: bad shape");
    GLOWTTS_CHECK_ARG(slab_B >= B && (slab_B == B || !cond), "glowtts_wn_fwd: bad slab batch");
    const long BHT = (long)slab_B * H * T;
    if (!io && !cond && slab_B == B) {        // the layer-resident kernel (csrc/wn_fused.hip): fp32 tensors, bf16x6 arithmetic, H = 192, 5 taps
        const int rc = wn_fused_dispatch(layers, n_layers, static_cast<const float *>(x), mask, drop, drop_scale, static_cast<float *>(xs),
                                         static_cast<float *>(acts), static_cast<float *>(ts), static_cast<float *>(skip), B, H, T, taps,
                                         dil_rate, (hipStream_t)stream);
        if (rc >= 0) return rc;
    }
    long dil = 1;
    for (int i = 0; i < n_layers; ++i, dil *= dil_rate) {
        const glowtts_wn_layer &L = layers[i];
        const bool last = i == n_layers - 1;
        const void *x_i = i == 0 ? x : at((const void *)xs, (long)(i - 1) * BHT, io);
        const int pad = (int)((taps * dil - dil) / 2);
        WN_TRY(glowtts_conv_gate_fwd_io(x_i, L.wf_in, L.b_in, cond ? cond + (long)i * B * 2 * H : nullptr,
                                        drop ? drop + (long)i * 2 * BHT : nullptr, drop_scale,
                                        at(acts, (long)i * BHT, io), at(ts, (long)i * 2 * BHT, io), B, H, T, taps, (int)dil, pad,
                                        io, stream));
        // x_{i+1} = (x_i + rs[:H]) mask ; skip += rs[H:]   (skip accumulates in place: same thread reads and writes an element)
        WN_TRY(glowtts_conv_res_skip_fwd_io(at((const void *)acts, (long)i * BHT, io), L.wf_rs, L.b_rs, mask, last ? nullptr : x_i,
                                            i == 0 ? nullptr : skip, last ? nullptr : at(xs, (long)i * BHT, io), skip, B, H, T,
                                            last ? 1 : 0, io, stream));
    }
    return 0;
}

extern "C" int glowtts_wn_fwd_io(const glowtts_wn_layer *layers, int n_layers, const void *x, const float *mask,
                                 const float *cond, const unsigned char *drop, float drop_scale, void *xs, void *acts,
                                 void *ts, void *skip, int B, int H, int T, int taps, int dil_rate, int io,
                                 glowtts_stream_t stream) {
    return wn_fwd_impl(layers, n_layers, x, mask, cond, drop, drop_scale, xs, acts, ts, skip, B, B, H, T, taps, dil_rate, io, stream);
}

extern "C" int glowtts_wn_fwd(const glowtts_wn_layer *layers, int n_layers, const float *x, const float *mask,
                              const unsigned char *drop, float drop_scale, float *xs, float *acts, float *ts, float *skip,
                              int B, int H, int T, int taps, int dil_rate, glowtts_stream_t stream) {
    return glowtts_wn_fwd_io(layers, n_layers, x, mask, nullptr, drop, drop_scale, xs, acts, ts, skip, B, H, T, taps, dil_rate, 0,
                             stream);
}

extern "C" int glowtts_wn_bwd_io(const glowtts_wn_layer *layers, int n_layers, const void *x, const void *xs,
                                 const void *acts, const void *ts, const float *mask, const unsigned char *drop,
                                 float drop_scale, const void *dskip, void *d_rs, void *d_xin, void *dx, float *dcond,
                                 const long long *unpack_desc, const int *unpack_prefix, int n_conv, int total_rows, int B,
                                 int H, int T, int taps, int dil_rate, int two_source, int mask_input_grad, int io,
                                 glowtts_stream_t wgrad_stream, glowtts_stream_t stream) {
    GLOWTTS_CHECK_ARG(layers && x && acts && ts && mask && dskip && d_rs && d_xin && dx, "glowtts_wn_bwd: null pointer");
    GLOWTTS_CHECK_ARG(n_layers >= 1 && (n_layers == 1 || xs), "glowtts_wn_bwd: bad layer count / missing xs");
    GLOWTTS_CHECK_ARG(B >= 0 && H > 0 && T >= 0 && taps >= 1 && (taps & 1) && dil_rate >= 1, "glowtts_wn_bwd: bad shape");
    GLOWTTS_CHECK_ARG(!io || (!two_source && mask_input_grad), "glowtts_wn_bwd: bf16 tensors use the d_rs form with a masked input gradient");
    hipStream_t ms = (hipStream_t)stream;
    hipStream_t ws = wgrad_stream ? (hipStream_t)wgrad_stream : ms;
    const long BHT = (long)B * H * T;
    long dil = 1;
    for (int i = 1; i < n_layers; ++i) dil *= dil_rate;
    const void *dsk = dskip;
    WN_TRY(order_after(ms, ws));                                    // accumulators were cleared on the main stream
    // two_source bit 1 (set by glowtts_flow_block_bwd): `dskip` arrives MASKED already — the end conv's backward-data epilogue
    // multiplied it by the mask — so the last layer's d_rs = dskip mask IS dskip and its res_skip_bwd launch (a 20 MB pass on the
    // backward's chain, 12 per step) is not queued
    const bool pre_masked = (two_source & 2) != 0;
    // bit 2 (with bits 0 and 1): the caller queues the stack's 1x1 weight gradients itself (glowtts_flow_block_bwd: one
    // glowtts_conv_wrw1_multi launch for them and the start / end convs')
    const bool skip_wrw1 = (two_source & 4) != 0 && pre_masked;
    two_source &= 1;
    if (two_source) {
        // The launch sequence of convops.WNFn._backward_layers in its two-source form, layer by layer: d_rs = [dx_{i+1} mask ;
        // dskip] is never written (gate_bwd and wrw2 read its halves from the two tensors), each weight-gradient kernel goes to
        // the second stream as soon as its operands exist, and only d_xin / dx live per layer.  d_rs needs B*H*T floats (the
        // last layer's masked dskip).  fp32 tensors only.
        GLOWTTS_CHECK_ARG(dil_rate == 1 && H % 192 == 0 && T % 4 == 0, "glowtts_wn_bwd: two-source form needs dilation 1, H %% 192 == 0, T %% 4 == 0");
        const int pad = (taps - 1) / 2;
        const float *xf = static_cast<const float *>(x), *xsf = static_cast<const float *>(xs);
        const float *actsf = static_cast<const float *>(acts), *tsf = static_cast<const float *>(ts);
        float *d_rsf = static_cast<float *>(d_rs), *d_xinf = static_cast<float *>(d_xin), *dxf = static_cast<float *>(dx);
        const float *dskf = static_cast<const float *>(dskip);
        // Default: the dx chain of the whole stack first, then the stack's weight gradients as batched launches
        // (glowtts_conv_wrw_batch: the 5-tap ones of all layers in one launch, the 1x1 two-source ones of layers 0 .. n-2 in
        // another): a launch per problem ends with its split-K atomics draining before the stream's next kernel may start; in one
        // launch the next problem's workgroups take the compute units as they free up — 15.54 -> 15.30 ms per step (three
        // alternating runs).  GLOWTTS_WRW_BATCH=0: a launch per layer, each as soon as its operands exist.
        const bool batch = knob(K_WRW_BATCH) != 0;
        if (batch && n_layers <= 8) {
            const float *bx5[8], *bd5[8], *bx1[8], *bd1[8], *bd1b[8];
            float *bw5[8], *bb5[8], *bw1[8], *bb1[8];
            for (int i = n_layers - 1; i >= 0; --i) {
                const glowtts_wn_layer &L = layers[i];
                const bool last = i == n_layers - 1;
                const float *ts_i = tsf + (long)i * 2 * BHT;
                const unsigned char *drop_i = drop ? drop + (long)i * 2 * BHT : nullptr;
                float *dxin_i = d_xinf + (long)i * 2 * BHT, *dx_i = dxf + (long)i * BHT;
                const float *half = last ? nullptr : dxf + (long)(i + 1) * BHT;
                if (last) {
                    if (!pre_masked) {
                        WN_TRY(glowtts_res_skip_bwd(nullptr, dskf, mask, nullptr, d_rsf, B, H, T, 1, stream));
                        dskf = d_rsf;
                    }
                    WN_TRY(glowtts_conv_gate_bwd_io(dskf, nullptr, L.wb_rs, ts_i, drop_i, drop_scale, dxin_i,
                                                    dcond ? dcond + (long)i * B * 2 * H : nullptr, B, H, H, T, 0, stream));
                } else {
                    WN_TRY(glowtts_conv_gate_bwd_io(half, dskf, L.wb_rs, ts_i, drop_i, drop_scale, dxin_i,
                                                    dcond ? dcond + (long)i * B * 2 * H : nullptr, B, 2 * H, H, T, 0, stream));
                }
                const bool m_out = i > 0 || mask_input_grad;
                WN_TRY(glowtts_conv_fwd(dxin_i, (long)2 * H * T, L.wb_in, nullptr, m_out ? mask : nullptr, half, (long)H * T, dx_i,
                                        (long)H * T, B, 2 * H, H, T, taps, 1, (taps - 1) - pad, 0, m_out ? 1 : 0, 0, stream));
                bx5[i] = i == 0 ? xf : xsf + (long)(i - 1) * BHT; bd5[i] = dxin_i; bw5[i] = L.dwp_in; bb5[i] = L.db_in;
                bx1[i] = actsf + (long)i * BHT; bd1[i] = half; bd1b[i] = dskf; bw1[i] = L.dwp_rs; bb1[i] = L.db_rs;
            }
            WN_TRY(order_after(ms, ws));
            const int nl = n_layers - 1;
            if (!skip_wrw1) {
                WN_TRY(glowtts_conv_wrw(bx1[nl], (long)H * T, dskf, (long)H * T, nullptr, nullptr, bw1[nl], bb1[nl], B, H, H, T, 1, 1, 0,
                                        (glowtts_stream_t)ws));
                if (nl > 0)
                    WN_TRY(glowtts_conv_wrw_batch(nl, bx1, (long)H * T, bd1, (long)H * T, bd1b, (long)H * T, H, nullptr, nullptr, bw1, bb1, B,
                                                  H, 2 * H, T, 1, 1, 0, (glowtts_stream_t)ws));
            }
            WN_TRY(glowtts_conv_wrw_batch(n_layers, bx5, (long)H * T, bd5, (long)2 * H * T, nullptr, 0, 0, nullptr, nullptr, bw5, bb5, B, H,
                                          2 * H, T, taps, 1, pad, (glowtts_stream_t)ws));
            if (unpack_desc)
                WN_TRY(glowtts_unpack_weight_grad_multi(unpack_desc, unpack_prefix, n_conv, total_rows, (glowtts_stream_t)ws));
            return 0;
        }
        for (int i = n_layers - 1; i >= 0; --i) {
            const glowtts_wn_layer &L = layers[i];
            const bool last = i == n_layers - 1;
            const float *x_i = i == 0 ? xf : xsf + (long)(i - 1) * BHT;
            const float *acts_i = actsf + (long)i * BHT, *ts_i = tsf + (long)i * 2 * BHT;
            const unsigned char *drop_i = drop ? drop + (long)i * 2 * BHT : nullptr;
            float *dxin_i = d_xinf + (long)i * 2 * BHT, *dx_i = dxf + (long)i * BHT;
            const float *half = last ? nullptr : dxf + (long)(i + 1) * BHT;       // left the layer above already masked
            if (last) {
                if (!pre_masked) {
                    WN_TRY(glowtts_res_skip_bwd(nullptr, dskf, mask, nullptr, d_rsf, B, H, T, 1, stream));
                    dskf = d_rsf;
                }
                if (!skip_wrw1) {
                    WN_TRY(order_after(ms, ws));
                    WN_TRY(glowtts_conv_wrw(acts_i, (long)H * T, dskf, (long)H * T, nullptr, nullptr, L.dwp_rs, L.db_rs, B, H, H, T, 1,
                                            1, 0, (glowtts_stream_t)ws));
                }
                WN_TRY(glowtts_conv_gate_bwd_io(dskf, nullptr, L.wb_rs, ts_i, drop_i, drop_scale, dxin_i,
                                                dcond ? dcond + (long)i * B * 2 * H : nullptr, B, H, H, T, 0, stream));
            } else {
                if (!skip_wrw1) {
                    WN_TRY(order_after(ms, ws));
                    WN_TRY(glowtts_conv_wrw2(acts_i, (long)H * T, half, (long)H * T, dskf, (long)H * T, H, L.dwp_rs, L.db_rs, B, H, 2 * H,
                                             T, 1, 1, 0, (glowtts_stream_t)ws));
                }
                WN_TRY(glowtts_conv_gate_bwd_io(half, dskf, L.wb_rs, ts_i, drop_i, drop_scale, dxin_i,
                                                dcond ? dcond + (long)i * B * 2 * H : nullptr, B, 2 * H, H, T, 0, stream));
            }
            WN_TRY(order_after(ms, ws));
            WN_TRY(glowtts_conv_wrw(x_i, (long)H * T, dxin_i, (long)2 * H * T, nullptr, nullptr, L.dwp_in, L.db_in, B, H, 2 * H, T,
                                    taps, 1, pad, (glowtts_stream_t)ws));
            // dx_i = (residual path) dx_{i+1} + (conv path) W_in^T (*) d_xin, masked unless it is the stack's own input gradient
            const bool m_out = i > 0 || mask_input_grad;
            WN_TRY(glowtts_conv_fwd(dxin_i, (long)2 * H * T, L.wb_in, nullptr, m_out ? mask : nullptr, half, (long)H * T, dx_i,
                                    (long)H * T, B, 2 * H, H, T, taps, 1, (taps - 1) - pad, 0, m_out ? 1 : 0, 0, stream));
        }
        if (unpack_desc)
            WN_TRY(glowtts_unpack_weight_grad_multi(unpack_desc, unpack_prefix, n_conv, total_rows, (glowtts_stream_t)ws));
        return 0;
    }
    for (int i = n_layers - 1; i >= 0; --i, dil /= dil_rate) {
        const glowtts_wn_layer &L = layers[i];
        const bool last = i == n_layers - 1;
        const int m_rs = last ? H : 2 * H;
        const int pad = (int)((taps * dil - dil) / 2);
        const void *ts_i = at(ts, (long)i * 2 * BHT, io);
        void *drs_i = at(d_rs, (long)i * 2 * BHT, io), *dxin_i = at(d_xin, (long)i * 2 * BHT, io), *dx_i = at(dx, (long)i * BHT, io);
        // d_rs = [dx_{i+1} mask ; dskip]   (last layer: dskip mask, which also becomes dskip of the layers below)
        WN_TRY(glowtts_res_skip_bwd_io(last ? nullptr : at((const void *)dx, (long)(i + 1) * BHT, io), dsk, mask, nullptr, drs_i, B, H, T,
                                       last ? 1 : 0, io, stream));
        if (last) dsk = drs_i;
        // d(pre-activation) = gate'(stored tanh / sigmoid) * (W_rs^T d_rs)
        WN_TRY(glowtts_conv_gate_bwd_io(drs_i, nullptr, L.wb_rs, ts_i, drop ? drop + (long)i * 2 * BHT : nullptr, drop_scale, dxin_i,
                                        dcond ? dcond + (long)i * B * 2 * H : nullptr, B, m_rs, H, T, io, stream));
        // dx_i = (residual path) d_rs[:H] + (conv path) W_in^T (*) d_xin ; the last layer has no residual path
        const bool m_out = i == 0 && mask_input_grad;
        WN_TRY(glowtts_conv_fwd_io(dxin_i, (long)2 * H * T, L.wb_in, nullptr, m_out ? mask : nullptr, last ? nullptr : drs_i,
                                   last ? 0 : (long)2 * H * T, dx_i, (long)H * T, B, 2 * H, H, T, taps, (int)dil,
                                   (int)((taps - 1) * dil - pad), 0, m_out ? 1 : 0, 0, io, io, stream));
    }
    // the whole dx chain of the stack is queued first; the weight gradients follow on their own stream, where they run
    // beside the chain of whatever backward comes next (per-layer interleaving made the two streams fight for the
    // same CUs at the same time and cost 1 ms per step; a THIRD stream for the 1x1 weight gradients, beside the 5-tap ones, is
    // worth 0.05 ms per step at most — measured in round 3, not kept)
    WN_TRY(order_after(ms, ws));
    dil = 1;
    for (int i = 1; i < n_layers; ++i) dil *= dil_rate;
    for (int i = n_layers - 1; i >= 0; --i, dil /= dil_rate) {
        const glowtts_wn_layer &L = layers[i];
        const bool last = i == n_layers - 1;
        const int m_rs = last ? H : 2 * H;
        const int pad = (int)((taps * dil - dil) / 2);
        const void *x_i = i == 0 ? x : at(xs, (long)(i - 1) * BHT, io);
        const void *acts_i = at(acts, (long)i * BHT, io);
        const void *drs_i = at((const void *)d_rs, (long)i * 2 * BHT, io), *dxin_i = at((const void *)d_xin, (long)i * 2 * BHT, io);
        WN_TRY(wrw_any(acts_i, (long)H * T, drs_i, (long)m_rs * T, nullptr, L.dwp_rs, L.db_rs, B, H, m_rs, T, 1, 1, 0, io,
                       (glowtts_stream_t)ws));
        WN_TRY(wrw_any(x_i, (long)H * T, dxin_i, (long)2 * H * T, nullptr, L.dwp_in, L.db_in, B, H, 2 * H, T, taps, (int)dil, pad, io,
                       (glowtts_stream_t)ws));
    }
    if (unpack_desc)                                                // behind the weight-gradient kernels, on their stream
        WN_TRY(glowtts_unpack_weight_grad_multi(unpack_desc, unpack_prefix, n_conv, total_rows, (glowtts_stream_t)ws));
    return 0;
}

extern "C" int glowtts_wn_bwd(const glowtts_wn_layer *layers, int n_layers, const float *x, const float *xs,
                              const float *acts, const float *ts, const float *mask, const unsigned char *drop,
                              float drop_scale, const float *dskip, float *d_rs, float *d_xin, float *dx,
                              const long long *unpack_desc, const int *unpack_prefix, int n_conv, int total_rows, int B,
                              int H, int T, int taps, int dil_rate, int two_source, glowtts_stream_t wgrad_stream,
                              glowtts_stream_t stream) {
    return glowtts_wn_bwd_io(layers, n_layers, x, xs, acts, ts, mask, drop, drop_scale, dskip, d_rs, d_xin, dx, nullptr, unpack_desc,
                             unpack_prefix, n_conv, total_rows, B, H, T, taps, dil_rate, two_source, 0, 0, wgrad_stream, stream);
}

// ---- a whole flow block per call ---------------------------------------------------------------------------------------
// [ActNorm, InvConvNear, CouplingBlock] (reference models.py:176-190, forward direction) is ONE host call each way.  The
// Python path drove it as five autograd nodes per block (fused ActNorm+InvConv, start conv, WN, end conv, affine apply):
// ~120 us of interpreter / autograd time per block forward and ~180 us backward on top of the launches themselves,
// 3 ms of the ~19 ms host time of a config-2 step.  The launch sequence below is the same one, kernel for kernel.
// io bit 0: the HIDDEN tensors of the coupling network (h0, xs, acts, ts, skip and every gradient among them, dout) are bf16
// in HBM; io bit 1 (needs bit 0): the FLOW tensor (x, y, z and their gradients) as well.  `out` = (m, logs) feeds the
// log-determinant and stays fp32 together with every log-det, mask, parameter and parameter gradient; all arithmetic
// accumulates in fp32.  With io == 1 the start conv reads y0h, a bf16 copy (B, C/2, T) of y's first half written by the
// fused ActNorm + InvConv kernel (unused otherwise, may be NULL).
extern "C" int glowtts_flow_block_fwd_io(const glowtts_flow_block *blk, const void *x, const float *mask, const float *x_len,
                                         const float *cond, const unsigned char *drop, float drop_scale, void *y, void *y0h, void *h0, void *xs,
                                         void *acts, void *ts, void *skip, float *out, void *z, float *logdet, int B, int C,
                                         int H, int T, int taps, int dil_rate, int n_split, int sigmoid_scale, int io,
                                         glowtts_stream_t stream) {
    // io bits 8 / 9 (round 4, convops.FlowStackFn with fp32 tensors): the caller has produced y itself (the previous block's affine
    // apply fused with this block's ActNorm + InvConv: glowtts_coupling_actnorm_invconv_fwd, which also needs W^-1 / log det W made
    // beforehand) / the caller runs this block's affine apply itself, fused into the next block: `z` is not written
    // io bit 10: W^-1 and log det W of this block are in place already (glowtts_invconv_prepare_multi: one launch for the stack)
    // io bits 11 / 12: the end conv (needs bit 9) / the start conv (needs bit 8) are the caller's as well — glowtts_flow_boundary_fwd
    // runs end conv(k), the flows between and start conv(k + 1) in one launch; this call is then the WN stack alone
    const bool skip_head = (io & 256) != 0, skip_tail = (io & 512) != 0, w_ready = (io & 1024) != 0;
    const bool skip_end = (io & 2048) != 0, skip_start = (io & 4096) != 0;
    GLOWTTS_CHECK_ARG((!skip_end || skip_tail) && (!skip_start || skip_head), "glowtts_flow_block_fwd: io bits 11 / 12 need bits 9 / 8");
    io &= 255;
    const int io_h = io & 1, io_f = (io >> 1) & 1;
    GLOWTTS_CHECK_ARG(io_h || !io_f, "glowtts_flow_block: a bf16 flow tensor needs bf16 hidden tensors (io = 0, 1 or 3)");
    GLOWTTS_CHECK_ARG(!(skip_head || skip_tail) || io == 0, "glowtts_flow_block_fwd: the fused-flow flags go with fp32 tensors");
    GLOWTTS_CHECK_ARG(io != 1 || y0h, "glowtts_flow_block_fwd: io = 1 needs the y0h buffer");
    const void *start_in = io == 1 ? y0h : y;
    const long start_bs = io == 1 ? (long)(C / 2) * T : (long)C * T;
    GLOWTTS_CHECK_ARG(blk && x && mask && x_len && y && h0 && acts && ts && skip && out && z && logdet,
                      "glowtts_flow_block_fwd: null pointer");
    GLOWTTS_CHECK_ARG(blk->layers && blk->n_layers >= 1 && blk->w_inv && blk->logdet_w && (!blk->pack_desc || blk->pack_prefix),
                      "glowtts_flow_block_fwd: incomplete block table");
    GLOWTTS_CHECK_ARG(C > 0 && (C % 2) == 0 && (n_split == 2 || n_split == 4) && C % n_split == 0,
                      "glowtts_flow_block_fwd: C=%d n_split=%d", C, n_split);
    const long CT = (long)C * T, HT = (long)H * T;
    // weight norm + k-packing of all 2 + 2 n_layers convolutions (pack_desc == NULL: the caller has packed them already,
    // e.g. because it also refreshes bf16 planes of the packed weights); W^-1 and log det W of the invertible 1x1
    if (blk->pack_desc)
        WN_TRY(glowtts_pack_weight_multi(blk->pack_desc, blk->pack_prefix, blk->n_conv, blk->total_rows, stream));
    if (!skip_head) {
        if (!w_ready) WN_TRY(glowtts_invconv_prepare(blk->w, blk->w_inv, blk->logdet_w, n_split, stream));
        // flows 3i, 3i+1: y = W ((bias + e^logs x) mask) mask ; logdet = (sum logs + log det W * C/n) x_len
        WN_TRY(glowtts_actnorm_invconv_fwd_io(x, mask, blk->logs, blk->bias, blk->w, blk->logdet_w, x_len, y, logdet,
                                              io == 1 ? y0h : nullptr, B, C, T, n_split, io_f, stream));
    }
    // flow 3i+2: h = start(y[:, :C/2]) mask  ->  WN  ->  out = end(h)  ->  z = [y0 ; (m + e^logs y1) mask], logdet += sum logs mask
    if (!skip_start)
        WN_TRY(glowtts_conv_fwd_io(start_in, start_bs, blk->wf_start, blk->b_start, mask, nullptr, 0, h0, HT, B, C / 2, H, T, 1, 1, 0, 0,
                                   1, 0, io_h, io_h, stream));
    // (blk->reserved > 0: the layer slabs hold that many utterances and this call covers B of them — the stack node's two
    // half-batch forward chains)
    WN_TRY(wn_fwd_impl(blk->layers, blk->n_layers, h0, mask, cond, drop, drop_scale, xs, acts, ts, skip, B,
                       blk->reserved > B ? blk->reserved : B, H, T, taps, dil_rate, io_h, stream));
    if (!skip_end)
        WN_TRY(glowtts_conv_fwd_io(skip, HT, blk->wf_end, blk->b_end, nullptr, nullptr, 0, out, CT, B, H, C, T, 1, 1, 0, 0, 0, 0, io_h, 0,
                                   stream));
    if (skip_tail) return 0;
    return glowtts_coupling_fwd_io(y, out, mask, z, logdet, B, C, T, sigmoid_scale, 0, io_f, stream);
}

extern "C" int glowtts_flow_block_fwd(const glowtts_flow_block *blk, const float *x, const float *mask, const float *x_len,
                                      const unsigned char *drop, float drop_scale, float *y, float *h0, float *xs,
                                      float *acts, float *ts, float *skip, float *out, float *z, float *logdet, int B, int C,
                                      int H, int T, int taps, int dil_rate, int n_split, int sigmoid_scale,
                                      glowtts_stream_t stream) {
    return glowtts_flow_block_fwd_io(blk, x, mask, x_len, nullptr, drop, drop_scale, y, nullptr, h0, xs, acts, ts, skip, out, z, logdet,
                                     B, C, H, T, taps, dil_rate, n_split, sigmoid_scale, 0, stream);
}

extern "C" int glowtts_flow_block_bwd_io(const glowtts_flow_block *blk, const void *x, const float *mask, const float *x_len,
                                         const unsigned char *drop, float drop_scale, const void *y, const void *y0h, const void *h0,
                                         const void *xs, const void *acts, const void *ts, const void *skip, const float *out,
                                         const void *dz, const float *dlogdet, void *dy, void *dout, void *dskip, void *d_rs,
                                         void *d_xin, void *dx_wn, void *dx, float *dcond, int B, int C, int H, int T, int taps, int dil_rate,
                                         int n_split, int sigmoid_scale, int two_source, int io, glowtts_stream_t wgrad_stream,
                                         glowtts_stream_t stream) {
    GLOWTTS_CHECK_ARG(blk && x && mask && x_len && y && h0 && acts && ts && skip && out && dz && dy && dout && dskip && d_rs &&
                      d_xin && dx_wn && dx, "glowtts_flow_block_bwd: null pointer");
    GLOWTTS_CHECK_ARG(blk->layers && blk->n_layers >= 1 && blk->unpack_desc && blk->pack_prefix && blk->dwp_all && blk->dlogs &&
                      blk->dbias && blk->dw, "glowtts_flow_block_bwd: incomplete block table");
    // io bits 8 / 9 (see glowtts_flow_block_fwd_io): the ActNorm + InvConv backward at the END of this block's chain is left to the
    // caller (fused with the previous block's coupling backward: glowtts_coupling_actnorm_invconv_bwd) / dy and dout have been
    // produced by the caller (the same fused kernel one block later in the flow): no coupling backward at the start
    // io bits 11 / 12 (glowtts_flow_boundary_bwd): dskip has been produced by the caller too (with bit 9: no end-conv backward-data
    // here) / the start conv's backward-data is the caller's (with bit 8: this call ends with the WN stack's dx_wn)
    const bool skip_ai = (io & 256) != 0, skip_cpl = (io & 512) != 0;
    const bool skip_end_bd = (io & 2048) != 0, skip_start_bd = (io & 4096) != 0;
    GLOWTTS_CHECK_ARG((!skip_end_bd || skip_cpl) && (!skip_start_bd || skip_ai), "glowtts_flow_block_bwd: io bits 11 / 12 need bits 9 / 8");
    io &= 255;
    GLOWTTS_CHECK_ARG(!io || !two_source, "glowtts_flow_block_bwd: bf16 tensors use the d_rs form");
    GLOWTTS_CHECK_ARG(!(skip_ai || skip_cpl) || io == 0, "glowtts_flow_block_bwd: the fused-flow flags go with fp32 tensors");
    const int io_h = io & 1, io_f = (io >> 1) & 1;
    GLOWTTS_CHECK_ARG(io_h || !io_f, "glowtts_flow_block: a bf16 flow tensor needs bf16 hidden tensors (io = 0, 1 or 3)");
    GLOWTTS_CHECK_ARG(io != 1 || y0h, "glowtts_flow_block_bwd: io = 1 needs the y0h buffer");
    const void *start_in = io == 1 ? y0h : y;
    const long start_bs = io == 1 ? (long)(C / 2) * T : (long)C * T;
    hipStream_t ms = (hipStream_t)stream;
    hipStream_t ws = wgrad_stream ? (hipStream_t)wgrad_stream : ms;
    const long CT = (long)C * T, HT = (long)H * T;
    // every packed weight-gradient accumulator of the block in one fill (the kernels add into it with atomics) — on the stream the
    // weight-gradient kernels run on: nothing else touches the accumulators (the previous step's un-pack is earlier on that stream),
    // and on the chain the fill was 5 us per block between two backward-data launches
    hipError_t e = hipMemsetAsync(blk->dwp_all, 0, (size_t)blk->dwp_floats * sizeof(float), ws);
    if (e != hipSuccess) { set_error("glowtts_flow_block_bwd: memset: %s", hipGetErrorString(e)); return (int)e; }
    // affine apply backwards: dy = [dz0 ; dz1 e^logs mask], dout = [dm ; dlogs]
    if (!skip_cpl) WN_TRY(glowtts_coupling_bwd_io(y, out, mask, dz, dlogdet, dy, dout, B, C, T, sigmoid_scale, io_f, io_h, stream));
    // end conv (H -> C, 1x1): weight gradient on the second stream, d(skip) on the chain.
    // Round 4, two-source form: the block's SIX 1x1 weight gradients (end conv, start conv, the stack's last and two-source
    // res/skip convs) are one launch behind the block's chain (glowtts_conv_wrw1_multi: 9 tiles of 192 x 192 sharing the compute
    // units) — alone they were launches of 22 us for < 1 us of matrix work each, and 88 us for the batch of three.
    const bool multi1 = two_source && !io && blk->n_layers >= 1 && blk->n_layers + 2 <= 8 && H % 64 == 0;
    if (!multi1) {
        WN_TRY(order_after(ms, ws));
        WN_TRY(wrw_any(skip, HT, dout, CT, nullptr, blk->dwp_end, blk->db_end, B, H, C, T, 1, 1, 0, io_h, (glowtts_stream_t)ws));
    }
    // (two-source form: dskip leaves this conv masked, which is the last WN layer's d_rs — see glowtts_wn_bwd_io)
    const int pre_mask = (two_source && !io) ? 1 : 0;
    if (!skip_end_bd)
        WN_TRY(glowtts_conv_fwd_io(dout, CT, blk->wb_end, nullptr, pre_mask ? mask : nullptr, nullptr, 0, dskip, HT, B, C, H, T, 1, 1, 0, 0,
                                   pre_mask, 0, io_h, io_h, stream));
    // the gated conv stack (its weight gradients go to the second stream as well; un-packing is done below for the block).
    // With bf16 tensors the stack masks its own input gradient: the start conv's weight gradient then needs no mask.
    WN_TRY(glowtts_wn_bwd_io(blk->layers, blk->n_layers, h0, xs, acts, ts, mask, drop, drop_scale, dskip, d_rs, d_xin, dx_wn, dcond,
                             nullptr, nullptr, 0, 0, B, H, T, taps, dil_rate,
                             two_source ? (1 | (pre_mask << 1) | ((multi1 && pre_mask) ? 4 : 0)) : 0, io_h, io_h, wgrad_stream, stream));
    // start conv (C/2 -> H, 1x1, output masked): its input gradient is ADDED into dy[:, :C/2] where the affine apply left dz0
    WN_TRY(order_after(ms, ws));
    if (multi1 && pre_mask) {
        const int nl = blk->n_layers;
        const long BHT = (long)B * H * T;
        const float *actsf = static_cast<const float *>(acts), *dxw = static_cast<const float *>(dx_wn);
        const float *dskf = static_cast<const float *>(dskip);                 // masked by the end conv's backward-data epilogue
        glowtts_wrw1_problem pr[8] = {};
        int np = 0;
        for (int i = 0; i < nl; ++i) {                                        // res/skip convs: x = acts_i, d = [dx_{i+1} mask ; dskip]
            const glowtts_wn_layer &L = blk->layers[i];
            glowtts_wrw1_problem &q = pr[np++];
            q.x = actsf + (long)i * BHT; q.x_bs = HT; q.Cin = H; q.dwp = L.dwp_rs; q.dbias = L.db_rs;
            if (i == nl - 1) { q.d = dskf; q.d_bs = HT; q.M = H; }
            else { q.d = dxw + (long)(i + 1) * BHT; q.d_bs = HT; q.d2 = dskf; q.d2_bs = HT; q.d_split = H; q.M = 2 * H; }
        }
        {   // end conv: x = the skip sum, d = dout
            glowtts_wrw1_problem &q = pr[np++];
            q.x = static_cast<const float *>(skip); q.x_bs = HT; q.Cin = H; q.d = static_cast<const float *>(dout); q.d_bs = CT; q.M = C;
            q.dwp = blk->dwp_end; q.dbias = blk->db_end;
        }
        {   // start conv: x = y[:, :C/2], d = the stack's input gradient times the mask (the conv's output was masked)
            glowtts_wrw1_problem &q = pr[np++];
            q.x = static_cast<const float *>(start_in); q.x_bs = start_bs; q.Cin = C / 2; q.d = dxw; q.d_bs = HT; q.M = H; q.mask_d = mask;
            q.dwp = blk->dwp_start; q.dbias = blk->db_start;
        }
        WN_TRY(glowtts_conv_wrw1_multi(np, pr, B, T, (glowtts_stream_t)ws));
    } else {
        WN_TRY(wrw_any(start_in, start_bs, dx_wn, HT, io_h ? nullptr : mask, blk->dwp_start, blk->db_start, B, C / 2, H, T, 1, 1, 0, io_h,
                       (glowtts_stream_t)ws));
    }
    if (!skip_start_bd)
        WN_TRY(glowtts_conv_fwd_io(dx_wn, HT, blk->wb_start, nullptr, mask, dy, CT, dy, CT, B, H, C / 2, T, 1, 1, 0, 1, 0, 0, io_h, io_f,
                                   stream));
    // flows 3i+1, 3i backwards in one pass; parameter gradients accumulate straight into their targets
    if (!skip_ai)
        WN_TRY(glowtts_actnorm_invconv_bwd_io(x, mask, blk->logs, blk->bias, blk->w, blk->w_inv, dy, dlogdet, x_len, dx, blk->dlogs,
                                              blk->dbias, blk->dw, B, C, T, n_split, io_f, stream));
    // the second stream finishes the block: after this point on `ws` EVERY parameter gradient of the block is complete
    // (the three ActNorm / InvConv gradients were produced on the chain, hence the ordering edge)
    WN_TRY(order_after(ms, ws));
    return glowtts_unpack_weight_grad_multi(blk->unpack_desc, blk->pack_prefix, blk->n_conv, blk->total_rows, (glowtts_stream_t)ws);
}

extern "C" int glowtts_flow_block_bwd(const glowtts_flow_block *blk, const float *x, const float *mask, const float *x_len,
                                      const unsigned char *drop, float drop_scale, const float *y, const float *h0,
                                      const float *xs, const float *acts, const float *ts, const float *skip,
                                      const float *out, const float *dz, const float *dlogdet, float *dy, float *dout,
                                      float *dskip, float *d_rs, float *d_xin, float *dx_wn, float *dx, int B, int C, int H,
                                      int T, int taps, int dil_rate, int n_split, int sigmoid_scale, int two_source,
                                      glowtts_stream_t wgrad_stream, glowtts_stream_t stream) {
    return glowtts_flow_block_bwd_io(blk, x, mask, x_len, drop, drop_scale, y, nullptr, h0, xs, acts, ts, skip, out, dz, dlogdet, dy, dout,
                                     dskip, d_rs, d_xin, dx_wn, dx, nullptr, B, C, H, T, taps, dil_rate, n_split, sigmoid_scale,
                                     two_source, 0, wgrad_stream, stream);
}


// ---- a whole transformer layer of the text encoder per call --------------------------------------------------------------
// attentions.py:63-73: x = x * mask ; y = attn(x) ; x = LN1(x + drop(y)) ; y = ffn(x) ; x = LN2(x + drop(y)), with
// attn = conv_o(rel_attention(conv_q(x), conv_k(x), conv_v(x))) (attentions.py:204-211) and ffn = conv_2(drop(relu(conv_1(x * mask)))
// * mask) * mask (attentions.py:373-381).  The per-operator path was ~16 autograd nodes and ~8 torch elementwise launches per
// layer each way; here the elementwise neighbours ride in kernel epilogues (x * mask as mask_in of the q / k / v convs and as
// mask_x of LN1; both dropouts of the residual branches inside the LayerNorm kernels; ReLU + dropout as the epilogue of
// conv_1 and as the gate of conv_2's backward-data) and the sequence is one host call each way.
extern "C" int glowtts_encoder_layer_fwd(const glowtts_enc_layer *L, const float *x, const float *mask,
                                         const unsigned char *drop_a, const unsigned char *drop_o, const unsigned char *drop_h,
                                         const unsigned char *drop_2, float drop_scale, float *q, float *k, float *v,
                                         float *p_attn, float *y_att, float *o, float *x1, float *stats1, float *h, float *y2,
                                         float *x2, float *stats2, int B, int H, int F, int T, int heads, int taps, int window,
                                         int heads_share, int block_len, float eps, glowtts_stream_t stream) {
    GLOWTTS_CHECK_ARG(L && x && mask && q && k && v && p_attn && y_att && o && x1 && stats1 && h && y2 && x2 && stats2,
                      "glowtts_encoder_layer_fwd: null pointer");
    GLOWTTS_CHECK_ARG(heads > 0 && H % heads == 0 && (taps & 1), "glowtts_encoder_layer_fwd: bad shape");
    const long HT = (long)H * T, FT = (long)F * T;
    const int pad = (taps - 1) / 2, dk = H / heads;
    if (L->pack_desc)
        WN_TRY(glowtts_pack_weight_multi(L->pack_desc, L->pack_prefix, L->n_conv, L->total_rows, stream));
    // q, k, v = 1x1 convs of x * mask
    WN_TRY(glowtts_conv_fwd(x, HT, L->wf_q, L->b_q, mask, nullptr, 0, q, HT, B, H, H, T, 1, 1, 0, 1, 0, 0, stream));
    WN_TRY(glowtts_conv_fwd(x, HT, L->wf_k, L->b_k, mask, nullptr, 0, k, HT, B, H, H, T, 1, 1, 0, 1, 0, 0, stream));
    WN_TRY(glowtts_conv_fwd(x, HT, L->wf_v, L->b_v, mask, nullptr, 0, v, HT, B, H, H, T, 1, 1, 0, 1, 0, 0, stream));
    WN_TRY(glowtts_rel_attn_fwd_ex(q, k, v, L->emb_k, L->emb_v, mask, drop_a, drop_scale, p_attn, y_att, B, heads, T, dk, window,
                                heads_share, block_len, L->attn_bf16, stream));
    WN_TRY(glowtts_conv_fwd(y_att, HT, L->wf_o, L->b_o, nullptr, nullptr, 0, o, HT, B, H, H, T, 1, 1, 0, 0, 0, 0, stream));
    // x1 = LN1(x * mask + dropout(o))
    WN_TRY(glowtts_chan_layernorm_fwd_ex(x, o, mask, drop_o, drop_scale, L->gamma1, L->beta1, x1, stats1, B, H, T, eps, stream));
    // h = dropout(relu(conv_1(x1 * mask))) ; y2 = conv_2(h * mask) * mask
    WN_TRY(glowtts_conv_fwd_act(x1, HT, L->wf_1, L->b_1, mask, nullptr, 0, h, FT, B, H, F, T, taps, 1, pad, 1, 0, 0, 1, drop_h,
                                drop_scale, nullptr, 1.f, stream));
    WN_TRY(glowtts_conv_fwd(h, FT, L->wf_2, L->b_2, mask, nullptr, 0, y2, HT, B, F, H, T, taps, 1, pad, 1, 1, 0, stream));
    // x2 = LN2(x1 + dropout(y2))
    return glowtts_chan_layernorm_fwd_ex(x1, y2, nullptr, drop_2, drop_scale, L->gamma2, L->beta2, x2, stats2, B, H, T, eps, stream);
}

extern "C" int glowtts_encoder_layer_bwd(const glowtts_enc_layer *L, const float *x, const float *mask,
                                         const unsigned char *drop_a, const unsigned char *drop_o, const unsigned char *drop_h,
                                         const unsigned char *drop_2, float drop_scale, const float *q, const float *k,
                                         const float *v, const float *p_attn, const float *y_att, const float *o, const float *x1,
                                         const float *stats1, const float *h, const float *y2, const float *stats2,
                                         const float *dx2, float *dx1a, float *dy2, float *d_pre1, float *dx1, float *dxa,
                                         float *d_o, float *dy_att, float *ds, float *dq, float *dkk, float *dv, float *dx, int B,
                                         int H, int F, int T, int heads, int taps, int window, int heads_share, int block_len,
                                         glowtts_stream_t wgrad_stream, glowtts_stream_t stream) {
    GLOWTTS_CHECK_ARG(L && x && mask && q && k && v && p_attn && y_att && o && x1 && stats1 && h && y2 && stats2 && dx2 && dx1a &&
                      dy2 && d_pre1 && dx1 && dxa && d_o && dy_att && ds && dq && dkk && dv && dx,
                      "glowtts_encoder_layer_bwd: null pointer");
    GLOWTTS_CHECK_ARG(L->unpack_desc && L->pack_prefix && L->dwp_all, "glowtts_encoder_layer_bwd: incomplete layer table");
    hipStream_t ms = (hipStream_t)stream;
    hipStream_t ws = wgrad_stream ? (hipStream_t)wgrad_stream : ms;
    glowtts_stream_t wss = (glowtts_stream_t)ws;
    const long HT = (long)H * T, FT = (long)F * T;
    const int pad = (taps - 1) / 2, dk = H / heads;
    hipError_t e = hipMemsetAsync(L->dwp_all, 0, (size_t)L->dwp_floats * sizeof(float), ws);   // (on the weight-gradient stream: see glowtts_flow_block_bwd_io)
    if (e != hipSuccess) { set_error("glowtts_encoder_layer_bwd: memset: %s", hipGetErrorString(e)); return (int)e; }
    // LN2 backwards: dx1a = d(x1) through the residual path, dy2 = d(y2) through the dropout
    WN_TRY(glowtts_chan_layernorm_bwd_ex(x1, y2, nullptr, drop_2, drop_scale, L->gamma2, stats2, dx2, dx1a, drop_2 ? dy2 : nullptr,
                                         L->dgamma2, L->dbeta2, B, H, T, stream));
    const float *g2 = drop_2 ? dy2 : dx1a;
    // conv_2 (y2 = conv(h * mask) * mask): weight gradient aside; input gradient gated by ReLU' and conv_1's dropout in one test
    WN_TRY(order_after(ms, ws));
    WN_TRY(glowtts_conv_wrw(h, FT, g2, HT, mask, mask, L->dwp_2, L->db_2, B, F, H, T, taps, 1, pad, wss));
    WN_TRY(glowtts_conv_fwd_act(g2, HT, L->wb_2, nullptr, mask, nullptr, 0, d_pre1, FT, B, H, F, T, taps, 1, (taps - 1) - pad, 1, 1, 0,
                                0, nullptr, 1.f, h, drop_h ? drop_scale : 1.f, stream));
    // conv_1 (pre = conv(x1 * mask)): dx1 = dx1a + mask * (W1^T d_pre1) = mask * (dx1a + W1^T d_pre1): dx1a vanishes beyond the
    // utterance, because the gradient arriving at a layer's output does (every consumer of it multiplies by the mask first)
    WN_TRY(order_after(ms, ws));
    WN_TRY(glowtts_conv_wrw(x1, HT, d_pre1, FT, nullptr, mask, L->dwp_1, L->db_1, B, H, F, T, taps, 1, pad, wss));
    WN_TRY(glowtts_conv_fwd(d_pre1, FT, L->wb_1, nullptr, mask, dx1a, HT, dx1, HT, B, F, H, T, taps, 1, (taps - 1) - pad, 0, 1, 0,
                            stream));
    // LN1 backwards: dxa = d(x) through the residual path (times mask), d_o through the dropout
    WN_TRY(glowtts_chan_layernorm_bwd_ex(x, o, mask, drop_o, drop_scale, L->gamma1, stats1, dx1, dxa, drop_o ? d_o : nullptr,
                                         L->dgamma1, L->dbeta1, B, H, T, stream));
    const float *go = drop_o ? d_o : dxa;
    // conv_o (its weight gradient joins those of q / k / v below: one multi-problem launch, csrc/convwrw1.hip)
    WN_TRY(glowtts_conv_fwd(go, HT, L->wb_o, nullptr, nullptr, nullptr, 0, dy_att, HT, B, H, H, T, 1, 1, 0, 0, 0, 0, stream));
    // attention
    WN_TRY(glowtts_rel_attn_bwd_ex(dy_att, q, k, v, L->emb_k, L->emb_v, mask, drop_a, drop_scale, p_attn, ds, dq, dkk, dv, L->demb_k,
                                L->demb_v, B, heads, T, dk, window, heads_share, block_len, L->attn_bf16, stream));
    // q, k, v convs of x * mask: dx = dxa + mask * (Wq^T dq + Wk^T dk + Wv^T dv)
    WN_TRY(order_after(ms, ws));
    {
        glowtts_wrw1_problem pr[4] = {};
        const float *bd[3] = {dq, dkk, dv};
        float *bw[3] = {L->dwp_q, L->dwp_k, L->dwp_v}, *bb[3] = {L->db_q, L->db_k, L->db_v};
        for (int j = 0; j < 3; ++j) {                 // q, k, v = conv(x * mask)
            pr[j].x = x; pr[j].x_bs = HT; pr[j].Cin = H; pr[j].d = bd[j]; pr[j].d_bs = HT; pr[j].M = H; pr[j].mask_x = mask;
            pr[j].dwp = bw[j]; pr[j].dbias = bb[j];
        }
        pr[3].x = y_att; pr[3].x_bs = HT; pr[3].Cin = H; pr[3].d = go; pr[3].d_bs = HT; pr[3].M = H; pr[3].dwp = L->dwp_o; pr[3].dbias = L->db_o;
        WN_TRY(glowtts_conv_wrw1_multi(4, pr, B, T, wss));
    }
    // dxa is masked already and mask * mask = mask, so dxa + mask * S = mask * (dxa + S): the chain starts from dxa and the
    // last convolution masks the sum
    WN_TRY(glowtts_conv_fwd(dq, HT, L->wb_q, nullptr, nullptr, dxa, HT, dx, HT, B, H, H, T, 1, 1, 0, 0, 0, 0, stream));
    WN_TRY(glowtts_conv_fwd(dkk, HT, L->wb_k, nullptr, nullptr, dx, HT, dx, HT, B, H, H, T, 1, 1, 0, 0, 0, 0, stream));
    WN_TRY(glowtts_conv_fwd(dv, HT, L->wb_v, nullptr, mask, dx, HT, dx, HT, B, H, H, T, 1, 1, 0, 0, 1, 0, stream));
    WN_TRY(order_after(ms, ws));
    return glowtts_unpack_weight_grad_multi(L->unpack_desc, L->pack_prefix, L->n_conv, L->total_rows, wss);
}
