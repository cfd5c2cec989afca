/*
 * glowtts_hip.h — C ABI of libglowtts_hip.so: the MI355X (gfx950) kernels behind the Glow-TTS training hot path.
 *
 * This is the drop-in boundary.  The reference has exactly one native entry point on this path,
 * the Cython `maximum_path_c` (glow_tts_train/monotonic_align/core.pyx:40); everything else on the path is a
 * PyTorch op called from its Python operators.  Each function below names the reference site it replaces.
 *
 * Conventions (all functions):
 *   - plain pointers + sizes, no torch types; every pointer is DEVICE memory owned by the caller;
 *     no allocation, no ownership transfer; re-entrant from autograd's backward threads.  State the library keeps:
 *       process-wide, mutable : the arithmetic mode of glowtts_conv_math and the glowtts_wn_fused switch (explicit setters;
 *                               atomic — a launch uses the setting in force when it is queued, so flip them between steps,
 *                               not while a backward runs);
 *       process-wide, latched : the tuning switches below — ALL of them are read from the environment ONCE, at the library's
 *                               first use of any of them, into a table of atomics (csrc/error.hip; no getenv on the launch
 *                               path); glowtts_set_knob(name, value) changes one afterwards (the A/B tools flip them between
 *                               blocks of steps; like the two setters above: not while a backward runs).  They select between
 *                               kernels / launch geometries with IDENTICAL results; defaults in brackets (-1 = unset):
 *                                 GLOWTTS_WRW_TR      [-1]    0 = 5-tap weight gradient on the frame-packed kernel, 1 = 64x32
 *                                                     tiles in the 16x16x32 form            (csrc/convwrw_tr.hip)
 *                                 GLOWTTS_WRW_TR_MT   [4]     2 = 64x32 tiles (32x32x16 form) instead of 64x64
 *                                 GLOWTTS_WRW_TR3     [1]     0 = 3-tap weight gradients on the frame-packed kernel
 *                                 GLOWTTS_WRW_TR3_MT  [2]     4 = 3-tap weight gradients on 64x64 tiles instead of 64x32
 *                                 GLOWTTS_WRW_TR_PRIO [2]     which wave group of the 5-tap weight gradient runs at raised priority
 *                                                     (0 = the multiplying waves, 1 = nobody, 2 = the storing waves)
 *                                 GLOWTTS_WRW_BATCH   [1]     0 = one weight-gradient launch per WN layer (csrc/wn_stack.hip)
 *                                 GLOWTTS_WRW1_PIPE   [1]     0 = 1x1 weight gradient without the software-pipelined plane split
 *                                                     (csrc/convgemm_split.hip)
 *                                 GLOWTTS_WRW5_BSPLIT [1]     0 = a batched 5-tap weight-gradient launch sizes its split-K per problem
 *                                                     (one round of workgroups per problem) instead of sharing the compute units
 *                                                     between the problems of the batch (csrc/convwrw_tr.hip)
 *                                 GLOWTTS_WRW5_CUS    [-1]    compute units a BATCHED 5-tap weight-gradient launch sizes its split-K for
 *                                                     (-1 = all of the device's)
 *                                 GLOWTTS_WINO        [1]     1 = the WN stack's gated 5-tap in-conv in its Winograd F(4, 5) form
 *                                                     (csrc/convwino.hip) wherever Winograd-domain planes are bound to the launching
 *                                                     thread (glowtts_conv_bind_wino); results differ from the direct form by the
 *                                                     fp32 roundings of the transforms (same tolerance against the oracle)
 *                                 GLOWTTS_WRW1_MULTI  [1]     0 = the 1x1 weight gradients of a flow block / transformer layer as separate
 *                                                     launches instead of one multi-problem launch (csrc/convwrw1.hip)
 *                                 GLOWTTS_WRW1_CUS    [-1]    compute units the multi-problem 1x1 weight gradient sizes its split-K for
 *                                                     (-1 = half of the device's: DESIGN.md 4j)
 *                                 GLOWTTS_WRW1_XCD    [1]     0 = its split count is not rounded down to a multiple of 8 (one split's
 *                                                     tiles then no longer share an XCD)
 *                                 GLOWTTS_WRW_TR_NG   [2]     1 = the 5-tap 64 x 64 weight gradient with ONE 4-wave group per workgroup (half the
 *                                                     LDS; measured 0.5 ms per step slower: DESIGN.md lesson 36;
 *                                                     GLOWTTS_WRW_TR_NG_SPLITS [1] multiplies its split-K workgroup count)
 *                                 GLOWTTS_CONV_ROW_ADJ [1]    0 = the bf16-plane convolution kernels take their workgroups in grid order
 *                                                     (all frame tiles of row tile 0, then row tile 1, ..) instead of numbering the
 *                                                     row tiles of one frame tile into consecutive slots of one XCD
 *                                                     (csrc/convgemm_split.hip)
 *                                 GLOWTTS_CONV32_1X1  [1]     0 = plain 1x1 convolutions (native fp32 kernels) on 80- / 64-frame tiles
 *                                                     instead of 32-frame tiles (csrc/convgemm.hip)
 *                                 GLOWTTS_WN_FUSED    [0]     initial value of the glowtts_wn_fused switch (csrc/wn_fused.hip)
 *                                 GLOWTTS_MAS_WAVES   [1]     0 = the alignment search on the single-wave kernel for every lattice
 *                                                     (csrc/mas.hip; 1: up to four DP waves where glowtts_mas_spans_supported)
 *                               Only in the tuning build (`make -C csrc trace`, -DGLOWTTS_TRACE, tools/libglowtts_trace.bin; the
 *                               shipped library contains neither the names nor the code paths), timing experiments that make
 *                               kernels SKIP work and so give WRONG results:
 *                                 GLOWTTS_BND_EXP     [0]     csrc/flow_boundary.hip: bit 0 / 1 = without the first / second contraction's
 *                                                     MFMAs, 2 = without the element-wise phase, 3 = backward without the group
 *                                                     reduction (tools/boundary_bench.py)
 *                                 GLOWTTS_WRW1_EXP    [0]     csrc/convwrw1.hip: bit 0 = no atomics, 2 = no MFMAs, 3 = no loads after the first
 *                                                     step (tools/wrw1_bench.py)
 *                                 GLOWTTS_CONV_EXP    [0]     csrc/convgemm_split.hip: bit 0 = no activation loads, 1 = no split / LDS stores,
 *                                                     2 = no weight loads after the first three, 3 = no epilogue, 4-6 = WHEN the next chunk's
 *                                                     activation loads are issued (tools/conv_exp.py)
 *       per device            : high-water marks of kernel LDS limits (hipFuncSetAttribute called once per kernel);
 *       per thread            : event rings of the timing mode, bf16-plane bindings (glowtts_conv_bind_planes);
 *   - activations are fp32, contiguous (B, C, T) with T fastest (the reference layout); masks are fp32 (B, T)
 *     holding 0/1 (the reference's (B,1,T) float mask viewed flat); log-determinants are fp32 (B);
 *   - `stream` is a hipStream_t (pass PyTorch's current stream): launches are asynchronous and ordered on it,
 *     nothing synchronises, so every call is hipGraph-capturable;
 *   - outputs documented "accumulated" are atomically ADDED to (caller zeroes them or passes a running sum);
 *   - return 0 on success, non-zero on argument error or hipError_t; glowtts_last_error() gives the text.
 *     The Python wrappers raise RuntimeError on non-zero, mirroring the reference's Python asserts
 *     (layers.py:227,240; attentions.py:162,226-228).
 */
#ifndef GLOWTTS_HIP_H
#define GLOWTTS_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef void *glowtts_stream_t; /* hipStream_t */

const char *glowtts_last_error(void);
int glowtts_abi_version(void);

/* ---- tuning switches (conventions block above; no reference counterpart: the reference has no native tuning state) ----------
 * set_knob : set the switch `name` ("GLOWTTS_WRW_TR_MT" or "WRW_TR_MT") for every launch queued from now on; non-zero (and
 *            glowtts_last_error) for a name the library does not know — a misspelt switch is an error, not a silent no-op.
 * get_knob : the value in force (environment value, default, or the last set_knob). */
int glowtts_set_knob(const char *name, int value);
int glowtts_get_knob(const char *name, int *value);

/* ---- monotonic alignment search ----------------------------------------------------------------------------
 * replaces maximum_path_c / maximum_path_each (monotonic_align/core.pyx:9-45) and the D2H/H2D round trip of its
 * wrapper (monotonic_align/__init__.py:11-21).
 * value : (B, Tx, Ty) fp32 log-likelihoods, READ ONLY (the reference mutates it as scratch; here the running
 *         column lives in registers).  Only cells inside each utterance's band are read, so `value` need not be
 *         pre-multiplied by the mask.
 * path  : (B, Tx, Ty) fp32, fully written with 0/1 (no pre-zeroing needed).
 * t_x,t_y: (B) int32 valid lengths (the reference derives them from the mask, __init__.py:18-19).
 * Bit-exact with the reference for identical `value` (one fp32 add per cell, max = (prev > cur) ? prev : cur,
 * max_neg_val = -1e9).  Limits: Tx <= 2048 (the reference has none); lattices whose 1-bit back-pointer image exceeds the
 * LDS (e.g. 500 x 4000) keep it in `path` until the path itself is written.
 */
int glowtts_mas_path(const float *value, float *path, const int32_t *t_x, const int32_t *t_y,
                     int B, int Tx, int Ty, glowtts_stream_t stream);
/* the same search, also handing out what its backtrack already knows (either may be NULL):
 * first (B, Tx + 1) int32: first[b, x] = first frame of text row x (rows >= t_x and first[b, Tx]: t_y) — row x owns the
 *   frames [first[x], first[x+1]), so sum_y path[x, y] = first[x+1] - first[x]  (models.py:393);
 * tok (B, Ty) int32: the text row of every frame, -1 for frames >= t_y. */
int glowtts_mas_path_spans(const float *value, float *path, int32_t *first, int32_t *tok, const int32_t *t_x,
                           const int32_t *t_y, int B, int Tx, int Ty, glowtts_stream_t stream);
/* Round 5 — the search on up to four wavefronts (csrc/mas.hip: mas_wave_kernel; core.pyx:9-35 unchanged in arithmetic): text rows
 * split over the waves, a skewed pipeline over 16-frame slabs, cells read straight from `value`.  Lattices with
 * glowtts_mas_spans_supported(Tx, Ty) != 0 (Ty % 4 == 0, Tx <= 512, back-pointer bits within the LDS; 16-byte aligned value / path)
 * take it inside glowtts_mas_path / glowtts_mas_path_spans; every other lattice the single-wave kernel.  With a span table the path
 * is not written by the search but expanded from `first` by a second launch over the whole chip:
 *   - glowtts_mas_path_spans(..., path = NULL, first != NULL, ...) on a supported lattice runs the search alone (spans, tok);
 *   - glowtts_mas_path_from_spans(first, path, B, Tx, Ty, stream): path[b, x, y] = 1 for first[b, x] <= y < first[b, x + 1], else 0,
 *     on any stream ordered behind the search (the Python host: a side stream, off the step's serial stretch; Ty % 4 == 0,
 *     16-byte aligned path).
 * GLOWTTS_MAS_WAVES = 0 (tuning switch) keeps every lattice on the single-wave kernel. */
int glowtts_mas_spans_supported(int Tx, int Ty);
int glowtts_mas_path_from_spans(const int32_t *first, float *path, int B, int Tx, int Ty, glowtts_stream_t stream);

/* ---- alignment-side glue of FlowGenerator.forward (csrc/align.hip; reference models.py:361-393) -------------------------
 * align_logp : logp[b, x, y] = log N(z[b, :, y]; x_m[b, :, x], exp(x_logs[b, :, x])) summed over the C channels — the
 *   (B, Tx, Ty) lattice of the alignment search (models.py:362-376: two bmm's + two channel sums + elementwise ops) as one
 *   fp32-MFMA contraction over k = 2C per utterance.  x_logs may be NULL (mean_only: log-std 0).  x_m, x_logs (B, C, Tx),
 *   z (B, C, Ty).
 * align_expand_fwd : out[b, d, y] = stats[b, d, tok[b, y]] (0 where tok < 0) = attn^T stats for the hard path the search
 *   returned (models.py:383-392: a bmm with one-hot rows) ; stats (B, D, Tx), out (B, D, Ty)
 * align_expand_bwd : dstats[b, d, x] = sum of dout[b, d, y] over y in [first[b, x], first[b, x+1])  (WRITTEN, no atomics) */
int glowtts_align_logp(const float *x_m, const float *x_logs, const float *z, float *logp, int B, int C, int Tx, int Ty,
                       glowtts_stream_t stream);
int glowtts_align_expand_fwd(const float *stats, const int32_t *tok, float *out, int B, int D, int Tx, int Ty,
                             glowtts_stream_t stream);
int glowtts_align_expand_bwd(const float *dout, const int32_t *first, float *dstats, int B, int D, int Tx, int Ty,
                             glowtts_stream_t stream);

/* ---- sequence lengths from a mask: x_len[b] = sum_t mask[b, t]  (layers.py:187,245) ------------------------ */
int glowtts_mask_len(const float *mask, float *x_len, int B, int T, glowtts_stream_t stream);

/* ---- ActNorm (layers.py:182-199; init :207-221) ------------------------------------------------------------
 * fwd : z = (bias + exp(logs) * x) * mask ; logdet[b] = sum(logs) * x_len[b]   (logdet may be NULL)
 * rev : z = (x - bias) * exp(-logs) * mask
 * bwd : dx = dz * exp(logs) * mask ; dlogs[c] += sum_{b,t} dz*x*exp(logs)*mask + sum_b dlogdet[b]*x_len[b] ;
 *       dbias[c] += sum_{b,t} dz*mask      (dlogs, dbias accumulated; dlogdet may be NULL)
 * stats: sum_x[c] += sum x*mask ; sum_x2[c] += sum x*x*mask  (accumulated; denominators via glowtts_mask_len) */
int glowtts_actnorm_fwd(const float *x, const float *mask, const float *logs, const float *bias,
                        const float *x_len, float *z, float *logdet, int B, int C, int T, int reverse,
                        glowtts_stream_t stream);
int glowtts_actnorm_bwd(const float *x, const float *mask, const float *logs, const float *dz,
                        const float *dlogdet, const float *x_len, float *dx, float *dlogs, float *dbias,
                        int B, int C, int T, glowtts_stream_t stream);
int glowtts_actnorm_stats(const float *x, const float *mask, float *sum_x, float *sum_x2, int B, int C, int T,
                          glowtts_stream_t stream);

/* ---- InvConvNear (layers.py:238-275) ------------------------------------------------------------------------
 * prepare: one wavefront factorises the n x n weight (n = n_split) by Gauss-Jordan with partial pivoting, fp64 — held
 *          across lanes for n <= 8, in LDS for n <= 32: w_inv (n*n) and logdet_w[0] = log(det W) (NaN if det <= 0, as
 *          torch.logdet, layers.py:265).  Replaces torch.logdet / torch.inverse (layers.py:258,265,275).
 * n_split: any even value <= 32 that divides C (layers.py:227,240).  2 / 4 / 8 keep a group's n x n mix in registers; other
 *          group sizes run on run-time-n kernels (one thread per output row of a group, dW entry by entry).
 * fwd    : per (b, group g, t): z[k_out] = sum_k W[k_out,k] x[k], rows k = h*(n/2)+s <-> channel
 *          h*(C/2) + g*(n/2) + s  (layers.py:247-252, 267-271), times mask;
 *          logdet[b] = logdet_w * (C/n) * x_len[b]   (logdet may be NULL; pass w = w_inv for reverse)
 * bwd    : dx = W^T (dz*mask) ; dw[o,k] += sum dz[o]*mask*x[k] + w_inv[k,o] * (C/n) * sum_b dlogdet[b]*x_len[b]
 *          (dw accumulated; dlogdet may be NULL) */
int glowtts_invconv_prepare(const float *w, float *w_inv, float *logdet_w, int n, glowtts_stream_t stream);
int glowtts_invconv_fwd(const float *x, const float *mask, const float *w, const float *logdet_w,
                        const float *x_len, float *z, float *logdet, int B, int C, int T, int n_split,
                        glowtts_stream_t stream);
int glowtts_invconv_bwd(const float *x, const float *mask, const float *w, const float *w_inv, const float *dz,
                        const float *dlogdet, const float *x_len, float *dx, float *dw, int B, int C, int T,
                        int n_split, glowtts_stream_t stream);

/* ---- ActNorm + InvConvNear in one pass (flows 3i and 3i+1 of every decoder block, models.py:176-179) ------------
 * The two flows are consecutive and elementwise per (b, group, t); fused, the (B, C, T) tensor crosses HBM twice in
 * forward (read x, write z) and three times in backward (read x, dz; write dx) instead of 4 / 6 times.
 * fwd : y = (bias + exp(logs) x) mask ; z = (W y) mask ; logdet[b] = (sum(logs) + logdet_w * C/n) * x_len[b]
 * bwd : dx, and ACCUMULATED dlogs, dbias (C each) and dw (n*n) -- the sums of the two separate backward kernels.
 * n_split in {2, 4}; other values use the separate entry points above. */
int glowtts_actnorm_invconv_fwd(const float *x, const float *mask, const float *logs, const float *bias,
                                const float *w, const float *logdet_w, const float *x_len, float *z, float *logdet,
                                int B, int C, int T, int n_split, glowtts_stream_t stream);
int glowtts_actnorm_invconv_bwd(const float *x, const float *mask, const float *logs, const float *bias,
                                const float *w, const float *w_inv, const float *dz, const float *dlogdet,
                                const float *x_len, float *dx, float *dlogs, float *dbias, float *dw, int B, int C,
                                int T, int n_split, glowtts_stream_t stream);
/* W^-1 and log|det W| of `n_problems` invertible 1x1 convolutions in ONE launch (round 4: convops.FlowStackFn, one launch per
 * step instead of one per flow block): w_table[i] = device address of the i-th (n, n) fp32 weight; results at
 * w_inv + i * out_stride: n * n floats of W^-1 followed by log|det W| (out_stride >= n * n + 1).  Same arithmetic as
 * glowtts_invconv_prepare (fp64 Gauss-Jordan with partial pivoting; NaN log-det for a negative determinant). */
int glowtts_invconv_prepare_multi(const long long *w_table, float *w_inv, long out_stride, int n_problems, int n,
                                  glowtts_stream_t stream);

/* Everything between the WN stacks of two consecutive flow blocks in ONE launch (round 4, csrc/flow_boundary.hip; fp32 tensors):
 *   out = end_k(skip_k) (attentions.py:131-133) ; z = [y_k0 ; (m + e^logs' y_k1) mask], logdet_prev += sum logs' mask (135-142) ;
 *   y = W ((bias + e^logs z) mask) mask, logdet = (sum logs + logdet_w C / n_split) x_len (layers.py:182-199, 238-272) ;
 *   h0 = (start_{k+1}(y[:, :C/2]) + b_start) mask (attentions.py:122-123).
 * skip (B, H, T): block k's WN output; wp_end / wp_start: the convolutions' packed fp32 weights ([G][M][16], as glowtts_pack_weight*
 * writes them); logs, bias, w, logdet_w: block k + 1's ActNorm / InvConvNear (logdet_w from glowtts_invconv_prepare*).
 * Written: out (B, C, T), y (B, C, T), h0 (B, H, T), logdet (B); logdet_prev (B) is accumulated.  z is never written.
 * The matrix products are convgemm_wd_kernel's (fp32 MFMA, same order): results equal those of the three launches it replaces.
 * Limits: C <= 192, H <= 192, T % 4 == 0, n_split 2 or 4, 16-byte aligned tensors (anything else: argument error). */
int glowtts_flow_boundary_fwd(const float *skip, const float *wp_end, const float *b_end, const float *y_prev, const float *mask,
                              const float *logs, const float *bias, const float *w, const float *logdet_w, const float *x_len,
                              const float *wp_start, const float *b_start, float *out, float *y, float *h0, float *logdet_prev,
                              float *logdet, int B, int C, int H, int T, int n_split, int sigmoid_scale, glowtts_stream_t stream);

/* The same boundary backwards, ONE launch on the chain (csrc/flow_boundary.hip; fp32 tensors):
 *   dyf = dy_next + [W_start^T (dx_wn mask) ; 0]  (the start conv's backward-data of block k + 1, mask_in form) ; the ActNorm +
 *   InvConvNear backward of block k + 1 on z recomputed from (y_prev, out_prev) and block k's coupling backward — the arithmetic of
 *   glowtts_coupling_actnorm_invconv_bwd — writing dy_prev, dout_prev (B, C, T) ; dskip = W_end^T dout_prev, times the mask if
 *   mask_dskip (the end conv's backward-data of block k).  wb_start / wb_end: the convolutions' packed backward weights.
 * The ActNorm / InvConv parameter gradients leave as per-workgroup partials in `partial`
 * (B * ceil(T / 32) * (C / n_split) * (2 n_split + n_split^2) floats); glowtts_flow_boundary_bwd_reduce ADDS their sums (and the log-determinant
 * terms, as glowtts_actnorm_invconv_bwd does) into dlogs (C), dbias (C), dw (n, n) — any stream that waits for the first launch.
 * Limits as glowtts_flow_boundary_fwd. */
int glowtts_flow_boundary_bwd(const float *dx_wn, const float *wb_start, const float *dy_next, const float *y_prev, const float *out_prev,
                              const float *mask, const float *logs, const float *bias, const float *w, const float *dlogdet,
                              const float *wb_end, float *dy_prev, float *dout_prev, float *dskip, float *partial, int B, int C, int H,
                              int T, int n_split, int sigmoid_scale, int mask_dskip, glowtts_stream_t stream);
int glowtts_flow_boundary_bwd_reduce(const float *partial, const float *w_inv, const float *dlogdet, const float *x_len, float *dlogs,
                                     float *dbias, float *dw, int B, int C, int T, int n_split, glowtts_stream_t stream);

/* ---- the affine apply of block k fused with ActNorm + InvConvNear of block k + 1 (round 4; attentions.py:128-142 followed by
 * layers.py:182-199, 238-272): adjacent element-wise passes over the same flow tensor.  fp32 tensors, n_split in {2, 4}.
 * fwd: z = [y0 ; (m + e^logs' y1) mask] with (m, logs') = out_prev is formed in registers and never written;
 *      y = W ((bias + e^logs z) mask) mask ; logdet_prev[b] += sum logs' mask (accumulated) ; logdet[b] = (sum logs + logdet_w C/n)
 *      x_len[b] (written).  logdet_w / w_inv from glowtts_invconv_prepare.
 * bwd: dz = gradient of y.  z is recomputed from (y_prev, out_prev); dlogs / dbias / dw accumulated as by
 *      glowtts_actnorm_invconv_bwd; then the coupling's backward on the gradient of z: dy_prev = [dz0 ; dz1 e^logs' mask] (written;
 *      the start conv's input gradient is added to its first half afterwards), dout_prev = [dz1 mask ; d logs'] (written). */
int glowtts_coupling_actnorm_invconv_fwd(const float *y_prev, const float *out_prev, const float *mask, const float *logs,
                                         const float *bias, const float *w, const float *logdet_w, const float *x_len, float *y,
                                         float *logdet_prev, float *logdet, int B, int C, int T, int n_split, int sigmoid_scale,
                                         glowtts_stream_t stream);
int glowtts_coupling_actnorm_invconv_bwd(const float *y_prev, const float *out_prev, const float *mask, const float *logs,
                                         const float *bias, const float *w, const float *w_inv, const float *dz,
                                         const float *dlogdet, const float *x_len, float *dy_prev, float *dout_prev, float *dlogs,
                                         float *dbias, float *dw, int B, int C, int T, int n_split, int sigmoid_scale,
                                         glowtts_stream_t stream);


/* ---- affine coupling apply (attentions.py:128-142) ----------------------------------------------------------
 * x   : (B, C, T) flow input; out : (B, C, T) = end-conv output, rows [0,C/2) = m, [C/2,C) = logs
 * fwd : z[:, :C/2] = x[:, :C/2] ; z[:, C/2:] = (m + exp(logs') * x1) * mask ; logdet[b] += sum logs' * mask
 *       logs' = log(1e-6 + sigmoid(logs + 2)) if sigmoid_scale else logs      (logdet accumulated)
 * rev : z[:, C/2:] = (x1 - m) * exp(-logs') * mask
 * bwd : dx[:, :C/2] = dz0 ; dx1 = dz1 * exp(logs') * mask ; dm = dz1 * mask ;
 *       dlogs' = (dz1 * exp(logs') * x1 + dlogdet[b]) * mask, chained through sigmoid_scale ; dout = [dm ; dlogs] */
int glowtts_coupling_fwd(const float *x, const float *out, const float *mask, float *z, float *logdet,
                         int B, int C, int T, int sigmoid_scale, int reverse, glowtts_stream_t stream);
int glowtts_coupling_bwd(const float *x, const float *out, const float *mask, const float *dz,
                         const float *dlogdet, float *dx, float *dout, int B, int C, int T, int sigmoid_scale,
                         glowtts_stream_t stream);

/* ---- WN gate (utils.py:31-38) and residual/skip update (layers.py:157-161) ----------------------------------
 * gate fwd: acts[b,c,t] = tanh(a[b,c,t] + g[b,c]) * sigmoid(a[b,H+c,t] + g[b,H+c]) ; a (B,2H,T), g (B,2H) or NULL
 * gate bwd: da (B,2H,T) from dacts (B,H,T); the activations are recomputed from a, g (nothing else is saved).
 *           (the conditioning gradient is the row sum of da over t, left to the caller: it exists only with speakers)
 * res_skip fwd: x_out = (x + rs[:, :H]) * mask ; skip_out = skip_in + rs[:, H:]       (rs is (B,2H,T))
 *               last layer (`last` = 1): rs is (B,H,T) and skip_out = (skip_in + rs) * mask, x_out untouched (may be NULL)
 *               skip_in may be NULL (= 0); skip_out may alias skip_in.
 * res_skip bwd: drs[:, :H] = dx_out * mask ; drs[:, H:] = dskip ; dx = dx_out * mask
 *               last: drs = dskip * mask (H rows) ; dx / dx_out unused (may be NULL) */
int glowtts_gate_fwd(const float *a, const float *g, float *acts, int B, int H, int T, glowtts_stream_t stream);
int glowtts_gate_bwd(const float *a, const float *g, const float *dacts, float *da, int B, int H, int T,
                     glowtts_stream_t stream);
int glowtts_res_skip_fwd(const float *x, const float *rs, const float *mask, const float *skip_in, float *x_out,
                         float *skip_out, int B, int H, int T, int last, glowtts_stream_t stream);
int glowtts_res_skip_bwd(const float *dx_out, const float *dskip, const float *mask, float *dx, float *drs, int B,
                         int H, int T, int last, glowtts_stream_t stream);

/* ---- fp32-MFMA implicit-GEMM 1-D convolutions (csrc/convgemm.hip) --------------------------------------------
 * replace F.conv1d / conv-transpose / weight-gradient calls of WN and the coupling 1x1 convs (layers.py:146,156;
 * attentions.py:124-126) — MIOpen / rocBLAS through PyTorch in the reference.  All tensors (B, C, T), T contiguous;
 * `*_bs` = batch stride in elements (lets a channel slice such as x[:, :C/2] be consumed in place).
 * Packed weights ("k-packed": 16 consecutive reduction channels per 64-byte row, zero beyond the channel count):
 *   wp_f[tap][ceil(Cin/16)][Cout][16] (forward) and wp_b[taps-1-tap][ceil(Cout/16)][Cin][16] (backward-data), produced by
 * glowtts_pack_weight, which also applies torch's weight_norm (w = g v / ||v||, dim 0) when g != NULL.  The weight
 * gradient (conv_wrw) is produced in the plain [tap][Cin][Cout] order that unpack_weight_grad consumes.
 *
 * conv_fwd      : y[b,m,t] = sum_tap sum_k wp[tap][k][m] * (x[b,k,t + tap*dil - pad] * (mask_in ? mask : 1)) + bias[m]
 *                 then (* mask if mask_out) ; if addend != NULL: y += addend (* mask if mask_add)
 *                 (backward-data = the same call with wp_b, Cin/M swapped, pad' = (taps-1)*dil - pad)
 * conv_gate_fwd : WN in-layer + gate: pre = conv(x) + bias, dropout keep-mask bytes `drop` (scale 1/(1-p)) applied to
 *                 pre, + cond[b, :] ; acts = tanh(pre[:H]) * sigmoid(pre[H:]) ; ts (B,2H,T) = the tanh / sigmoid values
 * conv_res_skip_fwd : WN res/skip 1x1 + update: rs = W acts + b ; x_out = (x_in + rs[:H]) * mask ; skip_out = skip_in + rs[H:]
 *                 last = 1: W is (H x H), skip_out = (skip_in + rs) * mask   (folds layers.py:162)
 * conv_wrw      : dwp[tap][k][m] += sum_{b,t} x[b,k,t'] (* mask_x[b,t'] if mask_x) * d[b,m,t] (* mask[b,t] if mask),
 *                 t' = t + tap*dil - pad   (accumulated)
 *                 and, if dbias != NULL, dbias[m] += sum_{b,t} d[b,m,t] (* mask)  (the bias gradient, same pass)
 * unpack_weight_grad : packed gradient -> dv (+= , weight layout [Cout][Cin][taps]) and dg (+=) through the weight norm
 * rowsum        : out[m] += sum_{b,t} d[b,m,t] (* mask)                                  (bias gradients, accumulated)
 * gate_bwd_ts   : da (B,2H,T) from dacts (B,H,T) and the saved ts; the forward's dropout mask is re-applied
 * conv_gate_bwd : the backward of "res/skip 1x1 conv after the gate" in one kernel (layers.py:152-156 backwards):
 *                 dacts = W_rs^T d_rs (backward-data of the 1x1 conv, packed wp_b, d_rs is (B, M_rs, T)), then
 *                 gate_bwd_ts in the epilogue -> d_pre (B,2H,T); dacts never reaches HBM */
int glowtts_conv_fwd(const float *x, long x_bs, const float *wp, const float *bias, const float *mask,
                     const float *addend, long addend_bs, float *y, long y_bs, int B, int Cin, int M, int T, int taps,
                     int dil, int pad, int mask_in, int mask_out, int mask_add, glowtts_stream_t stream);
int glowtts_conv_gate_fwd(const float *x, const float *wp, const float *bias, const float *cond,
                          const unsigned char *drop, float drop_scale, float *acts, float *ts, int B, int H, int T,
                          int taps, int dil, int pad, glowtts_stream_t stream);
int glowtts_conv_res_skip_fwd(const float *acts, const float *wp, const float *bias, const float *mask,
                              const float *x_in, const float *skip_in, float *x_out, float *skip_out, int B, int H, int T,
                              int last, glowtts_stream_t stream);
int glowtts_conv_gate_bwd(const float *d_rs, const float *d_rs2, const float *wp_b, const float *ts,
                          const unsigned char *drop, float drop_scale, float *d_pre, int B, int M_rs, int H, int T,
                          glowtts_stream_t stream);
/* two-source forms: d_rs = [dx_next * mask ; dskip] of a WN layer is never concatenated in memory — conv_gate_bwd reads
 * rows [0,H) from d_rs (B,H,T) and rows [H,2H) from d_rs2 (B,H,T) when d_rs2 != NULL, and conv_wrw2 takes the output
 * gradient rows [0,d_split) from d and [d_split,M) from d2 (d_split % 64 == 0; dilation 1, 'same' padding, no masks) */
int glowtts_conv_wrw2(const float *x, long x_bs, const float *d, long d_bs, const float *d2, long d2_bs, int d_split,
                      float *dwp, float *dbias, int B, int Cin, int M, int T, int taps, int dil, int pad,
                      glowtts_stream_t stream);
int glowtts_conv_wrw(const float *x, long x_bs, const float *d, long d_bs, const float *mask, const float *mask_x,
                     float *dwp, float *dbias, int B, int Cin, int M, int T, int taps, int dil, int pad,
                     glowtts_stream_t stream);
/* n weight gradients of ONE shape in one launch (the layers of a WN stack, whose operands all exist once the stack's dx chain has
 * run): problem q is glowtts_conv_wrw (d2 == NULL) or glowtts_conv_wrw2 on (x[q], d[q], d2[q]) -> dwp[q], dbias[q] (dbias or its
 * entries may be NULL); mask / mask_x as in glowtts_conv_wrw, shared by the problems (single-source form only).  x, d, d2, dwp, dbias are HOST arrays of n device pointers.  In the bf16-plane arithmetic the
 * problems share a launch (the next problem's workgroups start while the previous one's split-K atomics drain); otherwise, and
 * for shapes without such a kernel, they are launched one by one — the results are the same either way. */
int glowtts_conv_wrw_batch(int n, const float *const *x, long x_bs, const float *const *d, long d_bs, const float *const *d2,
                           long d2_bs, int d_split, const float *mask, const float *mask_x, float *const *dwp,
                           float *const *dbias, int B, int Cin, int M, int T, int taps, int dil, int pad, glowtts_stream_t stream);
/* n weight gradients of 1x1 convolutions of DIFFERENT shapes in one launch (autograd of the 1x1 convolutions of layers.py:155-156
 * and attentions.py:97-113, 128-129 — a flow block's three two-source res/skip gradients, its last layer's, the start conv's and
 * the end conv's — and of the encoder's q / k / v / o projections, attentions.py:204-211): problem q is glowtts_conv_wrw (d2 ==
 * NULL; mask_d multiplies d, mask_x multiplies x, either may be NULL) or glowtts_conv_wrw2 (rows [d_split, M) of the output
 * gradient from d2, d_split % 64 == 0, T % 4 == 0, no masks) with taps = 1; all problems share B and T (the one-launch form needs
 * T % 4 == 0 and 16-byte aligned rows; otherwise the problems are launched one by one).
 * `problems` is a HOST array.  In the bf16-plane arithmetic (glowtts_conv_math weight-gradient mode 3) the problems' 192 x 192
 * tiles share one round of workgroups (csrc/convwrw1.hip); otherwise they are launched one by one: same results either way. */
typedef struct glowtts_wrw1_problem {
    const float *x;          /* (B, Cin, T), batch stride x_bs elements */
    const float *d;          /* (B, M, T) — or (B, d_split, T) when d2 != NULL —, batch stride d_bs */
    const float *d2;         /* NULL, or (B, M - d_split, T): rows [d_split, M) of the output gradient, batch stride d2_bs */
    const float *mask_d;     /* NULL or (B, T) */
    const float *mask_x;     /* NULL or (B, T) */
    float *dwp;              /* [Cin][M] accumulated */
    float *dbias;            /* NULL or [M] accumulated row sums of the (masked) output gradient */
    long x_bs, d_bs, d2_bs;
    int Cin, M, d_split, reserved;
} glowtts_wrw1_problem;
int glowtts_conv_wrw1_multi(int n, const glowtts_wrw1_problem *problems, int B, int T, glowtts_stream_t stream);
int glowtts_pack_weight(const float *v, const float *g, float *wp_f, float *wp_b, float *inv_norm, int Cout, int Cin,
                        int taps, glowtts_stream_t stream);

/* Arithmetic of the WN convolutions (convgemm_split.hip): fp32 operands split into bf16 planes, products on the
 * bf16 matrix pipe, fp32 accumulation.  mode 0 = native fp32 MFMA (the LIBRARY's initial mode; the Python host selects
 * 3 + 4 * 3, "bf16x6+wrw", at import — convops.DEFAULT_CONV_MATH), 1 = bf16 operands,
 * 2 = bf16x3 (products good to 2^-16), 3 = bf16x6 (fp32-equivalent: dropped terms <= 2^-24 |x w|); the weight-gradient
 * kernel has its own code in bits 2-3 (mode = forward_code + 4 * wrw_code).  A negative mode only returns the current one.
 *   glowtts_conv_split_weights: planes[pl * n + i] = plane pl of wp[i] (caller-owned, 3 * n uint16; call after every
 *     re-packing — the planes are a snapshot of the weights);
 *   glowtts_conv_bind_planes: the CALLING THREAD's forward-type convolutions whose packed weights lie in [wp, wp + n) use
 *     these planes until the next bind (wp = NULL unbinds); everything else, and shapes without a split instantiation,
 *     runs native.  The weight-gradient kernel needs no planes (both its operands are activations). */
int glowtts_conv_math(int mode);

/* The forward of a WN stack as ONE layer-resident kernel (csrc/wn_fused.hip; reference layers.py:138-162): when switched on,
 * taken by glowtts_wn_fwd / glowtts_flow_block_fwd whenever it applies — fp32 tensors, no conditioning input, H = 192, 5 taps,
 * dilation 1, <= 4 layers, T % 4 == 0, arithmetic mode bf16x6 with the stack's planes bound — and otherwise the per-layer launch
 * sequence runs; results agree to fp32 round-off (same six-product arithmetic, another summation order).  Initially OFF (at
 * the benchmark's shape it is at parity with the per-layer launches) unless GLOWTTS_WN_FUSED=1.
 * enable: 1 / 0 sets the process-wide switch, -1 only queries; returns the
 * setting in force BEFORE the call.  enable = -2 returns the number of launches of the kernel so far in this process (tests
 * use it to see that the kernel, not the fallback, ran). */
int glowtts_wn_fused(int enable);
int glowtts_conv_split_weights(const float *wp, long n_floats, uint16_t *planes, glowtts_stream_t stream);
int glowtts_conv_bind_planes(const float *wp, long n_floats, const uint16_t *planes);

/* The weight gradient from operands that are bf16 planes already (3 planes: the bf16x6 arithmetic above).  Splitting costs
 * ~15 vector instructions per value and the weight-gradient tile reuses a staged value in few MFMAs, so the split is done
 * ONCE per tensor (glowtts_split_planes: planes[pl * n + i] = plane pl of x[i]; later by the producing kernel's epilogue)
 * instead of once per workgroup.  x_planes / d_planes index like the fp32 tensors (B, Cin, T) / (B, M, T) with batch strides
 * x_bs / d_bs in elements; dilation 1, 'same' padding, taps 5 (M % 32 == 0) or 1; dwp [taps][Cin][M] and dbias [M] are
 * accumulated (atomics) as in glowtts_conv_wrw (reference: autograd of layers.py:143,153 F.conv1d). */
int glowtts_split_planes(const float *x, long n, uint16_t *planes, int n_planes, glowtts_stream_t stream);
int glowtts_conv_wrw_planes(const uint16_t *x_planes, long x_plane_stride, long x_bs, const uint16_t *d_planes,
                            long d_plane_stride, long d_bs, float *dwp, float *dbias, int B, int Cin, int M, int T, int taps,
                            int n_planes, glowtts_stream_t stream);
int glowtts_unpack_weight_grad(const float *dwp, const float *v, const float *g, const float *inv_norm, float *dv,
                               float *dg, int Cout, int Cin, int taps, glowtts_stream_t stream);
int glowtts_rowsum(const float *d, long d_bs, const float *mask, float *out, int B, int M, int T,
                   glowtts_stream_t stream);
/* several convolutions per launch (a WN stack: 2 per layer).  desc = n_conv rows of int64:
 *   pack  : {v, g, wp_f, wp_b, inv_norm, Cout, Cin, taps}            unpack: {dwp, v, g, inv_norm, dv, dg, Cout, Cin, taps}
 * row_prefix[c] = sum of Cout of the convolutions before c (int32, n_conv + 1 entries); both tables in DEVICE memory. */
int glowtts_pack_weight_multi(const long long *desc, const int *row_prefix, int n_conv, int total_rows,
                              glowtts_stream_t stream);
/* The same, and the three bf16 planes of every packed value in the same pass (what glowtts_conv_split_weights would make of
 * the packed buffers afterwards): all wp_f / wp_b of the table lie inside [arena, arena + n_floats), planes is 3 * n_floats
 * uint16 with planes[pl * n_floats + (p - arena)] = plane pl of *p.  Positions of the arena that no packing covers are not
 * written.  (csrc/packw.hip: 16 output channels per workgroup, so both packings are written — and the packed gradient is read
 * by the un-packing — as whole 64-byte segments; packed buffers must be 8-byte aligned.) */
int glowtts_pack_weight_planes_multi(const long long *desc, const int *row_prefix, int n_conv, int total_rows,
                                     const float *arena, long n_floats, uint16_t *planes, glowtts_stream_t stream);
int glowtts_unpack_weight_grad_multi(const long long *desc, const int *row_prefix, int n_conv, int total_rows,
                                     glowtts_stream_t stream);
int glowtts_gate_bwd_ts(const float *ts, const float *dacts, const unsigned char *drop, float drop_scale, float *da,
                        int B, int H, int T, glowtts_stream_t stream);

/* ---- a whole WN stack per call (csrc/wn_stack.hip): the host-side launch sequence of layers.py:134-162 ------------
 * One coupling block's gated conv stack without conditioning input (g == None): n_layers x [k-tap dilated conv + gate,
 * 1x1 res/skip conv], dilation dil_rate^i, 'same' padding, hidden width H.  `layers` is a HOST array.
 * fwd : x (B,H,T) -> skip (B,H,T) = WN output (mask folded in); slabs written for the backward:
 *       xs (n_layers-1, B,H,T) inputs of layers 1.., acts (n_layers, B,H,T), ts (n_layers, B,2H,T) stored tanh / sigmoid;
 *       drop (n_layers, B,2H,T) keep-mask bytes or NULL, drop_scale = 1/(1-p)
 * bwd : dskip (B,H,T) = gradient of the output -> dx slab (n_layers, B,H,T), dx[0] = gradient of x; workspaces
 *       d_rs, d_xin (n_layers, B,2H,T); packed weight gradients accumulate into layers[i].dwp_* (zeroed by the caller),
 *       bias gradients into layers[i].db_*; the weight-gradient kernels and, if unpack_desc != NULL, the final
 *       glowtts_unpack_weight_grad_multi(unpack_desc, unpack_prefix, n_conv, total_rows) run on wgrad_stream behind
 *       events (NULL: everything on `stream`).  The caller keeps every buffer alive until wgrad_stream has drained.
 *       two_source: bit 0 = the two-source launch sequence (d_rs = [dx mask ; dskip] is never materialised; fp32 tensors,
 *       dilation 1, H % 192 == 0, T % 4 == 0); bit 1 (with bit 0) = `dskip` is masked already (glowtts_flow_block_bwd masks it in
 *       the end conv's backward-data epilogue), so the last layer's dskip * mask pass is not queued. */
typedef struct glowtts_wn_layer {
    const float *wf_in, *wb_in, *b_in;   /* packed forward / backward-data weights and bias of the k-tap in-conv */
    const float *wf_rs, *wb_rs, *b_rs;   /* ... of the 1x1 res/skip conv (2H rows, H in the last layer) */
    float *dwp_in, *dwp_rs;              /* packed weight-gradient accumulators [taps][H][2H] / [1][H][2H or H] */
    float *db_in, *db_rs;                /* bias gradients, accumulated */
} glowtts_wn_layer;
int glowtts_wn_fwd(const glowtts_wn_layer *layers, int n_layers, const float *x, const float *mask,
                   const unsigned char *drop, float drop_scale, float *xs, float *acts, float *ts, float *skip, int B,
                   int H, int T, int taps, int dil_rate, glowtts_stream_t stream);
int glowtts_wn_bwd(const glowtts_wn_layer *layers, int n_layers, const float *x, const float *xs, const float *acts,
                   const float *ts, const float *mask, const unsigned char *drop, float drop_scale, const float *dskip,
                   float *d_rs, float *d_xin, float *dx, const long long *unpack_desc, const int *unpack_prefix,
                   int n_conv, int total_rows, int B, int H, int T, int taps, int dil_rate, int two_source,
                   glowtts_stream_t wgrad_stream, glowtts_stream_t stream);

/* ---- a whole flow block per call (csrc/wn_stack.hip): [ActNorm, InvConvNear, CouplingBlock], models.py:176-190 ---------
 * The launch sequence of one decoder block in the forward (training) direction, without conditioning input, n_split in
 * {2, 4}.  `blk` is a HOST struct of DEVICE pointers; `layers` inside it a host array as for glowtts_wn_fwd.
 * fwd : x (B,C,T) -> z (B,C,T), logdet (B) WRITTEN (ActNorm + InvConv + coupling terms summed).  Packs every convolution's
 *       weights first (pack_desc: 2 + 2 n_layers rows, start / end / WN in-conv and res-skip per layer; NULL = the caller
 *       has packed them), factorises W
 *       into blk->w_inv / blk->logdet_w (kept for the backward), and writes the slabs the backward reads:
 *       y (B,C,T) after the invertible 1x1, h0 (B,H,T) start-conv output, xs / acts / ts / skip as glowtts_wn_fwd,
 *       out (B,C,T) end-conv output (m ; logs).
 * bwd : dz (B,C,T), dlogdet (B) -> dx (B,C,T).  Workspaces: dy, dout (B,C,T), dskip (B,H,T), d_rs / d_xin / dx_wn as
 *       glowtts_wn_bwd.  Packed weight gradients accumulate in blk->dwp_all (cleared here), are un-packed through the weight
 *       norm into the parameters' gradient buffers by unpack_desc, bias gradients go to db_*, ActNorm / InvConv gradients
 *       are ACCUMULATED into dlogs, dbias (C) and dw (n*n).  Weight-gradient kernels and the un-packing run on wgrad_stream
 *       (NULL: `stream`); once everything queued there has run, every parameter gradient of the block is complete. */
typedef struct glowtts_flow_block {
    const float *logs, *bias, *w;                 /* ActNorm (C each), InvConvNear weight (n x n) */
    float *w_inv, *logdet_w;                      /* n*n + 1 floats written by the forward, read by the backward */
    const float *wf_start, *wb_start, *b_start;   /* packed 1x1 start conv (C/2 -> H), bias */
    const float *wf_end, *wb_end, *b_end;         /* packed 1x1 end conv (H -> C), bias */
    float *dwp_start, *dwp_end, *db_start, *db_end;
    float *dlogs, *dbias, *dw;
    const glowtts_wn_layer *layers;               /* host array, n_layers entries */
    const long long *pack_desc, *unpack_desc;     /* device tables (glowtts_pack_weight_multi / _unpack_weight_grad_multi) */
    const int *pack_prefix;                       /* device, n_conv + 1 entries */
    float *dwp_all;                               /* all packed weight-gradient accumulators of the block, contiguous */
    long long dwp_floats;
    int n_layers, n_conv, total_rows;
    int reserved;                                 /* forward only: > B = utterances per LAYER slab of xs / acts / ts / drop when a call covers
                                                     only B of them (two half-batch forward chains writing into one set of slabs); else 0 */
} glowtts_flow_block;
int glowtts_flow_block_fwd(const glowtts_flow_block *blk, const float *x, const float *mask, const float *x_len,
                           const unsigned char *drop, float drop_scale, float *y, float *h0, float *xs, float *acts,
                           float *ts, float *skip, float *out, float *z, float *logdet, int B, int C, int H, int T, int taps,
                           int dil_rate, int n_split, int sigmoid_scale, glowtts_stream_t stream);
int glowtts_flow_block_bwd(const glowtts_flow_block *blk, const float *x, const float *mask, const float *x_len,
                           const unsigned char *drop, float drop_scale, const float *y, const float *h0, const float *xs,
                           const float *acts, const float *ts, const float *skip, const float *out, const float *dz,
                           const float *dlogdet, float *dy, float *dout, float *dskip, float *d_rs, float *d_xin,
                           float *dx_wn, float *dx, int B, int C, int H, int T, int taps, int dil_rate, int n_split,
                           int sigmoid_scale, int two_source, glowtts_stream_t wgrad_stream, glowtts_stream_t stream);

/* ---- text-encoder neighbours folded into kernels, and a whole transformer layer per call -------------------------------
 * conv_fwd_act : conv_fwd (fp32 tensors) with the elementwise ops around the encoder's convolutions in the epilogue:
 *   y = dropout(relu((conv(x) [gated] [+ addend]) [* mask]))  — relu flag; drop = keep bytes (B, M, T) or NULL with drop_scale;
 *   gate_pos (B, M, T) or NULL: the conv result is multiplied by gate_scale where gate_pos > 0 and zeroed elsewhere, BEFORE
 *   the addend — the backward of "ReLU then dropout" in one test when gate_pos is that layer's output
 *   (reference attentions.py:373-381, layers.py:73-80).
 * chan_layernorm_{fwd,bwd}_ex : the value normalised is x * mask_x[b, t] + res * keep * drop_scale (mask_x, drop may be NULL):
 *   the `x * x_mask` that opens a transformer layer and the dropout on the branch that is added back (attentions.py:64-72)
 *   never reach HBM; bwd: dx = dv * mask_x and, if dres != NULL, dres = dv * keep * drop_scale.
 * encoder_layer_{fwd,bwd} (csrc/wn_stack.hip): one post-LN transformer layer of attentions.Encoder (attentions.py:63-73 with
 *   MultiHeadAttention :204-264 and FFN :373-381) as ONE host call each way.  `L` is a HOST struct of DEVICE pointers (packed
 *   weights as for conv_fwd: q, k, v, o are 1x1, the FFN convs `taps` wide with 'same' padding, F = filter channels).
 *   fwd writes the slabs the backward reads: q, k, v, y_att, o, x1, h, y2, x2 (B, ., T), p_attn (B, heads, T, T), stats1/2
 *   (B, 2, T); drop_a (B, heads, T, T), drop_o / drop_2 (B, H, T), drop_h (B, F, T) are keep bytes (all NULL = no dropout).
 *   bwd: dx2 must vanish beyond each utterance (it does inside Encoder: every consumer of a layer's output masks it first);
 *   workspaces dx1a, dy2, dx1, dxa, d_o, dy_att, dq, dkk, dv (B, H, T), d_pre1 (B, F, T), ds (B, heads, T, T); dx (B, H, T) out.
 *   Parameter gradients: packed weight gradients accumulate in L->dwp_all (cleared here) and are un-packed by unpack_desc on
 *   wgrad_stream; biases, embeddings and LayerNorm parameters are ACCUMULATED into db_*, demb_*, dgamma*, dbeta*. */
int glowtts_conv_fwd_act(const float *x, long x_bs, const float *wp, const float *bias, const float *mask,
                         const float *addend, long addend_bs, float *y, long y_bs, int B, int Cin, int M, int T, int taps,
                         int dil, int pad, int mask_in, int mask_out, int mask_add, int relu, const unsigned char *drop,
                         float drop_scale, const float *gate_pos, float gate_scale, glowtts_stream_t stream);
int glowtts_chan_layernorm_fwd_ex(const float *x, const float *res, const float *mask_x, const unsigned char *drop,
                                  float drop_scale, const float *gamma, const float *beta, float *y, float *stats, int B,
                                  int C, int T, float eps, glowtts_stream_t stream);
int glowtts_chan_layernorm_bwd_ex(const float *x, const float *res, const float *mask_x, const unsigned char *drop,
                                  float drop_scale, const float *gamma, const float *stats, const float *dy, float *dx,
                                  float *dres, float *dgamma, float *dbeta, int B, int C, int T, glowtts_stream_t stream);
/* `_act` forms (reference layers.py:73-80: pre-net conv -> LayerNorm -> ReLU -> Dropout; models.py:44-50: duration predictor
 * conv -> ReLU -> LayerNorm -> Dropout): relu_in: the value normalised is relu(x) * mask_x (+ res ...) and dx is gated by x > 0;
 * relu_out / (odrop (B, C, T) keep bytes, oscale): y = dropout(relu(LN(v))), the backward gates dy the same way (relu_out: through
 * the sign of the stored y, which it then needs).  Everything else as the `_ex` forms. */
int glowtts_chan_layernorm_fwd_act(const float *x, const float *res, const float *mask_x, const unsigned char *drop,
                                   float drop_scale, const float *gamma, const float *beta, float *y, float *stats, int relu_in,
                                   int relu_out, const unsigned char *odrop, float oscale, int B, int C, int T, float eps,
                                   glowtts_stream_t stream);
int glowtts_chan_layernorm_bwd_act(const float *x, const float *res, const float *mask_x, const unsigned char *drop,
                                   float drop_scale, const float *gamma, const float *stats, const float *y, const float *dy,
                                   int relu_in, int relu_out, const unsigned char *odrop, float oscale, float *dx, float *dres,
                                   float *dgamma, float *dbeta, int B, int C, int T, glowtts_stream_t stream);
/* Phoneme embedding (reference models.py:90,121: self.emb(x) * sqrt(hidden), transposed to (B, H, T)) and its backward:
 * out[b][h][t] = weight[ids[b][t]][h] * scale;  dweight[v][h] += scale * sum over the positions holding id v of dout[b][h][t]
 * (ids int64 (B, T); one workgroup per vocabulary entry, no atomics, no sort). */
/* Dropout keep-masks for many tensors from one launch: out[i] = 1 with probability 1 - p_drop (else 0), i < n; Philox4x32-7 keyed
 * by `seed`, counter = byte group index, so a (seed, n) pair always gives the same mask.  Stands in for the generator behind
 * F.dropout (layers.py:147, attentions.py:248): the kernels that apply dropout take such byte masks.  `out` 8-byte aligned. */
int glowtts_keep_mask(unsigned char *out, long n, unsigned long long seed, float p_drop, glowtts_stream_t stream);
int glowtts_embed_fwd(const long long *ids, const float *weight, float scale, float *out, int B, int T, int H, int V,
                      glowtts_stream_t stream);
int glowtts_embed_bwd(const long long *ids, const float *dout, float scale, float *dweight, int B, int T, int H, int V,
                      glowtts_stream_t stream);
typedef struct glowtts_enc_layer {
    const float *wf_q, *wb_q, *b_q, *wf_k, *wb_k, *b_k, *wf_v, *wb_v, *b_v, *wf_o, *wb_o, *b_o;   /* attention 1x1 convs */
    const float *wf_1, *wb_1, *b_1, *wf_2, *wb_2, *b_2;                                           /* FFN convs */
    const float *emb_k, *emb_v;                       /* relative-position embeddings (1 or heads, 2w+1, dk) or NULL */
    const float *gamma1, *beta1, *gamma2, *beta2;     /* the two LayerNorms */
    float *dwp_q, *dwp_k, *dwp_v, *dwp_o, *dwp_1, *dwp_2;
    float *db_q, *db_k, *db_v, *db_o, *db_1, *db_2;
    float *demb_k, *demb_v, *dgamma1, *dbeta1, *dgamma2, *dbeta2;
    const long long *pack_desc, *unpack_desc;         /* device tables (6 rows) or pack_desc = NULL: already packed */
    const int *pack_prefix;
    float *dwp_all;
    long long dwp_floats;
    int n_conv, total_rows;
    int attn_bf16;                                    /* 1: the attention contractions on the bf16 MFMA (glowtts_rel_attn_*_ex) */
} glowtts_enc_layer;
int glowtts_encoder_layer_fwd(const glowtts_enc_layer *L, const float *x, const float *mask, const unsigned char *drop_a,
                              const unsigned char *drop_o, const unsigned char *drop_h, const unsigned char *drop_2,
                              float drop_scale, float *q, float *k, float *v, float *p_attn, float *y_att, float *o, float *x1,
                              float *stats1, float *h, float *y2, float *x2, float *stats2, int B, int H, int F, int T,
                              int heads, int taps, int window, int heads_share, int block_len, float eps,
                              glowtts_stream_t stream);
int glowtts_encoder_layer_bwd(const glowtts_enc_layer *L, const float *x, const float *mask, const unsigned char *drop_a,
                              const unsigned char *drop_o, const unsigned char *drop_h, const unsigned char *drop_2,
                              float drop_scale, const float *q, const float *k, const float *v, const float *p_attn,
                              const float *y_att, const float *o, const float *x1, const float *stats1, const float *h,
                              const float *y2, const float *stats2, const float *dx2, float *dx1a, float *dy2, float *d_pre1,
                              float *dx1, float *dxa, float *d_o, float *dy_att, float *ds, float *dq, float *dkk, float *dv,
                              float *dx, int B, int H, int F, int T, int heads, int taps, int window, int heads_share,
                              int block_len, glowtts_stream_t wgrad_stream, glowtts_stream_t stream);

/* ---- bf16 tensors in HBM (BASELINE configs[2]): the `_io` forms ---------------------------------------------------------
 * Same operators, same shapes and element indexing; `io` (or io_x / io_y) = 1 declares the ACTIVATION tensors bf16 in HBM
 * (void pointers), 0 = fp32 (then each is exactly the function without the suffix).  What stays fp32 in either case:
 * parameters, packed weights, biases, masks, conditioning rows, log-determinants, the coupling's (m, logs) = `out`, every
 * parameter gradient, and all arithmetic: bf16 operands go through v_mfma_f32_16x16x32_bf16 with fp32 accumulation, the
 * elementwise flows widen to fp32 in registers, results are rounded to nearest even on the way out
 * (reference: the autocast branch train.py:116-121, whose conv / matmul outputs are half precision).
 * Convolutions read bf16 planes of the packed weights: glowtts_split_planes(wp, n, planes, 3) after every packing, then
 * glowtts_conv_bind_planes_ns(wp, n, planes, 3) on the launching thread (wp = NULL unbinds).  Plane 0 is the weight rounded
 * to bf16 and is all a convolution with bf16 results reads; one with fp32 results (io_y = 0: the coupling's end conv, whose
 * `logs` rows are summed into the log-determinant) multiplies the bf16 activations by all three planes, i.e. by the exact
 * fp32 weights — a weight rounding repeats in every frame and would add up coherently in that sum.  The weight gradient for
 * bf16 tensors is glowtts_conv_wrw_planes with n_planes = 1 (the "plane" is the tensor itself; no masks: callers pass
 * gradients that are masked already).  Limits: T % 4 == 0, dilation 1 for the weight gradient; shapes of a flow block.
 *   conv_fwd_io        : io_x = x (and the implied second source), io_y = y and addend
 *   conv_gate_fwd_io / conv_res_skip_fwd_io / conv_gate_bwd_io / res_skip_bwd_io : io = every activation tensor
 *   actnorm_invconv_*_io, coupling_*_io : io = the flow tensors x, z, dz, dx; `out` stays fp32; coupling_bwd's io_dout = its
 *       dout (the end conv's output gradient: a hidden-side tensor); actnorm_invconv_fwd's z0h (may be NULL) receives a
 *       bf16 copy (B, C/2, T) of z's first half for a bf16 start conv behind an fp32 flow tensor
 *   squeeze_io / unsqueeze_io : io = the SQUEEZED tensor only (mel frames / the latent handed to the loss stay fp32)
 *   wn_*_io : io = every slab; wn_bwd_io's mask_input_grad = 1 multiplies the stack's own input gradient by the mask
 *       (required with io = 1, where the two-source form is not available).  Speaker conditioning (layers.py:142-153): cond
 *       (n_layers, B, 2H) fp32 or NULL = the rows added to each layer's pre-activation AFTER its dropout; dcond (same shape,
 *       ACCUMULATED, may be NULL) = their gradient, the row sums over t of the un-dropped pre-activation gradient, reduced
 *       in the gate-backward kernel's epilogue (conv_gate_bwd_io's dcond: one layer's (B, 2H))
 *   flow_block_*_io : io bit 0 = the coupling network's hidden tensors (h0, xs, acts, ts, skip, their gradients, dout), bit 1
 *       (needs bit 0) = the flow tensor too (x, y, z, dz, dy, dx).  io = 1 keeps the invertible chain in fp32, as the
 *       reference's autocast does, and needs y0h, the bf16 copy of y's first half that the start conv reads.
 *       fp32 tensors only (io & 3 == 0), for a caller that fuses a block's affine apply with the next block's ActNorm + InvConv
 *       (glowtts_coupling_actnorm_invconv_fwd / _bwd): bits 11 / 12 = the end conv (with bit 9) / the start conv (with bit 8) — in the backward their backward-data launches — are the caller's too (glowtts_flow_boundary_fwd / _bwd); bit 10 = forward: W^-1 / log det W are in place (glowtts_invconv_prepare_multi); bit 8 = forward: y has been written by the caller (no W^-1 factorisation, no
 *       ActNorm + InvConv launch) / backward: no ActNorm + InvConv backward at the end (dx is not written); bit 9 = forward: no
 *       affine apply at the end (z is not written) / backward: dy and dout have been written by the caller (no coupling backward) */
int glowtts_conv_bind_planes_ns(const float *wp, long n_floats, const uint16_t *planes, int n_planes);

/* ---- Winograd F(4, 5) form of the gated in-conv (reference layers.py:146 + utils.py:31-38; csrc/convwino.hip) -------------------
 * out[m][4j + i] = sum_p AT[i][p] (sum_c U[m][c][p] V[c][p][j]): U = G w (8 points from the 5 taps), V = BT d (8 points from the
 * input frames 4j - 2 .. 4j + 5), points {0, +-1, +-2, +-1/2, inf}: 8 products per (row, channel, 4 frames) where the direct
 * form needs 20.  U and V are fp32 values, each split into three bf16 planes (six products per fp32 product, fp32 accumulation).
 *   wino_plane_elems(n) : bf16 elements per plane of the U planes of a packed-weight buffer of n floats
 *   wino_weights : U planes of the listed convolutions of a packed buffer; table (device, int64) rows = (offset of the convolution's
 *       packed FORWARD weights in floats, Cin / 16, M); planes = 3 x plane_stride bf16 (caller's memory); after every packing
 *   conv_bind_wino : bind (wp = NULL: unbind) those planes to the calling thread; glowtts_conv_gate_fwd then takes the
 *       Winograd kernel when GLOWTTS_WINO = 1, arithmetic mode bf16x6, fp32 tensors, 5 taps, dilation 1, H % 64 == 0, T % 4 == 0,
 *       16-byte-aligned tensors — and the direct kernels otherwise. */
long glowtts_wino_plane_elems(long n_floats);
long glowtts_wino_launches(void);        /* launches of the Winograd kernel by this process so far (tests, bench.py) */
int glowtts_wino_weights(const float *wp, long n_floats, const long *table, int n_conv, uint16_t *planes, long plane_stride,
                         glowtts_stream_t stream);
int glowtts_conv_bind_wino(const float *wp, long n_floats, const uint16_t *planes, long plane_stride);
int glowtts_conv_fwd_io(const void *x, long x_bs, const float *wp, const float *bias, const float *mask, const void *addend,
                        long addend_bs, void *y, long y_bs, int B, int Cin, int M, int T, int taps, int dil, int pad,
                        int mask_in, int mask_out, int mask_add, int io_x, int io_y, glowtts_stream_t stream);
int glowtts_conv_gate_fwd_io(const void *x, const float *wp, const float *bias, const float *cond, const unsigned char *drop,
                             float drop_scale, void *acts, void *ts, int B, int H, int T, int taps, int dil, int pad, int io,
                             glowtts_stream_t stream);
int glowtts_conv_res_skip_fwd_io(const void *acts, const float *wp, const float *bias, const float *mask, const void *x_in,
                                 const void *skip_in, void *x_out, void *skip_out, int B, int H, int T, int last, int io,
                                 glowtts_stream_t stream);
int glowtts_conv_gate_bwd_io(const void *d_rs, const void *d_rs2, const float *wp_b, const void *ts,
                             const unsigned char *drop, float drop_scale, void *d_pre, float *dcond, int B, int M_rs, int H,
                             int T, int io, glowtts_stream_t stream);
int glowtts_res_skip_bwd_io(const void *dx_out, const void *dskip, const float *mask, void *dx, void *drs, int B, int H,
                            int T, int last, int io, glowtts_stream_t stream);
int glowtts_actnorm_invconv_fwd_io(const void *x, const float *mask, const float *logs, const float *bias, const float *w,
                                   const float *logdet_w, const float *x_len, void *z, float *logdet, void *z0h, int B, int C,
                                   int T, int n_split, int io, glowtts_stream_t stream);
int glowtts_actnorm_invconv_bwd_io(const void *x, const float *mask, const float *logs, const float *bias, const float *w,
                                   const float *w_inv, const void *dz, const float *dlogdet, const float *x_len, void *dx,
                                   float *dlogs, float *dbias, float *dw, int B, int C, int T, int n_split, int io,
                                   glowtts_stream_t stream);
int glowtts_coupling_fwd_io(const void *x, const float *out, const float *mask, void *z, float *logdet, int B, int C, int T,
                            int sigmoid_scale, int reverse, int io, glowtts_stream_t stream);
int glowtts_coupling_bwd_io(const void *x, const float *out, const float *mask, const void *dz, const float *dlogdet,
                            void *dx, void *dout, int B, int C, int T, int sigmoid_scale, int io, int io_dout,
                            glowtts_stream_t stream);
int glowtts_squeeze_io(const float *x, const float *mask, void *xs, float *ms, int B, int C, int T, int n, int io,
                       glowtts_stream_t stream);
int glowtts_unsqueeze_io(const void *xs, const float *ms, float *x, float *mask_out, int B, int C, int Tsq, int n, int io,
                         glowtts_stream_t stream);
int glowtts_wn_fwd_io(const glowtts_wn_layer *layers, int n_layers, const void *x, const float *mask, const float *cond,
                      const unsigned char *drop, float drop_scale, void *xs, void *acts, void *ts, void *skip, int B, int H,
                      int T, int taps, int dil_rate, int io, glowtts_stream_t stream);
int glowtts_wn_bwd_io(const glowtts_wn_layer *layers, int n_layers, const void *x, const void *xs, const void *acts,
                      const void *ts, const float *mask, const unsigned char *drop, float drop_scale, const void *dskip,
                      void *d_rs, void *d_xin, void *dx, float *dcond, const long long *unpack_desc, const int *unpack_prefix, int n_conv,
                      int total_rows, int B, int H, int T, int taps, int dil_rate, int two_source, int mask_input_grad, int io,
                      glowtts_stream_t wgrad_stream, glowtts_stream_t stream);
int glowtts_flow_block_fwd_io(const glowtts_flow_block *blk, const void *x, const float *mask, const float *x_len,
                              const float *cond, const unsigned char *drop, float drop_scale, void *y, void *y0h, void *h0, void *xs, void *acts, void *ts,
                              void *skip, float *out, void *z, float *logdet, int B, int C, int H, int T, int taps,
                              int dil_rate, int n_split, int sigmoid_scale, int io, glowtts_stream_t stream);
int glowtts_flow_block_bwd_io(const glowtts_flow_block *blk, const void *x, const float *mask, const float *x_len,
                              const unsigned char *drop, float drop_scale, const void *y, const void *y0h, const void *h0, const void *xs,
                              const void *acts, const void *ts, const void *skip, const float *out, const void *dz,
                              const float *dlogdet, void *dy, void *dout, void *dskip, void *d_rs, void *d_xin, void *dx_wn,
                              void *dx, float *dcond, int B, int C, int H, int T, int taps, int dil_rate, int n_split,
                              int sigmoid_scale, int two_source, int io, glowtts_stream_t wgrad_stream, glowtts_stream_t stream);

/* ---- relative-position multi-head self-attention (csrc/attention.hip) ------------------------------------------
 * replaces MultiHeadAttention.attention and its pad/reshape helpers (attentions.py:214-333).  q, k, v, out: (B, H*dk, T)
 * (head h = channels [h*dk, (h+1)*dk)); emb_k / emb_v: (1 or H, 2*window+1, dk) or NULL (no relative terms);
 * mask (B, T): pair (i, j) is kept iff mask[i]*mask[j] != 0 (the reference's attn_mask) and, if block_len >= 0,
 * |i - j| <= block_len; other scores are set to -1e4.  drop: (B, H, T, T) keep bytes applied to softmax(P) with scale
 * drop_scale, or NULL.  p_attn (B, H, T, T) receives softmax(P) BEFORE dropout (saved for the backward).
 * fwd : scores = (q_i.k_j + q_i.emb_k[j-i+w]) / sqrt(dk) ; out_i = sum_j Pd_ij v_j + sum_r Pd[i][i+r-w] emb_v[r]
 * bwd : dq, dk, dv (B, H*dk, T) written; demb_k / demb_v accumulated; ds (B, H, T, T) is scratch (scaled score grads).
 * Limits: dk % 16 == 0, dk <= 128, window <= 7; any T: MFMA kernels with the 64-query score strip on the chip up to 512 tokens
 * (in LDS to 256, in registers to 512), plain tiled kernels through p_attn / ds beyond (csrc/attention_long.hip; the bf16_mma
 * switch of the _ex forms has no effect there).
 * The `_ex` forms take bf16_mma: 0 = exactly the functions above (v_mfma_f32_16x16x4_f32); 1 = the contractions
 * (q k^T, q emb_k^T, P v, P_w emb_v, and dP, dq, dk, dv in the backward) on v_mfma_f32_16x16x16_bf16 with fp32
 * accumulation — operands are rounded to bf16 (nearest even) in registers as they leave LDS; q, k, v, p_attn, the
 * softmax, the relative-embedding gradients and every tensor in HBM stay fp32 (BASELINE configs[2], with bf16 tensors). */
int glowtts_rel_attn_fwd(const float *q, const float *k, const float *v, const float *emb_k, const float *emb_v,
                         const float *mask, const unsigned char *drop, float drop_scale, float *p_attn, float *out,
                         int B, int H, int T, int dk, int window, int heads_share, int block_len,
                         glowtts_stream_t stream);
int glowtts_rel_attn_bwd(const float *dout, const float *q, const float *k, const float *v, const float *emb_k,
                         const float *emb_v, const float *mask, const unsigned char *drop, float drop_scale,
                         const float *p_attn, float *ds, float *dq, float *dk_out, float *dv, float *demb_k,
                         float *demb_v, int B, int H, int T, int dk, int window, int heads_share, int block_len,
                         glowtts_stream_t stream);
int glowtts_rel_attn_fwd_ex(const float *q, const float *k, const float *v, const float *emb_k, const float *emb_v,
                            const float *mask, const unsigned char *drop, float drop_scale, float *p_attn, float *out,
                            int B, int H, int T, int dk, int window, int heads_share, int block_len, int bf16_mma,
                            glowtts_stream_t stream);
int glowtts_rel_attn_bwd_ex(const float *dout, const float *q, const float *k, const float *v, const float *emb_k,
                            const float *emb_v, const float *mask, const unsigned char *drop, float drop_scale,
                            const float *p_attn, float *ds, float *dq, float *dk_out, float *dv, float *demb_k,
                            float *demb_v, int B, int H, int T, int dk, int window, int heads_share, int block_len,
                            int bf16_mma, glowtts_stream_t stream);

/* ---- channel LayerNorm with fused residual (csrc/norm.hip) -----------------------------------------------------
 * replaces LayerNorm.forward (layers.py:19-28) and the `x + y` before it (attentions.py:68,72):
 * fwd : v = x (+ res) ; y = gamma * (v - mean_c v) * rsqrt(var_c v + eps) + beta ; stats (B, 2, T) = mean, rstd (may be NULL)
 * bwd : dx (= d res) written ; dgamma, dbeta accumulated */
int glowtts_chan_layernorm_fwd(const float *x, const float *res, const float *gamma, const float *beta, float *y,
                               float *stats, int B, int C, int T, float eps, glowtts_stream_t stream);
int glowtts_chan_layernorm_bwd(const float *x, const float *res, const float *gamma, const float *stats, const float *dy,
                               float *dx, float *dgamma, float *dbeta, int B, int C, int T, glowtts_stream_t stream);

/* ---- squeeze / unsqueeze (utils.py:135-160) -----------------------------------------------------------------
 * squeeze  : x (B,C,T) -> xs (B, n*C, T/n): xs[b, s*C+c, t'] = x[b,c,n*t'+s] * mask[b, n*t'+n-1] ; ms[b,t'] = mask[b, n*t'+n-1]
 * unsqueeze: xs (B, n*C, T') -> x (B,C,n*T'): x[b,c,n*t'+s] = xs[b,s*C+c,t'] * ms[b,t'] ; mask_out[b,n*t'+s] = ms[b,t']
 * (each is the other's adjoint up to the mask, so they also serve as each other's backward) */
int glowtts_squeeze(const float *x, const float *mask, float *xs, float *ms, int B, int C, int T, int n,
                    glowtts_stream_t stream);
int glowtts_unsqueeze(const float *xs, const float *ms, float *x, float *mask_out, int B, int C, int Tsq, int n,
                      glowtts_stream_t stream);

/* ---- mle_loss (utils.py:14-23) ------------------------------------------------------------------------------
 * fwd : acc[0] += sum(logs) + 0.5*sum(exp(-2 logs) (z-m)^2) ; acc[1] += sum over (b,t) of mask  (accumulated;
 *       the scalar loss = (acc[0] - sum(logdet)) / (C*acc[1]) + 0.5*log(2*pi) is finished by the caller)
 * bwd : with s = dloss / (C * sum mask) read from scale[0] on device:
 *       dz = s*exp(-2logs)(z-m) ; dm = -dz ; dlogs = s*(1 - exp(-2logs)(z-m)^2)   (mask is NOT applied, as in the reference) */
int glowtts_mle_fwd(const float *z, const float *m, const float *logs, const float *mask, float *acc, int B,
                    int C, int T, glowtts_stream_t stream);
int glowtts_mle_bwd(const float *z, const float *m, const float *logs, const float *scale, float *dz, float *dm,
                    float *dlogs, int64_t n, glowtts_stream_t stream);
/* The same losses with their scalar tails on the device (one autograd node, no one-element torch launches):
 *   mle_loss_fwd : acc (2, zero-filled) as glowtts_mle_fwd; out[0] = (acc[0] - sum logdet) / (acc[1] C) + 0.5 log(2 pi),
 *                  out[1] = acc[1] C (utils.py:14-23);  mle_loss_bwd : dz, dm, dlogs as glowtts_mle_bwd with
 *                  scale = dloss[0] / denom[0], and dlogdet[b] = -scale
 *   duration_loss_fwd : out[0] = sum (logw - logw_)^2 / sum lengths, out[1] = sum lengths (utils.py:26-28; lengths int64);
 *   duration_loss_bwd : dlogw = 2 (logw - logw_) dloss[0] / denom[0]
 *   span_logw : logw_[b][x] = log(1e-8 + first[b][x+1] - first[b][x]) for x < t_x[b], else 0 — the reference's
 *               log(1e-8 + sum(attn, -1)) * x_mask (models.py:392) from the span table of glowtts_mas_path_spans */
int glowtts_mle_loss_fwd(const float *z, const float *m, const float *logs, const float *mask, const float *logdet, float *acc,
                         float *out, int B, int C, int T, glowtts_stream_t stream);
int glowtts_mle_loss_bwd(const float *z, const float *m, const float *logs, const float *dloss, const float *denom, float *dz,
                         float *dm, float *dlogs, float *dlogdet, int B, int64_t n, glowtts_stream_t stream);
int glowtts_duration_loss_fwd(const float *logw, const float *logw_, const long long *lengths, float *out, int B, int64_t n,
                              glowtts_stream_t stream);
int glowtts_duration_loss_bwd(const float *logw, const float *logw_, const float *dloss, const float *denom, float *dlogw,
                              int64_t n, glowtts_stream_t stream);
int glowtts_span_logw(const int32_t *first, const int32_t *t_x, float *logw_, int B, int Tx, glowtts_stream_t stream);


/* ---- clip_grad_value_ (utils.py:118-132) and Adam + Noam (optimize.py:8-64) over FLAT buffers ---------------
 * clip : sumsq[0] += sum g^2 (pre-clamp, as the reference's norm) ; g = clamp(g, -clip, clip)
 * adam : state[0] = Adam step t (float, >= 1 at the call), state[1] = Noam step_num; the learning rate
 *        lr * dim_model^-0.5 * min(s^-0.5, s * warmup^-1.5) (or `lr` if warmup <= 0) is computed ON DEVICE from
 *        state so a captured graph replays with the right rate; torch.optim.Adam arithmetic (no amsgrad/decay).
 * adam_advance: state[0] += 1; state[1] += 1; state[2] = learning rate of the NEXT update (optimize.py:43-48) */
int glowtts_clip_grad_value(float *g, int64_t n, float clip, float *sumsq, glowtts_stream_t stream);
int glowtts_adam_noam(float *p, const float *g, float *m, float *v, int64_t n, const float *state, float lr,
                      float beta1, float beta2, float eps, float dim_model, float warmup,
                      glowtts_stream_t stream);
int glowtts_adam_advance(float *state, float lr, float dim_model, float warmup, glowtts_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* GLOWTTS_HIP_H */
