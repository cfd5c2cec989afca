// wn_fused.hip — the forward of a whole WN stack (reference layers.py:138-162) as ONE kernel: a workgroup owns all 192 hidden
// channels of a 52-frame tile of one utterance and runs the stack's layers back to back — k-tap in-conv, gate, 1x1 res/skip
// conv, residual update — with x, `acts` and the skip sum never leaving the compute unit between layers.
//
// Why: per layer the stack was a gated-conv launch + a res/skip launch; `acts` and x made an
// HBM / Infinity-Cache round trip per layer and each of the 8 launches of a block paid its own prologue, epilogue and dispatch
// skew (DESIGN.md 4a', lesson 21).  Here a block's forward is one launch of 8 x B workgroups — one per CU at config 2.
//
// Arithmetic: "bf16x6" only (convgemm_split.hip): every fp32 operand as three bf16 planes, six products per fp32 product on
// v_mfma_f32_16x16x32_bf16, fp32 accumulation.  Weights are the per-step planes the stack's other kernels use.
//
// Geometry (H = 192, 5 taps, dilation 1, <= 4 layers):
//   compute window : 64 frames (4 column tiles of 16) = the 52 owned frames + 6 on each side; a layer consumes 2 frames of halo
//                    per side, so after 4 layers exactly the owned frames are still right — the margins are recomputed by the
//                    neighbouring workgroups (23 % more MFMA work than the per-layer kernels for 7 fewer launches per block).
//   LDS (148 KB)   : R1 = bf16 plane image [3 planes][6 group pairs][68 frames][40] (pitch 80 B, the conflict-free pitch of
//                    convgemm_split) — holds the x planes while the in-conv runs, the `acts` planes while the res/skip conv runs;
//                    X32 = x in fp32 [64 frames][196] (the residual update's operand; R1 is overwritten by `acts`);
//                    row masks and the two bias vectors.
//   waves          : 12 (768 threads, three per SIMD).  Wave w owns hidden channels 16w .. 16w+15: in the in-conv the tanh row tile and
//                    the sigmoid row tile of those channels — the two halves of a gate are the same register of the same lane, no
//                    exchange; in the res/skip conv the residual row tile and the skip row tile.  The skip accumulators PERSIST
//                    across the layers — the skip sum is never stored until the end.  (First form: 8 waves x 24 channels as mixed
//                    tanh / sigmoid tiles with a lane exchange — its gate phase took 17.7 us per layer, this one's N.)
//   weights        : never touch LDS: each wave's A tiles are its own, loaded from L2 with range-checked buffer loads into a
//                    two-slot register ring one 32-deep step ahead (convgemm_split's k relabelling).
//
// Outputs are those of the per-layer path, for the unchanged backward: xs[l] = x_{l+1}, acts[l], ts[l] = (tanh, sigmoid),
// skip = (sum of skip rows + biases) * mask.  Only owned frames are written.
#include "convgemm_common.hpp"
#include "split_planes.hpp"

namespace glowtts {

struct WnFusedParams {
    const float *x;                 // (B, H, T) input of layer 0
    const float *mask;              // (B, T)
    const unsigned char *drop;      // (L, B, 2H, T) keep bytes or null
    float drop_scale;
    float *xs, *acts, *ts, *skip;   // (L-1, B, H, T), (L, B, H, T), (L, B, 2H, T), (B, H, T)
    const unsigned short *win[4];   // plane 0 of layer l's packed in-conv weights [tap][12][2H][16] (bf16)
    const unsigned short *wrs[4];   // plane 0 of layer l's packed res/skip weights [1][12][M][16]
    const float *bin[4], *brs[4];   // biases (2H) / (2H, H in the last layer)
    long plane_stride;              // elements between planes
    int B, T, n_layers, ntiles;
};

namespace wnf {
constexpr int H = 192, G = 12, GP = 6, XR = 68, RP = 40, W = 64, NCT = 4, NT = 52, XP = 196, TAPS = 5;
constexpr int PLANE16 = GP * XR * RP;                      // bf16 elements of one plane image
constexpr int PLANE_B = PLANE16 * 2;                       // bytes
constexpr size_t LDS_BYTES = 3 * (size_t)PLANE_B + ((size_t)W * XP + 72 + 2 * 2 * H) * sizeof(float);

__device__ __forceinline__ void lds_store8(void *p, unsigned lo, unsigned hi) {   // see convgemm_split.hip: never ds_write_b64
    asm volatile("ds_write2_b32 %0, %1, %2 offset1:1"
                 :: "v"((unsigned)(size_t)(__attribute__((address_space(3))) void *)p), "v"(lo), "v"(hi) : "memory");
}
__device__ __forceinline__ void lds_stores_done() { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); }

// position (in bf16 elements) of channel c inside the 32-channel row of its group pair: lane slot lk of an MFMA consumes channels
// 4 lk .. 4 lk + 3 of both groups of the pair (convgemm_split.hip)
__device__ __forceinline__ int row_pos(int c) { const int kk = c & 15; return (kk >> 2) * 8 + ((c >> 4) & 1) * 4 + (kk & 3); }

#ifdef GLOWTTS_TRACE
// 32 words per workgroup: 0 start, 1 prologue done; layer l at 2 + 6 l: in-conv start / end, gate end, res-skip start / end, update end;
// 26 end; 27 / 28 shader cycle counter around layer 0's in-conv; 29 XCC id
#define WNF_TRACE(i) do { if (threadIdx.x == 0) g_trace[(blockIdx.x & 4095) * 32 + (i)] = wall_clock64(); } while (0)
#define WNF_CYCLES(i) do { if (threadIdx.x == 0) g_trace[(blockIdx.x & 4095) * 32 + (i)] = __builtin_readcyclecounter(); } while (0)
#else
#define WNF_TRACE(i) do { } while (0)
#define WNF_CYCLES(i) do { } while (0)
#endif
}  // namespace wnf

__global__ __launch_bounds__(768, 3) void wn_fused_kernel(WnFusedParams p) {
    using namespace wnf;
    extern __shared__ __align__(16) float smem[];
    unsigned short *Pl = reinterpret_cast<unsigned short *>(smem);         // [3][GP][XR][RP]
    float *X32 = smem + 3 * PLANE16 / 2;                                   // [W][XP]
    float *Ms = X32 + W * XP;                                              // [72]: mask of LDS row r (0 outside the utterance)
    float *Bi = Ms + 72;                                                   // [2H] in-conv bias of the layer
    float *Br = Bi + 2 * H;                                                // [2H] res/skip bias of the layer

    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int lrow_ = lane & 15, lk_ = lane >> 4;
    const int lrow = lrow_, lk = lk_;
    const int b = blockIdx.x / p.ntiles;
    const int t0 = (blockIdx.x - b * p.ntiles) * NT;
    const int fs = t0 - 8;                                                 // frame of LDS row 0 (a multiple of 4)
    const int T = p.T;
    const long HT = (long)H * T;

    WNF_TRACE(0);
#ifdef GLOWTTS_TRACE
    if (threadIdx.x == 0) g_trace[(blockIdx.x & 4095) * 32 + 29] = __builtin_amdgcn_s_getreg(20 | (0 << 6) | (3 << 11));
#endif
    // ------------------------------------------------------------------------------------------------ prologue: x_0 -> R1, X32
    // pieces of 4 channels x 4 frames (four 16-byte loads; a frame of the piece is one 8-byte LDS store per plane): 6 group pairs
    // x 20 frame quads (17 used) x 8 channel quads = 960 slots; 16 lanes = 8 channel quads x 2 frame quads 4 frames apart
    {
        const char *xb8 = reinterpret_cast<const char *>(p.x + (long)b * HT);
        const __amdgpu_buffer_rsrc_t xrs = __builtin_amdgcn_make_buffer_rsrc(const_cast<char *>(xb8), 0, (int)(HT * 4), 0x00020000);
        const int c4l = tid & 7, c4 = c4l & 3, gs = c4l >> 2, fql = (tid >> 3) & 3;
        f32x4 v[2][4];
        int hi_[2];
#pragma unroll
        for (int r = 0; r < 2; ++r) {
            const int hi = (tid >> 5) + 24 * r;                          // 24 half-wave slots per round: 30 (group pair, quad group)s
            hi_[r] = hi;
            const int gp = hi / 5, fq = (hi - gp * 5) * 4 + fql;
            const int t = fs + fq * 4;
            const bool ok = hi < 30 && fq < XR / 4 && t >= 0 && t < T;
            const int ch = gp * 32 + gs * 16 + c4 * 4;
#pragma unroll
            for (int i = 0; i < 4; ++i)
                v[r][i] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(
                    xrs, ok ? ((ch + i) * T + t) * 4 : 0x7fffffff, 0, 0));
        }
        if (tid < 72) {
            const int t = fs + tid;
            Ms[tid] = (tid < XR && t >= 0 && t < T) ? p.mask[(long)b * T + t] : 0.f;
        }
#pragma unroll
        for (int r = 0; r < 2; ++r) {
            const int hi = hi_[r];
            const int gp = hi / 5, fq = (hi - gp * 5) * 4 + fql;
            if (hi < 30 && fq < XR / 4) {
                const int ch = gp * 32 + gs * 16 + c4 * 4;
#pragma unroll
                for (int f = 0; f < 4; ++f) {
                    const int row = fq * 4 + f;
                    unsigned oa[3], ob[3];
                    split_planes2<3>(v[r][0][f], v[r][1][f], oa);
                    split_planes2<3>(v[r][2][f], v[r][3][f], ob);
#pragma unroll
                    for (int pl = 0; pl < 3; ++pl) lds_store8(Pl + pl * PLANE16 + (gp * XR + row) * RP + c4 * 8 + gs * 4, oa[pl], ob[pl]);
                    if (row >= 2 && row < 2 + W)
                        *reinterpret_cast<f32x4 *>(X32 + (row - 2) * XP + ch) = f32x4{v[r][0][f], v[r][1][f], v[r][2][f], v[r][3][f]};
                }
            }
        }
        lds_stores_done();
    }

    // ---- per-lane constants -------------------------------------------------------------------------------------------
    // wave w owns hidden channels 16 w .. 16 w + 15: in-conv row tiles T (tanh rows 16 w + lrow) and S (sigmoid rows H + 16 w + lrow)
    // — the two halves of a gate are the same register of the same lane; res/skip row tiles R (residual rows) and K (skip rows)
    const int cw = 16 * wave;
    const int wvo_t = ((cw + lrow) * 16 + lk * 4) * 2, wvo_s = wvo_t + H * 32;
    constexpr int WTAP_IN = G * 2 * H * 32, WGRP_IN = 2 * H * 32, WBYTES_IN = TAPS * WTAP_IN;      // bytes of one plane
    const char *xdb = reinterpret_cast<const char *>(Pl) + lrow * (RP * 2) + lk * 16;
    const int T4 = T * 4, HT4 = (int)(HT * 4);

    f32x4 racc[2][NCT];                                  // res/skip accumulators: [0] residual rows, [1] skip rows — PERSISTS
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int c = 0; c < NCT; ++c) racc[j][c] = f32x4{0.f, 0.f, 0.f, 0.f};

    i32x4 a[2][2][3];                                    // weight ring: [slot][row tile][plane]
    i32x4 bv[2][3];                                      // B operands: [slot][plane]


    for (int l = 0; l < p.n_layers; ++l) {
        const bool last = l == p.n_layers - 1;
        // static selects: a run-time index into by-value kernel arguments would put the tables in scratch (lesson 20)
        const unsigned short *win = p.win[0], *wrs = p.wrs[0];
        const float *bin = p.bin[0], *brs = p.brs[0];
#pragma unroll
        for (int q = 1; q < 4; ++q)
            if (l == q) { win = p.win[q]; wrs = p.wrs[q]; bin = p.bin[q]; brs = p.brs[q]; }
        const int Mrs = last ? H : 2 * H;
        __amdgpu_buffer_rsrc_t wr_in[3], wr_rs[3];
#pragma unroll
        for (int pl = 0; pl < 3; ++pl) {
            wr_in[pl] = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned short *>(win + pl * p.plane_stride), 0, WBYTES_IN, 0x00020000);
            wr_rs[pl] = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned short *>(wrs + pl * p.plane_stride), 0, G * Mrs * 32, 0x00020000);
        }
        auto wload_in = [&](int s, int slot) {           // step s = 5 gp + tap of the in-conv (beyond 29: the last step again, unused)
            s = s < GP * TAPS ? s : GP * TAPS - 1;
            const int gp = s / TAPS, tap = s - gp * TAPS;
            const int so0 = tap * WTAP_IN + 2 * gp * WGRP_IN;
            const int so1 = so0 + WGRP_IN;
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int pl = 0; pl < 3; ++pl) {
                    const i32x2 lo = __builtin_bit_cast(i32x2, __builtin_amdgcn_raw_buffer_load_b64(wr_in[pl], j ? wvo_s : wvo_t, so0, 0));
                    const i32x2 hi = __builtin_bit_cast(i32x2, __builtin_amdgcn_raw_buffer_load_b64(wr_in[pl], j ? wvo_s : wvo_t, so1, 0));
                    a[slot][j][pl] = i32x4{lo[0], lo[1], hi[0], hi[1]};
                }
        };
        auto bfetch = [&](int gp, int row0, int slot) {  // the B operand of column rows row0 + lrow, group pair gp
#pragma unroll
            for (int pl = 0; pl < 3; ++pl)
                bv[slot][pl] = *reinterpret_cast<const i32x4 *>(xdb + pl * PLANE_B + (gp * XR + row0) * (RP * 2));
        };

        // ------------------------------------------------------------------------------------------ in-conv: 30 steps of 48 MFMAs
        f32x4 acc[2][NCT];
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int c = 0; c < NCT; ++c) acc[j][c] = f32x4{0.f, 0.f, 0.f, 0.f};
        wload_in(0, 0);
        wload_in(1, 1);
        if (l == 0) WNF_TRACE(1);
        __syncthreads();                                 // B4 of the previous layer / the prologue's stores
        WNF_TRACE(2 + 6 * l);
        if (l == 0) WNF_CYCLES(27);
        // this layer's biases (read behind B1 / B3; the previous layer's last readers are behind the barrier above)
        if (tid < 2 * H) Bi[tid] = bin[tid];
        else if (tid - 2 * H < Mrs) Br[tid - 2 * H] = brs[tid - 2 * H];
        bfetch(0, 0, 0);
        for (int it = 0; it < GP / 2; ++it) {
#pragma unroll
            for (int i = 0; i < 2 * TAPS; ++i) {
#pragma unroll
                for (int cc = 0; cc < NCT; ++cc) {
                    const int q = i * NCT + cc;
                    {                                    // next tile's LDS reads ahead of this tile's MFMAs
                        const int qn = q + 1, in_ = qn / NCT, cn = qn % NCT;
                        const int gpn = 2 * it + in_ / TAPS, tapn = in_ % TAPS;      // (in_ == 10: the next iteration's first tile)
                        bfetch(gpn, cn * 16 + tapn, qn & 1);
                    }
                    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                    for (int j = 0; j < 2; ++j)
#pragma unroll
                        for (int k = 0; k < 6; ++k)
                            acc[j][cc] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(
                                __builtin_bit_cast(bf16x8, a[i & 1][j][product_a(3, k)]),
                                __builtin_bit_cast(bf16x8, bv[q & 1][product_b(3, k)]), acc[j][cc], 0, 0, 0);
                    __builtin_amdgcn_sched_barrier(0);
                }
                wload_in(it * 2 * TAPS + i + 2, i & 1);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        WNF_TRACE(3 + 6 * l);
        if (l == 0) WNF_CYCLES(28);
        __syncthreads();                                 // B1: every wave is through with the x planes
        if (l == 1) WNF_TRACE(30);                       // (layer 1's release of B1: splits "B1 wait" from the gate phase)
        // (Measured: letting a wave that leaves the MFMA loop early run its gate arithmetic beside the other waves' MFMAs — planes
        //  held in registers until this barrier — is SLOWER: 385 us per stack against 300; the element-wise instructions of the
        //  SIMD's oldest wave take issue slots from the younger waves' MFMA streams, with or without s_setprio on the loops.)

        const int wvo_r = last ? G * Mrs * 32 : wvo_t;                      // residual rows 16 w + lrow (none in the last layer)
        const int wvo_k = last ? wvo_t : wvo_t + H * 32;                    // skip rows (H +) 16 w + lrow
        const int wgrp_rs = Mrs * 32;
        auto wload_rs = [&](int s, int slot) {
            s = s < GP ? s : GP - 1;                     // (beyond the last step: that step again, unused)
            const int so0 = 2 * s * wgrp_rs;
            const int so1 = so0 + wgrp_rs;
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int pl = 0; pl < 3; ++pl) {
                    const i32x2 lo = __builtin_bit_cast(i32x2, __builtin_amdgcn_raw_buffer_load_b64(wr_rs[pl], j ? wvo_k : wvo_r, so0, 0));
                    const i32x2 hi = __builtin_bit_cast(i32x2, __builtin_amdgcn_raw_buffer_load_b64(wr_rs[pl], j ? wvo_k : wvo_r, so1, 0));
                    a[slot][j][pl] = i32x4{lo[0], lo[1], hi[0], hi[1]};
                }
        };

        // ------------------------------------------------------------------------------------------ gate -> ts, acts, acts planes
        // lane: channels ch .. ch + 3 (registers), frame r of every column tile.  One byte offset per column tile (out of range for
        // frames this workgroup does not own); the register's row and the sigmoid half ride in the scalar offset.
        {
            int lrow = lrow_, lk = lk_;                  // (laundered: this phase's address arithmetic must not be hoisted out of
            asm volatile("" : "+v"(lrow), "+v"(lk));     //  the layer loop and spilled in front of the first MFMA loop)
            const long slab2 = ((long)l * p.B + b) * 2 * HT;              // (l, b) slab of ts / drop, in elements
            const __amdgpu_buffer_rsrc_t ts_rs = __builtin_amdgcn_make_buffer_rsrc(p.ts + slab2, 0, 2 * HT4, 0x00020000);
            const __amdgpu_buffer_rsrc_t ac_rs = __builtin_amdgcn_make_buffer_rsrc(p.acts + slab2 / 2, 0, HT4, 0x00020000);
            const __amdgpu_buffer_rsrc_t dr_rs = __builtin_amdgcn_make_buffer_rsrc(
                const_cast<unsigned char *>(p.drop ? p.drop + slab2 : reinterpret_cast<const unsigned char *>(p.x)), 0,
                p.drop ? (int)(2 * HT) : 0, 0x00020000);
            const int ch = cw + 4 * lk;
            const int tl = fs + 2 + lrow;                                 // frame of column tile 0
            const int ppos = ((ch >> 5) * XR + 2 + lrow) * RP + row_pos(ch);
            unsigned kt[NCT][4], ks[NCT][4];
            if (p.drop) {                                // keep bytes of EVERY frame of the window inside the utterance (the margins
#pragma unroll                                           // feed the next layers), all 32 loads in flight before the first use
                for (int cc = 0; cc < NCT; ++cc) {
                    const int t = tl + 16 * cc;
                    const int ok = (t >= 0 && t < T) ? ch * T + t : 0x7fffffff;
#pragma unroll
                    for (int g = 0; g < 4; ++g) {
                        kt[cc][g] = __builtin_amdgcn_raw_buffer_load_b8(dr_rs, ok, g * T, 0);
                        ks[cc][g] = __builtin_amdgcn_raw_buffer_load_b8(dr_rs, ok, g * T + (int)HT, 0);
                    }
                }
            }
            const f32x4 bt = *reinterpret_cast<const f32x4 *>(Bi + ch), bs = *reinterpret_cast<const f32x4 *>(Bi + H + ch);
#pragma unroll
            for (int cc = 0; cc < NCT; ++cc) {
                const int t = tl + 16 * cc;
                const int vo = (t >= t0 && t < t0 + NT && t < T) ? (ch * T + t) * 4 : 0x7fffffff;
                float av[4];
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    float vt = acc[0][cc][g] + bt[g], vs = acc[1][cc][g] + bs[g];
                    if (p.drop) {
                        vt = (kt[cc][g] & 0xffu) ? vt * p.drop_scale : 0.f;
                        vs = (ks[cc][g] & 0xffu) ? vs * p.drop_scale : 0.f;
                    }
                    const float th = fast_tanh(vt), sg = fast_sigmoid(vs);
                    av[g] = th * sg;
                    __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(th), ts_rs, vo, g * T4, 0);
                    __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(sg), ts_rs, vo, g * T4 + HT4, 0);
                    __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(av[g]), ac_rs, vo, g * T4, 0);
                }
                unsigned oa[3], ob[3];
                split_planes2<3>(av[0], av[1], oa);
                split_planes2<3>(av[2], av[3], ob);
#pragma unroll
                for (int pl = 0; pl < 3; ++pl) lds_store8(Pl + pl * PLANE16 + ppos + 16 * cc * RP, oa[pl], ob[pl]);
            }
            wload_rs(0, 0);                              // in flight across the barrier (issued earlier they cost the gate phase
            wload_rs(1, 1);                              // 48 registers it does not have at three waves per SIMD)
            lds_stores_done();
        }

        // ------------------------------------------------------------------------------------------ res/skip conv: 6 steps
        WNF_TRACE(4 + 6 * l);
        __syncthreads();                                 // B2: the acts planes are complete
        WNF_TRACE(5 + 6 * l);
        bfetch(0, 2, 0);
#pragma unroll
        for (int s = 0; s < GP; ++s) {
#pragma unroll
            for (int cc = 0; cc < NCT; ++cc) {
                const int q = s * NCT + cc;
                if (q + 1 < GP * NCT) bfetch((q + 1) / NCT, 2 + ((q + 1) % NCT) * 16, (q + 1) & 1);
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int j = 0; j < 2; ++j)          // (the last layer has no residual rows: their weights read as zeros — a uniform
#pragma unroll                                        //  branch around MFMAs would keep the LDS reads from moving ahead, lesson 19)
                    for (int k = 0; k < 6; ++k)
                        racc[j][cc] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(
                            __builtin_bit_cast(bf16x8, a[s & 1][j][product_a(3, k)]),
                            __builtin_bit_cast(bf16x8, bv[q & 1][product_b(3, k)]), racc[j][cc], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
            }
            wload_rs(s + 2, s & 1);
            __builtin_amdgcn_sched_barrier(0);
        }
        WNF_TRACE(6 + 6 * l);
        __syncthreads();                                 // B3: every wave is through with the acts planes

        // ------------------------------------------------------------------------------------------ residual update, skip biases
        {
            int lrow = lrow_, lk = lk_;
            asm volatile("" : "+v"(lrow), "+v"(lk));
            const __amdgpu_buffer_rsrc_t xs_rs = __builtin_amdgcn_make_buffer_rsrc(
                last ? p.skip + (long)b * HT : p.xs + ((long)l * p.B + b) * HT, 0, HT4, 0x00020000);
            const int ch = cw + 4 * lk;
            const int tl = fs + 2 + lrow;
            const f32x4 bk = *reinterpret_cast<const f32x4 *>(Br + (last ? 0 : H) + ch);
            if (!last) {
                const f32x4 bb = *reinterpret_cast<const f32x4 *>(Br + ch);
                const int ppos = ((ch >> 5) * XR + 2 + lrow) * RP + row_pos(ch);
#pragma unroll
                for (int cc = 0; cc < NCT; ++cc) {
                    const int t = tl + 16 * cc;
                    const int vo = (t >= t0 && t < t0 + NT && t < T) ? (ch * T + t) * 4 : 0x7fffffff;
                    const float m = Ms[2 + 16 * cc + lrow];
                    f32x4 xo = *reinterpret_cast<const f32x4 *>(X32 + (16 * cc + lrow) * XP + ch);
#pragma unroll
                    for (int g = 0; g < 4; ++g) xo[g] = (xo[g] + racc[0][cc][g] + bb[g]) * m;
                    *reinterpret_cast<f32x4 *>(X32 + (16 * cc + lrow) * XP + ch) = xo;
                    unsigned oa[3], ob[3];
                    split_planes2<3>(xo[0], xo[1], oa);
                    split_planes2<3>(xo[2], xo[3], ob);
#pragma unroll
                    for (int pl = 0; pl < 3; ++pl) lds_store8(Pl + pl * PLANE16 + ppos + 16 * cc * RP, oa[pl], ob[pl]);
#pragma unroll
                    for (int g = 0; g < 4; ++g) {
                        const float xv = xo[g];          // (a scalar copy: __builtin_bit_cast applied to the vector ELEMENT stored element 0 four times)
                        __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(xv), xs_rs, vo, g * T4, 0);
                    }
                    racc[0][cc] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
                    for (int g = 0; g < 4; ++g) racc[1][cc][g] += bk[g];
                }
                lds_stores_done();
            } else {                                     // WN's final `output * x_mask` (layers.py:161-162)
#pragma unroll
                for (int cc = 0; cc < NCT; ++cc) {
                    const int t = tl + 16 * cc;
                    const int vo = (t >= t0 && t < t0 + NT && t < T) ? (ch * T + t) * 4 : 0x7fffffff;
                    const float m = Ms[2 + 16 * cc + lrow];
#pragma unroll
                    for (int g = 0; g < 4; ++g)
                        __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint((racc[1][cc][g] + bk[g]) * m), xs_rs, vo, g * T4, 0);
                }
            }
        }
        WNF_TRACE(7 + 6 * l);
        // (B4 is the barrier at the top of the next layer's in-conv)
    }
    WNF_TRACE(26);
}

// OFF unless GLOWTTS_WN_FUSED=1 (read once) or glowtts_wn_fused(1) said otherwise: measured at config 2 the kernel is at parity
// with the per-layer launches (300 us per stack either way, 15.55 vs 15.56 ms per step — DESIGN.md 4i), and its 8 x B grid
// only fills the chip when 8 B is close to a multiple of 256
static std::atomic<int> g_wn_fused{-1};
static std::atomic<int> g_wn_fused_launches{0};
static bool wn_fused_enabled() {
    int v = g_wn_fused.load(std::memory_order_relaxed);
    if (v < 0) {
        v = knob(K_WN_FUSED) == 1;
        g_wn_fused.store(v, std::memory_order_relaxed);
    }
    return v != 0;
}

// convgemm_split.hip
bool conv_find_planes(const float *wp, int ns, const unsigned short **out, long *stride);
int conv_math_forward();

// -1: not applicable (shape, arithmetic, planes not bound, switched off) — the caller runs the per-layer sequence
int wn_fused_dispatch(const glowtts_wn_layer *layers, int n_layers, const float *x, const float *mask, const unsigned char *drop,
                      float drop_scale, float *xs, float *acts, float *ts, float *skip, int B, int H, int T, int taps, int dil_rate,
                      hipStream_t s) {
    if (!wn_fused_enabled() || H != wnf::H || taps != wnf::TAPS || dil_rate != 1 || n_layers < 1 || n_layers > 4 || (T & 3) || B <= 0 || T <= 0) return -1;
    if (conv_math_forward() != 3) return -1;
    if ((long)2 * H * T * 4 > 0x7fffffffL) return -1;
    if (!aligned16(x) || !aligned16(mask)) return -1;
    WnFusedParams p{};
    p.x = x; p.mask = mask; p.drop = drop; p.drop_scale = drop_scale;
    p.xs = xs; p.acts = acts; p.ts = ts; p.skip = skip;
    p.B = B; p.T = T; p.n_layers = n_layers; p.ntiles = (T + wnf::NT - 1) / wnf::NT;
    long stride = 0;
    for (int l = 0; l < 4; ++l) {
        const glowtts_wn_layer &L = layers[l < n_layers ? l : n_layers - 1];
        const unsigned short *pin = nullptr, *prs = nullptr;
        long s1 = 0, s2 = 0;
        if (!L.wf_in || !L.wf_rs || !L.b_in || !L.b_rs) return -1;
        if (!conv_find_planes(L.wf_in, 3, &pin, &s1) || !conv_find_planes(L.wf_rs, 3, &prs, &s2) || s1 != s2) return -1;
        if (l > 0 && s1 != stride) return -1;
        stride = s1;
        p.win[l] = pin; p.wrs[l] = prs; p.bin[l] = L.b_in; p.brs[l] = L.b_rs;
    }
    p.plane_stride = stride;
    static LdsLimit attr_max_e;
    if (int rc_ = attr_max_e.ensure(reinterpret_cast<const void *>(&wn_fused_kernel), wnf::LDS_BYTES, "glowtts_wn_fwd (fused)")) return rc_;
    g_wn_fused_launches.fetch_add(1, std::memory_order_relaxed);
    hipLaunchKernelGGL(wn_fused_kernel, dim3((unsigned)(p.ntiles * B)), dim3(768), wnf::LDS_BYTES, s, p);
    GLOWTTS_LAUNCH_CHECK("glowtts_wn_fwd (fused)");
}

}  // namespace glowtts

extern "C" int glowtts_wn_fused(int enable) {
    if (enable == -2) return glowtts::g_wn_fused_launches.load(std::memory_order_relaxed);
    const int before = glowtts::wn_fused_enabled() ? 1 : 0;
    if (enable >= 0) glowtts::g_wn_fused.store(enable != 0, std::memory_order_relaxed);
    return before;
}

#ifdef GLOWTTS_TRACE
extern "C" int glowtts_debug_trace_read_wnf(unsigned long long *host, int n_words, int clear) {
    hipError_t e = hipMemcpyFromSymbol(host, HIP_SYMBOL(glowtts::g_trace), (size_t)n_words * 8);
    if (e != hipSuccess) return (int)e;
    if (clear) {
        static unsigned long long zeros[8192 * 16];
        e = hipMemcpyToSymbol(HIP_SYMBOL(glowtts::g_trace), zeros, sizeof(zeros));
    }
    return (int)e;
}
#endif
