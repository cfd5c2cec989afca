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
//   waves          : 8 (512 threads, two per SIMD).  In-conv: wave w owns hidden channels 24w .. 24w+23 as three MIXED row tiles
//                    (rows 0-7 tanh rows of 8 channels, rows 8-15 the sigmoid rows of the same channels): the pair of a gate
//                    meets in lanes l / l + 32 of one register -> v_permlane32_swap, no LDS exchange.  Res/skip conv: wave w owns
//                    residual rows 24w .. 24w+23 and skip rows 24w .. 24w+23 (tile 0: residual, tile 1: 8 + 8, tile 2: skip);
//                    the skip accumulators PERSIST across the layers — the skip sum is never stored until the end.
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
#define WNF_TRACE(i) do { if (threadIdx.x == 0) g_trace[(blockIdx.x & 8191) * 16 + (i)] = wall_clock64(); } while (0)
#else
#define WNF_TRACE(i) do { } while (0)
#endif
}  // namespace wnf

__global__ __launch_bounds__(512, 2) void wn_fused_kernel(WnFusedParams p) {
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
    // ------------------------------------------------------------------------------------------------ prologue: x_0 -> R1, X32
    {
        const char *xb8 = reinterpret_cast<const char *>(p.x + (long)b * HT);
        const __amdgpu_buffer_rsrc_t xrs = __builtin_amdgcn_make_buffer_rsrc(const_cast<char *>(xb8), 0, (int)(HT * 4), 0x00020000);
        const int c4l = tid & 7, c4 = c4l & 3, gs = c4l >> 2, fql = (tid >> 3) & 3;
        f32x4 v[2][4];
        int hi_[2];
#pragma unroll
        for (int r = 0; r < 2; ++r) {
            const int hi = (tid >> 5) + 16 * r;
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
    // in-conv A rows: tile j, lane row lrow: lrow < 8 -> tanh row of channel 24 w + 8 j + lrow, else the sigmoid row of channel
    // 24 w + 8 j + lrow - 8
    int wvo_in[3];
#pragma unroll
    for (int j = 0; j < 3; ++j) {
        const int chn = 24 * wave + 8 * j + (lrow & 7);
        const int row = lrow < 8 ? chn : H + chn;
        wvo_in[j] = (row * 16 + lk * 4) * 2;
    }
    constexpr int WTAP_IN = G * 2 * H * 32, WGRP_IN = 2 * H * 32, WBYTES_IN = TAPS * WTAP_IN;      // bytes of one plane
    const char *xdb = reinterpret_cast<const char *>(Pl) + lrow * (RP * 2) + lk * 16;

    f32x4 racc[3][NCT];                                  // res/skip accumulators: tile 1 lanes >= 32 and tile 2 persist (skip)
#pragma unroll
    for (int j = 0; j < 3; ++j)
#pragma unroll
        for (int c = 0; c < NCT; ++c) racc[j][c] = f32x4{0.f, 0.f, 0.f, 0.f};

    i32x4 a[2][3][3];                                    // weight ring: [slot][row tile][plane]
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
            for (int j = 0; j < 3; ++j)
#pragma unroll
                for (int pl = 0; pl < 3; ++pl) {
                    const i32x2 lo = __builtin_bit_cast(i32x2, __builtin_amdgcn_raw_buffer_load_b64(wr_in[pl], wvo_in[j], so0, 0));
                    const i32x2 hi = __builtin_bit_cast(i32x2, __builtin_amdgcn_raw_buffer_load_b64(wr_in[pl], wvo_in[j], so1, 0));
                    a[slot][j][pl] = i32x4{lo[0], lo[1], hi[0], hi[1]};
                }
        };
        auto bfetch = [&](int gp, int row0, int slot) {  // the B operand of column rows row0 + lrow, group pair gp
#pragma unroll
            for (int pl = 0; pl < 3; ++pl)
                bv[slot][pl] = *reinterpret_cast<const i32x4 *>(xdb + pl * PLANE_B + (gp * XR + row0) * (RP * 2));
        };

        // ------------------------------------------------------------------------------------------ in-conv: 30 steps of 72 MFMAs
        f32x4 acc[3][NCT];
#pragma unroll
        for (int j = 0; j < 3; ++j)
#pragma unroll
            for (int c = 0; c < NCT; ++c) acc[j][c] = f32x4{0.f, 0.f, 0.f, 0.f};
        wload_in(0, 0);
        wload_in(1, 1);
        __syncthreads();                                 // B4 of the previous layer / the prologue's stores
        WNF_TRACE(1);
        // this layer's biases (read behind B1 / B3; the previous layer's last readers are behind the barrier above)
        if (tid < 2 * H) Bi[tid] = bin[tid];
        else if (tid - 2 * H < Mrs) Br[tid - 2 * H] = brs[tid - 2 * H];
        if (tid < Mrs - (512 - 2 * H)) Br[tid + 512 - 2 * H] = brs[tid + 512 - 2 * H];
        bfetch(0, 0, 0);
        for (int it = 0; it < GP / 2; ++it) {
#pragma unroll
            for (int i = 0; i < 2 * TAPS; ++i) {
                const int gp = 2 * it + i / TAPS, tap = i % TAPS;
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
                    for (int j = 0; j < 3; ++j)
#pragma unroll
                        for (int k = 0; k < 6; ++k)
                            acc[j][cc] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(
                                __builtin_bit_cast(bf16x8, a[i & 1][j][product_a(3, k)]),
                                __builtin_bit_cast(bf16x8, bv[q & 1][product_b(3, k)]), acc[j][cc], 0, 0, 0);
                    __builtin_amdgcn_sched_barrier(0);
                }
                (void)gp; (void)tap;
                wload_in(it * 2 * TAPS + i + 2, i & 1);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        WNF_TRACE(2);
        __syncthreads();                                 // B1: every wave is through with the x planes

        // ------------------------------------------------------------------------------------------ gate -> ts, acts, acts planes
        {
            const long slab2 = ((long)l * p.B + b) * 2 * HT;              // (l, b) slab of ts / drop, in elements
            const __amdgpu_buffer_rsrc_t ts_rs = __builtin_amdgcn_make_buffer_rsrc(p.ts + slab2, 0, (int)(2 * HT * 4), 0x00020000);
            const __amdgpu_buffer_rsrc_t ac_rs = __builtin_amdgcn_make_buffer_rsrc(p.acts + slab2 / 2, 0, (int)(HT * 4), 0x00020000);
            const __amdgpu_buffer_rsrc_t dr_rs = __builtin_amdgcn_make_buffer_rsrc(
                const_cast<unsigned char *>(p.drop ? p.drop + slab2 : reinterpret_cast<const unsigned char *>(p.x)), 0,
                p.drop ? (int)(2 * HT) : 0, 0x00020000);
            // (laundered lane ids: the address arithmetic of this phase must not be hoisted out of the layer loop — hoisted, its
            //  ~150 per-lane offsets were spilled to scratch in front of the first MFMA loop)
            int lrow = lrow_, lk = lk_;
            asm volatile("" : "+v"(lrow), "+v"(lk));
            const int chl = 4 * (lk & 1) + 2 * (lk >> 1);                 // this lane's channel pair inside a tile's 8 channels
            // keep bytes of EVERY frame of the compute window inside the utterance (the margins feed the next layers); a tile's 16
            // loads are issued one tile ahead of their use
            unsigned kt[3][NCT][2], ks[3][NCT][2];
            auto keep_load = [&](int j) {
#pragma unroll
                for (int cc = 0; cc < NCT; ++cc)
#pragma unroll
                    for (int e = 0; e < 2; ++e) {
                        const int t = fs + 2 + 16 * cc + lrow;
                        const int ot = (t >= 0 && t < T) ? (24 * wave + 8 * j + chl + e) * T + t : 0x7fffffff;
                        kt[j][cc][e] = __builtin_amdgcn_raw_buffer_load_b8(dr_rs, ot, 0, 0);
                        ks[j][cc][e] = __builtin_amdgcn_raw_buffer_load_b8(dr_rs, ot == 0x7fffffff ? ot : ot + (int)HT, 0, 0);
                    }
            };
            if (p.drop) keep_load(0);
#pragma unroll
            for (int j = 0; j < 3; ++j) {
                if (p.drop && j + 1 < 3) keep_load(j + 1);
                const int ch0 = 24 * wave + 8 * j + chl;
                const f32x2 bt = *reinterpret_cast<const f32x2 *>(Bi + ch0), bs = *reinterpret_cast<const f32x2 *>(Bi + H + ch0);
                const int ppos = ((ch0 >> 5) * XR) * RP + row_pos(ch0);
#pragma unroll
                for (int cc = 0; cc < NCT; ++cc) {
                    const int r = 2 + 16 * cc + lrow, t = fs + r;
                    const bool own = t >= t0 && t < t0 + NT && t < T;
                    // lanes l < 32 keep their tanh registers 0, 1 and receive the sigmoid registers 0, 1 of lane l + 32; lanes
                    // l >= 32 keep their sigmoid registers 2, 3 and receive the tanh registers 2, 3 of lane l - 32
                    float pt[2], ps[2];
#pragma unroll
                    for (int e = 0; e < 2; ++e) {
                        const float mine_lo = acc[j][cc][e], mine_hi = acc[j][cc][e + 2];
                        const float send = lane < 32 ? mine_hi : mine_lo;
                        const float got = __shfl_xor(send, 32, 64);
                        pt[e] = lane < 32 ? mine_lo : got;
                        ps[e] = lane < 32 ? got : mine_hi;
                    }
                    float th[2], sg[2], av[2];
#pragma unroll
                    for (int e = 0; e < 2; ++e) {
                        float vt = pt[e] + bt[e], vs = ps[e] + bs[e];
                        if (p.drop) {
                            vt = (kt[j][cc][e] & 0xffu) ? vt * p.drop_scale : 0.f;
                            vs = (ks[j][cc][e] & 0xffu) ? vs * p.drop_scale : 0.f;
                        }
                        th[e] = fast_tanh(vt);
                        sg[e] = fast_sigmoid(vs);
                        av[e] = th[e] * sg[e];
                        const int ot4 = own ? ((ch0 + e) * T + t) * 4 : 0x7fffffff;
                        __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(th[e]), ts_rs, ot4, 0, 0);
                        __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(sg[e]), ts_rs, own ? ot4 + (int)(HT * 4) : 0x7fffffff, 0, 0);
                        __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(av[e]), ac_rs, ot4, 0, 0);
                    }
                    unsigned o[3];
                    split_planes2<3>(av[0], av[1], o);
#pragma unroll
                    for (int pl = 0; pl < 3; ++pl)
                        *reinterpret_cast<unsigned *>(Pl + pl * PLANE16 + ppos + r * RP) = o[pl];
                }
            }
        }

        // ------------------------------------------------------------------------------------------ res/skip conv: 6 steps
        int wvo_rs[3];
        {
            const int r0 = 24 * wave + lrow;                                            // tile 0: residual rows
            const int r1 = lrow < 8 ? 24 * wave + 16 + lrow : (last ? 0 : H) + 24 * wave + 16 + (lrow - 8);
            const int r2 = (last ? 0 : H) + 24 * wave + lrow;                           // tile 2: skip rows
            const int oob = G * Mrs * 32;
            wvo_rs[0] = last ? oob : (r0 * 16 + lk * 4) * 2;
            wvo_rs[1] = (last && lrow < 8) ? oob : (r1 * 16 + lk * 4) * 2;
            wvo_rs[2] = (r2 * 16 + lk * 4) * 2;
        }
        const int wgrp_rs = Mrs * 32;
        auto wload_rs = [&](int s, int slot) {
            s = s < GP ? s : GP - 1;                     // (beyond the last step: that step again, unused)
            const int so0 = 2 * s * wgrp_rs;
            const int so1 = so0 + wgrp_rs;
#pragma unroll
            for (int j = 0; j < 3; ++j)
#pragma unroll
                for (int pl = 0; pl < 3; ++pl) {
                    const i32x2 lo = __builtin_bit_cast(i32x2, __builtin_amdgcn_raw_buffer_load_b64(wr_rs[pl], wvo_rs[j], so0, 0));
                    const i32x2 hi = __builtin_bit_cast(i32x2, __builtin_amdgcn_raw_buffer_load_b64(wr_rs[pl], wvo_rs[j], so1, 0));
                    a[slot][j][pl] = i32x4{lo[0], lo[1], hi[0], hi[1]};
                }
        };
        wload_rs(0, 0);
        wload_rs(1, 1);
        WNF_TRACE(3);
        __syncthreads();                                 // B2: the acts planes are complete
        bfetch(0, 2, 0);
#pragma unroll
        for (int s = 0; s < GP; ++s) {
#pragma unroll
            for (int cc = 0; cc < NCT; ++cc) {
                const int q = s * NCT + cc;
                if (q + 1 < GP * NCT) bfetch((q + 1) / NCT, 2 + ((q + 1) % NCT) * 16, (q + 1) & 1);
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int j = 0; j < 3; ++j)          // (the last layer has no residual rows: their weights read as zeros — a uniform
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
        WNF_TRACE(4);
        __syncthreads();                                 // B3: every wave is through with the acts planes

        // ------------------------------------------------------------------------------------------ residual update, skip biases
        {
            const __amdgpu_buffer_rsrc_t xs_rs = __builtin_amdgcn_make_buffer_rsrc(
                last ? p.skip + (long)b * HT : p.xs + ((long)l * p.B + b) * HT, 0, (int)(HT * 4), 0x00020000);
            int lrow = lrow_, lk = lk_;
            asm volatile("" : "+v"(lrow), "+v"(lk));
            if (!last) {
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    if (j == 1 && lane >= 32) continue;
                    const int ch = 24 * wave + 16 * j + 4 * lk;
                    const f32x4 bb = *reinterpret_cast<const f32x4 *>(Br + ch);
                    const int ppos = ((ch >> 5) * XR) * RP + row_pos(ch);
#pragma unroll
                    for (int cc = 0; cc < NCT; ++cc) {
                        const int r = 2 + 16 * cc + lrow, t = fs + r;
                        const bool own = t >= t0 && t < t0 + NT && t < T;
                        const float m = Ms[r];
                        f32x4 xo = *reinterpret_cast<const f32x4 *>(X32 + (r - 2) * XP + ch);
#pragma unroll
                        for (int g = 0; g < 4; ++g) xo[g] = (xo[g] + racc[j][cc][g] + bb[g]) * m;
                        *reinterpret_cast<f32x4 *>(X32 + (r - 2) * XP + ch) = xo;
                        unsigned oa[3], ob[3];
                        split_planes2<3>(xo[0], xo[1], oa);
                        split_planes2<3>(xo[2], xo[3], ob);
#pragma unroll
                        for (int pl = 0; pl < 3; ++pl) lds_store8(Pl + pl * PLANE16 + ppos + r * RP, oa[pl], ob[pl]);
#pragma unroll
                        for (int g = 0; g < 4; ++g) {
                            const float xv = xo[g];      // (a scalar copy: __builtin_bit_cast applied to the vector ELEMENT stored element 0 four times)
                            __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(xv), xs_rs, own ? ((ch + g) * T + t) * 4 : 0x7fffffff, 0, 0);
                        }
                        racc[j][cc] = f32x4{0.f, 0.f, 0.f, 0.f};
                    }
                }
                lds_stores_done();
            }
            // skip rows: tile 1 lanes >= 32 (channels 24 w + 16 + 4 (lk - 2) + reg) and tile 2 (channels 24 w + 4 lk + reg)
#pragma unroll
            for (int j = 1; j < 3; ++j) {
                if (j == 1 && lane < 32) continue;
                const int ch = j == 1 ? 24 * wave + 16 + 4 * (lk - 2) : 24 * wave + 4 * lk;
                const f32x4 bb = *reinterpret_cast<const f32x4 *>(Br + (last ? 0 : H) + ch);
#pragma unroll
                for (int cc = 0; cc < NCT; ++cc) {
#pragma unroll
                    for (int g = 0; g < 4; ++g) racc[j][cc][g] += bb[g];
                    if (last) {                          // WN's final `output * x_mask` (layers.py:161-162)
                        const int r = 2 + 16 * cc + lrow, t = fs + r;
                        const bool own = t >= t0 && t < t0 + NT && t < T;
                        const float m = Ms[r];
#pragma unroll
                        for (int g = 0; g < 4; ++g)
                            __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(racc[j][cc][g] * m), xs_rs,
                                                                  own ? ((ch + g) * T + t) * 4 : 0x7fffffff, 0, 0);
                    }
                }
            }
        }
        WNF_TRACE(5 + (l & 1));
        // (B4 is the barrier at the top of the next layer's in-conv)
    }
    WNF_TRACE(10);
}

// on unless GLOWTTS_WN_FUSED=0 (read once) or glowtts_wn_fused(0) said otherwise
static std::atomic<int> g_wn_fused{-1};
static std::atomic<int> g_wn_fused_launches{0};
static bool wn_fused_enabled() {
    int v = g_wn_fused.load(std::memory_order_relaxed);
    if (v < 0) {
        const char *e = getenv("GLOWTTS_WN_FUSED");
        v = !(e && e[0] == '0');
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
    hipLaunchKernelGGL(wn_fused_kernel, dim3((unsigned)(p.ntiles * B)), dim3(512), wnf::LDS_BYTES, s, p);
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
