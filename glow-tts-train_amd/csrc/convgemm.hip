// convgemm.hip — fp32 MFMA implicit-GEMM 1-D convolution family for gfx950 (the dense contractions of the WN
// coupling network, layers.py:138-162, and of the 1x1 start/end convs, attentions.py:124-126).
//
// Why hand-written: the first profile (profiles/r01_*_baseline.csv) shows PyTorch-ROCm's path spends, per
// training step, ~18 ms in MIOpen igemm kernels PLUS ~5 ms transposing NCHW<->NHWC around every call, ~10 ms in
// ~1 700 tiny bias / mask / add / reduce launches, and falls back to "naive_conv" kernels for strided inputs.
// These kernels consume the reference layout (B, C, T) directly, keep T contiguous for coalesced 64-B row
// segments, and fuse bias, mask, the WN gate (tanh * sigmoid, optional dropout + conditioning) and the
// residual/skip update into the GEMM epilogue.
//
// Arithmetic: v_mfma_f32_16x16x4_f32 — exact fp32 FMA chains (MI355X_MICROARCH.md: 157 TFLOP/s dense peak, same
// as the vector rate, no reduced-precision path), so results match an fp32 reference to rounding-order noise.
//
// Forward-type kernel (conv fwd, conv bwd-data with transposed/flipped packed weights, 1x1 GEMMs):
//     out[m, t] = sum_tap sum_k Wp[tap][k][m] * X[k][t + tap*dil - pad]            per utterance b
//   workgroup = 4 waves = (64*RTW rows) x (16*NCT columns) of one utterance; wave = RTW x NCT tiles of 16x16;
//   K loop over 16-channel chunks: packed weights [tap][16][rows] and the activation slab [16][cols + halo] are
//   staged in LDS (row pitches chosen == 16 mod 32 so the two k-rows a 32-lane group touches fall on disjoint
//   banks); every tap re-reads the same activation slab at a shifted column, which is what makes the k-tap
//   convolution an "implicit" GEMM with no im2col traffic.
// Weight-gradient kernel:
//     dWp[tap][k][m] += sum_{b,t} X[b][k][t + tap*dil - pad] * D[b][m][t]
//   A = X rows (k), B = D rows (m), contraction over time; split over utterance groups, one float atomic per
//   element and workgroup (64-B contiguous segments along m).
#include "convgemm_common.hpp"

namespace glowtts {


// ------------------------------------------------------------------------------------------------------------
// Forward-type kernel history (profiles/, tools/mfma_rate.hip, tools/trace_conv.py):
//   v1 staged through run-time loops (serialised loads, 40 % of the fp32 MFMA peak);
//   v2 prefetched chunk c+1 into registers behind the MFMAs of chunk c but fed every MFMA pair from ds_read_b32 —
//      7 LDS instructions per 10 MFMAs, a mix whose measured ceiling is 69 % of peak;
//   v3 laid both operands out with 16 consecutive k per LDS row ("k-packed"), so ONE ds_read_b128 per operand tile
//      feeds FOUR k-steps: lane (row = l & 15, slot = l >> 4) consumes k = 4*slot + j in step j for A and B alike — a
//      relabelling of the reduction index the MFMA is free to make.  Weights AND activations went through LDS, 16
//      channels per chunk, two barriers per chunk (94 TFLOP/s on the gated in-conv);
//   v4 (below) keeps v3's activation image and takes the weights out of LDS.
// ------------------------------------------------------------------------------------------------------------
// v4: weights straight from L2 into MFMA registers ("weights-direct").
//
// A wave's A operand (its own 16-row weight tiles) is used by no other wave, so routing it through LDS only costs LDS
// bandwidth and ties every 16-channel step to a pair of workgroup barriers.  The packed layout [tap][g][M][16] is
// already the MFMA A layout: lane (row = l & 15, slot = l >> 4) needs the 16 bytes at row*64 + slot*16 of a 1 KB
// contiguous block — one fully coalesced global_load_dwordx4 per 16x16 tile and (tap, g) step, feeding 4*NCT MFMAs.
// A 3-slot register ring keeps the load three steps ahead.  LDS then holds activations only, so a chunk can span
// KG = 6 groups (96 channels, all taps): 2 chunks and 4 barriers for Cin = 192 instead of 12 chunks and 24 barriers.
//   Xs[g][frame][20] : 16 consecutive channels per row, pitch 20 floats (conflict-free ds_read_b128); activations
//                      arrive [k][frame], each 16-byte load is scattered as 4 ds_write_b32; taps shift the frame row.
// vmcnt retires in order: the activation prefetch of chunk c+1 is issued at step 0 AFTER the weight load for step 3,
// so no weight wait before step 4 has to drain it.
// ------------------------------------------------------------------------------------------------------------
template <int RTW, int NCT, int EPI, int TAPS>
__global__ __launch_bounds__(256, 2) void convgemm_wd_kernel(ConvGemmParams p) {
    constexpr int WGR = 64 * RTW, NT = 16 * NCT, XC = NT + 16, KP = 20, KG = 6;
    constexpr int X4 = KG * 16 * (XC / 4);           // 16-byte loads per activation chunk
    constexpr int NX = (X4 + 255) / 256;
    constexpr int S = KG * TAPS;                     // MFMA steps per chunk (a multiple of 3: the ring stays static)
    static_assert(S % 3 == 0, "ring of 3");
    extern __shared__ __align__(16) float smem[];
    float *Xs = smem;                                // [KG][XC][KP]

    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int lrow = lane & 15, lk = lane >> 4;
    const int ntile_t = (p.T + NT - 1) / NT;
    const int b = blockIdx.x / ntile_t;
    const int t0 = (blockIdx.x - b * ntile_t) * NT;
    const int tile_m = blockIdx.y;
    const int off = (4 - (p.pad & 3)) & 3;           // window start ts = t0 - pad - off is a multiple of 4
    const int ts = t0 - p.pad - off;
    const int G = (p.Cin + 15) / 16;
    const int nchunks = (G + KG - 1) / KG;

    auto grow = [&](int lr) -> int {
        if (EPI == EPI_GATE) return lr < 64 ? tile_m * 64 + lr : p.H + tile_m * 64 + (lr - 64);
        return tile_m * WGR + lr;
    };
    auto row_ok = [&](int lr) -> bool {
        if (EPI == EPI_GATE) return tile_m * 64 + (lr & 63) < p.H;
        return tile_m * WGR + lr < p.M;
    };
    auto ltile = [&](int r) -> int { return (EPI == EPI_GATE) ? (r == 0 ? wave : 4 + wave) : wave * RTW + r; };

    f32x4 acc[RTW][NCT];
#pragma unroll
    for (int r = 0; r < RTW; ++r)
#pragma unroll
        for (int c = 0; c < NCT; ++c) acc[r][c] = f32x4{0.f, 0.f, 0.f, 0.f};

    // ---- weights: buffer loads (range-checked by the hardware: an offset past the descriptor returns zeros, so rows
    // beyond M and groups beyond G need no branch and no select of pointers — a select between two address spaces
    // makes the compiler emit FLAT loads, which also count against lgkmcnt and stall every LDS wait behind L2 latency)
    const int wbytes = TAPS * G * p.M * 64;
    const __amdgpu_buffer_rsrc_t wrs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(p.wp), 0, wbytes, 0x00020000);
    int wvo[RTW];                                    // this lane's byte offset inside a (tap, g) block, per row tile
#pragma unroll
    for (int r = 0; r < RTW; ++r) {
        const int lr = ltile(r) * 16 + lrow;
        wvo[r] = row_ok(lr) ? (grow(lr) * 16 + lk * 4) * 4 : wbytes;
    }
    const int wtap = G * p.M * 64, wgrp = p.M * 64;  // bytes
    f32x4 a[3][RTW];
    auto wload = [&](int c, int s, int slot) {       // weights of step s of chunk c (s may run past the chunk: next chunk)
        if (s >= S) { s -= S; c += 1; }
        const int g = c * KG + s / TAPS, tap = s % TAPS;
        const int so = g < G ? tap * wtap + g * wgrp : wbytes;
#pragma unroll
        for (int r = 0; r < RTW; ++r)
            a[slot][r] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(wrs, wvo[r], so, 0));
    };

    // ---- activations: thread = (channel kk of a group, frame quad qq, group parity gsel); its 9 pieces are
    // (group gsel + 2 gi, quad qq + 8 jq): every source / LDS offset is one per-thread base plus compile-time multiples,
    // so the staging costs no address registers.  The mask window sits in LDS (read at store time, not from HBM).
    const float *xb = p.x + (long)b * p.x_bs;
    const float *mk = p.mask ? p.mask + (long)b * p.T : nullptr;
    float *Ms = smem + KG * XC * KP;                 // [XC]
    constexpr int NQ = (XC / 4 + 7) / 8, NG = KG / 2;
    static_assert(NQ * NG * 256 >= X4, "piece map covers the chunk");
    const int kk = tid & 15, qq = (tid >> 4) & 7, gsel = tid >> 7;
    const int c_first = p.x2 ? p.x_split : p.Cin;    // channels served by the first source
    const int xbytes = c_first * p.T * 4;            // channels past the source's count fall outside the descriptor: zeros
    const __amdgpu_buffer_rsrc_t xrs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(xb), 0, xbytes, 0x00020000);
    const int x2bytes = p.x2 ? (p.Cin - p.x_split) * p.T * 4 : 0;
    const __amdgpu_buffer_rsrc_t xrs2 = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float *>(p.x2 ? p.x2 + (long)b * p.x2_bs : xb), 0, x2bytes, 0x00020000);
    int xvo[NQ];                                     // byte offset of (channel gsel*16 + kk, frame quad qq + 8 jq)
#pragma unroll
    for (int jq = 0; jq < NQ; ++jq) {
        const int t = ts + (qq + 8 * jq) * 4;
        const bool ok = (qq + 8 * jq < XC / 4) && t >= 0 && t < p.T;
        xvo[jq] = ok ? ((gsel * 16 + kk) * p.T + t) * 4 : 0x7fffffff;
    }
    const int dbase = (gsel * XC + qq * 4) * KP + kk;
    f32x4 xreg[NG][NQ];
    if (p.mask_in && tid < XC) {
        const int t = ts + tid;
        Ms[tid] = (t >= 0 && t < p.T) ? mk[t] : 0.f;
    }
    auto xload = [&](int c) {
        // a chunk (96 channels) lies entirely in one source: the choice is uniform over the workgroup
        const bool second = p.x2 != nullptr && c * KG * 16 >= p.x_split;
        const int cbase = c * KG * 16 - (second ? p.x_split : 0);
#pragma unroll
        for (int gi = 0; gi < NG; ++gi)
#pragma unroll
            for (int jq = 0; jq < NQ; ++jq)
                xreg[gi][jq] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(
                    second ? xrs2 : xrs, xvo[jq], (cbase + 2 * gi * 16) * p.T * 4, 0));
    };
    auto xstore = [&]() {
#pragma unroll
        for (int gi = 0; gi < NG; ++gi)
#pragma unroll
            for (int jq = 0; jq < NQ; ++jq)
                if (qq + 8 * jq < XC / 4) {
                    f32x4 v = xreg[gi][jq];
                    if (p.mask_in) v *= *reinterpret_cast<const f32x4 *>(Ms + (qq + 8 * jq) * 4);
                    float *d = Xs + dbase + (2 * gi * XC + 32 * jq) * KP;
                    d[0] = v[0]; d[KP] = v[1]; d[2 * KP] = v[2]; d[3 * KP] = v[3];
                }
    };
    const float *xd = Xs + (off + lrow) * KP + lk * 4;
    f32x4 bv[2][NCT];
    auto bfetch = [&](int s, int slot) {
        const int g = s / TAPS, tap = s % TAPS;
#pragma unroll
        for (int c = 0; c < NCT; ++c)
            bv[slot][c] = *reinterpret_cast<const f32x4 *>(xd + (g * XC + c * 16 + tap * p.dil) * KP);
    };

    GLOWTTS_TRACE_POINT(0);
    wload(0, 0, 0);
    wload(0, 1, 1);
    wload(0, 2, 2);
    xload(0);
    if (p.mask_in) __syncthreads();                 // mask window visible
    xstore();
    __syncthreads();
    GLOWTTS_TRACE_POINT(1);
    for (int c = 0; c < nchunks; ++c) {
        const bool more = c + 1 < nchunks;
        bfetch(0, 0);
#pragma unroll
        for (int s = 0; s < S; ++s) {
            if (s + 1 < S) bfetch(s + 1, (s + 1) & 1);
            __builtin_amdgcn_sched_barrier(0);      // keep the next step's LDS reads ahead of this step's MFMAs
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int r = 0; r < RTW; ++r)
#pragma unroll
                    for (int cc = 0; cc < NCT; ++cc)
                        acc[r][cc] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[s % 3][r][j], bv[s & 1][cc][j], acc[r][cc], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
            wload(c, s + 3, s % 3);                 // refill the slot just consumed: three steps of lead
            if (s == 0 && more) xload(c + 1);       // HBM/L2 -> registers, behind the MFMAs of this chunk
            __builtin_amdgcn_sched_barrier(0);
        }
        GLOWTTS_TRACE_POINT(2 + 2 * (c & 3));
        mfma_settle();                              // (the loop's branches stand between the last MFMAs and the accumulators' next read)
        __syncthreads();                            // every wave is done reading the LDS image
        if (more) {
            xstore();
            __syncthreads();
        }
        GLOWTTS_TRACE_POINT(3 + 2 * (c & 3));
    }
    if (p.vec_epilogue) {
        conv_epilogue_lds<RTW, NCT, EPI>(p, acc, smem, b, t0, tile_m, wave, lane);   // LDS is free after the last barrier
    } else {
        conv_epilogue<RTW, NCT, EPI>(p, acc, b, t0, tile_m, wave, lane);
    }
    GLOWTTS_TRACE_POINT(10);
}

// ------------------------------------------------------------------------------------------------------------
// weight gradient
// ------------------------------------------------------------------------------------------------------------

// workgroup: 64 input channels (4 k-tiles, one per wave) x 128 output channels (8 m-tiles per wave), one tap,
// nb utterances.  Time is consumed in chunks of 64 frames.
__global__ __launch_bounds__(256) void convwrw_kernel(ConvWrwParams p) {
    constexpr int CT = 64;
    extern __shared__ __align__(16) float smem[];
    const int XP = p.xs_pitch, DP = p.ds_pitch;
    float *Xs = smem;                 // [64][XP]   rows = input channel, cols = frames (shifted by the tap)
    float *Ds = smem + 64 * XP;       // [128][DP]  rows = output channel
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int lrow = lane & 15, lk = lane >> 4;
    const int nkt = (p.Cin + 63) / 64;
    const int kt = blockIdx.x % nkt, mt = blockIdx.x / nkt;
    const int tap = blockIdx.y;
    const int b0 = blockIdx.z * p.nb;
    const int b1 = min(p.B, b0 + p.nb);
    const int k0 = kt * 64, m0 = mt * 128;
    const int shift = tap * p.dil - p.pad;

    f32x4 acc[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};

    for (int b = b0; b < b1; ++b) {
        const float *xb = p.x + (long)b * p.x_bs;
        const float *db = p.d + (long)b * p.d_bs;
        for (int tc = 0; tc < p.T; tc += CT) {
            for (int idx = tid; idx < 64 * CT; idx += 256) {
                const int r = idx / CT, j = idx - r * CT;
                const int t = tc + j + shift;
                float v = 0.f;
                if (k0 + r < p.Cin && tc + j < p.T && t >= 0 && t < p.T) {
                    v = xb[(long)(k0 + r) * p.T + t];
                    if (p.mask_x) v *= p.mask_x[(long)b * p.T + t];
                }
                Xs[r * XP + j] = v;
            }
            for (int idx = tid; idx < 128 * CT; idx += 256) {
                const int r = idx / CT, j = idx - r * CT;
                float v = 0.f;
                if (m0 + r < p.M && tc + j < p.T) {
                    v = db[(long)(m0 + r) * p.T + tc + j];
                    if (p.mask) v *= p.mask[(long)b * p.T + tc + j];
                }
                Ds[r * DP + j] = v;
            }
            __syncthreads();
#pragma unroll 4
            for (int c4 = 0; c4 < CT / 4; ++c4) {
                const float a = Xs[(wave * 16 + lrow) * XP + c4 * 4 + lk];          // A[row = channel][k = frame]
#pragma unroll
                for (int i = 0; i < 8; ++i) {
                    const float bv = Ds[(i * 16 + lrow) * DP + c4 * 4 + lk];         // B[k = frame][col = out channel]
                    acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, bv, acc[i], 0, 0, 0);
                }
            }
            __syncthreads();
        }
    }
#pragma unroll
    for (int i = 0; i < 8; ++i) {
#pragma unroll
        for (int reg = 0; reg < 4; ++reg) {
            const int k = k0 + wave * 16 + lk * 4 + reg;
            const int m = m0 + i * 16 + lrow;
            if (k < p.Cin && m < p.M) atomicAdd(p.dwp + ((long)tap * p.Cin + k) * p.M + m, acc[i][reg]);
        }
    }
}

// software-pipelined weight gradient: workgroup = 64 input channels (one 16-row k-tile per wave) x 64 output channels
// (4 m-tiles per wave) x ALL taps (each tap is the same staged X slab read at a shifted column), contraction over the
// frames of `nb` utterances in chunks of CT frames; chunk c+1 travels HBM/L2 -> registers while chunk c feeds the MFMAs.
constexpr int cpitch2(int n) { return ((n - 2 + 31) / 32 * 32 + 2) < n ? ((n - 2 + 31) / 32 * 32 + 2) + 32 : ((n - 2 + 31) / 32 * 32 + 2); }

template <int TAPS, int CT>
__global__ __launch_bounds__(256) void convwrw_pipe_kernel(ConvWrwParams p) {
    constexpr int XC = CT + 16;
    constexpr int XP = cpitch2(XC), DP = cpitch2(CT);          // pitches == 2 (mod 32): conflict-free operand reads
    constexpr int X4 = 64 * (XC / 4), D4 = 64 * (CT / 4);
    constexpr int NX = (X4 + 255) / 256, ND = (D4 + 255) / 256;
    constexpr int XSZ = 64 * XP, DSZ = 64 * DP;
    extern __shared__ __align__(16) float smem[];
    float *Xs = smem;                // [2][XSZ]
    float *Ds = smem + 2 * XSZ;      // [2][DSZ]
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int lrow = lane & 15, lk = lane >> 4;
    const int nkt = (p.Cin + 63) / 64;
    // XCD-aware order: hardware deals workgroup ids round-robin over the 8 XCDs (own L2 each).  All tiles of one split
    // read the same x / d chunks, so logical work items are numbered split-major and XCD x takes a contiguous range of
    // them: a chunk is then fetched by one or two XCDs instead of all eight.
    const int ntiles = gridDim.x, nwg = gridDim.x * gridDim.z;
    const int id = blockIdx.x + blockIdx.z * gridDim.x;
    const int xcd = id & 7, slot = id >> 3;
    const int item = xcd * (nwg >> 3) + min(xcd, nwg & 7) + slot;
    const int tile = item % ntiles, split = item / ntiles;
    const int kt = tile % nkt, mt = tile / nkt;
    const int k0 = kt * 64, m0 = mt * 64;
    const int off = (4 - (p.pad & 3)) & 3;
    const int nct = (p.T + CT - 1) / CT;             // chunks per utterance
    const int c0 = split * p.nb;                     // this workgroup's range of the B * nct (utterance, chunk) pairs
    const int nchunks = min(p.B * nct, c0 + p.nb) - c0;

    f32x4 acc[TAPS][4];
#pragma unroll
    for (int tp = 0; tp < TAPS; ++tp)
#pragma unroll
        for (int i = 0; i < 4; ++i) acc[tp][i] = f32x4{0.f, 0.f, 0.f, 0.f};
    float4 xreg[NX], dreg[ND], mxreg[NX], mdreg[ND];
    float bsum[ND];                                   // bias gradient: running sums of this thread's D pieces
#pragma unroll
    for (int i = 0; i < ND; ++i) bsum[i] = 0.f;
    const bool do_bias = (p.dbias != nullptr) && (kt == 0);

    // Range-checked buffer loads: a piece outside the tensor (row past Cin / M, frame outside [0, T), index past the
    // tile) gets an offset beyond the descriptor and reads zeros — no exec-mask branch around any load, and nothing
    // touches a loaded value before the MFMAs (masks are applied when the chunk is stored to LDS).
    const int xbytes = (int)(((long)(p.B - 1) * p.x_bs + (long)p.Cin * p.T) * 4);
    const int dbytes = (int)(((long)(p.B - 1) * p.d_bs + (long)p.M * p.T) * 4);
    const __amdgpu_buffer_rsrc_t xrs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(p.x), 0, xbytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t drs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(p.d), 0, dbytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t mdrs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(p.mask), 0, p.mask ? p.B * p.T * 4 : 0, 0x00020000);
    const __amdgpu_buffer_rsrc_t mxrs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(p.mask_x), 0, p.mask_x ? p.B * p.T * 4 : 0, 0x00020000);
    constexpr int kOOB = 0x7fffffff;                  // beyond any descriptor
    int xrow[NX], xq[NX], drow[ND], dq[ND];           // element offset of the piece's row / frame offset in the window
#pragma unroll
    for (int i = 0; i < NX; ++i) {
        const int idx = tid + i * 256;
        const int q = idx % (XC / 4), r = idx / (XC / 4);
        xq[i] = q * 4;
        xrow[i] = (idx < X4 && k0 + r < p.Cin) ? (k0 + r) * p.T : -1;
    }
#pragma unroll
    for (int i = 0; i < ND; ++i) {
        const int idx = tid + i * 256;
        const int q = idx % (CT / 4), r = idx / (CT / 4);
        dq[i] = q * 4;
        drow[i] = (idx < D4 && m0 + r < p.M) ? (m0 + r) * p.T : -1;
    }
    auto ld16 = [&](const __amdgpu_buffer_rsrc_t &rs, int byte_off) -> float4 {
        return __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(rs, byte_off, 0, 0));
    };
    auto load_chunk = [&](int c) {
        const int b = (c0 + c) / nct;
        const int tc = ((c0 + c) % nct) * CT;
        const int ts = tc - p.pad - off;
        const int xb = b * (int)p.x_bs, db = b * (int)p.d_bs;
#pragma unroll
        for (int i = 0; i < NX; ++i) {
            const int t = ts + xq[i];
            const bool ok = xrow[i] >= 0 && t >= 0 && t < p.T;
            xreg[i] = ld16(xrs, ok ? (xb + xrow[i] + t) * 4 : kOOB);
            if (p.mask_x) mxreg[i] = ld16(mxrs, ok ? (b * p.T + t) * 4 : kOOB);
        }
#pragma unroll
        for (int i = 0; i < ND; ++i) {
            const int t = tc + dq[i];
            const bool ok = drow[i] >= 0 && t < p.T;
            dreg[i] = ld16(drs, ok ? (db + drow[i] + t) * 4 : kOOB);
            if (p.mask) mdreg[i] = ld16(mdrs, ok ? (b * p.T + t) * 4 : kOOB);
        }
    };
    auto store_chunk = [&](int buf) {
        float *xd = Xs + buf * XSZ;
        float *dd = Ds + buf * DSZ;
#pragma unroll
        for (int i = 0; i < NX; ++i) {
            const int idx = tid + i * 256;
            const int q = idx % (XC / 4), r = idx / (XC / 4);
            if (idx < X4) {           // pitch is even, not a multiple of 4: two 8-byte stores
                float4 v = xreg[i];
                if (p.mask_x) { v.x *= mxreg[i].x; v.y *= mxreg[i].y; v.z *= mxreg[i].z; v.w *= mxreg[i].w; }
                float2 *d2 = reinterpret_cast<float2 *>(xd + r * XP + q * 4);
                d2[0] = make_float2(v.x, v.y);
                d2[1] = make_float2(v.z, v.w);
            }
        }
#pragma unroll
        for (int i = 0; i < ND; ++i) {
            const int idx = tid + i * 256;
            const int q = idx % (CT / 4), r = idx / (CT / 4);
            if (idx < D4) {
                float4 v = dreg[i];
                if (p.mask) { v.x *= mdreg[i].x; v.y *= mdreg[i].y; v.z *= mdreg[i].z; v.w *= mdreg[i].w; }
                if (do_bias) bsum[i] += (v.x + v.y) + (v.z + v.w);
                float2 *d2 = reinterpret_cast<float2 *>(dd + r * DP + q * 4);
                d2[0] = make_float2(v.x, v.y);
                d2[1] = make_float2(v.z, v.w);
            }
        }
    };
    auto compute = [&](int buf) {
        const float *xd = Xs + buf * XSZ + (wave * 16 + lrow) * XP + off + lk;
        const float *dd = Ds + buf * DSZ + lrow * DP + lk;
        constexpr int S = CT / 4;
        float a[2][TAPS], bv[2][4];
        auto fetch = [&](int c4, int slot) {
#pragma unroll
            for (int tp = 0; tp < TAPS; ++tp) a[slot][tp] = xd[c4 * 4 + tp * p.dil];   // A[row = channel][k = frame + tap shift]
#pragma unroll
            for (int i = 0; i < 4; ++i) bv[slot][i] = dd[i * 16 * DP + c4 * 4];        // B[k = frame][col = out channel]
        };
        fetch(0, 0);
#pragma unroll
        for (int c4 = 0; c4 < S; ++c4) {
            if (c4 + 1 < S) fetch(c4 + 1, (c4 + 1) & 1);
#pragma unroll
            for (int tp = 0; tp < TAPS; ++tp)
#pragma unroll
                for (int i = 0; i < 4; ++i)
                    acc[tp][i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[c4 & 1][tp], bv[c4 & 1][i], acc[tp][i], 0, 0, 0);
        }
    };

    if (nchunks > 0) {
        load_chunk(0);
        store_chunk(0);
    }
    __syncthreads();
    for (int c = 0; c < nchunks; ++c) {
        const bool more = c + 1 < nchunks;
        if (more) load_chunk(c + 1);
        compute(c & 1);
        if (more) store_chunk((c + 1) & 1);
        __syncthreads();
    }
    if (k0 + 64 <= p.Cin && m0 + 64 <= p.M) {         // whole tile inside (workgroup-uniform): no per-lane predicates
        float *base = p.dwp + (long)(k0 + wave * 16 + lk * 4) * p.M + m0 + lrow;
#pragma unroll
        for (int tp = 0; tp < TAPS; ++tp)
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int reg = 0; reg < 4; ++reg)
                    atomicAdd(base + ((long)tp * p.Cin + reg) * p.M + i * 16, acc[tp][i][reg]);
    } else {
#pragma unroll
        for (int tp = 0; tp < TAPS; ++tp)
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int reg = 0; reg < 4; ++reg) {
                    const int k = k0 + wave * 16 + lk * 4 + reg;
                    const int m = m0 + i * 16 + lrow;
                    if (k < p.Cin && m < p.M) atomicAdd(p.dwp + ((long)tp * p.Cin + k) * p.M + m, acc[tp][i][reg]);
                }
    }
    if (do_bias) {      // a thread's piece i always belongs to row (tid + 256 i) / (CT/4): reduce rows in LDS, then one atomic each
        float *rowacc = smem;
        if (tid < 64) rowacc[tid] = 0.f;
        __syncthreads();
#pragma unroll
        for (int i = 0; i < ND; ++i) {
            const int idx = tid + i * 256;
            if (idx < D4) atomicAdd(rowacc + idx / (CT / 4), bsum[i]);
        }
        __syncthreads();
        if (tid < 64 && m0 + tid < p.M) atomicAdd(p.dbias + m0 + tid, rowacc[tid]);
    }
}

// ------------------------------------------------------------------------------------------------------------
// weight gradient, "frame-packed" (dil == 1, 'same' padding): the contraction index is the frame, and the same
// relabelling as in the forward kernel applies — lane slot lk supplies frames 16G + 4 lk + j to MFMA j of a 16-frame
// group G, for A (x rows) and B (d rows) alike.  One ds_read_b128 per d tile then feeds 4 MFMAs, and the 5 tap shifts
// of an x row are 12 consecutive floats = 3 aligned ds_read_b128 feeding 20 (tap, j) pairs: 7 LDS instructions per
// 80 MFMAs instead of 36 ds_read_b32 (a mix measured at <= 80 % of the MFMA rate, tools/mfma_rate.hip).
//   Xs[64 k-rows][XP], Ds[64 m-rows][DP], pitches == 4 (mod 8) floats: conflict-free b128 reads, and the staging
//   store is the 16-byte load written back as one ds_write_b128.
// One LDS image; chunk c+1 waits in registers behind the MFMAs of chunk c.  Masks go through a small LDS window.
// ------------------------------------------------------------------------------------------------------------
__host__ __device__ constexpr int cpitch4(int w) { int p = (w + 3) / 4 * 4; while (p % 8 != 4) p += 4; return p; }

template <int TAPS, int NGRP, int MT>      // MT = 16-row output-gradient tiles per wave: workgroup tile 64 k x 16*MT m
__global__ __launch_bounds__(256, (MT <= 2 ? 3 : 2)) void convwrw_fp_kernel(ConvWrwParams p) {
    constexpr int MR = 16 * MT;
    constexpr int CT = 16 * NGRP, XC = CT + 16;
    constexpr int XP = cpitch4(XC), DP = cpitch4(CT);
    constexpr int PAD = (TAPS - 1) / 2, OFF = (4 - (PAD & 3)) & 3;
    constexpr int NA = (3 + TAPS + OFF + 3) / 4;                 // b128 reads covering floats [0, 3 + TAPS - 1 + OFF]
    constexpr int X4 = 64 * (XC / 4), D4 = MR * (CT / 4);
    constexpr int NX = (X4 + 255) / 256, ND = (D4 + 255) / 256;
    extern __shared__ __align__(16) float smem[];
    float *Xs = smem;                        // [64][XP]
    float *Ds = smem + 64 * XP;              // [MR][DP]
    float *Mx = Ds + MR * DP;                // [XC]  x-mask window of the chunk being stored
    float *Md = Mx + XC;                     // [CT]
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int lrow = lane & 15, lk = lane >> 4;
    const int nkt = (p.Cin + 63) / 64;
    // XCD-aware order (see convwrw_pipe_kernel): split-major work items, contiguous ranges per XCD
    const int ntiles = gridDim.x, nwg = gridDim.x * gridDim.z;
    const int id = blockIdx.x + blockIdx.z * gridDim.x;
    const int xcd = id & 7, slot = id >> 3;
    const int item = xcd * (nwg >> 3) + min(xcd, nwg & 7) + slot;
    const int tile = item % ntiles, split = item / ntiles;
    const int kt = tile % nkt, mt = tile / nkt;
    const int k0 = kt * 64, m0 = mt * MR;
    const int nct = (p.T + CT - 1) / CT;
    const int c0 = split * p.nb;
    const int nchunks = min(p.B * nct, c0 + p.nb) - c0;

    f32x4 acc[TAPS][MT];
#pragma unroll
    for (int tp = 0; tp < TAPS; ++tp)
#pragma unroll
        for (int i = 0; i < MT; ++i) acc[tp][i] = f32x4{0.f, 0.f, 0.f, 0.f};
    f32x4 xreg[NX], dreg[ND], mreg;
    float bsum[ND];
#pragma unroll
    for (int i = 0; i < ND; ++i) bsum[i] = 0.f;
    const bool do_bias = (p.dbias != nullptr) && (kt == 0);
    const bool masked = (p.mask != nullptr) || (p.mask_x != nullptr);

    const int xbytes = (int)(((long)(p.B - 1) * p.x_bs + (long)p.Cin * p.T) * 4);
    // the 64 output-gradient rows of this workgroup lie entirely in one of the (up to) two d sources
    const bool d_second = p.d2 != nullptr && m0 >= p.d_split;
    const int m_rows = d_second ? p.M - p.d_split : (p.d2 ? p.d_split : p.M);   // rows of the chosen source
    const int m_base = d_second ? m0 - p.d_split : m0;                        // first row of this tile inside it
    const long d_bs = d_second ? p.d2_bs : p.d_bs;
    const int dbytes = (int)(((long)(p.B - 1) * d_bs + (long)m_rows * p.T) * 4);
    const __amdgpu_buffer_rsrc_t xrs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(p.x), 0, xbytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t drs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(d_second ? p.d2 : p.d), 0, dbytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t mdrs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(p.mask), 0, p.mask ? p.B * p.T * 4 : 0, 0x00020000);
    const __amdgpu_buffer_rsrc_t mxrs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(p.mask_x), 0, p.mask_x ? p.B * p.T * 4 : 0, 0x00020000);
    constexpr int kOOB = 0x7fffffff;
    auto ld16 = [&](const __amdgpu_buffer_rsrc_t &rs, int byte_off) -> f32x4 {
        return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs, byte_off, 0, 0));
    };
    int xrow[NX], xq[NX], drow[ND], dq[ND];
#pragma unroll
    for (int i = 0; i < NX; ++i) {
        const int idx = tid + i * 256;
        const int q = idx % (XC / 4), r = idx / (XC / 4);
        xq[i] = q * 4;
        xrow[i] = (idx < X4 && k0 + r < p.Cin) ? (k0 + r) * p.T : -1;
    }
#pragma unroll
    for (int i = 0; i < ND; ++i) {
        const int idx = tid + i * 256;
        const int q = idx % (CT / 4), r = idx / (CT / 4);
        dq[i] = q * 4;
        drow[i] = (idx < D4 && m_base + r < m_rows) ? (m_base + r) * p.T : -1;
    }
    // mask windows: thread tid < XC/4 carries 4 x-mask frames, thread 64 + j < 64 + CT/4 carries 4 d-mask frames
    const bool mx_thread = tid < XC / 4, md_thread = tid >= 64 && tid < 64 + CT / 4;

    auto load_chunk = [&](int c) {
        const int b = (c0 + c) / nct;
        const int tc = ((c0 + c) % nct) * CT;
        const int ts = tc - PAD - OFF;
        const int xb = b * (int)p.x_bs, db = b * (int)d_bs;
#pragma unroll
        for (int i = 0; i < NX; ++i) {
            const int t = ts + xq[i];
            const bool ok = xrow[i] >= 0 && t >= 0 && t < p.T;
            xreg[i] = ld16(xrs, ok ? (xb + xrow[i] + t) * 4 : kOOB);
        }
#pragma unroll
        for (int i = 0; i < ND; ++i) {
            const int t = tc + dq[i];
            const bool ok = drow[i] >= 0 && t < p.T;
            dreg[i] = ld16(drs, ok ? (db + drow[i] + t) * 4 : kOOB);
        }
        if (masked) {
            const int tx = ts + tid * 4, td = tc + (tid - 64) * 4;
            if (mx_thread) mreg = ld16(mxrs, (tx >= 0 && tx < p.T) ? (b * p.T + tx) * 4 : kOOB);
            if (md_thread) mreg = ld16(mdrs, (td < p.T) ? (b * p.T + td) * 4 : kOOB);
        }
    };
    auto store_chunk = [&]() {
        if (masked) {                                   // workgroup-uniform
            if (mx_thread) *reinterpret_cast<f32x4 *>(Mx + tid * 4) = mreg;
            if (md_thread) *reinterpret_cast<f32x4 *>(Md + (tid - 64) * 4) = mreg;
            __syncthreads();
        }
#pragma unroll
        for (int i = 0; i < NX; ++i) {
            const int idx = tid + i * 256;
            const int q = idx % (XC / 4), r = idx / (XC / 4);
            if (idx < X4) {
                f32x4 v = xreg[i];
                if (p.mask_x) v *= *reinterpret_cast<const f32x4 *>(Mx + q * 4);
                *reinterpret_cast<f32x4 *>(Xs + r * XP + q * 4) = v;
            }
        }
#pragma unroll
        for (int i = 0; i < ND; ++i) {
            const int idx = tid + i * 256;
            const int q = idx % (CT / 4), r = idx / (CT / 4);
            if (idx < D4) {
                f32x4 v = dreg[i];
                if (p.mask) v *= *reinterpret_cast<const f32x4 *>(Md + q * 4);
                if (do_bias) bsum[i] += (v[0] + v[1]) + (v[2] + v[3]);
                *reinterpret_cast<f32x4 *>(Ds + r * DP + q * 4) = v;
            }
        }
    };
    const float *xa = Xs + (wave * 16 + lrow) * XP + lk * 4;
    const float *db_ = Ds + lrow * DP + lk * 4;
    auto compute = [&]() {
        f32x4 av[2][NA], bv[2][MT];
        auto fetch = [&](int g, int sl) {
#pragma unroll
            for (int n = 0; n < NA; ++n) av[sl][n] = *reinterpret_cast<const f32x4 *>(xa + g * 16 + n * 4);
#pragma unroll
            for (int i = 0; i < MT; ++i) bv[sl][i] = *reinterpret_cast<const f32x4 *>(db_ + i * 16 * DP + g * 16);
        };
        fetch(0, 0);
#pragma unroll
        for (int g = 0; g < NGRP; ++g) {
            if (g + 1 < NGRP) fetch(g + 1, (g + 1) & 1);
            __builtin_amdgcn_sched_barrier(0);
            const int sl = g & 1;
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int tp = 0; tp < TAPS; ++tp)
#pragma unroll
                    for (int i = 0; i < MT; ++i)
                        acc[tp][i] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[sl][(j + tp + OFF) >> 2][(j + tp + OFF) & 3],
                                                                          bv[sl][i][j], acc[tp][i], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
        }
    };

    GLOWTTS_TRACE_POINT_Z(0);
    if (nchunks > 0) {
        load_chunk(0);
        store_chunk();
    }
    __syncthreads();
    GLOWTTS_TRACE_POINT_Z(1);
    for (int c = 0; c < nchunks; ++c) {
        const bool more = c + 1 < nchunks;
        if (more) load_chunk(c + 1);
        compute();
        if (c == 0) GLOWTTS_TRACE_POINT_Z(2);
        __syncthreads();
        if (more) {
            store_chunk();
            __syncthreads();
        }
        if (c == 0) GLOWTTS_TRACE_POINT_Z(3);
    }
    GLOWTTS_TRACE_POINT_Z(4);
    if (k0 + 64 <= p.Cin && m0 + MR <= p.M) {         // whole tile inside (workgroup-uniform): no per-lane predicates
        float *base = p.dwp + (long)(k0 + wave * 16 + lk * 4) * p.M + m0 + lrow;
#pragma unroll
        for (int tp = 0; tp < TAPS; ++tp)
#pragma unroll
            for (int i = 0; i < MT; ++i)
#pragma unroll
                for (int reg = 0; reg < 4; ++reg)
                    atomicAdd(base + ((long)tp * p.Cin + reg) * p.M + i * 16, acc[tp][i][reg]);
    } else {
#pragma unroll
        for (int tp = 0; tp < TAPS; ++tp)
#pragma unroll
            for (int i = 0; i < MT; ++i)
#pragma unroll
                for (int reg = 0; reg < 4; ++reg) {
                    const int k = k0 + wave * 16 + lk * 4 + reg;
                    const int m = m0 + i * 16 + lrow;
                    if (k < p.Cin && m < p.M) atomicAdd(p.dwp + ((long)tp * p.Cin + k) * p.M + m, acc[tp][i][reg]);
                }
    }
    if (do_bias) {
        float *rowacc = smem;                           // the image is dead: every wave passed the last barrier
        if (tid < MR) rowacc[tid] = 0.f;
        __syncthreads();
#pragma unroll
        for (int i = 0; i < ND; ++i) {
            const int idx = tid + i * 256;
            if (idx < D4) atomicAdd(rowacc + idx / (CT / 4), bsum[i]);
        }
        __syncthreads();
        if (tid < MR && m0 + tid < p.M) atomicAdd(p.dbias + m0 + tid, rowacc[tid]);
    }
    GLOWTTS_TRACE_POINT_Z(10);
}

// ------------------------------------------------------------------------------------------------------------
// weight packing (+ weight norm) and its backward; row sums for bias gradients
// ------------------------------------------------------------------------------------------------------------
// one workgroup per output channel o:  w[o] = v[o] * g[o] / ||v[o]||  (torch.nn.utils.weight_norm, dim 0) or w = v
//   wp_f[tap][c/16][o][c%16]        = w[o][c][tap]     forward packing   (rows = o, K = c, 16 k per row)
//   wp_b[taps-1-tap][o/16][c][o%16] = w[o][c][tap]     backward-data packing (rows = c, K = o, taps flipped)
// (k positions beyond the channel count must be zero: the caller zero-fills the buffers when a count is not a multiple of 16)
__device__ __forceinline__ void pack_weight_row(const float *__restrict__ v, const float *__restrict__ g,
                                                float *__restrict__ wp_f, float *__restrict__ wp_b,
                                                float *__restrict__ inv_norm, int o, int Cout, int Cin, int taps, float *red) {
    const int n = Cin * taps;
    const float *vo = v + (long)o * n;
    float scale = 1.f;
    if (g != nullptr) {
        float s = 0.f;
        for (int i = threadIdx.x; i < n; i += 256) s += vo[i] * vo[i];
        s = block_sum_256(s, red);
        const float inv = 1.0f / sqrtf(s);
        scale = g[o] * inv;
        if (threadIdx.x == 0 && inv_norm) inv_norm[o] = inv;
    }
    const int Gi = (Cin + 15) / 16, Go = (Cout + 15) / 16;
    for (int i = threadIdx.x; i < n; i += 256) {
        const int c = i / taps, tap = i - c * taps;
        const float w = vo[i] * scale;
        if (wp_f) wp_f[(((long)tap * Gi + (c >> 4)) * Cout + o) * 16 + (c & 15)] = w;
        if (wp_b) wp_b[(((long)(taps - 1 - tap) * Go + (o >> 4)) * Cin + c) * 16 + (o & 15)] = w;
    }
}

__global__ __launch_bounds__(256) void pack_weight_kernel(const float *__restrict__ v, const float *__restrict__ g,
                                                          float *__restrict__ wp_f, float *__restrict__ wp_b,
                                                          float *__restrict__ inv_norm, int Cout, int Cin, int taps) {
    __shared__ float red[4];
    pack_weight_row(v, g, wp_f, wp_b, inv_norm, blockIdx.x, Cout, Cin, taps, red);
}

// (several convolutions per launch: packw.hip, on 16-row tiles)

// backward of the packing: dw[o][c][tap] = dwp[tap][c][o];  plain conv: dweight += dw
// weight norm: dg[o] += sum(dw * v) / n ;  dv += (g / n) * (dw - v * sum(dw * v) / n^2)      (n = ||v[o]||)
__device__ __forceinline__ void unpack_weight_grad_row(const float *__restrict__ dwp, const float *__restrict__ v,
                                                       const float *__restrict__ g, const float *__restrict__ inv_norm,
                                                       float *__restrict__ dv, float *__restrict__ dg, int o, int Cout,
                                                       int Cin, int taps, float *red) {
    const int n = Cin * taps;
    const float *vo = v + (long)o * n;
    if (g == nullptr) {
        for (int i = threadIdx.x; i < n; i += 256) {
            const int c = i / taps, tap = i - c * taps;
            dv[(long)o * n + i] += dwp[((long)tap * Cin + c) * Cout + o];
        }
        return;
    }
    // the row of dW is a gather with stride Cout (one 4-byte element per cache line): read it ONCE and keep it in registers
    // between the dot product and the update (rows up to 12 x 256 elements: every convolution of the model)
    constexpr int KEEP = 12;
    float wreg[KEEP], vreg[KEEP];
    const bool keep = n <= KEEP * 256;
    float dot = 0.f;
    if (keep) {
#pragma unroll
        for (int k = 0; k < KEEP; ++k) {
            const int i = threadIdx.x + 256 * k;
            wreg[k] = 0.f;
            vreg[k] = 0.f;
            if (i < n) {
                const int c = i / taps, tap = i - c * taps;
                wreg[k] = dwp[((long)tap * Cin + c) * Cout + o];
                vreg[k] = vo[i];
            }
        }
#pragma unroll
        for (int k = 0; k < KEEP; ++k) dot += wreg[k] * vreg[k];
    } else {
        for (int i = threadIdx.x; i < n; i += 256) {
            const int c = i / taps, tap = i - c * taps;
            dot += dwp[((long)tap * Cin + c) * Cout + o] * vo[i];
        }
    }
    dot = block_sum_256(dot, red);
    const float inv = inv_norm[o];
    const float gn = g[o] * inv;
    if (threadIdx.x == 0) dg[o] += dot * inv;
    const float proj = dot * inv * inv;
    if (keep) {
#pragma unroll
        for (int k = 0; k < KEEP; ++k) {
            const int i = threadIdx.x + 256 * k;
            if (i < n) dv[(long)o * n + i] += gn * (wreg[k] - vreg[k] * proj);
        }
        return;
    }
    for (int i = threadIdx.x; i < n; i += 256) {
        const int c = i / taps, tap = i - c * taps;
        dv[(long)o * n + i] += gn * (dwp[((long)tap * Cin + c) * Cout + o] - vo[i] * proj);
    }
}

__global__ __launch_bounds__(256) void unpack_weight_grad_kernel(const float *__restrict__ dwp, const float *__restrict__ v,
                                                                 const float *__restrict__ g, const float *__restrict__ inv_norm,
                                                                 float *__restrict__ dv, float *__restrict__ dg, int Cout,
                                                                 int Cin, int taps) {
    __shared__ float red[4];
    unpack_weight_grad_row(dwp, v, g, inv_norm, dv, dg, blockIdx.x, Cout, Cin, taps, red);
}

// out[m] += sum_{b,t} d[b][m][t]   (bias gradients) — grid (M, slabs of utterances)
__global__ __launch_bounds__(256) void rowsum_kernel(const float *__restrict__ d, const float *__restrict__ mask,
                                                     float *__restrict__ out, long d_bs, int B, int M, int T, int nb) {
    __shared__ float red[4];
    const int m = blockIdx.x;
    const int b0 = blockIdx.y * nb, b1 = min(B, b0 + nb);
    float s = 0.f;
    for (int b = b0; b < b1; ++b) {
        const float *row = d + (long)b * d_bs + (long)m * T;
        for (int t = threadIdx.x; t < T; t += 256) s += mask ? row[t] * mask[(long)b * T + t] : row[t];
    }
    s = block_sum_256(s, red);
    if (threadIdx.x == 0) atomicAdd(out + m, s);
}

static inline int pitch16(int n) {          // smallest pitch >= n with pitch % 32 == 16
    int p = (n + 15) / 32 * 32 + 16;
    if (p - 32 >= n) p -= 32;
    return p;
}
static inline int pitch2(int n) {           // smallest pitch >= n with pitch % 32 == 2
    int p = (n - 2 + 31) / 32 * 32 + 2;
    return p < n ? p + 32 : p;
}

template <int RTW, int NCT, int EPI>
static int launch_convgemm(ConvGemmParams &p, hipStream_t s) {
    constexpr int WGR = 64 * RTW, NT = 16 * NCT;
    p.xp_pitch = pitch16(NT + (p.taps - 1) * p.dil);
    const size_t lds = ((size_t)p.taps * 16 * (WGR + 16) + (size_t)16 * p.xp_pitch) * sizeof(float);
    GLOWTTS_CHECK_ARG(lds <= 160 * 1024, "glowtts_conv: %d taps x dilation %d needs %zu B of LDS", p.taps, p.dil, lds);
    static LdsLimit attr_max_e;   // per device: raised only when a launch needs more than any earlier one
    if (int rc_ = attr_max_e.ensure(reinterpret_cast<const void *>(&convgemm_kernel<RTW, NCT, EPI>), lds, "glowtts_conv")) return rc_;
    const int ntile_t = (p.T + NT - 1) / NT;
    const int rows = (EPI == EPI_GATE) ? p.H : p.M;
    const int per = (EPI == EPI_GATE) ? 64 : WGR;
    dim3 grid(ntile_t * p.B, (rows + per - 1) / per);
    hipLaunchKernelGGL((convgemm_kernel<RTW, NCT, EPI>), grid, dim3(256), lds, s, p);
    GLOWTTS_LAUNCH_CHECK("glowtts_conv");
}

template <int RTW, int NCT, int EPI, int TAPS>
static int launch_convgemm_wd(ConvGemmParams &p, hipStream_t s) {
    constexpr int WGR = 64 * RTW, NT = 16 * NCT;
    constexpr size_t lds_pipe = ((size_t)6 * (NT + 16) * 20 + (NT + 16)) * sizeof(float);
    constexpr size_t lds_epi = ((size_t)WGR * (NT + 4) + (EPI == EPI_GATEBWD ? 2 * WGR : 0)) * sizeof(float);
    constexpr size_t lds = lds_pipe > lds_epi ? lds_pipe : lds_epi;
    static_assert(lds <= 80 * 1024, "two workgroups per CU");
    p.vec_epilogue = aligned16(p.y0) && aligned16(p.y1) && aligned16(p.r0) && aligned16(p.r1) && aligned16(p.mask) &&
                     aligned16(p.drop) && aligned16(p.gate_pos) && (p.y_bs % 4 == 0) && (p.r_bs % 4 == 0) &&
                     (EPI != EPI_GATE || p.H % 4 == 0);
    static LdsLimit attr_max_e;   // per device: raised only when a launch needs more than any earlier one
    if (int rc_ = attr_max_e.ensure(reinterpret_cast<const void *>(&convgemm_wd_kernel<RTW, NCT, EPI, TAPS>), lds, "glowtts_conv")) return rc_;
    const int ntile_t = (p.T + NT - 1) / NT;
    const int rows = (EPI == EPI_GATE) ? p.H : p.M;
    const int per = (EPI == EPI_GATE) ? 64 : WGR;
    dim3 grid(ntile_t * p.B, (rows + per - 1) / per);
    hipLaunchKernelGGL((convgemm_wd_kernel<RTW, NCT, EPI, TAPS>), grid, dim3(256), lds, s, p);
    GLOWTTS_LAUNCH_CHECK("glowtts_conv");
}

template <int RTW, int NCT, int EPI>
static int dispatch_taps(ConvGemmParams &p, hipStream_t s, bool pipe_ok) {
    if (pipe_ok) {
        if (p.taps == 5) return launch_convgemm_wd<RTW, NCT, EPI, 5>(p, s);
        if (p.taps == 3) return launch_convgemm_wd<RTW, NCT, EPI, 3>(p, s);
        if (p.taps == 1) return launch_convgemm_wd<RTW, NCT, EPI, 1>(p, s);
    }
    return launch_convgemm<RTW, NCT, EPI>(p, s);
}

template <int EPI>
static int dispatch_convgemm(ConvGemmParams &p, hipStream_t s) {
    // 80-frame tiles when T divides evenly (e.g. T' = 400), else 64; 128-row workgroups unless M is small
    const bool n5 = (p.T % 80 == 0) || (p.T % 64 != 0 && ((p.T + 79) / 80) * 80 < ((p.T + 63) / 64) * 64);
    const bool big = (EPI == EPI_GATE) || (p.M % 128 == 0) || p.M > 192;
    const bool pipe_ok = (p.T % 4 == 0) && aligned16(p.x) && (p.x_bs % 4 == 0) &&
                         (!p.mask_in || aligned16(p.mask)) && ((p.taps - 1) * p.dil <= 12) &&
                         (EPI != EPI_GATE || p.H % 4 == 0);
    if (p.xb) return conv_bf16_dispatch(p, EPI, big, n5, pipe_ok, s);                     // bf16 tensors in HBM
    // Small problems (the text encoder: T = 160) give only ~190 workgroups with 80-frame tiles — less than one per CU.
    // 32-frame tiles fill the chip (480+ workgroups): -15..20 % on the encoder's 3-tap and 1-tap convs.  At the decoder's
    // T' = 400 (480 workgroups already) the smaller tiles only lose operand reuse, so the switch is on the grid size.
    bool use32 = false;
    if constexpr (EPI == EPI_PLAIN) if (pipe_ok) {
        const int rows_per = big ? 128 : 64;
        const long wg5 = (long)((p.T + 79) / 80) * p.B * ((p.M + rows_per - 1) / rows_per);
        const int t32 = ((p.T + 31) / 32) * 32;
        // (threshold 440 -> 380 in round 4: the FFN's 768 <- 192-channel 3-tap convolutions at T = 160 have 384 tiles of 128 rows
        // x 80 frames, which the bf16-plane kernel runs in 80-frame form — its 32-frame 128-row form does not exist, and the native
        // 32-frame kernel that served them took 49 us: 15.00 -> 14.81 ms per step, tools/ab_flags.py envs=GLOWTTS_CONV32_WG:440,..:380;
        // 190 and 0 measured 14.87 / 14.96 against 14.90)
        use32 = wg5 < 380 && t32 * 10 <= p.T * 11;
        // plain 1x1 convolutions (the coupling's start / end convs and the end conv's backward-data: 12 800 columns, 480 tiles of
        // 80 frames) take 32-frame tiles as well: four consecutive A/B pairs 15.04 -> 15.00 ms per step (GLOWTTS_CONV32_1X1=0 / 1)
        if (p.taps == 1 && t32 * 10 <= p.T * 11 && knob(K_CONV32_1X1) == 1) use32 = true;
    }
    if (int rc = conv_split_dispatch(p, EPI, big, use32 ? 2 : (n5 ? 5 : 4), pipe_ok, s); rc >= 0) return rc;   // bf16-plane arithmetic
    if (use32) {
        if (big) return dispatch_taps<2, 2, EPI>(p, s, pipe_ok);
        return dispatch_taps<1, 2, EPI>(p, s, pipe_ok);
    }
    if (EPI == EPI_GATE) return n5 ? dispatch_taps<2, 5, EPI>(p, s, pipe_ok) : dispatch_taps<2, 4, EPI>(p, s, pipe_ok);
    if (big) return n5 ? dispatch_taps<2, 5, EPI>(p, s, pipe_ok) : dispatch_taps<2, 4, EPI>(p, s, pipe_ok);
    return n5 ? dispatch_taps<1, 5, EPI>(p, s, pipe_ok) : dispatch_taps<1, 4, EPI>(p, s, pipe_ok);
}

template <int TAPS, int NGRP, int MT>
static int launch_wrw_fp_mt(ConvWrwParams &p, hipStream_t s) {
    constexpr int CT = 16 * NGRP, MR = 16 * MT;
    constexpr size_t lds = ((size_t)64 * cpitch4(CT + 16) + (size_t)MR * cpitch4(CT) + (CT + 16) + CT) * sizeof(float);
    static_assert(lds <= 80 * 1024, "two workgroups per CU");
    static LdsLimit attr_max_e;   // per device: raised only when a launch needs more than any earlier one
    if (int rc_ = attr_max_e.ensure(reinterpret_cast<const void *>(&convwrw_fp_kernel<TAPS, NGRP, MT>), lds, "glowtts_conv_wrw")) return rc_;
    // all workgroups resident at once (2 per CU: 512 slots); p.nb = chunks of CT frames per workgroup
    const int tiles = ((p.Cin + 63) / 64) * ((p.M + MR - 1) / MR);
    const int total = p.B * ((p.T + CT - 1) / CT);
    int splits = (MT <= 2 ? 768 : 512) / tiles;
    if (splits > total) splits = total;
    if (splits < 1) splits = 1;
    p.nb = (total + splits - 1) / splits;
    dim3 grid(tiles, 1, (total + p.nb - 1) / p.nb);
    hipLaunchKernelGGL((convwrw_fp_kernel<TAPS, NGRP, MT>), grid, dim3(256), lds, s, p);
    GLOWTTS_LAUNCH_CHECK("glowtts_conv_wrw");
}

template <int TAPS, int NGRP>
static int launch_wrw_fp(ConvWrwParams &p, hipStream_t s) {
    // 5 taps: 64 k x 32 m tiles — 40 accumulator registers per lane instead of 80, so THREE workgroups fit a CU (168
    // VGPRs, 36 KB LDS) and hide each other's prologue / atomics epilogue: 107 -> 96 us at config 2.  With 3 taps or 1 the
    // 64 x 64 tile already leaves room for three, and the smaller tile only costs operand reuse.
    if constexpr (TAPS == 5)
        if (p.M % 32 == 0 && (!p.d2 || p.d_split % 32 == 0)) return launch_wrw_fp_mt<TAPS, NGRP, 2>(p, s);
    return launch_wrw_fp_mt<TAPS, NGRP, 4>(p, s);
}

template <int TAPS, int CT>
static int launch_wrw_pipe(ConvWrwParams &p, hipStream_t s) {
    constexpr size_t lds = 2 * ((size_t)64 * cpitch2(CT + 16) + (size_t)64 * cpitch2(CT)) * sizeof(float);
    static LdsLimit attr_max_e;   // per device: raised only when a launch needs more than any earlier one
    if (int rc_ = attr_max_e.ensure(reinterpret_cast<const void *>(&convwrw_pipe_kernel<TAPS, CT>), lds, "glowtts_conv_wrw")) return rc_;
    // The contraction runs over B * T frames = `total` chunks of CT; it is split so that ALL workgroups are resident at
    // once (2 per CU by LDS, 512 slots): a grid of 576 on 512 slots costs two full rounds.  p.nb = chunks per workgroup.
    const int tiles = ((p.Cin + 63) / 64) * ((p.M + 63) / 64);
    const int total = p.B * ((p.T + CT - 1) / CT);
    int splits = 512 / tiles;
    if (splits > total) splits = total;
    if (splits < 1) splits = 1;
    p.nb = (total + splits - 1) / splits;
    dim3 grid(tiles, 1, (total + p.nb - 1) / p.nb);
    hipLaunchKernelGGL((convwrw_pipe_kernel<TAPS, CT>), grid, dim3(256), lds, s, p);
    GLOWTTS_LAUNCH_CHECK("glowtts_conv_wrw");
}

}  // namespace glowtts

using namespace glowtts;

extern "C" int glowtts_rowsum(const float *d, long d_bs, const float *mask, float *out, int B, int M, int T,
                              glowtts_stream_t stream);

static int check_conv_common(const char *name, const void *x, const void *wp, int B, int Cin, int M, int T, int taps,
                             int dil, int pad) {
    GLOWTTS_CHECK_ARG(x && wp, "%s: null pointer", name);
    GLOWTTS_CHECK_ARG(B >= 0 && Cin > 0 && M > 0 && T >= 0 && taps >= 1 && dil >= 1 && pad >= 0,
                      "%s: bad shape B=%d Cin=%d M=%d T=%d taps=%d dil=%d pad=%d", name, B, Cin, M, T, taps, dil, pad);
    GLOWTTS_CHECK_ARG(aligned16(wp), "%s: packed weights must be 16-byte aligned", name);
    return 0;
}

// `_io` forms: the same operators on bf16 tensors (io_x: x / second source; io_y: every tensor the epilogue reads or writes);
// packed weights as always plus ONE bf16 plane bound to the calling thread (glowtts_conv_bind_planes_ns).  io 0 = fp32.
extern "C" int glowtts_conv_fwd_io(const void *x, long x_bs, const float *wp, const float *bias, const float *mask,
                                   const void *addend, long addend_bs, void *y, long y_bs, int B, int Cin, int M, int T,
                                   int taps, int dil, int pad, int mask_in, int mask_out, int mask_add, int io_x, int io_y,
                                   glowtts_stream_t stream) {
    if (int rc = check_conv_common("glowtts_conv_fwd", x, wp, B, Cin, M, T, taps, dil, pad)) return rc;
    GLOWTTS_CHECK_ARG(y, "glowtts_conv_fwd: null output");
    GLOWTTS_CHECK_ARG(!(mask_in || mask_out || mask_add) || mask, "glowtts_conv_fwd: mask flag without mask");
    GLOWTTS_CHECK_ARG(io_x || !io_y, "glowtts_conv_fwd: bf16 results need bf16 operands (io_x = 1)");
    if ((long)B * T == 0) return 0;
    ConvGemmParams p{};
    p.x = static_cast<const float *>(x); p.wp = wp; p.bias = bias; p.mask = mask; p.r0 = static_cast<const float *>(addend);
    p.y0 = static_cast<float *>(y); p.xb = io_x; p.yb = io_y;
    p.x_bs = x_bs; p.y_bs = y_bs; p.B = B; p.Cin = Cin; p.M = M; p.T = T; p.taps = taps; p.dil = dil; p.pad = pad;
    p.mask_in = mask_in; p.mask_out = mask_out; p.mask_add = mask_add; p.r_bs = addend_bs;
    return addend ? dispatch_convgemm<EPI_ADD>(p, (hipStream_t)stream) : dispatch_convgemm<EPI_PLAIN>(p, (hipStream_t)stream);
}

// conv_fwd with the elementwise neighbours of the text encoder's convolutions folded into the epilogue (fp32 tensors):
//   forward : y = dropout(relu(conv(x) [+ addend] [* mask]))     relu, (drop keep bytes (B, M, T), drop_scale) optional
//   backward-data of such a conv's CONSUMER: gate_pos = the forward's output g (B, M, T): the conv result is multiplied by
//   gate_scale where g > 0 and zeroed elsewhere (ReLU' and the dropout in one test: g = 0 wherever either cut), then
//   [+ addend] [* mask].   (reference: attentions.py:373-381 FFN, layers.py:73-80 prenet)
extern "C" int glowtts_conv_fwd_act(const float *x, long x_bs, const float *wp, const float *bias, const float *mask,
                                    const float *addend, long addend_bs, float *y, long y_bs, int B, int Cin, int M, int T,
                                    int taps, int dil, int pad, int mask_in, int mask_out, int mask_add, int relu,
                                    const unsigned char *drop, float drop_scale, const float *gate_pos, float gate_scale,
                                    glowtts_stream_t stream) {
    if (int rc = check_conv_common("glowtts_conv_fwd_act", x, wp, B, Cin, M, T, taps, dil, pad)) return rc;
    GLOWTTS_CHECK_ARG(y, "glowtts_conv_fwd_act: null output");
    GLOWTTS_CHECK_ARG(!(mask_in || mask_out || mask_add) || mask, "glowtts_conv_fwd_act: mask flag without mask");
    if ((long)B * T == 0) return 0;
    ConvGemmParams p{};
    p.x = x; p.wp = wp; p.bias = bias; p.mask = mask; p.r0 = addend; p.y0 = y;
    p.x_bs = x_bs; p.y_bs = y_bs; p.B = B; p.Cin = Cin; p.M = M; p.T = T; p.taps = taps; p.dil = dil; p.pad = pad;
    p.mask_in = mask_in; p.mask_out = mask_out; p.mask_add = mask_add; p.r_bs = addend_bs;
    p.relu = relu; p.drop = drop; p.drop_scale = drop_scale; p.gate_pos = gate_pos; p.gate_scale = gate_scale;
    return addend ? dispatch_convgemm<EPI_ADD>(p, (hipStream_t)stream) : dispatch_convgemm<EPI_PLAIN>(p, (hipStream_t)stream);
}

extern "C" int glowtts_conv_fwd(const float *x, long x_bs, const float *wp, const float *bias, const float *mask,
                                const float *addend, long addend_bs, float *y, long y_bs, int B, int Cin, int M, int T,
                                int taps, int dil, int pad, int mask_in, int mask_out, int mask_add,
                                glowtts_stream_t stream) {
    return glowtts_conv_fwd_io(x, x_bs, wp, bias, mask, addend, addend_bs, y, y_bs, B, Cin, M, T, taps, dil, pad, mask_in,
                               mask_out, mask_add, 0, 0, stream);
}

extern "C" int glowtts_conv_gate_fwd_io(const void *x, const float *wp, const float *bias, const float *cond,
                                        const unsigned char *drop, float drop_scale, void *acts, void *ts, int B, int H,
                                        int T, int taps, int dil, int pad, int io, glowtts_stream_t stream) {
    if (int rc = check_conv_common("glowtts_conv_gate_fwd", x, wp, B, H, 2 * H, T, taps, dil, pad)) return rc;
    GLOWTTS_CHECK_ARG(acts, "glowtts_conv_gate_fwd: null output");
    GLOWTTS_CHECK_ARG(H % 4 == 0, "glowtts_conv_gate_fwd: hidden width %d must be a multiple of 4", H);
    if ((long)B * T == 0) return 0;
    ConvGemmParams p{};
    p.x = static_cast<const float *>(x); p.wp = wp; p.bias = bias; p.cond = cond; p.drop = drop; p.drop_scale = drop_scale;
    p.y0 = static_cast<float *>(acts); p.y1 = static_cast<float *>(ts); p.xb = io; p.yb = io;
    p.x_bs = (long)H * T; p.B = B; p.Cin = H; p.M = 2 * H; p.H = H; p.T = T; p.taps = taps; p.dil = dil; p.pad = pad;
    if (io == 0) {                                   // Winograd form where U planes are bound and the switch is on (convwino.hip)
        const int rc = conv_wino_gate_dispatch(p, (hipStream_t)stream);
        if (rc >= 0) return rc;
    }
    return dispatch_convgemm<EPI_GATE>(p, (hipStream_t)stream);
}

extern "C" int glowtts_conv_gate_fwd(const float *x, const float *wp, const float *bias, const float *cond,
                                     const unsigned char *drop, float drop_scale, float *acts, float *ts, int B, int H,
                                     int T, int taps, int dil, int pad, glowtts_stream_t stream) {
    return glowtts_conv_gate_fwd_io(x, wp, bias, cond, drop, drop_scale, acts, ts, B, H, T, taps, dil, pad, 0, stream);
}

extern "C" int glowtts_conv_res_skip_fwd_io(const void *acts, const float *wp, const float *bias, const float *mask,
                                            const void *x_in, const void *skip_in, void *x_out, void *skip_out, int B,
                                            int H, int T, int last, int io, glowtts_stream_t stream) {
    const int M = last ? H : 2 * H;
    if (int rc = check_conv_common("glowtts_conv_res_skip_fwd", acts, wp, B, H, M, T, 1, 1, 0)) return rc;
    GLOWTTS_CHECK_ARG(mask && skip_out && (last || (x_in && x_out)), "glowtts_conv_res_skip_fwd: null pointer");
    if ((long)B * T == 0) return 0;
    ConvGemmParams p{};
    p.x = static_cast<const float *>(acts); p.wp = wp; p.bias = bias; p.mask = mask; p.r0 = static_cast<const float *>(x_in);
    p.r1 = static_cast<const float *>(skip_in); p.y0 = static_cast<float *>(x_out); p.y1 = static_cast<float *>(skip_out);
    p.xb = io; p.yb = io;
    p.x_bs = (long)H * T; p.B = B; p.Cin = H; p.M = M; p.H = H; p.T = T; p.taps = 1; p.dil = 1; p.pad = 0;
    return last ? dispatch_convgemm<EPI_RESSKIP_LAST>(p, (hipStream_t)stream)
                : dispatch_convgemm<EPI_RESSKIP>(p, (hipStream_t)stream);
}

extern "C" int glowtts_conv_res_skip_fwd(const float *acts, const float *wp, const float *bias, const float *mask,
                                         const float *x_in, const float *skip_in, float *x_out, float *skip_out, int B,
                                         int H, int T, int last, glowtts_stream_t stream) {
    return glowtts_conv_res_skip_fwd_io(acts, wp, bias, mask, x_in, skip_in, x_out, skip_out, B, H, T, last, 0, stream);
}

extern "C" int glowtts_conv_gate_bwd_io(const void *d_rs, const void *d_rs2, const float *wp_b, const void *ts,
                                        const unsigned char *drop, float drop_scale, void *d_pre, float *dcond, int B,
                                        int M_rs, int H, int T, int io, glowtts_stream_t stream) {
    if (int rc = check_conv_common("glowtts_conv_gate_bwd", d_rs, wp_b, B, M_rs, H, T, 1, 1, 0)) return rc;
    GLOWTTS_CHECK_ARG(ts && d_pre, "glowtts_conv_gate_bwd: null pointer");
    GLOWTTS_CHECK_ARG(H % 4 == 0, "glowtts_conv_gate_bwd: hidden width %d must be a multiple of 4", H);
    if ((long)B * T == 0) return 0;
    ConvGemmParams p{};
    p.x = static_cast<const float *>(d_rs); p.wp = wp_b; p.r0 = static_cast<const float *>(ts); p.drop = drop;
    p.drop_scale = drop_scale; p.y0 = static_cast<float *>(d_pre); p.xb = io; p.yb = io; p.dcond = dcond;
    p.x_bs = (long)M_rs * T; p.B = B; p.Cin = M_rs; p.M = H; p.H = H; p.T = T; p.taps = 1; p.dil = 1; p.pad = 0;
    if (d_rs2) {    // d_rs = rows [0, H) as (B,H,T), d_rs2 = rows [H, 2H) as (B,H,T): never concatenated in memory
        GLOWTTS_CHECK_ARG(M_rs == 2 * H && H % 96 == 0 && T % 4 == 0 && aligned16(d_rs) && aligned16(d_rs2),
                          "glowtts_conv_gate_bwd: two-source input needs M_rs == 2H, H %% 96 == 0, T %% 4 == 0, 16-byte rows");
        p.x_bs = (long)H * T; p.x2 = static_cast<const float *>(d_rs2); p.x2_bs = (long)H * T; p.x_split = H;
    }
    return dispatch_convgemm<EPI_GATEBWD>(p, (hipStream_t)stream);
}

extern "C" int glowtts_conv_gate_bwd(const float *d_rs, const float *d_rs2, const float *wp_b, const float *ts,
                                     const unsigned char *drop, float drop_scale, float *d_pre, int B, int M_rs, int H,
                                     int T, glowtts_stream_t stream) {
    return glowtts_conv_gate_bwd_io(d_rs, d_rs2, wp_b, ts, drop, drop_scale, d_pre, nullptr, B, M_rs, H, T, 0, stream);
}

extern "C" int glowtts_conv_wrw2(const float *x, long x_bs, const float *d, long d_bs, const float *d2, long d2_bs,
                                 int d_split, float *dwp, float *dbias, int B, int Cin, int M, int T, int taps, int dil,
                                 int pad, glowtts_stream_t stream) {
    GLOWTTS_CHECK_ARG(x && d && d2 && dwp, "glowtts_conv_wrw2: null pointer");
    GLOWTTS_CHECK_ARG(B >= 0 && Cin > 0 && M > 0 && T >= 0 && d_split > 0 && d_split < M, "glowtts_conv_wrw2: bad shape");
    GLOWTTS_CHECK_ARG((taps == 1 || taps == 3 || taps == 5) && dil == 1 && pad == (taps - 1) / 2 && T % 4 == 0 &&
                          d_split % 64 == 0 && aligned16(x) && aligned16(d) && aligned16(d2) && x_bs % 4 == 0 &&
                          d_bs % 4 == 0 && d2_bs % 4 == 0,
                      "glowtts_conv_wrw2: the two-source weight gradient needs taps in {1,3,5}, dilation 1, 'same' padding, "
                      "T %% 4 == 0, d_split %% 64 == 0 and 16-byte aligned rows");
    if ((long)B * T == 0) return 0;
    ConvWrwParams p{};
    p.x = x; p.d = d; p.d2 = d2; p.d2_bs = d2_bs; p.d_split = d_split; p.dwp = dwp; p.dbias = dbias; p.x_bs = x_bs; p.d_bs = d_bs;
    p.B = B; p.Cin = Cin; p.M = M; p.T = T; p.taps = taps; p.dil = dil; p.pad = pad;
    hipStream_t s = (hipStream_t)stream;
    if (int rc = conv_wrw_split_dispatch(p, s); rc >= 0) return rc;        // opt-in bf16-plane arithmetic
    const bool n5 = (T % 80 == 0) || ((T + 79) / 80) * 80 <= ((T + 63) / 64) * 64;
    if (taps == 5) return n5 ? launch_wrw_fp<5, 5>(p, s) : launch_wrw_fp<5, 4>(p, s);
    if (taps == 3) return n5 ? launch_wrw_fp<3, 5>(p, s) : launch_wrw_fp<3, 4>(p, s);
    return n5 ? launch_wrw_fp<1, 5>(p, s) : launch_wrw_fp<1, 4>(p, s);
}

extern "C" int glowtts_conv_wrw(const float *x, long x_bs, const float *d, long d_bs, const float *mask,
                                const float *mask_x, float *dwp, float *dbias, int B, int Cin, int M, int T, int taps,
                                int dil, int pad, glowtts_stream_t stream) {
    GLOWTTS_CHECK_ARG(x && d && dwp, "glowtts_conv_wrw: null pointer");
    GLOWTTS_CHECK_ARG(B >= 0 && Cin > 0 && M > 0 && T >= 0 && taps >= 1 && dil >= 1 && pad >= 0, "glowtts_conv_wrw: bad shape");
    if ((long)B * T == 0) return 0;
    ConvWrwParams p{};
    p.x = x; p.d = d; p.dwp = dwp; p.dbias = dbias; p.mask = mask; p.mask_x = mask_x; p.x_bs = x_bs; p.d_bs = d_bs;
    p.B = B; p.Cin = Cin; p.M = M; p.T = T; p.taps = taps; p.dil = dil; p.pad = pad;
    const bool pipe_ok = (T % 4 == 0) && aligned16(x) && aligned16(d) && (x_bs % 4 == 0) && (d_bs % 4 == 0) &&
                         (!mask || aligned16(mask)) && (!mask_x || aligned16(mask_x)) && ((taps - 1) * dil <= 12);
    if (pipe_ok && (taps == 1 || taps == 3 || taps == 5)) {
        hipStream_t s = (hipStream_t)stream;
        if (dil == 1 && pad == (taps - 1) / 2) {         // frame-packed kernel: 80-frame chunks, or 64 when that wastes less
            if (int rc = conv_wrw_split_dispatch(p, s); rc >= 0) return rc;        // opt-in bf16-plane arithmetic
            const bool n5 = (T % 80 == 0) || ((T + 79) / 80) * 80 <= ((T + 63) / 64) * 64;
            if (taps == 5) return n5 ? launch_wrw_fp<5, 5>(p, s) : launch_wrw_fp<5, 4>(p, s);
            if (taps == 3) return n5 ? launch_wrw_fp<3, 5>(p, s) : launch_wrw_fp<3, 4>(p, s);
            return n5 ? launch_wrw_fp<1, 5>(p, s) : launch_wrw_fp<1, 4>(p, s);
        }
        const bool c40 = (T % 40 == 0);
        if (taps == 5) return c40 ? launch_wrw_pipe<5, 40>(p, s) : launch_wrw_pipe<5, 32>(p, s);
        if (taps == 3) return c40 ? launch_wrw_pipe<3, 40>(p, s) : launch_wrw_pipe<3, 32>(p, s);
        return c40 ? launch_wrw_pipe<1, 40>(p, s) : launch_wrw_pipe<1, 32>(p, s);
    }
    const int tiles = ((Cin + 63) / 64) * ((M + 127) / 128) * taps;
    int splits = (768 + tiles - 1) / tiles;          // aim at ~3 workgroups per CU
    if (splits > B) splits = B;
    if (splits < 1) splits = 1;
    p.nb = (B + splits - 1) / splits;
    p.xs_pitch = pitch2(64);
    p.ds_pitch = pitch2(64);
    const size_t lds = ((size_t)64 * p.xs_pitch + (size_t)128 * p.ds_pitch) * sizeof(float);
    dim3 grid(((Cin + 63) / 64) * ((M + 127) / 128), taps, (B + p.nb - 1) / p.nb);
    hipLaunchKernelGGL(convwrw_kernel, grid, dim3(256), lds, (hipStream_t)stream, p);
    if (dbias) {
        hipError_t e0 = hipGetLastError();
        if (e0 != hipSuccess) { set_error("glowtts_conv_wrw: launch failed: %s", hipGetErrorString(e0)); return (int)e0; }
        return glowtts_rowsum(d, d_bs, mask, dbias, B, M, T, stream);
    }
    GLOWTTS_LAUNCH_CHECK("glowtts_conv_wrw");
}

// Several weight gradients of ONE shape in one launch (dilation 1, 'same' padding): x[q] (B, Cin, T), d[q] (B, M, T) or, with d2,
// rows [d_split, M) from d2[q]; dwp[q] / dbias[q] accumulated as in glowtts_conv_wrw; mask / mask_x (single-source form only) are
// shared by the problems.  The arrays are HOST arrays of device pointers.  Without a kernel for the batch (native fp32 arithmetic,
// other shapes) the problems are launched one by one.
extern "C" int glowtts_conv_wrw_batch(int n, const float *const *x, long x_bs, const float *const *d, long d_bs,
                                      const float *const *d2, long d2_bs, int d_split, const float *mask, const float *mask_x,
                                      float *const *dwp, float *const *dbias, int B, int Cin, int M, int T, int taps, int dil,
                                      int pad, glowtts_stream_t stream) {
    GLOWTTS_CHECK_ARG(n >= 1 && x && d && dwp, "glowtts_conv_wrw_batch: null pointer");
    GLOWTTS_CHECK_ARG(!d2 || (!mask && !mask_x), "glowtts_conv_wrw_batch: the two-source form takes no masks");
    GLOWTTS_CHECK_ARG(B >= 0 && Cin > 0 && M > 0 && T >= 0 && taps >= 1 && dil >= 1 && pad >= 0, "glowtts_conv_wrw_batch: bad shape");
    GLOWTTS_CHECK_ARG(!d2 || (d_split > 0 && d_split < M), "glowtts_conv_wrw_batch: bad d_split");
    if ((long)B * T == 0) return 0;
    hipStream_t s = (hipStream_t)stream;
    for (int q0 = 0; q0 < n; q0 += ConvWrwParams::kMaxBatch) {
        const int nb = n - q0 < ConvWrwParams::kMaxBatch ? n - q0 : ConvWrwParams::kMaxBatch;
        bool ok = nb > 1 && dil == 1 && pad == (taps - 1) / 2 && (taps == 1 || taps == 3 || taps == 5) && T % 4 == 0 &&
                  x_bs % 4 == 0 && d_bs % 4 == 0 && (!d2 || (d2_bs % 4 == 0 && d_split % 64 == 0)) &&
                  (!mask || aligned16(mask)) && (!mask_x || aligned16(mask_x));
        ConvWrwParams p{};
        for (int q = 0; q < nb && ok; ++q) {
            const float *dq2 = d2 ? d2[q0 + q] : nullptr;
            GLOWTTS_CHECK_ARG(x[q0 + q] && d[q0 + q] && dwp[q0 + q] && (!d2 || dq2), "glowtts_conv_wrw_batch: null pointer in problem %d", q0 + q);
            ok = aligned16(x[q0 + q]) && aligned16(d[q0 + q]) && (!dq2 || aligned16(dq2));
            p.bx[q] = x[q0 + q]; p.bd[q] = d[q0 + q]; p.bd2[q] = dq2; p.bdwp[q] = dwp[q0 + q];
            p.bdbias[q] = dbias ? dbias[q0 + q] : nullptr;
        }
        if (ok) {
            p.x = p.bx[0]; p.d = p.bd[0]; p.d2 = p.bd2[0]; p.dwp = p.bdwp[0]; p.dbias = p.bdbias[0];
            p.d2_bs = d2_bs; p.d_split = d_split; p.x_bs = x_bs; p.d_bs = d_bs; p.mask = mask; p.mask_x = mask_x;
            p.B = B; p.Cin = Cin; p.M = M; p.T = T; p.taps = taps; p.dil = dil; p.pad = pad;
            p.nbatch = nb;
            const int rc = conv_wrw_split_dispatch(p, s);
            if (rc > 0) return rc;
            if (rc == 0) continue;
        }
        for (int q = q0; q < q0 + nb; ++q) {               // one by one
            const int rc = d2 ? glowtts_conv_wrw2(x[q], x_bs, d[q], d_bs, d2[q], d2_bs, d_split, dwp[q], dbias ? dbias[q] : nullptr, B, Cin,
                                                  M, T, taps, dil, pad, stream)
                              : glowtts_conv_wrw(x[q], x_bs, d[q], d_bs, mask, mask_x, dwp[q], dbias ? dbias[q] : nullptr, B, Cin, M,
                                                 T, taps, dil, pad, stream);
            if (rc != 0) return rc;
        }
    }
    return 0;
}

extern "C" int glowtts_pack_weight(const float *v, const float *g, float *wp_f, float *wp_b, float *inv_norm, int Cout,
                                   int Cin, int taps, glowtts_stream_t stream) {
    GLOWTTS_CHECK_ARG(v && (wp_f || wp_b), "glowtts_pack_weight: null pointer");
    GLOWTTS_CHECK_ARG(!g || inv_norm, "glowtts_pack_weight: weight norm needs inv_norm");
    GLOWTTS_CHECK_ARG(Cout > 0 && Cin > 0 && taps > 0, "glowtts_pack_weight: bad shape");
    hipLaunchKernelGGL(pack_weight_kernel, dim3(Cout), dim3(256), 0, (hipStream_t)stream, v, g, wp_f, wp_b, inv_norm, Cout, Cin, taps);
    GLOWTTS_LAUNCH_CHECK("glowtts_pack_weight");
}

extern "C" int glowtts_unpack_weight_grad(const float *dwp, const float *v, const float *g, const float *inv_norm,
                                          float *dv, float *dg, int Cout, int Cin, int taps, glowtts_stream_t stream) {
    GLOWTTS_CHECK_ARG(dwp && v && dv, "glowtts_unpack_weight_grad: null pointer");
    GLOWTTS_CHECK_ARG(!g || (inv_norm && dg), "glowtts_unpack_weight_grad: weight norm needs inv_norm and dg");
    GLOWTTS_CHECK_ARG(Cout > 0 && Cin > 0 && taps > 0, "glowtts_unpack_weight_grad: bad shape");
    hipLaunchKernelGGL(unpack_weight_grad_kernel, dim3(Cout), dim3(256), 0, (hipStream_t)stream, dwp, v, g, inv_norm, dv, dg, Cout, Cin, taps);
    GLOWTTS_LAUNCH_CHECK("glowtts_unpack_weight_grad");
}

extern "C" int glowtts_rowsum(const float *d, long d_bs, const float *mask, float *out, int B, int M, int T,
                              glowtts_stream_t stream) {
    GLOWTTS_CHECK_ARG(d && out, "glowtts_rowsum: null pointer");
    GLOWTTS_CHECK_ARG(B >= 0 && M > 0 && T >= 0, "glowtts_rowsum: bad shape");
    if ((long)B * T == 0) return 0;
    int slabs = (1024 + M - 1) / M;
    if (slabs > B) slabs = B;
    if (slabs < 1) slabs = 1;
    const int nb = (B + slabs - 1) / slabs;
    hipLaunchKernelGGL(rowsum_kernel, dim3(M, (B + nb - 1) / nb), dim3(256), 0, (hipStream_t)stream, d, mask, out, d_bs, B, M, T, nb);
    GLOWTTS_LAUNCH_CHECK("glowtts_rowsum");
}

#ifdef GLOWTTS_TRACE
extern "C" int glowtts_debug_trace_read(unsigned long long *host, int n_words, int clear) {
    hipError_t e = hipMemcpyFromSymbol(host, HIP_SYMBOL(glowtts::g_trace), (size_t)n_words * 8);
    if (e != hipSuccess) return (int)e;
    if (clear) {
        static unsigned long long zeros[8192 * 16];
        e = hipMemcpyToSymbol(HIP_SYMBOL(glowtts::g_trace), zeros, sizeof(zeros));
    }
    return (int)e;
}
#endif
