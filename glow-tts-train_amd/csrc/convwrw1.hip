// convwrw1.hip — the weight gradients of 1x1 convolutions, several problems of DIFFERENT shapes in one launch (round 4).
//
//   dW[k][m] = sum_{b, t} x[b][k][t] * d[b][m][t]          (reference: autograd of the 1x1 convs of layers.py:155-156 — the WN
//   stack's res/skip convs — and attentions.py:97-113 — the coupling's start / end convs; the encoder's q / k / v / o convs)
//
// What bounded the frame-packed kernel (convwrw_split_kernel<3, 1, ...>, 64 x 64 tiles, 29 us per 384 x 192 problem at 12 800
// frames, matrix pipe 13-16 % busy): every staged fp32 value is split into three bf16 planes (5.5 vector instructions) and
// then used by only 64 output channels — ~3.3 vector instructions per MFMA on the issue port the MFMAs share; and the small
// problems of a flow block (start / end conv, last layer) were launches of their own, 22 us each for < 1 us of matrix work.
// Here:
//   tile      192 x channels x 128 d channels per workgroup: a staged value feeds 128-192 outputs (1.5 vector instructions per MFMA);
//             K <= 192 is ONE k tile, so d is staged exactly once and x once per 128 output channels;
//   step      32 frames = one v_mfma_f32_16x16x32_bf16 step: 320 rows x 32 frames = 10 x 16-byte loads per thread of a group;
//   groups    a workgroup = 2 groups of 4 waves on the same tile, each with its own LDS image [3 planes][320 rows][32 frames + 8]
//             bf16 (80-byte pitch: conflict-free ds_read_b128; 2 x 77 KB); the groups take alternate steps and run HALF A PERIOD apart
//             (convwrw_tr.hip's schedule): while one multiplies (a wave owns 48 x 128 = 3 x 8 accumulator tiles: 144 MFMAs per
//             step), the other splits its next step into planes and stores them into ITS image, then issues the loads of the step
//             after that; one LDS-only barrier per half period, the loads stay in flight across it.  The first form of this kernel
//             (one group, 192 x 192 tiles, the split interleaved with the MFMA groups, two barriers per step) left the matrix pipe
//             idle while both waves of a SIMD stored planes and waited: 91 us for a flow block's six problems, this one 79;
//   batch     up to 8 problems per launch, each with its own channel counts, strides, masks, two-source gradient; the tiles of all
//             problems share the compute units: splits = CUs / (tiles of the batch), one round of workgroups.  A flow block's six
//             1x1 weight gradients (3 two-source res/skip, last res/skip, start, end) are ONE launch of 15 tiles;
//   epilogue  the two groups' sums meet in LDS, one set of float atomics per workgroup.
// Arithmetic: the six bf16 x bf16 products per fp32 product of convgemm_split.hip (exact 3-plane split, fp32 accumulation).
#include "convgemm_common.hpp"
#include "split_planes.hpp"

namespace glowtts {

constexpr int kW1Max = 8;

struct Wrw1Params {
    const float *x[kW1Max], *d[kW1Max], *d2[kW1Max], *mask_d[kW1Max], *mask_x[kW1Max];
    float *dwp[kW1Max], *dbias[kW1Max];
    long x_bs[kW1Max], d_bs[kW1Max], d2_bs[kW1Max];
    int Cin[kW1Max], M[kW1Max], d_split[kW1Max];
    int tile0[kW1Max + 1];       // first tile of each problem in the launch's tile numbering
    int n, B, T, steps_u, total_steps, nb, total_tiles, splits;
    int exp;                     // timing experiments (tools/wrw1_bench.py): bit 0 = no atomics, bit 1 = no plane split
};

int conv_math_wrw();             // convgemm_split.hip: planes per fp32 operand of the weight-gradient kernels (0 = native fp32)

__device__ __forceinline__ void w1_store8(void *p, unsigned lo, unsigned hi) {      // (see lds_store8 in convgemm_split.hip)
    // no "memory" clobber: the plane arithmetic of the other items may be scheduled around a store (with it every item's
    // dependent split chain ran alone, ~8 cycles per instruction: the staging phase took 1.2 us per step).  Volatile asms keep
    // their order among themselves, and the LDS-only barrier between a staging phase and the reads of the image does clobber memory.
    asm volatile("ds_write2_b32 %0, %1, %2 offset1:1"
                 :: "v"((unsigned)(size_t)(__attribute__((address_space(3))) void *)p), "v"(lo), "v"(hi));
}

template <int NS>
__global__ __launch_bounds__(512, 2) void convwrw1_kernel(Wrw1Params P) {
    constexpr int TK = 192, TM = 128, ROWS = TK + TM, RP = 40, PLANE = ROWS * RP, IMG = NS * PLANE;
    constexpr int NI = ROWS * 8 / 256, NXI = TK / 32;     // 16-byte staging items per thread and step: 6 of x, 4 of d
    static_assert(NI == 10 && NXI == 6, "ten items per thread");
    extern __shared__ __align__(16) unsigned short w1_lds[];                 // [2 groups][NS][ROWS][RP]

    const int tid = threadIdx.x, grp = tid >> 8, gt = tid & 255, wv = (tid >> 6) & 3, lane = tid & 63;
    const int lrow = lane & 15, lk = lane >> 4;
    unsigned short *img = w1_lds + grp * IMG;

    // ---- which problem, which tile, which steps -----------------------------------------------------------------------------
    // workgroup id = tile * splits + split with splits a multiple of 8 where the grid allows it: blocks are dealt round-robin over
    // the 8 XCDs, so the tiles of ONE split — which read the same frames: x for a problem's m tiles, dskip for a block's four
    // res/skip problems — share an XCD and its L2 (speed only, never correctness)
    const int id = blockIdx.x;
    const int tile = id / P.splits, split = id - tile * P.splits;
    int pi = 0;
#pragma unroll
    for (int j = 1; j < kW1Max; ++j)
        if (j < P.n && tile >= P.tile0[j]) pi = j;
    const float *px = P.x[0], *pd = P.d[0], *pd2 = P.d2[0], *pmd = P.mask_d[0], *pmx = P.mask_x[0];
    float *pdw = P.dwp[0], *pdb = P.dbias[0];
    long x_bs = P.x_bs[0], d_bs = P.d_bs[0], d2_bs = P.d2_bs[0];
    int Cin = P.Cin[0], M = P.M[0], d_split = P.d_split[0], t0p = P.tile0[0];
#pragma unroll
    for (int j = 1; j < kW1Max; ++j)                 // static indices: a run-time index would put the tables in scratch
        if (pi == j) {
            px = P.x[j]; pd = P.d[j]; pd2 = P.d2[j]; pmd = P.mask_d[j]; pmx = P.mask_x[j]; pdw = P.dwp[j]; pdb = P.dbias[j];
            x_bs = P.x_bs[j]; d_bs = P.d_bs[j]; d2_bs = P.d2_bs[j]; Cin = P.Cin[j]; M = P.M[j]; d_split = P.d_split[j];
            t0p = P.tile0[j];
        }
    const int ktiles = (Cin + TK - 1) / TK;
    const int lt = tile - t0p, kt = lt % ktiles, mt = lt / ktiles;
    const int k0 = kt * TK, m0 = mt * TM;
    const int T = P.T;
    const int s_begin = split * P.nb, s_end = min(P.total_steps, s_begin + P.nb);
    const int n_steps = max(0, s_end - s_begin);
    const int n_my = (n_steps - grp + 1) / 2;         // this group: steps s_begin + grp, s_begin + grp + 2, ...
    const int n_max = (n_steps + 1) / 2;

    const __amdgpu_buffer_rsrc_t xrs = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float *>(px), 0, (int)(((long)(P.B - 1) * x_bs + (long)Cin * T) * 4), 0x00020000);
    const int d_rows1 = pd2 ? d_split : M;
    const __amdgpu_buffer_rsrc_t drs = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float *>(pd), 0, (int)(((long)(P.B - 1) * d_bs + (long)d_rows1 * T) * 4), 0x00020000);
    const __amdgpu_buffer_rsrc_t d2rs = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float *>(pd2 ? pd2 : pd), 0, pd2 ? (int)(((long)(P.B - 1) * d2_bs + (long)(M - d_split) * T) * 4) : 0, 0x00020000);
    const __amdgpu_buffer_rsrc_t mdrs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(pmd), 0, pmd ? P.B * T * 4 : 0, 0x00020000);
    const __amdgpu_buffer_rsrc_t mxrs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(pmx), 0, pmx ? P.B * T * 4 : 0, 0x00020000);
    constexpr int kOOB = 0x7fffffff;
    auto ld16 = [&](const __amdgpu_buffer_rsrc_t &rs, int byte_off) -> f32x4 {
        return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs, byte_off, 0, 0));
    };

    // ---- staging map of a group: item i = rows 32 i .. 32 i + 31 of its image (i < 6: x rows, else d rows), thread = (row, quad) --
    const int srow = gt >> 3, q = gt & 7;
    // row of item i: x row k0 + 32 i + srow, d row m0 + 32 (i - 6) + srow (second source: minus d_split); `valid` bit i = the row exists
    const int xoff0 = (k0 + srow) * T, doff0 = (m0 + srow) * T;
    unsigned valid = 0;
    bool second[NI - NXI];                           // d items that come from the second source (uniform per item)
#pragma unroll
    for (int i = 0; i < NXI; ++i)
        if (k0 + 32 * i + srow < Cin) valid |= 1u << i;
#pragma unroll
    for (int i = NXI; i < NI; ++i) {
        const int mbase = m0 + 32 * (i - NXI);
        second[i - NXI] = pd2 != nullptr && mbase >= d_split;
        if (mbase + srow < M) valid |= 1u << i;
    }
    const int dsT = d_split * T;
    const bool do_bias = pdb != nullptr && kt == 0;
    float bsum[NI - NXI] = {0.f, 0.f, 0.f, 0.f};

    f32x4 raw[NI], mxv, mdv;
    int ld_i = 0;                                     // next item of this group to load
    auto load_next = [&]() {
        const int s = s_begin + grp + 2 * ld_i;
        ++ld_i;
        if ((GLOWTTS_EXP_BITS(P.exp) & 8) && ld_i > 1) return;          // (timing experiment: no global loads after the first step)
        const int b = s / P.steps_u, t = (s - b * P.steps_u) * 32 + q * 4;
        const bool tok = t < T;
#pragma unroll
        for (int i = 0; i < NXI; ++i)
            raw[i] = ld16(xrs, (tok && ((valid >> i) & 1)) ? (int)(((long)b * x_bs + xoff0 + 32 * i * T + t) * 4) : kOOB);
#pragma unroll
        for (int i = NXI; i < NI; ++i) {
            const bool ok = tok && ((valid >> i) & 1);
            const int ro = doff0 + 32 * (i - NXI) * T + t;
            if (second[i - NXI]) raw[i] = ld16(d2rs, ok ? (int)(((long)b * d2_bs + ro - dsT) * 4) : kOOB);
            else                 raw[i] = ld16(drs, ok ? (int)(((long)b * d_bs + ro) * 4) : kOOB);
        }
        if (pmx) mxv = ld16(mxrs, tok ? (b * T + t) * 4 : kOOB);
        if (pmd) mdv = ld16(mdrs, tok ? (b * T + t) * 4 : kOOB);
    };
    auto stage = [&]() {                              // the loaded step: split into planes, into this group's image
#pragma unroll
        for (int i = 0; i < NI; ++i) {
            f32x4 v = raw[i];
            if (i < NXI) { if (pmx) v *= mxv; }
            else {
                if (pmd) v *= mdv;
                if (do_bias) bsum[i - NXI] += (v[0] + v[1]) + (v[2] + v[3]);
            }
            unsigned o01[NS], o23[NS];
            split_planes2<NS>(v[0], v[1], o01);
            split_planes2<NS>(v[2], v[3], o23);
#pragma unroll
            for (int p = 0; p < NS; ++p) w1_store8(img + p * PLANE + (32 * i + srow) * RP + q * 4, o01[p], o23[p]);
        }
    };
    auto lds_barrier = [&]() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); };

    // ---- accumulators: wave wv of a group owns x rows 48 wv .. + 48 and all 128 d rows of the tile ---------------------------------
    f32x4 acc[3][8];
#pragma unroll
    for (int a = 0; a < 3; ++a)
#pragma unroll
        for (int c = 0; c < 8; ++c) acc[a][c] = f32x4{0.f, 0.f, 0.f, 0.f};
    bool a_ok[3], c_ok[8];
#pragma unroll
    for (int a = 0; a < 3; ++a) a_ok[a] = k0 + 48 * wv + 16 * a < Cin;
#pragma unroll
    for (int c = 0; c < 8; ++c) c_ok[c] = m0 + 16 * c < M;

    const unsigned short *xa = img + (48 * wv + lrow) * RP + lk * 8;
    const unsigned short *da = img + (TK + lrow) * RP + lk * 8;
    auto compute = [&]() {
        i32x4 A[3][NS], Bv[2][NS];
#pragma unroll
        for (int a = 0; a < 3; ++a)
#pragma unroll
            for (int p = 0; p < NS; ++p) A[a][p] = *reinterpret_cast<const i32x4 *>(xa + p * PLANE + 16 * a * RP);
#pragma unroll
        for (int p = 0; p < NS; ++p) Bv[0][p] = *reinterpret_cast<const i32x4 *>(da + p * PLANE);
#pragma unroll
        for (int c = 0; c < 8; ++c) {
            if (c + 1 < 8)
#pragma unroll
                for (int p = 0; p < NS; ++p) Bv[(c + 1) & 1][p] = *reinterpret_cast<const i32x4 *>(da + p * PLANE + 16 * (c + 1) * RP);
            __builtin_amdgcn_sched_barrier(0);
            if (c_ok[c]) {
#pragma unroll
                for (int a = 0; a < 3; ++a)
                    if (a_ok[a]) {
#pragma unroll
                        for (int k = 0; k < n_products(NS); ++k)
                            acc[a][c] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(
                                __builtin_bit_cast(bf16x8, A[a][product_a(NS, k)]),
                                __builtin_bit_cast(bf16x8, Bv[c & 1][product_b(NS, k)]), acc[a][c], 0, 0, 0);
                    }
            }
            __builtin_amdgcn_sched_barrier(0);
        }
    };

    // ---- half-period schedule (convwrw_tr.hip's): group g multiplies its item i in phase 2 i + g and, in phase 2 i + g + 1, stages
    // item i + 1 into its own image and issues the loads of item i + 2: every SIMD has one wave multiplying and one staging
    if (n_my > 0) load_next();
    if (grp == 0 && n_my > 0) {
        stage();
        if (n_my > 1) load_next();
    }
    lds_barrier();
    for (int ph = 0; ph < 2 * n_max; ++ph) {
        const int rel = ph - grp;
        if (rel >= 0 && (rel & 1) == 0) {
            if ((rel >> 1) < n_my && !(GLOWTTS_EXP_BITS(P.exp) & 4)) compute();
        } else {
            const int i = (rel + 1) >> 1;
            if (i < n_my && (i > 0 || grp == 1)) {
                stage();
                if (i + 1 < n_my) load_next();
            }
        }
        lds_barrier();
    }

    // ---- the two groups' sums meet in LDS (the images are dead); group 0 sends the tile's split-K atomics -----------------------------
    if (GLOWTTS_EXP_BITS(P.exp) & 1) { if (acc[0][0][0] == 123.456f) pdw[0] = 1.f; return; }
    float *red = reinterpret_cast<float *>(w1_lds);
    if (grp == 1) {
#pragma unroll
        for (int a = 0; a < 3; ++a)
#pragma unroll
            for (int c = 0; c < 8; ++c)
                *reinterpret_cast<f32x4 *>(red + (((a * 8 + c) * 256) + gt) * 4) = acc[a][c];
    }
    __syncthreads();
    if (grp == 0) {
#pragma unroll
        for (int a = 0; a < 3; ++a)
#pragma unroll
            for (int c = 0; c < 8; ++c) {
                const f32x4 o = *reinterpret_cast<const f32x4 *>(red + (((a * 8 + c) * 256) + gt) * 4);
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int k = k0 + 48 * wv + 16 * a + 4 * lk + r;
                    const int m = m0 + 16 * c + lrow;
                    if (k < Cin && m < M) atomicAdd(pdw + (long)k * M + m, acc[a][c][r] + o[r]);
                }
            }
    }
    if (do_bias) {                                    // row sums of the (masked) d rows each group staged
#pragma unroll
        for (int i = 0; i < NI - NXI; ++i) {
            float v = bsum[i];
            v += __shfl_xor(v, 1, 64);
            v += __shfl_xor(v, 2, 64);
            v += __shfl_xor(v, 4, 64);
            const int m = m0 + 32 * i + srow;
            if (q == 0 && m < M) atomicAdd(pdb + m, v);
        }
    }
}

static int w1_compute_units() {
    static int n[kMaxDevices] = {0};
    int dev = 0;
    (void)hipGetDevice(&dev);
    if (dev < 0 || dev >= kMaxDevices) dev = 0;
    if (n[dev] == 0) {
        int v = 0;
        if (hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || v <= 0) v = 256;
        n[dev] = v;
    }
    return n[dev];
}

// 0 = launched; -1 = not handled (arithmetic mode, shapes, alignment): the caller launches the problems one by one
int conv_wrw1_multi_dispatch(int n, const glowtts_wrw1_problem *pr, int B, int T, hipStream_t s) {
    if (conv_math_wrw() != 3 || knob(K_WRW1_MULTI) == 0) return -1;
    if (n < 1 || n > kW1Max || (T & 3) != 0 || B < 1 || T < 4) return -1;
    Wrw1Params P{};
    int tiles = 0;
    for (int j = 0; j < n; ++j) {
        const glowtts_wrw1_problem &q = pr[j];
        if (!q.x || !q.d || !q.dwp || q.Cin < 1 || q.M < 1) return -1;
        if (!aligned16(q.x) || !aligned16(q.d) || (q.d2 && !aligned16(q.d2)) || (q.mask_d && !aligned16(q.mask_d)) ||
            (q.mask_x && !aligned16(q.mask_x)) || (q.x_bs & 3) || (q.d_bs & 3) || (q.d2 && (q.d2_bs & 3)))
            return -1;
        if (q.d2 && (q.d_split % 64 != 0 || q.d_split <= 0 || q.d_split >= q.M)) return -1;
        const long xb = ((long)(B - 1) * q.x_bs + (long)q.Cin * T) * 4, db = ((long)(B - 1) * q.d_bs + (long)q.M * T) * 4;
        const long d2b = q.d2 ? ((long)(B - 1) * q.d2_bs + (long)q.M * T) * 4 : 0;
        if (xb > 0x7ffffff0L || db > 0x7ffffff0L || d2b > 0x7ffffff0L || (long)B * T * 4 > 0x7ffffff0L) return -1;
        P.x[j] = q.x; P.d[j] = q.d; P.d2[j] = q.d2; P.mask_d[j] = q.mask_d; P.mask_x[j] = q.mask_x; P.dwp[j] = q.dwp; P.dbias[j] = q.dbias;
        P.x_bs[j] = q.x_bs; P.d_bs[j] = q.d_bs; P.d2_bs[j] = q.d2_bs; P.Cin[j] = q.Cin; P.M[j] = q.M; P.d_split[j] = q.d_split;
        P.tile0[j] = tiles;
        tiles += ((q.Cin + 191) / 192) * ((q.M + 127) / 128);
    }
    for (int j = n; j <= kW1Max; ++j) P.tile0[j] = tiles;
    P.n = n; P.B = B; P.T = T; P.total_tiles = tiles;
    #ifdef GLOWTTS_TRACE
    P.exp = knob(K_WRW1_EXP);
#endif
    P.steps_u = (T + 31) / 32;
    P.total_steps = B * P.steps_u;
    // split-K sized for HALF the compute units: every workgroup ends with 98 KB of float atomics (a tile is the whole problem, so
    // splits x tiles x 98 KB leave the chip at ~1.3 TB/s whatever the kernel does), and the launch shares the GPU with the
    // backward's chain on the other stream — alone the launch is faster on all CUs (79 against 117 us for a flow block's six
    // problems), in the step it is not: 14.34 / 14.24 / 14.17 ms per step for 256 / 192 / 128 (tools/ab_flags.py envs=GLOWTTS_WRW1_CUS:..)
    const int cus = knob(K_WRW1_CUS);                                  // -1 (default): half the compute units
    int splits = (cus > 0 ? cus : w1_compute_units() / 2) / tiles;
    if (splits >= 16 && knob(K_WRW1_XCD)) splits &= ~7;      // (see the kernel's workgroup numbering)
    if (splits > P.total_steps) splits = P.total_steps;
    if (splits < 1) splits = 1;
    P.nb = (P.total_steps + splits - 1) / splits;
    if (!(splits >= 16 && (splits & 7) == 0)) splits = (P.total_steps + P.nb - 1) / P.nb;       // (a multiple of 8 stays: a split may be empty)
    P.splits = splits;
    constexpr size_t lds = (size_t)2 * 3 * 320 * 40 * 2;             // two groups' images; the group reduction reuses them
    static LdsLimit attr_max_e;
    if (int rc_ = attr_max_e.ensure(reinterpret_cast<const void *>(&convwrw1_kernel<3>), lds, "glowtts_conv_wrw1_multi")) return rc_;
    hipLaunchKernelGGL((convwrw1_kernel<3>), dim3(tiles * splits), dim3(512), lds, s, P);
    hipError_t e_ = hipGetLastError();
    if (e_ != hipSuccess) { set_error("glowtts_conv_wrw1_multi: launch failed: %s", hipGetErrorString(e_)); return (int)e_; }
    return 0;
}

}  // namespace glowtts

using namespace glowtts;

extern "C" int glowtts_conv_wrw1_multi(int n, const glowtts_wrw1_problem *problems, int B, int T, glowtts_stream_t stream) {
    GLOWTTS_CHECK_ARG(n >= 1 && problems, "glowtts_conv_wrw1_multi: no problems");
    GLOWTTS_CHECK_ARG(B >= 0 && T >= 0, "glowtts_conv_wrw1_multi: bad shape");
    if ((long)B * T == 0) return 0;
    for (int q0 = 0; q0 < n; q0 += kW1Max) {
        const int nb = n - q0 < kW1Max ? n - q0 : kW1Max;
        const int rc = conv_wrw1_multi_dispatch(nb, problems + q0, B, T, (hipStream_t)stream);
        if (rc > 0) return rc;
        if (rc == 0) continue;
        for (int j = q0; j < q0 + nb; ++j) {          // one by one on the single-problem kernels
            const glowtts_wrw1_problem &q = problems[j];
            GLOWTTS_CHECK_ARG(q.x && q.d && q.dwp, "glowtts_conv_wrw1_multi: null pointer in problem %d", j);
            const int r1 = q.d2 ? glowtts_conv_wrw2(q.x, q.x_bs, q.d, q.d_bs, q.d2, q.d2_bs, q.d_split, q.dwp, q.dbias, B, q.Cin, q.M, T, 1,
                                                    1, 0, stream)
                                : glowtts_conv_wrw(q.x, q.x_bs, q.d, q.d_bs, q.mask_d, q.mask_x, q.dwp, q.dbias, B, q.Cin, q.M, T, 1, 1, 0,
                                                   stream);
            if (r1 != 0) return r1;
        }
    }
    return 0;
}
