// convwrw_tr.hip — weight gradient of the WN stack's 5-tap convolutions on the bf16 matrix pipe, frame-major LDS images read
// with gfx950's transposing LDS load (reference op: autograd of F.conv1d, layers.py:146,156).
//
//   dW[tap][k][m] = sum_{b,t} x[b][k][t + tap - pad] * d[b][m][t]          (k: input channel, m: output channel)
//
// The frame-packed kernel of convgemm_split.hip keeps x as [channel][frame] in LDS, so a tap is a shift by whole bf16
// elements INSIDE a lane's register window: ~100 vector instructions (v_alignbit / v_perm / moves to even-aligned register
// tuples) per 60 MFMAs, on the issue port the MFMAs share — and both co-resident workgroups of a CU stage (split fp32 into
// bf16 planes: ~1 vector instruction per MFMA) and multiply in lock-step, so the matrix pipe idles while they stage
// (measured 72 us at B=32 / T'=400 where its MFMAs alone need ~36).  Here:
//   * LDS images are FRAME-major, [plane][frame][channel]: a tap is a ROW offset, i.e. an immediate in the read's address —
//     the multiply phase has no vector instruction besides the MFMAs.  An MFMA operand (8 frames of one channel per lane)
//     comes from two ds_read_b64_tr_b16 (each: 4 frames x 16 channels, transposed by the LDS on the way out).
//   * a workgroup = 2 groups of 4 waves on the same 64 x 32 tile; the groups take alternate chunks of the workgroup's frames
//     and run half a period apart: while one multiplies, the other splits and stores its next chunk (one barrier per half
//     period).  A SIMD always has one wave in each role, so the matrix pipe never waits for staging.
//   * the two groups' partial sums meet in LDS and leave as ONE set of atomics: 10 MB per launch instead of 20.
//   * chunks are 64 frames (two 32-frame MFMA steps; the last chunk of an utterance runs only the steps it has frames for):
//     T' = 400 costs 13 steps per utterance instead of 15 with the 80-frame chunks padded to 96.
//
// Frame <-> MFMA k index (the same for both operands, so any bijection is legal): element e of lane group g (g = lane >> 4)
// of the first transposed read is frame 8 (g & 1) + 2 e + (g >> 1) of a 16-frame window, the second read the next 16 frames.
// A 32-lane half therefore reads 8 frames of equal parity, which with the pitches below (36 / 20 dwords: 4 mod 8) fall on
// 8 disjoint groups of 8 banks; the staging stores (one dword = a channel PAIR of one frame, as v_cvt_pk_bf16_f32 leaves it)
// go 16 pairs x 2 frame quads per half-wave, again conflict-free with these pitches.
#include "convgemm_common.hpp"
#include "split_planes.hpp"

#include <cstdlib>

namespace glowtts {

typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ bf16x8 lds_tr8(const char *p0, const char *p1) {
    const bf16x4 a = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((__attribute__((address_space(3))) bf16x4 *)(p0));
    const bf16x4 b = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((__attribute__((address_space(3))) bf16x4 *)(p1));
    return __builtin_shufflevector(a, b, 0, 1, 2, 3, 4, 5, 6, 7);
}

template <int TAPS, int MT>
struct WrwTrGeom {
    static constexpr int MR = 16 * MT;                  // d rows (output channels) of a tile
    static constexpr int CF = 64;                       // frames per chunk
    static constexpr int PAD = (TAPS - 1) / 2, OFF = (4 - (PAD & 3)) & 3;
    static constexpr int XR = CF + 8;                   // x rows of a chunk: frames [tc - PAD - OFF, + XR)
    static constexpr int XP = 144, DP = MR * 2 + 16;    // row pitches in bytes (36 / 20 or 36 dwords)
    static constexpr int XPL = XR * XP, DPL = CF * DP;  // one plane image
    static_assert(TAPS - 1 + OFF < 8 && (XP / 4) % 8 == 4 && (DP / 4) % 8 == 4, "window and bank layout");
    static constexpr size_t group_bytes(int ns) { return (size_t)ns * (XPL + DPL); }
    static constexpr size_t lds_bytes(int ns, bool w32 = false, int ng = 2) {
        const size_t img = ng * group_bytes(ns), red = ng == 2 ? (size_t)TAPS * (w32 ? 16 : MT * 4) * 256 * 4 : 0;
        return (img > red ? img : red) + MR * sizeof(float);
    }
};

typedef float f32x16 __attribute__((ext_vector_type(16)));

// W32: the multiply phase on v_mfma_f32_32x32x16_bf16 — a wave owns 32 x channels x 32 d channels of ONE 32-frame step of the chunk
// (waves 0, 1: the tile's two k halves on step 0; waves 2, 3: on step 1), 60 MFMAs of 32 cycles per chunk instead of 120 of 16.
// Same matrix work; but a 32x32x16 MFMA holds the SIMD's vector issue port 8 of its 32 cycles where the 16x16x32 one holds it 8 of
// 16, so the SIMD's OTHER wave — splitting and storing the next chunk, ~2 vector instructions per 16 MFMA cycles, the pace-setter of
// the W32 = false form — gets three quarters of the issue slots instead of half.  Four partial sums per output meet in LDS.
// NG = 1 (round 4, opt-in: GLOWTTS_WRW_TR_NG=1): ONE group per workgroup — 256 threads, one LDS image (59 KB for the 5-tap 64 x 64
// form instead of 118): the multiply and the store phase of a workgroup then alternate instead of overlapping, and what fills the matrix
// pipe meanwhile is whatever else the CU holds — in the backward a workgroup of the chain's convolution kernels, which the two-group
// form's LDS keeps off the CU (DESIGN.md lesson 36).
template <int NS, int TAPS, int MT, bool W32 = false, int NG = 2>
__global__ __launch_bounds__(256 * NG, 2) void convwrw_tr_kernel(ConvWrwParams p) {
    static_assert(!W32 || MT == 2, "the 32x32 form is the 64 x 32 tile");
    static_assert(NG == 2 || !W32, "one group: the 16x16x32 forms only");
    using G = WrwTrGeom<TAPS, MT>;
    constexpr int MR = G::MR, CF = G::CF, PAD = G::PAD, OFF = G::OFF, XR = G::XR, XP = G::XP, DP = G::DP;
    constexpr int XPL = G::XPL, DPL = G::DPL;
    constexpr int NXI = 32 * (XR / 4), NXR = (NXI + 255) / 256;            // x items: (channel pair, frame quad)
    constexpr int NDI = (MR / 2) * (CF / 4), NDR = (NDI + 255) / 256;      // d items
    constexpr int CPH = MR / 32;                                           // 16-pair halves of a d row
    extern __shared__ __align__(16) char smem_tr[];
    const int tid = threadIdx.x, grp = NG == 2 ? tid >> 8 : 0, gt = tid & 255, wave = (tid >> 6) & 3, lane = tid & 63;
    char *Xg = smem_tr + grp * G::group_bytes(NS);
    char *Dg = Xg + NS * XPL;
    float *rowacc = reinterpret_cast<float *>(smem_tr + G::lds_bytes(NS, W32, NG) - MR * sizeof(float));

    // ---- which tile, which frames (workgroups numbered XCD-major: the splits of one tile share frames with the other tiles
    // of the same split, which then sit on one or two XCDs' L2s)
    const int nkt = (p.Cin + 63) / 64;
    const int ntiles = gridDim.x, nwg = gridDim.x * gridDim.z;
    const int id = blockIdx.x + blockIdx.z * gridDim.x;
    const int xcd = id & 7, slot = id >> 3;
    int witem = xcd * (nwg >> 3) + min(xcd, nwg & 7) + slot;
    if (p.nbatch > 0) {                               // XCD-major over the whole batch: an XCD works on one problem at a time
        const int per = nwg / p.nbatch, q = witem / per;
        witem -= q * per;
#pragma unroll
        for (int j = 0; j < ConvWrwParams::kMaxBatch; ++j)         // static indices: a run-time index would put the tables in scratch
            if (j == q) { p.x = p.bx[j]; p.d = p.bd[j]; p.d2 = p.bd2[j]; p.dwp = p.bdwp[j]; p.dbias = p.bdbias[j]; }
    }
    const int tile = witem % ntiles, split = witem / ntiles;
    const int kt = tile % nkt, mt = tile / nkt;
    const int k0 = kt * 64, m0 = mt * MR;
    const int nct = (p.T + CF - 1) / CF;
    const int c0 = split * p.nb;
    const int n_split = max(0, min(p.B * nct, c0 + p.nb) - c0);           // (utterance, chunk) items of this workgroup
    const int n_my = NG == 2 ? (n_split - grp + 1) / 2 : n_split;          // this group: items c0 + grp, c0 + grp + 2, ... (NG = 1: all)
    const int n_max = NG == 2 ? (n_split + 1) / 2 : n_split;

    f32x4 acc[W32 ? 1 : TAPS][W32 ? 1 : MT];
    f32x16 acc32[W32 ? TAPS : 1];
#pragma unroll
    for (int tp = 0; tp < (W32 ? 1 : TAPS); ++tp)
#pragma unroll
        for (int i = 0; i < (W32 ? 1 : MT); ++i) acc[tp][i] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int tp = 0; tp < (W32 ? TAPS : 1); ++tp)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc32[tp][e] = 0.f;

    const bool do_bias = (p.dbias != nullptr) && (kt == 0);
    const int xbytes = (int)(((long)(p.B - 1) * p.x_bs + (long)p.Cin * p.T) * 4);
    const bool d_second = p.d2 != nullptr && m0 >= p.d_split;
    const int m_rows = d_second ? p.M - p.d_split : (p.d2 ? p.d_split : p.M);
    const int m_base = d_second ? m0 - p.d_split : m0;
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

    // ---- staging map.  item idx = gt + 256 r: bits 0-3 channel pair (low), bit 4 frame quad (low), then the pair's high
    // bit(s), then the quad's high bits — a half-wave stores 16 pairs x 2 quads (and loads 32 B runs of 32 rows)
    int xq[NXR], xc0[NXR], xc1[NXR], xw[NXR];          // frame quad; BYTE offsets of the pair's two rows + quad (kOOB: no such row); LDS byte offset (-1: no item)
#pragma unroll
    for (int r = 0; r < NXR; ++r) {
        const int idx = gt + 256 * r;
        const int cp = (idx & 15) + 16 * ((idx >> 5) & 1), fq = ((idx >> 4) & 1) + 2 * (idx >> 6);
        const bool ok = idx < NXI;
        xq[r] = fq * 4;
        xc0[r] = (ok && k0 + 2 * cp < p.Cin) ? ((k0 + 2 * cp) * p.T + fq * 4) * 4 : kOOB;
        xc1[r] = (ok && k0 + 2 * cp + 1 < p.Cin) ? ((k0 + 2 * cp + 1) * p.T + fq * 4) * 4 : kOOB;
        xw[r] = ok ? (fq * 4) * XP + cp * 4 : -1;
    }
    int dq[NDR], dc0[NDR], dc1[NDR], dw_[NDR];
#pragma unroll
    for (int r = 0; r < NDR; ++r) {
        const int idx = gt + 256 * r;
        const int cp = (idx & 15) + 16 * ((idx >> 5) & (CPH - 1));
        const int fq = ((idx >> 4) & 1) + 2 * (idx >> (CPH == 2 ? 6 : 5));
        const bool ok = idx < NDI;
        dq[r] = fq * 4;
        dc0[r] = (ok && m_base + 2 * cp < m_rows) ? ((m_base + 2 * cp) * p.T + fq * 4) * 4 : kOOB;
        dc1[r] = (ok && m_base + 2 * cp + 1 < m_rows) ? ((m_base + 2 * cp + 1) * p.T + fq * 4) * 4 : kOOB;
        dw_[r] = ok ? (fq * 4) * DP + cp * 4 : -1;
    }
    f32x4 xa[NXR], xb[NXR], da[NDR], db[NDR], mxr[NXR], mdr[NDR];
    float bs0[NDR], bs1[NDR];
#pragma unroll
    for (int r = 0; r < NDR; ++r) bs0[r] = bs1[r] = 0.f;

    // the group's items in order: (b, ci) of the next item to load, advanced by two items at a time without divisions
    int nb_ = (c0 + grp) / nct, nci = (c0 + grp) - nb_ * nct;
    int steps_loaded = 0, steps_staged = 0;           // MFMA steps of the item in registers / in the LDS images
    auto load_next = [&]() {                          // the group's next item -> registers
        const int b = nb_, tc = nci * CF;
        const int ts = tc - PAD - OFF;
        steps_loaded = (min(CF, p.T - tc) + 31) >> 5;
        nci += NG;
        while (nci >= nct) { nci -= nct; ++nb_; }
        const int xbo = (b * (int)p.x_bs + ts) * 4, dbo = (b * (int)d_bs + tc) * 4;     // (ts may be -4: only added to valid quads)
        const int xlo = -ts, xhi = p.T - ts, dhi = p.T - tc;                             // valid quads: xlo <= 4 fq < xhi (T % 4 == 0)
#pragma unroll
        for (int r = 0; r < NXR; ++r) {
            const bool in = xq[r] >= xlo && xq[r] < xhi;
            xa[r] = ld16(xrs, (in && xc0[r] != kOOB) ? xc0[r] + xbo : kOOB);
            xb[r] = ld16(xrs, (in && xc1[r] != kOOB) ? xc1[r] + xbo : kOOB);
            if (p.mask_x) mxr[r] = ld16(mxrs, in ? (b * p.T + ts + xq[r]) * 4 : kOOB);
        }
#pragma unroll
        for (int r = 0; r < NDR; ++r) {
            const bool in = dq[r] < dhi;
            da[r] = ld16(drs, (in && dc0[r] != kOOB) ? dc0[r] + dbo : kOOB);
            db[r] = ld16(drs, (in && dc1[r] != kOOB) ? dc1[r] + dbo : kOOB);
            if (p.mask) mdr[r] = ld16(mdrs, in ? (b * p.T + tc + dq[r]) * 4 : kOOB);
        }
    };
    auto stage = [&]() {                               // registers -> bf16 planes in this group's images
        steps_staged = steps_loaded;
#pragma unroll
        for (int r = 0; r < NXR; ++r)
            if (xw[r] >= 0) {
                f32x4 va = xa[r], vb = xb[r];
                if (p.mask_x) { va *= mxr[r]; vb *= mxr[r]; }
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    unsigned o[NS];
                    split_planes2<NS>(va[j], vb[j], o);
#pragma unroll
                    for (int pl = 0; pl < NS; ++pl) *reinterpret_cast<unsigned *>(Xg + pl * XPL + xw[r] + j * XP) = o[pl];
                }
            }
#pragma unroll
        for (int r = 0; r < NDR; ++r)
            if (dw_[r] >= 0) {
                f32x4 va = da[r], vb = db[r];
                if (p.mask) { va *= mdr[r]; vb *= mdr[r]; }
                if (do_bias) {
                    bs0[r] += (va[0] + va[1]) + (va[2] + va[3]);
                    bs1[r] += (vb[0] + vb[1]) + (vb[2] + vb[3]);
                }
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    unsigned o[NS];
                    split_planes2<NS>(va[j], vb[j], o);
#pragma unroll
                    for (int pl = 0; pl < NS; ++pl) *reinterpret_cast<unsigned *>(Dg + pl * DPL + dw_[r] + j * DP) = o[pl];
                }
            }
    };

    // ---- operand addresses: lane (g, q, p) supplies row 8 (g & 1) + 2 q + (g >> 1) of a 16-row window, columns 4 p .. 4 p + 3
    // of its wave's 16 channels (x) / of m-tile i (d)
    const int lg = lane >> 4, lq = (lane >> 2) & 3, lp = lane & 3;
    const int lrow16 = 8 * (lg & 1) + 2 * lq + (lg >> 1);
    const char *xl = Xg + lrow16 * XP + wave * 32 + lp * 8;
    const char *dl = Dg + lrow16 * DP + lp * 8;
    // W32 addresses: lane group g = lane >> 4: channel half cg = g & 1 (16 of the wave's 32 channels), k slot h = g >> 1; the
    // 16 frames of a substep map to (h, element j) as 4 (j & 3) + 2 h + (j >> 2): a transposed read takes 4 rows spaced by FOUR
    // (36 x 4 and 20 x 4 dwords are 16 mod 64), its two 16-lane groups neighbouring 32-byte column chunks: 64 banks, no conflict
    const int kh = wave & 1, fh = wave >> 1;
    const char *xl32 = Xg + (4 * lq + 2 * (lg >> 1)) * XP + (32 * kh + 16 * (lg & 1)) * 2 + lp * 8;
    const char *dl32 = Dg + (4 * lq + 2 * (lg >> 1)) * DP + (16 * (lg & 1)) * 2 + lp * 8;
    auto compute32 = [&](int nsteps) {
        if (fh >= nsteps) return;
#pragma unroll
        for (int ss = 0; ss < 2; ++ss) {
            const int f0 = fh * 32 + ss * 16;              // (fh is wave-uniform: two copies of the loop body)
            bf16x8 bv[NS];
#pragma unroll
            for (int pl = 0; pl < NS; ++pl) {
                const char *q = dl32 + pl * DPL + f0 * DP;
                bv[pl] = lds_tr8(q, q + DP);
            }
            bf16x8 av[2][NS];
            auto afetch = [&](int tp, int sl) {
#pragma unroll
                for (int pl = 0; pl < NS; ++pl) {
                    const char *q = xl32 + pl * XPL + (f0 + tp + OFF) * XP;
                    av[sl][pl] = lds_tr8(q, q + XP);
                }
            };
            afetch(0, 0);
#pragma unroll
            for (int tp = 0; tp < TAPS; ++tp) {
                if (tp + 1 < TAPS) afetch(tp + 1, (tp + 1) & 1);
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int k = 0; k < n_products(NS); ++k)
                    acc32[W32 ? tp : 0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(av[tp & 1][product_a(NS, k)], bv[product_b(NS, k)],
                                                                                 acc32[W32 ? tp : 0], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
    };
    auto compute = [&](int nsteps) {
        if constexpr (W32) { compute32(nsteps); return; }
        // (a software pipeline ACROSS the two steps — the next step's operands read before the last tap's MFMAs — was measured
        //  slower: 60.6 us of main loop against 52.4; 40 more registers and a longer dependence-free window did not help the
        //  scheduler)
#pragma unroll
        for (int s_ = 0; s_ < CF / 32; ++s_) {
            if (s_ < nsteps) {
                bf16x8 bv[MT][NS];
#pragma unroll
                for (int i = 0; i < MT; ++i)
#pragma unroll
                    for (int pl = 0; pl < NS; ++pl) {
                        const char *q = dl + pl * DPL + (s_ * 32) * DP + i * 32;
                        bv[i][pl] = lds_tr8(q, q + 16 * DP);
                    }
                bf16x8 av[2][NS];
                auto afetch = [&](int tp, int sl) {
#pragma unroll
                    for (int pl = 0; pl < NS; ++pl) {
                        const char *q = xl + pl * XPL + (s_ * 32 + tp + OFF) * XP;
                        av[sl][pl] = lds_tr8(q, q + 16 * XP);
                    }
                };
                afetch(0, 0);
#pragma unroll
                for (int tp = 0; tp < TAPS; ++tp) {
                    if (tp + 1 < TAPS) afetch(tp + 1, (tp + 1) & 1);
                    __builtin_amdgcn_sched_barrier(0);     // the next tap's reads stay ahead of this tap's MFMAs
#pragma unroll
                    for (int i = 0; i < MT; ++i)
#pragma unroll
                        for (int k = 0; k < n_products(NS); ++k)
                            acc[W32 ? 0 : tp][W32 ? 0 : i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(
                                av[tp & 1][product_a(NS, k)], bv[i][product_b(NS, k)], acc[W32 ? 0 : tp][W32 ? 0 : i], 0, 0, 0);
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
        }
    };
    // LDS-only barrier: the next item's global loads stay in flight across it (__syncthreads would drain vmcnt as well)
    auto lds_barrier = [&]() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); };

    // ---- half-period schedule: group g multiplies its item i in phase 2 i + g and, in phase 2 i + g + 1, stores item i + 1 and
    // issues the loads of item i + 2 (item 0: group 0 before the loop, group 1 in phase 0).  A multiplying wave issues nothing
    // but LDS reads and MFMAs; the SIMD's other wave (storing) fills the issue slots the MFMAs leave and runs at RAISED priority:
    // per half period the storing wave needs ~3 000 cycles, the multiplying one ~2 650 (trace build, tools/trace_conv.py wrw5:
    // whichever side is raised, the sum stays — the SIMD's issue port and the LDS are shared — but the storing side is the longer).
    GLOWTTS_TRACE_POINT_Z(0);
    if (tid < MR) rowacc[tid] = 0.f;
    if (n_my > 0) load_next();
    if (grp == 0 && n_my > 0) {
        stage();
        if (n_my > 1) load_next();
    }
    lds_barrier();
    GLOWTTS_TRACE_POINT_Z(1);
#ifdef GLOWTTS_TRACE   // where a half period goes (shader cycles, summed over the loop; wave 0 of each group): slots 5-8 / 9, 11-13
    unsigned long long tr_mul = 0, tr_mul_wait = 0, tr_store = 0, tr_store_wait = 0;
#define TR_NOW() __builtin_readcyclecounter()
#else
#define TR_NOW() 0ull
#endif
    for (int ph = 0; ph < 2 * n_max; ++ph) {
        const int rel = ph - grp;
        [[maybe_unused]] const unsigned long long t_a = TR_NOW();
        [[maybe_unused]] bool mul = false;
        if (rel >= 0 && (rel & 1) == 0) {
            mul = true;
            if ((rel >> 1) < n_my) {
                if (p.xs_pitch == 0) __builtin_amdgcn_s_setprio(1);
                compute(steps_staged);
                if (p.xs_pitch == 0) __builtin_amdgcn_s_setprio(0);
            }
        } else {
            const int i = (rel + 1) >> 1;
            if (i < n_my && (i > 0 || grp == 1)) {
                if (p.xs_pitch == 2) __builtin_amdgcn_s_setprio(1);
                stage();
                if (i + 1 < n_my) load_next();
                if (p.xs_pitch == 2) __builtin_amdgcn_s_setprio(0);
            }
        }
        [[maybe_unused]] const unsigned long long t_b = TR_NOW();
        lds_barrier();
#ifdef GLOWTTS_TRACE
        const unsigned long long t_c = TR_NOW();
        if (mul) { tr_mul += t_b - t_a; tr_mul_wait += t_c - t_b; } else { tr_store += t_b - t_a; tr_store_wait += t_c - t_b; }
#endif
    }
#ifdef GLOWTTS_TRACE
    if (gt == 0) {
        unsigned long long *rec = g_trace + ((blockIdx.z * gridDim.x + blockIdx.x) & 8191) * 16;
        const int o = grp == 0 ? 5 : 9;
        rec[o] = tr_mul;
        rec[grp == 0 ? 6 : 11] = tr_mul_wait;
        rec[grp == 0 ? 7 : 12] = tr_store;
        rec[grp == 0 ? 8 : 13] = tr_store_wait;
    }
#endif

    GLOWTTS_TRACE_POINT_Z(2);
    // ---- the two groups' sums meet in LDS (the images are dead), group 0 sends the tile's atomics
    float *red = reinterpret_cast<float *>(smem_tr);
    if (do_bias) {
#pragma unroll
        for (int r = 0; r < NDR; ++r)
            if (dw_[r] >= 0) {
                const int idx = gt + 256 * r;
                const int cp = (idx & 15) + 16 * ((idx >> 5) & (CPH - 1));
                atomicAdd(rowacc + 2 * cp, bs0[r]);
                atomicAdd(rowacc + 2 * cp + 1, bs1[r]);
            }
    }
    if constexpr (W32) {
        // four partial sums per output: (group, frame half).  Group 1 -> LDS -> the same wave of group 0; then group 0's
        // step-1 waves -> LDS -> its step-0 waves, which send the atomics (128 lanes x 80 values = the tile once)
        if (grp == 1) {
#pragma unroll
            for (int tp = 0; tp < TAPS; ++tp)
#pragma unroll
                for (int e = 0; e < 16; ++e) red[(tp * 16 + e) * 256 + gt] = acc32[tp][e];
        }
        __syncthreads();
        if (grp == 0) {
#pragma unroll
            for (int tp = 0; tp < TAPS; ++tp)
#pragma unroll
                for (int e = 0; e < 16; ++e) acc32[tp][e] += red[(tp * 16 + e) * 256 + gt];
        }
        __syncthreads();
        if (grp == 0 && fh == 1) {
#pragma unroll
            for (int tp = 0; tp < TAPS; ++tp)
#pragma unroll
                for (int e = 0; e < 16; ++e) red[(tp * 16 + e) * 128 + (gt - 128)] = acc32[tp][e];
        }
        __syncthreads();
        GLOWTTS_TRACE_POINT_Z(3);
        if (grp == 0 && fh == 0) {
#pragma unroll
            for (int tp = 0; tp < TAPS; ++tp)
#pragma unroll
                for (int e = 0; e < 16; ++e) acc32[tp][e] += red[(tp * 16 + e) * 128 + gt];
            // C layout of the 32x32 MFMA: column (d channel) = lane & 31, row (x channel) = (e & 3) + 8 (e >> 2) + 4 (lane >> 5)
            const int lm = lane & 31, lr4 = 4 * (lane >> 5);
            if (k0 + 64 <= p.Cin && m0 + MR <= p.M) {
                const int lane_off = lr4 * p.M + lm;
#pragma unroll
                for (int tp = 0; tp < TAPS; ++tp)
#pragma unroll
                    for (int e = 0; e < 16; ++e) {
                        float *u = p.dwp + ((long)tp * p.Cin + k0 + 32 * kh + (e & 3) + 8 * (e >> 2)) * p.M + m0;
                        atomicAdd(u + lane_off, acc32[tp][e]);
                    }
            } else {
#pragma unroll
                for (int tp = 0; tp < TAPS; ++tp)
#pragma unroll
                    for (int e = 0; e < 16; ++e) {
                        const int k = k0 + 32 * kh + (e & 3) + 8 * (e >> 2) + lr4;
                        const int m = m0 + lm;
                        if (k < p.Cin && m < p.M) atomicAdd(p.dwp + ((long)tp * p.Cin + k) * p.M + m, acc32[tp][e]);
                    }
            }
        }
    } else {
        if (grp == 1) {
#pragma unroll
            for (int tp = 0; tp < TAPS; ++tp)
#pragma unroll
                for (int i = 0; i < MT; ++i)
#pragma unroll
                    for (int reg = 0; reg < 4; ++reg) red[((tp * MT + i) * 4 + reg) * 256 + gt] = acc[tp][i][reg];
        }
        __syncthreads();
        GLOWTTS_TRACE_POINT_Z(3);
        if (grp == 0) {
            const int lrow = lane & 15, lk = lane >> 4;
            if constexpr (NG == 2) {
#pragma unroll
                for (int tp = 0; tp < TAPS; ++tp)
#pragma unroll
                    for (int i = 0; i < MT; ++i)
#pragma unroll
                        for (int reg = 0; reg < 4; ++reg) acc[tp][i][reg] += red[((tp * MT + i) * 4 + reg) * 256 + gt];
            }
            if (k0 + 64 <= p.Cin && m0 + MR <= p.M) {
                const int lane_off = (wave * 16 + lk * 4) * p.M + lrow;      // per lane, once; the rest of an address is uniform
#pragma unroll
                for (int tp = 0; tp < TAPS; ++tp)
#pragma unroll
                    for (int reg = 0; reg < 4; ++reg) {
                        float *u = p.dwp + ((long)tp * p.Cin + k0 + reg) * p.M + m0;
#pragma unroll
                        for (int i = 0; i < MT; ++i) atomicAdd(u + i * 16 + lane_off, acc[tp][i][reg]);
                    }
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
        }
    }
    if (do_bias && tid < MR && m0 + tid < p.M) atomicAdd(p.dbias + m0 + tid, rowacc[tid]);
#ifdef GLOWTTS_TRACE
    GLOWTTS_TRACE_POINT_Z(4);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // (trace builds only: point 10 = the atomics have left)
    GLOWTTS_TRACE_POINT_Z(10);
#endif
}

static int compute_units() {
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

template <int NS, int TAPS, int MT, bool W32 = false, int NG = 2>
static int launch_wrw_tr(ConvWrwParams &p, hipStream_t s) {
    using G = WrwTrGeom<TAPS, MT>;
    constexpr size_t lds = G::lds_bytes(NS, W32, NG);
    static_assert(lds <= 160 * 1024, "one workgroup per CU");
    static LdsLimit attr_max_e;
    if (int rc_ = attr_max_e.ensure(reinterpret_cast<const void *>(&convwrw_tr_kernel<NS, TAPS, MT, W32, NG>), lds, "glowtts_conv_wrw (tr)")) return rc_;
    const int tiles = ((p.Cin + 63) / 64) * ((p.M + G::MR - 1) / G::MR);
    const int total = p.B * ((p.T + G::CF - 1) / G::CF);
    int splits = compute_units() / tiles;               // ONE workgroup (8 waves) per CU, one round
    // A batch of nbatch problems shares the compute units: ONE round of workgroups for the whole batch (4 problems x 18 tiles x 3
    // splits = 216 workgroups of 75 items each at config 2) instead of one round per problem (4 x 252 workgroups of 16 items):
    // nbatch times fewer prologues, group reductions and split-K atomics per problem (82 KB per workgroup: 20.7 -> 4.4 MB per
    // problem) for 16 % of the compute units left to the other streams; 15.08 -> 14.98 ms per step (tools/ab_flags.py
    // envs=GLOWTTS_WRW5_BSPLIT:0,GLOWTTS_WRW5_BSPLIT:1; two launches of two problems x 7 splits: 15.01).  Read at every launch.
    if (p.nbatch > 1 && knob(K_WRW5_BSPLIT) == 1 && compute_units() >= tiles * p.nbatch) {
        const int cus = knob(K_WRW5_CUS);          // -1: every compute unit; a smaller number leaves the rest to the backward's chain
        splits = ((cus > 0 && cus < compute_units()) ? cus : compute_units()) / (tiles * p.nbatch);
    }
    if (NG == 1) splits *= knob(K_WRW_TR_NG_SPLITS);      // (one group: workgroups per CU's worth of items; 1 = same grid)
    if (splits > (total + 1) / 2) splits = (total + 1) / 2;      // a workgroup wants an item for each of its two groups
    if (splits < 1) splits = 1;
    p.nb = (total + splits - 1) / splits;
    p.xs_pitch = knob(K_WRW_TR_PRIO);         // (tuning switch: 0 = the multiplying waves run at raised priority, 1 = nobody, 2 = the storing waves)
    dim3 grid(tiles, 1, ((total + p.nb - 1) / p.nb) * (p.nbatch > 0 ? p.nbatch : 1));
    hipLaunchKernelGGL((convwrw_tr_kernel<NS, TAPS, MT, W32, NG>), grid, dim3(256 * NG), lds, s, p);
    GLOWTTS_LAUNCH_CHECK("glowtts_conv_wrw (tr)");
}

// called by conv_wrw_split_dispatch for fp32 tensors: -1 = not handled here
int conv_wrw_tr_dispatch(ConvWrwParams &p, int ns, hipStream_t s) {
    const int tr = knob(K_WRW_TR);                    // -1 unset, 0 = off, 1 = 64 x 32 tiles in the 16x16x32 form
    if (tr == 0) return -1;
    if (p.M % 32 != 0 || (p.d2 && p.d_split % 32 != 0)) return -1;
    if (p.taps == 3) {        // the text encoder's FFN convolutions (768 <-> 192 channels, T_text frames)
        const bool no3 = knob(K_WRW_TR3) == 0;
        if (ns == 3 && !no3 && knob(K_WRW_TR3_MT) == 4 && p.M % 64 == 0) return launch_wrw_tr<3, 3, 4>(p, s);
        return (ns == 3 && !no3) ? launch_wrw_tr<3, 3, 2, true>(p, s) : -1;
    }
    // 1x1 convolutions stay on the frame-packed kernel: this one is staging-bound there (measured at B=32 / T'=400, 384 <- 192
    // channels: 45 us as 64x32 tiles, 34 us as 64x64, against 30.5 us)
    if (p.taps != 5) return -1;
    // default: 64 x 64 tiles (x staged once per 64 output channels: the loop runs at 86 % of the pipe), whose 20 MB of atomics
    // drain while the other stream's kernels run — alone it is no faster than the 64 x 32 form (61 us either way), in the step it
    // is (16.25 -> 16.00 ms, three alternating runs).  GLOWTTS_WRW_TR_MT=2 selects 64 x 32 tiles in the 32x32x16 form with the
    // storing waves at raised priority (A/B at B=32 / T'=400, production build: 65.7 us per back-to-back launch; 16x16x32 form
    // 66.3; either form with the MULTIPLYING waves raised 67-69), GLOWTTS_WRW_TR=1 their 16x16x32 form
    const bool mt2 = knob(K_WRW_TR_MT) == 2, w16 = tr == 1;
    if (ns == 3 && !mt2 && !w16 && p.M % 64 == 0 && (!p.d2 || p.d_split % 64 == 0))
        return knob(K_WRW_TR_NG) == 1 ? launch_wrw_tr<3, 5, 4, false, 1>(p, s) : launch_wrw_tr<3, 5, 4>(p, s);
    if (ns == 3 && !w16) return launch_wrw_tr<3, 5, 2, true>(p, s);
    if (ns == 3) return launch_wrw_tr<3, 5, 2>(p, s);
    if (ns == 2) return launch_wrw_tr<2, 5, 2>(p, s);
    if (ns == 1) return launch_wrw_tr<1, 5, 2>(p, s);
    return -1;
}

}  // namespace glowtts

#ifdef GLOWTTS_TRACE
extern "C" int glowtts_debug_trace_read_tr(unsigned long long *host, int n_words, int clear) {
    hipError_t e = hipMemcpyFromSymbol(host, HIP_SYMBOL(glowtts::g_trace), (size_t)n_words * 8);
    if (e != hipSuccess) return (int)e;
    if (clear) {
        static unsigned long long zeros[8192 * 16];
        e = hipMemcpyToSymbol(HIP_SYMBOL(glowtts::g_trace), zeros, sizeof(zeros));
    }
    return (int)e;
}
#endif
