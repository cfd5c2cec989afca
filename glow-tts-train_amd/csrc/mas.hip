// mas.hip — monotonic alignment search (MAS) for gfx950: one (T_text, T_mel) lattice per workgroup, the dynamic
// programme itself on ONE wavefront.
//
// Replaces glow_tts_train/monotonic_align/core.pyx:9-45 (maximum_path_each / maximum_path_c) and the
// D2H -> CPU -> H2D round trip of monotonic_align/__init__.py:11-21.  Bit-exact with the reference given the same
// fp32 `value`: one fp32 add per cell, max(a,b) == (v_prev > v_cur) ? v_prev : v_cur (core.c:2697-2703),
// max_neg_val = -1e9f.
//
// Mapping (design, not a translation — the reference walks the lattice cell by cell on one CPU thread):
//   * the recurrence is sequential in y (mel frames) and parallel in x (text tokens).  Wave 0 owns the running
//     column v[x, y-1] in REGISTERS: lane l holds the R consecutive rows x = l*R .. l*R+R-1, so the only
//     cross-lane traffic per column is ONE DPP wave-shift (v[x-1] for each lane's first row);
//   * `value` is (B, Tx, Ty) with y contiguous, i.e. a column step would read 4 bytes from Tx different rows.
//     Waves 1..3 therefore stage 64-column tiles through LDS with coalesced 256-byte row reads, transposing on
//     the LDS write (odd row pitch => conflict-free), double-buffered so staging tile k+1 overlaps the DP on k;
//   * the back-pointer of cell (x, y) is one bit; each lane packs 32 columns per row into a register and spills
//     one word per 32 columns to LDS; the backtrack is a single lane that jumps from row change to row change with
//     clz on those words (~Tx + Ty/32 steps) and records the first frame of every text row;
//   * the 0/1 path is written by all four waves with coalesced 16-byte stores (no pre-zeroing pass).
//
// HBM traffic (algorithmic): 4 B read per in-band cell + 4 B written per path cell.
#include "common.hpp"

namespace glowtts {

constexpr float kMasNeg = -1e9f;

#ifdef GLOWTTS_TRACE   // tuning builds only: phase timestamps per utterance (100 MHz wall clock)
__device__ unsigned long long g_mas_trace[1024 * 8];
#define MAS_TRACE(i) do { if (threadIdx.x == 0) { g_mas_trace[(blockIdx.x & 1023) * 8 + (i)] = wall_clock64(); if ((i) < 2) g_mas_trace[(blockIdx.x & 1023) * 8 + 5 + (i)] = __builtin_readcyclecounter(); } } while (0)
#else
#define MAS_TRACE(i) do { } while (0)
#endif

// v_mov_b32_dpp wave_shr:1 (control 0x138): lane i receives lane i-1's value, lane 0 keeps the `old` operand.

template <int R, bool GD>
__global__ __launch_bounds__(256) void mas_kernel(const float *__restrict__ value, float *__restrict__ path,
                                                  const int *__restrict__ t_xs, const int *__restrict__ t_ys,
                                                  int Tx, int Ty, int log2tc, int nblk32,
                                                  int *__restrict__ first_out, int *__restrict__ tok_out) {
    constexpr bool gdirs = GD;      // (a template parameter: one pointer that may be LDS or global turns every access into a
                                    //  flat instruction, and those make each s_waitcnt in the DP loop wait for everything)
    extern __shared__ __align__(16) unsigned char smem[];
    constexpr int ROWPAD = R * 64 + 1;
    const int TC = 1 << log2tc;
    float *tile = reinterpret_cast<float *>(smem);                                // [2][TC][ROWPAD]
    uint32_t *dirs = reinterpret_cast<uint32_t *>(tile + 2 * TC * ROWPAD);        // [R][nblk32][64] (LDS form)
    int *first = reinterpret_cast<int *>(dirs + (gdirs ? 0 : R * nblk32 * 64));   // [Tx + 1]: first frame of every text row
    // Long lattices (bit image > LDS: e.g. 500 tokens x 4 000 frames): the back-pointer bits go to this utterance's own
    // slice of the OUTPUT buffer instead — 1 bit per cell fits 32x over in the 4-byte-per-cell path, nobody reads the
    // path before the kernel ends, and the bits are dead (spans are in `first`) before the first path element is written.
    // Written and read back by wave 0 only; the read-back is volatile (device-scope load, never a stale L1 line).
    uint32_t *gd = reinterpret_cast<uint32_t *>(path + (size_t)blockIdx.x * Tx * Ty);

    const int b = blockIdx.x;
    int tx = t_xs[b], ty = t_ys[b];
    tx = tx < 0 ? 0 : (tx > Tx ? Tx : tx);
    ty = ty < 0 ? 0 : (ty > Ty ? Ty : ty);
    const float *val = value + (size_t)b * Tx * Ty;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int ntiles = (ty + TC - 1) >> log2tc;

    // ---- tile staging by waves 1..3: coalesced row segments -> transposed LDS image ------------------------
    auto stage = [&](int k) {
        const int y0 = k << log2tc;
        int xlo = tx + y0 - ty;            // lowest row that is in band for some column of this tile
        xlo = xlo < 0 ? 0 : xlo;
        int xhi = y0 + TC;                 // rows >= y+1 are above the band
        xhi = xhi > tx ? tx : xhi;
        const int rpi = 64 >> log2tc;      // rows per wave-instruction
        const int sub = lane >> log2tc, yl = lane & (TC - 1);
        const int y = y0 + yl;
        const bool yok = y < ty;
        float *dst = tile + ((k & 1) * TC + yl) * ROWPAD;
#pragma unroll 8
        for (int x = xlo + (wave - 1) * rpi + sub; x < xhi; x += 3 * rpi) {
            float t = yok ? val[(size_t)x * Ty + y] : 0.0f;
            dst[(x % R) * 64 + x / R] = t;
        }
    };

    // 16-byte variant (Ty % 4 == 0, aligned rows, TC >= 16): a lane fetches 4 consecutive frames of one row, 64 / (TC/4)
    // rows per wave-instruction, and EVERY load of the tile is issued before the first LDS write — one HBM round trip
    // per tile instead of one per 8 rows (the scalar loop above made the staging, not the DP, the critical path:
    // 160 rows / 3 waves / 8 in flight = 7 round trips x 13 tiles).
    const bool vec_ok = ((Ty & 3) == 0) && ((reinterpret_cast<uintptr_t>(value) & 15u) == 0) && (log2tc >= 4);
    auto stage4 = [&](int k) {
        constexpr int MAXL = 16;                       // loads in flight per lane (covers Tx <= 768 at TC = 64)
        const int y0 = k << log2tc;
        int xlo = tx + y0 - ty;
        xlo = xlo < 0 ? 0 : xlo;
        int xhi = y0 + TC;
        xhi = xhi > tx ? tx : xhi;
        const int qpr = TC >> 2;                       // float4 per tile row
        const int rpi = 64 / qpr;                      // rows per wave-instruction
        const int sub = lane / qpr, q = lane - sub * qpr;
        const int y = y0 + q * 4;
        float *dst = tile + ((k & 1) * TC + q * 4) * ROWPAD;
        const int xstart = xlo + (wave - 1) * rpi + sub, xstep = 3 * rpi;
        for (int xb = xstart; xb < xhi; xb += MAXL * xstep) {
            float4 t[MAXL];
#pragma unroll
            for (int i = 0; i < MAXL; ++i) {
                const int x = xb + i * xstep;
                t[i] = make_float4(0.f, 0.f, 0.f, 0.f);
                if (x < xhi && y < Ty) t[i] = *reinterpret_cast<const float4 *>(val + (size_t)x * Ty + y);
            }
#pragma unroll
            for (int i = 0; i < MAXL; ++i) {
                const int x = xb + i * xstep;
                if (x < xhi) {
                    float *d = dst + (x % R) * 64 + x / R;
                    d[0] = (y + 0 < ty) ? t[i].x : 0.0f;
                    d[ROWPAD] = (y + 1 < ty) ? t[i].y : 0.0f;
                    d[2 * ROWPAD] = (y + 2 < ty) ? t[i].z : 0.0f;
                    d[3 * ROWPAD] = (y + 3 < ty) ? t[i].w : 0.0f;
                }
            }
        }
    };
    auto stage_any = [&](int k) { if (vec_ok) stage4(k); else stage(k); };

    MAS_TRACE(0);
    if (wave != 0 && ntiles > 0) stage_any(0);

    // ---- forward DP on wave 0 ---------------------------------------------------------------------------------
    float v[R];
    uint32_t dw[R];
#pragma unroll
    for (int r = 0; r < R; ++r) { v[r] = 0.0f; dw[r] = 0u; }

    for (int k = 0; k < ntiles; ++k) {
        __syncthreads();
        if (wave != 0) {
            if (k + 1 < ntiles) stage_any(k + 1);
        } else {
            const int y0 = k << log2tc;
            int ncols = ty - y0;
            ncols = ncols > TC ? TC : ncols;
            const float *tbase = tile + (k & 1) * TC * ROWPAD;
            // Only in-band cells are ever read back: an in-band (x, y) takes v[x][y-1] and v[x-1][y-1], both in band (or the
            // x == y / x == 0 sentinels), and the backtrack never leaves the band.  So rows outside the band may carry
            // garbage values and bits, and the loop needs no band test at all — 8 vector instructions per row and column.
            // Four columns per half-trip, two register buffers: while one buffer's columns are worked on, the other is being
            // filled from LDS (an LDS read is ~130 cycles and wave 0 has nobody to hide behind — one column of look-ahead
            // left the wave waiting every column: 300 cycles per column, 117 us at Ty = 800).
            // Per cell 6 vector instructions: diagonal compare + select, compare, select, add, and ONE add-with-carry that
            // shifts the new direction bit into the row's word (newest column in bit 0; normalised by a shift + bit reverse
            // when the word is stored every 32 columns).
            constexpr int U = 4;
            float bufa[U][R], bufb[U][R];
            const float *trow = tbase + lane;
            auto fill = [&](float (&buf)[U][R], int yl) {          // columns yl .. yl + U - 1 of the tile
                if (yl < TC) {
#pragma unroll
                    for (int u = 0; u < U; ++u)
#pragma unroll
                        for (int r = 0; r < R; ++r) buf[u][r] = trow[(yl + u) * ROWPAD + r * 64];
                }
            };
            // Scheduling (one wave, nobody to hide behind): nothing on the column-to-column chain goes through a scalar
            // register written in the same column — the diagonal masks of a half-trip are computed ahead, the running value is
            // select(diag) -> v_max -> add, and the `vprev > vcur` compare feeds only the add-with-carry that shifts the
            // direction bit into the row's word.  (max(a, b) is the reference's `a > b ? a : b` for everything but the sign of a
            // zero, which no later compare or sum can tell apart.)  The forced step on the diagonal (core.pyx:34, index == y)
            // is OR-ed into a word when it is stored, not per cell.  The R rows of a lane are independent within a column:
            // their instructions are issued stage by stage, not row by row; a half-trip that lies inside the utterance runs
            // without per-column branches.
            auto flush = [&](int y) {                             // after column y: store the rows' words of its 32-column block
                const int sh = 31 - (y & 31), blk = y >> 5;
#pragma unroll
                for (int r = 0; r < R; ++r) {
                    uint32_t w = __brev(dw[r] << sh);             // column 32 blk + j of the word in bit j
                    const int x = lane * R + r;
                    if ((x >> 5) == blk) w |= 1u << (x & 31);     // x == y: the path must step (bits past y are never read)
                    if (gdirs) gd[(r * nblk32 + blk) * 64 + lane] = w;
                    else dirs[(r * nblk32 + blk) * 64 + lane] = w;
                    dw[r] = 0u;
                }
            };
            auto column = [&](const float (&cells)[R], unsigned long long (&dg)[R], int y) {
                // v[x-1] of this lane's first row: ONE DPP; lane 0 (x = 0) receives the sentinel of core.pyx:24-27
                const float edge = (y == 0) ? 0.0f : kMasNeg;
                float vp[R], vc[R], best[R];
                vp[0] = __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(edge), __float_as_int(v[R - 1]),
                                                                   0x138, 0xf, 0xf, false));
#pragma unroll
                for (int r = 1; r < R; ++r) vp[r] = v[r - 1];
#pragma unroll
                for (int r = 0; r < R; ++r)
                    asm("v_cndmask_b32 %0, %1, %2, %3" : "=v"(vc[r]) : "v"(v[r]), "v"(kMasNeg), "s"(dg[r]));
                unsigned long long take[R];                       // (each compare well ahead of the add-with-carry that reads its mask)
#pragma unroll
                for (int r = 0; r < R; ++r) take[r] = __builtin_amdgcn_fcmpf(vp[r], vc[r], 2);      // vprev > vcur
#pragma unroll
                for (int r = 0; r < R; ++r) asm volatile("v_max_f32 %0, %1, %2" : "=v"(best[r]) : "v"(vp[r]), "v"(vc[r]));
#pragma unroll
                for (int r = 0; r < R; ++r) v[r] = best[r] + cells[r];
#pragma unroll
                for (int r = 0; r < R; ++r) {
                    unsigned long long carry_out;
                    asm volatile("v_addc_co_u32 %0, %1, %2, %2, %3" : "=v"(dw[r]), "=s"(carry_out) : "v"(dw[r]), "s"(take[r]));
                }
            };
            auto work = [&](const float (&buf)[U][R], int yl) {
                unsigned long long dg[U][R];
#pragma unroll
                for (int u = 0; u < U; ++u)
#pragma unroll
                    for (int r = 0; r < R; ++r)
                        dg[u][r] = __builtin_amdgcn_uicmp((unsigned)(lane * R + r), (unsigned)(y0 + yl + u), 32);   // x == y
                const int ya = y0 + yl;
                if (ya + U <= ty) {                               // the whole half-trip is inside the utterance
#pragma unroll
                    for (int u = 0; u < U; ++u) column(buf[u], dg[u], ya + u);
                    const int y = ya + U - 1;                     // (ya is a multiple of 4: only the last column can end a block)
                    if ((y & 31) == 31 || y == ty - 1) flush(y);
                } else {
#pragma unroll
                    for (int u = 0; u < U; ++u) {
                        const int y = ya + u;
                        if (y < ty) {
                            column(buf[u], dg[u], y);
                            if ((y & 31) == 31 || y == ty - 1) flush(y);
                        }
                    }
                }
            };
            fill(bufa, 0);
            for (int yl = 0; yl < ncols; yl += 2 * U) {
                fill(bufb, yl + U);
                work(bufa, yl);
                if (yl + U < ncols) {
                    fill(bufa, yl + 2 * U);
                    work(bufb, yl + U);
                }
            }
        }
    }
    MAS_TRACE(1);
    __syncthreads();
    MAS_TRACE(2);

    // ---- backtrack (core.pyx:32-35).  The path is monotone: text row x owns the frames [first[x], first[x+1]).  One lane
    // walks it row change by row change — the next frame at which the path steps down is the highest set bit at or
    // below the current frame in the row's 32-frame bit word (clz), so the walk takes ~Tx + Ty/32 steps, not Ty; the
    // word of the row below is fetched while the current row is walked.
    for (int x = threadIdx.x; x <= Tx; x += 256) first[x] = x >= tx ? ty : 0;
    __syncthreads();
    // The walk runs on wave 0 with its state in SCALAR registers (a lone wave issues a vector instruction every ~10 cycles,
    // a scalar one every cycle or two): all 64 lanes read the same word (an LDS broadcast), v_readfirstlane hands it to the
    // scalar unit, and mask / clz / compare / address arithmetic are SALU work.
    if (__builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)) == 0 && tx > 0 && ty > 0) {
        auto word_of = [&](int row, int blk) -> uint32_t {
            if (row <= 0) return 0u;
            const int at = ((row % R) * nblk32 + blk) * 64 + row / R;
            const uint32_t w = gdirs ? *reinterpret_cast<volatile uint32_t *>(gd + at) : dirs[at];
            return (uint32_t)__builtin_amdgcn_readfirstlane((int)w);
        };
        // (fetching two rows ahead so that the LDS latency passes under two steps was measured SLOWER, 39 us against 30: the
        //  extra address arithmetic costs more vector issue slots than the latency it hides)
        int index = tx - 1, y = ty - 1, blk = y >> 5;
        uint32_t word = word_of(index, blk), below = word_of(index - 1, blk);
        while (index > 0 && y > 0) {
            uint32_t m = word & (0xffffffffu >> (31 - (y & 31)));
            if (blk == 0) m &= ~1u;                       // no step at frame 0 (core.pyx:34 needs y > 0 ... index - 1 at y - 1)
            if (m == 0) {                                 // the path stays on this row for the rest of the 32-frame block
                if (blk == 0) break;
                y = (blk << 5) - 1;
                --blk;
                word = word_of(index, blk);
                below = word_of(index - 1, blk);
                continue;
            }
            y = (blk << 5) + (31 - __clz(m));             // frame at which the path leaves row `index` downwards
            first[index] = y;                             // (every lane stores the same value)
            --index;
            --y;
            word = below;
            if ((y >> 5) != blk) {                        // stepped across a block boundary (y >= 0 here)
                blk = y >> 5;
                word = word_of(index, blk);
            }
            below = word_of(index - 1, blk);
        }
    }
    MAS_TRACE(3);
    __syncthreads();

    // ---- optional by-products of the search: the spans (first frame of every text row; first[Tx] = t_y) and the token of
    // every frame (-1 past the utterance) — what the expansion z_m = attn^T x_m and log(sum attn) need instead of the path
    if (first_out != nullptr)
        for (int x = threadIdx.x; x <= Tx; x += 256) first_out[(size_t)b * (Tx + 1) + x] = first[x];
    if (tok_out != nullptr && ((Ty & 3) != 0 || (reinterpret_cast<uintptr_t>(path) & 15u) != 0)) {
        for (int y = threadIdx.x; y < Ty; y += 256) {
            int t = -1;
            for (int x = 0; x < tx; ++x)
                if (y >= first[x] && y < first[x + 1]) t = x;
            tok_out[(size_t)b * Ty + y] = t;
        }
    }

    // ---- path write: all waves, coalesced -------------------------------------------------------------------------
    float *pb = path + (size_t)b * Tx * Ty;
    if ((Ty & 3) == 0 && ((reinterpret_cast<uintptr_t>(path) & 15u) == 0)) {
        // thread = one quad of frames: its 4 path rows are read from LDS ONCE, then every text row gets one 16-byte
        // store (consecutive threads -> consecutive addresses within a row)
        const int ty4 = Ty >> 2;
        for (int q = threadIdx.x; q < ty4; q += 256) {
            const int y = q << 2;
            float4 *dst = reinterpret_cast<float4 *>(pb) + q;
            int lo = first[0];
            int tk0 = -1, tk1 = -1, tk2 = -1, tk3 = -1;
#pragma unroll 4
            for (int x = 0; x < Tx; ++x) {
                const int hi = first[x + 1];               // same address in every lane: an LDS broadcast
                const bool i0 = y + 0 >= lo && y + 0 < hi, i1 = y + 1 >= lo && y + 1 < hi, i2 = y + 2 >= lo && y + 2 < hi,
                           i3 = y + 3 >= lo && y + 3 < hi;
                dst[(size_t)x * ty4] = make_float4(i0 ? 1.0f : 0.0f, i1 ? 1.0f : 0.0f, i2 ? 1.0f : 0.0f, i3 ? 1.0f : 0.0f);
                tk0 = i0 ? x : tk0; tk1 = i1 ? x : tk1; tk2 = i2 ? x : tk2; tk3 = i3 ? x : tk3;
                lo = hi;
            }
            if (tok_out != nullptr) *reinterpret_cast<int4 *>(tok_out + (size_t)b * Ty + y) = make_int4(tk0, tk1, tk2, tk3);
        }
    } else {
        const int n = Tx * Ty;
        for (int i = threadIdx.x; i < n; i += 256) {
            const int x = i / Ty;
            const int y = i - x * Ty;
            pb[i] = (y >= first[x] && y < first[x + 1]) ? 1.0f : 0.0f;
        }
    }
    MAS_TRACE(4);
}

// ---------------------------------------------------------------------------------------------------------------------
// Round 5: the dynamic programme on UP TO FOUR waves.  The single DP wave above spends ~200 cycles per lattice
// column whatever it does (a lone wave issues a vector instruction every ~10 cycles; 5 instructions per row and column, three
// rows per lane at 160 tokens) and three waves only feed it.  Here
//   * the ROWS are split over the waves: global lane L = 64 w + lane owns rows L R .. L R + R - 1 (R = 1 up to 256 tokens), so a
//     column costs a wave 2 + 5 R vector instructions instead of 1 + 5 ceil(Tx / 64);
//   * the recurrence needs v[x - 1][y - 1] across the wave boundary: wave w runs ONE 16-column slab behind wave w - 1 (a skewed
//     pipeline over slabs, one workgroup barrier per step: ceil(ty / 16) + waves - 1 steps), and after every column each wave
//     leaves its lanes' running values in an LDS ring (64 columns deep) from which its successor takes lane 63's — all 16 of a
//     slab with one burst of broadcast reads;
//   * nothing is staged: a lane reads the 16 cells of its own row straight from `value` (four 16-byte loads per slab and row,
//     the next slab's in flight while this one is worked on) — the transposing LDS image and the staging waves are gone;
//   * only in-band cells are ever read back (see the kernel above), so rows above the diagonal and ring slots never written
//     may hold anything;
//   * the 0/1 path is NOT written here when the caller wants the spans: mas_path_from_spans_kernel expands them on the whole
//     chip (one utterance's 512 KB took its single workgroup ~20 us), on a stream of the caller's choice.
// Same arithmetic per cell (select, compare, max, add: bit-exact with the reference), same 1-bit back-pointers, same backtrack.
template <int R>
__global__ __launch_bounds__(256) void mas_wave_kernel(const float *__restrict__ value, float *__restrict__ path,
                                                       const int *__restrict__ t_xs, const int *__restrict__ t_ys,
                                                       int Tx, int Ty, int nblk32, int nw,
                                                       int *__restrict__ first_out, int *__restrict__ tok_out) {
    extern __shared__ __align__(16) unsigned char smem[];
    constexpr int D = 16, LW = 256;
    uint32_t *dirs = reinterpret_cast<uint32_t *>(smem);                         // [R][nblk32][LW]
    float *ring = reinterpret_cast<float *>(dirs + R * nblk32 * LW);             // [4 waves][64 columns][64 lanes]
    int *first = reinterpret_cast<int *>(ring + 4 * 64 * 64);                    // [Tx + 1]

    const int b = blockIdx.x;
    int tx = t_xs[b], ty = t_ys[b];
    tx = tx < 0 ? 0 : (tx > Tx ? Tx : tx);
    ty = ty < 0 ? 0 : (ty > Ty ? Ty : ty);
    const float *val = value + (size_t)b * Tx * Ty;
    const int L = threadIdx.x, lane = L & 63;
    // the wave index as a SCALAR: everything derived from it (the wave's slab, the `whole slab inside the utterance` test) then
    // branches on the scalar unit instead of becoming per-column exec-mask sequences (DESIGN.md lesson 32b)
    const int w = __builtin_amdgcn_readfirstlane(L >> 6);
    const int nslab = (tx > 0 && ty > 0) ? (ty + D - 1) / D : 0;

    float v[R];
    uint32_t dw[R];
    const float *rowp[R];
#pragma unroll
    for (int r = 0; r < R; ++r) {
        v[r] = 0.0f;
        dw[r] = 0u;
        int x = L * R + r;
        x = x < Tx ? x : Tx - 1;                   // rows past the lattice read the last row: out of band, never read back
        rowp[r] = val + (size_t)x * Ty;
    }
    float4 ca[R][4], cb[R][4];
    auto load = [&](float4 (&c)[R][4], int s) {
        const int y0 = s * D;
#pragma unroll
        for (int r = 0; r < R; ++r)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                // UNCONDITIONAL loads (frames past the lattice re-read its last four: columns >= ty are never worked on) — behind a
                // branch per load the compiler can no longer count them and waits for vmcnt(0), the NEXT slab's loads included
                int yq = y0 + 4 * q;
                yq = yq < Ty - 4 ? yq : Ty - 4;                                                                // (Ty % 4 == 0, Ty >= 4)
                c[r][q] = *reinterpret_cast<const float4 *>(rowp[r] + yq);
            }
    };
    auto flush = [&](int y) {                      // after column y: store the rows' words of its 32-column block
        const int sh = 31 - (y & 31), blk = y >> 5;
#pragma unroll
        for (int r = 0; r < R; ++r) {
            uint32_t wd = __brev(dw[r] << sh);     // column 32 blk + j of the word in bit j
            const int x = L * R + r;
            if ((x >> 5) == blk) wd |= 1u << (x & 31);      // x == y: the path must step (bits past y are never read)
            if (blk == 0) wd &= ~1u;                         // no step at frame 0 (core.pyx:34 needs y > 0): the walk need not test it
            dirs[(r * nblk32 + blk) * LW + L] = wd;
            dw[r] = 0u;
        }
    };
    // DIAG: the slab holds cells with x == y for this wave's rows (their `stay` predecessor lies above the diagonal: max_neg_val,
    // core.pyx:21-22).  A slab whose every column lies beyond the wave's last row has none — the select and its compare go
    auto column = [&](const float (&cells)[R], const unsigned long long (&dg)[R], float edge, const bool diag) {
        float vp[R], vc[R], best[R];
        vp[0] = __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(edge), __float_as_int(v[R - 1]), 0x138, 0xf, 0xf, false));
#pragma unroll
        for (int r = 1; r < R; ++r) vp[r] = v[r - 1];
#pragma unroll
        for (int r = 0; r < R; ++r) {
            // (`diag` is a constant at every call site of this inlined lambda; a generic lambda cannot hold the asm operands)
            if (diag) asm("v_cndmask_b32 %0, %1, %2, %3" : "=v"(vc[r]) : "v"(v[r]), "v"(kMasNeg), "s"(dg[r]));
            else vc[r] = v[r];
        }
        unsigned long long take[R];
#pragma unroll
        for (int r = 0; r < R; ++r) take[r] = __builtin_amdgcn_fcmpf(vp[r], vc[r], 2);      // vprev > vcur
#pragma unroll
        for (int r = 0; r < R; ++r) asm volatile("v_max_f32 %0, %1, %2" : "=v"(best[r]) : "v"(vp[r]), "v"(vc[r]));
#pragma unroll
        for (int r = 0; r < R; ++r) v[r] = best[r] + cells[r];
#pragma unroll
        for (int r = 0; r < R; ++r) {
            unsigned long long carry_out;
            asm volatile("v_addc_co_u32 %0, %1, %2, %2, %3" : "=v"(dw[r]), "=s"(carry_out) : "v"(dw[r]), "s"(take[r]));
        }
    };
    float *my_ring = ring + w * 64 * 64 + lane;      // (the last wave's ring is never read: an unconditional store beats a branch per column)
    const float *prev_ring = ring + (w > 0 ? w - 1 : 0) * 64 * 64 + 63;
    auto slab = [&](const float4 (&c)[R][4], int s) {
        const int y0 = s * D;
        if (y0 + D - 1 < 64 * R * w) return;       // every cell of the slab lies above the diagonal for this wave's rows: nothing is read back
        float bnd[D];                              // v[64 w R - 1][y - 1] for the slab's columns (wave 0: the sentinels of core.pyx:24-27)
        if (w > 0) {
#pragma unroll
            for (int u = 0; u < D; ++u) bnd[u] = prev_ring[((y0 + u) & 63) * 64];
        } else {
#pragma unroll
            for (int u = 0; u < D; ++u) bnd[u] = kMasNeg;
            if (s == 0) bnd[0] = 0.0f;
        }
        float *rb = my_ring + (y0 & 63) * 64;      // columns y0 + 1 .. y0 + 15 of the ring: immediate offsets; y0 + 16 may wrap
        auto whole = [&](const bool diag) {        // the whole slab lies inside the utterance: no per-column test
#pragma unroll
            for (int g = 0; g < D / 4; ++g) {
                unsigned long long dg[4][R];
#pragma unroll
                for (int u = 0; u < 4; ++u)
#pragma unroll
                    for (int r = 0; r < R; ++r)
                        dg[u][r] = diag ? __builtin_amdgcn_uicmp((unsigned)(L * R + r), (unsigned)(y0 + 4 * g + u), 32) : 0ull;
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    float cells[R];
#pragma unroll
                    for (int r = 0; r < R; ++r) cells[r] = u == 0 ? c[r][g].x : u == 1 ? c[r][g].y : u == 2 ? c[r][g].z : c[r][g].w;
                    column(cells, dg[u], bnd[4 * g + u], diag);
                    if (4 * g + u < D - 1) rb[(4 * g + u + 1) * 64] = v[R - 1];
                    else my_ring[((y0 + D) & 63) * 64] = v[R - 1];
                }
            }
            const int y = y0 + D - 1;              // (y0 is a multiple of 16: only the slab's last column can end a block)
            if ((y & 31) == 31 || y == ty - 1) flush(y);
        };
        if (y0 + D <= ty) {
            if (y0 >= 64 * R * (w + 1)) whole(false);
            else whole(true);
        } else {                                   // the utterance's last, partial slab
#pragma unroll
            for (int g = 0; g < D / 4; ++g)
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const int y = y0 + 4 * g + u;
                    if (y < ty) {
                        unsigned long long dg[R];
                        float cells[R];
#pragma unroll
                        for (int r = 0; r < R; ++r) {
                            dg[r] = __builtin_amdgcn_uicmp((unsigned)(L * R + r), (unsigned)y, 32);
                            cells[r] = u == 0 ? c[r][g].x : u == 1 ? c[r][g].y : u == 2 ? c[r][g].z : c[r][g].w;
                        }
                        column(cells, dg, bnd[4 * g + u], true);
                        my_ring[((y + 1) & 63) * 64] = v[R - 1];
                        if ((y & 31) == 31 || y == ty - 1) flush(y);
                    }
                }
        }
    };

    // Step t: wave w works on slab t - w out of buffer (t & 1) while the next slab's cells load into the other one — the roles of
    // the two buffers are tied to the parity of the STEP, so an unrolled pair of steps has them fixed (tied to the slab's parity the
    // compiler merged the two cases with a register copy of a whole buffer and a vmcnt(0) per step).
    const int nsteps = nslab > 0 ? nslab + nw - 1 : 0;
    // ring slots and direction words of a step are visible to the next through an LDS-ONLY barrier: __syncthreads() would also wait for
    // the cell loads just issued (vmcnt(0)), i.e. put a trip to memory into every one of the ~52 steps
    auto lds_barrier = [&]() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); };
    auto step = [&](int t, float4 (&cur)[R][4], float4 (&nxt)[R][4]) {
        const int sl = t - w;
        if (w < nw && sl >= 0 && sl < nslab) {
            if (sl + 1 < nslab) load(nxt, sl + 1);
            slab(cur, sl);
        }
    };
    MAS_TRACE(0);
    if (w < nw && nslab > 0) {                     // slab 0 is worked on in step w
        if (w & 1) load(cb, 0);
        else load(ca, 0);
    }
    for (int t = 0; t < nsteps; t += 2) {
        step(t, ca, cb);
        lds_barrier();
        if (t + 1 < nsteps) {
            step(t + 1, cb, ca);
            lds_barrier();
        }
    }
    MAS_TRACE(1);
    __syncthreads();
    MAS_TRACE(2);

    // ---- backtrack (core.pyx:32-35) with the direction words in REGISTERS.  mas_kernel's walk fetches a word from LDS, hands it
    // to the scalar unit and decides — ~390 cycles per text row, 30 us at 160 tokens.  Here the words of a 32-frame block sit where
    // the DP left them: lane L of wave w holds those of its own rows, so the wave that owns a row reads any of them with
    // v_readlane (a scalar lane index, no memory) and the walk passes from wave to wave as it descends through the rows
    // (nw - 1 hand-overs through LDS).  The next block's words are loaded while the current one is walked; the spans collect in a
    // register per lane and are stored once at the end.
    int fv[R];
#pragma unroll
    for (int r = 0; r < R; ++r) fv[r] = (L * R + r) >= tx ? ty : 0;
    int *hand = reinterpret_cast<int *>(ring);           // (index, y) between waves; the ring is dead after the DP's last barrier
    if (L == 0) { hand[0] = tx - 1; hand[1] = ty - 1; }
    __syncthreads();
    for (int j = nw - 1; j >= 0; --j) {
        if (w == j && tx > 0 && ty > 0) {
            int index = __builtin_amdgcn_readfirstlane(hand[0]), y = __builtin_amdgcn_readfirstlane(hand[1]);
            const int base = 64 * R * j;                 // this wave's first row
            if (index >= base && index > 0 && y > 0) {
                int blk = y >> 5;
                uint32_t wc[R], wn[R];
#pragma unroll
                for (int r = 0; r < R; ++r) {
                    wc[r] = dirs[(r * nblk32 + blk) * LW + L];
                    wn[r] = blk > 0 ? dirs[(r * nblk32 + blk - 1) * LW + L] : 0u;
                }
                // per text row: the row's word from its owner lane, the highest direction bit at or below frame y (shift the
                // frames above y out, count leading zeros), the span's first frame into the owner lane's register.  ~20 instructions,
                // nearly all scalar, per row; frame 0's bit is cleared in the stored words, so y >= 0 needs no test of its own.
                int li = index - base;
                const int li_min = base == 0 ? 1 : 0;    // (row 0 is never left)
                while (li >= li_min) {
                    uint32_t word;
                    if (R == 1) word = (uint32_t)__builtin_amdgcn_readlane((int)wc[0], li);
                    else {
                        const uint32_t w0 = (uint32_t)__builtin_amdgcn_readlane((int)wc[0], li >> 1);
                        const uint32_t w1 = (uint32_t)__builtin_amdgcn_readlane((int)wc[R - 1], li >> 1);
                        word = (li & 1) ? w1 : w0;
                    }
                    const uint32_t m = word << ((~y) & 31);       // frame y in bit 31, the frames below it after it
                    bool cross;
                    if (m == 0) {                        // the path stays on this row for the rest of the 32-frame block
                        if (blk == 0) { li = -base; break; }      // ... down to frame 0: index 0 ends every wave's walk
                        y = (y & ~31) - 1;
                        cross = true;
                    } else {
                        const int ys = y - __clz(m);              // frame at which the path leaves this row downwards (>= 1)
                        if (R == 1) fv[0] = (lane == li) ? ys : fv[0];
                        else if (li & 1) fv[R - 1] = (lane == (li >> 1)) ? ys : fv[R - 1];
                        else fv[0] = (lane == (li >> 1)) ? ys : fv[0];
                        --li;
                        y = ys - 1;
                        cross = (ys & 31) == 0;
                    }
                    if (cross) {                         // into the block below: its words are there, fetch the next
                        --blk;
#pragma unroll
                        for (int r = 0; r < R; ++r) {
                            wc[r] = wn[r];
                            wn[r] = blk > 0 ? dirs[(r * nblk32 + blk - 1) * LW + L] : 0u;
                        }
                    }
                }
                index = base + li;
            }
            if (lane == 0) { hand[0] = index; hand[1] = y; }
        }
        __syncthreads();
    }
#pragma unroll
    for (int r = 0; r < R; ++r)
        if (L * R + r <= Tx) first[L * R + r] = fv[r];
    if (L == 0) first[Tx] = ty;                          // (row Tx itself when Tx = 256 R)
    MAS_TRACE(3);
    __syncthreads();

    if (first_out != nullptr)
        for (int x = threadIdx.x; x <= Tx; x += 256) first_out[(size_t)b * (Tx + 1) + x] = first[x];
    if (tok_out != nullptr) {                      // frame -> token: every text row writes its span, frames past the utterance -1
        for (int y = ty + (int)threadIdx.x; y < Ty; y += 256) tok_out[(size_t)b * Ty + y] = -1;
        for (int x = threadIdx.x; x < tx; x += 256) {
            const int lo = first[x], hi = first[x + 1];
            for (int y = lo; y < hi; ++y) tok_out[(size_t)b * Ty + y] = x;
        }
    }
    if (path != nullptr) {                         // (callers without a span table: the path from this workgroup, 16-byte stores)
        float *pb = path + (size_t)b * Tx * Ty;
        const int ty4 = Ty >> 2;
        for (int q = threadIdx.x; q < ty4; q += 256) {
            const int y = q << 2;
            float4 *dst = reinterpret_cast<float4 *>(pb) + q;
            int lo = first[0];
#pragma unroll 4
            for (int x = 0; x < Tx; ++x) {
                const int hi = first[x + 1];
                dst[(size_t)x * ty4] = make_float4((y + 0 >= lo && y + 0 < hi) ? 1.0f : 0.0f, (y + 1 >= lo && y + 1 < hi) ? 1.0f : 0.0f,
                                                   (y + 2 >= lo && y + 2 < hi) ? 1.0f : 0.0f, (y + 3 >= lo && y + 3 < hi) ? 1.0f : 0.0f);
                lo = hi;
            }
        }
    }
    MAS_TRACE(4);
}

// path[b][x][y] = 1 for first[b][x] <= y < first[b][x + 1], else 0: a workgroup writes 8 text rows of one utterance with 16-byte
// stores (Ty % 4 == 0, 16-byte aligned path)
__global__ __launch_bounds__(256) void mas_path_from_spans_kernel(const int *__restrict__ first, float *__restrict__ path, int Tx, int Ty) {
    constexpr int RW = 8;
    __shared__ int span[RW + 1];
    const int b = blockIdx.y, x0 = blockIdx.x * RW;
    if (threadIdx.x <= RW) span[threadIdx.x] = first[(size_t)b * (Tx + 1) + min(x0 + (int)threadIdx.x, Tx)];
    __syncthreads();
    const int ty4 = Ty >> 2;
    const int rows = min(RW, Tx - x0);
    float4 *dst = reinterpret_cast<float4 *>(path + ((size_t)b * Tx + x0) * Ty);
    for (int i = threadIdx.x; i < rows * ty4; i += 256) {
        const int r = i / ty4, y = (i - r * ty4) << 2;
        const int lo = span[r], hi = span[r + 1];
        dst[i] = make_float4((y + 0 >= lo && y + 0 < hi) ? 1.0f : 0.0f, (y + 1 >= lo && y + 1 < hi) ? 1.0f : 0.0f,
                             (y + 2 >= lo && y + 2 < hi) ? 1.0f : 0.0f, (y + 3 >= lo && y + 3 < hi) ? 1.0f : 0.0f);
    }
}

// LDS the multi-wave kernel needs for a (Tx, Ty) lattice with R rows per lane; 0 = not eligible (shape / size)
static size_t mas_wave_lds(int Tx, int Ty, int R) {
    if (Tx < 1 || Ty < 4 || (Ty & 3) != 0 || Tx > 256 * R) return 0;
    const size_t nblk32 = (size_t)(Ty + 31) / 32;
    const size_t bytes = (size_t)R * nblk32 * 256 * 4 + 4 * 64 * 64 * 4 + (((size_t)(Tx + 1) * 4 + 15) & ~(size_t)15);
    return bytes <= 150 * 1024 ? bytes : 0;
}
static int mas_wave_rows(int Tx, int Ty) {          // rows per lane of the multi-wave kernel, 0 = use mas_kernel
    if (knob(K_MAS_WAVES) == 0) return 0;
    for (int R = 1; R <= 2; ++R)
        if (mas_wave_lds(Tx, Ty, R)) return R;
    return 0;
}

template <int R>
static int launch_mas_wave(const float *value, float *path, const int32_t *t_x, const int32_t *t_y, int B, int Tx, int Ty,
                           int *first_out, int *tok_out, hipStream_t stream) {
    const size_t bytes = mas_wave_lds(Tx, Ty, R);
    const int nw = (Tx + 64 * R - 1) / (64 * R);
    static LdsLimit limit;
    if (int rc_ = limit.ensure(reinterpret_cast<const void *>(&mas_wave_kernel<R>), bytes, "glowtts_mas_path")) return rc_;
    hipLaunchKernelGGL((mas_wave_kernel<R>), dim3(B), dim3(256), bytes, stream, value, path, t_x, t_y, Tx, Ty, (Ty + 31) / 32, nw,
                       first_out, tok_out);
    GLOWTTS_LAUNCH_CHECK("glowtts_mas_path");
}

template <int R>
static int launch_mas(const float *value, float *path, const int32_t *t_x, const int32_t *t_y, int B, int Tx,
                      int Ty, int *first_out, int *tok_out, hipStream_t stream) {
    const int nblk32 = (Ty + 31) / 32;
    const size_t budget = 150 * 1024;
    const size_t first_b = (((size_t)(Tx + 1) * 4 + 15) & ~(size_t)15);
    const size_t dirs_b = (size_t)R * nblk32 * 64 * 4;
    auto tile_b = [](int l2) { return (size_t)2 * (1 << l2) * (R * 64 + 1) * 4; };
    // back-pointer bits in LDS when they leave room for at least 16-column tiles, else in the output buffer (see kernel)
    const int gdirs = (dirs_b + first_b + tile_b(4) > budget) ? 1 : 0;
    const size_t fixed = first_b + (gdirs ? 0 : dirs_b);
    int log2tc = 6;
    while (log2tc > 3 && fixed + tile_b(log2tc) > budget) --log2tc;
    const size_t bytes = fixed + tile_b(log2tc);
    GLOWTTS_CHECK_ARG(bytes <= 160 * 1024, "glowtts_mas_path: lattice %dx%d needs %zu B of LDS (> 160 KiB)", Tx, Ty,
                      bytes);
    if (gdirs) {
        static LdsLimit limit_g;
        if (int rc_ = limit_g.ensure(reinterpret_cast<const void *>(&mas_kernel<R, true>), bytes, "glowtts_mas_path")) return rc_;
        hipLaunchKernelGGL((mas_kernel<R, true>), dim3(B), dim3(256), bytes, stream, value, path, t_x, t_y, Tx, Ty, log2tc,
                           nblk32, first_out, tok_out);
    } else {
        static LdsLimit limit;   // per device: raised only when a launch needs more than any earlier one
        if (int rc_ = limit.ensure(reinterpret_cast<const void *>(&mas_kernel<R, false>), bytes, "glowtts_mas_path")) return rc_;
        hipLaunchKernelGGL((mas_kernel<R, false>), dim3(B), dim3(256), bytes, stream, value, path, t_x, t_y, Tx, Ty, log2tc,
                           nblk32, first_out, tok_out);
    }
    GLOWTTS_LAUNCH_CHECK("glowtts_mas_path");
}

}  // namespace glowtts

extern "C" int glowtts_mas_spans_supported(int Tx, int Ty) { return glowtts::mas_wave_rows(Tx, Ty) != 0 ? 1 : 0; }

extern "C" int glowtts_mas_path_from_spans(const int32_t *first, float *path, int B, int Tx, int Ty, glowtts_stream_t stream) {
    using namespace glowtts;
    GLOWTTS_CHECK_ARG(first && path, "glowtts_mas_path_from_spans: null pointer");
    GLOWTTS_CHECK_ARG(B >= 0 && Tx >= 0 && Ty >= 0 && (Ty & 3) == 0 && aligned16(path),
                      "glowtts_mas_path_from_spans: needs Ty %% 4 == 0 and a 16-byte aligned path");
    if (B == 0 || Tx == 0 || Ty == 0) return 0;
    hipLaunchKernelGGL(mas_path_from_spans_kernel, dim3((Tx + 7) / 8, B), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), first, path,
                       Tx, Ty);
    GLOWTTS_LAUNCH_CHECK("glowtts_mas_path_from_spans");
}

extern "C" int glowtts_mas_path_spans(const float *value, float *path, int32_t *first, int32_t *tok, const int32_t *t_x,
                                      const int32_t *t_y, int B, int Tx, int Ty, glowtts_stream_t stream) {
    using namespace glowtts;
    GLOWTTS_CHECK_ARG(value && t_x && t_y, "glowtts_mas_path: null pointer");
    GLOWTTS_CHECK_ARG(B >= 0 && Tx >= 0 && Ty >= 0, "glowtts_mas_path: negative size");
    if (B == 0 || Tx == 0 || Ty == 0) return 0;
    GLOWTTS_CHECK_ARG(Tx <= 2048, "glowtts_mas_path: Tx=%d exceeds the 2048-token limit of this build", Tx);
    GLOWTTS_CHECK_ARG((long)Tx * Ty < (1L << 31), "glowtts_mas_path: lattice too large");
    GLOWTTS_CHECK_ARG(!tok || (Ty & 3) != 0 || aligned16(tok), "glowtts_mas_path: tok must be 16-byte aligned");
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    // the multi-wave search (mas_wave_kernel): spans from the search, the path from the spans on the whole chip
    const int rw = (aligned16(value) && (path == nullptr || aligned16(path))) ? mas_wave_rows(Tx, Ty) : 0;
    GLOWTTS_CHECK_ARG(path != nullptr || (first != nullptr && rw != 0),
                      "glowtts_mas_path_spans: path may be NULL only with a span table and a lattice glowtts_mas_spans_supported() accepts");
    if (rw != 0) {
        float *in_kernel = first ? nullptr : path;      // no span table to expand from: the search's workgroups write the path
        const int rc = rw == 1 ? launch_mas_wave<1>(value, in_kernel, t_x, t_y, B, Tx, Ty, first, tok, s)
                               : launch_mas_wave<2>(value, in_kernel, t_x, t_y, B, Tx, Ty, first, tok, s);
        if (rc != 0 || first == nullptr || path == nullptr) return rc;
        return glowtts_mas_path_from_spans(first, path, B, Tx, Ty, stream);
    }
    const int r = (Tx + 63) / 64;
#define GLOWTTS_MAS(R) return launch_mas<R>(value, path, t_x, t_y, B, Tx, Ty, first, tok, s)
    switch (r) {
        case 1: GLOWTTS_MAS(1);
        case 2: GLOWTTS_MAS(2);
        case 3: GLOWTTS_MAS(3);
        case 4: GLOWTTS_MAS(4);
        case 5: GLOWTTS_MAS(5);
        case 6: GLOWTTS_MAS(6);
        case 7: case 8: GLOWTTS_MAS(8);
        default: break;
    }
    // beyond 512 tokens (no realistic utterance; the reference's Cython loop has no limit, so neither fails here)
    if (r <= 12) GLOWTTS_MAS(12);
    if (r <= 16) GLOWTTS_MAS(16);
    if (r <= 24) GLOWTTS_MAS(24);
    GLOWTTS_MAS(32);
#undef GLOWTTS_MAS
}

extern "C" int glowtts_mas_path(const float *value, float *path, const int32_t *t_x, const int32_t *t_y, int B,
                                int Tx, int Ty, glowtts_stream_t stream) {
    return glowtts_mas_path_spans(value, path, nullptr, nullptr, t_x, t_y, B, Tx, Ty, stream);
}

#ifdef GLOWTTS_TRACE
extern "C" int glowtts_debug_mas_trace_read(unsigned long long *host, int n_words) {
    return (int)hipMemcpyFromSymbol(host, HIP_SYMBOL(glowtts::g_mas_trace), (size_t)n_words * 8);
}
#endif
