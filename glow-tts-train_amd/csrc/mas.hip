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
//     one word per 32 columns to LDS, so the backtrack is a single lane chasing bits in LDS and only re-reads a
//     word when its row or 32-column block changes;
//   * the 0/1 path is written by all four waves with coalesced 16-byte stores (no pre-zeroing pass).
//
// HBM traffic (algorithmic): 4 B read per in-band cell + 4 B written per path cell.
#include "common.hpp"

namespace glowtts {

constexpr float kMasNeg = -1e9f;

__device__ __forceinline__ float dpp_wave_shr1(float v) {
    // v_mov_b32_dpp wave_shr:1 — lane i receives lane i-1's value (lane 0 keeps `old` = 0).
    int r = __builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x138, 0xf, 0xf, false);
    return __int_as_float(r);
}

template <int R>
__global__ __launch_bounds__(256) void mas_kernel(const float *__restrict__ value, float *__restrict__ path,
                                                  const int *__restrict__ t_xs, const int *__restrict__ t_ys,
                                                  int Tx, int Ty, int log2tc, int nblk32) {
    extern __shared__ __align__(16) unsigned char smem[];
    constexpr int ROWPAD = R * 64 + 1;
    const int TC = 1 << log2tc;
    float *tile = reinterpret_cast<float *>(smem);                                // [2][TC][ROWPAD]
    uint32_t *dirs = reinterpret_cast<uint32_t *>(tile + 2 * TC * ROWPAD);        // [R][nblk32][64]
    short *idx = reinterpret_cast<short *>(dirs + R * nblk32 * 64);               // [Ty]

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

    if (wave != 0 && ntiles > 0) stage(0);

    // ---- forward DP on wave 0 ---------------------------------------------------------------------------------
    float v[R];
    uint32_t dw[R];
#pragma unroll
    for (int r = 0; r < R; ++r) { v[r] = 0.0f; dw[r] = 0u; }
    // self-check of the DPP shift direction (wave-uniform); fall back to ds_bpermute if it is not "from lane-1"
    const int probe = __builtin_amdgcn_update_dpp(0, lane, 0x138, 0xf, 0xf, false);
    const bool dpp_ok = __all((lane == 0) || (probe == lane - 1));

    for (int k = 0; k < ntiles; ++k) {
        __syncthreads();
        if (wave != 0) {
            if (k + 1 < ntiles) stage(k + 1);
        } else {
            const int y0 = k << log2tc;
            int ncols = ty - y0;
            ncols = ncols > TC ? TC : ncols;
            const float *tbase = tile + (k & 1) * TC * ROWPAD;
            for (int yl = 0; yl < ncols; ++yl) {
                const int y = y0 + yl;
                int lo = tx + y - ty;
                lo = lo < 0 ? 0 : lo;
                int hi = y + 1;
                hi = hi > tx ? tx : hi;
                const float *trow = tbase + yl * ROWPAD;
                float cell[R];
#pragma unroll
                for (int r = 0; r < R; ++r) cell[r] = trow[r * 64 + lane];
                const float up = dpp_ok ? dpp_wave_shr1(v[R - 1]) : __shfl_up(v[R - 1], 1, 64);
#pragma unroll
                for (int r = R - 1; r >= 0; --r) {
                    const int x = lane * R + r;
                    float vprev = (r == 0) ? up : v[r > 0 ? r - 1 : 0];
                    if (x == 0) vprev = (y == 0) ? 0.0f : kMasNeg;
                    const float vcur = (x == y) ? kMasNeg : v[r];
                    const bool take_prev = vprev > vcur;
                    const float nv = (take_prev ? vprev : vcur) + cell[r];
                    const bool inb = (x >= lo) && (x < hi);
                    const bool move = inb && (x != 0) && (y > 0) && ((x == y) || take_prev);
                    v[r] = inb ? nv : v[r];
                    dw[r] |= (move ? 1u : 0u) << (y & 31);
                }
                if ((y & 31) == 31 || y == ty - 1) {
#pragma unroll
                    for (int r = 0; r < R; ++r) {
                        dirs[(r * nblk32 + (y >> 5)) * 64 + lane] = dw[r];
                        dw[r] = 0u;
                    }
                }
            }
        }
    }
    __syncthreads();

    // ---- backtrack: one lane chases the back-pointer bits (core.pyx:32-35) -----------------------------------
    if (threadIdx.x == 0) {
        int index = tx - 1;
        int cblk = -1, cidx = -1;
        uint32_t word = 0u;
        for (int y = ty - 1; y >= 0; --y) {
            idx[y] = (short)index;
            if (index > 0 && y > 0) {
                const int blk = y >> 5;
                if (blk != cblk || index != cidx) {
                    word = dirs[((index % R) * nblk32 + blk) * 64 + index / R];
                    cblk = blk;
                    cidx = index;
                }
                index -= (int)((word >> (y & 31)) & 1u);
            }
        }
    }
    __syncthreads();

    // ---- path write: all waves, coalesced -------------------------------------------------------------------------
    float *pb = path + (size_t)b * Tx * Ty;
    if ((Ty & 3) == 0 && ((reinterpret_cast<uintptr_t>(path) & 15u) == 0)) {
        const int ty4 = Ty >> 2;
        const int n4 = Tx * ty4;
        for (int i = threadIdx.x; i < n4; i += 256) {
            const int x = i / ty4;
            const int y = (i - x * ty4) << 2;
            float4 o;
            o.x = (y + 0 < ty && idx[y + 0] == x) ? 1.0f : 0.0f;
            o.y = (y + 1 < ty && idx[y + 1] == x) ? 1.0f : 0.0f;
            o.z = (y + 2 < ty && idx[y + 2] == x) ? 1.0f : 0.0f;
            o.w = (y + 3 < ty && idx[y + 3] == x) ? 1.0f : 0.0f;
            reinterpret_cast<float4 *>(pb)[i] = o;
        }
    } else {
        const int n = Tx * Ty;
        for (int i = threadIdx.x; i < n; i += 256) {
            const int x = i / Ty;
            const int y = i - x * Ty;
            pb[i] = (y < ty && idx[y] == x) ? 1.0f : 0.0f;
        }
    }
}

template <int R>
static int launch_mas(const float *value, float *path, const int32_t *t_x, const int32_t *t_y, int B, int Tx,
                      int Ty, hipStream_t stream) {
    const int nblk32 = (Ty + 31) / 32;
    const size_t fixed = (size_t)R * nblk32 * 64 * 4 + (((size_t)Ty * 2 + 15) & ~(size_t)15);
    const size_t budget = 150 * 1024;
    int log2tc = 6;
    while (log2tc > 3 && fixed + (size_t)2 * (1 << log2tc) * (R * 64 + 1) * 4 > budget) --log2tc;
    const size_t bytes = fixed + (size_t)2 * (1 << log2tc) * (R * 64 + 1) * 4;
    GLOWTTS_CHECK_ARG(bytes <= 160 * 1024, "glowtts_mas_path: lattice %dx%d needs %zu B of LDS (> 160 KiB)", Tx, Ty,
                      bytes);
    static size_t attr_max_e = 0;   // raise the dynamic-LDS limit only when a larger size is needed (not per launch)
    if ((size_t)bytes > attr_max_e) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(&mas_kernel<R>),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
        if (e != hipSuccess) {
        set_error("glowtts_mas_path: cannot reserve %zu B LDS: %s", bytes, hipGetErrorString(e));
        return (int)e;
    }
        attr_max_e = (size_t)bytes;
    }
    hipLaunchKernelGGL(mas_kernel<R>, dim3(B), dim3(256), bytes, stream, value, path, t_x, t_y, Tx, Ty, log2tc,
                       nblk32);
    GLOWTTS_LAUNCH_CHECK("glowtts_mas_path");
}

}  // namespace glowtts

extern "C" int glowtts_mas_path(const float *value, float *path, const int32_t *t_x, const int32_t *t_y, int B,
                                int Tx, int Ty, glowtts_stream_t stream) {
    using namespace glowtts;
    GLOWTTS_CHECK_ARG(value && path && t_x && t_y, "glowtts_mas_path: null pointer");
    GLOWTTS_CHECK_ARG(B >= 0 && Tx >= 0 && Ty >= 0, "glowtts_mas_path: negative size");
    if (B == 0 || Tx == 0 || Ty == 0) return 0;
    GLOWTTS_CHECK_ARG(Tx <= 512, "glowtts_mas_path: Tx=%d exceeds the 512-token limit of this build", Tx);
    GLOWTTS_CHECK_ARG((long)Tx * Ty < (1L << 31), "glowtts_mas_path: lattice too large");
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    const int r = (Tx + 63) / 64;
    switch (r) {
        case 1: return launch_mas<1>(value, path, t_x, t_y, B, Tx, Ty, s);
        case 2: return launch_mas<2>(value, path, t_x, t_y, B, Tx, Ty, s);
        case 3: return launch_mas<3>(value, path, t_x, t_y, B, Tx, Ty, s);
        case 4: return launch_mas<4>(value, path, t_x, t_y, B, Tx, Ty, s);
        case 5: return launch_mas<5>(value, path, t_x, t_y, B, Tx, Ty, s);
        case 6: return launch_mas<6>(value, path, t_x, t_y, B, Tx, Ty, s);
        default: return launch_mas<8>(value, path, t_x, t_y, B, Tx, Ty, s);
    }
}
