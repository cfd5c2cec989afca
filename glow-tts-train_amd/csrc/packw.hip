// packw.hip — weight norm + MFMA packing of several convolutions per launch, and its backward, on 16-row tiles.
//
// The one-row-per-workgroup kernels of convgemm.hip (kept for the single-convolution entry points) write the backward-data
// packing wp_b[..][c][o % 16] and read the packed gradient dwp[tap][c][o] four bytes at a time with a stride of a cache
// line or more: 20 us per WN stack for 21 MB (pack), 19 us for 28 MB (un-pack), plus a second launch that re-reads the
// packed weights to make their bf16 planes.  A workgroup that owns 16 consecutive output channels sees both packings and the
// packed gradient as whole 64-byte segments:
//   wp_f[tap][c / 16][o][c % 16]           16 o x 16 c  = 1 KB contiguous per (tap, channel group)
//   wp_b[taps-1-tap][o / 16][c][o % 16]    16 c x 16 o  = 1 KB contiguous per (tap, channel group)
//   dwp [tap][c][o]                        16 o         = 64 B contiguous per (tap, c)
// and writes the three bf16 planes of every packed value (the operands of convgemm_split.hip) in the same pass.
// Reference: torch.nn.utils.weight_norm (dim 0) as applied in layers.py:113,125,135 and its autograd.
#include "common.hpp"
#include "split_planes.hpp"

namespace glowtts {

namespace {

constexpr int TP = 17;     // LDS pitch of a 16 x 16 transpose tile

struct PlaneOut {
    const float *arena;    // the packed buffers of a launch lie in [arena, arena + plane_stride)
    uint16_t *planes;      // planes[pl * plane_stride + (p - arena)] = plane pl of *p
    long plane_stride;
};

template <bool PLANES>
__device__ __forceinline__ void store_pair(float *dst, float a, float b, const PlaneOut &po) {
    *reinterpret_cast<float2 *>(dst) = make_float2(a, b);
    if constexpr (PLANES) {
        unsigned o[3];
        split_planes2<3>(a, b, o);
        const long e = dst - po.arena;
#pragma unroll
        for (int pl = 0; pl < 3; ++pl) *reinterpret_cast<unsigned *>(po.planes + pl * po.plane_stride + e) = o[pl];
    }
}

// rows o0 .. o0+15 of one convolution; blockIdx.y strides over pairs of 16-channel groups
template <int TAPS, bool PLANES>
__device__ __forceinline__ void pack_tile(const float *__restrict__ v, const float *__restrict__ g, float *__restrict__ wp_f,
                                          float *__restrict__ wp_b, float *__restrict__ inv_norm, int o0, int Cout, int Cin,
                                          const PlaneOut &po, float *scale_s, float *tile_s) {
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int n = Cin * TAPS;
    const int Gi = (Cin + 15) / 16, Go = (Cout + 15) / 16, og = o0 >> 4;
    const int h2 = tid >> 7, u = tid & 127;
    const int o = u >> 3, cl2 = (u & 7) * 2;          // forward packing: row o, channels cl2, cl2 + 1 of the group
    const int cb = u >> 3, ob2 = (u & 7) * 2;         // backward packing: channel cb, rows ob2, ob2 + 1
    const int row = o0 + o;
    const int npairs = (Gi + 1) / 2;
    float w[2][TAPS], wn[2][TAPS];
    auto load = [&](int pr, float (&dst)[2][TAPS]) {
        const int c0 = (pr * 2 + h2) * 16 + cl2;
#pragma unroll
        for (int k = 0; k < 2; ++k)
#pragma unroll
            for (int tap = 0; tap < TAPS; ++tap)
                dst[k][tap] = (row < Cout && c0 + k < Cin) ? v[(long)row * n + (c0 + k) * TAPS + tap] : 0.f;
    };
    if ((int)blockIdx.y < npairs) load(blockIdx.y, w);        // in flight during the norm pass
    if (g != nullptr) {                               // ||v[o]||: a wave takes four rows, their loads interleaved
        float s[4] = {0.f, 0.f, 0.f, 0.f};
        const int r0 = o0 + wave * 4;
        const float g_r = (lane < 4 && r0 + lane < Cout) ? g[r0 + lane] : 0.f;
        if ((n & 3) == 0 && ((size_t)v & 15) == 0) {
            const int n4 = n >> 2;
#pragma unroll 4
            for (int i = lane; i < n4; i += 64) {
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const float4 a = (r0 + r < Cout) ? reinterpret_cast<const float4 *>(v + (long)(r0 + r) * n)[i] : make_float4(0.f, 0.f, 0.f, 0.f);
                    s[r] += (a.x * a.x + a.y * a.y) + (a.z * a.z + a.w * a.w);
                }
            }
        } else {
#pragma unroll 4
            for (int i = lane; i < n; i += 64) {
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const float a = (r0 + r < Cout) ? v[(long)(r0 + r) * n + i] : 0.f;
                    s[r] += a * a;
                }
            }
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const float t = wave_sum(s[r]);
            if (lane == 0) {
                const bool ok = r0 + r < Cout;
                const float inv = ok ? 1.0f / sqrtf(t) : 0.f;
                scale_s[wave * 4 + r] = inv;
                if (ok && inv_norm && blockIdx.y == 0) inv_norm[r0 + r] = inv;
            }
        }
        __builtin_amdgcn_wave_barrier();
        if (lane < 4) scale_s[wave * 4 + lane] *= g_r;
    } else if (tid < 16) {
        scale_s[tid] = 1.f;
    }
    __syncthreads();

    const float sc = scale_s[o];
    for (int pr = blockIdx.y; pr < npairs; pr += gridDim.y) {
        const int cg = pr * 2 + h2;
        if (pr + (int)gridDim.y < npairs) load(pr + gridDim.y, wn);     // in flight across this pair's stores and barriers
#pragma unroll
        for (int k = 0; k < 2; ++k)
#pragma unroll
            for (int tap = 0; tap < TAPS; ++tap) w[k][tap] *= sc;
        if (wp_f != nullptr && cg < Gi && row < Cout) {
#pragma unroll
            for (int tap = 0; tap < TAPS; ++tap)
                store_pair<PLANES>(wp_f + (((long)tap * Gi + cg) * Cout + row) * 16 + cl2, w[0][tap], w[1][tap], po);
        }
        if (wp_b != nullptr) {                        // uniform: the barriers below are reached by every thread
#pragma unroll
            for (int tap = 0; tap < TAPS; ++tap) {
                tile_s[((tap * 2 + h2) * 16 + cl2) * TP + o] = w[0][tap];
                tile_s[((tap * 2 + h2) * 16 + cl2 + 1) * TP + o] = w[1][tap];
            }
            __syncthreads();
            const int c = cg * 16 + cb;
            if (cg < Gi && c < Cin) {
#pragma unroll
                for (int tap = 0; tap < TAPS; ++tap) {
                    const float a = tile_s[((tap * 2 + h2) * 16 + cb) * TP + ob2];
                    const float b = tile_s[((tap * 2 + h2) * 16 + cb) * TP + ob2 + 1];
                    store_pair<PLANES>(wp_b + (((long)(TAPS - 1 - tap) * Go + og) * Cin + c) * 16 + ob2, a, b, po);
                }
            }
            __syncthreads();
        }
#pragma unroll
        for (int k = 0; k < 2; ++k)
#pragma unroll
            for (int tap = 0; tap < TAPS; ++tap) w[k][tap] = wn[k][tap];
    }
}

// any tap count: one row after the other, element by element (no convolution of the model comes here)
template <bool PLANES>
__device__ void pack_tile_generic(const float *__restrict__ v, const float *__restrict__ g, float *__restrict__ wp_f,
                                  float *__restrict__ wp_b, float *__restrict__ inv_norm, int o0, int Cout, int Cin, int taps,
                                  const PlaneOut &po, float *red) {
    if (blockIdx.y != 0) return;
    const int n = Cin * taps, Gi = (Cin + 15) / 16, Go = (Cout + 15) / 16;
    for (int o = o0; o < min(Cout, o0 + 16); ++o) {
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
        for (int i = threadIdx.x; i < n; i += 256) {
            const int c = i / taps, tap = i - c * taps;
            const float w = vo[i] * scale;
            float *d[2] = {wp_f ? wp_f + (((long)tap * Gi + (c >> 4)) * Cout + o) * 16 + (c & 15) : nullptr,
                           wp_b ? wp_b + (((long)(taps - 1 - tap) * Go + (o >> 4)) * Cin + c) * 16 + (o & 15) : nullptr};
            for (int k = 0; k < 2; ++k)
                if (d[k]) {
                    *d[k] = w;
                    if constexpr (PLANES) {
                        unsigned pl3[3];
                        split_planes<3>(w, pl3);
                        for (int pl = 0; pl < 3; ++pl) po.planes[pl * po.plane_stride + (d[k] - po.arena)] = (uint16_t)pl3[pl];
                    }
                }
        }
    }
}

// Which convolution and which of its 16-row tiles blockIdx.x is: lane c reads convolution c's row count (one trip to memory
// for the whole table instead of one per convolution), an inclusive scan over the lanes gives the tile prefix.
__device__ __forceinline__ bool find_tile(const long long *__restrict__ desc, int n_conv, int words, int cout_word, int &c_out,
                                          int &tile_out) {
    const int lane = threadIdx.x & 63;
    int base = 0;
    for (int c0 = 0; c0 < n_conv; c0 += 64) {          // (a launch has at most a few dozen convolutions: one round)
        const int c = c0 + lane;
        const int nt = c < n_conv ? (((int)desc[(long)c * words + cout_word] + 15) >> 4) : 0;
        int incl = nt;
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) {
            const int up = __shfl_up(incl, off, 64);
            if (lane >= off) incl += up;
        }
        const int tile = (int)blockIdx.x - base;
        const unsigned long long hit = __ballot(tile < incl);
        if (hit != 0ull) {
            const int l = __ffsll((long long)hit) - 1;
            c_out = c0 + l;
            tile_out = tile - (__shfl(incl, l, 64) - __shfl(nt, l, 64));
            return true;
        }
        base += __shfl(incl, 63, 64);
    }
    return false;
}

// desc[c] = {v, g, wp_f, wp_b, inv_norm, Cout, Cin, taps}; blockIdx.x walks the 16-row tiles of the convolutions in order
template <bool PLANES>
__global__ __launch_bounds__(256) void pack_weight_tile_kernel(const long long *__restrict__ desc, int n_conv, PlaneOut po) {
    __shared__ float scale_s[16];
    __shared__ float tile_s[5 * 2 * 16 * TP];
    int tile, c;
    if (!find_tile(desc, n_conv, 8, 5, c, tile)) return;      // the grid is an upper bound of the tile count
    const long long *d = desc + (long)c * 8;
    const float *v = reinterpret_cast<const float *>(d[0]), *g = reinterpret_cast<const float *>(d[1]);
    float *wp_f = reinterpret_cast<float *>(d[2]), *wp_b = reinterpret_cast<float *>(d[3]);
    float *inv_norm = reinterpret_cast<float *>(d[4]);
    const int Cout = (int)d[5], Cin = (int)d[6], taps = (int)d[7];
    switch (taps) {
    case 1: pack_tile<1, PLANES>(v, g, wp_f, wp_b, inv_norm, tile * 16, Cout, Cin, po, scale_s, tile_s); break;
    case 3: pack_tile<3, PLANES>(v, g, wp_f, wp_b, inv_norm, tile * 16, Cout, Cin, po, scale_s, tile_s); break;
    case 5: pack_tile<5, PLANES>(v, g, wp_f, wp_b, inv_norm, tile * 16, Cout, Cin, po, scale_s, tile_s); break;
    default: pack_tile_generic<PLANES>(v, g, wp_f, wp_b, inv_norm, tile * 16, Cout, Cin, taps, po, scale_s); break;
    }
}

// ---- backward: dw[o][c][tap] = dwp[tap][c][o];  plain conv: dv += dw
// weight norm: dg[o] += sum(dw * v) / ||v|| ;  dv += (g / ||v||) * (dw - v * sum(dw * v) / ||v||^2)
// A stage = 64 channels x TAPS of the 16 rows: read as 64-byte segments of dwp, turned through LDS so that the v / dv rows
// are walked along their contiguous index i = c * TAPS + tap.
constexpr int kStageCh = 64;

template <int TAPS>
struct UnpackGeom {
    static constexpr int W = kStageCh * TAPS, PITCH = W + 1, PER = W / 16;   // PER values of a stage per thread, either role
};

// loads of a stage, all issued before anything waits: the 64-byte segments of dwp (thread = row ol of item it0 + 16 k,
// item = tap * 64 + channel) and the v row pieces this thread will pair them with (row o, i = j + 16 k)
template <int TAPS>
__device__ __forceinline__ void unpack_stage_load(const float *__restrict__ dwp, const float *__restrict__ v, int o0, int Cout,
                                                  int Cin, int sgi, float (&r)[UnpackGeom<TAPS>::PER],
                                                  float (&vv)[UnpackGeom<TAPS>::PER]) {
    using G = UnpackGeom<TAPS>;
    const int ol = threadIdx.x & 15, it0 = threadIdx.x >> 4;
    const int ch0 = sgi * kStageCh, n = Cin * TAPS;
#pragma unroll
    for (int k = 0; k < G::PER; ++k) {
        const int item = it0 + 16 * k;
        const int tap = item / kStageCh, c = ch0 + (item - tap * kStageCh);
        r[k] = (c < Cin && o0 + ol < Cout) ? dwp[((long)tap * Cin + c) * Cout + o0 + ol] : 0.f;
    }
    if (v != nullptr) {
        const int row = o0 + it0, j = ol;              // the consumer's (o, j) are the same two fields of the thread index
#pragma unroll
        for (int k = 0; k < G::PER; ++k) {
            const int i = sgi * G::W + j + 16 * k;
            vv[k] = (row < Cout && i < n) ? v[(long)row * n + i] : 0.f;
        }
    }
}

// through LDS: r (segments of dwp) -> dw (this thread's pieces of row o along i)
template <int TAPS>
__device__ __forceinline__ void unpack_stage_turn(float (&r)[UnpackGeom<TAPS>::PER], float *st) {
    using G = UnpackGeom<TAPS>;
    const int ol = threadIdx.x & 15, it0 = threadIdx.x >> 4;
#pragma unroll
    for (int k = 0; k < G::PER; ++k) {
        const int item = it0 + 16 * k;
        const int tap = item / kStageCh, cl = item - tap * kStageCh;
        st[ol * G::PITCH + cl * TAPS + tap] = r[k];
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < G::PER; ++k) r[k] = st[it0 * G::PITCH + ol + 16 * k];
    __syncthreads();
}

template <int TAPS>
__device__ __forceinline__ void unpack_tile(const float *__restrict__ dwp, const float *__restrict__ v,
                                            const float *__restrict__ g, const float *__restrict__ inv_norm,
                                            float *__restrict__ dv, float *__restrict__ dg, int o0, int Cout, int Cin, float *st) {
    using G = UnpackGeom<TAPS>;
    constexpr int KEEP = 3;                           // stages whose values stay in registers (Cin <= 192: the WN stacks)
    const int o = threadIdx.x >> 4, j = threadIdx.x & 15;
    const int row = o0 + o, n = Cin * TAPS;
    const bool rok = row < Cout;
    const int nstage = (Cin + kStageCh - 1) / kStageCh;
    const float *vsrc = g != nullptr ? v : nullptr;
    const float inv_r = (g != nullptr && rok) ? inv_norm[row] : 0.f;
    const float g_r = (g != nullptr && rok) ? g[row] : 0.f;
    auto finish = [&](float dot, float &gn, float &proj) {
#pragma unroll
        for (int off = 8; off > 0; off >>= 1) dot += __shfl_xor(dot, off, 64);
        gn = g_r * inv_r;
        proj = dot * inv_r * inv_r;
        if (rok && j == 0) dg[row] += dot * inv_r;
    };
    if (nstage <= KEEP) {                             // one trip to memory for dwp, v and dv
        float dw[KEEP][G::PER], vv[KEEP][G::PER], old[KEEP][G::PER];
#pragma unroll
        for (int sgi = 0; sgi < KEEP; ++sgi)
            if (sgi < nstage) {
                unpack_stage_load<TAPS>(dwp, vsrc, o0, Cout, Cin, sgi, dw[sgi], vv[sgi]);
#pragma unroll
                for (int k = 0; k < G::PER; ++k) {
                    const int i = sgi * G::W + j + 16 * k;
                    old[sgi][k] = (rok && i < n) ? dv[(long)row * n + i] : 0.f;
                }
            }
        float dot = 0.f;
#pragma unroll
        for (int sgi = 0; sgi < KEEP; ++sgi)
            if (sgi < nstage) {
                unpack_stage_turn<TAPS>(dw[sgi], st);
                if (g != nullptr) {
#pragma unroll
                    for (int k = 0; k < G::PER; ++k) dot += dw[sgi][k] * vv[sgi][k];
                }
            }
        float gn = 1.f, proj = 0.f;
        if (g != nullptr) finish(dot, gn, proj);
#pragma unroll
        for (int sgi = 0; sgi < KEEP; ++sgi)
            if (sgi < nstage) {
#pragma unroll
                for (int k = 0; k < G::PER; ++k) {
                    const int i = sgi * G::W + j + 16 * k;
                    if (rok && i < n) dv[(long)row * n + i] = old[sgi][k] + ((g != nullptr) ? gn * (dw[sgi][k] - vv[sgi][k] * proj) : dw[sgi][k]);
                }
            }
        return;
    }
    float gn = 1.f, proj = 0.f;
    float r[G::PER], vv[G::PER];
    if (g != nullptr) {                               // long rows: first pass sum(dw * v), second pass from L2
        float dot = 0.f;
        for (int sgi = 0; sgi < nstage; ++sgi) {
            unpack_stage_load<TAPS>(dwp, vsrc, o0, Cout, Cin, sgi, r, vv);
            unpack_stage_turn<TAPS>(r, st);
#pragma unroll
            for (int k = 0; k < G::PER; ++k) dot += r[k] * vv[k];
        }
        finish(dot, gn, proj);
    }
    for (int sgi = 0; sgi < nstage; ++sgi) {
        unpack_stage_load<TAPS>(dwp, vsrc, o0, Cout, Cin, sgi, r, vv);
        float old[G::PER];
#pragma unroll
        for (int k = 0; k < G::PER; ++k) {
            const int i = sgi * G::W + j + 16 * k;
            old[k] = (rok && i < n) ? dv[(long)row * n + i] : 0.f;
        }
        unpack_stage_turn<TAPS>(r, st);
#pragma unroll
        for (int k = 0; k < G::PER; ++k) {
            const int i = sgi * G::W + j + 16 * k;
            if (rok && i < n) dv[(long)row * n + i] = old[k] + ((g != nullptr) ? gn * (r[k] - vv[k] * proj) : r[k]);
        }
    }
}

__device__ void unpack_tile_generic(const float *__restrict__ dwp, const float *__restrict__ v, const float *__restrict__ g,
                                    const float *__restrict__ inv_norm, float *__restrict__ dv, float *__restrict__ dg, int o0,
                                    int Cout, int Cin, int taps, float *red) {
    const int n = Cin * taps;
    for (int o = o0; o < min(Cout, o0 + 16); ++o) {
        float proj = 0.f, gn = 1.f;
        if (g != nullptr) {
            float dot = 0.f;
            for (int i = threadIdx.x; i < n; i += 256) {
                const int c = i / taps, tap = i - c * taps;
                dot += dwp[((long)tap * Cin + c) * Cout + o] * v[(long)o * n + i];
            }
            dot = block_sum_256(dot, red);
            const float inv = inv_norm[o];
            gn = g[o] * inv;
            proj = dot * inv * inv;
            if (threadIdx.x == 0) dg[o] += dot * inv;
        }
        for (int i = threadIdx.x; i < n; i += 256) {
            const int c = i / taps, tap = i - c * taps;
            const float dw = dwp[((long)tap * Cin + c) * Cout + o];
            dv[(long)o * n + i] += (g != nullptr) ? gn * (dw - v[(long)o * n + i] * proj) : dw;
        }
    }
}

// desc[c] = {dwp, v, g, inv_norm, dv, dg, Cout, Cin, taps}
__global__ __launch_bounds__(256) void unpack_weight_grad_tile_kernel(const long long *__restrict__ desc, int n_conv) {
    extern __shared__ float st[];                     // 16 x (64 * 5 + 1) floats
    int tile, c;
    if (!find_tile(desc, n_conv, 9, 6, c, tile)) return;
    const long long *d = desc + (long)c * 9;
    const float *dwp = reinterpret_cast<const float *>(d[0]), *v = reinterpret_cast<const float *>(d[1]);
    const float *g = reinterpret_cast<const float *>(d[2]), *inv_norm = reinterpret_cast<const float *>(d[3]);
    float *dv = reinterpret_cast<float *>(d[4]), *dg = reinterpret_cast<float *>(d[5]);
    const int Cout = (int)d[6], Cin = (int)d[7], taps = (int)d[8];
    switch (taps) {
    case 1: unpack_tile<1>(dwp, v, g, inv_norm, dv, dg, tile * 16, Cout, Cin, st); break;
    case 3: unpack_tile<3>(dwp, v, g, inv_norm, dv, dg, tile * 16, Cout, Cin, st); break;
    case 5: unpack_tile<5>(dwp, v, g, inv_norm, dv, dg, tile * 16, Cout, Cin, st); break;
    default: unpack_tile_generic(dwp, v, g, inv_norm, dv, dg, tile * 16, Cout, Cin, taps, st); break;
    }
}

// the tile count of a launch is not known on the host (the tables live in device memory): an upper bound from the row count
inline unsigned tile_bound(int n_conv, int total_rows) { return (unsigned)(total_rows / 16 + n_conv); }

}  // namespace

}  // namespace glowtts

using namespace glowtts;

extern "C" int glowtts_pack_weight_multi(const long long *desc, const int *row_prefix, int n_conv, int total_rows,
                                         glowtts_stream_t stream) {
    GLOWTTS_CHECK_ARG(desc && row_prefix && n_conv > 0 && total_rows > 0, "glowtts_pack_weight_multi: bad argument");
    hipLaunchKernelGGL(pack_weight_tile_kernel<false>, dim3(tile_bound(n_conv, total_rows), 2), dim3(256), 0, (hipStream_t)stream,
                       desc, n_conv, PlaneOut{nullptr, nullptr, 0});
    GLOWTTS_LAUNCH_CHECK("glowtts_pack_weight_multi");
}

extern "C" int glowtts_pack_weight_planes_multi(const long long *desc, const int *row_prefix, int n_conv, int total_rows,
                                                const float *arena, long n_floats, uint16_t *planes, glowtts_stream_t stream) {
    GLOWTTS_CHECK_ARG(desc && row_prefix && n_conv > 0 && total_rows > 0 && arena && planes && n_floats > 0 && n_floats % 2 == 0 &&
                          ((size_t)arena & 7) == 0 && ((size_t)planes & 3) == 0,
                      "glowtts_pack_weight_planes_multi: bad argument");
    hipLaunchKernelGGL(pack_weight_tile_kernel<true>, dim3(tile_bound(n_conv, total_rows), 2), dim3(256), 0, (hipStream_t)stream,
                       desc, n_conv, PlaneOut{arena, planes, n_floats});
    GLOWTTS_LAUNCH_CHECK("glowtts_pack_weight_planes_multi");
}

extern "C" int glowtts_unpack_weight_grad_multi(const long long *desc, const int *row_prefix, int n_conv, int total_rows,
                                                glowtts_stream_t stream) {
    GLOWTTS_CHECK_ARG(desc && row_prefix && n_conv > 0 && total_rows > 0, "glowtts_unpack_weight_grad_multi: bad argument");
    constexpr size_t lds = (size_t)16 * (kStageCh * 5 + 1) * sizeof(float);
    hipLaunchKernelGGL(unpack_weight_grad_tile_kernel, dim3(tile_bound(n_conv, total_rows)), dim3(256), lds, (hipStream_t)stream, desc,
                       n_conv);
    GLOWTTS_LAUNCH_CHECK("glowtts_unpack_weight_grad_multi");
}
