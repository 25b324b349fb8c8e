// owwhip_px.h -- weight-stationary, layer-per-SIMD form of the f16-split stage kernels (included by owwhip.hip).
//
// A stage B..E is four convolution layers; a CU has four SIMDs with a 512-entry register file each.  Here a workgroup is
// four waves, one per SIMD, and wave l IS layer l: it loads the packed hi/lo weights of its layer into registers once
// (B: 72 / 144 registers, C: 240 / 360) and keeps them for the whole launch; the workgroup is persistent and walks over
// its groups of streams ("units") as a four-deep pipeline,
//      iteration i:   wave 0: layer a of unit i      wave 1: layer b of unit i-1
//                     wave 2: layer c of unit i-2    wave 3: layer d of unit i-3 (+ pooling, store)
// handing the activations of a unit from wave to wave through double-buffered LDS tiles in MFMA operand form (f16 hi/lo,
// the producer's lane layout is the consumer's, owwhip_hx.h), one workgroup barrier per iteration.  The waves that own a
// 1x3 layer (0 and 2) also fetch the unit's conv history, forward it to the following 3x1 layer through the same LDS tiles and
// store the new history; they fetch one unit ahead, so no HBM latency sits in front of an MFMA.
// Compared with hstage_kernel: no weight traffic at all after the prologue (there: every workgroup re-streams the stage's
// weights L2 -> LDS -> registers for its 4..32 streams), no per-chunk barrier, no LDS read per MFMA triple.
// Same arithmetic, same operand order, same state arrays and hand-over layouts as hstage_kernel: results are bit-identical.
#pragma once
#include "owwhip_hx.h"

#ifndef OWP_SB
#define OWP_SB 1          // 1: scheduling barrier after every output-channel tile
#endif
#ifndef OWP_PIPE
#define OWP_PIPE 0        // n > 0: pin 1 MFMA : n VALU between the MFMAs of tile k and the epilogue of tile k-1
#endif
#if OWP_SB
#define OWP_OCT_SB() __builtin_amdgcn_sched_barrier(0)
#else
#define OWP_OCT_SB() do {} while (0)
#endif

namespace owp {

using owh::Op;
using owh::f16x8;
using owr::f32x4;

// one layer's packed weights, resident in registers: [oct][tap][ks][hi/lo] blocks of the hx packing (64 lanes x 16 B)
template <int NCTO, int KSI>
struct WRegs {
    f16x8 w[NCTO][3][KSI][2];
    __device__ __forceinline__ void load(const float* __restrict__ g, int lane) {
#pragma unroll
        for (int oct = 0; oct < NCTO; ++oct)
#pragma unroll
            for (int tap = 0; tap < 3; ++tap)
#pragma unroll
                for (int ks = 0; ks < KSI; ++ks)
#pragma unroll
                    for (int part = 0; part < 2; ++part)
                    {
                        // the weights live in the accumulation half of the register file (AGPRs): the MFMA reads its A operand from
                        // there directly; without the "a" constraint the allocator parks them there and copies them back per use
                        typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
                        u32x4 v = *reinterpret_cast<const u32x4*>(g + ((size_t)((((oct * 3 + tap) * KSI + ks) * 2 + part) * 64 + lane)) * 4);
                        asm volatile("" : "+a"(v));
                        w[oct][tap][ks][part] = __builtin_bit_cast(f16x8, v);
                    }
    }
};

// folded BatchNorm of one layer for this lane's channels (4j .. 4j+3 of every output-channel tile)
template <int NCTO>
struct BnRegs {
    f32x4 sc[NCTO], sh[NCTO];
    __device__ __forceinline__ void load(const float* __restrict__ scale, const float* __restrict__ shift, int j) {
#pragma unroll
        for (int oct = 0; oct < NCTO; ++oct) {
            sc[oct] = *reinterpret_cast<const f32x4*>(scale + oct * 16 + 4 * j);
            sh[oct] = *reinterpret_cast<const f32x4*>(shift + oct * 16 + 4 * j);
        }
    }
    __device__ __forceinline__ f32x4 act(const f32x4 v, int oct) const {
        f32x4 r;
#pragma unroll
        for (int e = 0; e < 4; ++e) r[e] = owr::leaky_clamp(v[e] * sc[oct][e] + sh[oct][e]);
        return r;
    }
};

// LDS tile of one row in operand form: [ks][hi/lo][64 lanes][8 halves]
template <int KS>
__device__ __forceinline__ void put_row(float* buf, const Op (&o)[KS], int lane) {
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
        *reinterpret_cast<f16x8*>(buf + ((ks * 2 + 0) * 64 + lane) * 4) = o[ks].h;
        *reinterpret_cast<f16x8*>(buf + ((ks * 2 + 1) * 64 + lane) * 4) = o[ks].l;
    }
}
template <int KS>
__device__ __forceinline__ void get_row(const float* buf, Op (&o)[KS], int lane) {
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
        o[ks].h = *reinterpret_cast<const f16x8*>(buf + ((ks * 2 + 0) * 64 + lane) * 4);
        o[ks].l = *reinterpret_cast<const f16x8*>(buf + ((ks * 2 + 1) * 64 + lane) * 4);
    }
}

// 1x3 (mel) layer on NT row tiles, weights in registers (arithmetic and operand order of owh::conv_mel_hx)
template <int KSI, int NCTO, int NT, int F>
__device__ __forceinline__ void conv_mel_px(const Op (&in)[NT][KSI], f32x4 (&out)[NT][NCTO], const WRegs<NCTO, KSI>& W,
                                            const BnRegs<NCTO>& bn, int lane) {
    using namespace owr;
    const int pos = lane & 15;
    const bool first = (pos & (F - 1)) == 0, last = (pos & (F - 1)) == F - 1;
#pragma unroll
    for (int oct = 0; oct < NCTO; ++oct) {
        f32x4 res[NT], accs[2][NT];
#pragma unroll
        for (int ti = 0; ti < 3; ++ti) {                                  // tap order 0, 2, 1
            const int tap = ti == 0 ? 0 : (ti == 1 ? 2 : 1);
            f32x4 acc[NT];
#pragma unroll
            for (int t = 0; t < NT; ++t) {
                if (ti < 2) acc[t] = f32x4{0.f, 0.f, 0.f, 0.f};
                else {
#pragma unroll
                    for (int e = 0; e < 4; ++e) { const float l = dpp_shr1_zero(accs[0][t][e]); acc[t][e] = (F < 16 && first) ? 0.f : l; }
                }
            }
#pragma unroll
            for (int ks = 0; ks < KSI; ++ks) {
                const f16x8 ah = W.w[oct][tap][ks][0], al = W.w[oct][tap][ks][1];
#pragma unroll
                for (int t = 0; t < NT; ++t) acc[t] = OWH_MFMA(ah, in[t][ks].h, acc[t]);
#pragma unroll
                for (int t = 0; t < NT; ++t) acc[t] = OWH_MFMA(ah, in[t][ks].l, acc[t]);
#pragma unroll
                for (int t = 0; t < NT; ++t) acc[t] = OWH_MFMA(al, in[t][ks].h, acc[t]);
            }
#pragma unroll
            for (int t = 0; t < NT; ++t) {
                if (ti < 2) accs[ti][t] = acc[t];
                else {
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const float hh = dpp_shl1_zero(accs[1][t][e]);
                        res[t][e] = acc[t][e] + ((F < 16 && last) ? 0.f : hh);
                    }
                }
            }
        }
#pragma unroll
        for (int t = 0; t < NT; ++t) { out[t][oct] = bn.act(res[t], oct); pin(out[t][oct]); }
        OWP_OCT_SB();
    }
}

// 3x1 (time) layer: rows[0..NR+1] = two history rows then the NR new rows; out row r uses rows r, r+1, r+2
// (arithmetic and operand order of owh::conv_time_hx; epilogue of tile k-1 issued behind the MFMAs of tile k)
template <int KSI, int NCTO, int NR>
__device__ __forceinline__ void conv_time_px(const Op (&rows)[NR + 2][KSI], f32x4 (&out)[NR][NCTO], const WRegs<NCTO, KSI>& W,
                                             const BnRegs<NCTO>& bn) {
    using namespace owr;
    f32x4 prev[NR];
#pragma unroll
    for (int oct = 0; oct <= NCTO; ++oct) {
        f32x4 acc[NR];
        if (oct < NCTO) {
#pragma unroll
            for (int r = 0; r < NR; ++r) acc[r] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int tap = 0; tap < 3; ++tap)
#pragma unroll
                for (int ks = 0; ks < KSI; ++ks) {
                    const f16x8 ah = W.w[oct][tap][ks][0], al = W.w[oct][tap][ks][1];
#pragma unroll
                    for (int part = 0; part < 3; ++part)
#pragma unroll
                        for (int r = 0; r < NR; ++r) {
                            const Op& b = rows[r + tap][ks];
                            acc[r] = OWH_MFMA(part == 2 ? al : ah, part == 1 ? b.l : b.h, acc[r]);
                        }
                }
        }
        if (oct > 0) {
#pragma unroll
            for (int r = 0; r < NR; ++r) { out[r][oct - 1] = bn.act(prev[r], oct - 1); pin(out[r][oct - 1]); }
        }
#if OWP_PIPE
        if (oct > 0 && oct < NCTO) {
#pragma unroll
            for (int i = 0; i < 9 * KSI * NR; ++i) {
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                __builtin_amdgcn_sched_group_barrier(0x002, OWP_PIPE, 0);
            }
        }
#endif
        if (oct < NCTO) {
#pragma unroll
            for (int r = 0; r < NR; ++r) prev[r] = acc[r];
            OWP_OCT_SB();
        }
    }
}

template <class C>
struct PLayout {
    static constexpr int KSA = (C::NCTI + 1) / 2, KS = (C::NCT + 1) / 2;
    static constexpr int ROW = KS * 2 * 256;                  // floats per row tile in operand form
    static constexpr int AB = (C::R + 2) * ROW, BC = C::R * ROW, CD = (C::R + 2) * ROW;      // one buffer of each hand-over
    static constexpr int LDS_FLOATS = 2 * (AB + BC + CD);
    static constexpr int LDS_BYTES = LDS_FLOATS * 4;
};

// stages B..D (not the last stage), one pass over the R new rows
template <class C, bool DBG>
__global__ __launch_bounds__(256, 1) void pstage_kernel(owr::RStageParams p) {
    using namespace owr;
    using L = PLayout<C>;
    static_assert(C::NPASS == 1, "the pipelined form carries all R rows of a unit at once");
    constexpr int NCTI = C::NCTI, NCT = C::NCT, R = C::R, F = C::F, KSA = L::KSA, KS = L::KS;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* bufAB = smem;                                       // [2][R+2][ROW]
    float* bufBC = bufAB + 2 * L::AB;                          // [2][R][ROW]
    float* bufCD = bufBC + 2 * L::BC;                          // [2][R+2][ROW]
    const int lane = threadIdx.x & 63, j = lane >> 4;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int n_units = (p.n_groups - (int)blockIdx.x + (int)gridDim.x - 1) / (int)gridDim.x;     // groups blockIdx.x, + gridDim.x, ...
    const int n_iter = n_units + 3;

    if (wave == 0) {
        // ---------------- layer a: 1x3, CIN -> C; forwards the old hist_b rows, stores the new ones
        WRegs<NCT, KSA> W; W.load(p.w[0], lane);
        BnRegs<NCT> bn; bn.load(p.scale[0], p.shift[0], j);
        f32x4 Xn[R][NCTI], Hn[2][NCT];                        // fetched one unit ahead
        auto fetch = [&](int u) {
            const int g = blockIdx.x + u * gridDim.x;
#pragma unroll
            for (int r = 0; r < R; ++r) load_tile<NCTI>(Xn[r], p.xin + ((size_t)g * R + r) * (NCTI * 4 * 64), lane);
            const float* hb = p.hist_b + (size_t)g * C::HIST_FLOATS;
            load_tile<NCT>(Hn[0], hb, lane);
            load_tile<NCT>(Hn[1], hb + NCT * 4 * 64, lane);
        };
        if (n_units > 0) fetch(0);
        for (int i = 0; i < n_iter; ++i) {
            const int u = i;
            if (u < n_units) {
                const int g = blockIdx.x + u * gridDim.x;
                float* dst = bufAB + (u & 1) * L::AB;
                Op Xo[R][KSA];
#pragma unroll
                for (int r = 0; r < R; ++r) owh::to_ops<NCTI>(Xn[r], Xo[r]);
                {
                    Op H[KS];
                    owh::to_ops<NCT>(Hn[0], H); put_row<KS>(dst, H, lane);
                    owh::to_ops<NCT>(Hn[1], H); put_row<KS>(dst + L::ROW, H, lane);
                }
                if (u + 1 < n_units) fetch(u + 1);
                f32x4 Y[R][NCT];
                conv_mel_px<KSA, NCT, R, F>(Xo, Y, W, bn, lane);
                float* hb = p.hist_b + (size_t)g * C::HIST_FLOATS;
                store_tile<NCT>(Y[R - 2], hb, lane);
                store_tile<NCT>(Y[R - 1], hb + NCT * 4 * 64, lane);
#pragma unroll
                for (int r = 0; r < R; ++r) { Op o[KS]; owh::to_ops<NCT>(Y[r], o); put_row<KS>(dst + (2 + r) * L::ROW, o, lane); }
                if (DBG && p.dbg) {
#pragma unroll
                    for (int r = 0; r < R; ++r) dump_tile<NCT, F, C::C>(Y[r], p.dbg, p.dbg_stride, p.dbg_off[0], g * C::SPT, r, p.S, lane);
                }
            }
            __syncthreads();
        }
    } else if (wave == 1) {
        // ---------------- layer b: 3x1 over [hist_b(2) ; Ya]
        WRegs<NCT, KS> W; W.load(p.w[1], lane);
        BnRegs<NCT> bn; bn.load(p.scale[1], p.shift[1], j);
        for (int i = 0; i < n_iter; ++i) {
            const int u = i - 1;
            if (u >= 0 && u < n_units) {
                const int g = blockIdx.x + u * gridDim.x;
                const float* src = bufAB + (u & 1) * L::AB;
                Op rows[R + 2][KS];
#pragma unroll
                for (int r = 0; r < R + 2; ++r) get_row<KS>(src + r * L::ROW, rows[r], lane);
                f32x4 Y[R][NCT];
                conv_time_px<KS, NCT, R>(rows, Y, W, bn);
                float* dst = bufBC + (u & 1) * L::BC;
#pragma unroll
                for (int r = 0; r < R; ++r) { Op o[KS]; owh::to_ops<NCT>(Y[r], o); put_row<KS>(dst + r * L::ROW, o, lane); }
                if (DBG && p.dbg) {
#pragma unroll
                    for (int r = 0; r < R; ++r) dump_tile<NCT, F, C::C>(Y[r], p.dbg, p.dbg_stride, p.dbg_off[1], g * C::SPT, r, p.S, lane);
                }
            }
            __syncthreads();
        }
    } else if (wave == 2) {
        // ---------------- layer c: 1x3, C -> C; forwards the old hist_d rows, stores the new ones
        WRegs<NCT, KS> W; W.load(p.w[2], lane);
        BnRegs<NCT> bn; bn.load(p.scale[2], p.shift[2], j);
        f32x4 Hn[2][NCT];
        auto fetch = [&](int u) {
            const int g = blockIdx.x + u * gridDim.x;
            const float* hd = p.hist_d + (size_t)g * C::HIST_FLOATS;
            load_tile<NCT>(Hn[0], hd, lane);
            load_tile<NCT>(Hn[1], hd + NCT * 4 * 64, lane);
        };
        if (n_units > 0) fetch(0);
        for (int i = 0; i < n_iter; ++i) {
            const int u = i - 2;
            if (u >= 0 && u < n_units) {
                const int g = blockIdx.x + u * gridDim.x;
                const float* src = bufBC + (u & 1) * L::BC;
                float* dst = bufCD + (u & 1) * L::CD;
                Op in[R][KS];
#pragma unroll
                for (int r = 0; r < R; ++r) get_row<KS>(src + r * L::ROW, in[r], lane);
                {
                    Op H[KS];
                    owh::to_ops<NCT>(Hn[0], H); put_row<KS>(dst, H, lane);
                    owh::to_ops<NCT>(Hn[1], H); put_row<KS>(dst + L::ROW, H, lane);
                }
                if (u + 1 < n_units) fetch(u + 1);
                f32x4 Y[R][NCT];
                conv_mel_px<KS, NCT, R, F>(in, Y, W, bn, lane);
                float* hd = p.hist_d + (size_t)g * C::HIST_FLOATS;
                store_tile<NCT>(Y[R - 2], hd, lane);
                store_tile<NCT>(Y[R - 1], hd + NCT * 4 * 64, lane);
#pragma unroll
                for (int r = 0; r < R; ++r) { Op o[KS]; owh::to_ops<NCT>(Y[r], o); put_row<KS>(dst + (2 + r) * L::ROW, o, lane); }
                if (DBG && p.dbg) {
#pragma unroll
                    for (int r = 0; r < R; ++r) dump_tile<NCT, F, C::C>(Y[r], p.dbg, p.dbg_stride, p.dbg_off[2], g * C::SPT, r, p.S, lane);
                }
            }
            __syncthreads();
        }
    } else {
        // ---------------- layer d: 3x1 over [hist_d(2) ; Yc], pooling, hand-over to the next stage
        WRegs<NCT, KS> W; W.load(p.w[3], lane);
        BnRegs<NCT> bn; bn.load(p.scale[3], p.shift[3], j);
        for (int i = 0; i < n_iter; ++i) {
            const int u = i - 3;
            if (u >= 0 && u < n_units) {
                const int g = blockIdx.x + u * gridDim.x;
                const float* src = bufCD + (u & 1) * L::CD;
                Op rows[R + 2][KS];
#pragma unroll
                for (int r = 0; r < R + 2; ++r) get_row<KS>(src + r * L::ROW, rows[r], lane);
                f32x4 Y[R][NCT];
                conv_time_px<KS, NCT, R>(rows, Y, W, bn);
                if (DBG && p.dbg) {
#pragma unroll
                    for (int r = 0; r < R; ++r) dump_tile<NCT, F, C::C>(Y[r], p.dbg, p.dbg_stride, p.dbg_off[3], g * C::SPT, r, p.S, lane);
                }
                pool_store<C>(Y, p.xout, g, 0, lane);
            }
            __syncthreads();
        }
    }
}

}  // namespace owp
