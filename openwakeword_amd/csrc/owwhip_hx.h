// owwhip_hx.h -- fp16-split ("f16x3") form of the register-resident embedding CNN (included by owwhip.hip).
//
// Same dataflow, tiles, state arrays and memory layouts as owwhip_rr.h; only the inner product changes.  The fp32 MFMA
// runs at the fp32 VECTOR rate (157 TFLOP/s); the f16 MFMA is 16x faster.  Every fp32 operand is split into two f16
// halves, x = xh + xl with xh = f16(x), xl = f16(x - xh) (22 mantissa bits together), and
//        x * w  ~=  xh*wh + xh*wl + xl*wh                       (the dropped xl*wl term is < 2^-22 |x w|)
// is evaluated by three v_mfma_f32_16x16x32_f16 with fp32 accumulation: f16 x f16 products are exact in fp32, so the
// result differs from the fp32 kernels by ~2^-22 relative per product -- the same class as fp32 round-off itself
// (oracle emulation: |embedding error| 4.7e-6 vs 4.5e-6 for plain fp32 against float64; tests hold it to the same
// tolerances as the fp32 paths).  The embedding CNN's weights are pre-split on the host with the BatchNorm scale and the layers'
// activation-scale ratio folded in (round 3, see act1 below and calibrate_hx in owwhip.hip); the heads' and the VAD stand-in's
// weights are pre-scaled by 2^8 (keeps their low halves out of the f16 subnormal range; undone in their bias fma).
//
// Operand form.  16x16x32: A lane (i, g) holds A[i][k = 8g..8g+7], B lane (p, g) holds B[k = 8g..8g+7][p] (8 halves =
// 4 VGPRs each), D as in the fp32 form (lane (p, j), register e <-> D[4j+e][p]).  A k-step carries 32 input channels =
// two channel tiles: k = 8g + q  <->  channel 16*(2ks + q/4) + 4g + q%4, i.e. the B operand of k-step ks is the
// f16 split of D registers 0..3 of channel tiles 2ks and 2ks+1 of the same lane: again no data movement between layers,
// only the split (3 VALU ops per value) after the BatchNorm/activation epilogue.
#pragma once
#include <type_traits>
#include <utility>
#include "owwhip_rr.h"

namespace owh {

using owr::f32x4;
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

constexpr float WSCALE = 256.0f;            // heads / VAD: weights are stored as f16 halves of 2^8 * w
constexpr float WUNSCALE = 1.0f / 256.0f;

// BatchNorm in the f16-split family is FOLDED: the per-channel scale (sign included) goes into the f16-split weights, the shift is
// the accumulator's start value, and every layer's activations are carried multiplied by a per-layer power of two K = 2^e chosen at
// oww_commit from a calibration run on the exact-fp32 kernels (owwhip.hip: calibrate_hx) -- a ladder that climbs by 2^2 per layer
// inside a stage (so the folded weights keep precise low halves) and is brought back at the pooled hand-over:
//      acc = sum_k W'_k X_k + K h,   W' = s w K_out / K_in,  X = K_in a(y_in)   =>   acc = K_out y
//      K a(y) = max(max(0.2 acc, acc), -0.4 K)                 (the activation is positively homogeneous up to its clamp constant)
// Two VALU per value (v_mul, v_max3) instead of three, no silent underflow for small-valued networks, and max-pooling commutes
// with the (monotone) activation.  Results differ from the unfolded evaluation by fp32 round-off (s w is formed in double and split
// into hi + lo, 22 bits, like the plain weights were).
__device__ __forceinline__ float act1(float x, float cl) {
    return fmaxf(fmaxf(0.2f * x, x), cl);      // v_mul (paired: v_pk_mul) + v_max3
}
// activation of one output tile; HALF: registers 2, 3 are padding and stay zero
template <bool ACT, bool HALF>
__device__ __forceinline__ f32x4 act_t(const f32x4 v, float cl) {
    if (!ACT) return HALF ? f32x4{v[0], v[1], 0.f, 0.f} : v;
    if (HALF) return f32x4{act1(v[0], cl), act1(v[1], cl), 0.f, 0.f};
    return f32x4{act1(v[0], cl), act1(v[1], cl), act1(v[2], cl), act1(v[3], cl)};
}
// owr::pin for a tile whose registers 2, 3 are padding zeros (HALF): only the two live registers pass through the opaque asm -- pinning the
// constant zeros would materialise them (two v_mov per tile)
template <bool HALF>
__device__ __forceinline__ void pin_t(f32x4& v) {
    if (!HALF) { owr::pin(v); return; }
    float a = v[0], b = v[1];
    asm volatile("" : "+v"(a), "+v"(b));
    v = f32x4{a, b, 0.f, 0.f};
}
// the accumulator's start value of output tile oct: K * BatchNorm shift in tile row order (zero in padding rows); nullptr = 0
__device__ __forceinline__ f32x4 acc_init(const float* init, int oct, int j) {
    return init ? *reinterpret_cast<const f32x4*>(init + oct * 16 + 4 * j) : f32x4{0.f, 0.f, 0.f, 0.f};
}

// operand pair (hi, lo) of one k-step of one position tile
struct Op { f16x8 h, l; };

// Range guard of the f16 split.  An activation beyond the f16 range (|x| >= 65520) splits into hi = inf, lo = -inf (or NaN);
// in the layer that consumes it every product with it is +-inf or NaN (0 * inf included) and hi + lo contributions cancel to
// NaN, so EVERY output channel of that position has a NaN accumulator -- which the max()-based activation would then swallow
// (max3(NaN, NaN, -0.4) = -0.4): a silently wrong answer.  Testing ONE accumulator register per position tile and layer
// before the activation therefore catches the first overflow anywhere upstream; the lane masks are OR-ed in SGPRs
// (v_cmp_u_f32 + s_or_b64 per tile) and the wave raises the handle's sticky flag (host-mapped word) at kernel end.
typedef unsigned long long lanemask_t;
__device__ __forceinline__ void nan_guard(lanemask_t& bad, float x) {
    // v_cmp_u_f32 + s_or_b64.  The compare is the builtin (not inline asm) so that the compiler's hazard recogniser sees a VALU read of
    // an MFMA result and SCC / VCC stay modelled; the empty asm pins the OR here -- without it every tile's lane mask stays alive
    // until the end of the kernel (+50 SGPRs).
    bad |= __builtin_amdgcn_fcmpf(x, x, 8 /* FCMP_UNO */);
    asm volatile("" : "+s"(bad));
}
// flag[0] = raised; flag[1] = ONE packed word (first stream << 6 | stream count, count <= 32; -1 = position unknown) of one of the
// waves that saw it.  Several waves -- of different kernels, reporting 1, 2, 4, 8 or 32 streams -- may write concurrently; a single
// aligned 32-bit store is indivisible, so whichever write lands last, first and count always belong to the same wave (any offender is
// enough for a caller that resets the streams concerned instead of the whole handle: oww_range_where).
__device__ __forceinline__ void raise_range_flag(lanemask_t bad, int* flag, int first_stream = -1, int n_streams = 0) {
    if (bad != 0 && flag != nullptr && (threadIdx.x & 63) == 0) {
        flag[1] = first_stream < 0 ? -1 : (int)(((unsigned)first_stream << 6) | (unsigned)(n_streams & 63));
        *flag = 1;
    }
}

#ifndef OWH_MIXSPLIT
#define OWH_MIXSPLIT 1     // residual halves by v_fma_mix{lo,hi}_f16 (f16 source read in place): 1.5 instead of 2.6 VALU ops per value
#endif
template <bool MIX = (OWH_MIXSPLIT != 0)>
__device__ __forceinline__ Op split_pair(const f32x4 a, const f32x4 b) {
    Op o;
    if constexpr (MIX) {
    typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
    typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
    u32x4 h, l;
#pragma unroll
    for (int v = 0; v < 4; ++v) {
        const float x0 = v < 2 ? a[2 * v] : b[2 * v - 4], x1 = v < 2 ? a[2 * v + 1] : b[2 * v - 3];
        const unsigned hp = __builtin_bit_cast(unsigned, f16x2{(_Float16)x0, (_Float16)x1});       // v_cvt_pk_f16_f32
        unsigned lp;
        // lo16 = f16(x0 - f32(hp.lo16)), hi16 = f16(x1 - f32(hp.hi16)); op_sel_hi marks source 0 as f16, op_sel picks its half
        asm("v_fma_mixlo_f16 %0, %1, -1.0, %2 op_sel_hi:[1,0,0]\n\t"
            "v_fma_mixhi_f16 %0, %1, -1.0, %3 op_sel:[1,0,0] op_sel_hi:[1,0,0]"
            : "=&v"(lp) : "v"(hp), "v"(x0), "v"(x1));
        h[v] = hp; l[v] = lp;
    }
    o.h = __builtin_bit_cast(f16x8, h);
    o.l = __builtin_bit_cast(f16x8, l);
    } else {
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        const _Float16 ha = (_Float16)a[e], hb = (_Float16)b[e];
        o.h[e] = ha; o.h[4 + e] = hb;
        o.l[e] = (_Float16)(a[e] - (float)ha);
        o.l[4 + e] = (_Float16)(b[e] - (float)hb);
    }
    }
    return o;
}
__device__ __forceinline__ void pin_op(Op& o) {
    // f16x8 = 4 VGPRs; keep the split where it was written (see owr::pin)
    typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
    u32x4 a = __builtin_bit_cast(u32x4, o.h), b = __builtin_bit_cast(u32x4, o.l);
    asm volatile("" : "+v"(a), "+v"(b));
    o.h = __builtin_bit_cast(f16x8, a); o.l = __builtin_bit_cast(f16x8, b);
}

// the same for a k-step of which only the first NP (1..3) of the four packed pairs carry channels: (a0 a1)(a2 a3)(b0 b1)(b2 b3);
// the others are zero operands and cost nothing (an odd last channel tile pairs with nothing: NP = 2; a HALF last tile -- see
// pack_hx in owwhip.hip: its registers 2, 3 are padding -- ends the list one pair earlier)
template <int NP>
__device__ __forceinline__ Op split_some(const f32x4 a, const f32x4 b) {
    typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
    typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
    u32x4 h = {0u, 0u, 0u, 0u}, l = {0u, 0u, 0u, 0u};
#pragma unroll
    for (int v = 0; v < NP; ++v) {
        const float x0 = v < 2 ? a[2 * v] : b[2 * v - 4], x1 = v < 2 ? a[2 * v + 1] : b[2 * v - 3];
        const unsigned hp = __builtin_bit_cast(unsigned, f16x2{(_Float16)x0, (_Float16)x1});
        unsigned lp;
        asm("v_fma_mixlo_f16 %0, %1, -1.0, %2 op_sel_hi:[1,0,0]\n\t"
            "v_fma_mixhi_f16 %0, %1, -1.0, %3 op_sel:[1,0,0] op_sel_hi:[1,0,0]"
            : "=&v"(lp) : "v"(hp), "v"(x0), "v"(x1));
        h[v] = hp; l[v] = lp;
    }
    Op o;
    o.h = __builtin_bit_cast(f16x8, h);
    o.l = __builtin_bit_cast(f16x8, l);
    return o;
}

// A k-step that carries ONE full channel tile (an odd last tile of 16 channels: 48 = 32 + 16) leaves half of its 32 k-slots empty.
// The REM2 form uses them for the f16 split itself: the operand pair becomes
//      .l = (xh pairs 0, 1 | xl pairs 0, 1)      against weights (wh | wh):  xh wh + xl wh
//      .h = (xh pairs 0, 1 |  0,  0)             against weights (wl | 0):   xh wl
// i.e. two MFMAs for that k-step instead of three (pack_hx(..., rem2) packs the weight blocks accordingly).  Stage B: 5 instead of 6
// MFMAs per tap, output tile and position tile in its three 48-channel-input layers (756 -> 648 per stream-step).
#ifndef OWH_REM2
#define OWH_REM2 1
#endif
__device__ __forceinline__ Op split_dup(const f32x4 a) {
    typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
    typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
    unsigned hp[2], lp[2];
#pragma unroll
    for (int v = 0; v < 2; ++v) {
        const float x0 = a[2 * v], x1 = a[2 * v + 1];
        hp[v] = __builtin_bit_cast(unsigned, f16x2{(_Float16)x0, (_Float16)x1});
        asm("v_fma_mixlo_f16 %0, %1, -1.0, %2 op_sel_hi:[1,0,0]\n\t"
            "v_fma_mixhi_f16 %0, %1, -1.0, %3 op_sel:[1,0,0] op_sel_hi:[1,0,0]"
            : "=&v"(lp[v]) : "v"(hp[v]), "v"(x0), "v"(x1));
    }
    Op o;
    o.h = __builtin_bit_cast(f16x8, u32x4{hp[0], hp[1], 0u, 0u});
    o.l = __builtin_bit_cast(f16x8, u32x4{hp[0], hp[1], lp[0], lp[1]});
    return o;
}

// operand form of a whole fp32 tile (NCT channel tiles -> KS = ceil(NCT/2) k-steps; an odd last tile pairs with zeros).
// HALF: the last channel tile is a half tile (8 channels in its registers 0, 1).  REM2: see split_dup.
template <int NCT, bool HALF = false, bool REM2 = false>
__device__ __forceinline__ void to_ops(const f32x4 (&t)[NCT], Op (&o)[(NCT + 1) / 2]) {
    constexpr int KS = (NCT + 1) / 2;
    constexpr int NP_LAST = (NCT % 2 ? 2 : 4) - (HALF ? 1 : 0);       // valid pairs of the last k-step
    static_assert(!REM2 || (NCT % 2 == 1 && !HALF), "REM2: an odd, full last channel tile");
#pragma unroll
    for (int k = 0; k < KS; ++k) {
        const f32x4 b = 2 * k + 1 < NCT ? t[2 * k + 1] : f32x4{0.f, 0.f, 0.f, 0.f};
        if (REM2 && k == KS - 1) o[k] = split_dup(t[2 * k]);
        else if (k == KS - 1 && NP_LAST < 4) o[k] = split_some<NP_LAST>(t[2 * k], b);
        else o[k] = split_pair(t[2 * k], b);
        pin_op(o[k]);
    }
}

// register-dump I/O of the f16-split family: like owr::load_tile / store_tile, minus the two padding registers of a half last tile
template <int NCT, bool HALF>
__device__ __forceinline__ void load_tile_h(f32x4 (&t)[NCT], const float* __restrict__ base, int lane) {
#pragma unroll
    for (int ct = 0; ct < NCT; ++ct)
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            if (HALF && ct == NCT - 1 && e >= 2) t[ct][e] = 0.f;
            else t[ct][e] = base[(ct * 4 + e) * 64 + lane];
        }
}
template <int NCT, bool HALF>
__device__ __forceinline__ void store_tile_h(const f32x4 (&t)[NCT], float* __restrict__ base, int lane) {
#pragma unroll
    for (int ct = 0; ct < NCT; ++ct)
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            if (HALF && ct == NCT - 1 && e >= 2) continue;
            base[(ct * 4 + e) * 64 + lane] = t[ct][e];
        }
}

// Position order inside a 16-position tile of stages C, D, E (F = 8, 4, 2 mel positions per stream, SPT = 16 / F streams per tile).
// The fp32 family keeps a stream's positions together (position p = F * stream + mel), so a +-1 mel shift of the per-tap
// accumulators of a 1x3 layer is a DPP row shift by one lane plus a select that zeroes what crossed a stream boundary: 2 extra
// VALU per value and tap.  The f16-split family INTERLEAVES the streams instead: p = SPT * mel + stream.  A mel shift is then a row
// shift by SPT lanes, and exactly the lanes that would read across a stream's edge are the ones the shift leaves without a source:
// DPP bound_ctrl zero-fills them and no select is needed (5.3 -> 2.5 VALU per value in the tap combine of stages C, D, E).
// Everything that maps a lane to a (stream, mel position) goes through tile_stream / tile_mel: the pooled hand-over stores, the
// per-lane participation mask, the debug dump, and owk::reset_kernel on the host side (owh::kInterleave).
#ifndef OWH_INTERLEAVE
#define OWH_INTERLEAVE 1
#endif
constexpr bool kInterleave = OWH_INTERLEAVE != 0;
template <int F> __device__ __forceinline__ int tile_stream(int pos) { return kInterleave ? pos % (16 / F) : pos / F; }
template <int F> __device__ __forceinline__ int tile_mel(int pos) { return kInterleave ? pos / (16 / F) : pos % F; }
template <int F> __device__ __forceinline__ int tile_pos(int stream, int mel) { return kInterleave ? mel * (16 / F) + stream : stream * F + mel; }
// lane p <- x[p - N] / x[p + N] inside the 16-lane row, lanes without a source <- 0
template <int N> __device__ __forceinline__ float dpp_shr_zero(float x) {
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), 0x110 + N, 0xf, 0xf, true));
}
template <int N> __device__ __forceinline__ float dpp_shl_zero(float x) {
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), 0x100 + N, 0xf, 0xf, true));
}
// A sum that becomes an element of an MFMA C operand is a build-vector seed for the SLP vectoriser, which pairs the adds into
// v_pk_add_f32 -- two issue passes all the same, and a form the DPP combiner cannot fold the row shift into (v_mov_b32_dpp stays:
// 8 passes per tile instead of 4).  Passing the scalar through an empty asm keeps it a plain v_add_f32 (dpp): same arithmetic,
// bit-identical results, 5..7 % fewer VALU passes per stage.  Same-box A/B at 131,072 streams (profiles/r05_unpaired_ab.txt):
// stage B -0.9 %, C -1.1 %, D +0.5 %, E +2.8 % (two more spilled registers at three waves per SIMD) -- so B and C (F >= 8) take it.
#ifndef OWH_UNPAIRED_MIN_F
#define OWH_UNPAIRED_MIN_F 8
#endif
template <int F> __device__ __forceinline__ float unpaired(float s) {
    if (F >= OWH_UNPAIRED_MIN_F) asm("" : "+v"(s));
    return s;
}

// debug dump of the f16-split family (cf. owr::dump_tile): a half last tile keeps channel 16 ct + 2j + e in register e < 2
template <int NCT, int F, int C>
__device__ __forceinline__ void dump_tile_ht(const f32x4 (&t)[NCT], float* __restrict__ dbg, size_t stride, int off, int s_first,
                                             int row, int S, int lane, float mul) {
    const int pos = lane & 15, j = lane >> 4;
    const int sp = tile_stream<F>(pos), f = tile_mel<F>(pos), s = s_first + sp;
    if (s >= S) return;
#pragma unroll
    for (int ct = 0; ct < NCT; ++ct)
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            int c = ct * 16 + 4 * j + e;
            if (C % 16 == 8 && ct == NCT - 1) c = e < 2 ? ct * 16 + 2 * j + e : C;
            if (c < C) dbg[(size_t)s * stride + off + (row * F + f) * C + c] = t[ct][e] * mul;      // (mul = 1 / K of the layer: dumps are in true units)
        }
}

__device__ __forceinline__ f16x8 lds_h(const float* buf, int blk, int lane) {
    return *reinterpret_cast<const f16x8*>(buf + (blk * 64 + lane) * 4);
}

#ifndef OWH_WG
#define OWH_WG 4           // waves per workgroup of stages B..E (they share one weight chunk stream through LDS)
#endif
#ifndef OWH_WG_B
#define OWH_WG_B OWH_WG
#endif
#ifndef OWH_WG_C
#define OWH_WG_C OWH_WG
#endif
#ifndef OWH_WG_D
#define OWH_WG_D OWH_WG
#endif
#ifndef OWH_WG_E
#define OWH_WG_E OWH_WG
#endif
#ifndef OWH_WPS_A
#define OWH_WPS_A 3
#endif
#ifndef OWH_WPS_B
#define OWH_WPS_B 3
#endif
#ifndef OWH_WPS_C
#define OWH_WPS_C 2
#endif
#ifndef OWH_WPS_D
#define OWH_WPS_D 2
#endif
#ifndef OWH_WPS_E
#define OWH_WPS_E 3
#endif
#ifndef OWH_RC_RP
#define OWH_RC_RP 4
#endif
#ifndef OWH_RB_RP
#define OWH_RB_RP 4
#endif
using HB = owr::RCfg<24, 48, 4, 16, 1, 2, OWH_RB_RP, OWH_WPS_B>;
using HC = owr::RCfg<48, 72, 4, 8, 2, 2, OWH_RC_RP, OWH_WPS_C>;
using HD = owr::RCfg<72, 96, 2, 4, 1, 2, 2, OWH_WPS_D>;
using HE = owr::RCfg<96, 96, 2, 2, 2, 2, 2, OWH_WPS_E>;

#define OWH_MFMA(a, b, c) __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, 0, 0, 0)

// PRICING BUILDS ONLY (tools/price_handover.sh; results are garbage): what would a fused A -> B launch save at most?  Bit 0 folds
// stage A's hand-over STORES onto 512 streams' rows (4 MB), bit 1 stage B's hand-over LOADS -- the rows then live in L2 and the 8 KB written
// and 8 KB read per stream-step never reach HBM, while every instruction of both kernels still executes.  The time such a build
// saves is the upper bound of a fusion's gain from the hand-over traffic (the launch boundary itself is priced by the 4,096-stream run).
#ifndef OWH_PRICE_HANDOVER
#define OWH_PRICE_HANDOVER 0
#endif
template <bool ON> __device__ __forceinline__ int price_alias(int i) { return ON ? (i & 511) : i; }

// ---- weight-chunk ring of the stage kernels ----------------------------------------------------------------------------------------
// A layer's weights arrive as NCTO chunks (one output-channel tile each: 6..18 KB), L2 -> LDS by global_load_lds, shared by the
// workgroup's waves.  NS slots, P = NS - 1 chunks in flight ahead of the one being consumed; chunks past a layer's end are the first
// chunks of the next layer (w_next), so the stream never drains at a layer boundary.
//  * every slot is its own LDS object: the compiler orders each LDS read behind every pending LDS-DMA write it cannot prove disjoint
//    (s_waitcnt vmcnt(0)).  With one array of two halves (rounds 1-3) a wave waited for the chunk it had JUST issued before it read
//    the current one, in every second chunk step: the prefetch never overlapped the wave's own MFMAs.  Distinct objects carry distinct
//    alias scopes and the wait is gone (B -7 %, C -6 %, D -3 %, E -4 % at 131,072 streams; E -12 % at 4,096).
//  * NS = 2 (large launches): end of a step = s_waitcnt vmcnt(0) + __syncthreads().  Other workgroups of the CU cover the rest.
//  * NS = 3 (small launches, where a workgroup is alone on its CU and each chunk step is ~0.3 us of MFMA work behind a ~1 us
//    L2 -> LDS round trip): end of step i = s_waitcnt vmcnt(K) with K = the DMA instructions this wave issued in step i (chunk
//    i + 2 may stay in flight; VMEM reads return in order, so at most K outstanding means chunk i + 1 has landed -- stores that are
//    still in flight can only make the wait longer) + a BARE s_barrier: __syncthreads() carries a workgroup release fence, which on
//    this target is another vmcnt(0).  Same arithmetic in the same order: results do not depend on NS (tested bit for bit).
// s_waitcnt vmcnt(N) as the BUILTIN (gfx9 encoding: vmcnt in bits 3:0 and 15:14, expcnt 6:4 and lgkmcnt 11:8 left at "no wait"), not inline
// asm: the compiler's own wait-count bookkeeping reads a real S_WAITCNT and stops believing the older LDS-DMA writes are pending --
// behind an asm wait it adds its own (full) vmcnt(0) in front of the next LDS read of a slot it thinks is still being written
template <int N> __device__ __forceinline__ void wait_vmcnt() {
    static_assert(N >= 0 && N < 64, "vmcnt is a 6-bit field");
    __builtin_amdgcn_s_waitcnt((N & 0xF) | ((N >> 4) << 14) | (0x7 << 4) | (0xF << 8));
    asm volatile("" ::: "memory");
}
template <int NS> struct WRing { float* s[5]; };
// s_waitcnt vmcnt(n) for an n that is a compile-time value only AFTER unrolling (a sum over the chunks that may stay in flight): the
// builtin wants an integer constant expression, so the value is matched against 0 .. MAX and the dead branches fold away
template <int MAX> __device__ __forceinline__ void wait_vmcnt_of(int n) {
    if (n >= MAX) wait_vmcnt<MAX>();
    else if constexpr (MAX > 0) wait_vmcnt_of<MAX - 1>(n);
}
// a chunk with the SAME number of DMA instructions in every wave (no branch on the wave index): blocks past the end re-load the last
// block (same source, same destination, same data).  Straight-line issue keeps the compiler's wait counts exact -- behind a branch it
// falls back to vmcnt(0) in front of the next read of that slot -- and makes the counted wait below exact for every wave.
template <int NBLK, int WG>
__device__ __forceinline__ void issue_chunk_uniform(const float* __restrict__ gsrc, float* ldst, int wave, int lane) {
    const unsigned voff = lane * 4;
#pragma unroll
    for (int u = 0; u < (NBLK + WG - 1) / WG; ++u) {
        const int i = min(u * WG + wave, NBLK - 1);
        const float* base = owr::uniform_ptr(gsrc + i * 256);
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(base + voff),
                                         (__attribute__((address_space(3))) void*)(ldst + i * 256), 16, 0, 0);
    }
}
template <int NS, int NBLK, int WG>
__device__ __forceinline__ void ring_issue(const float* __restrict__ gsrc, float* ldst, int wave, int lane) {
    if constexpr (NS == 2) owr::issue_chunk<NBLK, WG>(gsrc, ldst, wave, lane);
    else issue_chunk_uniform<NBLK, WG>(gsrc, ldst, wave, lane);
}
template <int NS, int CH, int NBLK, int NCTO, int NEXT_NBLK, int WG>
__device__ __forceinline__ const float* ring_begin(const WRing<NS>& r, int oct, const float* __restrict__ w, const float* __restrict__ w_next,
                                                   int wave, int lane) {
    constexpr int P = NS - 1;
    const int c = oct + P;                                   // the chunk issued in this step (past NCTO: the next layer's)
    float* dst = r.s[(CH + oct + P) % NS];                   // its slot held chunk oct - 1: every wave passed the last barrier
    if (c < NCTO) ring_issue<NS, NBLK, WG>(w + (size_t)c * NBLK * 256, dst, wave, lane);
    else if (NEXT_NBLK > 0 && c - NCTO < P) ring_issue<NS, (NEXT_NBLK > 0 ? NEXT_NBLK : 1), WG>(w_next + (size_t)(c - NCTO) * NEXT_NBLK * 256, dst, wave, lane);
    return r.s[(CH + oct) % NS];
}
template <int NS, int NBLK, int NCTO, int NEXT_NBLK, int WG>
__device__ __forceinline__ void ring_end(int oct) {
    if (!(oct + 1 < NCTO || NEXT_NBLK > 0)) return;          // no chunk oct + 1 in this launch
    if constexpr (NS == 2) owr::chunk_sync();
    else {
        // chunk oct + 1 must have landed; chunks oct + 2 .. oct + P (issued after it: VMEM returns in order) may stay in flight.  NS = 3:
        // that is the chunk issued in this step; deeper rings (NS = 4, 5: launches of at most ONE workgroup per CU, where nothing else
        // covers the L2 -> LDS round trip and the per-CU DMA rate is what a chunk step costs) leave P - 1 = 2, 3 chunks in flight
        constexpr int P = NS - 1, A = (NBLK + WG - 1) / WG, B = NEXT_NBLK > 0 ? ((NEXT_NBLK > 0 ? NEXT_NBLK : 1) + WG - 1) / WG : 0;
        int n = 0;
#pragma unroll
        for (int c = oct + 2; c <= oct + P; ++c) n += c < NCTO ? A : ((NEXT_NBLK > 0 && c - NCTO < P) ? B : 0);
        wait_vmcnt_of<(P - 1) * (A > B ? A : B)>(n);
        // lgkmcnt(0): this wave's LDS reads of chunk oct have RETURNED before the barrier lets another wave restage that slot in the
        // next step (the MFMAs that consume them -- and the waits in front of those -- may be scheduled behind the bare barrier)
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    }
}

// ---- A operands one k-group ahead (OWH_APIPE) ---------------------------------------------------------------------------------------
// A k-group = the (hi, lo) weight blocks of one (tap, k-step): two ds_read_b128 feeding 3 x NT MFMAs (NT = 2 or 4 position tiles).  Left
// alone the compiler issues a group's reads right in front of the s_waitcnt lgkmcnt(0) that precedes its MFMAs -- in stage D the
// schedule reads  r [wait] MM r [wait] MMMM  : every LDS round trip (~100 cycles) in the open behind 32-64 cycles of MFMA work, at two
// waves per SIMD.  With OWH_APIPE the reads of group g + 1 are issued before the MFMAs of group g: `apipe_step` first makes the CURRENT
// operands a use (an empty asm: the compiler's wait for them lands here), then loads the next pair, and a scheduling barrier keeps both
// in front of the group's MFMAs.  Same MFMAs, same order: bit-identical results.  8 more VGPRs per wave.
// Per stage: bit 0 = the 1x3 (mel) layers, bit 1 = the 3x1 (time) layers.  Same-box alternating runs at 131,072 streams
// (profiles/r06_apipe_ab.txt): stage C with both 1.387 -> 1.351 ms (-2.6 %); everywhere (B, D, E: time layers too) B +5 %, D +7 %, E +9 %
// -- their time layers lose the MFMA : VALU interleave of OWH_PIPE behind the scheduling barriers, D loses its third wave per SIMD
// (172 registers), B and E spill; mel layers only: B, D unchanged (off), E -1.5 % (on).
#ifndef OWH_APIPE_B
#define OWH_APIPE_B 0
#endif
#ifndef OWH_APIPE_C
#define OWH_APIPE_C 3
#endif
#ifndef OWH_APIPE_D
#define OWH_APIPE_D 0
#endif
#ifndef OWH_APIPE_E
#define OWH_APIPE_E 1
#endif
struct APair { f16x8 h, l; };
__device__ __forceinline__ APair apair_load(const float* cur, int blk, int lane) { return APair{lds_h(cur, blk, lane), lds_h(cur, blk + 1, lane)}; }
__device__ __forceinline__ void apair_pin(APair& a) {
    typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
    u32x4 x = __builtin_bit_cast(u32x4, a.h), y = __builtin_bit_cast(u32x4, a.l);
    asm volatile("" : "+v"(x), "+v"(y));
    a.h = __builtin_bit_cast(f16x8, x); a.l = __builtin_bit_cast(f16x8, y);
}
// current pair ready, next pair (block `next_blk`, < 0: none) on its way, both ahead of what follows
__device__ __forceinline__ void apipe_step(APair& curp, APair& nextp, const float* cur, int next_blk, int lane) {
    apair_pin(curp);
    if (next_blk >= 0) nextp = apair_load(cur, next_blk, lane);
    __builtin_amdgcn_sched_barrier(0);
}

// chunk of one output-channel tile: [tap 3][ks KSI][part 2] blocks of 1 KB
// 1x3 (mel) layer: NT tiles in operand form -> NT fp32 D tiles (BatchNorm + activation applied)
template <int KSI, int NCTO, int NT, int F, bool BN, int CH0, int NEXT_NBLK, int WG = OWH_WG, bool HOUT = false, bool REM2 = false, int NS = 2, bool AP = false>
__device__ __forceinline__ void conv_mel_hx(const Op (&in)[NT][KSI], f32x4 (&out)[NT][NCTO], const WRing<NS>& ring,
                                            const float* __restrict__ w, const float* __restrict__ w_next,
                                            const float* __restrict__ init, float cl, int wave, int lane,
                                            lanemask_t& bad) {
    using namespace owr;
    const int pos = lane & 15, j = lane >> 4;
    // (interleaved order: the row shift by SH = 16 / F lanes zero-fills exactly the stream-edge lanes and the masks are not used)
    constexpr int SH = kInterleave ? 16 / F : 1;
    constexpr bool MASKS = F < 16 && !kInterleave;
    const bool first = (pos & (F - 1)) == 0, last = (pos & (F - 1)) == F - 1;
    constexpr int NBLK = 3 * KSI * 2;
#pragma unroll
    for (int oct = 0; oct < NCTO; ++oct) {
        const float* cur = ring_begin<NS, CH0, NBLK, NCTO, NEXT_NBLK, WG>(ring, oct, w, w_next, wave, lane);
        f32x4 res[NT], accs[2][NT];
        APair ap[2];
        if constexpr (AP) ap[0] = apair_load(cur, 0, lane);           // (group 0: tap 0, k-step 0)
#pragma unroll
        for (int ti = 0; ti < 3; ++ti) {                                  // tap order 0, 2, 1 (see conv_mel_lds)
            const int tap = ti == 0 ? 0 : (ti == 1 ? 2 : 1);
            f32x4 acc[NT];
            f32x4 I = {0.f, 0.f, 0.f, 0.f};
            if (ti == 2) I = acc_init(BN ? init : nullptr, oct, j);
#pragma unroll
            for (int t = 0; t < NT; ++t) {
                if (ti < 2) acc[t] = f32x4{0.f, 0.f, 0.f, 0.f};
                else {
                    // the centre chain starts from the folded BatchNorm shift plus the shifted tap-0 sums (one v_add_f32 dpp)
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        if (HOUT && oct == NCTO - 1 && e >= 2) { acc[t][e] = 0.f; continue; }      // padding rows of a half tile
                        const float l = dpp_shr_zero<SH>(accs[0][t][e]);
                        acc[t][e] = unpaired<F>(I[e] + ((MASKS && first) ? 0.f : l));
                    }
                }
            }
#pragma unroll
            for (int ks = 0; ks < KSI; ++ks) {
                f16x8 ah, al;
                if constexpr (AP) {
                    const int g = ti * KSI + ks;                  // position in the issue order (taps 0, 2, 1)
                    const int ntap = ks + 1 < KSI ? tap : (ti == 0 ? 2 : 1), nks = ks + 1 < KSI ? ks + 1 : 0;
                    apipe_step(ap[g & 1], ap[(g + 1) & 1], cur, g + 1 < 3 * KSI ? (ntap * KSI + nks) * 2 : -1, lane);
                    ah = ap[g & 1].h; al = ap[g & 1].l;
                } else {
                    ah = lds_h(cur, (tap * KSI + ks) * 2 + 0, lane);
                    al = lds_h(cur, (tap * KSI + ks) * 2 + 1, lane);
                }
                if (REM2 && ks == KSI - 1) {                      // blocks (wh | wh), (wl | 0) against (xh | xl), (xh | 0): see split_dup
#pragma unroll
                    for (int t = 0; t < NT; ++t) acc[t] = OWH_MFMA(ah, in[t][ks].l, acc[t]);
#pragma unroll
                    for (int t = 0; t < NT; ++t) acc[t] = OWH_MFMA(al, in[t][ks].h, acc[t]);
                    continue;
                }
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
                        if (HOUT && oct == NCTO - 1 && e >= 2) { res[t][e] = 0.f; continue; }
                        const float hh = dpp_shl_zero<SH>(accs[1][t][e]);
                        res[t][e] = acc[t][e] + ((MASKS && last) ? 0.f : hh);
                    }
                }
            }
        }
        if (oct == 0) {
#pragma unroll
            for (int t = 0; t < NT; ++t) nan_guard(bad, res[t][0]);
        }
#pragma unroll
        for (int t = 0; t < NT; ++t) {
            if (HOUT && oct == NCTO - 1) { out[t][oct] = act_t<BN, true>(res[t], cl); pin_t<true>(out[t][oct]); }
            else { out[t][oct] = act_t<BN, false>(res[t], cl); pin_t<false>(out[t][oct]); }
        }
        ring_end<NS, NBLK, NCTO, NEXT_NBLK, WG>(oct);
    }
}

// K-merged form of a 1x3 (mel) layer whose input has a half remainder tile (72 = 64 + 8 channels: stage C layer c, stage D layer a).
// The per-tap accumulator chains keep their KSF full k-steps; the three taps' 8-channel remainders -- each a single operand dword
// per lane -- share ONE k-step that feeds the centre chain directly: its B operand carries, for position p, the remainder channels of
// positions p-1, p, p+1 in dwords 0, 1, 2 (two DPP row shifts per tile and part, done once per layer: the operand is the same for
// every output tile; interleaved position order, so the shifts zero-fill the stream edges themselves).  7 instead of 9 k-steps per
// output tile; the weights come in pack_hx_tm order (same pair index = tap rule as the time layers).
#ifndef OWH_KMERGE_MEL
#define OWH_KMERGE_MEL 1
#endif
// NPR = operand dwords (f16 pairs) of the remainder tile per lane: 1 for a half tile (72 = 64 + 8), 2 for a full one (48 = 32 + 16: the
// three taps' 16-channel remainders fill 2 k-steps instead of 3; 8 more operand registers per tile, so only where the budget has them:
// layer a of stage C, OWH_KMERGE_MEL2)
#ifndef OWH_KMERGE_MEL2
#define OWH_KMERGE_MEL2 1
#endif
template <int KSI, int NT, int F, int NPR>
__device__ __forceinline__ void merge_mel_rems(const Op (&in)[NT][KSI], Op (&M)[NT][(3 * NPR + 3) / 4]) {
    typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
    static_assert(kInterleave || F == 16, "the operand shift relies on the zero fill at the stream edges");
    constexpr int SH = 16 / F, NMK = (3 * NPR + 3) / 4;
#pragma unroll
    for (int t = 0; t < NT; ++t) {
        const u32x4 sh = __builtin_bit_cast(u32x4, in[t][KSI - 1].h), sl = __builtin_bit_cast(u32x4, in[t][KSI - 1].l);
        unsigned h[NMK * 4], l[NMK * 4];
#pragma unroll
        for (int i = 0; i < NMK * 4; ++i) {                   // pair index i = tap * NPR + v (pack_hx_tm)
            const int tap = i / NPR, v = i % NPR;
            if (i >= 3 * NPR) { h[i] = 0u; l[i] = 0u; }
            else if (tap == 0) { h[i] = __builtin_amdgcn_update_dpp(0, sh[v], 0x110 + SH, 0xf, 0xf, true); l[i] = __builtin_amdgcn_update_dpp(0, sl[v], 0x110 + SH, 0xf, 0xf, true); }
            else if (tap == 2) { h[i] = __builtin_amdgcn_update_dpp(0, sh[v], 0x100 + SH, 0xf, 0xf, true); l[i] = __builtin_amdgcn_update_dpp(0, sl[v], 0x100 + SH, 0xf, 0xf, true); }
            else { h[i] = sh[v]; l[i] = sl[v]; }
        }
#pragma unroll
        for (int mk = 0; mk < NMK; ++mk) {
            M[t][mk].h = __builtin_bit_cast(f16x8, u32x4{h[4 * mk], h[4 * mk + 1], h[4 * mk + 2], h[4 * mk + 3]});
            M[t][mk].l = __builtin_bit_cast(f16x8, u32x4{l[4 * mk], l[4 * mk + 1], l[4 * mk + 2], l[4 * mk + 3]});
            pin_op(M[t][mk]);
        }
    }
}
template <int KSI, int NMK, int NCTO, int NT, int F, bool BN, int CH0, int NEXT_NBLK, int WG = OWH_WG, bool HOUT = false, int NS = 2, bool AP = false>
__device__ __forceinline__ void conv_mel_hxm(const Op (&in)[NT][KSI], const Op (&M)[NT][NMK], f32x4 (&out)[NT][NCTO], const WRing<NS>& ring,
                                             const float* __restrict__ w, const float* __restrict__ w_next,
                                             const float* __restrict__ init, float cl, int wave, int lane,
                                             lanemask_t& bad) {
    using namespace owr;
    const int j = lane >> 4;
    constexpr int SH = 16 / F, KSF = KSI - 1;
    constexpr int NBLK = (3 * KSF + NMK) * 2;
#pragma unroll
    for (int oct = 0; oct < NCTO; ++oct) {
        const float* cur = ring_begin<NS, CH0, NBLK, NCTO, NEXT_NBLK, WG>(ring, oct, w, w_next, wave, lane);
        f32x4 res[NT], accs[2][NT];
        APair ap[2];
        if constexpr (AP) ap[0] = apair_load(cur, 0, lane);
        int g = 0;                                                        // (a compile-time value after unrolling)
#pragma unroll
        for (int ti = 0; ti < 3; ++ti) {                                  // tap order 0, 2, 1 (see conv_mel_lds)
            const int tap = ti == 0 ? 0 : (ti == 1 ? 2 : 1);
            f32x4 acc[NT];
            f32x4 I = {0.f, 0.f, 0.f, 0.f};
            if (ti == 2) I = acc_init(BN ? init : nullptr, oct, j);
#pragma unroll
            for (int t = 0; t < NT; ++t) {
                if (ti < 2) acc[t] = f32x4{0.f, 0.f, 0.f, 0.f};
                else {
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        if (HOUT && oct == NCTO - 1 && e >= 2) { acc[t][e] = 0.f; continue; }      // padding rows of a half tile
                        acc[t][e] = unpaired<F>(I[e] + dpp_shr_zero<SH>(accs[0][t][e]));
                    }
                }
            }
#pragma unroll
            for (int ks = 0; ks < KSF + (ti == 2 ? NMK : 0); ++ks) {
                const int mk = ks < KSF ? 0 : ks - KSF;
                const int blk = ks < KSF ? (tap * KSF + ks) * 2 : (3 * KSF + mk) * 2;
                f16x8 ah, al;
                if constexpr (AP) {
                    // next group in issue order: the next k-step of this tap, else k-step 0 of the next tap (2 after 0, 1 after 2);
                    // behind the centre tap's full k-steps come its NMK merged ones
                    const int nsteps = KSF + (ti == 2 ? NMK : 0);
                    int nblk = -1;
                    if (ks + 1 < nsteps) nblk = ks + 1 < KSF ? (tap * KSF + ks + 1) * 2 : (3 * KSF + (ks + 1 - KSF)) * 2;
                    else if (ti < 2) nblk = KSF > 0 ? ((ti == 0 ? 2 : 1) * KSF) * 2 : (3 * KSF) * 2;
                    apipe_step(ap[g & 1], ap[(g + 1) & 1], cur, nblk, lane);
                    ah = ap[g & 1].h; al = ap[g & 1].l;
                    ++g;
                } else {
                    ah = lds_h(cur, blk + 0, lane);
                    al = lds_h(cur, blk + 1, lane);
                }
#pragma unroll
                for (int t = 0; t < NT; ++t) acc[t] = OWH_MFMA(ah, ks < KSF ? in[t][ks].h : M[t][mk].h, acc[t]);
#pragma unroll
                for (int t = 0; t < NT; ++t) acc[t] = OWH_MFMA(ah, ks < KSF ? in[t][ks].l : M[t][mk].l, acc[t]);
#pragma unroll
                for (int t = 0; t < NT; ++t) acc[t] = OWH_MFMA(al, ks < KSF ? in[t][ks].h : M[t][mk].h, acc[t]);
            }
#pragma unroll
            for (int t = 0; t < NT; ++t) {
                if (ti < 2) accs[ti][t] = acc[t];
                else {
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        if (HOUT && oct == NCTO - 1 && e >= 2) { res[t][e] = 0.f; continue; }
                        res[t][e] = acc[t][e] + dpp_shl_zero<SH>(accs[1][t][e]);
                    }
                }
            }
        }
        if (oct == 0) {
#pragma unroll
            for (int t = 0; t < NT; ++t) nan_guard(bad, res[t][0]);
        }
#pragma unroll
        for (int t = 0; t < NT; ++t) {
            if (HOUT && oct == NCTO - 1) { out[t][oct] = act_t<BN, true>(res[t], cl); pin_t<true>(out[t][oct]); }
            else { out[t][oct] = act_t<BN, false>(res[t], cl); pin_t<false>(out[t][oct]); }
        }
        ring_end<NS, NBLK, NCTO, NEXT_NBLK, WG>(oct);
    }
}

// 3x1 (time) layer: rows h0, h1 (history) and in[0..NR-1] in operand form; out row r uses rows r, r+1, r+2.
// Software pipelined over the output-channel tiles: the MFMA chain of tile k is issued interleaved with the BatchNorm /
// activation epilogue of tile k-1 (a wave issues in order, so VALU work only overlaps a MFMA if it sits between two MFMAs
// in program order; OWH_PIPE pins one MFMA : OWH_PIPE_VALU VALU instructions with sched_group_barrier).
#ifndef OWH_PIPE
#define OWH_PIPE 1
#endif
#ifndef OWH_PIPE_VALU
#define OWH_PIPE_VALU 2
#endif
#ifndef OWH_PIPE_M
#define OWH_PIPE_M 0       // the same interleave hint in the K-merged time layers (stage C): same-box A/B 1.555 -> 1.525 ms WITHOUT it
#endif
template <int KSI, int NCTO, int NR, bool BN, int CH0, int NEXT_NBLK, int WG = OWH_WG, bool GUARD = true, bool HOUT = false, bool PIPE = OWH_PIPE != 0, bool REM2 = false, int NS = 2, bool AP = false>
__device__ __forceinline__ void conv_time_hx(const Op (&h0)[KSI], const Op (&h1)[KSI], const Op (&in)[NR][KSI], f32x4 (&out)[NR][NCTO],
                                             const WRing<NS>& ring, const float* __restrict__ w, const float* __restrict__ w_next,
                                             const float* __restrict__ init, float cl, int wave, int lane,
                                             lanemask_t& bad) {
    using namespace owr;
    const int j = lane >> 4;
    constexpr int NBLK = 3 * KSI * 2;
    f32x4 prev[NR];
#pragma unroll
    for (int oct = 0; oct <= NCTO; ++oct) {
        f32x4 acc[NR];
        if (oct < NCTO) {
            const float* cur = ring_begin<NS, CH0, NBLK, NCTO, NEXT_NBLK, WG>(ring, oct, w, w_next, wave, lane);
            const f32x4 I = acc_init(BN ? init : nullptr, oct, j);           // folded BatchNorm shift = the chain's start value
#pragma unroll
            for (int r = 0; r < NR; ++r) acc[r] = I;
            APair ap[2];
            if constexpr (AP) ap[0] = apair_load(cur, 0, lane);
#pragma unroll
            for (int tap = 0; tap < 3; ++tap)
#pragma unroll
                for (int ks = 0; ks < KSI; ++ks) {
                    f16x8 ah, al;
                    if constexpr (AP) {
                        const int g = tap * KSI + ks;
                        apipe_step(ap[g & 1], ap[(g + 1) & 1], cur, g + 1 < 3 * KSI ? (g + 1) * 2 : -1, lane);
                        ah = ap[g & 1].h; al = ap[g & 1].l;
                    } else {
                        ah = lds_h(cur, (tap * KSI + ks) * 2 + 0, lane);
                        al = lds_h(cur, (tap * KSI + ks) * 2 + 1, lane);
                    }
                    const bool rem = REM2 && ks == KSI - 1;       // two MFMAs: (wh | wh) x (xh | xl), (wl | 0) x (xh | 0)
#pragma unroll
                    for (int part = 0; part < (rem ? 2 : 3); ++part)
#pragma unroll
                        for (int r = 0; r < NR; ++r) {
                            const int src = r + tap;
                            const int ri = src >= 2 ? src - 2 : 0;
                            const Op& b = src == 0 ? h0[ks] : (src == 1 ? h1[ks] : in[ri][ks]);
                            if (rem) acc[r] = OWH_MFMA(part == 0 ? ah : al, part == 0 ? b.l : b.h, acc[r]);
                            else acc[r] = OWH_MFMA(part == 2 ? al : ah, part == 1 ? b.l : b.h, acc[r]);
                        }
                }
        }
        if (oct > 0) {                                       // epilogue of the previous tile
            if (GUARD && oct == 1) {
#pragma unroll
                for (int r = 0; r < NR; ++r) nan_guard(bad, prev[r][0]);
            }
#pragma unroll
            for (int r = 0; r < NR; ++r) {
                if (HOUT && oct == NCTO) { out[r][oct - 1] = act_t<BN, true>(prev[r], cl); pin_t<true>(out[r][oct - 1]); }
                else { out[r][oct - 1] = act_t<BN, false>(prev[r], cl); pin_t<false>(out[r][oct - 1]); }
            }
        }
        if (PIPE && !AP && oct > 0 && oct < NCTO) {                // (the A-operand pipeline's scheduling barriers fix the order themselves)
#pragma unroll
            for (int i = 0; i < (9 * KSI - (REM2 ? 3 : 0)) * NR; ++i) {
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                __builtin_amdgcn_sched_group_barrier(0x002, OWH_PIPE_VALU, 0);
            }
        }
        if (oct < NCTO) {
#pragma unroll
            for (int r = 0; r < NR; ++r) prev[r] = acc[r];
                ring_end<NS, NBLK, NCTO, NEXT_NBLK, WG>(oct);
        }
    }
}

// ------------------------------------------------------------------------------------------------
// K-merged form of the 3x1 (time) layers of stages whose channel count is not a multiple of 32 (B: 48 = 32 + 16, C: 72 = 64 + 8).
// A k-step carries 32 input channels; with one k-step sequence per tap the remainder tile of every tap costs a whole, mostly
// empty k-step (B: 16 of 32 slots, C: 8 of 32).  The K order of a convolution is free, and for a TIME conv all three taps feed the
// same accumulator, so the remainder tiles of the three taps are packed together: B 3 x 16 channels -> 2 k-steps instead of 3,
// C 3 x 8 -> 1 instead of 3 (C: 7 k-steps per output tile instead of 9, B: 5 instead of 6: -22 % / -17 % MFMAs and weight bytes in
// these layers).  Operand dwords (f16 pairs) of the remainder tiles are copied into the merged operands in the order
// pair index = tap * NPR + v  (NPR = pairs per remainder tile: 2, or 1 for a half tile); pack_hx_tm (owwhip.hip) packs the weights
// in the same order.  The 1x3 (mel) layers keep one accumulator chain per tap and cannot merge.
// ------------------------------------------------------------------------------------------------
#ifndef OWH_KMERGE
#define OWH_KMERGE 1
#endif
#ifndef OWH_KMERGE_B
#define OWH_KMERGE_B 0     // stage B too (48 = 32 + 16 channels: 2 merged k-steps per OUTPUT row cost 16 more registers than the 6 source
                           // rows' separate remainder k-steps -> spills at 3 waves per SIMD); stage C (72 = 64 + 8) always
#endif
template <int NCT, bool HALF> struct TimeK {
    static constexpr int KSF = NCT / 2;                         // full k-steps per tap
    static constexpr int NPR = NCT % 2 ? (HALF ? 1 : 2) : 0;    // f16 pairs of the remainder tile
    static constexpr int NMK = (3 * NPR + 3) / 4;               // merged k-steps for the three taps' remainders
    static constexpr int NBLK = (3 * KSF + NMK) * 2;            // 1 KB blocks per output-channel tile
};

// operand form of a row for a time conv: the full k-steps, and the remainder tile's NPR pairs as bare dwords (hi, lo)
struct RemPairs { unsigned h[2], l[2]; };
template <int NCT, bool HALF>
__device__ __forceinline__ void to_ops_time(const f32x4 (&t)[NCT], Op (&full)[TimeK<NCT, HALF>::KSF], RemPairs& rem) {
    typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
#pragma unroll
    for (int k = 0; k < NCT / 2; ++k) { full[k] = split_pair(t[2 * k], t[2 * k + 1]); pin_op(full[k]); }
    const Op o = split_some<TimeK<NCT, HALF>::NPR>(t[NCT - 1], f32x4{0.f, 0.f, 0.f, 0.f});
    const u32x4 oh = __builtin_bit_cast(u32x4, o.h), ol = __builtin_bit_cast(u32x4, o.l);
#pragma unroll
    for (int v = 0; v < 2; ++v) { rem.h[v] = v < TimeK<NCT, HALF>::NPR ? oh[v] : 0u; rem.l[v] = v < TimeK<NCT, HALF>::NPR ? ol[v] : 0u; }
}
// merged operands of output row r from the remainders of rows r, r+1, r+2 (rows 0, 1 = history)
template <int NR, int NPR, int NMK>
__device__ __forceinline__ void merge_rems(const RemPairs (&rem)[NR + 2], Op (&M)[NR][NMK]) {
    typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
#pragma unroll
    for (int r = 0; r < NR; ++r) {
        // (scalar arrays with compile-time indices after unrolling: a vector element written through a loop index is not promoted to
        //  registers reliably and ended up in scratch)
        unsigned h[NMK * 4], l[NMK * 4];
#pragma unroll
        for (int i = 0; i < NMK * 4; ++i) {
            const int tap = i / NPR, v = i % NPR;
            h[i] = i < 3 * NPR ? rem[r + (tap < 3 ? tap : 0)].h[v] : 0u;
            l[i] = i < 3 * NPR ? rem[r + (tap < 3 ? tap : 0)].l[v] : 0u;
        }
#pragma unroll
        for (int mk = 0; mk < NMK; ++mk) {
            M[r][mk].h = __builtin_bit_cast(f16x8, u32x4{h[4 * mk], h[4 * mk + 1], h[4 * mk + 2], h[4 * mk + 3]});
            M[r][mk].l = __builtin_bit_cast(f16x8, u32x4{l[4 * mk], l[4 * mk + 1], l[4 * mk + 2], l[4 * mk + 3]});
            pin_op(M[r][mk]);
        }
    }
}

template <int KSF, int NMK, int NCTO, int NR, bool BN, int CH0, int NEXT_NBLK, int WG = OWH_WG, bool HOUT = false, int NS = 2, bool AP = false>
__device__ __forceinline__ void conv_time_hxm(const Op (&h0)[KSF], const Op (&h1)[KSF], const Op (&in)[NR][KSF], const Op (&M)[NR][NMK],
                                              f32x4 (&out)[NR][NCTO], const WRing<NS>& ring, const float* __restrict__ w, const float* __restrict__ w_next,
                                              const float* __restrict__ init, float cl, int wave, int lane,
                                              lanemask_t& bad) {
    using namespace owr;
    const int j = lane >> 4;
    constexpr int NBLK = (3 * KSF + NMK) * 2;
    f32x4 prev[NR];
#pragma unroll
    for (int oct = 0; oct <= NCTO; ++oct) {
        f32x4 acc[NR];
        if (oct < NCTO) {
            const float* cur = ring_begin<NS, CH0, NBLK, NCTO, NEXT_NBLK, WG>(ring, oct, w, w_next, wave, lane);
            const f32x4 I = acc_init(BN ? init : nullptr, oct, j);
#pragma unroll
            for (int r = 0; r < NR; ++r) acc[r] = I;
            APair ap[2];
            if constexpr (AP) ap[0] = apair_load(cur, 0, lane);  // groups in block order: 3 KSF full k-steps, then NMK merged ones
#pragma unroll
            for (int tap = 0; tap < 3; ++tap)
#pragma unroll
                for (int ks = 0; ks < KSF; ++ks) {
                    f16x8 ah, al;
                    if constexpr (AP) {
                        const int g = tap * KSF + ks;
                        apipe_step(ap[g & 1], ap[(g + 1) & 1], cur, g + 1 < 3 * KSF + NMK ? (g + 1) * 2 : -1, lane);
                        ah = ap[g & 1].h; al = ap[g & 1].l;
                    } else {
                        ah = lds_h(cur, (tap * KSF + ks) * 2 + 0, lane);
                        al = lds_h(cur, (tap * KSF + ks) * 2 + 1, lane);
                    }
#pragma unroll
                    for (int part = 0; part < 3; ++part)
#pragma unroll
                        for (int r = 0; r < NR; ++r) {
                            const int src = r + tap;
                            const int ri = src >= 2 ? src - 2 : 0;
                            const Op& b = src == 0 ? h0[ks] : (src == 1 ? h1[ks] : in[ri][ks]);
                            acc[r] = OWH_MFMA(part == 2 ? al : ah, part == 1 ? b.l : b.h, acc[r]);
                        }
                }
#pragma unroll
            for (int mk = 0; mk < NMK; ++mk) {
                f16x8 ah, al;
                if constexpr (AP) {
                    const int g = 3 * KSF + mk;
                    apipe_step(ap[g & 1], ap[(g + 1) & 1], cur, g + 1 < 3 * KSF + NMK ? (g + 1) * 2 : -1, lane);
                    ah = ap[g & 1].h; al = ap[g & 1].l;
                } else {
                    ah = lds_h(cur, (3 * KSF + mk) * 2 + 0, lane);
                    al = lds_h(cur, (3 * KSF + mk) * 2 + 1, lane);
                }
#pragma unroll
                for (int part = 0; part < 3; ++part)
#pragma unroll
                    for (int r = 0; r < NR; ++r) acc[r] = OWH_MFMA(part == 2 ? al : ah, part == 1 ? M[r][mk].l : M[r][mk].h, acc[r]);
            }
        }
        if (oct > 0) {                                       // epilogue of the previous tile
            if (oct == 1) {
#pragma unroll
                for (int r = 0; r < NR; ++r) nan_guard(bad, prev[r][0]);
            }
#pragma unroll
            for (int r = 0; r < NR; ++r) {
                if (HOUT && oct == NCTO) { out[r][oct - 1] = act_t<BN, true>(prev[r], cl); pin_t<true>(out[r][oct - 1]); }
                else { out[r][oct - 1] = act_t<BN, false>(prev[r], cl); pin_t<false>(out[r][oct - 1]); }
            }
        }
#if OWH_PIPE_M
        if (oct > 0 && oct < NCTO) {
#pragma unroll
            for (int i = 0; i < 3 * (3 * KSF + NMK) * NR; ++i) {
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                __builtin_amdgcn_sched_group_barrier(0x002, OWH_PIPE_VALU, 0);
            }
        }
#endif
        if (oct < NCTO) {
#pragma unroll
            for (int r = 0; r < NR; ++r) prev[r] = acc[r];
                ring_end<NS, NBLK, NCTO, NEXT_NBLK, WG>(oct);
        }
    }
}

// ------------------------------------------------------------------------------------------------
// stages B..E (parameters, geometry and memory layouts: owr::RStageParams / owr::RCfg, channel tiles NOT re-packed)
// ------------------------------------------------------------------------------------------------
#ifndef OWH_KMERGE_MEL2B
#define OWH_KMERGE_MEL2B 0     // layer c of stage B (48 -> 48) in the same form: see DESIGN.md 5.2 for the measurement
#endif
#ifndef OWH_HIST_LDS
#define OWH_HIST_LDS 1
#endif
#ifndef OWH_HIST_LDS_C
#define OWH_HIST_LDS_C 0       // stage C too: measured +1.5 % on C (same box: 1.496 -> 1.520 ms) against -2.5 % on B (1.378 -> 1.345)
#endif
// Wave priority by layer inside the stage kernels (s_setprio; profiles/r06_prio_ab.txt).  A code of four digits = the levels of conv a, b, c, d
// (0 = off; the prologue's loads and the pooled store run at level 0).  The two or three workgroups of a CU are at different layers at any
// time; with FALLING levels (3210) a workgroup that has just started -- the youngest wave of its SIMD, served last by the age-ordered arbiter
// -- gets through its first layers ahead of the others and the SIMD's waves spread over the layers instead of bunching behind the oldest.
// Same-box alternating runs, bit-identical: C -1 %, D -2 % with 3210; B -0.6 %, E -1 % with one level above the load / store phases (1111);
// falling levels in B cost it +9 %, a high prologue costs everywhere.  Together -0.4 % of the step (5.59-5.61 -> 5.57-5.58 ms).
#ifndef OWH_SPRIO_B
#define OWH_SPRIO_B 1111
#endif
#ifndef OWH_SPRIO_C
#define OWH_SPRIO_C 3210
#endif
#ifndef OWH_SPRIO_D
#define OWH_SPRIO_D 3210
#endif
#ifndef OWH_SPRIO_E
#define OWH_SPRIO_E 1111
#endif
template <int CODE, int LAYER> __device__ __forceinline__ void sprio() {
    constexpr int div = LAYER == 0 ? 1000 : LAYER == 1 ? 100 : LAYER == 2 ? 10 : 1;
    if constexpr (CODE != 0 && LAYER >= 0) __builtin_amdgcn_s_setprio((CODE / div) % 10);
    if constexpr (CODE != 0 && LAYER < 0) __builtin_amdgcn_s_setprio(0);
}
// NBLK KB of a wave's own history block, HBM -> LDS (linear image: register-dump rows of 256 B, four rows per DMA instruction)
template <int NBLK>
__device__ __forceinline__ void issue_hist(const float* __restrict__ gsrc, float* ldst, int lane) {
    const float* base = owr::uniform_ptr(gsrc);
#pragma unroll
    for (int i = 0; i < NBLK; ++i)
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(base + i * 256 + lane * 4),
                                         (__attribute__((address_space(3))) void*)(ldst + i * 256), 16, 0, 0);
}
template <int NCT, bool HALF>
__device__ __forceinline__ void load_tile_lds(f32x4 (&t)[NCT], const float* base, int lane) {
#pragma unroll
    for (int ct = 0; ct < NCT; ++ct)
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            if (HALF && ct == NCT - 1 && e >= 2) t[ct][e] = 0.f;
            else t[ct][e] = base[(ct * 4 + e) * 64 + lane];
        }
}

template <class C, bool LAST, bool DBG, int WG = OWH_WG, int NS = 2>
__global__ __launch_bounds__(64 * WG, (DBG || NS > 3 ? 1 : (NS == 3 && C::WPS > 2 ? 2 : C::WPS))) void hstage_kernel(owr::RStageParams p) {
    using namespace owr;
    constexpr int NCTI = C::NCTI, NCT = C::NCT, R = C::RP, F = C::F;   // R = rows per pass (see owr::RCfg::RP)
    static_assert(!LAST || C::NPASS == 1, "the last stage runs in one pass");
    // every pass starts at ring phase CH0 = 0 (conv a) and advances the ring by 4 * NCT chunks: with more than one pass the slot the
    // chunk prefetched by conv d lands in is the one conv a of the next pass reads only if that advance is a multiple of the ring length
    static_assert(NS == 2 || C::NPASS == 1 || (4 * NCT) % NS == 0, "multi-pass stages need a ring phase that returns to 0 after each pass");
    static_assert(NS >= 2 && NS <= 5 && NS - 1 <= NCT, "the chunks in flight at a layer boundary belong to at most the next layer");
    constexpr int KSA = (NCTI + 1) / 2, KS = (NCT + 1) / 2;          // k-steps per tap: first layer / other layers
    constexpr int NBA = 3 * KSA * 2, NB = 3 * KS * 2;                // 1 KB blocks per chunk
    using TK = TimeK<NCT, C::HOUT>;
    constexpr bool MERGE = OWH_KMERGE && (NCT % 2 == 1) && !LAST && (C::HOUT || OWH_KMERGE_B);   // time layers in the K-merged form
    constexpr int NBT = MERGE ? TK::NBLK : NB;                       // blocks per chunk of the 3x1 layers
    constexpr bool PIPE = OWH_PIPE != 0;
    constexpr int APCFG = NCT == 3 ? OWH_APIPE_B : NCT == 5 ? OWH_APIPE_C : LAST ? OWH_APIPE_E : OWH_APIPE_D;
    constexpr bool APM = (APCFG & 1) != 0, APT = (APCFG & 2) != 0;       // A operands one k-group ahead (apipe_step): mel / time layers
    constexpr int SPCODE = NCT == 3 ? OWH_SPRIO_B : NCT == 5 ? OWH_SPRIO_C : LAST ? OWH_SPRIO_E : OWH_SPRIO_D;   // wave priority by layer (sprio)
    // a full odd last channel tile (stage B: 48 = 32 + 16) in the two-MFMA remainder form of split_dup
    constexpr bool REM2 = OWH_REM2 && NCT % 2 == 1 && !C::HOUT && !LAST && !(OWH_KMERGE && OWH_KMERGE_B) && !(OWH_KMERGE_MEL && OWH_KMERGE_MEL2B);
    // 1x3 layers whose 72-channel input leaves a half remainder tile, in the K-merged form (conv_mel_hxm): layer a of stage D, c of C
    constexpr bool MMA2 = OWH_KMERGE_MEL && OWH_KMERGE_MEL2 && kInterleave && !C::HIN && NCTI % 2 == 1 && NCTI >= 3 && C::WPS == 2;   // C layer a (48 in)
    constexpr bool MMA = (OWH_KMERGE_MEL && kInterleave && C::HIN && NCTI % 2 == 1 && NCTI >= 3) || MMA2;
    constexpr int NPRA = MMA2 ? 2 : 1, NMKA = (3 * NPRA + 3) / 4;
    constexpr bool MMC2 = OWH_KMERGE_MEL && OWH_KMERGE_MEL2B && !C::HOUT && NCT % 2 == 1 && NCT >= 3 && !LAST;      // B layer c
    constexpr bool MMC = (OWH_KMERGE_MEL && kInterleave && C::HOUT && NCT % 2 == 1 && NCT >= 3) || MMC2;
    constexpr int NPRC = MMC2 ? 2 : 1, NMKC = (3 * NPRC + 3) / 4;
    constexpr int NBAM = MMA ? (3 * (KSA - 1) + NMKA) * 2 : NBA, NBCM = MMC ? (3 * (KS - 1) + NMKC) * 2 : NB;
    // the largest weight chunk of THIS stage (blocks of 256 floats): the double buffer is sized per stage, so that stages B and C have
    // LDS left for their conv histories (below)
    constexpr int NBMAX = (NBAM > NBT ? NBAM : NBT) > (NBCM > (LAST ? NB : 0) ? NBCM : (LAST ? NB : 0)) ? (NBAM > NBT ? NBAM : NBT) : (NBCM > (LAST ? NB : 0) ? NBCM : (LAST ? NB : 0));
    constexpr int WBS = NBMAX * 256;
    // HLDS: the two history rows of each 3x1 layer are fetched HBM -> LDS by the DMA path (global_load_lds: no VGPRs, one pass
    // through the L2 -> CU path) while the preceding 1x3 layer computes, instead of by register loads at the layer boundary where
    // their latency is exposed with only 2-3 waves per SIMD.  Stage B: 6 KB per wave, -2.5 % (same-box A/B); stage C (10 KB per wave,
    // OWH_HIST_LDS_C) measured slower; D and E have no LDS left for it at three workgroups per CU.
    constexpr bool HLDS = OWH_HIST_LDS && !LAST && C::NPASS == 1 && (C::SPT == 1 || (C::SPT == 2 && OWH_HIST_LDS_C));
    constexpr int HROW = NCT * 4 * 64;                                // floats of one history row of a wave's group
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    int g = blockIdx.x * WG + wave;
    // the slots of the weight ring are DISTINCT LDS objects (see WRing above)
    __shared__ __attribute__((aligned(16))) float wbuf[WBS];
    __shared__ __attribute__((aligned(16))) float wbuf1[WBS];
    __shared__ __attribute__((aligned(16))) float wbuf2[NS >= 3 ? WBS : 4];
    __shared__ __attribute__((aligned(16))) float wbuf3[NS >= 4 ? WBS : 4];
    __shared__ __attribute__((aligned(16))) float wbuf4[NS >= 5 ? WBS : 4];
    const WRing<NS> ring{{wbuf, wbuf1, wbuf2, wbuf3, wbuf4}};
    __shared__ __attribute__((aligned(16))) float sbn[4][NCT * 16];      // per layer: K * BatchNorm shift in tile row order = accumulator start values
    __shared__ __attribute__((aligned(16))) float hlds[HLDS ? WG * 2 * HROW : 4];
    float* const hl = hlds + (HLDS ? wave * 2 * HROW : 0);
    const bool active = g < p.n_groups;
    if (!active) g = p.n_groups - 1;
    if (p.glist) g = p.glist[g];          // a masked step with few participants runs only the groups that hold one (owwhip.hip: build_active_lists)
    else g += p.g_base;                   // block-pipelined step: this launch covers groups g_base .. g_base + n_groups - 1
    lanemask_t bad = 0;
    ring_issue<NS, NBAM, WG>(p.w[0], wbuf, wave, lane);
    if (NS >= 3) ring_issue<NS, NBAM, WG>(p.w[0] + (size_t)NBAM * 256, wbuf1, wave, lane);   // (NS - 1 chunks in flight from the start)
    if (NS >= 4) ring_issue<NS, NBAM, WG>(p.w[0] + (size_t)2 * NBAM * 256, wbuf2, wave, lane);
    if (NS >= 5) ring_issue<NS, NBAM, WG>(p.w[0] + (size_t)3 * NBAM * 256, wbuf3, wave, lane);
    for (int i = threadIdx.x; i < 4 * NCT * 16; i += 64 * WG) {
        const int l = i / (NCT * 16), c = i % (NCT * 16);
        sbn[l][c] = p.shift[l][c];
    }
    const int s_first = g * C::SPT;
    // masked steps (oww_step_masked): a stream that sits this step out is computed like any other (the workgroup shares its weight
    // stream) but none of its state, hand-over or output is stored; per lane, because a tile holds 16 / F streams
    const bool lane_on = active && (p.stream_on == nullptr || p.stream_on[min(s_first + tile_stream<F>(lane & 15), p.S - 1)] != 0);
    float* hb = p.hist_b + (size_t)g * C::HIST_FLOATS;
    float* hd = p.hist_d + (size_t)g * C::HIST_FLOATS;
    if (HLDS) issue_hist<2 * NCT>(hb, hl, lane);                      // lands before the first chunk_sync (vmcnt(0)) at the latest

#pragma unroll
    for (int pass = 0; pass < C::NPASS; ++pass) {
    f32x4 Y[R][NCT];
    Op Xo[R][KSA];
#pragma unroll
    for (int r = 0; r < R; ++r) {
        f32x4 X[NCTI];
        load_tile_h<NCTI, C::HIN>(X, p.xin + ((size_t)price_alias<(OWH_PRICE_HANDOVER & 2) != 0 && C::CIN == 24>(g) * C::R + pass * R + r) * (NCTI * 4 * 64), lane);
        to_ops<NCTI, C::HIN>(X, Xo[r]);
    }
    if (pass == 0) chunk_sync();
    sprio<SPCODE, 0>();

    // conv a: 1x3, CIN -> C
    if constexpr (MMA) {
        Op Mx[R][NMKA];
        merge_mel_rems<KSA, R, F, NPRA>(Xo, Mx);
        conv_mel_hxm<KSA, NMKA, NCT, R, F, true, 0, NBT, WG, C::HOUT, NS, APM>(Xo, Mx, Y, ring, p.w[0], p.w[1], sbn[0], p.clampv[0], wave, lane, bad);
    } else
    conv_mel_hx<KSA, NCT, R, F, true, 0, NBT, WG, C::HOUT, false, NS, APM>(Xo, Y, ring, p.w[0], p.w[1], sbn[0], p.clampv[0], wave, lane, bad);
    if (DBG && p.dbg && active) {
#pragma unroll
        for (int r = 0; r < R; ++r) dump_tile_ht<NCT, F, C::C>(Y[r], p.dbg, p.dbg_stride, p.dbg_off[0], s_first, pass * R + r, p.S, lane, p.dbg_mul[0]);
    }
    Op Ao[R][KS], H0[KS], H1[KS];
    sprio<SPCODE, 1>();
    if constexpr (MERGE) {
        // conv b: 3x1 over [hist_b(2) ; Ya], K-merged form
        Op AoF[R][TK::KSF], H0F[TK::KSF], H1F[TK::KSF], M[R][TK::NMK];
        RemPairs rem[R + 2];
        {
            f32x4 T0[NCT], T1[NCT];
            if (HLDS) { load_tile_lds<NCT, C::HOUT>(T0, hl, lane); load_tile_lds<NCT, C::HOUT>(T1, hl + HROW, lane); }
            else { load_tile_h<NCT, C::HOUT>(T0, hb, lane); load_tile_h<NCT, C::HOUT>(T1, hb + NCT * 4 * 64, lane); }
            to_ops_time<NCT, C::HOUT>(T0, H0F, rem[0]);
            to_ops_time<NCT, C::HOUT>(T1, H1F, rem[1]);
        }
        if (HLDS) { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); issue_hist<2 * NCT>(hd, hl, lane); }   // the slot is free again: conv d's history flies during conv b and c
        if (lane_on) {
            store_tile_h<NCT, C::HOUT>(Y[R - 2], hb, lane);
            store_tile_h<NCT, C::HOUT>(Y[R - 1], hb + NCT * 4 * 64, lane);
        }
#pragma unroll
        for (int r = 0; r < R; ++r) to_ops_time<NCT, C::HOUT>(Y[r], AoF[r], rem[2 + r]);
        merge_rems<R, TK::NPR, TK::NMK>(rem, M);
        __builtin_amdgcn_sched_barrier(0);
        conv_time_hxm<TK::KSF, TK::NMK, NCT, R, true, NCT, NBCM, WG, C::HOUT, NS, APT>(H0F, H1F, AoF, M, Y, ring, p.w[1], p.w[2], sbn[1], p.clampv[1], wave, lane, bad);
    } else {
    {
        f32x4 T0[NCT], T1[NCT];
        if (HLDS) { load_tile_lds<NCT, C::HOUT>(T0, hl, lane); load_tile_lds<NCT, C::HOUT>(T1, hl + HROW, lane); }
        else { load_tile_h<NCT, C::HOUT>(T0, hb, lane); load_tile_h<NCT, C::HOUT>(T1, hb + NCT * 4 * 64, lane); }
        to_ops<NCT, C::HOUT, REM2>(T0, H0);
        to_ops<NCT, C::HOUT, REM2>(T1, H1);
    }
    if (HLDS) { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); issue_hist<2 * NCT>(hd, hl, lane); }
    if (lane_on) {
        store_tile_h<NCT, C::HOUT>(Y[R - 2], hb, lane);
        store_tile_h<NCT, C::HOUT>(Y[R - 1], hb + NCT * 4 * 64, lane);
    }
#pragma unroll
    for (int r = 0; r < R; ++r) to_ops<NCT, C::HOUT, REM2>(Y[r], Ao[r]);
    __builtin_amdgcn_sched_barrier(0);
    // conv b: 3x1 over [hist_b(2) ; Ya]
    conv_time_hx<KS, NCT, R, true, NCT, NBCM, WG, true, C::HOUT, PIPE, REM2, NS, APT>(H0, H1, Ao, Y, ring, p.w[1], p.w[2], sbn[1], p.clampv[1], wave, lane, bad);
    }
    if (DBG && p.dbg && active) {
#pragma unroll
        for (int r = 0; r < R; ++r) dump_tile_ht<NCT, F, C::C>(Y[r], p.dbg, p.dbg_stride, p.dbg_off[1], s_first, pass * R + r, p.S, lane, p.dbg_mul[1]);
    }
#pragma unroll
    for (int r = 0; r < R; ++r) to_ops<NCT, C::HOUT, REM2>(Y[r], Ao[r]);
    __builtin_amdgcn_sched_barrier(0);
    sprio<SPCODE, 2>();
    // conv c: 1x3
    if constexpr (MMC) {
        Op Mc[R][NMKC];
        merge_mel_rems<KS, R, F, NPRC>(Ao, Mc);
        conv_mel_hxm<KS, NMKC, NCT, R, F, true, 2 * NCT, NBT, WG, C::HOUT, NS, APM>(Ao, Mc, Y, ring, p.w[2], p.w[3], sbn[2], p.clampv[2], wave, lane, bad);
    } else
    conv_mel_hx<KS, NCT, R, F, true, 2 * NCT, NBT, WG, C::HOUT, REM2, NS, APM>(Ao, Y, ring, p.w[2], p.w[3], sbn[2], p.clampv[2], wave, lane, bad);
    if (DBG && p.dbg && active) {
#pragma unroll
        for (int r = 0; r < R; ++r) dump_tile_ht<NCT, F, C::C>(Y[r], p.dbg, p.dbg_stride, p.dbg_off[2], s_first, pass * R + r, p.S, lane, p.dbg_mul[2]);
    }
    sprio<SPCODE, 3>();
    if constexpr (MERGE) {
        // conv d: 3x1 over [hist_d(2) ; Yc], K-merged form
        Op AoF[R][TK::KSF], H0F[TK::KSF], H1F[TK::KSF], M[R][TK::NMK];
        RemPairs rem[R + 2];
        {
            f32x4 T0[NCT], T1[NCT];
            if (HLDS) { load_tile_lds<NCT, C::HOUT>(T0, hl, lane); load_tile_lds<NCT, C::HOUT>(T1, hl + HROW, lane); }
            else { load_tile_h<NCT, C::HOUT>(T0, hd, lane); load_tile_h<NCT, C::HOUT>(T1, hd + NCT * 4 * 64, lane); }
            to_ops_time<NCT, C::HOUT>(T0, H0F, rem[0]);
            to_ops_time<NCT, C::HOUT>(T1, H1F, rem[1]);
        }
        if (lane_on) {
            store_tile_h<NCT, C::HOUT>(Y[R - 2], hd, lane);
            store_tile_h<NCT, C::HOUT>(Y[R - 1], hd + NCT * 4 * 64, lane);
        }
#pragma unroll
        for (int r = 0; r < R; ++r) to_ops_time<NCT, C::HOUT>(Y[r], AoF[r], rem[2 + r]);
        merge_rems<R, TK::NPR, TK::NMK>(rem, M);
        __builtin_amdgcn_sched_barrier(0);
        conv_time_hxm<TK::KSF, TK::NMK, NCT, R, true, 3 * NCT, (C::NPASS > 1 ? NBAM : 0), WG, C::HOUT, NS, APT>(H0F, H1F, AoF, M, Y, ring, p.w[3], p.w[0], sbn[3], p.clampv[3], wave, lane, bad);
    } else {
    {
        f32x4 T0[NCT], T1[NCT];
        if (HLDS) { load_tile_lds<NCT, C::HOUT>(T0, hl, lane); load_tile_lds<NCT, C::HOUT>(T1, hl + HROW, lane); }
        else { load_tile_h<NCT, C::HOUT>(T0, hd, lane); load_tile_h<NCT, C::HOUT>(T1, hd + NCT * 4 * 64, lane); }
        to_ops<NCT, C::HOUT, REM2>(T0, H0);
        to_ops<NCT, C::HOUT, REM2>(T1, H1);
    }
    if (lane_on) {
        store_tile_h<NCT, C::HOUT>(Y[R - 2], hd, lane);
        store_tile_h<NCT, C::HOUT>(Y[R - 1], hd + NCT * 4 * 64, lane);
    }
#pragma unroll
    for (int r = 0; r < R; ++r) to_ops<NCT, C::HOUT, REM2>(Y[r], Ao[r]);
    __builtin_amdgcn_sched_barrier(0);
    // conv d: 3x1 over [hist_d(2) ; Yc]
    conv_time_hx<KS, NCT, R, true, 3 * NCT, (LAST ? NB : (C::NPASS > 1 ? NBAM : 0)), WG, true, C::HOUT, PIPE, REM2, NS, APT>(H0, H1, Ao, Y, ring, p.w[3], LAST ? p.w19 : p.w[0], sbn[3], p.clampv[3], wave, lane, bad);
    }
    if (DBG && p.dbg && active) {
#pragma unroll
        for (int r = 0; r < R; ++r) dump_tile_ht<NCT, F, C::C>(Y[r], p.dbg, p.dbg_stride, p.dbg_off[3], s_first, pass * R + r, p.S, lane, p.dbg_mul[3]);
    }

    sprio<SPCODE, -1>();
    if (!LAST && lane_on) {
        constexpr int FO = C::FO, RO = C::RO;
        constexpr int SPTN = 16 / FO;
        const int pos = lane & 15, j = lane >> 4;
        const int sp = tile_stream<F>(pos), f = tile_mel<F>(pos);
        const int s = g * C::SPT + sp;
        const int gn = s / SPTN, spn = s % SPTN;
        const int posn = tile_pos<FO>(spn, f / 2);                     // (the consumer's own position order)
        constexpr int SH = kInterleave ? C::SPT : 1;                   // lanes between a stream's neighbouring mel positions
#pragma unroll
        for (int ro = 0; ro < R / C::PT; ++ro) {
            float pm[NCT][4];
#pragma unroll
            for (int ct = 0; ct < NCT; ++ct)
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    if (C::HOUT && ct == NCT - 1 && e >= 2) { pm[ct][e] = 0.f; continue; }      // padding registers of the half tile
                    float m = Y[ro * C::PT][ct][e];
                    if (C::PT == 2) m = fmax_nc(m, Y[ro * C::PT + 1][ct][e]);
                    pm[ct][e] = fmax_nc(m, dpp_shl_zero<SH>(m)) * p.xmul;      // (the next stage's input scale: calibrate_hx's ladder)
                }
            if ((f & 1) == 0) {                                        // one predicated region per pooled row
                float* xo = p.xout + ((size_t)(gn * RO + pass * (R / C::PT) + ro) * (NCT * 4)) * 64 + j * 16 + posn;
#pragma unroll
                for (int ct = 0; ct < NCT; ++ct)
#pragma unroll
                    for (int e = 0; e < 4; ++e)
                        if (!(C::HOUT && ct == NCT - 1 && e >= 2)) xo[(ct * 4 + e) * 64] = pm[ct][e];
            }
        }
    }
    if (LAST) {
        static_assert(!LAST || (C::RO == 1 && C::FO == 1 && NCT == 6), "last stage pools to one position, 96 channels");
        const int pos = lane & 15, j = lane >> 4;
        f32x4 Pl[NCT];
        // the pooled value of stream sp sits at the lane of its mel position 0; lanes 8..15 mirror lanes 0..7
        const int src = (j * 16 + tile_pos<F>(pos & 7, 0)) * 4;
        constexpr int SH = kInterleave ? C::SPT : 1;
#pragma unroll
        for (int ct = 0; ct < NCT; ++ct)
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                float m = fmax_nc(Y[0][ct][e], Y[1][ct][e]);
                m = fmax_nc(m, dpp_shl_zero<SH>(m)) * p.xmul;
                Pl[ct][e] = __builtin_bit_cast(float, __builtin_amdgcn_ds_bpermute(src, __builtin_bit_cast(int, m)));
            }
        float* h19 = p.hist19 + (size_t)g * (2 * NCT * 4 * 64);
        f32x4 T0[NCT], T1[NCT];
        Op H0[KS], H1[KS];
        load_tile<NCT>(T0, h19, lane);
        load_tile<NCT>(T1, h19 + NCT * 4 * 64, lane);
        to_ops<NCT>(T0, H0);
        to_ops<NCT>(T1, H1);
        Op Po[1][KS];
        to_ops<NCT>(Pl, Po[0]);
        f32x4 E[1][NCT];
        // (no guard here: an out-of-range input of conv19 yields a NaN embedding, which the heads kernel's guard reports)
        conv_time_hx<KS, NCT, 1, false, 4 * NCT, 0, WG, false, false, OWH_PIPE != 0, false, NS>(H0, H1, Po, E, ring, p.w19, nullptr, nullptr, 0.f, wave, lane, bad);      // (conv19's packed weights carry 1 / K of its input: true embeddings)
        const bool on19 = active && (p.stream_on == nullptr || p.stream_on[min(s_first + (pos & 7), p.S - 1)] != 0);   // lanes 8..15 mirror 0..7
        if (on19) {
            store_tile<NCT>(T1, h19, lane);
            store_tile<NCT>(Pl, h19 + NCT * 4 * 64, lane);
        }
        const int s = s_first + pos;
        if (on19 && pos < C::SPT && s < p.S) {
            const uint32_t slot = p.nfeat[s] % (uint32_t)p.TR;
#pragma unroll
            for (int ct = 0; ct < NCT; ++ct) {
                const f32x4 ev = E[0][ct] * p.emb_mul;                    // true units (conv19 runs at its input's scale x 4)
                *reinterpret_cast<f32x4*>(p.feat + ((size_t)s * p.TR + slot) * 96 + ct * 16 + 4 * j) = ev;
                *reinterpret_cast<f32x4*>(p.emb + (size_t)s * 96 + ct * 16 + 4 * j) = ev;
                if (DBG && p.dbg) *reinterpret_cast<f32x4*>(p.dbg + (size_t)s * p.dbg_stride + p.dbg_off[4] + ct * 16 + 4 * j) = ev;
            }
        }
    }
    }   // pass
    if (!LAST && C::NPASS > 1) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");    // drain the chunk the last pass prefetched
    raise_range_flag(bad, p.range_flag, s_first, C::SPT);
}

// ------------------------------------------------------------------------------------------------
// stage A (parameters: owr::RAParams; w0 = conv0 in the K-folded form of pack_hx_conv0, w1 / w2 = hx-packed)
//
// Round-3 form.  The 32 mel bins of a row are split over the row's two position tiles by PARITY: position p of tile h is mel bin
// 2p + h (rounds 1-2: bin 16h + p).  The +-1 mel neighbours of a position then sit in the OTHER tile of the row -- in the same
// lane, or one lane over with the zero padding of the mel axis falling exactly on the lane a row shift leaves without a source:
//   conv1 (1x3):  out_T0[p] = W0 X_T1[p-1] + W1 X_T0[p] + W2 X_T1[p]       out_T1[p] = W0 X_T0[p] + W1 X_T1[p] + W2 X_T0[p+1]
//     -> per tile ONE main chain (start value K h, two taps, six MFMAs) + one side chain (three MFMAs) that enters through a single
//        v_add_f32 dpp: 1 VALU per value for the tap combine (rounds 1-2: 5.3 -- two shifts with carries across the seam + adds);
//   pool 2x2:     the four values of a pooling window sit in the SAME lane of the row pair's four tiles: three max, no shift, and the
//                 16 pooled bins of a row fill one whole tile of stage B's input (unpredicated, fully coalesced stores);
//   the activation of conv2 runs after the pooling (the fold makes acc = K y with K > 0: max-pooling commutes with it): 1/4 of the values.
// conv0 (3x3, ONE input channel, K = 9): the three products of the f16 split share a single k-step,
//        k-slots 0..8 = xh_tap wh_tap, 9..17 = xh_tap wl_tap, 18..26 = xl_tap wh_tap   (27 of 32 slots; pack_hx_conv0)
//   so conv0 is one MFMA per output tile instead of three, and its B operand is gathered as f16 halves straight from two LDS planes
//   (hi / lo of the mel rows, split ONCE when a row is written) -- no VALU split per tap position (rounds 1-2: 12 VALU per tile).
// The conv1 outputs of the last two rows (conv2's history) and the last two mel rows live in HBM in operand form -- (hi, lo) f16
// pairs, the same 4 bytes per value as fp32 -- so nothing is split again when a step starts.
// ------------------------------------------------------------------------------------------------
// levels of the four row pairs of the fused front end's matrix phase (see owwhip_fused.h: OWF_PRIO 6; 0 = leave the priority alone)
#ifndef OWF_QPRIO
#if !defined(OWF_PRIO) || OWF_PRIO == 6
#define OWF_QPRIO 2233
#else
#define OWF_QPRIO 0
#endif
#endif
namespace sa {
constexpr int RS = 34;                 // halves per mel row in a plane: 32 bins + a zero column either side
constexpr int PLANE = 344;             // 10 rows (2 history + 8 new) x 34, rounded up to a multiple of 8 halves
constexpr int WAVE_HALVES = 2 * PLANE; // hi plane, lo plane
}
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

// gather offsets of conv0's B operand (bytes into a wave's planes, tile row 0 / parity 0): lane (p, g), half q <-> k-slot 8g + q
__device__ __forceinline__ void stageA_fill_gather_table(int* gtab /*[64][8]*/, int tid, int nt) {
    for (int i = tid; i < 512; i += nt) {
        const int lane = i >> 3, q = i & 7, p = lane & 15, g = lane >> 4, slot = 8 * g + q;
        const int tap = slot % 9, plane = slot >= 18 ? 1 : 0;
        gtab[i] = slot < 27 ? 2 * (plane * sa::PLANE + (tap / 3) * sa::RS + (tap % 3) + 2 * p) : 0;
    }
}
// one mel value -> its (hi, lo) halves in the planes
__device__ __forceinline__ void stageA_put_mel(_Float16* sP, int row, int bin, float v) {
    const _Float16 hi = (_Float16)v;
    sP[row * sa::RS + 1 + bin] = hi;
    sP[sa::PLANE + row * sa::RS + 1 + bin] = (_Float16)(v - (float)hi);
}
__device__ __forceinline__ f32x4 mfma3(const f16x8 ah, const f16x8 al, const Op& x, f32x4 c) {
    c = OWH_MFMA(ah, x.h, c);
    c = OWH_MFMA(ah, x.l, c);
    return OWH_MFMA(al, x.h, c);
}
// debug dump of a stage-A tile (parity order): position p of tile h <-> mel bin 2p + h
__device__ __forceinline__ void dump_tile_a(const f32x4 (&t)[2], float* __restrict__ dbg, size_t stride, int off, int s, int row, int h,
                                            int S, int lane, float mul) {
    if (s >= S) return;
    const int pos = lane & 15, j = lane >> 4;
#pragma unroll
    for (int ct = 0; ct < 2; ++ct)
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int c = ct == 0 ? 4 * j + e : (e < 2 ? 16 + 2 * j + e : 24);
            if (c < 24) dbg[(size_t)s * stride + off + (row * 32 + 2 * pos + h) * 24 + c] = t[ct][e] * mul;
        }
}

// one stream-step of stage A for the wave.  sP: the wave's planes; rows 0, 1 = history (filled here), rows 2..9 = the step's eight new
// rows (MEL_IN_LDS: already written by the caller -- the fused mel front end, owf::hmelA_kernel -- else read from p.mel).
template <bool DBG, bool MEL_IN_LDS>
__device__ __forceinline__ void hstageA_stream(const owr::RAParams& p, int s, _Float16* sP, const float* sW0, const float* sW1,
                                               const float* sW2, const float* sbn0, const int* gtab, lanemask_t& bad, int lane) {
    using namespace owr;
    const int j = lane >> 4;
    int z = 0;
    asm volatile("" : "+s"(z));
    const float* w0s = sW0 + z;
    const float* w1s = sW1 + z;
    const float* w2s = sW2 + z;
    const float* bn = sbn0 + z;
    // (wave-uniform bases in SGPRs, lane offsets in one VGPR: as 64-bit per-lane pointers their loop-invariant parts were hoisted out of
    //  the stream loop and spilled -- the only scratch traffic of the default step)
    unsigned* hm = const_cast<unsigned*>(reinterpret_cast<const unsigned*>(uniform_ptr(p.hist_mel + (size_t)s * 64)));           // (hi | lo << 16) of the last two mel rows
    unsigned* h2 = const_cast<unsigned*>(reinterpret_cast<const unsigned*>(uniform_ptr(p.hist2 + (size_t)s * (2 * 2 * 8 * 64))));   // [row 2][parity 2][dword 8: 3 hi, 3 lo, 2 unused][64]
    {
        const unsigned w = hm[lane];
        const int o = (lane >> 5) * sa::RS + 1 + (lane & 31);
        reinterpret_cast<unsigned short*>(sP)[o] = (unsigned short)(w & 0xffffu);
        reinterpret_cast<unsigned short*>(sP)[sa::PLANE + o] = (unsigned short)(w >> 16);
    }
    if (!MEL_IN_LDS) {
        const f32x4 m4 = *reinterpret_cast<const f32x4*>(p.mel + (size_t)s * p.mel_stride + p.mel_off + lane * 4);
        const int row = lane >> 3, col = (lane & 7) * 4;
#pragma unroll
        for (int e = 0; e < 4; ++e) stageA_put_mel(sP, 2 + row, col + e, m4[e]);
    }
    int go[8];
    {
        const int lz = (lane + z) * 8;             // (iteration-local: not hoisted and kept alive across the mel phase)
        const u32x4 g0 = *reinterpret_cast<const u32x4*>(gtab + lz), g1 = *reinterpret_cast<const u32x4*>(gtab + lz + 4);
#pragma unroll
        for (int q = 0; q < 4; ++q) { go[q] = (int)g0[q]; go[4 + q] = (int)g1[q]; }
    }
    Op Yh[2][2];                                  // conv1 output rows r-2, r-1 in operand form: [row][parity]
#pragma unroll
    for (int r = 0; r < 2; ++r)
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const unsigned* src = h2 + (r * 2 + h) * 512 + lane;
            Yh[r][h].h = __builtin_bit_cast(f16x8, u32x4{src[0], src[64], src[128], 0u});
            Yh[r][h].l = __builtin_bit_cast(f16x8, u32x4{src[192], src[256], src[320], 0u});
        }
    const char* planes = reinterpret_cast<const char*>(sP);
#pragma unroll
    for (int q = 0; q < 4; ++q) {                 // conv rows 2q, 2q+1; tile t = 2 * (row & 1) + parity
        OWR_SB();
        if (MEL_IN_LDS && OWF_QPRIO != 0) {       // the fused front end's wave priority rises with the stream-step's progress (owwhip_fused.h: OWF_PRIO)
            constexpr int code = OWF_QPRIO;
            if (q == 0) __builtin_amdgcn_s_setprio((code / 1000) % 10);
            else if (q == 1) __builtin_amdgcn_s_setprio((code / 100) % 10);
            else if (q == 2) __builtin_amdgcn_s_setprio((code / 10) % 10);
            else __builtin_amdgcn_s_setprio(code % 10);
        }
        // ---- conv0: one K-folded MFMA per output tile
        Op Y0o[4][1];
        {
            const f16x8 a0 = lds_h(w0s, 0, lane), a1 = lds_h(w0s, 1, lane);
            const f32x4 I0 = acc_init(bn + 32, 0, j), I1 = acc_init(bn + 32, 1, j), B0 = acc_init(bn, 0, j), B1 = acc_init(bn, 1, j);
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                const int r = 2 * q + (t >> 1), h = t & 1;
                const char* base = planes + (r * sa::RS + h) * 2;
                f16x8 b;                          // (element-wise f16 loads: ds_read_u16_d16 / _d16_hi fill the operand dwords in place)
#pragma unroll
                for (int v = 0; v < 8; ++v) b[v] = *reinterpret_cast<const _Float16*>(base + go[v]);
                // conv0 has a ReLU between the convolution and its BatchNorm (s relu(v) + h): with s folded into the weights and the
                // shift as start value, acc = K s v + K h and K (s relu(v) + h) = max(acc, K h) for s >= 0, min(acc, K h) for s < 0
                // = med3(acc, K h, +-inf): one VALU (bn[0..31] = the per-channel +-inf, bn[32..63] = K h)
                f32x4 Y0[2];
                Y0[0] = OWH_MFMA(a0, b, I0);
                Y0[1] = OWH_MFMA(a1, b, I1);
#pragma unroll
                for (int e = 0; e < 4; ++e) Y0[0][e] = __builtin_amdgcn_fmed3f(Y0[0][e], I0[e], B0[e]);
#pragma unroll
                for (int e = 0; e < 2; ++e) Y0[1][e] = __builtin_amdgcn_fmed3f(Y0[1][e], I1[e], B1[e]);       // (tile 1 = the half tile: 8 channels)
                Y0[0] = act_t<true, false>(Y0[0], p.clampv[0]);
                Y0[1] = act_t<true, true>(Y0[1], p.clampv[0]);
                pin_t<false>(Y0[0]); pin_t<true>(Y0[1]);
                if (DBG && p.dbg) dump_tile_a(Y0, p.dbg, p.dbg_stride, p.dbg_off[0], s, r, h, p.S, lane, p.dbg_mul[0]);
                to_ops<2, true>(Y0, Y0o[t]);
            }
        }
        // ---- conv1: 1x3 over the row's two parity tiles (see the header of this section)
        f32x4 Y1[4][2];
        {   // channel tile 0 (16 channels): per output tile one main chain (start value + two taps) and one side chain (one tap)
            const f32x4 I = acc_init(bn + 96, 0, j);
            const f32x4 Z = {0.f, 0.f, 0.f, 0.f};
            f32x4 sd[4], mn[4];
            {   // tap 0: W0 X_T1 -> side of T0 (enters shifted right), W0 X_T0 -> main of T1
                const f16x8 ah = lds_h(w1s, 0, lane), al = lds_h(w1s, 1, lane);
                sd[0] = OWH_MFMA(ah, Y0o[1][0].h, Z); mn[1] = OWH_MFMA(ah, Y0o[0][0].h, I);
                sd[2] = OWH_MFMA(ah, Y0o[3][0].h, Z); mn[3] = OWH_MFMA(ah, Y0o[2][0].h, I);
                sd[0] = OWH_MFMA(ah, Y0o[1][0].l, sd[0]); mn[1] = OWH_MFMA(ah, Y0o[0][0].l, mn[1]);
                sd[2] = OWH_MFMA(ah, Y0o[3][0].l, sd[2]); mn[3] = OWH_MFMA(ah, Y0o[2][0].l, mn[3]);
                sd[0] = OWH_MFMA(al, Y0o[1][0].h, sd[0]); mn[1] = OWH_MFMA(al, Y0o[0][0].h, mn[1]);
                sd[2] = OWH_MFMA(al, Y0o[3][0].h, sd[2]); mn[3] = OWH_MFMA(al, Y0o[2][0].h, mn[3]);
            }
            {   // tap 2: W2 X_T0 -> side of T1 (enters shifted left), W2 X_T1 -> main of T0
                const f16x8 ah = lds_h(w1s, 4, lane), al = lds_h(w1s, 5, lane);
                sd[1] = OWH_MFMA(ah, Y0o[0][0].h, Z); mn[0] = OWH_MFMA(ah, Y0o[1][0].h, I);
                sd[3] = OWH_MFMA(ah, Y0o[2][0].h, Z); mn[2] = OWH_MFMA(ah, Y0o[3][0].h, I);
                sd[1] = OWH_MFMA(ah, Y0o[0][0].l, sd[1]); mn[0] = OWH_MFMA(ah, Y0o[1][0].l, mn[0]);
                sd[3] = OWH_MFMA(ah, Y0o[2][0].l, sd[3]); mn[2] = OWH_MFMA(ah, Y0o[3][0].l, mn[2]);
                sd[1] = OWH_MFMA(al, Y0o[0][0].h, sd[1]); mn[0] = OWH_MFMA(al, Y0o[1][0].h, mn[0]);
                sd[3] = OWH_MFMA(al, Y0o[2][0].h, sd[3]); mn[2] = OWH_MFMA(al, Y0o[3][0].h, mn[2]);
            }
            {   // tap 1: the centre tap of every tile
                const f16x8 ah = lds_h(w1s, 2, lane), al = lds_h(w1s, 3, lane);
#pragma unroll
                for (int t = 0; t < 4; ++t) mn[t] = OWH_MFMA(ah, Y0o[t][0].h, mn[t]);
#pragma unroll
                for (int t = 0; t < 4; ++t) mn[t] = OWH_MFMA(ah, Y0o[t][0].l, mn[t]);
#pragma unroll
                for (int t = 0; t < 4; ++t) mn[t] = OWH_MFMA(al, Y0o[t][0].h, mn[t]);
            }
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                f32x4 r;
#pragma unroll
                for (int e = 0; e < 4; ++e) r[e] = mn[t][e] + ((t & 1) ? dpp_shl1_zero(sd[t][e]) : dpp_shr1_zero(sd[t][e]));
                nan_guard(bad, r[0]);
                Y1[t][0] = act_t<true, false>(r, p.clampv[1]); pin_t<false>(Y1[t][0]);
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        {   // channel tile 1 (8 channels in registers 0, 1): the side chain rides in the tile's free registers 2, 3 -- the weight
            // blocks stack two taps in M (pack_hx_stage_a: (W2 | W0) for T0, (W0 | W2) for T1, (W1 | 0) for the centre tap), so a
            // tile costs six MFMAs instead of nine
            const f32x4 I = acc_init(bn + 96, 1, j);
            const f32x4 I2 = {I[0], I[1], 0.f, 0.f};
            f32x4 Q[4];
            {
                const f16x8 ah = lds_h(w1s, 6, lane), al = lds_h(w1s, 7, lane);          // (W2 | W0) against X_T1 -> T0
                Q[0] = OWH_MFMA(ah, Y0o[1][0].h, I2); Q[2] = OWH_MFMA(ah, Y0o[3][0].h, I2);
                Q[0] = OWH_MFMA(ah, Y0o[1][0].l, Q[0]); Q[2] = OWH_MFMA(ah, Y0o[3][0].l, Q[2]);
                Q[0] = OWH_MFMA(al, Y0o[1][0].h, Q[0]); Q[2] = OWH_MFMA(al, Y0o[3][0].h, Q[2]);
            }
            {
                const f16x8 ah = lds_h(w1s, 10, lane), al = lds_h(w1s, 11, lane);        // (W0 | W2) against X_T0 -> T1
                Q[1] = OWH_MFMA(ah, Y0o[0][0].h, I2); Q[3] = OWH_MFMA(ah, Y0o[2][0].h, I2);
                Q[1] = OWH_MFMA(ah, Y0o[0][0].l, Q[1]); Q[3] = OWH_MFMA(ah, Y0o[2][0].l, Q[3]);
                Q[1] = OWH_MFMA(al, Y0o[0][0].h, Q[1]); Q[3] = OWH_MFMA(al, Y0o[2][0].h, Q[3]);
            }
            {
                const f16x8 ah = lds_h(w1s, 8, lane), al = lds_h(w1s, 9, lane);          // (W1 | 0): the centre tap
#pragma unroll
                for (int t = 0; t < 4; ++t) Q[t] = OWH_MFMA(ah, Y0o[t][0].h, Q[t]);
#pragma unroll
                for (int t = 0; t < 4; ++t) Q[t] = OWH_MFMA(ah, Y0o[t][0].l, Q[t]);
#pragma unroll
                for (int t = 0; t < 4; ++t) Q[t] = OWH_MFMA(al, Y0o[t][0].h, Q[t]);
            }
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                f32x4 r = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int e = 0; e < 2; ++e) r[e] = Q[t][e] + ((t & 1) ? dpp_shl1_zero(Q[t][e + 2]) : dpp_shr1_zero(Q[t][e + 2]));
                Y1[t][1] = act_t<true, true>(r, p.clampv[1]); pin_t<true>(Y1[t][1]);
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        if (DBG && p.dbg) {
#pragma unroll
            for (int t = 0; t < 4; ++t) dump_tile_a(Y1[t], p.dbg, p.dbg_stride, p.dbg_off[1], s, 2 * q + (t >> 1), t & 1, p.S, lane, p.dbg_mul[1]);
        }
        Op Y1o[4][1];
#pragma unroll
        for (int t = 0; t < 4; ++t) to_ops<2, true>(Y1[t], Y1o[t]);
        // ---- conv2: 3x1 over [Yh0, Yh1, Y1 row 2q, Y1 row 2q+1], then pool 2x2 (same lane of the four tiles), then the activation
        f32x4 PA[2];
        {   // channel tile 0
            const f32x4 I = acc_init(bn + 160, 0, j);
            f32x4 acc[4];
#pragma unroll
            for (int t = 0; t < 4; ++t) acc[t] = I;
#pragma unroll
            for (int tap = 0; tap < 3; ++tap) {
                const f16x8 ah = lds_h(w2s, tap * 2 + 0, lane), al = lds_h(w2s, tap * 2 + 1, lane);
#pragma unroll
                for (int part = 0; part < 3; ++part)
#pragma unroll
                    for (int t = 0; t < 4; ++t) {
                        const int src = (t >> 1) + tap, h = t & 1;
                        const Op& b = src < 2 ? Yh[src][h] : Y1o[(src - 2) * 2 + h][0];
                        acc[t] = OWH_MFMA(part == 2 ? al : ah, part == 1 ? b.l : b.h, acc[t]);
                    }
            }
#pragma unroll
            for (int t = 0; t < 4; ++t) nan_guard(bad, acc[t][0]);
            if (DBG && p.dbg) {
                const int pos = lane & 15;
#pragma unroll
                for (int t = 0; t < 4; ++t) {
                    const f32x4 y = act_t<true, false>(acc[t], p.clampv[2]);
#pragma unroll
                    for (int e = 0; e < 4; ++e)
                        if (s < p.S) p.dbg[(size_t)s * p.dbg_stride + p.dbg_off[2] + ((2 * q + (t >> 1)) * 32 + 2 * pos + (t & 1)) * 24 + 4 * j + e] = y[e] * p.dbg_mul[2];
                }
            }
            f32x4 m;
#pragma unroll
            for (int e = 0; e < 4; ++e) m[e] = fmax_nc(fmax_nc(acc[0][e], acc[1][e]), fmax_nc(acc[2][e], acc[3][e]));
            m = act_t<true, false>(m, p.clampv[2]);
            PA[0] = m * p.xmul;                   // stage B's input scale (calibrate_hx's ladder)
            pin_t<false>(PA[0]);
            __builtin_amdgcn_sched_barrier(0);
        }
        {   // channel tile 1 (8 channels): the two output rows of the pair share one accumulator tile per parity -- registers 0, 1 =
            // row 2q, registers 2, 3 = row 2q+1 -- and the weight blocks stack the two rows' taps in M (pack_hx_stage_a: input row i of
            // [Yh0, Yh1, Y1 row 2q, Y1 row 2q+1] meets (W_i | W_{i-1})): 12 MFMAs per parity instead of 18
            const f32x4 I = acc_init(bn + 160, 1, j);
            f32x4 Q[2] = {f32x4{I[0], I[1], I[0], I[1]}, f32x4{I[0], I[1], I[0], I[1]}};
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const f16x8 ah = lds_h(w2s, 6 + i * 2 + 0, lane), al = lds_h(w2s, 6 + i * 2 + 1, lane);
#pragma unroll
                for (int part = 0; part < 3; ++part)
#pragma unroll
                    for (int h = 0; h < 2; ++h) {
                        const Op& b = i < 2 ? Yh[i][h] : Y1o[(i - 2) * 2 + h][0];
                        Q[h] = OWH_MFMA(part == 2 ? al : ah, part == 1 ? b.l : b.h, Q[h]);
                    }
            }
            if (DBG && p.dbg) {
                const int pos = lane & 15;
#pragma unroll
                for (int h = 0; h < 2; ++h) {
                    const f32x4 y = act_t<true, false>(Q[h], p.clampv[2]);
#pragma unroll
                    for (int e = 0; e < 4; ++e)
                        if (s < p.S) p.dbg[(size_t)s * p.dbg_stride + p.dbg_off[2] + ((2 * q + (e >> 1)) * 32 + 2 * pos + h) * 24 + 16 + 2 * j + (e & 1)] = y[e] * p.dbg_mul[2];
                }
            }
            f32x4 m = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int e = 0; e < 2; ++e) m[e] = fmax_nc(fmax_nc(Q[0][e], Q[0][e + 2]), fmax_nc(Q[1][e], Q[1][e + 2]));
            m = act_t<true, true>(m, p.clampv[2]);
            PA[1] = m * p.xmul;
            pin_t<true>(PA[1]);
            __builtin_amdgcn_sched_barrier(0);
        }
#pragma unroll
        for (int h = 0; h < 2; ++h) { Yh[0][h] = Y1o[h][0]; Yh[1][h] = Y1o[2 + h][0]; }
        // ---- stage B input row q: the 16 pooled bins are the 16 positions of one tile
        float* xo = p.xout + ((size_t)price_alias<(OWH_PRICE_HANDOVER & 1) != 0>(s) * 4 + q) * (8 * 64) + lane;
#pragma unroll
        for (int e = 0; e < 4; ++e) xo[e * 64] = PA[0][e];
        xo[4 * 64] = PA[1][0];
        xo[5 * 64] = PA[1][1];
    }
#pragma unroll
    for (int r = 0; r < 2; ++r)
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            unsigned* dst = h2 + (r * 2 + h) * 512 + lane;
            const u32x4 hh = __builtin_bit_cast(u32x4, Yh[r][h].h), ll = __builtin_bit_cast(u32x4, Yh[r][h].l);
            dst[0] = hh[0]; dst[64] = hh[1]; dst[128] = hh[2];
            dst[192] = ll[0]; dst[256] = ll[1]; dst[320] = ll[2];
        }
    {
        const int o = (8 + (lane >> 5)) * sa::RS + 1 + (lane & 31);
        const unsigned short* u = reinterpret_cast<const unsigned short*>(sP);
        hm[lane] = (unsigned)u[o] | ((unsigned)u[sa::PLANE + o] << 16);
    }
}

template <bool DBG>
__global__ __launch_bounds__(256, DBG ? 1 : OWH_WPS_A) void hstageA_kernel(owr::RAParams p) {
    using namespace owr;
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int gw = blockIdx.x * 4 + wave, nw = gridDim.x * 4;

    // weights: conv0 [2 oct] K-folded blocks (2 KB), conv1 / conv2 [2 oct][3 taps][part 2] (12 KB each); start values; planes; gather table
    __shared__ __attribute__((aligned(16))) float sW0[2 * 256];
    __shared__ __attribute__((aligned(16))) float sW[2][14 * 256];     // conv1: 12 blocks, conv2: 14 (pack_hx_stage_a)
    __shared__ __attribute__((aligned(16))) float sbn[3][2][32];
    __shared__ __attribute__((aligned(16))) _Float16 sPl[4][sa::WAVE_HALVES];
    __shared__ __attribute__((aligned(16))) int gtab[512];
    for (int i = threadIdx.x; i < 2 * 64; i += 256) reinterpret_cast<f32x4*>(sW0)[i] = reinterpret_cast<const f32x4*>(p.w0)[i];
    for (int i = threadIdx.x; i < 12 * 64; i += 256) reinterpret_cast<f32x4*>(sW[0])[i] = reinterpret_cast<const f32x4*>(p.w1)[i];
    for (int i = threadIdx.x; i < 14 * 64; i += 256) reinterpret_cast<f32x4*>(sW[1])[i] = reinterpret_cast<const f32x4*>(p.w2)[i];
    if (threadIdx.x < 96) {
        const int l = threadIdx.x / 32, c = threadIdx.x % 32;
        sbn[l][0][c] = p.scale[l][c];
        sbn[l][1][c] = p.shift[l][c];
    }
    for (int i = threadIdx.x; i < 4 * sa::WAVE_HALVES; i += 256) sPl[0][i] = (_Float16)0.f;
    stageA_fill_gather_table(gtab, threadIdx.x, 256);
    __syncthreads();
    lanemask_t bad = 0;

    for (int s0 = gw; s0 < p.n_streams; s0 += nw) {
        const int s = s0 + p.s_base;
        if (p.stream_on && !p.stream_on[s]) continue;            // masked step: this stream sits it out
        hstageA_stream<DBG, false>(p, s, sPl[wave], sW0, sW[0], sW[1], &sbn[0][0][0], gtab, bad, lane);
        if (bad) { raise_range_flag(bad, p.range_flag, s, 1); bad = 0; }
    }
}

// ------------------------------------------------------------------------------------------------
// wake-word heads, fp16-split GEMM form (model.py:299-302; architecture train.py:56-83)
//   layer 1: D[64*NN hidden][stream] = W1[hidden][K = T*96] * F[K][stream]  -- K streamed 32 channels (one k-step) at a time:
//            the weights of a k-step (NN*4 hidden tiles x hi/lo x 1 KB) go global -> LDS once per 4-wave workgroup (double
//            buffered, global_load_lds), the feature operands of a wave's own 32 streams come straight from the feature ring
//   LayerNorm + ReLU in registers (cross-lane part of the sums by two xor-shuffles), layer 2 (64x64 per net) chained on the
//   D registers like the CNN layers, LayerNorm + ReLU, layer 3 dot product + sigmoid, hey_jarvis-style gating.
// A workgroup = 4 waves x 2 stream tiles = 128 streams; NN <= 4 nets of hidden 64 per launch.
// ------------------------------------------------------------------------------------------------
struct HeadHxNet {
    const float *w2hx;                       // [4 oct][2 ks][2 part][64][8 halves]
    const float *b1, *ln1g, *ln1b, *b2, *ln2g, *ln2b, *w3, *b3;
    int has_ln, role, head, out_col;
    float u1, u2;                            // exact power-of-two un-scales of the two accumulators: 2^-(e_feat + e_w1), 2^-e_w2
    int hidden;                              // real hidden units (<= 64; wide form <= 128): the units beyond are zero padding, outside the LayerNorm statistics
    float inv_hidden;                        // 1 / hidden
    // wide form (HT = 8: up to 128 hidden units, up to 8 outputs -- the multiclass `timer` model, docs/models/timers.md:9-27):
    const float* w3hx;                       // output layer [4 ks][2 part][64][8 halves]: 16 output rows, rows n_out .. 15 zero (b3: 16 floats)
    float u3;                                // 2^-e_w3
    int n_out, final_act;                    // final_act 0: sigmoid per output; 1: ReLU + softmax over the n_out outputs (train.py:79,152-165)
};
// Optional tail of the heads kernel: Model.predict's post-processing (model.py:330-381 -- first-5 zeroing, patience / debounce
// over the 30-deep score ring, ring append, VAD gate) and the step's frame-counter advance for the same streams, instead of two
// more launches (owk::postproc_kernel, owk::advance_kernel).  Valid when every label of the handle is produced by THIS launch
// (one group of sigmoid heads) and the step carries one chunk; otherwise the separate kernels run.
struct HeadHxPost {
    int enabled;
    float* scores;           // [S][NL]
    float* ring;             // [S][NL][30]
    uint32_t* npred;         // [S]
    uint32_t* nfeat;         // [S] (advanced here)
    const int* patience;     // [NL]
    const float* threshold;  // [NL] (NaN = none)
    int debounce_frames;
    const float* vad_ring;   // [S][8]
    const uint32_t* n_vad;   // [S]
    float vad_threshold;     // <= 0: gate off
};
struct HeadHxParams {
    const float* feat;      // ring [S][TR][96] or external [B][T][96] when ext != 0
    int ext, TR, T;
    const uint32_t* nfeat;
    const float* w1hx;      // [T*3 ksteps][NN*4 ct][2 part][64][8 halves]
    HeadHxNet net[4];
    float* raw;             // [S][NL]
    int NL, S, accumulate_max;
    int* range_flag;        // sticky f16-range flag of the handle (see nan_guard)
    HeadHxPost post;
    const uint8_t* stream_on;   // oww_step_masked: [S] 1 = the stream takes part in this step; nullptr = all do
    const int* ids;             // oww_step_masked with few participants: the n_ids participating streams (position k of the launch = stream
    int n_ids;                  // ids[k]); nullptr = streams s_base .. S-1
    int s_base;                 // block-pipelined step: first stream of this launch (S = one past its last)
    // Like every CNN layer, the first GEMM runs on a calibrated power-of-two scale: the features are multiplied by fscale = 2^e_feat
    // as they are loaded (oww_commit puts the probe set's largest |embedding| at 2^9..2^10 -- a factor 64 below the f16 overflow, and
    // embeddings of order 1e-4 keep the low halves of their split), each net's weights are stored as halves of 2^e_w w with the
    // largest at 2^11..2^12 (no fixed 2^8: weights of any magnitude fit), and HeadHxNet::u1 / u2 undo both on the fp32 accumulator.
    float fscale;
};

__device__ __forceinline__ float xsum4(float v) {      // sum over the four j groups (lanes p, p+16, p+32, p+48)
    v += __shfl_xor(v, 16);
    v += __shfl_xor(v, 32);
    return v;
}

template <int NN>
__device__ __forceinline__ void ln_relu(f32x4 (&h)[4], const float* __restrict__ bias, const float* __restrict__ g,
                                        const float* __restrict__ b, int has_ln, int j, float unscale, int hidden, float inv_hidden) {
    // h: the 64 hidden values of one net for this lane's stream: tile ct, register e <-> hidden 16ct + 4j + e.  Units >= `hidden` are
    // padding: zero weights and bias make them exactly 0 here, zero gamma / beta keep them 0 behind the LayerNorm; only the variance has
    // to leave them out (their (0 - mu)^2 is not part of the net)
#pragma unroll
    for (int ct = 0; ct < 4; ++ct) {
        const f32x4 bb = *reinterpret_cast<const f32x4*>(bias + ct * 16 + 4 * j);
        h[ct] = h[ct] * unscale + bb;
    }
    if (has_ln) {
        float sum = 0.f;
#pragma unroll
        for (int ct = 0; ct < 4; ++ct)
#pragma unroll
            for (int e = 0; e < 4; ++e) sum += h[ct][e];
        const float mu = xsum4(sum) * inv_hidden;
        float var = 0.f;
#pragma unroll
        for (int ct = 0; ct < 4; ++ct)
#pragma unroll
            for (int e = 0; e < 4; ++e) { const float d = (ct * 16 + 4 * j + e < hidden) ? h[ct][e] - mu : 0.f; var = fmaf(d, d, var); }
        const float rs = 1.0f / sqrtf(xsum4(var) * inv_hidden + 1e-5f);
#pragma unroll
        for (int ct = 0; ct < 4; ++ct) {
            const f32x4 gg = *reinterpret_cast<const f32x4*>(g + ct * 16 + 4 * j);
            const f32x4 be = *reinterpret_cast<const f32x4*>(b + ct * 16 + 4 * j);
#pragma unroll
            for (int e = 0; e < 4; ++e) h[ct][e] = fmaxf((h[ct][e] - mu) * rs * gg[e] + be[e], 0.f);
        }
    } else {
#pragma unroll
        for (int ct = 0; ct < 4; ++ct)
#pragma unroll
            for (int e = 0; e < 4; ++e) h[ct][e] = owr::fmax_nc(h[ct][e], 0.f);
    }
}

// ---- wide form of the heads kernel (HT = 8 hidden tiles) --------------------------------------------------------------------------
// Nets of up to 128 hidden units and up to 8 outputs, with or without LayerNorm, sigmoid or ReLU + softmax at the end: train.py's class
// default width (train.py:67, layer_dim = 128) and the released multiclass `timer` model (T = 34: 3264 -> 128 -> 128 -> 7, ReLU between
// the layers, softmax added at export: docs/models/timers.md:9-27, train.py:152-165).  Same first GEMM as the 64-unit form -- a net is
// eight hidden tiles instead of four -- then 128 x 128 and 128 x 16 (outputs zero-padded) on the matrix pipe as well.  Without a
// LayerNorm nothing bounds a hidden vector, so each stream's vector is carried into the next GEMM multiplied by ITS OWN power of two
// (largest unit at 2^9..2^10, undone exactly on the fp32 accumulator by v_ldexp): no hidden magnitude leaves the f16 range or loses the
// low halves of its split, and a stream's bits do not depend on its neighbours.
template <int HT>
__device__ __forceinline__ void ln_relu_w(f32x4 (&h)[HT], const float* __restrict__ bias, const float* __restrict__ g, const float* __restrict__ b,
                                          int has_ln, int j, float unscale, int unexp, int hidden, float inv_hidden) {
#pragma unroll
    for (int ct = 0; ct < HT; ++ct) {
        const f32x4 bb = *reinterpret_cast<const f32x4*>(bias + ct * 16 + 4 * j);
#pragma unroll
        for (int e = 0; e < 4; ++e) h[ct][e] = ldexpf(h[ct][e] * unscale, unexp) + bb[e];
    }
    if (has_ln) {
        float sum = 0.f;
#pragma unroll
        for (int ct = 0; ct < HT; ++ct)
#pragma unroll
            for (int e = 0; e < 4; ++e) sum += h[ct][e];
        const float mu = xsum4(sum) * inv_hidden;
        float var = 0.f;
#pragma unroll
        for (int ct = 0; ct < HT; ++ct)
#pragma unroll
            for (int e = 0; e < 4; ++e) { const float d = (ct * 16 + 4 * j + e < hidden) ? h[ct][e] - mu : 0.f; var = fmaf(d, d, var); }
        const float rs = 1.0f / sqrtf(xsum4(var) * inv_hidden + 1e-5f);
#pragma unroll
        for (int ct = 0; ct < HT; ++ct) {
            const f32x4 gg = *reinterpret_cast<const f32x4*>(g + ct * 16 + 4 * j);
            const f32x4 be = *reinterpret_cast<const f32x4*>(b + ct * 16 + 4 * j);
#pragma unroll
            for (int e = 0; e < 4; ++e) h[ct][e] = fmaxf((h[ct][e] - mu) * rs * gg[e] + be[e], 0.f);
        }
    } else {
#pragma unroll
        for (int ct = 0; ct < HT; ++ct)
#pragma unroll
            for (int e = 0; e < 4; ++e) h[ct][e] = owr::fmax_nc(h[ct][e], 0.f);
    }
}
// the stream's own scale: multiplies the (non-negative) hidden vector by 2^e with its largest unit at 2^9 .. 2^10; returns e
template <int HT>
__device__ __forceinline__ int scale_own(f32x4 (&h)[HT]) {
    float m = 0.f;
#pragma unroll
    for (int ct = 0; ct < HT; ++ct)
#pragma unroll
        for (int e = 0; e < 4; ++e) m = fmaxf(m, h[ct][e]);
    m = fmaxf(m, __shfl_xor(m, 16));
    m = fmaxf(m, __shfl_xor(m, 32));
    int ex = __builtin_amdgcn_frexp_expf(m);                     // m = f 2^ex, f in [0.5, 1); 0 for m = 0 (an all-zero vector stays zero)
    ex = 10 - min(max(ex, -110), 120);
#pragma unroll
    for (int ct = 0; ct < HT; ++ct)
#pragma unroll
        for (int e = 0; e < 4; ++e) h[ct][e] = ldexpf(h[ct][e], ex);
    return ex;
}

template <int NN, int WG, int HT, int NCT>
__device__ __forceinline__ void heads_wide_tail(const HeadHxParams& p, f32x4 (&acc)[NCT][2], lanemask_t& bad, int lane, int wave) {
    constexpr int KS = HT / 2;                  // k-steps of a hidden vector
    const int pos = lane & 15, j = lane >> 4;
#pragma unroll
    for (int n = 0; n < NN; ++n) {
        const HeadHxNet& net = p.net[n];
        Op ho[2][KS];
        int ex[2];
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            f32x4 h1[HT];
#pragma unroll
            for (int ct = 0; ct < HT; ++ct) h1[ct] = acc[HT * n + ct][t];
            ln_relu_w<HT>(h1, net.b1, net.ln1g, net.ln1b, net.has_ln, j, net.u1, 0, net.hidden, net.inv_hidden);
            ex[t] = scale_own<HT>(h1);
            to_ops<HT>(h1, ho[t]);
        }
        // ---- hidden block: 128 x 128, one output tile at a time, each weight block read once for both stream tiles
        f32x4 h2[2][HT];
#pragma unroll
        for (int oct = 0; oct < HT; ++oct) {
            f32x4 a2[2] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
#pragma unroll
            for (int k2 = 0; k2 < KS; ++k2) {
                const f16x8 wh = *reinterpret_cast<const f16x8*>(net.w2hx + (((oct * KS + k2) * 2 + 0) * 64 + lane) * 4);
                const f16x8 wl = *reinterpret_cast<const f16x8*>(net.w2hx + (((oct * KS + k2) * 2 + 1) * 64 + lane) * 4);
#pragma unroll
                for (int t = 0; t < 2; ++t) {
                    a2[t] = OWH_MFMA(wh, ho[t][k2].h, a2[t]);
                    a2[t] = OWH_MFMA(wh, ho[t][k2].l, a2[t]);
                    a2[t] = OWH_MFMA(wl, ho[t][k2].h, a2[t]);
                }
            }
            h2[0][oct] = a2[0]; h2[1][oct] = a2[1];
            if (oct == 0) { nan_guard(bad, a2[0][0]); nan_guard(bad, a2[1][0]); }
        }
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            ln_relu_w<HT>(h2[t], net.b2, net.ln2g, net.ln2b, net.has_ln, j, net.u2, -ex[t], net.hidden, net.inv_hidden);
            ex[t] = scale_own<HT>(h2[t]);
            to_ops<HT>(h2[t], ho[t]);
        }
        // ---- output layer: 16 rows (n_out real ones) x 128 -- lane (pos, j), register e <-> output 4j + e of stream pos
        f32x4 z[2] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
#pragma unroll
        for (int k2 = 0; k2 < KS; ++k2) {
            const f16x8 wh = *reinterpret_cast<const f16x8*>(net.w3hx + ((k2 * 2 + 0) * 64 + lane) * 4);
            const f16x8 wl = *reinterpret_cast<const f16x8*>(net.w3hx + ((k2 * 2 + 1) * 64 + lane) * 4);
#pragma unroll
            for (int t = 0; t < 2; ++t) {
                z[t] = OWH_MFMA(wh, ho[t][k2].h, z[t]);
                z[t] = OWH_MFMA(wh, ho[t][k2].l, z[t]);
                z[t] = OWH_MFMA(wl, ho[t][k2].h, z[t]);
            }
        }
        const f32x4 b3 = *reinterpret_cast<const f32x4*>(net.b3 + 4 * j);
        const int O = net.n_out;
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            nan_guard(bad, z[t][0]);
            float v[4];
            if (net.final_act == 1) {                                   // ReLU, then softmax over the n_out outputs (rows 0..7: lane groups 0, 1)
                float mx = -INFINITY;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    v[e] = fmaxf(ldexpf(z[t][e] * net.u3, -ex[t]) + b3[e], 0.f);
                    if (4 * j + e < O) mx = fmaxf(mx, v[e]);
                }
                mx = fmaxf(mx, __shfl_xor(mx, 16));
                float sum = 0.f;
#pragma unroll
                for (int e = 0; e < 4; ++e) { v[e] = 4 * j + e < O ? expf(v[e] - mx) : 0.f; sum += v[e]; }
                sum += __shfl_xor(sum, 16);
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] /= sum;
            } else {
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] = 1.0f / (1.0f + expf(-(ldexpf(z[t][e] * net.u3, -ex[t]) + b3[e])));
            }
            const int idx = (blockIdx.x * WG + wave) * 32 + t * 16 + pos;
            if (j >= 2 || idx >= (p.ids ? p.n_ids : p.S - p.s_base)) continue;
            const int st = p.ids ? p.ids[idx] : idx + p.s_base;
            if (p.stream_on && !p.stream_on[st]) continue;              // sits this step out: its scores stay as they are
#pragma unroll
            for (int e = 0; e < 4; ++e)
                if (4 * j + e < O) {
                    float* o = p.raw + (size_t)st * p.NL + net.out_col + 4 * j + e;
                    *o = p.accumulate_max ? fmaxf(*o, v[e]) : v[e];
                }
        }
    }
}

// Waves per workgroup and weight chunks in the LDS ring.  The first layer is a K = 96 T GEMM whose weights (NN x 8 KB per k-step
// of 32 features) stream L2 -> LDS once per workgroup.  Defaults: 4 waves (128 streams), double buffer.  A deeper pipeline --
// 8 waves = 256 streams per chunk and a ring of 3 or 4 chunks with a counted s_waitcnt vmcnt(n) in front of the barrier, so
// that two or three chunks stay in flight -- measured the same or 2 % slower (0.353 vs 0.347 ms): the waves' wait time (59 % of
// their cycles, PMC) is the LDS read of the A operands (32 KB per wave and k-step for 96 MFMAs), not the DMA latency.
#ifndef OWH_HEADS_WG
#define OWH_HEADS_WG 4
#endif
#ifndef OWH_HEADS_NBUF
#define OWH_HEADS_NBUF 2
#endif
#ifndef OWH_HEADS_APIPE
#define OWH_HEADS_APIPE 1      // A operands one pair of hidden tiles ahead of their MFMAs: heads launch 0.374 -> 0.340 ms (profiles/r06_apipe_ab.txt)
#endif
constexpr int HX_WG = OWH_HEADS_WG, HX_NBUF = OWH_HEADS_NBUF;
// NBUF = slots of the weight ring in LDS (and of the feature ring in registers); D = NBUF - 1 k-steps are in flight ahead of the one
// being consumed: the weight chunk of k-step i + D (L2 -> LDS, global_load_lds) and, issued right AFTER it, the feature rows of the
// same k-step (HBM / L2 -> VGPRs).  VMEM reads return in order, so the wait the compiler places in front of the f16 split of k-step
// i + 1's features also covers that k-step's weight chunk -- no counted s_waitcnt is needed, only the workgroup barrier that
// publishes the other waves' chunk parts.
// Every slot is its OWN LDS object (hslot<>): the compiler orders an LDS read behind every pending LDS-DMA write it cannot prove
// disjoint (s_waitcnt vmcnt(0)); with one ring array that made each wave wait for the chunk it had just issued before it read the
// current one -- the prefetch never overlapped the wave's own MFMAs.  The k loop is unrolled by NBUF so that slots are static.
// Large launches run several workgroups per CU, which cover each other's latencies: two slots (HX_NBUF) are enough.  A SMALL launch
// (BASELINE configs[1]: 4,096 streams = 32 workgroups on 256 CUs) is one workgroup alone on its CU walking 48 k-steps of ~0.35 us of
// MFMA work each behind a 1.5-2 us memory round trip: the deep instantiation keeps 5 (two nets or fewer) or 3 k-steps in flight.
// Same arithmetic in the same order, so a stream's scores do not depend on which instantiation ran (batch invariance is tested bit
// for bit: test_large_batch_properties).
template <int NN> struct HeadsDeep { static constexpr int NBUF = NN <= 2 ? 6 : 4; };
template <int U, int N, class F>
__device__ __forceinline__ void static_for_while(F&& f) {        // f(integral_constant<U>) for U = 0 .. N-1 while it returns true
    if constexpr (U < N) {
        if (f(std::integral_constant<int, U>{})) static_for_while<U + 1, N>(f);
    }
}
template <int FLOATS, int I>
__device__ __forceinline__ float* hslot() {
    __shared__ __attribute__((aligned(16))) float slot[FLOATS];
    return slot;
}
template <int FLOATS, int I>
__device__ __forceinline__ float* hslot_at(int i) {         // slot i of 0..I (i is a compile-time value after unrolling)
    if constexpr (I == 0) return hslot<FLOATS, 0>();
    else return i == I ? hslot<FLOATS, I>() : hslot_at<FLOATS, I - 1>(i);
}
// WG = waves per workgroup (32 streams each).  Round 5 A/B at 4,096 streams x hey_jarvis (same box, per-launch hipEvents): one wave per
// workgroup (128 workgroups instead of 32 on the 256 CUs) 79 us against 71 us for WG = 4 -- every workgroup then streams the weights
// itself and the launch is bound by exactly that stream; two-wave workgroups in stages D / E: 36.6 / 39.3 against 35.2 / 38.6 us.
template <int NN, int NBUF = HX_NBUF, int WG = HX_WG, int HT = 4>       // HT = hidden tiles per net: 4 (<= 64 units) or 8 (wide form, <= 128)
__global__ __launch_bounds__(64 * WG, (NBUF > 2 ? 1 : 2)) void heads_hx_kernel(HeadHxParams p) {
    using namespace owr;
    constexpr int NCT = NN * HT;                // hidden tiles of 16
    constexpr int NBLK = NCT * 2;               // 1 KB blocks per k-step chunk
    constexpr int CHUNK = NBLK * 256;           // floats
    constexpr int D = NBUF - 1;                 // chunks in flight ahead of the one being consumed
    const int lane = threadIdx.x & 63, pos = lane & 15, j = lane >> 4;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int KST = p.T * 3;
    issue_chunk<NBLK, WG>(p.w1hx, hslot<CHUNK, 0>(), wave, lane);      // chunk 0 flies while the stream addresses are set up

    // this lane's streams (two tiles of 16) and the address of ring row t
    int s[2];
    const float* frow[2];
    uint32_t slot0[2];
#pragma unroll
    for (int t = 0; t < 2; ++t) {
        const int idx = (blockIdx.x * WG + wave) * 32 + t * 16 + pos;
        s[t] = p.ids ? p.ids[min(idx, p.n_ids - 1)] : min(idx + p.s_base, p.S - 1);
        if (p.ext) { frow[t] = p.feat + (size_t)s[t] * p.T * 96; slot0[t] = 0; }
        else { frow[t] = p.feat + (size_t)s[t] * p.TR * 96; slot0[t] = p.nfeat[s[t]] + (uint32_t)(2 * p.TR - p.T + 1); }
    }
    // features of k-step ks as loaded (two 16-byte pieces per tile); the f16 split happens when the k-step is next in line
    auto load_raw = [&](int ks, f32x4 (&r)[2][2]) {
        const int tr = ks / 3, c0 = (ks % 3) * 32 + 8 * j;
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            const uint32_t slot = p.ext ? (uint32_t)tr : (slot0[t] + (uint32_t)tr) % (uint32_t)p.TR;
            const float* src = frow[t] + (size_t)slot * 96 + c0;
            r[t][0] = *reinterpret_cast<const f32x4*>(src);        // (scaled when they are split: nothing may wait for a load here)
            r[t][1] = *reinterpret_cast<const f32x4*>(src + 4);
        }
    };
    f32x4 acc[NCT][2];
#pragma unroll
    for (int ct = 0; ct < NCT; ++ct) { acc[ct][0] = f32x4{0.f, 0.f, 0.f, 0.f}; acc[ct][1] = f32x4{0.f, 0.f, 0.f, 0.f}; }
    Op bcur[2];
    f32x4 raw[NBUF][2][2];                      // feature rows of the k-steps in flight (slot = k-step mod NBUF, static after unrolling)
#pragma unroll
    for (int c = 0; c < D; ++c)
        if (c < KST) {
            if (c > 0) issue_chunk<NBLK, WG>(p.w1hx + (size_t)c * CHUNK, hslot_at<CHUNK, NBUF - 1>(c), wave, lane);     // (chunk 0: issued at kernel start)
            asm volatile("" ::: "memory");      // program order = issue order: chunk c, then the features of k-step c
            load_raw(c, raw[c]);
        }
    const float fsc = p.fscale;
    bcur[0] = split_pair<false>(raw[0][0][0] * fsc, raw[0][0][1] * fsc);
    bcur[1] = split_pair<false>(raw[0][1][0] * fsc, raw[0][1][1] * fsc);    // (waiting for these features = chunk 0 has landed: in-order return)
    pin_op(bcur[0]); pin_op(bcur[1]);
    __syncthreads();
    // one k-step: ring slot u (static), prefetch of k-step ks + D into slot (u + D) % NBUF, MFMAs, hand-over to k-step ks + 1
    auto kstep = [&](int ks, auto uc, auto always) {
        constexpr int u = decltype(uc)::value;
        constexpr bool ALWAYS = decltype(always)::value;        // main loop: every k-step of the group prefetches and has a successor
        const float* cur = hslot_at<CHUNK, NBUF - 1>(u);
        if (ALWAYS || ks + D < KST) {                           // slot of k-step ks-1: every wave passed the last barrier, its features are split
            constexpr int un = (u + D) % NBUF;
            issue_chunk<NBLK, WG>(p.w1hx + (size_t)(ks + D) * CHUNK, hslot_at<CHUNK, NBUF - 1>(un), wave, lane);
            asm volatile("" ::: "memory");
            load_raw(ks + D, raw[un]);
        }
#if OWH_HEADS_APIPE
        // A operands (four 1 KB weight blocks per pair of hidden tiles) one pair AHEAD of the MFMAs that consume them: left alone the
        // compiler issues a pair's four ds_read_b128 next to the last MFMA of the pair before and waits lgkmcnt(0) in front of the next
        // twelve -- ~100 cycles of LDS latency in the open per 192 MFMA cycles.  LDS reads return in order, so the wait in front of a
        // pair is lgkmcnt(4): its own reads done, the next pair's in flight.  Same MFMAs in the same order: bit-identical.
        f16x8 A[2][4];
#pragma unroll
        for (int b = 0; b < 4; ++b) A[0][b] = lds_h(cur, b, lane);
#pragma unroll
        for (int c2 = 0; c2 < NCT; c2 += 2) {
            const int pc = (c2 / 2) & 1;
            if (c2 + 2 < NCT) {
#pragma unroll
                for (int b = 0; b < 4; ++b) A[pc ^ 1][b] = lds_h(cur, (c2 + 2) * 2 + b, lane);
            }
            __builtin_amdgcn_sched_barrier(0);                  // (the reads stay in front of this pair's MFMAs)
#pragma unroll
            for (int part = 0; part < 3; ++part) {
#pragma unroll
                for (int t = 0; t < 2; ++t) {
                    acc[c2][t] = OWH_MFMA(part == 2 ? A[pc][1] : A[pc][0], part == 1 ? bcur[t].l : bcur[t].h, acc[c2][t]);
                    acc[c2 + 1][t] = OWH_MFMA(part == 2 ? A[pc][3] : A[pc][2], part == 1 ? bcur[t].l : bcur[t].h, acc[c2 + 1][t]);
                }
            }
        }
#else
#pragma unroll
        for (int c2 = 0; c2 < NCT; c2 += 2) {
            const f16x8 ah0 = lds_h(cur, c2 * 2 + 0, lane), al0 = lds_h(cur, c2 * 2 + 1, lane);
            const f16x8 ah1 = lds_h(cur, c2 * 2 + 2, lane), al1 = lds_h(cur, c2 * 2 + 3, lane);
#pragma unroll
            for (int part = 0; part < 3; ++part) {
#pragma unroll
                for (int t = 0; t < 2; ++t) {
                    acc[c2][t] = OWH_MFMA(part == 2 ? al0 : ah0, part == 1 ? bcur[t].l : bcur[t].h, acc[c2][t]);
                    acc[c2 + 1][t] = OWH_MFMA(part == 2 ? al1 : ah1, part == 1 ? bcur[t].l : bcur[t].h, acc[c2 + 1][t]);
                }
            }
        }
#endif
        if (ALWAYS || ks + 1 < KST) {
            // k-step ks+1: its features are younger than its weight chunk, so the split's wait covers this wave's part of the chunk;
            // the barrier covers the other waves' parts.  Everything issued for k-steps ks+2 .. ks+D stays in flight.
            constexpr int u1 = (u + 1) % NBUF;
            __builtin_amdgcn_sched_barrier(0);                  // the split (and the wait for its features) stays BEHIND this k-step's MFMAs
            bcur[0] = split_pair<false>(raw[u1][0][0] * fsc, raw[u1][0][1] * fsc);
            bcur[1] = split_pair<false>(raw[u1][1][0] * fsc, raw[u1][1][1] * fsc);
            pin_op(bcur[0]); pin_op(bcur[1]);                   // (the split -- and with it the wait -- stays in front of the barrier)
            // a bare s_barrier: __syncthreads() carries a workgroup-scope release fence, which on this target is s_waitcnt vmcnt(0) --
            // it would drain every prefetch in flight at every k-step.  What the barrier has to order is already ordered: this wave's
            // part of chunk ks+1 has landed (the wait above), its LDS reads of chunk ks have returned (they fed the MFMAs above), and
            // the memory clobber keeps the compiler from moving LDS accesses across it.
            asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");     // (lgkmcnt(0): this wave's reads of the slot have returned)
        }
    };
    auto group = [&](int ks0, auto always) {                    // NBUF consecutive k-steps, slots 0 .. NBUF-1
        constexpr bool ALWAYS = decltype(always)::value;
        static_for_while<0, NBUF>([&](auto uc) {
            constexpr int U = decltype(uc)::value;
            if (!ALWAYS && ks0 + U >= KST) return false;
            kstep(ks0 + U, uc, always);
            return true;
        });
    };
    int ks0 = 0;
    // main loop: straight-line groups (no conditional issue: the compiler's wait counts stay exact, nothing waits for the newest prefetch)
    for (; ks0 + NBUF - 1 + D < KST; ks0 += NBUF) group(ks0, std::true_type{});
    for (; ks0 < KST; ks0 += NBUF) group(ks0, std::false_type{});        // the last D .. NBUF + D - 1 k-steps
    lanemask_t bad = 0;
    nan_guard(bad, acc[0][0][0]);               // a feature beyond the f16 range
    nan_guard(bad, acc[0][1][0]);
    if constexpr (HT != 4) {                    // wide nets: their own tail (no gating, no fused post-processing)
        heads_wide_tail<NN, WG, HT, NCT>(p, acc, bad, lane, wave);
        if (p.ids) raise_range_flag(bad, p.range_flag);
        else raise_range_flag(bad, p.range_flag, (blockIdx.x * WG + wave) * 32 + p.s_base, 32);
        return;
    } else {
    // ---- per net: bias, LayerNorm, ReLU, 64x64, bias, LayerNorm, ReLU, dot, sigmoid
    float score[NN][2];
#pragma unroll
    for (int n = 0; n < NN; ++n) {
        const HeadHxNet& net = p.net[n];
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            f32x4 h1[4] = {acc[4 * n][t], acc[4 * n + 1][t], acc[4 * n + 2][t], acc[4 * n + 3][t]};
            ln_relu<NN>(h1, net.b1, net.ln1g, net.ln1b, net.has_ln, j, net.u1, net.hidden, net.inv_hidden);
            Op ho[2];
            to_ops<4>(h1, ho);
            f32x4 h2[4];
#pragma unroll
            for (int oct = 0; oct < 4; ++oct) {
                f32x4 a2 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int k2 = 0; k2 < 2; ++k2) {
                    const f16x8 wh = *reinterpret_cast<const f16x8*>(net.w2hx + (((oct * 2 + k2) * 2 + 0) * 64 + lane) * 4);
                    const f16x8 wl = *reinterpret_cast<const f16x8*>(net.w2hx + (((oct * 2 + k2) * 2 + 1) * 64 + lane) * 4);
                    a2 = OWH_MFMA(wh, ho[k2].h, a2);
                    a2 = OWH_MFMA(wh, ho[k2].l, a2);
                    a2 = OWH_MFMA(wl, ho[k2].h, a2);
                }
                h2[oct] = a2;
                if (oct == 0) nan_guard(bad, a2[0]);
            }
            ln_relu<NN>(h2, net.b2, net.ln2g, net.ln2b, net.has_ln, j, net.u2, net.hidden, net.inv_hidden);
            float z = 0.f;
#pragma unroll
            for (int ct = 0; ct < 4; ++ct) {
                const f32x4 w3 = *reinterpret_cast<const f32x4*>(net.w3 + ct * 16 + 4 * j);
#pragma unroll
                for (int e = 0; e < 4; ++e) z = fmaf(h2[ct][e], w3[e], z);
            }
            z = xsum4(z) + net.b3[0];
            score[n][t] = 1.0f / (1.0f + expf(-z));
        }
    }
    // ---- gating + store (one lane group per stream: j == 0)
    if (j == 0) {
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            const int idx = (blockIdx.x * WG + wave) * 32 + t * 16 + pos;
            if (idx >= (p.ids ? p.n_ids : p.S - p.s_base)) continue;
            const int st = p.ids ? p.ids[idx] : idx + p.s_base;
            if (p.stream_on && !p.stream_on[st]) continue;          // sits this step out: scores, rings and counters stay as they are
            const uint32_t cnt = p.post.enabled ? p.post.npred[st] : 0u;
            const int have = cnt < 30u ? (int)cnt : 30;
            float fin[NN];
#pragma unroll
            for (int n = 0; n < NN; ++n) {
                fin[n] = 0.f;
                if (p.net[n].role != 0) continue;
                float sc = score[n][t];
                if (n + 1 < NN && p.net[n + 1].role == 1 && p.net[n + 1].head == p.net[n].head && sc > 0.5f) sc = score[n + 1][t];
                const int l = p.net[n].out_col;
                float* o = p.raw + (size_t)st * p.NL + l;
                sc = p.accumulate_max ? fmaxf(*o, sc) : sc;
                *o = sc;
                if (p.post.enabled) {                                       // the rules of owk::postproc_kernel, same order
                    float* ring = p.post.ring + ((size_t)st * p.NL + l) * 30;
                    if (cnt < 5u) sc = 0.0f;                                // model.py:331-333
                    if (sc != 0.0f) {
                        const int pat = p.post.patience[l];
                        const float thr = p.post.threshold[l];
                        if (pat > 0) {                                      // model.py:349-352
                            const int look = pat < have ? pat : have;
                            int n_ok = 0;
                            for (int i = 1; i <= look; ++i) n_ok += ring[(cnt - i) % 30u] >= thr ? 1 : 0;
                            if (n_ok < pat) sc = 0.0f;
                        } else if (p.post.debounce_frames > 0 && thr == thr) {   // model.py:353-359
                            const int look = p.post.debounce_frames < have ? p.post.debounce_frames : have;
                            int n_hit = 0;
                            for (int i = 1; i <= look; ++i) n_hit += ring[(cnt - i) % 30u] >= thr ? 1 : 0;
                            if (sc >= thr && n_hit > 0) sc = 0.0f;
                        }
                    }
                    ring[cnt % 30u] = sc;                                   // model.py:362-363
                    fin[n] = sc;
                }
            }
            if (p.post.enabled) {
                p.post.npred[st] = cnt + 1u;
                p.post.nfeat[st] += 1u;                                     // (this stream's ring slot was read at kernel start)
                bool gate = false;
                if (p.post.vad_threshold > 0.0f) {                          // model.py:375-381
                    const uint32_t L = p.post.n_vad[st];
                    float vmax = 0.0f;
                    if (L >= 5u) {
                        vmax = -INFINITY;
                        for (uint32_t i = (L >= 7u ? L - 7u : 0u); i + 5u <= L; ++i) vmax = fmaxf(vmax, p.post.vad_ring[(size_t)st * 8 + (i & 7u)]);
                    }
                    gate = vmax < p.post.vad_threshold;
                }
#pragma unroll
                for (int n = 0; n < NN; ++n)
                    if (p.net[n].role == 0) p.post.scores[(size_t)st * p.NL + p.net[n].out_col] = gate ? 0.0f : fin[n];
            }
        }
    }
    // (which streams: the wave's 32 positions; with a participant list they are not contiguous -- reported as unknown)
    if (p.ids) raise_range_flag(bad, p.range_flag);
    else raise_range_flag(bad, p.range_flag, (blockIdx.x * WG + wave) * 32 + p.s_base, 32);
    }
}

}  // namespace owh
