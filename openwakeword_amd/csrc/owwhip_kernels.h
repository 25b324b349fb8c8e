// owwhip_kernels.h -- gfx950 device code of the streaming wake-word path (included by owwhip.hip only).
//
// Dataflow per 80 ms step, all S streams at once:
//   mel_kernel      int16 PCM (+480-sample tail) -> 8 new log-mel rows           (HBM bound, radix-8 FFT)
//   stageA_kernel   mel rows -> conv0(3x3) conv1(1x3) conv2(3x1) pool2x2         (fp32 MFMA)
//   stage_kernel<B> 4 convs + pool, three instances (48 / 72 / 96 channels)      (fp32 MFMA)
//   stage_kernel<E> 4 convs + pool + conv19 -> 96-d embedding -> feature ring    (fp32 MFMA)
//   heads64_kernel  16x96 features -> Linear/LN/ReLU x2 -> Linear -> sigmoid      (fp32 MFMA)
//   postproc_kernel first-5 zeroing, patience / debounce, score ring
//
// The embedding CNN is evaluated INCREMENTALLY: every time-axis op is 'valid' and the product of the
// time strides (2*1*2*1*2) equals the window hop (8 mel rows), so a step only computes the rows that are
// new (8,8,8 | 4,4,4,4 | 4,4,4,4 | 2,2,2,2 | 2,2,2,2 | 1) and keeps the last two input rows of every 3x1
// (and the 3x3) convolution as per-stream state (9,472 floats).  Reference semantics being replaced:
// /root/reference/openwakeword/utils.py:409-452 (full 76-row window re-evaluated through
// embedding_model.onnx every step) -- same outputs up to fp32 summation order.
//
// MFMA formulation of a conv layer (v_mfma_f32_16x16x4_f32, exact fp32):
//   D[cout 16][position 16] += A[cout][k 4] * B[k][position],  k = (tap, cin)
//   A = weights, pre-packed on the host so that one coalesced dword load per lane per k-step fills the
//       operand register; a wave keeps the whole K x 16 slab of its cout tile in VGPRs (<= 72 regs)
//   B = activations read from LDS with ds_read_b64 (two k-steps per read); channel stride C+4 floats
//       makes the 32-lane read groups conflict free (stride == 4*odd mod 64 banks)
//   D lane layout: lane&15 = position, (lane>>4)*4+reg = cout  -> one ds_write_b128 per lane epilogue
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace owk {

typedef float f32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ float leaky_clamp(float x) {
    // max(max(0.2x, x), -0.4)  (converting_google_speech_embedding_model.ipynb cell 18)
    return fmaxf(fmaxf(0.2f * x, x), -0.4f);
}

// ------------------------------------------------------------------------------------------------
// generic conv layer, LDS -> LDS
//   TIME=false : 1x3 conv along mel axis.  in  layout P[b][R][F+2][CIN+4]   (cols 0 and F+1 are zeros)
//                                           out layout Q[b][R+2][F][COUT+4]  rows 2.. (rows 0,1 = history)
//   TIME=true  : 3x1 conv along time axis.  in  layout Q[b][R+2][F][CIN+4]
//                                           out layout P[b][R][F+2][COUT+4]  cols 1..F
// ------------------------------------------------------------------------------------------------
template <int R, int F, bool TIME, int CPI, int CPO>
__device__ __forceinline__ void pos_addr(int p, int& rd, int& wr) {
    const int f = p % F;
    const int r = (p / F) % R;
    const int b = p / (F * R);
    if (TIME) {
        rd = ((b * (R + 2) + r) * F + f) * CPI;
        wr = ((b * R + r) * (F + 2) + f + 1) * CPO;
    } else {
        rd = ((b * R + r) * (F + 2) + f) * CPI;
        wr = ((b * (R + 2) + r + 2) * F + f) * CPO;
    }
}

#ifndef OWW_PF
#define OWW_PF 1          // LDS operand reads run this many 8-channel k-blocks ahead of the MFMAs (1 = compiler's choice)
#endif
#ifndef OWW_SCHED
#define OWW_SCHED 0       // pin the DS-read / MFMA interleave with sched_group_barrier
#endif
#ifndef OWW_PIN
#define OWW_PIN 0         // 1: weights loaded early (before the barrier) and pinned in VGPRs; 0: compiler streams them
#endif
#ifndef OWW_SPLIT
#define OWW_SPLIT 0       // 0: one cout tile per wave (NW = NCT*PSPLIT); 1: equal runs of (cout,position) tiles per wave
#endif

// Weights of the cout tile a wave is working on: K/4 MFMA operand registers per lane.
template <int CIN, int COUT, int NPT, int NW>
struct ConvW {
    static constexpr int KS = 3 * CIN / 4;
    static constexpr int NCT = (COUT + 15) / 16;
    static constexpr int ITEMS = NCT * NPT;
#if OWW_SPLIT
    static constexpr int PER = (ITEMS + NW - 1) / NW;
#else
    static constexpr int PSPLIT = NW / NCT;
    static_assert(PSPLIT >= 1, "need at least one wave per cout tile");
#endif
    float w[KS];
    __device__ __forceinline__ static int first_ct(int wave) {
#if OWW_SPLIT
        return wave * PER < ITEMS ? (wave * PER) / NPT : -1;
#else
        return wave < NCT * PSPLIT ? wave % NCT : -1;
#endif
    }
    // issue the loads of the wave's first cout tile early (before the barrier that precedes the layer)
    __device__ __forceinline__ void load(const float* __restrict__ wpk, int tid) {
#if OWW_PIN
        asm volatile("" ::: "memory");      // do not hoist these loads above the previous layer's MFMA loop
        const int wave = tid >> 6, lane = tid & 63;
        const int ct = first_ct(wave);
        if (ct < 0) return;
#pragma unroll
        for (int s = 0; s < KS; ++s) w[s] = wpk[(ct * KS + s) * 64 + lane];
#endif
    }
};

// MFMA body for one or two position tiles against the cout tile whose weights sit in W.w
template <int CIN, int COUT, int R, int F, int B, bool TIME, bool BN_ACT, bool PAIR, class WT>
__device__ __forceinline__ void conv_tiles(const float* __restrict__ in, float* __restrict__ out, WT& W, int ct, int t0, int t1,
                                           const f32x4 sc, const f32x4 sh, int lane) {
    constexpr int CPI = CIN + 4, CPO = COUT + 4;
    constexpr int NP = B * R * F;
    constexpr int TAPSTRIDE = TIME ? F * CPI : CPI;
    const int pl = lane & 15, j = lane >> 4;
    const int c0 = ct * 16 + j * 4;
    const bool cvalid = c0 < COUT;
    const int p0 = t0 * 16 + pl, p1 = t1 * 16 + pl;
    const bool v0ok = p0 < NP, v1ok = PAIR && (p1 < NP);
    int rd0, wr0, rd1 = 0, wr1 = 0;
    pos_addr<R, F, TIME, CPI, CPO>(v0ok ? p0 : NP - 1, rd0, wr0);
    if (PAIR) pos_addr<R, F, TIME, CPI, CPO>(v1ok ? p1 : NP - 1, rd1, wr1);
    f32x4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};
    constexpr int NKB = 3 * CIN / 8;           // k-blocks of 8 channels
    constexpr int PF = OWW_PF < 1 ? 1 : OWW_PF;
    auto kb_off = [&](int kb) { return (kb / (CIN / 8)) * TAPSTRIDE + (kb % (CIN / 8)) * 8 + 2 * j; };
    float2 q0[PF], q1[PF];
#pragma unroll
    for (int u = 0; u < PF; ++u) {
        q0[u] = *reinterpret_cast<const float2*>(in + rd0 + kb_off(u < NKB ? u : 0));
        if (PAIR) q1[u] = *reinterpret_cast<const float2*>(in + rd1 + kb_off(u < NKB ? u : 0));
    }
#pragma unroll
    for (int kb = 0; kb < NKB; ++kb) {
        const float2 x0 = q0[kb % PF];
        float2 x1 = x0;
        if (PAIR) x1 = q1[kb % PF];
        if (kb + PF < NKB) {
            q0[kb % PF] = *reinterpret_cast<const float2*>(in + rd0 + kb_off(kb + PF));
            if (PAIR) q1[kb % PF] = *reinterpret_cast<const float2*>(in + rd1 + kb_off(kb + PF));
        }
        acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(W.w[2 * kb], x0.x, acc0, 0, 0, 0);
        if (PAIR) acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(W.w[2 * kb], x1.x, acc1, 0, 0, 0);
        acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(W.w[2 * kb + 1], x0.y, acc0, 0, 0, 0);
        if (PAIR) acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(W.w[2 * kb + 1], x1.y, acc1, 0, 0, 0);
#if OWW_SCHED
        __builtin_amdgcn_sched_group_barrier(0x100, PAIR ? 2 : 1, 0);   // DS reads
        __builtin_amdgcn_sched_group_barrier(0x008, PAIR ? 4 : 2, 0);   // MFMA
#endif
    }
    if (cvalid) {
        if (BN_ACT) {
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                acc0[e] = leaky_clamp(acc0[e] * sc[e] + sh[e]);
                if (PAIR) acc1[e] = leaky_clamp(acc1[e] * sc[e] + sh[e]);
            }
        }
        if (v0ok) *reinterpret_cast<f32x4*>(out + wr0 + c0) = acc0;
        if (v1ok) *reinterpret_cast<f32x4*>(out + wr1 + c0) = acc1;
    }
}

template <int CIN, int COUT, int R, int F, int B, bool TIME, int NW, bool BN_ACT>
__device__ __forceinline__ void conv_mfma(const float* __restrict__ in, float* __restrict__ out,
                                          const float* __restrict__ wpk,
                                          ConvW<CIN, COUT, (B * R * F + 15) / 16, NW>& W,
                                          const float* __restrict__ scale, const float* __restrict__ shift, int tid) {
    static_assert(CIN % 8 == 0, "k-blocks of 8 channels");
    constexpr int NPT = (B * R * F + 15) / 16;
    using WT = ConvW<CIN, COUT, NPT, NW>;
    constexpr int KS = WT::KS;
    const int wave = tid >> 6, lane = tid & 63;
    const int j = lane >> 4;
    auto fetch = [&](int ct, bool preloaded) {
        if (!(OWW_PIN && preloaded)) {
#pragma unroll
            for (int s = 0; s < KS; ++s) W.w[s] = wpk[(ct * KS + s) * 64 + lane];
        }
#if OWW_PIN
#pragma unroll
        for (int s = 0; s < KS; ++s) asm volatile("" : "+v"(W.w[s]));
#endif
    };
    auto bn = [&](int ct, f32x4& sc, f32x4& sh) {
        sc = f32x4{1.f, 1.f, 1.f, 1.f}; sh = f32x4{0.f, 0.f, 0.f, 0.f};
        const int c0 = ct * 16 + j * 4;
        if (BN_ACT && c0 < COUT) {
            sc = *reinterpret_cast<const f32x4*>(scale + c0);
            sh = *reinterpret_cast<const f32x4*>(shift + c0);
        }
    };
#if OWW_SPLIT
    int i = wave * WT::PER;
    const int iend = min(i + WT::PER, WT::ITEMS);
    bool first = true;
    while (i < iend) {
        const int ct = i / NPT;
        int t = i - ct * NPT;
        const int tend = min(NPT, t + (iend - i));
        i += tend - t;
        fetch(ct, first);
        first = false;
        f32x4 sc, sh;
        bn(ct, sc, sh);
        for (; t + 1 < tend; t += 2)
            conv_tiles<CIN, COUT, R, F, B, TIME, BN_ACT, true>(in, out, W, ct, t, t + 1, sc, sh, lane);
        if (t < tend)
            conv_tiles<CIN, COUT, R, F, B, TIME, BN_ACT, false>(in, out, W, ct, t, t, sc, sh, lane);
    }
#else
    constexpr int PSPLIT = WT::PSPLIT;
    if (wave >= WT::NCT * PSPLIT) return;
    const int ct = wave % WT::NCT, part = wave / WT::NCT;
    fetch(ct, true);
    f32x4 sc, sh;
    bn(ct, sc, sh);
    int t = part;
    for (; t + PSPLIT < NPT; t += 2 * PSPLIT)
        conv_tiles<CIN, COUT, R, F, B, TIME, BN_ACT, true>(in, out, W, ct, t, t + PSPLIT, sc, sh, lane);
    if (t < NPT)
        conv_tiles<CIN, COUT, R, F, B, TIME, BN_ACT, false>(in, out, W, ct, t, t, sc, sh, lane);
#endif
}

// plain-VALU version of the same layer (same LDS layouts; weights in natural [tap][cin][cout] order)
template <int CIN, int COUT, int R, int F, int B, bool TIME, int NW, bool BN_ACT>
__device__ __forceinline__ void conv_valu(const float* __restrict__ in, float* __restrict__ out,
                                          const float* __restrict__ wraw, const float* __restrict__ scale,
                                          const float* __restrict__ shift, int tid) {
    constexpr int CPI = CIN + 4, CPO = COUT + 4;
    constexpr int NP = B * R * F;
    constexpr int TAPSTRIDE = TIME ? F * CPI : CPI;
    for (int idx = tid; idx < NP * COUT; idx += NW * 64) {
        const int p = idx / COUT, c = idx - p * COUT;
        int rd, wr;
        pos_addr<R, F, TIME, CPI, CPO>(p, rd, wr);
        float acc = 0.f;
        for (int tap = 0; tap < 3; ++tap)
            for (int ci = 0; ci < CIN; ++ci)
                acc = fmaf(in[rd + tap * TAPSTRIDE + ci], wraw[(tap * CIN + ci) * COUT + c], acc);
        out[wr + c] = BN_ACT ? leaky_clamp(acc * scale[c] + shift[c]) : acc;
    }
}

template <bool MFMA, int CIN, int COUT, int R, int F, int B, bool TIME, int NW, bool BN_ACT>
__device__ __forceinline__ void conv_layer(const float* in, float* out, const float* w,
                                           ConvW<CIN, COUT, (B * R * F + 15) / 16, NW>& W,
                                           const float* scale, const float* shift, int tid) {
    if (MFMA) conv_mfma<CIN, COUT, R, F, B, TIME, NW, BN_ACT>(in, out, w, W, scale, shift, tid);
    else      conv_valu<CIN, COUT, R, F, B, TIME, NW, BN_ACT>(in, out, w, scale, shift, tid);
}

// dense global [B][ROWS][F][C] <-> LDS helpers -----------------------------------------------------
// copy global dense rows into an LDS buffer laid out [b][BROWS][FW][CP] at (row0 + row, f0 + f)
template <int B, int ROWS, int F, int C, int BROWS, int FW, int CP, int NT>
__device__ __forceinline__ void g2l(const float* __restrict__ g, float* __restrict__ l, int row0, int f0, int tid) {
    constexpr int N4 = B * ROWS * F * C / 4;
    for (int i = tid; i < N4; i += NT) {
        const int e = i * 4;
        const int c = e % C;
        const int pos = e / C;
        const int f = pos % F;
        const int rb = pos / F;
        const int row = rb % ROWS, b = rb / ROWS;
        const f32x4 v = reinterpret_cast<const f32x4*>(g)[i];
        *reinterpret_cast<f32x4*>(l + ((b * BROWS + row0 + row) * FW + f0 + f) * CP + c) = v;
    }
}
template <int B, int ROWS, int F, int C, int BROWS, int FW, int CP, int NT>
__device__ __forceinline__ void l2g(const float* __restrict__ l, float* __restrict__ g, int row0, int f0, int tid) {
    constexpr int N4 = B * ROWS * F * C / 4;
    for (int i = tid; i < N4; i += NT) {
        const int e = i * 4;
        const int c = e % C;
        const int pos = e / C;
        const int f = pos % F;
        const int rb = pos / F;
        const int row = rb % ROWS, b = rb / ROWS;
        reinterpret_cast<f32x4*>(g)[i] =
            *reinterpret_cast<const f32x4*>(l + ((b * BROWS + row0 + row) * FW + f0 + f) * CP + c);
    }
}
// debug copy: LDS [b][BROWS][FW][CP] rows row0.. -> per-stream dense blocks g[(s0+b)*stride + off + ...]
template <int B, int ROWS, int F, int C, int BROWS, int FW, int CP, int NT>
__device__ __forceinline__ void l2dbg(const float* __restrict__ l, float* __restrict__ dbg, size_t stride, int off,
                                      int s0, int row0, int f0, int tid) {
    constexpr int N = B * ROWS * F * C;
    for (int i = tid; i < N; i += NT) {
        const int c = i % C;
        const int pos = i / C;
        const int f = pos % F;
        const int rb = pos / F;
        const int row = rb % ROWS, b = rb / ROWS;
        dbg[(size_t)(s0 + b) * stride + off + (row * F + f) * C + c] =
            l[((b * BROWS + row0 + row) * FW + f0 + f) * CP + c];
    }
}
// zero the two padding columns (f=0 and f=F+1) of a P buffer [b*R rows][F+2][CP] for channels < C
template <int BR, int F, int C, int CP, int NT>
__device__ __forceinline__ void zero_pads(float* __restrict__ P, int tid) {
    for (int i = tid; i < BR * 2 * C; i += NT) {
        const int c = i % C;
        const int t = i / C;
        const int side = t & 1, row = t >> 1;
        P[(row * (F + 2) + (side ? F + 1 : 0)) * CP + c] = 0.f;
    }
}

// ------------------------------------------------------------------------------------------------
// stage B / C / D / E
// ------------------------------------------------------------------------------------------------
template <int CIN_, int C_, int R_, int F_, int PT_, int PF_, int B_, int NW_>
struct StageCfg {
    static constexpr int CIN = CIN_, C = C_, R = R_, F = F_, PT = PT_, PF = PF_, B = B_, NW = NW_;
    static constexpr int CP = C + 4;
    static constexpr int P_FLOATS = B * R * (F + 2) * CP;
    static constexpr int Q_FLOATS = B * (R + 2) * F * CP;
    static constexpr int LDS_BYTES = (P_FLOATS + Q_FLOATS) * 4;
    static constexpr int NT = NW * 64;
};
#if OWW_SPLIT
using CfgB = StageCfg<24, 48, 4, 16, 1, 2, 2, 8>;
using CfgC = StageCfg<48, 72, 4, 8, 2, 2, 2, 4>;
using CfgD = StageCfg<72, 96, 2, 4, 1, 2, 4, 4>;
using CfgE = StageCfg<96, 96, 2, 2, 2, 2, 8, 4>;
#else
using CfgB = StageCfg<24, 48, 4, 16, 1, 2, 2, 6>;
using CfgC = StageCfg<48, 72, 4, 8, 2, 2, 2, 5>;
using CfgD = StageCfg<72, 96, 2, 4, 1, 2, 4, 6>;
using CfgE = StageCfg<96, 96, 2, 2, 2, 2, 8, 6>;
#endif

struct StageParams {
    const float* xin;      // [S][R][F][CIN]
    float* xout;           // [S][R/PT][F/PF][C]           (unused by the last stage)
    float* hist_b;         // [S][2][F][C]  input history of the 1st 3x1 conv
    float* hist_d;         // [S][2][F][C]  input history of the 2nd 3x1 conv
    const float* w[4];     // MFMA-packed or natural weights of the four convs
    const float* scale[4];
    const float* shift[4];
    // last stage only
    float* hist19;         // [S][2][96]
    const float* w19;
    float* feat;           // [S][TR][96] feature ring
    float* emb;            // [S][96]     newest embedding (dense copy)
    const uint32_t* nfeat; // [S] embeddings appended since reset
    int TR;
    // debug
    float* dbg;            // [S][dbg_stride] or null
    size_t dbg_stride;
    int dbg_off[5];
    long long* prof;       // [16 waves][16 marks] shader-clock stamps of workgroup prof_block, or null
    int prof_block;
};
#define OWW_MARK(k) do { if (p.prof && (int)blockIdx.x == p.prof_block && (tid & 63) == 0) p.prof[(tid >> 6) * 16 + (k)] = clock64(); } while (0)

template <class C, bool MFMA, bool LAST>
__global__ __launch_bounds__(C::NT) void stage_kernel(StageParams p) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    constexpr int CIN = C::CIN, CC = C::C, R = C::R, F = C::F, B = C::B, NT = C::NT, NW = C::NW;
    constexpr int CPI = CIN + 4, CP = C::CP;
    float* P = smem;
    float* Q = smem + C::P_FLOATS;
    const int tid = threadIdx.x;
    const int s0 = blockIdx.x * B;

    constexpr int NPT = (B * R * F + 15) / 16;
    ConvW<CIN, CC, NPT, NW> Wa;
    ConvW<CC, CC, NPT, NW> Wb, Wc, Wd;
    ConvW<CC, CC, (B + 15) / 16, NW> W19;
    OWW_MARK(0);
    if (MFMA) Wa.load(p.w[0], tid);
    // history of conv d is needed two barriers from now: fetch it into registers right away
    constexpr int HN4 = B * 2 * F * CC / 4;
    constexpr int HPT = (HN4 + NT - 1) / NT;
    f32x4 hd[HPT];
#pragma unroll
    for (int u = 0; u < HPT; ++u) {
        const int i = tid + u * NT;
        if (i < HN4) hd[u] = reinterpret_cast<const f32x4*>(p.hist_d + (size_t)s0 * 2 * F * CC)[i];
    }
    // phase 0: stage input (CIN layout) + history of conv b
    zero_pads<B * R, F, CIN, CPI, NT>(P, tid);
    g2l<B, R, F, CIN, R, F + 2, CPI, NT>(p.xin + (size_t)s0 * R * F * CIN, P, 0, 1, tid);
    g2l<B, 2, F, CC, R + 2, F, CP, NT>(p.hist_b + (size_t)s0 * 2 * F * CC, Q, 0, 0, tid);
    OWW_MARK(1);
    __syncthreads();
    OWW_MARK(2);
    // conv a: 1x3 CIN -> C
    conv_layer<MFMA, CIN, CC, R, F, B, false, NW, true>(P, Q, p.w[0], Wa, p.scale[0], p.shift[0], tid);
    if (MFMA) Wb.load(p.w[1], tid);
    OWW_MARK(3);
    __syncthreads();
    OWW_MARK(4);
    // new history of conv b = last two rows of its input; conv b: 3x1
    l2g<B, 2, F, CC, R + 2, F, CP, NT>(Q, p.hist_b + (size_t)s0 * 2 * F * CC, R, 0, tid);
    if (p.dbg) l2dbg<B, R, F, CC, R + 2, F, CP, NT>(Q, p.dbg, p.dbg_stride, p.dbg_off[0], s0, 2, 0, tid);
    zero_pads<B * R, F, CC, CP, NT>(P, tid);
    conv_layer<MFMA, CC, CC, R, F, B, true, NW, true>(Q, P, p.w[1], Wb, p.scale[1], p.shift[1], tid);
    if (MFMA) Wc.load(p.w[2], tid);
    OWW_MARK(5);
    __syncthreads();
    OWW_MARK(6);
    // conv c: 1x3 (its output rows 2.. of Q; rows 0,1 = prefetched history of conv d)
#pragma unroll
    for (int u = 0; u < HPT; ++u) {
        const int i = tid + u * NT;
        if (i < HN4) {
            const int e = i * 4;
            const int c = e % CC;
            const int pos = e / CC;
            const int f = pos % F;
            const int rb = pos / F;
            const int row = rb % 2, b = rb / 2;
            *reinterpret_cast<f32x4*>(Q + ((b * (R + 2) + row) * F + f) * CP + c) = hd[u];
        }
    }
    if (p.dbg) l2dbg<B, R, F, CC, R, F + 2, CP, NT>(P, p.dbg, p.dbg_stride, p.dbg_off[1], s0, 0, 1, tid);
    conv_layer<MFMA, CC, CC, R, F, B, false, NW, true>(P, Q, p.w[2], Wc, p.scale[2], p.shift[2], tid);
    if (MFMA) Wd.load(p.w[3], tid);
    OWW_MARK(7);
    __syncthreads();
    OWW_MARK(8);
    // conv d: 3x1
    l2g<B, 2, F, CC, R + 2, F, CP, NT>(Q, p.hist_d + (size_t)s0 * 2 * F * CC, R, 0, tid);
    if (p.dbg) l2dbg<B, R, F, CC, R + 2, F, CP, NT>(Q, p.dbg, p.dbg_stride, p.dbg_off[2], s0, 2, 0, tid);
    conv_layer<MFMA, CC, CC, R, F, B, true, NW, true>(Q, P, p.w[3], Wd, p.scale[3], p.shift[3], tid);
    if (MFMA && LAST) W19.load(p.w19, tid);
    OWW_MARK(9);
    __syncthreads();
    OWW_MARK(10);
    if (p.dbg) l2dbg<B, R, F, CC, R, F + 2, CP, NT>(P, p.dbg, p.dbg_stride, p.dbg_off[3], s0, 0, 1, tid);

    // max pool PT x PF
    constexpr int RO = R / C::PT, FO = F / C::PF;
    if (!LAST) {
        float* xo = p.xout + (size_t)s0 * RO * FO * CC;
        for (int i = tid; i < B * RO * FO * CC; i += NT) {
            const int c = i % CC;
            const int pos = i / CC;
            const int fo = pos % FO;
            const int t = pos / FO;
            const int ro = t % RO, b = t / RO;
            float m = -INFINITY;
#pragma unroll
            for (int dt = 0; dt < C::PT; ++dt)
#pragma unroll
                for (int df = 0; df < C::PF; ++df)
                    m = fmaxf(m, P[((b * R + ro * C::PT + dt) * (F + 2) + fo * C::PF + df + 1) * CP + c]);
            xo[i] = m;
        }
    } else {
        // RO == FO == 1: pooled row feeds conv19 (3x1, 96->96, no BN/activation) over [hist19(2) ; pooled]
        static_assert(!LAST || (RO == 1 && FO == 1), "last stage pools to one position");
        // Q19 layout [b][3][1][CP] in Q (free: conv d finished reading it before the last barrier)
        for (int i = tid; i < B * CC; i += NT) {
            const int c = i % CC, b = i / CC;
            float m = -INFINITY;
#pragma unroll
            for (int dt = 0; dt < C::PT; ++dt)
#pragma unroll
                for (int df = 0; df < C::PF; ++df)
                    m = fmaxf(m, P[((b * R + dt) * (F + 2) + df + 1) * CP + c]);
            Q[(b * 3 + 2) * CP + c] = m;
        }
        g2l<B, 2, 1, CC, 3, 1, CP, NT>(p.hist19 + (size_t)s0 * 2 * CC, Q, 0, 0, tid);
        __syncthreads();
        // P19 layout [b][1][3][CP] in P (pool finished reading P before the barrier above)
        conv_layer<MFMA, CC, CC, 1, 1, B, true, NW, false>(Q, P, p.w19, W19, nullptr, nullptr, tid);
        l2g<B, 2, 1, CC, 3, 1, CP, NT>(Q, p.hist19 + (size_t)s0 * 2 * CC, 1, 0, tid);
        __syncthreads();
        for (int i = tid; i < B * CC; i += NT) {
            const int c = i % CC, b = i / CC;
            const float v = P[(b * 3 + 1) * CP + c];
            const int s = s0 + b;
            const uint32_t slot = p.nfeat[s] % (uint32_t)p.TR;
            p.feat[((size_t)s * p.TR + slot) * CC + c] = v;
            p.emb[(size_t)s * CC + c] = v;
            if (p.dbg) p.dbg[(size_t)s * p.dbg_stride + p.dbg_off[4] + c] = v;
        }
    }
    OWW_MARK(11);
}

// ------------------------------------------------------------------------------------------------
// stage A: mel rows -> conv0 (3x3, 1->24, ReLU, BN, act) -> conv1 (1x3) -> conv2 (3x1) -> pool 2x2
// one stream per workgroup, 8 waves
// ------------------------------------------------------------------------------------------------
struct StageAParams {
    const float* mel;      // [S][mel_stride] transformed mel rows of this call; chunk rows start at mel_off
    int mel_stride, mel_off;
    float* hist_mel;       // [S][2][32]
    float* hist2;          // [S][2][32][24]
    const float* w0;       // natural [9][24] (VALU) or MFMA-packed [2][3][64] (K padded 9 -> 12)
    const float* w1;       // packed / natural
    const float* w2;
    const float* scale[3];
    const float* shift[3];
    float* xout;           // [S][4][16][24]
    float* dbg;
    size_t dbg_stride;
    int dbg_off[3];
};
struct CfgA {
    static constexpr int NW = 8, NT = 512, R = 8, F = 32, C = 24, CP = 28;
    static constexpr int M_FLOATS = 352;                       // [10][34] padded to a 16-byte multiple
    static constexpr int P_FLOATS = R * (F + 2) * CP;          // 7616
    static constexpr int Q_FLOATS = (R + 2) * F * CP;          // 8960
    static constexpr int LDS_BYTES = (M_FLOATS + P_FLOATS + Q_FLOATS) * 4;
};

template <bool MFMA>
__global__ __launch_bounds__(CfgA::NT) void stageA_kernel(StageAParams p) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    constexpr int R = 8, F = 32, CC = 24, CP = 28, NT = CfgA::NT, NW = CfgA::NW;
    float* M = smem;                       // [10][34]
    float* P = smem + CfgA::M_FLOATS;      // [8][34][28]
    float* Q = P + CfgA::P_FLOATS;         // [10][32][28]
    const int tid = threadIdx.x;
    const int s = blockIdx.x;

    ConvW<CC, CC, 16, NW> W1, W2;
    float w0r[3];
    if (MFMA) {
        const int wave = tid >> 6, lane = tid & 63;
#pragma unroll
        for (int k = 0; k < 3; ++k) w0r[k] = p.w0[((wave & 1) * 3 + k) * 64 + lane];
        W1.load(p.w1, tid);
    }
    // phase 0
    for (int i = tid; i < 10 * 34; i += NT) {
        const int row = i / 34, col = i % 34;
        float v = 0.f;
        if (col >= 1 && col <= 32) {
            v = (row < 2) ? p.hist_mel[(size_t)s * 64 + row * 32 + col - 1]
                          : p.mel[(size_t)s * p.mel_stride + p.mel_off + (row - 2) * 32 + col - 1];
        }
        M[i] = v;
    }
    zero_pads<R, F, CC, CP, NT>(P, tid);
    g2l<1, 2, F, CC, R + 2, F, CP, NT>(p.hist2 + (size_t)s * 2 * F * CC, Q, 0, 0, tid);
    __syncthreads();
    // conv0: 3x3 valid in time, zero padded in mel; ReLU; BN; activation
    if (MFMA) {
        // K = 9 taps padded to 12: three k-steps of v_mfma_f32_16x16x4_f32, operand B gathered from the mel tile
        const int wave = tid >> 6, lane = tid & 63;
        const int ct = wave & 1, part = wave >> 1, pl = lane & 15, j = lane >> 4;
        const int c0 = ct * 16 + j * 4;
        const bool cvalid = c0 < CC;
        f32x4 sc = {1.f, 1.f, 1.f, 1.f}, sh = {0.f, 0.f, 0.f, 0.f};
        if (cvalid) { sc = *reinterpret_cast<const f32x4*>(p.scale[0] + c0); sh = *reinterpret_cast<const f32x4*>(p.shift[0] + c0); }
        int koff[3];
#pragma unroll
        for (int k = 0; k < 3; ++k) { const int kk = min(4 * k + j, 8); koff[k] = (kk / 3) * 34 + (kk % 3); }
#pragma unroll
        for (int it = 0; it < 4; ++it) {
            const int pidx = (part + 4 * it) * 16 + pl;
            const int f = pidx % F, r = pidx / F;
            f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int k = 0; k < 3; ++k) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(w0r[k], M[r * 34 + f + koff[k]], acc, 0, 0, 0);
            if (cvalid) {
#pragma unroll
                for (int e = 0; e < 4; ++e) acc[e] = leaky_clamp(fmaxf(acc[e], 0.f) * sc[e] + sh[e]);
                *reinterpret_cast<f32x4*>(P + (r * (F + 2) + f + 1) * CP + c0) = acc;
            }
        }
    } else {
        const float* __restrict__ w0 = p.w0;
        const float* __restrict__ sc = p.scale[0];
        const float* __restrict__ sh = p.shift[0];
        for (int idx = tid; idx < R * F * CC; idx += NT) {
            const int c = idx % CC;
            const int pos = idx / CC;
            const int f = pos % F, r = pos / F;
            float acc = 0.f;
#pragma unroll
            for (int dt = 0; dt < 3; ++dt)
#pragma unroll
                for (int df = 0; df < 3; ++df)
                    acc = fmaf(M[(r + dt) * 34 + f + df], w0[(dt * 3 + df) * CC + c], acc);
            acc = fmaxf(acc, 0.f);
            P[(r * (F + 2) + f + 1) * CP + c] = leaky_clamp(acc * sc[c] + sh[c]);
        }
    }
    if (tid < 64) p.hist_mel[(size_t)s * 64 + tid] = M[(8 + tid / 32) * 34 + 1 + (tid % 32)];
    __syncthreads();
    if (p.dbg) l2dbg<1, R, F, CC, R, F + 2, CP, NT>(P, p.dbg, p.dbg_stride, p.dbg_off[0], s, 0, 1, tid);
    conv_layer<MFMA, CC, CC, R, F, 1, false, NW, true>(P, Q, p.w1, W1, p.scale[1], p.shift[1], tid);
    if (MFMA) W2.load(p.w2, tid);
    __syncthreads();
    l2g<1, 2, F, CC, R + 2, F, CP, NT>(Q, p.hist2 + (size_t)s * 2 * F * CC, R, 0, tid);
    if (p.dbg) l2dbg<1, R, F, CC, R + 2, F, CP, NT>(Q, p.dbg, p.dbg_stride, p.dbg_off[1], s, 2, 0, tid);
    conv_layer<MFMA, CC, CC, R, F, 1, true, NW, true>(Q, P, p.w2, W2, p.scale[2], p.shift[2], tid);
    __syncthreads();
    if (p.dbg) l2dbg<1, R, F, CC, R, F + 2, CP, NT>(P, p.dbg, p.dbg_stride, p.dbg_off[2], s, 0, 1, tid);
    float* xo = p.xout + (size_t)s * 4 * 16 * CC;
    for (int i = tid; i < 4 * 16 * CC; i += NT) {
        const int c = i % CC;
        const int pos = i / CC;
        const int fo = pos % 16, ro = pos / 16;
        float m = -INFINITY;
#pragma unroll
        for (int dt = 0; dt < 2; ++dt)
#pragma unroll
            for (int df = 0; df < 2; ++df)
                m = fmaxf(m, P[((ro * 2 + dt) * (F + 2) + fo * 2 + df + 1) * CP + c]);
        xo[i] = m;
    }
}

// ------------------------------------------------------------------------------------------------
// mel front end: one workgroup (4 waves) per stream-step; wave w turns frames 2w and 2w+1 of an
// 8-frame group into ONE 512-point complex FFT (z = frame_a + i*frame_b), radix-8 x 3 with two LDS
// transposes, then splits the two real spectra, |.|^2 for bins 2..121, sparse mel (<=16 taps),
// 10*log10, clamp at (max over the call) - 80 dB, x/10 + 2.
// Replaces melspectrogram.onnx + utils.py:180-208,387-401 (SURVEY Appendix A is the recipe).
// ------------------------------------------------------------------------------------------------
struct MelParams {
    const int16_t* pcm;     // [S][n_samples]
    int n_samples;          // samples per stream in this call
    int n_frames;           // frames to produce per stream
    int streaming;          // 1: virtual buffer = tail(480) ++ pcm, first-call masking, tail update, x/10+2
                            // 0: clip mode: frames straight from pcm, raw dB out, per-stream max to `smax`
    int16_t* tail;          // [S][480]
    const uint32_t* nfeat;  // [S]
    float* out;             // [S][n_frames][32]
    float* smax;            // [S] (clip mode)
    const float* hann;      // [400]
    const int* mel_start;   // [32]
    const float* mel_taps;  // [32][16]
    int S;
    const uint8_t* stream_on;   // oww_step_masked (streaming mode): a stream with stream_on[s] == 0 keeps its sample tail; nullptr = all take part
    // A call longer than the handle's mel buffer (oww_step with n_chunks > max_chunks) is evaluated in slices that share ONE clamp floor,
    // the reference's: melspectrogram.onnx runs once over the whole call (utils.py:387-401), so "max - 80 dB" is the call's maximum.
    int pcm_stride;             // samples between two streams' rows of `pcm` (0 = n_samples): a slice of a longer call
    int max_only;               // streaming mode, first pass over the WHOLE call: only its maximum (-> smax); no rows, no tail update
    const float* floor_max;     // streaming mode, the slices: [S] the call's maximum from that pass (replaces the slice's own)
};

// dB value and host transform of the mel front end (ipynb cell 15: 10 log10(max(p, 1e-10)); utils.py:180,206: x / 10 + 2) with the two
// divisions by constants written as multiplications and the hardware log2 (a true fp32 division / the full logf are ~10 VALU
// instructions each here; the result moves by a few ulp, i.e. < 2e-5 dB / 2e-6 mel units -- tests hold the rows to 5e-4)
// (v_log_f32 -- log2, 1 ulp -- times 10 log10(2): the full-precision logf costs ~10 VALU more per value for digits below 1e-5 dB)
__device__ __forceinline__ float db10(float p) { return __builtin_amdgcn_logf(fmaxf(p, 1e-10f)) * 3.0102999566398120f; }
__device__ __forceinline__ float mel_units(float db, float floor_db) { return fmaf(fmaxf(db, floor_db), 0.1f, 2.0f); }

__device__ __forceinline__ void dft8(float* re, float* im) {
    // forward 8-point DFT, natural order in and out
    const float h = 0.70710678118654752440f;
    float a0r = re[0] + re[4], a0i = im[0] + im[4], a1r = re[0] - re[4], a1i = im[0] - im[4];
    float a2r = re[2] + re[6], a2i = im[2] + im[6], t3r = re[2] - re[6], t3i = im[2] - im[6];
    float a3r = t3i, a3i = -t3r;                                   // * (-i)
    float b0r = re[1] + re[5], b0i = im[1] + im[5], b1r = re[1] - re[5], b1i = im[1] - im[5];
    float b2r = re[3] + re[7], b2i = im[3] + im[7], u3r = re[3] - re[7], u3i = im[3] - im[7];
    float b3r = u3i, b3i = -u3r;                                   // * (-i)
    float E0r = a0r + a2r, E0i = a0i + a2i, E2r = a0r - a2r, E2i = a0i - a2i;
    float E1r = a1r + a3r, E1i = a1i + a3i, E3r = a1r - a3r, E3i = a1i - a3i;
    float O0r = b0r + b2r, O0i = b0i + b2i, O2r = b0r - b2r, O2i = b0i - b2i;
    float O1r = b1r + b3r, O1i = b1i + b3i, O3r = b1r - b3r, O3i = b1i - b3i;
    // twiddles W8^k: 1, (1-i)/sqrt2, -i, (-1-i)/sqrt2
    float T1r = h * (O1r + O1i), T1i = h * (O1i - O1r);
    float T2r = O2i, T2i = -O2r;
    float T3r = h * (O3i - O3r), T3i = -h * (O3r + O3i);
    re[0] = E0r + O0r; im[0] = E0i + O0i; re[4] = E0r - O0r; im[4] = E0i - O0i;
    re[1] = E1r + T1r; im[1] = E1i + T1i; re[5] = E1r - T1r; im[5] = E1i - T1i;
    re[2] = E2r + T2r; im[2] = E2i + T2i; re[6] = E2r - T2r; im[6] = E2i - T2i;
    re[3] = E3r + T3r; im[3] = E3i + T3i; re[7] = E3r - T3r; im[7] = E3i - T3i;
}

constexpr int MEL_NT = 256;
constexpr int MEL_WX = 672 + 8;         // samples a wave needs for its two frames (160 + 512), padded
constexpr int MEL_PBINS = 120;          // FFT bins 2..121 are the only ones the filterbank touches

// wave-local ordering point: the LDS operations of one wave execute in order, so data exchanged between the lanes of ONE
// wave through LDS needs no s_barrier -- only the outstanding LDS operations must have been issued/completed and the
// compiler must not move accesses across this point
__device__ __forceinline__ void wave_sync() {
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
}

// The 672(+8) samples wave `wave` needs for frames 2*wave, 2*wave+1 of group g of stream s, as raw int16 in registers
// (lane l: samples 8l..8l+7 and, lanes < 21, 512+8l..): virtual index c of [tail(hist) ; pcm], zero beyond the input.
// Issued one loop iteration ahead of its use so the HBM latency overlaps the previous FFT.
__device__ __forceinline__ void mel_fetch(const MelParams& p, int s, int g, int wave, int lane, int hist, int4 (&raw)[2]) {
    const int16_t* pcm = p.pcm + (size_t)s * (p.pcm_stride ? p.pcm_stride : p.n_samples);
    const int16_t* tail = p.tail + (size_t)s * 480;
    const int c0 = g * 1280 + 320 * wave;
#pragma unroll
    for (int u = 0; u < 2; ++u) {
        const int i = lane * 8 + u * 512;
        if (i >= MEL_WX) continue;
        const int c = c0 + i;
        if (c + 8 <= hist) {
            raw[u] = *reinterpret_cast<const int4*>(tail + c);
        } else if (c >= hist && c - hist + 8 <= p.n_samples && ((reinterpret_cast<uintptr_t>(pcm + (c - hist)) & 15) == 0)) {
            raw[u] = *reinterpret_cast<const int4*>(pcm + (c - hist));
        } else {
            alignas(16) int16_t h[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const int cc = c + e;
                int16_t x = 0;
                if (cc < hist) x = tail[cc];
                else if (cc - hist < p.n_samples) x = pcm[cc - hist];
                h[e] = x;
            }
            raw[u] = *reinterpret_cast<const int4*>(h);
        }
    }
}

// One workgroup (4 waves) per stream-step; wave w owns frames 2w and 2w+1 of every 8-frame group from the PCM samples to
// the log-mel values (its own LDS regions, no workgroup barrier); the waves only meet once per call for the clamp maximum.
__global__ __launch_bounds__(MEL_NT) void mel_kernel(MelParams p) {
    __shared__ float s_hann[400];
    // a wave's PCM samples are dead once the windowed values are in registers: they share the LDS of the FFT transposes
    // (re and im planes, contiguous per wave) -- 24 KB instead of 35 KB per workgroup, six resident workgroups per CU
    __shared__ float s_z[4][2 * 576];
    static_assert(MEL_WX <= 2 * 576, "sample window fits the transpose planes");
    __shared__ float s_red[2][4];
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;

    for (int i = tid; i < 400; i += MEL_NT) s_hann[i] = p.hann[i];
    // per-lane twiddles (registers, computed once per workgroup lifetime)
    float tw1r[8], tw1i[8], tw2r[8], tw2i[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        sincospif(-(float)(lane * k) / 256.f, &tw1i[k], &tw1r[k]);          // exp(-2 pi i lane k / 512)
        sincospif(-(float)((lane & 7) * k) / 32.f, &tw2i[k], &tw2r[k]);     // exp(-2 pi i m0 k / 64)
    }
    // mel filter of this thread's output bin; thread (fr, mbin) with fr = tid >> 5 in {2 wave, 2 wave + 1}: its own wave's frames
    const int mbin = tid & 31, fr = tid >> 5;
    const int mstart = p.mel_start[mbin] - 2;
    float taps[16];
#pragma unroll
    for (int t = 0; t < 16; ++t) taps[t] = p.mel_taps[mbin * 16 + t];
    const int n_groups = (p.n_frames + 7) / 8;
    const int hist = p.streaming ? 480 : 0;
    float* sx = s_z[wave];
    float* xr = s_z[wave];
    float* xi = s_z[wave] + 576;
    // power spectra of the wave's two frames: the last FFT stage leaves xr[128..383] unused (only bins < 128 and >= 384 are
    // written back), exactly 2 x 128 floats -- 20 KB of LDS per workgroup, seven resident workgroups per CU
    static_assert(MEL_PBINS + 8 <= 128, "power rows fit the unused middle of the re plane");
    float* pw0 = xr + 128;
    float* pw1 = xr + 256;
    __syncthreads();                             // s_hann

    int it = 0;
    int4 raw[2] = {};
    if ((int)blockIdx.x < p.S) mel_fetch(p, blockIdx.x, 0, wave, lane, hist, raw);
    for (int s = blockIdx.x; s < p.S; s += gridDim.x, ++it) {
        const int16_t* pcm = p.pcm + (size_t)s * (p.pcm_stride ? p.pcm_stride : p.n_samples);
        const bool first = p.streaming && (p.nfeat[s] == 0);
        float vmax = -INFINITY;
        float last_db = 0.f;
        for (int g = 0; g < n_groups; ++g) {
            // ---- this wave's samples (fetched one iteration ahead, see mel_fetch): int16 -> float into the wave's LDS window
            wave_sync();                         // previous group's readers of sx / s_pow (same wave) are done
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                const int i = lane * 8 + u * 512;
                if (i < MEL_WX) {
                    const int16_t* h = reinterpret_cast<const int16_t*>(&raw[u]);
#pragma unroll
                    for (int e = 0; e < 8; ++e) sx[i + e] = (float)h[e];
                }
            }
            {   // the next unit of work of this wave: next group of this stream, else group 0 of the workgroup's next stream
                const bool same = g + 1 < n_groups;
                const int sn = same ? s : s + (int)gridDim.x, gn = same ? g + 1 : 0;
                if (sn < p.S) mel_fetch(p, sn, gn, wave, lane, hist, raw);
            }
            wave_sync();
            // ---- one complex FFT per wave: z = frame_a + i * frame_b (frames 2*wave and 2*wave + 1 of this group)
            float re[8], im[8];
#pragma unroll
            for (int n2 = 0; n2 < 8; ++n2) {
                const int n = 64 * n2 + lane;
                const bool in = (n >= 56) && (n < 456);
                const float w = in ? s_hann[in ? n - 56 : 0] : 0.f;
                re[n2] = w * sx[n];
                im[n2] = w * sx[160 + n];
            }
            dft8(re, im);
#pragma unroll
            for (int k = 1; k < 8; ++k) {
                const float r = re[k] * tw1r[k] - im[k] * tw1i[k];
                im[k] = re[k] * tw1i[k] + im[k] * tw1r[k];
                re[k] = r;
            }
#pragma unroll
            for (int k = 0; k < 8; ++k) { xr[k * 72 + lane] = re[k]; xi[k * 72 + lane] = im[k]; }
            wave_sync();
            {
                const int q = lane >> 3, m0 = lane & 7;
#pragma unroll
                for (int m1 = 0; m1 < 8; ++m1) { re[m1] = xr[q * 72 + m1 * 8 + m0]; im[m1] = xi[q * 72 + m1 * 8 + m0]; }
            }
            dft8(re, im);
#pragma unroll
            for (int k = 1; k < 8; ++k) {
                const float r = re[k] * tw2r[k] - im[k] * tw2i[k];
                im[k] = re[k] * tw2i[k] + im[k] * tw2r[k];
                re[k] = r;
            }
            wave_sync();
            {
                const int q = lane >> 3, m0 = lane & 7;
#pragma unroll
                for (int k1 = 0; k1 < 8; ++k1) { xr[(q * 8 + k1) * 9 + m0] = re[k1]; xi[(q * 8 + k1) * 9 + m0] = im[k1]; }
            }
            wave_sync();
#pragma unroll
            for (int m0 = 0; m0 < 8; ++m0) { re[m0] = xr[lane * 9 + m0]; im[m0] = xi[lane * 9 + m0]; }
            dft8(re, im);
            wave_sync();
            {
                // lane = k0*8 + k1 holds Z[k0 + 8*k1 + 64*k2], k2 = 0..7; only k<128 and k>=384 are needed
                const int kb = (lane >> 3) + 8 * (lane & 7);
                xr[kb] = re[0];        xi[kb] = im[0];
                xr[kb + 64] = re[1];   xi[kb + 64] = im[1];
                xr[kb + 384] = re[6];  xi[kb + 384] = im[6];
                xr[kb + 448] = re[7];  xi[kb + 448] = im[7];
            }
            wave_sync();
            for (int i = lane; i < MEL_PBINS; i += 64) {
                const int k = i + 2;
                const float zr = xr[k], zi = xi[k], yr = xr[512 - k], yi = xi[512 - k];
                const float ar = zr + yr, ai = zi - yi;        // 2*A[k]
                const float br = zi + yi, bi = zr - yr;        // 2*|B[k]| components
                pw0[i] = 0.25f * (ar * ar + ai * ai);
                pw1[i] = 0.25f * (br * br + bi * bi);
            }
            wave_sync();
            // ---- mel + log: thread (fr, mbin), frames of its own wave
            const int frame = g * 8 + fr;
            if (frame < p.n_frames) {
                float acc = 0.f;
                const float* pw = (fr & 1) ? pw1 : pw0;
#pragma unroll
                for (int t = 0; t < 16; ++t) acc = fmaf(pw[mstart + t], taps[t], acc);
                float db = db10(acc);
                const bool masked = first && frame < 3;
                if (!masked) vmax = fmaxf(vmax, db);
                if (masked) db = INFINITY;                 // marker: becomes 1.0 below
                last_db = db;
                if ((n_groups > 1 || !p.streaming) && !p.max_only) p.out[((size_t)s * p.n_frames + frame) * 32 + mbin] = db;
            }
        }
        // ---- workgroup maximum over the whole call (the only point where the four waves meet)
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) vmax = fmaxf(vmax, __shfl_xor(vmax, o));
        if (lane == 0) s_red[it & 1][wave] = vmax;
        __syncthreads();
        vmax = fmaxf(fmaxf(s_red[it & 1][0], s_red[it & 1][1]), fmaxf(s_red[it & 1][2], s_red[it & 1][3]));
        if (p.max_only) {
            if (tid == 0) p.smax[s] = vmax;
        } else if (p.streaming) {
            const float floor_db = (p.floor_max ? p.floor_max[s] : vmax) - 80.0f;
            for (int g = 0; g < n_groups; ++g) {
                const int frame = g * 8 + fr;
                if (frame >= p.n_frames) break;
                float* o = p.out + ((size_t)s * p.n_frames + frame) * 32 + mbin;
                const float db = (n_groups > 1) ? *o : last_db;
                *o = (db == INFINITY) ? 1.0f : mel_units(db, floor_db);
            }
            // new 480-sample tail = last 480 samples of [tail ; pcm] (a stream that sits a masked step out keeps its tail: the rows
            // computed above from its stale samples are scratch nobody stores from)
            if (p.stream_on && !p.stream_on[s]) {
            } else if (p.n_samples >= 480) {
                for (int i = tid; i < 480; i += MEL_NT) p.tail[(size_t)s * 480 + i] = pcm[p.n_samples - 480 + i];
            } else {
                // shorter call: the tail shifts; read everything before anyone overwrites it
                int16_t keep[2];
                int n_keep = 0;
                for (int i = tid; i < 480; i += MEL_NT) {
                    const int src = p.n_samples - 480 + i;
                    keep[n_keep++] = (src >= 0) ? pcm[src] : p.tail[(size_t)s * 480 + p.n_samples + i];
                }
                __syncthreads();
                n_keep = 0;
                for (int i = tid; i < 480; i += MEL_NT) p.tail[(size_t)s * 480 + i] = keep[n_keep++];
            }
        } else if (tid == 0) {
            p.smax[s] = vmax;
        }
    }
}

__global__ void clamp_db_rows_kernel(float* x, int per_clip, const float* smax) {       // one floor per clip
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < per_clip) { float* q = x + (size_t)blockIdx.y * per_clip + i; *q = fmaxf(*q, smax[blockIdx.y] - 80.0f); }
}

// per-clip clamp + the host transform of utils.py:180,206 (x/10 + 2): what embed_clips feeds the embedding model
__global__ void clamp_transform_rows_kernel(float* x, int per_clip, const float* smax) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < per_clip) { float* q = x + (size_t)blockIdx.y * per_clip + i; *q = mel_units(*q, smax[blockIdx.y] - 80.0f); }
}

__global__ void clamp_db_kernel(float* x, size_t n, float floor_db) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) x[i] = fmaxf(x[i], floor_db);
}

// ------------------------------------------------------------------------------------------------
// wake-word heads (model.py:299-302; architecture train.py:56-83)
// ------------------------------------------------------------------------------------------------
struct NetDesc {
    int hidden, n_out, has_ln, final_act;   // final_act: 0 sigmoid, 1 relu+softmax
    int T;
    int head, role;                          // role 0 = primary net, 1 = gate verifier (hey_jarvis style)
    int out_col;                             // first score column of the owning head
    int hid_off;                             // column offset inside the group's hidden matrix (fast path)
    const float *w1, *b1, *ln1g, *ln1b, *w2, *b2, *ln2g, *ln2b, *w3, *b3;   // natural layouts (w2 .. ln2b: hidden block 0)
    int n_blocks;                            // hidden blocks (train.py:73); the MFMA head kernels take nets with exactly one
    const float* blocks;                     // all of them back to back, w[H][H] b[H] (g[H] be[H]) each: heads_generic_kernel
    const float* rnn;                        // model_type "rnn" (train.py:85-98; heads_rnn_kernel): the head's whole blob -- per layer, per
                                             // direction w [in + 64][256], b [256]; then w_out [128][n_out], b_out [n_out]; nullptr = an MLP net
    const float *w2pk;                       // MFMA-packed [hidden/16][hidden/4][64] (fast path)
};

struct HeadParams {
    const float* feat;      // ring [S][TR][96]  or external [B][T][96] when ext != 0
    int ext;
    int TR;
    const uint32_t* nfeat;
    const NetDesc* nets;    // device array
    int n_nets;
    int T;                  // common T of the group (fast path)
    int NH;                 // total hidden columns of the group (fast path)
    const float* w1pk;      // [NH/16][T*24][64]   (fast path)
    const float* b1cat;     // [NH]
    float* raw;             // [S][NL]
    int NL;
    int S;
    int accumulate_max;     // 1: raw = max(raw, new)  (multi-chunk calls, model.py:298)
    const uint8_t* stream_on;   // oww_step_masked: [S] 1 = the stream takes part in this step (its raw scores are stored); nullptr = all do
};

__device__ __forceinline__ const float* feat_row(const HeadParams& p, int s, int T, int t) {
    if (p.ext) return p.feat + ((size_t)s * T + t) * 96;
    const uint32_t slot = (p.nfeat[s] + (uint32_t)(2 * p.TR - T + 1 + t)) % (uint32_t)p.TR;
    return p.feat + ((size_t)s * p.TR + slot) * 96;
}

__device__ __forceinline__ void store_raw(const HeadParams& p, int s, int col, float v) {
    if (p.stream_on && !p.ext && !p.stream_on[s]) return;
    float* o = p.raw + (size_t)s * p.NL + col;
    *o = p.accumulate_max ? fmaxf(*o, v) : v;
}

// generic (any T / hidden <= 512 / n_out <= 8 / LN / softmax / 0..8 hidden blocks): ONE WAVE per SPW streams, the hidden units spread
// over the lanes (unit o = lane + 64 i).  Layer 1 walks the T x 96 features in order -- every hidden unit's sum is the k-ordered fmaf
// chain of the reference formula -- with the feature values broadcast from LDS (four k per read) and the weight row w1[k][.] read
// coalesced, once per SPW streams; the hidden blocks the same on the hidden vector in LDS; LayerNorm sums by wave reduction.
// Two shapes of one template: <4 streams, 4 waves> for small batches (calibration probes, single streams: more waves in flight) and
// <16, 2> from GH_BIG_STREAMS streams on -- every weight value read from L2 then feeds 16 streams' FMAs instead of 4 (the multiclass
// `timer` head, T = 34 x 128 hidden units, at 131,072 streams: 24.3 ms with four streams per wave).  Per-stream arithmetic is the
// same in both, so results do not depend on the shape.  LDS is dynamic: [WAVES][SPW] x (96 features + hs hidden + 8 outputs).
// The big shape is built with accumulator registers for 128 hidden units (HPL = 2: train.py's default width and the released
// multiclass models); with registers for 512 it spills inside the feature loop, so wider nets keep the small shape at any batch.
// (Rounds 1-4 ran one THREAD per stream with its hidden vectors in a global scratch buffer: 0.4 M dependent memory round trips per
// stream for `timer` -- 70 ms per launch on 32 streams, which oww_commit's calibration and self-test issue 66 times.)
constexpr int GH_HMAX = 512;         // oww_add_head refuses hidden > 512
constexpr int GH_HPL = GH_HMAX / 64; // hidden units per lane
constexpr int GH_BIG_STREAMS = 2048; // launches of at least this many streams take the <16, 2> shape
inline size_t gh_lds_bytes(int spw, int waves, int hs) { return (size_t)waves * spw * (96 + hs + 8) * sizeof(float); }

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    return v;
}

// LayerNorm (optional) + ReLU of the hidden vectors held as acc[stream][i] <-> unit lane + 64 i; then into LDS hv[stream][unit]
template <int SPW, int HPL>
__device__ __forceinline__ void gh_norm_relu_store(float (&acc)[SPW][HPL], int H, int has_ln, const float* __restrict__ g,
                                                   const float* __restrict__ b, float* hv /*[SPW][hs]*/, int hs, int lane) {
#pragma unroll
    for (int s = 0; s < SPW; ++s) {
        if (has_ln) {
            float sum = 0.f;
#pragma unroll
            for (int i = 0; i < HPL; ++i) if (lane + 64 * i < H) sum += acc[s][i];
            const float mu = wave_sum(sum) / (float)H;
            float var = 0.f;
#pragma unroll
            for (int i = 0; i < HPL; ++i) if (lane + 64 * i < H) { const float d = acc[s][i] - mu; var = fmaf(d, d, var); }
            const float rs = 1.0f / sqrtf(wave_sum(var) / (float)H + 1e-5f);
#pragma unroll
            for (int i = 0; i < HPL; ++i) if (lane + 64 * i < H) acc[s][i] = (acc[s][i] - mu) * rs * g[lane + 64 * i] + b[lane + 64 * i];
        }
#pragma unroll
        for (int i = 0; i < HPL; ++i) if (lane + 64 * i < H) hv[s * hs + lane + 64 * i] = fmaxf(acc[s][i], 0.f);
    }
}

template <int SPW, int WAVES, int HPL = GH_HPL>       // HPL = hidden units per lane the registers are sized for (hidden <= 64 HPL)
// (the big shape is held to two waves per SIMD: left alone the compiler unrolls the feature loop into 300-400 registers, one wave per SIMD)
__global__ __launch_bounds__(64 * WAVES, SPW >= 16 ? 2 : 1) void heads_generic_kernel(HeadParams p, int net_begin, int net_end, int hs) {
    typedef float f32x4g __attribute__((ext_vector_type(4)));
    static_assert(SPW <= 64, "one lane per stream finishes the output layer");
    extern __shared__ __attribute__((aligned(16))) float gh_lds[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int s0 = (blockIdx.x * WAVES + wave) * SPW;
    if (s0 >= p.S) return;                                   // (whole waves leave; no workgroup barrier below)
    float* xs = gh_lds + (size_t)wave * SPW * 96;                                    // [SPW][96] feature row t of the wave's streams
    float* hv = gh_lds + (size_t)WAVES * SPW * 96 + (size_t)wave * SPW * hs;         // [SPW][hs] hidden vectors
    float* sz = gh_lds + (size_t)WAVES * SPW * (96 + hs) + (size_t)wave * SPW * 8;   // [SPW][8]  output-layer sums
    int sid[SPW];
#pragma unroll
    for (int s = 0; s < SPW; ++s) sid[s] = min(s0 + s, p.S - 1);
    for (int ni = net_begin; ni < net_end; ++ni) {
        const NetDesc& n = p.nets[ni];
        if (n.role != 0 || n.rnn != nullptr) continue;       // (recurrent heads: heads_rnn_kernel)
        float result[8];                                     // lane s < SPW: the scores of stream s0 + s
        float gate_score = 0.f;
#pragma unroll
        for (int o = 0; o < 8; ++o) result[o] = 0.f;
        for (int pass = 0; pass < 2; ++pass) {
            if (pass == 1) {
                if (ni + 1 >= net_end) break;
                if (p.nets[ni + 1].role != 1 || p.nets[ni + 1].head != n.head) break;
            }
            const NetDesc& m = p.nets[ni + pass];
            const int H = m.hidden, T = m.T, O = m.n_out;
            float acc[SPW][HPL];
            // ---- layer 1
#pragma unroll
            for (int i = 0; i < HPL; ++i) {
                const float bb = lane + 64 * i < H ? m.b1[lane + 64 * i] : 0.f;
#pragma unroll
                for (int s = 0; s < SPW; ++s) acc[s][i] = bb;
            }
            for (int t = 0; t < T; ++t) {
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");           // (wave-local LDS tile: the previous row's readers are done)
#pragma unroll
                for (int s = 0; s < SPW; ++s) {
                    const float* row = feat_row(p, sid[s], T, t);
                    xs[s * 96 + lane] = row[lane];
                    if (lane < 32) xs[s * 96 + 64 + lane] = row[64 + lane];
                }
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                // four feature columns per trip; the NEXT trip's weight rows are requested before this trip's FMAs (the loop is kept
                // rolled: unrolled, its 16 feature vectors per trip push the kernel past the register file)
                const float* w = m.w1 + (size_t)t * 96 * H;
                float wcur[4][HPL], wnxt[4][HPL];
#pragma unroll
                for (int cc = 0; cc < 4; ++cc)
#pragma unroll
                    for (int i = 0; i < HPL; ++i) wcur[cc][i] = lane + 64 * i < H ? w[(size_t)cc * H + lane + 64 * i] : 0.f;
#pragma unroll 1
                for (int c = 0; c < 96; c += 4) {
                    w += 4 * (size_t)H;
#pragma unroll
                    for (int cc = 0; cc < 4; ++cc)
#pragma unroll
                        for (int i = 0; i < HPL; ++i) wnxt[cc][i] = (c + 4 < 96 && lane + 64 * i < H) ? w[(size_t)cc * H + lane + 64 * i] : 0.f;
                    f32x4g xv[SPW];
#pragma unroll
                    for (int s = 0; s < SPW; ++s) xv[s] = *reinterpret_cast<const f32x4g*>(xs + s * 96 + c);
#pragma unroll
                    for (int cc = 0; cc < 4; ++cc)
#pragma unroll
                        for (int i = 0; i < HPL; ++i)
                            if (lane + 64 * i < H) {
#pragma unroll
                                for (int s = 0; s < SPW; ++s) acc[s][i] = fmaf(xv[s][cc], wcur[cc][i], acc[s][i]);
                            }
#pragma unroll
                    for (int cc = 0; cc < 4; ++cc)
#pragma unroll
                        for (int i = 0; i < HPL; ++i) wcur[cc][i] = wnxt[cc][i];
                }
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            gh_norm_relu_store<SPW, HPL>(acc, H, m.has_ln, m.ln1g, m.ln1b, hv, hs, lane);
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            // ---- hidden blocks (train.py:56-65 FCNBlock: Linear -> LayerNorm -> ReLU; one in the released models)
            for (int blk = 0; blk < m.n_blocks; ++blk) {
                const float* w2 = m.blocks + (size_t)blk * ((size_t)H * H + H + (m.has_ln ? 2 * H : 0));
                const float* b2 = w2 + (size_t)H * H;
#pragma unroll
                for (int i = 0; i < HPL; ++i) {
                    const float bb = lane + 64 * i < H ? b2[lane + 64 * i] : 0.f;
#pragma unroll
                    for (int s = 0; s < SPW; ++s) acc[s][i] = bb;
                }
#pragma unroll 1
                for (int k = 0; k < H; ++k) {
                    float xv[SPW];
#pragma unroll
                    for (int s = 0; s < SPW; ++s) xv[s] = hv[s * hs + k];
                    const float* w = w2 + (size_t)k * H;
#pragma unroll
                    for (int i = 0; i < HPL; ++i)
                        if (lane + 64 * i < H) {
                            const float wv = w[lane + 64 * i];
#pragma unroll
                            for (int s = 0; s < SPW; ++s) acc[s][i] = fmaf(xv[s], wv, acc[s][i]);
                        }
                }
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");           // every lane has read the previous layer's vector
                gh_norm_relu_store<SPW, HPL>(acc, H, m.has_ln, b2 + H, b2 + 2 * H, hv, hs, lane);
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            }
            // ---- output layer: lane o < n_out, the k-ordered chain of the formula; then the final activation, one lane per stream
            if (lane < O) {
#pragma unroll
                for (int s = 0; s < SPW; ++s) {
                    float a = m.b3[lane];
                    for (int i = 0; i < H; ++i) a = fmaf(hv[s * hs + i], m.w3[i * O + lane], a);
                    sz[s * 8 + lane] = a;
                }
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            if (lane < SPW) {
                float z[8];
#pragma unroll
                for (int o = 0; o < 8; ++o) z[o] = o < O ? sz[lane * 8 + o] : 0.f;
                if (m.final_act == 1) {
                    float mx = -INFINITY, sum = 0.f;
#pragma unroll
                    for (int o = 0; o < 8; ++o) if (o < O) { z[o] = fmaxf(z[o], 0.f); mx = fmaxf(mx, z[o]); }
#pragma unroll
                    for (int o = 0; o < 8; ++o) if (o < O) { z[o] = expf(z[o] - mx); sum += z[o]; }
#pragma unroll
                    for (int o = 0; o < 8; ++o) if (o < O) z[o] /= sum;
                } else {
#pragma unroll
                    for (int o = 0; o < 8; ++o) if (o < O) z[o] = 1.0f / (1.0f + expf(-z[o]));
                }
                if (pass == 0) {
#pragma unroll
                    for (int o = 0; o < 8; ++o) result[o] = z[o];
                    gate_score = z[0];
                } else if (gate_score > 0.5f) {
#pragma unroll
                    for (int o = 0; o < 8; ++o) result[o] = z[o];
                }
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        }
        if (lane < SPW && s0 + lane < p.S) {
#pragma unroll
            for (int o = 0; o < 8; ++o)
                if (o < n.n_out) store_raw(p, s0 + lane, n.out_col + o, result[o]);
        }
    }
}

// ------------------------------------------------------------------------------------------------
// The reference's other model_type, "rnn" (train.py:85-98): x [T, 96] -> 2-layer bidirectional LSTM(64) -> Linear(128, n_out) on the
// LAST time step's output -> Sigmoid (one class) | ReLU + softmax (train.py:152-165).  No released model is recurrent; this kernel is the
// plain form for any kernel family: ONE WAVE per SPW streams, lane = hidden unit (its four gates, its cell and hidden state in
// registers), the step's input vector and the previous hidden vector broadcast from the wave's LDS, weight rows read coalesced from L2
// (torch's gate order i | f | g | o in 64-column blocks).  Layer 0 runs both directions over all T rows; layer 1 runs forward over all
// of them and backward for ONE step only -- out[:, -1] of the reverse direction is its first step, from a zero state.
// LDS (dynamic): [SPW] x (T x 96 features + T x 128 layer-0 outputs + 64 hidden + 128 last + 8 outputs).
// ------------------------------------------------------------------------------------------------
constexpr int RNN_H = 64, RNN_SPW = 2, RNN_TMAX = 64;
inline size_t rnn_lds_bytes(int T) { return (size_t)RNN_SPW * ((size_t)T * (96 + 128) + 64 + 128 + 8) * sizeof(float); }

__device__ __forceinline__ float sigm(float v) { return 1.0f / (1.0f + expf(-v)); }

// one LSTM step of one direction for the wave's streams: x = xin[s] (n_in floats, LDS), hp[s] (64 floats, LDS; skipped when `first`)
template <int SPW>
__device__ __forceinline__ void rnn_step(const float* __restrict__ w, const float* __restrict__ b, int n_in, const float* const (&xin)[SPW],
                                         float* hp, bool first, float (&c)[SPW], float (&hcur)[SPW], int lane) {
    float acc[SPW][4];
#pragma unroll
    for (int g = 0; g < 4; ++g) {
        const float bb = b[g * RNN_H + lane];
#pragma unroll
        for (int s = 0; s < SPW; ++s) acc[s][g] = bb;
    }
#pragma unroll 4
    for (int k = 0; k < n_in; ++k) {
        float wv[4];
#pragma unroll
        for (int g = 0; g < 4; ++g) wv[g] = w[(size_t)k * 256 + g * RNN_H + lane];
#pragma unroll
        for (int s = 0; s < SPW; ++s) {
            const float xv = xin[s][k];
#pragma unroll
            for (int g = 0; g < 4; ++g) acc[s][g] = fmaf(xv, wv[g], acc[s][g]);
        }
    }
    if (!first) {
        const float* wh = w + (size_t)n_in * 256;
#pragma unroll 4
        for (int k = 0; k < RNN_H; ++k) {
            float wv[4];
#pragma unroll
            for (int g = 0; g < 4; ++g) wv[g] = wh[(size_t)k * 256 + g * RNN_H + lane];
#pragma unroll
            for (int s = 0; s < SPW; ++s) {
                const float hv = hp[s * RNN_H + k];
#pragma unroll
                for (int g = 0; g < 4; ++g) acc[s][g] = fmaf(hv, wv[g], acc[s][g]);
            }
        }
    }
#pragma unroll
    for (int s = 0; s < SPW; ++s) {
        const float i = sigm(acc[s][0]), f = sigm(acc[s][1]), gg = tanhf(acc[s][2]), o = sigm(acc[s][3]);
        c[s] = first ? i * gg : fmaf(f, c[s], i * gg);
        hcur[s] = o * tanhf(c[s]);
    }
}

template <int SPW>
__global__ __launch_bounds__(64) void heads_rnn_kernel(HeadParams p, int ni) {
    extern __shared__ __attribute__((aligned(16))) float rnn_lds[];
    const NetDesc& n = p.nets[ni];
    const int lane = threadIdx.x, T = n.T, O = n.n_out;
    const int s0 = blockIdx.x * SPW;
    float* xs = rnn_lds;                                 // [SPW][T][96]
    float* y0 = xs + (size_t)SPW * T * 96;               // [SPW][T][128]  layer-0 outputs (forward | backward)
    float* hp = y0 + (size_t)SPW * T * 128;              // [SPW][64]      previous hidden vector of the running direction
    float* last = hp + SPW * RNN_H;                      // [SPW][128]     out[:, -1] of layer 1
    float* sz = last + SPW * 128;                        // [SPW][8]
    int sid[SPW];
#pragma unroll
    for (int s = 0; s < SPW; ++s) sid[s] = min(s0 + s, p.S - 1);
    for (int t = 0; t < T; ++t)
#pragma unroll
        for (int s = 0; s < SPW; ++s) {
            const float* row = feat_row(p, sid[s], T, t);
            xs[((size_t)s * T + t) * 96 + lane] = row[lane];
            if (lane < 32) xs[((size_t)s * T + t) * 96 + 64 + lane] = row[64 + lane];
        }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    const float* q = n.rnn;
    for (int layer = 0; layer < 2; ++layer) {
        const int n_in = layer == 0 ? 96 : 128;
        const float* src = layer == 0 ? xs : y0;
        for (int dir = 0; dir < 2; ++dir) {
            const float* w = q;
            const float* b = q + (size_t)(n_in + RNN_H) * 256;
            q = b + 256;
            float c[SPW], hc[SPW];
#pragma unroll
            for (int s = 0; s < SPW; ++s) { c[s] = 0.f; hc[s] = 0.f; }
            const int n_steps = (layer == 1 && dir == 1) ? 1 : T;          // out[:, -1] of the reverse direction = its first step
            for (int it = 0; it < n_steps; ++it) {
                const int t = dir == 0 ? it : T - 1 - it;
                const float* xin[SPW];
#pragma unroll
                for (int s = 0; s < SPW; ++s) xin[s] = src + ((size_t)s * T + t) * n_in;
                rnn_step<SPW>(w, b, n_in, xin, hp, it == 0, c, hc, lane);
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");          // every lane has read the previous hidden vector
#pragma unroll
                for (int s = 0; s < SPW; ++s) {
                    hp[s * RNN_H + lane] = hc[s];
                    if (layer == 0) y0[((size_t)s * T + t) * 128 + dir * RNN_H + lane] = hc[s];
                }
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            }
#pragma unroll
            for (int s = 0; s < SPW; ++s) if (layer == 1) last[s * 128 + dir * RNN_H + lane] = hc[s];
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        }
    }
    const float* w_out = q;
    const float* b_out = q + (size_t)128 * O;
    if (lane < O) {
#pragma unroll
        for (int s = 0; s < SPW; ++s) {
            float a = b_out[lane];
            for (int k = 0; k < 128; ++k) a = fmaf(last[s * 128 + k], w_out[k * O + lane], a);
            sz[s * 8 + lane] = a;
        }
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    if (lane < SPW && s0 + lane < p.S) {
        float z[8];
#pragma unroll
        for (int o = 0; o < 8; ++o) z[o] = o < O ? sz[lane * 8 + o] : 0.f;
        if (n.final_act == 1) {
            float mx = -INFINITY, sum = 0.f;
#pragma unroll
            for (int o = 0; o < 8; ++o) if (o < O) { z[o] = fmaxf(z[o], 0.f); mx = fmaxf(mx, z[o]); }
#pragma unroll
            for (int o = 0; o < 8; ++o) if (o < O) { z[o] = expf(z[o] - mx); sum += z[o]; }
#pragma unroll
            for (int o = 0; o < 8; ++o) if (o < O) z[o] /= sum;
        } else {
#pragma unroll
            for (int o = 0; o < 8; ++o) if (o < O) z[o] = sigm(z[o]);
        }
#pragma unroll
        for (int o = 0; o < 8; ++o)
            if (o < O) store_raw(p, s0 + lane, n.out_col + o, z[o]);
    }
}

// fast path: every net of the group has hidden == 64, n_out == 1, sigmoid, the same T.
constexpr int HD_SB = 32;          // streams per workgroup
constexpr int HD_NW = 8;
constexpr int HD_NT = HD_NW * 64;
constexpr int HD_MAXNETS = 8;

__global__ __launch_bounds__(HD_NT) void heads64_kernel(HeadParams p) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int NH = p.NH, HS = NH + 4;             // hidden row stride in LDS
    float* X = smem;                               // [SB][100]
    float* H1 = X + HD_SB * 100;                   // [SB][HS]
    float* H2 = H1 + HD_SB * HS;                   // [SB][HS]
    float* SC = H2 + HD_SB * HS;                   // [SB][MAXNETS]
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int pl = lane & 15, j = lane >> 4;
    const int s0 = blockIdx.x * HD_SB;
    const int NCT = NH / 16;
    const int T = p.T;
    const int KST = T * 24;                        // k-steps of the first layer

    // ---- layer 1: D[hidden][stream] over K = T*96, staged one ring row (96 k) at a time
    f32x4 acc[4][2];
#pragma unroll
    for (int a = 0; a < 4; ++a) { acc[a][0] = f32x4{0.f, 0.f, 0.f, 0.f}; acc[a][1] = f32x4{0.f, 0.f, 0.f, 0.f}; }
    for (int t = 0; t < T; ++t) {
        __syncthreads();
        for (int i = tid; i < HD_SB * 24; i += HD_NT) {
            const int b = i / 24, c4 = i % 24;
            int s = s0 + b;
            if (s >= p.S) s = p.S - 1;
            *reinterpret_cast<f32x4*>(X + b * 100 + c4 * 4) = *reinterpret_cast<const f32x4*>(feat_row(p, s, T, t) + c4 * 4);
        }
        __syncthreads();
#pragma unroll
        for (int a = 0; a < 4; ++a) {
            const int ct = wave + a * HD_NW;
            if (ct < NCT) {
                const float* wp = p.w1pk + ((size_t)ct * KST + t * 24) * 64 + lane;
#pragma unroll
                for (int cb = 0; cb < 96; cb += 8) {
                    const float2 x0 = *reinterpret_cast<const float2*>(X + pl * 100 + cb + 2 * j);
                    const float2 x1 = *reinterpret_cast<const float2*>(X + (16 + pl) * 100 + cb + 2 * j);
                    const float w0 = wp[(cb / 4) * 64], w1 = wp[(cb / 4 + 1) * 64];
                    acc[a][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(w0, x0.x, acc[a][0], 0, 0, 0);
                    acc[a][1] = __builtin_amdgcn_mfma_f32_16x16x4f32(w0, x1.x, acc[a][1], 0, 0, 0);
                    acc[a][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(w1, x0.y, acc[a][0], 0, 0, 0);
                    acc[a][1] = __builtin_amdgcn_mfma_f32_16x16x4f32(w1, x1.y, acc[a][1], 0, 0, 0);
                }
            }
        }
    }
#pragma unroll
    for (int a = 0; a < 4; ++a) {
        const int ct = wave + a * HD_NW;
        if (ct < NCT) {
            const int c0 = ct * 16 + j * 4;
            const f32x4 bias = *reinterpret_cast<const f32x4*>(p.b1cat + c0);
            *reinterpret_cast<f32x4*>(H1 + pl * HS + c0) = acc[a][0] + bias;
            *reinterpret_cast<f32x4*>(H1 + (16 + pl) * HS + c0) = acc[a][1] + bias;
        }
    }
    __syncthreads();
    // ---- LayerNorm + ReLU, one thread per (stream, net)
    const int n_nets = p.n_nets;
    for (int layer = 0; layer < 2; ++layer) {
        float* H = layer ? H2 : H1;
        if (tid < HD_SB * n_nets) {
            const int b = tid % HD_SB, ni = tid / HD_SB;
            const NetDesc& n = p.nets[ni];
            float* h = H + b * HS + n.hid_off;
            if (n.has_ln) {
                const float* g = layer ? n.ln2g : n.ln1g;
                const float* be = layer ? n.ln2b : n.ln1b;
                float mu = 0.f;
                for (int o = 0; o < 64; ++o) mu += h[o];
                mu /= 64.f;
                float var = 0.f;
                for (int o = 0; o < 64; ++o) { const float d = h[o] - mu; var = fmaf(d, d, var); }
                var /= 64.f;
                const float rs = 1.0f / sqrtf(var + 1e-5f);
                for (int o = 0; o < 64; ++o) h[o] = fmaxf((h[o] - mu) * rs * g[o] + be[o], 0.f);
            } else {
                for (int o = 0; o < 64; ++o) h[o] = fmaxf(h[o], 0.f);
            }
            if (layer == 1) {
                float z = n.b3[0];
                for (int i = 0; i < 64; ++i) z = fmaf(h[i], n.w3[i], z);
                SC[b * HD_MAXNETS + ni] = 1.0f / (1.0f + expf(-z));
            }
        }
        __syncthreads();
        if (layer == 0) {
            // ---- layer 2 (64x64 per net) on MFMA: combos (net, cout tile) spread over the waves
            for (int combo = wave; combo < n_nets * 4; combo += HD_NW) {
                const int ni = combo >> 2, ct = combo & 3;
                const NetDesc& n = p.nets[ni];
                const float* wp = n.w2pk + (size_t)ct * 16 * 64 + lane;
                const float* hb = H1 + n.hid_off;
                f32x4 a0 = {0.f, 0.f, 0.f, 0.f}, a1 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int cb = 0; cb < 64; cb += 8) {
                    const float2 x0 = *reinterpret_cast<const float2*>(hb + pl * HS + cb + 2 * j);
                    const float2 x1 = *reinterpret_cast<const float2*>(hb + (16 + pl) * HS + cb + 2 * j);
                    const float w0 = wp[(cb / 4) * 64], w1 = wp[(cb / 4 + 1) * 64];
                    a0 = __builtin_amdgcn_mfma_f32_16x16x4f32(w0, x0.x, a0, 0, 0, 0);
                    a1 = __builtin_amdgcn_mfma_f32_16x16x4f32(w0, x1.x, a1, 0, 0, 0);
                    a0 = __builtin_amdgcn_mfma_f32_16x16x4f32(w1, x0.y, a0, 0, 0, 0);
                    a1 = __builtin_amdgcn_mfma_f32_16x16x4f32(w1, x1.y, a1, 0, 0, 0);
                }
                const int c0 = ct * 16 + j * 4;
                const f32x4 bias = *reinterpret_cast<const f32x4*>(n.b2 + c0);
                *reinterpret_cast<f32x4*>(H2 + pl * HS + n.hid_off + c0) = a0 + bias;
                *reinterpret_cast<f32x4*>(H2 + (16 + pl) * HS + n.hid_off + c0) = a1 + bias;
            }
            __syncthreads();
        }
    }
    // ---- gating + store
    if (tid < HD_SB * n_nets) {
        const int b = tid % HD_SB, ni = tid / HD_SB;
        const NetDesc& n = p.nets[ni];
        const int s = s0 + b;
        if (n.role == 0 && s < p.S) {
            float sc = SC[b * HD_MAXNETS + ni];
            if (ni + 1 < n_nets && p.nets[ni + 1].role == 1 && p.nets[ni + 1].head == n.head && sc > 0.5f)
                sc = SC[b * HD_MAXNETS + ni + 1];
            store_raw(p, s, n.out_col, sc);
        }
    }
}

// ------------------------------------------------------------------------------------------------
// post-processing (model.py:330-363): one thread per stream
// ------------------------------------------------------------------------------------------------
struct PostParams {
    const float* raw;        // [S][NL]
    float* scores;           // [S][NL]
    float* ring;             // [S][NL][30]
    uint32_t* npred;         // [S]
    const int* patience;     // [NL]
    const float* threshold;  // [NL]  (NaN = no threshold for this label)
    int debounce_frames;
    int NL, S;
    // VAD gate (model.py:366-381): per-stream ring of the caller-supplied voice-activity scores, 8 deep
    const float* vad_ring;   // [S][8]
    const uint32_t* n_vad;   // [S] scores pushed so far
    float vad_threshold;     // <= 0: gate off
    const uint8_t* stream_on;    // oww_step_masked: [S] 1 = the stream takes part in this step; nullptr = all do
};

__global__ void postproc_kernel(PostParams p) {
    const int s = blockIdx.x * blockDim.x + threadIdx.x;
    if (s >= p.S) return;
    if (p.stream_on && !p.stream_on[s]) return;
    const uint32_t cnt = p.npred[s];
    const int have = cnt < 30u ? (int)cnt : 30;
    for (int l = 0; l < p.NL; ++l) {
        float sc = p.raw[(size_t)s * p.NL + l];
        float* ring = p.ring + ((size_t)s * p.NL + l) * 30;
        if (cnt < 5u) sc = 0.0f;                                            // model.py:331-333
        if (sc != 0.0f) {
            const int pat = p.patience[l];
            const float thr = p.threshold[l];
            if (pat > 0) {                                                  // model.py:349-352
                const int look = pat < have ? pat : have;
                int n_ok = 0;
                for (int i = 1; i <= look; ++i) n_ok += ring[(cnt - i) % 30u] >= thr ? 1 : 0;
                if (n_ok < pat) sc = 0.0f;
            } else if (p.debounce_frames > 0 && thr == thr) {               // model.py:353-359
                const int look = p.debounce_frames < have ? p.debounce_frames : have;
                int n_hit = 0;
                for (int i = 1; i <= look; ++i) n_hit += ring[(cnt - i) % 30u] >= thr ? 1 : 0;
                if (sc >= thr && n_hit > 0) sc = 0.0f;
            }
        }
        ring[cnt % 30u] = sc;                                               // model.py:362-363
        p.scores[(size_t)s * p.NL + l] = sc;
    }
    p.npred[s] = cnt + 1u;
    if (p.vad_threshold > 0.0f) {
        // model.py:375-381: the score ring above keeps the ungated values; the returned scores of this step are zeroed when the
        // largest VAD score of ring[-7:-4] (entries L-7 .. L-5 of the L pushed so far; none while L < 5) is below the threshold
        const uint32_t L = p.n_vad[s];
        float vmax = 0.0f;
        if (L >= 5u) {
            vmax = -INFINITY;
            for (uint32_t i = (L >= 7u ? L - 7u : 0u); i + 5u <= L; ++i) vmax = fmaxf(vmax, p.vad_ring[(size_t)s * 8 + (i & 7u)]);
        }
        if (vmax < p.vad_threshold)
            for (int l = 0; l < p.NL; ++l) p.scores[(size_t)s * p.NL + l] = 0.0f;
    }
}

__global__ void push_vad_kernel(float* ring, uint32_t* n_vad, const float* scores, int S) {
    const int s = blockIdx.x * blockDim.x + threadIdx.x;
    if (s >= S) return;
    const uint32_t L = n_vad[s];
    ring[(size_t)s * 8 + (L & 7u)] = scores[s];
    n_vad[s] = L + 1u;
}

// ------------------------------------------------------------------------------------------------
// custom verifier on the device (model.py:320-328; custom_verifier_model.py:95-113): the reference re-scores a label whose
// base score reaches custom_verifier_threshold with a pickled scikit-learn pipeline -- flatten -> StandardScaler ->
// LogisticRegression on the last T feature rows.  Scaler and regression fold into ONE affine map, so the device form is a
// dot product of the stream's T x 96 ring rows with w' = coef / scale plus b' = intercept - sum(coef * mean / scale), then a
// sigmoid (= predict_proba(...)[0][-1]).  One wave per stream; runs between the heads and the post-processing.
// ------------------------------------------------------------------------------------------------
struct VerifierParams {
    float* raw;              // [S][NL] head outputs of this step (re-scored in place)
    const float* feat;       // feature ring [S][TR][96]
    const uint32_t* nfeat;   // [S] (not yet advanced for this step)
    const float* w;          // [NL][wstride]
    const float* bias;       // [NL]
    const float* thr;        // [NL] custom_verifier_threshold
    const int* T;            // [NL] feature rows of the label's model; 0 = no verifier for this label
    int wstride, NL, TR, S;
    const uint8_t* stream_on;    // oww_step_masked
};

__global__ __launch_bounds__(256) void verifier_kernel(VerifierParams p) {
    const int lane = threadIdx.x & 63, s = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (s >= p.S) return;
    if (p.stream_on && !p.stream_on[s]) return;
    for (int l = 0; l < p.NL; ++l) {
        const int T = p.T[l];
        if (T <= 0) continue;
        const float base = p.raw[(size_t)s * p.NL + l];
        if (!(base >= p.thr[l])) continue;                                   // model.py:322
        const uint32_t slot0 = p.nfeat[s] + (uint32_t)(2 * p.TR - T + 1);    // oldest of the last T rows (cf. heads_generic_kernel)
        float acc = 0.f;
        for (int i = lane; i < T * 96; i += 64) {
            const uint32_t slot = (slot0 + (uint32_t)(i / 96)) % (uint32_t)p.TR;
            acc = fmaf(p.feat[((size_t)s * p.TR + slot) * 96 + i % 96], p.w[(size_t)l * p.wstride + i], acc);
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) acc += __shfl_xor(acc, o);
        if (lane == 0) p.raw[(size_t)s * p.NL + l] = 1.0f / (1.0f + expf(-(acc + p.bias[l])));
    }
}

// polyphase FIR rate conversion of every stream's message to 16 kHz (oww_resample; filter design: openwakeword_amd/resample.py).
//   out[s][j] = sat_int16(rint(sum_k taps[(j p) % q][k] * in[s][(j p) / q + k - half + 1])),   zero outside the message.
// A workgroup produces `opb` consecutive outputs of one stream (up to a whole 1280-sample chunk): the input span they touch is staged in LDS once, converted to float
// (coalesced int16 reads), and so is the filter bank when it fits (rows padded to a multiple of 4 taps so that a lane reads 4 taps with
// one ds_read_b128; with q = 1 -- 48, 32, 96 kHz -- all lanes read the same row: a broadcast); one thread per output, k-ordered fmaf
// chain (the order the host restatement documents).
constexpr int RS_NT = 256;
struct ResampleParams {
    const int16_t* in; int16_t* out; const float* taps;   // taps: [q][ntp] (ntp = n_taps rounded up to 4, zero padded)
    int n_in, n_out, p, q, n_taps, ntp, S, span, taps_in_lds;
    int opb;                 // outputs per workgroup (a multiple of RS_NT: a whole 1280-sample chunk when the input span fits the LDS)
};
__global__ __launch_bounds__(RS_NT) void resample_kernel(ResampleParams a) {
    extern __shared__ __attribute__((aligned(16))) float rs_lds[];
    float* xs = rs_lds;                                  // [span]
    float* ts = rs_lds + (a.span + 3) / 4 * 4;           // [q][ntp] when taps_in_lds
    const int s = blockIdx.y, j0 = blockIdx.x * a.opb, t = threadIdx.x;
    const int half = a.n_taps / 2;
    const int base = (int)(((long long)j0 * a.p) / a.q) - (half - 1);       // input index of xs[0]
    const int16_t* x = a.in + (size_t)s * a.n_in;
    for (int i = t; i < a.span; i += RS_NT) {
        const int g = base + i;
        xs[i] = (g >= 0 && g < a.n_in) ? (float)x[g] : 0.f;
    }
    if (a.taps_in_lds)
        for (int i = t; i < a.q * a.ntp; i += RS_NT) ts[i] = a.taps[i];
    __syncthreads();
    for (int j = j0 + t; j < min(j0 + a.opb, a.n_out); j += RS_NT) {
        const long long jp = (long long)j * a.p;
        const int ph = (int)(jp % a.q);
        const float* xw = xs + ((int)(jp / a.q) - (half - 1) - base);
        const float* w = (a.taps_in_lds ? ts : a.taps) + (size_t)ph * a.ntp;
        float acc = 0.f;
        for (int k = 0; k < a.ntp; k += 4) {
            const float4 w4 = *reinterpret_cast<const float4*>(w + k);
            acc = fmaf(w4.x, xw[k], acc);
            acc = fmaf(w4.y, xw[k + 1], acc);
            acc = fmaf(w4.z, xw[k + 2], acc);
            acc = fmaf(w4.w, xw[k + 3], acc);
        }
        a.out[(size_t)s * a.n_out + j] = (int16_t)fminf(fmaxf(rintf(acc), -32768.f), 32767.f);
    }
}

// forget the VAD score history of the listed streams (ids == nullptr: streams [0, n)); oww_reset itself leaves it alone (model.py:226-230)
__global__ void vad_ring_reset_kernel(float* ring, uint32_t* n_vad, const int* ids, int n) {
    const int k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= n) return;
    const int s = ids ? ids[k] : k;
    for (int i = 0; i < 8; ++i) ring[(size_t)s * 8 + i] = 0.f;
    n_vad[s] = 0u;
}

__global__ void advance_kernel(uint32_t* nfeat, int S, const uint8_t* stream_on) {
    const int s = blockIdx.x * blockDim.x + threadIdx.x;
    if (s < S && (stream_on == nullptr || stream_on[s])) nfeat[s] += 1u;
}

__global__ void fill_kernel(float* x, size_t n, float v) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) x[i] = v;
}

// largest |x| of every layer's debug dump (oww_commit's calibration of the f16-split family): grid (20 layers, streams), the
// result as float bit patterns (non-negative floats order like their bit patterns; a NaN ends up above every finite value)
__global__ void layer_absmax_kernel(const float* __restrict__ dbg, size_t stride, const int* __restrict__ off /*[21]*/, unsigned* __restrict__ out /*[20]*/) {
    const int l = blockIdx.x;
    const float* x = dbg + (size_t)blockIdx.y * stride;
    unsigned m = 0u;
    for (int i = off[l] + threadIdx.x; i < off[l + 1]; i += blockDim.x) m = max(m, __float_as_uint(fabsf(x[i])));
    for (int o = 32; o > 0; o >>= 1) m = max(m, (unsigned)__shfl_xor((int)m, o));
    if ((threadIdx.x & 63) == 0 && m) atomicMax(out + l, m);
}

// reset: copy per-layer templates into the listed streams' state
struct ResetParams {
    const int* ids;          // device list or null (all)
    int n;
    int n_arrays;
    float* dst[16];
    const float* tmpl[16];
    int len[16];             // floats per stream
    int spg[16];             // streams per group block (register-dump layouts interleave the streams of a wave; 1 = plain)
    int fpos[16];            // positions per stream inside the 16 positions of a tile (when spg > 1)
    int interleaved;         // position order of such a tile: 0 = fpos * stream + mel (fp32 family), 1 = spg * mel + stream (f16-split family)
    int16_t* tail;
    uint32_t* nfeat;
    uint32_t* npred;
    float* ring;  int ring_len;      // score ring floats per stream
    float* feat;  int feat_len;      // feature ring floats per stream
    const float* feat_init;          // [feat_len] or null (zeros)
};

__global__ void reset_kernel(ResetParams p) {
    const int k = blockIdx.x;
    const int s = p.ids ? p.ids[k] : k;
    for (int a = 0; a < p.n_arrays; ++a) {
        if (p.spg[a] <= 1) {
            for (int i = threadIdx.x; i < p.len[a]; i += blockDim.x) p.dst[a][(size_t)s * p.len[a] + i] = p.tmpl[a][i];
        } else {
            // block of spg streams in register-dump order [..][64 lanes]: stream sp owns the lanes whose position maps to it
            const int bl = p.len[a] * p.spg[a], g = s / p.spg[a], sp = s % p.spg[a];
            for (int i = threadIdx.x; i < bl; i += blockDim.x)
                if ((p.interleaved ? (i & 15) % p.spg[a] : (i & 15) / p.fpos[a]) == sp) p.dst[a][(size_t)g * bl + i] = p.tmpl[a][i];
        }
    }
    for (int i = threadIdx.x; i < 480; i += blockDim.x) p.tail[(size_t)s * 480 + i] = 0;
    for (int i = threadIdx.x; i < p.ring_len; i += blockDim.x) p.ring[(size_t)s * p.ring_len + i] = 0.f;
    for (int i = threadIdx.x; i < p.feat_len; i += blockDim.x)
        p.feat[(size_t)s * p.feat_len + i] = p.feat_init ? p.feat_init[i] : 0.f;
    if (threadIdx.x == 0) { p.nfeat[s] = 0u; p.npred[s] = 0u; }
}

}  // namespace owk
