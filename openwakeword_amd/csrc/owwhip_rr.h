// owwhip_rr.h -- register-resident form of the incremental embedding CNN for gfx950 (included by owwhip.hip).
//
// Idea.  v_mfma_f32_16x16x4_f32 computes D[16 cout][16 pos] += A[16 cout][4 k] * B[4 k][16 pos] with the lane maps
//     A: lane (i = l&15, j = l>>4) holds A[i][k=j]          B: lane (p = l&15, j) holds B[k=j][p]
//     D: lane (p, j), register e of 4 holds D[cout = 4j+e][p]
// The k index of a convolution is (tap, input channel) and its order is free.  If the k-step "(ct, e)" is defined
// to carry the four input channels {16ct + 4j + e : j = 0..3}, then the B operand of that k-step is *exactly*
// register e of the previous layer's output tile ct -- the output registers of one layer are the input operands of
// the next, with no data movement.  A wave therefore owns whole position tiles (16 positions x all channels) and
// carries them through the four convolutions of a stage in registers:
//   * 3x1 (time) taps address other row tiles held by the same wave: a different register, same lane;
//   * 1x3 (mel) taps are computed as three per-tap accumulators on the UNshifted input and combined in the epilogue
//     with two DPP row shifts of the 4 accumulator registers (zero fill = the zero padding of the mel axis);
//   * weights stream from L2 in MFMA operand order (one 16-byte load per lane = four k-steps), each loaded register
//     feeding NT MFMAs; stage A keeps all of its weights in registers for the whole launch;
//   * per-stream conv history and the arrays handed from stage to stage are stored in register-dump order
//     [group][tile][register][64 lanes], i.e. every load/store is one fully coalesced 256-byte row.
// No LDS, no barrier: every wave is independent, the matrix pipe sees a dense stream of independent MFMAs.
//
// Tile geometry per stage (F = mel positions per row, R = new rows per step, C channels):
//   A: F=32 R=8 C=24   one stream per wave, a row = two tiles (halves), rows processed one after the other
//   B: F=16 R=4 C=48   tile = one row of one stream,            NT = 4 tiles, 1 stream  per wave
//   C: F=8  R=4 C=72   tile = one row of 2 streams (8 pos each) NT = 4,       2 streams per wave
//   D: F=4  R=2 C=96   tile = one row of 4 streams              NT = 2,       4 streams per wave
//   E: F=2  R=2 C=96   tile = one row of 8 streams              NT = 2,       8 streams per wave (+ conv19)
// Channel counts that are not multiples of 16 (24, 72) are padded to whole tiles with zero weights / zero BN.
//
// Numerics: exact fp32 (the MFMA is a k-ordered fmaf chain); the only difference to the reference graph is the
// summation order inside a dot product (per-tap partial sums for the 1x3 layers).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace owr {

typedef float f32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ float leaky_clamp(float x) { return fmaxf(fmaxf(0.2f * x, x), -0.4f); }
// max(a, b) as v_med3_f32(a, b, +inf): fmaxf() follows IEEE maxNum, for which LLVM first quiets every input it cannot prove
// canonical (values coming out of an MFMA, a DPP move or an opaque asm) with an extra "v_max_f32 x, x, x" -- 12 % of stage A's
// VALU instructions were such no-ops.  NaNs are not expected here (they would already have poisoned the convolution).
// (+inf comes from an opaque scalar move: with a literal the optimiser rewrites the median back into maxNum.)
__device__ __forceinline__ float fmax_nc(float a, float b) {
    float inf;
    asm("s_mov_b32 %0, 0x7f800000" : "=s"(inf));
    return __builtin_amdgcn_fmed3f(a, b, inf);
}

// ---- DPP helpers: shifts inside the 16-lane rows (= the 16 positions of a tile; j groups shift alike) ----------
__device__ __forceinline__ float dpp_shr1_zero(float x) {      // lane p <- x[p-1], lane 0 <- 0
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), 0x111, 0xf, 0xf, true));
}
__device__ __forceinline__ float dpp_shl1_zero(float x) {      // lane p <- x[p+1], lane 15 <- 0
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), 0x101, 0xf, 0xf, true));
}
__device__ __forceinline__ float dpp_shr1_carry(float x, float c) {   // lane p <- x[p-1], lane 0 <- c[15]
    const int t = __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, c), 0x121 /*row_ror:1*/, 0xf, 0xf, false);
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(t, __builtin_bit_cast(int, x), 0x111, 0xf, 0xf, false));
}
__device__ __forceinline__ float dpp_shl1_carry(float x, float c) {   // lane p <- x[p+1], lane 15 <- c[0]
    const int t = __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, c), 0x12F /*row_ror:15*/, 0xf, 0xf, false);
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(t, __builtin_bit_cast(int, x), 0x101, 0xf, 0xf, false));
}

// combine the three per-tap accumulators of a 1x3 layer: out[p] = a0[p-1] + a1[p] + a2[p+1], zero beyond a
// stream's F positions (F < 16: several streams share the 16 positions of a tile)
template <int F>
__device__ __forceinline__ f32x4 combine_taps(const f32x4 a0, const f32x4 a1, const f32x4 a2, int pos) {
    f32x4 r;
    const bool first = (pos & (F - 1)) == 0, last = (pos & (F - 1)) == F - 1;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        float l = dpp_shr1_zero(a0[e]);
        float h = dpp_shl1_zero(a2[e]);
        if (F < 16) { l = first ? 0.f : l; h = last ? 0.f : h; }
        r[e] = (a1[e] + l) + h;
    }
    return r;
}

// An opaque def/use: keeps the epilogue that produced v at this point of the program.  Without it LLVM's IR-level
// sinking moves the whole per-tile epilogue (tap combine, BatchNorm, activation) down to the first use in the NEXT
// layer, i.e. keeps the raw accumulators of every output tile alive (~2x the registers, spills).
__device__ __forceinline__ void pin(f32x4& v) {
    float a = v[0], b = v[1], c = v[2], d = v[3];
    asm volatile("" : "+v"(a), "+v"(b), "+v"(c), "+v"(d));
    v = f32x4{a, b, c, d};
}

// Channel counts with C % 16 == 8 (24, 72) leave the last channel tile half empty: lanes j = 2,3 of its four registers
// carry no channel.  As an MFMA B operand that would waste half of four k-steps, so the tile is re-packed into TWO full
// registers with v_permlane32_swap: P0 = [e0 | e1], P1 = [e2 | e3] (lower 32 lanes | upper 32 lanes), i.e. lane (p, j)
// of P_e' carries channel 16ct + 4(j&1) + 2e' + (j>>1).  The weight packing (pack_rr, host) uses the same k order.
__device__ __forceinline__ f32x4 pack_half(const f32x4 v) {
    // inline asm on purpose: __builtin_amdgcn_permlane32_swap mis-models its second result on this toolchain (tools/ubench/
    // lane_test.hip); the instruction swaps lanes 32..63 of its first operand with lanes 0..31 of the second, in place.
    // s_nop: hipcc inserts no hazard wait states around asm operands (VALU write -> lane-crossing read, and -> MFMA read).
    float a = v[0], b = v[1], c = v[2], d = v[3];
    asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1\n\tv_permlane32_swap_b32 %2, %3\n\ts_nop 1"
                 : "+v"(a), "+v"(b), "+v"(c), "+v"(d));
    return f32x4{a, c, 0.f, 0.f};
}

template <bool BN>
__device__ __forceinline__ f32x4 bn_act(const f32x4 v, const float* __restrict__ scale, const float* __restrict__ shift,
                                        int oct, int j) {
    if (!BN) return v;
    const f32x4 sc = *reinterpret_cast<const f32x4*>(scale + oct * 16 + 4 * j);
    const f32x4 sh = *reinterpret_cast<const f32x4*>(shift + oct * 16 + 4 * j);
    f32x4 r;
#pragma unroll
    for (int e = 0; e < 4; ++e) r[e] = leaky_clamp(v[e] * sc[e] + sh[e]);
    return r;
}

// packed weights of one layer: [oct][tap][ct][lane][e]  (lane (i,j), e -> w[tap][cin = 16ct+4j+e][cout = 16oct+i])
template <int NCTI>
__device__ __forceinline__ f32x4 load_w(const float* __restrict__ w, int oct, int tap, int ct, int lane) {
    return *reinterpret_cast<const f32x4*>(w + ((size_t)(((oct * 3 + tap) * NCTI + ct) * 64 + lane)) * 4);
}

#ifndef OWR_PD
#define OWR_PD 2      // weight prefetch distance in (oct,tap,ct) iterations
#endif
#ifndef OWR_WPS
#define OWR_WPS 2     // waves per SIMD the register budget is sized for
#endif
#ifndef OWR_SCHEDBAR
#define OWR_SCHEDBAR 1
#endif
#if OWR_SCHEDBAR
#define OWR_SB() __builtin_amdgcn_sched_barrier(0)
#else
#define OWR_SB() do {} while (0)
#endif

// 1x3 (mel axis) layer on NT tiles held in registers
template <int NCTI, int NCTO, int NT, int F, bool BN>
__device__ __forceinline__ void conv_mel(const f32x4 (&in)[NT][NCTI], f32x4 (&out)[NT][NCTO], const float* __restrict__ w,
                                         const float* __restrict__ scale, const float* __restrict__ shift, int lane) {
    const int pos = lane & 15, j = lane >> 4;
    const bool first = (pos & (F - 1)) == 0, last = (pos & (F - 1)) == F - 1;
    constexpr int NIT = NCTO * 3 * NCTI;
    f32x4 wq[OWR_PD];
#pragma unroll
    for (int u = 0; u < OWR_PD; ++u) wq[u] = load_w<NCTI>(w, (u / NCTI) / 3, (u / NCTI) % 3, u % NCTI, lane);
#pragma unroll
    for (int oct = 0; oct < NCTO; ++oct) {
        f32x4 res[NT];
#pragma unroll
        for (int tap = 0; tap < 3; ++tap) {
            f32x4 acc[NT];
#pragma unroll
            for (int t = 0; t < NT; ++t) acc[t] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int ct = 0; ct < NCTI; ++ct) {
                const int it = (oct * 3 + tap) * NCTI + ct;
                const f32x4 a = wq[it % OWR_PD];
                const int nx = it + OWR_PD;
                if (nx < NIT) wq[it % OWR_PD] = load_w<NCTI>(w, (nx / NCTI) / 3, (nx / NCTI) % 3, nx % NCTI, lane);
#pragma unroll
                for (int e = 0; e < 4; ++e)
#pragma unroll
                    for (int t = 0; t < NT; ++t)
                        acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[e], in[t][ct][e], acc[t], 0, 0, 0);
            }
            // out[p] = tap0[p-1] + tap1[p] + tap2[p+1], zero beyond a stream's F positions
#pragma unroll
            for (int t = 0; t < NT; ++t)
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    if (tap == 0) { const float l = dpp_shr1_zero(acc[t][e]); res[t][e] = (F < 16 && first) ? 0.f : l; }
                    else if (tap == 1) res[t][e] += acc[t][e];
                    else { const float hh = dpp_shl1_zero(acc[t][e]); res[t][e] += (F < 16 && last) ? 0.f : hh; }
                }
        }
#pragma unroll
        for (int t = 0; t < NT; ++t) { out[t][oct] = bn_act<BN>(res[t], scale, shift, oct, j); pin(out[t][oct]); }
    }
}

// 3x1 (time axis) layer: rows[0..NR+1] are tiles of consecutive rows (two history rows first); out row r uses rows r..r+2
template <int NCTI, int NCTO, int NR, bool BN>
__device__ __forceinline__ void conv_time(const f32x4 (&h0)[NCTI], const f32x4 (&h1)[NCTI], const f32x4 (&in)[NR][NCTI],
                                          f32x4 (&out)[NR][NCTO], const float* __restrict__ w,
                                          const float* __restrict__ scale, const float* __restrict__ shift, int lane) {
    const int j = lane >> 4;
    constexpr int NIT = NCTO * 3 * NCTI;
    f32x4 wq[OWR_PD];
#pragma unroll
    for (int u = 0; u < OWR_PD; ++u) wq[u] = load_w<NCTI>(w, (u / NCTI) / 3, (u / NCTI) % 3, u % NCTI, lane);
#pragma unroll
    for (int oct = 0; oct < NCTO; ++oct) {
        f32x4 acc[NR];
#pragma unroll
        for (int r = 0; r < NR; ++r) acc[r] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int tap = 0; tap < 3; ++tap)
#pragma unroll
            for (int ct = 0; ct < NCTI; ++ct) {
                const int it = (oct * 3 + tap) * NCTI + ct;
                const f32x4 a = wq[it % OWR_PD];
                const int nx = it + OWR_PD;
                if (nx < NIT) wq[it % OWR_PD] = load_w<NCTI>(w, (nx / NCTI) / 3, (nx / NCTI) % 3, nx % NCTI, lane);
#pragma unroll
                for (int e = 0; e < 4; ++e)
#pragma unroll
                    for (int r = 0; r < NR; ++r) {
                        const int src = r + tap;                     // 0,1 = history rows
                        const int ri = src >= 2 ? src - 2 : 0;
                        const float b = src == 0 ? h0[ct][e] : (src == 1 ? h1[ct][e] : in[ri][ct][e]);
                        acc[r] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[e], b, acc[r], 0, 0, 0);
                    }
            }
#pragma unroll
        for (int r = 0; r < NR; ++r) { out[r][oct] = bn_act<BN>(acc[r], scale, shift, oct, j); pin(out[r][oct]); }
    }
}

// ---- weights streamed through LDS ---------------------------------------------------------------------------
// Loading every weight register from L2 per wave caps the matrix pipe at ~80 % (all 2048 resident waves pull the same
// 45-110 KB per layer; measured with tools/ubench).  Instead the four waves of a workgroup share ONE copy: the weights
// of one output-channel tile ("chunk": 3 taps x NCTI blocks of 1 KB in MFMA operand order) are moved global -> LDS
// by global_load_lds_dwordx4 (no VGPR staging) into a double buffer while the previous chunk is being consumed with
// ds_read_b128; one s_waitcnt vmcnt(0) + s_barrier per chunk (every 48*NCTI*NT MFMAs).
#ifndef OWR_SGB
#define OWR_SGB 1     // pin the LDS-operand-read / MFMA interleave: reads run OWR_LDSPD blocks ahead, one read per 4*NT MFMAs
#endif
#ifndef OWR_LDSPD
#define OWR_LDSPD 2
#endif
#if OWR_SGB
#define OWR_SGB_PROLOGUE() __builtin_amdgcn_sched_group_barrier(0x100, OWR_LDSPD, 0)
#define OWR_SGB_STEP(NMFMA) do { __builtin_amdgcn_sched_group_barrier(0x008, NMFMA, 0); __builtin_amdgcn_sched_group_barrier(0x100, 1, 0); } while (0)
#else
#define OWR_SGB_PROLOGUE() do {} while (0)
#define OWR_SGB_STEP(NMFMA) do {} while (0)
#endif
constexpr int WBUF_FLOATS = 3 * 6 * 256;          // largest chunk: 3 taps x 6 channel tiles x 1 KB
constexpr int WG_WAVES = 4;

// a pointer the compiler can keep in SGPRs (the DMA below then uses the scalar-base + lane-offset address form instead of
// one 64-bit VGPR address pair per block)
__device__ __forceinline__ const float* uniform_ptr(const float* p) {
    const uint64_t v = reinterpret_cast<uint64_t>(p);
    const uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)v), hi = __builtin_amdgcn_readfirstlane((uint32_t)(v >> 32));
    return reinterpret_cast<const float*>(((uint64_t)hi << 32) | lo);
}

template <int NBLK, int WG_WAVES = owr::WG_WAVES>
__device__ __forceinline__ void issue_chunk(const float* __restrict__ gsrc, float* ldst, int wave, int lane) {
    const float* base = uniform_ptr(gsrc + wave * 256);
    const unsigned voff = lane * 4;
#pragma unroll
    for (int u = 0; u < (NBLK + WG_WAVES - 1) / WG_WAVES; ++u) {
        const int i = u * WG_WAVES + wave;
        if (i < NBLK)
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(base + u * WG_WAVES * 256 + voff),
                                             (__attribute__((address_space(3))) void*)(ldst + i * 256), 16, 0, 0);
    }
}
__device__ __forceinline__ void chunk_sync() {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
}
__device__ __forceinline__ f32x4 lds_w(const float* buf, int blk, int lane) {
    return *reinterpret_cast<const f32x4*>(buf + (blk * 64 + lane) * 4);
}

// CH0 = running chunk number of this layer's first chunk (selects the buffer parity); NEXT_NBLK = blocks of the chunk
// that follows this layer's last one (first chunk of the next layer), 0 = none.
template <int NCTI, int NCTO, int NT, int F, bool BN, int CH0, int NEXT_NBLK, bool HIN = false, bool HOUT = false>
__device__ __forceinline__ void conv_mel_lds(const f32x4 (&in)[NT][NCTI], f32x4 (&out)[NT][NCTO], float* wbuf,
                                             const float* __restrict__ w, const float* __restrict__ w_next,
                                             const float* __restrict__ scale, const float* __restrict__ shift, int wave, int lane, float* wbuf1) {
    const int pos = lane & 15, j = lane >> 4;
    const bool first = (pos & (F - 1)) == 0, last = (pos & (F - 1)) == F - 1;
#pragma unroll
    for (int oct = 0; oct < NCTO; ++oct) {
        const float* cur = ((CH0 + oct) & 1) ? wbuf1 : wbuf;
        float* nxt = ((CH0 + oct + 1) & 1) ? wbuf1 : wbuf;
        if (oct + 1 < NCTO) issue_chunk<3 * NCTI>(w + (size_t)(oct + 1) * 3 * NCTI * 256, nxt, wave, lane);
        else if (NEXT_NBLK > 0) issue_chunk<NEXT_NBLK>(w_next, nxt, wave, lane);
        // out[p] = tap0[p-1] + tap1[p] + tap2[p+1] (zero beyond a stream's F positions).  Tap order 0, 2, 1: the two taps
        // that need a lane shift of their accumulators run first, so that their DPP epilogues overlap with the MFMA
        // chain of the following tap; the shifted tap-0 sum is the starting accumulator of the tap-1 chain.
        f32x4 res[NT], accs[2][NT];
        OWR_SGB_PROLOGUE();
#pragma unroll
        for (int ti = 0; ti < 3; ++ti) {
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
            for (int ct = 0; ct < NCTI; ++ct) {
                const f32x4 a = lds_w(cur, tap * NCTI + ct, lane);
                const int ne = (HIN && ct == NCTI - 1) ? 2 : 4;       // packed half tile: two k-steps
#pragma unroll
                for (int e = 0; e < 4; ++e)
#pragma unroll
                    for (int t = 0; t < NT; ++t)
                        if (e < ne) acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[e], in[t][ct][e], acc[t], 0, 0, 0);
                if (HIN && ct == NCTI - 1) OWR_SGB_STEP(2 * NT); else OWR_SGB_STEP(4 * NT);
            }
#pragma unroll
            for (int t = 0; t < NT; ++t) {
                if (ti < 2) accs[ti][t] = acc[t];
                else {
#pragma unroll
                    for (int e = 0; e < 4; ++e) { const float hh = dpp_shl1_zero(accs[1][t][e]); res[t][e] = acc[t][e] + ((F < 16 && last) ? 0.f : hh); }
                }
            }
        }
#pragma unroll
        for (int t = 0; t < NT; ++t) {
            out[t][oct] = bn_act<BN>(res[t], scale, shift, oct, j);
            pin(out[t][oct]);
            if (HOUT && oct == NCTO - 1) { out[t][oct] = pack_half(out[t][oct]); pin(out[t][oct]); }
        }
        __builtin_amdgcn_sched_barrier(0);     // the epilogue of this tile is finished here, not sunk to the end of the layer
        if (oct + 1 < NCTO || NEXT_NBLK > 0) chunk_sync();
    }
}

template <int NCTI, int NCTO, int NR, bool BN, int CH0, int NEXT_NBLK, bool HIN = false, bool HOUT = false>
__device__ __forceinline__ void conv_time_lds(const f32x4 (&h0)[NCTI], const f32x4 (&h1)[NCTI], const f32x4 (&in)[NR][NCTI],
                                              f32x4 (&out)[NR][NCTO], float* wbuf, const float* __restrict__ w,
                                              const float* __restrict__ w_next, const float* __restrict__ scale,
                                              const float* __restrict__ shift, int wave, int lane, float* wbuf1) {
    const int j = lane >> 4;
#pragma unroll
    for (int oct = 0; oct < NCTO; ++oct) {
        const float* cur = ((CH0 + oct) & 1) ? wbuf1 : wbuf;
        float* nxt = ((CH0 + oct + 1) & 1) ? wbuf1 : wbuf;
        if (oct + 1 < NCTO) issue_chunk<3 * NCTI>(w + (size_t)(oct + 1) * 3 * NCTI * 256, nxt, wave, lane);
        else if (NEXT_NBLK > 0) issue_chunk<NEXT_NBLK>(w_next, nxt, wave, lane);
        f32x4 acc[NR];
#pragma unroll
        for (int r = 0; r < NR; ++r) acc[r] = f32x4{0.f, 0.f, 0.f, 0.f};
        OWR_SGB_PROLOGUE();
#pragma unroll
        for (int tap = 0; tap < 3; ++tap)
#pragma unroll
            for (int ct = 0; ct < NCTI; ++ct) {
                const f32x4 a = lds_w(cur, tap * NCTI + ct, lane);
                const int ne = (HIN && ct == NCTI - 1) ? 2 : 4;
#pragma unroll
                for (int e = 0; e < 4; ++e)
#pragma unroll
                    for (int r = 0; r < NR; ++r) {
                        const int src = r + tap;
                        const int ri = src >= 2 ? src - 2 : 0;
                        const float b = src == 0 ? h0[ct][e] : (src == 1 ? h1[ct][e] : in[ri][ct][e]);
                        if (e < ne) acc[r] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[e], b, acc[r], 0, 0, 0);
                    }
                if (HIN && ct == NCTI - 1) OWR_SGB_STEP(2 * NR); else OWR_SGB_STEP(4 * NR);
            }
#pragma unroll
        for (int r = 0; r < NR; ++r) {
            out[r][oct] = bn_act<BN>(acc[r], scale, shift, oct, j);
            pin(out[r][oct]);
            if (HOUT && oct == NCTO - 1) { out[r][oct] = pack_half(out[r][oct]); pin(out[r][oct]); }
        }
        __builtin_amdgcn_sched_barrier(0);     // the epilogue of this tile is finished here, not sunk to the end of the layer
        if (oct + 1 < NCTO || NEXT_NBLK > 0) chunk_sync();
    }
}

// register-dump I/O: tile = NCT x f32x4 per lane; memory [..][4*NCT registers][64 lanes]
template <int NCT>
__device__ __forceinline__ void load_tile(f32x4 (&t)[NCT], const float* __restrict__ base, int lane) {
#pragma unroll
    for (int ct = 0; ct < NCT; ++ct)
#pragma unroll
        for (int e = 0; e < 4; ++e) t[ct][e] = base[(ct * 4 + e) * 64 + lane];
}
template <int NCT>
__device__ __forceinline__ void store_tile(const f32x4 (&t)[NCT], float* __restrict__ base, int lane) {
#pragma unroll
    for (int ct = 0; ct < NCT; ++ct)
#pragma unroll
        for (int e = 0; e < 4; ++e) base[(ct * 4 + e) * 64 + lane] = t[ct][e];
}

// debug: dense [rows][F][C] dump of a tile row for the streams it holds (tests only)
template <int NCT, int F, int C, bool PACKED = false>
__device__ __forceinline__ void dump_tile(const f32x4 (&t)[NCT], float* __restrict__ dbg, size_t stride, int off, int s_first,
                                          int row, int S, int lane) {
    const int pos = lane & 15, j = lane >> 4;
    const int sp = pos / F, f = pos % F, s = s_first + sp;
    if (s >= S) return;
#pragma unroll
    for (int ct = 0; ct < NCT; ++ct)
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            int c = ct * 16 + 4 * j + e;
            if (PACKED && ct == NCT - 1) c = e < 2 ? ct * 16 + 4 * (j & 1) + 2 * e + (j >> 1) : C;   // pack_half order
            if (c < C) dbg[(size_t)s * stride + off + (row * F + f) * C + c] = t[ct][e];
        }
}

// ------------------------------------------------------------------------------------------------
// stages B..E
// ------------------------------------------------------------------------------------------------
template <int CIN_, int C_, int R_, int F_, int PT_, int PF_, int RP_, int WPS_>
struct RCfg {
    static constexpr int WPS = WPS_;                      // waves per SIMD the register allocation is sized for
    static constexpr int CIN = CIN_, C = C_, R = R_, F = F_, PT = PT_, PF = PF_;
    // rows per pass: the R new rows are carried through the four layers RP at a time (R/RP passes inside the kernel,
    // the conv histories going through memory in between) -- halves the live registers of the widest stage
    static constexpr int RP = RP_, NPASS = R_ / RP_;
    static_assert(R_ % RP_ == 0 && RP_ % PT_ == 0, "passes are whole pooling groups");
    static constexpr int NCTI = (CIN + 15) / 16, NCT = (C + 15) / 16;
    static constexpr bool HIN = CIN % 16 == 8, HOUT = C % 16 == 8;     // half last channel tile: kept re-packed (pack_half)
    static constexpr int SPT = 16 / F;                    // streams per tile = streams per wave
    static constexpr int XIN_FLOATS = R * NCTI * 4 * 64;  // per group
    static constexpr int HIST_FLOATS = 2 * NCT * 4 * 64;  // per group, per history array
    static constexpr int RO = R / PT, FO = F / PF;
    static constexpr int WAVES = 4;
};
#ifndef OWR_RC_RP
#define OWR_RC_RP 4
#endif
#ifndef OWR_WPS_B
#define OWR_WPS_B 2
#endif
#ifndef OWR_WPS_C
#define OWR_WPS_C 2
#endif
#ifndef OWR_WPS_D
#define OWR_WPS_D 3
#endif
#ifndef OWR_WPS_E
#define OWR_WPS_E 3
#endif
using RB = RCfg<24, 48, 4, 16, 1, 2, 4, OWR_WPS_B>;
using RC = RCfg<48, 72, 4, 8, 2, 2, OWR_RC_RP, OWR_WPS_C>;
using RD = RCfg<72, 96, 2, 4, 1, 2, 2, OWR_WPS_D>;
using RE = RCfg<96, 96, 2, 2, 2, 2, 2, OWR_WPS_E>;

struct RStageParams {
    const float* xin;      // [G][R][4*NCTI][64]
    float* xout;           // next stage's xin (its own geometry)
    float* hist_b;         // [G][2][4*NCT][64]
    float* hist_d;
    const float* w[4];     // rr-packed
    const float* scale[4]; // padded to NCT*16
    const float* shift[4];
    int n_groups;          // groups (waves) to run
    int S;                 // streams (debug / ring bounds)
    // last stage
    float* hist19;         // [G][2][24][64]
    const float* w19;
    float* feat;           // [S][TR][96]
    float* emb;            // [S][96]
    const uint32_t* nfeat;
    int TR;
    float* dbg;
    size_t dbg_stride;
    int dbg_off[5];
    float clampv[4];       // f16-split family: -0.4 K of each layer (activation clamp in the layer's scaled domain, owwhip_hx.h act1)
    float dbg_mul[5];      // f16-split family: 1 / K of each layer (debug dumps are written in true units); [4] = conv19
    float xmul;            // f16-split family: factor of the pooled hand-over (K of the next stage's input / K of this stage's last layer)
    float emb_mul;         // f16-split family, last stage: 1 / K of conv19 (embeddings are stored in true units)
    int* range_flag;       // f16-split family: sticky out-of-range flag of the handle (owwhip_hx.h nan_guard); nullptr otherwise
    const uint8_t* stream_on;  // oww_step_masked (register-resident families): [S] 1 = the stream takes part in this step; nullptr = all do
    const int* glist;          // f16-split family, oww_step_masked with few participants: the n_groups groups (of this stage's SPT streams)
                               // that hold at least one participating stream; nullptr = groups g_base .. g_base + n_groups-1
    int g_base;                // f16-split family: first group of the block this launch covers (block-pipelined step; 0 otherwise)
};

// max-pool PT x PF of the stage output and scatter into the next stage's register-dump layout
//   next geometry: Fn = F/PF positions per stream, SPTn = 16/Fn streams per tile, rows RO, NCT channel tiles
template <class C>
__device__ __forceinline__ void pool_store(const f32x4 (&y)[C::RP][C::NCT], float* __restrict__ xout, int g, int ro0, int lane) {
    constexpr int F = C::F, FO = C::FO, RO = C::RO, NCT = C::NCT;
    constexpr int SPTN = 16 / FO;
    const int pos = lane & 15, j = lane >> 4;
    const int sp = pos / F, f = pos % F;
    const int s = g * C::SPT + sp;                         // global stream of this lane
    const int gn = s / SPTN, spn = s % SPTN;
    const int posn = spn * FO + f / 2;
    const bool writer = (f & 1) == 0;
#pragma unroll
    for (int ro = 0; ro < C::RP / C::PT; ++ro)
#pragma unroll
        for (int ct = 0; ct < NCT; ++ct)
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                float m = y[ro * C::PT][ct][e];
                if (C::PT == 2) m = fmax_nc(m, y[ro * C::PT + 1][ct][e]);
                m = fmax_nc(m, dpp_shl1_zero(m));            // pair (f, f+1); valid on even f
                if (writer) xout[((size_t)(gn * RO + ro0 + ro) * (NCT * 4) + ct * 4 + e) * 64 + j * 16 + posn] = m;
            }
}

template <class C, bool LAST, bool DBG>
// (DBG: the per-layer dumps of tests and of oww_commit's calibration run keep every layer's tiles alive; at the production occupancy
//  they spilled 100+ bytes per lane into scratch -- a debug launch is a 32-stream grid, so it simply takes the whole register file)
__global__ __launch_bounds__(256, DBG ? 1 : C::WPS) void rstage_kernel(RStageParams p) {
    constexpr int NCTI = C::NCTI, NCT = C::NCT, R = C::R, RP = C::RP, F = C::F;
    static_assert(!LAST || C::NPASS == 1, "the last stage runs in one pass");
    constexpr int NBA = 3 * NCTI, NB = 3 * NCT;             // 1 KB blocks per chunk: first layer / other layers
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);      // wave-uniform: addresses built from it stay in SGPRs
    int g = blockIdx.x * C::WAVES + wave;
    // (two distinct LDS objects, not one array of two halves: see owh::hstage_kernel -- the compiler otherwise waits for the chunk a
    //  wave has just issued before it lets the wave read the current one)
    __shared__ __attribute__((aligned(16))) float wbuf[WBUF_FLOATS];
    __shared__ __attribute__((aligned(16))) float wbuf1[WBUF_FLOATS];
    // folded BatchNorm of the four layers in LDS: global loads of them would be hoisted over the chunk barriers into
    // ~50 live registers; LDS reads stay inside their chunk
    __shared__ __attribute__((aligned(16))) float sbn[4][2][NCT * 16];
    const bool active = g < p.n_groups;                      // every wave keeps running (workgroup barriers); idle ones recompute
    if (!active) g = p.n_groups - 1;                         // the last group and store nothing
    issue_chunk<NBA>(p.w[0], wbuf, wave, lane);
    for (int i = threadIdx.x; i < 4 * NCT * 16; i += 256) {
        const int l = i / (NCT * 16), c = i % (NCT * 16);
        sbn[l][0][c] = p.scale[l][c];
        sbn[l][1][c] = p.shift[l][c];
    }
    const int s_first = g * C::SPT;
    // masked steps (oww_step_masked): a stream that sits the step out is computed like any other and stores none of its state;
    // per lane, because a tile holds SPT streams (position p of a tile belongs to stream p / F in this family)
    const bool lane_on = active && (p.stream_on == nullptr || p.stream_on[min(s_first + (lane & 15) / F, p.S - 1)] != 0);

    float* hb = p.hist_b + (size_t)g * C::HIST_FLOATS;
    float* hd = p.hist_d + (size_t)g * C::HIST_FLOATS;
    f32x4 Yd[RP][NCT];
#pragma unroll 1
    for (int pass = 0; pass < C::NPASS; ++pass) {
    f32x4 X[RP][NCTI];
#pragma unroll
    for (int r = 0; r < RP; ++r) load_tile<NCTI>(X[r], p.xin + ((size_t)g * R + pass * RP + r) * (NCTI * 4 * 64), lane);
    if (pass == 0) chunk_sync();

    // conv a: 1x3, CIN -> C
    f32x4 Ya[RP][NCT];
    conv_mel_lds<NCTI, NCT, RP, F, true, 0, NB, C::HIN, C::HOUT>(X, Ya, wbuf, p.w[0], p.w[1], sbn[0][0], sbn[0][1], wave, lane, wbuf1);
    if (DBG && p.dbg && active) {
#pragma unroll
        for (int r = 0; r < RP; ++r) dump_tile<NCT, F, C::C, C::HOUT>(Ya[r], p.dbg, p.dbg_stride, p.dbg_off[0], s_first, pass * RP + r, p.S, lane);
    }
    OWR_SB();
    // conv b: 3x1 over [hist_b(2) ; Ya]
    f32x4 H0[NCT], H1[NCT];
    load_tile<NCT>(H0, hb, lane);
    load_tile<NCT>(H1, hb + NCT * 4 * 64, lane);
    f32x4 Yb[RP][NCT];
    conv_time_lds<NCT, NCT, RP, true, NCT, NB, C::HOUT, C::HOUT>(H0, H1, Ya, Yb, wbuf, p.w[1], p.w[2], sbn[1][0], sbn[1][1], wave, lane, wbuf1);
    if (lane_on) {
        store_tile<NCT>(Ya[RP - 2], hb, lane);
        store_tile<NCT>(Ya[RP - 1], hb + NCT * 4 * 64, lane);
    }
    if (DBG && p.dbg && active) {
#pragma unroll
        for (int r = 0; r < RP; ++r) dump_tile<NCT, F, C::C, C::HOUT>(Yb[r], p.dbg, p.dbg_stride, p.dbg_off[1], s_first, pass * RP + r, p.S, lane);
    }
    OWR_SB();
    // conv c: 1x3
    f32x4 Yc[RP][NCT];
    conv_mel_lds<NCT, NCT, RP, F, true, 2 * NCT, NB, C::HOUT, C::HOUT>(Yb, Yc, wbuf, p.w[2], p.w[3], sbn[2][0], sbn[2][1], wave, lane, wbuf1);
    if (DBG && p.dbg && active) {
#pragma unroll
        for (int r = 0; r < RP; ++r) dump_tile<NCT, F, C::C, C::HOUT>(Yc[r], p.dbg, p.dbg_stride, p.dbg_off[2], s_first, pass * RP + r, p.S, lane);
    }
    OWR_SB();
    // conv d: 3x1 over [hist_d(2) ; Yc]
    load_tile<NCT>(H0, hd, lane);
    load_tile<NCT>(H1, hd + NCT * 4 * 64, lane);
    conv_time_lds<NCT, NCT, RP, true, 3 * NCT, (LAST ? NB : (C::NPASS > 1 ? NBA : 0)), C::HOUT, C::HOUT>(H0, H1, Yc, Yd, wbuf, p.w[3], LAST ? p.w19 : p.w[0], sbn[3][0], sbn[3][1], wave, lane, wbuf1);
    if (lane_on) {
        store_tile<NCT>(Yc[RP - 2], hd, lane);
        store_tile<NCT>(Yc[RP - 1], hd + NCT * 4 * 64, lane);
    }
    if (DBG && p.dbg && active) {
#pragma unroll
        for (int r = 0; r < RP; ++r) dump_tile<NCT, F, C::C, C::HOUT>(Yd[r], p.dbg, p.dbg_stride, p.dbg_off[3], s_first, pass * RP + r, p.S, lane);
    }

    if (!LAST && active) pool_store<C>(Yd, p.xout, g, pass * (RP / C::PT), lane);
    }   // pass
    if (!LAST && C::NPASS > 1) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // the last pass prefetched a chunk nobody uses: drain it

    if (LAST) {
        // pool 2x2 -> one position per stream (even lanes), compact to lanes pos' = stream via the LDS crossbar,
        // then conv19: 3x1 96->96 without BN/activation over [hist19(2) ; pooled]
        static_assert(!LAST || (C::RO == 1 && C::FO == 1 && NCT == 6), "last stage pools to one position, 96 channels");
        const int pos = lane & 15, j = lane >> 4;
        f32x4 Pl[1][NCT];
        const int src = (j * 16 + 2 * (pos & 7)) * 4;                    // byte address of the source lane
#pragma unroll
        for (int ct = 0; ct < NCT; ++ct)
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                float m = fmax_nc(Yd[0][ct][e], Yd[1][ct][e]);
                m = fmax_nc(m, dpp_shl1_zero(m));
                Pl[0][ct][e] = __builtin_bit_cast(float, __builtin_amdgcn_ds_bpermute(src, __builtin_bit_cast(int, m)));
            }
        float* h19 = p.hist19 + (size_t)g * (2 * NCT * 4 * 64);
        f32x4 H0[NCT], H1[NCT];
        load_tile<NCT>(H0, h19, lane);
        load_tile<NCT>(H1, h19 + NCT * 4 * 64, lane);
        f32x4 E[1][NCT];
        conv_time_lds<NCT, NCT, 1, false, 4 * NCT, 0>(H0, H1, Pl, E, wbuf, p.w19, nullptr, nullptr, nullptr, wave, lane, wbuf1);
        const bool on19 = active && (p.stream_on == nullptr || p.stream_on[min(s_first + (pos & 7), p.S - 1)] != 0);   // lanes 8..15 mirror 0..7
        if (on19) {
            store_tile<NCT>(H1, h19, lane);
            store_tile<NCT>(Pl[0], h19 + NCT * 4 * 64, lane);
        }
        const int s = s_first + pos;
        if (on19 && pos < C::SPT && s < p.S) {
            const uint32_t slot = p.nfeat[s] % (uint32_t)p.TR;
#pragma unroll
            for (int ct = 0; ct < NCT; ++ct) {
                *reinterpret_cast<f32x4*>(p.feat + ((size_t)s * p.TR + slot) * 96 + ct * 16 + 4 * j) = E[0][ct];
                *reinterpret_cast<f32x4*>(p.emb + (size_t)s * 96 + ct * 16 + 4 * j) = E[0][ct];
                if (DBG && p.dbg) *reinterpret_cast<f32x4*>(p.dbg + (size_t)s * p.dbg_stride + p.dbg_off[4] + ct * 16 + 4 * j) = E[0][ct];
            }
        }
    }
}

// ------------------------------------------------------------------------------------------------
// stage A: mel rows -> conv0 3x3 (1->24, ReLU, BN, act) -> conv1 1x3 -> conv2 3x1 -> pool 2x2
// one stream per wave at a time, persistent over streams; conv1/conv2 weights live in registers
// ------------------------------------------------------------------------------------------------
struct RAParams {
    const float* mel;      // [S][mel_stride], chunk rows at mel_off
    int mel_stride, mel_off;
    float* hist_mel;       // [S][2][32]
    float* hist2;          // [S][2 rows][2 halves][8][64]  conv2 input history (register dump)
    const float* w0;       // [2 oct][3 ksteps][64]
    const float* w1;       // rr-packed, NCTI = 2
    const float* w2;
    const float* scale[3]; // padded to 32
    const float* shift[3];
    float* xout;           // stage B xin: [S][4][8][64]
    int n_streams;         // streams to run
    int s_base;            // f16-split family: first stream of the block this launch covers (block-pipelined step; 0 otherwise)
    int S;
    float* dbg;
    size_t dbg_stride;
    int dbg_off[3];
    float clampv[3], dbg_mul[3], xmul;   // see RStageParams (f16-split family; there scale[0] = conv0's med3 bound, shift[l] = K * BatchNorm shift)
    int* range_flag;       // see RStageParams::range_flag
    const uint8_t* stream_on;  // see RStageParams::stream_on
};

#ifndef OWR_WPS_A
#define OWR_WPS_A 2
#endif

template <bool DBG>
__global__ __launch_bounds__(256, DBG ? 1 : OWR_WPS_A) void rstageA_kernel(RAParams p) {
    const int lane = threadIdx.x & 63, pos = lane & 15, j = lane >> 4;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int gw = blockIdx.x * 4 + wave, nw = gridDim.x * 4;

    // conv1 / conv2 weights (2 x 12 KB, MFMA operand order) and the three folded BatchNorms -> LDS once per workgroup;
    // per wave a [10][34] tile of mel rows (2 history + 8 new, a zero column either side = the mel-axis padding)
    __shared__ __attribute__((aligned(16))) float sW[2][2 * 3 * 2 * 64 * 4];
    __shared__ __attribute__((aligned(16))) float sbn[3][2][32];
    __shared__ float sMel[4][10 * 34];
    for (int i = threadIdx.x; i < 2 * 3 * 2 * 64; i += 256) {
        reinterpret_cast<f32x4*>(sW[0])[i] = reinterpret_cast<const f32x4*>(p.w1)[i];
        reinterpret_cast<f32x4*>(sW[1])[i] = reinterpret_cast<const f32x4*>(p.w2)[i];
    }
    if (threadIdx.x < 96) {
        const int l = threadIdx.x / 32, c = threadIdx.x % 32;
        sbn[l][0][c] = p.scale[l][c];
        sbn[l][1][c] = p.shift[l][c];
    }
    for (int i = threadIdx.x; i < 4 * 10 * 34; i += 256) sMel[0][i] = 0.f;
    float W0[2][3];
#pragma unroll
    for (int oct = 0; oct < 2; ++oct)
#pragma unroll
        for (int k = 0; k < 3; ++k) W0[oct][k] = p.w0[(oct * 3 + k) * 64 + lane];
    __syncthreads();
    const f32x4* sW1b = reinterpret_cast<const f32x4*>(sW[0]) + lane;    // [(oct*3+tap)*2+ct][64 lanes]
    const f32x4* sW2b = reinterpret_cast<const f32x4*>(sW[1]) + lane;
    float* sM = sMel[wave];
    // conv0 operand gather: k-step ks carries tap k = 4ks+j = (dt, df); k >= 9 has zero weight (any finite operand will do)
    int goff[3];
#pragma unroll
    for (int ks = 0; ks < 3; ++ks) { const int k = min(4 * ks + j, 8); goff[ks] = (k / 3) * 34 + (k % 3) + pos; }

    for (int s = gw; s < p.n_streams; s += nw) {
        // an opaque zero per stream: keeps the loop-invariant LDS reads (weights, BatchNorm) from being hoisted out of
        // the stream loop into ~150 live registers
        int z = 0;
        asm volatile("" : "+s"(z));
        const f32x4* sW1 = sW1b + z;
        const f32x4* sW2 = sW2b + z;
        const float* bn = &sbn[0][0][0] + z;
        const float* mel = p.mel + (size_t)s * p.mel_stride + p.mel_off;
        float* hm = p.hist_mel + (size_t)s * 64;
        float* h2 = p.hist2 + (size_t)s * (2 * 2 * 8 * 64);
        // mel tile -> LDS (wave-private region; LDS operations of one wave execute in order)
        {
            const f32x4 m4 = *reinterpret_cast<const f32x4*>(mel + lane * 4);
            const float hv = hm[lane];
            const int row = lane >> 3, col = (lane & 7) * 4;
#pragma unroll
            for (int e = 0; e < 4; ++e) sM[(2 + row) * 34 + 1 + col + e] = m4[e];
            sM[(lane >> 5) * 34 + 1 + (lane & 31)] = hv;
        }
        f32x4 Yh[2][2][2];                            // conv1 output rows r-2, r-1: [row][half][channel block], packed form
#pragma unroll
        for (int r = 0; r < 2; ++r)
#pragma unroll
            for (int h = 0; h < 2; ++h) load_tile<2>(Yh[r][h], h2 + (r * 2 + h) * 512, lane);
#pragma unroll
        for (int q = 0; q < 4; ++q) {                 // rows 2q, 2q+1
            OWR_SB();
            // ---- conv0: 3x3, 1 -> 24 (K = 9 padded to 12), ReLU, BN, activation.  tile t = (row rr, half h)
            f32x4 Y0[4][2];
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                const int r = 2 * q + (t >> 1), h = t & 1;
                float b[3];
#pragma unroll
                for (int ks = 0; ks < 3; ++ks) b[ks] = sM[r * 34 + h * 16 + goff[ks]];
#pragma unroll
                for (int oct = 0; oct < 2; ++oct) {
                    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                    for (int ks = 0; ks < 3; ++ks) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(W0[oct][ks], b[ks], acc, 0, 0, 0);
#pragma unroll
                    for (int e = 0; e < 4; ++e) acc[e] = fmax_nc(acc[e], 0.f);
                    Y0[t][oct] = bn_act<true>(acc, bn, bn + 32, oct, j);
                    pin(Y0[t][oct]);
                }
                Y0[t][1] = pack_half(Y0[t][1]);
                pin(Y0[t][1]);
                if (DBG && p.dbg) dump_tile<2, 16, 24, true>(Y0[t], p.dbg + h * 16 * 24, p.dbg_stride, p.dbg_off[0], s, r * 2, p.S, lane);
            }
            // ---- conv1: 1x3 over the 32 mel positions of a row = two halves with carries across the seam
            f32x4 Y1[4][2];
#pragma unroll
            for (int oct = 0; oct < 2; ++oct) {
                // tap order 0, 2, 1 (see conv_mel_lds): the shifted tap-0 sums start the tap-1 accumulators
                f32x4 acc[3][4];
#pragma unroll
                for (int ti = 0; ti < 3; ++ti) {
                    const int tap = ti == 0 ? 0 : (ti == 1 ? 2 : 1);
#pragma unroll
                    for (int t = 0; t < 4; ++t) {
                        if (ti < 2) acc[tap][t] = f32x4{0.f, 0.f, 0.f, 0.f};
                        else {
#pragma unroll
                            for (int e = 0; e < 4; ++e)
                                acc[1][t][e] = (t & 1) ? dpp_shr1_carry(acc[0][t][e], acc[0][t - 1][e]) : dpp_shr1_zero(acc[0][t][e]);
                        }
                    }
#pragma unroll
                    for (int ct = 0; ct < 2; ++ct) {
                        const f32x4 a = sW1[((oct * 3 + tap) * 2 + ct) * 64];
#pragma unroll
                        for (int e = 0; e < (ct ? 2 : 4); ++e)
#pragma unroll
                            for (int t = 0; t < 4; ++t)
                                acc[tap][t] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[e], Y0[t][ct][e], acc[tap][t], 0, 0, 0);
                    }
                }
#pragma unroll
                for (int rr = 0; rr < 2; ++rr) {
                    const int t0 = 2 * rr, t1 = 2 * rr + 1;
                    f32x4 r0, r1;
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        r0[e] = acc[1][t0][e] + dpp_shl1_carry(acc[2][t0][e], acc[2][t1][e]);
                        r1[e] = acc[1][t1][e] + dpp_shl1_zero(acc[2][t1][e]);
                    }
                    Y1[t0][oct] = bn_act<true>(r0, bn + 64, bn + 96, oct, j);
                    Y1[t1][oct] = bn_act<true>(r1, bn + 64, bn + 96, oct, j);
                    pin(Y1[t0][oct]); pin(Y1[t1][oct]);
                    if (oct == 1) { Y1[t0][1] = pack_half(Y1[t0][1]); Y1[t1][1] = pack_half(Y1[t1][1]); pin(Y1[t0][1]); pin(Y1[t1][1]); }
                }
                __builtin_amdgcn_sched_barrier(0);
            }
            if (DBG && p.dbg) {
#pragma unroll
                for (int t = 0; t < 4; ++t)
                    dump_tile<2, 16, 24, true>(Y1[t], p.dbg + (t & 1) * 16 * 24, p.dbg_stride, p.dbg_off[1], s, (2 * q + (t >> 1)) * 2, p.S, lane);
            }
            // ---- conv2: 3x1 over conv1 rows (r-2, r-1, r): rows of this step see [Yh0, Yh1, Y1 row 2q, Y1 row 2q+1]
            f32x4 Y2[4][2];
#pragma unroll
            for (int oct = 0; oct < 2; ++oct) {
                f32x4 acc[4];
#pragma unroll
                for (int t = 0; t < 4; ++t) acc[t] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int tap = 0; tap < 3; ++tap)
#pragma unroll
                    for (int ct = 0; ct < 2; ++ct) {
                        const f32x4 a = sW2[((oct * 3 + tap) * 2 + ct) * 64];
#pragma unroll
                        for (int e = 0; e < (ct ? 2 : 4); ++e)
#pragma unroll
                            for (int t = 0; t < 4; ++t) {
                                const int src = (t >> 1) + tap, h = t & 1;      // 0,1 = history rows, 2,3 = this step's rows
                                const float b = src < 2 ? Yh[src][h][ct][e] : Y1[(src - 2) * 2 + h][ct][e];
                                acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[e], b, acc[t], 0, 0, 0);
                            }
                    }
#pragma unroll
                for (int t = 0; t < 4; ++t) {
                    Y2[t][oct] = bn_act<true>(acc[t], bn + 128, bn + 160, oct, j);
                    pin(Y2[t][oct]);
                    if (oct == 1) { Y2[t][1] = pack_half(Y2[t][1]); pin(Y2[t][1]); }
                }
                __builtin_amdgcn_sched_barrier(0);
            }
            if (DBG && p.dbg) {
#pragma unroll
                for (int t = 0; t < 4; ++t)
                    dump_tile<2, 16, 24, true>(Y2[t], p.dbg + (t & 1) * 16 * 24, p.dbg_stride, p.dbg_off[2], s, (2 * q + (t >> 1)) * 2, p.S, lane);
            }
            // the conv2 input window moves on by two rows
#pragma unroll
            for (int h = 0; h < 2; ++h)
#pragma unroll
                for (int ct = 0; ct < 2; ++ct) { Yh[0][h][ct] = Y1[h][ct]; Yh[1][h][ct] = Y1[2 + h][ct]; }
            // ---- pool 2x2 -> stage B input row q: 16 positions = the pooled f of both halves (channels stay in pack_half order)
            float* xo = p.xout + ((size_t)s * 4 + q) * (8 * 64);
#pragma unroll
            for (int h = 0; h < 2; ++h)
#pragma unroll
                for (int ct = 0; ct < 2; ++ct)
#pragma unroll
                    for (int e = 0; e < (ct ? 2 : 4); ++e) {
                        float m = fmax_nc(Y2[h][ct][e], Y2[2 + h][ct][e]);
                        m = fmax_nc(m, dpp_shl1_zero(m));
                        if ((pos & 1) == 0) xo[(ct * 4 + e) * 64 + j * 16 + h * 8 + (pos >> 1)] = m;
                    }
        }
        // new histories: conv1 rows 6,7 (packed) and mel rows 6,7 (not for a stream that sits a masked step out)
        if (p.stream_on && !p.stream_on[s]) continue;
#pragma unroll
        for (int r = 0; r < 2; ++r)
#pragma unroll
            for (int h = 0; h < 2; ++h) store_tile<2>(Yh[r][h], h2 + (r * 2 + h) * 512, lane);
        hm[lane] = sM[(8 + (lane >> 5)) * 34 + 1 + (lane & 31)];
    }
}

}  // namespace owr
