// owwhip_fused.h -- mel front end fused into stage A of the embedding CNN (included by owwhip.hip).
//
// BASELINE configs[2] "mel+embedding fused": the reference's hand-over from melspectrogram.onnx to embedding_model.onnx goes
// through a 970-row host ring (utils.py:387-401 -> 437-443).  In the unfused device path the mel kernel writes the step's 8 x 32
// transformed rows to HBM and stage A reads them back; here the wave that runs stage A for a stream computes the rows itself and
// hands them over in its LDS tile: they never reach HBM (1 KB written + 1 KB read per stream-step, and one launch, saved), and the
// latency-bound FFT phases of one wave overlap the matrix phases of the two other waves of its SIMD.
//
// One wave = one stream-step from PCM to the pooled stage-A output:
//   4 passes x (fetch 672 int16 samples of [480-sample tail ; 1280 new] -> Hann(400) -> one complex 512-point FFT of two frames,
//   radix-8 x 3 through the wave's own LDS planes -> |.|^2 of bins 2..121 -> sparse mel -> 10 log10) -> per-call maximum over the
//   8 x 32 values (wave reduction) -> clamp at max - 80 dB, x/10 + 2, first-call masking (utils.py:180-208, 387-401; the same
//   arithmetic and operation order as owk::mel_kernel) -> rows 2..9 of the wave's two f16 planes (hi / lo halves of every mel value,
//   split once here: conv0 gathers its operands from them as halves) -> owh::hstageA_stream.
// Twiddles, window and the sparse filter bank live in LDS (shared by the 12 waves of the workgroup) instead of 48 registers per
// lane, so that the register budget of stage A (3 waves per SIMD) is untouched.  Streaming steps of one chunk only; multi-chunk
// calls, clip embedding and the other kernel families keep the separate mel kernel.
#pragma once
#include "owwhip_hx.h"
#include "owwhip_kernels.h"

namespace owf {

using owh::lanemask_t;
using owr::f32x4;

#ifndef OWF_COMPACT_TAPS
#define OWF_COMPACT_TAPS 1 // the sparse mel taps read conflict-free compact power tables (0: the plain power rows, three bins per bank)
#endif
// Wave priority by progress.  The three waves of a SIMD run the same program on equal shares of the streams, but the SIMD's arbiter serves the
// oldest wave first -- for the whole life of this persistent launch.  Left alone, the favoured wave finishes its share early and the launch
// ends on the youngest wave's tail (PMC: the launch's cycles fall by 11 % with the priorities at an unchanged sum of wave cycles and unchanged
// instruction counts).  Default (6): the level RISES with the stream-step's progress -- FFT passes 0, 0, 1, 1 (OWF_MPRIO), row pairs of the
// matrix phase 2, 2, 3, 3 (OWF_QPRIO, owwhip_hx.h) -- so no wave is the favourite for long, and the wave nearest to handing its stream over
// wins VALU arbitration (its dependent epilogues keep the MFMAs coming) while the others' butterflies fill the gaps: front-end launch -10 %
// against no priorities (1.54-1.57 -> 1.38-1.41 ms), the step -2 ... -2.6 %, bit-identical (profiles/r06_prio_ab.txt).  Flat forms: 1 = one
// level for the whole matrix phase (-5 %), 2 = the inverse (-4 %), 4 = three levels rotating by stream (-5 %); three STATIC levels only move
// the favour to another wave (nothing).  Handing the workgroup's streams to its waves from a counter in LDS instead of the static
// partition evens the waves out as well (-7.5 % alone) and adds nothing beside the priorities; a device-wide ticket counter costs more in
// atomic round trips than it evens out (+3 %).  0 = off.
#ifndef OWF_PRIO
#define OWF_PRIO 6
#endif
#ifndef OWF_PRIO_HI
#define OWF_PRIO_HI 2
#endif
#ifndef OWF_MPRIO
#define OWF_MPRIO 11       // levels of FFT passes 0..3 as digits (0011)
#endif
#ifndef OWF_WG
#define OWF_WG 12          // waves per workgroup (one workgroup per CU: 3 waves per SIMD)
#endif
constexpr int FA_WG = OWF_WG;

struct MelAParams {
    owr::RAParams a;        // stage A (a.mel unused)
    const int16_t* pcm;     // [S][1280]
    int16_t* tail;          // [S][480]
    const uint32_t* nfeat;  // [S] frames seen (0 = first call after a reset: 5 mel rows, the first three read 1.0)
    const float* hann;      // [400]
    const int* mel_start;   // [32]
    const float* mel_taps;  // [32][16]
    const int* mel_off;     // [32]  start of mel bin m's segment in a frame's compact power table (residues mod 32 pairwise different)
    const unsigned* mel_dst;// [120] the (at most two) table slots power bin i feeds: lo / hi 16 bits (250 = none)
    float* mel_out;         // optional [S][8][32]: keep the rows for oww_get_mel (debug handles)
};

// LDS layout (floats)
constexpr int FA_W0 = 2 * 256, FA_W1 = 12 * 256, FA_W2 = 14 * 256, FA_BN = 3 * 2 * 32, FA_MT = owh::sa::WAVE_HALVES / 2, FA_Z = 2 * 576, FA_GT = 512;
constexpr int FA_OFF_W0 = 0, FA_OFF_W1 = FA_OFF_W0 + FA_W0, FA_OFF_W2 = FA_OFF_W1 + FA_W1, FA_OFF_BN = FA_OFF_W2 + FA_W2;
constexpr int FA_OFF_HANN = FA_OFF_BN + FA_BN, FA_OFF_TW1 = FA_OFF_HANN + 512, FA_OFF_TW2 = FA_OFF_TW1 + 2 * 8 * 64;
constexpr int FA_OFF_TAPS = FA_OFF_TW2 + 2 * 8 * 8, FA_OFF_MS = FA_OFF_TAPS + 16 * 32, FA_OFF_DST = FA_OFF_MS + 32, FA_OFF_GT = FA_OFF_DST + 128, FA_OFF_MEL = FA_OFF_GT + FA_GT;
constexpr int FA_OFF_Z = FA_OFF_MEL + FA_WG * FA_MT + ((4 - (FA_WG * FA_MT) % 4) % 4);
constexpr int FA_LDS_BYTES = (FA_OFF_Z + FA_WG * FA_Z) * 4;
static_assert(FA_OFF_Z % 4 == 0 && FA_OFF_W1 % 4 == 0 && FA_OFF_W2 % 4 == 0 && FA_OFF_GT % 4 == 0 && FA_OFF_MEL % 4 == 0 && FA_MT % 4 == 0, "16-byte aligned blocks");
static_assert(owk::MEL_WX <= FA_Z, "sample window fits the transpose planes");

// the 672 (+8) samples of pass f2 (frames 2 f2, 2 f2 + 1) of [tail(480) ; pcm(1280)] as raw int16 (lane l: samples 8l.., 512 + 8l..):
// every piece of 8 samples lies wholly in the tail or wholly in the chunk, and both rows are 16-byte aligned (the host checks)
__device__ __forceinline__ void fetch_pass(const int16_t* __restrict__ tail_row, const int16_t* __restrict__ pcm_row, int f2, int lane,
                                           int4 (&raw)[2]) {
#pragma unroll
    for (int u = 0; u < 2; ++u) {
        const int i = lane * 8 + u * 512;
        if (i < owk::MEL_WX) {
            const int c = 320 * f2 + i;
            raw[u] = c < 480 ? *reinterpret_cast<const int4*>(tail_row + c) : *reinterpret_cast<const int4*>(pcm_row + (c - 480));
        }
    }
}

template <bool DBG>
__global__ __launch_bounds__(64 * FA_WG, (FA_WG + 3) / 4) void hmelA_kernel(MelAParams q) {
    using namespace owr;
    using owk::dft8;
    using owk::wave_sync;
    extern __shared__ __attribute__((aligned(16))) float fl[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int gw = blockIdx.x * FA_WG + wave, nw = gridDim.x * FA_WG;
    constexpr int NT = 64 * FA_WG;
    float* sW0 = fl + FA_OFF_W0;
    float* sW1 = fl + FA_OFF_W1;
    float* sW2 = fl + FA_OFF_W2;
    float* sbn = fl + FA_OFF_BN;
    float* t_hann = fl + FA_OFF_HANN;
    float* t_tw1 = fl + FA_OFF_TW1;
    float* t_tw2 = fl + FA_OFF_TW2;
    float* t_taps = fl + FA_OFF_TAPS;
    int* s_ms = reinterpret_cast<int*>(fl + FA_OFF_MS);
    for (int i = tid; i < FA_W0 / 4; i += NT) reinterpret_cast<f32x4*>(sW0)[i] = reinterpret_cast<const f32x4*>(q.a.w0)[i];
    for (int i = tid; i < FA_W1 / 4; i += NT) reinterpret_cast<f32x4*>(sW1)[i] = reinterpret_cast<const f32x4*>(q.a.w1)[i];
    for (int i = tid; i < FA_W2 / 4; i += NT) reinterpret_cast<f32x4*>(sW2)[i] = reinterpret_cast<const f32x4*>(q.a.w2)[i];
    if (tid < 96) {
        const int l = tid / 32, c = tid % 32;
        sbn[(l * 2 + 0) * 32 + c] = q.a.scale[l][c];
        sbn[(l * 2 + 1) * 32 + c] = q.a.shift[l][c];
    }
    for (int i = tid; i < 512; i += NT) t_hann[i] = (i >= 56 && i < 456) ? q.hann[i - 56] : 0.f;      // Hann(400) centred in the 512-point frame
    for (int i = tid; i < 512; i += NT) {
        float sn, cs;
        sincospif(-(float)((i & 63) * (i >> 6)) / 256.f, &sn, &cs);
        t_tw1[i] = cs; t_tw1[512 + i] = sn;
    }
    if (tid < 64) {
        float sn, cs;
        sincospif(-(float)((tid & 7) * (tid >> 3)) / 32.f, &sn, &cs);
        t_tw2[tid] = cs; t_tw2[64 + tid] = sn;
    }
    for (int i = tid; i < 512; i += NT) t_taps[i] = q.mel_taps[(i & 31) * 16 + (i >> 5)];
    if (tid < 32) s_ms[tid] = OWF_COMPACT_TAPS ? q.mel_off[tid] : q.mel_start[tid] - 2;
    if (tid < 128) reinterpret_cast<unsigned*>(fl + FA_OFF_DST)[tid] = q.mel_dst[tid];
    for (int i = tid; i < FA_WG * FA_MT; i += NT) fl[FA_OFF_MEL + i] = 0.f;      // (hi / lo planes of every wave: zero columns = the mel-axis padding)
    owh::stageA_fill_gather_table(reinterpret_cast<int*>(fl + FA_OFF_GT), tid, NT);
    // the FFT planes start finite: a tap read past the end of a mel bin's segment (against a zero tap) may fall into a gap of the
    // compact power tables that no FFT stage has written yet, and 0 x NaN would poison the sum
    for (int i = tid; i < FA_WG * FA_Z; i += NT) fl[FA_OFF_Z + i] = 0.f;
    __syncthreads();
    _Float16* sP = reinterpret_cast<_Float16*>(fl + FA_OFF_MEL + wave * FA_MT);
    const int* gtab = reinterpret_cast<const int*>(fl + FA_OFF_GT);
    float* const planes = fl + FA_OFF_Z + wave * FA_Z;
    lanemask_t bad = 0;


    for (int s0 = gw; s0 < q.a.n_streams; s0 += nw) {
        // the stream index as an opaque scalar: row addresses are then formed per iteration as SGPR base + lane offset instead of
        // strength-reduced 64-bit per-lane pointers that would stay alive (and spill) across the whole loop body
        int s = __builtin_amdgcn_readfirstlane(s0 + q.a.s_base);
        asm volatile("" : "+s"(s));
#if OWF_PRIO == 4
        {   // (variant: the three waves of a SIMD take turns at the top priority, one stream each)
            const int turn = __builtin_amdgcn_readfirstlane(((s0 - gw) / nw + (wave >> 2)) % 3);
            if (turn == 0) __builtin_amdgcn_s_setprio(0);
            else if (turn == 1) __builtin_amdgcn_s_setprio(1);
            else __builtin_amdgcn_s_setprio(2);
        }
#endif
        if (q.a.stream_on && !q.a.stream_on[s]) continue;        // masked step (oww_step_masked): no sample is consumed, no state touched

        // an opaque zero added to every LDS base of the mel phase: its per-lane addresses are then recomputed in each iteration
        // (a few VALU adds) instead of being hoisted out of the stream loop and kept alive -- i.e. spilled -- across stage A
        int z = 0;
        asm volatile("" : "+s"(z));
        const int lane_all = lane;
        const int lane = lane_all + z;            // (the mel phase's lane-derived offsets are iteration-local for the same reason)
        float* xr = planes + z;                   // FFT planes; the sample window and the two power rows alias them (owk::mel_kernel)
        float* xi = xr + 576;
        float* sx = xr;
        // compact power tables of the pass's two frames (256 words each: the words 128..383 of either plane, which the last FFT
        // stage leaves unused): one segment per mel bin, segment starts on 32 different banks -- the tap reads below are conflict-free
        float* pw0 = xr + 128;
        float* pw1 = OWF_COMPACT_TAPS ? xi + 128 : xr + 257;
        const unsigned* s_dst = reinterpret_cast<const unsigned*>(fl + FA_OFF_DST + z);
        const float* s_hann = fl + FA_OFF_HANN + z;
        const float* s_tw1 = fl + FA_OFF_TW1 + z;   // [re / im][k 8][lane 64]: exp(-2 pi i lane k / 512)
        const float* s_tw2 = fl + FA_OFF_TW2 + z;   // [re / im][k 8][m0 8]:    exp(-2 pi i m0 k / 64)
        const float* s_taps = fl + FA_OFF_TAPS + z; // [tap 16][mel bin 32]
        const int mbin = lane & 31, hf = lane >> 5;   // this lane's output in every pass: frame 2 pass + hf, mel bin mbin
        const int mstart = s_ms[mbin + z];
        const bool first = q.nfeat[s] == 0;
        float db[4];
        float vmax = -INFINITY;
        int4 raw[2] = {};
        const int16_t* tail_row = q.tail + (size_t)s * 480;
        const int16_t* pcm_row = q.pcm + (size_t)s * 1280;
        fetch_pass(tail_row, pcm_row, 0, lane, raw);
#pragma unroll
        for (int f2 = 0; f2 < 4; ++f2) {
#if OWF_PRIO == 6
            {   // the level rises with the stream-step's progress: FFT passes here, row pairs of the matrix phase in hstageA_stream
                constexpr int code = OWF_MPRIO;
                if (f2 == 0) __builtin_amdgcn_s_setprio((code / 1000) % 10);
                else if (f2 == 1) __builtin_amdgcn_s_setprio((code / 100) % 10);
                else if (f2 == 2) __builtin_amdgcn_s_setprio((code / 10) % 10);
                else __builtin_amdgcn_s_setprio(code % 10);
            }
#endif
            wave_sync();                         // the previous pass's readers of the planes / power rows are done (same wave)
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                const int i = lane * 8 + u * 512;
                if (i < owk::MEL_WX) {
                    // two 16-byte stores per lane (2-way bank conflicts) instead of the eight scalar ones the compiler pairs into
                    // ds_write2_b32 at a stride of 8 floats (8-way conflicts: a quarter of this kernel's LDS-active cycles).
                    // (staging the raw int16, or a transposed [8][89] float window, measured slower: round 2)
                    const int16_t* h = reinterpret_cast<const int16_t*>(&raw[u]);
                    *reinterpret_cast<f32x4*>(sx + i) = f32x4{(float)h[0], (float)h[1], (float)h[2], (float)h[3]};
                    *reinterpret_cast<f32x4*>(sx + i + 4) = f32x4{(float)h[4], (float)h[5], (float)h[6], (float)h[7]};
                }
            }
            if (f2 < 3) fetch_pass(tail_row, pcm_row, f2 + 1, lane, raw);     // next pass's samples fly during this FFT
            wave_sync();
            float re[8], im[8];
#pragma unroll
            for (int n2 = 0; n2 < 8; ++n2) {
                const int n = 64 * n2 + lane;
                const float w = s_hann[n];               // (zero outside [56, 456): the table is padded to the frame)
                re[n2] = w * sx[n];
                im[n2] = w * sx[160 + n];
            }
            dft8(re, im);
#pragma unroll
            for (int k = 1; k < 8; ++k) {
                const float cr = s_tw1[k * 64 + lane], ci = s_tw1[512 + k * 64 + lane];
                const float r = re[k] * cr - im[k] * ci;
                im[k] = re[k] * ci + im[k] * cr;
                re[k] = r;
            }
            wave_sync();                         // every lane has read its samples: the planes may be overwritten
#pragma unroll
            for (int k = 0; k < 8; ++k) { xr[k * 72 + lane] = re[k]; xi[k * 72 + lane] = im[k]; }
            wave_sync();
            {
                const int qq = lane >> 3, m0 = lane & 7;
#pragma unroll
                for (int m1 = 0; m1 < 8; ++m1) { re[m1] = xr[qq * 72 + m1 * 8 + m0]; im[m1] = xi[qq * 72 + m1 * 8 + m0]; }
            }
            dft8(re, im);
#pragma unroll
            for (int k = 1; k < 8; ++k) {
                const float cr = s_tw2[k * 8 + (lane & 7)], ci = s_tw2[64 + k * 8 + (lane & 7)];
                const float r = re[k] * cr - im[k] * ci;
                im[k] = re[k] * ci + im[k] * cr;
                re[k] = r;
            }
            wave_sync();
            {
                const int qq = lane >> 3, m0 = lane & 7;
#pragma unroll
                for (int k1 = 0; k1 < 8; ++k1) { xr[(qq * 8 + k1) * 9 + m0] = re[k1]; xi[(qq * 8 + k1) * 9 + m0] = im[k1]; }
            }
            wave_sync();
#pragma unroll
            for (int m0 = 0; m0 < 8; ++m0) { re[m0] = xr[lane * 9 + m0]; im[m0] = xi[lane * 9 + m0]; }
            dft8(re, im);
            wave_sync();
            {
                // lane = k0*8 + k1 holds Z[k0 + 8*k1 + 64*k2], k2 = 0..7; only k < 128 and k >= 384 are needed
                const int kb = (lane >> 3) + 8 * (lane & 7);
                xr[kb] = re[0];        xi[kb] = im[0];
                xr[kb + 64] = re[1];   xi[kb + 64] = im[1];
                xr[kb + 384] = re[6];  xi[kb + 384] = im[6];
                xr[kb + 448] = re[7];  xi[kb + 448] = im[7];
            }
            wave_sync();
            // (same loop as owk::mel_kernel: the power rows alias xr[128..383], which the last FFT stage leaves unused.  A variant that
            //  first collected all four powers in registers and wrote them after one more ordering point produced rare wrong top-bin
            //  values on the GPU -- the compiler had packed it into v_pk_*_f32 under exec masks -- so this stays as it is)
            for (int i = lane; i < owk::MEL_PBINS; i += 64) {
                const int k = i + 2;
                const float zr = xr[k], zi = xi[k], yr = xr[512 - k], yi = xi[512 - k];
                const float ar = zr + yr, ai = zi - yi;
                const float br = zi + yi, bi = zr - yr;
                const float p0 = 0.25f * (ar * ar + ai * ai), p1 = 0.25f * (br * br + bi * bi);
#if OWF_COMPACT_TAPS
                const unsigned d = s_dst[i];                  // every FFT bin feeds at most two (neighbouring) triangular filters
                pw0[d & 0xffffu] = p0; pw0[d >> 16] = p0;
                pw1[d & 0xffffu] = p1; pw1[d >> 16] = p1;
#else
                pw0[i] = p0; pw1[i] = p1;
#endif
            }
            wave_sync();
            {
                float acc = 0.f;
                const float* pw = hf ? pw1 : pw0;
#pragma unroll
                for (int t = 0; t < 16; ++t) acc = fmaf(pw[mstart + t], s_taps[t * 32 + mbin], acc);
                float d = owk::db10(acc);
                const bool masked = first && (2 * f2 + hf) < 3;
                if (!masked) vmax = fmaxf(vmax, d);
                db[f2] = masked ? INFINITY : d;
            }
        }
        // ---- clamp floor of the call (utils.py:202: the maximum over everything this call produced) and the host transform
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) vmax = fmaxf(vmax, __shfl_xor(vmax, o));
        const float floor_db = vmax - 80.0f;
#pragma unroll
        for (int f2 = 0; f2 < 4; ++f2) {
            const float v = (db[f2] == INFINITY) ? 1.0f : owk::mel_units(db[f2], floor_db);
            owh::stageA_put_mel(sP, 2 + 2 * f2 + hf, mbin, v);       // rows 2..9 of the wave's planes, as (hi, lo) f16 halves
            if (q.mel_out) q.mel_out[((size_t)s * 8 + 2 * f2 + hf) * 32 + mbin] = v;
        }
        // new 480-sample tail = the last 480 samples of the chunk (every tail read of this step has completed: its data was used)
        if (lane < 60) *reinterpret_cast<int4*>(q.tail + (size_t)s * 480 + lane * 8) =
                           *reinterpret_cast<const int4*>(q.pcm + (size_t)s * 1280 + 800 + lane * 8);
        wave_sync();
        __builtin_amdgcn_sched_barrier(0);       // the mel phase ends here: none of its values stays live into stage A
#if OWF_PRIO == 1
        __builtin_amdgcn_s_setprio(OWF_PRIO_HI); // the matrix phase outranks the other waves' log-mel phases on this SIMD
#elif OWF_PRIO == 2
        __builtin_amdgcn_s_setprio(0);
#endif
        owh::hstageA_stream<DBG, true>(q.a, s, sP, sW0, sW1, sW2, sbn, gtab, bad, lane_all);
#if OWF_PRIO == 1
        __builtin_amdgcn_s_setprio(0);
#elif OWF_PRIO == 2
        __builtin_amdgcn_s_setprio(OWF_PRIO_HI); // (variant, inverse: the log-mel phase outranks)
#endif
        if (bad) { owh::raise_range_flag(bad, q.a.range_flag, s, 1); bad = 0; }
        __builtin_amdgcn_sched_barrier(0);
    }
    owh::raise_range_flag(bad, q.a.range_flag);
}

}  // namespace owf
