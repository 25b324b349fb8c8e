// owwhip_vad.h -- voice-activity network on the device (included by owwhip.hip): row I / K5 of SURVEY section 8.
//
// The reference runs Silero's silero_vad.onnx per stream through onnxruntime (openwakeword/vad.py:98-130): every predict()
// splits the frame into 640-sample sub-frames, scales by 1/32767, calls the network with the carried state h, c [2, 1, 64],
// and appends the mean sub-frame score to a 125-deep ring that gates the wake-word scores (model.py:366-381).  The graph of
// that file is not in the reference checkout, so what runs here is a STRUCTURAL STAND-IN with the same interface and state
// (numpy restatement: oracle/vad_standin.py; weights: openwakeword_amd/weights.py synthetic_vad):
//
//   vad_front_kernel   per stream-step: 14 Hann(256)/hop-64 frames of the two 640-sample sub-frames -> |STFT| of bins 1..128 by
//                      radix-4 x 8 x 8 complex FFTs (two complex 256-point FFTs = four real frames per wave pass, through the
//                      wave's own LDS planes, no workgroup barrier) -> log(1 + 50 |X|) -> four Conv1d(k=3)+ReLU encoder layers as
//                      fp16-split MFMAs (positions of the 16x16 tile = the 2 x 7 frames; strides are evaluated as dilations, so
//                      nothing is compacted) -> the 64-channel LSTM inputs of the 2 x 2 remaining time steps, scattered into the
//                      16-stream tiles the LSTM kernel consumes.  Encoder weights (72 KB) stay in LDS for the persistent launch.
//   vad_lstm_kernel    per 16 streams: 2-layer LSTM(64) over the 4 time steps of the step (state h, c in HBM, register-dump
//                      order), gates as fp16-split MFMAs with the weights streamed L2 -> LDS in 16 KB chunks shared by the
//                      workgroup (same scheme as the CNN stages), sigmoid decoder, mean -> the stream's VAD ring (postproc_kernel
//                      applies the gate).
#pragma once
#include "owwhip_hx.h"

namespace owv {

using owh::f16x8;
using owh::lanemask_t;
using owh::Op;
using owr::f32x4;

#ifndef OWV_WG
#define OWV_WG 12
#endif
constexpr int V_WG = OWV_WG;              // waves per workgroup of the front kernel (12: one workgroup per CU, 3 waves per SIMD)
constexpr int V_FEAT_STRIDE = 144;        // floats per staged feature row (128 bins + 16: the gather of 16 lanes x 16 B is conflict-free)
// encoder weight blocks of 1 KB in LDS: layer l = [oct][tap][ks][part]
constexpr int V_BLK1 = 1 * 3 * 4 * 2, V_BLK2 = 2 * 3 * 1 * 2, V_BLK3 = 2 * 3 * 1 * 2, V_BLK4 = 4 * 3 * 1 * 2;
constexpr int V_WFLOATS = (V_BLK1 + V_BLK2 + V_BLK3 + V_BLK4) * 256;
constexpr int V_WAVE_FLOATS = 640 + 2 * 576;       // samples (1280 int16) + FFT planes (the feature staging rows alias the re plane)
static_assert(4 * V_FEAT_STRIDE <= 576, "feature rows fit the re plane");
constexpr int V_LDS_BYTES = (V_WFLOATS + 4 * 64 + 256 + V_WG * V_WAVE_FLOATS) * 4;

struct VadFrontParams {
    const int16_t* pcm;     // [S][n_samples], the step's new samples (the first 1280 are used)
    int n_samples, S;
    const float* hann;      // [256]
    float mag_gain;
    const float* w;         // hx-packed encoder weights, layers back to back (V_WFLOATS floats)
    const float* bias;      // [4][64], zero padded
    float* xout;            // [ceil(S/16)][4 (sub-frame, time)][16 registers][64 lanes]: LSTM input tiles
    int* range_flag;
    const uint8_t* stream_on;   // oww_step_masked: [S] 1 = the stream takes part in this step; nullptr = all do
};

template <int D> __device__ __forceinline__ float dpp_shr_zero(float x) {       // lane p <- x[p - D] inside the 16-lane row, else 0
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), 0x110 + D, 0xf, 0xf, true));
}
template <int D> __device__ __forceinline__ float dpp_shl_zero(float x) {       // lane p <- x[p + D]
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), 0x100 + D, 0xf, 0xf, true));
}

// Conv1d(k = 3, zero padding) + bias + ReLU over the time axis = the 16 positions of the tile (two sub-frames of 8 slots, the valid
// ones being 0, D0, 2 D0 ... MAXPOS), evaluated at every position with the neighbours D lanes away: out[p] = w0 x[p-D] + w1 x[p] +
// w2 x[p+D].  Three per-tap accumulator chains on the unshifted input, combined with two DPP row shifts (cf. owh::conv_mel_hx).
template <int KSI, int NCTO, int D, int MAXPOS>
__device__ __forceinline__ void conv_t(const Op (&in)[KSI], f32x4 (&out)[NCTO], const float* w, const float* bias, int lane, lanemask_t& bad) {
    const int pos = lane & 15, j = lane >> 4, t = pos & 7;
    const bool lo_ok = t >= D, hi_ok = t + D <= MAXPOS;
#pragma unroll
    for (int oct = 0; oct < NCTO; ++oct) {
        f32x4 acc[3];
#pragma unroll
        for (int tap = 0; tap < 3; ++tap) acc[tap] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int ks = 0; ks < KSI; ++ks)
#pragma unroll
            for (int part = 0; part < 3; ++part)
#pragma unroll
                for (int tap = 0; tap < 3; ++tap) {
                    const f16x8 a = owh::lds_h(w, ((oct * 3 + tap) * KSI + ks) * 2 + (part == 2 ? 1 : 0), lane);
                    acc[tap] = OWH_MFMA(a, part == 1 ? in[ks].l : in[ks].h, acc[tap]);
                }
        if (oct == 0) owh::nan_guard(bad, acc[1][0]);
        const f32x4 b = *reinterpret_cast<const f32x4*>(bias + oct * 16 + 4 * j);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const float l = dpp_shr_zero<D>(acc[0][e]), h = dpp_shl_zero<D>(acc[2][e]);
            const float v = acc[1][e] + (lo_ok ? l : 0.f) + (hi_ok ? h : 0.f);
            out[oct][e] = owr::fmax_nc(fmaf(v, owh::WUNSCALE, b[e]), 0.f);
        }
        owr::pin(out[oct]);
    }
}

// the 1280 samples of stream s's step as raw int16 in registers (lane l: samples 8l.., 512 + 8l.., and 1024 + 8l.. for l < 32)
__device__ __forceinline__ void vad_fetch(const VadFrontParams& p, int s, int lane, int4 (&raw)[3]) {
    const int16_t* src = p.pcm + (size_t)s * p.n_samples;
#pragma unroll
    for (int u = 0; u < 3; ++u) {
        const int i = lane * 8 + u * 512;
        if (i >= 1280) continue;
        if ((reinterpret_cast<uintptr_t>(src + i) & 15) == 0) raw[u] = *reinterpret_cast<const int4*>(src + i);
        else {
            alignas(16) int16_t h[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) h[e] = src[i + e];
            raw[u] = *reinterpret_cast<const int4*>(h);
        }
    }
}

__global__ __launch_bounds__(64 * V_WG, (V_WG + 3) / 4) void vad_front_kernel(VadFrontParams p) {
    using owk::dft8;
    using owk::wave_sync;
    extern __shared__ __attribute__((aligned(16))) float vlds[];
    float* sW = vlds;                               // encoder weights, operand order
    float* sB = sW + V_WFLOATS;                     // biases [4][64]
    float* sHann = sB + 4 * 64;
    const int tid = threadIdx.x, lane = tid & 63, pos = lane & 15, j = lane >> 4;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    float* mine = sHann + 256 + wave * V_WAVE_FLOATS;
    int16_t* sx = reinterpret_cast<int16_t*>(mine);                   // 1280 samples
    float* xr = mine + 640;                                         // FFT planes (re, im), 576 floats each
    float* xi = xr + 576;
    float* sF = xr;                                                 // features of the pass's four frames [4][V_FEAT_STRIDE]: written over the
                                                                    // re plane once every lane has read its spectrum values
    for (int i = tid; i < V_WFLOATS / 4; i += 64 * V_WG) reinterpret_cast<f32x4*>(sW)[i] = reinterpret_cast<const f32x4*>(p.w)[i];
    for (int i = tid; i < 4 * 64; i += 64 * V_WG) sB[i] = p.bias[i];
    for (int i = tid; i < 256; i += 64 * V_WG) sHann[i] = p.hann[i];
    // per-lane twiddles: exp(-2 pi i lane k / 256) (first, radix-4 stage), exp(-2 pi i (lane & 7) k / 64) (second stage)
    float tw1r[4], tw1i[4], tw2r[8], tw2i[8];
#pragma unroll
    for (int k = 0; k < 4; ++k) sincospif(-(float)(lane * k) / 128.f, &tw1i[k], &tw1r[k]);
#pragma unroll
    for (int k = 0; k < 8; ++k) sincospif(-(float)((lane & 7) * k) / 32.f, &tw2i[k], &tw2r[k]);
    __syncthreads();
    float hw[4];                                     // the window at this lane's four points n = 64 a + lane, times 1 / 32767
#pragma unroll
    for (int a = 0; a < 4; ++a) hw[a] = sHann[64 * a + lane] * (1.0f / 32767.0f);
    const float* w1 = sW;
    const float* w2 = w1 + V_BLK1 * 256;
    const float* w3 = w2 + V_BLK2 * 256;
    const float* w4 = w3 + V_BLK3 * 256;
    lanemask_t bad = 0;

    const int gw = blockIdx.x * V_WG + wave, nw = gridDim.x * V_WG;
    int4 raw[3] = {};
    if (gw < p.S) vad_fetch(p, gw, lane, raw);
    for (int s = gw; s < p.S; s += nw) {
        if (p.stream_on && !p.stream_on[s]) {        // sits this step out (its LSTM lanes store nothing either); keep the prefetch chain going
            if (s + nw < p.S) vad_fetch(p, s + nw, lane, raw);
            continue;
        }
        wave_sync();                                 // the previous stream's readers of sx are done (same wave)
#pragma unroll
        for (int u = 0; u < 3; ++u) {
            const int i = lane * 8 + u * 512;
            if (i < 1280) *reinterpret_cast<int4*>(sx + i) = raw[u];
        }
        if (s + nw < p.S) vad_fetch(p, s + nw, lane, raw);          // next stream's samples fly during this one's FFTs
        wave_sync();

        Op X[4];                                     // encoder input: lane (position, g) holds its frame's 128 features as 4 k-steps
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) { X[ks].h = f16x8{}; X[ks].l = f16x8{}; }
#pragma unroll
        for (int q = 0; q < 4; ++q) {                // pass q: tile positions 4q .. 4q+3 (position = 8 * sub-frame + frame)
            float re[8], im[8];                      // rows 0..3: FFT A (positions 4q, 4q+1), rows 4..7: FFT B (4q+2, 4q+3)
#pragma unroll
            for (int f = 0; f < 2; ++f) {
                const int pr = 4 * q + 2 * f, pi = pr + 1;
                const int offr = 640 * (pr >> 3) + 64 * (pr & 7);
                const bool vi = (pi & 7) < 7;                       // slot 7 of a sub-frame is padding: a zero frame
                const int offi = vi ? 640 * (pi >> 3) + 64 * (pi & 7) : offr;
                float zr[4], zi[4];
#pragma unroll
                for (int a = 0; a < 4; ++a) {
                    zr[a] = hw[a] * (float)sx[offr + 64 * a + lane];
                    zi[a] = vi ? hw[a] * (float)sx[offi + 64 * a + lane] : 0.f;
                }
                // radix-4 over a: y[k1] = sum_a z[a] (-i)^(a k1)
                const float t0r = zr[0] + zr[2], t0i = zi[0] + zi[2], t1r = zr[0] - zr[2], t1i = zi[0] - zi[2];
                const float t2r = zr[1] + zr[3], t2i = zi[1] + zi[3], t3r = zr[1] - zr[3], t3i = zi[1] - zi[3];
                float yr[4], yi[4];
                yr[0] = t0r + t2r; yi[0] = t0i + t2i;
                yr[2] = t0r - t2r; yi[2] = t0i - t2i;
                yr[1] = t1r + t3i; yi[1] = t1i - t3r;
                yr[3] = t1r - t3i; yi[3] = t1i + t3r;
                re[4 * f] = yr[0]; im[4 * f] = yi[0];
#pragma unroll
                for (int k = 1; k < 4; ++k) {
                    re[4 * f + k] = yr[k] * tw1r[k] - yi[k] * tw1i[k];
                    im[4 * f + k] = yr[k] * tw1i[k] + yi[k] * tw1r[k];
                }
            }
            wave_sync();                             // the previous pass's readers of the planes are done
            // eight rows of 64 values (one per lane): a 64-point DFT along each row, as 8 x 8 through two LDS transposes
#pragma unroll
            for (int k = 0; k < 8; ++k) { xr[k * 72 + lane] = re[k]; xi[k * 72 + lane] = im[k]; }
            wave_sync();
            {
                const int r = lane >> 3, m0 = lane & 7;
#pragma unroll
                for (int m1 = 0; m1 < 8; ++m1) { re[m1] = xr[r * 72 + m1 * 8 + m0]; im[m1] = xi[r * 72 + m1 * 8 + m0]; }
            }
            dft8(re, im);
#pragma unroll
            for (int k = 1; k < 8; ++k) {
                const float t = re[k] * tw2r[k] - im[k] * tw2i[k];
                im[k] = re[k] * tw2i[k] + im[k] * tw2r[k];
                re[k] = t;
            }
            wave_sync();
            {
                const int r = lane >> 3, m0 = lane & 7;
#pragma unroll
                for (int k1 = 0; k1 < 8; ++k1) { xr[(r * 8 + k1) * 9 + m0] = re[k1]; xi[(r * 8 + k1) * 9 + m0] = im[k1]; }
            }
            wave_sync();
#pragma unroll
            for (int m0 = 0; m0 < 8; ++m0) { re[m0] = xr[lane * 9 + m0]; im[m0] = xi[lane * 9 + m0]; }
            dft8(re, im);
            wave_sync();
            {
                // lane (row r, kappa1 = lane & 7), register kappa2 holds Z_fft[(r & 3) + 4 kappa1 + 32 kappa2], fft = r >> 2
                const int r = lane >> 3;
                const int base = (r >> 2) * 256 + (r & 3) + 4 * (lane & 7);
#pragma unroll
                for (int k2 = 0; k2 < 8; ++k2) { xr[base + 32 * k2] = re[k2]; xi[base + 32 * k2] = im[k2]; }
            }
            wave_sync();
            // two real spectra per complex FFT: |A[k]|, |B[k]| for k = 1..128, compressed; 4 frames x 128 bins over 64 lanes
            float fa[4], fb[4];
#pragma unroll
            for (int it = 0; it < 4; ++it) {
                const int idx = it * 64 + lane, f = idx >> 7, kb = idx & 127, k = kb + 1;
                const float zr_ = xr[f * 256 + k], zi_ = xi[f * 256 + k], yr_ = xr[f * 256 + 256 - k], yi_ = xi[f * 256 + 256 - k];
                const float ar = zr_ + yr_, ai = zi_ - yi_, br = zi_ + yi_, bi = zr_ - yr_;
                // (hardware v_sqrt_f32 / v_log_f32, 1 ulp each: the IEEE-exact sqrtf and the full logf expand to ~10 VALU instructions
                //  apiece here -- a third of this kernel's instruction count -- for digits the 1e-4 score tolerance never sees)
                fa[it] = __builtin_amdgcn_logf(1.0f + p.mag_gain * 0.5f * __builtin_amdgcn_sqrtf(ar * ar + ai * ai)) * 0.69314718055994531f;
                fb[it] = __builtin_amdgcn_logf(1.0f + p.mag_gain * 0.5f * __builtin_amdgcn_sqrtf(br * br + bi * bi)) * 0.69314718055994531f;
            }
            wave_sync();                             // every spectrum value has been read: the re plane becomes the staging rows
#pragma unroll
            for (int it = 0; it < 4; ++it) {
                const int idx = it * 64 + lane, f = idx >> 7, kb = idx & 127;
                sF[(2 * f) * V_FEAT_STRIDE + kb] = fa[it];
                sF[(2 * f + 1) * V_FEAT_STRIDE + kb] = fb[it];
            }
            wave_sync();
            if ((pos >> 2) == q) {                   // the 16 lanes whose tile position is one of this pass's frames take their operands
                const float* fr = sF + (pos & 3) * V_FEAT_STRIDE + 4 * j;
#pragma unroll
                for (int ks = 0; ks < 4; ++ks) {
                    const f32x4 a = *reinterpret_cast<const f32x4*>(fr + 32 * ks);
                    const f32x4 b = *reinterpret_cast<const f32x4*>(fr + 32 * ks + 16);
                    X[ks] = owh::split_pair(a, b);
                }
            }
        }
        // ---- encoder: 128 -> 16 (stride 1) -> 32 (stride 2) -> 32 (stride 2) -> 64 (stride 1), strides as dilations 1, 1, 2, 4
        f32x4 y1[1], y2[2], y3[2], y4[4];
        Op o1[1], o2[1], o3[1];
        conv_t<4, 1, 1, 6>(X, y1, w1, sB, lane, bad);
        owh::to_ops<1>(y1, o1);
        conv_t<1, 2, 1, 6>(o1, y2, w2, sB + 64, lane, bad);
        owh::to_ops<2>(y2, o2);
        conv_t<1, 2, 2, 6>(o2, y3, w3, sB + 128, lane, bad);
        owh::to_ops<2>(y3, o3);
        conv_t<1, 4, 4, 4>(o3, y4, w4, sB + 192, lane, bad);
        if ((pos & 3) == 0) {                        // positions 0, 4, 8, 12 = (sub-frame, time) 0..3
            float* xo = p.xout + ((size_t)(s >> 4) * 4 + (pos >> 2)) * 1024 + j * 16 + (s & 15);
#pragma unroll
            for (int u = 0; u < 4; ++u)
#pragma unroll
                for (int e = 0; e < 4; ++e) xo[(u * 4 + e) * 64] = y4[u][e];
        }
    }
    owh::raise_range_flag(bad, p.range_flag);
}

// ------------------------------------------------------------------------------------------------------------------------------
struct VadLstmParams {
    const float* xin;       // [G][4][16][64]
    float* hc;              // [G][4 (h1, c1, h2, c2)][16][64]: recurrent state, register-dump order
    const float* w;         // both layers: [layer][hidden tile u][gate i,f,g,o][ks 4][part 2] blocks of 1 KB (hx operand order)
    const float* bias;      // [2][256], PyTorch gate order
    const float* wd;        // [64] decoder
    float bd;
    float* ring;            // [S][8] VAD score ring (postproc_kernel)
    uint32_t* n_vad;        // [S]
    float* last;            // [S] the score just pushed
    int S, n_groups;
    const uint8_t* stream_on;   // see VadFrontParams::stream_on
    const int* glist;           // see owr::RStageParams::glist (groups of 16 streams)
};

constexpr int L_WG = 4;                       // waves per workgroup: they share one weight chunk stream
constexpr int L_CHUNK_BLOCKS = 2 * 4 * 2;     // two gate tiles x 4 k-steps x hi/lo
constexpr int L_CHUNK = L_CHUNK_BLOCKS * 256; // floats

// (v_rcp_f32, 1 ulp: a true fp32 division is ~10 VALU instructions here -- 58 % of this kernel's instruction count before)
__device__ __forceinline__ float sigm(float x) { return __builtin_amdgcn_rcpf(1.0f + __expf(-x)); }
__device__ __forceinline__ float tanh_fast(float x) { return fmaf(2.0f, __builtin_amdgcn_rcpf(1.0f + __expf(-2.0f * x)), -1.0f); }

__global__ __launch_bounds__(64 * L_WG, 2) void vad_lstm_kernel(VadLstmParams p) {
    using namespace owr;
    __shared__ __attribute__((aligned(16))) float wbuf[L_CHUNK];          // (two distinct LDS objects: see owh::hstage_kernel)
    __shared__ __attribute__((aligned(16))) float wbuf1[L_CHUNK];
    __shared__ __attribute__((aligned(16))) float sb[2 * 256];
    __shared__ __attribute__((aligned(16))) float swd[64];
    const int lane = threadIdx.x & 63, pos = lane & 15, j = lane >> 4;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    int g = blockIdx.x * L_WG + wave;
    const bool active = g < p.n_groups;
    if (!active) g = p.n_groups - 1;
    if (p.glist) g = p.glist[g];          // masked step with few participants: only the 16-stream groups that hold one
    issue_chunk<L_CHUNK_BLOCKS, L_WG>(p.w, wbuf, wave, lane);
    for (int i = threadIdx.x; i < 512; i += 64 * L_WG) sb[i] = p.bias[i];
    if (threadIdx.x < 64) swd[threadIdx.x] = p.wd[threadIdx.x];

    float* hc = p.hc + (size_t)g * 4096;
    f32x4 hf[2][4], cf[2][4];
    Op H[2][2];
#pragma unroll
    for (int l = 0; l < 2; ++l) {
        load_tile<4>(hf[l], hc + (2 * l) * 1024, lane);
        load_tile<4>(cf[l], hc + (2 * l + 1) * 1024, lane);
        owh::to_ops<4>(hf[l], H[l]);
    }
    float yacc = 0.f;
    chunk_sync();
#pragma unroll 1
    for (int bt = 0; bt < 4; ++bt) {
        Op Xo[2];
        {
            f32x4 X[4];
            load_tile<4>(X, p.xin + ((size_t)g * 4 + bt) * 1024, lane);
            owh::to_ops<4>(X, Xo);
        }
#pragma unroll
        for (int l = 0; l < 2; ++l) {
            f32x4 hn[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                f32x4 acc[4];
#pragma unroll
                for (int hh = 0; hh < 2; ++hh) {
                    const int cc = (l * 4 + u) * 2 + hh;                    // chunk number inside this time step (16 per step)
                    const float* cur = (cc & 1) ? wbuf1 : wbuf;
                    float* nxt = ((cc + 1) & 1) ? wbuf1 : wbuf;
                    issue_chunk<L_CHUNK_BLOCKS, L_WG>(p.w + (size_t)((cc + 1) & 15) * L_CHUNK, nxt, wave, lane);
#pragma unroll
                    for (int gt = 0; gt < 2; ++gt) {
                        f32x4 a = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                        for (int ks = 0; ks < 4; ++ks) {
                            const Op& b = ks < 2 ? (l == 0 ? Xo[ks] : H[0][ks]) : H[l][ks - 2];
                            const f16x8 ah = owh::lds_h(cur, (gt * 4 + ks) * 2 + 0, lane), al = owh::lds_h(cur, (gt * 4 + ks) * 2 + 1, lane);
                            a = OWH_MFMA(ah, b.h, a);
                            a = OWH_MFMA(ah, b.l, a);
                            a = OWH_MFMA(al, b.h, a);
                        }
                        acc[2 * hh + gt] = a;
                    }
                    chunk_sync();
                }
                // gates of hidden units 16u + 4j + e (register e): i, f, g, o = acc[0..3]
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const int c = 16 * u + 4 * j + e;
                    const float gi = sigm(fmaf(acc[0][e], owh::WUNSCALE, sb[l * 256 + c]));
                    const float gf = sigm(fmaf(acc[1][e], owh::WUNSCALE, sb[l * 256 + 64 + c]));
                    const float gg = tanh_fast(fmaf(acc[2][e], owh::WUNSCALE, sb[l * 256 + 128 + c]));
                    const float go = sigm(fmaf(acc[3][e], owh::WUNSCALE, sb[l * 256 + 192 + c]));
                    const float cn = fmaf(gf, cf[l][u][e], gi * gg);
                    cf[l][u][e] = cn;
                    hn[u][e] = go * tanh_fast(cn);
                }
            }
            // the layer's new h replaces the old one only now: every gate of this time step saw h(t-1)
#pragma unroll
            for (int u = 0; u < 4; ++u) hf[l][u] = hn[u];
            owh::to_ops<4>(hn, H[l]);
        }
        float z = 0.f;
#pragma unroll
        for (int u = 0; u < 4; ++u)
#pragma unroll
            for (int e = 0; e < 4; ++e) z = fmaf(fmaxf(hf[1][u][e], 0.f), swd[16 * u + 4 * j + e], z);
        z = owh::xsum4(z) + p.bd;
        yacc += sigm(z);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");          // drain the chunk prefetched after the last one
    const int s = g * 16 + pos;
    if (active && s < p.S && (p.stream_on == nullptr || p.stream_on[s] != 0)) {      // per lane: position = stream
#pragma unroll
        for (int l = 0; l < 2; ++l) {
            store_tile<4>(hf[l], hc + (2 * l) * 1024, lane);
            store_tile<4>(cf[l], hc + (2 * l + 1) * 1024, lane);
        }
        if (j == 0) {
            const float score = yacc * 0.25f;                  // mean over the 2 sub-frames x 2 time steps (vad.py:127)
            const uint32_t L = p.n_vad[s];
            p.ring[(size_t)s * 8 + (L & 7u)] = score;
            p.n_vad[s] = L + 1u;
            p.last[s] = score;
        }
    }
}

// zero the recurrent state / score ring of the listed streams (ids == nullptr: streams [0, n))
__global__ void vad_reset_kernel(float* hc, float* ring, uint32_t* n_vad, float* last, const int* ids, int n) {
    const int k = blockIdx.x;
    if (k >= n) return;
    const int s = ids ? ids[k] : k;
    float* base = hc + (size_t)(s >> 4) * 4096;
    for (int i = threadIdx.x; i < 4 * 16 * 4; i += blockDim.x) {          // [array 4][register 16][j 4] at position s & 15
        const int a = i >> 6, r = (i >> 2) & 15, jj = i & 3;
        base[a * 1024 + r * 64 + jj * 16 + (s & 15)] = 0.f;
    }
    if (threadIdx.x < 8) ring[(size_t)s * 8 + threadIdx.x] = 0.f;
    if (threadIdx.x == 0) { n_vad[s] = 0u; last[s] = 0.f; }
}

}  // namespace owv
