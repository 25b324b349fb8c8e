// Would stages D / E gain from v_mfma_f32_32x32x16_f16 tiles?  The f16-split time-conv layer body of stage D
// (96 -> 96 channels, two new rows + two history rows, weights in LDS, epilogue = activation + re-split into operands) in both shapes:
//   A  16x16x32: tile = 16 positions (4 streams x 4 mel), a wave carries 2 row tiles = 4 streams   (what hstage_kernel<HD> runs)
//   B  32x32x16: tile = 32 positions (8 streams x 4 mel), a wave carries 2 row tiles = 8 streams
// Same MACs per stream, same LDS weight bytes per output channel; B reads each 1 KB weight block for twice the MFMA cycles and may
// hide two VALU per MFMA (overlap_ubench), but holds 192 operand + 96 result registers per wave instead of 96 + 48, so it runs at one
// wave per SIMD where A runs at three.  Output: ns per stream-layer at each occupancy that compiles without scratch.
// build: hipcc --offload-arch=gfx950 -O3 -std=c++17 tile32_ubench.hip -o tile32_ubench
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

__device__ __forceinline__ float act1(float x, float cl) { return fmaxf(fmaxf(0.2f * x, x), cl); }
// (hi, lo) operand halves of eight fp32 values, as owh::split_pair does it (v_cvt_pk + v_fma_mix)
__device__ __forceinline__ void split8(const float (&x)[8], f16x8& h, f16x8& l) {
    u32x4 hh, ll;
#pragma unroll
    for (int v = 0; v < 4; ++v) {
        const unsigned hp = __builtin_bit_cast(unsigned, f16x2{(_Float16)x[2 * v], (_Float16)x[2 * v + 1]});
        unsigned lp;
        asm("v_fma_mixlo_f16 %0, %1, -1.0, %2 op_sel_hi:[1,0,0]\n\tv_fma_mixhi_f16 %0, %1, -1.0, %3 op_sel:[1,0,0] op_sel_hi:[1,0,0]"
            : "=&v"(lp) : "v"(hp), "v"(x[2 * v]), "v"(x[2 * v + 1]));
        hh[v] = hp; ll[v] = lp;
    }
    h = __builtin_bit_cast(f16x8, hh); l = __builtin_bit_cast(f16x8, ll);
}

// ---- A: 16x16x32, NT = 2 row tiles, 96 input channels = 3 k-steps of 32, NOCT output tiles of 16 channels
template <int NOCT, int WPS>
__global__ __launch_bounds__(256, WPS) void k16(float* out, const float* in, const _Float16* w, int iters) {
    extern __shared__ __attribute__((aligned(16))) _Float16 sw[];       // [oct][tap 3][ks 3][part 2][64][8]
    const int lane = threadIdx.x & 63;
    constexpr int NBLK = NOCT * 3 * 3 * 2;
    for (int i = threadIdx.x; i < NBLK * 64; i += 256) reinterpret_cast<f16x8*>(sw)[i] = reinterpret_cast<const f16x8*>(w)[i];
    __syncthreads();
    f16x8 Bh[4][3], Bl[4][3];                                            // rows h0, h1, x0, x1
#pragma unroll
    for (int r = 0; r < 4; ++r)
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            float x[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) x[e] = in[((r * 3 + k) * 8 + e) * 256 + threadIdx.x];
            split8(x, Bh[r][k], Bl[r][k]);
        }
    for (int it = 0; it < iters; ++it) {
        int z = 0; asm volatile("" : "+s"(z));
        const f16x8* swz = reinterpret_cast<const f16x8*>(sw) + z + lane;
        f32x4 Y[2][6];
#pragma unroll
        for (int oct = 0; oct < 6; ++oct) {                               // all six output tiles (the LDS image holds NOCT of them: reused)
            const int ob = oct % NOCT;
            f32x4 acc[2] = {f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}};
#pragma unroll
            for (int tap = 0; tap < 3; ++tap)
#pragma unroll
                for (int k = 0; k < 3; ++k) {
                    const f16x8 ah = swz[(((ob * 3 + tap) * 3 + k) * 2 + 0) * 64], al = swz[(((ob * 3 + tap) * 3 + k) * 2 + 1) * 64];
#pragma unroll
                    for (int t = 0; t < 2; ++t) acc[t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah, Bh[t + tap][k], acc[t], 0, 0, 0);
#pragma unroll
                    for (int t = 0; t < 2; ++t) acc[t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah, Bl[t + tap][k], acc[t], 0, 0, 0);
#pragma unroll
                    for (int t = 0; t < 2; ++t) acc[t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(al, Bh[t + tap][k], acc[t], 0, 0, 0);
                }
#pragma unroll
            for (int t = 0; t < 2; ++t) {
#pragma unroll
                for (int e = 0; e < 4; ++e) Y[t][oct][e] = act1(acc[t][e], -3.2f);
                float a = Y[t][oct][0], b = Y[t][oct][1], c = Y[t][oct][2], d = Y[t][oct][3];
                asm volatile("" : "+v"(a), "+v"(b), "+v"(c), "+v"(d));
                Y[t][oct] = f32x4{a, b, c, d};
            }
        }
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int k = 0; k < 3; ++k) {
                const float x[8] = {Y[t][2 * k][0], Y[t][2 * k][1], Y[t][2 * k][2], Y[t][2 * k][3], Y[t][2 * k + 1][0], Y[t][2 * k + 1][1], Y[t][2 * k + 1][2], Y[t][2 * k + 1][3]};
                split8(x, Bh[t + 2][k], Bl[t + 2][k]);
            }
    }
    float r = 0.f;
#pragma unroll
    for (int k = 0; k < 3; ++k) r += (float)Bh[2][k][0] + (float)Bl[3][k][7];
    out[blockIdx.x * 256 + threadIdx.x] = r;
}

// ---- B: 32x32x16, NT = 2 row tiles of 32 positions, 96 input channels = 6 k-steps of 16, three output tiles of 32 channels
template <int NOCT, int WPS>
__global__ __launch_bounds__(256, WPS) void k32(float* out, const float* in, const _Float16* w, int iters) {
    extern __shared__ __attribute__((aligned(16))) _Float16 sw[];       // [oct][tap 3][ks 6][part 2][64][8]
    const int lane = threadIdx.x & 63;
    constexpr int NBLK = NOCT * 3 * 6 * 2;
    for (int i = threadIdx.x; i < NBLK * 64; i += 256) reinterpret_cast<f16x8*>(sw)[i] = reinterpret_cast<const f16x8*>(w)[i];
    __syncthreads();
    f16x8 Bh[4][6], Bl[4][6];
#pragma unroll
    for (int r = 0; r < 4; ++r)
#pragma unroll
        for (int k = 0; k < 6; ++k) {
            float x[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) x[e] = in[((r * 6 + k) * 8 + e) * 256 + threadIdx.x];
            split8(x, Bh[r][k], Bl[r][k]);
        }
    for (int it = 0; it < iters; ++it) {
        int z = 0; asm volatile("" : "+s"(z));
        const f16x8* swz = reinterpret_cast<const f16x8*>(sw) + z + lane;
#pragma unroll
        for (int oct = 0; oct < 3; ++oct) {
            const int ob = oct % NOCT;
            f32x16 acc[2];
#pragma unroll
            for (int t = 0; t < 2; ++t)
#pragma unroll
                for (int e = 0; e < 16; ++e) acc[t][e] = 0.f;
#pragma unroll
            for (int tap = 0; tap < 3; ++tap)
#pragma unroll
                for (int k = 0; k < 6; ++k) {
                    const f16x8 ah = swz[(((ob * 3 + tap) * 6 + k) * 2 + 0) * 64], al = swz[(((ob * 3 + tap) * 6 + k) * 2 + 1) * 64];
#pragma unroll
                    for (int t = 0; t < 2; ++t) acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, Bh[t + tap][k], acc[t], 0, 0, 0);
#pragma unroll
                    for (int t = 0; t < 2; ++t) acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, Bl[t + tap][k], acc[t], 0, 0, 0);
#pragma unroll
                    for (int t = 0; t < 2; ++t) acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al, Bh[t + tap][k], acc[t], 0, 0, 0);
                }
            // epilogue of this 32-channel output tile: activation, then straight into the NEXT layer's operands (k-steps 2 oct, 2 oct + 1 of
            // rows 2, 3: D registers 8s .. 8s+7 of the tile are the eight K slots of k-step s) -- the fp32 results need not stay alive
#pragma unroll
            for (int t = 0; t < 2; ++t)
#pragma unroll
                for (int s = 0; s < 2; ++s) {
                    float x[8];
#pragma unroll
                    for (int e = 0; e < 8; ++e) x[e] = act1(acc[t][8 * s + e], -3.2f);
                    // (written to a shadow set: the taps of later output tiles still read this iteration's inputs)
                    split8(x, Bh[t][2 * oct + s], Bl[t][2 * oct + s]);
                }
        }
        // rotate: the freshly written rows 0, 1 become the new rows 2, 3 (register renaming, free after unrolling by two would be ideal;
        // here the swap costs v_movs on the operand registers -- counted against B, it is what a real kernel pays for its history too)
#pragma unroll
        for (int k = 0; k < 6; ++k) {
            f16x8 th = Bh[0][k], tl = Bl[0][k]; Bh[0][k] = Bh[2][k]; Bl[0][k] = Bl[2][k]; Bh[2][k] = th; Bl[2][k] = tl;
            th = Bh[1][k]; tl = Bl[1][k]; Bh[1][k] = Bh[3][k]; Bl[1][k] = Bl[3][k]; Bh[3][k] = th; Bl[3][k] = tl;
        }
    }
    float r = 0.f;
#pragma unroll
    for (int k = 0; k < 6; ++k) r += (float)Bh[2][k][0] + (float)Bl[3][k][7];
    out[blockIdx.x * 256 + threadIdx.x] = r;
}

// ---- B with the weights STREAMED like the real stage kernels do it: one 36 KB chunk (one 32-channel output tile) per step, L2 -> LDS by
// global_load_lds into a three-slot ring shared by the workgroup's four waves, two chunks in flight, counted wait + bare barrier per step
// (owwhip_hx.h: WRing, NS = 3).  One workgroup per CU (the registers allow one wave per SIMD), so nothing else covers the ring's waits.
__global__ __launch_bounds__(256, 1) void k32s(float* out, const float* in, const _Float16* w, int iters) {
    __shared__ __attribute__((aligned(16))) float slot0[9216];
    __shared__ __attribute__((aligned(16))) float slot1[9216];
    __shared__ __attribute__((aligned(16))) float slot2[9216];
    float* const slots[3] = {slot0, slot1, slot2};
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const float* wf = reinterpret_cast<const float*>(w);
    auto issue = [&](int chunk, float* dst) {                 // 36 blocks of 1 KB, 9 per wave
        const float* src = wf + (size_t)(chunk % 3) * 9216;
#pragma unroll
        for (int u = 0; u < 9; ++u) {
            const int i = u * 4 + wave;
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src + i * 256 + lane * 4),
                                             (__attribute__((address_space(3))) void*)(dst + i * 256), 16, 0, 0);
        }
    };
    f16x8 Bh[4][6], Bl[4][6];
#pragma unroll
    for (int r = 0; r < 4; ++r)
#pragma unroll
        for (int k = 0; k < 6; ++k) {
            float x[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) x[e] = in[((r * 6 + k) * 8 + e) * 256 + threadIdx.x];
            split8(x, Bh[r][k], Bl[r][k]);
        }
    issue(0, slot0);
    issue(1, slot1);
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int oct = 0; oct < 3; ++oct) {
            // chunk c = 3 it + oct: wait until it has landed (the newest one, 9 DMA instructions of this wave, may stay in flight), meet the
            // other waves, then refill the slot everybody has just left
            __builtin_amdgcn_s_waitcnt((9 & 0xF) | ((9 >> 4) << 14) | (0x7 << 4) | (0xF << 8));
            asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
            issue(oct + 2, slots[(oct + 2) % 3]);
            const f16x8* swz = reinterpret_cast<const f16x8*>(slots[oct]) + lane;
            f32x16 acc[2];
#pragma unroll
            for (int t = 0; t < 2; ++t)
#pragma unroll
                for (int e = 0; e < 16; ++e) acc[t][e] = 0.f;
#pragma unroll
            for (int tap = 0; tap < 3; ++tap)
#pragma unroll
                for (int k = 0; k < 6; ++k) {
                    const f16x8 ah = swz[((tap * 6 + k) * 2 + 0) * 64], al = swz[((tap * 6 + k) * 2 + 1) * 64];
#pragma unroll
                    for (int t = 0; t < 2; ++t) acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, Bh[t + tap][k], acc[t], 0, 0, 0);
#pragma unroll
                    for (int t = 0; t < 2; ++t) acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, Bl[t + tap][k], acc[t], 0, 0, 0);
#pragma unroll
                    for (int t = 0; t < 2; ++t) acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al, Bh[t + tap][k], acc[t], 0, 0, 0);
                }
#pragma unroll
            for (int t = 0; t < 2; ++t)
#pragma unroll
                for (int s = 0; s < 2; ++s) {
                    float x[8];
#pragma unroll
                    for (int e = 0; e < 8; ++e) x[e] = act1(acc[t][8 * s + e], -3.2f);
                    split8(x, Bh[t][2 * oct + s], Bl[t][2 * oct + s]);
                }
        }
#pragma unroll
        for (int k = 0; k < 6; ++k) {
            f16x8 th = Bh[0][k], tl = Bl[0][k]; Bh[0][k] = Bh[2][k]; Bl[0][k] = Bl[2][k]; Bh[2][k] = th; Bl[2][k] = tl;
            th = Bh[1][k]; tl = Bl[1][k]; Bh[1][k] = Bh[3][k]; Bl[1][k] = Bl[3][k]; Bh[3][k] = th; Bl[3][k] = tl;
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    float r = 0.f;
#pragma unroll
    for (int k = 0; k < 6; ++k) r += (float)Bh[2][k][0] + (float)Bl[3][k][7];
    out[blockIdx.x * 256 + threadIdx.x] = r;
}

template <class K>
int timeit(const char* name, K kern, int wps, int lds, int streams_per_wave, float* out, const float* in, const _Float16* w) {
    const int iters = 300;
    if (lds > 0) CHECK(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    hipEvent_t a, b; CHECK(hipEventCreate(&a)); CHECK(hipEventCreate(&b));
    const int grid = 256 * wps;
    hipLaunchKernelGGL(kern, dim3(grid), dim3(256), lds, 0, out, in, w, iters);
    CHECK(hipDeviceSynchronize());
    CHECK(hipEventRecord(a));
    hipLaunchKernelGGL(kern, dim3(grid), dim3(256), lds, 0, out, in, w, iters);
    CHECK(hipEventRecord(b)); CHECK(hipEventSynchronize(b));
    float ms; CHECK(hipEventElapsedTime(&ms, a, b));
    const double stream_layers = (double)grid * 4 * streams_per_wave * iters;
    printf("%-34s waves/SIMD %d  %8.3f ms  %7.2f ns per stream-layer (chip)  %6.2f us per wave-layer\n", name, wps, ms, ms * 1e6 / stream_layers,
           ms * 1e3 / iters);
    return 0;
}

int main() {
    float *out, *in; _Float16* w;
    CHECK(hipMalloc(&out, 1 << 22)); CHECK(hipMalloc(&in, 1 << 24)); CHECK(hipMalloc(&w, 1 << 22));
    std::vector<float> h(1 << 22);
    for (size_t i = 0; i < h.size(); ++i) h[i] = ((i * 2654435761u) % 2001) * 1e-3f - 1.0f;
    CHECK(hipMemcpy(in, h.data(), h.size() * 4, hipMemcpyHostToDevice));
    std::vector<_Float16> hw(1 << 21);
    for (size_t i = 0; i < hw.size(); ++i) hw[i] = (_Float16)((((i * 40503u) % 2001) * 1e-3f - 1.0f) * 0.03f);
    CHECK(hipMemcpy(w, hw.data(), hw.size() * 2, hipMemcpyHostToDevice));
    // LDS: 16x16 tiles, NOCT output tiles of 16 channels = NOCT * 18 KB; 32x32 tiles, NOCT of 32 channels = NOCT * 36 KB
    timeit("A 16x16x32, 4 streams/wave", k16<2, 3>, 3, 2 * 18 * 1024 + 15 * 1024, 4, out, in, w);     // (padded to 51 KB: three workgroups per CU)
    timeit("A 16x16x32, 4 streams/wave", k16<2, 2>, 2, 72 * 1024, 4, out, in, w);
    timeit("A 16x16x32, 4 streams/wave", k16<2, 1>, 1, 100 * 1024, 4, out, in, w);
    timeit("B 32x32x16, 8 streams/wave", k32<1, 2>, 2, 72 * 1024, 8, out, in, w);
    timeit("B 32x32x16, 8 streams/wave", k32<1, 1>, 1, 100 * 1024, 8, out, in, w);
    timeit("B 32x32x16 + streamed weights", k32s, 1, 0, 8, out, in, w);
    return 0;
}
