// Micro-benchmarks for the fp32-MFMA instruction streams of owwhip_rr.h: cycles per v_mfma_f32_16x16x4_f32 seen by one
// wave, for 1 and 2 waves per SIMD, as pieces of the real layer bodies are added.
// build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -I../../openwakeword_amd/csrc mfma_ubench.hip -o mfma_ubench
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include "owwhip_rr.h"
using namespace owr;

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

// K1: 4 accumulators, constant operands
template <int WPS>
__global__ __launch_bounds__(256, WPS) void k_pure(float* out, int iters) {
    extern __shared__ float pad[];
    float a = threadIdx.x * 0.001f, b = threadIdx.x * 0.002f;
    f32x4 acc[4] = {};
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 16; ++u)
#pragma unroll
            for (int t = 0; t < 4; ++t) acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[t], 0, 0, 0);
    }
    out[blockIdx.x * 256 + threadIdx.x] = acc[0][0] + acc[1][1] + acc[2][2] + acc[3][3];
}

// K2: operand B cycles through 80 registers (4 tiles x 20), operand A through 4
template <int WPS>
__global__ __launch_bounds__(256, WPS) void k_regs(float* out, const float* in, int iters) {
    extern __shared__ float pad[];
    f32x4 X[4][5];
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
        for (int c = 0; c < 5; ++c) X[t][c] = *reinterpret_cast<const f32x4*>(in + ((t * 5 + c) * 256 + threadIdx.x) * 4);
    f32x4 a = X[0][0];
    f32x4 acc[4] = {};
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int c = 0; c < 5; ++c)
#pragma unroll
            for (int e = 0; e < 4; ++e)
#pragma unroll
                for (int t = 0; t < 4; ++t) acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[e], X[t][c][e], acc[t], 0, 0, 0);
    }
    out[blockIdx.x * 256 + threadIdx.x] = acc[0][0] + acc[1][1] + acc[2][2] + acc[3][3];
}

// K3: K2 + a fresh 16-byte weight load per 16 MFMAs (prefetch distance 2), weights = 75 KB L2-resident block
template <int WPS>
__global__ __launch_bounds__(256, WPS) void k_wload(float* out, const float* in, const float* w_, int iters) {
    extern __shared__ float pad[];
    const int lane = threadIdx.x & 63;
    f32x4 X[4][5];
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
        for (int c = 0; c < 5; ++c) X[t][c] = *reinterpret_cast<const f32x4*>(in + ((t * 5 + c) * 256 + threadIdx.x) * 4);
    f32x4 acc[4] = {};
    for (int it = 0; it < iters; ++it) {
        int z = 0; asm volatile("" : "+s"(z)); const float* w = w_ + z;     // opaque: no hoisting of the weight loads
        f32x4 wq[2];
        wq[0] = load_w<5>(w, 0, 0, 0, lane);
        wq[1] = load_w<5>(w, 0, 0, 1, lane);
#pragma unroll
        for (int i = 0; i < 75; ++i) {
            const f32x4 a = wq[i % 2];
            if (i + 2 < 75) wq[i % 2] = load_w<5>(w, (i + 2) / 15, ((i + 2) / 5) % 3, (i + 2) % 5, lane);
#pragma unroll
            for (int e = 0; e < 4; ++e)
#pragma unroll
                for (int t = 0; t < 4; ++t) acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[e], X[t][i % 5][e], acc[t], 0, 0, 0);
        }
    }
    out[blockIdx.x * 256 + threadIdx.x] = acc[0][0] + acc[1][1] + acc[2][2] + acc[3][3];
}

// K4 / K5: the real layer bodies on register-resident tiles (no tile I/O), NT = 4, 80 channels
template <int WPS, bool MEL>
__global__ __launch_bounds__(256, WPS) void k_layer(float* out, const float* in, const float* w_, const float* sc_, const float* sh_, int iters) {
    extern __shared__ float pad[];
    const int lane = threadIdx.x & 63;
    f32x4 X[4][5], Y[4][5], H0[5], H1[5];
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
        for (int c = 0; c < 5; ++c) X[t][c] = *reinterpret_cast<const f32x4*>(in + ((t * 5 + c) * 256 + threadIdx.x) * 4);
#pragma unroll
    for (int c = 0; c < 5; ++c) { H0[c] = X[0][c]; H1[c] = X[1][c]; }
    for (int it = 0; it < iters; ++it) {
        int z = 0; asm volatile("" : "+s"(z)); const float* w = w_ + z; const float* sc = sc_ + z; const float* sh = sh_ + z;
        if (MEL) conv_mel<5, 5, 4, 8, true>(X, Y, w, sc, sh, lane);
        else conv_time<5, 5, 4, true>(H0, H1, X, Y, w, sc, sh, lane);
#pragma unroll
        for (int t = 0; t < 4; ++t)
#pragma unroll
            for (int c = 0; c < 5; ++c) X[t][c] = Y[t][c];
    }
    float r = 0.f;
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
        for (int c = 0; c < 5; ++c) r += X[t][c][0] + X[t][c][3];
    out[blockIdx.x * 256 + threadIdx.x] = r;
}

// NT = 2, 96 channels (stages D / E)
template <int WPS, bool MEL>
__global__ __launch_bounds__(256, WPS) void k_layer2(float* out, const float* in, const float* w_, const float* sc_, const float* sh_, int iters) {
    extern __shared__ float pad[];
    const int lane = threadIdx.x & 63;
    f32x4 X[2][6], Y[2][6], H0[6], H1[6];
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int c = 0; c < 6; ++c) X[t][c] = *reinterpret_cast<const f32x4*>(in + ((t * 6 + c) * 256 + threadIdx.x) * 4);
#pragma unroll
    for (int c = 0; c < 6; ++c) { H0[c] = X[0][c]; H1[c] = X[1][c]; }
    for (int it = 0; it < iters; ++it) {
        int z = 0; asm volatile("" : "+s"(z)); const float* w = w_ + z; const float* sc = sc_ + z; const float* sh = sh_ + z;
        if (MEL) conv_mel<6, 6, 2, 4, true>(X, Y, w, sc, sh, lane);
        else conv_time<6, 6, 2, true>(H0, H1, X, Y, w, sc, sh, lane);
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int c = 0; c < 6; ++c) X[t][c] = Y[t][c];
    }
    float r = 0.f;
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int c = 0; c < 6; ++c) r += X[t][c][0] + X[t][c][3];
    out[blockIdx.x * 256 + threadIdx.x] = r;
}

// K6 / K7: the LDS-streamed layer bodies (weights once per workgroup through a double-buffered LDS chunk)
template <int WPS, bool MEL, int NT, int NC, int F>
__global__ __launch_bounds__(256, WPS) void k_layer_lds(float* out, const float* in, const float* w_, const float* sc_, const float* sh_, int iters) {
    extern __shared__ float pad[];
    __shared__ __attribute__((aligned(16))) float wbuf[2 * WBUF_FLOATS];
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    f32x4 X[NT][NC], Y[NT][NC], H0[NC], H1[NC];
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int c = 0; c < NC; ++c) X[t][c] = *reinterpret_cast<const f32x4*>(in + ((t * NC + c) * 256 + threadIdx.x) * 4);
#pragma unroll
    for (int c = 0; c < NC; ++c) { H0[c] = X[0][c]; H1[c] = X[1][c]; }
    issue_chunk<3 * NC>(w_, wbuf, wave, lane);
    chunk_sync();
    for (int it = 0; it < iters; ++it) {
        int z = 0; asm volatile("" : "+s"(z)); const float* w = w_ + z; const float* sc = sc_ + z; const float* sh = sh_ + z;
        // NC chunks per layer: with an even chunk count the parity is the same every iteration; odd NC flips it, so run two layers
        if (MEL) { conv_mel_lds<NC, NC, NT, F, true, 0, 3 * NC>(X, Y, wbuf, w, w, sc, sh, wave, lane);
                   conv_mel_lds<NC, NC, NT, F, true, NC, 3 * NC>(Y, X, wbuf, w, w, sc, sh, wave, lane); }
        else     { conv_time_lds<NC, NC, NT, true, 0, 3 * NC>(H0, H1, X, Y, wbuf, w, w, sc, sh, wave, lane);
                   conv_time_lds<NC, NC, NT, true, NC, 3 * NC>(H0, H1, Y, X, wbuf, w, w, sc, sh, wave, lane); }
    }
    float r = 0.f;
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int c = 0; c < NC; ++c) r += X[t][c][0] + X[t][c][3];
    out[blockIdx.x * 256 + threadIdx.x] = r;
}

template <class F>
int timeit(const char* name, int wps, double mfma_per_wave, F launch) {
    hipEvent_t a, b;
    CHECK(hipEventCreate(&a)); CHECK(hipEventCreate(&b));
    launch();
    CHECK(hipDeviceSynchronize());
    CHECK(hipEventRecord(a));
    launch();
    CHECK(hipEventRecord(b));
    CHECK(hipEventSynchronize(b));
    float ms = 0;
    CHECK(hipEventElapsedTime(&ms, a, b));
    // every SIMD runs `wps` waves concurrently; cycles per MFMA as seen by the SIMD's matrix pipe (32 = peak)
    const double cyc = ms * 1e-3 * 2.4e9;
    printf("%-28s wps=%d  %8.3f ms  pipe cycles/MFMA @2.4GHz = %6.1f  (util %.1f%%)\n", name, wps, ms, cyc / (mfma_per_wave * wps),
           100.0 * 32.0 * mfma_per_wave * wps / cyc);
    return 0;
}

int main() {
    float *out, *in, *w, *sc, *sh;
    const size_t nw = 6 * 3 * 6 * 64 * 4;
    CHECK(hipMalloc(&out, 1 << 22)); CHECK(hipMalloc(&in, 64 * 256 * 4 * 4)); CHECK(hipMalloc(&w, nw * 4)); CHECK(hipMalloc(&sc, 4096)); CHECK(hipMalloc(&sh, 4096));
    std::vector<float> h(64 * 256 * 4);
    for (size_t i = 0; i < h.size(); ++i) h[i] = ((i * 2654435761u) % 2001) * 1e-3f - 1.0f;
    CHECK(hipMemcpy(in, h.data(), h.size() * 4, hipMemcpyHostToDevice));
    std::vector<float> hw(nw);
    for (size_t i = 0; i < nw; ++i) hw[i] = (((i * 40503u) % 2001) * 1e-3f - 1.0f) * 0.06f;
    CHECK(hipMemcpy(w, hw.data(), nw * 4, hipMemcpyHostToDevice));
    std::vector<float> one(1024, 1.0f), zero(1024, 0.01f);
    CHECK(hipMemcpy(sc, one.data(), 4096, hipMemcpyHostToDevice)); CHECK(hipMemcpy(sh, zero.data(), 4096, hipMemcpyHostToDevice));
    const int iters = 40;
    for (int wps = 1; wps <= 2; ++wps) {
        const int grid = 256 * wps;                 // one (two) 4-wave workgroups per CU
        int lds = wps == 1 ? 100 * 1024 : 70 * 1024;    // pins the residency
#define L(K, ...) [&] { if (wps == 1) hipLaunchKernelGGL((K<1 __VA_OPT__(,) __VA_ARGS__>), dim3(grid), dim3(256), lds, 0, ARGS); else hipLaunchKernelGGL((K<2 __VA_OPT__(,) __VA_ARGS__>), dim3(grid), dim3(256), lds, 0, ARGS); }
#define ARGS out, iters * 20
        hipFuncSetAttribute((const void*)k_pure<1>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        hipFuncSetAttribute((const void*)k_pure<2>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        timeit("pure (const operands)", wps, iters * 20 * 64.0, L(k_pure));
#undef ARGS
#define ARGS out, in, iters * 16
        hipFuncSetAttribute((const void*)k_regs<1>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        hipFuncSetAttribute((const void*)k_regs<2>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        timeit("80 B-operand registers", wps, iters * 16 * 80.0, L(k_regs));
#undef ARGS
#define ARGS out, in, w, iters
        hipFuncSetAttribute((const void*)k_wload<1>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        hipFuncSetAttribute((const void*)k_wload<2>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        timeit("+ weight load / 16 MFMA", wps, iters * 75 * 16.0, L(k_wload));
#undef ARGS
#define ARGS out, in, w, sc, sh, iters
        hipFuncSetAttribute((const void*)k_layer<1, false>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        hipFuncSetAttribute((const void*)k_layer<2, false>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        hipFuncSetAttribute((const void*)k_layer<1, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        hipFuncSetAttribute((const void*)k_layer<2, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        hipFuncSetAttribute((const void*)k_layer2<1, false>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        hipFuncSetAttribute((const void*)k_layer2<2, false>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        hipFuncSetAttribute((const void*)k_layer2<1, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        hipFuncSetAttribute((const void*)k_layer2<2, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        timeit("conv_time NT=4 C=80", wps, iters * 5 * 3 * 5 * 16.0, L(k_layer, false));
        timeit("conv_mel  NT=4 C=80", wps, iters * 5 * 3 * 5 * 16.0, L(k_layer, true));
        timeit("conv_time NT=2 C=96", wps, iters * 6 * 3 * 6 * 8.0, L(k_layer2, false));
        timeit("conv_mel  NT=2 C=96", wps, iters * 6 * 3 * 6 * 8.0, L(k_layer2, true));
#define LL(MEL, NT, NC, FF, name) do { \
        hipFuncSetAttribute((const void*)k_layer_lds<1, MEL, NT, NC, FF>, hipFuncAttributeMaxDynamicSharedMemorySize, 120 * 1024); \
        hipFuncSetAttribute((const void*)k_layer_lds<2, MEL, NT, NC, FF>, hipFuncAttributeMaxDynamicSharedMemorySize, 120 * 1024); \
        timeit(name, wps, iters * 2.0 * NC * 3 * NC * 4 * NT, L(k_layer_lds, MEL, NT, NC, FF)); } while (0)
        lds = wps == 1 ? 64 * 1024 : 36 * 1024;     // + 36 KB static
        LL(false, 4, 5, 8, "LDS conv_time NT=4 C=80");
        LL(true, 4, 5, 8, "LDS conv_mel  NT=4 C=80");
        LL(false, 2, 6, 4, "LDS conv_time NT=2 C=96");
        LL(true, 2, 6, 4, "LDS conv_mel  NT=2 C=96");
        LL(false, 4, 3, 16, "LDS conv_time NT=4 C=48");
        LL(true, 4, 3, 16, "LDS conv_mel  NT=4 C=48");
#undef ARGS
    }
    return 0;
}
