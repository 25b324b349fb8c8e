// Feasibility micro-benchmark for an fp16-split ("f16x3") form of the register-resident conv stack:
// x*w ~= xh*wh + xh*wl + xl*wh with x = xh + xl, w = wh + wl (f16 halves, 22-bit mantissa), fp32 accumulate,
// on v_mfma_f32_16x16x32_f16.  One layer body = time conv, C channels (multiple of 32), NT position tiles per wave,
// weights static in LDS, epilogue = BN + activation + re-split of the outputs into f16 hi/lo operand registers.
#include <hip/hip_runtime.h>
#include <hip/hip_fp16.h>
#include <cstdio>
#include <vector>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

__device__ __forceinline__ float leaky_clamp(float x) { return fmaxf(fmaxf(0.2f * x, x), -0.4f); }

// split four fp32 values (one D register quad = channels 4j..4j+3 of a channel tile) into f16 hi / lo quads
__device__ __forceinline__ void split4(const f32x4 v, _Float16 (&hi)[4], _Float16 (&lo)[4]) {
#pragma unroll
    for (int e = 0; e < 4; ++e) { hi[e] = (_Float16)v[e]; lo[e] = (_Float16)(v[e] - (float)hi[e]); }
}

template <int NT, int KS /*k-steps of 32 channels per tap*/, int NCTO, int WPS>
__global__ __launch_bounds__(256, WPS) void k_f16x3(float* out, const float* in, const _Float16* w, const float* sc, const float* sh, int iters) {
    extern __shared__ __attribute__((aligned(16))) _Float16 sw[];     // [oct][tap][ks][part 2][64 lanes][8]
    const int lane = threadIdx.x & 63, j = lane >> 4;
    constexpr int WHALFS = NCTO * 3 * KS * 2 * 64 * 8;
    for (int i = threadIdx.x; i < WHALFS / 8; i += 256) reinterpret_cast<f16x8*>(sw)[i] = reinterpret_cast<const f16x8*>(w)[i];
    __syncthreads();
    // activations: rows [h0, h1, x0 .. x(NT-1)], each KS k-steps of hi and lo operand quads
    f16x8 Bh[NT + 2][KS], Bl[NT + 2][KS];
#pragma unroll
    for (int r = 0; r < NT + 2; ++r)
#pragma unroll
        for (int k = 0; k < KS; ++k) {
            const f32x4 a = *reinterpret_cast<const f32x4*>(in + ((r * KS + k) * 256 + threadIdx.x) * 4);
            const f32x4 b = *reinterpret_cast<const f32x4*>(in + ((r * KS + k) * 256 + threadIdx.x) * 4 + 8192);
            _Float16 h0[4], l0[4], h1[4], l1[4];
            split4(a, h0, l0); split4(b, h1, l1);
            Bh[r][k] = f16x8{h0[0], h0[1], h0[2], h0[3], h1[0], h1[1], h1[2], h1[3]};
            Bl[r][k] = f16x8{l0[0], l0[1], l0[2], l0[3], l1[0], l1[1], l1[2], l1[3]};
        }
    for (int it = 0; it < iters; ++it) {
        int z = 0; asm volatile("" : "+s"(z));
        const f16x8* swz = reinterpret_cast<const f16x8*>(sw) + z + lane;
        f32x4 Y[NT][NCTO];
#pragma unroll
        for (int oct = 0; oct < NCTO; ++oct) {
            f32x4 acc[NT];
#pragma unroll
            for (int t = 0; t < NT; ++t) acc[t] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int tap = 0; tap < 3; ++tap)
#pragma unroll
                for (int k = 0; k < KS; ++k) {
                    const f16x8 ah = swz[(((oct * 3 + tap) * KS + k) * 2 + 0) * 64];
                    const f16x8 al = swz[(((oct * 3 + tap) * KS + k) * 2 + 1) * 64];
#pragma unroll
                    for (int t = 0; t < NT; ++t) acc[t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah, Bh[t + tap][k], acc[t], 0, 0, 0);
#pragma unroll
                    for (int t = 0; t < NT; ++t) acc[t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah, Bl[t + tap][k], acc[t], 0, 0, 0);
#pragma unroll
                    for (int t = 0; t < NT; ++t) acc[t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(al, Bh[t + tap][k], acc[t], 0, 0, 0);
                }
            const f32x4 s4 = *reinterpret_cast<const f32x4*>(sc + z + oct * 16 + 4 * j);
            const f32x4 b4 = *reinterpret_cast<const f32x4*>(sh + z + oct * 16 + 4 * j);
#pragma unroll
            for (int t = 0; t < NT; ++t) {
#pragma unroll
                for (int e = 0; e < 4; ++e) Y[t][oct][e] = leaky_clamp(acc[t][e] * s4[e] + b4[e]);
                float a = Y[t][oct][0], b = Y[t][oct][1], c = Y[t][oct][2], d = Y[t][oct][3];
                asm volatile("" : "+v"(a), "+v"(b), "+v"(c), "+v"(d));
                Y[t][oct] = f32x4{a, b, c, d};
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        // re-split the outputs into next-layer operands (rows 2.. ; history rows stay)
#pragma unroll
        for (int t = 0; t < NT; ++t)
#pragma unroll
            for (int k = 0; k < KS; ++k) {
                _Float16 h0[4], l0[4], h1[4], l1[4];
                split4(Y[t][(2 * k) % NCTO], h0, l0); split4(Y[t][(2 * k + 1) % NCTO], h1, l1);
                Bh[t + 2][k] = f16x8{h0[0], h0[1], h0[2], h0[3], h1[0], h1[1], h1[2], h1[3]};
                Bl[t + 2][k] = f16x8{l0[0], l0[1], l0[2], l0[3], l1[0], l1[1], l1[2], l1[3]};
            }
    }
    float r = 0.f;
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int k = 0; k < KS; ++k) r += (float)Bh[t + 2][k][0] + (float)Bl[t + 2][k][7];
    out[blockIdx.x * 256 + threadIdx.x] = r;
}

// pure issue-rate probes: MODE 0 = constant operands, 1 = B from 24 different register quads, 2 = + A quads read from LDS
template <int MODE, int WPS>
__global__ __launch_bounds__(256, WPS) void k_pure16(float* out, const float* in, const _Float16* w, int iters) {
    extern __shared__ __attribute__((aligned(16))) _Float16 sw[];
    const int lane = threadIdx.x & 63;
    for (int i = threadIdx.x; i < 18 * 2 * 64; i += 256) reinterpret_cast<f16x8*>(sw)[i] = reinterpret_cast<const f16x8*>(w)[i];
    __syncthreads();
    f16x8 B[24];
#pragma unroll
    for (int r = 0; r < 24; ++r) {
        const f32x4 a = *reinterpret_cast<const f32x4*>(in + (r * 256 + threadIdx.x) * 4);
        B[r] = f16x8{(_Float16)a[0], (_Float16)a[1], (_Float16)a[2], (_Float16)a[3], (_Float16)a[0], (_Float16)a[1], (_Float16)a[2], (_Float16)a[3]};
    }
    f32x4 acc[4] = {};
    f16x8 a0 = B[0];
    for (int it = 0; it < iters; ++it) {
        int z = 0; asm volatile("" : "+s"(z));
        const f16x8* swz = reinterpret_cast<const f16x8*>(sw) + z + lane;
#pragma unroll
        for (int k = 0; k < 18; ++k) {
            const f16x8 a = MODE == 2 ? swz[k * 64] : a0;
#pragma unroll
            for (int u = 0; u < 3; ++u)
#pragma unroll
                for (int t = 0; t < 4; ++t)
                    acc[t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, MODE == 0 ? B[1] : B[(k * 4 + t + u * 7) % 24], acc[t], 0, 0, 0);
        }
    }
    out[blockIdx.x * 256 + threadIdx.x] = acc[0][0] + acc[1][1] + acc[2][2] + acc[3][3];
}
template <int MODE, int WPS>
int run_pure(const char* name, float* out, const float* in, const _Float16* w) {
    const int iters = 400;
    auto kern = k_pure16<MODE, WPS>;
    CHECK(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    const int lds = WPS == 1 ? 100 * 1024 : (WPS == 2 ? 70 * 1024 : 50 * 1024);
    hipEvent_t a, b; CHECK(hipEventCreate(&a)); CHECK(hipEventCreate(&b));
    hipLaunchKernelGGL(kern, dim3(256 * WPS), dim3(256), lds, 0, out, in, w, iters);
    CHECK(hipDeviceSynchronize());
    CHECK(hipEventRecord(a));
    hipLaunchKernelGGL(kern, dim3(256 * WPS), dim3(256), lds, 0, out, in, w, iters);
    CHECK(hipEventRecord(b)); CHECK(hipEventSynchronize(b));
    float ms; CHECK(hipEventElapsedTime(&ms, a, b));
    printf("%-26s wps=%d %7.3f ms  cycles per f16 MFMA %5.1f\n", name, WPS, ms, ms * 1e-3 * 2.4e9 / (iters * 18.0 * 12 * WPS));
    return 0;
}

template <int NT, int KS, int NCTO, int WPS>
int run(const char* name, float* out, const float* in, const _Float16* w, const float* sc, const float* sh) {
    const int iters = 200;
    auto kern = k_f16x3<NT, KS, NCTO, WPS>;
    const int wbytes = NCTO * 3 * KS * 2 * 64 * 8 * 2;
    CHECK(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    int lds = WPS == 1 ? 100 * 1024 : (WPS == 2 ? 70 * 1024 : 50 * 1024);
    if (wbytes > lds) { if (WPS > 1) { printf("%s: weights %d B exceed the LDS slice\n", name, wbytes); return 0; } lds = wbytes; }
    hipEvent_t a, b; CHECK(hipEventCreate(&a)); CHECK(hipEventCreate(&b));
    hipLaunchKernelGGL(kern, dim3(256 * WPS), dim3(256), lds, 0, out, in, w, sc, sh, iters);
    CHECK(hipDeviceSynchronize());
    CHECK(hipEventRecord(a));
    hipLaunchKernelGGL(kern, dim3(256 * WPS), dim3(256), lds, 0, out, in, w, sc, sh, iters);
    CHECK(hipEventRecord(b)); CHECK(hipEventSynchronize(b));
    float ms; CHECK(hipEventElapsedTime(&ms, a, b));
    const double mfma = (double)iters * NCTO * 3 * KS * 3 * NT;            // per wave
    const double cyc = ms * 1e-3 * 2.4e9 / (mfma * WPS);
    // fp32-equivalent work: each triple of f16 MFMAs does 16x16x32 MACs = 8 fp32 16x16x4 MFMAs (256 pipe cycles)
    printf("%-26s wps=%d %7.3f ms  cycles per f16 MFMA %5.1f   speed-up over a perfect fp32-MFMA stream %4.2fx\n", name, WPS, ms, cyc,
           256.0 / (3.0 * cyc));
    return 0;
}

int main() {
    float *out, *in, *sc, *sh; _Float16* w;
    CHECK(hipMalloc(&out, 1 << 22)); CHECK(hipMalloc(&in, 1 << 24)); CHECK(hipMalloc(&w, 1 << 22)); CHECK(hipMalloc(&sc, 4096)); CHECK(hipMalloc(&sh, 4096));
    std::vector<float> h(1 << 22);
    for (size_t i = 0; i < h.size(); ++i) h[i] = ((i * 2654435761u) % 2001) * 1e-3f - 1.0f;
    CHECK(hipMemcpy(in, h.data(), h.size() * 4, hipMemcpyHostToDevice));
    std::vector<_Float16> hw(1 << 21);
    for (size_t i = 0; i < hw.size(); ++i) hw[i] = (_Float16)((((i * 40503u) % 2001) * 1e-3f - 1.0f) * 0.06f);
    CHECK(hipMemcpy(w, hw.data(), hw.size() * 2, hipMemcpyHostToDevice));
    std::vector<float> one(1024, 1.0f), zero(1024, 0.01f);
    CHECK(hipMemcpy(sc, one.data(), 4096, hipMemcpyHostToDevice)); CHECK(hipMemcpy(sh, zero.data(), 4096, hipMemcpyHostToDevice));
    run_pure<0, 1>("pure const", out, in, w); run_pure<0, 2>("pure const", out, in, w);
    run_pure<1, 1>("pure 24 B quads", out, in, w); run_pure<1, 2>("pure 24 B quads", out, in, w);
    run_pure<2, 1>("pure + A from LDS", out, in, w); run_pure<2, 2>("pure + A from LDS", out, in, w);
    run<4, 3, 6, 1>("time NT=4 C=96", out, in, w, sc, sh);
    run<2, 3, 6, 1>("time NT=2 C=96", out, in, w, sc, sh);
    run<4, 3, 3, 2>("time NT=4 C=96 (3 oct)", out, in, w, sc, sh);
    run<2, 3, 3, 2>("time NT=2 C=96 (3 oct)", out, in, w, sc, sh);
    run<2, 3, 3, 3>("time NT=2 C=96 (3 oct)", out, in, w, sc, sh);
    run<8, 2, 4, 1>("time NT=8 C=64", out, in, w, sc, sh);
    run<4, 2, 4, 2>("time NT=4 C=64", out, in, w, sc, sh);
    run<4, 2, 4, 3>("time NT=4 C=64", out, in, w, sc, sh);
    run<4, 1, 2, 3>("time NT=4 C=32", out, in, w, sc, sh);
    return 0;
}
