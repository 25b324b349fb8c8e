// Issue cost of packed fp32 VALU on gfx950, measured ("v_pk_mul_f32 + v_pk_max_f32 for the leaky / floor pair ...
// would halve the first block").  A wave64 VALU instruction issues over 2 passes of the SIMD-32; v_pk_* f32 does TWO values per lane,
// so if it takes twice the passes nothing is gained by pairing -- only the instruction COUNT (SQ_INSTS_VALU) drops.
// Streams of independent instructions, one wave per SIMD and four, cycles per instruction by s_memtime.
//   hipcc --offload-arch=gfx950 -O3 pk_ubench.hip -o pk_ubench && ./pk_ubench
#include <hip/hip_runtime.h>
#include <cstdio>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
typedef float f32x2 __attribute__((ext_vector_type(2)));

template <int KIND>      // 0: v_mul_f32 x2 (two values), 1: v_pk_mul_f32 (two values), 2: v_max3_f32 x2, 3: v_mul + v_max3 per value x2, 4: v_pk_mul + 2 v_max3, 5: v_pk_fma_f32, 6: v_fma_f32 x2
__global__ __launch_bounds__(1024) void k(float* out, long long* cyc, int iters) {
    float a[16], b[16];
    for (int i = 0; i < 16; ++i) { a[i] = threadIdx.x * 0.001f + i; b[i] = 1.0f + i * 0.01f; }
    const long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            float &x = a[2 * u], &y = a[2 * u + 1];
            if (KIND == 0) asm volatile("v_mul_f32 %0, %0, %2\n\tv_mul_f32 %1, %1, %2" : "+v"(x), "+v"(y) : "v"(b[u]));
            if (KIND == 1) { f32x2 v = {x, y}, w = {b[u], b[u]}; asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(v) : "v"(w)); x = v[0]; y = v[1]; }
            if (KIND == 2) asm volatile("v_max3_f32 %0, %0, %2, %3\n\tv_max3_f32 %1, %1, %2, %3" : "+v"(x), "+v"(y) : "v"(b[u]), "v"(b[u + 8]));
            if (KIND == 3) asm volatile("v_mul_f32 %2, 0x3e4ccccd, %0\n\tv_max3_f32 %0, %2, %0, %4\n\tv_mul_f32 %3, 0x3e4ccccd, %1\n\tv_max3_f32 %1, %3, %1, %4"
                                        : "+v"(x), "+v"(y), "=&v"(b[u]), "=&v"(b[u + 8]) : "v"(b[15]));
            if (KIND == 4) { f32x2 v = {x, y}, w = {0.2f, 0.2f}, m; asm volatile("v_pk_mul_f32 %0, %1, %2" : "=&v"(m) : "v"(v), "v"(w));
                             asm volatile("v_max3_f32 %0, %2, %0, %4\n\tv_max3_f32 %1, %3, %1, %4" : "+v"(x), "+v"(y) : "v"(m[0]), "v"(m[1]), "v"(b[15])); }
            if (KIND == 5) { f32x2 v = {x, y}, w = {b[u], b[u]}; asm volatile("v_pk_fma_f32 %0, %0, %1, %1" : "+v"(v) : "v"(w)); x = v[0]; y = v[1]; }
            if (KIND == 6) asm volatile("v_fma_f32 %0, %0, %2, %2\n\tv_fma_f32 %1, %1, %2, %2" : "+v"(x), "+v"(y) : "v"(b[u]));
        }
    }
    const long long t1 = __builtin_readcyclecounter();
    float r = 0.f;
    for (int i = 0; i < 16; ++i) r += a[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = r;
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

template <int KIND> int run(const char* name, int waves_per_simd, int values_per_iter, int insts_per_iter) {
    float* out; long long* cyc;
    const int iters = 4000, nb = 256;
    CHECK(hipMalloc(&out, nb * 1024 * sizeof(float)));
    CHECK(hipMalloc(&cyc, nb * sizeof(long long)));
    const int threads = 256 * waves_per_simd;          // 4 SIMDs x waves_per_simd waves
    hipLaunchKernelGGL(k<KIND>, dim3(nb), dim3(threads), 0, 0, out, cyc, 10);
    hipLaunchKernelGGL(k<KIND>, dim3(nb), dim3(threads), 0, 0, out, cyc, iters);
    CHECK(hipDeviceSynchronize());
    long long h[256];
    CHECK(hipMemcpy(h, cyc, sizeof h, hipMemcpyDeviceToHost));
    double mean = 0; for (int i = 0; i < nb; ++i) mean += h[i]; mean /= nb;
    const double per_iter = mean / iters;
    printf("%-44s %d wave(s)/SIMD: %7.1f cycles per wave for %2d values (%d instructions) = %.2f cycles per value-pair per wave, %.2f per value-pair per SIMD\n",
           name, waves_per_simd, per_iter, values_per_iter, insts_per_iter, per_iter / (values_per_iter / 2.0), per_iter / (values_per_iter / 2.0) / waves_per_simd);
    hipFree(out); hipFree(cyc);
    return 0;
}

int main() {
    for (int w = 1; w <= 4; w *= 2) {
        if (run<0>("2 x v_mul_f32", w, 16, 16)) return 1;
        if (run<1>("1 x v_pk_mul_f32", w, 16, 8)) return 1;
        if (run<6>("2 x v_fma_f32", w, 16, 16)) return 1;
        if (run<5>("1 x v_pk_fma_f32", w, 16, 8)) return 1;
        if (run<2>("2 x v_max3_f32", w, 16, 16)) return 1;
        if (run<3>("activation: 2 x (v_mul_f32 + v_max3_f32)", w, 16, 32)) return 1;
        if (run<4>("activation: v_pk_mul_f32 + 2 x v_max3_f32", w, 16, 24)) return 1;
    }
    return 0;
}
