// Issue rate of the f16 MFMA shapes on gfx950 (cycles per instruction relative to v_mfma_f32_16x16x32_f16 = 16 cycles):
// decides whether a short K tail (16 channels) is cheaper as one 16x16x16 instruction than as a half-empty 16x16x32.
// Also: a wave alternating the two shapes, as a K = 32 + 16 layer would issue them.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

template <int SHAPE>      // 0: 16x16x32, 1: 16x16x16, 2: alternate 32,16,  3: 32x32x16, 4: 32x32x8, 5: 2x(16x16x32) + 1x(16x16x16)
__global__ __launch_bounds__(256, 2) void k(float* out, int iters) {
    f16x8 a, b;
    for (int i = 0; i < 8; ++i) { a[i] = (_Float16)(threadIdx.x * 0.001f + i); b[i] = (_Float16)(i * 0.5f); }
    const f16x4 a4 = {a[0], a[1], a[2], a[3]}, b4 = {b[0], b[1], b[2], b[3]};
    float r = 0.f;
    if (SHAPE <= 2 || SHAPE == 5) {
        f32x4 acc[4] = {};
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int u = 0; u < 12; ++u)
#pragma unroll
                for (int t = 0; t < 4; ++t) {
                    const bool k32 = SHAPE == 0 || (SHAPE == 2 && (u & 1) == 0) || (SHAPE == 5 && (u % 3) != 2);
                    if (k32) acc[t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, acc[t], 0, 0, 0);
                    else acc[t] = __builtin_amdgcn_mfma_f32_16x16x16f16(a4, b4, acc[t], 0, 0, 0);
                }
        }
        r = acc[0][0] + acc[1][1] + acc[2][2] + acc[3][3];
    } else {
        f32x16 acc[2] = {};
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int u = 0; u < 24; ++u)
#pragma unroll
                for (int t = 0; t < 2; ++t) {
                    if (SHAPE == 3) acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc[t], 0, 0, 0);
                    else acc[t] = __builtin_amdgcn_mfma_f32_32x32x8f16(a4, b4, acc[t], 0, 0, 0);
                }
        }
        r = acc[0][0] + acc[1][5];
    }
    out[blockIdx.x * 256 + threadIdx.x] = r;
}

template <int SHAPE>
static float run(float* d, int iters) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(k<SHAPE>, dim3(2048), dim3(256), 0, 0, d, iters);
    hipEventRecord(e0);
    hipLaunchKernelGGL(k<SHAPE>, dim3(2048), dim3(256), 0, 0, d, iters);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    return ms;
}
int main() {
    float* d; CHECK(hipMalloc(&d, 2048 * 256 * 4));
    const int iters = 2000;
    const char* names[] = {"16x16x32_f16", "16x16x16_f16", "alternate 32/16", "32x32x16_f16", "32x32x8_f16", "32,32,16 pattern"};
    float ms[6] = {run<0>(d, iters), run<1>(d, iters), run<2>(d, iters), run<3>(d, iters), run<4>(d, iters), run<5>(d, iters)};
    // 2048 WGs x 4 waves = 8 waves per SIMD in 4 rounds of 2 (launch bounds) -> instructions per SIMD:
    for (int s = 0; s < 6; ++s) {
        const double n = (double)iters * 48 * 8;      // MFMAs issued per SIMD (all shapes: 48 per iteration per wave)
        printf("%-18s %8.3f ms  %.2f ns per MFMA per SIMD  (x%.2f of 16x16x32)\n", names[s], ms[s], ms[s] * 1e6 / n, ms[s] / ms[0]);
    }
    return 0;
}
