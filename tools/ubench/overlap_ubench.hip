// Can a SIMD overlap one wave's f16 MFMA stream with another wave's VALU stream?  8-wave workgroups (two waves per SIMD):
// mode 0 = both waves MFMA, 1 = both VALU, 2 = one of each.  If mode 2 takes max(MFMA, VALU) the pipes overlap across waves.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

typedef float f32x16 __attribute__((ext_vector_type(16)));
template <int SHAPE>      // 0: 16x16x32 f16 (16 cycles), 1: 32x32x16 f16 (32 cycles, same flop rate), 2: 16x16x4 f32 (32 cycles)
__global__ __launch_bounds__(512, 2) void k(float* out, int mode, int it_mfma, int it_valu) {
    const int wave = threadIdx.x >> 6;
    const bool do_mfma = mode == 0 || (mode == 2 && wave < 4);
    const bool do_valu = mode == 1 || (mode == 2 && wave >= 4);
    float r = 0.f;
    if (do_mfma) {
        f16x8 a, b;
        for (int i = 0; i < 8; ++i) { a[i] = (_Float16)(threadIdx.x * 0.001f + i); b[i] = (_Float16)(i * 0.5f); }
        if (SHAPE == 0) {
            f32x4 acc[4] = {};
            for (int it = 0; it < it_mfma; ++it) {
#pragma unroll
                for (int u = 0; u < 16; ++u)
#pragma unroll
                    for (int t = 0; t < 4; ++t) acc[t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, acc[t], 0, 0, 0);
            }
            r = acc[0][0] + acc[1][1] + acc[2][2] + acc[3][3];
        } else if (SHAPE == 1) {
            f32x16 acc[2] = {};
            for (int it = 0; it < it_mfma; ++it) {
#pragma unroll
                for (int u = 0; u < 16; ++u)
#pragma unroll
                    for (int t = 0; t < 2; ++t) acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc[t], 0, 0, 0);
            }
            r = acc[0][0] + acc[1][5];
        } else {
            f32x4 acc[4] = {};
            const float fa = (float)a[0], fb = (float)b[1];
            for (int it = 0; it < it_mfma; ++it) {
#pragma unroll
                for (int u = 0; u < 8; ++u)
#pragma unroll
                    for (int t = 0; t < 4; ++t) acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(fa, fb, acc[t], 0, 0, 0);
            }
            r = acc[0][0] + acc[1][1] + acc[2][2] + acc[3][3];
        }
    }
    if (do_valu) {
        float x[8];
        for (int i = 0; i < 8; ++i) x[i] = threadIdx.x * 0.01f + i;
        for (int it = 0; it < it_valu; ++it) {
#pragma unroll
            for (int u = 0; u < 16; ++u)
#pragma unroll
                for (int i = 0; i < 8; ++i) x[i] = fmaf(x[i], 0.999f, 0.5f);
        }
        for (int i = 0; i < 8; ++i) r += x[i];
    }
    out[blockIdx.x * 512 + threadIdx.x] = r;
}

// in-wave interleave: K VALU (independent v_fma chains) after every MFMA in program order, both waves of a SIMD alike
template <int SHAPE, int K>
__global__ __launch_bounds__(512, 2) void k_inwave(float* out, int iters) {
    f16x8 a, b;
    for (int i = 0; i < 8; ++i) { a[i] = (_Float16)(threadIdx.x * 0.001f + i); b[i] = (_Float16)(i * 0.5f); }
    float x[8];
    for (int i = 0; i < 8; ++i) x[i] = threadIdx.x * 0.01f + i;
    float r = 0.f;
    if (SHAPE == 0) {
        f32x4 acc[4] = {};
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int u = 0; u < 16; ++u)
#pragma unroll
                for (int t = 0; t < 4; ++t) {
                    acc[t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, acc[t], 0, 0, 0);
#pragma unroll
                    for (int q = 0; q < K; ++q) x[(t * K + q) % 8] = fmaf(x[(t * K + q) % 8], 0.999f, 0.5f);
                    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                    __builtin_amdgcn_sched_group_barrier(0x002, K, 0);
                }
        }
        r = acc[0][0] + acc[1][1] + acc[2][2] + acc[3][3];
    } else {
        f32x16 acc[2] = {};
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int u = 0; u < 16; ++u)
#pragma unroll
                for (int t = 0; t < 2; ++t) {
                    acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc[t], 0, 0, 0);
#pragma unroll
                    for (int q = 0; q < K; ++q) x[(t * K + q) % 8] = fmaf(x[(t * K + q) % 8], 0.999f, 0.5f);
                    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                    __builtin_amdgcn_sched_group_barrier(0x002, K, 0);
                }
        }
        r = acc[0][0] + acc[1][5];
    }
    for (int i = 0; i < 8; ++i) r += x[i];
    out[blockIdx.x * 512 + threadIdx.x] = r;
}
template <int SHAPE, int K>
int run_inwave(float* out) {
    hipEvent_t a, b; CHECK(hipEventCreate(&a)); CHECK(hipEventCreate(&b));
    const int iters = 1000;
    hipLaunchKernelGGL((k_inwave<SHAPE, K>), dim3(256), dim3(512), 0, 0, out, iters);
    CHECK(hipDeviceSynchronize());
    CHECK(hipEventRecord(a));
    hipLaunchKernelGGL((k_inwave<SHAPE, K>), dim3(256), dim3(512), 0, 0, out, iters);
    CHECK(hipEventRecord(b)); CHECK(hipEventSynchronize(b));
    float ms; CHECK(hipEventElapsedTime(&ms, a, b));
    const int n_mfma = iters * (SHAPE == 0 ? 64 : 32);
    printf("in-wave %s + %d VALU per MFMA: %.3f ms  (%d MFMAs per wave, two waves per SIMD; MFMA-only time = K=0 line)\n",
           SHAPE == 0 ? "16x16x32" : "32x32x16", K, ms, n_mfma);
    return 0;
}

int main() {
    float* out; CHECK(hipMalloc(&out, 1 << 22));
    run_inwave<0, 0>(out); run_inwave<0, 1>(out); run_inwave<0, 2>(out); run_inwave<0, 4>(out);
    run_inwave<1, 0>(out); run_inwave<1, 2>(out); run_inwave<1, 4>(out); run_inwave<1, 8>(out);
    const int im = 2000, iv = 2000;           // 128k MFMAs / 256k VALU per wave
    for (int shape = 0; shape < 3; ++shape)
    for (int mode = 0; mode < 3; ++mode) {
        hipEvent_t a, b; CHECK(hipEventCreate(&a)); CHECK(hipEventCreate(&b));
        auto launch = [&] {
            if (shape == 0) hipLaunchKernelGGL(k<0>, dim3(256), dim3(512), 0, 0, out, mode, im, iv);
            else if (shape == 1) hipLaunchKernelGGL(k<1>, dim3(256), dim3(512), 0, 0, out, mode, im, iv);
            else hipLaunchKernelGGL(k<2>, dim3(256), dim3(512), 0, 0, out, mode, im, iv);
        };
        launch();
        CHECK(hipDeviceSynchronize());
        CHECK(hipEventRecord(a));
        launch();
        CHECK(hipEventRecord(b)); CHECK(hipEventSynchronize(b));
        float ms; CHECK(hipEventElapsedTime(&ms, a, b));
        const char* names[3] = {"MFMA + MFMA", "VALU + VALU", "MFMA + VALU"};
        const char* shapes[3] = {"16x16x32 f16 (64/iter)", "32x32x16 f16 (32/iter)", "16x16x4 f32 (32/iter)"};
        printf("%-24s %s (two waves per SIMD): %.3f ms\n", shapes[shape], names[mode], ms);
    }
    return 0;
}
