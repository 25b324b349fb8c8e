// Does v_mfma_f32_16x16x32_f16 honour f16 subnormal INPUTS?  (tools/ubench: hipcc --offload-arch=gfx950 -O2 denorm_test.hip -o denorm_test)
#include <hip/hip_runtime.h>
#include <cstdio>
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
__global__ void k(float* out, float a_val, float b_val) {
    f16x8 a, b;
    for (int i = 0; i < 8; ++i) { a[i] = (_Float16)0.f; b[i] = (_Float16)0.f; }
    a[0] = (_Float16)a_val;            // A[i][k = 8g] for every lane
    b[0] = (_Float16)b_val;
    f32x4 c = {0.f, 0.f, 0.f, 0.f};
    c = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, 0, 0, 0);
    out[threadIdx.x] = c[0];
}
int main() {
    float* d; hipMalloc(&d, 64 * 4);
    const float tests[][2] = {{1.0f, 1.0f}, {6.0e-5f, 1024.f}, {3.0e-5f, 1024.f}, {1.0e-6f, 1024.f}, {6.0e-8f, 32768.f}, {1024.f, 3.0e-5f}, {1024.f, 1.0e-6f}};
    for (auto& t : tests) {
        hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d, t[0], t[1]);
        float h[64]; hipMemcpy(h, d, sizeof h, hipMemcpyDeviceToHost);
        printf("a = %.3g (f16 %.6g)  b = %.3g (f16 %.6g): mfma = %.9g   exact product of the f16 values x 4 k-groups = %.9g\n", t[0], (double)(float)(_Float16)t[0], t[1],
               (double)(float)(_Float16)t[1], h[0], 4.0 * (double)(float)(_Float16)t[0] * (double)(float)(_Float16)t[1]);
    }
    return 0;
}
