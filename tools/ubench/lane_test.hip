// prints what the lane-crossing helpers of owwhip_rr.h do to registers holding lane ids
#include <hip/hip_runtime.h>
#include <cstdio>
#include "owwhip_rr.h"
using namespace owr;
__global__ void k(float* o) {
    const int l = threadIdx.x;
    f32x4 v = {(float)l, 100.f + l, 200.f + l, 300.f + l};
    const f32x4 p = pack_half(v);
    o[l] = p[0]; o[64 + l] = p[1];
    o[128 + l] = dpp_shr1_zero((float)l); o[192 + l] = dpp_shl1_zero((float)l);
    o[256 + l] = dpp_shr1_carry((float)l, 1000.f + l); o[320 + l] = dpp_shl1_carry((float)l, 1000.f + l);
}
int main() {
    float* d; hipMalloc(&d, 384 * 4);
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d);
    float h[384]; hipMemcpy(h, d, sizeof h, hipMemcpyDeviceToHost);
    const char* names[6] = {"pack P0", "pack P1", "shr1_zero", "shl1_zero", "shr1_carry", "shl1_carry"};
    for (int r = 0; r < 6; ++r) { printf("%-10s", names[r]); for (int l = 0; l < 64; ++l) printf(" %g", h[r * 64 + l]); printf("\n"); }
    bool ok = true;
    for (int l = 0; l < 64; ++l) ok = ok && h[l] == (l < 32 ? l : 100 + l - 32) && h[64 + l] == (l < 32 ? 200 + l : 300 + l - 32);
    printf(ok ? "pack_half OK\n" : "pack_half WRONG\n");
    return ok ? 0 : 1;
}
