// Does hipMemcpyAsync between host memory and a hipMemMap'ed device range deliver every byte?  (a probe behind OWW_GUARD_ALLOC)
// hipcc --offload-arch=gfx950 -O2 vmm_copy_probe.hip -o vmm_copy_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
__global__ void fill(unsigned* p, size_t n, unsigned seed) { size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; if (i < n) p[i] = seed + (unsigned)i; }
int main() {
    hipMemAllocationProp prop = {};
    prop.type = hipMemAllocationTypePinned; prop.location.type = hipMemLocationTypeDevice; prop.location.id = 0;
    size_t gran = 0;
    CK(hipMemGetAllocationGranularity(&gran, &prop, hipMemAllocationGranularityMinimum));
    printf("granularity %zu\n", gran);
    hipStream_t st; CK(hipStreamCreate(&st));
    const size_t sizes[] = {15000, 65536, 161408, 1 << 20, 5000000};
    for (size_t nb : sizes) for (int end = 0; end < 2; ++end) for (int pinned = 0; pinned < 2; ++pinned) {
        size_t mapped = (nb + gran - 1) / gran * gran;
        void* base; CK(hipMemAddressReserve(&base, mapped + 2 * gran, gran, nullptr, 0));
        hipMemGenericAllocationHandle_t hnd; CK(hipMemCreate(&hnd, mapped, &prop, 0));
        char* lo = (char*)base + gran;
        CK(hipMemMap(lo, mapped, 0, hnd, 0));
        hipMemAccessDesc acc = {}; acc.location = prop.location; acc.flags = hipMemAccessFlagsProtReadWrite;
        CK(hipMemSetAccess(lo, mapped, &acc, 1));
        unsigned* d = (unsigned*)(end ? lo + mapped - (nb + 15) / 16 * 16 : lo);
        size_t n = nb / 4;
        unsigned* hbuf; std::vector<unsigned> pag(n);
        if (pinned) CK(hipHostMalloc((void**)&hbuf, n * 4, hipHostMallocDefault)); else hbuf = pag.data();
        hipLaunchKernelGGL(fill, dim3((n + 255) / 256), dim3(256), 0, st, d, n, 7u);
        CK(hipMemcpyAsync(hbuf, d, n * 4, hipMemcpyDeviceToHost, st));
        CK(hipStreamSynchronize(st));
        size_t bad = 0, first = 0;
        for (size_t i = 0; i < n; ++i) if (hbuf[i] != 7u + (unsigned)i) { if (!bad) first = i; ++bad; }
        for (size_t i = 0; i < n; ++i) hbuf[i] = 99u + (unsigned)i;
        CK(hipMemcpyAsync(d, hbuf, n * 4, hipMemcpyHostToDevice, st));
        unsigned* d2; CK(hipMalloc(&d2, n * 4));
        CK(hipMemcpyAsync(d2, d, n * 4, hipMemcpyDeviceToDevice, st));
        std::vector<unsigned> chk(n);
        CK(hipMemcpyAsync(chk.data(), d2, n * 4, hipMemcpyDeviceToHost, st));
        CK(hipStreamSynchronize(st));
        size_t bad2 = 0;
        for (size_t i = 0; i < n; ++i) if (chk[i] != 99u + (unsigned)i) ++bad2;
        printf("bytes %8zu at_end %d pinned %d: D2H bad %zu (first word %zu)  H2D bad %zu\n", nb, end, pinned, bad, first, bad2);
        CK(hipFree(d2)); if (pinned) CK(hipHostFree(hbuf));
        CK(hipMemUnmap(lo, mapped)); CK(hipMemRelease(hnd)); CK(hipMemAddressFree(base, mapped + 2 * gran));
    }
    return 0;
}
