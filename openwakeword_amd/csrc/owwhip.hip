// owwhip.hip -- host side of libowwhip.so: context, weight packing, kernel launches, C ABI.
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -shared -fPIC owwhip.hip -I../../include -o libowwhip.so
// Interface contract and the reference call sites each entry replaces: include/owwhip.h.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <ctime>
#include <exception>
#include <mutex>
#include <new>
#include <unordered_map>
#include <dlfcn.h>
#include <string>
#include <vector>

#include "owwhip.h"
#include "owwhip_kernels.h"
#include "owwhip_rr.h"
#include "owwhip_hx.h"
#include "owwhip_vad.h"
#include "owwhip_fused.h"

using namespace owk;

namespace {

thread_local std::string g_err;

int fail(int code, const char* fmt, ...) {
    char buf[1024];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    try { g_err = buf; } catch (...) {}                // (the message is best effort; the code is what callers branch on)
    return code;
}

// Nothing throws across the C ABI: every extern "C" body that can allocate (std::vector / std::string packing buffers, new) runs
// between these two, which turn a C++ exception into an error code + message like any other failure.
#define OWW_GUARD_BEGIN try {
#define OWW_GUARD_END                                                                                              \
    } catch (const std::bad_alloc&) { return fail(OWW_ENOMEM, "%s: out of host memory", __func__);                 \
    } catch (const std::exception& e__) { return fail(OWW_ESTATE, "%s: unexpected C++ exception: %s", __func__, e__.what()); \
    } catch (...) { return fail(OWW_ESTATE, "%s: unexpected C++ exception", __func__); }

#define HIPCHK(expr)                                                                         \
    do {                                                                                     \
        hipError_t e__ = (expr);                                                             \
        if (e__ != hipSuccess) return fail(OWW_EHIP, "%s failed: %s", #expr, hipGetErrorString(e__)); \
    } while (0)

// ---- device allocations.  OWW_GUARD_ALLOC=1|2 is a debugging aid: every device buffer of the library then lives in its own
// virtual range (hipMemAddressReserve / hipMemMap) with an UNMAPPED granule on both sides and the buffer pushed against the upper (1)
// or the lower (2) end of its mapping, so that a kernel that reads or writes outside a buffer faults on the spot instead of touching
// whatever hipMalloc happened to place next to it; each allocation prints its address range and the source line that made it.
struct GuardRange { char* base; size_t reserve, mapped; hipMemGenericAllocationHandle_t hnd; };
std::mutex g_guard_mu;
std::unordered_map<void*, GuardRange> g_guard;
int guard_mode() {
    static const int mode = [] { const char* e = getenv("OWW_GUARD_ALLOC"); return e ? atoi(e) : 0; }();
    return mode;
}
hipError_t guard_alloc(void** p, size_t n, int line) {
    int dev = 0;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess) return e;
    hipMemAllocationProp prop = {};
    prop.type = hipMemAllocationTypePinned;
    prop.location.type = hipMemLocationTypeDevice;
    prop.location.id = dev;
    size_t gran = 0;
    if ((e = hipMemGetAllocationGranularity(&gran, &prop, hipMemAllocationGranularityMinimum)) != hipSuccess) return e;
    GuardRange g{};
    g.mapped = (std::max<size_t>(n, 1) + gran - 1) / gran * gran;
    g.reserve = g.mapped + 2 * gran;
    void* base = nullptr;
    if ((e = hipMemAddressReserve(&base, g.reserve, gran, nullptr, 0)) != hipSuccess) return e;
    g.base = static_cast<char*>(base);
    if ((e = hipMemCreate(&g.hnd, g.mapped, &prop, 0)) != hipSuccess) { (void)hipMemAddressFree(base, g.reserve); return e; }
    hipMemAccessDesc acc = {};
    acc.location = prop.location;
    acc.flags = hipMemAccessFlagsProtReadWrite;
    if ((e = hipMemMap(g.base + gran, g.mapped, 0, g.hnd, 0)) != hipSuccess ||
        (e = hipMemSetAccess(g.base + gran, g.mapped, &acc, 1)) != hipSuccess) {
        (void)hipMemRelease(g.hnd); (void)hipMemAddressFree(base, g.reserve); return e;
    }
    char* user = guard_mode() == 2 ? g.base + gran : g.base + gran + g.mapped - (n + 15) / 16 * 16;
    fprintf(stderr, "[owwhip guard] line %d: %zu bytes at [%p, %p), mapped [%p, %p)\n", line, n, (void*)user, (void*)(user + n),
            (void*)(g.base + gran), (void*)(g.base + gran + g.mapped));
    std::lock_guard<std::mutex> lk(g_guard_mu);
    g_guard[user] = g;
    *p = user;
    return hipSuccess;
}
template <class T>
hipError_t dev_alloc(T** p, size_t n, int line = __builtin_LINE()) {
    if (!guard_mode()) return hipMalloc(reinterpret_cast<void**>(p), n);
    return guard_alloc(reinterpret_cast<void**>(p), n, line);
}
hipError_t dev_free(void* p) {
    if (!guard_mode() || !p) return hipFree(p);
    GuardRange g;
    {
        std::lock_guard<std::mutex> lk(g_guard_mu);
        auto it = g_guard.find(p);
        if (it == g_guard.end()) return hipFree(p);
        g = it->second;
        g_guard.erase(it);
    }
    (void)hipDeviceSynchronize();
    const size_t gran = (g.reserve - g.mapped) / 2;
    (void)hipMemUnmap(g.base + gran, g.mapped);
    (void)hipMemRelease(g.hnd);
    return hipMemAddressFree(g.base, g.reserve);
}

// Copies between host memory and a hipMemMap'ed range: the runtime's path for PAGEABLE host memory drops bytes at some sizes /
// alignments (tools/experiments/vmm_copy_probe.hip: 5,000,000 bytes from a range that ends at its mapping's end), its pinned path does
// not -- under OWW_GUARD_ALLOC host <-> device copies therefore go through a page-locked bounce buffer, synchronously.
hipError_t guard_copy(void* dst, const void* src, size_t n, hipMemcpyKind kind, hipStream_t st) {
    void* pin = nullptr;
    hipError_t e = hipHostMalloc(&pin, std::max<size_t>(n, 1), hipHostMallocDefault);
    if (e != hipSuccess) return e;
    if (kind == hipMemcpyHostToDevice) {
        memcpy(pin, src, n);
        e = hipMemcpyAsync(dst, pin, n, kind, st);
        if (e == hipSuccess) e = hipStreamSynchronize(st);
    } else {
        e = hipMemcpyAsync(pin, src, n, kind, st);
        if (e == hipSuccess) e = hipStreamSynchronize(st);
        if (e == hipSuccess) memcpy(dst, pin, n);
    }
    (void)hipHostFree(pin);
    return e;
}
inline hipError_t copy_async(void* dst, const void* src, size_t n, hipMemcpyKind kind, hipStream_t st) {
    if (guard_mode() && n && (kind == hipMemcpyHostToDevice || kind == hipMemcpyDeviceToHost)) return guard_copy(dst, src, n, kind, st);
    return hipMemcpyAsync(dst, src, n, kind, st);
}
inline hipError_t copy_sync(void* dst, const void* src, size_t n, hipMemcpyKind kind) {
    if (guard_mode() && n && (kind == hipMemcpyHostToDevice || kind == hipMemcpyDeviceToHost)) return guard_copy(dst, src, n, kind, nullptr);
    const hipError_t e = hipMemcpy(dst, src, n, kind);
    // (set-up time only: the consumers run on non-blocking streams, which nothing orders behind the legacy stream)
    return e != hipSuccess || kind != hipMemcpyHostToDevice ? e : hipStreamSynchronize(nullptr);
}

struct LayerDef { int kh, kw, cin, cout; };
constexpr int kSmallLaunchWgs = 2 * 256;      // stage / heads launches of at most two workgroups per CU (MI355X: 256 CUs) use the deep weight rings
const LayerDef kLayers[20] = {
    {3, 3, 1, 24},
    {1, 3, 24, 24}, {3, 1, 24, 24},
    {1, 3, 24, 48}, {3, 1, 48, 48}, {1, 3, 48, 48}, {3, 1, 48, 48},
    {1, 3, 48, 72}, {3, 1, 72, 72}, {1, 3, 72, 72}, {3, 1, 72, 72},
    {1, 3, 72, 96}, {3, 1, 96, 96}, {1, 3, 96, 96}, {3, 1, 96, 96},
    {1, 3, 96, 96}, {3, 1, 96, 96}, {1, 3, 96, 96}, {3, 1, 96, 96},
    {3, 1, 96, 96},
};
// new rows x F x C of every layer's output per step (debug layout)
const int kLayerOut[20][3] = {
    {8, 32, 24}, {8, 32, 24}, {8, 32, 24},
    {4, 16, 48}, {4, 16, 48}, {4, 16, 48}, {4, 16, 48},
    {4, 8, 72}, {4, 8, 72}, {4, 8, 72}, {4, 8, 72},
    {2, 4, 96}, {2, 4, 96}, {2, 4, 96}, {2, 4, 96},
    {2, 2, 96}, {2, 2, 96}, {2, 2, 96}, {2, 2, 96},
    {1, 1, 96},
};

// pack [ntaps][cin][cout] for conv_mfma / heads64: out[(ct*KS + s)*64 + lane]
void pack_mfma(const float* w, int ntaps, int cin, int cout, std::vector<float>& out) {
    const int nct = (cout + 15) / 16, ks = ntaps * cin / 4;
    out.assign((size_t)nct * ks * 64, 0.f);
    for (int ct = 0; ct < nct; ++ct)
        for (int tap = 0; tap < ntaps; ++tap)
            for (int cb = 0; cb < cin; cb += 8)
                for (int q = 0; q < 2; ++q) {
                    const int s = (tap * cin + cb) / 4 + q;
                    for (int lane = 0; lane < 64; ++lane) {
                        const int i = lane & 15, j = lane >> 4;
                        const int co = ct * 16 + i, ci = cb + 2 * j + q;
                        out[((size_t)ct * ks + s) * 64 + lane] = co < cout ? w[((size_t)tap * cin + ci) * cout + co] : 0.f;
                    }
                }
}

// register-resident layout (owwhip_rr.h): out[((((oct*ntaps + tap)*ncti + ct)*64 + lane)*4 + e], lane = (i, j):
// weight of input channel 16ct+4j+e and output channel 16oct+i
// a half-filled last input-channel tile (cin % 16 == 8) is consumed in pack_half order: two k-steps, lane (i, j) of
// k-step e' < 2 carrying channel 16ct + 4(j&1) + 2e' + (j>>1); k-steps 2,3 of that block are not executed
void pack_rr(const float* w, int ntaps, int cin, int cout, std::vector<float>& out) {
    const int ncti = (cin + 15) / 16, ncto = (cout + 15) / 16;
    const bool half_in = cin % 16 == 8;
    out.assign((size_t)ncto * ntaps * ncti * 64 * 4, 0.f);
    for (int oct = 0; oct < ncto; ++oct)
        for (int tap = 0; tap < ntaps; ++tap)
            for (int ct = 0; ct < ncti; ++ct)
                for (int lane = 0; lane < 64; ++lane)
                    for (int e = 0; e < 4; ++e) {
                        const int i = lane & 15, j = lane >> 4;
                        int ci = ct * 16 + 4 * j + e;
                        if (half_in && ct == ncti - 1) ci = e < 2 ? ct * 16 + 4 * (j & 1) + 2 * e + (j >> 1) : cin;
                        const int co = oct * 16 + i;
                        if (ci < cin && co < cout)
                            out[((((size_t)oct * ntaps + tap) * ncti + ct) * 64 + lane) * 4 + e] = w[((size_t)tap * cin + ci) * cout + co];
                    }
}

// the f16 halves of 2^8 * w must stay finite: |w| < 255 (trained linear weights are orders of magnitude below); VAD network
bool hx_in_range(const float* w, size_t n) {
    for (size_t i = 0; i < n; ++i) if (!(std::fabs(w[i]) * owh::WSCALE < 65000.f)) return false;
    return true;
}
// heads: the power-of-two exponent e that puts the largest |w| of a matrix at 2^11 .. 2^12 (its f16 halves then carry 22 bits for
// every weight down to 2^-13 of the largest; no weight magnitude is refused); -1000 when a weight is not finite
int hx_weight_exp(const float* w, size_t n) {
    float m = 0.f;
    for (size_t i = 0; i < n; ++i) { if (!std::isfinite(w[i])) return -1000; m = std::max(m, std::fabs(w[i])); }
    int e2 = 0;
    if (m > 0.f) std::frexp(m, &e2);                 // m < 2^e2
    return std::min(100, std::max(-100, 12 - e2));
}

// Half channel tiles of the fp16-split family (24 = 16 + 8, 72 = 64 + 8 channels).  The MFMA D layout puts row 4j + e of a tile into
// register e of lane group j; with the natural order the 8 real channels of the last tile would sit in registers 0..3 of lane groups
// 0, 1 and no register would be all padding.  The f16-split kernels instead place them in registers 0, 1 of ALL four lane groups:
//     row 4j + e of the half tile  <->  channel 16 ct + 2j + e   (e < 2),   rows with e >= 2: padding
// so that registers 2, 3 of that tile are identically zero and their epilogue, operand split, loads and stores can be skipped.
// The same order is the K order of the layer that consumes the tile (operand halves q % 4 = e of lane group g = j), of the folded
// BatchNorm arrays and of the debug dump (owh::dump_tile_ht).
inline int hx_row_channel(int tile, int row, int C) {            // channel in row `row` (0..15) of channel tile `tile`, or -1
    const bool half = C % 16 == 8 && tile == (C + 15) / 16 - 1;
    if (!half) { const int c = tile * 16 + row; return c < C ? c : -1; }
    const int j = row >> 2, e = row & 3;
    return e < 2 ? tile * 16 + 2 * j + e : -1;
}

// one weight as an f16 (hi, lo) pair: w * mul formed in double, hi = f16(v), lo = f16(v - hi) (22 bits together).
// mul = colmul[cout] for the embedding CNN (folded BatchNorm scale x the layer's activation-scale ratio, see fold_cnn), else 2^8.
struct HxFold {
    const double* colmul = nullptr;      // per output channel; nullptr = owh::WSCALE for every channel
    double absmax = 0.0;                 // largest |w * mul| seen (range check by the caller)
    inline void split(float w, int co, _Float16& hi, _Float16& lo) {
        const double v = (double)w * (colmul ? colmul[co] : (double)owh::WSCALE);
        absmax = std::max(absmax, std::fabs(v));
        hi = (_Float16)v;
        lo = (_Float16)(v - (double)hi);
    }
};

// fp16-split operand order (owwhip_hx.h): blocks [oct][tap][ks][part hi/lo] of 64 lanes x 8 halves; lane (i, g), half q
// <-> weight of the input channel in row 4g + q%4 of channel tile 2ks + q/4 and the output channel in row i of tile oct
// (hx_row_channel)
// rem2 (cin = odd number of FULL channel tiles, stage B's 48): the last k-step in the two-MFMA form of owh::split_dup -- its empty
// half carries the same channels again: block part 0 = (wh | wh), part 1 = (wl | 0)
void pack_hx(const float* w, int ntaps, int cin, int cout, std::vector<float>& out, HxFold* fold = nullptr, bool rem2 = false) {
    HxFold dflt; if (!fold) fold = &dflt;
    const int ks_n = ((cin + 15) / 16 + 1) / 2, ncto = (cout + 15) / 16;
    rem2 = rem2 && ((cin + 15) / 16) % 2 == 1 && cin % 16 == 0;
    std::vector<_Float16> hbuf((size_t)ncto * ntaps * ks_n * 2 * 64 * 8, (_Float16)0.f);
    for (int oct = 0; oct < ncto; ++oct)
        for (int tap = 0; tap < ntaps; ++tap)
            for (int ks = 0; ks < ks_n; ++ks)
                for (int lane = 0; lane < 64; ++lane)
                    for (int q = 0; q < 8; ++q) {
                        const int i = lane & 15, g = lane >> 4;
                        const bool dup = rem2 && ks == ks_n - 1 && q >= 4;            // the empty half of the remainder k-step
                        const int ci = dup ? hx_row_channel(2 * ks, 4 * g + q % 4, cin) :
                                       (2 * ks + q / 4 < (cin + 15) / 16 ? hx_row_channel(2 * ks + q / 4, 4 * g + q % 4, cin) : -1);
                        const int co = hx_row_channel(oct, i, cout);
                        if (ci < 0 || co < 0) continue;
                        _Float16 hi, lo;
                        fold->split(w[((size_t)tap * cin + ci) * cout + co], co, hi, lo);
                        const size_t blk = (((size_t)oct * ntaps + tap) * ks_n + ks) * 2;
                        hbuf[(blk * 64 + lane) * 8 + q] = hi;
                        hbuf[((blk + 1) * 64 + lane) * 8 + q] = dup ? (_Float16)0.f : lo;
                    }
    out.assign(hbuf.size() / 2, 0.f);
    memcpy(out.data(), hbuf.data(), hbuf.size() * sizeof(_Float16));
}
// K-merged 3x1 (time) layer of a stage with an odd number of channel tiles (owh::conv_time_hxm): per output tile the blocks
// [tap][full k-step][part], then [merged k-step][part]; merged k-step mk, lane (i, g), half q: pair index pi = 4 mk + q/2 carries
// tap pi / NPR, pair v = pi % NPR of the remainder tile, i.e. its row 4g + 2v + q%2
void pack_hx_tm(const float* w, int cin, int cout, std::vector<float>& out, HxFold* fold) {
    const int ncti = (cin + 15) / 16, ncto = (cout + 15) / 16;
    const bool half = cin % 16 == 8;
    const int ksf = ncti / 2, npr = half ? 1 : 2, nmk = (3 * npr + 3) / 4, nb = (3 * ksf + nmk) * 2;
    std::vector<_Float16> hbuf((size_t)ncto * nb * 64 * 8, (_Float16)0.f);
    auto put = [&](size_t blk, int lane, int q, int tap, int ci, int co) {
        if (ci < 0 || co < 0) return;
        _Float16 hi, lo;
        fold->split(w[((size_t)tap * cin + ci) * cout + co], co, hi, lo);
        hbuf[(blk * 64 + lane) * 8 + q] = hi;
        hbuf[((blk + 1) * 64 + lane) * 8 + q] = lo;
    };
    for (int oct = 0; oct < ncto; ++oct)
        for (int lane = 0; lane < 64; ++lane)
            for (int q = 0; q < 8; ++q) {
                const int i = lane & 15, g = lane >> 4, co = hx_row_channel(oct, i, cout);
                for (int tap = 0; tap < 3; ++tap)
                    for (int ks = 0; ks < ksf; ++ks)
                        put(((size_t)oct * nb) + (tap * ksf + ks) * 2, lane, q, tap, hx_row_channel(2 * ks + q / 4, 4 * g + q % 4, cin), co);
                for (int mk = 0; mk < nmk; ++mk) {
                    const int pi = 4 * mk + q / 2;
                    if (pi >= 3 * npr) continue;
                    const int tap = pi / npr, v = pi % npr;
                    put(((size_t)oct * nb) + (3 * ksf + mk) * 2, lane, q, tap, hx_row_channel(ncti - 1, 4 * g + 2 * v + q % 2, cin), co);
                }
            }
    out.assign(hbuf.size() / 2, 0.f);
    memcpy(out.data(), hbuf.data(), hbuf.size() * sizeof(_Float16));
}
// per-channel array (folded BatchNorm scale / shift) in the row order of the f16-split tiles, zero in padding rows
void pad_hx_rows(const float* v, int C, float mul, std::vector<float>& out) {
    const int nct = (C + 15) / 16;
    out.assign((size_t)nct * 16, 0.f);
    for (int t = 0; t < nct; ++t)
        for (int r = 0; r < 16; ++r) { const int c = hx_row_channel(t, r, C); if (c >= 0) out[t * 16 + r] = v[c] * mul; }
}
// heads layer 1 (owh::heads_hx_kernel): k-step major [K/32][NH/16][part][64][8]; lane (i, g), half q <-> input 32ks + 8g + q
void pack_hx_w1(const float* wcat /*[K][NH]*/, int K, int NH, const double* colmul /*[NH]*/, std::vector<float>& out) {
    const int nks = K / 32, nct = NH / 16;
    std::vector<_Float16> hbuf((size_t)nks * nct * 2 * 64 * 8, (_Float16)0.f);
    for (int ks = 0; ks < nks; ++ks)
        for (int ct = 0; ct < nct; ++ct)
            for (int lane = 0; lane < 64; ++lane)
                for (int q = 0; q < 8; ++q) {
                    const int i = lane & 15, g = lane >> 4;
                    const double v = (double)wcat[(size_t)(32 * ks + 8 * g + q) * NH + 16 * ct + i] * colmul[16 * ct + i];
                    const _Float16 hi = (_Float16)v, lo = (_Float16)(v - (double)hi);
                    const size_t blk = ((size_t)ks * nct + ct) * 2;
                    hbuf[(blk * 64 + lane) * 8 + q] = hi;
                    hbuf[((blk + 1) * 64 + lane) * 8 + q] = lo;
                }
    out.assign(hbuf.size() / 2, 0.f);
    memcpy(out.data(), hbuf.data(), hbuf.size() * sizeof(_Float16));
}
// conv0 (3x3, one input channel, K = 9) in the K-folded form of owh::hstageA_stream: the three products of the f16 split share ONE
// k-step -- k-slot 8g + q of lane (i, g): slots 0..8 = wh[tap] (against xh), 9..17 = wl[tap] (against xh), 18..26 = wh[tap] (against
// xl), 27..31 = 0.  One 1 KB block per output-channel tile.
void pack_hx_conv0(const float* w /*[9][24]*/, std::vector<float>& out, HxFold* fold) {
    std::vector<_Float16> hbuf((size_t)2 * 64 * 8, (_Float16)0.f);
    for (int oct = 0; oct < 2; ++oct)
        for (int lane = 0; lane < 64; ++lane)
            for (int q = 0; q < 8; ++q) {
                const int i = lane & 15, g = lane >> 4, slot = 8 * g + q, co = hx_row_channel(oct, i, 24);
                if (slot >= 27 || co < 0) continue;
                _Float16 hi, lo;
                fold->split(w[(slot % 9) * 24 + co], co, hi, lo);
                hbuf[((size_t)oct * 64 + lane) * 8 + q] = (slot / 9 == 1) ? lo : hi;
            }
    out.assign(hbuf.size() / 2, 0.f);
    memcpy(out.data(), hbuf.data(), hbuf.size() * sizeof(_Float16));
}

// stage A's two 24 -> 24 layers (owh::hstageA_stream): channel tile 0 as in pack_hx -- blocks [tap][part] -- and the HALF tile (channels
// 16..23 in rows 4j + e, e < 2) with a second tap stacked into its free rows 4j + e, e >= 2:
//   conv1 (1x3): blocks (W2 | W0), (W1 | 0), (W0 | W2): the side chain of a parity tile rides in the main chain's registers 2, 3;
//   conv2 (3x1): blocks (W_i | W_{i-1}) for input row i = 0..3 of [history 0, history 1, row 2q, row 2q+1]: registers 0, 1 accumulate
//                output row 2q, registers 2, 3 output row 2q+1.
void pack_hx_stage_a(const float* w /*[3][24][24]*/, int layer, std::vector<float>& out, HxFold* fold) {
    const int nblk = layer == 1 ? 12 : 14;
    std::vector<_Float16> hbuf((size_t)nblk * 64 * 8, (_Float16)0.f);
    auto put = [&](int blk, int lane, int q, int tap, int ci, int co) {
        if (tap < 0 || tap > 2 || ci < 0 || co < 0) return;
        _Float16 hi, lo;
        fold->split(w[((size_t)tap * 24 + ci) * 24 + co], co, hi, lo);
        hbuf[((size_t)blk * 64 + lane) * 8 + q] = hi;
        hbuf[((size_t)(blk + 1) * 64 + lane) * 8 + q] = lo;
    };
    for (int lane = 0; lane < 64; ++lane)
        for (int q = 0; q < 8; ++q) {
            const int i = lane & 15, g = lane >> 4, ci = hx_row_channel(q / 4, 4 * g + q % 4, 24);
            for (int tap = 0; tap < 3; ++tap) put(tap * 2, lane, q, tap, ci, i);                       // channel tile 0: output channel i
            const int j = i >> 2, e = i & 3, co = 16 + 2 * j + (e & 1);
            if (layer == 1) {
                const int lo_tap[3] = {2, 1, 0}, hi_tap[3] = {0, -1, 2};
                for (int v = 0; v < 3; ++v) put(6 + v * 2, lane, q, e < 2 ? lo_tap[v] : hi_tap[v], ci, co);
            } else {
                for (int r = 0; r < 4; ++r) put(6 + r * 2, lane, q, e < 2 ? r : r - 1, ci, co);
            }
        }
    out.assign(hbuf.size() / 2, 0.f);
    memcpy(out.data(), hbuf.data(), hbuf.size() * sizeof(_Float16));
}

struct HostBuf {                      // host image of the device weight buffer (256-byte aligned pieces)
    std::vector<float> data;
    size_t add(const float* p, size_t n) {
        const size_t off = (data.size() + 63) / 64 * 64;
        data.resize(off + n);
        if (p) memcpy(data.data() + off, p, n * sizeof(float));
        return off;
    }
    size_t add(const std::vector<float>& v) { return add(v.data(), v.size()); }
};

struct NetHost {
    int hidden, n_out, has_ln, T, final_act, head, role, out_col;
    int n_blocks;                     // hidden blocks behind the first layer (train.py:73: Net's n_blocks; 1 in every released model)
    const float *w1, *b1, *ln1g, *ln1b, *w2, *b2, *ln2g, *ln2b, *w3, *b3;   // into the owning head blob (w2 .. ln2b: block 0)
    const float* blocks;              // the n_blocks blocks back to back: w[H][H] b[H] (g[H] be[H])
    const float* rnn = nullptr;       // model_type "rnn" (kind 3): the head's blob (owk::heads_rnn_kernel); its size in rnn_floats
    size_t rnn_floats = 0;
    int hx_e1 = 0, hx_e2 = 0, hx_e3 = 0;   // fp16-split heads: power-of-two scales of w1 / w2 (/ w3: wide form) (hx_weight_exp)
};
struct HeadHost {
    int kind, T, hidden, n_out, has_ln, n_blocks;
    std::vector<float> blob;
    int out_col;
};
struct FastGroup {
    int T, NH, n_nets;
    std::vector<int> nets;            // indices into all nets
    NetDesc* d_nets = nullptr;
    const float* d_w1pk = nullptr;
    const float* d_b1cat = nullptr;
    const float* d_w1hx = nullptr;    // fp16-split k-step-major layer-1 weights (heads_hx_kernel)
    std::vector<const float*> d_w2hx; // per net
    // fp16-split fast path: per net the seven per-unit arrays (b1, ln1 g / b, b2, ln2 g / b, w3) padded with zeros to the kernel's 64
    // hidden units -- a narrower net (the reference's training pipeline defaults to 32: examples/custom_model.yml:89) runs as a
    // 64-unit net whose padding units are identically zero and are left out of the LayerNorm statistics (owh::HeadHxNet::hidden)
    std::vector<const float*> d_pad;  // per net: 7 x 64 floats (wide form: 6 x 128 + 16, see below)
    // wide form (owh::heads_wide_tail: nets of up to 128 hidden units / 8 outputs, sigmoid or ReLU + softmax -- the multiclass `timer`
    // model): ht = 8 hidden tiles per net, at most two nets per launch; d_pad = b1, ln1 g / b, b2, ln2 g / b padded to 128 units + b3
    // padded to 16 outputs; the output layer as a third f16-split matrix
    int ht = 4;
    std::vector<const float*> d_w3hx; // per net
};

constexpr int N_STATE = 11;
// per-stream floats of every state array: hist_mel, hist2, B:b,d  C:b,d  D:b,d  E:b,d  hist19
const int kStateLenLds[N_STATE] = {64, 1536, 1536, 1536, 1152, 1152, 768, 768, 384, 384, 192};
// register-resident layout: channel tiles padded to 16 (24->32, 72->80), streams of one wave interleaved per block
const int kStateLenRr[N_STATE] = {64, 2048, 1536, 1536, 1280, 1280, 768, 768, 384, 384, 384};
const int kStateSpgRr[N_STATE] = {1, 1, 1, 1, 2, 2, 4, 4, 8, 8, 8};
const int kStateFposRr[N_STATE] = {16, 16, 16, 16, 8, 8, 4, 4, 2, 2, 1};
// pooled activations handed from stage to stage, floats per stream: xA, xB, xC, xD
const int kXLenLds[4] = {1536, 1536, 576, 384};
const int kXLenRr[4] = {2048, 1536, 640, 384};
constexpr int DBG_FLOATS = 3 * 6144 + 4 * 3072 + 4 * 2304 + 4 * 768 + 4 * 384 + 96;

struct EventRec { hipEvent_t a, b; int cls; };

}  // namespace

struct oww_ctx {
    oww_config cfg{};
    int S = 0, Spad = 0, kmax = 1, TR = 16, NL = 0;
    hipStream_t stream = nullptr;
    bool own_stream = false, committed = false, mfma = true;
    bool rr = true;                   // register-resident CNN kernels (owwhip_rr.h); false = LDS-tiled kernels (MFMA or VALU)
    bool hx = false;                  // fp16-split form of the register-resident kernels (owwhip_hx.h); shares rr's layouts
    const int* state_len = kStateLenRr;
    // host side weights as loaded
    std::vector<float> mel_blob, emb_blob;
    std::vector<HeadHost> heads;
    std::vector<NetHost> nets;
    std::vector<std::pair<int, int>> head_nets;      // [begin,end) into nets per head
    // device weights
    float* d_w = nullptr;
    const float *d_hann = nullptr, *d_taps = nullptr;
    const int* d_mstart = nullptr;
    const int* d_meloff = nullptr; const unsigned* d_meldst = nullptr;   // fused front end: compact tap table (oww_commit)
    const float* d_conv[20] = {};     // layer 0: natural [9][24]; 1..19: packed (mfma) or natural (valu)
    const float* d_scale[20] = {};
    const float* d_shift[20] = {};
    const float* d_conv0_mfma = nullptr;   // conv0 in MFMA k-step order (shared by the LDS-MFMA and register-resident kernels)
    NetDesc* d_allnets = nullptr;
    std::vector<NetDesc> host_descs;  // device pointers of every net's arrays (host copy of d_allnets)
    std::vector<FastGroup> groups;
    std::vector<int> generic_nets;    // indices into nets (with verifier right after its primary)
    std::vector<int> rnn_nets;        // recurrent heads (train.py:85-98): owk::heads_rnn_kernel, one launch per head
    NetDesc* d_generic = nullptr;
    int generic_hmax = 0;
    int generic_spw = 0;              // OWW_GENERIC_SPW: 4 / 16 pins the generic heads kernel's shape, 0 = by launch size
    // state
    float* d_state[N_STATE] = {};
    float* d_tmpl[N_STATE] = {};
    float *d_xA = nullptr, *d_xB = nullptr, *d_xC = nullptr, *d_xD = nullptr;
    float *d_mel = nullptr, *d_feat = nullptr, *d_emb = nullptr, *d_raw = nullptr, *d_scores = nullptr, *d_ring = nullptr;
    // host-fed pipeline (oww_submit / oww_collect): two steps in flight, uploads and score downloads on their own streams
    struct IngestSlot {
        int16_t* d_pcm = nullptr; float* d_scores = nullptr; float* h_scores = nullptr;
        uint8_t* h_on = nullptr;     // page-locked staging of a masked submit's participation mask
        hipEvent_t up = nullptr, done = nullptr, down = nullptr;
        bool busy = false;
    } slot[2];
    hipStream_t up_stream = nullptr, down_stream = nullptr;
    uint64_t n_submit = 0, n_collect = 0;
    const float* mel_src = nullptr;   // when set: the CNN reads its mel rows from here instead of d_mel (oww_embed_clips)
    float* d_featinit = nullptr;
    float* d_dbg = nullptr;
    long long* d_prof = nullptr;     // [4 stages][16 waves][16 marks], allocated when OWW_PROF_BLOCK is set
    int prof_block = -1;
    uint32_t *d_nfeat = nullptr, *d_npred = nullptr;
    float* d_vadring = nullptr; uint32_t* d_nvad = nullptr; float* d_vadin = nullptr; float vad_threshold = 0.f;   // VAD gate (oww_push_vad)
    // voice-activity stand-in network on the device (oww_load_vad; owwhip_vad.h)
    std::vector<float> vad_blob;
    bool vad = false;
    const float *d_vad_hann = nullptr, *d_vad_encw = nullptr, *d_vad_encb = nullptr, *d_vad_lstmw = nullptr, *d_vad_lstmb = nullptr, *d_vad_wd = nullptr;
    float vad_bd = 0.f, vad_gain = 50.f;
    float *d_vadx = nullptr, *d_vadhc = nullptr, *d_vadlast = nullptr;
    int16_t *d_tail = nullptr, *d_pcm = nullptr;
    int16_t* d_long = nullptr; size_t long_cap = 0;      // oww_step with n_chunks > max_chunks: the whole call's PCM when it arrives from the host
    float* d_callmax = nullptr;                          // ... and the call's per-stream mel maximum (launch_step_long)
    int* h_range = nullptr;          // sticky f16-range flag: one page-locked, device-mapped word the f16-split kernels raise
    int* d_range = nullptr;          // the same word as the kernels address it
    void* d_rs = nullptr; size_t rs_bytes = 0;     // oww_resample scratch: [taps | in | out] as needed
    std::vector<float> rs_taps;                    // the padded filter bank of the last oww_resample call (upload source)
    uint8_t* d_on = nullptr;         // oww_step_masked: [Spad] participation mask of the step being launched (pad streams 0)
    const uint8_t* on_now = nullptr; // = d_on (or the caller's device mask) while a masked step is being launched, else nullptr
    // masked step with few participants (host-resident mask, <= half of the streams): lists of the participating streams and of the
    // stage groups (2 / 4 / 8 / 16 streams) that hold one; the launches of stages B..E, the heads and the VAD LSTM then cover only
    // those (build_active_lists).  [0] = stream ids (stage B's groups and the heads' positions), [1] C, [2] D, [3] E, [4] VAD LSTM
    int* d_lists = nullptr; int* h_lists[2] = {nullptr, nullptr}; hipEvent_t lists_ev[2] = {nullptr, nullptr}; size_t lists_cap = 0; unsigned lists_turn = 0;
    const int* gl_now[5] = {}; int gn_now[5] = {}; bool lists_now = false;
    int k_last = 1;                  // n_chunks of the last step (row stride of d_mel)
    // f16-split family: the output of layer l is carried multiplied by 2^hx_e[l], its input arrives multiplied by 2^hx_ein[l]
    // (oww_commit: calibrate_hx; owwhip_hx.h act1).  Inside a stage hx_ein[l] = hx_e[l - 1]; the pooled hand-over between two stages
    // (and the pooled input of conv19) is re-scaled by 2^hx_xexp[stage], the embedding by 2^-hx_e[19] when it is stored.
    int hx_e[20] = {}, hx_ein[20] = {}, hx_xexp[5] = {};
    float hx_absmax[20] = {};        // largest |activation| of each layer in the calibration run (exact-fp32 kernels)
    std::vector<int16_t> cal_user;   // oww_set_calibration: caller's calibration audio as [n_seg][CAL_T * 1280] segments
    // builds with -DOWH_DEEP_RING only (round 6 experiment, profiles/r06_deep_ring_c1.txt: bit-identical, no faster at 4,096 / 16,384 streams):
    // launches of at most this many workgroups run a weight ring of NS = 4 / 5 slots, three / four chunks in flight; OWW_DEEP_WGS
    int deep_wgs = 0;
    int small_wgs = kSmallLaunchWgs, small_wgs_heads = kSmallLaunchWgs;   // A/B aids: OWW_SMALL_WGS / OWW_SMALL_WGS_HEADS (0 = never the deep rings)
    bool ring3_always[4] = {};       // A/B aid (OWW_RING3_ALWAYS="BCDE"): the three-slot weight ring of stages B..E at any launch size
    int hx_efeat = 0;                // heads: the features enter the first GEMM multiplied by 2^hx_efeat (largest probe |embedding| at 2^9..2^10)
    float hx_selftest_err = 0.f, hx_selftest_ref = 0.f, hx_selftest_score_err = 0.f;   // commit-time f16-split vs exact-fp32 comparison
    bool fuse = false;               // f16-split family: mel front end fused into stage A for one-chunk streaming steps (owwhip_fused.h)
    const int16_t* fuse_pcm = nullptr;   // set by launch_step for the duration of a fused step
    // custom verifiers on the device (oww_set_verifier)
    float *d_verw = nullptr, *d_verb = nullptr, *d_verthr = nullptr; int* d_verT = nullptr;
    int ver_stride = 0, n_verifiers = 0;
    std::vector<int> ver_T;
    bool post_in_heads = false;          // one group of sigmoid heads covers every label: post-processing + counter advance ride in the heads launch
    bool post_in_heads_now = false;      // ... for the step being launched (one-chunk steps only)
    float* d_save = nullptr; size_t save_floats = 0;      // streaming state parked by oww_embed / oww_embed_clips
    int* d_ids = nullptr;
    int ids_cap = 0;
    int *d_patience = nullptr;
    float* d_threshold = nullptr;
    int debounce_frames = 0;
    // block-pipelined step: the one-chunk fused step of a large handle is launched as n_blocks stream ranges on their own HIP streams
    // (forked from / joined to the handle's stream with events), so that one block's latency-bound phases and partly filled last
    // wave rounds overlap the other block's kernels.  blk_s0 / blk_s1 = the range being launched (0 / 0 = everything).
    int n_blocks = 1; hipStream_t blk_stream[4] = {}; hipEvent_t blk_fork = nullptr, blk_done[4] = {};
    int blk_s0 = 0, blk_s1 = 0;
    // RCCL communicator of oww_comm_init (multi-GPU delivery of results without Python)
    void* comm = nullptr; int comm_rank = 0, comm_world = 1;
    // timing
    bool timing = false;
    std::vector<EventRec> ev;
    size_t ev_used = 0;
    double t_ms[OWW_N_KERNEL_CLASSES] = {};
    int64_t t_n[OWW_N_KERNEL_CLASSES] = {};
    // graph
    bool want_graph = false;
    hipGraph_t graph = nullptr;
    hipGraphExec_t graph_exec = nullptr;
};

namespace {

int flush_events(oww_ctx* h) {
    if (!h->ev_used) return 0;
    HIPCHK(hipStreamSynchronize(h->stream));
    for (size_t i = 0; i < h->ev_used; ++i) {
        float ms = 0.f;
        HIPCHK(hipEventElapsedTime(&ms, h->ev[i].a, h->ev[i].b));
        h->t_ms[h->ev[i].cls] += ms;
        h->t_n[h->ev[i].cls] += 1;
    }
    h->ev_used = 0;
    return 0;
}

struct Timed {          // RAII: records a start event now and a stop event at scope exit
    oww_ctx* h;
    EventRec* r = nullptr;
    Timed(oww_ctx* h_, int cls) : h(h_) {
        if (!h->timing) return;
        if (h->ev_used == h->ev.size()) {
            if (h->ev.size() >= 8192) { flush_events(h); }
            else {
                EventRec e{};
                if (hipEventCreate(&e.a) != hipSuccess || hipEventCreate(&e.b) != hipSuccess) return;
                h->ev.push_back(e);
            }
        }
        r = &h->ev[h->ev_used++];
        r->cls = cls;
        (void)hipEventRecord(r->a, h->stream);
    }
    ~Timed() { if (r) (void)hipEventRecord(r->b, h->stream); }
};

template <class K>
int set_lds(K kernel, int bytes) {
    HIPCHK(hipFuncSetAttribute(reinterpret_cast<const void*>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, bytes));
    return 0;
}

// ---- CNN over the first n_active streams, mel rows at d_mel + s*mel_stride + mel_off ----------------
template <bool MFMA>
int run_cnn_t(oww_ctx* h, int n_active, int mel_stride, int mel_off) {
    hipStream_t st = h->stream;
    const bool dbg = h->d_dbg != nullptr;
    int off = 0;
    int dbg_off[20];
    for (int l = 0; l < 20; ++l) { dbg_off[l] = off; off += kLayerOut[l][0] * kLayerOut[l][1] * kLayerOut[l][2]; }
    {
        StageAParams p{};
        p.mel = h->mel_src ? h->mel_src : h->d_mel; p.mel_stride = mel_stride; p.mel_off = mel_off;
        p.hist_mel = h->d_state[0]; p.hist2 = h->d_state[1];
        p.w0 = h->d_conv[0]; p.w1 = h->d_conv[1]; p.w2 = h->d_conv[2];
        for (int i = 0; i < 3; ++i) { p.scale[i] = h->d_scale[i]; p.shift[i] = h->d_shift[i]; p.dbg_off[i] = dbg_off[i]; }
        p.xout = h->d_xA; p.dbg = dbg ? h->d_dbg : nullptr; p.dbg_stride = DBG_FLOATS;
        Timed t(h, 1);
        hipLaunchKernelGGL(stageA_kernel<MFMA>, dim3(n_active), dim3(CfgA::NT), CfgA::LDS_BYTES, st, p);
    }
    auto fill = [&](StageParams& p, const float* xin, float* xout, int first_layer, int sb, int sd) {
        p.xin = xin; p.xout = xout; p.hist_b = h->d_state[sb]; p.hist_d = h->d_state[sd];
        for (int i = 0; i < 4; ++i) {
            p.w[i] = h->d_conv[first_layer + i]; p.scale[i] = h->d_scale[first_layer + i];
            p.shift[i] = h->d_shift[first_layer + i]; p.dbg_off[i] = dbg_off[first_layer + i];
        }
        p.dbg = dbg ? h->d_dbg : nullptr; p.dbg_stride = DBG_FLOATS;
        p.prof = h->d_prof ? h->d_prof + (first_layer / 4) * 256 : nullptr;      // first_layer 3,7,11,15 -> slot 0..3
        p.prof_block = h->prof_block;
    };
    {
        StageParams p{}; fill(p, h->d_xA, h->d_xB, 3, 2, 3);
        Timed t(h, 2);
        hipLaunchKernelGGL((stage_kernel<CfgB, MFMA, false>), dim3((n_active + CfgB::B - 1) / CfgB::B), dim3(CfgB::NT), CfgB::LDS_BYTES, st, p);
    }
    {
        StageParams p{}; fill(p, h->d_xB, h->d_xC, 7, 4, 5);
        Timed t(h, 3);
        hipLaunchKernelGGL((stage_kernel<CfgC, MFMA, false>), dim3((n_active + CfgC::B - 1) / CfgC::B), dim3(CfgC::NT), CfgC::LDS_BYTES, st, p);
    }
    {
        StageParams p{}; fill(p, h->d_xC, h->d_xD, 11, 6, 7);
        Timed t(h, 4);
        hipLaunchKernelGGL((stage_kernel<CfgD, MFMA, false>), dim3((n_active + CfgD::B - 1) / CfgD::B), dim3(CfgD::NT), CfgD::LDS_BYTES, st, p);
    }
    {
        StageParams p{}; fill(p, h->d_xD, nullptr, 15, 8, 9);
        p.hist19 = h->d_state[10]; p.w19 = h->d_conv[19]; p.feat = h->d_feat; p.emb = h->d_emb; p.nfeat = h->d_nfeat; p.TR = h->TR;
        p.dbg_off[4] = dbg_off[19];
        Timed t(h, 5);
        hipLaunchKernelGGL((stage_kernel<CfgE, MFMA, true>), dim3((n_active + CfgE::B - 1) / CfgE::B), dim3(CfgE::NT), CfgE::LDS_BYTES, st, p);
    }
    HIPCHK(hipGetLastError());
    return 0;
}

// register-resident kernels (owwhip_rr.h): every wave independent, groups of 1 / 1 / 2 / 4 / 8 streams per wave
template <bool DBG, bool HX>
int run_cnn_rr(oww_ctx* h, int n_active, int mel_stride, int mel_off) {
    using namespace owr;
    hipStream_t st = h->stream;
    int off = 0, dbg_off[20];
    for (int l = 0; l < 20; ++l) { dbg_off[l] = off; off += kLayerOut[l][0] * kLayerOut[l][1] * kLayerOut[l][2]; }
    {
        RAParams p{};
        p.mel = h->mel_src ? h->mel_src : h->d_mel; p.mel_stride = mel_stride; p.mel_off = mel_off;
        p.hist_mel = h->d_state[0]; p.hist2 = h->d_state[1];
        p.w0 = h->d_conv[0]; p.w1 = h->d_conv[1]; p.w2 = h->d_conv[2];
        for (int i = 0; i < 3; ++i) { p.scale[i] = h->d_scale[i]; p.shift[i] = h->d_shift[i]; p.dbg_off[i] = dbg_off[i]; }
        p.xout = h->d_xA; p.n_streams = n_active; p.S = h->Spad;
        p.dbg = DBG ? h->d_dbg : nullptr; p.dbg_stride = DBG_FLOATS;
        for (int i = 0; i < 3; ++i) { p.clampv[i] = -0.4f * ldexpf(1.f, h->hx_e[i]); p.dbg_mul[i] = ldexpf(1.f, -h->hx_e[i]); }
        p.xmul = ldexpf(1.f, h->hx_xexp[0]);
        p.range_flag = HX ? h->d_range : nullptr;
        p.stream_on = h->on_now;
        const int grid = std::min((n_active + 3) / 4, 768);           // persistent: 3 workgroups of 4 waves per CU
        Timed t(h, 1);
        if (HX && h->fuse_pcm) {
            // mel front end + stage A in one launch: PCM in, pooled stage-A activations out (BASELINE configs[2] "mel+embedding fused")
            owf::MelAParams q{};
            q.a = p; q.a.mel = nullptr; q.a.n_streams = h->S;          // the PCM buffer holds the S real streams only
            if (h->blk_s1 > 0) { q.a.s_base = h->blk_s0; q.a.n_streams = std::min(h->S, h->blk_s1) - h->blk_s0; }
            q.pcm = h->fuse_pcm; q.tail = h->d_tail; q.nfeat = h->d_nfeat; q.hann = h->d_hann; q.mel_start = h->d_mstart; q.mel_taps = h->d_taps; q.mel_off = h->d_meloff; q.mel_dst = h->d_meldst;
            q.mel_out = h->cfg.debug_layers ? h->d_mel : nullptr;
            const int per_cu = std::max(1, std::min(12 / owf::FA_WG, 163840 / owf::FA_LDS_BYTES));     // persistent: 12 waves per CU
            const int g2 = std::max(1, std::min((q.a.n_streams + owf::FA_WG - 1) / owf::FA_WG, 256 * per_cu));
            hipLaunchKernelGGL(owf::hmelA_kernel<DBG>, dim3(g2), dim3(64 * owf::FA_WG), owf::FA_LDS_BYTES, st, q);
        }
        else if (HX) hipLaunchKernelGGL(owh::hstageA_kernel<DBG>, dim3(std::min((n_active + 3) / 4, 256 * OWH_WPS_A)), dim3(256), 0, st, p);
        else hipLaunchKernelGGL(rstageA_kernel<DBG>, dim3(grid), dim3(256), 0, st, p);
    }
    auto fill = [&](RStageParams& p, const float* xin, float* xout, int first_layer, int sb, int sd, int spt) {
        p.xin = xin; p.xout = xout; p.hist_b = h->d_state[sb]; p.hist_d = h->d_state[sd];
        for (int i = 0; i < 4; ++i) {
            p.w[i] = h->d_conv[first_layer + i]; p.scale[i] = h->d_scale[first_layer + i];
            p.shift[i] = h->d_shift[first_layer + i]; p.dbg_off[i] = dbg_off[first_layer + i];
            p.clampv[i] = -0.4f * ldexpf(1.f, h->hx_e[first_layer + i]); p.dbg_mul[i] = ldexpf(1.f, -h->hx_e[first_layer + i]);
        }
        p.dbg_mul[4] = ldexpf(1.f, -h->hx_e[19]);
        p.xmul = ldexpf(1.f, h->hx_xexp[first_layer / 4 + 1]);      // first_layer 3, 7, 11, 15 -> hand-over 1..4
        p.emb_mul = ldexpf(1.f, -h->hx_e[19]);
        p.n_groups = (n_active + spt - 1) / spt; p.S = h->Spad;
        p.dbg = DBG ? h->d_dbg : nullptr; p.dbg_stride = DBG_FLOATS;
        p.range_flag = HX ? h->d_range : nullptr;
        p.stream_on = h->on_now;
        if (HX && h->blk_s1 > 0) { p.g_base = h->blk_s0 / spt; p.n_groups = (h->blk_s1 - h->blk_s0 + spt - 1) / spt; }
        if (HX && h->lists_now) {                                   // spt 1, 2, 4, 8 -> list 0, 1, 2, 3
            const int k = spt == 1 ? 0 : spt == 2 ? 1 : spt == 4 ? 2 : 3;
            p.glist = h->gl_now[k]; p.n_groups = h->gn_now[k];
        }
    };
    {
        RStageParams p{}; fill(p, h->d_xA, h->d_xB, 3, 2, 3, RB::SPT);
        Timed t(h, 2);
        // a launch whose workgroups are (nearly) alone on their CUs runs the three-slot weight ring (owwhip_hx.h WRing): same results
        const int nwg = (p.n_groups + OWH_WG_B - 1) / OWH_WG_B;
#ifdef OWH_DEEP_RING
        if (HX && !DBG && nwg <= h->deep_wgs) { hipLaunchKernelGGL((owh::hstage_kernel<owh::HB, false, false, OWH_WG_B, 4>), dim3(nwg), dim3(64 * OWH_WG_B), 0, st, p); } else
#endif
        if (HX && !DBG && (nwg <= h->small_wgs || h->ring3_always[0])) hipLaunchKernelGGL((owh::hstage_kernel<owh::HB, false, false, OWH_WG_B, 3>), dim3(nwg), dim3(64 * OWH_WG_B), 0, st, p);
        else if (HX) hipLaunchKernelGGL((owh::hstage_kernel<owh::HB, false, DBG, OWH_WG_B>), dim3(nwg), dim3(64 * OWH_WG_B), 0, st, p);
        else hipLaunchKernelGGL((rstage_kernel<RB, false, DBG>), dim3((p.n_groups + 3) / 4), dim3(256), 0, st, p);
    }
    {
        RStageParams p{}; fill(p, h->d_xB, h->d_xC, 7, 4, 5, RC::SPT);
        Timed t(h, 3);
        // a launch whose workgroups are (nearly) alone on their CUs runs the three-slot weight ring (owwhip_hx.h WRing): same results
        const int nwg = (p.n_groups + OWH_WG_C - 1) / OWH_WG_C;
#ifdef OWH_DEEP_RING
        if (HX && !DBG && nwg <= h->deep_wgs) { hipLaunchKernelGGL((owh::hstage_kernel<owh::HC, false, false, OWH_WG_C, 5>), dim3(nwg), dim3(64 * OWH_WG_C), 0, st, p); } else
#endif
        if (HX && !DBG && (nwg <= h->small_wgs || h->ring3_always[1])) hipLaunchKernelGGL((owh::hstage_kernel<owh::HC, false, false, OWH_WG_C, 3>), dim3(nwg), dim3(64 * OWH_WG_C), 0, st, p);
        else if (HX) hipLaunchKernelGGL((owh::hstage_kernel<owh::HC, false, DBG, OWH_WG_C>), dim3(nwg), dim3(64 * OWH_WG_C), 0, st, p);
        else hipLaunchKernelGGL((rstage_kernel<RC, false, DBG>), dim3((p.n_groups + 3) / 4), dim3(256), 0, st, p);
    }
    {
        RStageParams p{}; fill(p, h->d_xC, h->d_xD, 11, 6, 7, RD::SPT);
        Timed t(h, 4);
        // a launch whose workgroups are (nearly) alone on their CUs runs the three-slot weight ring (owwhip_hx.h WRing): same results
        const int nwg = (p.n_groups + OWH_WG_D - 1) / OWH_WG_D;
#ifdef OWH_DEEP_RING
        if (HX && !DBG && nwg <= h->deep_wgs) { hipLaunchKernelGGL((owh::hstage_kernel<owh::HD, false, false, OWH_WG_D, 5>), dim3(nwg), dim3(64 * OWH_WG_D), 0, st, p); } else
#endif
        if (HX && !DBG && (nwg <= h->small_wgs || h->ring3_always[2])) hipLaunchKernelGGL((owh::hstage_kernel<owh::HD, false, false, OWH_WG_D, 3>), dim3(nwg), dim3(64 * OWH_WG_D), 0, st, p);
        else if (HX) hipLaunchKernelGGL((owh::hstage_kernel<owh::HD, false, DBG, OWH_WG_D>), dim3(nwg), dim3(64 * OWH_WG_D), 0, st, p);
        else hipLaunchKernelGGL((rstage_kernel<RD, false, DBG>), dim3((p.n_groups + 3) / 4), dim3(256), 0, st, p);
    }
    {
        RStageParams p{}; fill(p, h->d_xD, nullptr, 15, 8, 9, RE::SPT);
        p.hist19 = h->d_state[10]; p.w19 = h->d_conv[19]; p.feat = h->d_feat; p.emb = h->d_emb; p.nfeat = h->d_nfeat; p.TR = h->TR;
        p.dbg_off[4] = dbg_off[19];
        Timed t(h, 5);
        // a launch whose workgroups are (nearly) alone on their CUs runs the three-slot weight ring (owwhip_hx.h WRing): same results
        const int nwg = (p.n_groups + OWH_WG_E - 1) / OWH_WG_E;
#ifdef OWH_DEEP_RING
        if (HX && !DBG && nwg <= h->deep_wgs) { hipLaunchKernelGGL((owh::hstage_kernel<owh::HE, true, false, OWH_WG_E, 5>), dim3(nwg), dim3(64 * OWH_WG_E), 0, st, p); } else
#endif
        if (HX && !DBG && (nwg <= h->small_wgs || h->ring3_always[3])) hipLaunchKernelGGL((owh::hstage_kernel<owh::HE, true, false, OWH_WG_E, 3>), dim3(nwg), dim3(64 * OWH_WG_E), 0, st, p);
        else if (HX) hipLaunchKernelGGL((owh::hstage_kernel<owh::HE, true, DBG, OWH_WG_E>), dim3(nwg), dim3(64 * OWH_WG_E), 0, st, p);
        else hipLaunchKernelGGL((rstage_kernel<RE, true, DBG>), dim3((p.n_groups + 3) / 4), dim3(256), 0, st, p);
    }
    HIPCHK(hipGetLastError());
    return 0;
}

int run_cnn(oww_ctx* h, int n_active, int mel_stride, int mel_off) {
    // grids cover whole workgroups of streams: round up to the largest per-workgroup stream count
    n_active = std::min(h->Spad, (n_active + 7) / 8 * 8);
    if (h->hx) return h->d_dbg ? run_cnn_rr<true, true>(h, n_active, mel_stride, mel_off) : run_cnn_rr<false, true>(h, n_active, mel_stride, mel_off);
    if (h->rr) return h->d_dbg ? run_cnn_rr<true, false>(h, n_active, mel_stride, mel_off) : run_cnn_rr<false, false>(h, n_active, mel_stride, mel_off);
    return h->mfma ? run_cnn_t<true>(h, n_active, mel_stride, mel_off) : run_cnn_t<false>(h, n_active, mel_stride, mel_off);
}

int heads_lds_bytes(int NH) { return (HD_SB * 100 + 2 * HD_SB * (NH + 4) + HD_SB * HD_MAXNETS) * 4; }

// heads over streams [0,n_active): ring mode (ext == nullptr) or external features
// heads_generic_kernel in the shape that fits the launch (owwhip_kernels.h): <4 streams per wave, 4 waves> below GH_BIG_STREAMS
// streams, <16, 2> from there on when no head is wider than 128 hidden units (its accumulator registers are sized for that:
// train.py's default width and the released multiclass models).  OWW_GENERIC_SPW=4|16 pins one -- tests run both on the same inputs.
// hs = LDS stride of a hidden vector.
void launch_generic_heads(oww_ctx* h, const HeadParams& p, int n_active, int nb, int ne, hipStream_t st) {
    const int hs = (std::max(h->generic_hmax, 1) + 3) & ~3;
    const bool big = h->generic_hmax <= 128 && (h->generic_spw == 16 || (h->generic_spw == 0 && n_active >= GH_BIG_STREAMS));
    if (big) hipLaunchKernelGGL((heads_generic_kernel<16, 2, 2>), dim3((n_active + 31) / 32), dim3(128), gh_lds_bytes(16, 2, hs), st, p, nb, ne, hs);
    else hipLaunchKernelGGL((heads_generic_kernel<4, 4>), dim3((n_active + 15) / 16), dim3(256), gh_lds_bytes(4, 4, hs), st, p, nb, ne, hs);
}

int run_heads(oww_ctx* h, int n_active, bool accumulate_max, const float* ext, int only_head, float* raw_out, int force_generic) {
    hipStream_t st = h->stream;
    Timed t(h, 6);
    HeadParams base{};
    base.feat = ext ? ext : h->d_feat; base.ext = ext ? 1 : 0; base.TR = h->TR; base.nfeat = h->d_nfeat;
    base.raw = raw_out; base.NL = h->NL; base.S = n_active; base.accumulate_max = accumulate_max ? 1 : 0;
    base.stream_on = ext ? nullptr : h->on_now;
    const bool fast_ok = h->mfma && !force_generic;
    if (fast_ok) {
        for (auto& g : h->groups) {
            if (only_head >= 0) {
                bool has = false;
                for (int ni : g.nets) has |= h->nets[ni].head == only_head;
                if (!has) continue;
            }
            if (h->hx) {
                owh::HeadHxParams q{};
                q.feat = base.feat; q.ext = base.ext; q.TR = base.TR; q.T = g.T; q.nfeat = base.nfeat; q.w1hx = g.d_w1hx;
                q.raw = raw_out; q.NL = h->NL; q.S = n_active; q.accumulate_max = base.accumulate_max;
                q.range_flag = h->d_range; q.stream_on = h->on_now;
                if (h->lists_now && !ext) { q.ids = h->gl_now[0]; q.n_ids = h->gn_now[0]; }
                if (h->blk_s1 > 0 && !ext) { q.s_base = h->blk_s0; q.S = h->blk_s1; }
                const int n_pos = q.ids ? q.n_ids : q.S - q.s_base;
                if (h->post_in_heads_now) {
                    owh::HeadHxPost& pp = q.post;
                    pp.enabled = 1; pp.scores = h->d_scores; pp.ring = h->d_ring; pp.npred = h->d_npred; pp.nfeat = h->d_nfeat;
                    pp.patience = h->d_patience; pp.threshold = h->d_threshold; pp.debounce_frames = h->debounce_frames;
                    pp.vad_ring = h->d_vadring; pp.n_vad = h->d_nvad; pp.vad_threshold = h->vad_threshold;
                }
                for (int i = 0; i < g.n_nets; ++i) {
                    const NetHost& n = h->nets[g.nets[i]];
                    const NetDesc d = h->host_descs[g.nets[i]];
                    owh::HeadHxNet& o = q.net[i];
                    const float* pd = g.d_pad[i];                      // the per-unit arrays padded to 64 (wide form: 128) units
                    const int HP = 16 * g.ht;
                    o.w2hx = g.d_w2hx[i]; o.b1 = pd; o.ln1g = pd + HP; o.ln1b = pd + 2 * HP; o.b2 = pd + 3 * HP; o.ln2g = pd + 4 * HP; o.ln2b = pd + 5 * HP;
                    o.w3 = pd + 6 * HP; o.b3 = d.b3; o.has_ln = n.has_ln; o.role = n.role; o.head = n.head; o.out_col = n.out_col;
                    o.hidden = n.hidden; o.inv_hidden = 1.0f / (float)n.hidden;
                    o.u1 = std::ldexp(1.0f, -(h->hx_efeat + n.hx_e1)); o.u2 = std::ldexp(1.0f, -n.hx_e2);
                    o.n_out = n.n_out; o.final_act = n.final_act;
                    if (g.ht == 8) { o.w3 = nullptr; o.b3 = pd + 6 * HP; o.w3hx = g.d_w3hx[i]; o.u3 = std::ldexp(1.0f, -n.hx_e3); }
                }
                q.fscale = std::ldexp(1.0f, h->hx_efeat);
                const dim3 grid((n_pos + 32 * owh::HX_WG - 1) / (32 * owh::HX_WG)), block(64 * owh::HX_WG);
                // a launch that leaves workgroups alone on their CUs runs the deep weight ring (owwhip_hx.h: HX_NBUF_DEEP); same results
                const bool deep = (int)grid.x <= h->small_wgs_heads;
                const int nn = std::min(g.n_nets, 4);
                const int lds = 0;                                  // (the ring slots are static LDS objects: owwhip_hx.h hslot)
                if (g.ht == 8) {                                    // wide nets (<= 128 hidden units, <= 8 outputs): one or two per launch
                    if (nn == 1 && !deep) hipLaunchKernelGGL((owh::heads_hx_kernel<1, owh::HX_NBUF, owh::HX_WG, 8>), grid, block, lds, st, q);
                    else if (nn == 1) hipLaunchKernelGGL((owh::heads_hx_kernel<1, owh::HeadsDeep<2>::NBUF, owh::HX_WG, 8>), grid, block, lds, st, q);
                    else if (!deep) hipLaunchKernelGGL((owh::heads_hx_kernel<2, owh::HX_NBUF, owh::HX_WG, 8>), grid, block, lds, st, q);
                    else hipLaunchKernelGGL((owh::heads_hx_kernel<2, owh::HeadsDeep<4>::NBUF, owh::HX_WG, 8>), grid, block, lds, st, q);
                    continue;
                }
                switch (nn * 2 + (deep ? 1 : 0)) {
                    case 2: hipLaunchKernelGGL(owh::heads_hx_kernel<1>, grid, block, lds, st, q); break;
                    case 3: hipLaunchKernelGGL((owh::heads_hx_kernel<1, owh::HeadsDeep<1>::NBUF>), grid, block, lds, st, q); break;
                    case 4: hipLaunchKernelGGL(owh::heads_hx_kernel<2>, grid, block, lds, st, q); break;
                    case 5: hipLaunchKernelGGL((owh::heads_hx_kernel<2, owh::HeadsDeep<2>::NBUF>), grid, block, lds, st, q); break;
                    case 6: hipLaunchKernelGGL(owh::heads_hx_kernel<3>, grid, block, lds, st, q); break;
                    case 7: hipLaunchKernelGGL((owh::heads_hx_kernel<3, owh::HeadsDeep<3>::NBUF>), grid, block, lds, st, q); break;
                    case 8: hipLaunchKernelGGL(owh::heads_hx_kernel<4>, grid, block, lds, st, q); break;
                    default: hipLaunchKernelGGL((owh::heads_hx_kernel<4, owh::HeadsDeep<4>::NBUF>), grid, block, lds, st, q); break;
                }
                continue;
            }
            HeadParams p = base;
            p.nets = g.d_nets; p.n_nets = g.n_nets; p.T = g.T; p.NH = g.NH; p.w1pk = g.d_w1pk; p.b1cat = g.d_b1cat;
            hipLaunchKernelGGL(heads64_kernel, dim3((n_active + HD_SB - 1) / HD_SB), dim3(HD_NT), heads_lds_bytes(g.NH), st, p);
        }
    }
    // generic kernel: nets that have no fast group, or everything when the fast path is off
    if (!fast_ok) {
        HeadParams p = base;
        p.nets = h->d_allnets; p.n_nets = (int)h->nets.size();
        int nb = 0, ne = (int)h->nets.size();
        if (only_head >= 0) { nb = h->head_nets[only_head].first; ne = h->head_nets[only_head].second; }
        launch_generic_heads(h, p, n_active, nb, ne, st);
    } else if (!h->generic_nets.empty()) {
        for (size_t hi = 0; hi < h->heads.size(); ++hi) {
            if (only_head >= 0 && (int)hi != only_head) continue;
            const int nb = h->head_nets[hi].first, ne = h->head_nets[hi].second;
            if (std::find(h->generic_nets.begin(), h->generic_nets.end(), nb) == h->generic_nets.end()) continue;
            HeadParams p = base;
            p.nets = h->d_allnets; p.n_nets = (int)h->nets.size();
            launch_generic_heads(h, p, n_active, nb, ne, st);
        }
    }
    // recurrent heads (model_type "rnn"): the same kernel in every family
    for (int ni : h->rnn_nets) {
        if (only_head >= 0 && h->nets[ni].head != only_head) continue;
        HeadParams p = base;
        p.nets = h->d_allnets; p.n_nets = (int)h->nets.size();
        hipLaunchKernelGGL((heads_rnn_kernel<RNN_SPW>), dim3((n_active + RNN_SPW - 1) / RNN_SPW), dim3(64), rnn_lds_bytes(h->nets[ni].T), st, p, ni);
    }
    HIPCHK(hipGetLastError());
    return 0;
}

int launch_mel(oww_ctx* h, const int16_t* d_pcm, int n_streams, int n_samples, int n_frames, int streaming, float* out, float* smax,
               int pcm_stride = 0, int max_only = 0, const float* floor_max = nullptr) {
    MelParams p{};
    p.pcm = d_pcm; p.n_samples = n_samples; p.n_frames = n_frames; p.streaming = streaming;
    p.pcm_stride = pcm_stride; p.max_only = max_only; p.floor_max = floor_max;
    p.tail = h->d_tail; p.nfeat = h->d_nfeat; p.out = out; p.smax = smax;
    p.hann = h->d_hann; p.mel_start = h->d_mstart; p.mel_taps = h->d_taps; p.S = n_streams;
    p.stream_on = streaming ? h->on_now : nullptr;
#ifndef OWK_MEL_WGS
#define OWK_MEL_WGS 6      // mel workgroups per CU in the persistent grid (20 KB LDS, 68 VGPRs each; 7 and 8 measured slower: 0.85 / 0.77 vs 0.73 ms)
#endif
    const int grid = std::min(n_streams, 256 * OWK_MEL_WGS);
    Timed t(h, 0);
    hipLaunchKernelGGL(mel_kernel, dim3(grid), dim3(MEL_NT), 0, h->stream, p);
    HIPCHK(hipGetLastError());
    return 0;
}

int do_reset(oww_ctx* h, const int* d_ids, int n, const float* d_featinit) {
    ResetParams p{};
    p.ids = d_ids; p.n = n; p.n_arrays = N_STATE;
    for (int a = 0; a < N_STATE; ++a) {
        p.dst[a] = h->d_state[a]; p.tmpl[a] = h->d_tmpl[a]; p.len[a] = h->state_len[a];
        p.spg[a] = h->rr ? kStateSpgRr[a] : 1; p.fpos[a] = h->rr ? kStateFposRr[a] : 16;
    }
    p.interleaved = h->hx && owh::kInterleave;
    p.tail = h->d_tail; p.nfeat = h->d_nfeat; p.npred = h->d_npred;
    p.ring = h->d_ring; p.ring_len = h->NL * OWW_SCORE_RING;
    p.feat = h->d_feat; p.feat_len = h->TR * OWW_EMB_DIM; p.feat_init = d_featinit;
    hipLaunchKernelGGL(reset_kernel, dim3(n), dim3(256), 0, h->stream, p);
    HIPCHK(hipGetLastError());
    return 0;
}

template <class T>
int dalloc(hipStream_t st, T** p, size_t n, bool zero = true, int line = __builtin_LINE()) {
    HIPCHK(dev_alloc(p, std::max<size_t>(n, 1) * sizeof(T), line));
    // the zero fill runs ON THE HANDLE'S STREAM: ordered in front of every kernel the handle will launch on it (its other streams are
    // forked from it by events), and no host round trip per buffer -- a legacy-stream fill would need one, because nothing orders
    // the handle's non-blocking streams behind the legacy stream (round 4: ~90 synchronisations per handle creation)
    if (zero) HIPCHK(hipMemsetAsync(*p, 0, std::max<size_t>(n, 1) * sizeof(T), st));
    return 0;
}

// ---- RCCL (librccl.so: ncclSend / ncclRecv over xGMI), bound at run time: the library is only needed by callers that shard streams
//      over GPUs WITHOUT torch.distributed (oww_comm_init / oww_gather_scores); nothing else in libowwhip touches it
struct RcclId { char b[128]; };           // ncclUniqueId (passed BY VALUE to ncclCommInitRank)
struct Rccl {
    void* lib = nullptr;
    int (*GetUniqueId)(void*) = nullptr;
    int (*CommInitRank)(void**, int, RcclId, int) = nullptr;
    int (*CommDestroy)(void*) = nullptr;
    int (*CommCount)(void*, int*) = nullptr;
    int (*GroupStart)() = nullptr;
    int (*GroupEnd)() = nullptr;
    int (*Send)(const void*, size_t, int, int, void*, hipStream_t) = nullptr;
    int (*Recv)(void*, size_t, int, int, void*, hipStream_t) = nullptr;
    const char* (*GetErrorString)(int) = nullptr;
};
Rccl g_rccl;
int rccl_load() {
    if (g_rccl.lib) return 0;
    void* lib = nullptr;
    // a copy the process already holds (torch bundles one) wins: two RCCL instances in one process is asking for trouble
    for (const char* name : {"librccl.so.1", "librccl.so"}) if (!lib) lib = dlopen(name, RTLD_NOW | RTLD_NOLOAD);
    for (const char* name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"}) if (!lib) lib = dlopen(name, RTLD_NOW | RTLD_LOCAL);
    if (!lib) return fail(OWW_ESTATE, "librccl.so could not be loaded: %s", dlerror());
    Rccl r; r.lib = lib;
    bool ok = true;
    auto sym = [&](const char* n) { void* p = dlsym(lib, n); ok = ok && p != nullptr; return p; };
    r.GetUniqueId = reinterpret_cast<decltype(r.GetUniqueId)>(sym("ncclGetUniqueId"));
    r.CommInitRank = reinterpret_cast<decltype(r.CommInitRank)>(sym("ncclCommInitRank"));
    r.CommDestroy = reinterpret_cast<decltype(r.CommDestroy)>(sym("ncclCommDestroy"));
    r.CommCount = reinterpret_cast<decltype(r.CommCount)>(sym("ncclCommCount"));
    r.GroupStart = reinterpret_cast<decltype(r.GroupStart)>(sym("ncclGroupStart"));
    r.GroupEnd = reinterpret_cast<decltype(r.GroupEnd)>(sym("ncclGroupEnd"));
    r.Send = reinterpret_cast<decltype(r.Send)>(sym("ncclSend"));
    r.Recv = reinterpret_cast<decltype(r.Recv)>(sym("ncclRecv"));
    r.GetErrorString = reinterpret_cast<decltype(r.GetErrorString)>(sym("ncclGetErrorString"));
    if (!ok) return fail(OWW_ESTATE, "librccl.so lacks one of ncclGetUniqueId / ncclCommInitRank / ncclSend / ncclRecv / ncclGroupStart / ncclGroupEnd");
    g_rccl = r;
    return 0;
}
#define RCCLCHK(expr)                                                                                         \
    do {                                                                                                      \
        int e__ = (expr);                                                                                     \
        if (e__ != 0) return fail(OWW_EHIP, "%s failed: %s", #expr, g_rccl.GetErrorString ? g_rccl.GetErrorString(e__) : "?"); \
    } while (0)
void comm_release(oww_ctx* h);

void free_all(oww_ctx* h) {
    // Nothing of this handle may still be queued when its buffers go: every stream the handle ever launched on is drained first
    // (hipFree would wait for the whole device as well, but the page-locked words -- h_range, which the f16-split kernels write at
    // exit, h_lists, the ingest slots -- and the streams and events themselves are released by calls that promise no such wait).
    if (h->stream) (void)hipStreamSynchronize(h->stream);
    for (hipStream_t st : {h->up_stream, h->down_stream, h->blk_stream[0], h->blk_stream[1], h->blk_stream[2], h->blk_stream[3]})
        if (st) (void)hipStreamSynchronize(st);
    auto fr = [](auto*& p) { if (p) { (void)dev_free((void*)p); p = nullptr; } };
    fr(h->d_w); fr(h->d_allnets); fr(h->d_generic);
    for (auto& g : h->groups) fr(g.d_nets);
    for (int a = 0; a < N_STATE; ++a) { fr(h->d_state[a]); fr(h->d_tmpl[a]); }
    fr(h->d_xA); fr(h->d_xB); fr(h->d_xC); fr(h->d_xD); fr(h->d_mel); fr(h->d_feat); fr(h->d_emb); fr(h->d_raw);
    fr(h->d_scores); fr(h->d_ring); fr(h->d_featinit); fr(h->d_dbg); fr(h->d_nfeat); fr(h->d_npred); fr(h->d_tail); fr(h->d_vadring); fr(h->d_nvad); fr(h->d_vadin); fr(h->d_vadx); fr(h->d_vadhc); fr(h->d_vadlast); fr(h->d_verw); fr(h->d_verb); fr(h->d_verthr); fr(h->d_verT);
    fr(h->d_prof); fr(h->d_pcm); fr(h->d_ids); fr(h->d_patience); fr(h->d_threshold); fr(h->d_save); fr(h->d_long); fr(h->d_callmax);
    h->long_cap = 0;
    h->save_floats = 0;
    if (h->d_on) { (void)dev_free(h->d_on); h->d_on = nullptr; }
    for (int b = 0; b < 4; ++b) {
        if (h->blk_stream[b]) { (void)hipStreamSynchronize(h->blk_stream[b]); (void)hipStreamDestroy(h->blk_stream[b]); h->blk_stream[b] = nullptr; }
        if (h->blk_done[b]) { (void)hipEventDestroy(h->blk_done[b]); h->blk_done[b] = nullptr; }
    }
    if (h->blk_fork) { (void)hipEventDestroy(h->blk_fork); h->blk_fork = nullptr; }
    if (h->d_lists) { (void)dev_free(h->d_lists); h->d_lists = nullptr; }
    for (int i = 0; i < 2; ++i) {
        if (h->h_lists[i]) { (void)hipHostFree(h->h_lists[i]); h->h_lists[i] = nullptr; }
        if (h->lists_ev[i]) { (void)hipEventDestroy(h->lists_ev[i]); h->lists_ev[i] = nullptr; }
    }
    h->lists_cap = 0;
    if (h->d_rs) { (void)dev_free(h->d_rs); h->d_rs = nullptr; h->rs_bytes = 0; }
    if (h->h_range) { (void)hipHostFree(h->h_range); h->h_range = nullptr; h->d_range = nullptr; }
    for (auto& sl : h->slot) {
        fr(sl.d_pcm); fr(sl.d_scores);
        if (sl.h_scores) { (void)hipHostFree(sl.h_scores); sl.h_scores = nullptr; }
        if (sl.h_on) { (void)hipHostFree(sl.h_on); sl.h_on = nullptr; }
        for (hipEvent_t* e : {&sl.up, &sl.done, &sl.down}) if (*e) { (void)hipEventDestroy(*e); *e = nullptr; }
        sl.busy = false;
    }
    if (h->up_stream) { (void)hipStreamDestroy(h->up_stream); h->up_stream = nullptr; }
    if (h->down_stream) { (void)hipStreamDestroy(h->down_stream); h->down_stream = nullptr; }
    for (auto& e : h->ev) { (void)hipEventDestroy(e.a); (void)hipEventDestroy(e.b); }
    h->ev.clear();
    if (h->graph_exec) { (void)hipGraphExecDestroy(h->graph_exec); h->graph_exec = nullptr; }
    if (h->graph) { (void)hipGraphDestroy(h->graph); h->graph = nullptr; }
    comm_release(h);
}

// one chunk of the streaming step on device-resident mel rows
// (k, c: the mel rows of this chunk sit at row 8c of 8k per stream; first / last: of the CALL, which may span several mel slices)
int step_chunk(oww_ctx* h, int k, int c, bool first, bool last, bool single);
int step_chunk(oww_ctx* h, int k, int c) { return step_chunk(h, k, c, c == 0, c == k - 1, k == 1); }
int step_chunk(oww_ctx* h, int k, int c, bool first, bool last, bool single) {
    if (int rc = run_cnn(h, h->Spad, 8 * k * 32, c * 8 * 32)) return rc;
    h->post_in_heads_now = h->post_in_heads && single && h->n_verifiers == 0;
    const int rc = run_heads(h, h->Spad, !first, nullptr, -1, h->d_raw, 0);
    const bool done_in_heads = h->post_in_heads_now;
    h->post_in_heads_now = false;
    if (rc) return rc;
    if (h->n_verifiers > 0 && last) {                // after the maximum over the call's chunks, on the newest feature rows
        VerifierParams v{};
        v.raw = h->d_raw; v.feat = h->d_feat; v.nfeat = h->d_nfeat; v.w = h->d_verw; v.bias = h->d_verb; v.thr = h->d_verthr; v.T = h->d_verT;
        v.wstride = h->ver_stride; v.NL = h->NL; v.TR = h->TR; v.S = h->S; v.stream_on = h->on_now;
        hipLaunchKernelGGL(verifier_kernel, dim3((h->S + 3) / 4), dim3(256), 0, h->stream, v);
    }
    if (!done_in_heads)
        hipLaunchKernelGGL(advance_kernel, dim3((h->Spad + 255) / 256), dim3(256), 0, h->stream, h->d_nfeat, h->Spad, h->on_now);
    return 0;
}

// voice-activity stand-in network for this step's 1280 new samples of every stream -> one score per stream into the VAD ring
int launch_vad(oww_ctx* h, const int16_t* d_pcm, int n_samples) {
    const int G = (h->S + 15) / 16;
    {
        owv::VadFrontParams p{};
        p.pcm = d_pcm; p.n_samples = n_samples; p.S = h->S; p.hann = h->d_vad_hann; p.mag_gain = h->vad_gain;
        p.w = h->d_vad_encw; p.bias = h->d_vad_encb; p.xout = h->d_vadx; p.range_flag = h->d_range; p.stream_on = h->on_now;
        const int grid = std::min((h->S + owv::V_WG - 1) / owv::V_WG, 256);          // persistent: one 8-wave workgroup per CU
        Timed t(h, 8);
        hipLaunchKernelGGL(owv::vad_front_kernel, dim3(grid), dim3(64 * owv::V_WG), owv::V_LDS_BYTES, h->stream, p);
    }
    {
        owv::VadLstmParams p{};
        p.xin = h->d_vadx; p.hc = h->d_vadhc; p.w = h->d_vad_lstmw; p.bias = h->d_vad_lstmb; p.wd = h->d_vad_wd; p.bd = h->vad_bd;
        p.ring = h->d_vadring; p.n_vad = h->d_nvad; p.last = h->d_vadlast; p.S = h->S; p.n_groups = G; p.stream_on = h->on_now;
        if (h->lists_now) { p.glist = h->gl_now[4]; p.n_groups = h->gn_now[4]; }
        Timed t(h, 9);
        hipLaunchKernelGGL(owv::vad_lstm_kernel, dim3((p.n_groups + owv::L_WG - 1) / owv::L_WG), dim3(64 * owv::L_WG), 0, h->stream, p);
    }
    HIPCHK(hipGetLastError());
    return 0;
}

// Lists for a masked step with few participants: [stream ids | C groups | D groups | E groups | VAD groups] into one pinned staging
// buffer, copied to the device on the compute stream (ordered before the step's kernels).  Returns the number of participants, -1
// when the dense launches should be used (more than 7/8 of the streams take part: the lists then save nothing; below that they do even
// for a mask that touches most groups, and a serving edge that places connections by cohort -- serve.py::SlotAllocator -- makes
// participation per group all-or-nothing, so the launches shrink with the mask), or an OWW_E* code - 100 on failure.
int build_active_lists(oww_ctx* h, const uint8_t* on) {
    const int S = h->S;
    int n_act = 0;
    for (int s = 0; s < S; ++s) n_act += on[s] != 0;          // (vectorised by the compiler)
    if (n_act == 0) return 0;
    if (!h->hx || !h->fuse || !h->generic_nets.empty() || !h->rnn_nets.empty()) return -1;       // the group lists are read by the default family's launches only
    if ((long long)n_act * 8 > (long long)S * 7) return -1;
    const size_t need = (size_t)S + S / 2 + S / 4 + S / 8 + S / 16 + 64;       // (regions of the five lists, see below)
    if (need > h->lists_cap) {
        if (h->d_lists) (void)dev_free(h->d_lists);
        h->d_lists = nullptr;
        for (int i = 0; i < 2; ++i) {
            if (h->lists_ev[i]) (void)hipEventSynchronize(h->lists_ev[i]);
            if (h->h_lists[i]) { (void)hipHostFree(h->h_lists[i]); h->h_lists[i] = nullptr; }
        }
        h->lists_cap = 0;
        if (dev_alloc(&h->d_lists, need * sizeof(int)) != hipSuccess) return fail(OWW_ENOMEM, "oww_step_masked: out of device memory") - 100;
        for (int i = 0; i < 2; ++i) {
            if (hipHostMalloc((void**)&h->h_lists[i], need * sizeof(int), hipHostMallocDefault) != hipSuccess) return fail(OWW_ENOMEM, "oww_step_masked: out of page-locked memory") - 100;
            if (!h->lists_ev[i] && hipEventCreateWithFlags(&h->lists_ev[i], hipEventDisableTiming) != hipSuccess) return fail(OWW_EHIP, "hipEventCreate failed") - 100;
        }
        h->lists_cap = need;
    }
    const unsigned turn = h->lists_turn++ & 1u;
    (void)hipEventSynchronize(h->lists_ev[turn]);              // the copy that last read this staging buffer has run
    int* out = h->h_lists[turn];
    // one pass over the mask, eight streams (one 64-bit word = one stage-E group) at a time; the five lists grow side by side in
    // fixed regions of the staging buffer and are packed afterwards
    const size_t base[5] = {0, (size_t)S, (size_t)S + S / 2 + 8, (size_t)S + S / 2 + S / 4 + 16, (size_t)S + S / 2 + S / 4 + S / 8 + 24};
    int n[5] = {0, 0, 0, 0, 0};
    const int W8 = S / 8;
    int last_v = -1;
    auto visit = [&](int s0, const uint8_t* b, int cnt) {        // streams s0 .. s0+cnt-1 (cnt <= 8), at least one of them on
        const int g8 = s0 / 8;
        out[base[3] + n[3]++] = g8;
        if (g8 / 2 != last_v) { last_v = g8 / 2; out[base[4] + n[4]++] = g8 / 2; }
        for (int q = 0; q < cnt; q += 4) {
            bool any4 = false;
            for (int c = q; c < std::min(cnt, q + 4); c += 2) {
                bool any2 = false;
                for (int i = c; i < std::min(cnt, c + 2); ++i) if (b[i]) { out[base[0] + n[0]++] = s0 + i; any2 = true; }
                if (any2) { out[base[1] + n[1]++] = (s0 + c) / 2; any4 = true; }
            }
            if (any4) out[base[2] + n[2]++] = (s0 + q) / 4;
        }
    };
    for (int w = 0; w < W8; ++w) {
        uint64_t v;
        memcpy(&v, on + (size_t)w * 8, 8);
        if (v) visit(w * 8, on + (size_t)w * 8, 8);
    }
    if (S % 8) {
        bool any = false;
        for (int s = W8 * 8; s < S; ++s) any = any || on[s] != 0;
        if (any) visit(W8 * 8, on + (size_t)W8 * 8, S - W8 * 8);
    }
    size_t off = 0;
    for (int k = 0; k < 5; ++k) {
        if (base[k] != off) memmove(out + off, out + base[k], (size_t)n[k] * sizeof(int));
        h->gl_now[k] = h->d_lists + off; h->gn_now[k] = n[k];
        off += (size_t)n[k];
    }
    if (off) {
        if (copy_async(h->d_lists, out, off * sizeof(int), hipMemcpyHostToDevice, h->stream) != hipSuccess) return fail(OWW_EHIP, "oww_step_masked: list upload failed") - 100;
        (void)hipEventRecord(h->lists_ev[turn], h->stream);
    }
    return n_act;
}

int launch_postproc(oww_ctx* h);
int launch_step(oww_ctx* h, const int16_t* d_pcm, int k) {
    if (h->vad) {
        if (k != 1) return fail(OWW_EINVAL, "with the on-device VAD network a step carries exactly one 1280-sample chunk per stream (got %d)", k);
        if (int rc = launch_vad(h, d_pcm, OWW_CHUNK * k)) return rc;
    }
    if (h->fuse && k == 1 && (reinterpret_cast<uintptr_t>(d_pcm) & 15) == 0) {      // (the fused front end uses 16-byte sample loads)
        h->fuse_pcm = d_pcm;
        int rc = 0;
        const bool blocks = h->n_blocks > 1 && h->post_in_heads && h->n_verifiers == 0 && !h->lists_now && !h->timing && !h->want_graph && !h->d_dbg;
        if (blocks) {
            // fork: the block streams wait for everything already queued on the handle's stream (PCM upload, VAD launches, masks)
            hipStream_t main_stream = h->stream;
            if (hipEventRecord(h->blk_fork, main_stream) != hipSuccess) rc = fail(OWW_EHIP, "block fork failed");
            const int per = (h->Spad / h->n_blocks + 127) / 128 * 128;
            for (int b = 0; b < h->n_blocks && !rc; ++b) {
                h->blk_s0 = b * per; h->blk_s1 = b == h->n_blocks - 1 ? h->Spad : std::min(h->Spad, (b + 1) * per);
                if (h->blk_s0 >= h->blk_s1 || h->blk_s0 >= h->S) break;
                if (hipStreamWaitEvent(h->blk_stream[b], h->blk_fork, 0) != hipSuccess) { rc = fail(OWW_EHIP, "block fork failed"); break; }
                h->stream = h->blk_stream[b];
                rc = step_chunk(h, 1, 0);
                h->stream = main_stream;
                if (!rc && (hipEventRecord(h->blk_done[b], h->blk_stream[b]) != hipSuccess || hipStreamWaitEvent(main_stream, h->blk_done[b], 0) != hipSuccess))
                    rc = fail(OWW_EHIP, "block join failed");
            }
            h->stream = main_stream; h->blk_s0 = h->blk_s1 = 0;
        } else rc = step_chunk(h, 1, 0);
        h->fuse_pcm = nullptr;
        if (rc) return rc;
    } else {
        if (int rc = launch_mel(h, d_pcm, h->S, OWW_CHUNK * k, 8 * k, 1, h->d_mel, nullptr)) return rc;
        for (int c = 0; c < k; ++c)
            if (int rc = step_chunk(h, k, c)) return rc;
    }
    if (h->post_in_heads && k == 1 && h->n_verifiers == 0) { HIPCHK(hipGetLastError()); return 0; }      // post-processing already ran inside the heads launch
    return launch_postproc(h);
}

// A call of more chunks than the handle's mel buffer holds (n_chunks > max_chunks; the reference takes any length: model.py:287-298,
// utils.py:387-401).  The reference runs its melspectrogram graph ONCE over the call, so the clamp floor "maximum - 80 dB" is the
// call's; evaluating the call in slices of max_chunks chunks with each slice's own maximum would move the floor (a quiet start of a
// call whose loud part comes later).  Two passes: the mel kernel over the whole call for its per-stream maximum only, then the
// slices -- mel rows with that shared floor, one embedding and one heads evaluation per chunk, raw scores max-combined over ALL
// chunks of the call -- and one post-processing pass.  d_pcm: [S][1280 K] on the device.
int launch_step_long(oww_ctx* h, const int16_t* d_pcm, int K) {
    if (h->vad) return fail(OWW_EINVAL, "with the on-device VAD network a step carries exactly one 1280-sample chunk per stream (got %d)", K);
    if (!h->d_callmax) if (int rc = dalloc(h->stream, &h->d_callmax, (size_t)h->Spad)) return rc;
    if (int rc = launch_mel(h, d_pcm, h->S, OWW_CHUNK * K, 8 * K, 1, nullptr, h->d_callmax, 0, 1, nullptr)) return rc;
    for (int o = 0; o < K; o += h->kmax) {
        const int ks = std::min(h->kmax, K - o);
        if (int rc = launch_mel(h, d_pcm + (size_t)o * OWW_CHUNK, h->S, OWW_CHUNK * ks, 8 * ks, 1, h->d_mel, nullptr, OWW_CHUNK * K, 0, h->d_callmax)) return rc;
        for (int c = 0; c < ks; ++c)
            if (int rc = step_chunk(h, ks, c, o + c == 0, o + c == K - 1, false)) return rc;
        h->k_last = ks;
    }
    return launch_postproc(h);
}

int launch_postproc(oww_ctx* h) {
    PostParams pp{};
    pp.raw = h->d_raw; pp.scores = h->d_scores; pp.ring = h->d_ring; pp.npred = h->d_npred;
    pp.patience = h->d_patience; pp.threshold = h->d_threshold; pp.debounce_frames = h->debounce_frames;
    pp.NL = h->NL; pp.S = h->Spad;
    pp.vad_ring = h->d_vadring; pp.n_vad = h->d_nvad; pp.vad_threshold = h->vad_threshold; pp.stream_on = h->on_now;
    {
        Timed t(h, 7);
        hipLaunchKernelGGL(postproc_kernel, dim3((h->Spad + 127) / 128), dim3(128), 0, h->stream, pp);
    }
    HIPCHK(hipGetLastError());
    return 0;
}

// sticky out-of-range flag of the f16-split kernels (owwhip_hx.h nan_guard): read straight from the mapped host word
int range_check(oww_ctx* h, const char* where) {
    if (h->h_range && *(volatile int*)h->h_range)
        return fail(OWW_ERANGE, "%s: an activation left the f16 range of the fp16-split kernels (use_mfma = 3) -- scores since then are not "
                    "trustworthy; create the handle with use_mfma = 1 (exact fp32) for these weights, or clear with oww_range_status(h, 1)", where);
    return 0;
}

// oww_embed / oww_embed_clips borrow the streaming machinery of the first streams: their state (conv histories, feature ring
// rows, frame counters) is parked in a scratch buffer for the duration of the call and put back afterwards
int park_state(oww_ctx* h, int n_streams, bool save) {
    const size_t n8 = std::min<size_t>(h->Spad, ((size_t)n_streams + 7) / 8 * 8);
    size_t need = h->Spad + n8 * (size_t)h->TR * 96;
    for (int a = 0; a < N_STATE; ++a) need += n8 * (size_t)h->state_len[a];
    if (save && need > h->save_floats) {
        if (h->d_save) (void)dev_free(h->d_save);
        h->d_save = nullptr; h->save_floats = 0;
        if (dev_alloc(&h->d_save, need * sizeof(float)) != hipSuccess) return fail(OWW_ENOMEM, "out of device memory for %zu parked state bytes", need * sizeof(float));
        h->save_floats = need;
    }
    if (!h->d_save || need > h->save_floats) return fail(OWW_ESTATE, "park_state: nothing parked");
    float* q = h->d_save;
    auto cp = [&](void* live, size_t nfl) -> int {
        HIPCHK(copy_async(save ? (void*)q : live, save ? live : (void*)q, nfl * sizeof(float), hipMemcpyDeviceToDevice, h->stream));
        q += nfl;
        return 0;
    };
    for (int a = 0; a < N_STATE; ++a) if (int rc = cp(h->d_state[a], n8 * (size_t)h->state_len[a])) return rc;
    if (int rc = cp(h->d_feat, n8 * (size_t)h->TR * 96)) return rc;
    if (int rc = cp(h->d_nfeat, h->Spad)) return rc;         // the frame counters of EVERY stream advance with the borrowed steps
    return 0;
}

// OWW_COMMIT_TIMING=1: wall-clock of the phases of oww_commit on stderr (development aid; handle creation should stay in the tens of
// milliseconds -- the reference constructs Model objects freely, utils.py:502-536)
struct CommitClock {
    bool on; double t0; const char* tag;
    static double now() { timespec ts; clock_gettime(CLOCK_MONOTONIC, &ts); return ts.tv_sec * 1e3 + ts.tv_nsec * 1e-6; }
    explicit CommitClock(const char* tag_) : on(getenv("OWW_COMMIT_TIMING") != nullptr), t0(now()), tag(tag_) {}
    void lap(const char* what) { if (on) { const double t = now(); fprintf(stderr, "[owwhip commit %s] %-28s %8.2f ms\n", tag, what, t - t0); t0 = t; } }
};

// ---- f16-split family: commit-time calibration and self-test against the exact-fp32 kernels -------------------------------------
// The reference's graphs are fp32 (onnxruntime CPU kernels, utils.py:84-93): they have no range to leave.  The f16-split kernels
// carry every activation as an f16 (hi, lo) pair, which is exact to 22 bits only inside the f16 exponent range, so oww_commit
//  (1) runs 32 probe streams (silence ... full-scale noise and square waves) plus the all-ones mel history through a scratch handle
//      of the exact-fp32 family with per-layer dumps and records every layer's largest |activation|,
//  (2) gives every layer the power-of-two scale K = 2^e that puts that maximum at 512..1024 -- a factor 64 below the f16 overflow
//      and 2^13 above the point where the low half goes subnormal -- and folds K, the BatchNorm scale and shift into the packed
//      weights / accumulator start values (fold_cnn; owwhip_hx.h act1),
//  (3) replays the probes through the handle's own f16-split kernels and compares the embeddings (and, with heads loaded, the raw
//      head outputs) with the fp32 run: weights for which the two differ by more than the north-star tolerance are refused with
//      OWW_ERANGE at commit instead of scoring differently later.
// Probes run in batches of CAL_NP streams x CAL_T frames (every handle has at least 32 padded streams): batch 0 is the built-in
// synthetic set, further batches carry the caller's calibration audio (oww_set_calibration: speech), cut into CAL_T-frame segments.
constexpr int CAL_NP = 32, CAL_T = 16, CAL_MAX_BATCHES = 8;
struct HxCalib {
    int nb = 1;                      // batches
    std::vector<int16_t> pcm;        // [nb][CAL_T][CAL_NP][1280]
    int16_t* d_pcm = nullptr;        // the same on the device: uploaded once, read by the calibration run and by the self-test replay
    std::vector<float> ref_emb;      // [nb][CAL_T][CAL_NP][96]   exact-fp32 embeddings of the probe run
    std::vector<float> ref_raw;      // [nb][CAL_T][CAL_NP][NL]   exact-fp32 raw head outputs
    int NL = 0;
    HxCalib() = default;
    HxCalib(const HxCalib&) = delete;
    HxCalib& operator=(const HxCalib&) = delete;
    ~HxCalib() { if (d_pcm) (void)dev_free(d_pcm); }
};

void make_probe_pcm(std::vector<int16_t>& pcm, const std::vector<int16_t>& user /*[n_seg][CAL_T * 1280]*/) {
    const size_t seg = (size_t)CAL_T * OWW_CHUNK, n_seg = user.size() / seg;
    const int nb = 1 + (int)((n_seg + CAL_NP - 1) / CAL_NP);
    pcm.assign((size_t)nb * CAL_T * CAL_NP * OWW_CHUNK, 0);
    for (size_t k = 0; k < n_seg; ++k) {
        const size_t b = 1 + k / CAL_NP, i = k % CAL_NP;
        for (int it = 0; it < CAL_T; ++it)
            memcpy(&pcm[((b * CAL_T + it) * CAL_NP + i) * OWW_CHUNK], &user[k * seg + (size_t)it * OWW_CHUNK], OWW_CHUNK * sizeof(int16_t));
    }
    // batch 0 (the synthetic set) is the same for every handle: computed once per process (655,360 Gaussian samples in double)
    static std::once_flag once;
    static std::vector<int16_t> synth;
    std::call_once(once, [] {
        synth.assign((size_t)CAL_T * CAL_NP * OWW_CHUNK, 0);
        uint64_t st = 0x9E3779B97F4A7C15ull;
        auto u01 = [&]() { st ^= st << 13; st ^= st >> 7; st ^= st << 17; return ((st >> 11) + 1) * (1.0 / 9007199254740993.0); };
        const double amps[5] = {30, 300, 3000, 12000, 32767};
        for (int i = 1; i < CAL_NP; ++i) {                 // stream 0: silence
            const double amp = amps[i % 5];
            for (int n = 0; n < CAL_T * OWW_CHUNK; ++n) {
                double v;
                if (i % 3 == 0) v = ((n / (8 << (i % 4))) % 2) ? amp : -amp;                        // square waves, 1 kHz .. 125 Hz
                else v = std::nearbyint(amp * std::sqrt(-2.0 * std::log(u01())) * std::cos(6.283185307179586 * u01()));
                v = std::min(32767.0, std::max(-32768.0, v));
                synth[((size_t)(n / OWW_CHUNK) * CAL_NP + i) * OWW_CHUNK + n % OWW_CHUNK] = (int16_t)v;
            }
        }
    });
    memcpy(pcm.data(), synth.data(), synth.size() * sizeof(int16_t));
}

// one probe step on handle t: mel of the chunk (separate kernel), CNN, frame counters, optionally the heads
int probe_step(oww_ctx* t, const int16_t* d_chunk, bool heads) {
    if (int rc = launch_mel(t, d_chunk, CAL_NP, OWW_CHUNK, 8, 1, t->d_mel, nullptr)) return rc;
    if (int rc = run_cnn(t, CAL_NP, 256, 0)) return rc;
    if (heads && t->NL > 0) {
        const bool save = t->post_in_heads_now;
        t->post_in_heads_now = false;
        const int rc = run_heads(t, CAL_NP, false, nullptr, -1, t->d_raw, 0);
        t->post_in_heads_now = save;
        if (rc) return rc;
    }
    hipLaunchKernelGGL(advance_kernel, dim3((t->Spad + 255) / 256), dim3(256), 0, t->stream, t->d_nfeat, t->Spad, (const uint8_t*)nullptr);
    return 0;
}

int calibrate_hx(oww_ctx* h, HxCalib& cal) {
    (void)hipGetLastError();                 // (a stale error of the caller's thread is not this function's to report)
    make_probe_pcm(cal.pcm, h->cal_user);
    cal.nb = (int)(cal.pcm.size() / ((size_t)CAL_T * CAL_NP * OWW_CHUNK));
    oww_config c2 = h->cfg;
    c2.n_streams = CAL_NP; c2.max_chunks = 1; c2.use_mfma = 1; c2.debug_layers = 1; c2.stream = nullptr;
    c2.feature_ring = h->TR;
    oww_ctx* t = nullptr;
    if (int rc = oww_create(&c2, &t)) return rc;
    int rc = 0;
    unsigned* d_max = nullptr; int* d_off = nullptr; float* d_ref = nullptr;
    CommitClock clk("calibrate");
    do {
        if ((rc = oww_load_mel(t, h->mel_blob.data(), h->mel_blob.size() * sizeof(float)))) break;
        if ((rc = oww_load_embedding(t, h->emb_blob.data(), h->emb_blob.size() * sizeof(float)))) break;
        for (const HeadHost& hh : h->heads) {
            std::vector<float> blob(8 + hh.blob.size());
            const int32_t hdr[8] = {hh.kind, hh.T, hh.hidden, hh.n_out, hh.has_ln, hh.n_blocks - 1, 0, 0};
            memcpy(blob.data(), hdr, sizeof hdr);
            memcpy(blob.data() + 8, hh.blob.data(), hh.blob.size() * sizeof(float));
            if ((rc = oww_add_head(t, blob.data(), blob.size() * sizeof(float))) < 0) break;
            rc = 0;
        }
        if (rc) break;
        if ((rc = oww_commit(t))) break;
        clk.lap("scratch fp32 handle");
        cal.NL = t->NL;
        int off[21]; off[0] = 0;
        for (int l = 0; l < 20; ++l) off[l + 1] = off[l] + kLayerOut[l][0] * kLayerOut[l][1] * kLayerOut[l][2];
        // probe audio up once, results down once: the run in between is stream-ordered, without a host round trip per probe step
        const size_t n_steps = (size_t)cal.nb * CAL_T, n_emb = n_steps * CAL_NP * 96, n_raw = n_steps * CAL_NP * std::max(t->NL, 1);
        if (dev_alloc(&d_max, 20 * sizeof(unsigned)) != hipSuccess || dev_alloc(&d_off, sizeof off) != hipSuccess ||
            dev_alloc(&cal.d_pcm, cal.pcm.size() * sizeof(int16_t)) != hipSuccess ||
            dev_alloc(&d_ref, (n_emb + n_raw) * sizeof(float)) != hipSuccess) { rc = fail(OWW_ENOMEM, "oww_commit: out of device memory (calibration)"); break; }
        if (hipMemsetAsync(d_max, 0, 20 * sizeof(unsigned), t->stream) != hipSuccess ||
            copy_async(d_off, off, sizeof off, hipMemcpyHostToDevice, t->stream) != hipSuccess ||
            copy_async(cal.d_pcm, cal.pcm.data(), cal.pcm.size() * sizeof(int16_t), hipMemcpyHostToDevice, t->stream) != hipSuccess) { rc = fail(OWW_EHIP, "oww_commit: calibration setup failed"); break; }
        auto absmax = [&]() { hipLaunchKernelGGL(layer_absmax_kernel, dim3(20, CAL_NP), dim3(256), 0, t->stream, t->d_dbg, (size_t)DBG_FLOATS, d_off, d_max); };
        // (a) the all-ones mel history every stream starts from (utils.py:165): the handle sits in that steady state after its commit
        hipLaunchKernelGGL(fill_kernel, dim3(CAL_NP), dim3(256), 0, t->stream, t->d_mel, (size_t)CAL_NP * 256, 1.0f);
        if ((rc = run_cnn(t, CAL_NP, 256, 0))) break;
        absmax();
        // (b) the probe audio, batch by batch from the reset state
        cal.ref_emb.assign((size_t)cal.nb * CAL_T * CAL_NP * 96, 0.f);
        cal.ref_raw.assign((size_t)cal.nb * CAL_T * CAL_NP * std::max(t->NL, 1), 0.f);
        for (int bt = 0; bt < cal.nb * CAL_T && !rc; ++bt) {
            const int it = bt % CAL_T;
            if (it == 0 && bt > 0 && (rc = do_reset(t, nullptr, CAL_NP, nullptr))) break;
            if (getenv("OWW_DEBUG_CALIB")) { const hipError_t e = hipStreamSynchronize(t->stream); fprintf(stderr, "calibrate: probe step %d of %d (%s)\n", bt, cal.nb * CAL_T, hipGetErrorString(e)); }
            if ((rc = probe_step(t, cal.d_pcm + (size_t)bt * CAL_NP * OWW_CHUNK, true))) break;
            absmax();
            if (hipMemcpyAsync(d_ref + (size_t)bt * CAL_NP * 96, t->d_emb, (size_t)CAL_NP * 96 * sizeof(float), hipMemcpyDeviceToDevice, t->stream) != hipSuccess ||
                (t->NL > 0 && hipMemcpyAsync(d_ref + n_emb + (size_t)bt * CAL_NP * t->NL, t->d_raw, (size_t)CAL_NP * t->NL * sizeof(float), hipMemcpyDeviceToDevice, t->stream) != hipSuccess)) { rc = fail(OWW_EHIP, "oww_commit: probe gather failed"); break; }
        }
        if (rc) break;
        unsigned mx[20];
        if (copy_async(cal.ref_emb.data(), d_ref, n_emb * sizeof(float), hipMemcpyDeviceToHost, t->stream) != hipSuccess ||
            (t->NL > 0 && copy_async(cal.ref_raw.data(), d_ref + n_emb, n_raw * sizeof(float), hipMemcpyDeviceToHost, t->stream) != hipSuccess) ||
            copy_async(mx, d_max, sizeof mx, hipMemcpyDeviceToHost, t->stream) != hipSuccess || hipStreamSynchronize(t->stream) != hipSuccess) { rc = fail(OWW_EHIP, "oww_commit: calibration run failed: %s", hipGetErrorString(hipGetLastError())); break; }
        clk.lap("probe run (exact fp32)");
        for (int l = 0; l < 20; ++l) {
            float m; memcpy(&m, &mx[l], 4);
            if (!std::isfinite(m)) { rc = fail(OWW_EINVAL, "oww_commit: layer %d of the embedding network produces non-finite activations in exact fp32 -- the weights are broken", l); break; }
            h->hx_absmax[l] = m;
        }
        if (rc) break;
        // Scale ladder.  acc = sum W' X needs no multiply after it only if W' = s w 2^(e_out - e_in), so the weights' magnitude is
        // pinned by the exponent step of the layer; their low halves stay precise (abs error 2^-25 against sums of magnitude
        // 2^(e_out) |y|) when that step is >= ~2.  Inside a stage the activation maxima therefore climb 2^3 (pooled input) ->
        // 2^5 -> 2^7 -> 2^9 -> 2^11 (a factor 32 below the f16 overflow for the loudest probe) and the pooled hand-over, which
        // is multiplied once per stored value anyway, brings the next stage's input back to 2^3.
        auto ex = [&](int l) { int e2 = 0; if (h->hx_absmax[l] > 0.f) std::frexp(h->hx_absmax[l], &e2); return e2; };   // max < 2^ex
        auto cl = [](int e2) { return std::min(100, std::max(-100, e2)); };
        const int first[5] = {0, 3, 7, 11, 15};
        for (int st = 0; st < 5; ++st) {
            const int n = st == 0 ? 3 : 4;
            for (int i = 0; i < n; ++i) {
                const int l = first[st] + i;
                h->hx_e[l] = cl((st == 0 ? 7 : 5) + 2 * i - ex(l));
                h->hx_ein[l] = i == 0 ? (st == 0 ? 0 : cl(3 - ex(l - 1))) : h->hx_e[l - 1];     // (max-pooling keeps the maximum)
            }
        }
        h->hx_ein[19] = cl(3 - ex(18));
        // conv19: no BatchNorm, no activation, and its accumulator is un-scaled in fp32 when the embedding is stored -- nothing pins its
        // output range, so the exponent step is chosen for the WEIGHTS: the largest |w| 2^step at 2^11..2^12 (a network whose last
        // layer is 1e-4 x weaker keeps 22 bits per weight: tests/test_weight_regimes.py, tiny_embedding)
        {
            const float* q = h->emb_blob.data();
            for (int l = 0; l < 19; ++l) q += (size_t)kLayers[l].kh * kLayers[l].kw * kLayers[l].cin * kLayers[l].cout + 2 * kLayers[l].cout;
            const int step = hx_weight_exp(q, (size_t)kLayers[19].kh * kLayers[19].kw * kLayers[19].cin * kLayers[19].cout);
            h->hx_e[19] = cl(h->hx_ein[19] + (step == -1000 ? 2 : step));
        }
        h->hx_efeat = cl(10 - ex(19));                          // the heads' GEMM takes the (true-unit) feature ring at this scale
        for (int st = 0; st < 5; ++st) {
            const int last = st == 0 ? 2 : first[st] + 3, nxt = last + 1;
            h->hx_xexp[st] = h->hx_ein[nxt] - h->hx_e[last];
        }
    } while (0);
    if (d_max) (void)dev_free(d_max);
    if (d_off) (void)dev_free(d_off);
    if (d_ref) (void)dev_free(d_ref);
    const std::string keep = g_err;
    // the scratch handle's whole life -- launches whose status nobody looked at, its frees -- must have left no HIP error behind
    const hipError_t e_run = hipStreamSynchronize(t->stream);
    (void)oww_destroy(t);
    const hipError_t e_last = hipGetLastError();
    if (rc) g_err = keep;
    (void)hipSetDevice(h->cfg.device);
    clk.lap("scratch handle destroyed");
    if (!rc && (e_run != hipSuccess || e_last != hipSuccess))
        rc = fail(OWW_EHIP, "oww_commit: the calibration handle left a HIP error behind (run: %s, last: %s)", hipGetErrorString(e_run), hipGetErrorString(e_last));
    return rc;
}

// replay of the probes on the handle's own (f16-split) kernels; leaves the first CAL_NP streams dirty -- the caller resets all state
int selftest_hx(oww_ctx* h, const HxCalib& cal) {
    const size_t n_steps = (size_t)cal.nb * CAL_T, n_emb = n_steps * CAL_NP * 96, n_raw = n_steps * CAL_NP * std::max(h->NL, 1);
    float* d_out = nullptr;
    if (!cal.d_pcm) return fail(OWW_ESTATE, "oww_commit: self-test without calibration probes");
    if (dev_alloc(&d_out, (n_emb + n_raw) * sizeof(float)) != hipSuccess) return fail(OWW_ENOMEM, "oww_commit: out of device memory (self-test)");
    std::vector<float> emb(n_emb), raw(n_raw);
    int rc = 0;
    float* saved_dbg = h->d_dbg; h->d_dbg = nullptr;
    for (int bt = 0; bt < cal.nb * CAL_T && !rc; ++bt) {
        if (bt % CAL_T == 0 && bt > 0 && (rc = do_reset(h, nullptr, CAL_NP, nullptr))) break;
        if (getenv("OWW_DEBUG_CALIB")) { const hipError_t e = hipStreamSynchronize(h->stream); fprintf(stderr, "self-test: probe step %d of %d (%s)\n", bt, cal.nb * CAL_T, hipGetErrorString(e)); }
        if ((rc = probe_step(h, cal.d_pcm + (size_t)bt * CAL_NP * OWW_CHUNK, true))) break;
        if (hipMemcpyAsync(d_out + (size_t)bt * CAL_NP * 96, h->d_emb, (size_t)CAL_NP * 96 * sizeof(float), hipMemcpyDeviceToDevice, h->stream) != hipSuccess ||
            (h->NL > 0 && hipMemcpyAsync(d_out + n_emb + (size_t)bt * CAL_NP * h->NL, h->d_raw, (size_t)CAL_NP * h->NL * sizeof(float), hipMemcpyDeviceToDevice, h->stream) != hipSuccess)) { rc = fail(OWW_EHIP, "oww_commit: probe gather failed"); break; }
    }
    h->d_dbg = saved_dbg;
    if (!rc && (copy_async(emb.data(), d_out, n_emb * sizeof(float), hipMemcpyDeviceToHost, h->stream) != hipSuccess ||
                (h->NL > 0 && copy_async(raw.data(), d_out + n_emb, n_raw * sizeof(float), hipMemcpyDeviceToHost, h->stream) != hipSuccess) ||
                hipStreamSynchronize(h->stream) != hipSuccess)) rc = fail(OWW_EHIP, "oww_commit: self-test run failed: %s", hipGetErrorString(hipGetLastError()));
    (void)dev_free(d_out);
    if (rc) return rc;
    float err = 0.f, ref = 0.f, serr = 0.f;
    bool finite = true;
    for (size_t i = 0; i < emb.size(); ++i) {
        finite = finite && std::isfinite(emb[i]);
        err = std::max(err, std::fabs(emb[i] - cal.ref_emb[i])); ref = std::max(ref, std::fabs(cal.ref_emb[i]));
    }
    if (h->NL > 0 && cal.NL == h->NL)
        for (size_t i = 0; i < raw.size(); ++i) { finite = finite && std::isfinite(raw[i]); serr = std::max(serr, std::fabs(raw[i] - cal.ref_raw[i])); }
    h->hx_selftest_err = err; h->hx_selftest_ref = ref; h->hx_selftest_score_err = serr;
    if (getenv("OWW_DEBUG_CALIB")) fprintf(stderr, "commit self-test: max |emb - fp32| %.3g on |emb| <= %.3g, max |raw score - fp32| %.3g\n", (double)err, (double)ref, (double)serr);
    const float tol = 1e-3f;                     // the north-star score tolerance
    if (!finite || err > tol * std::max(1.f, ref) || serr > tol || (h->h_range && *(volatile int*)h->h_range)) {
        if (h->h_range) *(volatile int*)h->h_range = 0;
        return fail(OWW_ERANGE, "oww_commit: with these weights the fp16-split kernels (use_mfma = 3) differ from the exact-fp32 kernels by %.3g on "
                    "embeddings of magnitude %.3g and by %.3g on raw scores over the probe set (tolerance %.0e): create the handle with "
                    "use_mfma = 1", (double)err, (double)ref, (double)serr, (double)tol);
    }
    return 0;
}

void comm_release(oww_ctx* h) {
    if (h->comm && g_rccl.CommDestroy) (void)g_rccl.CommDestroy(h->comm);
    h->comm = nullptr; h->comm_rank = 0; h->comm_world = 1;
}

}  // namespace

// =====================================================================================================
extern "C" {

int oww_abi_version(void) { return OWW_ABI_VERSION; }
#ifndef OWW_SRC_SHA16
#define OWW_SRC_SHA16 "unknown"            /* (set by openwakeword_amd/_build.py: hash of csrc/ + include/owwhip.h) */
#endif
const char* oww_build_info(void) { return "src=" OWW_SRC_SHA16 " arch=gfx950"; }
const char* oww_last_error(void) { return g_err.c_str(); }

int oww_create(const oww_config* cfg, oww_ctx** out) {
    OWW_GUARD_BEGIN
    if (!cfg || !out) return fail(OWW_EINVAL, "oww_create: null argument");
    if (cfg->n_streams < 1) return fail(OWW_EINVAL, "oww_create: n_streams must be >= 1");
    int ndev = 0;
    HIPCHK(hipGetDeviceCount(&ndev));
    if (cfg->device < 0 || cfg->device >= ndev) return fail(OWW_EINVAL, "oww_create: device %d of %d", cfg->device, ndev);
    HIPCHK(hipSetDevice(cfg->device));
    oww_ctx* h = new (std::nothrow) oww_ctx();
    if (!h) return fail(OWW_ENOMEM, "oww_create: out of host memory");
    h->cfg = *cfg;
    h->S = cfg->n_streams;
    h->Spad = (h->S + 31) / 32 * 32;
    h->kmax = std::max(1, cfg->max_chunks);
    h->mfma = cfg->use_mfma != 0;
    h->rr = cfg->use_mfma == 1 || cfg->use_mfma == 3;
    h->hx = cfg->use_mfma == 3;
    h->state_len = h->rr ? kStateLenRr : kStateLenLds;
    if (cfg->stream) h->stream = reinterpret_cast<hipStream_t>(cfg->stream);
    else {
        hipError_t e = hipStreamCreateWithFlags(&h->stream, hipStreamNonBlocking);
        if (e != hipSuccess) { delete h; return fail(OWW_EHIP, "hipStreamCreate: %s", hipGetErrorString(e)); }
        h->own_stream = true;
    }
    *out = h;
    return OWW_OK;
    OWW_GUARD_END
}

int oww_destroy(oww_ctx* h) {
    OWW_GUARD_BEGIN
    if (!h) return OWW_OK;
    (void)hipSetDevice(h->cfg.device);
    (void)hipStreamSynchronize(h->stream);
    if (h->up_stream) (void)hipStreamSynchronize(h->up_stream);
    if (h->down_stream) (void)hipStreamSynchronize(h->down_stream);
    free_all(h);
    if (h->own_stream) (void)hipStreamDestroy(h->stream);
    delete h;
    return OWW_OK;
    OWW_GUARD_END
}

int oww_load_mel(oww_ctx* h, const void* blob, size_t nbytes) {
    OWW_GUARD_BEGIN
    if (!h || !blob) return fail(OWW_EINVAL, "oww_load_mel: null argument");
    if (h->committed) return fail(OWW_ESTATE, "weights already committed");
    const size_t want = (400 + 32 + 32 * 16) * 4;
    if (nbytes != want) return fail(OWW_EINVAL, "oww_load_mel: blob is %zu bytes, expected %zu", nbytes, want);
    h->mel_blob.assign((const float*)blob, (const float*)blob + want / 4);
    return OWW_OK;
    OWW_GUARD_END
}

int oww_load_embedding(oww_ctx* h, const void* blob, size_t nbytes) {
    OWW_GUARD_BEGIN
    if (!h || !blob) return fail(OWW_EINVAL, "oww_load_embedding: null argument");
    if (h->committed) return fail(OWW_ESTATE, "weights already committed");
    size_t want = 0;
    for (int l = 0; l < 20; ++l) {
        want += (size_t)kLayers[l].kh * kLayers[l].kw * kLayers[l].cin * kLayers[l].cout;
        if (l < 19) want += 2 * (size_t)kLayers[l].cout;
    }
    if (nbytes != want * 4) return fail(OWW_EINVAL, "oww_load_embedding: blob is %zu bytes, expected %zu", nbytes, want * 4);
    // (a NaN weight would not fail later: the max()-based activation swallows it, in every kernel family)
    for (size_t i = 0; i < want; ++i)
        if (!std::isfinite(((const float*)blob)[i])) return fail(OWW_EINVAL, "oww_load_embedding: weights are not finite (float %zu of the blob)", i);
    h->emb_blob.assign((const float*)blob, (const float*)blob + want);
    return OWW_OK;
    OWW_GUARD_END
}

int oww_add_head(oww_ctx* h, const void* blob, size_t nbytes) {
    OWW_GUARD_BEGIN
    if (!h || !blob || nbytes < 32) return fail(OWW_EINVAL, "oww_add_head: bad argument");
    if (h->committed) return fail(OWW_ESTATE, "weights already committed");
    if ((int)h->heads.size() >= OWW_MAX_HEADS) return fail(OWW_EINVAL, "too many heads");
    const int32_t* hdr = (const int32_t*)blob;
    HeadHost hh{};
    hh.kind = hdr[0]; hh.T = hdr[1]; hh.hidden = hdr[2]; hh.n_out = hdr[3]; hh.has_ln = hdr[4];
    hh.n_blocks = 1 + hdr[5];                  // (hdr[5] = hidden blocks beyond the one every released model has; -1 = none)
    if (hh.kind == 3) {                        // train.py:85-98: 2-layer bidirectional LSTM(64) + Linear(128, n_out)
        if (hh.T < 1 || hh.T > RNN_TMAX || hh.hidden != RNN_H || hh.n_out < 1 || hh.n_out > 8 || hh.has_ln || hdr[5] != 0)
            return fail(OWW_EINVAL, "oww_add_head: bad rnn header T=%d (<= %d) hidden=%d (= %d) n_out=%d", hh.T, RNN_TMAX, hh.hidden, RNN_H, hh.n_out);
        const size_t want = 2 * ((size_t)(96 + RNN_H) * 256 + 256) + 2 * ((size_t)(128 + RNN_H) * 256 + 256) + (size_t)128 * hh.n_out + hh.n_out;
        if (nbytes != 32 + want * 4) return fail(OWW_EINVAL, "oww_add_head: rnn blob is %zu bytes, expected %zu", nbytes, 32 + want * 4);
        hh.n_blocks = 1;
        hh.blob.assign((const float*)((const char*)blob + 32), (const float*)((const char*)blob + nbytes));
        for (size_t i = 0; i < hh.blob.size(); ++i)
            if (!std::isfinite(hh.blob[i])) return fail(OWW_EINVAL, "oww_add_head: head weights are not finite (float %zu of the blob)", i);
        h->heads.push_back(std::move(hh));
        return (int)h->heads.size() - 1;
    }
    if (hh.kind < 0 || hh.kind > 2 || hh.T < 1 || hh.T > 120 || hh.hidden < 1 || hh.hidden > 512 || hh.n_out < 1 || hh.n_out > 8 ||
        hh.n_blocks < 0 || hh.n_blocks > OWW_MAX_HEAD_BLOCKS)
        return fail(OWW_EINVAL, "oww_add_head: bad header kind=%d T=%d hidden=%d n_out=%d blocks=%d", hh.kind, hh.T, hh.hidden, hh.n_out, hh.n_blocks);
    const size_t in = (size_t)hh.T * 96, H = hh.hidden, O = hh.n_out;
    const size_t per_net = in * H + H + (hh.has_ln ? 2 * H : 0) + (size_t)hh.n_blocks * (H * H + H + (hh.has_ln ? 2 * H : 0)) + H * O + O;
    const size_t n_nets = hh.kind == 1 ? 2 : 1;
    if (nbytes != 32 + per_net * n_nets * 4)
        return fail(OWW_EINVAL, "oww_add_head: blob is %zu bytes, expected %zu", nbytes, 32 + per_net * n_nets * 4);
    hh.blob.assign((const float*)((const char*)blob + 32), (const float*)((const char*)blob + nbytes));
    for (size_t i = 0; i < hh.blob.size(); ++i)
        if (!std::isfinite(hh.blob[i])) return fail(OWW_EINVAL, "oww_add_head: head weights are not finite (float %zu of the blob)", i);
    h->heads.push_back(std::move(hh));
    return (int)h->heads.size() - 1;
    OWW_GUARD_END
}

namespace {
// blob of oww_load_vad (floats after the 8-int header): gain, hann[256], 4 x (w[3][cin][cout], b[cout]), 2 x (w[128][256], b[256]), wd[64], bd
const int kVadEnc[4][2] = {{128, 16}, {16, 32}, {32, 32}, {32, 64}};
size_t vad_blob_floats() {
    size_t n = 1 + 256;
    for (auto& e : kVadEnc) n += (size_t)3 * e[0] * e[1] + e[1];
    n += 2 * ((size_t)128 * 256 + 256) + 64 + 1;
    return n;
}
}  // namespace

int oww_load_vad(oww_ctx* h, const void* blob, size_t nbytes) {
    OWW_GUARD_BEGIN
    if (!h || !blob) return fail(OWW_EINVAL, "oww_load_vad: null argument");
    if (h->committed) return fail(OWW_ESTATE, "weights already committed");
    const size_t want = 32 + vad_blob_floats() * 4;
    if (nbytes != want) return fail(OWW_EINVAL, "oww_load_vad: blob is %zu bytes, expected %zu", nbytes, want);
    const int32_t* hdr = (const int32_t*)blob;
    if (hdr[0] != 1 || hdr[1] != 256 || hdr[2] != 64 || hdr[3] != 128 || hdr[4] != 64)
        return fail(OWW_EINVAL, "oww_load_vad: unsupported geometry (version %d, n_fft %d, hop %d, bins %d, hidden %d)", hdr[0], hdr[1], hdr[2], hdr[3], hdr[4]);
    h->vad_blob.assign((const float*)((const char*)blob + 32), (const float*)((const char*)blob + nbytes));
    return OWW_OK;
    OWW_GUARD_END
}

int oww_set_calibration(oww_ctx* h, const int16_t* pcm, int32_t n_streams, int32_t n_frames) {
    OWW_GUARD_BEGIN
    if (!h) return fail(OWW_EINVAL, "oww_set_calibration: null handle");
    if (h->committed) return fail(OWW_ESTATE, "oww_set_calibration: call before oww_commit");
    h->cal_user.clear();
    if (!pcm || n_streams < 1 || n_frames < 1) return OWW_OK;          // (clears the set)
    const size_t seg = (size_t)CAL_T * OWW_CHUNK;
    const int per = (n_frames + CAL_T - 1) / CAL_T;                     // CAL_T-frame segments per stream (the last one zero-padded)
    const size_t n_seg = std::min<size_t>((size_t)n_streams * per, (size_t)(CAL_MAX_BATCHES - 1) * CAL_NP);
    h->cal_user.assign(n_seg * seg, 0);
    for (size_t k = 0; k < n_seg; ++k) {
        const size_t s = k / per, part = k % per;
        const size_t first = part * seg, n = std::min(seg, (size_t)n_frames * OWW_CHUNK - first);
        memcpy(&h->cal_user[k * seg], pcm + s * (size_t)n_frames * OWW_CHUNK + first, n * sizeof(int16_t));
    }
    return OWW_OK;
    OWW_GUARD_END
}

int oww_calibration_info(oww_ctx* h, float absmax[20], int32_t exps[21], int32_t* n_probe_streams, float selftest[3]) {
    OWW_GUARD_BEGIN
    if (!h || !h->committed) return fail(OWW_ESTATE, "oww_calibration_info: handle not committed");
    if (!h->hx) return fail(OWW_ESTATE, "oww_calibration_info: only the fp16-split family (use_mfma = 3) calibrates");
    if (absmax) memcpy(absmax, h->hx_absmax, sizeof h->hx_absmax);
    if (exps) { for (int l = 0; l < 20; ++l) exps[l] = h->hx_e[l]; exps[20] = h->hx_efeat; }
    if (n_probe_streams) *n_probe_streams = CAL_NP + (int32_t)(h->cal_user.size() / ((size_t)CAL_T * OWW_CHUNK));
    if (selftest) { selftest[0] = h->hx_selftest_err; selftest[1] = h->hx_selftest_ref; selftest[2] = h->hx_selftest_score_err; }
    return OWW_OK;
    OWW_GUARD_END
}

int oww_n_labels(const oww_ctx* h) { return h ? h->NL : 0; }

int oww_commit(oww_ctx* h) {
    OWW_GUARD_BEGIN
    if (!h) return fail(OWW_EINVAL, "null handle");
    if (h->committed) return fail(OWW_ESTATE, "already committed");
    if (h->mel_blob.empty() || h->emb_blob.empty()) return fail(OWW_ESTATE, "mel and embedding weights must be loaded before commit");
    HIPCHK(hipSetDevice(h->cfg.device));

    // ---- nets ----
    h->nets.clear(); h->head_nets.clear();
    int col = 0, maxT = 16, hmax = 1;
    for (size_t hi = 0; hi < h->heads.size(); ++hi) {
        HeadHost& hh = h->heads[hi];
        hh.out_col = col;
        const size_t in = (size_t)hh.T * 96, H = hh.hidden, O = hh.n_out;
        const float* q = hh.blob.data();
        const int begin = (int)h->nets.size();
        if (hh.kind == 3) {                                                  // recurrent head: one net, its blob as a whole
            NetHost n{};
            n.hidden = hh.hidden; n.n_out = hh.n_out; n.has_ln = 0; n.T = hh.T; n.final_act = hh.n_out == 1 ? 0 : 1;
            n.head = (int)hi; n.role = 0; n.out_col = col; n.n_blocks = 0;
            n.rnn = q; n.rnn_floats = hh.blob.size();
            h->nets.push_back(n);
        }
        for (int r = 0; r < (hh.kind == 3 ? 0 : hh.kind == 1 ? 2 : 1); ++r) {
            NetHost n{};
            n.hidden = hh.hidden; n.n_out = hh.n_out; n.has_ln = hh.has_ln; n.T = hh.T;
            n.final_act = hh.kind == 2 ? 1 : 0; n.head = (int)hi; n.role = r; n.out_col = col;
            n.n_blocks = hh.n_blocks;
            n.w1 = q; q += in * H; n.b1 = q; q += H;
            if (hh.has_ln) { n.ln1g = q; q += H; n.ln1b = q; q += H; }
            n.blocks = q;
            if (hh.n_blocks > 0) {                                           // (block 0 by name: what the MFMA head kernels read)
                n.w2 = q; n.b2 = q + H * H;
                if (hh.has_ln) { n.ln2g = n.b2 + H; n.ln2b = n.ln2g + H; }
            }
            q += (size_t)hh.n_blocks * (H * H + H + (hh.has_ln ? 2 * H : 0));
            n.w3 = q; q += H * O; n.b3 = q; q += O;
            h->nets.push_back(n);
        }
        h->head_nets.push_back({begin, (int)h->nets.size()});
        col += hh.n_out;
        maxT = std::max(maxT, hh.T);
        hmax = std::max(hmax, hh.hidden);
    }
    h->NL = col;
    if (h->NL > OWW_MAX_LABELS) return fail(OWW_EINVAL, "too many labels (%d)", h->NL);
    h->TR = h->cfg.feature_ring > 0 ? std::max(h->cfg.feature_ring, maxT) : maxT;
    h->generic_hmax = hmax;

    // ---- f16-split family: per-layer activation scales from a calibration run on the exact-fp32 kernels (calibrate_hx) ----
    CommitClock clk(h->hx ? "f16-split" : "family");
    HxCalib cal;
    if (h->hx) {
        if (int rc = calibrate_hx(h, cal)) return rc;
        clk.lap("calibration (total)");
        if (getenv("OWW_DEBUG_CALIB"))
            for (int l = 0; l < 20; ++l) fprintf(stderr, "calib layer %2d: max|a| %-12.5g e_in %4d e_out %4d\n", l, h->hx_absmax[l], h->hx_ein[l], h->hx_e[l]);
    }

    // ---- device weight image ----
    HostBuf hb;
    const size_t o_hann = hb.add(h->mel_blob.data(), 400);
    const size_t o_start = hb.add(h->mel_blob.data() + 400, 32);          // int32 bit patterns
    const size_t o_taps = hb.add(h->mel_blob.data() + 432, 512);
    // fused front end: the sparse mel taps read a COMPACT copy of each frame's power row -- one segment per mel bin, [first tap ..
    // last non-zero tap] -- whose segment starts have pairwise different residues mod 32, so that the 32 lanes of a tap read hit 32
    // different LDS banks (the plain power row gave three bins per bank for every tap: most of the launch's bank conflicts).  Every
    // power bin belongs to at most two triangular filters, hence two destinations per bin (mel_dst: lo / hi 16 bits; kMelJunk = none).
    size_t o_meloff = 0, o_meldst = 0;
    {
        constexpr int kMelTable = 250, kMelJunk = 250;           // floats of a frame's table; bins without a second filter store here
        const int32_t* start = reinterpret_cast<const int32_t*>(h->mel_blob.data() + 400);
        const float* taps = h->mel_blob.data() + 432;
        int nz[32], first[32];
        for (int m = 0; m < 32; ++m) {
            nz[m] = 0;
            for (int t = 0; t < 16; ++t) if (taps[m * 16 + t] != 0.f) nz[m] = t + 1;
            first[m] = start[m] - 2;                              // power-row index of tap 0 (the kernels keep FFT bins 2..121)
        }
        int off[32], best[32], best_end = 1 << 30;
        uint64_t st = 0x9E3779B97F4A7C15ull;
        for (int it = 0; it < 20000 && best_end > kMelTable; ++it) {                // randomised first-fit; a few hundred tries are enough
            int order[32];
            for (int i = 0; i < 32; ++i) order[i] = i;
            for (int i = 31; i > 0; --i) { st ^= st << 13; st ^= st >> 7; st ^= st << 17; std::swap(order[i], order[st % (uint64_t)(i + 1)]); }
            unsigned used = 0; int cur = 0, end = 0;
            for (int i = 0; i < 32; ++i) {
                const int m = order[i];
                int o = cur;
                while (used >> (o & 31) & 1u) ++o;
                used |= 1u << (o & 31); off[m] = o; cur = o + nz[m];
                end = std::max(end, o + 16);                      // a lane reads 16 taps from its start
            }
            if (end < best_end) { best_end = end; memcpy(best, off, sizeof off); }
        }
        if (best_end > kMelTable) {                               // (cannot happen for a 32-filter bank of <= 16 taps; plain prefix layout)
            int cur = 0;
            for (int m = 0; m < 32; ++m) { best[m] = cur; cur += nz[m]; }
            if (cur + 16 > kMelTable) return fail(OWW_EINVAL, "oww_commit: the mel filter bank has more than %d taps", kMelTable - 16);
        }
        std::vector<float> dst(128, 0.f);
        for (int i = 0; i < 120; ++i) {
            uint32_t d[2] = {kMelJunk, kMelJunk}; int n = 0;
            for (int m = 0; m < 32; ++m) {
                const int t = i - first[m];
                if (t >= 0 && t < nz[m] && taps[m * 16 + t] != 0.f) {
                    if (n == 2) return fail(OWW_EINVAL, "oww_commit: FFT bin %d feeds more than two mel filters (not a triangular filter bank)", i + 2);
                    d[n++] = (uint32_t)(best[m] + t);
                }
            }
            const uint32_t packed = d[0] | (d[1] << 16);
            memcpy(&dst[i], &packed, 4);
        }
        std::vector<float> offf(32);
        for (int m = 0; m < 32; ++m) { const int32_t v = best[m]; memcpy(&offf[m], &v, 4); }
        o_meloff = hb.add(offf.data(), 32);
        o_meldst = hb.add(dst.data(), 128);
    }
    size_t o_conv[20], o_scale[20] = {}, o_shift[20] = {};
    {
        const float* q = h->emb_blob.data();
        std::vector<float> pk;
        for (int l = 0; l < 20; ++l) {
            const LayerDef& L = kLayers[l];
            const size_t nw = (size_t)L.kh * L.kw * L.cin * L.cout;
            const float* bn_scale = l < 19 ? q + nw : nullptr;             // folded inference BatchNorm of this layer (blob order: w, scale, shift)
            if (!h->mfma) o_conv[l] = hb.add(q, nw);
            else if (h->hx) {
                // fold_cnn: W' = s w 2^(e_out - e_in) per output channel (calibrate_hx's scale ladder; e_in = 0 for the mel input)
                std::vector<double> colmul(L.cout);
                const int de = h->hx_e[l] - h->hx_ein[l];
                for (int c = 0; c < L.cout; ++c) colmul[c] = std::ldexp(bn_scale ? (double)bn_scale[c] : 1.0, de);
                HxFold fold; fold.colmul = colmul.data();
                const bool time_merged = OWH_KMERGE && L.kh == 3 && L.kw == 1 && ((L.cin + 15) / 16) % 2 == 1 &&
                                         (L.cin % 16 == 8 || OWH_KMERGE_B);                     // stage C (and B): layers b, d
                // 1x3 layers with a 72-channel input (stage C layer c, stage D layer a): owh::conv_mel_hxm, same packing rule
                const bool mel_merged = OWH_KMERGE_MEL && owh::kInterleave && L.kh == 1 && L.kw == 3 &&
                                        ((L.cin + 15) / 16) % 2 == 1 && (L.cin + 15) / 16 >= 3 &&
                                        (L.cin % 16 == 8 || (OWH_KMERGE_MEL2 && l == 7 && OWH_WPS_C == 2) ||   // (l == 7: stage C layer a, 48 -> 72)
                                         (OWH_KMERGE_MEL2B && l == 5));                                   // (l == 5: stage B layer c, A/B switch)
                if (l == 0) pack_hx_conv0(q, pk, &fold);
                else if (l <= 2) pack_hx_stage_a(q, l, pk, &fold);
                else if (time_merged || mel_merged) pack_hx_tm(q, L.cin, L.cout, pk, &fold);
                else pack_hx(q, 3, L.cin, L.cout, pk, &fold, OWH_REM2 && !OWH_KMERGE_B && !OWH_KMERGE_MEL2B && l >= 4 && l <= 6);     // (stage B layers b, c, d)
                if (!(fold.absmax < 65000.0))
                    return fail(OWW_ERANGE, "conv layer %d: folded weight magnitude %.3g (BatchNorm scale x weight x activation-scale ratio 2^%d) is outside "
                                "the f16 range of the fp16-split kernels (use_mfma = 3); use use_mfma = 1", l, fold.absmax, de);
                o_conv[l] = hb.add(pk);
            }
            else if (h->rr && l > 0) { pack_rr(q, 3, L.cin, L.cout, pk); o_conv[l] = hb.add(pk); }
            else if (l == 0) {
                // conv0: K = 9 taps padded to 12 -> three k-steps; lane (i, j) of k-step s holds w[k = 4s+j][cout = 16ct+i]
                pk.assign(2 * 3 * 64, 0.f);
                for (int ct = 0; ct < 2; ++ct)
                    for (int s = 0; s < 3; ++s)
                        for (int lane = 0; lane < 64; ++lane) {
                            const int k = 4 * s + (lane >> 4), co = ct * 16 + (lane & 15);
                            if (k < 9 && co < 24) pk[(ct * 3 + s) * 64 + lane] = q[k * 24 + co];
                        }
                o_conv[l] = hb.add(pk);
            } else { pack_mfma(q, 3, L.cin, L.cout, pk); o_conv[l] = hb.add(pk); }
            q += nw;
            if (l < 19) {
                // zero padded to whole 16-channel tiles: the register-resident kernels evaluate the pad channels (as zeros)
                std::vector<float> pad((size_t)(L.cout + 15) / 16 * 16, 0.f);
                if (h->hx) {
                    // f16-split family: the scale lives in the weights.  Slot "scale" is only read by conv0: the third operand of the
                    // med3 that applies its ReLU in the folded form (+inf where the BatchNorm scale is >= 0, -inf where it is negative)
                    std::vector<float> bound(L.cout);
                    for (int c = 0; c < L.cout; ++c) bound[c] = q[c] < 0.f ? -INFINITY : INFINITY;
                    pad_hx_rows(bound.data(), L.cout, 1.0f, pad);
                }
                else memcpy(pad.data(), q, L.cout * sizeof(float));
                o_scale[l] = hb.add(pad); q += L.cout;
                std::fill(pad.begin(), pad.end(), 0.f);
                if (h->hx) pad_hx_rows(q, L.cout, std::ldexp(1.0f, h->hx_e[l]), pad);       // accumulator start values K h, tile row order
                else memcpy(pad.data(), q, L.cout * sizeof(float));
                o_shift[l] = hb.add(pad); q += L.cout;
            }
        }
    }
    // heads: natural arrays for every net (+ packed w2 for hidden==64), fast groups
    struct NetOff { size_t w1, b1, ln1g, ln1b, w2, b2, ln2g, ln2b, w3, b3, w2pk, blocks, rnn; };
    std::vector<NetOff> noff(h->nets.size());
    for (size_t ni = 0; ni < h->nets.size(); ++ni) {
        const NetHost& n = h->nets[ni];
        if (n.rnn) { noff[ni] = NetOff{}; noff[ni].rnn = hb.add(n.rnn, n.rnn_floats); continue; }
        const size_t in = (size_t)n.T * 96, H = n.hidden, O = n.n_out;
        NetOff& o = noff[ni];
        o.w1 = hb.add(n.w1, in * H); o.b1 = hb.add(n.b1, H);
        o.ln1g = n.has_ln ? hb.add(n.ln1g, H) : 0; o.ln1b = n.has_ln ? hb.add(n.ln1b, H) : 0;
        const size_t blk = H * H + H + (n.has_ln ? 2 * H : 0);
        o.blocks = n.n_blocks > 0 ? hb.add(n.blocks, (size_t)n.n_blocks * blk) : 0;      // (the generic kernel walks them in place)
        o.w2 = o.b2 = o.ln2g = o.ln2b = 0;
        if (n.n_blocks > 0) {
            o.w2 = hb.add(n.w2, H * H); o.b2 = hb.add(n.b2, H);
            o.ln2g = n.has_ln ? hb.add(n.ln2g, H) : 0; o.ln2b = n.has_ln ? hb.add(n.ln2b, H) : 0;
        }
        o.w3 = hb.add(n.w3, H * O); o.b3 = hb.add(n.b3, O);
        o.w2pk = 0;
        if (n.hidden == 64 && n.n_blocks == 1) { std::vector<float> pk; pack_mfma(n.w2, 1, 64, 64, pk); o.w2pk = hb.add(pk); }
    }
    // grouping: heads whose nets are all (hidden 64, n_out 1, sigmoid) share a fast group per T (<= 8 nets each)
    h->groups.clear(); h->generic_nets.clear(); h->rnn_nets.clear();
    struct GOff { size_t w1pk, b1cat, w1hx; std::vector<size_t> w2hx, pad, w3hx; };
    std::vector<GOff> goff;
    for (size_t hi = 0; hi < h->heads.size(); ++hi) {
        const auto [nb, ne] = h->head_nets[hi];
        if (h->nets[nb].rnn) { h->rnn_nets.push_back(nb); continue; }       // recurrent heads have their own kernel in every family
        bool fast = h->mfma;
        // the MFMA head kernels: sigmoid nets of one output and one hidden block; exactly 64 hidden units in the fp32 family
        // (heads64_kernel), up to 64 in the fp16-split family (zero-padded, see FastGroup::d_pad)
        for (int ni = nb; ni < ne; ++ni)
            fast = fast && (h->nets[ni].hidden == 64 || (h->hx && h->nets[ni].hidden <= 64)) && h->nets[ni].n_out == 1 &&
                   h->nets[ni].final_act == 0 && h->nets[ni].n_blocks == 1;
        // the wide form of the fp16-split heads kernel: ungated nets of up to 128 hidden units and 8 outputs with one hidden block
        // (the released multiclass `timer`: docs/models/timers.md:9-27; train.py's default layer_dim = 128)
        bool wide = !fast && h->hx && h->mfma && ne - nb == 1 && !getenv("OWW_NO_WIDE_HEADS");
        for (int ni = nb; ni < ne; ++ni) wide = wide && h->nets[ni].hidden <= 128 && h->nets[ni].n_out <= 8 && h->nets[ni].n_blocks == 1 && h->nets[ni].role == 0;
        if (!fast && !wide) { for (int ni = nb; ni < ne; ++ni) h->generic_nets.push_back(ni); continue; }
        FastGroup* g = nullptr;
        const int ht = wide ? 8 : 4;
        const int cap = h->hx ? 16 / ht : HD_MAXNETS;                // heads_hx_kernel: at most sixteen hidden tiles per launch
        for (auto& gg : h->groups) if (gg.T == h->nets[nb].T && gg.ht == ht && gg.n_nets + (ne - nb) <= cap) { g = &gg; break; }
        if (!g) { h->groups.push_back(FastGroup{}); g = &h->groups.back(); g->T = h->nets[nb].T; g->n_nets = 0; g->ht = ht; }
        for (int ni = nb; ni < ne; ++ni) { g->nets.push_back(ni); g->n_nets++; }
    }
    for (auto& g : h->groups) {
        if (g.ht == 8) {
            // ---- wide group: hidden columns padded to 128 per net
            const int HP = 128;
            g.NH = HP * g.n_nets;
            const size_t K = (size_t)g.T * 96;
            std::vector<float> wcat(K * g.NH, 0.f), pk;
            std::vector<double> colmul(g.NH);
            GOff go{0, 0, 0, {}, {}, {}};
            for (int gi = 0; gi < g.n_nets; ++gi) {
                NetHost& n = h->nets[g.nets[gi]];
                const size_t H = n.hidden, O = n.n_out;
                for (size_t k = 0; k < K; ++k) memcpy(&wcat[k * g.NH + HP * gi], n.w1 + k * H, H * sizeof(float));
                n.hx_e1 = hx_weight_exp(n.w1, K * H); n.hx_e2 = hx_weight_exp(n.w2, H * H); n.hx_e3 = hx_weight_exp(n.w3, H * O);
                if (n.hx_e1 == -1000 || n.hx_e2 == -1000 || n.hx_e3 == -1000) return fail(OWW_EINVAL, "head weights are not finite");
                for (int c = 0; c < HP; ++c) colmul[HP * gi + c] = std::ldexp(1.0, n.hx_e1);
            }
            pack_hx_w1(wcat.data(), (int)K, g.NH, colmul.data(), pk);
            go.w1hx = hb.add(pk);
            for (int gi = 0; gi < g.n_nets; ++gi) {
                const NetHost& n = h->nets[g.nets[gi]];
                const size_t H = n.hidden, O = n.n_out;
                std::vector<double> cm2(HP, std::ldexp(1.0, n.hx_e2)), cm3(16, std::ldexp(1.0, n.hx_e3));
                HxFold f2; f2.colmul = cm2.data();
                std::vector<float> w2p((size_t)HP * HP, 0.f);            // [in 128][out 128], zero rows / columns beyond H
                for (size_t i = 0; i < H; ++i) memcpy(&w2p[i * HP], n.w2 + i * H, H * sizeof(float));
                pack_hx(w2p.data(), 1, HP, HP, pk, &f2); go.w2hx.push_back(hb.add(pk));
                HxFold f3; f3.colmul = cm3.data();
                std::vector<float> w3p((size_t)HP * 16, 0.f);            // [in 128][out 16], outputs O .. 15 zero
                for (size_t i = 0; i < H; ++i) memcpy(&w3p[i * 16], n.w3 + i * O, O * sizeof(float));
                pack_hx(w3p.data(), 1, HP, 16, pk, &f3); go.w3hx.push_back(hb.add(pk));
                std::vector<float> pad(6 * HP + 16, 0.f);                // b1, ln1g, ln1b, b2, ln2g, ln2b | b3
                const float* src[6] = {n.b1, n.has_ln ? n.ln1g : nullptr, n.has_ln ? n.ln1b : nullptr, n.b2,
                                       n.has_ln ? n.ln2g : nullptr, n.has_ln ? n.ln2b : nullptr};
                for (int a = 0; a < 6; ++a) if (src[a]) memcpy(&pad[a * HP], src[a], H * sizeof(float));
                memcpy(&pad[6 * HP], n.b3, O * sizeof(float));
                go.pad.push_back(hb.add(pad));
            }
            goff.push_back(go);
            continue;
        }
        g.NH = 64 * g.n_nets;
        const size_t K = (size_t)g.T * 96;
        std::vector<float> wcat(K * g.NH, 0.f), bcat(g.NH, 0.f), pk;
        for (int gi = 0; gi < g.n_nets; ++gi) {
            const NetHost& n = h->nets[g.nets[gi]];
            const size_t H = n.hidden;                                   // (<= 64; columns H .. 63 of the net's block stay zero)
            for (size_t k = 0; k < K; ++k) memcpy(&wcat[k * g.NH + 64 * gi], n.w1 + k * H, H * sizeof(float));
            memcpy(&bcat[64 * gi], n.b1, H * sizeof(float));
        }
        pack_mfma(wcat.data(), g.T, 96, g.NH, pk);
        GOff go{hb.add(pk), hb.add(bcat), 0, {}, {}, {}};
        if (h->hx) {
            // every net's two matrices on their own power-of-two scale (hx_weight_exp); undone on the fp32 accumulators (HeadHxNet::u1, u2)
            std::vector<double> colmul(g.NH);
            for (int gi = 0; gi < g.n_nets; ++gi) {
                NetHost& n = h->nets[g.nets[gi]];
                const size_t H = n.hidden;
                n.hx_e1 = hx_weight_exp(n.w1, K * H); n.hx_e2 = hx_weight_exp(n.w2, H * H);
                if (n.hx_e1 == -1000 || n.hx_e2 == -1000) return fail(OWW_EINVAL, "head weights are not finite");
                for (int c = 0; c < 64; ++c) colmul[64 * gi + c] = std::ldexp(1.0, n.hx_e1);
            }
            pack_hx_w1(wcat.data(), (int)K, g.NH, colmul.data(), pk);
            go.w1hx = hb.add(pk);
            for (int gi = 0; gi < g.n_nets; ++gi) {
                const NetHost& n = h->nets[g.nets[gi]];
                const size_t H = n.hidden;
                std::vector<double> cm2(64, std::ldexp(1.0, n.hx_e2));
                HxFold fold; fold.colmul = cm2.data();
                std::vector<float> w2p(64 * 64, 0.f);                    // [in 64][out 64], zero rows / columns beyond H
                for (size_t i = 0; i < H; ++i) memcpy(&w2p[i * 64], n.w2 + i * H, H * sizeof(float));
                pack_hx(w2p.data(), 1, 64, 64, pk, &fold); go.w2hx.push_back(hb.add(pk));
                std::vector<float> pad(7 * 64, 0.f);                     // b1, ln1g, ln1b, b2, ln2g, ln2b, w3
                const float* src[7] = {n.b1, n.has_ln ? n.ln1g : nullptr, n.has_ln ? n.ln1b : nullptr, n.b2,
                                       n.has_ln ? n.ln2g : nullptr, n.has_ln ? n.ln2b : nullptr, n.w3};
                for (int a = 0; a < 7; ++a) if (src[a]) memcpy(&pad[a * 64], src[a], H * sizeof(float));
                go.pad.push_back(hb.add(pad));
            }
        }
        goff.push_back(go);
    }
    // voice-activity stand-in (always fp16-split MFMA kernels, whatever the CNN family)
    size_t o_vhann = 0, o_vencw = 0, o_vencb = 0, o_vlw = 0, o_vlb = 0, o_vwd = 0;
    h->vad = !h->vad_blob.empty();
    if (h->vad) {
        const float* q = h->vad_blob.data();
        h->vad_gain = *q++;
        o_vhann = hb.add(q, 256); q += 256;
        std::vector<float> encw, encb(4 * 64, 0.f), pk;
        for (int l = 0; l < 4; ++l) {
            const int cin = kVadEnc[l][0], cout = kVadEnc[l][1];
            if (!hx_in_range(q, (size_t)3 * cin * cout)) return fail(OWW_EINVAL, "VAD encoder weights too large for the fp16-split kernels");
            pack_hx(q, 3, cin, cout, pk);
            encw.insert(encw.end(), pk.begin(), pk.end());
            q += (size_t)3 * cin * cout;
            memcpy(&encb[l * 64], q, cout * sizeof(float)); q += cout;
        }
        if (encw.size() != (size_t)owv::V_WFLOATS) return fail(OWW_EINVAL, "internal: VAD encoder image is %zu floats, expected %d", encw.size(), owv::V_WFLOATS);
        o_vencw = hb.add(encw); o_vencb = hb.add(encb);
        std::vector<float> lw, lb;
        for (int l = 0; l < 2; ++l) {
            if (!hx_in_range(q, (size_t)128 * 256)) return fail(OWW_EINVAL, "VAD LSTM weights too large for the fp16-split kernels");
            // columns regrouped so that the four gates of hidden tile u are neighbours: column 16 (4u + gate) + i <- gate * 64 + 16u + i
            std::vector<float> perm((size_t)128 * 256);
            for (int k = 0; k < 128; ++k)
                for (int u = 0; u < 4; ++u)
                    for (int gt = 0; gt < 4; ++gt)
                        for (int i = 0; i < 16; ++i) perm[(size_t)k * 256 + 16 * (4 * u + gt) + i] = q[(size_t)k * 256 + gt * 64 + 16 * u + i];
            pack_hx(perm.data(), 1, 128, 256, pk);
            lw.insert(lw.end(), pk.begin(), pk.end());
            q += (size_t)128 * 256;
            lb.insert(lb.end(), q, q + 256); q += 256;
        }
        o_vlw = hb.add(lw); o_vlb = hb.add(lb);
        o_vwd = hb.add(q, 64); q += 64;
        h->vad_bd = *q;
    }
    clk.lap("weight packing (host)");
    HIPCHK(dev_alloc(&h->d_w, hb.data.size() * sizeof(float)));
    HIPCHK(copy_sync(h->d_w, hb.data.data(), hb.data.size() * sizeof(float), hipMemcpyHostToDevice));
    h->d_hann = h->d_w + o_hann; h->d_mstart = reinterpret_cast<const int*>(h->d_w + o_start); h->d_taps = h->d_w + o_taps;
    h->d_meloff = reinterpret_cast<const int*>(h->d_w + o_meloff); h->d_meldst = reinterpret_cast<const unsigned*>(h->d_w + o_meldst);
    if (h->vad) {
        h->d_vad_hann = h->d_w + o_vhann; h->d_vad_encw = h->d_w + o_vencw; h->d_vad_encb = h->d_w + o_vencb;
        h->d_vad_lstmw = h->d_w + o_vlw; h->d_vad_lstmb = h->d_w + o_vlb; h->d_vad_wd = h->d_w + o_vwd;
    }
    for (int l = 0; l < 20; ++l) {
        h->d_conv[l] = h->d_w + o_conv[l];
        h->d_scale[l] = l < 19 ? h->d_w + o_scale[l] : nullptr;
        h->d_shift[l] = l < 19 ? h->d_w + o_shift[l] : nullptr;
    }
    auto make_desc = [&](int ni, int hid_off) {
        const NetHost& n = h->nets[ni];
        const NetOff& o = noff[ni];
        NetDesc d{};
        d.hidden = n.hidden; d.n_out = n.n_out; d.has_ln = n.has_ln; d.final_act = n.final_act; d.T = n.T;
        d.head = n.head; d.role = n.role; d.out_col = n.out_col; d.hid_off = hid_off;
        if (n.rnn) { d.rnn = h->d_w + o.rnn; return d; }
        d.w1 = h->d_w + o.w1; d.b1 = h->d_w + o.b1;
        d.ln1g = n.has_ln ? h->d_w + o.ln1g : nullptr; d.ln1b = n.has_ln ? h->d_w + o.ln1b : nullptr;
        d.n_blocks = n.n_blocks;
        d.blocks = n.n_blocks > 0 ? h->d_w + o.blocks : nullptr;
        d.w2 = n.n_blocks > 0 ? h->d_w + o.w2 : nullptr; d.b2 = n.n_blocks > 0 ? h->d_w + o.b2 : nullptr;
        d.ln2g = n.has_ln && n.n_blocks > 0 ? h->d_w + o.ln2g : nullptr; d.ln2b = n.has_ln && n.n_blocks > 0 ? h->d_w + o.ln2b : nullptr;
        d.w3 = h->d_w + o.w3; d.b3 = h->d_w + o.b3;
        d.w2pk = n.hidden == 64 && n.n_blocks == 1 ? h->d_w + o.w2pk : nullptr;
        return d;
    };
    if (!h->nets.empty()) {
        std::vector<NetDesc> all;
        for (size_t ni = 0; ni < h->nets.size(); ++ni) all.push_back(make_desc((int)ni, 0));
        h->host_descs = all;
        HIPCHK(dev_alloc(&h->d_allnets, all.size() * sizeof(NetDesc)));
        HIPCHK(copy_sync(h->d_allnets, all.data(), all.size() * sizeof(NetDesc), hipMemcpyHostToDevice));
    }
    for (size_t gi = 0; gi < h->groups.size(); ++gi) {
        FastGroup& g = h->groups[gi];
        std::vector<NetDesc> ds;
        for (int i = 0; i < g.n_nets; ++i) ds.push_back(make_desc(g.nets[i], 64 * i));
        HIPCHK(dev_alloc(&g.d_nets, ds.size() * sizeof(NetDesc)));
        HIPCHK(copy_sync(g.d_nets, ds.data(), ds.size() * sizeof(NetDesc), hipMemcpyHostToDevice));
        g.d_w1pk = h->d_w + goff[gi].w1pk; g.d_b1cat = h->d_w + goff[gi].b1cat;
        if (h->hx) {
            g.d_w1hx = h->d_w + goff[gi].w1hx;
            for (size_t o : goff[gi].w2hx) g.d_w2hx.push_back(h->d_w + o);
            for (size_t o : goff[gi].pad) g.d_pad.push_back(h->d_w + o);
            for (size_t o : goff[gi].w3hx) g.d_w3hx.push_back(h->d_w + o);
        }
        if (g.ht == 4) if (int rc = set_lds(heads64_kernel, heads_lds_bytes(g.NH))) return rc;
    }
    if (!h->rnn_nets.empty()) if (int rc = set_lds(heads_rnn_kernel<RNN_SPW>, (int)rnn_lds_bytes(RNN_TMAX))) return rc;

    // ---- sticky range flag of the f16-split kernels: page-locked + device-mapped, so the host reads it without a copy ----
    {
        void* dp = nullptr;
        HIPCHK(hipHostMalloc((void**)&h->h_range, 64, hipHostMallocMapped));
        h->h_range[0] = 0; h->h_range[1] = -1;
        HIPCHK(hipHostGetDevicePointer(&dp, h->h_range, 0));
        h->d_range = (int*)dp;
    }
    clk.lap("weight upload");
    // ---- state ----
    const size_t SP = h->Spad;
    for (int a = 0; a < N_STATE; ++a) {
        if (int rc = dalloc(h->stream, &h->d_state[a], SP * h->state_len[a])) return rc;
        if (int rc = dalloc(h->stream, &h->d_tmpl[a], (size_t)h->state_len[a] * (h->rr ? kStateSpgRr[a] : 1))) return rc;
    }
    const int* xlen = h->rr ? kXLenRr : kXLenLds;
    if (int rc = dalloc(h->stream, &h->d_xA, SP * xlen[0])) return rc;
    if (int rc = dalloc(h->stream, &h->d_xB, SP * xlen[1])) return rc;
    if (int rc = dalloc(h->stream, &h->d_xC, SP * xlen[2])) return rc;
    if (int rc = dalloc(h->stream, &h->d_xD, SP * xlen[3])) return rc;
    if (int rc = dalloc(h->stream, &h->d_mel, SP * 8 * h->kmax * 32)) return rc;
    if (int rc = dalloc(h->stream, &h->d_feat, SP * h->TR * 96)) return rc;
    if (int rc = dalloc(h->stream, &h->d_emb, SP * 96)) return rc;
    if (int rc = dalloc(h->stream, &h->d_raw, SP * std::max(h->NL, 1))) return rc;
    if (int rc = dalloc(h->stream, &h->d_scores, SP * std::max(h->NL, 1))) return rc;
    if (int rc = dalloc(h->stream, &h->d_ring, SP * std::max(h->NL, 1) * OWW_SCORE_RING)) return rc;
    if (int rc = dalloc(h->stream, &h->d_featinit, (size_t)h->TR * 96)) return rc;
    if (int rc = dalloc(h->stream, &h->d_nfeat, SP)) return rc;
    if (int rc = dalloc(h->stream, &h->d_npred, SP)) return rc;
    if (int rc = dalloc(h->stream, &h->d_vadring, SP * 8)) return rc;
    if (int rc = dalloc(h->stream, &h->d_nvad, SP)) return rc;
    if (int rc = dalloc(h->stream, &h->d_vadin, SP)) return rc;
    if (h->vad) {
        const size_t G = (SP + 15) / 16;
        if (int rc = dalloc(h->stream, &h->d_vadx, G * 4 * 1024)) return rc;
        if (int rc = dalloc(h->stream, &h->d_vadhc, G * 4096)) return rc;
        if (int rc = dalloc(h->stream, &h->d_vadlast, SP)) return rc;
        if (int rc = set_lds(owv::vad_front_kernel, owv::V_LDS_BYTES)) return rc;
    }
    if (int rc = dalloc(h->stream, &h->d_tail, SP * 480)) return rc;
    if (int rc = dalloc(h->stream, &h->d_pcm, (size_t)h->S * OWW_CHUNK * h->kmax)) return rc;
    if (int rc = dalloc(h->stream, &h->d_patience, (size_t)std::max(h->NL, 1))) return rc;
    if (int rc = dalloc(h->stream, &h->d_threshold, (size_t)std::max(h->NL, 1))) return rc;
    {
        std::vector<float> nanv(std::max(h->NL, 1), NAN);
        HIPCHK(copy_async(h->d_threshold, nanv.data(), nanv.size() * sizeof(float), hipMemcpyHostToDevice, h->stream));   // (behind the buffer's zero fill, same stream)
    }
    if (h->cfg.debug_layers) if (int rc = dalloc(h->stream, &h->d_dbg, SP * DBG_FLOATS)) return rc;
    if (const char* e = getenv("OWW_PROF_BLOCK")) { h->prof_block = atoi(e); if (int rc = dalloc(h->stream, &h->d_prof, (size_t)4 * 256)) return rc; }

    // block-pipelined step: OFF by default.  Measured with the round-3 kernels at 131,072 x 3 (same box, OWW_BLOCKS = 1 / 2 / 3 / 4):
    // 6.04 / 6.10 / 6.25 / 6.32 ms per step -- two kernels sharing the chip gain nothing now that the front end no longer stalls
    // (round 2, two handles on two streams: -3.6 %).  OWW_BLOCKS=2..4 switches it on for large handles (experiments).
    if (h->hx && h->S >= 16384) {
        h->n_blocks = 1;
        if (const char* e = getenv("OWW_BLOCKS")) h->n_blocks = std::min(4, std::max(1, atoi(e)));
        if (h->n_blocks > 1) {
            HIPCHK(hipEventCreateWithFlags(&h->blk_fork, hipEventDisableTiming));
            for (int b = 0; b < h->n_blocks; ++b) {
                HIPCHK(hipStreamCreateWithFlags(&h->blk_stream[b], hipStreamNonBlocking));
                HIPCHK(hipEventCreateWithFlags(&h->blk_done[b], hipEventDisableTiming));
            }
        }
    }
    h->fuse = h->hx && !getenv("OWW_NO_FUSE");
    if (const char* e = getenv("OWW_DEEP_WGS")) h->deep_wgs = atoi(e);
    if (const char* e = getenv("OWW_SMALL_WGS")) h->small_wgs = atoi(e);
    if (const char* e = getenv("OWW_SMALL_WGS_HEADS")) h->small_wgs_heads = atoi(e);
    if (const char* e = getenv("OWW_GENERIC_SPW")) { const int v = atoi(e); h->generic_spw = v == 4 || v == 16 ? v : 0; }
    if (const char* e = getenv("OWW_RING3_ALWAYS")) for (int i = 0; i < 4; ++i) h->ring3_always[i] = strchr(e, "BCDE"[i]) != nullptr;
    h->post_in_heads = h->hx && !getenv("OWW_NO_FUSE") && h->groups.size() == 1 && h->groups[0].ht == 4 && h->generic_nets.empty() && h->rnn_nets.empty() && h->NL > 0;                  // (A/B switch: OWW_NO_FUSE=1 keeps the separate mel kernel)
    if (int rc = set_lds(owf::hmelA_kernel<false>, owf::FA_LDS_BYTES)) return rc;
    if (int rc = set_lds(owf::hmelA_kernel<true>, owf::FA_LDS_BYTES)) return rc;
    if (int rc = set_lds(stageA_kernel<true>, CfgA::LDS_BYTES)) return rc;
    if (int rc = set_lds(stageA_kernel<false>, CfgA::LDS_BYTES)) return rc;
    if (int rc = set_lds(stage_kernel<CfgB, true, false>, CfgB::LDS_BYTES)) return rc;
    if (int rc = set_lds(stage_kernel<CfgB, false, false>, CfgB::LDS_BYTES)) return rc;
    if (int rc = set_lds(stage_kernel<CfgC, true, false>, CfgC::LDS_BYTES)) return rc;
    if (int rc = set_lds(stage_kernel<CfgC, false, false>, CfgC::LDS_BYTES)) return rc;
    if (int rc = set_lds(stage_kernel<CfgD, true, false>, CfgD::LDS_BYTES)) return rc;
    if (int rc = set_lds(stage_kernel<CfgD, false, false>, CfgD::LDS_BYTES)) return rc;
    if (int rc = set_lds(stage_kernel<CfgE, true, true>, CfgE::LDS_BYTES)) return rc;
    if (int rc = set_lds(stage_kernel<CfgE, false, true>, CfgE::LDS_BYTES)) return rc;

    // ---- reset state = what an all-ones mel history leaves behind (utils.py:165 melspectrogram_buffer =
    //      ones((76,32))): run the incremental CNN on ones rows until the zero start is flushed out ----
    clk.lap("state allocation");
    {
        const int warm = std::min<int>(32, (int)SP);
        hipLaunchKernelGGL(fill_kernel, dim3((warm * 256 + 255) / 256), dim3(256), 0, h->stream, h->d_mel, (size_t)warm * 256, 1.0f);
        float* saved_dbg = h->d_dbg; h->d_dbg = nullptr;
        for (int it = 0; it < 12; ++it)
            if (int rc = run_cnn(h, warm, 256, 0)) return rc;
        h->d_dbg = saved_dbg;
        for (int a = 0; a < N_STATE; ++a)
            HIPCHK(copy_async(h->d_tmpl[a], h->d_state[a], (size_t)h->state_len[a] * (h->rr ? kStateSpgRr[a] : 1) * sizeof(float),
                                  hipMemcpyDeviceToDevice, h->stream));
        HIPCHK(hipMemsetAsync(h->d_mel, 0, SP * 8 * h->kmax * 32 * sizeof(float), h->stream));
        HIPCHK(hipMemsetAsync(h->d_emb, 0, SP * 96 * sizeof(float), h->stream));
        if (int rc = do_reset(h, nullptr, (int)SP, nullptr)) return rc;
        HIPCHK(hipStreamSynchronize(h->stream));
        clk.lap("warm-up + reset");
        // the warm-up already drove the network with an all-ones mel history: weights that overflow the f16 range there are refused now
        if (int rc = range_check(h, "oww_commit")) return rc;
        // f16-split family: replay the calibration probes and hold the result to the exact-fp32 run (refuses weights the split loses)
        if (h->hx && !getenv("OWW_NO_COMMIT_SELFTEST")) {
            if (int rc = selftest_hx(h, cal)) return rc;
            HIPCHK(hipMemsetAsync(h->d_mel, 0, SP * 8 * h->kmax * 32 * sizeof(float), h->stream));
            HIPCHK(hipMemsetAsync(h->d_emb, 0, SP * 96 * sizeof(float), h->stream));
            HIPCHK(hipMemsetAsync(h->d_raw, 0, SP * std::max(h->NL, 1) * sizeof(float), h->stream));
            if (int rc = do_reset(h, nullptr, (int)SP, nullptr)) return rc;
            HIPCHK(hipStreamSynchronize(h->stream));
            clk.lap("self-test replay + reset");
        }
    }
    h->committed = true;
    return OWW_OK;
    OWW_GUARD_END
}

int oww_reset(oww_ctx* h, const int32_t* stream_ids, int32_t n, const float* init_features) {
    OWW_GUARD_BEGIN
    if (!h || !h->committed) return fail(OWW_ESTATE, "oww_reset: handle not committed");
    HIPCHK(hipSetDevice(h->cfg.device));
    const float* d_init = nullptr;
    if (init_features) {
        HIPCHK(copy_async(h->d_featinit, init_features, (size_t)h->TR * 96 * sizeof(float), hipMemcpyHostToDevice, h->stream));
        d_init = h->d_featinit;
    }
    if (!stream_ids) {
        if (int rc = do_reset(h, nullptr, h->Spad, d_init)) return rc;
    } else {
        if (n < 1) return OWW_OK;
        for (int i = 0; i < n; ++i)
            if (stream_ids[i] < 0 || stream_ids[i] >= h->S) return fail(OWW_EINVAL, "oww_reset: stream id %d out of range", stream_ids[i]);
        if (n > h->ids_cap) {
            if (h->d_ids) (void)dev_free(h->d_ids);
            h->d_ids = nullptr; h->ids_cap = 0;
            HIPCHK(dev_alloc(&h->d_ids, (size_t)n * sizeof(int)));
            h->ids_cap = n;
        }
        HIPCHK(copy_async(h->d_ids, stream_ids, (size_t)n * sizeof(int), hipMemcpyHostToDevice, h->stream));
        if (int rc = do_reset(h, h->d_ids, n, d_init)) return rc;
    }
    HIPCHK(hipStreamSynchronize(h->stream));     // host buffers may be reused by the caller
    return OWW_OK;
    OWW_GUARD_END
}

int oww_set_postproc(oww_ctx* h, const int32_t* patience, const float* threshold, int32_t debounce_frames) {
    OWW_GUARD_BEGIN
    if (!h || !h->committed) return fail(OWW_ESTATE, "oww_set_postproc: handle not committed");
    HIPCHK(hipSetDevice(h->cfg.device));
    std::vector<int> pat(std::max(h->NL, 1), 0);
    std::vector<float> thr(std::max(h->NL, 1), NAN);
    if (patience) for (int i = 0; i < h->NL; ++i) pat[i] = patience[i];
    if (threshold) for (int i = 0; i < h->NL; ++i) thr[i] = threshold[i];
    HIPCHK(copy_async(h->d_patience, pat.data(), pat.size() * sizeof(int), hipMemcpyHostToDevice, h->stream));
    HIPCHK(copy_async(h->d_threshold, thr.data(), thr.size() * sizeof(float), hipMemcpyHostToDevice, h->stream));
    HIPCHK(hipStreamSynchronize(h->stream));
    h->debounce_frames = debounce_frames;
    if (h->graph_exec) { (void)hipGraphExecDestroy(h->graph_exec); h->graph_exec = nullptr; }   // baked-in scalar changed
    return OWW_OK;
    OWW_GUARD_END
}

int oww_step(oww_ctx* h, const int16_t* pcm, int pcm_on_device, int32_t n_chunks, float* scores, int scores_on_device) {
    OWW_GUARD_BEGIN
    if (!h || !h->committed) return fail(OWW_ESTATE, "oww_step: handle not committed");
    if (!pcm) return fail(OWW_EINVAL, "oww_step: pcm is null");
    if (n_chunks < 1 || n_chunks > OWW_MAX_CALL_CHUNKS) return fail(OWW_EINVAL, "oww_step: n_chunks=%d outside [1,%d]", n_chunks, OWW_MAX_CALL_CHUNKS);
    if (int rc = range_check(h, "oww_step")) return rc;          // raised by an earlier (asynchronous) step: sticky
    HIPCHK(hipSetDevice(h->cfg.device));
    h->k_last = n_chunks;
    const size_t n_pcm = (size_t)h->S * OWW_CHUNK * n_chunks;
    if (n_chunks > h->kmax) {                                    // longer than the mel buffer: slices that share the call's clamp floor
        const int16_t* d_call = pcm;
        if (!pcm_on_device) {
            if (n_pcm > h->long_cap) {
                if (h->d_long) { HIPCHK(hipStreamSynchronize(h->stream)); (void)dev_free(h->d_long); h->d_long = nullptr; h->long_cap = 0; }
                if (dev_alloc(&h->d_long, n_pcm * sizeof(int16_t)) != hipSuccess) return fail(OWW_ENOMEM, "oww_step: out of device memory for a call of %d chunks", n_chunks);
                h->long_cap = n_pcm;
            }
            HIPCHK(copy_async(h->d_long, pcm, n_pcm * sizeof(int16_t), hipMemcpyHostToDevice, h->stream));
            d_call = h->d_long;
        }
        if (int rc = launch_step_long(h, d_call, n_chunks)) return rc;
        if (scores) {
            const size_t nb = (size_t)h->S * h->NL * sizeof(float);
            if (nb) HIPCHK(copy_async(scores, h->d_scores, nb, scores_on_device ? hipMemcpyDeviceToDevice : hipMemcpyDeviceToHost, h->stream));
        }
        if (!scores_on_device || !pcm_on_device) {               // (the caller's host buffer is free to change when this returns)
            HIPCHK(hipStreamSynchronize(h->stream));
            if (int rc = range_check(h, "oww_step")) return rc;
        }
        return OWW_OK;
    }
    const bool graphable = h->want_graph && n_chunks == 1 && !h->timing;
    const int16_t* d_pcm = pcm;
    if (!pcm_on_device || graphable) {
        HIPCHK(copy_async(h->d_pcm, pcm, n_pcm * sizeof(int16_t), pcm_on_device ? hipMemcpyDeviceToDevice : hipMemcpyHostToDevice, h->stream));
        d_pcm = h->d_pcm;
    }
    if (graphable) {
        if (!h->graph_exec) {
            HIPCHK(hipStreamBeginCapture(h->stream, hipStreamCaptureModeThreadLocal));
            const int rc = launch_step(h, h->d_pcm, 1);
            hipGraph_t g = nullptr;
            const hipError_t e = hipStreamEndCapture(h->stream, &g);
            if (rc) return rc;
            if (e != hipSuccess) return fail(OWW_EHIP, "hipStreamEndCapture: %s", hipGetErrorString(e));
            if (h->graph) (void)hipGraphDestroy(h->graph);
            h->graph = g;
            HIPCHK(hipGraphInstantiate(&h->graph_exec, h->graph, nullptr, nullptr, 0));
        }
        HIPCHK(hipGraphLaunch(h->graph_exec, h->stream));
    } else {
        if (int rc = launch_step(h, d_pcm, n_chunks)) return rc;
    }
    if (scores) {
        const size_t nb = (size_t)h->S * h->NL * sizeof(float);
        if (nb) HIPCHK(copy_async(scores, h->d_scores, nb, scores_on_device ? hipMemcpyDeviceToDevice : hipMemcpyDeviceToHost, h->stream));
        if (!scores_on_device) {
            HIPCHK(hipStreamSynchronize(h->stream));
            if (int rc = range_check(h, "oww_step")) return rc;
        }
    }
    return OWW_OK;
    OWW_GUARD_END
}

int oww_step_masked(oww_ctx* h, const int16_t* pcm, int pcm_on_device, const uint8_t* stream_on, int stream_on_on_device,
                    float* scores, int scores_on_device) {
    OWW_GUARD_BEGIN
    if (!h || !h->committed) return fail(OWW_ESTATE, "oww_step_masked: handle not committed");
    if (!pcm || !stream_on) return fail(OWW_EINVAL, "oww_step_masked: null argument");
    if (!h->rr) return fail(OWW_EINVAL, "oww_step_masked: needs the register-resident kernel families (use_mfma = 3 or 1)");
    if (int rc = range_check(h, "oww_step_masked")) return rc;
    HIPCHK(hipSetDevice(h->cfg.device));
    h->k_last = 1;
    if (!h->d_on) {
        HIPCHK(dev_alloc(&h->d_on, h->Spad));
        HIPCHK(hipMemsetAsync(h->d_on, 0, h->Spad, h->stream));
    }
    // (always through the handle's own buffer: the kernels index it up to the padded stream count)
    HIPCHK(copy_async(h->d_on, stream_on, h->S, stream_on_on_device ? hipMemcpyDeviceToDevice : hipMemcpyHostToDevice, h->stream));
    const int16_t* d_pcm = pcm;
    if (!pcm_on_device || (reinterpret_cast<uintptr_t>(pcm) & 15)) {
        HIPCHK(copy_async(h->d_pcm, pcm, (size_t)h->S * OWW_CHUNK * sizeof(int16_t),
                              pcm_on_device ? hipMemcpyDeviceToDevice : hipMemcpyHostToDevice, h->stream));
        d_pcm = h->d_pcm;
    }
    h->on_now = h->d_on;
    int n_act = -1;
    if (!stream_on_on_device) {                                 // few participants: only their groups are launched (build_active_lists)
        n_act = build_active_lists(h, stream_on);
        if (n_act < -1) { h->on_now = nullptr; return n_act + 100; }
    }
    h->lists_now = n_act >= 0;
    const int rc = n_act == 0 ? 0 : launch_step(h, d_pcm, 1);     // nobody takes part: nothing moves
    h->on_now = nullptr; h->lists_now = false;
    if (rc) return rc;
    if (scores) {
        const size_t nb = (size_t)h->S * h->NL * sizeof(float);
        if (nb) HIPCHK(copy_async(scores, h->d_scores, nb, scores_on_device ? hipMemcpyDeviceToDevice : hipMemcpyDeviceToHost, h->stream));
        if (!scores_on_device) {
            HIPCHK(hipStreamSynchronize(h->stream));
            if (int rc2 = range_check(h, "oww_step_masked")) return rc2;
        }
    }
    return OWW_OK;
    OWW_GUARD_END
}

static int ensure_ingest(oww_ctx* h) {
    if (h->up_stream) return 0;
    HIPCHK(hipStreamCreateWithFlags(&h->up_stream, hipStreamNonBlocking));
    HIPCHK(hipStreamCreateWithFlags(&h->down_stream, hipStreamNonBlocking));
    const size_t nb = (size_t)h->S * std::max(h->NL, 1) * sizeof(float);
    for (auto& sl : h->slot) {
        HIPCHK(dev_alloc(&sl.d_pcm, (size_t)h->S * OWW_CHUNK * h->kmax * sizeof(int16_t)));
        HIPCHK(dev_alloc(&sl.d_scores, nb));
        HIPCHK(hipHostMalloc((void**)&sl.h_scores, nb, hipHostMallocDefault));
        HIPCHK(hipHostMalloc((void**)&sl.h_on, h->S, hipHostMallocDefault));
        HIPCHK(hipEventCreateWithFlags(&sl.up, hipEventDisableTiming));
        HIPCHK(hipEventCreateWithFlags(&sl.done, hipEventDisableTiming));
        HIPCHK(hipEventCreateWithFlags(&sl.down, hipEventDisableTiming));
    }
    return 0;
}

static int submit_impl(oww_ctx* h, const int16_t* pcm, int32_t n_chunks, const uint8_t* stream_on);

int oww_submit(oww_ctx* h, const int16_t* pcm, int32_t n_chunks) { OWW_GUARD_BEGIN return submit_impl(h, pcm, n_chunks, nullptr); OWW_GUARD_END }

int oww_submit_masked(oww_ctx* h, const int16_t* pcm, const uint8_t* stream_on) {
    OWW_GUARD_BEGIN
    if (!stream_on) return fail(OWW_EINVAL, "oww_submit_masked: stream_on is null");
    if (h && h->committed && !h->rr)
        return fail(OWW_EINVAL, "oww_submit_masked: needs the register-resident kernel families (use_mfma = 3 or 1; see oww_step_masked)");
    return submit_impl(h, pcm, 1, stream_on);
    OWW_GUARD_END
}

static int submit_impl(oww_ctx* h, const int16_t* pcm, int32_t n_chunks, const uint8_t* stream_on) {
    if (!h || !h->committed) return fail(OWW_ESTATE, "oww_submit: handle not committed");
    if (!pcm) return fail(OWW_EINVAL, "oww_submit: pcm is null");
    if (n_chunks < 1 || n_chunks > h->kmax) return fail(OWW_EINVAL, "oww_submit: n_chunks=%d outside [1,%d]", n_chunks, h->kmax);
    if (int rc = range_check(h, "oww_submit")) return rc;
    HIPCHK(hipSetDevice(h->cfg.device));
    h->k_last = n_chunks;
    if (int rc = ensure_ingest(h)) return rc;
    auto& sl = h->slot[h->n_submit & 1];
    if (sl.busy) return fail(OWW_ESTATE, "oww_submit: two steps already in flight, call oww_collect first");
    const size_t n_pcm = (size_t)h->S * OWW_CHUNK * n_chunks;
    HIPCHK(copy_async(sl.d_pcm, pcm, n_pcm * sizeof(int16_t), hipMemcpyHostToDevice, h->up_stream));
    HIPCHK(hipEventRecord(sl.up, h->up_stream));
    HIPCHK(hipStreamWaitEvent(h->stream, sl.up, 0));
    if (stream_on) {                                   // (the mask is small: copied on the compute stream, ordered before this step's kernels)
        if (!h->d_on) {
            HIPCHK(dev_alloc(&h->d_on, h->Spad));
            HIPCHK(hipMemsetAsync(h->d_on, 0, h->Spad, h->stream));
        }
        memcpy(sl.h_on, stream_on, h->S);              // the caller's array is free again when this call returns
        HIPCHK(copy_async(h->d_on, sl.h_on, h->S, hipMemcpyHostToDevice, h->stream));
        h->on_now = h->d_on;
    }
    int n_act = -1;
    if (stream_on) {
        n_act = build_active_lists(h, stream_on);
        if (n_act < -1) { h->on_now = nullptr; return n_act + 100; }
    }
    h->lists_now = n_act >= 0;
    const int rc_step = n_act == 0 ? 0 : launch_step(h, sl.d_pcm, n_chunks);
    h->on_now = nullptr; h->lists_now = false;
    if (rc_step) return rc_step;
    const size_t nb = (size_t)h->S * h->NL * sizeof(float);
    if (nb) HIPCHK(copy_async(sl.d_scores, h->d_scores, nb, hipMemcpyDeviceToDevice, h->stream));   // d_scores is rewritten by the next step
    HIPCHK(hipEventRecord(sl.done, h->stream));
    HIPCHK(hipStreamWaitEvent(h->down_stream, sl.done, 0));
    if (nb) HIPCHK(copy_async(sl.h_scores, sl.d_scores, nb, hipMemcpyDeviceToHost, h->down_stream));
    HIPCHK(hipEventRecord(sl.down, h->down_stream));
    sl.busy = true;
    ++h->n_submit;
    return OWW_OK;
}

int oww_collect(oww_ctx* h, float* scores) {
    OWW_GUARD_BEGIN
    if (!h || !h->committed) return fail(OWW_ESTATE, "oww_collect: handle not committed");
    auto& sl = h->slot[h->n_collect & 1];
    if (h->n_collect == h->n_submit || !sl.busy) return fail(OWW_ESTATE, "oww_collect: no step in flight");
    HIPCHK(hipSetDevice(h->cfg.device));
    HIPCHK(hipEventSynchronize(sl.down));
    if (scores) memcpy(scores, sl.h_scores, (size_t)h->S * h->NL * sizeof(float));
    sl.busy = false;
    ++h->n_collect;
    if (int rc = range_check(h, "oww_collect")) return rc;      // the step is consumed; its scores are suspect
    return OWW_OK;
    OWW_GUARD_END
}

int oww_host_alloc(void** out, size_t nbytes) {
    OWW_GUARD_BEGIN
    if (!out || !nbytes) return fail(OWW_EINVAL, "oww_host_alloc: bad argument");
    if (hipHostMalloc(out, nbytes, hipHostMallocDefault) != hipSuccess) { *out = nullptr; return fail(OWW_ENOMEM, "oww_host_alloc: %zu bytes of page-locked memory not available", nbytes); }
    return OWW_OK;
    OWW_GUARD_END
}

int oww_host_free(void* p) {
    OWW_GUARD_BEGIN
    if (p) HIPCHK(hipHostFree(p));
    return OWW_OK;
    OWW_GUARD_END
}

int oww_set_verifier(oww_ctx* h, int32_t label, const float* w, int32_t n_w, float bias, float threshold) {
    OWW_GUARD_BEGIN
    if (!h || !h->committed) return fail(OWW_ESTATE, "oww_set_verifier: handle not committed");
    if (label < 0 || label >= h->NL) return fail(OWW_EINVAL, "oww_set_verifier: label %d outside [0,%d)", label, h->NL);
    HIPCHK(hipSetDevice(h->cfg.device));
    int T = 0;                                         // feature rows of the model that owns this label
    for (const auto& hh : h->heads) if (label >= hh.out_col && label < hh.out_col + hh.n_out) T = hh.T;
    if (w && n_w != T * OWW_EMB_DIM) return fail(OWW_EINVAL, "oww_set_verifier: %d weights given, the label's model has %d x 96 = %d features", n_w, T, T * OWW_EMB_DIM);
    if (!h->d_verw) {
        int maxT = 1;
        for (const auto& hh : h->heads) maxT = std::max(maxT, hh.T);
        h->ver_stride = maxT * OWW_EMB_DIM;
        if (int rc = dalloc(h->stream, &h->d_verw, (size_t)h->NL * h->ver_stride)) return rc;
        if (int rc = dalloc(h->stream, &h->d_verb, (size_t)h->NL)) return rc;
        if (int rc = dalloc(h->stream, &h->d_verthr, (size_t)h->NL)) return rc;
        if (int rc = dalloc(h->stream, &h->d_verT, (size_t)h->NL)) return rc;
        h->ver_T.assign(h->NL, 0);
    }
    HIPCHK(hipStreamSynchronize(h->stream));
    if (w) {
        HIPCHK(copy_sync(h->d_verw + (size_t)label * h->ver_stride, w, (size_t)n_w * sizeof(float), hipMemcpyHostToDevice));
        HIPCHK(copy_sync(h->d_verb + label, &bias, sizeof(float), hipMemcpyHostToDevice));
        HIPCHK(copy_sync(h->d_verthr + label, &threshold, sizeof(float), hipMemcpyHostToDevice));
    }
    h->ver_T[label] = w ? T : 0;
    HIPCHK(copy_sync(h->d_verT, h->ver_T.data(), (size_t)h->NL * sizeof(int), hipMemcpyHostToDevice));
    h->n_verifiers = 0;
    for (int t : h->ver_T) h->n_verifiers += t > 0;
    if (h->graph_exec) { (void)hipGraphExecDestroy(h->graph_exec); h->graph_exec = nullptr; }       // the launch list changed
    return OWW_OK;
    OWW_GUARD_END
}

int oww_set_vad_threshold(oww_ctx* h, float threshold) {
    OWW_GUARD_BEGIN
    if (!h || !h->committed) return fail(OWW_ESTATE, "oww_set_vad_threshold: handle not committed");
    if (!(threshold == threshold)) return fail(OWW_EINVAL, "oww_set_vad_threshold: NaN");
    h->vad_threshold = threshold;
    if (h->graph_exec) { (void)hipGraphExecDestroy(h->graph_exec); h->graph_exec = nullptr; }     // the threshold is a kernel argument
    return OWW_OK;
    OWW_GUARD_END
}

int oww_push_vad(oww_ctx* h, const float* vad_scores, int on_device) {
    OWW_GUARD_BEGIN
    if (!h || !h->committed) return fail(OWW_ESTATE, "oww_push_vad: handle not committed");
    if (!vad_scores) return fail(OWW_EINVAL, "oww_push_vad: null argument");
    if (h->vad) return fail(OWW_ESTATE, "oww_push_vad: this handle computes its own voice-activity scores (oww_load_vad)");
    HIPCHK(hipSetDevice(h->cfg.device));
    const float* src = vad_scores;
    if (!on_device) {
        HIPCHK(copy_async(h->d_vadin, vad_scores, (size_t)h->S * sizeof(float), hipMemcpyHostToDevice, h->stream));
        src = h->d_vadin;
    }
    hipLaunchKernelGGL(push_vad_kernel, dim3((h->S + 255) / 256), dim3(256), 0, h->stream, h->d_vadring, h->d_nvad, src, h->S);
    HIPCHK(hipGetLastError());
    if (!on_device) HIPCHK(hipStreamSynchronize(h->stream));          // the caller's buffer may be reused at once
    return OWW_OK;
    OWW_GUARD_END
}

int oww_get_vad(oww_ctx* h, float* out) {
    OWW_GUARD_BEGIN
    if (!h || !h->committed) return fail(OWW_ESTATE, "oww_get_vad: handle not committed");
    if (!h->vad) return fail(OWW_ESTATE, "oww_get_vad: no voice-activity network loaded (oww_load_vad)");
    if (!out) return fail(OWW_EINVAL, "oww_get_vad: null argument");
    HIPCHK(hipSetDevice(h->cfg.device));
    HIPCHK(copy_async(out, h->d_vadlast, (size_t)h->S * sizeof(float), hipMemcpyDeviceToHost, h->stream));
    HIPCHK(hipStreamSynchronize(h->stream));
    return OWW_OK;
    OWW_GUARD_END
}

int oww_reset_vad(oww_ctx* h, const int32_t* stream_ids, int32_t n) {
    OWW_GUARD_BEGIN
    if (!h || !h->committed) return fail(OWW_ESTATE, "oww_reset_vad: handle not committed");
    HIPCHK(hipSetDevice(h->cfg.device));
    const int* d_ids = nullptr;
    int count = h->S;
    if (stream_ids) {
        if (n < 1) return OWW_OK;
        for (int i = 0; i < n; ++i)
            if (stream_ids[i] < 0 || stream_ids[i] >= h->S) return fail(OWW_EINVAL, "oww_reset_vad: stream id %d out of range", stream_ids[i]);
        if (n > h->ids_cap) {
            if (h->d_ids) (void)dev_free(h->d_ids);
            h->d_ids = nullptr; h->ids_cap = 0;
            HIPCHK(dev_alloc(&h->d_ids, (size_t)n * sizeof(int)));
            h->ids_cap = n;
        }
        HIPCHK(copy_async(h->d_ids, stream_ids, (size_t)n * sizeof(int), hipMemcpyHostToDevice, h->stream));
        d_ids = h->d_ids; count = n;
    }
    if (h->vad) {
        hipLaunchKernelGGL(owv::vad_reset_kernel, dim3(count), dim3(64), 0, h->stream, h->d_vadhc, h->d_vadring, h->d_nvad, h->d_vadlast, d_ids, count);
    } else {
        hipLaunchKernelGGL(vad_ring_reset_kernel, dim3((count + 255) / 256), dim3(256), 0, h->stream, h->d_vadring, h->d_nvad, d_ids, count);
    }
    HIPCHK(hipGetLastError());
    HIPCHK(hipStreamSynchronize(h->stream));
    return OWW_OK;
    OWW_GUARD_END
}

int oww_sync(oww_ctx* h) {
    OWW_GUARD_BEGIN
    if (!h) return fail(OWW_EINVAL, "null handle");
    HIPCHK(hipSetDevice(h->cfg.device));
    HIPCHK(hipStreamSynchronize(h->stream));
    return range_check(h, "oww_sync");
    OWW_GUARD_END
}

int oww_range_status(oww_ctx* h, int clear) {
    OWW_GUARD_BEGIN
    if (!h || !h->committed) return fail(OWW_ESTATE, "oww_range_status: handle not committed");
    HIPCHK(hipSetDevice(h->cfg.device));
    HIPCHK(hipStreamSynchronize(h->stream));
    const int rc = range_check(h, "oww_range_status");
    if (clear && h->h_range) { volatile int* f = (volatile int*)h->h_range; f[1] = -1; f[0] = 0; }     // the position goes with the flag
    return rc;
    OWW_GUARD_END
}

int oww_range_where(oww_ctx* h, int32_t* first_stream, int32_t* n_streams) {
    OWW_GUARD_BEGIN
    if (!h || !h->committed || !first_stream || !n_streams) return fail(OWW_EINVAL, "oww_range_where: bad argument");
    HIPCHK(hipSetDevice(h->cfg.device));
    HIPCHK(hipStreamSynchronize(h->stream));
    const volatile int* f = (const volatile int*)h->h_range;
    *first_stream = -1; *n_streams = 0;
    if (f && f[0]) {
        const int packed = f[1];                            // first << 6 | count, written by one wave in one store (owwhip_hx.h)
        if (packed >= 0) {
            const int first = (int)((unsigned)packed >> 6), cnt = packed & 63;
            if (first < h->S && cnt > 0) { *first_stream = first; *n_streams = std::min(cnt, h->S - first); }
        }
    }
    return OWW_OK;
    OWW_GUARD_END
}

const float* oww_scores_dev(const oww_ctx* h) { return h ? h->d_scores : nullptr; }

int oww_resample(oww_ctx* h, const int16_t* in, int in_on_device, int32_t n_in, int32_t p, int32_t q, const float* taps, int32_t n_taps,
                 int16_t* out, int out_on_device, int32_t n_out) {
    OWW_GUARD_BEGIN
    if (!h || !h->committed) return fail(OWW_ESTATE, "oww_resample: handle not committed");
    if (!in || !out || !taps) return fail(OWW_EINVAL, "oww_resample: null argument");
    if (n_in < 1 || p < 1 || q < 1 || n_taps < 2 || (n_taps & 1) || n_taps > 4096 || q > 65536)
        return fail(OWW_EINVAL, "oww_resample: bad argument (n_in=%d p=%d q=%d n_taps=%d)", n_in, p, q, n_taps);
    if (n_out != (int32_t)(((long long)n_in * q) / p) || n_out < 1)
        return fail(OWW_EINVAL, "oww_resample: n_out must be n_in * q / p = %lld, got %d", ((long long)n_in * q) / p, n_out);
    HIPCHK(hipSetDevice(h->cfg.device));
    const int ntp = (n_taps + 3) / 4 * 4;
    // outputs per workgroup: as many as a 48 KB input span allows, at most a chunk's worth (the filter bank is staged once per workgroup)
    int opb = RS_NT * 5;
    auto span_of = [&](int o) { return (int)(((long long)(o - 1) * p) / q) + 1 + n_taps + 4; };   // (+ 4: the zero-padded taps read 3 words on)
    while (opb > RS_NT && (size_t)span_of(opb) * sizeof(float) > 48 * 1024) opb -= RS_NT;
    opb = std::min(opb, (n_out + RS_NT - 1) / RS_NT * RS_NT);
    const int span = span_of(opb);
    const size_t lds_x = (size_t)((span + 3) / 4 * 4) * sizeof(float), lds_t = (size_t)q * ntp * sizeof(float);
    if (lds_x > 96 * 1024) return fail(OWW_EINVAL, "oww_resample: input rate too high for the staging buffer (p / q = %d / %d)", p, q);
    const int taps_in_lds = lds_x + lds_t <= 150 * 1024;
    const size_t lds = lds_x + (taps_in_lds ? lds_t : 0);
    HIPCHK(hipFuncSetAttribute(reinterpret_cast<const void*>(resample_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    const size_t b_taps = ((size_t)q * ntp * sizeof(float) + 255) / 256 * 256;
    const size_t b_in = in_on_device ? 0 : ((size_t)h->S * n_in * sizeof(int16_t) + 255) / 256 * 256;
    const size_t b_out = out_on_device ? 0 : (size_t)h->S * n_out * sizeof(int16_t);
    if (b_taps + b_in + b_out > h->rs_bytes) {
        HIPCHK(hipStreamSynchronize(h->stream));
        if (h->d_rs) (void)dev_free(h->d_rs);
        h->d_rs = nullptr; h->rs_bytes = 0;
        if (dev_alloc(&h->d_rs, b_taps + b_in + b_out) != hipSuccess) return fail(OWW_ENOMEM, "oww_resample: out of device memory");
        h->rs_bytes = b_taps + b_in + b_out;
    }
    char* base = (char*)h->d_rs;
    // (stream-ordered upload from a buffer that lives in the handle: an earlier oww_resample may still be reading the old bank)
    HIPCHK(hipStreamSynchronize(h->stream));
    h->rs_taps.assign((size_t)q * ntp, 0.f);
    for (int r = 0; r < q; ++r) memcpy(&h->rs_taps[(size_t)r * ntp], taps + (size_t)r * n_taps, n_taps * sizeof(float));
    HIPCHK(copy_async(base, h->rs_taps.data(), h->rs_taps.size() * sizeof(float), hipMemcpyHostToDevice, h->stream));
    ResampleParams a{};
    a.taps = (const float*)base; a.n_in = n_in; a.n_out = n_out; a.p = p; a.q = q; a.n_taps = n_taps; a.ntp = ntp; a.S = h->S;
    a.span = span; a.taps_in_lds = taps_in_lds; a.opb = opb;
    a.in = in;
    if (!in_on_device) {
        HIPCHK(copy_async(base + b_taps, in, (size_t)h->S * n_in * sizeof(int16_t), hipMemcpyHostToDevice, h->stream));
        a.in = (const int16_t*)(base + b_taps);
    }
    a.out = out_on_device ? out : (int16_t*)(base + b_taps + b_in);
    hipLaunchKernelGGL(resample_kernel, dim3((n_out + opb - 1) / opb, h->S), dim3(RS_NT), lds, h->stream, a);
    HIPCHK(hipGetLastError());
    if (!out_on_device) {
        HIPCHK(copy_async(out, a.out, (size_t)h->S * n_out * sizeof(int16_t), hipMemcpyDeviceToHost, h->stream));
        HIPCHK(hipStreamSynchronize(h->stream));
    }
    return OWW_OK;
    OWW_GUARD_END
}

int oww_get_raw(oww_ctx* h, float* out) {
    OWW_GUARD_BEGIN
    if (!h || !h->committed) return fail(OWW_ESTATE, "oww_get_raw: handle not committed");
    if (!out) return fail(OWW_EINVAL, "oww_get_raw: null argument");
    HIPCHK(hipSetDevice(h->cfg.device));
    const size_t nb = (size_t)h->S * h->NL * sizeof(float);
    if (nb) HIPCHK(copy_async(out, h->d_raw, nb, hipMemcpyDeviceToHost, h->stream));
    HIPCHK(hipStreamSynchronize(h->stream));
    return OWW_OK;
    OWW_GUARD_END
}

static int mel_impl(oww_ctx* h, const int16_t* pcm, int32_t B, int32_t n, float* out_db, bool per_clip) {
    if (!h || !h->committed) return fail(OWW_ESTATE, "oww_mel: handle not committed");
    if (!pcm || !out_db || B < 1 || n < 512) return fail(OWW_EINVAL, "oww_mel: bad argument (B=%d n=%d)", B, n);
    HIPCHK(hipSetDevice(h->cfg.device));
    const int F = (n - 512) / 160 + 1;
    int16_t* d_in = nullptr; float* d_out = nullptr; float* d_max = nullptr;
    int rc = 0;
    do {
        if (dev_alloc(&d_in, (size_t)B * n * sizeof(int16_t)) != hipSuccess || dev_alloc(&d_out, (size_t)B * F * 32 * sizeof(float)) != hipSuccess ||
            dev_alloc(&d_max, (size_t)B * sizeof(float)) != hipSuccess) { rc = fail(OWW_ENOMEM, "oww_mel: out of device memory"); break; }
        if (copy_async(d_in, pcm, (size_t)B * n * sizeof(int16_t), hipMemcpyHostToDevice, h->stream) != hipSuccess) { rc = fail(OWW_EHIP, "oww_mel: H2D failed"); break; }
        if ((rc = launch_mel(h, d_in, B, n, F, 0, d_out, d_max))) break;
        const size_t tot = (size_t)B * F * 32;
        if (per_clip) {
            hipLaunchKernelGGL(clamp_db_rows_kernel, dim3((F * 32 + 255) / 256, B), dim3(256), 0, h->stream, d_out, F * 32, d_max);
        } else {
            std::vector<float> mx(B);
            if (copy_async(mx.data(), d_max, B * sizeof(float), hipMemcpyDeviceToHost, h->stream) != hipSuccess ||
                hipStreamSynchronize(h->stream) != hipSuccess) { rc = fail(OWW_EHIP, "oww_mel: D2H failed"); break; }
            float gmax = -INFINITY;                       // one clamp floor for the whole call (ipynb cell 15: log_spec.max())
            for (float v : mx) gmax = std::max(gmax, v);
            hipLaunchKernelGGL(clamp_db_kernel, dim3((unsigned)((tot + 255) / 256)), dim3(256), 0, h->stream, d_out, tot, gmax - 80.0f);
        }
        if (copy_async(out_db, d_out, tot * sizeof(float), hipMemcpyDeviceToHost, h->stream) != hipSuccess ||
            hipStreamSynchronize(h->stream) != hipSuccess) { rc = fail(OWW_EHIP, "oww_mel: D2H failed"); break; }
    } while (0);
    (void)dev_free(d_in); (void)dev_free(d_out); (void)dev_free(d_max);
    return rc;
}

int oww_mel(oww_ctx* h, const int16_t* pcm, int32_t B, int32_t n, float* out_db) { OWW_GUARD_BEGIN return mel_impl(h, pcm, B, n, out_db, false); OWW_GUARD_END }
int oww_mel_clips(oww_ctx* h, const int16_t* pcm, int32_t B, int32_t n, float* out_db) { OWW_GUARD_BEGIN return mel_impl(h, pcm, B, n, out_db, true); OWW_GUARD_END }

int oww_embed(oww_ctx* h, const float* mel_rows, int32_t B, int32_t rows, float* out) {
    OWW_GUARD_BEGIN
    if (!h || !h->committed) return fail(OWW_ESTATE, "oww_embed: handle not committed");
    if (!mel_rows || !out || B < 1 || B > h->Spad || rows < 76 || (rows - 76) % 8) return fail(OWW_EINVAL, "oww_embed: bad argument (B=%d rows=%d, need B<=%d)", B, rows, h->Spad);
    HIPCHK(hipSetDevice(h->cfg.device));
    const int n_out = (rows - 76) / 8 + 1;
    const int n_steps = (rows + 4) / 8;                 // 4 lead-in rows + rows, 8 per step
    std::vector<float> slab((size_t)B * 256), emb((size_t)B * 96);
    if (int rc = park_state(h, B, true)) return rc;
    int rc_all = 0;
    for (int it = 0; it < n_steps && !rc_all; ++it) {
        for (int b = 0; b < B; ++b)
            for (int r = 0; r < 8; ++r) {
                const int src = it * 8 + r - 4;
                float* d = &slab[((size_t)b * 8 + r) * 32];
                if (src < 0) memset(d, 0, 32 * sizeof(float));
                else memcpy(d, mel_rows + ((size_t)b * rows + src) * 32, 32 * sizeof(float));
            }
        if (copy_async(h->d_mel, slab.data(), slab.size() * sizeof(float), hipMemcpyHostToDevice, h->stream) != hipSuccess) { rc_all = fail(OWW_EHIP, "oww_embed: H2D failed"); break; }
        if ((rc_all = run_cnn(h, B, 256, 0))) break;
        hipLaunchKernelGGL(advance_kernel, dim3((h->Spad + 255) / 256), dim3(256), 0, h->stream, h->d_nfeat, h->Spad, (const uint8_t*)nullptr);
        if (it >= 9 && copy_async(emb.data(), h->d_emb, emb.size() * sizeof(float), hipMemcpyDeviceToHost, h->stream) != hipSuccess) { rc_all = fail(OWW_EHIP, "oww_embed: D2H failed"); break; }
        if (hipStreamSynchronize(h->stream) != hipSuccess) { rc_all = fail(OWW_EHIP, "oww_embed: device error"); break; }   // slab is reused next iteration
        if (it >= 9)
            for (int b = 0; b < B; ++b) memcpy(out + ((size_t)b * n_out + (it - 9)) * 96, &emb[(size_t)b * 96], 96 * sizeof(float));
    }
    const int rc_restore = park_state(h, B, false);
    (void)hipStreamSynchronize(h->stream);
    if (rc_all) return rc_all;
    if (rc_restore) return rc_restore;
    return range_check(h, "oww_embed");
    OWW_GUARD_END
}

int oww_embed_clips(oww_ctx* h, const int16_t* pcm, int32_t pcm_on_device, int32_t B, int32_t n, float* out, int32_t out_on_device) {
    OWW_GUARD_BEGIN
    if (!h || !h->committed) return fail(OWW_ESTATE, "oww_embed_clips: handle not committed");
    if (!pcm || !out || B < 1 || B > h->Spad || n < 512) return fail(OWW_EINVAL, "oww_embed_clips: bad argument (B=%d n=%d, need B<=%d)", B, n, h->Spad);
    const int F = (n - 512) / 160 + 1;
    if (F < 76) return fail(OWW_EINVAL, "oww_embed_clips: %d samples give %d mel frames, one embedding window needs 76", n, F);
    if ((int64_t)F * 32 > INT32_MAX / 2) return fail(OWW_EINVAL, "oww_embed_clips: clip too long");
    HIPCHK(hipSetDevice(h->cfg.device));
    const int n_out = (F - 76) / 8 + 1;
    const int n_steps = 9 + n_out;                      // step it consumes mel rows [8 it - 4, 8 it + 4) of every clip
    const int Bp = std::min(h->Spad, (B + 7) / 8 * 8);   // run_cnn covers whole groups of 8 streams: their (zero) mel rows must exist
    const size_t lead = 4 * 32;                         // four lead-in rows in front of clip 0 (other clips: the previous clip's tail;
                                                        // their content never reaches a returned embedding)
    int16_t* d_in = nullptr; float* d_mel = nullptr; float* d_max = nullptr; float* d_out = nullptr;
    int rc = park_state(h, B, true);
    if (rc) return rc;
    do {
        if (!pcm_on_device) {
            if (dev_alloc(&d_in, (size_t)B * n * sizeof(int16_t)) != hipSuccess) { rc = fail(OWW_ENOMEM, "oww_embed_clips: out of device memory"); break; }
            if (copy_async(d_in, pcm, (size_t)B * n * sizeof(int16_t), hipMemcpyHostToDevice, h->stream) != hipSuccess) { rc = fail(OWW_EHIP, "oww_embed_clips: H2D failed"); break; }
        }
        if (dev_alloc(&d_mel, (lead + (size_t)Bp * F * 32) * sizeof(float)) != hipSuccess ||
            dev_alloc(&d_max, (size_t)B * sizeof(float)) != hipSuccess ||
            (!out_on_device && dev_alloc(&d_out, (size_t)B * n_out * 96 * sizeof(float)) != hipSuccess)) { rc = fail(OWW_ENOMEM, "oww_embed_clips: out of device memory"); break; }
        float* o = out_on_device ? out : d_out;
        if (hipMemsetAsync(d_mel, 0, lead * sizeof(float), h->stream) != hipSuccess ||
            (Bp > B && hipMemsetAsync(d_mel + lead + (size_t)B * F * 32, 0, (size_t)(Bp - B) * F * 32 * sizeof(float), h->stream) != hipSuccess)) { rc = fail(OWW_EHIP, "oww_embed_clips: memset failed"); break; }
        if ((rc = launch_mel(h, pcm_on_device ? pcm : d_in, B, n, F, 0, d_mel + lead, d_max))) break;
        hipLaunchKernelGGL(clamp_transform_rows_kernel, dim3((F * 32 + 255) / 256, B), dim3(256), 0, h->stream, d_mel + lead, F * 32, d_max);
        h->mel_src = d_mel + lead;
        for (int it = 0; it < n_steps && !rc; ++it) {
            rc = run_cnn(h, B, F * 32, (it * 8 - 4) * 32);
            if (rc) break;
            hipLaunchKernelGGL(advance_kernel, dim3((h->Spad + 255) / 256), dim3(256), 0, h->stream, h->d_nfeat, h->Spad, (const uint8_t*)nullptr);
            if (it >= 9 && hipMemcpy2DAsync(o + (size_t)(it - 9) * 96, (size_t)n_out * 96 * sizeof(float), h->d_emb, 96 * sizeof(float),
                                            96 * sizeof(float), B, hipMemcpyDeviceToDevice, h->stream) != hipSuccess)
                rc = fail(OWW_EHIP, "oww_embed_clips: gather failed");
        }
        h->mel_src = nullptr;
        if (rc) break;
        if (!out_on_device && copy_async(out, d_out, (size_t)B * n_out * 96 * sizeof(float), hipMemcpyDeviceToHost, h->stream) != hipSuccess) { rc = fail(OWW_EHIP, "oww_embed_clips: D2H failed"); break; }
        if (hipStreamSynchronize(h->stream) != hipSuccess) { rc = fail(OWW_EHIP, "oww_embed_clips: device error: %s", hipGetErrorString(hipGetLastError())); break; }
    } while (0);
    h->mel_src = nullptr;
    const int rc_restore = park_state(h, B, false);
    (void)hipStreamSynchronize(h->stream);
    (void)dev_free(d_in); (void)dev_free(d_mel); (void)dev_free(d_max); (void)dev_free(d_out);
    if (!rc) rc = rc_restore;
    if (!rc) rc = range_check(h, "oww_embed_clips");
    return rc;
    OWW_GUARD_END
}

int oww_head(oww_ctx* h, int32_t head, const float* features, int32_t B, float* out) {
    OWW_GUARD_BEGIN
    if (!h || !h->committed) return fail(OWW_ESTATE, "oww_head: handle not committed");
    if (head < 0 || head >= (int)h->heads.size() || !features || !out || B < 1) return fail(OWW_EINVAL, "oww_head: bad argument");
    HIPCHK(hipSetDevice(h->cfg.device));
    const HeadHost& hh = h->heads[head];
    const size_t nf = (size_t)B * hh.T * 96;
    float* d_f = nullptr; float* d_raw = nullptr;
    int rc = 0;
    do {
        if (dev_alloc(&d_f, nf * sizeof(float)) != hipSuccess || dev_alloc(&d_raw, (size_t)B * h->NL * sizeof(float)) != hipSuccess) {
            rc = fail(OWW_ENOMEM, "oww_head: out of device memory"); break;
        }
        if (copy_async(d_f, features, nf * sizeof(float), hipMemcpyHostToDevice, h->stream) != hipSuccess) { rc = fail(OWW_EHIP, "oww_head: H2D failed"); break; }
        if (hipMemsetAsync(d_raw, 0, (size_t)B * h->NL * sizeof(float), h->stream) != hipSuccess) { rc = fail(OWW_EHIP, "oww_head: memset failed"); break; }
        if ((rc = run_heads(h, B, false, d_f, head, d_raw, 0))) break;
        std::vector<float> raw((size_t)B * h->NL);
        if (copy_async(raw.data(), d_raw, raw.size() * sizeof(float), hipMemcpyDeviceToHost, h->stream) != hipSuccess ||
            hipStreamSynchronize(h->stream) != hipSuccess) { rc = fail(OWW_EHIP, "oww_head: D2H failed"); break; }
        for (int b = 0; b < B; ++b)
            for (int o = 0; o < hh.n_out; ++o) out[(size_t)b * hh.n_out + o] = raw[(size_t)b * h->NL + hh.out_col + o];
        rc = range_check(h, "oww_head");
    } while (0);
    (void)dev_free(d_f); (void)dev_free(d_raw);
    return rc;
    OWW_GUARD_END
}

int oww_get_features(oww_ctx* h, int32_t sid, int32_t T, float* out) {
    OWW_GUARD_BEGIN
    if (!h || !h->committed) return fail(OWW_ESTATE, "oww_get_features: handle not committed");
    if (sid < 0 || sid >= h->S || T < 1 || T > h->TR || !out) return fail(OWW_EINVAL, "oww_get_features: bad argument (sid=%d T=%d ring=%d)", sid, T, h->TR);
    HIPCHK(hipSetDevice(h->cfg.device));
    std::vector<float> ring((size_t)h->TR * 96);
    uint32_t cnt = 0;
    HIPCHK(copy_async(ring.data(), h->d_feat + (size_t)sid * h->TR * 96, ring.size() * sizeof(float), hipMemcpyDeviceToHost, h->stream));
    HIPCHK(copy_async(&cnt, h->d_nfeat + sid, sizeof cnt, hipMemcpyDeviceToHost, h->stream));
    HIPCHK(hipStreamSynchronize(h->stream));
    // after the step's advance the newest row sits at slot (cnt-1) % TR; oldest-first order of the last T rows
    for (int t = 0; t < T; ++t) {
        const uint32_t slot = (cnt + (uint32_t)(2 * h->TR - T + t)) % (uint32_t)h->TR;
        memcpy(out + (size_t)t * 96, &ring[(size_t)slot * 96], 96 * sizeof(float));
    }
    return OWW_OK;
    OWW_GUARD_END
}

int oww_get_mel(oww_ctx* h, int32_t sid, float* out, int32_t n_rows) {
    OWW_GUARD_BEGIN
    if (!h || !h->committed) return fail(OWW_ESTATE, "oww_get_mel: handle not committed");
    if (h->fuse && h->k_last == 1 && !h->cfg.debug_layers)
        return fail(OWW_ESTATE, "oww_get_mel: with the mel front end fused into stage A the rows of a one-chunk step never reach HBM; "
                    "create the handle with debug_layers = 1 to keep them");
    const int rows_last = 8 * h->k_last;           // the last step wrote [S][8 * n_chunks][32]
    if (sid < 0 || sid >= h->S || !out || n_rows < 1 || n_rows > rows_last)
        return fail(OWW_EINVAL, "oww_get_mel: bad argument (sid=%d n_rows=%d; the last step produced %d rows per stream)", sid, n_rows, rows_last);
    HIPCHK(hipSetDevice(h->cfg.device));
    HIPCHK(copy_async(out, h->d_mel + ((size_t)sid * rows_last + (rows_last - n_rows)) * 32, (size_t)n_rows * 32 * sizeof(float), hipMemcpyDeviceToHost, h->stream));
    HIPCHK(hipStreamSynchronize(h->stream));
    return OWW_OK;
    OWW_GUARD_END
}

int oww_debug_read(oww_ctx* h, int32_t sid, int32_t layer, float* out, int32_t cap) {
    OWW_GUARD_BEGIN
    if (!h || !h->committed || !h->d_dbg) return fail(OWW_ESTATE, "oww_debug_read: needs a committed handle created with debug_layers=1");
    if (sid < 0 || sid >= h->S || layer < 0 || layer > 19 || !out) return fail(OWW_EINVAL, "oww_debug_read: bad argument");
    int off = 0;
    for (int l = 0; l < layer; ++l) off += kLayerOut[l][0] * kLayerOut[l][1] * kLayerOut[l][2];
    const int n = kLayerOut[layer][0] * kLayerOut[layer][1] * kLayerOut[layer][2];
    if (cap < n) return fail(OWW_EINVAL, "oww_debug_read: need room for %d floats", n);
    HIPCHK(hipSetDevice(h->cfg.device));
    HIPCHK(copy_async(out, h->d_dbg + (size_t)sid * DBG_FLOATS + off, (size_t)n * sizeof(float), hipMemcpyDeviceToHost, h->stream));
    HIPCHK(hipStreamSynchronize(h->stream));
    return n;
    OWW_GUARD_END
}

int oww_debug_profile(oww_ctx* h, int64_t* out, int32_t cap) {
    OWW_GUARD_BEGIN
    if (!h || !h->committed || !h->d_prof) return fail(OWW_ESTATE, "oww_debug_profile: set OWW_PROF_BLOCK before creating the handle");
    if (!out || cap < 4 * 256) return fail(OWW_EINVAL, "oww_debug_profile: need room for 1024 values");
    HIPCHK(hipSetDevice(h->cfg.device));
    HIPCHK(hipStreamSynchronize(h->stream));
    HIPCHK(copy_sync(out, h->d_prof, (size_t)4 * 256 * sizeof(long long), hipMemcpyDeviceToHost));
    return 4 * 256;
    OWW_GUARD_END
}

int oww_enable_timing(oww_ctx* h, int on) {
    OWW_GUARD_BEGIN
    if (!h) return fail(OWW_EINVAL, "null handle");
    if (!on && h->timing) { if (int rc = flush_events(h)) return rc; }
    h->timing = on != 0;
    return OWW_OK;
    OWW_GUARD_END
}

int oww_kernel_times(oww_ctx* h, double ms[OWW_N_KERNEL_CLASSES], int64_t n[OWW_N_KERNEL_CLASSES]) {
    OWW_GUARD_BEGIN
    if (!h || !ms || !n) return fail(OWW_EINVAL, "null argument");
    HIPCHK(hipSetDevice(h->cfg.device));
    if (int rc = flush_events(h)) return rc;
    for (int i = 0; i < OWW_N_KERNEL_CLASSES; ++i) { ms[i] = h->t_ms[i]; n[i] = h->t_n[i]; h->t_ms[i] = 0; h->t_n[i] = 0; }
    return OWW_OK;
    OWW_GUARD_END
}

int oww_use_graph(oww_ctx* h, int on) {
    OWW_GUARD_BEGIN
    if (!h) return fail(OWW_EINVAL, "null handle");
    h->want_graph = on != 0;
    if (!on && h->graph_exec) { (void)hipGraphExecDestroy(h->graph_exec); h->graph_exec = nullptr; }
    return OWW_OK;
    OWW_GUARD_END
}

// ---- multi-GPU delivery of results over RCCL, without torch.distributed -----------------------------------------------------------
int oww_comm_id(void* id) {
    OWW_GUARD_BEGIN
    if (!id) return fail(OWW_EINVAL, "oww_comm_id: null argument");
    if (int rc = rccl_load()) return rc;
    RCCLCHK(g_rccl.GetUniqueId(id));
    return OWW_OK;
    OWW_GUARD_END
}

int oww_comm_init(oww_ctx* h, const void* id, int32_t rank, int32_t world) {
    OWW_GUARD_BEGIN
    if (!h || !h->committed) return fail(OWW_ESTATE, "oww_comm_init: handle not committed");
    if (!id || world < 1 || rank < 0 || rank >= world) return fail(OWW_EINVAL, "oww_comm_init: bad argument (rank %d of %d)", rank, world);
    if (h->comm) return fail(OWW_ESTATE, "oww_comm_init: the handle already has a communicator");
    if (int rc = rccl_load()) return rc;
    HIPCHK(hipSetDevice(h->cfg.device));
    RcclId uid;
    memcpy(uid.b, id, sizeof uid.b);
    void* comm = nullptr;
    RCCLCHK(g_rccl.CommInitRank(&comm, world, uid, rank));
    h->comm = comm; h->comm_rank = rank; h->comm_world = world;
    return OWW_OK;
    OWW_GUARD_END
}

int oww_gather_scores(oww_ctx* h, float* out, const int32_t* counts) {
    OWW_GUARD_BEGIN
    if (!h || !h->committed) return fail(OWW_ESTATE, "oww_gather_scores: handle not committed");
    if (!h->comm) return fail(OWW_ESTATE, "oww_gather_scores: call oww_comm_init first");
    if (!counts) return fail(OWW_EINVAL, "oww_gather_scores: counts is null");
    if (counts[h->comm_rank] != h->S) return fail(OWW_EINVAL, "oww_gather_scores: counts[%d] = %d, this handle owns %d streams", h->comm_rank, counts[h->comm_rank], h->S);
    if (h->comm_rank == 0 && !out) return fail(OWW_EINVAL, "oww_gather_scores: rank 0 needs the output buffer");
    HIPCHK(hipSetDevice(h->cfg.device));
    const size_t NL = (size_t)h->NL;
    if (NL == 0) return OWW_OK;
    // one grouped exchange on the handle's stream (ordered after the step that produced d_scores): every rank sends its [S_r][NL]
    // block to rank 0 -- rank 0 to itself as well, so the path is the same RCCL kernel at any world size
    RCCLCHK(g_rccl.GroupStart());
    int rc_send = g_rccl.Send(h->d_scores, (size_t)h->S * NL, 7 /* ncclFloat32 */, 0, h->comm, h->stream);
    int rc_recv = 0;
    if (h->comm_rank == 0) {
        size_t off = 0;
        for (int r = 0; r < h->comm_world && !rc_recv; ++r) {
            if (counts[r] < 0) { rc_recv = -1; break; }
            if (counts[r] > 0) rc_recv = g_rccl.Recv(out + off, (size_t)counts[r] * NL, 7, r, h->comm, h->stream);
            off += (size_t)counts[r] * NL;
        }
    }
    const int rc_end = g_rccl.GroupEnd();
    if (rc_send) RCCLCHK(rc_send);
    if (rc_recv < 0) return fail(OWW_EINVAL, "oww_gather_scores: negative count");
    if (rc_recv) RCCLCHK(rc_recv);
    RCCLCHK(rc_end);
    return OWW_OK;
    OWW_GUARD_END
}

int oww_comm_count(oww_ctx* h, int32_t* ranks) {
    OWW_GUARD_BEGIN
    if (!h || !ranks) return fail(OWW_EINVAL, "oww_comm_count: null argument");
    if (!h->comm) return fail(OWW_ESTATE, "oww_comm_count: call oww_comm_init first");
    int n = 0;
    RCCLCHK(g_rccl.CommCount(h->comm, &n));
    *ranks = n;
    return OWW_OK;
    OWW_GUARD_END
}

int oww_comm_destroy(oww_ctx* h) {
    OWW_GUARD_BEGIN
    if (!h) return OWW_OK;
    if (h->comm) { (void)hipSetDevice(h->cfg.device); (void)hipStreamSynchronize(h->stream); }
    comm_release(h);
    return OWW_OK;
    OWW_GUARD_END
}

}  // extern "C"
