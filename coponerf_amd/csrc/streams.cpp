// HIP streams restricted to a share of the compute units (CU masks), and the CU count a launch on a stream may use.
//
// The reference's evaluation loop is serial — get_z, then the chunked forward() calls, pair after pair
// (/root/reference test.py:164-212, wrapper.py:176-211).  On an MI355X the two halves are complementary: the render
// pass is a handful of HBM-bound launches whose persistent grids take every CU (one workgroup per CU, most of its LDS),
// get_z is ~550 small launches that rarely fill a quarter of the chip.  Issued on two ordinary streams they alternate
// instead of overlapping: each small kernel waits until a chip-filling one drains.  With two CU-masked streams the
// chip is PARTITIONED — the render pass keeps e.g. 24 CUs of every XCD, get_z the other 8 — and both run all the time.
//
// KFD spreads the bits of a CU mask round-robin over the XCDs (bit k -> XCD k % 8), then over that XCD's 4 shader
// engines, so a contiguous bit range [first, first + n) with first and n multiples of 32 is an equal share of every
// shader engine of every XCD: the workgroup -> XCD round-robin that the XCD-aware tile walks of the GEMM / encoder rely
// on is unchanged, and a persistent grid of one workgroup per CU still lands one workgroup on each CU.  (A share of
// 208 = 26 per XCD leaves the shader engines 7/7/6/6 CUs: the dispatcher hands every engine the same number of
// workgroups, two of them meet on one CU, cannot co-reside (LDS) and the render pass takes 37 ms instead of 27.)
//
// Persistent kernels size their grid by the CU count: cpn_stream_cus(stream) is what their launchers use — the device's
// count for ordinary streams, the share for streams created here.
#include <hip/hip_runtime.h>

#include <mutex>
#include <vector>

#include "common.h"

namespace {
struct Masked {
    hipStream_t stream;
    int cus;
};
std::mutex g_mu;
std::vector<Masked> g_masked;

int device_cus() {
    static int num_cu = 0;
    if (num_cu == 0) {
        int dev = 0, n = 0;
        if (hipGetDevice(&dev) != hipSuccess ||
            hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n <= 0)
            n = 256;
        num_cu = n;
    }
    return num_cu;
}
}  // namespace

int cpn_stream_cus(void* stream) {
    if (stream) {
        std::lock_guard<std::mutex> lk(g_mu);
        for (const Masked& m : g_masked)
            if ((void*)m.stream == stream) return m.cus;
    }
    return device_cus();
}

extern "C" int cpn_device_cu_count(void) { return device_cus(); }

extern "C" int cpn_stream_cu_count(void* stream) { return cpn_stream_cus(stream); }

extern "C" int cpn_stream_create_cu_range(int first_cu, int num_cus, void** stream_out) {
    CPN_REQUIRE(stream_out, CPN_E_ARG, "cpn_stream_create_cu_range: null pointer");
    const int total = device_cus();
    CPN_REQUIRE(first_cu >= 0 && num_cus > 0 && first_cu + num_cus <= total, CPN_E_ARG,
                "cpn_stream_create_cu_range: CUs [%d, %d) outside the device's %d", first_cu, first_cu + num_cus, total);
    CPN_REQUIRE((first_cu % 32) == 0 && (num_cus % 32) == 0, CPN_E_ARG,
                "cpn_stream_create_cu_range: first_cu and num_cus must be multiples of 32 (an equal share of each of the "
                "4 shader engines of each of the 8 XCDs)");
    std::vector<uint32_t> mask((size_t)(total + 31) / 32, 0u);
    for (int k = first_cu; k < first_cu + num_cus; ++k) mask[(size_t)k >> 5] |= 1u << (k & 31);
    hipStream_t s = nullptr;
    const hipError_t e = hipExtStreamCreateWithCUMask(&s, (uint32_t)mask.size(), mask.data());
    if (e != hipSuccess) {
        cpn_set_error("cpn_stream_create_cu_range: hipExtStreamCreateWithCUMask failed: %s", hipGetErrorString(e));
        return (int)e;
    }
    {
        std::lock_guard<std::mutex> lk(g_mu);
        g_masked.push_back({s, num_cus});
    }
    *stream_out = (void*)s;
    return 0;
}

extern "C" int cpn_stream_destroy(void* stream) {
    CPN_REQUIRE(stream, CPN_E_ARG, "cpn_stream_destroy: null stream");
    bool ours = false;
    {
        std::lock_guard<std::mutex> lk(g_mu);
        for (size_t i = 0; i < g_masked.size(); ++i)
            if ((void*)g_masked[i].stream == stream) {
                g_masked.erase(g_masked.begin() + (long)i);
                ours = true;
                break;
            }
    }
    CPN_REQUIRE(ours, CPN_E_ARG, "cpn_stream_destroy: not a stream of cpn_stream_create_cu_range");
    const hipError_t e = hipStreamDestroy((hipStream_t)stream);
    if (e != hipSuccess) {
        cpn_set_error("cpn_stream_destroy: %s", hipGetErrorString(e));
        return (int)e;
    }
    return 0;
}
