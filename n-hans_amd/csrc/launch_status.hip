// Launch-error channel of libnhans_hip.so (declared in nhans_kernels.h): the launchers are plain
// void functions deep inside the launch sequences; instead of threading a status through every one
// of them, each records the outcome of its launch here and the C-ABI entry point collects it.
#include "nhans_kernels.h"

#include <atomic>
#include <cstdlib>

namespace nhans {

namespace {
thread_local hipError_t g_first_err = hipSuccess;
thread_local const char* g_first_where = "";

void keep(hipError_t e, const char* where) {
    // (hipErrorNotReady is what an event/stream QUERY of the application leaves behind: not a failure)
    if (e != hipSuccess && e != hipErrorNotReady && g_first_err == hipSuccess) {
        g_first_err = e;
        g_first_where = where;
    }
}
}  // namespace

void note_launch(const char* kernel, hipError_t launch_rc) {
    if (launch_rc == hipSuccess) return;
    (void)hipGetLastError();        // (a failed launch of OURS also sets the sticky error: take it back out, it is reported by return code)
    keep(launch_rc, kernel);
}

// A launch the LIBRARY refuses (arguments it will not run a kernel on): no HIP call failed, so the runtime's sticky
// per-thread error -- which may hold something the application left there -- is neither read nor cleared.
void note_refusal(const char* what) { keep(hipErrorInvalidValue, what); }

bool launch_error_pending() { return g_first_err != hipSuccess; }

void set_max_dynamic_lds(const void* fn, size_t bytes, unsigned long long* done_mask, const char* kernel) {
    int dev = 0;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess) { keep(e, kernel); return; }
    auto* mask = reinterpret_cast<std::atomic<unsigned long long>*>(done_mask);
    const unsigned long long bit = 1ull << (dev & 63);
    if (dev < 64 && (mask->load(std::memory_order_acquire) & bit)) return;
    e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
    if (e != hipSuccess) { (void)hipGetLastError(); keep(e, kernel); return; }
    if (dev < 64) mask->fetch_or(bit, std::memory_order_release);
}

hipError_t take_launch_error(const char** kernel) {
    const hipError_t e = g_first_err;
    if (kernel) *kernel = g_first_where;
    g_first_err = hipSuccess;
    g_first_where = "";
    return e;
}

#ifdef NHANS_DEV
int dev_ablate() {
    static const int v = [] { const char* e = getenv("NHANS_ABLATE"); return e ? atoi(e) : 0; }();
    return v;
}
#endif

}  // namespace nhans
