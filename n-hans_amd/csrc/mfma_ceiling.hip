// nhans_debug_mfma_ceiling: what the f16 matrix pipes of THIS device sustain at its socket power cap.
//
// The conv kernels of the split-f16 path are priced against the 2.5 PFLOP/s dense f16 MFMA peak of the
// data sheet, which the chip reaches only on operands that do not toggle the multipliers: on data-like
// operands it sits at its power cap at a clock far below 2.4 GHz ("The power wall").
// This entry point measures that ceiling where the claim is made -- on the box and in the process that
// reports the roofline: every wave issues back-to-back independent v_mfma_f32_32x32x16_f16 (the shape the
// conv kernels issue) on pseudo-random register operands, no LDS, no memory, launch after launch for the
// requested time; the sustained rate is the mean over the second half of the launches (the first ones run
// at the boost clock before the power controller has settled).
#include "../../include/nhans_hip.h"
#include "nhans_kernels.h"

#include <vector>

namespace nhans {
namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

constexpr int kCeilIters = 20000, kCeilWavesPerCU = 8, kCeilCUs = 256;

__global__ void __launch_bounds__(256) mfma_ceiling_kernel(float* out, const _Float16* src, int iters) {
    f16x8 a[4], b[4];
    for (int i = 0; i < 4; ++i) {
        a[i] = *reinterpret_cast<const f16x8*>(src + ((threadIdx.x * 8 + i * 2048) & 16383));
        b[i] = *reinterpret_cast<const f16x8*>(src + 16384 + ((threadIdx.x * 8 + i * 2048 + 1024) & 16383));
    }
    f32x16 acc[4];
    for (int i = 0; i < 4; ++i)
        for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 4; ++u) {
#pragma unroll
            for (int i = 0; i < 4; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[(i + u) & 3], b[i], acc[i], 0, 0, 0);
        }
    }
    float s = 0;
    for (int i = 0; i < 4; ++i)
        for (int r = 0; r < 16; ++r) s += acc[i][r];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}

}  // namespace
}  // namespace nhans

extern "C" int nhans_debug_mfma_ceiling(double seconds, void* stream, double* sustained_tflops, double* first_tflops,
                                        int* launches_out) {
    using namespace nhans;
    if (!sustained_tflops || !(seconds > 0) || seconds > 60) return NHANS_EINVAL;
    hipStream_t s = static_cast<hipStream_t>(stream);
    const int blocks = kCeilCUs * kCeilWavesPerCU / 4;
    float* out = nullptr;
    _Float16* src = nullptr;
    if (hipMalloc(reinterpret_cast<void**>(&out), (size_t)blocks * 256 * 4) != hipSuccess) return NHANS_ENOMEM;
    if (hipMalloc(reinterpret_cast<void**>(&src), 32768 * 2) != hipSuccess) { (void)hipFree(out); return NHANS_ENOMEM; }
    std::vector<_Float16> h(32768);
    unsigned x = 12345;
    for (int i = 0; i < 32768; ++i) {       // uniform in [-1, 1], like mantissas of normalised data
        x = x * 1664525u + 1013904223u;
        h[i] = (_Float16)(((int)(x >> 16) % 2001 - 1000) * 1e-3f);
    }
    int rc = NHANS_OK;
    hipEvent_t e0 = nullptr, e1 = nullptr;
    if (hipMemcpy(src, h.data(), 32768 * 2, hipMemcpyHostToDevice) != hipSuccess || hipEventCreate(&e0) != hipSuccess ||
        hipEventCreate(&e1) != hipSuccess)
        rc = NHANS_EHIP;
    const double flop = (double)blocks * 4 * kCeilIters * 16 * 2.0 * 32 * 32 * 16;
    std::vector<double> rate;
    double t = 0;
    (void)take_launch_error(nullptr);
    while (rc == NHANS_OK && t < seconds * 1e3) {
        (void)hipEventRecord(e0, s);
        NHANS_LAUNCH("mfma_ceiling", mfma_ceiling_kernel, dim3(blocks), dim3(256), 0, s, out, src, kCeilIters);
        (void)hipEventRecord(e1, s);
        float ms = 0.f;
        if (take_launch_error(nullptr) != hipSuccess || hipEventSynchronize(e1) != hipSuccess ||
            hipEventElapsedTime(&ms, e0, e1) != hipSuccess || !(ms > 0.f)) { rc = NHANS_EHIP; break; }
        t += ms;
        rate.push_back(flop / ms / 1e9);
    }
    if (rc == NHANS_OK && !rate.empty()) {
        double sum = 0;
        const size_t from = rate.size() / 2;
        for (size_t i = from; i < rate.size(); ++i) sum += rate[i];
        *sustained_tflops = sum / (double)(rate.size() - from);
        if (first_tflops) *first_tflops = rate[0];
        if (launches_out) *launches_out = (int)rate.size();
    }
    if (e0) (void)hipEventDestroy(e0);
    if (e1) (void)hipEventDestroy(e1);
    (void)hipFree(src);
    (void)hipFree(out);
    return rc;
}
