// Dev micro-benchmark: issue rate of v_mfma_f32_32x32x16_f16 with NACC independent accumulators per
// wave and WAVES waves per SIMD.   hipcc --offload-arch=gfx950 -O3 mfma_rate.hip -o mfma_rate && ./mfma_rate
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

template <int NACC>
__global__ void __launch_bounds__(1024) k(float* out, int iters, long long* cyc) {
    f32x16 acc[NACC];
    for (int i = 0; i < NACC; ++i) for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
    f16x8 a, b;
    for (int e = 0; e < 8; ++e) { a[e] = (_Float16)(threadIdx.x * 0.001f + e); b[e] = (_Float16)(e * 0.5f); }
    const long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int rep = 0; rep < 12 / NACC; ++rep)
#pragma unroll
            for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc[i], 0, 0, 0);
    }
    const long long t1 = __builtin_amdgcn_s_memtime();
    float s = 0;
    for (int i = 0; i < NACC; ++i) for (int r = 0; r < 16; ++r) s += acc[i][r];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

template <int NACC> void run(int waves_per_simd) {
    const int threads = waves_per_simd * 4 * 64, blocks = 256, iters = 2000;
    float* out; long long* cyc;
    hipMalloc(&out, blocks * threads * 4); hipMalloc(&cyc, blocks * 8);
    hipLaunchKernelGGL(k<NACC>, dim3(blocks), dim3(threads), 0, 0, out, iters, cyc);
    hipDeviceSynchronize();
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0); hipLaunchKernelGGL(k<NACC>, dim3(blocks), dim3(threads), 0, 0, out, iters, cyc); hipEventRecord(e1);
    hipDeviceSynchronize();
    float ms; hipEventElapsedTime(&ms, e0, e1);
    long long h[256]; hipMemcpy(h, cyc, sizeof h, hipMemcpyDeviceToHost);
    double c = 0; for (int i = 0; i < 256; ++i) c += h[i]; c /= 256;
    const double mf = (double)iters * 12;    // MFMAs per wave
    printf("NACC %d, %d waves/SIMD: %.1f cycles per MFMA per wave, %.1f cycles per MFMA per SIMD, %.0f TFLOP/s, %.2f GHz\n", NACC,
           waves_per_simd, c / mf, c / (mf * waves_per_simd), 2.0 * 32 * 32 * 16 * mf * waves_per_simd * 4 * blocks / (ms * 1e-3) / 1e12,
           c / (ms * 1e-3) / 1e9);
    hipFree(out); hipFree(cyc);
}

int main() {
    for (int w = 1; w <= 2; ++w) { run<1>(w); run<2>(w); run<3>(w); run<4>(w); run<6>(w); }
    return 0;
}
