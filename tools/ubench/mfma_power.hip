// Dev micro-benchmark: what the f16 matrix pipes of ONE MI355X sustain at the socket power cap.  Every wave
// issues back-to-back independent v_mfma_f32_32x32x16_f16 on register operands (no LDS, no memory) for several
// seconds; the host prints TFLOP/s per launch, and tools/clock_power_trace.sh beside it shows clock and power.
//   hipcc --offload-arch=gfx950 -O3 mfma_power.hip -o mfma_power && ./mfma_power [waves per CU: 4|8|12] [data] [seconds] [shape]
//   data: 0 zeros, 1 random, 2 random with half of the activation-side operand's elements zero (post-ReLU-like);
//   shape: 0 = v_mfma_f32_32x32x16_f16 (what the conv kernels issue), 1 = v_mfma_f32_16x16x32_f16
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

typedef float f32x4 __attribute__((ext_vector_type(4)));
__global__ void __launch_bounds__(256) k16(float* out, const _Float16* src, int iters) {
    f16x8 a[4], b[4];
    for (int i = 0; i < 4; ++i) {
        a[i] = *reinterpret_cast<const f16x8*>(src + ((threadIdx.x * 8 + i * 2048) & 16383));
        b[i] = *reinterpret_cast<const f16x8*>(src + 16384 + ((threadIdx.x * 8 + i * 2048 + 1024) & 16383));
    }
    f32x4 acc[8];
    for (int i = 0; i < 8; ++i) for (int r = 0; r < 4; ++r) acc[i][r] = 0.f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 4; ++u) {
#pragma unroll
            for (int i = 0; i < 8; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a[(i + u) & 3], b[i & 3], acc[i], 0, 0, 0);
        }
    }
    float s = 0;
    for (int i = 0; i < 8; ++i) for (int r = 0; r < 4; ++r) s += acc[i][r];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}

__global__ void __launch_bounds__(256) k(float* out, const _Float16* src, int iters) {
    f16x8 a[4], b[4];
    for (int i = 0; i < 4; ++i) {
        a[i] = *reinterpret_cast<const f16x8*>(src + ((threadIdx.x * 8 + i * 2048) & 16383));
        b[i] = *reinterpret_cast<const f16x8*>(src + 16384 + ((threadIdx.x * 8 + i * 2048 + 1024) & 16383));
    }
    f32x16 acc[4];
    for (int i = 0; i < 4; ++i) for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 4; ++u) {
#pragma unroll
            for (int i = 0; i < 4; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[(i + u) & 3], b[i], acc[i], 0, 0, 0);
        }
    }
    float s = 0;
    for (int i = 0; i < 4; ++i) for (int r = 0; r < 16; ++r) s += acc[i][r];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}

int main(int argc, char** argv) {
    const int wpc = argc > 1 ? atoi(argv[1]) : 8, rnd = argc > 2 ? atoi(argv[2]) : 1;
    const double secs = argc > 3 ? atof(argv[3]) : 6.0;
    const int shape = argc > 4 ? atoi(argv[4]) : 0;
    const int blocks = 256 * wpc / 4, iters = 20000;
    float* out; _Float16* src;
    (void)hipMalloc(&out, blocks * 256 * 4); (void)hipMalloc(&src, 32768 * 2);
    static _Float16 h[32768];                        // [0, 16384): activation-side operand, [16384, 32768): weight side
    unsigned x = 12345;
    for (int i = 0; i < 32768; ++i) {
        x = x * 1664525u + 1013904223u;
        h[i] = rnd ? (_Float16)(((int)(x >> 16) % 2001 - 1000) * 1e-3f) : (_Float16)0.f;
        if (rnd == 2 && i < 16384 && (x >> 9 & 1)) h[i] = (_Float16)0.f;
    }
    (void)hipMemcpy(src, h, sizeof h, hipMemcpyHostToDevice);
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    const double flop = shape ? (double)blocks * 4 * iters * 32 * 2.0 * 16 * 16 * 32 : (double)blocks * 4 * iters * 16 * 2.0 * 32 * 32 * 16;
    double t = 0; int n = 0;
    printf("%d waves per CU, %s operands, %s, %d MFMAs per wave per launch\n", wpc, rnd == 2 ? "random (activation side half zeros)" : rnd ? "random" : "zero",
           shape ? "16x16x32" : "32x32x16", iters * (shape ? 32 : 16));
    while (t < secs * 1e3) {
        (void)hipEventRecord(e0, 0);
        if (shape) hipLaunchKernelGGL(k16, dim3(blocks), dim3(256), 0, 0, out, src, iters);
        else hipLaunchKernelGGL(k, dim3(blocks), dim3(256), 0, 0, out, src, iters);
        (void)hipEventRecord(e1, 0); (void)hipEventSynchronize(e1);
        float ms; (void)hipEventElapsedTime(&ms, e0, e1);
        t += ms;
        if (n++ % 8 == 0) printf("  t = %6.0f ms: %7.1f TFLOP/s\n", t, flop / ms / 1e9);
    }
    return 0;
}
