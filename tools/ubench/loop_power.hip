// Dev micro-benchmark: the shipped consumer/producer loop (8 MFMA waves + 4 DMA waves, one workgroup per CU, as
// fat_loop.hip) SUSTAINED for seconds at the socket power cap, as a function of the LDS-DMA volume per tap and its
// source -- what a byte of L2->LDS or HBM->LDS traffic costs in throughput once the chip is power-limited (the
// s_memtime ticks of fat_loop.hip say what it costs in stalls; this says what it costs in wall time).
//   hipcc --offload-arch=gfx950 -O3 loop_power.hip -o loop_power && ./loop_power [seconds per point]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

template <int I, int NM, int ND> __device__ __forceinline__ void pin() {
    if constexpr (I < ND) {
        __builtin_amdgcn_sched_group_barrier(0x008, (I + 1) * NM / ND - I * NM / ND, 0);
        __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
        pin<I + 1, NM, ND>();
    }
}

// NDMA: global_load_lds instructions (1 KB each) per producer wave per tap; SHARED: all workgroups stream the same
// 64 KB window (L2-resident, like the weights) instead of a private window per wave (HBM, like the activations)
template <int NDMA, int SHARED>
__global__ void __launch_bounds__(768) k(float* out, const float* src, int taps) {
    constexpr int TM = 2, TN = 2;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    // operand data: random-looking f16 pairs (the power draw of an MFMA depends on its operands)
    for (int i = tid; i < 36864; i += 768) { unsigned x = (i + 1) * 2654435761u; smem[i] = __uint_as_float(0x38003800u ^ (x & 0x03ff03ffu)); }
    __syncthreads();
    if (wave >= 8) {
        __builtin_amdgcn_s_setprio(3);
        const int pw = wave - 8;
        const float* p = SHARED ? src + pw * 4096 + lane * 4 : src + (1 << 20) + ((size_t)blockIdx.x * 4 + pw) * (1 << 18) + lane * 4;
        for (int it = 0; it < taps; ++it) {
#pragma unroll
            for (int d = 0; d < NDMA; ++d)
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(p + ((it * NDMA + d) & (SHARED ? 15 : 1023)) * 256),
                                                 (__attribute__((address_space(3))) void*)(smem + 20480 + ((it & 3) * 4096) + (pw * NDMA + d) % 16 * 256),
                                                 16, 0, 0);
            if constexpr (NDMA > 0) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NDMA * 2) : "memory");
            __builtin_amdgcn_s_barrier();
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        return;
    }
    const int wm = wave >> 1, wn = wave & 1, g8 = lane >> 5;
    int jb[TM];
    for (int t = 0; t < TM; ++t) jb[t] = (wm * TM + t) * 32 % 256 + (lane & 31);
    const int bcol = 20480 + (wn * TN) * 1024 + lane * 4;
    f32x16 acc[TM][TN];
    for (int t = 0; t < TM; ++t) for (int j = 0; j < TN; ++j) for (int r = 0; r < 16; ++r) acc[t][j][r] = 0.f;
    f32x4 fa_hi[2][TM], fa_lo[2][TM], fb_hi[2][TN], fb_lo[2][TN];
#define RD(H, KW, STG)                                                                             \
    {                                                                                              \
        _Pragma("unroll") for (int t = 0; t < TM; ++t) {                                           \
            const int jr_ = jb[t] + (KW);                                                          \
            const float* ar_ = smem + ((STG) & 1) * 10240 + jr_ * 32;                              \
            const int rs_ = (jr_ >> 1) & 7;                                                        \
            fa_hi[H][t] = *reinterpret_cast<const f32x4*>(ar_ + (((2 * (H) + g8) ^ rs_) * 4));     \
            fa_lo[H][t] = *reinterpret_cast<const f32x4*>(ar_ + (((2 * (H) + g8 + 4) ^ rs_) * 4)); \
        }                                                                                          \
        _Pragma("unroll") for (int j = 0; j < TN; ++j) {                                           \
            fb_hi[H][j] = *reinterpret_cast<const f32x4*>(smem + bcol + ((STG) & 3) * 4096 + j * 1024 + (H) * 512);       \
            fb_lo[H][j] = *reinterpret_cast<const f32x4*>(smem + bcol + ((STG) & 3) * 4096 + j * 1024 + (H) * 512 + 256); \
        }                                                                                          \
    }
#define MM(H)                                                                                      \
    _Pragma("unroll") for (int p = 0; p < 3; ++p)                                                  \
        _Pragma("unroll") for (int t = 0; t < TM; ++t)                                             \
            _Pragma("unroll") for (int j = 0; j < TN; ++j) {                                       \
                const f16x8 a_ = __builtin_bit_cast(f16x8, p == 0 ? fa_lo[H][t] : fa_hi[H][t]);    \
                const f16x8 b_ = __builtin_bit_cast(f16x8, p == 1 ? fb_lo[H][j] : fb_hi[H][j]);    \
                acc[t][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(b_, a_, acc[t][j], 0, 0, 0);    \
            }
    RD(0, 0, 0)
    for (int it = 0; it < taps; ++it) {
        const int kw = it & 3;
        __builtin_amdgcn_sched_barrier(0);
        MM(0)
        RD(1, kw, it)
        pin<0, 12, 8>();
        __builtin_amdgcn_sched_barrier(0);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
        MM(1)
        RD(0, (kw + 1) & 3, it + 1)
        pin<0, 12, 8>();
        __builtin_amdgcn_sched_barrier(0);
    }
    float s = 0;
    for (int t = 0; t < TM; ++t) for (int j = 0; j < TN; ++j) for (int r = 0; r < 16; ++r) s += acc[t][j][r];
    out[blockIdx.x * 512 + tid] = s;
}

// the DMA'd bytes become MFMA operands: they must look like data too (an all-zero source would make the MFMAs cheap)
__global__ void fill(float* p, size_t n) {
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
        unsigned x = (unsigned)(i + 1) * 2654435761u;
        p[i] = __uint_as_float(0x38003800u ^ (x & 0x03ff03ffu));
    }
}

template <int NDMA, int SHARED> void run(const float* src, double secs) {
    const int blocks = 256, taps = 40000;
    float* out; (void)hipMalloc(&out, blocks * 512 * 4);
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&k<NDMA, SHARED>), hipFuncAttributeMaxDynamicSharedMemorySize, 147456);
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    const double flop = (double)blocks * taps * 8 * 24 * 2.0 * 32 * 32 * 16;       // executed MFMA flops per launch
    double t = 0, last = 0; int n = 0;
    while (t < secs * 1e3) {
        (void)hipEventRecord(e0, 0);
        hipLaunchKernelGGL((k<NDMA, SHARED>), dim3(blocks), dim3(768), 147456, 0, out, src, taps);
        (void)hipEventRecord(e1, 0); (void)hipEventSynchronize(e1);
        float ms; (void)hipEventElapsedTime(&ms, e0, e1);
        t += ms; last = flop / ms / 1e9; ++n;
    }
    printf("%2d KB of LDS-DMA per tap per CU from %-3s: sustained %6.0f executed TFLOP/s (launch %d, %.0f ms each)\n", NDMA * 4,
           SHARED ? "L2" : "HBM", last, n, t / n);
    fflush(stdout);
    (void)hipFree(out);
}

int main(int argc, char** argv) {
    const double secs = argc > 1 ? atof(argv[1]) : 4.0;
    const size_t n = (size_t)(1 << 20) + (size_t)256 * 4 * (1 << 18);
    float* src; (void)hipMalloc(&src, n * 4);
    hipLaunchKernelGGL(fill, dim3(4096), dim3(256), 0, 0, src, n);
    (void)hipDeviceSynchronize();
    run<0, 1>(src, secs); run<3, 1>(src, secs); run<7, 1>(src, secs); run<10, 1>(src, secs); run<14, 1>(src, secs);
    run<3, 0>(src, secs); run<7, 0>(src, secs); run<10, 0>(src, secs);
    return 0;
}
