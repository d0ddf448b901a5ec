// Dev micro-benchmark: how much do LDS-DMA writes (issued by 4 producer waves, one barrier per tap with
// the 8 consumer waves) slow the consumer loop of conv_igemm_halo*.hip down?  KB of DMA per tap swept.
//   hipcc --offload-arch=gfx950 -O3 dma_interference.hip -o dma_interference && ./dma_interference
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

template <int TN, int NDMA, int PRIO>   // NDMA: global_load_lds instructions per producer wave per tap (1 KB each)
__global__ void __launch_bounds__(768) k(float* out, const float* src, int taps, long long* cyc) {
    constexpr int TM = 2, KS = 2;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    for (int i = tid; i < 36864; i += 768) smem[i] = (float)((i * 2654435761u) >> 20) * 1e-3f;
    __syncthreads();
    if (wave >= 8) {
        if constexpr (PRIO) __builtin_amdgcn_s_setprio(3);
        const int pw = wave - 8;
        const float* p = src + ((size_t)blockIdx.x * 4 + pw) * (1 << 16) + lane * 4;   // 256 KB private window per wave
        for (int it = 0; it < taps; ++it) {
#pragma unroll
            for (int d = 0; d < NDMA; ++d)
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(p + ((it * NDMA + d) & 255) * 256),
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
    for (int t = 0; t < TM; ++t) jb[t] = wm * 64 + t * 32 + (lane & 31);
    const int bcol = 20480 + (wn * TN) * 1024 + lane * 4;
    f32x16 acc[TM][TN];
    for (int t = 0; t < TM; ++t) for (int j = 0; j < TN; ++j) for (int r = 0; r < 16; ++r) acc[t][j][r] = 0.f;
    f32x4 fa_hi[KS][TM], fa_lo[KS][TM], fb_hi[KS][TN], fb_lo[KS][TN];
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
    const long long t0 = __builtin_amdgcn_s_memtime();
    RD(0, 0, 0)
    for (int it = 0; it < taps; ++it) {
        const int kw = it & 3;
        RD(1, kw, it)
        __builtin_amdgcn_sched_barrier(0);
        MM(0)
        __builtin_amdgcn_sched_barrier(0);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
        RD(0, (kw + 1) & 3, it + 1)
        __builtin_amdgcn_sched_barrier(0);
        MM(1)
        __builtin_amdgcn_sched_barrier(0);
    }
    const long long t1 = __builtin_amdgcn_s_memtime();
    float s = 0;
    for (int t = 0; t < TM; ++t) for (int j = 0; j < TN; ++j) for (int r = 0; r < 16; ++r) s += acc[t][j][r];
    out[blockIdx.x * 512 + tid] = s;
    if (lane == 0) cyc[blockIdx.x * 8 + wave] = t1 - t0;
}

template <int TN, int NDMA, int PRIO> void run(const float* src) {
    const int blocks = 256, taps = 2000;
    float* out; long long* cyc;
    hipMalloc(&out, blocks * 512 * 4); hipMalloc(&cyc, blocks * 64);
    hipFuncSetAttribute(reinterpret_cast<const void*>(&k<TN, NDMA, PRIO>), hipFuncAttributeMaxDynamicSharedMemorySize, 147456);
    for (int r = 0; r < 2; ++r) hipLaunchKernelGGL((k<TN, NDMA, PRIO>), dim3(blocks), dim3(768), 147456, 0, out, src, taps, cyc);
    hipDeviceSynchronize();
    long long h[2048]; hipMemcpy(h, cyc, sizeof h, hipMemcpyDeviceToHost);
    double c4 = 0; for (int i = 0; i < 256; ++i) c4 += h[i * 8 + 4];
    printf("TN %d, %2d KB of LDS-DMA per tap, producer prio %d: %.0f cycles per tap (MFMA floor %d)\n", TN, NDMA * 4, PRIO,
           c4 / 256 / taps, 2 * 2 * TN * 2 * 3 * 33);
    hipFree(out); hipFree(cyc);
}

int main() {
    float* src; hipMalloc(&src, (size_t)256 * 4 * (1 << 16) * 4); hipMemset(src, 0, (size_t)256 * 4 * (1 << 16) * 4);
    run<1, 0, 1>(src); run<1, 2, 1>(src); run<1, 3, 1>(src); run<1, 5, 1>(src); run<1, 5, 0>(src); run<1, 10, 1>(src);
    run<2, 0, 1>(src); run<2, 4, 1>(src); run<2, 7, 1>(src); run<2, 7, 0>(src); run<2, 12, 1>(src);
    return 0;
}
