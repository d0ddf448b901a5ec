// Dev micro-benchmark: the consumer half-tap loop of conv_igemm_halo*.hip in isolation (8 waves, LDS
// preloaded, no DMA, no barriers unless asked): cycles per tap for reads+MFMAs / MFMAs only / reads only.
//   hipcc --offload-arch=gfx950 -O3 consumer_loop.hip -o consumer_loop && ./consumer_loop
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

template <int TN, int MODE, int BAR, int NW = 8, int SWZ = 0, int ORD = 0>   // ORD 1: older wave of a SIMD issues MFMAs first, younger reads first; 2: all MFMAs first; SWZ 1: all lanes one row (no conflicts), 2: row&7 swizzle; MODE 0: reads + MFMAs, 1: MFMAs only, 2: reads only
__global__ void __launch_bounds__(NW * 64) k(float* out, int taps, long long* cyc) {
    constexpr int TM = 2, KS = 2;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, wm = NW == 8 ? wave >> 1 : wave, wn = NW == 8 ? wave & 1 : 0, g8 = lane >> 5;
    for (int i = tid; i < 36864; i += NW * 64) smem[i] = (float)((i * 2654435761u) >> 20) * 1e-3f;
    __syncthreads();
    int jb[TM];
    for (int t = 0; t < TM; ++t) jb[t] = SWZ == 1 ? wm * 64 + t * 32 : wm * 64 + t * 32 + (lane & 31);
    const int bcol = 20480 + (wn * TN) * 1024 + lane * 4;
    f32x16 acc[TM][TN];
    for (int t = 0; t < TM; ++t) for (int j = 0; j < TN; ++j) for (int r = 0; r < 16; ++r) acc[t][j][r] = 0.f;
    f32x4 fa_hi[KS][TM], fa_lo[KS][TM], fb_hi[KS][TN], fb_lo[KS][TN];
    for (int s = 0; s < KS; ++s) {
        for (int t = 0; t < TM; ++t) { fa_hi[s][t] = f32x4{1, 2, 3, 4}; fa_lo[s][t] = f32x4{1, 2, 3, 4}; }
        for (int j = 0; j < TN; ++j) { fb_hi[s][j] = f32x4{1, 2, 3, 4}; fb_lo[s][j] = f32x4{1, 2, 3, 4}; }
    }
#define RD(H, KW, STG)                                                                             \
    if constexpr (MODE != 1) {                                                                     \
        _Pragma("unroll") for (int t = 0; t < TM; ++t) {                                           \
            const int jr_ = jb[t] + (KW);                                                          \
            const float* ar_ = smem + ((STG) & 1) * 10240 + jr_ * 32;                              \
            const int rs_ = SWZ == 2 ? (jr_ & 7) : (jr_ >> 1) & 7;                                 \
            fa_hi[H][t] = *reinterpret_cast<const f32x4*>(ar_ + (((2 * (H) + g8) ^ rs_) * 4));     \
            fa_lo[H][t] = *reinterpret_cast<const f32x4*>(ar_ + (((2 * (H) + g8 + 4) ^ rs_) * 4)); \
        }                                                                                          \
        _Pragma("unroll") for (int j = 0; j < TN; ++j) {                                           \
            fb_hi[H][j] = *reinterpret_cast<const f32x4*>(smem + bcol + ((STG) & 3) * 4096 + j * 1024 + (H) * 512);       \
            fb_lo[H][j] = *reinterpret_cast<const f32x4*>(smem + bcol + ((STG) & 3) * 4096 + j * 1024 + (H) * 512 + 256); \
        }                                                                                          \
    }
#define MM(H)                                                                                      \
    if constexpr (MODE != 2) {                                                                     \
        _Pragma("unroll") for (int p = 0; p < 3; ++p)                                              \
            _Pragma("unroll") for (int t = 0; t < TM; ++t)                                         \
                _Pragma("unroll") for (int j = 0; j < TN; ++j) {                                   \
                    const f16x8 a_ = __builtin_bit_cast(f16x8, p == 0 ? fa_lo[H][t] : fa_hi[H][t]); \
                    const f16x8 b_ = __builtin_bit_cast(f16x8, p == 1 ? fb_lo[H][j] : fb_hi[H][j]); \
                    acc[t][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(b_, a_, acc[t][j], 0, 0, 0); \
                }                                                                                  \
    } else {                                                                                       \
        _Pragma("unroll") for (int t = 0; t < TM; ++t) acc[t][0][0] += fa_hi[H][t][0] + fa_lo[H][t][1]; \
        _Pragma("unroll") for (int j = 0; j < TN; ++j) acc[0][j][1] += fb_hi[H][j][0] + fb_lo[H][j][1]; \
    }
    const long long t0 = __builtin_amdgcn_s_memtime();
    RD(0, 0, 0)
    const bool mfma_first = ORD == 2 || (ORD == 1 && wave < NW / 2);
    for (int it = 0; it < taps; ++it) {
        const int kw = it & 3;
        if (mfma_first) {
            MM(0)
            __builtin_amdgcn_sched_barrier(0);
            RD(1, kw, it)
        } else {
            RD(1, kw, it)
            __builtin_amdgcn_sched_barrier(0);
            MM(0)
        }
        __builtin_amdgcn_sched_barrier(0);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        if constexpr (BAR) __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
        if (mfma_first) {
            MM(1)
            __builtin_amdgcn_sched_barrier(0);
            RD(0, (kw + 1) & 3, it + 1)
        } else {
            RD(0, (kw + 1) & 3, it + 1)
            __builtin_amdgcn_sched_barrier(0);
            MM(1)
        }
        __builtin_amdgcn_sched_barrier(0);
    }
    const long long t1 = __builtin_amdgcn_s_memtime();
    float s = 0;
    for (int t = 0; t < TM; ++t) for (int j = 0; j < TN; ++j) for (int r = 0; r < 16; ++r) s += acc[t][j][r];
    out[blockIdx.x * NW * 64 + tid] = s;
    if (lane == 0) cyc[blockIdx.x * 8 + wave] = t1 - t0;
}

template <int TN, int MODE, int BAR, int NW = 8, int SWZ = 0, int ORD = 0> void run(const char* what) {
    const int blocks = 256, taps = 2000;
    float* out; long long* cyc;
    hipMalloc(&out, blocks * 512 * 4); hipMalloc(&cyc, blocks * 64);
    hipFuncSetAttribute(reinterpret_cast<const void*>(&k<TN, MODE, BAR, NW, SWZ, ORD>), hipFuncAttributeMaxDynamicSharedMemorySize, 147456);
    for (int r = 0; r < 2; ++r) hipLaunchKernelGGL((k<TN, MODE, BAR, NW, SWZ, ORD>), dim3(blocks), dim3(NW * 64), 147456, 0, out, taps, cyc);
    hipDeviceSynchronize();
    long long h[2048]; hipMemcpy(h, cyc, sizeof h, hipMemcpyDeviceToHost);
    double c0 = 0, c4 = 0; for (int i = 0; i < 256; ++i) { c0 += h[i * 8]; c4 += h[i * 8 + (NW == 8 ? 4 : 3)]; }
    printf("%d waves, TN %d %-12s barrier %d: %.0f cycles per tap (wave 0), %.0f (wave %d); MFMA floor %d\n", NW, TN, what, BAR,
           c0 / 256 / taps, c4 / 256 / taps, NW == 8 ? 4 : 3, (NW / 4) * 2 * TN * 2 * 3 * 33);
    hipFree(out); hipFree(cyc);
}

int main() {
    run<1, 0, 1, 8, 0, 0>("reads first");
    run<1, 0, 1, 8, 0, 1>("old MFMA first");
    run<1, 0, 1, 8, 0, 2>("all MFMA first");
    run<2, 0, 1, 8, 0, 0>("reads first");
    run<2, 0, 1, 8, 0, 1>("old MFMA first");
    run<2, 0, 1, 8, 0, 2>("all MFMA first");
    return 0;
}
