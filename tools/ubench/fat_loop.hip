// Dev micro-benchmark: ONE consumer wave per SIMD with a 128-pixel x 64-channel wave tile (TM = 4, TN = 2:
// 12 operand reads feed 24 MFMAs per k-step) against the shipped arrangement (two consumer waves per SIMD,
// 64 x 64 wave tiles, 8 reads per 12 MFMAs), with 4 producer waves streaming LDS-DMA beside them and one
// barrier per tap.  Question: does a single in-order wave whose operand reads are interleaved between its
// MFMAs keep the matrix pipe busier than two waves that alternate read blocks and MFMA blocks in lockstep?
//   hipcc --offload-arch=gfx950 -O3 fat_loop.hip -o fat_loop && ./fat_loop
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

// NCW consumer waves: 8 -> TM 2 (two per SIMD), 4 -> TM 4 (one per SIMD).  ILV: 0 = read block then MFMA
// block, 1 = reads interleaved between the MFMAs (sched_group_barrier), 2 = MFMA block first then reads.
// NDMA: global_load_lds instructions (1 KB each) per producer wave per tap; SHARED: every workgroup streams
// the same 64 KB window (L2-resident, like the weights) instead of a private 256 KB window per wave.
template <int NCW, int ILV, int NDMA, int SHARED>
__global__ void __launch_bounds__((NCW + 4) * 64) k(float* out, const float* src, int taps, long long* cyc) {
    constexpr int TM = NCW == 8 ? 2 : 4, TN = 2;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    for (int i = tid; i < 36864; i += (NCW + 4) * 64) smem[i] = (float)((i * 2654435761u) >> 20) * 1e-3f;
    __syncthreads();
    if (wave >= NCW) {
        __builtin_amdgcn_s_setprio(3);
        const int pw = wave - NCW;
        const float* p = SHARED ? src + pw * 4096 + lane * 4 : src + ((size_t)blockIdx.x * 4 + pw) * (1 << 16) + lane * 4;
        for (int it = 0; it < taps; ++it) {
#pragma unroll
            for (int d = 0; d < NDMA; ++d)
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(p + ((it * NDMA + d) & (SHARED ? 15 : 255)) * 256),
                                                 (__attribute__((address_space(3))) void*)(smem + 20480 + ((it & 3) * 4096) + (pw * NDMA + d) % 16 * 256),
                                                 16, 0, 0);
            if constexpr (NDMA > 0) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NDMA * 2) : "memory");
            __builtin_amdgcn_s_barrier();
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        return;
    }
    const int wm = NCW == 8 ? wave >> 1 : wave >> 1, wn = wave & 1, g8 = lane >> 5;
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
    constexpr int NM = TM * TN * 3, ND = (TM + TN) * 2;
    // pin ND reads evenly between NM MFMAs (one read after every NM/ND MFMAs)
#define ILV_PIN()                                                                                  \
    _Pragma("unroll") for (int i = 0; i < ND; ++i) {                                               \
        __builtin_amdgcn_sched_group_barrier(0x008, NM / ND, 0);                                   \
        __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);                                         \
    }
    const long long t0 = __builtin_amdgcn_s_memtime();
    RD(0, 0, 0)
    for (int it = 0; it < taps; ++it) {
        const int kw = it & 3;
        if constexpr (ILV == 0) {
            RD(1, kw, it)
            __builtin_amdgcn_sched_barrier(0);
            MM(0)
        } else if constexpr (ILV == 1 || ILV == 3) {
            MM(0)
            RD(1, kw, it)
            ILV_PIN()
        } else {
            MM(0)
            __builtin_amdgcn_sched_barrier(0);
            RD(1, kw, it)
        }
        __builtin_amdgcn_sched_barrier(0);
        if constexpr (ILV == 0) {
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            __builtin_amdgcn_sched_barrier(0);
            RD(0, (kw + 1) & 3, it + 1)
            __builtin_amdgcn_sched_barrier(0);
            MM(1)
        } else if constexpr (ILV == 3) {
            // interleaved, but the barrier stays between the halves (the shipped pipeline depth)
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            __builtin_amdgcn_sched_barrier(0);
            MM(1)
            RD(0, (kw + 1) & 3, it + 1)
            ILV_PIN()
        } else if constexpr (ILV == 1) {
            // everything tap it+1 reads landed a tap ago (deeper ring): no barrier between the halves
            MM(1)
            RD(0, (kw + 1) & 3, it + 1)
            ILV_PIN()
            __builtin_amdgcn_sched_barrier(0);
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
        } else {
            MM(1)
            __builtin_amdgcn_sched_barrier(0);
            RD(0, (kw + 1) & 3, it + 1)
            __builtin_amdgcn_sched_barrier(0);
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
        }
        __builtin_amdgcn_sched_barrier(0);
    }
    const long long t1 = __builtin_amdgcn_s_memtime();
    float s = 0;
    for (int t = 0; t < TM; ++t) for (int j = 0; j < TN; ++j) for (int r = 0; r < 16; ++r) s += acc[t][j][r];
    out[blockIdx.x * 512 + tid] = s;
    if (lane == 0) cyc[blockIdx.x * 8 + wave] = t1 - t0;
}

template <int NCW, int ILV, int NDMA, int SHARED> void run(const float* src) {
    const int blocks = 256, taps = 2000;
    float* out; long long* cyc;
    (void)hipMalloc(&out, blocks * 512 * 4); (void)hipMalloc(&cyc, blocks * 64);
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&k<NCW, ILV, NDMA, SHARED>), hipFuncAttributeMaxDynamicSharedMemorySize, 147456);
    for (int r = 0; r < 2; ++r) hipLaunchKernelGGL((k<NCW, ILV, NDMA, SHARED>), dim3(blocks), dim3((NCW + 4) * 64), 147456, 0, out, src, taps, cyc);
    (void)hipDeviceSynchronize();
    long long h[2048]; (void)hipMemcpy(h, cyc, sizeof h, hipMemcpyDeviceToHost);
    double c = 0; for (int i = 0; i < 256; ++i) c += h[i * 8 + NCW - 1];
    printf("%d consumer waves (%s), %-22s %2d KB of LDS-DMA per tap (%s): %5.0f cycles per tap; MFMA floor %d\n", NCW,
           NCW == 8 ? "64x64 tiles" : "128x64 tiles", ILV == 0 ? "read block, MFMA block," : ILV == 1 ? "reads between MFMAs," : ILV == 3 ? "interleaved, mid barrier," : "MFMA block, read block,",
           NDMA * 4, SHARED ? "L2-resident source" : "HBM source", c / 256 / taps, 2 * 2 * 2 * 2 * 3 * 32);
    (void)hipFree(out); (void)hipFree(cyc);
}

int main() {
    float* src; (void)hipMalloc(&src, (size_t)256 * 4 * (1 << 16) * 4); (void)hipMemset(src, 0, (size_t)256 * 4 * (1 << 16) * 4);
    run<8, 0, 0, 0>(src); run<8, 1, 0, 0>(src); run<8, 3, 0, 0>(src); run<8, 2, 0, 0>(src);
    run<4, 0, 0, 0>(src); run<4, 1, 0, 0>(src); run<4, 3, 0, 0>(src); run<4, 2, 0, 0>(src);
    run<8, 0, 7, 0>(src); run<8, 0, 7, 1>(src); run<8, 1, 7, 0>(src); run<8, 1, 7, 1>(src); run<8, 3, 7, 0>(src); run<8, 3, 7, 1>(src);
    run<4, 1, 7, 0>(src); run<4, 1, 7, 1>(src); run<4, 3, 7, 1>(src); run<4, 0, 7, 1>(src); run<4, 2, 7, 1>(src);
    run<8, 0, 3, 1>(src); run<8, 1, 3, 1>(src); run<8, 3, 3, 1>(src);
    return 0;
}
