// Dev micro-benchmark: the consumer loop of the halo conv kernel on FOUR-wave workgroups, TWO resident per CU
// (4 waves x 160 VGPRs co-reside; 6 x 160 do not -- residency.hip), every wave both multiplying (64 x 64 tile,
// 24 MFMAs + 16 operand reads per tap) and issuing its share of the LDS-DMA (ND global_load_lds per wave per
// tap), against the shipped arrangement measured by fat_loop.hip (8 consumer + 4 producer waves, one workgroup
// per CU: 2,069 cycles per tap beside 28 KB of DMA; MFMA floor 1,536 for either).
//   hipcc --offload-arch=gfx950 -O3 quad_loop.hip -o quad_loop && ./quad_loop
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

template <int I, int NM, int NR, int NDMA> __device__ __forceinline__ void pin() {
    if constexpr (I < NR) {
        __builtin_amdgcn_sched_group_barrier(0x008, (I + 1) * NM / NR - I * NM / NR, 0);
        __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
        if constexpr (I < NDMA) __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);
        pin<I + 1, NM, NR, NDMA>();
    }
}

// LDS (floats): image 2 x 5120 (160 rows x 32), weight ring 2 x 4096  = 18432 floats = 72 KB
// ND: DMA instructions per wave per tap (1 KB each); NH of them from a private HBM window, the rest from a
// 64 KB window every workgroup shares (L2-resident, like the weights).  WHERE: 0 = DMA issued in a block after
// the barrier, 1 = spread between the MFMAs of the second half.
template <int ND, int NH, int WHERE>
__global__ void __launch_bounds__(256, 2) k(float* out, const float* src, int taps, long long* cyc) {
    constexpr int TM = 2, TN = 2;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    for (int i = tid; i < 18432; i += 256) smem[i] = (float)((i * 2654435761u) >> 20) * 1e-3f;
    __syncthreads();
    const int wm = wave >> 1, wn = wave & 1, g8 = lane >> 5;
    int jb[TM];
    for (int t = 0; t < TM; ++t) jb[t] = (wm * TM + t) * 32 + (lane & 31);
    const int bcol = 10240 + (wn * TN) * 1024 + lane * 4;
    const float* ps = src + wave * 4096 + lane * 4;                                             // shared window
    const float* ph = src + (1 << 20) + ((size_t)blockIdx.x * 4 + wave) * (1 << 16) + lane * 4; // private window
    f32x16 acc[TM][TN];
    for (int t = 0; t < TM; ++t) for (int j = 0; j < TN; ++j) for (int r = 0; r < 16; ++r) acc[t][j][r] = 0.f;
    f32x4 fa_hi[2][TM], fa_lo[2][TM], fb_hi[2][TN], fb_lo[2][TN];
#define RD(H, KW, STG)                                                                             \
    {                                                                                              \
        _Pragma("unroll") for (int t = 0; t < TM; ++t) {                                           \
            const int jr_ = jb[t] + (KW);                                                          \
            const float* ar_ = smem + ((STG) & 1) * 5120 + jr_ * 32;                               \
            const int rs_ = (jr_ >> 1) & 7;                                                        \
            fa_hi[H][t] = *reinterpret_cast<const f32x4*>(ar_ + (((2 * (H) + g8) ^ rs_) * 4));     \
            fa_lo[H][t] = *reinterpret_cast<const f32x4*>(ar_ + (((2 * (H) + g8 + 4) ^ rs_) * 4)); \
        }                                                                                          \
        _Pragma("unroll") for (int j = 0; j < TN; ++j) {                                           \
            fb_hi[H][j] = *reinterpret_cast<const f32x4*>(smem + bcol + ((STG) & 1) * 4096 + j * 1024 + (H) * 512);       \
            fb_lo[H][j] = *reinterpret_cast<const f32x4*>(smem + bcol + ((STG) & 1) * 4096 + j * 1024 + (H) * 512 + 256); \
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
#define DMA(IT)                                                                                    \
    _Pragma("unroll") for (int d = 0; d < ND; ++d) {                                               \
        const float* p_ = d < NH ? ph + (((IT) * NH + d) & 255) * 256 : ps + (((IT) * (ND - NH) + d) & 15) * 256; \
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)p_,       \
                                         (__attribute__((address_space(3))) void*)(smem + 10240 + (((IT) + 1) & 1) * 4096 + ((wave * ND + d) & 3) * 1024 + (d >> 2) * 256), \
                                         16, 0, 0);                                                \
    }
#define PIN_RD() pin<0, 12, 8, 0>();
    // second half: 12 MFMAs, 8 reads, ND DMA instructions (VMEM read mask 0x020)
#define PIN_RD_DMA() pin<0, 12, 8, ND>();
    const long long t0 = __builtin_amdgcn_s_memtime();
    RD(0, 0, 0)
    for (int it = 0; it < taps; ++it) {
        const int kw = it & 3;
        __builtin_amdgcn_sched_barrier(0);
        MM(0)
        RD(1, kw, it)
        PIN_RD()
        __builtin_amdgcn_sched_barrier(0);
        // everything tap it+1 reads was issued a tap ago by all four waves: own share landed, then the barrier
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
        if constexpr (WHERE == 0) {
            DMA(it)
            __builtin_amdgcn_sched_barrier(0);
            MM(1)
            RD(0, (kw + 1) & 3, it + 1)
            PIN_RD()
        } else {
            MM(1)
            RD(0, (kw + 1) & 3, it + 1)
            DMA(it)
            PIN_RD_DMA()
        }
        __builtin_amdgcn_sched_barrier(0);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    const long long t1 = __builtin_amdgcn_s_memtime();
    float s = 0;
    for (int t = 0; t < TM; ++t) for (int j = 0; j < TN; ++j) for (int r = 0; r < 16; ++r) s += acc[t][j][r];
    out[blockIdx.x * 256 + tid] = s;
    if (lane == 0) cyc[blockIdx.x * 4 + wave] = t1 - t0;
}

template <int ND, int NH, int WHERE> void run(const float* src, int blocks) {
    const int taps = 2000;
    float* out; long long* cyc;
    (void)hipMalloc(&out, blocks * 256 * 4); (void)hipMalloc(&cyc, blocks * 32);
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&k<ND, NH, WHERE>), hipFuncAttributeMaxDynamicSharedMemorySize, 73728);
    for (int r = 0; r < 2; ++r) hipLaunchKernelGGL((k<ND, NH, WHERE>), dim3(blocks), dim3(256), 73728, 0, out, src, taps, cyc);
    (void)hipDeviceSynchronize();
    static long long h[4096]; (void)hipMemcpy(h, cyc, blocks * 32, hipMemcpyDeviceToHost);
    double c = 0; for (int i = 0; i < blocks; ++i) c += h[i * 4 + 3];
    printf("%3d workgroups of 4 waves (%d per CU), %d KB of LDS-DMA per workgroup per tap (%d KB from HBM), DMA %s: %5.0f cycles per 128x128 tap"
           " = per 256x128-equivalent %5.0f; MFMA floor %d\n", blocks, blocks / 256, ND * 4, NH * 4,
           WHERE ? "between the MFMAs" : "in a block after the barrier", c / blocks / taps, c / blocks / taps * (blocks == 256 ? 2 : 1), 1536);
    (void)hipFree(out); (void)hipFree(cyc);
}

int main() {
    const size_t n = (size_t)(1 << 20) + (size_t)512 * 4 * (1 << 16);
    float* src; (void)hipMalloc(&src, n * 4); (void)hipMemset(src, 0, n * 4);
    run<0, 0, 0>(src, 512); run<0, 0, 0>(src, 256);
    run<5, 1, 0>(src, 512); run<5, 1, 1>(src, 512); run<5, 1, 0>(src, 256); run<5, 1, 1>(src, 256);
    run<6, 2, 0>(src, 512); run<6, 2, 1>(src, 512); run<4, 0, 1>(src, 512); run<8, 4, 1>(src, 512);
    return 0;
}
