// Dev micro-benchmark: what does ONE CU's vector-memory path deliver when every CU of the chip streams at once?
// The question behind conv_wino128.hip's K loop: 64 KB of transformed weights (L2-resident,
// the same 512 KB for all CUs) + 25 KB of input (HBM) per period of 3,072 MFMA cycles.
//   hipcc --offload-arch=gfx950 -O3 cu_vmem_rate.hip -o cu_vmem_rate && ./cu_vmem_rate
// One workgroup of 8 waves per CU (160 KB of LDS keeps a second one out), every wave loops over 1 KB wave-loads
// (global_load_dwordx4, 16 B per lane) keeping DEPTH of them in flight; MODE 0: all CUs walk the same SHARED_KB
// region (L2 hits), 1: every wave walks its own 4 MB window of a 8 GB buffer (HBM), 2: 70 % shared + 30 % private
// (the kernel's mix), 3: as 0 through LDS-DMA (global_load_lds) instead of into registers, 4: the 70 / 30 mix with
// the two streams in DIFFERENT waves (waves 0-5 shared with 14 iterations for every 18 of waves 6-7, private): do an
// L2-hit stream and an HBM stream overlap inside one CU when no wave's in-order return couples them?  5: as 1 through
// LDS-DMA.  6: as 1, every loaded KB also stored (non-temporal) to a second private window: the epilogue's mix.
// 7 / 8: as 1 / 6 with conv_wino's epilogue pattern instead of 1 KB contiguous per instruction: lane = (tile-pixel q =
// lane >> 3, channel group c8 = lane & 7) reads 16 B at q * 1280 + (c8 >> 2) * 128 + (c8 & 3) * 16 (+ 64 for the lo
// halves, + 256 per column): ten instructions cover 10 KB contiguous, each one sixteen 64-byte half lines.
// 9 / 10: as 7 / 8 with non-temporal LOADS (the kernel's residual); 11: as 8 with plain loads and plain stores.
// 12..15: the pattern with the roles of the +64 and +128 strides swapped -- an instruction covers whole 128-byte lines
// (lanes 4..7 of a group take the lo halves of lanes 0..3's channels): 12 nt loads, 13 copy nt + nt, 14 plain loads, 15 copy plain + nt.
// Modes 1 and 6 are also run on 32 / 64 / 128 workgroups: is 24 GB/s per CU the CU's limit or the chip's share?
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int DEPTH, int MODE>
__global__ void __launch_bounds__(512) k(const float* shared_src, const float* priv_src, int shared_kb, int iters, long long* cyc, float* sink, float* wr) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int gw = blockIdx.x * 8 + wave;
    const char* sp = reinterpret_cast<const char*>(shared_src) + lane * 16;
    const char* pp = reinterpret_cast<const char*>(priv_src) + (size_t)gw * (4u << 20) + lane * 16;
    const unsigned smask = (unsigned)shared_kb * 1024u - 1u;
    f32x4 r[DEPTH];
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    unsigned so = (unsigned)(gw * 1024 * 37) & smask, po = 0;
    __syncthreads();
    const long long t0 = (long long)__builtin_amdgcn_s_memtime();
    const int my_iters = MODE == 4 ? (wave < 6 ? iters * 14 / 16 : iters * 18 / 16) : iters;
    for (int it = 0; it < my_iters; ++it) {
#pragma unroll
        for (int d = 0; d < DEPTH; ++d) {
            const bool use_shared = MODE == 0 || MODE == 3 || (MODE == 2 && ((it * DEPTH + d) % 10) < 7) || (MODE == 4 && wave < 6);
            const char* p = use_shared ? sp + so : pp + po;
            if constexpr (MODE >= 7) {
                // DEPTH = 10: instruction d = (column d >> 1, half d & 1) of a 10 KB piece
                const unsigned piece = (unsigned)it * 10240u & ((4u << 20) - 1u);
                p = reinterpret_cast<const char*>(priv_src) + (size_t)gw * (4u << 20) + (piece + (unsigned)(lane >> 3) * 1280u + (unsigned)((lane & 7) >> 2) * (MODE >= 12 ? 64u : 128u) +
                                                            (unsigned)(lane & 3) * 16u + (unsigned)(d & 1) * (MODE >= 12 ? 128u : 64u) + (unsigned)(d >> 1) * 256u) % (4u << 20);
            }
            if (use_shared) so = (so + 8192u) & smask; else po = (po + 1024u) & ((4u << 20) - 1u);
            if constexpr (MODE == 3 || MODE == 5) {
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)p,
                                                 (__attribute__((address_space(3))) void*)(smem + (wave * DEPTH + d) * 256), 16, 0, 0);
            } else {
                // (asm: the compiler would otherwise wait for each load where its value is consumed)
                if constexpr (MODE == 9 || MODE == 10 || MODE == 12 || MODE == 13) asm volatile("global_load_dwordx4 %0, %1, off nt" : "=v"(r[d]) : "v"(p) : "memory");
                else asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(r[d]) : "v"(p) : "memory");
            }
        }
        __builtin_amdgcn_sched_barrier(0);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
        if constexpr (MODE == 8 || MODE == 10 || MODE == 11 || MODE == 13 || MODE == 15) {
            const unsigned piece = (unsigned)it * 10240u & ((4u << 20) - 1u);
#pragma unroll
            for (int d = 0; d < DEPTH; ++d) {
                f32x4* q = reinterpret_cast<f32x4*>(reinterpret_cast<char*>(wr) + (size_t)gw * (4u << 20) +
                    (piece + (unsigned)(lane >> 3) * 1280u + (unsigned)((lane & 7) >> 2) * (MODE >= 12 ? 64u : 128u) + (unsigned)(lane & 3) * 16u + (unsigned)(d & 1) * (MODE >= 12 ? 128u : 64u) + (unsigned)(d >> 1) * 256u) % (4u << 20));
                if constexpr (MODE == 11) *q = r[d]; else __builtin_nontemporal_store(r[d], q);
            }
        } else if constexpr (MODE == 6) {
            char* wp = reinterpret_cast<char*>(wr) + (size_t)gw * (4u << 20) + lane * 16;
#pragma unroll
            for (int d = 0; d < DEPTH; ++d)
                __builtin_nontemporal_store(r[d], reinterpret_cast<f32x4*>(wp + ((po + (unsigned)(d - DEPTH) * 1024u) & ((4u << 20) - 1u))));
        } else if constexpr (MODE != 3 && MODE != 5) {
#pragma unroll
            for (int d = 0; d < DEPTH; ++d) acc += r[d];
        }
    }
    const long long t1 = (long long)__builtin_amdgcn_s_memtime();
    if (lane == 0) cyc[gw] = t1 - t0;
    if (acc.x == 123.456f) sink[gw] = acc.x + acc.y + acc.z + acc.w;
}

template <int DEPTH, int MODE> void run(const float* s, const float* p, int shared_kb, long long* cyc, float* sink, const char* what, int ncu = 256, float* wr = nullptr) {
    const int iters = 4096 / DEPTH;
    hipFuncSetAttribute(reinterpret_cast<const void*>(&k<DEPTH, MODE>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    for (int rep = 0; rep < 2; ++rep) {
        hipEventRecord(e0);
        hipLaunchKernelGGL((k<DEPTH, MODE>), dim3(ncu), dim3(512), 160 * 1024, 0, s, p, shared_kb, iters, cyc, sink, wr);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
    }
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    std::vector<long long> h(ncu * 8);
    hipMemcpy(h.data(), cyc, h.size() * 8, hipMemcpyDeviceToHost);
    double mean = 0;
    for (long long v : h) mean += (double)v;
    mean /= h.size();
    const double bytes_cu = 8.0 * iters * DEPTH * 1024.0;
    // s_memtime ticks at 100 MHz: convert with the launch's wall time instead (bytes per CU / time / clock is what matters)
    printf("%-34s %3d WGs depth %2d shared %5d KB: %7.3f ms, %6.1f GB/s per CU, %5.2f TB/s chip (memtime ticks per wave %.0f)\n", what, ncu, DEPTH, shared_kb,
           ms, bytes_cu / (ms * 1e-3) / 1e9, bytes_cu * ncu / (ms * 1e-3) / 1e12, mean);
}

int main() {
    float *s, *p, *sink;
    long long* cyc;
    hipMalloc(&s, 64u << 20);
    hipMalloc(&p, (size_t)256 * 8 * (4u << 20));
    hipMalloc(&sink, 1 << 20);
    hipMalloc(&cyc, 1 << 20);
    hipMemset(s, 0, 64u << 20);
    hipMemset(p, 0, (size_t)256 * 8 * (4u << 20));
    for (int kb : {512, 2048, 16384}) {
        run<4, 0>(s, p, kb, cyc, sink, "L2-shared -> registers");
        run<8, 0>(s, p, kb, cyc, sink, "L2-shared -> registers");
        run<16, 0>(s, p, kb, cyc, sink, "L2-shared -> registers");
        run<32, 0>(s, p, kb, cyc, sink, "L2-shared -> registers");
    }
    run<8, 3>(s, p, 512, cyc, sink, "L2-shared -> LDS (DMA)");
    run<16, 3>(s, p, 512, cyc, sink, "L2-shared -> LDS (DMA)");
    run<4, 1>(s, p, 512, cyc, sink, "private (HBM) -> registers");
    run<8, 1>(s, p, 512, cyc, sink, "private (HBM) -> registers");
    run<16, 1>(s, p, 512, cyc, sink, "private (HBM) -> registers");
    run<32, 1>(s, p, 512, cyc, sink, "private (HBM) -> registers");
    run<8, 2>(s, p, 512, cyc, sink, "70 % shared + 30 % private");
    run<16, 2>(s, p, 512, cyc, sink, "70 % shared + 30 % private");
    run<32, 2>(s, p, 512, cyc, sink, "70 % shared + 30 % private");
    run<4, 5>(s, p, 512, cyc, sink, "private (HBM) -> LDS (DMA)");
    run<8, 5>(s, p, 512, cyc, sink, "private (HBM) -> LDS (DMA)");
    run<16, 5>(s, p, 512, cyc, sink, "private (HBM) -> LDS (DMA)");
    run<8, 4>(s, p, 512, cyc, sink, "the same mix, streams in separate waves");
    run<16, 4>(s, p, 512, cyc, sink, "the same mix, streams in separate waves");
    run<32, 4>(s, p, 512, cyc, sink, "the same mix, streams in separate waves");
    float* wr;
    hipMalloc(&wr, (size_t)256 * 8 * (4u << 20));
    for (int ncu : {85, 256}) {
        run<10, 1>(s, p, 512, cyc, sink, "HBM -> registers, contiguous", ncu);
        run<10, 7>(s, p, 512, cyc, sink, "HBM -> registers, epilogue pattern", ncu);
        run<10, 6>(s, p, 512, cyc, sink, "copy, contiguous", ncu, wr);
        run<10, 8>(s, p, 512, cyc, sink, "copy, epilogue pattern", ncu, wr);
        run<10, 9>(s, p, 512, cyc, sink, "HBM -> regs, pattern, nt loads", ncu);
        run<10, 10>(s, p, 512, cyc, sink, "copy, pattern, nt loads + nt stores", ncu, wr);
        run<10, 11>(s, p, 512, cyc, sink, "copy, pattern, plain loads + stores", ncu, wr);
        run<10, 12>(s, p, 512, cyc, sink, "HBM -> regs, full-line pattern, nt loads", ncu);
        run<10, 14>(s, p, 512, cyc, sink, "HBM -> regs, full-line pattern, plain", ncu);
        run<10, 13>(s, p, 512, cyc, sink, "copy, full-line pattern, nt + nt", ncu, wr);
        run<10, 15>(s, p, 512, cyc, sink, "copy, full-line pattern, plain + nt", ncu, wr);
    }
    for (int ncu : {32, 64, 128, 256}) {
        run<8, 1>(s, p, 512, cyc, sink, "private (HBM) -> registers", ncu);
        run<16, 1>(s, p, 512, cyc, sink, "private (HBM) -> registers", ncu);
        run<32, 1>(s, p, 512, cyc, sink, "private (HBM) -> registers", ncu);
        run<8, 6>(s, p, 512, cyc, sink, "HBM -> registers -> HBM (nt store)", ncu, wr);
        run<16, 6>(s, p, 512, cyc, sink, "HBM -> registers -> HBM (nt store)", ncu, wr);
        run<32, 6>(s, p, 512, cyc, sink, "HBM -> registers -> HBM (nt store)", ncu, wr);
    }
    return 0;
}
