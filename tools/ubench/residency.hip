// Dev micro-benchmark: which workgroup shapes does the dispatcher keep TWO of per CU?  Each workgroup spins for
// ~100 k cycles and records where and when it ran; the host counts, per CU, the time with 0/1/2/... resident.
//   hipcc --offload-arch=gfx950 -O3 residency.hip -o residency && ./residency
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <map>
#include <vector>

template <int NW, int VG> __global__ void __launch_bounds__(NW * 64) k(long long* rec) {
    extern __shared__ float smem[];
    const long long t0 = __builtin_amdgcn_s_memtime();
    // pin the VGPR allocation to VG registers
    if constexpr (VG == 160) asm volatile("v_mov_b32 v159, 0" ::: "v159");
    if constexpr (VG == 128) asm volatile("v_mov_b32 v127, 0" ::: "v127");
    if constexpr (VG == 96) asm volatile("v_mov_b32 v95, 0" ::: "v95");
    smem[threadIdx.x] = 1.f;
    while ((long long)__builtin_amdgcn_s_memtime() - t0 < 100000) __builtin_amdgcn_s_sleep(32);
    __syncthreads();
    if (threadIdx.x == 0) {
        rec[blockIdx.x * 4 + 0] = t0;
        rec[blockIdx.x * 4 + 1] = __builtin_amdgcn_s_memtime();
        rec[blockIdx.x * 4 + 2] = ((long long)__builtin_amdgcn_s_getreg((31 << 11) | 20) << 32) | (unsigned)__builtin_amdgcn_s_getreg((31 << 11) | 4);
    }
}

template <int NW, int VG> void run(int lds) {
    const int blocks = 2048;
    long long* rec; (void)hipMalloc(&rec, blocks * 32);
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&k<NW, VG>), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    hipLaunchKernelGGL((k<NW, VG>), dim3(blocks), dim3(NW * 64), lds, 0, rec);
    (void)hipDeviceSynchronize();
    std::vector<long long> h(blocks * 4); (void)hipMemcpy(h.data(), rec, blocks * 32, hipMemcpyDeviceToHost);
    std::map<long long, std::vector<std::pair<long long, int>>> ev;
    for (int b = 0; b < blocks; ++b) {
        const long long hw = h[b * 4 + 2];
        const long long cu = ((hw >> 32) & 0xF) * 100000 + ((hw >> 8) & 0xFF);   // xcc, (se, sh, cu)
        ev[cu].push_back({h[b * 4], 1}); ev[cu].push_back({h[b * 4 + 1], -1});
    }
    std::map<int, double> share; double tot = 0;
    for (auto& kv : ev) {
        auto& v = kv.second; std::sort(v.begin(), v.end());
        int live = 0; long long last = v[0].first;
        for (auto& e : v) { share[live] += (double)(e.first - last); tot += (double)(e.first - last); live += e.second; last = e.first; }
    }
    printf("%d waves, %3d VGPRs, %6d B LDS: %zu CUs; residency share:", NW, VG, lds, ev.size());
    for (auto& kv : share) printf("  %d: %.1f%%", kv.first, 100 * kv.second / tot);
    printf("\n");
    (void)hipFree(rec);
}

int main() {
    run<6, 160>(73728); run<6, 160>(65536); run<6, 160>(32768); run<6, 128>(73728); run<6, 96>(73728);
    run<5, 160>(73728); run<4, 160>(73728); run<4, 160>(49152); run<8, 128>(73728); run<12, 160>(147456); run<3, 160>(36864);
    return 0;
}
