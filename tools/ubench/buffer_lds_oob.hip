// Dev micro-benchmark / semantics probe: what does an LDS-DMA load through a buffer descriptor write for a lane whose
// offset is out of range -- zeros (then image padding needs no zero page and no address select), or nothing?
//   hipcc --offload-arch=gfx950 -O3 buffer_lds_oob.hip -o buffer_lds_oob && ./buffer_lds_oob
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

__global__ void __launch_bounds__(64) k(const unsigned* src, int nbytes, unsigned* out) {
    __shared__ __attribute__((aligned(16))) unsigned lds[64 * 4];
    const int lane = threadIdx.x;
    for (int i = 0; i < 4; ++i) lds[lane * 4 + i] = 0xFFFFFFFFu;
    __syncthreads();
    __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned*>(src), 0, nbytes, 0x00020000);
    // odd lanes: out of range
    const unsigned off = (lane & 1) ? 0x80000000u : (unsigned)lane * 16u;
    __builtin_amdgcn_raw_ptr_buffer_load_lds(r, (__attribute__((address_space(3))) void*)lds, 16, off, 0, 0, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    for (int i = 0; i < 4; ++i) out[lane * 4 + i] = lds[lane * 4 + i];
}

int main() {
    std::vector<unsigned> h(64 * 4);
    for (size_t i = 0; i < h.size(); ++i) h[i] = 0x1000u + (unsigned)i;
    unsigned *s, *o;
    hipMalloc(&s, h.size() * 4); hipMalloc(&o, h.size() * 4);
    hipMemcpy(s, h.data(), h.size() * 4, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, s, (int)(h.size() * 4), o);
    std::vector<unsigned> g(64 * 4);
    hipMemcpy(g.data(), o, g.size() * 4, hipMemcpyDeviceToHost);
    int ok_valid = 0, oob_zero = 0, oob_untouched = 0, oob_other = 0;
    for (int l = 0; l < 64; ++l)
        for (int i = 0; i < 4; ++i) {
            const unsigned v = g[l * 4 + i];
            if (!(l & 1)) ok_valid += v == h[l * 4 + i];
            else if (v == 0) ++oob_zero; else if (v == 0xFFFFFFFFu) ++oob_untouched; else ++oob_other;
        }
    printf("valid lanes correct: %d / 128; out-of-range lanes: %d zero, %d untouched, %d other\n", ok_valid, oob_zero, oob_untouched, oob_other);
    return 0;
}
