// Dev micro-benchmark (round 6): what does a pure STORE stream achieve on one MI355X?  direct_conv64_4x4 writes 6.7 GB
// and conv_igemm_dma<128> (_transform) 3.5 GB per launch, both at ~3.97 TB/s, while a copy reaches ~6.3 TB/s of traffic
// (= 3.15 read + 3.15 written).  Is 4 TB/s the chip's store ceiling, or are the two kernels short of it?
//   hipcc --offload-arch=gfx950 -O3 store_stream.hip -o store_stream && ./store_stream [GB]
// Variants: store width (4 / 8 / 16 bytes per lane), cache policy (default / nt), grid (persistent 256 x G workgroups
// striding over the buffer / one workgroup per 64 KB), the row-strided pattern of direct_conv64_4x4 (a wave's instruction
// = 1 KB contiguous, consecutive instructions of a wave one image row apart), a 1 : 2 read : write mix (_transform's),
// and hipMemsetAsync beside them.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <algorithm>
typedef float fx4 __attribute__((ext_vector_type(4)));
typedef float fx2 __attribute__((ext_vector_type(2)));

template <int WIDTH, int NT>
__device__ __forceinline__ void st(char* p, float v) {
    if constexpr (WIDTH == 16) {
        fx4 x = {v, v, v, v};
        if constexpr (NT) __builtin_nontemporal_store(x, reinterpret_cast<fx4*>(p)); else *reinterpret_cast<fx4*>(p) = x;
    } else if constexpr (WIDTH == 8) {
        fx2 x = {v, v};
        if constexpr (NT) __builtin_nontemporal_store(x, reinterpret_cast<fx2*>(p)); else *reinterpret_cast<fx2*>(p) = x;
    } else {
        if constexpr (NT) __builtin_nontemporal_store(v, reinterpret_cast<float*>(p)); else *reinterpret_cast<float*>(p) = v;
    }
}

// every workgroup walks the buffer with stride gridDim * 256 * WIDTH (whole-chip wavefront of contiguous bytes)
template <int WIDTH, int NT>
__global__ void __launch_bounds__(256) k_stride(char* dst, size_t bytes, float v) {
    const size_t step = (size_t)gridDim.x * 256 * WIDTH;
    for (size_t o = ((size_t)blockIdx.x * 256 + threadIdx.x) * WIDTH; o < bytes; o += step) st<WIDTH, NT>(dst + o, v);
}

// every workgroup owns one contiguous span of the buffer (what a tile-per-workgroup kernel does)
template <int WIDTH, int NT>
__global__ void __launch_bounds__(256) k_span(char* dst, size_t bytes, size_t span, float v) {
    for (size_t s0 = (size_t)blockIdx.x * span; s0 < bytes; s0 += (size_t)gridDim.x * span)
        for (size_t o = (size_t)threadIdx.x * WIDTH; o < span && s0 + o < bytes; o += 256 * WIDTH) st<WIDTH, NT>(dst + s0 + o, v);
}

// direct_conv64_4x4's pattern: a tile = 4 rows x 16 pixels x 256 B; a wave's store = 4 pixels of one row (1 KB), the
// workgroup's four waves cover 16 pixels (4 KB) of a row, four passes walk four rows (row pitch = 201 pixels)
template <int NT>
__global__ void __launch_bounds__(256) k_rows(char* dst, int ntiles, float v) {
    const int tiles_c = 13, tiles_r = 9;                       // 35 x 201 image: 9 x 13 tiles per frame
    for (int tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        const int b = tile / (tiles_c * tiles_r), q = tile - b * (tiles_c * tiles_r);
        const int tr = q / tiles_c, tc = q - tr * tiles_c;
        for (int pass = 0; pass < 4; ++pass) {
            const int ho = tr * 4 + pass, wo = tc * 16 + (threadIdx.x >> 4);
            if (ho >= 35 || wo >= 201) continue;
            st<16, NT>(dst + ((size_t)b * 35 * 201 + (size_t)ho * 201 + wo) * 256 + (threadIdx.x & 15) * 16, v);
        }
    }
}

// 1 : 2 read : write (the 2-tap _transform conv reads a 64-channel pixel and writes a 128-channel one)
template <int NT>
__global__ void __launch_bounds__(256) k_rw(const char* src, char* dst, size_t bytes_in) {
    const size_t step = (size_t)gridDim.x * 256 * 16;
    for (size_t o = ((size_t)blockIdx.x * 256 + threadIdx.x) * 16; o < bytes_in; o += step) {
        const fx4 x = __builtin_nontemporal_load(reinterpret_cast<const fx4*>(src + o));
        const size_t d = (o >> 8) * 512 + (o & 255);
        if constexpr (NT) {
            __builtin_nontemporal_store(x, reinterpret_cast<fx4*>(dst + d));
            __builtin_nontemporal_store(x, reinterpret_cast<fx4*>(dst + d + 256));
        } else {
            *reinterpret_cast<fx4*>(dst + d) = x;
            *reinterpret_cast<fx4*>(dst + d + 256) = x;
        }
    }
}

// _transform's REAL access pattern (resblock2_1: 1x1 conv, stride 2, 64 -> 128 channels): input image 35 x 201 pixels of 256 B,
// output 18 x 101 pixels of 512 B; output pixel (ho, wo) reads input pixel (2 ho, 2 wo) -- every other 256-byte pixel of every
// other image row -- and writes 512 contiguous bytes.  16 lanes per input pixel (16 B each).
template <int NT>
__global__ void __launch_bounds__(256) k_rw_strided(const char* src, char* dst, int frames) {
    const int per = 18 * 101;
    const long long total = (long long)frames * per * 16;            // 16-byte pieces to read
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
        const long long px = i >> 4;
        const int piece = (int)(i & 15);
        const int f = (int)(px / per), r = (int)(px - (long long)f * per);
        const int ho = r / 101, wo = r - ho * 101;
        const size_t in = (((size_t)f * 35 + 2 * ho) * 201 + 2 * wo) * 256 + piece * 16;
        const fx4 x = __builtin_nontemporal_load(reinterpret_cast<const fx4*>(src + in));
        const size_t out = (size_t)px * 512 + piece * 16;
        if constexpr (NT) {
            __builtin_nontemporal_store(x, reinterpret_cast<fx4*>(dst + out));
            __builtin_nontemporal_store(x, reinterpret_cast<fx4*>(dst + out + 256));
        } else {
            *reinterpret_cast<fx4*>(dst + out) = x;
            *reinterpret_cast<fx4*>(dst + out + 256) = x;
        }
    }
}

static hipEvent_t e0, e1;
template <class F>
static void run(const char* name, double bytes, F f) {
    float best = 1e30f, sum = 0;
    for (int i = 0; i < 6; ++i) {
        (void)hipEventRecord(e0, 0);
        f();
        (void)hipEventRecord(e1, 0); (void)hipEventSynchronize(e1);
        float ms; (void)hipEventElapsedTime(&ms, e0, e1);
        if (i) { sum += ms; if (ms < best) best = ms; }
    }
    printf("%-78s %8.3f ms avg  %7.1f GB/s avg  %7.1f GB/s best\n", name, sum / 5, bytes / (sum / 5) / 1e6, bytes / best / 1e6);
    fflush(stdout);
}

int main(int argc, char** argv) {
    const double gb = argc > 1 ? atof(argv[1]) : 6.7;
    const size_t bytes = ((size_t)(gb * 1e9) >> 16) << 16;
    char *dst, *src;
    (void)hipMalloc(&dst, bytes); (void)hipMalloc(&src, bytes / 2);
    (void)hipMemset(src, 1, bytes / 2);
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    char name[160];
    printf("# %.2f GB written per launch\n", bytes / 1e9);
    run("hipMemsetAsync", (double)bytes, [&] { (void)hipMemsetAsync(dst, 0, bytes, 0); });
    run("hipMemcpyAsync D2D of half the size (bytes = read + written)", (double)bytes, [&] { (void)hipMemcpyAsync(dst, src, bytes / 2, hipMemcpyDeviceToDevice, 0); });
    for (int g : {2, 4, 8, 16, 32}) {
        snprintf(name, sizeof name, "strided, 16 B, nt, 256 x %d workgroups", g);
        run(name, (double)bytes, [&] { hipLaunchKernelGGL((k_stride<16, 1>), dim3(256 * g), dim3(256), 0, 0, dst, bytes, 1.f); });
        snprintf(name, sizeof name, "strided, 16 B, default policy, 256 x %d workgroups", g);
        run(name, (double)bytes, [&] { hipLaunchKernelGGL((k_stride<16, 0>), dim3(256 * g), dim3(256), 0, 0, dst, bytes, 1.f); });
    }
    run("strided, 8 B, nt, 256 x 8", (double)bytes, [&] { hipLaunchKernelGGL((k_stride<8, 1>), dim3(2048), dim3(256), 0, 0, dst, bytes, 1.f); });
    run("strided, 4 B, nt, 256 x 8", (double)bytes, [&] { hipLaunchKernelGGL((k_stride<4, 1>), dim3(2048), dim3(256), 0, 0, dst, bytes, 1.f); });
    for (size_t span : {(size_t)16384, (size_t)65536, (size_t)1 << 20}) {
        snprintf(name, sizeof name, "one span of %zu KB per workgroup turn, 16 B, nt, 256 x 8 persistent", span >> 10);
        run(name, (double)bytes, [&] { hipLaunchKernelGGL((k_span<16, 1>), dim3(2048), dim3(256), 0, 0, dst, bytes, span, 1.f); });
        snprintf(name, sizeof name, "one span of %zu KB per workgroup, 16 B, nt, grid = bytes / span", span >> 10);
        run(name, (double)bytes, [&] { hipLaunchKernelGGL((k_span<16, 1>), dim3((unsigned)(bytes / span)), dim3(256), 0, 0, dst, bytes, span, 1.f); });
    }
    {
        const int frames = (int)(bytes / (35.0 * 201 * 256)), ntiles = frames * 9 * 13;
        const double wb = (double)frames * 35 * 201 * 256;
        run("direct_conv64_4x4's row pattern, nt, 256 x 8 persistent", wb, [&] { hipLaunchKernelGGL((k_rows<1>), dim3(2048), dim3(256), 0, 0, dst, ntiles, 1.f); });
        run("direct_conv64_4x4's row pattern, default policy", wb, [&] { hipLaunchKernelGGL((k_rows<0>), dim3(2048), dim3(256), 0, 0, dst, ntiles, 1.f); });
        run("direct_conv64_4x4's row pattern, nt, 256 x 16 persistent", wb, [&] { hipLaunchKernelGGL((k_rows<1>), dim3(4096), dim3(256), 0, 0, dst, ntiles, 1.f); });
    }
    {
        const size_t in = bytes / 2 / 2;                      // reads `in`, writes 2 * in
        run("1 : 2 read : write, nt stores (bytes = read + written)", 3.0 * in, [&] { hipLaunchKernelGGL((k_rw<1>), dim3(2048), dim3(256), 0, 0, src, dst, in); });
        run("1 : 2 read : write, default stores (bytes = read + written)", 3.0 * in, [&] { hipLaunchKernelGGL((k_rw<0>), dim3(2048), dim3(256), 0, 0, src, dst, in); });
    }
    {
        // as many frames as the buffers hold: input 35*201*256 B per frame must fit src (bytes / 2), output 18*101*512 B dst
        const int frames = (int)std::min((bytes / 2) / (35.0 * 201 * 256), bytes / (18.0 * 101 * 512));
        const double moved = (double)frames * 18 * 101 * (256 + 512);
        run("_transform's pattern: stride-2 pixel reads (256 of every 512 B, every other row), 512 B written each, nt (bytes = read + written)",
            moved, [&] { hipLaunchKernelGGL((k_rw_strided<1>), dim3(2048), dim3(256), 0, 0, src, dst, frames); });
        run("_transform's pattern, default stores (bytes = read + written)",
            moved, [&] { hipLaunchKernelGGL((k_rw_strided<0>), dim3(2048), dim3(256), 0, 0, src, dst, frames); });
        run("_transform's pattern, nt, 256 x 16 workgroups (bytes = read + written)",
            moved, [&] { hipLaunchKernelGGL((k_rw_strided<1>), dim3(4096), dim3(256), 0, 0, src, dst, frames); });
    }
    return 0;
}
