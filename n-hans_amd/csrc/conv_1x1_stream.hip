// The stand-alone 1 x 1 strided `_transform` convolution of a channel-changing residual block whose conv2 runs in its
// Winograd form (resblock2_1: 64 -> 128 channels, stride 2, SN/main.py:176-181; nhans_api.hip run_stack_chunk) as what it
// is: a STREAM.  1.75 GB read (every other pixel of every other image row of the block input, split NHWC) and 3.5 GB
// written (f32 NHWC, added like a residual by conv2's epilogue) per pass of 3,776 frames against 2 x 64 x 128 MACs per
// output element: the generic implicit-GEMM kernels -- 256-pixel tiles, one workgroup per CU, load -> multiply -> epilogue
// through LDS, one after the other -- all ran it at 3.95 TB/s (three of them in the same time:
// profiles/r06/ab_transform_conv_three_kernels_same_time.txt) where a copy kernel with exactly this access pattern
// reaches 5.2 TB/s (tools/ubench/store_stream.hip).  Here a WAVE owns 32 output pixels at a time: the 128 x 64 weights
// (hi + lo f16 fragments, 128 registers) stay in registers for the whole kernel, the 32 pixels' 8 KB come in by eight
// coalesced 16-byte loads per lane (a lane instruction covers four whole pixels), cross a wave-private padded LDS
// image into MFMA operand order, 48 MFMAs, and the 16 KB of results leave straight from the accumulators.  Eight waves
// per CU, each with ~20 k cycles per group before the chip's HBM rate is the limit: nothing here needs pipelining.
//
// Arithmetic: exactly conv_igemm_dma.hip's for this layer -- per accumulator the k-steps in ascending order, each as
// W_hi X_lo, W_lo X_hi, W_hi X_hi (v_mfma_f32_32x32x16_f16, f32 accumulate), then acc * (ws * in_scale) -- bit for bit.
#include "conv_epilogue.h"      // (vector types)
#include <algorithm>

namespace nhans {

namespace {
__device__ __forceinline__ void wave_lds_sync_1x1() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}
}  // namespace

constexpr int S1_LDO = 132;                      // floats per pixel row of the output image in LDS (128 + 4: conflict-free 16-byte writes)
constexpr int S1_WAVE = 32 * S1_LDO;             // floats per wave: the output image (16.5 KB); the input image (32 x 17 x 16 B) aliases it
static_assert(32 * 17 * 4 <= S1_WAVE, "input image inside the wave's area");

__global__ void __launch_bounds__(256, 2) conv_1x1_stream_kernel(const Stream1x1Args a) {
    // per wave one LDS area, used twice per group: (1) the 32 input pixels, 16 pieces of 16 B each in rows of 17 (272 B: the 32
    // lanes that read one piece of 32 pixels hit 16 x 4 different banks; every address is one base register + an immediate);
    // (2) the 32 x 128 f32 results, so that the stores leave as whole 512-byte pixels -- straight from
    // the accumulators a store instruction scatters 32-byte pieces over 32 pixels, and the kernel ran at 1.5 TB/s
    __shared__ __attribute__((aligned(16))) float lds[4][S1_WAVE];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);

    // weights -> registers: fold.py pack_igemm_h3 [chunk 2][n-tile 4][s 2][h 2][64 lanes][8 halves]; k-step ks = 2 chunk + s
    f16x8 wh[4][4], wl[4][4];
#pragma unroll
    for (int ks = 0; ks < 4; ++ks)
#pragma unroll
        for (int nt = 0; nt < 4; ++nt) {
            const int base = ((((ks >> 1) * 4 + nt) * 2 + (ks & 1)) * 2) * 64 + lane;       // in 16-byte units
            wh[ks][nt] = __builtin_bit_cast(f16x8, *reinterpret_cast<const f32x4*>(a.wpk + (size_t)base * 4));
            wl[ks][nt] = __builtin_bit_cast(f16x8, *reinterpret_cast<const f32x4*>(a.wpk + (size_t)(base + 64) * 4));
        }
    // on the way out a lane always carries channels 4 (lane & 31) .. + 3: its four unscale factors live in registers
    const f32x4 w4 = *reinterpret_cast<const f32x4*>(a.ws + 4 * (lane & 31)) * a.in_scale;

    const int p = lane & 31, kh = lane >> 5;
    const int ngroups = (a.M + 31) >> 5;
    float* const lw = lds[wave];
    f32x4* const xw = reinterpret_cast<f32x4*>(lw);
    for (int g = blockIdx.x * 4 + wave; g < ngroups; g += gridDim.x * 4) {
        const int m0 = g << 5;
        // ---- the group's 32 input pixels (256 B each): instruction i fetches pixels 4i .. 4i+3, 16 lanes x 16 B per pixel
        // (a pixel past the end reads the last one: never stored)
        // (two halves of four instructions: 128 of the 256 registers hold the weights)
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            f32x4 r[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int pi = 4 * (4 * h + i) + (lane >> 4), q = lane & 15;
                const int m = m0 + pi < a.M ? m0 + pi : a.M - 1;
                const uint32_t b = fd_div((uint32_t)m, a.fdHoWo);
                const uint32_t rem = (uint32_t)m - b * a.fdHoWo.d;
                const uint32_t ho = fd_div(rem, a.fdWo);
                const uint32_t wo = rem - ho * a.fdWo.d;
                const size_t px = ((size_t)b * a.H + ho * a.sh) * a.W + wo * a.sw;
                r[i] = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(a.src + px * 64) + q);
            }
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int pi = 4 * (4 * h + i) + (lane >> 4), q = lane & 15;
                xw[pi * 17 + q] = r[i];
            }
        }
        wave_lds_sync_1x1();
        // ---- MFMA operands: lane = (pixel p, k half kh); k-step ks = channels 16 ks .. + 15 of the pixel: group ks >> 1,
        // hi halves at piece (ks >> 1) * 8 + (ks & 1) * 2 + kh, lo halves four pieces on
        __builtin_amdgcn_sched_barrier(0);
        f16x8 xh[4], xl[4];
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            const int qh = (ks >> 1) * 8 + (ks & 1) * 2 + kh;
            xh[ks] = __builtin_bit_cast(f16x8, xw[p * 17 + qh]);
            xl[ks] = __builtin_bit_cast(f16x8, xw[p * 17 + qh + 4]);
        }
        // ---- two n-tiles at a time (32 accumulator registers beside the 128 of the weights), then accumulators -> LDS: the
        // lane's quad q4 of n-tile nt is channels 32 nt + 8 q4 + 4 kh .. + 3 of pixel p.  (The operand reads above have
        // returned -- the first MFMA consumed them -- before the first write into the area they came from.)
#pragma unroll
        for (int half = 0; half < 2; ++half) {
            f32x16 acc[2];
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int e = 0; e < 16; ++e) acc[j][e] = 0.f;
#pragma unroll
            for (int ks = 0; ks < 4; ++ks)
#pragma unroll
                for (int pr = 0; pr < 3; ++pr)
#pragma unroll
                    for (int j = 0; j < 2; ++j)
                        acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(pr == 1 ? wl[ks][2 * half + j] : wh[ks][2 * half + j],
                                                                        pr == 0 ? xl[ks] : xh[ks], acc[j], 0, 0, 0);
            if (half == 0) wave_lds_sync_1x1();
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int q4 = 0; q4 < 4; ++q4)
                    *reinterpret_cast<f32x4*>(lw + p * S1_LDO + 32 * (2 * half + j) + 8 * q4 + 4 * kh) =
                        f32x4{acc[j][4 * q4], acc[j][4 * q4 + 1], acc[j][4 * q4 + 2], acc[j][4 * q4 + 3]};
            __builtin_amdgcn_sched_barrier(0);               // (the halves one after the other: 32 accumulator registers, not 64)
        }
        wave_lds_sync_1x1();
        // ---- out[m, n] = acc * (ws[n] * in_scale), two whole pixels (2 x 512 B, contiguous) per store instruction.
        // (+ 0.0f and the -3e38 floor: the generic epilogue's zero bias, absent table, absent residual and its branch-free
        // "no ReLU", which turn a -0.0 into +0.0 and a NaN into -3e38)
        float* const o = a.out + (size_t)m0 * 128 + 4 * (lane & 31);
#pragma unroll
        for (int j = 0; j < 16; ++j) {
            const int pj = 2 * j + (lane >> 5);
            const f32x4 v = *reinterpret_cast<const f32x4*>(lw + pj * S1_LDO + 4 * (lane & 31));
            f32x4 y;
#pragma unroll
            for (int e = 0; e < 4; ++e) y[e] = fmaxf((__builtin_fmaf(v[e], w4[e], 0.f) + 0.f) + 0.f, -3.0e38f);
            if (m0 + pj < a.M) __builtin_nontemporal_store(y, reinterpret_cast<f32x4*>(o + (size_t)pj * 128));
            if ((j & 3) == 3) __builtin_amdgcn_sched_barrier(0);     // (four at a time: the weights hold half of the registers)
        }
        wave_lds_sync_1x1();                                // (the next group's input image goes into the same area)
    }
}

bool conv_1x1_stream_eligible(const ConvArgs& t) {
    const ConvSeg& g = t.seg[0];
    return t.prec == 1 && t.nseg == 1 && g.KH == 1 && g.KW == 1 && g.C == 64 && t.N == 128 && t.Nreal == 128 && t.ldo == 128 &&
           !t.in_f32 && !t.out_split && t.id_mode == 0 && !t.tf && !t.relu && !t.aux && t.cb_stride == 0 && t.out_scale == 1.f &&
           g.pt == 0 && g.pl == 0 && t.ws != nullptr && t.cb == t.zero && t.kgroup == 0 && t.variant >= 1 &&
           (double)t.M * 128.0 < 2147483648.0 * 4.0;
}

void launch_conv_1x1_stream(const ConvArgs& t, hipStream_t s) {
    const ConvSeg& g = t.seg[0];
    Stream1x1Args a{};
    a.src = g.src; a.wpk = g.wpk; a.ws = t.ws; a.in_scale = t.in_scale; a.out = t.out;
    a.H = g.H; a.W = g.W; a.sh = g.sh; a.sw = g.sw; a.M = t.M;
    a.fdHoWo = t.fdHoWo; a.fdWo = t.fdWo;
    const int groups = (t.M + 31) / 32;
    const int grid = std::min(256 * 2, (groups + 3) / 4);
    NHANS_LAUNCH("conv_1x1_stream", conv_1x1_stream_kernel, dim3(grid), dim3(256), 0, s, a);
}

}  // namespace nhans
