// Bandwidth-bound helper kernels of the N-HANS hot path: the 1-channel first-layer convolution,
// sliding-window gather, global average pool, conditioning projections, frame index.
#include "nhans_kernels.h"

namespace nhans {

// Clamp to the f16 range before a split (hi/lo) store; true if anything was out of range or NaN.
__device__ __forceinline__ bool split_clamp(float4& v, float limit) {
    const bool sat = !(fmaxf(fmaxf(fabsf(v.x), fabsf(v.y)), fmaxf(fabsf(v.z), fabsf(v.w))) < limit);
    v.x = fminf(fmaxf(v.x, -65504.f), 65504.f); v.y = fminf(fmaxf(v.y, -65504.f), 65504.f);
    v.z = fminf(fmaxf(v.z, -65504.f), 65504.f); v.w = fminf(fmaxf(v.w, -65504.f), 65504.f);
    return sat;
}

// ---------------------------------------------------------------------------------------------
// First conv of a block whose input is the 1-channel log-magnitude image (K = KH*KW <= 32):
// main stack resblock1_1_conv1 (4x4, SN/main.py:162-168) and the tower's noise_resblock1_1_conv1
// (8x4 stride (3,2), SN/main.py:104-107).  16 lanes x float4 cover the 64 output channels of one
// pixel, so every store is a full 256-byte NHWC pixel; the taps live in LDS.
__global__ void __launch_bounds__(256) direct_conv64(const DirectArgs a) {
    __shared__ __attribute__((aligned(16))) float wsh[32 * 64];
    const int taps = a.KH * a.KW;
    for (int i = threadIdx.x; i < taps * 64; i += 256) wsh[i] = a.w[i];
    __syncthreads();
    const int cq = threadIdx.x & 15;
    const int c = cq * 4;
    for (int m = blockIdx.x * 16 + (threadIdx.x >> 4); m < a.M; m += gridDim.x * 16) {
        const uint32_t b = fd_div((uint32_t)m, a.fdHoWo);
        const uint32_t rem = (uint32_t)m - b * a.fdHoWo.d;
        const uint32_t ho = fd_div(rem, a.fdWo);
        const uint32_t wo = rem - ho * a.fdWo.d;
        const float* img = a.src + (size_t)b * a.H * a.W;
        const int hi0 = (int)ho * a.sh - a.pt, wi0 = (int)wo * a.sw - a.pl;
        float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
        for (int kh = 0; kh < a.KH; ++kh) {
            const int hi = hi0 + kh;
            if ((unsigned)hi >= (unsigned)a.H) continue;
            for (int kw = 0; kw < a.KW; ++kw) {
                const int wi = wi0 + kw;
                if ((unsigned)wi >= (unsigned)a.W) continue;
                const float x = img[hi * a.W + wi];
                const float4 w = *reinterpret_cast<const float4*>(&wsh[(kh * a.KW + kw) * 64 + c]);
                acc.x = fmaf(x, w.x, acc.x); acc.y = fmaf(x, w.y, acc.y);
                acc.z = fmaf(x, w.z, acc.z); acc.w = fmaf(x, w.w, acc.w);
            }
        }
        const int clip = a.img_clip ? a.img_clip[b] : 0;
        const float4 cb = *reinterpret_cast<const float4*>(a.cb + (size_t)clip * a.cb_stride + c);
        acc.x += cb.x; acc.y += cb.y; acc.z += cb.z; acc.w += cb.w;
        if (a.tt) {                                       // the table's two terms (60 KB: cache-resident)
            const float4 t = *reinterpret_cast<const float4*>(a.tt + ho * 64 + c);
            const float4 f = *reinterpret_cast<const float4*>(a.ff + wo * 64 + c);
            acc.x = (acc.x + t.x) + f.x; acc.y = (acc.y + t.y) + f.y; acc.z = (acc.z + t.z) + f.z; acc.w = (acc.w + t.w) + f.w;
        } else if (a.tf) {
            const float4 t = *reinterpret_cast<const float4*>(a.tf + (size_t)rem * 64 + c);
            acc.x += t.x; acc.y += t.y; acc.z += t.z; acc.w += t.w;
        }
        if (a.relu) {
            acc.x = fmaxf(acc.x, 0.f); acc.y = fmaxf(acc.y, 0.f);
            acc.z = fmaxf(acc.z, 0.f); acc.w = fmaxf(acc.w, 0.f);
        }
        acc.x *= a.out_scale; acc.y *= a.out_scale; acc.z *= a.out_scale; acc.w *= a.out_scale;   // (1 for f32 outputs)
        if (a.out_split) {
            // split NHWC: group g = c/32 of pixel m is one 128-byte line, 32 hi halfs then 32 lo halfs
            typedef _Float16 h4 __attribute__((ext_vector_type(4)));
            h4 hi, lo;
            if (split_clamp(acc, a.sat_limit) && a.sat) atomicOr(a.sat, kSatActivation);
            hi.x = (_Float16)acc.x; hi.y = (_Float16)acc.y; hi.z = (_Float16)acc.z; hi.w = (_Float16)acc.w;
            lo.x = (_Float16)(acc.x - (float)hi.x); lo.y = (_Float16)(acc.y - (float)hi.y);
            lo.z = (_Float16)(acc.z - (float)hi.z); lo.w = (_Float16)(acc.w - (float)hi.w);
            _Float16* line = reinterpret_cast<_Float16*>(a.out + (size_t)m * 64) + (c >> 5) * 64 + (c & 31);
            __builtin_nontemporal_store(hi, reinterpret_cast<h4*>(line));
            __builtin_nontemporal_store(lo, reinterpret_cast<h4*>(line + 32));
        } else {
            *reinterpret_cast<float4*>(a.out + (size_t)m * 64 + c) = acc;
        }
    }
}

// The 4x4 stride-1 case (resblock1_1_conv1: 7,035 pixels per frame-window; a pure HBM writer -- 6.7 GB per pass of 3,776
// frames against 0.1 GB read): persistent blocks, the thread's 16 x 4 weights live in registers, a block takes a strip
// of D4_TR rows x 16 output pixels at a time from its zero-padded input patch in LDS; a pass = one image row of the
// strip = 16 LDS values + 64 FMAs (32 v_pk_fma_f32) per 16-byte store, the rows the next pass shares stay in registers.
// Same taps in the same order as the generic kernel (a padded tap adds 0 * w): bitwise the same.
//
// Round 6.  (i) The input is read where it lies: `win` makes frame b's 35 x 201 image a sliding window of the
// log-magnitude spectrogram (SN/apply.py:378's strided_crop is never materialised: the gather_windows kernel and its
// 104 MB per pass are gone).  (ii) What held this kernel at 3.97 TB/s where a pure store stream of its shape reaches
// 6.4-6.6 (tools/ubench/store_stream.hip, profiles/r06) was neither bytes nor arithmetic: on gfx9 a wave's loads and
// stores retire through ONE in-order counter (vmcnt) -- a load issued behind a store cannot be waited for without that
// store's acknowledgement from the L2 as well -- and inside a loop, where the compiler cannot count the stores in
// flight, every wait for a load becomes vmcnt(0).  The old kernel loaded two position-table rows per pass: one full
// write round trip per 16 bytes stored per lane.  Now nothing is loaded from global memory inside a strip: the time term of the table sits in LDS
// (35 x 64 floats, staged once per block), the frequency term and the clip's bias are fetched once per strip BEFORE its
// first store, and the next strip's patch is prefetched there too and parked in the other LDS buffer at the strip's
// end -- the stores drain once per 35 passes instead of once per pass.
constexpr int D4_TR = 35, D4_PW = 19, D4_PH = D4_TR + 3, D4_NPE = (D4_PH * D4_PW + 255) / 256, D4_PATCH = D4_NPE * 256;
constexpr int D4_U = 5;                             // passes unrolled together (35 = 7 x 5)
static_assert(D4_TR % D4_U == 0, "whole groups of passes");
template <int SPLIT>
__global__ void __launch_bounds__(256) direct_conv64_4x4(const DirectArgs a, int tiles_r, int tiles_c, int ntiles) {
    __shared__ float patch[2][D4_PATCH];
    __shared__ __attribute__((aligned(16))) float ttl[D4_TR * 64];
    const int cq = threadIdx.x & 15, c = cq * 4, pc = threadIdx.x >> 4;
    float4 w[16];
#pragma unroll
    for (int t = 0; t < 16; ++t) w[t] = *reinterpret_cast<const float4*>(a.w + t * 64 + c);
    const int tpi = tiles_r * tiles_c;
    // element e of tile `tile`'s patch (0.0: padding, a row outside the clip, no such tile)
    auto fetch = [&](int tile, int e) -> float {
        if (tile >= ntiles || e >= D4_PH * D4_PW) return 0.f;
        const int pi_h = e / D4_PW, pi_w = e - pi_h * D4_PW;
        const int b = tile / tpi, q = tile - b * tpi;
        const int tr = q / tiles_c;
        const int hi = tr * D4_TR - a.pt + pi_h, wi = (q - tr * tiles_c) * 16 - a.pl + pi_w;
        if ((unsigned)hi >= (unsigned)a.H || (unsigned)wi >= (unsigned)a.W) return 0.f;
        if (a.win.t) return win_row_ok(a.win, b, hi) ? a.src[((ptrdiff_t)a.win.row0 + b + hi) * a.W + wi] : 0.f;
        return a.src[((size_t)b * a.H + hi) * a.W + wi];
    };
    // the table's time term for the rows of the strip (the launcher admits Ho <= D4_TR: one strip per image column block)
    if (a.tt)
        for (int i = threadIdx.x; i < a.Ho * 64; i += 256) ttl[i] = a.tt[i];
    const int out_step = a.Wo * 64;
    int cur = 0;
#pragma unroll
    for (int k = 0; k < D4_NPE; ++k) patch[0][k * 256 + threadIdx.x] = fetch(blockIdx.x, k * 256 + (int)threadIdx.x);
    __syncthreads();
    for (int tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        // every global load of this strip AND the next strip's patch: issued here, before the strip's first store
        float nxt[D4_NPE];
#pragma unroll
        for (int k = 0; k < D4_NPE; ++k) nxt[k] = fetch(tile + gridDim.x, k * 256 + (int)threadIdx.x);
        const int b = tile / tpi;
        const int q = tile - b * tpi;
        const int tr = q / tiles_c;
        const int ho0 = tr * D4_TR, wo = (q - tr * tiles_c) * 16 + pc;
        const int npass = a.Ho - ho0 < D4_TR ? a.Ho - ho0 : D4_TR;
        const bool live = wo < a.Wo;
        const int wo_c = live ? wo : 0;
        const int clip = a.img_clip ? a.img_clip[b] : 0;
        float4 cb_, fq_;
        const float4 cb = *reinterpret_cast<const float4*>(a.cb + (size_t)clip * a.cb_stride + c);
        const float4 fq = a.tt ? *reinterpret_cast<const float4*>(a.ff + wo_c * 64 + c) : make_float4(0.f, 0.f, 0.f, 0.f);
        // (the compiler waits for these two loads HERE -- in front of the strip's first store, with only the previous
        // strip's last stores to drain -- and the loop below sees registers no load is pending on)
        {
            float4 cbv = cb, fqv = fq;
            asm volatile("" : "+v"(cbv.x), "+v"(cbv.y), "+v"(cbv.z), "+v"(cbv.w), "+v"(fqv.x), "+v"(fqv.y), "+v"(fqv.z), "+v"(fqv.w));
            cb_ = cbv; fq_ = fqv;
        }
        if (live) {
            const int rem0 = ho0 * a.Wo + wo;
            float* op = a.out + ((size_t)b * a.Ho * a.Wo + rem0) * 64 + c;
            const float* tfp = (!a.tt && a.tf) ? a.tf + (size_t)rem0 * 64 + c : nullptr;    // (unsplit table: a global load per pass, the slow way)
            const float* const px = patch[cur] + pc;
#pragma unroll 1
            for (int p0 = 0; p0 < D4_TR; p0 += D4_U) {
                if (p0 >= npass) break;
#pragma unroll
                for (int u = 0; u < D4_U; ++u) {
                    const int pass = p0 + u;
                    if (pass >= npass) break;
                    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
                    for (int t = 0; t < 16; ++t) {
                        const float x = px[(pass + (t >> 2)) * D4_PW + (t & 3)];
                        acc.x = fmaf(x, w[t].x, acc.x); acc.y = fmaf(x, w[t].y, acc.y);
                        acc.z = fmaf(x, w[t].z, acc.z); acc.w = fmaf(x, w[t].w, acc.w);
                    }
                    acc.x += cb_.x; acc.y += cb_.y; acc.z += cb_.z; acc.w += cb_.w;
                    if (a.tt) {
                        const float4 t = *reinterpret_cast<const float4*>(ttl + pass * 64 + c);
                        acc.x = (acc.x + t.x) + fq_.x; acc.y = (acc.y + t.y) + fq_.y; acc.z = (acc.z + t.z) + fq_.z; acc.w = (acc.w + t.w) + fq_.w;
                    } else if (tfp) {
                        const float4 t = *reinterpret_cast<const float4*>(tfp + (size_t)pass * out_step);
                        acc.x += t.x; acc.y += t.y; acc.z += t.z; acc.w += t.w;
                    }
                    if (a.relu) {
                        acc.x = fmaxf(acc.x, 0.f); acc.y = fmaxf(acc.y, 0.f);
                        acc.z = fmaxf(acc.z, 0.f); acc.w = fmaxf(acc.w, 0.f);
                    }
                    acc.x *= a.out_scale; acc.y *= a.out_scale; acc.z *= a.out_scale; acc.w *= a.out_scale;
                    if constexpr (SPLIT) {
                        typedef _Float16 h4 __attribute__((ext_vector_type(4)));
                        h4 hi, lo;
                        if (split_clamp(acc, a.sat_limit) && a.sat) atomicOr(a.sat, kSatActivation);
                        hi.x = (_Float16)acc.x; hi.y = (_Float16)acc.y; hi.z = (_Float16)acc.z; hi.w = (_Float16)acc.w;
                        lo.x = (_Float16)(acc.x - (float)hi.x); lo.y = (_Float16)(acc.y - (float)hi.y);
                        lo.z = (_Float16)(acc.z - (float)hi.z); lo.w = (_Float16)(acc.w - (float)hi.w);
                        _Float16* line = reinterpret_cast<_Float16*>(op - c) + (c >> 5) * 64 + (c & 31);
                        __builtin_nontemporal_store(hi, reinterpret_cast<h4*>(line));
                        __builtin_nontemporal_store(lo, reinterpret_cast<h4*>(line + 32));
                    } else {
                        // (f32 storage of a tensor the Winograd kernel reads, or the f32 mode: the same flag and clamp as
                        // a split store -- the reader's transform re-splits what it reads)
                        if (a.sat && split_clamp(acc, a.sat_limit)) atomicOr(a.sat, kSatActivation);
                        typedef float fx4 __attribute__((ext_vector_type(4)));
                        __builtin_nontemporal_store(fx4{acc.x, acc.y, acc.z, acc.w}, reinterpret_cast<fx4*>(op));
                    }
                    op += out_step;
                }
            }
        }
        // the next strip's patch into the other buffer: its last readers passed the barrier at the end of the previous turn
#pragma unroll
        for (int k = 0; k < D4_NPE; ++k) patch[cur ^ 1][k * 256 + threadIdx.x] = nxt[k];
        __syncthreads();
        cur ^= 1;
    }
}

__global__ void __launch_bounds__(256) unsplit_kernel(const float* src, int64_t total, int C, float scale, float* dst) {
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        const int64_t m = i / C;
        const int n = (int)(i - m * C);
        const _Float16* p = reinterpret_cast<const _Float16*>(src + m * C) + (n >> 5) * 64 + (n & 31);
        dst[i] = ((float)p[0] + (float)p[32]) * scale;
    }
}

void launch_unsplit(const float* src, int64_t M, int C, float scale, float* dst, hipStream_t s) {
    const int64_t total = M * C;
    if (total <= 0) return;
    int64_t blocks = (total + 255) / 256;
    if (blocks > 256 * 32) blocks = 256 * 32;
    NHANS_LAUNCH("unsplit", unsplit_kernel, dim3((unsigned)blocks), dim3(256), 0, s, src, total, C, scale, dst);
}

__global__ void __launch_bounds__(256) scale_copy_kernel(const float* src, size_t n, float scale, float* dst) {
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) dst[i] = src[i] * scale;
}

void launch_scale_copy(const float* src, size_t n, float scale, float* dst, hipStream_t s) {
    if (n == 0) return;
    size_t blocks = (n + 255) / 256;
    if (blocks > 256 * 16) blocks = 256 * 16;
    NHANS_LAUNCH("scale_copy", scale_copy_kernel, dim3((unsigned)blocks), dim3(256), 0, s, src, n, scale, dst);
}

// Calibration tap (nhans_api.hip: activation exponents): running maximum of |x| * scale over a tensor.  Split tensors
// are read as plain halfs -- the hi half of a value bounds it to 2^-11, and a lo half is never larger than its hi.
__global__ void __launch_bounds__(256) absmax_kernel(const uint32_t* x, size_t nwords, int split, float scale, unsigned* slot) {
    float m = 0.f;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < nwords; i += (size_t)gridDim.x * 256) {
        const uint32_t w = x[i];
        float v;
        if (split) {
            typedef _Float16 h2 __attribute__((ext_vector_type(2)));
            const h2 h = __builtin_bit_cast(h2, w);
            v = fmaxf(fabsf((float)h.x), fabsf((float)h.y));
        } else {
            v = fabsf(__builtin_bit_cast(float, w));
        }
        m = v > m || v != v ? v : m;                       // (a NaN sticks: the host sees it)
    }
    m *= scale;
    for (int o = 32; o > 0; o >>= 1) {
        const float t = __shfl_xor(m, o);
        m = t > m || t != t ? t : m;
    }
    // non-negative floats order like their bit patterns; a NaN has the largest pattern of all
    if ((threadIdx.x & 63) == 0) atomicMax(slot, __builtin_bit_cast(unsigned, m));
}

void launch_absmax(const float* x, size_t nwords, int split, float scale, unsigned* slot, hipStream_t s) {
    if (nwords == 0) return;
    size_t blocks = (nwords + 255) / 256;
    if (blocks > 256 * 16) blocks = 256 * 16;
    NHANS_LAUNCH("absmax", absmax_kernel, dim3((unsigned)blocks), dim3(256), 0, s, reinterpret_cast<const uint32_t*>(x), nwords, split, scale, slot);
}

void launch_direct_conv64(const DirectArgs& a, hipStream_t s) {
    if (a.KH == 4 && a.KW == 4 && a.sh == 1 && a.sw == 1 && a.Ho <= D4_TR && a.M % (a.Ho * a.Wo) == 0) {
        const int tiles_r = (a.Ho + D4_TR - 1) / D4_TR, tiles_c = (a.Wo + 15) / 16;
        const int ntiles = (a.M / (a.Ho * a.Wo)) * tiles_r * tiles_c;
        const int grid4 = ntiles < 256 * 8 ? ntiles : 256 * 8;
        if (a.out_split) NHANS_LAUNCH("direct_conv64_4x4", direct_conv64_4x4<1>, dim3(grid4), dim3(256), 0, s, a, tiles_r, tiles_c, ntiles);
        else NHANS_LAUNCH("direct_conv64_4x4", direct_conv64_4x4<0>, dim3(grid4), dim3(256), 0, s, a, tiles_r, tiles_c, ntiles);
        return;
    }
    if (a.win.t) {                  // (sliding-window images are read by the 4x4 stride-1 kernel only: the one first conv that has them)
        note_refusal("direct_conv64 (sliding-window input with a filter the 4x4 kernel does not take)");
        return;
    }
    int grid = (a.M + 15) / 16;
    if (grid > 256 * 16) grid = 256 * 16;
    NHANS_LAUNCH("direct_conv64", direct_conv64, dim3(grid), dim3(256), 0, s, a);
}

// ---------------------------------------------------------------------------------------------
__global__ void frame_index_kernel(const int64_t* off, int nclips, int64_t total, int* f_clip, int* f_t,
                                   int* f_T) {
    const int64_t g = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (g >= total) return;
    int lo = 0, hi = nclips;          // largest c with off[c] <= g
    while (hi - lo > 1) {
        const int mid = (lo + hi) >> 1;
        if (off[mid] <= g) lo = mid; else hi = mid;
    }
    f_clip[g] = lo;
    f_t[g] = (int)(g - off[lo]);
    f_T[g] = (int)(off[lo + 1] - off[lo]);
}

void launch_frame_index(const int64_t* frame_offsets_dev, int nclips, int64_t total, int* f_clip, int* f_t,
                        int* f_T, hipStream_t s) {
    if (total <= 0) return;
    NHANS_LAUNCH("frame_index", frame_index_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s,
                       frame_offsets_dev, nclips, total, f_clip, f_t, f_T);
}

// ---------------------------------------------------------------------------------------------
// tf.nn.avg_pool2d over the whole map (SN/main.py:199-202).  One block per (image, 64 channels);
// fixed summation order -> deterministic.
__global__ void __launch_bounds__(256) avgpool_kernel(const float* x, int HW, int C, int split, float scale, float* out) {
    __shared__ float part[4][64];
    const int b = blockIdx.x, cg = blockIdx.y;
    const int c = cg * 64 + (threadIdx.x & 63), p = threadIdx.x >> 6;
    const float* base = x + (size_t)b * HW * C + c;
    float s = 0.f;
    if (split) {
        const _Float16* hb = reinterpret_cast<const _Float16*>(x + (size_t)b * HW * C) + (c >> 5) * 64 + (c & 31);
        for (int i = p; i < HW; i += 4) s += (float)hb[(size_t)i * 2 * C] + (float)hb[(size_t)i * 2 * C + 32];
    } else {
        for (int i = p; i < HW; i += 4) s += base[(size_t)i * C];
    }
    part[p][threadIdx.x & 63] = s;
    __syncthreads();
    if (p == 0) {
        const int l = threadIdx.x & 63;
        out[(size_t)b * C + c] = (((part[0][l] + part[1][l]) + part[2][l]) + part[3][l]) / (float)HW * scale;
    }
}

void launch_avgpool(const float* x, int B, int HW, int C, int split, float scale, float* out, hipStream_t s) {
    if (B <= 0) return;
    NHANS_LAUNCH("avgpool", avgpool_kernel, dim3(B, C / 64), dim3(256), 0, s, x, HW, C, split, scale, out);
}

// ---------------------------------------------------------------------------------------------
// All 16 conditioning projections of a clip at once (process_noise_t_f, SN/main.py:139-148), with
// the BatchNorm scale/shift, conv bias and transform bias folded into Wc/base on the host.
// Block = 64 columns x 4 slices of K; every thread keeps 4 independent partial sums so that 4 loads
// are in flight, and the 16 partials of a column are added in a fixed order (deterministic, and the
// same for any batch).  One clip is 60 blocks instead of 15 threads-with-1024-serial-loads.
__global__ void __launch_bounds__(256) cond_kernel(const float* ea, const float* eb, const float* Wc,
                                                   const float* base, int ncols, float* cb) {
    __shared__ float e[2 * kEmb];
    __shared__ float part[4][64];
    const int clip = blockIdx.x;
    for (int i = threadIdx.x; i < kEmb; i += 256) {
        e[i] = ea[(size_t)clip * kEmb + i];
        e[kEmb + i] = eb[(size_t)clip * kEmb + i];
    }
    __syncthreads();
    const int col = threadIdx.x & 63, ks = threadIdx.x >> 6;
    const int n = blockIdx.y * 64 + col;
    float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
    if (n < ncols) {
        const int k0 = ks * (2 * kEmb / 4);
        const float* w = Wc + (size_t)k0 * ncols + n;
        for (int k = 0; k < 2 * kEmb / 4; k += 4) {
            a0 = fmaf(e[k0 + k], w[(size_t)k * ncols], a0);
            a1 = fmaf(e[k0 + k + 1], w[(size_t)(k + 1) * ncols], a1);
            a2 = fmaf(e[k0 + k + 2], w[(size_t)(k + 2) * ncols], a2);
            a3 = fmaf(e[k0 + k + 3], w[(size_t)(k + 3) * ncols], a3);
        }
    }
    part[ks][col] = (a0 + a1) + (a2 + a3);
    __syncthreads();
    if (ks == 0 && n < ncols)
        cb[(size_t)clip * ncols + n] = base[n] + ((part[0][col] + part[1][col]) + (part[2][col] + part[3][col]));
}

void launch_cond(const float* ea, const float* eb, int nclips, const float* Wc, const float* base, int ncols,
                 float* cb, hipStream_t s) {
    if (nclips <= 0) return;
    NHANS_LAUNCH("cond_proj", cond_kernel, dim3(nclips, (ncols + 63) / 64), dim3(256), 0, s, ea, eb, Wc, base, ncols, cb);
}

}  // namespace nhans
