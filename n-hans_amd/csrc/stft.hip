// STFT feature extraction and inverse STFT / overlap-add for N-HANS (16 kHz, 400-sample periodic
// Hann window, hop 160, 201 bins).
//
//   stft_features_kernel : tf.signal.stft(wav, 400, 160, fft_length=400) -> log(|X| + 1e-5), angle(X)
//                          (SN/apply.py:368-375, SN/reader.py:334-350)
//   istft_ola_kernel     : exp(logmag) * e^{j phase} -> tf.signal.inverse_stft(400, 160, 400,
//                          window_fn=inverse_stft_window_fn(160, hann periodic))   (SN/apply.py:189-204)
//
// Both are HBM-bound by their algorithmic bytes (2,248 per frame) and VALU-instruction-bound in practice.
// The 400-point transform is 20 x 20 (fft400.h): a lane computes one 20-point DFT in registers, and the 20 x 20
// transpose between the two passes goes through LDS rows padded to 21 complex values.  Complex values are native
// 2-vectors (packed-f32 math), global accesses are a wave-uniform base plus a 32-bit byte offset per lane, and
// workgroups are persistent: they walk the list of runs of the batch.
//   STFT: runs of 23 frames; the run's sample span is staged in LDS once (each sample crosses HBM once although
//   it is in 2.5 frames) and the NEXT run's samples are fetched into registers before the current run is
//   transformed.  The transform is REAL-input: pass 1 (460 tasks = frame x column, dealt to all 256 lanes)
//   computes rows 0..10 only, pass 2 (253 tasks = frame x row) runs complex DFTs on those and stores its outputs
//   k2 >= 10 as the conjugates of bins 400 - k; the bins are written straight from the pass-2 registers.
//   iSTFT: TWO frames share one complex transform (Z = A + iB), 3 transforms = 6 frames per wavefront, one pass
//   of 24 frames per run at two waves per SIMD; the next run's log-magnitudes and phases are fetched into
//   registers before the current run is transformed; the windowed frames land in an LDS buffer that aliases the
//   (wavefront-private) transform areas; overlap-add is in gather form (each output sample summed from its <= 3
//   frames in ascending order, four samples per lane: no atomics, bitwise deterministic).
// log/atan2/exp/sincos run on the hardware transcendental units (below).
#include "nhans_kernels.h"
#include "fft400.h"

namespace nhans {

constexpr int kFpw = 3;                 // iSTFT: transforms per wavefront
constexpr int kTRow = 21;               // padded transpose row (complex values)
constexpr int kTFrame = 20 * kTRow;     // 420 complex per frame
// ---- feature math on the hardware transcendental units.  Both kernels are VALU-bound once their loads
// are pipelined (PMC: 490 VALU wave-instructions per frame, two thirds of them libm's atan2f / logf /
// sincosf / expf with their full-range reductions); the ranges here are narrow and float32-level accuracy
// is all the path can use (phase 1.7e-7 rad, magnitudes ~2e-7 relative).
__device__ __forceinline__ float fast_atan2(float y, float x) {
    const float ax = fabsf(x), ay = fabsf(y);
    const float mx = fmaxf(ax, ay), mn = fminf(ax, ay);
    const float r = mx > 0.f ? mn * __builtin_amdgcn_rcpf(mx) : 0.f;     // atan2(0, 0) = 0 like libm
    const float t = r * r;
    // minimax fit of atan(r)/r in r^2 on [0, 1]: 1.7e-7 rad in float32 arithmetic
    float p = -0.004733146633952856f;
    p = fmaf(p, t, 0.024376805871725082f);
    p = fmaf(p, t, -0.0596301406621933f);
    p = fmaf(p, t, 0.09921535104513168f);
    p = fmaf(p, t, -0.140206977725029f);
    p = fmaf(p, t, 0.1996956467628479f);
    p = fmaf(p, t, -0.3333193361759186f);
    p = fmaf(p, t, 0.9999998807907104f);
    float a = p * r;
    a = ay > ax ? 1.57079632679489662f - a : a;
    a = x < 0.f ? 3.14159265358979324f - a : a;
    return copysignf(a, y);
}
__device__ __forceinline__ float fast_log(float v) {           // v >= 1e-5: v_log_f32 is log2, 1 ulp
    return __builtin_amdgcn_logf(v) * 0.69314718055994531f;
}
__device__ __forceinline__ float fast_exp(float x) {           // |x| < 60; the product x*log2(e) carried as hi + lo
    const float hi = x * 1.44269504088896341f;
    const float lo = fmaf(x, 1.44269504088896341f, -hi) + x * 1.925963033500e-8f;
    return __builtin_amdgcn_exp2f(hi) * fmaf(lo, 0.69314718055994531f, 1.0f);
}
__device__ __forceinline__ void fast_sincos(float a, float* sn, float* cs) {   // |a| <= pi: v_sin/v_cos take revolutions
    const float rev = a * 0.15915494309189535f;
    *sn = __builtin_amdgcn_sinf(rev);
    *cs = __builtin_amdgcn_cosf(rev);
}


// Global access as (wave-uniform 64-bit base) + (32-bit BYTE offset per lane): the form the scalar-base global
// instructions take directly; element indexing would make the compiler widen every offset to 64 bits first.
__device__ __forceinline__ float ldg32(const float* base, unsigned byte_off) {
    return *reinterpret_cast<const float*>(reinterpret_cast<const char*>(base) + byte_off);
}
__device__ __forceinline__ void stg32(float* base, unsigned byte_off, float v) {
    *reinterpret_cast<float*>(reinterpret_cast<char*>(base) + byte_off) = v;
}

// A wavefront's transpose / spectrum area is private to it: its lanes exchange data through it with no
// workgroup barrier, only "my LDS traffic has completed" (lanes of a wave run in lockstep).
__device__ __forceinline__ void wave_lds_sync() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// The ANALYSIS transform is one real frame per transform, but a REAL one: pass 1 computes rows k1 = 0..10 of the
// 20 x 20 decomposition from real columns (fft400.h: rdft20_half -- the conjugate-symmetric half, with the k1 = 3
// radix-5 branch never computed), pass 2 runs full complex 20-point DFTs on those 11 rows only, and a row's
// outputs k2 >= 10 are written as the conjugates of bins 400 - k (rows 1..9), which covers rows 11..19.  460
// pass-1 tasks and 253 pass-2 tasks per run of 23 frames, spread over the workgroup's 256 lanes.
// (Packing TWO frames into one complex transform, z = a + i b untangled as A = (Z[k] + conj Z[400-k]) / 2, was
// built and taken out: it leaks float32 rounding of the louder frame into the quieter one, log(|X| + 1e-5)
// amplifies that at silent bins -- 1.25e-4 instead of 7e-5 on the separator's context embeddings, past the 1e-4
// bar.  The inverse transform below does pack two frames: there the leak is 1e-7 of a waveform sample.)
constexpr int kRows = 11;                    // rows k1 = 0..10 of the transposed intermediate
constexpr int kTReal = kRows * kTRow;        // 231 complex per frame

__global__ void __launch_bounds__(256) stft_features_kernel(
    const float* __restrict__ wav, ClipTable tab, const int* __restrict__ block_clip,
    const int* __restrict__ block_f0, int nblocks, const cplx* __restrict__ tw400g, const float* __restrict__ windowg,
    float* __restrict__ logmag, float* __restrict__ phase) {
    constexpr int F = kStftFramesPerBlock;
    constexpr int SPAN = kWin + kHop * (F - 1);
    __shared__ __attribute__((aligned(16))) float xs[SPAN];
    __shared__ float win[kWin];
    __shared__ cplx tw[400];
    __shared__ cplx tbuf[F * kTReal];              // T[frame][k1][n2], rows padded to 21

    const int tid = threadIdx.x;
    constexpr int NPRE = (SPAN + 255) / 256;        // samples per thread of one run
    float pre[NPRE];
    auto fetch = [&](int blk) {                     // the run's samples -> registers (loads stay in flight)
        const int c = block_clip[blk];
        const int64_t base = tab.sample_off[c] + (int64_t)block_f0[blk] * kHop;
        const int left = (int)min((int64_t)SPAN, tab.sample_off[c + 1] - base);      // samples of the run inside the clip
        const float* __restrict__ wb = wav + base;                                   // uniform base + 32-bit lane offset
#pragma unroll
        for (int u = 0; u < NPRE; ++u) {
            const int i = u * 256 + tid;
            pre[u] = i < left ? ldg32(wb, (unsigned)i * 4u) : 0.f;
        }
    };
    for (int i = tid; i < 400; i += 256) { tw[i] = tw400g[i]; win[i] = windowg[i]; }
    fetch(blockIdx.x);

#pragma unroll 1
  for (int blk = blockIdx.x; blk < nblocks; blk += gridDim.x) {
    const int clip = block_clip[blk], f0 = block_f0[blk];
    const int64_t fr_beg = tab.frame_off[clip];
    const int T = (int)(tab.frame_off[clip + 1] - fr_beg);
#pragma unroll
    for (int u = 0; u < NPRE; ++u) {
        const int i = u * 256 + tid;
        if (i < SPAN) xs[i] = pre[u];
    }
    __syncthreads();                                // (also: every lane is through with the previous run's rows)
    if (blk + (int)gridDim.x < nblocks) fetch(blk + gridDim.x);

    // pass 1: task (frame f, column n2) = real 20-point DFT over n1 of the windowed samples x[20*n1 + n2]
#pragma unroll 1
    for (int id = tid; id < F * 20; id += 256) {
        const int f = id / 20, n2 = id - f * 20;
        float col[20];
        cplx y[kRows];
        const float* xf = xs + f * kHop + n2;
#pragma unroll
        for (int n1 = 0; n1 < 20; ++n1) col[n1] = xf[20 * n1] * win[20 * n1 + n2];
        fft400_pass1_real(col, n2, tw, y);
        cplx* tf = tbuf + f * kTReal + n2;
#pragma unroll
        for (int k1 = 0; k1 < kRows; ++k1) tf[k1 * kTRow] = y[k1];
    }
    __syncthreads();
    // pass 2: task (frame f, row k1 <= 10) = complex 20-point DFT over n2 -> X[k1 + 20*k2], k2 = 0..19
    if (tid < F * kRows) {
        const int f = tid / kRows, k1 = tid - f * kRows;
        cplx row[20], X[20];
        const cplx* tf = tbuf + f * kTReal + k1 * kTRow;
#pragma unroll
        for (int n2 = 0; n2 < 20; ++n2) row[n2] = tf[n2];
        fft400_pass2<false>(row, X);
        if (f0 + f < T) {
            float* __restrict__ lo_ = logmag + (fr_beg + f0 + f) * kBins;         // the frame's row (64-bit once per lane)
            float* __restrict__ po_ = phase ? phase + (fr_beg + f0 + f) * kBins : nullptr;
            // k2 < 10: bin k1 + 20*k2 as computed.  k2 >= 10: bin 400 - (k1 + 20*k2) as the conjugate (rows 1..9);
            // row 0 has the Nyquist bin 200 at k2 = 10 and nothing else there, row 10 nothing.
#pragma unroll
            for (int k2 = 0; k2 < 20; ++k2) {
                const bool mirror = k2 >= 10 && k1 != 0;
                const bool live = k2 < 10 || (k1 == 0 ? k2 == 10 : k1 != 10);
                if (!live) continue;
                const int bin = mirror ? 400 - k1 - 20 * k2 : k1 + 20 * k2;
                const cplx v = X[k2];
                const float mag = __builtin_amdgcn_sqrtf(v.x * v.x + v.y * v.y);
                lo_[bin] = fast_log(mag + 1e-5f);
                if (po_) po_[bin] = fast_atan2(mirror ? -v.y : v.y, v.x);
            }
        }
    }
  }
}

void launch_stft(const float* wav, ClipTable t, const int* block_clip, const int* block_f0, int nblocks,
                 const float* tw400, const float* window, float* logmag, float* phase, hipStream_t s) {
    if (nblocks <= 0) return;
    constexpr int wg_per_cu = kStftFramesPerBlock <= 19 ? 3 : 2;            // resident workgroups (LDS: 53 KB at 19 frames)
    const int grid = nblocks < (256 * wg_per_cu) ? nblocks : (256 * wg_per_cu);
    NHANS_LAUNCH("stft_features", stft_features_kernel, dim3(grid), dim3(256), 0, s, wav, t, block_clip, block_f0, nblocks,
                 reinterpret_cast<const cplx*>(tw400), window, logmag, phase);
}

// ---------------------------------------------------------------------------------------------
// Inverse: two frames per complex transform as well.  With A, B the (Hermitian) spectra of frames a and b,
// Z = A + iB has the inverse transform z = a + i b: real part = frame a, imaginary part = frame b.  The DC and
// Nyquist bins enter with their imaginary parts dropped, which is what a real inverse FFT does with them.
__global__ void __launch_bounds__(256, 2) istft_ola_kernel(
    const float* __restrict__ logmag, const float* __restrict__ phase, ClipTable tab,
    const int* __restrict__ block_clip, const int* __restrict__ block_h0, int nblocks, const cplx* __restrict__ tw400g,
    const float* __restrict__ wsyng, float* __restrict__ wav_out) {
    constexpr int HB = kIstftHopsPerBlock;          // output hops per block
    constexpr int F = HB + 2;                        // frames needed: h0-2 .. h0+HB-1
    static_assert(F == 4 * kFpw * 2, "24 frames = 4 waves x 3 transforms x 2 frames");
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    cplx* tw = reinterpret_cast<cplx*>(smem_raw);                 // 400
    float* wsyn = reinterpret_cast<float*>(tw + 400);             // 400
    cplx* area = reinterpret_cast<cplx*>(wsyn + kWin);            // per wave 3 x 420: spectra Z, then transpose rows
    float* y = reinterpret_cast<float*>(area);                    // [F][400] windowed frames, aliasing the areas
    static_assert((size_t)F * kWin * sizeof(float) <= (size_t)4 * kFpw * kTFrame * sizeof(cplx), "y fits the areas");

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int j = lane / 20, q = lane - j * 20;
    const bool active = lane < 60;
    cplx* a_wave = area + wave * kFpw * kTFrame;
    const int lf0 = wave * kFpw * 2;                 // first local frame of this wave; clip frame = h0 - 2 + lf

    constexpr int NIN = (kFpw * kBins + 63) / 64;    // (transform, bin) items per lane
    float la[NIN], pa[NIN], lb[NIN], pb[NIN];        // log-magnitude / phase of frames a and b of the NEXT run
    unsigned va = 0, vb = 0;
    auto fetch = [&](int blk) {
        const int c = block_clip[blk];
        const int64_t fb_ = tab.frame_off[c];
        const int Tc = (int)(tab.frame_off[c + 1] - fb_);
        const int fr0 = block_h0[blk] - 2 + lf0;
        // wave-uniform 64-bit base of the clip + 32-bit element offsets per lane (a clip is far below 2^31 / 201
        // frames): the loads take the scalar-base form instead of a 64-bit address computation per element
        const float* __restrict__ lmc = logmag + fb_ * kBins;
        const float* __restrict__ phc = phase + fb_ * kBins;
        va = 0; vb = 0;
#pragma unroll
        for (int u = 0; u < NIN; ++u) {
            const int idx = u * 64 + lane;
            const int tr = idx / kBins, k = idx - tr * kBins;
            const int fa = fr0 + 2 * tr;
            const bool oka = idx < kFpw * kBins && fa >= 0 && fa < Tc;
            const bool okb = idx < kFpw * kBins && fa + 1 >= 0 && fa + 1 < Tc;
            const unsigned oa = oka ? (unsigned)(fa * kBins + k) * 4u : 0u, ob = okb ? (unsigned)((fa + 1) * kBins + k) * 4u : 0u;
            la[u] = ldg32(lmc, oa); pa[u] = ldg32(phc, oa);
            lb[u] = ldg32(lmc, ob); pb[u] = ldg32(phc, ob);
            va |= oka ? 1u << u : 0u;
            vb |= okb ? 1u << u : 0u;
        }
    };

    for (int i = tid; i < 400; i += 256) { tw[i] = tw400g[i]; wsyn[i] = wsyng[i]; }
    fetch(blockIdx.x);
    __syncthreads();

#pragma unroll 1
    for (int blk = blockIdx.x; blk < nblocks; blk += gridDim.x) {
        const int clip = block_clip[blk], h0 = block_h0[blk];
        const int64_t fr_beg = tab.frame_off[clip];
        const int T = (int)(tab.frame_off[clip + 1] - fr_beg);
        const int nout = (T - 1) * kHop + kWin;
        // Z = A + iB over all 400 bins from the 201 given ones (frames outside [0, T) contribute zero)
#pragma unroll
        for (int u = 0; u < NIN; ++u) {
            const int idx = u * 64 + lane;
            const int tr = idx / kBins, k = idx - tr * kBins;
            float sn, cs;
            fast_sincos(pa[u], &sn, &cs);
            const float ma = ((va >> u) & 1u) ? fast_exp(la[u]) : 0.f;
            float are = ma * cs, aim = ma * sn;
            fast_sincos(pb[u], &sn, &cs);
            const float mb = ((vb >> u) & 1u) ? fast_exp(lb[u]) : 0.f;
            float bre = mb * cs, bim = mb * sn;
            const bool edge = k == 0 || k == 200;
            if (edge) { aim = 0.f; bim = 0.f; }
            if (idx < kFpw * kBins) {
                cplx* z = a_wave + tr * 400;
                z[k] = cmake(are - bim, aim + bre);
                if (!edge) z[400 - k] = cmake(are + bim, bre - aim);
            }
        }
        if (blk + (int)gridDim.x < nblocks) fetch(blk + gridDim.x);
        wave_lds_sync();
        cplx col[20];
        if (active) {
            const cplx* z = a_wave + j * 400;
#pragma unroll
            for (int a = 0; a < 20; ++a) col[a] = z[20 * a + q];
        }
        wave_lds_sync();                             // every lane holds its column: the area becomes the transpose buffer
        if (active) {
            cplx yv[20];
            fft400_pass1<true>(col, q, tw, yv);
            cplx* tf = a_wave + j * kTFrame;
#pragma unroll
            for (int c1 = 0; c1 < 20; ++c1) tf[c1 * kTRow + q] = yv[c1];
        }
        wave_lds_sync();
        cplx x[20];
        if (active) {
            cplx row[20];
            const cplx* tf = a_wave + j * kTFrame + q * kTRow;
#pragma unroll
            for (int b = 0; b < 20; ++b) row[b] = tf[b];
            fft400_pass2<true>(row, x);
        }
        __syncthreads();                             // every wave is through with its area: it becomes y
        if (active) {
            float* ya = y + (lf0 + 2 * j) * kWin;
#pragma unroll
            for (int c2 = 0; c2 < 20; ++c2) {
                const int n = q + 20 * c2;
                const float w = (1.0f / 400.0f) * wsyn[n];
                ya[n] = x[c2].x * w;                 // real part: frame a
                ya[kWin + n] = x[c2].y * w;          // imaginary part: frame b
            }
        }
        __syncthreads();
        // gather-form overlap-add: sample i of hop u sums frames u-2, u-1, u (ascending)
        float* __restrict__ wo = wav_out + tab.out_off[clip];    // uniform base + 32-bit sample position within the clip
        if ((reinterpret_cast<uintptr_t>(wo) & 15) == 0) {
            // four consecutive samples per lane (hop, window and the 80-sample head are multiples of 4, and a clip's
            // length is too): 16-byte LDS reads and global stores, the same three-term sum per sample
            typedef float f32x4v __attribute__((ext_vector_type(4)));
            for (int i = tid * 4; i < HB * kHop; i += 1024) {
                const int hl = i / kHop, r = i - hl * kHop;
                const int pos = (h0 + hl) * kHop + r;
                if (pos >= nout) break;
                f32x4v acc = {0.f, 0.f, 0.f, 0.f};
                if (r < kWin - 2 * kHop) acc = *reinterpret_cast<const f32x4v*>(y + hl * kWin + 2 * kHop + r);
                acc += *reinterpret_cast<const f32x4v*>(y + (hl + 1) * kWin + kHop + r);
                acc += *reinterpret_cast<const f32x4v*>(y + (hl + 2) * kWin + r);
                *reinterpret_cast<f32x4v*>(reinterpret_cast<char*>(wo) + (unsigned)pos * 4u) = acc;
            }
        } else {                                     // an output offset the caller did not align to 16 bytes
            for (int i = tid; i < HB * kHop; i += 256) {
                const int hl = i / kHop, r = i - hl * kHop;
                const int pos = (h0 + hl) * kHop + r;
                if (pos >= nout) break;
                float acc = 0.f;
                if (r < kWin - 2 * kHop) acc = y[hl * kWin + 2 * kHop + r];
                acc += y[(hl + 1) * kWin + kHop + r];
                acc += y[(hl + 2) * kWin + r];
                stg32(wo, (unsigned)pos * 4u, acc);
            }
        }
        __syncthreads();                             // y has been read: the areas may be rewritten
    }
}

void launch_istft(const float* logmag, const float* phase, ClipTable t, const int* block_clip,
                  const int* block_h0, int nblocks, const float* tw400, const float* wsyn, float* wav_out,
                  hipStream_t s) {
    if (nblocks <= 0) return;
    constexpr size_t lds = (400 + 4 * kFpw * kTFrame) * sizeof(cplx) + kWin * sizeof(float);     // 45 KB
    static unsigned long long attr_devices = 0;
    set_max_dynamic_lds(reinterpret_cast<const void*>(&istft_ola_kernel), lds, &attr_devices, "istft_ola");
    const int grid = nblocks < 256 * 2 ? nblocks : 256 * 2;     // two resident workgroups per CU (256 registers per lane; LDS would allow three)
    NHANS_LAUNCH("istft_ola", istft_ola_kernel, dim3(grid), dim3(256), lds, s, logmag, phase, t, block_clip,
                 block_h0, nblocks, reinterpret_cast<const cplx*>(tw400), wsyn, wav_out);
}

}  // namespace nhans
