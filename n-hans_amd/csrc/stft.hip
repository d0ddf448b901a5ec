// STFT feature extraction and inverse STFT / overlap-add for N-HANS (16 kHz, 400-sample periodic
// Hann window, hop 160, 201 bins).
//
//   stft_features_kernel : tf.signal.stft(wav, 400, 160, fft_length=400) -> log(|X| + 1e-5), angle(X)
//                          (SN/apply.py:368-375, SN/reader.py:334-350)
//   istft_ola_kernel     : exp(logmag) * e^{j phase} -> tf.signal.inverse_stft(400, 160, 400,
//                          window_fn=inverse_stft_window_fn(160, hann periodic))   (SN/apply.py:189-204)
//
// Both are HBM-bound by their algorithmic bytes (2,248 per frame) and VALU-instruction-bound in practice.
// The 400-point transform is 20 x 20 (fft400.h): every lane computes one 20-point DFT in registers, and the
// 20x20 transpose between the two passes goes through a wavefront-private LDS area (rows padded to 21 complex
// values), synchronised at wavefront level only.  TWO REAL FRAMES SHARE ONE COMPLEX TRANSFORM (z = a + i b), so
// a wavefront carries 3 transforms = 6 frames (60 of 64 lanes busy) and a workgroup 24 frames per pass.
// Workgroups are persistent and walk the list of 24-frame runs of the batch.
//   STFT: a lane's 2 x 20 samples come straight from global memory (20 lanes read 80 contiguous bytes; the overlap
//   of neighbouring frames is served by the caches); bins are untangled (A = (Z[k] + conj Z[400-k]) / 2, ...) and
//   turned into log-magnitude / phase by all 64 lanes, 201 contiguous floats per frame and array.
//   iSTFT: the next run's log-magnitudes and phases are fetched into registers before the current run is
//   transformed; Z = A + iB is built for all 400 bins in LDS; the windowed frames land in an LDS buffer that
//   aliases the transform areas; overlap-add is in gather form (each output sample summed by one thread from
//   its <= 3 frames in ascending order: no atomics, bitwise deterministic).
// log/atan2/exp/sincos run on the hardware transcendental units (below).
#include "nhans_kernels.h"
#include "fft400.h"

namespace nhans {

constexpr int kFpw = 3;                 // frames per wavefront per pass
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


// A wavefront's transpose / spectrum area is private to it: its lanes exchange data through it with no
// workgroup barrier, only "my LDS traffic has completed" (lanes of a wave run in lockstep).
__device__ __forceinline__ void wave_lds_sync() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// Two real frames per complex transform: z = a + i b  ->  Z = FFT(z);  A[k] = (Z[k] + conj Z[400-k]) / 2,
// B[k] = (Z[k] - conj Z[400-k]) / 2i.  A wavefront carries 3 transforms = 6 frames, a workgroup 24 frames in
// ONE pass (round 1: one frame per transform, two passes, five workgroup barriers per pass).
__global__ void __launch_bounds__(256) stft_features_kernel(
    const float* __restrict__ wav, ClipTable tab, const int* __restrict__ block_clip,
    const int* __restrict__ block_f0, int nblocks, const cplx* __restrict__ tw400g, const float* __restrict__ windowg,
    float* __restrict__ logmag, float* __restrict__ phase) {
    constexpr int F = kStftFramesPerBlock;
    static_assert(F == 4 * kFpw * 2, "24 frames = 4 waves x 3 transforms x 2 frames");
    __shared__ float win[kWin];
    __shared__ cplx tw[400];
    __shared__ cplx tbuf[4 * kFpw * kTFrame];      // per wave: 3 x 420 transpose rows, later 3 x 400 spectrum values

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int j = lane / 20, q = lane - j * 20;     // transform slot within the wave, DFT column/row
    const bool active = lane < 60;
    cplx* tw_wave = tbuf + wave * kFpw * kTFrame;
    const int lf0 = wave * kFpw * 2;                // first local frame of this wave; transform j: frames lf0+2j, +1

    // The samples of a lane's column (x[20*n1 + n2] of frame a and of frame b) come straight from global memory
    // into registers -- 20 lanes read 80 contiguous bytes, neighbouring frames overlap in the L1/L2.  No sample
    // staging in LDS, so no workgroup barrier in the loop and three workgroups per CU (45 KB of LDS each); their
    // waves hide each other's load latency (a register prefetch of the next run costs the third wave per SIMD).
    float nx[40];
    auto fetch = [&](int blk) {
        const int c = block_clip[blk];
        const int64_t base = tab.sample_off[c] + (int64_t)(block_f0[blk] + lf0 + 2 * j) * kHop + q, end = tab.sample_off[c + 1];
#pragma unroll
        for (int n1 = 0; n1 < 20; ++n1) {
            const int64_t ia = base + 20 * n1, ib = ia + kHop;
            nx[2 * n1] = (active && ia < end) ? wav[ia] : 0.f;
            nx[2 * n1 + 1] = (active && ib < end) ? wav[ib] : 0.f;
        }
    };
    for (int i = tid; i < 400; i += 256) { tw[i] = tw400g[i]; win[i] = windowg[i]; }
    __syncthreads();

#pragma unroll 1
    for (int blk = blockIdx.x; blk < nblocks; blk += gridDim.x) {
        const int clip = block_clip[blk], f0 = block_f0[blk];
        const int64_t fr_beg = tab.frame_off[clip];
        const int T = (int)(tab.frame_off[clip + 1] - fr_beg);
        fetch(blk);
        cplx col[20];
#pragma unroll
        for (int n1 = 0; n1 < 20; ++n1) {
            const float w = win[20 * n1 + q];
            col[n1] = cmake(nx[2 * n1] * w, nx[2 * n1 + 1] * w);
        }
        if (active) {
            // pass 1: lane = n2; 20-point DFT over n1 of z[20*n1 + n2] = (frame a, frame b) windowed
            cplx y[20];
            fft400_pass1<false>(col, q, tw, y);
            cplx* tf = tw_wave + j * kTFrame;
#pragma unroll
            for (int k1 = 0; k1 < 20; ++k1) tf[k1 * kTRow + q] = y[k1];
        }
        wave_lds_sync();
        cplx Z[20];
        if (active) {
            // pass 2: lane = k1; 20-point DFT over n2 -> Z[k1 + 20*k2]
            cplx row[20];
            const cplx* tf = tw_wave + j * kTFrame + q * kTRow;
#pragma unroll
            for (int n2 = 0; n2 < 20; ++n2) row[n2] = tf[n2];
            fft400_pass2<false>(row, Z);
        }
        wave_lds_sync();                            // every lane has its row: the area is free again
        if (active) {
            cplx* sp = tw_wave + j * 400;           // [3][400] spectra of the wave's transforms
#pragma unroll
            for (int k2 = 0; k2 < 20; ++k2) sp[q + 20 * k2] = Z[k2];
        }
        wave_lds_sync();
        {
            // untangle + features: item = (transform, bin 0..200) -> bins of frame a and of frame b
            const int fw0 = f0 + lf0;               // clip-relative frame of the wave's first frame
            for (int idx = lane; idx < kFpw * kBins; idx += 64) {
                const int tr = idx / kBins, k = idx - tr * kBins;
                const int fa = fw0 + 2 * tr;
                if (fa >= T) break;
                const cplx zk = tw_wave[tr * 400 + k];
                const cplx zm = tw_wave[tr * 400 + (k == 0 ? 0 : 400 - k)];
                const float are = 0.5f * (zk.x + zm.x), aim = 0.5f * (zk.y - zm.y);
                const float bre = 0.5f * (zk.y + zm.y), bim = -0.5f * (zk.x - zm.x);
                const int64_t o = (fr_beg + fa) * kBins + k;
                logmag[o] = fast_log(__builtin_amdgcn_sqrtf(are * are + aim * aim) + 1e-5f);
                if (phase) phase[o] = fast_atan2(aim, are);
                if (fa + 1 < T) {
                    logmag[o + kBins] = fast_log(__builtin_amdgcn_sqrtf(bre * bre + bim * bim) + 1e-5f);
                    if (phase) phase[o + kBins] = fast_atan2(bim, bre);
                }
            }
        }
        wave_lds_sync();                            // the spectra have been read before the next run's pass 1 writes
    }
}

void launch_stft(const float* wav, ClipTable t, const int* block_clip, const int* block_f0, int nblocks,
                 const float* tw400, const float* window, float* logmag, float* phase, hipStream_t s) {
    if (nblocks <= 0) return;
    const int grid = nblocks < 256 * 3 ? nblocks : 256 * 3;     // three resident workgroups per CU
    NHANS_LAUNCH("stft_features", stft_features_kernel, dim3(grid), dim3(256), 0, s, wav, t, block_clip, block_f0, nblocks,
                 reinterpret_cast<const cplx*>(tw400), window, logmag, phase);
}

// ---------------------------------------------------------------------------------------------
// Inverse: two frames per complex transform as well.  With A, B the (Hermitian) spectra of frames a and b,
// Z = A + iB has the inverse transform z = a + i b: real part = frame a, imaginary part = frame b.  The DC and
// Nyquist bins enter with their imaginary parts dropped, which is what a real inverse FFT does with them.
__global__ void __launch_bounds__(256) istft_ola_kernel(
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
        va = 0; vb = 0;
#pragma unroll
        for (int u = 0; u < NIN; ++u) {
            const int idx = u * 64 + lane;
            const int tr = idx / kBins, k = idx - tr * kBins;
            const int fa = fr0 + 2 * tr;
            const bool oka = idx < kFpw * kBins && fa >= 0 && fa < Tc;
            const bool okb = idx < kFpw * kBins && fa + 1 >= 0 && fa + 1 < Tc;
            const int64_t oa = oka ? (fb_ + fa) * kBins + k : 0, ob = okb ? (fb_ + fa + 1) * kBins + k : 0;
            la[u] = logmag[oa]; pa[u] = phase[oa];
            lb[u] = logmag[ob]; pb[u] = phase[ob];
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
        const int64_t nout = (int64_t)(T - 1) * kHop + kWin;
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
        const int64_t obase = tab.out_off[clip];
        for (int i = tid; i < HB * kHop; i += 256) {
            const int hl = i / kHop, r = i - hl * kHop;
            const int64_t pos = (int64_t)(h0 + hl) * kHop + r;
            if (pos >= nout) break;
            float acc = 0.f;
            if (r < kWin - 2 * kHop) acc = y[hl * kWin + 2 * kHop + r];
            acc += y[(hl + 1) * kWin + kHop + r];
            acc += y[(hl + 2) * kWin + r];
            wav_out[obase + pos] = acc;
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
    const int grid = nblocks < 256 * 3 ? nblocks : 256 * 3;     // three resident workgroups per CU (LDS), two by registers
    NHANS_LAUNCH("istft_ola", istft_ola_kernel, dim3(grid), dim3(256), lds, s, logmag, phase, t, block_clip,
                 block_h0, nblocks, reinterpret_cast<const cplx*>(tw400), wsyn, wav_out);
}

}  // namespace nhans
