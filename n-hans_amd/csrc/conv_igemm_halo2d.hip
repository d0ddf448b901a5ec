// Implicit-GEMM convolution, 2-D halo variant of conv_igemm_halo.hip for the layers whose launches are
// bound by the memory pipe rather than the matrix pipe: the stride-1 k x k convs with 64 output
// channels (resblock1: 38 % of a step).  Same contract, packed weights, MFMA arrangement, producer /
// consumer wave split, register pipelining and epilogue; different pixel tile.
//
// conv_igemm_halo.hip tiles the flattened pixel run (256 consecutive pixels ~ 1.3 image rows of a
// 35 x 201 map) and stages one halo image per (filter row, channel chunk): every input pixel of the
// tile crosses the CU's memory pipe KH times (+25 % halo).  Measured there (tools/halo_phase_cycles.py):
// a tap of the N = 64 layers costs 800 cycles of DMA alone against 768 of MFMAs, and the two add up.
// Here the workgroup owns a th x tw rectangle of ONE image (5 x 51 for 35 x 201: 98 % of the 256 tile
// slots used) and stages, per 32-channel chunk, the (th+KH-1) x (tw+KW-1) input rectangle once; all
// KH*KW taps read it at a row offset of kh*(tw+KW-1) + kw.  Activation bytes per tile: 2 x 55 KB
// instead of 8 x 40 KB.  Padding rows / columns come from the zero page as before; slots of the tile
// that fall outside the image compute on whatever the rectangle holds and are dropped by the epilogue.
//
//   LDS: 2 image buffers x HR2 rows x 128 B + 4 weight stages x BN x 128 B (448 rows, BN 64: 144 KB).
//   K order: channel chunk, filter row, filter column -- the order the packed weights are stored in
//   (fold.py kmat) since round 3.
//   One super-chunk = KH*KW >= 9 taps, so the standard counted waits of the halo kernel apply.
//
// Eligibility (launcher): one segment, stride 1, SAME padding, 64 output channels per tile column
// (N % 128 != 0), a tile shape with >= 93 % slot use whose rectangle fits HR2 rows, < 2^31 elements.
#include "conv_epilogue.h"
#include <cstdlib>

namespace nhans {

namespace {
constexpr int T2_SLOTS = 256;  // tile slots (MFMA rows) per workgroup
constexpr int T2_BST = 4;      // weight ring stages
constexpr int T2_NCW = 8;      // consumer (MFMA) waves
constexpr int T2_NPW = 4;      // producer (DMA) waves

template <int N> __device__ __forceinline__ void t2_wait_vmcnt() {
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

// scheduling pattern of a consumer half: one MFMA, then the LDS reads' even share (0x008 MFMA, 0x100 DS read)
template <int I, int NM, int ND> __device__ __forceinline__ void t2_interleave() {
    if constexpr (I < NM) {
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
        constexpr int nd = (I + 1) * ND / NM - I * ND / NM;
        if constexpr (nd > 0) __builtin_amdgcn_sched_group_barrier(0x100, nd, 0);
        t2_interleave<I + 1, NM, ND>();
    }
}
}  // namespace

template <int BN, int PREC, int HR2, int ABL = 0>   // ABL 1: no per-tap barrier (timing experiment, wrong results)
__global__ void __launch_bounds__((T2_NCW + T2_NPW) * 64) conv_igemm_halo2d(const ConvArgs a) {
    constexpr int TM = 2;
    constexpr int TN = BN / 64;
    constexpr int A_BUF = HR2 * 32;                    // floats
    constexpr int B_STAGE = 32 * BN;                   // floats
    constexpr int B_BASE = 2 * A_BUF;
    constexpr int GBP = B_STAGE / 4 / (T2_NPW * 64);   // weight DMA instructions per producer thread per tap
    constexpr int NAP = HR2 * 8 / (T2_NPW * 64);       // activation DMA instructions per producer thread per image
    extern __shared__ __attribute__((aligned(16))) float smem[];

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);

    // XCD-aware, bijective remap of the linear workgroup id: an XCD owns a contiguous range of tiles,
    // i.e. neighbouring rectangles of the same images, whose halos overlap in its L2
    const int ntn = a.N / BN;
    int L;
    {
        const int nblk = gridDim.x, bid = blockIdx.x;
        const int q = nblk >> 3, r = nblk & 7, xcd = bid & 7, idx = bid >> 3;
        L = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
    }
    const int mt = L / ntn, nt = L - mt * ntn;
    const int nt0 = nt * (BN / 32);
    const int th = a.t2_th, tw = a.t2_tw;
    const int tpi = a.t2_ntr * a.t2_ntc;
    const int img = mt / tpi;
    const int tq = mt - img * tpi;
    const int tr = tq / a.t2_ntc;
    const int r0 = tr * th, c0 = (tq - tr * a.t2_ntc) * tw;

    const ConvSeg& g = a.seg[0];
    const int KW0 = g.KW, KH0 = g.KH, CC0 = g.C >> 5;
    const int IW = tw + KW0 - 1, IH = th + KH0 - 1;    // the staged rectangle
    const int total = CC0 * KH0 * KW0;                 // taps = weight chunks

    // cursor of the current tap: filter column, filter row, image buffer (one image per channel chunk)
    int kwC = 0, khC = 0, bufC = 0;
#define NH_NEXT_TAP()                                                                              \
    if (++kwC >= KW0) {                                                                            \
        kwC = 0;                                                                                   \
        if (++khC >= KH0) {                                                                        \
            khC = 0;                                                                               \
            bufC ^= 1;                                                                             \
        }                                                                                          \
    }

    if (wave >= T2_NCW) {
        // =========================================================================================
        // Producer waves (see conv_igemm_halo.hip for the protocol).  Queue order per iteration j:
        // [image of the next chunk if tap j opens one], weights of tap j+3.
        const int pw = wave - T2_NCW, ptid = tid - T2_NCW * 64;
        __builtin_amdgcn_s_setprio(3);
        const int slot = lane & 7;
        const size_t bstride = (size_t)(a.N / 32) * 1024;
        // LDS row j of the image = input pixel (r0 - pt + j / IW, c0 - pl + j % IW); 8 lanes per row,
        // lane slot s fetches source piece s ^ ((j>>1)&7).  Element offset for chunk 0, or -1 for
        // padding / rows past the rectangle (zero page).
        int poff[NAP];
#pragma unroll
        for (int d = 0; d < NAP; ++d) {
            const int j = d * 32 + pw * 8 + (lane >> 3);
            const int sp = (slot ^ ((j >> 1) & 7)) * 4;
            const int ir = j / IW, ic = j - ir * IW;
            const int hi = r0 + ir - g.pt, wi = c0 + ic - g.pl;
            const bool ok = ir < IH && (unsigned)hi < (unsigned)g.H && (unsigned)wi < (unsigned)g.W;
            poff[d] = ok ? ((img * g.H + hi) * g.W + wi) * g.C + sp : -1;
        }
#define NH_GLDS(SRC, DST)                                                                          \
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(SRC),        \
                                     (__attribute__((address_space(3))) void*)(DST), 16, 0, 0);
        int ccA = 0;                                    // next channel chunk to stage
        const float* const zp = a.zero + (slot ^ ((lane >> 4) & 7)) * 4;
#define NH_ISSUE_A(BUF)                                                                            \
    {                                                                                              \
        const bool live_ = ccA < CC0;                                                              \
        float* sa_ = smem + (BUF) * A_BUF + pw * 8 * 32;                                           \
        _Pragma("unroll") for (int d = 0; d < NAP; ++d) {                                          \
            const float* p_ = (live_ && poff[d] >= 0) ? g.src + (poff[d] + ccA * 32) : zp;         \
            NH_GLDS(p_, sa_ + d * 32 * 32)                                                         \
        }                                                                                          \
        ++ccA;                                                                                     \
    }
        // weights of the taps in the order (chunk, row, column) -- the order fold.py kmat() packs them in
        int kwB = 0, khB = 0, ccB = 0, tapB = 0;
        const float* const wbase = g.wpk + (size_t)nt0 * 1024;
        const float* bp_ = wbase;
#define NH_ISSUE_B(ST)                                                                             \
    {                                                                                              \
        if (tapB < total) {                                                                        \
            bp_ = wbase + (size_t)((ccB * KH0 + khB) * KW0 + kwB) * bstride;                       \
            ++tapB;                                                                                \
            if (++kwB >= KW0) {                                                                    \
                kwB = 0;                                                                           \
                if (++khB >= KH0) { khB = 0; ++ccB; }                                              \
            }                                                                                      \
        }                                                                                          \
        float* sb_ = smem + B_BASE + (ST) * B_STAGE;                                               \
        _Pragma("unroll") for (int j = 0; j < GBP; ++j)                                            \
            NH_GLDS(bp_ + (j * (T2_NPW * 64) + ptid) * 4, sb_ + (j * (T2_NPW * 64) + pw * 64) * 4) \
    }

        NH_ISSUE_A(0)
        NH_ISSUE_B(0)
        NH_ISSUE_B(1)
        NH_ISSUE_B(2)
        t2_wait_vmcnt<2 * GBP>();                       // image 0 and tap 0
        __builtin_amdgcn_s_barrier();
        bool prev_first = false;
        for (int it = 0; it < total; ++it) {
            const bool first = kwC == 0 && khC == 0;
            if (first) NH_ISSUE_A(bufC ^ 1)
            NH_ISSUE_B((it + 3) & (T2_BST - 1))
            if (first || prev_first) t2_wait_vmcnt<2 * GBP + NAP>();
            else t2_wait_vmcnt<2 * GBP>();
            if constexpr (!(ABL & 1)) __builtin_amdgcn_s_barrier();
            prev_first = first;
            NH_NEXT_TAP()
        }
        t2_wait_vmcnt<0>();                             // dummy DMAs past the end still target LDS
        __builtin_amdgcn_s_barrier();
        return;                                         // the epilogue's barriers count live waves only
#undef NH_GLDS
#undef NH_ISSUE_A
#undef NH_ISSUE_B
    }

    // =============================================================================================
    // Consumer waves.  Tile slot p = wm*64 + t*32 + (lane&31) is pixel (p / tw, p % tw) of the
    // rectangle; for tap (kh, kw) it reads image row (p/tw + kh) * IW + p%tw + kw.
    const int wm = wave >> 1, wn = wave & 1;
    const int g8 = lane >> 5;
    int jb[TM];
#pragma unroll
    for (int t = 0; t < TM; ++t) {
        int p = wm * 64 + t * 32 + (lane & 31);
        p = p < th * tw ? p : th * tw - 1;              // spare slots read a valid row, their result is dropped
        const int r = p / tw;
        jb[t] = r * IW + (p - r * tw);
    }
    const int bcol = (wn * TN) * 1024 + lane * 4;

    constexpr int KS = PREC == 1 ? 2 : 4;              // k-steps per chunk (16 k each / 8 k each)
    constexpr int KH_ = KS / 2;
    f32x4 fa_hi[KS][TM], fa_lo[PREC == 1 ? KS : 1][TM], fb_hi[KS][TN], fb_lo[PREC == 1 ? KS : 1][TN];
#define NH_READ_HALF(H, STG)                                                                       \
    {                                                                                              \
        const float* Sa_ = smem + bufC * A_BUF;                                                    \
        const float* Sb_ = smem + B_BASE + (STG) * B_STAGE + bcol;                                 \
        const int tapoff_ = khC * IW + kwC;                                                        \
        _Pragma("unroll") for (int t = 0; t < TM; ++t) {                                           \
            const int jr_ = jb[t] + tapoff_;                                                       \
            const float* ar_ = Sa_ + jr_ * 32;                                                     \
            const int rs_ = (jr_ >> 1) & 7;                                                        \
            _Pragma("unroll") for (int s = (H) * KH_; s < ((H) + 1) * KH_; ++s) {                  \
                fa_hi[s][t] = *reinterpret_cast<const f32x4*>(ar_ + (((2 * s + g8) ^ rs_) * 4));   \
                if constexpr (PREC == 1)                                                           \
                    fa_lo[s][t] = *reinterpret_cast<const f32x4*>(ar_ + (((2 * s + g8 + 4) ^ rs_) * 4)); \
            }                                                                                      \
        }                                                                                          \
        _Pragma("unroll") for (int s = (H) * KH_; s < ((H) + 1) * KH_; ++s)                        \
            _Pragma("unroll") for (int j = 0; j < TN; ++j) {                                       \
                if constexpr (PREC == 1) {                                                         \
                    fb_hi[s][j] = *reinterpret_cast<const f32x4*>(Sb_ + j * 1024 + s * 512);       \
                    fb_lo[s][j] = *reinterpret_cast<const f32x4*>(Sb_ + j * 1024 + s * 512 + 256); \
                } else {                                                                           \
                    fb_hi[s][j] = *reinterpret_cast<const f32x4*>(Sb_ + j * 1024 + s * 256);       \
                }                                                                                  \
            }                                                                                      \
    }
#define NH_MFMA_HALF(H)                                                                            \
    {                                                                                              \
        _Pragma("unroll") for (int s = (H) * KH_; s < ((H) + 1) * KH_; ++s)                        \
            _Pragma("unroll") for (int p = 0; p < (PREC == 1 ? 3 : 4); ++p)                        \
                _Pragma("unroll") for (int t = 0; t < TM; ++t)                                     \
                    _Pragma("unroll") for (int j = 0; j < TN; ++j) {                               \
                        if constexpr (PREC == 1) {                                                 \
                            const f16x8 a_ = __builtin_bit_cast(f16x8, p == 0 ? fa_lo[s][t] : fa_hi[s][t]); \
                            const f16x8 b_ = __builtin_bit_cast(f16x8, p == 1 ? fb_lo[s][j] : fb_hi[s][j]); \
                            acc[t][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(b_, a_, acc[t][j], 0, 0, 0); \
                        } else {                                                                   \
                            acc[t][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(fb_hi[s][j][p], fa_hi[s][t][p], acc[t][j], 0, 0, 0); \
                        }                                                                          \
                    }                                                                              \
    }

    f32x16 acc[TM][TN];
#pragma unroll
    for (int t = 0; t < TM; ++t)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[t][j][r] = 0.f;

    constexpr int NM = KH_ * TM * TN * (PREC == 1 ? 3 : 4);         // MFMAs per half
    constexpr int ND = KH_ * (TM + TN) * (PREC == 1 ? 2 : 1);       // ds_read_b128 per half
    __builtin_amdgcn_s_barrier();                       // image 0 and tap 0 have landed
    NH_READ_HALF(0, 0)
    for (int it = 0; it < total; ++it) {
        __builtin_amdgcn_sched_barrier(0);
        NH_READ_HALF(1, it & (T2_BST - 1))
        if constexpr (!(ABL & 2)) __builtin_amdgcn_sched_barrier(0);
        NH_MFMA_HALF(0)
        if constexpr ((ABL & 2) != 0) t2_interleave<0, NM, ND>();
        __builtin_amdgcn_sched_barrier(0);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        if constexpr (!(ABL & 1)) __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
        NH_NEXT_TAP()
        NH_READ_HALF(0, (it + 1) & (T2_BST - 1))        // (past the last tap: a harmless read)
        if constexpr (!(ABL & 2)) __builtin_amdgcn_sched_barrier(0);
        NH_MFMA_HALF(1)
        if constexpr ((ABL & 2) != 0) t2_interleave<0, NM, ND>();
        __builtin_amdgcn_sched_barrier(0);
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();                       // the producers have drained every DMA
#undef NH_READ_HALF
#undef NH_MFMA_HALF
#undef NH_NEXT_TAP

    static_assert(conv_epilogue_lds_bytes<T2_SLOTS, BN>() <= (size_t)(2 * A_BUF + T2_BST * B_STAGE) * sizeof(float), "epilogue LDS");
    conv_epilogue<TM, TN, PREC, T2_NCW * 64, T2_SLOTS, BN>(a, acc, smem, EpiTile{0, tw, th, img, r0, c0}, wm * 64,
                                                           wn * TN * 32, nt * BN, tid, lane);
}

namespace {
constexpr int kHR2 = 448;      // image rows per buffer for the BN = 64 instantiation (2 x 56 KB + 32 KB of weights)

// best th x tw <= 256 with (th+KH-1)(tw+KW-1) <= rows: most of the image per tile slot
bool plan_tiles(int Ho, int Wo, int KH, int KW, int rows, int* th_, int* tw_, double* eff_) {
    double best = 0;
    for (int th = 1; th <= Ho && th <= T2_SLOTS; ++th) {
        for (int tw = 1; tw <= Wo && th * tw <= T2_SLOTS; ++tw) {
            if ((th + KH - 1) * (tw + KW - 1) > rows) continue;
            const int ntr = (Ho + th - 1) / th, ntc = (Wo + tw - 1) / tw;
            const double eff = (double)Ho * Wo / ((double)ntr * ntc * T2_SLOTS);
            if (eff > best) { best = eff; *th_ = th; *tw_ = tw; }
        }
    }
    *eff_ = best;
    return best > 0;
}

template <int BN, int PREC, int ABL = 0> void launch_t2(const ConvArgs& a, int images, hipStream_t s) {
    constexpr size_t lds = (size_t)(2 * kHR2 * 32 + T2_BST * 32 * BN) * sizeof(float);
    static unsigned long long attr_devices = 0;
    set_max_dynamic_lds(reinterpret_cast<const void*>(&conv_igemm_halo2d<BN, PREC, kHR2, ABL>), lds, &attr_devices, "conv_igemm_halo2d");
    const int grid = images * a.t2_ntr * a.t2_ntc * (a.N / BN);
    NHANS_LAUNCH("conv_igemm_halo2d", (conv_igemm_halo2d<BN, PREC, kHR2, ABL>), dim3(grid), dim3((T2_NCW + T2_NPW) * 64), lds, s, a);
}
}  // namespace

bool launch_conv_igemm_halo2d(const ConvArgs& a0, hipStream_t s) {
    const bool enabled = dev_halo2d_enabled();
    const ConvSeg& g = a0.seg[0];
    if (!enabled || a0.nseg != 1 || a0.N % 64 != 0 || a0.N % 128 == 0) return false;
    if (g.sh != 1 || g.sw != 1 || a0.Ho != g.H || a0.Wo != g.W || g.KW < 2 || g.KH * g.KW < 3) return false;
    if (g.pl < 0 || g.pl >= g.KW || g.pt < 0 || g.pt >= g.KH || a0.M % (a0.Ho * a0.Wo) != 0) return false;
    const int images = a0.M / (a0.Ho * a0.Wo);
    if ((double)images * g.H * g.W * g.C + 65536.0 >= 2147483648.0) return false;
    int th = 0, tw = 0;
    double eff = 0;
    if (!plan_tiles(a0.Ho, a0.Wo, g.KH, g.KW, kHR2, &th, &tw, &eff) || eff < 0.93) return false;
    ConvArgs a = a0;
    a.t2_th = th; a.t2_tw = tw;
    a.t2_ntr = (a.Ho + th - 1) / th; a.t2_ntc = (a.Wo + tw - 1) / tw;
#ifdef NHANS_DEV
    const int abl = dev_ablate();
    if (a.prec == 1 && abl == 512) { launch_t2<64, 1, 1>(a, images, s); return true; }     // timing experiment: no per-tap barrier
    if (a.prec == 1 && abl == 1024) { launch_t2<64, 1, 2>(a, images, s); return true; }   // reads interleaved 1:1 with the MFMAs
#endif
    if (a.prec == 1) launch_t2<64, 1>(a, images, s);
    else launch_t2<64, 0>(a, images, s);
    return true;
}

}  // namespace nhans
