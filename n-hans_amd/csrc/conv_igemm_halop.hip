// Persistent form of the halo-reuse / wave-specialised convolution kernel (conv_igemm_halo.hip): same tile
// shapes, data layouts, staging, MFMA arrangement and arithmetic -- every output bit is the same -- but a
// workgroup stays on its CU and walks a list of tiles, and the DMA pipeline does not stop at a tile's end:
//
//   * conv_igemm_halo.hip keeps its vmcnt immediates static by issuing DUMMY loads past the last tap
//     (the image of a super-chunk that does not exist, the weights of taps total .. total+PFD-1).  Here
//     those slots carry the first halo image and the first PFD weight chunks of the workgroup's NEXT
//     tile: the ring of weight stages and the two image buffers simply keep turning, and the counted
//     waits need no change at all.
//   * While the consumer waves run the epilogue of tile i, the producers sit at the barrier with exactly
//     those loads in flight, so the "prologue" of tile i+1 (6-9 k cycles of a 107-170 k cycle tile: its
//     first image comes from HBM) is hidden behind the epilogue of tile i.
//   * The epilogue can therefore no longer take the whole LDS for its transposition buffer.  It works in
//     four rounds inside what is free at that moment, the image buffer of the last super-chunk: round
//     (i, j) takes accumulator tile [i][j] (32 pixels x 32 channels) of EVERY consumer wave -- so the
//     accumulator registers die a quarter at a time in all waves alike -- i.e. 128 x 64 (256 x 32 in the
//     512 x 64 shape) output values, whole 128-byte lines of the split-NHWC output; write, barrier, sweep
//     by all eight waves (8 channels per thread, arithmetic of conv_epilogue.h), barrier.  The producers
//     execute the same eight barriers and issue nothing meanwhile.
//
// MEASURED (32 clips, same process, tools/ab_variants.py --option persistent_tiles): 2.9 % / 1.6 % SLOWER
// than one tile per workgroup (517 vs 502 ms, 267 vs 263 ms): the prologue is hidden, but the epilogue in
// rounds keeps only one pass of global loads in flight ahead of the one being combined -- the registers the
// still-live accumulators leave allow no more (a second pass ahead spills inside the K loop: 5.7 % slower) --
// and pays eight workgroup barriers.  NOT the default; kept selectable (option "persistent_tiles" = 1) so that
// the result can be reproduced.  Used for the split-f16 main path (PREC 1, split-NHWC output, residual none /
// split tensor / 1-channel image) when a launch has at least two tiles per CU.
#include "conv_epilogue.h"

namespace nhans {

namespace {
constexpr int NCW = 8;        // consumer (MFMA) waves
constexpr int NPW = 4;        // producer (DMA) waves

template <int HBM_> struct HaloShapeP {          // as HaloShape in conv_igemm_halo.hip
    static constexpr int HBM = HBM_;
    static constexpr int HR = HBM_ == 512 ? 544 : 320;
    static constexpr int BST = HBM_ == 512 ? 3 : 4;
    static constexpr int WN = HBM_ == 512 ? 1 : 2;
    static constexpr int NAP = HR * 8 / (NPW * 64);
};

template <int N> __device__ __forceinline__ void halop_wait_vmcnt() {
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

// pixel row -> what the epilogue needs to know about it
struct RowInfo {
    int cbx;      // clip * cb_stride
    int tfy;      // (ho*Wo + wo) * N
    int m;        // output pixel or -1
    int ids;      // 1-channel residual index
};
__device__ __forceinline__ RowInfo row_info(const ConvArgs& a, int m0, int p) {
    const int mz = m0 + p < a.M ? m0 + p : -1;
    const int m = mz < 0 ? m0 : mz;
    const uint32_t b = fd_div((uint32_t)m, a.fdHoWo);
    const uint32_t rem = (uint32_t)m - b * a.fdHoWo.d;
    const uint32_t ho = fd_div(rem, a.fdWo);
    const uint32_t wo = rem - ho * a.fdWo.d;
    const int clip = a.img_clip ? a.img_clip[b] : 0;
    RowInfo r;
    r.cbx = clip * a.cb_stride;
    r.tfy = (int)rem * a.N;
    r.m = mz;
    r.ids = (int)((b * a.idH + ho * a.idsh) * a.idW + wo * a.idsw);
    return r;
}

template <int IDM> struct RoundRaw {              // what one pass of the sweep has in flight
    f32x4 t0, t1;
    f16x8 h, l;
    float sv;
    int m, prow;
};
}  // namespace

template <int BN, int HBM_>
__global__ void __launch_bounds__((NCW + NPW) * 64) conv_igemm_halo_persist(const ConvArgs a, int ntiles) {
    using SH = HaloShapeP<HBM_>;
    constexpr int HBM = SH::HBM, HR = SH::HR, BST = SH::BST, WN = SH::WN, NAP = SH::NAP;
    constexpr int PFD = BST - 1;
    constexpr int TM = 2;
    constexpr int TN = BN / (32 * WN);
    static_assert(HBM == (NCW / WN) * TM * 32 && BN == WN * TN * 32, "wave grid");
    constexpr int A_BUF = HR * 32;                     // floats
    constexpr int B_STAGE = 32 * BN;                   // floats
    constexpr int B_BASE = 2 * A_BUF;
    constexpr int GBP = B_STAGE / 4 / (NPW * 64);
    constexpr int NR = 4;                              // epilogue rounds: TM x TN accumulator tiles per wave
    extern __shared__ __attribute__((aligned(16))) float smem[];

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);

    // ---- this workgroup's tiles: XCD x owns the contiguous range [xs, xs + xc) of the ntiles tiles
    // (as the one-tile-per-workgroup kernels); its W = gridDim.x / 8 workgroups take every W-th of them,
    // so neighbouring workgroups of an XCD work on neighbouring tiles at the same time.
    const int ntn = a.N / BN;
    const int W = gridDim.x >> 3;
    int tile, tile_end;
    {
        const int q = ntiles >> 3, r = ntiles & 7, xcd = blockIdx.x & 7, idx = blockIdx.x >> 3;
        const int xs = xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
        tile = xs + idx;
        tile_end = xs + q + (xcd < r ? 1 : 0);
    }
    if (tile >= tile_end) return;                      // (never: the launcher asks for >= 2 tiles per workgroup)

    const int Wo = (int)a.fdWo.d;
    const int nseg = a.nseg;
    const int KW0 = a.seg[0].KW, KH0 = a.seg[0].KH, CC0 = a.seg[0].C >> 5;
    const int KH1 = nseg > 1 ? a.seg[1].KH : 0, CC1 = nseg > 1 ? (a.seg[1].C >> 5) : 0;
    const int nsup0 = KH0 * CC0;
    const int nsup = nsup0 + KH1 * CC1;
    const int ntap0 = nsup0 * KW0;
    const int total = ntap0 + KH1 * CC1;

    // cursor of the current tap (both roles step it identically): segment, kw within the super-chunk,
    // halo buffer, super-chunk count.  bufC is NOT reset between tiles: the buffers keep alternating.
    int segC = 0, kwC = 0, bufC = 0, supC = 0;
#define NH_NEXT_TAP()                                                                              \
    if (++kwC >= (segC ? 1 : KW0)) {                                                               \
        kwC = 0;                                                                                   \
        bufC ^= 1;                                                                                 \
        if (++supC == nsup0) segC = 1;                                                             \
    }

    if (wave >= NCW) {
        // =========================================================================================
        // Producer waves (see conv_igemm_halo.hip for the per-iteration protocol and the wait counts).
        const int pw = wave - NCW, ptid = tid - NCW * 64;
        __builtin_amdgcn_s_setprio(3);
        const int slot = lane & 7;
        const int nrows_all = a.M / Wo;
        const size_t bstride = (size_t)(a.N / 32) * 1024;
        int poff[NAP], hov[NAP];
        int sH = 0, sW = 0, sC = 0;
        const float* ssrc = nullptr;
        // activation context: the tile whose images are being issued
        int tileA = tile, m0A = 0, R0A = 0, w0A = 0;
#define NH_TILE_A()                                                                                \
    {                                                                                              \
        m0A = (tileA / ntn) * HBM;                                                                 \
        R0A = (int)fd_div((uint32_t)m0A, a.fdWo);                                                  \
        w0A = m0A - R0A * Wo;                                                                      \
    }
#define NH_MAP_SEGMENT(S)                                                                          \
    {                                                                                              \
        const ConvSeg& g = a.seg[S];                                                               \
        sH = g.H; sW = g.W; sC = g.C; ssrc = g.src;                                                \
        _Pragma("unroll") for (int d = 0; d < NAP; ++d) {                                          \
            const int j = d * 32 + pw * 8 + (lane >> 3);                                           \
            const int sp = (slot ^ ((j >> 1) & 7)) * 4;                                            \
            int Rg, wi;                                                                            \
            bool ok;                                                                               \
            if (g.KW > 1) {                                                                        \
                const int n0 = Wo - w0A + g.KW - 1;                                                \
                int i = 0, cj = w0A + j;                                                           \
                if (j >= n0) {                                                                     \
                    const int jj = j - n0;                                                         \
                    const int q = (int)fd_div((uint32_t)jj, a.fdWP);                               \
                    i = 1 + q;                                                                     \
                    cj = jj - q * (int)a.fdWP.d;                                                   \
                }                                                                                  \
                wi = cj - g.pl;                                                                    \
                Rg = R0A + i;                                                                      \
                ok = Rg < nrows_all && (unsigned)wi < (unsigned)g.W;                               \
            } else {                                                                               \
                const int m = m0A + j;                                                             \
                ok = j < HBM && m < a.M;                                                           \
                Rg = (int)fd_div((uint32_t)(ok ? m : 0), a.fdWo);                                  \
                wi = ((ok ? m : 0) - Rg * Wo) * g.sw - g.pl;                                       \
            }                                                                                      \
            if (!ok) { Rg = 0; wi = 0; }                                                           \
            const int b = (int)fd_div((uint32_t)(Rg * Wo), a.fdHoWo);                              \
            const int hi0 = (Rg - b * a.Ho) * g.sh - g.pt;                                         \
            poff[d] = ((b * g.H + hi0) * g.W + wi) * g.C + sp;                                     \
            hov[d] = ok ? hi0 : -(1 << 28);                                                        \
        }                                                                                          \
    }
#define NH_GLDS(SRC, DST)                                                                          \
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(SRC),        \
                                     (__attribute__((address_space(3))) void*)(DST), 16, 0, 0);

        int segA = 0, khA = 0, ccA = 0, supA = 0;
        bool liveA = true;                              // false: past the workgroup's last tile (zero-page loads)
        const float* const zp = a.zero + (slot ^ ((lane >> 4) & 7)) * 4;
#define NH_ISSUE_A(BUF)                                                                            \
    {                                                                                              \
        if (supA == nsup) {                             /* this image opens the NEXT tile */       \
            tileA += W;                                                                            \
            liveA = tileA < tile_end;                                                              \
            if (liveA) {                                                                           \
                NH_TILE_A()                                                                        \
                segA = 0; khA = 0; ccA = 0; supA = 0;                                              \
                NH_MAP_SEGMENT(0)                                                                  \
            }                                                                                      \
        }                                                                                          \
        const int khoff_ = khA * sW * sC + ccA * 32;                                               \
        float* sa_ = smem + (BUF) * A_BUF + pw * 8 * 32;                                           \
        _Pragma("unroll") for (int d = 0; d < NAP; ++d) {                                          \
            const float* p_ = (liveA && (unsigned)(hov[d] + khA) < (unsigned)sH) ? ssrc + (poff[d] + khoff_) : zp; \
            NH_GLDS(p_, sa_ + d * 32 * 32)                                                         \
        }                                                                                          \
        if (liveA) {                                                                               \
            ++supA;                                                                                \
            if (++khA >= (segA ? KH1 : KH0)) {                                                     \
                khA = 0;                                                                           \
                if (++ccA >= (segA ? CC1 : CC0)) {                                                 \
                    ccA = 0;                                                                       \
                    if (segA == 0 && nseg > 1) {                                                   \
                        segA = 1;                                                                  \
                        NH_MAP_SEGMENT(1)                                                          \
                    }                                                                              \
                }                                                                                  \
            }                                                                                      \
        }                                                                                          \
    }
        // weight context: the tile whose chunks are being issued (PFD taps ahead of the consumers)
        int tileB = tile, tapB = 0;
        bool liveB = true;
        const float* bp_;
        const float* wpk1;
#define NH_TILE_B()                                                                                \
    {                                                                                              \
        const size_t nt0_ = (size_t)(tileB % ntn) * (BN / 32);                                     \
        bp_ = a.seg[0].wpk + nt0_ * 1024 - bstride;                                                \
        wpk1 = (nseg > 1 ? a.seg[1].wpk : a.seg[0].wpk) + nt0_ * 1024;                             \
        tapB = 0;                                                                                  \
    }
#define NH_ISSUE_B(ST)                                                                             \
    {                                                                                              \
        if (tapB == total) {                            /* this chunk opens the NEXT tile */       \
            tileB += W;                                                                            \
            liveB = tileB < tile_end;                                                              \
            if (liveB) NH_TILE_B()                                                                 \
        }                                                                                          \
        if (liveB) {                                                                               \
            bp_ = tapB == ntap0 ? wpk1 : bp_ + bstride;                                            \
            ++tapB;                                                                                \
        }                                                                                          \
        float* sb_ = smem + B_BASE + (ST) * B_STAGE;                                               \
        _Pragma("unroll") for (int j = 0; j < GBP; ++j)                                            \
            NH_GLDS(bp_ + (j * (NPW * 64) + ptid) * 4, sb_ + (j * (NPW * 64) + pw * 64) * 4)       \
    }

        NH_TILE_A()
        NH_MAP_SEGMENT(0)
        NH_TILE_B()
        NH_ISSUE_A(0)
        NH_ISSUE_B(0)
        NH_ISSUE_B(1)
        if constexpr (PFD == 3) NH_ISSUE_B(2)
        halop_wait_vmcnt<(PFD - 1) * GBP>();            // image 0 and tap 0 of the first tile
        __builtin_amdgcn_s_barrier();
        bool prev_first = false;
        int stB = PFD;
        for (;;) {
            for (int it = 0; it < total; ++it) {
                const int KWc = segC ? 1 : KW0;
                const bool first = kwC == 0;
                if (first) NH_ISSUE_A(bufC ^ 1)
                NH_ISSUE_B(stB)
                if (++stB == BST) stB = 0;
                if (KWc == 1) halop_wait_vmcnt<GBP>();
                else if (first || (PFD == 3 && prev_first)) halop_wait_vmcnt<(PFD - 1) * GBP + NAP>();
                else halop_wait_vmcnt<(PFD - 1) * GBP>();
                __builtin_amdgcn_s_barrier();
                prev_first = first;
                NH_NEXT_TAP()
            }
            tile += W;
            if (tile >= tile_end) break;
            // the consumers' epilogue: 2 barriers per round; nothing is issued meanwhile (the next
            // iteration's loads target the very buffers the epilogue works in)
#pragma unroll
            for (int e = 0; e < 2 * NR; ++e) __builtin_amdgcn_s_barrier();
            segC = 0; supC = 0;                         // (kwC is 0, bufC keeps alternating)
        }
        halop_wait_vmcnt<0>();                          // zero-page loads past the end still target LDS
        __builtin_amdgcn_s_barrier();
        return;                                         // the last epilogue's barriers count live waves only
#undef NH_TILE_A
#undef NH_TILE_B
#undef NH_MAP_SEGMENT
#undef NH_GLDS
#undef NH_ISSUE_A
#undef NH_ISSUE_B
    }

    // =============================================================================================
    // Consumer waves.
    const int wm = wave / WN, wn = wave % WN;
    const int g8 = lane >> 5;
    const int bcol = (wn * TN) * 1024 + lane * 4;
    constexpr int KS = 2, KH_ = 1;                     // PREC 1: two k-steps of 16 per chunk
    f32x4 fa_hi[KS][TM], fa_lo[KS][TM], fb_hi[KS][TN], fb_lo[KS][TN];
    int jb0[TM], jb1[TM];
#define NH_READ_HALF(H, STG)                                                                       \
    {                                                                                              \
        const float* Sa_ = smem + bufC * A_BUF;                                                    \
        const float* Sb_ = smem + B_BASE + (STG) * B_STAGE + bcol;                                 \
        _Pragma("unroll") for (int t = 0; t < TM; ++t) {                                           \
            const int jr_ = (segC ? jb1[t] : jb0[t]) + kwC;                                        \
            const float* ar_ = Sa_ + jr_ * 32;                                                     \
            const int rs_ = (jr_ >> 1) & 7;                                                        \
            _Pragma("unroll") for (int s = (H) * KH_; s < ((H) + 1) * KH_; ++s) {                  \
                fa_hi[s][t] = *reinterpret_cast<const f32x4*>(ar_ + (((2 * s + g8) ^ rs_) * 4));   \
                fa_lo[s][t] = *reinterpret_cast<const f32x4*>(ar_ + (((2 * s + g8 + 4) ^ rs_) * 4)); \
            }                                                                                      \
        }                                                                                          \
        _Pragma("unroll") for (int s = (H) * KH_; s < ((H) + 1) * KH_; ++s)                        \
            _Pragma("unroll") for (int j = 0; j < TN; ++j) {                                       \
                fb_hi[s][j] = *reinterpret_cast<const f32x4*>(Sb_ + j * 1024 + s * 512);           \
                fb_lo[s][j] = *reinterpret_cast<const f32x4*>(Sb_ + j * 1024 + s * 512 + 256);     \
            }                                                                                      \
    }
#define NH_MFMA_HALF(H)                                                                            \
    {                                                                                              \
        _Pragma("unroll") for (int s = (H) * KH_; s < ((H) + 1) * KH_; ++s)                        \
            _Pragma("unroll") for (int p = 0; p < 3; ++p)                                          \
                _Pragma("unroll") for (int t = 0; t < TM; ++t)                                     \
                    _Pragma("unroll") for (int j = 0; j < TN; ++j) {                               \
                        const f16x8 a_ = __builtin_bit_cast(f16x8, p == 0 ? fa_lo[s][t] : fa_hi[s][t]); \
                        const f16x8 b_ = __builtin_bit_cast(f16x8, p == 1 ? fb_lo[s][j] : fb_hi[s][j]); \
                        acc[t][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(b_, a_, acc[t][j], 0, 0, 0); \
                    }                                                                              \
    }
    constexpr int NM = KH_ * TM * TN * 3;              // MFMAs per half
    constexpr int ND = KH_ * (TM + TN) * 2;            // ds_read_b128 per half

    __builtin_amdgcn_s_barrier();                       // image 0 and tap 0 of the first tile have landed
    int stC = 0;                                        // ring stage of the current tap (never reset)
    for (;;) {
        const int mt = tile / ntn, nt = tile - mt * ntn;
        const int m0 = mt * HBM;
        {
            const int R0 = (int)fd_div((uint32_t)m0, a.fdWo);
#pragma unroll
            for (int t = 0; t < TM; ++t) {
                const int r = wm * 64 + t * 32 + (lane & 31);
                const int irow = (int)fd_div((uint32_t)(m0 + r), a.fdWo) - R0;
                jb0[t] = r + (KW0 - 1) * irow;
                jb1[t] = r;
            }
        }
        f32x16 acc[TM][TN];
#pragma unroll
        for (int t = 0; t < TM; ++t)
#pragma unroll
            for (int j = 0; j < TN; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[t][j][r] = 0.f;

        NH_READ_HALF(0, stC)
        for (int it = 0; it < total; ++it) {
            __builtin_amdgcn_sched_barrier(0);
            NH_MFMA_HALF(0)
            NH_READ_HALF(1, stC)
            pin_reads_between_mfmas<0, NM, ND>();
            if (++stC == BST) stC = 0;
            __builtin_amdgcn_sched_barrier(0);
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            __builtin_amdgcn_sched_barrier(0);
            NH_NEXT_TAP()
            NH_MFMA_HALF(1)
            NH_READ_HALF(0, stC)                        // (after the last tap: a harmless read; the next tile reads its
            pin_reads_between_mfmas<0, NM, ND>();       //  own first half again, with its own row map, after the epilogue)
            __builtin_amdgcn_sched_barrier(0);
        }
        // Here: every consumer has passed the barrier of the last tap, so nobody reads the last
        // super-chunk's image buffer (bufC ^ 1 after the final NH_NEXT_TAP) any more; what is in
        // flight goes to the other buffer and to the weight stages.
        const bool last_tile = tile + W >= tile_end;
        if (last_tile) {
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();               // the producers have drained everything and leave
        }
        float* const ct = smem + (bufC ^ 1) * A_BUF;    // transposition buffer of a round
        // (the epilogue's index arithmetic must not be hoisted above the K loop, where every register counts:
        // its inputs become opaque here)
        int m0e = m0, nte = nt, tide = tid;
        asm volatile("" : "+s"(m0e), "+s"(nte));
        asm volatile("" : "+v"(tide));

        // ---- epilogue in rounds -----------------------------------------------------------------
        // Round (i, j) takes accumulator tile [i][j] (32 pixels x 32 channels) of EVERY wave, so the
        // accumulator registers die a quarter at a time in all waves alike: LROWS = 32 x (waves along
        // the pixels) rows by LCOLS = 32 x (waves along the channels) columns, whole 32-channel groups =
        // whole 128-byte lines of the split-NHWC output.
        {
            constexpr int LROWS = (NCW / WN) * 32, LCOLS = WN * 32;
            constexpr int LDR = LCOLS + 4;                         // padded row of the round buffer
            constexpr int CG = LCOLS / 8;                          // threads along the channels of a row
            constexpr int RPP = NCW * 64 / CG;                     // rows per pass
            constexpr int NPS = LROWS / RPP;                       // passes per round (2)
            static_assert((size_t)LROWS * LDR <= (size_t)A_BUF && NPS * RPP == LROWS, "round buffer");
            const int f_tf = a.tf ? 1 : 0;
            const float* __restrict__ cbp = a.cb;
            const float* __restrict__ tfp = a.tf ? a.tf : a.zero;
            const float lo_clamp = a.relu ? 0.f : -3.0e38f;
            const int idm = a.id_mode;                 // 0 none, 1 split tensor, 2 one-channel image
            // one clip for the whole tile (pixels are ordered by clip): its bias is loaded once per round; a
            // tile that straddles two clips (one in ~14,000) reloads it per row
            const int last_p = (m0e + HBM <= a.M ? HBM : a.M - m0e) - 1;
            const int cbx_first = row_info(a, m0e, 0).cbx;
            const bool one_clip = cbx_first == row_info(a, m0e, last_p).cbx;
            const int cg = tide % CG, rl0 = tide / CG;               // this thread's column group and first local row
            bool sat = false;

            // pass ps of round (i, j): local row -> pixel row of the tile, column group -> first channel
            auto tile_row = [&](int i, int ps) { const int rl = ps * RPP + rl0; return (rl >> 5) * 64 + i * 32 + (rl & 31); };
            auto chan0 = [&](int j) { const int lc = cg * 8; return nte * BN + (lc >> 5) * (TN * 32) + j * 32 + (lc & 31); };

            auto issue = [&](int i, int j, int ps, RoundRaw<1>& r) {
                const int p = tile_row(i, ps);
                const int n = chan0(j);
                const RowInfo ri = row_info(a, m0e, p);
                r.prow = p;
                r.m = ri.m;
                const int mc = ri.m < 0 ? 0 : ri.m;
                r.t0 = *reinterpret_cast<const f32x4*>(tfp + (ri.tfy + n) * f_tf);
                r.t1 = *reinterpret_cast<const f32x4*>(tfp + (ri.tfy + n) * f_tf + 4 * f_tf);
                if (idm == 1) {
                    const _Float16* hp = reinterpret_cast<const _Float16*>(a.id + (size_t)mc * a.id_ld) + (n >> 5) * 64 + (n & 31);
                    r.h = *reinterpret_cast<const f16x8*>(hp);
                    r.l = *reinterpret_cast<const f16x8*>(hp + 32);
                } else if (idm == 2) {
                    r.sv = a.id[ri.ids];
                }
            };
            auto finish = [&](int j, int ps, const RoundRaw<1>& r) {
                const int n = chan0(j);
                const int hoff = (n >> 5) * 64 + (n & 31);
                const int rl = ps * RPP + rl0;
                const f32x4 ws0 = *reinterpret_cast<const f32x4*>(a.ws + n) * a.in_scale, ws1 = *reinterpret_cast<const f32x4*>(a.ws + n + 4) * a.in_scale;
                const f32x4 av0 = *reinterpret_cast<const f32x4*>(ct + rl * LDR + cg * 8);
                const f32x4 av1 = *reinterpret_cast<const f32x4*>(ct + rl * LDR + cg * 8 + 4);
                f32x4 i0 = {0.f, 0.f, 0.f, 0.f}, i1 = i0, iw0 = i0, iw1 = i0;
                if (idm != 0) { iw0 = *reinterpret_cast<const f32x4*>(a.idw + n) * a.id_scale; iw1 = *reinterpret_cast<const f32x4*>(a.idw + n + 4) * a.id_scale; }
                if (idm == 1) {
                    const f16x8 h = r.h, l = r.l;
                    i0 = f32x4{(float)h[0] + (float)l[0], (float)h[1] + (float)l[1], (float)h[2] + (float)l[2], (float)h[3] + (float)l[3]};
                    i1 = f32x4{(float)h[4] + (float)l[4], (float)h[5] + (float)l[5], (float)h[6] + (float)l[6], (float)h[7] + (float)l[7]};
                } else if (idm == 2) {
                    i0 = f32x4{r.sv, r.sv, r.sv, r.sv};
                    i1 = i0;
                }
                const int cx = one_clip ? cbx_first : row_info(a, m0e, r.prow).cbx;
                const f32x4 c0 = *reinterpret_cast<const f32x4*>(cbp + cx + n);
                const f32x4 c1 = *reinterpret_cast<const f32x4*>(cbp + cx + n + 4);
                const f32x4 r0 = epi_combine(av0, ws0, c0, r.t0, iw0, i0);
                const f32x4 r1 = epi_combine(av1, ws1, c1, r.t1, iw1, i1);
                if (r.m >= 0) {
                    f16x8 h, l;
#pragma unroll
                    for (int e = 0; e < 8; ++e) {
                        const float y = fmaxf(e < 4 ? r0[e] : r1[e - 4], lo_clamp) * a.out_scale;
                        sat |= !(fabsf(y) < a.sat_limit);
                        const float yc = fminf(fmaxf(y, -65504.f), 65504.f);
                        h[e] = (_Float16)yc;
                        l[e] = (_Float16)(yc - (float)h[e]);
                    }
                    _Float16* dst = reinterpret_cast<_Float16*>(a.out + (size_t)r.m * a.ldo) + hoff;
                    *reinterpret_cast<f16x8*>(dst) = h;
                    *reinterpret_cast<f16x8*>(dst + 32) = l;
                }
            };

            // the passes of all rounds form one sequence; pass s+1's global loads are issued before pass s is
            // finished (static double buffer; a third buffer spills inside the K loop)
            RoundRaw<1> ra, rb;
            issue(0, 0, 0, ra);
#pragma unroll
            for (int round = 0; round < TM * TN; ++round) {
                const int i = round / TN, j = round % TN;
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const f32x4 v = {acc[i][j][4 * g], acc[i][j][4 * g + 1], acc[i][j][4 * g + 2], acc[i][j][4 * g + 3]};
                    *reinterpret_cast<f32x4*>(ct + (wm * 32 + (lane & 31)) * LDR + wn * 32 + 8 * g + 4 * (lane >> 5)) = v;
                }
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                __builtin_amdgcn_s_barrier();
#pragma unroll
                for (int ps = 0; ps < NPS; ++ps) {
                    const int sq = round * NPS + ps;               // position in the sequence of passes
                    const int nq = sq + 1, nr = nq / NPS, np = nq % NPS;
                    if (sq & 1) {
                        if (nq < TM * TN * NPS) issue(nr / TN, nr % TN, np, ra);
                        finish(j, ps, rb);
                    } else {
                        if (nq < TM * TN * NPS) issue(nr / TN, nr % TN, np, rb);
                        finish(j, ps, ra);
                    }
                }
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // this round's buffer has been read
                __builtin_amdgcn_s_barrier();
            }
            if (sat && a.sat) atomicOr(a.sat, kSatActivation);
        }

        tile += W;
        if (tile >= tile_end) break;
        segC = 0; supC = 0;
    }
#undef NH_READ_HALF
#undef NH_MFMA_HALF
#undef NH_NEXT_TAP
}

// Tiles per launch below which the one-tile-per-workgroup kernel is used (2 tiles per CU).
static constexpr int kPersistMinTiles = 512;

template <int BN, int HBM_> static void launch_persist_t(const ConvArgs& a, int ntiles, hipStream_t s) {
    using SH = HaloShapeP<HBM_>;
    constexpr size_t lds = (size_t)(2 * SH::HR * 32 + SH::BST * 32 * BN) * sizeof(float);
    static_assert(lds <= 160 * 1024, "LDS of a gfx950 CU");
    static unsigned long long attr_devices = 0;
    set_max_dynamic_lds(reinterpret_cast<const void*>(&conv_igemm_halo_persist<BN, HBM_>), lds, &attr_devices, "conv_igemm_halo_persist");
    NHANS_LAUNCH("conv_igemm_halo_persist", (conv_igemm_halo_persist<BN, HBM_>), dim3(256), dim3((NCW + NPW) * 64), lds, s, a, ntiles);
}

// true = launched.  `a` must already have passed conv_igemm_halo_eligible() with the same tile choice.
bool launch_conv_igemm_halo_persist(const ConvArgs& a0, hipStream_t s) {
    if (a0.prec != 1 || !a0.out_split || !a0.persist) return false;
    if (!(a0.id_mode == 0 || (a0.id_mode == 1 && a0.id_split) || a0.id_mode == 2) || a0.aux || a0.Nreal != a0.N) return false;
    const bool wide = a0.N % 128 == 0;
    if (!wide && !a0.halo64_tile512) return false;      // (the 256 x 64 shape exists for A/B only)
    const int hbm = wide ? 256 : 512, bn = wide ? 128 : 64;
    const int ntiles = ((a0.M + hbm - 1) / hbm) * (a0.N / bn);
    if (ntiles < kPersistMinTiles) return false;
    ConvArgs a = a0;
    a.fdWP = make_fastdiv((uint32_t)(a.Wo + a.seg[0].KW - 1));
    if (wide) launch_persist_t<128, 256>(a, ntiles, s); else launch_persist_t<64, 512>(a, ntiles, s);
    return true;
}

}  // namespace nhans
