// Implicit-GEMM convolution, LDS-DMA variant ("v4"): same contract, data layouts, packed weights and
// epilogue as conv_igemm.hip, different K loop.  Built because the 128-pixel / 4-wave kernel turned
// out latency-bound in the split-f16 mode: its MFMAs per 32-channel chunk take ~770 cycles per
// wave while an L2/HBM round trip under load is ~3,000 cycles and its double-buffered LDS allows
// only one chunk of prefetch (profiles/r01: MFMA busy 37-44 %; removing the MFMAs altogether left
// 70 % of the run time).
//
//   * Workgroup = 8 wavefronts (2 per SIMD), tile = 256 output pixels x BN channels (BN 128 / 64),
//     wave tile 64 x 64 (BN 128) or 64 x 32 (BN 64): twice the MFMA work per staged byte of B.
//   * Every operand goes global -> LDS by LDS-DMA (`global_load_lds_dwordx4`): no staging VGPRs,
//     no ds_write pass, no select for padding -- padded taps read a zero page, resolved once per
//     filter tap into per-row pointers.
//   * The LDS-DMA destination is lane-linear, so the A image cannot be padded against bank
//     conflicts; it is XOR-swizzled instead (16-byte piece p of pixel row r sits at slot
//     p^((r>>1)&7): two 128-byte rows share a 256-byte bank row, and a ds_read_b128 lane group covers
//     rows {r..r+3, r+12..r+15, r+20..r+27} -- with (r>>1)&7 its 16 slots are distinct; the obvious
//     r&7 is 2-way conflicted and measured 18 % slower in tools/ubench/consumer_loop.hip), applied on
//     the SOURCE address of the DMA and on the ds_read address.
//   * 3-stage LDS ring, prefetch distance 2: iteration `it` issues the DMA of chunk it+2, multiplies
//     chunk it, then waits with a COUNTED `s_waitcnt vmcnt(G)` (G = DMA instructions per chunk) so
//     that chunk it+1 has landed while chunk it+2 stays in flight across the raw `s_barrier`.
//     (`__syncthreads()` would drain the queue: hipcc fences LDS-DMA with vmcnt(0).)
#include "conv_epilogue.h"
#include <algorithm>
#include <cstdlib>

namespace nhans {

namespace {
constexpr int DBM = 256, DBK = 32, DSTAGES = 3;

template <int N> __device__ __forceinline__ void wait_vmcnt() {
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}
}  // namespace

// GRP: the K loop sums in groups of a.kgroup chunks (each group from a zeroed accumulator, groups
// added in order) -- the arithmetic of a split-K launch, so that a layer's results do not depend on
// whether its launch was small enough to be split (batch- and shard-invariance stay bitwise).
template <int BN, int PREC, int ABL = 0, int PINGPONG = 0, int GRP = 0>   // ABL: timing ablations (1: A from the zero page, 2: B always chunk 0, 4: no epilogue)
__global__ void __launch_bounds__(512) conv_igemm_dma(const ConvArgs a) {
    constexpr int WN = 2, WM = 4;                     // wave grid: 4 (pixels) x 2 (channels)
    constexpr int TM = DBM / WM / 32;                  // 2
    constexpr int TN = BN / WN / 32;                   // 2 (BN 128) or 1 (BN 64)
    constexpr int A_STAGE = DBM * 32;                  // floats: 256 rows x 128 B
    constexpr int B_STAGE = DBK * BN;                  // floats
    constexpr int STAGE = A_STAGE + B_STAGE;
    constexpr int GB = BN / 64;                        // B DMA instructions per thread per chunk
    constexpr int G = 4 + GB;                          // DMA instructions per thread per chunk
    extern __shared__ __attribute__((aligned(16))) float smem[];

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1;
    const long long t_start = (kDev && a.dbg) ? (long long)__builtin_amdgcn_s_memtime() : 0;

    // XCD-aware, bijective remap of the linear workgroup id
    const int ntn = a.N / BN;
    int L;
    {
        const int nblk = gridDim.x, bid = blockIdx.x;
        const int q = nblk >> 3, r = nblk & 7, xcd = bid & 7, idx = bid >> 3;
        L = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
    }
    const int mt = L / ntn, nt = L - mt * ntn;
    const int m0 = mt * DBM;
    const int nt0 = nt * (BN / 32);

    // ---- A DMA assignment: instruction j of wave w moves pixel rows j*64 + w*8 .. +7 (8 lanes per
    // row); lane slot s = lane&7 fetches source piece s ^ (row&7) so that the linear LDS image is
    // the swizzled one.
    const int slot = lane & 7;
    int rb[4], rho[4], rwo[4], spiece[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int row = j * 64 + wave * 8 + (lane >> 3);
        spiece[j] = (slot ^ ((row >> 1) & 7)) * 4;    // float offset of the source piece
        const int m = m0 + row;
        if (m < a.M) {
            const uint32_t b = fd_div((uint32_t)m, a.fdHoWo);
            const uint32_t rem = (uint32_t)m - b * a.fdHoWo.d;
            const uint32_t ho = fd_div(rem, a.fdWo);
            rb[j] = (int)b; rho[j] = (int)ho; rwo[j] = (int)(rem - ho * a.fdWo.d);
        } else {
            rb[j] = -1; rho[j] = 0; rwo[j] = 0;
        }
    }

    int64_t roff0, roff1, roff2, roff3;
    int hi0[4], wi0[4];
    const float *pa0, *pa1, *pa2, *pa3;
    int seg = 0, kh = 0, kw = 0, c0 = 0;
    int sH, sW, sC, sKW, sKH;
    const float* ssrc;

#define NH_ROW(I, ROFF)                                                                            \
    if (rb[I] >= 0) {                                                                              \
        hi0[I] = rho[I] * g.sh - g.pt;                                                             \
        wi0[I] = rwo[I] * g.sw - g.pl;                                                             \
        ROFF = (((int64_t)rb[I] * g.H + hi0[I]) * g.W + wi0[I]) * (int64_t)g.C + spiece[I];        \
    } else {                                                                                       \
        hi0[I] = -(1 << 28); wi0[I] = 0; ROFF = 0;                                                 \
    }
#define NH_TAP_ROW(I, ROFF, PA)                                                                    \
    PA = (!(ABL & 1) && (unsigned)(hi0[I] + kh) < (unsigned)sH && (unsigned)(wi0[I] + kw) < (unsigned)sW) \
             ? ssrc + (ROFF + tapoff) : a.zero + spiece[I];
#define NH_TAP()                                                                                   \
    {                                                                                              \
        const int64_t tapoff = (int64_t)(kh * sW + kw) * sC;                                       \
        NH_TAP_ROW(0, roff0, pa0) NH_TAP_ROW(1, roff1, pa1)                                        \
        NH_TAP_ROW(2, roff2, pa2) NH_TAP_ROW(3, roff3, pa3)                                        \
    }
#define NH_ENTER_SEGMENT(S)                                                                        \
    {                                                                                              \
        const ConvSeg& g = a.seg[S];                                                               \
        sH = g.H; sW = g.W; sC = g.C; sKW = g.KW; sKH = g.KH; ssrc = g.src;                        \
        NH_ROW(0, roff0) NH_ROW(1, roff1) NH_ROW(2, roff2) NH_ROW(3, roff3)                        \
        kh = 0; kw = 0; c0 = 0;                                                                    \
        NH_TAP()                                                                                   \
    }
#define NH_ADVANCE_A()                                                                             \
    {                                                                                              \
        if (++kw >= sKW) {            /* K order (fold.py kmat): chunk, row, column, channel */    \
            kw = 0;                                                                                \
            if (++kh >= sKH) {                                                                     \
                kh = 0;                                                                            \
                c0 += DBK;                                                                         \
            }                                                                                      \
        }                                                                                          \
        if (c0 >= sC) {                                                                           \
            ++seg;                                                                                 \
            if (seg < a.nseg) NH_ENTER_SEGMENT(seg)                                                \
            else c0 = 0;              /* past the end: the cursor stays on valid memory */         \
        } else NH_TAP()                                                                            \
    }

#define NH_GLDS(SRC, DST)                                                                          \
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(SRC),        \
                                     (__attribute__((address_space(3))) void*)(DST), 16, 0, 0);

    // issue the DMA of the chunk under the cursor into ring stage ST, then advance the cursor
    const int n0chunks = a.seg[0].nchunks;
    const size_t bstride = (size_t)(a.N / 32) * 1024;
    int lchunk = 0;                                    // index of the next chunk to load
#define NH_ISSUE(ST)                                                                               \
    {                                                                                              \
        float* sa_ = smem + (ST) * STAGE + wave * 8 * 32;                                          \
        NH_GLDS(pa0 + c0, sa_)                                                                     \
        NH_GLDS(pa1 + c0, sa_ + 64 * 32)                                                           \
        NH_GLDS(pa2 + c0, sa_ + 128 * 32)                                                          \
        NH_GLDS(pa3 + c0, sa_ + 192 * 32)                                                          \
        const float* bp = ((ABL & 2) ? a.seg[0].wpk                                                \
                           : lchunk < n0chunks ? a.seg[0].wpk + (size_t)lchunk * bstride           \
                                             : a.seg[1].wpk + (size_t)(lchunk - n0chunks) * bstride) + \
                          (size_t)nt0 * 1024;                                                      \
        float* sb_ = smem + (ST) * STAGE + A_STAGE;                                                \
        _Pragma("unroll") for (int j = 0; j < GB; ++j)                                             \
            NH_GLDS(bp + (j * 512 + tid) * 4, sb_ + (j * 512 + wave * 64) * 4)                     \
        ++lchunk;                                                                                  \
        NH_ADVANCE_A()                                                                             \
    }

    // fragment addresses: A row r = wm*64 + t*32 + (lane&31); piece p of that row is at slot p^(r&7)
    const int g8 = lane >> 5;
    int aoff[TM];
#pragma unroll
    for (int t = 0; t < TM; ++t) aoff[t] = (wm * 64 + t * 32 + (lane & 31)) * 32;
    const int rsw = (lane >> 1) & 7;                   // ((row>>1) & 7): tile rows start at multiples of 32
    const int bcol = A_STAGE + (wn * TN) * 1024 + lane * 4;

    // MFMA operands of one whole chunk live in registers between the two phases
    constexpr int KS = PREC == 1 ? 2 : 4;              // k-steps per chunk (16 k each / 8 k each)
    f32x4 fa_hi[KS][TM], fa_lo[PREC == 1 ? KS : 1][TM], fb_hi[KS][TN], fb_lo[PREC == 1 ? KS : 1][TN];
#define NH_READ_FRAGS(ST)                                                                          \
    {                                                                                              \
        const float* Sb_ = smem + (ST) * STAGE;                                                    \
        _Pragma("unroll") for (int s = 0; s < KS; ++s) {                                           \
            _Pragma("unroll") for (int t = 0; t < TM; ++t) {                                       \
                fa_hi[s][t] = *reinterpret_cast<const f32x4*>(Sb_ + aoff[t] + (((2 * s + g8) ^ rsw) * 4)); \
                if constexpr (PREC == 1)                                                           \
                    fa_lo[s][t] = *reinterpret_cast<const f32x4*>(Sb_ + aoff[t] + (((2 * s + g8 + 4) ^ rsw) * 4)); \
            }                                                                                      \
            _Pragma("unroll") for (int j = 0; j < TN; ++j) {                                       \
                if constexpr (PREC == 1) {                                                         \
                    fb_hi[s][j] = *reinterpret_cast<const f32x4*>(Sb_ + bcol + j * 1024 + s * 512);       \
                    fb_lo[s][j] = *reinterpret_cast<const f32x4*>(Sb_ + bcol + j * 1024 + s * 512 + 256); \
                } else {                                                                           \
                    fb_hi[s][j] = *reinterpret_cast<const f32x4*>(Sb_ + bcol + j * 1024 + s * 256); \
                }                                                                                  \
            }                                                                                      \
        }                                                                                          \
    }
#define NH_MFMA_FRAGS()                                                                            \
    {                                                                                              \
        _Pragma("unroll") for (int s = 0; s < KS; ++s)                                             \
            _Pragma("unroll") for (int t = 0; t < TM; ++t)                                         \
                _Pragma("unroll") for (int j = 0; j < TN; ++j) {                                   \
                    if constexpr (PREC == 1) {                                                     \
                        const f16x8 ah_ = __builtin_bit_cast(f16x8, fa_hi[s][t]), al_ = __builtin_bit_cast(f16x8, fa_lo[s][t]); \
                        const f16x8 bh_ = __builtin_bit_cast(f16x8, fb_hi[s][j]), bl_ = __builtin_bit_cast(f16x8, fb_lo[s][j]); \
                        acc[t][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(bh_, al_, acc[t][j], 0, 0, 0); \
                        acc[t][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(bl_, ah_, acc[t][j], 0, 0, 0); \
                        acc[t][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(bh_, ah_, acc[t][j], 0, 0, 0); \
                    } else {                                                                       \
                        acc[t][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(fb_hi[s][j].x, fa_hi[s][t].x, acc[t][j], 0, 0, 0); \
                        acc[t][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(fb_hi[s][j].y, fa_hi[s][t].y, acc[t][j], 0, 0, 0); \
                        acc[t][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(fb_hi[s][j].z, fa_hi[s][t].z, acc[t][j], 0, 0, 0); \
                        acc[t][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(fb_hi[s][j].w, fa_hi[s][t].w, acc[t][j], 0, 0, 0); \
                    }                                                                              \
                }                                                                                  \
    }

    // The same, half a chunk at a time (k-steps [H*KS/2, (H+1)*KS/2)): one half is read while the other
    // is multiplied.  Per accumulator the products are added in the order of NH_MFMA_FRAGS (same bits);
    // consecutive MFMAs go to different accumulators.
    constexpr int KHD = KS / 2;
#define NH_READ_HALF_D(H, ST)                                                                      \
    {                                                                                              \
        const float* Sb_ = smem + (ST) * STAGE;                                                    \
        _Pragma("unroll") for (int s = (H) * KHD; s < ((H) + 1) * KHD; ++s) {                      \
            _Pragma("unroll") for (int t = 0; t < TM; ++t) {                                       \
                fa_hi[s][t] = *reinterpret_cast<const f32x4*>(Sb_ + aoff[t] + (((2 * s + g8) ^ rsw) * 4)); \
                if constexpr (PREC == 1)                                                           \
                    fa_lo[s][t] = *reinterpret_cast<const f32x4*>(Sb_ + aoff[t] + (((2 * s + g8 + 4) ^ rsw) * 4)); \
            }                                                                                      \
            _Pragma("unroll") for (int j = 0; j < TN; ++j) {                                       \
                if constexpr (PREC == 1) {                                                         \
                    fb_hi[s][j] = *reinterpret_cast<const f32x4*>(Sb_ + bcol + j * 1024 + s * 512);       \
                    fb_lo[s][j] = *reinterpret_cast<const f32x4*>(Sb_ + bcol + j * 1024 + s * 512 + 256); \
                } else {                                                                           \
                    fb_hi[s][j] = *reinterpret_cast<const f32x4*>(Sb_ + bcol + j * 1024 + s * 256); \
                }                                                                                  \
            }                                                                                      \
        }                                                                                          \
    }
#define NH_MFMA_HALF_D(H)                                                                          \
    {                                                                                              \
        _Pragma("unroll") for (int s = (H) * KHD; s < ((H) + 1) * KHD; ++s)                        \
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
    constexpr int NMD = KHD * TM * TN * (PREC == 1 ? 3 : 4);        // MFMAs per half
    constexpr int NDD = KHD * (TM + TN) * (PREC == 1 ? 2 : 1);      // ds_read_b128 per half

    f32x16 acc[TM][TN];
#pragma unroll
    for (int t = 0; t < TM; ++t)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[t][j][r] = 0.f;

    int total = 0;
    for (int s = 0; s < a.nseg; ++s) total += a.seg[s].nchunks;

    // Split-K (launches too small to fill the chip: the head's dense layer and the embedding tower at
    // one or two clips): blockIdx.y owns a contiguous range of the chunks of the (single) segment.
    const int ksplit = gridDim.y, kz = blockIdx.y;
    f32x16 tot[GRP ? TM : 1][GRP ? TN : 1];
    int gleft = GRP ? a.kgroup : 0;
    if constexpr (GRP) {
#pragma unroll
        for (int t = 0; t < TM; ++t)
#pragma unroll
            for (int j = 0; j < TN; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) tot[t][j][r] = 0.f;
    }
    int cbeg = 0;
    if (ksplit > 1) {
        const int cps = a.kgroup;                      // one group per split
        cbeg = kz * cps;
        total = total - cbeg < cps ? total - cbeg : cps;
    }

    // prologue: chunks 0 and 1 in flight, wait for chunk 0 only
    long long t_loop = 0, t_epi = 0;
    NH_ENTER_SEGMENT(0)
    if (cbeg) {                                        // cursor to chunk cbeg: K order (chunk, row, column)
        int cb_ = cbeg;
        if (cb_ >= n0chunks) {                         // the group starts inside the transform segment
            cb_ -= n0chunks;
            seg = 1;
            NH_ENTER_SEGMENT(1)
        }
        const int per_cc = sKH * sKW;
        const int cc_ = cb_ / per_cc;
        const int r_ = cb_ - cc_ * per_cc;
        kh = r_ / sKW;
        kw = r_ - kh * sKW;
        c0 = cc_ * DBK;
        lchunk = cbeg;
        NH_TAP()
    }
    NH_ISSUE(0)
    if (total > 1) {
        NH_ISSUE(1)
        wait_vmcnt<G>();
    } else {
        wait_vmcnt<0>();
    }
    __builtin_amdgcn_s_barrier();
    if (kDev && a.dbg) t_loop = (long long)__builtin_amdgcn_s_memtime();

    if constexpr (PINGPONG) {
        // Ping-pong: waves w and w+4 share a SIMD.  Each iteration has two phases separated by
        // workgroup barriers -- P1: issue the DMA share of chunk it+2, read every MFMA fragment of
        // chunk it from LDS into registers, counted wait for the own share of chunk it+1; P2: the
        // 24 (12) MFMAs from registers.  Waves 4-7 pass one extra barrier up front, so on every SIMD
        // one wave is always in P2 (matrix pipe) while the other is in P1 (memory / LDS): without the
        // stagger both run the same phase at the same time and the matrix pipe idles during P1.
        // Hazards (k = barrier index; group A runs P1(it) at k=2it, group B at k=2it+1):
        //   chunk c is read at k=2c (A), 2c+1 (B); its stage is rewritten for chunk c+3 from k=2c+2
        //   on, and every wave drains its ds_reads (lgkmcnt 0) before leaving P1;
        //   the shares of chunk c are issued at k=2c-4 (A), 2c-3 (B) and waited for at k=2c-2 (A),
        //   2c-1 (B), both before the barrier that precedes the first read.
        const bool grp_b = wave >= 4;
        if (grp_b) __builtin_amdgcn_s_barrier();
        int st = 0;
        for (int it = 0; it < total; ++it) {
            if (it + 2 < total) {
                const int st2 = st >= 1 ? st - 1 : st + 2;             // (st + 2) % 3
                NH_ISSUE(st2)
            }
            NH_READ_FRAGS(st)
            if (it + 2 < total) wait_vmcnt<G>(); else wait_vmcnt<0>();
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_sched_barrier(0);
            __builtin_amdgcn_s_barrier();
            __builtin_amdgcn_sched_barrier(0);
            if constexpr (ABL & 32) __builtin_amdgcn_s_setprio(1);
            NH_MFMA_FRAGS()
            if constexpr (ABL & 32) __builtin_amdgcn_s_setprio(0);
            __builtin_amdgcn_sched_barrier(0);
            __builtin_amdgcn_s_barrier();
            __builtin_amdgcn_sched_barrier(0);
            st = st == 2 ? 0 : st + 1;
        }
        if (!grp_b) __builtin_amdgcn_s_barrier();
    } else if (ABL == 0 && GRP == 0 && !(kDev && a.dbg)) {
        // (every launch that is not a grouped-sum / split-K one: the strided convs and the head conv)
        // Half-chunk register pipeline with the reads pinned between the MFMAs (as conv_igemm_halo.hip):
        //   issue the DMA of chunk it+2 | MFMAs of half 0 of chunk it + reads of its half 1 | counted wait,
        //   barrier: chunk it+1 has landed everywhere, nobody reads chunk it any more | MFMAs of half 1 +
        //   reads of half 0 of chunk it+1.
        int st = 0;
        NH_READ_HALF_D(0, 0)
        for (int it = 0; it < total; ++it) {
            if (it + 2 < total) {
                const int st2 = st >= 1 ? st - 1 : st + 2;             // (st + 2) % 3
                NH_ISSUE(st2)                                           // chunk it+2 -> stage freed at it-1
            }
            __builtin_amdgcn_sched_barrier(0);
            NH_MFMA_HALF_D(0)
            NH_READ_HALF_D(1, st)
            pin_reads_between_mfmas<0, NMD, NDD>();
            __builtin_amdgcn_sched_barrier(0);
            if (it + 2 < total) wait_vmcnt<G>(); else wait_vmcnt<0>();
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            __builtin_amdgcn_sched_barrier(0);
            st = st == 2 ? 0 : st + 1;
            NH_MFMA_HALF_D(1)
            NH_READ_HALF_D(0, st)                                       // (past the last chunk: a harmless read)
            pin_reads_between_mfmas<0, NMD, NDD>();
            __builtin_amdgcn_sched_barrier(0);
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();                  // the harmless read past the end has returned before the epilogue reuses the LDS
    } else {
        int st = 0;                                    // ring stage of chunk `it`
        long long ph0 = 0, ph1 = 0, ph2 = 0, ph3 = 0, tq = 0;          // dev tool: in-loop phase cycles
        for (int it = 0; it < total; ++it) {
            if (kDev && a.dbg) tq = (long long)__builtin_amdgcn_s_memtime();
            if (it + 2 < total) {
                const int st2 = st >= 1 ? st - 1 : st + 2;             // (st + 2) % 3
                NH_ISSUE(st2)                                           // chunk it+2 -> stage freed at it-1
            }
            if (kDev && a.dbg) { __builtin_amdgcn_sched_barrier(0); const long long t = (long long)__builtin_amdgcn_s_memtime(); ph0 += t - tq; tq = t; }
            NH_READ_FRAGS(st)
            if (kDev && a.dbg) { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); __builtin_amdgcn_sched_barrier(0); const long long t = (long long)__builtin_amdgcn_s_memtime(); ph1 += t - tq; tq = t; }
            NH_MFMA_FRAGS()
            if constexpr (GRP) {
                if (ksplit == 1 && --gleft == 0) {      // close the group: tot += acc, acc = 0
                    gleft = a.kgroup;
#pragma unroll
                    for (int t = 0; t < TM; ++t)
#pragma unroll
                        for (int j = 0; j < TN; ++j) {
                            tot[t][j] += acc[t][j];
#pragma unroll
                            for (int r = 0; r < 16; ++r) acc[t][j][r] = 0.f;
                        }
                }
            }
            __builtin_amdgcn_sched_barrier(0);
            if (kDev && a.dbg) { const long long t = (long long)__builtin_amdgcn_s_memtime(); ph2 += t - tq; tq = t; }
            // chunk it+1 must have landed (in every wave) before anyone reads it; chunk it+2 stays in flight
            if (it + 2 < total) wait_vmcnt<G>(); else wait_vmcnt<0>();
            __builtin_amdgcn_s_barrier();
            if (kDev && a.dbg) { const long long t = (long long)__builtin_amdgcn_s_memtime(); ph3 += t - tq; }
            st = st == 2 ? 0 : st + 1;
        }
        if (kDev && a.dbg && tid == 0) {
            long long* d = a.dbg + (size_t)(4 << 20) + (size_t)blockIdx.x * 4;
            d[0] = ph0; d[1] = ph1; d[2] = ph2; d[3] = ph3;
        }
    }

#undef NH_ROW
#undef NH_TAP_ROW
#undef NH_TAP
#undef NH_ENTER_SEGMENT
#undef NH_ADVANCE_A
#undef NH_GLDS
#undef NH_ISSUE
#undef NH_READ_FRAGS
#undef NH_MFMA_FRAGS

    if (kDev && a.dbg) t_epi = (long long)__builtin_amdgcn_s_memtime();
    if constexpr (GRP) {
        if (ksplit == 1) {                              // last (partial) group, then the sum is the result
#pragma unroll
            for (int t = 0; t < TM; ++t)
#pragma unroll
                for (int j = 0; j < TN; ++j) {
                    if (gleft != a.kgroup) tot[t][j] += acc[t][j];
                    acc[t][j] = tot[t][j];
                }
        }
    }
    if (ksplit > 1) {
        // Every workgroup parks its partial accumulator tile in scratch (thread-private slots, so no
        // transposition: piece q of thread t of split z), then takes a ticket; the last one to arrive
        // sums the partials in the fixed order z = 0..ksplit-1 -- bitwise reproducible whoever is last
        // -- and runs the epilogue.  Fences are agent scope: the splits of a tile may sit on different
        // XCDs, whose L2s are not coherent with each other inside a kernel.
        constexpr int NQ = TM * TN * 4;
        const size_t slot = (size_t)512 * NQ * 4;      // floats per (tile, split)
        float* part = a.kscratch + ((size_t)L * ksplit + kz) * slot + tid * 4;
#pragma unroll
        for (int t = 0; t < TM; ++t)
#pragma unroll
            for (int j = 0; j < TN; ++j)
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const f32x4 v = {acc[t][j][4 * q], acc[t][j][4 * q + 1], acc[t][j][4 * q + 2], acc[t][j][4 * q + 3]};
                    *reinterpret_cast<f32x4*>(part + ((t * TN + j) * 4 + q) * 2048) = v;
                }
        // One release per workgroup and an acquire in the last arriver only: an agent-scope fence writes the XCD's L2
        // back AND invalidates it -- issued by all eight waves of every split (7,680 times in the head's dense launch)
        // it kept emptying the L2 under the K loops of the workgroups still running
        // (0.53 -> 0.25 ms for that launch, profiles/r05/ab_splitk_scratch_and_fences.txt).
        //
        // Why ONE thread's release and a RELAXED ticket are enough -- by the memory model, not by luck:
        //  (1) HSA / C++ fence-fence synchronisation: a release fence F_r sequenced before an atomic write X, and an atomic
        //      read Y that sees X (here: the ticket fetch_add of a later arriver reads the value this one wrote -- RMWs on
        //      one address form a release sequence) sequenced before an acquire fence F_a, make F_r synchronise with F_a
        //      whatever the memory order of X and Y themselves: the ticket may be relaxed.  The last arriver's acquire
        //      fence pairs with EVERY earlier arriver's release fence through the chain of RMWs.
        //  (2) The other seven waves' stores happen-before thread 0's fence through the workgroup barrier
        //      (__syncthreads = workgroup-scope release / acquire), and happens-before is transitive across inclusive
        //      scopes: barrier (workgroup) then fence (agent) publishes them at agent scope.
        //  (3) What the hardware does for it (CDNA3 ISA, "Memory model" / LLVM AMDGPU memory model for gfx942/gfx950): a
        //      global store decrements vmcnt when the XCD's L2 has ACKNOWLEDGED the write (the vector L1 is write-through),
        //      so after `s_waitcnt vmcnt(0)` + s_barrier all eight waves' partials are in this XCD's L2; the agent-scope
        //      release is `buffer_wbl2 sc1` + `s_waitcnt vmcnt(0)`, which writes back EVERY dirty line of that L2 -- not
        //      only the issuing wave's -- to memory; the agent-scope RMW executes at the memory side (sc1), coherent across
        //      XCDs; the acquire is `buffer_inv sc1`, which drops the reader's L2 / L1 copies before it loads the partials
        //      (with non-temporal loads on top).  The explicit vmcnt(0) in front of the barrier is what (2) needs on this
        //      hardware: s_barrier alone does not wait for outstanding stores.
        // tests/test_gpu_abi_edges.py::test_split_k_stress_is_bit_identical_to_the_unsplit_walk runs 10,000 split launches at
        // the largest grids the rule admits against the unsplit walk, bit for bit.
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // (explicit: a workgroup-scope barrier alone need not wait for stores)
        __syncthreads();                                    // every wave's partial stores have reached this XCD's L2
        int* ticket = reinterpret_cast<int*>(smem);
        if (tid == 0) {
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // the write-back has completed before the ticket is taken
            ticket[0] = __hip_atomic_fetch_add(a.kcounter + L, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        __syncthreads();
        const int old = ticket[0];
        __syncthreads();
        if (old != ksplit - 1) return;
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        if (tid == 0) a.kcounter[L] = 0;               // ready for the next launch on this stream
#pragma unroll
        for (int t = 0; t < TM; ++t)
#pragma unroll
            for (int j = 0; j < TN; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[t][j][r] = 0.f;
        const float* all = a.kscratch + (size_t)L * ksplit * slot + tid * 4;
#pragma unroll 1
        for (int z = 0; z < ksplit; ++z) {
#pragma unroll
            for (int t = 0; t < TM; ++t)
#pragma unroll
                for (int j = 0; j < TN; ++j)
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        const f32x4 v = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(all + z * slot + ((t * TN + j) * 4 + q) * 2048));
                        acc[t][j][4 * q] += v.x; acc[t][j][4 * q + 1] += v.y;
                        acc[t][j][4 * q + 2] += v.z; acc[t][j][4 * q + 3] += v.w;
                    }
        }
    }
    if constexpr (ABL & 4) {
        float chk = 0.f;                                 // keep every MFMA alive
#pragma unroll
        for (int t = 0; t < TM; ++t)
#pragma unroll
            for (int j = 0; j < TN; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) chk += acc[t][j][r];
        if (chk == 123.456f) a.out[0] = chk;
    } else {
        static_assert(conv_epilogue_lds_bytes<DBM, BN>() <= (size_t)DSTAGES * STAGE * sizeof(float), "epilogue LDS");
        conv_epilogue<TM, TN, PREC, 512, DBM, BN>(a, acc, smem, EpiTile{m0, 0, 0, 0, 0, 0}, wm * 64, wn * TN * 32, nt * BN, tid, lane);
    }
    if (kDev && a.dbg) {                                       // dev tool (tools/conv_phase_cycles.py)
        __syncthreads();
        if (tid == 0) {
            long long* d = a.dbg + (size_t)blockIdx.x * 4;
            d[0] = t_start; d[1] = t_loop; d[2] = t_epi; d[3] = (long long)__builtin_amdgcn_s_memtime();
        }
    }
}

template <int BN, int PREC, int ABL = 0, int PINGPONG = 0, int GRP = 0>
static void launch_dma_g(const ConvArgs& a, int grid, int ks, hipStream_t s) {
    constexpr size_t lds = (size_t)DSTAGES * (DBM * 32 + DBK * BN) * sizeof(float);
    static unsigned long long attr_devices = 0;
    set_max_dynamic_lds(reinterpret_cast<const void*>(&conv_igemm_dma<BN, PREC, ABL, PINGPONG, GRP>), lds, &attr_devices, "conv_igemm_dma");
    NHANS_LAUNCH("conv_igemm_dma", (conv_igemm_dma<BN, PREC, ABL, PINGPONG, GRP>), dim3(grid, ks), dim3(512), lds, s, a);
}

// Grouped summation / split-K plan of one launch.  The group size depends on the layer only (K) -- never on the launch,
// and never on whether a scratch buffer exists: single-segment convs with >= 32 chunks sum in <= 32 groups of >= 8
// chunks.  A launch that would leave most CUs idle (the head's dense layer, the embedding tower at a few clips) runs
// one workgroup per (tile, group) -- split-K -- when the scratch holds grid x groups accumulator tiles (512 threads x
// TM*TN*16 floats); any other launch walks the groups in order inside the K loop.  Same additions in the same order
// either way: with no scratch (allocation failed, option split_k = 0) the results are the same bits.
namespace {
struct SplitKPlan { int kgroup, groups; size_t bytes; bool wants_split; };   // bytes: scratch the split form needs
SplitKPlan splitk_plan(const ConvArgs& a, int BN) {
    SplitKPlan p{0, 1, 0, false};
    int total = 0;
    for (int i = 0; i < a.nseg; ++i) total += a.seg[i].nchunks;
    // (a.kgroup < 0 on entry = the caller marks the layers whose launches can be that small)
    if (!(a.kgroup < 0 && total >= 32)) return p;
    p.kgroup = std::max(8, (total + 31) / 32);
    p.groups = (total + p.kgroup - 1) / p.kgroup;
    const int grid = ((a.M + DBM - 1) / DBM) * (a.N / BN);
    const size_t slot_bytes = (size_t)512 * (BN / 64) * 2 * 16 * sizeof(float);
    p.bytes = (size_t)grid * p.groups * slot_bytes;
    p.wants_split = grid <= 96 && p.groups > 1;
    return p;
}
}  // namespace

// scratch bytes the split-K form of this launch needs (0: the launch does not split) -- nhans_api.hip sizes the
// context's scratch from it, lazily
size_t conv_splitk_scratch_bytes(const ConvArgs& a) {
    if (a.variant < 1) return 0;
    const SplitKPlan p = splitk_plan(a, a.N % 128 == 0 ? 128 : 64);
    return p.wants_split ? p.bytes : 0;
}

template <int BN, int PREC, int ABL = 0, int PINGPONG = 0>
static void launch_dma_t(const ConvArgs& a0, hipStream_t s) {
    const int mtiles = (a0.M + DBM - 1) / DBM;
    const int grid = mtiles * (a0.N / BN);
    if constexpr (ABL || PINGPONG) {
        launch_dma_g<BN, PREC, ABL, PINGPONG, 0>(a0, grid, 1, s);
    } else {
        ConvArgs a = a0;
        const SplitKPlan p = splitk_plan(a0, BN);
        a.kgroup = p.kgroup;
        if (!a.kgroup) { launch_dma_g<BN, PREC, 0, 0, 0>(a, grid, 1, s); return; }
        const bool split = p.wants_split && a.kscratch && a.kcounter && grid <= a.kcounter_n && p.bytes <= a.kscratch_bytes;
        launch_dma_g<BN, PREC, 0, 0, 1>(a, grid, split ? p.groups : 1, s);
    }
}

void launch_conv_igemm_dma(const ConvArgs& a, hipStream_t s) {
    if (a.prec == 1) {
#ifdef NHANS_DEV
        const int abl = dev_ablate();
        if (a.N % 128 == 0) {
            switch (abl) {          // timing experiments only: results are wrong for abl != 0
                case 1: launch_dma_t<128, 1, 1>(a, s); break;
                case 2: launch_dma_t<128, 1, 2>(a, s); break;
                case 3: launch_dma_t<128, 1, 3>(a, s); break;
                case 4: launch_dma_t<128, 1, 4>(a, s); break;
                case 7: launch_dma_t<128, 1, 7>(a, s); break;
                case 8: launch_dma_t<128, 1, 8>(a, s); break;
                case 16: launch_dma_t<128, 1, 0, 1>(a, s); break;      // ping-pong phase stagger (slower, kept for reference)
                default: launch_dma_t<128, 1>(a, s);
            }
        } else if (abl == 16) launch_dma_t<64, 1, 0, 1>(a, s);
        else launch_dma_t<64, 1>(a, s);
#else
        if (a.N % 128 == 0) launch_dma_t<128, 1>(a, s); else launch_dma_t<64, 1>(a, s);
#endif
    } else {
        if (a.N % 128 == 0) launch_dma_t<128, 0>(a, s); else launch_dma_t<64, 0>(a, s);
    }
}

}  // namespace nhans
