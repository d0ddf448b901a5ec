// Implicit-GEMM convolution, halo-reuse variant ("v5") of the LDS-DMA kernel in conv_igemm_dma.hip:
// same contract, data layouts, packed weights, MFMA arrangement and epilogue; different staging of
// the activation operand and a software-pipelined K loop.
//
// Why: in conv_igemm_dma.hip every (filter tap, 32-channel chunk) re-stages its own 256 x 128 B
// activation tile, and a wave's K-loop iteration is strictly DMA issue -> fragment reads -> MFMAs ->
// wait/barrier.  Both waves of a SIMD run those phases in lockstep, so the matrix pipe idles while
// they issue DMAs and wait for LDS (profiles/r01: ~1,000 of ~2,500 cycles per chunk).  Here:
//
//   * For a stride-1 convolution the KW taps of one filter row read the SAME input pixels shifted by
//     one column.  The K loop runs over super-chunks (kh, 32-channel chunk); each stages ONE halo
//     image of the tile -- its pixel run plus the KW-1 padding columns per image row it touches --
//     and the KW taps read it at a row offset of kw.  Activation bytes and DMA instructions per MFMA
//     drop by KW (3 or 4).  Padding columns and rows are DMA'd from the zero page, so a tap needs no
//     per-lane validity test at all.
//         LDS row of tile pixel r (image row i(r) within the tile) for tap kw:  r + (KW-1)*i(r) + kw
//   * The halo image is double-buffered (2 x 320 rows x 128 B = 80 KB); the weight chunks (BN x 128 B
//     per tap) run through their own 4-stage ring.  The 1x1 strided `_transform` segment (KW = 1)
//     stages one row per output pixel.
//   * Register pipelining at half-tap granularity: the MFMA operands of the second half of tap `it`
//     are read from LDS (and the next weight DMA is issued) right before the MFMAs of its first half,
//     the operands of the first half of tap it+1 right before the MFMAs of the second half, so LDS
//     latency and DMA issue overlap matrix work of the same wave.  No extra registers, no extra
//     barrier: one counted `s_waitcnt vmcnt(n)` + raw `s_barrier` per tap, in the middle of it.
//   * DMA issue is unconditional (dummy loads from the zero page / the last weight chunk past the
//     end) so that the vmcnt immediates are static.
//
// Eligibility (checked by the launcher, otherwise conv_igemm_dma.hip runs): segment 0 has KW >= 3,
// column stride 1 and SAME padding (Wo == W), segment 1 (if any) has KW == 1, and the halo image of
// any 256-pixel run fits 320 rows.
#include "conv_epilogue.h"
#include <cstdlib>

namespace nhans {

namespace {
constexpr int HBM = 256;      // output pixels per workgroup
constexpr int HR = 320;       // rows of one halo image (5 DMA instructions x 64 rows)
constexpr int NA = 5;         // activation DMA instructions per thread per super-chunk
constexpr int BST = 4;        // weight ring stages

template <int N> __device__ __forceinline__ void halo_wait_vmcnt() {
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

}  // namespace

template <int BN, int PREC, int PP = 1, int DBG = 0>   // PP: ping-pong K loop; DBG: dev tool, per-workgroup cycle stamps (tools/conv_phase_cycles.py)
__global__ void __launch_bounds__(512) conv_igemm_halo(const ConvArgs a) {
    constexpr int TM = 2;                              // wave grid 4 (pixels) x 2 (channels), wave tile 64 x BN/2
    constexpr int TN = BN / 64;
    constexpr int A_BUF = HR * 32;                     // floats
    constexpr int B_STAGE = 32 * BN;                   // floats
    constexpr int B_BASE = 2 * A_BUF;
    constexpr int GB = BN / 64;                        // weight DMA instructions per thread per tap
    extern __shared__ __attribute__((aligned(16))) float smem[];

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1;

    // XCD-aware, bijective remap of the linear workgroup id
    const int ntn = a.N / BN;
    int L;
    {
        const int nblk = gridDim.x, bid = blockIdx.x;
        const int q = nblk >> 3, r = nblk & 7, xcd = bid & 7, idx = bid >> 3;
        L = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
    }
    const int mt = L / ntn, nt = L - mt * ntn;
    const int m0 = mt * HBM;
    const int nt0 = nt * (BN / 32);

    const int Wo = (int)a.fdWo.d;
    const int R0 = (int)fd_div((uint32_t)m0, a.fdWo);  // first output row (over all images) of the tile
    const int w0 = m0 - R0 * Wo;
    const int nrows_all = a.M / Wo;                    // B * Ho

    // ---- activation DMA assignment: instruction d of wave w fills LDS rows d*64 + w*8 .. +7, 8 lanes
    // per row; lane slot s = lane&7 fetches source piece s ^ ((row>>1)&7) (XOR swizzle applied at the
    // source).  A 256-B bank row holds two 128-B pixel rows, so the 16-byte bank slot of (row, piece) is
    // (row&1)*8 + piece^f(row); a ds_read_b128 lane group covers rows {r..r+3, r+12..r+15, r+20..r+27}
    // of one fragment, and f = (row>>1)&7 is the choice that makes those 16 slots distinct for any r.
    // Per row: element offset of its pixel for kh = 0 / chunk 0 (the launcher checks that tensors stay
    // below 2^31 elements) and the input row hi0 of kh = 0, or a sentinel for padding / unused rows.
    const int slot = lane & 7;
    int poff0, poff1, poff2, poff3, poff4;
    int hov0, hov1, hov2, hov3, hov4;
    int sH = 0, sW = 0, sC = 0;
    const float* ssrc = nullptr;

#define NH_MAP_ROW(D, POFF, HOV)                                                                   \
    {                                                                                              \
        const int j = (D) * 64 + wave * 8 + (lane >> 3);                                           \
        const int sp = (slot ^ ((j >> 1) & 7)) * 4;                                                \
        int Rg, wi;                                                                                \
        bool ok;                                                                                   \
        if (g.KW > 1) {                                                                            \
            const int n0 = Wo - w0 + g.KW - 1;                                                     \
            int i = 0, cj = w0 + j;                                                                \
            if (j >= n0) {                                                                         \
                const int jj = j - n0;                                                             \
                const int q = (int)fd_div((uint32_t)jj, a.fdWP);                                   \
                i = 1 + q;                                                                         \
                cj = jj - q * (int)a.fdWP.d;                                                       \
            }                                                                                      \
            wi = cj - g.pl;                                                                        \
            Rg = R0 + i;                                                                           \
            ok = Rg < nrows_all && (unsigned)wi < (unsigned)g.W;                                   \
        } else {                                                                                   \
            const int m = m0 + j;                                                                  \
            ok = j < HBM && m < a.M;                                                               \
            Rg = (int)fd_div((uint32_t)(ok ? m : 0), a.fdWo);                                      \
            wi = ((ok ? m : 0) - Rg * Wo) * g.sw - g.pl;                                           \
        }                                                                                          \
        if (!ok) { Rg = 0; wi = 0; }                                                               \
        const int b = (int)fd_div((uint32_t)(Rg * Wo), a.fdHoWo);                                  \
        const int hi0 = (Rg - b * a.Ho) * g.sh - g.pt;                                             \
        POFF = ((b * g.H + hi0) * g.W + wi) * g.C + sp;                                            \
        HOV = ok ? hi0 : -(1 << 28);                                                               \
    }
#define NH_MAP_SEGMENT(S)                                                                          \
    {                                                                                              \
        const ConvSeg& g = a.seg[S];                                                               \
        sH = g.H; sW = g.W; sC = g.C; ssrc = g.src;                                                \
        NH_MAP_ROW(0, poff0, hov0) NH_MAP_ROW(1, poff1, hov1) NH_MAP_ROW(2, poff2, hov2)           \
        NH_MAP_ROW(3, poff3, hov3) NH_MAP_ROW(4, poff4, hov4)                                      \
    }

#define NH_GLDS(SRC, DST)                                                                          \
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(SRC),        \
                                     (__attribute__((address_space(3))) void*)(DST), 16, 0, 0);

    // segment shapes as scalars (segment 1 = optional 1x1 `_transform`)
    const int nseg = a.nseg;
    const int KW0 = a.seg[0].KW, KH0 = a.seg[0].KH, CC0 = a.seg[0].C >> 5;
    const int KH1 = nseg > 1 ? a.seg[1].KH : 0, CC1 = nseg > 1 ? (a.seg[1].C >> 5) : 0;
    const int nsup0 = KH0 * CC0;                       // super-chunks of segment 0
    const int nsup = nsup0 + KH1 * CC1;
    const int total = nsup0 * KW0 + KH1 * CC1;         // taps = weight chunks
    const size_t bstride = (size_t)(a.N / 32) * 1024;

    // ---- activation cursor: next super-chunk to stage
    int segA = 0, khA = 0, ccA = 0, supA = 0;
    const int zpiece0 = (slot ^ ((lane >> 3) & 7)) * 4;   // any in-page offset will do for dummy rows
#define NH_A_PTR(POFF, HOV)                                                                        \
    ((supA < nsup && (unsigned)((HOV) + khA) < (unsigned)sH) ? ssrc + ((POFF) + khoff_) : a.zero + zpiece0)
#define NH_ISSUE_A(BUF)                                                                            \
    {                                                                                              \
        const int khoff_ = khA * sW * sC + ccA * 32;                                               \
        float* sa_ = smem + (BUF) * A_BUF + wave * 8 * 32;                                         \
        const float* p0_ = NH_A_PTR(poff0, hov0);                                                  \
        const float* p1_ = NH_A_PTR(poff1, hov1);                                                  \
        const float* p2_ = NH_A_PTR(poff2, hov2);                                                  \
        const float* p3_ = NH_A_PTR(poff3, hov3);                                                  \
        const float* p4_ = NH_A_PTR(poff4, hov4);                                                  \
        NH_GLDS(p0_, sa_)                                                                          \
        NH_GLDS(p1_, sa_ + 64 * 32)                                                                \
        NH_GLDS(p2_, sa_ + 128 * 32)                                                               \
        NH_GLDS(p3_, sa_ + 192 * 32)                                                               \
        NH_GLDS(p4_, sa_ + 256 * 32)                                                               \
    }
#define NH_ADVANCE_A()                                                                             \
    {                                                                                              \
        ++supA;                                                                                    \
        if (++ccA >= (segA ? CC1 : CC0)) {                                                         \
            ccA = 0;                                                                               \
            if (++khA >= (segA ? KH1 : KH0)) {                                                     \
                khA = 0;                                                                           \
                if (segA == 0 && nseg > 1) {                                                       \
                    segA = 1;                                                                      \
                    NH_MAP_SEGMENT(1)                                                              \
                }                                                                                  \
            }                                                                                      \
        }                                                                                          \
    }

    // ---- weight cursor: the packed weights are stored in the order the taps are walked (fold.py
    // kmat), so the next tap is the next `bstride` floats; one jump to the transform's array, and a
    // stall on the last chunk for the dummy loads past the end.  (Kept branch-free: the DMAs must sit
    // in one basic block with the MFMAs they are interleaved with.)
    const int ntap0 = nsup0 * KW0;
    int tapB = 0;
    const float* bp_ = a.seg[0].wpk + (size_t)nt0 * 1024 - bstride;
    const float* const wpk1 = (nseg > 1 ? a.seg[1].wpk : a.seg[0].wpk) + (size_t)nt0 * 1024;
#define NH_ISSUE_B(ST)                                                                             \
    {                                                                                              \
        const float* nx_ = tapB == ntap0 ? wpk1 : bp_ + bstride;                                   \
        bp_ = tapB < total ? nx_ : bp_;                                                            \
        ++tapB;                                                                                    \
        float* sb_ = smem + B_BASE + (ST) * B_STAGE;                                               \
        _Pragma("unroll") for (int j = 0; j < GB; ++j)                                             \
            NH_GLDS(bp_ + (j * 512 + tid) * 4, sb_ + (j * 512 + wave * 64) * 4)                    \
    }

    // ---- fragment addresses.  Tile pixel r = wm*64 + t*32 + (lane&31) sits in halo row
    // r + (KW-1)*i(r) (+ kw per tap) while segment 0 runs, in row r for the transform segment.
    const int g8 = lane >> 5;
    int jb0[TM], jb1[TM];
#pragma unroll
    for (int t = 0; t < TM; ++t) {
        const int r = wm * 64 + t * 32 + (lane & 31);
        const int irow = (int)fd_div((uint32_t)(m0 + r), a.fdWo) - R0;
        jb0[t] = r + (KW0 - 1) * irow;
        jb1[t] = r;
    }
    const int bcol = (wn * TN) * 1024 + lane * 4;

    // MFMA operands: k-steps [0, KH_) of a tap are "half 0", the rest "half 1"; each half has its
    // own registers so that one half is being read while the other is multiplied.
    constexpr int KS = PREC == 1 ? 2 : 4;              // k-steps per chunk (16 k each / 8 k each)
    constexpr int KH_ = KS / 2;
    f32x4 fa_hi[KS][TM], fa_lo[PREC == 1 ? KS : 1][TM], fb_hi[KS][TN], fb_lo[PREC == 1 ? KS : 1][TN];
#define NH_READ_HALF(H, ABUF, BSTG, SEG_, KW_)                                                     \
    {                                                                                              \
        const float* Sa_ = smem + (ABUF) * A_BUF;                                                  \
        const float* Sb_ = smem + B_BASE + (BSTG) * B_STAGE + bcol;                                \
        _Pragma("unroll") for (int t = 0; t < TM; ++t) {                                           \
            const int jr_ = ((SEG_) ? jb1[t] : jb0[t]) + (KW_);                                    \
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
    // (the three split products -- or the four k-pairs of an f32 quad -- of one accumulator are issued
    // TM*TN MFMAs apart, so consecutive MFMAs never chain on the same accumulator)
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

    // Cursor of the tap being multiplied: segment, kw, halo buffer, super-chunk count.
    int segC = 0, kwC = 0, bufC = 0, supC = 0;
    long long dbg_p1 = 0, dbg_vm = 0, dbg_bar = 0, dbg_t0 = 0;

    if constexpr (PP) {
        // ---- Ping-pong K loop.  Waves w and w+4 share a SIMD.  A tap is two phases per wave, each
        // closed by a workgroup barrier: P1 = memory (issue the DMAs of a later tap, read all MFMA
        // operands of this tap from LDS, counted wait for the own shares of the next tap); P2 = the
        // 24 (12) MFMAs from registers at raised priority.  Waves 4-7 run one barrier behind, so on
        // every SIMD one wave is in P2 while the other is in P1: the older wave of a SIMD otherwise
        // wins every arbitration, finishes early and idles at the barrier while the younger one does
        // its memory phase with the matrix pipe empty (measured: 650 vs 80 cycles of barrier wait per
        // tap).  Slot k = 2*tap (waves 0-3) / 2*tap+1 (waves 4-7):
        //   tap c is read in slots 2c and 2c+1; its weight stage is refilled with tap c+4 -- issued as
        //   "tap it+3" in slots 2c+2 / 2c+3 -- and the halo buffer of super-chunk u, last read by its
        //   last tap L, with image u+2, issued at the first tap of u+1 (slots 2L+2 / 2L+3);
        //   every wave waits for its own shares of tap it+1 at the end of P1(it), i.e. before the
        //   barriers that precede slot 2(it+1), and drains its ds_reads before leaving P1.
        // Queue order per iteration j: [image if first(j)], weights j+3.
        NH_MAP_SEGMENT(0)
        NH_ISSUE_A(0)
        NH_ADVANCE_A()
        NH_ISSUE_B(0)
        NH_ISSUE_B(1)
        NH_ISSUE_B(2)
        halo_wait_vmcnt<2 * GB>();
        __builtin_amdgcn_s_barrier();
        if constexpr (DBG) dbg_t0 = (long long)__builtin_amdgcn_s_memtime();
        const bool grp_y = wave >= 4;
        if (grp_y) __builtin_amdgcn_s_barrier();
        bool prev_first = false;
        for (int it = 0; it < total; ++it) {
            const int KWc = segC ? 1 : KW0;
            const bool first = kwC == 0;
            long long tq0 = 0, tq1 = 0, tq2 = 0;
            if constexpr (DBG) tq0 = (long long)__builtin_amdgcn_s_memtime();
            if (first) {
                NH_ISSUE_A(bufC ^ 1)
                NH_ADVANCE_A()
            }
            NH_ISSUE_B((it + 3) & (BST - 1))
            NH_READ_HALF(0, bufC, it & (BST - 1), segC, kwC)
            NH_READ_HALF(1, bufC, it & (BST - 1), segC, kwC)
            if constexpr (DBG) tq1 = (long long)__builtin_amdgcn_s_memtime();
            if (KWc == 1) halo_wait_vmcnt<GB>();
            else if (first || prev_first) halo_wait_vmcnt<2 * GB + NA>();
            else halo_wait_vmcnt<2 * GB>();
            if constexpr (DBG) tq2 = (long long)__builtin_amdgcn_s_memtime();
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_sched_barrier(0);
            __builtin_amdgcn_s_barrier();
            if constexpr (DBG) {
                const long long tq3 = (long long)__builtin_amdgcn_s_memtime();
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                dbg_p1 += tq1 - tq0; dbg_vm += tq2 - tq1; dbg_bar += tq3 - tq2;
            }
            __builtin_amdgcn_sched_barrier(0);
            __builtin_amdgcn_s_setprio(1);
            NH_MFMA_HALF(0)
            NH_MFMA_HALF(1)
            __builtin_amdgcn_s_setprio(0);
            __builtin_amdgcn_sched_barrier(0);
            __builtin_amdgcn_s_barrier();
            __builtin_amdgcn_sched_barrier(0);
            prev_first = first;
            if (++kwC >= KWc) {
                kwC = 0;
                bufC ^= 1;
                if (++supC == nsup0) segC = 1;
            }
        }
        if (!grp_y) __builtin_amdgcn_s_barrier();
    } else {
        // ---- Lockstep K loop with half-tap register pipelining.
        // prologue: halo images 0 and 1 and weight taps 0..2 in flight; wait for image 0 and tap 0
        // (image 1 precedes tap 0 in the queue) and read half 0 of tap 0.
        NH_MAP_SEGMENT(0)
        NH_ISSUE_A(0)
        NH_ADVANCE_A()
        NH_ISSUE_A(1)
        NH_ADVANCE_A()
        NH_ISSUE_B(0)
        NH_ISSUE_B(1)
        NH_ISSUE_B(2)
        halo_wait_vmcnt<2 * GB>();
        __builtin_amdgcn_s_barrier();
        bool last1 = false, last2 = false;              // tap it-1 / it-2 closed a super-chunk
        NH_READ_HALF(0, 0, 0, 0, 0)

        // One tap `it`:
        //   issue the weights of tap it+3 (their stage was freed by the barrier of it-1); read half 1
        //   of tap it; MFMAs of half 0;
        //   counted wait + barrier: the operands of tap it+1 have landed in every wave and every wave
        //   holds all of tap it in registers -- which frees the weight stage of tap it and, if tap it
        //   closes a super-chunk, its halo buffer: refill that one with the image after next;
        //   read half 0 of tap it+1; MFMAs of half 1.
        // Queue order per iteration j: weights j+3, [halo image if last(j)].  Needed at the barrier of
        // iteration it: the weights of tap it+1 (issued at it-2) and the image that tap it+1 may open,
        // issued >= KW >= 3 iterations ago -- except inside the KW = 1 transform segment, where it was
        // issued at it-1 and only the weights of tap it+3 may still be in flight.
        // (Interleaving the reads and DMAs 1:1 between the MFMAs with sched_group_barrier measured 4 %
        // slower than issuing them ahead of the burst.)
        if constexpr (DBG) dbg_t0 = (long long)__builtin_amdgcn_s_memtime();
        for (int it = 0; it < total; ++it) {
            const int KWc = segC ? 1 : KW0;
            const bool lastC = kwC + 1 >= KWc;
            NH_ISSUE_B((it + 3) & (BST - 1))
            NH_READ_HALF(1, bufC, it & (BST - 1), segC, kwC)
            __builtin_amdgcn_sched_barrier(0);
            NH_MFMA_HALF(0)
            __builtin_amdgcn_sched_barrier(0);
            long long tq0 = 0, tq1 = 0;
            if constexpr (DBG) tq0 = (long long)__builtin_amdgcn_s_memtime();
            if (KWc == 1) halo_wait_vmcnt<GB>();
            else if (last1 || last2) halo_wait_vmcnt<2 * GB + NA>();
            else halo_wait_vmcnt<2 * GB>();
            if constexpr (DBG) tq1 = (long long)__builtin_amdgcn_s_memtime();
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            if constexpr (DBG) {
                const long long tq2 = (long long)__builtin_amdgcn_s_memtime();
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                dbg_vm += tq1 - tq0; dbg_bar += tq2 - tq1;
            }
            __builtin_amdgcn_sched_barrier(0);
            if (lastC) {
                NH_ISSUE_A(bufC)
                NH_ADVANCE_A()
            }
            last2 = last1; last1 = lastC;
            if (lastC) {
                kwC = 0;
                bufC ^= 1;
                if (++supC == nsup0) segC = 1;
            } else ++kwC;
            NH_READ_HALF(0, bufC, (it + 1) & (BST - 1), segC, kwC)       // (past the last tap: a harmless read)
            __builtin_amdgcn_sched_barrier(0);
            NH_MFMA_HALF(1)
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    halo_wait_vmcnt<0>();                               // dummy DMAs past the end still target LDS
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    if constexpr (DBG) {        // per wave 0 / wave 7: [loop cycles, P1 issue+reads, vmcnt wait, lgkm+barrier wait]
        if (a.dbg && (tid == 0 || tid == 448)) {
            long long* d = a.dbg + ((size_t)blockIdx.x * 2 + (tid ? 1 : 0)) * 4;
            d[0] = (long long)__builtin_amdgcn_s_memtime() - dbg_t0; d[1] = dbg_p1; d[2] = dbg_vm; d[3] = dbg_bar;
        }
    }


#undef NH_MAP_ROW
#undef NH_MAP_SEGMENT
#undef NH_GLDS
#undef NH_A_PTR
#undef NH_ISSUE_A
#undef NH_ISSUE_B
#undef NH_READ_HALF
#undef NH_MFMA_HALF

    static_assert(conv_epilogue_lds_bytes<HBM, BN>() <= (size_t)(2 * A_BUF + BST * B_STAGE) * sizeof(float), "epilogue LDS");
    conv_epilogue<TM, TN, PREC, 512, HBM, BN>(a, acc, smem, m0, wm * 64, wn * TN * 32, nt * BN, tid, lane);
}

template <int BN, int PREC, int PP = 1, int DBG = 0> static void launch_halo_t(const ConvArgs& a, hipStream_t s) {
    constexpr size_t lds = (size_t)(2 * HR * 32 + BST * 32 * BN) * sizeof(float);
    static bool attr_set = false;
    if (!attr_set) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_igemm_halo<BN, PREC, PP, DBG>),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        attr_set = true;
    }
    const int mtiles = (a.M + HBM - 1) / HBM;
    const int grid = mtiles * (a.N / BN);
    hipLaunchKernelGGL((conv_igemm_halo<BN, PREC, PP, DBG>), dim3(grid), dim3(512), lds, s, a);
}

bool conv_igemm_halo_eligible(const ConvArgs& a) {
    const ConvSeg& g = a.seg[0];
    // KW >= 3: the counted waits assume a halo image is issued at least two taps before its first use
    if (g.KW < 3 || g.sw != 1 || a.Wo != g.W || g.pl < 0 || g.pl >= g.KW) return false;
    if (a.nseg > 1 && a.seg[1].KW != 1) return false;
    if (a.nseg > 2 || a.M % a.Wo != 0) return false;
    // 32-bit element offsets inside the kernel
    for (int i = 0; i < a.nseg; ++i) {
        const ConvSeg& q = a.seg[i];
        const double elems = (double)(a.M / (a.Ho * a.Wo)) * q.H * q.W * q.C;
        if (elems + 65536.0 >= 2147483648.0) return false;
    }
    // rows of the halo image of a 256-pixel run that starts at the last column of an image row
    const int nrows = (a.Wo - 1 + HBM - 1) / a.Wo + 1;
    return HBM + (g.KW - 1) * nrows <= HR;
}

void launch_conv_igemm_halo(const ConvArgs& a0, hipStream_t s) {
    ConvArgs a = a0;
    a.fdWP = make_fastdiv((uint32_t)(a.Wo + a.seg[0].KW - 1));
    // NHANS_HALO_LOCKSTEP=1 selects the lockstep K loop (A/B experiments)
    static const bool pp = [] { const char* e = getenv("NHANS_HALO_LOCKSTEP"); return !(e && atoi(e)); }();
    const bool wide = a.N % 128 == 0;
    if (a.prec == 1 && a.dbg) {
        if (pp) { if (wide) launch_halo_t<128, 1, 1, 1>(a, s); else launch_halo_t<64, 1, 1, 1>(a, s); }
        else { if (wide) launch_halo_t<128, 1, 0, 1>(a, s); else launch_halo_t<64, 1, 0, 1>(a, s); }
    } else if (a.prec == 1) {
        if (pp) { if (wide) launch_halo_t<128, 1, 1>(a, s); else launch_halo_t<64, 1, 1>(a, s); }
        else { if (wide) launch_halo_t<128, 1, 0>(a, s); else launch_halo_t<64, 1, 0>(a, s); }
    } else {
        if (pp) { if (wide) launch_halo_t<128, 0, 1>(a, s); else launch_halo_t<64, 0, 1>(a, s); }
        else { if (wide) launch_halo_t<128, 0, 0>(a, s); else launch_halo_t<64, 0, 0>(a, s); }
    }
}

}  // namespace nhans
