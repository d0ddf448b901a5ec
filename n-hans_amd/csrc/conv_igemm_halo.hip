// Implicit-GEMM convolution, halo-reuse / wave-specialised variant ("v5") of the LDS-DMA kernel in
// conv_igemm_dma.hip: same contract, data layouts, packed weights, MFMA arrangement and epilogue;
// different staging of the activation operand and a producer/consumer split of the workgroup.
//
// Why (profiles/r01, tools/halo_phase_cycles.py): in conv_igemm_dma.hip a K-loop iteration of a wave
// is DMA issue -> fragment reads -> MFMAs -> counted wait -> barrier, all eight waves in lockstep,
// and ~1,000 of its ~2,500 cycles per chunk pass with the matrix pipe idle.  Cycle stamps put the
// cost on the LDS-DMA issue itself: a `global_load_lds` wave-instruction holds its wave for 100-185
// cycles, a chunk needs 6 of them per wave, and neither register pipelining, nor interleaving them
// between MFMAs, nor a ping-pong of the two waves of a SIMD takes them off the critical path while
// the MFMA waves issue them.  Here:
//
//   * Halo reuse.  For a stride-1 convolution the KW taps of one filter row read the SAME input
//     pixels shifted by one column.  The K loop runs over super-chunks (kh, 32-channel chunk); each
//     stages ONE halo image of the tile -- its pixel run plus the KW-1 padding columns per image row
//     it touches -- and the KW taps read it at a row offset of kw.  Activation bytes and DMA
//     instructions per MFMA drop by KW (3 or 4).  Padding columns and rows are DMA'd from the zero
//     page, so a tap needs no per-lane validity test at all.
//         LDS row of tile pixel r (image row i(r) within the tile) for tap kw:  r + (KW-1)*i(r) + kw
//     The halo image is double-buffered (2 x 320 rows x 128 B = 80 KB); the weight chunks (BN x 128 B
//     per tap) run through their own 4-stage ring.  The 1x1 strided `_transform` segment (KW = 1)
//     stages one row per output pixel.
//   * Wave specialisation.  The workgroup is 8 consumer waves (2 per SIMD; wave grid 4 x 2, wave tile
//     64 x BN/2) that only read LDS and issue MFMAs, plus 4 producer waves (1 per SIMD) that issue
//     every DMA, keep the counted `s_waitcnt vmcnt(n)` bookkeeping and meet the consumers at one raw
//     `s_barrier` per tap.  A producer needs the SIMD's issue slots, not its matrix pipe.
//   * Consumers pipeline in registers at half-tap granularity: the operands of the second half of
//     tap `it` are read right before the MFMAs of its first half, those of the first half of tap it+1
//     right before the MFMAs of the second half; the barrier sits between the halves.
//   * DMA issue is unconditional (dummy loads from the zero page / the last weight chunk past the
//     end) so that the vmcnt immediates are static.
//
// Eligibility (checked by the launcher, otherwise conv_igemm_dma.hip runs): segment 0 has KW >= 3,
// column stride 1 and SAME padding (Wo == W), segment 1 (if any) has KW == 1, tensors stay below
// 2^31 elements and the halo image of any 256-pixel run fits 320 rows.
#include "conv_epilogue.h"
#include <cstdlib>

namespace nhans {

namespace {
constexpr int NCW = 8;        // consumer (MFMA) waves
constexpr int NPW = 4;        // producer (DMA) waves

// Tile shapes.  The work of a consumer wave is always 64 pixels x 64 channels (TM = TN = 2: 8 operand
// reads feed 12 MFMAs per k-step) except in the 256 x 64 shape:
//   HBM 256, BN 128: wave grid 4 x 2, halo image 320 rows, 4 weight stages (80 + 64 KB)   N >= 128
//   HBM 512, BN  64: wave grid 8 x 1, halo image 544 rows, 3 weight stages (136 + 24 KB = all 160 KB
//                    of the CU)                                                           N = 64
//   HBM 256, BN  64: wave grid 4 x 2, wave tile 64 x 32 (6 reads per 6 MFMAs)             kept for A/B
// The 512-pixel shape exists because the 64-channel convs (K = 1024: 32 taps) are the ones where the
// per-tile fixed costs -- first halo image, epilogue round trips -- weigh most and where a 64 x 32
// wave tile makes the LDS operand reads a co-bottleneck (profiles/r01).
template <int HBM_> struct HaloShape {
    static constexpr int HBM = HBM_;                         // output pixels per workgroup
    static constexpr int HR = HBM_ == 512 ? 544 : 320;       // rows of one halo image (multiple of 32)
    static constexpr int BST = HBM_ == 512 ? 3 : 4;          // weight ring stages; tap it+BST-1 is prefetched during tap it
    static constexpr int WN = HBM_ == 512 ? 1 : 2;           // consumer wave grid (NCW / WN) x WN
    static constexpr int NAP = HR * 8 / (NPW * 64);          // activation DMA instructions per producer thread per image
};

template <int N> __device__ __forceinline__ void halo_wait_vmcnt() {
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}
}  // namespace

// PWM ("pointwise mode"): the same producer / consumer pipeline for the convs WITHOUT halo reuse -- strided convs, the
// 5 x 1 VALID head conv -- that conv_igemm_dma.hip otherwise runs with every wave issuing DMA and MFMAs in lockstep.
// Every tap (chunk, kh, kw) stages its own image of one 128-byte row per output pixel; with 256 rows instead of 320
// THREE images fit beside the four weight stages (3 x 32 KB + 4 x 16 KB = the CU's 160 KB), so the image of tap it+2 is
// issued while tap it is multiplied: one tap of lookahead where the two-buffer scheme of the 1 x 1 `_transform`
// segment has none.
template <int BN, int PREC, int HBM_ = 256, int DBG = 0, int ABL = 0, int PWM = 0>   // DBG: dev tool, per-workgroup cycle stamps (tools/halo_phase_cycles.py); ABL: timing ablations (wrong results)
__global__ void __launch_bounds__((NCW + NPW) * 64) conv_igemm_halo(const ConvArgs a) {
    using SH = HaloShape<HBM_>;
    constexpr int HBM = SH::HBM, HR = SH::HR, BST = SH::BST, WN = SH::WN, NAP = SH::NAP;
    constexpr int PFD = BST - 1;                       // weight prefetch distance in taps
    constexpr int TM = 2;
    constexpr int TN = BN / (32 * WN);
    static_assert(HBM == (NCW / WN) * TM * 32 && BN == WN * TN * 32, "wave grid");
    static_assert(!PWM || (HBM == 256 && BST == 4), "pointwise mode: 256-pixel tiles, four weight stages");
    constexpr int A_BUF = (PWM ? HBM : HR) * 32;       // floats
    constexpr int NABUF = PWM ? 3 : 2;                 // image buffers
    constexpr int NAPW = PWM ? HBM * 8 / (NPW * 64) : NAP;   // activation DMA instructions per producer thread per image
    constexpr int B_STAGE = 32 * BN;                   // floats
    constexpr int B_BASE = NABUF * A_BUF;
    constexpr int GBP = B_STAGE / 4 / (NPW * 64);      // weight DMA instructions per producer thread per tap (4 / 2)
    extern __shared__ __attribute__((aligned(16))) float smem[];

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    long long dbg_entry = 0;
    if constexpr (DBG) dbg_entry = (long long)__builtin_amdgcn_s_memtime();

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

    // segment shapes as scalars (segment 1 = optional 1x1 `_transform`); taps are walked in the order
    // (segment, 32-channel chunk, kh, kw), which is the order fold.py kmat() packs the weights in: the KH halo
    // images of a chunk are the same pixels shifted by one image row, staged back to back, so all but the first come
    // out of the L2 (in the order (kh, chunk) of rounds 1-2 a whole pass over the channels lay between them and every
    // filter row fetched its image rows from HBM again: 2.2 x the algorithmic reads)
    const int nseg = a.nseg;
    const int KW0 = a.seg[0].KW, KH0 = a.seg[0].KH, CC0 = a.seg[0].C >> 5;
    const int KH1 = nseg > 1 ? a.seg[1].KH : 0, CC1 = nseg > 1 ? (a.seg[1].C >> 5) : 0;
    const int nsup0 = KH0 * CC0 * (PWM ? KW0 : 1);     // super-chunks (= staged images) of segment 0
    const int nsup = nsup0 + KH1 * CC1;
    const int ntap0 = PWM ? nsup0 : nsup0 * KW0;
    const int total = ntap0 + KH1 * CC1;               // taps = weight chunks

    // cursor of the current tap: segment, kw within the super-chunk, halo buffer, super-chunk count
    int segC = 0, kwC = 0, bufC = 0, supC = 0;
#define NH_NEXT_TAP()                                                                              \
    if constexpr (PWM) {                                                                           \
        bufC = bufC == NABUF - 1 ? 0 : bufC + 1;                                                   \
    } else if (++kwC >= (segC ? 1 : KW0)) {                                                        \
        kwC = 0;                                                                                   \
        bufC ^= 1;                                                                                 \
        if (++supC == nsup0) segC = 1;                                                             \
    }

    if (wave >= NCW) {
        // =========================================================================================
        // Producer waves.  Iteration `it` (between the barriers of taps it-1 and it): issue the halo
        // image of the NEXT super-chunk if tap it opens one (its buffer was last read by the tap
        // before), issue the weights of tap it+PFD (their stage was last read by tap it-1; PFD =
        // BST-1 = 3, or 2 in the 512-pixel shape), then wait until the own shares of everything tap
        // it+1 reads have landed.
        // Queue order per iteration j: [image if first(j)], weights j+PFD.  Needed at the barrier of
        // iteration it: the weights of tap it+1 (issued at it+1-PFD) and the image tap it+1 may open,
        // issued >= KW >= 3 iterations ago -- except inside the KW = 1 transform segment, where it
        // was issued in this very iteration, before the weights of tap it+PFD.  Younger than the
        // weights of tap it+1 are PFD-1 weight groups and the images issued in the last PFD-1
        // iterations (this one included).
        const int pw = wave - NCW, ptid = tid - NCW * 64;
        if constexpr (!(ABL & 64)) __builtin_amdgcn_s_setprio(3);   // the one wave per SIMD everybody waits for (measured +3 %)
        const int slot = lane & 7;
        const int nrows_all = a.M / Wo;                // B * Ho
        const size_t bstride = (size_t)(a.N / 32) * 1024;
        // Activation DMA assignment: instruction d of producer wave p fills LDS rows d*32 + p*8 .. +7,
        // 8 lanes per row; lane slot s = lane&7 fetches source piece s ^ ((row>>1)&7) (XOR swizzle
        // applied at the source).  A 256-B bank row holds two 128-B pixel rows, so the 16-byte bank
        // slot of (row, piece) is (row&1)*8 + piece^f(row); a ds_read_b128 lane group covers rows
        // {r..r+3, r+12..r+15, r+20..r+27} of one fragment, and f = (row>>1)&7 makes those 16 slots
        // distinct for any r.  Per row: element offset of its pixel for kh = 0 / chunk 0 and the input
        // row hi0 of kh = 0, or a sentinel for padding / unused rows.
        // The image requests go through a buffer descriptor over the input from the tile's first frame on: a lane whose
        // offset is out of range gets zeros written to LDS (tools/ubench/buffer_lds_oob.hip), padding needs no zero
        // page, a request no 64-bit pointer arithmetic and select.  (The tap's shift is added per lane, not passed as the
        // scalar offset: a padding row's own offset is negative and becomes valid only with the shift.)
        constexpr unsigned kOob = 0x80000000u;
        const int b0 = (int)fd_div((uint32_t)m0, a.fdHoWo);                      // (uniform) frame of the tile's first pixel
        int poff[NAPW], hov[NAPW], wiv[PWM ? NAPW : 1];
        int sH = 0, sW = 0, sC = 0;
        __amdgpu_buffer_rsrc_t srsrc;
#define NH_MAP_SEGMENT(S)                                                                          \
    {                                                                                              \
        const ConvSeg& g = a.seg[S];                                                               \
        sH = g.H; sW = g.W; sC = g.C;                                                              \
        {                                                                                          \
            const size_t fb_ = (size_t)g.H * g.W * g.C * 4;                                        \
            const size_t left_ = (size_t)(a.M / (a.Ho * Wo) - b0) * fb_;                           \
            srsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(g.src) + (size_t)b0 * (fb_ / 4), 0, \
                                                      (int)(left_ < 0x7FFFFFFFu ? left_ : 0x7FFFFFFFu), 0x00020000); \
        }                                                                                          \
        _Pragma("unroll") for (int d = 0; d < NAPW; ++d) {                                         \
            const int j = d * 32 + pw * 8 + (lane >> 3);                                           \
            const int sp = (slot ^ ((j >> 1) & 7)) * 4;                                            \
            int Rg, wi;                                                                            \
            bool ok;                                                                               \
            if (!PWM && g.KW > 1) {                                                                \
                const int n0 = Wo - w0 + g.KW - 1;                                                 \
                int i = 0, cj = w0 + j;                                                            \
                if (j >= n0) {                                                                     \
                    const int jj = j - n0;                                                         \
                    const int q = (int)fd_div((uint32_t)jj, a.fdWP);                               \
                    i = 1 + q;                                                                     \
                    cj = jj - q * (int)a.fdWP.d;                                                   \
                }                                                                                  \
                wi = cj - g.pl;                                                                    \
                Rg = R0 + i;                                                                       \
                ok = Rg < nrows_all && (unsigned)wi < (unsigned)g.W;                               \
            } else {                                                                               \
                const int m = m0 + j;                                                              \
                ok = j < HBM && m < a.M;                                                           \
                Rg = (int)fd_div((uint32_t)(ok ? m : 0), a.fdWo);                                  \
                wi = ((ok ? m : 0) - Rg * Wo) * g.sw - g.pl;                                       \
            }                                                                                      \
            if (!ok) { Rg = 0; wi = 0; }                                                           \
            const int b = (int)fd_div((uint32_t)(Rg * Wo), a.fdHoWo);                              \
            const int hi0 = (Rg - b * a.Ho) * g.sh - g.pt;                                         \
            poff[d] = ((((b - b0) * g.H + hi0) * g.W + wi) * g.C + sp) * 4;     /* bytes from the tile's first frame */ \
            hov[d] = ok ? hi0 : -(1 << 28);                                                        \
            if constexpr (PWM) wiv[d] = wi;              /* (column validity depends on the tap's kw) */ \
        }                                                                                          \
    }
#define NH_GLDS(SRC, DST)                                                                          \
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(SRC),        \
                                     (__attribute__((address_space(3))) void*)(DST), 16, 0, 0);

        // activation cursor: next super-chunk to stage
        int segA = 0, khA = 0, ccA = 0, supA = 0, kwA = 0;
#define NH_ISSUE_A(BUF)                                                                            \
    {                                                                                              \
        const int khoff_ = (khA * sW + (PWM ? kwA : 0)) * sC + ccA * 32;                           \
        const bool live_ = supA < nsup;                                                            \
        float* sa_ = smem + (BUF) * A_BUF + pw * 8 * 32;                                           \
        _Pragma("unroll") for (int d = 0; d < NAPW; ++d) {                                         \
            bool in_ = !(ABL & 1) && live_ && (unsigned)(hov[d] + khA) < (unsigned)sH;             \
            if constexpr (PWM) in_ = in_ && (unsigned)(wiv[d] + kwA) < (unsigned)sW;               \
            if constexpr (!(ABL & 2))                                                              \
                __builtin_amdgcn_raw_ptr_buffer_load_lds(srsrc, (__attribute__((address_space(3))) void*)(sa_ + d * 32 * 32), 16, \
                                                         in_ ? (unsigned)(poff[d] + khoff_ * 4) : kOob, 0, 0, 0); \
        }                                                                                          \
        ++supA;                                                                                    \
        if (PWM && ++kwA < KW0) {                 /* pointwise: the KW taps of a filter row first */ \
        } else if (kwA = 0, ++khA >= (segA ? KH1 : KH0)) { /* consecutive super-chunks shift by one row: L2 */ \
            khA = 0;                                                                               \
            if (++ccA >= (segA ? CC1 : CC0)) {                                                     \
                ccA = 0;                                                                           \
                if (segA == 0 && nseg > 1) {                                                       \
                    segA = 1;                                                                      \
                    NH_MAP_SEGMENT(1)                                                              \
                }                                                                                  \
            }                                                                                      \
        }                                                                                          \
    }
        // weight cursor: the next tap is the next `bstride` floats, with one jump to the transform's
        // array and a stall on the last chunk for the dummy loads past the end
        int tapB = 0;
        // (through buffer descriptors as well: the thread's 16 bytes of a tap's block at a fixed per-lane offset, the
        // tap in the scalar offset -- no per-request address arithmetic at all)
        const __amdgpu_buffer_rsrc_t wr0 = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.seg[0].wpk) + (size_t)nt0 * 1024, 0, 0x7FFFFFFF, 0x00020000);
        const __amdgpu_buffer_rsrc_t wr1 = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(nseg > 1 ? a.seg[1].wpk : a.seg[0].wpk) + (size_t)nt0 * 1024, 0, 0x7FFFFFFF, 0x00020000);
        const unsigned wvo = (unsigned)ptid * 16u;
        const int bstride_b = (int)(bstride * 4);
        int wtap = -1, wseg = 0;                        // tap within its segment's array (the last one again past the end)
#define NH_ISSUE_B(ST)                                                                             \
    {                                                                                              \
        if (tapB < total && !(ABL & 4)) {                                                          \
            if (tapB == ntap0) { wseg = 1; wtap = 0; } else ++wtap;                                \
        }                                                                                          \
        ++tapB;                                                                                    \
        float* sb_ = smem + B_BASE + (ST) * B_STAGE;                                               \
        const int so_ = (ABL & 4) ? 0 : wtap * bstride_b;                                          \
        _Pragma("unroll") for (int j = 0; j < GBP; ++j)                                            \
            if constexpr (!(ABL & 8))                                                              \
                __builtin_amdgcn_raw_ptr_buffer_load_lds((wseg || (ABL & 4)) ? wr1 : wr0,          \
                    (__attribute__((address_space(3))) void*)(sb_ + (j * (NPW * 64) + pw * 64) * 4), 16, wvo, so_ + j * (NPW * 64) * 16, 0, 0); \
    }

        NH_MAP_SEGMENT(0)
        NH_ISSUE_A(0)
        NH_ISSUE_B(0)
        if constexpr (PWM) NH_ISSUE_A(1)                // (behind tap 0's weights: image 0 and tap 0 are what the barrier waits for)
        NH_ISSUE_B(1)
        if constexpr (PFD == 3) NH_ISSUE_B(2)
        halo_wait_vmcnt<(PFD - 1) * GBP + (PWM ? NAPW : 0)>();   // image 0 and tap 0
        __builtin_amdgcn_s_barrier();
        long long dbg_is = 0, dbg_vm = 0, dbg_bar = 0, dbg_t0 = 0;
        if constexpr (DBG) dbg_t0 = (long long)__builtin_amdgcn_s_memtime();
        bool prev_first = false;
        int stB = PFD;                                  // ring stage of the tap being prefetched
        for (int it = 0; it < total; ++it) {
            const int KWc = segC ? 1 : KW0;
            const bool first = kwC == 0;
            long long tq0 = 0, tq1 = 0, tq2 = 0;
            if constexpr (DBG) tq0 = (long long)__builtin_amdgcn_s_memtime();
            if constexpr (PWM) {
                // the image of tap it+2, into the buffer tap it-1 has released; behind it in the queue only the weights
                // of tap it+3.  Needed at this iteration's barrier: image and weights of tap it+1 -- younger than those
                // are the weights of tap it+2 (issued one iteration ago), this image and these weights.
                NH_ISSUE_A(bufC == 0 ? NABUF - 1 : bufC - 1)
            } else if (first) NH_ISSUE_A(bufC ^ 1)
            NH_ISSUE_B(stB)
            if (++stB == BST) stB = 0;
            if constexpr (DBG) tq1 = (long long)__builtin_amdgcn_s_memtime();
            if constexpr (ABL & (2 | 8)) halo_wait_vmcnt<0>();
            else if constexpr (PWM) halo_wait_vmcnt<NAPW + 2 * GBP>();
            else if (KWc == 1) halo_wait_vmcnt<GBP>();
            else if (first || (PFD == 3 && prev_first)) halo_wait_vmcnt<(PFD - 1) * GBP + NAP>();
            else halo_wait_vmcnt<(PFD - 1) * GBP>();
            if constexpr (DBG) tq2 = (long long)__builtin_amdgcn_s_memtime();
            __builtin_amdgcn_s_barrier();
            if constexpr (DBG) {
                const long long tq3 = (long long)__builtin_amdgcn_s_memtime();
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                dbg_is += tq1 - tq0; dbg_vm += tq2 - tq1; dbg_bar += tq3 - tq2;
            }
            prev_first = first;
            NH_NEXT_TAP()
        }
        halo_wait_vmcnt<0>();                           // dummy DMAs past the end still target LDS
        __builtin_amdgcn_s_barrier();
        if constexpr (DBG) {                            // one record per wave
            if (a.dbg && lane == 0) {
                long long* d = a.dbg + ((size_t)blockIdx.x * (NCW + NPW) + wave) * 4;
                d[0] = (long long)__builtin_amdgcn_s_memtime() - dbg_t0; d[1] = dbg_is; d[2] = dbg_vm; d[3] = dbg_bar;
            }
        }
        return;                                         // the epilogue's barriers count live waves only
#undef NH_MAP_SEGMENT
#undef NH_GLDS
#undef NH_ISSUE_A
#undef NH_ISSUE_B
    }

    // =============================================================================================
    // Consumer waves.  Tile pixel r = wm*64 + t*32 + (lane&31) sits in halo row r + (KW-1)*i(r)
    // (+ kw per tap) while segment 0 runs, in row r for the transform segment; 16-byte piece p of a
    // row sits at slot p ^ ((row>>1)&7).
    const int wm = wave / WN, wn = wave % WN;
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
#define NH_READ_HALF(H, STG)                                                                       \
    {                                                                                              \
        const float* Sa_ = smem + bufC * A_BUF;                                                    \
        const float* Sb_ = smem + B_BASE + (STG) * B_STAGE + bcol;                                 \
        _Pragma("unroll") for (int t = 0; t < TM; ++t) {                                           \
            const int jr_ = PWM ? jb1[t] : (segC ? jb1[t] : jb0[t]) + kwC;                         \
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

    // One tap `it`: MFMAs of half 0 with the operand reads of half 1 of tap it issued BETWEEN them;
    // barrier (the producers have seen the operands of tap it+1 land, every consumer holds all of
    // tap it in registers -- which frees its weight stage and, after the last tap of a super-chunk, its
    // halo buffer); MFMAs of half 1 with the reads of half 0 of tap it+1 between them.
    // The interleave is pinned (sched_group_barrier): one ds_read behind each of the first ND MFMAs of the half,
    // the remaining MFMAs with nothing behind them, so that the last reads have landed when the wave reaches the
    // lgkmcnt(0) in front of the barrier (an even spread leaves the last read right there: +1 % wall).  With a read block
    // in front of an MFMA block -- the round-1 arrangement -- all eight waves hammer the LDS while the
    // matrix pipes idle and then the reverse; a wave's reads issued in the shadow of its own 32-cycle
    // MFMAs cost the pipe nothing (tools/ubench/fat_loop.hip: 2,436 -> 2,069 cycles per tap beside
    // 28 KB of LDS-DMA, MFMA floor 1,536).
    __builtin_amdgcn_s_barrier();                       // image 0 and tap 0 have landed
    long long dbg_bar = 0, dbg_t0 = 0;
    if constexpr (DBG) dbg_t0 = (long long)__builtin_amdgcn_s_memtime();
    constexpr int NM = KH_ * TM * TN * (PREC == 1 ? 3 : 4);         // MFMAs per half
    constexpr int ND = KH_ * (TM + TN) * (PREC == 1 ? 2 : 1);       // ds_read_b128 per half
    static_assert(NM >= ND, "interleave pattern");
#define NH_PIN_INTERLEAVE()                                                                        \
    if constexpr ((ABL & 256) != 0) pin_reads_between_mfmas<0, NM, ND>();                             \
    else if constexpr (!(ABL & (16 | 32 | 128))) pin_reads_front_loaded<0, NM, ND>();
    NH_READ_HALF(0, 0)
    int stC = 0;                                        // ring stage of tap `it`
    for (int it = 0; it < total; ++it) {
        __builtin_amdgcn_sched_barrier(0);
        if constexpr ((ABL & 128) != 0) {                // round-1 order: read block, then MFMA block
            NH_READ_HALF(1, stC)
            __builtin_amdgcn_sched_barrier(0);
            NH_MFMA_HALF(0)
        } else {
            if constexpr (!(ABL & 32)) NH_MFMA_HALF(0)
            if constexpr (!(ABL & 16)) NH_READ_HALF(1, stC)
            NH_PIN_INTERLEAVE()
        }
        if (++stC == BST) stC = 0;
        __builtin_amdgcn_sched_barrier(0);
        long long tq0 = 0;
        if constexpr (DBG) tq0 = (long long)__builtin_amdgcn_s_memtime();
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        if constexpr (DBG) {
            const long long tq1 = (long long)__builtin_amdgcn_s_memtime();
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            dbg_bar += tq1 - tq0;
        }
        __builtin_amdgcn_sched_barrier(0);
        NH_NEXT_TAP()
        if constexpr ((ABL & 128) != 0) {
            NH_READ_HALF(0, stC)
            __builtin_amdgcn_sched_barrier(0);
            NH_MFMA_HALF(1)
        } else {
            if constexpr (!(ABL & 32)) NH_MFMA_HALF(1)
            if constexpr (!(ABL & 16)) NH_READ_HALF(0, stC)   // (past the last tap: a harmless read)
            NH_PIN_INTERLEAVE()
        }
        if constexpr ((ABL & 32) != 0) {                 // keep the operand reads alive
            float k_ = 0.f;
            _Pragma("unroll") for (int s = 0; s < KS; ++s) {
                _Pragma("unroll") for (int t = 0; t < TM; ++t) k_ += fa_hi[s][t][0] + fa_lo[PREC == 1 ? s : 0][t][0];
                _Pragma("unroll") for (int j = 0; j < TN; ++j) k_ += fb_hi[s][j][0] + fb_lo[PREC == 1 ? s : 0][j][0];
            }
            acc[0][0][0] += k_;
        }
        __builtin_amdgcn_sched_barrier(0);
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();                       // the producers have drained every DMA
    if constexpr (DBG) {                                // one record per wave
        if (a.dbg && lane == 0) {
            long long* d = a.dbg + ((size_t)blockIdx.x * (NCW + NPW) + wave) * 4;
            d[0] = (long long)__builtin_amdgcn_s_memtime() - dbg_t0; d[1] = dbg_t0 - dbg_entry; d[2] = 0; d[3] = dbg_bar;
        }
    }
#undef NH_READ_HALF
#undef NH_MFMA_HALF
#undef NH_NEXT_TAP
#undef NH_PIN_INTERLEAVE

    long long dbg_epi = 0;
    long long es[3] = {0, 0, 0};
    if constexpr (DBG) dbg_epi = (long long)__builtin_amdgcn_s_memtime();
    static_assert(conv_epilogue_lds_bytes<HBM, BN>() <= (size_t)(NABUF * A_BUF + BST * B_STAGE) * sizeof(float), "epilogue LDS");
    conv_epilogue<TM, TN, PREC, NCW * 64, HBM, BN>(a, acc, smem, EpiTile{m0, 0, 0, 0, 0, 0}, wm * 64, wn * TN * 32, nt * BN, tid, lane,
                                                   DBG ? es : nullptr);
    if constexpr (DBG) {                                // [.., prologue cycles, epilogue cycles, ..]
        const long long t_issued = (long long)__builtin_amdgcn_s_memtime();
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        if (a.dbg && lane == 0) {
            const long long t_end = (long long)__builtin_amdgcn_s_memtime();
            a.dbg[((size_t)blockIdx.x * (NCW + NPW) + wave) * 4 + 2] = t_end - dbg_epi;
            // epilogue phases: LDS written | barrier passed | group 0 loaded | all stores issued | drained
            long long* e = a.dbg + (size_t)(4 << 20) + ((size_t)blockIdx.x * (NCW + NPW) + wave) * 8;
            e[0] = es[0] - dbg_epi; e[1] = es[1] - dbg_epi; e[2] = es[2] - dbg_epi; e[3] = t_issued - dbg_epi; e[4] = t_end - dbg_epi;
        }
    }
}

template <int BN, int PREC, int HBM_ = 256, int DBG = 0, int ABL = 0, int PWM = 0> static void launch_halo_t(const ConvArgs& a, hipStream_t s) {
    using SH = HaloShape<HBM_>;
    constexpr size_t lds = (size_t)((PWM ? 3 * SH::HBM : 2 * SH::HR) * 32 + SH::BST * 32 * BN) * sizeof(float);
    static_assert(lds <= 160 * 1024, "LDS of a gfx950 CU");
    static unsigned long long attr_devices = 0;
    set_max_dynamic_lds(reinterpret_cast<const void*>(&conv_igemm_halo<BN, PREC, HBM_, DBG, ABL, PWM>), lds, &attr_devices, "conv_igemm_halo");
    const int mtiles = (a.M + SH::HBM - 1) / SH::HBM;
    const int grid = mtiles * (a.N / BN);
    NHANS_LAUNCH("conv_igemm_halo", (conv_igemm_halo<BN, PREC, HBM_, DBG, ABL, PWM>), dim3(grid), dim3((NCW + NPW) * 64), lds, s, a);
}

// pixels per workgroup tile for this conv: 512 for the 64-channel layers, 256 otherwise
static int halo_tile_pixels(const ConvArgs& a) { return (a.N % 128 != 0 && a.halo64_tile512) ? 512 : 256; }

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
    const int hbm = halo_tile_pixels(a);
    const int nrows = (a.Wo - 1 + hbm - 1) / a.Wo + 1;
    return hbm + (g.KW - 1) * nrows <= (hbm == 512 ? HaloShape<512>::HR : HaloShape<256>::HR);
}

// Pointwise mode (template parameter PWM of the kernel): one segment, 128-channel tiles, any stride / padding / filter
// shape, at least four taps; layers marked for grouped summation stay on conv_igemm_dma.hip.
bool conv_igemm_halo_pw_eligible(const ConvArgs& a) {
    const ConvSeg& g = a.seg[0];
    if (a.nseg != 1 || a.N % 128 != 0 || a.kgroup != 0 || g.C % 32 != 0) return false;
    if (g.KH * g.KW * (g.C / 32) < 4 || a.M % (a.Ho * a.Wo) != 0) return false;
    const double elems = (double)(a.M / (a.Ho * a.Wo)) * g.H * g.W * g.C;      // 32-bit element offsets inside the kernel
    return elems + 65536.0 < 2147483648.0 && (double)a.M * a.N < 2147483648.0;
}

void launch_conv_igemm_halo_pw(const ConvArgs& a, hipStream_t s) {
    if (a.prec == 1) launch_halo_t<128, 1, 256, 0, 0, 1>(a, s);
    else launch_halo_t<128, 0, 256, 0, 0, 1>(a, s);
}

void launch_conv_igemm_halo(const ConvArgs& a0, hipStream_t s) {
    ConvArgs a = a0;
    a.fdWP = make_fastdiv((uint32_t)(a.Wo + a.seg[0].KW - 1));
    const bool wide = a.N % 128 == 0;
    if (a.ilv == 2 && a.prec == 1) {                    // A/B knob: operand reads spread evenly over the half (round 2's first pattern)
        if (!wide && halo_tile_pixels(a) == 512) launch_halo_t<64, 1, 512, 0, 256>(a, s);
        else if (wide) launch_halo_t<128, 1, 256, 0, 256>(a, s);
        else launch_halo_t<64, 1, 256, 0, 256>(a, s);
        return;
    }
    if (!a.ilv && a.prec == 1) {                        // A/B knob: round-1 instruction order
        if (!wide && halo_tile_pixels(a) == 512) launch_halo_t<64, 1, 512, 0, 128>(a, s);
        else if (wide) launch_halo_t<128, 1, 256, 0, 128>(a, s);
        else launch_halo_t<64, 1, 256, 0, 128>(a, s);
        return;
    }
    if (!wide && halo_tile_pixels(a) == 512) {          // 64-channel convs: 512-pixel tiles
#ifdef NHANS_DEV
        if (a.prec == 1 && a.dbg) { launch_halo_t<64, 1, 512, 1>(a, s); return; }
#endif
        if (a.prec == 1) launch_halo_t<64, 1, 512>(a, s); else launch_halo_t<64, 0, 512>(a, s);
        return;
    }
#ifdef NHANS_DEV
    const int abl = dev_ablate();
    if (a.prec == 1 && a.dbg) {     // cycle stamps, optionally of an ablated loop
#define NH_DBG_CASE(V) case V: if (wide) launch_halo_t<128, 1, 256, 1, V>(a, s); else launch_halo_t<64, 1, 256, 1, V>(a, s); break;
        switch (abl) {
            NH_DBG_CASE(1) NH_DBG_CASE(2) NH_DBG_CASE(4) NH_DBG_CASE(5) NH_DBG_CASE(8) NH_DBG_CASE(10) NH_DBG_CASE(16) NH_DBG_CASE(32) NH_DBG_CASE(48) NH_DBG_CASE(128)
            default: if (wide) launch_halo_t<128, 1, 256, 1>(a, s); else launch_halo_t<64, 1, 256, 1>(a, s);
        }
#undef NH_DBG_CASE
    } else if (a.prec == 1) {
#define NH_ABL_CASE(V) case V: if (wide) launch_halo_t<128, 1, 256, 0, V>(a, s); else launch_halo_t<64, 1, 256, 0, V>(a, s); break;
        switch (abl) {              // timing experiments only: results are wrong for abl != 0
            NH_ABL_CASE(1) NH_ABL_CASE(2) NH_ABL_CASE(4) NH_ABL_CASE(5) NH_ABL_CASE(8) NH_ABL_CASE(10)
            NH_ABL_CASE(16) NH_ABL_CASE(32) NH_ABL_CASE(48) NH_ABL_CASE(64)
            default: if (wide) launch_halo_t<128, 1>(a, s); else launch_halo_t<64, 1>(a, s);
        }
#undef NH_ABL_CASE
    } else {
        if (wide) launch_halo_t<128, 0>(a, s); else launch_halo_t<64, 0>(a, s);
    }
#else
    if (a.prec == 1) { if (wide) launch_halo_t<128, 1>(a, s); else launch_halo_t<64, 1>(a, s); }
    else { if (wide) launch_halo_t<128, 0>(a, s); else launch_halo_t<64, 0>(a, s); }
#endif
}

}  // namespace nhans
