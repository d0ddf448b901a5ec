// Implicit-GEMM convolution on FOUR-wave workgroups, two resident per CU ("quad"): the halo-reuse staging,
// data layouts, packed weights, MFMA arrangement, K order and epilogue of conv_igemm_halo.hip, but no
// producer/consumer split -- every wave multiplies its 64 x 64 tile AND issues a quarter of the LDS-DMA.
//
// Why (tools/ubench/residency.hip, tools/ubench/quad_loop.hip): with one 12-wave workgroup
// per CU the matrix pipes idle through every prologue and epilogue (15 % of a tile for the rb2 convs) and all
// eight consumer waves meet at one barrier per tap.  Two workgroups per CU overlap one's epilogue with the
// other's K loop and halve the barrier's span -- but the dispatcher only co-schedules workgroups whose waves
// spread evenly over the SIMDs (4 x 160 registers yes, 6 x 160 no), which rules out a 4 + 2 specialised
// shape.  Measured on the loop alone: 1,768 ticks per 256 x 128-equivalent tap beside 40 KB of DMA per CU
// against 2,069 for the 8 + 4 arrangement beside 28 KB.
//
//   * Tile 128 pixels x 128 channels, wave grid 2 x 2, wave tile 64 x 64 (TM = TN = 2).
//   * LDS 72 KB per workgroup: halo image double-buffered (2 x 160 rows x 128 B), weight ring of 2 stages
//     (2 x 16 KB).  The epilogue's transposed accumulator tile (128 x 132 floats) reuses it.
//   * One barrier per tap.  After barrier(it) every wave holds all of tap `it` in registers, so its weight
//     stage is free: each wave issues its quarter of the weights of tap it+2 into it (and, when tap it+1 opens
//     a super-chunk, its quarter of the NEXT super-chunk's halo image into the other image buffer, AFTER the
//     weights so that the counted wait at the next barrier can leave the image in flight for one more tap).
//   * Same K order and the same MFMA sequence per accumulator as every other conv kernel: bit-identical results.
//
// Eligibility as conv_igemm_halo.hip with a 128-pixel run (the halo image of a run must fit 160 rows), N % 128 == 0,
// split-f16 (f16x3) operands only.
#include "conv_epilogue.h"

namespace nhans {

namespace {
constexpr int QW = 4;            // waves per workgroup
constexpr int QBM = 128;         // output pixels per workgroup
constexpr int QBN = 128;         // output channels per workgroup
constexpr int QHR = 160;         // rows of one halo image
constexpr int QBST = 2;          // weight ring stages
constexpr int QNAP = QHR * 8 / (QW * 64);           // activation DMA instructions per thread per image (5)
constexpr int QGBP = 32 * QBN / 4 / (QW * 64);      // weight DMA instructions per thread per tap (4)

template <int N> __device__ __forceinline__ void quad_wait_vmcnt() {
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}
}  // namespace

template <int DBG = 0>
__global__ void __launch_bounds__(QW * 64, 2) conv_igemm_quad(const ConvArgs a) {
    constexpr int TM = 2, TN = 2, WN = 2;
    constexpr int A_BUF = QHR * 32;                    // floats
    constexpr int B_STAGE = 32 * QBN;                  // floats
    constexpr int B_BASE = 2 * A_BUF;
    extern __shared__ __attribute__((aligned(16))) float smem[];

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    long long dbg_entry = 0;
    if constexpr (DBG) dbg_entry = (long long)__builtin_amdgcn_s_memtime();

    // XCD-aware, bijective remap of the linear workgroup id
    const int ntn = a.N / QBN;
    int L;
    {
        const int nblk = gridDim.x, bid = blockIdx.x;
        const int q = nblk >> 3, r = nblk & 7, xcd = bid & 7, idx = bid >> 3;
        L = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
    }
    const int mt = L / ntn, nt = L - mt * ntn;
    const int m0 = mt * QBM;
    const int nt0 = nt * (QBN / 32);

    const int Wo = (int)a.fdWo.d;
    const int R0 = (int)fd_div((uint32_t)m0, a.fdWo);  // first output row (over all images) of the tile
    const int w0 = m0 - R0 * Wo;

    // taps are walked in the order (segment, 32-channel chunk, kh, kw) -- fold.py kmat()
    const int nseg = a.nseg;
    const int KW0 = a.seg[0].KW, KH0 = a.seg[0].KH, CC0 = a.seg[0].C >> 5;
    const int KH1 = nseg > 1 ? a.seg[1].KH : 0, CC1 = nseg > 1 ? (a.seg[1].C >> 5) : 0;
    const int nsup0 = KH0 * CC0;
    const int nsup = nsup0 + KH1 * CC1;
    const int ntap0 = nsup0 * KW0;
    const int total = ntap0 + KH1 * CC1;

    // cursor of the current tap: segment, kw within the super-chunk, halo buffer, super-chunk count
    int segC = 0, kwC = 0, bufC = 0, supC = 0;
#define NQ_NEXT_TAP()                                                                              \
    if (++kwC >= (segC ? 1 : KW0)) {                                                               \
        kwC = 0;                                                                                   \
        bufC ^= 1;                                                                                 \
        if (++supC == nsup0) segC = 1;                                                             \
    }

    // ---- DMA side (every wave): see conv_igemm_halo.hip for the row mapping and the source-side XOR swizzle
    const int slot = lane & 7;
    const int nrows_all = a.M / Wo;                    // B * Ho
    const size_t bstride = (size_t)(a.N / 32) * 1024;
    int poff[QNAP], hov[QNAP];
    int sH = 0, sW = 0, sC = 0;
    const float* ssrc = nullptr;
#define NQ_MAP_SEGMENT(S)                                                                          \
    {                                                                                              \
        const ConvSeg& g = a.seg[S];                                                               \
        sH = g.H; sW = g.W; sC = g.C; ssrc = g.src;                                                \
        _Pragma("unroll") for (int d = 0; d < QNAP; ++d) {                                         \
            const int j = d * (QW * 8) + wave * 8 + (lane >> 3);                                   \
            const int sp = (slot ^ ((j >> 1) & 7)) * 4;                                            \
            int Rg, wi;                                                                            \
            bool ok;                                                                               \
            if (g.KW > 1) {                                                                        \
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
                ok = j < QBM && m < a.M;                                                           \
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
#define NQ_GLDS(SRC, DST)                                                                          \
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(SRC),        \
                                     (__attribute__((address_space(3))) void*)(DST), 16, 0, 0);

    int segA = 0, khA = 0, ccA = 0, supA = 0;          // activation cursor: next super-chunk to stage
    const float* const zp = a.zero + (slot ^ ((lane >> 4) & 7)) * 4;
#define NQ_ISSUE_A(BUF)                                                                            \
    {                                                                                              \
        const int khoff_ = khA * sW * sC + ccA * 32;                                               \
        const bool live_ = supA < nsup;                                                            \
        float* sa_ = smem + (BUF) * A_BUF + wave * 8 * 32;                                         \
        _Pragma("unroll") for (int d = 0; d < QNAP; ++d) {                                         \
            const float* p_ = (live_ && (unsigned)(hov[d] + khA) < (unsigned)sH) ? ssrc + (poff[d] + khoff_) : zp; \
            NQ_GLDS(p_, sa_ + d * (QW * 8) * 32)                                                   \
        }                                                                                          \
        ++supA;                                                                                    \
        if (++khA >= (segA ? KH1 : KH0)) {                                                         \
            khA = 0;                                                                               \
            if (++ccA >= (segA ? CC1 : CC0)) {                                                     \
                ccA = 0;                                                                           \
                if (segA == 0 && nseg > 1) {                                                       \
                    segA = 1;                                                                      \
                    NQ_MAP_SEGMENT(1)                                                              \
                }                                                                                  \
            }                                                                                      \
        }                                                                                          \
    }
    // weight cursor: the next tap is the next `bstride` floats, with one jump to the transform's array and a
    // stall on the last chunk for the dummy loads past the end (unconditional issue keeps the waits static)
    int tapB = 0;
    const float* bp_ = a.seg[0].wpk + (size_t)nt0 * 1024 - bstride;
    const float* const wpk1 = (nseg > 1 ? a.seg[1].wpk : a.seg[0].wpk) + (size_t)nt0 * 1024;
#define NQ_ISSUE_B(ST)                                                                             \
    {                                                                                              \
        const float* nx_ = tapB == ntap0 ? wpk1 : bp_ + bstride;                                   \
        bp_ = tapB < total ? nx_ : bp_;                                                            \
        ++tapB;                                                                                    \
        float* sb_ = smem + B_BASE + (ST) * B_STAGE;                                               \
        _Pragma("unroll") for (int j = 0; j < QGBP; ++j)                                           \
            NQ_GLDS(bp_ + (j * (QW * 64) + tid) * 4, sb_ + (j * (QW * 64) + wave * 64) * 4)        \
    }

    // ---- MFMA side: tile pixel r = wm*64 + t*32 + (lane&31) sits in halo row r + (KW-1)*i(r) (+ kw per tap)
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
    constexpr int KS = 2, KH_ = 1;                     // k-steps per chunk (16 k each); one per half
    f32x4 fa_hi[KS][TM], fa_lo[KS][TM], fb_hi[KS][TN], fb_lo[KS][TN];
#define NQ_READ_HALF(H, STG)                                                                       \
    {                                                                                              \
        const float* Sa_ = smem + bufC * A_BUF;                                                    \
        const float* Sb_ = smem + B_BASE + (STG) * B_STAGE + bcol;                                 \
        _Pragma("unroll") for (int t = 0; t < TM; ++t) {                                           \
            const int jr_ = (segC ? jb1[t] : jb0[t]) + kwC;                                        \
            const float* ar_ = Sa_ + jr_ * 32;                                                     \
            const int rs_ = (jr_ >> 1) & 7;                                                        \
            fa_hi[H][t] = *reinterpret_cast<const f32x4*>(ar_ + (((2 * (H) + g8) ^ rs_) * 4));     \
            fa_lo[H][t] = *reinterpret_cast<const f32x4*>(ar_ + (((2 * (H) + g8 + 4) ^ rs_) * 4)); \
        }                                                                                          \
        _Pragma("unroll") for (int j = 0; j < TN; ++j) {                                           \
            fb_hi[H][j] = *reinterpret_cast<const f32x4*>(Sb_ + j * 1024 + (H) * 512);             \
            fb_lo[H][j] = *reinterpret_cast<const f32x4*>(Sb_ + j * 1024 + (H) * 512 + 256);       \
        }                                                                                          \
    }
    // (same product order per accumulator as conv_igemm_halo.hip: lo*hi, hi*lo, hi*hi)
#define NQ_MFMA_HALF(H)                                                                            \
    {                                                                                              \
        _Pragma("unroll") for (int p = 0; p < 3; ++p)                                              \
            _Pragma("unroll") for (int t = 0; t < TM; ++t)                                         \
                _Pragma("unroll") for (int j = 0; j < TN; ++j) {                                   \
                    const f16x8 a_ = __builtin_bit_cast(f16x8, p == 0 ? fa_lo[H][t] : fa_hi[H][t]); \
                    const f16x8 b_ = __builtin_bit_cast(f16x8, p == 1 ? fb_lo[H][j] : fb_hi[H][j]); \
                    acc[t][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(b_, a_, acc[t][j], 0, 0, 0); \
                }                                                                                  \
    }
    f32x16 acc[TM][TN];
#pragma unroll
    for (int t = 0; t < TM; ++t)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[t][j][r] = 0.f;
    constexpr int NM = KH_ * TM * TN * 3;              // MFMAs per half (12)
    constexpr int ND = KH_ * (TM + TN) * 2;            // ds_read_b128 per half (8)

    // ---- prologue: image 0, taps 0 and 1
    NQ_MAP_SEGMENT(0)
    NQ_ISSUE_A(0)
    NQ_ISSUE_B(0)
    quad_wait_vmcnt<0>();
    __builtin_amdgcn_s_barrier();
    long long dbg_t0 = 0, dbg_bar = 0;
    if constexpr (DBG) dbg_t0 = (long long)__builtin_amdgcn_s_memtime();
    // "DMA slot 0" (what the loop does after barrier(it-1) for it = 0): weights of tap 1, image of super-chunk 1
    NQ_ISSUE_B(1)
    NQ_ISSUE_A(1)
    bool img_in_flight = true;                          // super-chunk 1 opens KW0 >= 3 taps from now: may cross barrier(0)
    NQ_READ_HALF(0, 0)
    int stC = 0;                                        // ring stage of tap `it`
    for (int it = 0; it < total; ++it) {
        __builtin_amdgcn_sched_barrier(0);
        NQ_MFMA_HALF(0)
        NQ_READ_HALF(1, stC)
        pin_reads_between_mfmas<0, NM, ND>();
        __builtin_amdgcn_sched_barrier(0);
        long long tq0 = 0;
        if constexpr (DBG) tq0 = (long long)__builtin_amdgcn_s_memtime();
        // everything tap it+1 reads has landed: the weights issued in the last DMA slot always; an image issued
        // in that slot is needed only if tap it+1 is a one-tap super-chunk (the 1x1 transform segment)
        if (img_in_flight) quad_wait_vmcnt<QNAP>(); else quad_wait_vmcnt<0>();
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        if constexpr (DBG) dbg_bar += (long long)__builtin_amdgcn_s_memtime() - tq0;
        __builtin_amdgcn_sched_barrier(0);
        NQ_NEXT_TAP()                                   // cursor -> tap it+1
        // DMA slot it+1: weights of tap it+2 into the stage tap `it` has just left; if tap it+1 opens a
        // super-chunk, the image of the one after it into the buffer the tap before last read
        NQ_ISSUE_B(stC)
        const bool first = kwC == 0;
        if (first) NQ_ISSUE_A(bufC ^ 1)
        img_in_flight = first && segC == 0;
        stC ^= 1;
        __builtin_amdgcn_sched_barrier(0);
        NQ_MFMA_HALF(1)
        NQ_READ_HALF(0, stC)                            // (past the last tap: a harmless read)
        pin_reads_between_mfmas<0, NM, ND>();
        __builtin_amdgcn_sched_barrier(0);
    }
    quad_wait_vmcnt<0>();                               // dummy DMAs past the end still target LDS
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
#undef NQ_READ_HALF
#undef NQ_MFMA_HALF
#undef NQ_NEXT_TAP
#undef NQ_MAP_SEGMENT
#undef NQ_GLDS
#undef NQ_ISSUE_A
#undef NQ_ISSUE_B

    long long dbg_epi = 0;
    if constexpr (DBG) dbg_epi = (long long)__builtin_amdgcn_s_memtime();
    static_assert(conv_epilogue_lds_bytes<QBM, QBN>() <= (size_t)(2 * A_BUF + QBST * B_STAGE) * sizeof(float), "epilogue LDS");
    conv_epilogue<TM, TN, 1, QW * 64, QBM, QBN>(a, acc, smem, EpiTile{m0, 0, 0, 0, 0, 0}, wm * 64, wn * TN * 32, nt * QBN, tid, lane);
    if constexpr (DBG) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        if (a.dbg && lane == 0) {
            // per wave: K-loop ticks, prologue, epilogue, barrier waits | entry, exit, (XCC_ID, HW_ID)
            const long long t_end = (long long)__builtin_amdgcn_s_memtime();
            long long* d = a.dbg + ((size_t)blockIdx.x * QW + wave) * 8;
            d[0] = dbg_epi - dbg_t0; d[1] = dbg_t0 - dbg_entry; d[2] = t_end - dbg_epi; d[3] = dbg_bar;
            d[4] = dbg_entry; d[5] = t_end;
            d[6] = ((long long)__builtin_amdgcn_s_getreg((31 << 11) | 20) << 32) | (unsigned)__builtin_amdgcn_s_getreg((31 << 11) | 4);
        }
    }
}

bool conv_igemm_quad_eligible(const ConvArgs& a) {
    const ConvSeg& g = a.seg[0];
    if (a.prec != 1 || a.N % QBN != 0) return false;
    if (g.KW < 3 || g.sw != 1 || a.Wo != g.W || g.pl < 0 || g.pl >= g.KW) return false;
    if (a.nseg > 1 && a.seg[1].KW != 1) return false;
    if (a.nseg > 2 || a.M % a.Wo != 0) return false;
    for (int i = 0; i < a.nseg; ++i) {                  // 32-bit element offsets inside the kernel
        const ConvSeg& q = a.seg[i];
        const double elems = (double)(a.M / (a.Ho * a.Wo)) * q.H * q.W * q.C;
        if (elems + 65536.0 >= 2147483648.0) return false;
    }
    const int nrows = (a.Wo - 1 + QBM - 1) / a.Wo + 1;  // image rows a 128-pixel run can touch
    return QBM + (g.KW - 1) * nrows <= QHR;
}

void launch_conv_igemm_quad(const ConvArgs& a0, hipStream_t s) {
    ConvArgs a = a0;
    a.fdWP = make_fastdiv((uint32_t)(a.Wo + a.seg[0].KW - 1));
    constexpr size_t lds = (size_t)(2 * QHR * 32 + QBST * 32 * QBN) * sizeof(float);
    static_assert(lds <= 80 * 1024, "two workgroups per CU");
    const int grid = ((a.M + QBM - 1) / QBM) * (a.N / QBN);
#ifdef NHANS_DEV
    if (a.dbg) {
        static unsigned long long attr_devices_d = 0;
        set_max_dynamic_lds(reinterpret_cast<const void*>(&conv_igemm_quad<1>), lds, &attr_devices_d, "conv_igemm_quad");
        NHANS_LAUNCH("conv_igemm_quad", (conv_igemm_quad<1>), dim3(grid), dim3(QW * 64), lds, s, a);
        return;
    }
#endif
    static unsigned long long attr_devices = 0;
    set_max_dynamic_lds(reinterpret_cast<const void*>(&conv_igemm_quad<0>), lds, &attr_devices, "conv_igemm_quad");
    NHANS_LAUNCH("conv_igemm_quad", (conv_igemm_quad<0>), dim3(grid), dim3(QW * 64), lds, s, a);
}

}  // namespace nhans
