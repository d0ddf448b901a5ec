// Implicit-GEMM NHWC convolution on the gfx950 matrix cores, with the N-HANS block epilogue
// (conditioning bias + position tables + residual + ReLU; BatchNorm is folded into the weights
// and tables on the host) fused in.
//
// Replaces tf.nn.conv2d + bias + broadcast adds + tf.nn.batch_normalization + tf.nn.relu of
// SN/blocks.py:38-48,104-108 and SN/main.py:102-124,161-187,232-238.
//
// Two arithmetic modes share the tiling, staging and epilogue:
//   PREC 0  exact f32: v_mfma_f32_32x32x2_f32 (157 TF pipe).  Activations are f32 NHWC.
//   PREC 1  split f16 x3 on the 2.5 PF f16 pipe: every value is carried as hi + lo with
//           hi = f16(x), lo = f16(x - hi); a*b ~ ah*bh + ah*bl + al*bh accumulated in f32 by
//           v_mfma_f32_32x32x16_f16 (FP32-class result: the dropped al*bl term is 2^-22 relative).
//           Activations live in HBM already split ("split NHWC": per pixel and per group of 32
//           channels one 128-byte line = 32 hi halfs then 32 lo halfs), written that way by the
//           producing epilogue, so the K loop only copies bytes.  Weight columns are pre-scaled by a
//           power of two into [32,64) so their lo parts stay normal f16 numbers; the epilogue undoes it.
//   A 32-channel chunk of one pixel occupies the same 128 bytes in both layouts, so the staging
//   code is identical.
//
// Tiling (one workgroup = 4 wavefronts of 64 lanes, 128 output pixels x BN output channels):
//   * K is walked in chunks of 32 input channels of one filter tap.  The A chunk (128 pixels x 128
//     bytes, gathered with TF-SAME asymmetric zero padding) and the B chunk (pre-packed on the host
//     in MFMA fragment order, 4 KB per 32 output channels) are double-buffered in LDS.
//   * Padding costs nothing in the loop: once per filter tap every staging thread resolves each of
//     its 4 pixel rows to a pointer -- into the tensor, or into a page of zeros when the tap falls
//     outside the image -- and per chunk only adds the channel offset.
//   * 4-stage software pipeline, unrolled by two with two A register sets: at the top of iteration
//     `it` the B chunk it+1 is sent global -> LDS directly (LDS-DMA, lane-linear image) and the A
//     chunk it+2 is loaded into registers; chunk it is multiplied from LDS; at the bottom the A
//     chunk it+1 goes registers -> LDS.  One barrier per chunk; every memory operation has at least
//     one whole chunk of MFMAs to hide behind.
//   * f32: lane l supplies A[row = l&31][k = l>>5], B[k = l>>5][col = l&31]; K inside a chunk is
//     permuted (MFMA (q,e) takes k = 8q + 4(l>>5) + e) so operands are read as 16-byte vectors.
//     f16: lane l supplies 8 consecutive k of A row l&31 / B column l&31, k-group l>>5.
//   * LDS A rows are padded to 144 bytes: the 16 rows of a ds_read_b128 lane group start on 16
//     distinct 16-byte bank slots (36*i mod 64 covers all multiples of 4); B is lane-linear.
//   * Workgroup ids are remapped so each XCD (private L2) owns a contiguous range of pixel tiles:
//     vertically adjacent tiles re-read the same input rows for neighbouring taps.
//   * Epilogue: per-row (clip, h, w) is staged once in LDS; all table / residual loads of a row
//     group are issued branch-free before the first use; stores follow.
#include "conv_epilogue.h"
#include <cstdlib>

namespace nhans {

constexpr int BM = 128, BK = 32, LDA = 36;

template <int BN, int WM, int WN, int PREC, int ABL = 0>   // ABL: timing ablations (tools/ablate.py)
__global__ void __launch_bounds__(256) conv_igemm(const ConvArgs a) {
    constexpr int TM = BM / WM / 32, TN = BN / WN / 32;
    constexpr int A_BUF = BM * LDA, B_BUF = BK * BN;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* As = smem;
    float* Bs = smem + 2 * A_BUF;

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave / WN, wn = wave % WN;

    // XCD-aware, bijective remap of the linear workgroup id
    const int ntn = a.N / BN;
    int L;
    {
        const int nblk = gridDim.x, bid = blockIdx.x;
        const int q = nblk >> 3, r = nblk & 7, xcd = bid & 7, idx = bid >> 3;
        L = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
    }
    const int mt = L / ntn, nt = L - mt * ntn;
    const int m0 = mt * BM;
    const int nt0 = nt * (BN / 32);          // first 32-wide n-tile of this block

    // ---- A staging assignment: thread owns rows (tid>>3) + 32*i, 16 bytes at column (tid&7)*4
    const int col4 = tid & 7;
    int rb[4], rho[4], rwo[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int m = m0 + (tid >> 3) + 32 * i;
        if (m < a.M) {
            const uint32_t b = fd_div((uint32_t)m, a.fdHoWo);
            const uint32_t rem = (uint32_t)m - b * a.fdHoWo.d;
            const uint32_t ho = fd_div(rem, a.fdWo);
            rb[i] = (int)b; rho[i] = (int)ho; rwo[i] = (int)(rem - ho * a.fdWo.d);
        } else {
            rb[i] = -1; rho[i] = 0; rwo[i] = 0;
        }
    }

    // ---- A cursor (runs two chunks ahead of the MFMAs): segment, tap, channel offset, and per row
    // the pointer of the current tap (tensor pixel or zero page)
    int64_t roff0, roff1, roff2, roff3;
    int hi0[4], wi0[4];
    const float *pa0, *pa1, *pa2, *pa3;
    int seg = 0, kh = 0, kw = 0, c0 = 0;
    int sH, sW, sC, sKW, sKH;
    const float* ssrc;
    const float* zpage = a.zero + col4 * 4;

#define NH_ROW(I, ROFF)                                                                            \
    if (rb[I] >= 0) {                                                                              \
        hi0[I] = rho[I] * g.sh - g.pt;                                                             \
        wi0[I] = rwo[I] * g.sw - g.pl;                                                             \
        ROFF = (((int64_t)rb[I] * g.H + hi0[I]) * g.W + wi0[I]) * (int64_t)g.C + col4 * 4;          \
    } else {                                                                                       \
        hi0[I] = -(1 << 28); wi0[I] = 0; ROFF = 0;                                                 \
    }
#define NH_TAP_ROW(I, ROFF, PA)                                                                    \
    PA = ((unsigned)(hi0[I] + kh) < (unsigned)sH && (unsigned)(wi0[I] + kw) < (unsigned)sW)        \
             ? ssrc + (ROFF + tapoff) : zpage;
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
                c0 += BK;                                                                          \
            }                                                                                      \
        }                                                                                          \
        if (c0 >= sC) {                                                                           \
            ++seg;                                                                                 \
            if (seg < a.nseg) NH_ENTER_SEGMENT(seg)                                                \
            else c0 = 0;              /* past the end: the cursor stays on valid memory */         \
        } else NH_TAP()                                                                            \
    }
#define NH_LOAD_A(S)                                                                               \
    if constexpr (ABL & 2) {                                                                       \
        ra0##S = ra1##S = ra2##S = ra3##S = f32x4{0.f, 0.f, 0.f, 0.f};                              \
    } else {                                                                                       \
        ra0##S = *reinterpret_cast<const f32x4*>(pa0 + c0);                                        \
        ra1##S = *reinterpret_cast<const f32x4*>(pa1 + c0);                                        \
        ra2##S = *reinterpret_cast<const f32x4*>(pa2 + c0);                                        \
        ra3##S = *reinterpret_cast<const f32x4*>(pa3 + c0);                                        \
    }
#define NH_STORE_A(BUF, S)                                                                         \
    if constexpr (!(ABL & 8)) {                                                                    \
        float* Ar_ = As + (BUF) * A_BUF + (tid >> 3) * LDA + col4 * 4;                             \
        *reinterpret_cast<f32x4*>(Ar_) = ra0##S;                                                   \
        *reinterpret_cast<f32x4*>(Ar_ + 32 * LDA) = ra1##S;                                        \
        *reinterpret_cast<f32x4*>(Ar_ + 64 * LDA) = ra2##S;                                        \
        *reinterpret_cast<f32x4*>(Ar_ + 96 * LDA) = ra3##S;                                        \
    }

    // ---- B cursor (runs one chunk ahead): chunk `bchunk` of the concatenated K goes global -> LDS
    // by LDS-DMA; the LDS image is lane-linear, each wave-instruction moves one 1 KB piece.
    const int n0chunks = a.seg[0].nchunks;
    const size_t bstride = (size_t)(a.N / 32) * 1024;
    int bchunk = 0;
#define NH_LOAD_B(BUF)                                                                             \
    if constexpr (ABL & 4) { ++bchunk; } else {                                                    \
        const float* bp = (bchunk < n0chunks ? a.seg[0].wpk + (size_t)bchunk * bstride             \
                                             : a.seg[1].wpk + (size_t)(bchunk - n0chunks) * bstride) + \
                          (size_t)nt0 * 1024;                                                      \
        _Pragma("unroll") for (int j = 0; j < BN / 32; ++j)                                        \
            __builtin_amdgcn_global_load_lds(                                                      \
                (const __attribute__((address_space(1))) void*)(bp + (j * 256 + tid) * 4),         \
                (__attribute__((address_space(3))) void*)(Bs + (BUF) * B_BUF + (j * 256 + wave * 64) * 4), \
                16, 0, 0);                                                                         \
        ++bchunk;                                                                                  \
    }

    // f32: A piece (q) of row i at float offset q*8 + (lane>>5)*4; B piece at q*256 + lane*4.
#define NH_COMPUTE_F32(BUF)                                                                        \
    {                                                                                              \
        const float* Ab_ = As + (BUF) * A_BUF + arow;                                              \
        const float* Bb_ = Bs + (BUF) * B_BUF + bcol;                                              \
        _Pragma("unroll") for (int q = 0; q < 4; ++q) {                                            \
            f32x4 av[TM], bv[TN];                                                                  \
            _Pragma("unroll") for (int i = 0; i < TM; ++i)                                         \
                av[i] = *reinterpret_cast<const f32x4*>(Ab_ + i * 32 * LDA + q * 8);               \
            _Pragma("unroll") for (int j = 0; j < TN; ++j)                                         \
                bv[j] = *reinterpret_cast<const f32x4*>(Bb_ + j * 1024 + q * 256);                 \
            _Pragma("unroll") for (int i = 0; i < TM; ++i)                                         \
                _Pragma("unroll") for (int j = 0; j < TN; ++j) {                                   \
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(bv[j].x, av[i].x, acc[i][j], 0, 0, 0); \
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(bv[j].y, av[i].y, acc[i][j], 0, 0, 0); \
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(bv[j].z, av[i].z, acc[i][j], 0, 0, 0); \
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(bv[j].w, av[i].w, acc[i][j], 0, 0, 0); \
                }                                                                                  \
        }                                                                                          \
    }

    // f16 x3: k-step s (16 k): A hi piece of row i at float offset (2s + (lane>>5))*4, lo piece 16
    // floats further; B pieces at ((s*2 + h)*64 + lane)*4 with h = 0 (hi) / 1 (lo).
#define NH_COMPUTE_H3(BUF)                                                                         \
    {                                                                                              \
        const float* Ab_ = As + (BUF) * A_BUF + arow;                                              \
        const float* Bb_ = Bs + (BUF) * B_BUF + bcol;                                              \
        _Pragma("unroll") for (int s = 0; s < 2; ++s) {                                            \
            f16x8 ah[TM], al[TM], bh[TN], bl[TN];                                                  \
            _Pragma("unroll") for (int i = 0; i < TM; ++i) {                                       \
                if constexpr (ABL & 32) {                                                          \
                    const f32x4 cz = {1.0f + s, 2.0f, 3.0f, 4.0f + i};                             \
                    ah[i] = __builtin_bit_cast(f16x8, cz); al[i] = ah[i];                          \
                } else {                                                                           \
                ah[i] = __builtin_bit_cast(f16x8, *reinterpret_cast<const f32x4*>(Ab_ + i * 32 * LDA + s * 8));      \
                al[i] = __builtin_bit_cast(f16x8, *reinterpret_cast<const f32x4*>(Ab_ + i * 32 * LDA + 16 + s * 8)); \
                }                                                                                  \
            }                                                                                      \
            _Pragma("unroll") for (int j = 0; j < TN; ++j) {                                       \
                if constexpr (ABL & 32) {                                                          \
                    const f32x4 cz = {1.5f + s, 2.5f, 3.5f, 4.5f + j};                             \
                    bh[j] = __builtin_bit_cast(f16x8, cz); bl[j] = bh[j];                          \
                } else {                                                                           \
                bh[j] = __builtin_bit_cast(f16x8, *reinterpret_cast<const f32x4*>(Bb_ + j * 1024 + s * 512));        \
                bl[j] = __builtin_bit_cast(f16x8, *reinterpret_cast<const f32x4*>(Bb_ + j * 1024 + s * 512 + 256)); \
                }                                                                                  \
            }                                                                                      \
            _Pragma("unroll") for (int i = 0; i < TM; ++i)                                         \
                _Pragma("unroll") for (int j = 0; j < TN; ++j) {                                   \
                    if constexpr (ABL & 1) {                                                       \
                        asm volatile("" ::"v"(al[i]), "v"(ah[i]), "v"(bh[j]), "v"(bl[j]));         \
                    } else {                                                                       \
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(bh[j], al[i], acc[i][j], 0, 0, 0); \
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(bl[j], ah[i], acc[i][j], 0, 0, 0); \
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(bh[j], ah[i], acc[i][j], 0, 0, 0); \
                    }                                                                              \
                }                                                                                  \
        }                                                                                          \
    }

#define NH_COMPUTE(BUF)                                                                            \
    if constexpr (PREC == 0) NH_COMPUTE_F32(BUF) else NH_COMPUTE_H3(BUF)

    f32x4 ra0_x, ra1_x, ra2_x, ra3_x;
    f32x4 ra0_y, ra1_y, ra2_y, ra3_y;

    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    int total = 0;
    for (int s = 0; s < a.nseg; ++s) total += a.seg[s].nchunks;

    const int arow = (wm * TM * 32 + (lane & 31)) * LDA + (lane >> 5) * 4;
    const int bcol = (wn * TN) * 1024 + lane * 4;

    // LDS-DMA writes are ordered for other waves only by the issuing wave's vmcnt followed by a
    // barrier; hipcc emits that wait itself, the explicit one keeps the contract visible.
#define NH_SYNC()                                                                                  \
    {                                                                                              \
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                                           \
        if constexpr (!(ABL & 16)) __syncthreads();                                                \
    }

    // prologue: chunk 0 into LDS buffer 0, A chunk 1 into register set y
    NH_ENTER_SEGMENT(0)
    NH_LOAD_B(0)
    NH_LOAD_A(_x)
    NH_STORE_A(0, _x)
    if (total > 1) {
        NH_ADVANCE_A()
        NH_LOAD_A(_y)
    }
    NH_SYNC()

    int it = 0;
    for (; it + 3 < total; it += 2) {
        NH_LOAD_B(1)                                // B chunk it+1 -> buffer 1
        NH_ADVANCE_A()
        NH_LOAD_A(_x)                               // A chunk it+2 -> set x
        NH_COMPUTE(0)                               // chunk it (buffer 0)
        NH_STORE_A(1, _y)                           // A chunk it+1 -> buffer 1
        NH_SYNC()
        NH_LOAD_B(0)                                // B chunk it+2 -> buffer 0
        NH_ADVANCE_A()
        NH_LOAD_A(_y)                               // A chunk it+3 -> set y
        NH_COMPUTE(1)                               // chunk it+1
        NH_STORE_A(0, _x)                           // A chunk it+2 -> buffer 0
        NH_SYNC()
    }
    // tail: `it` is even, 1..3 chunks left; chunk it is in buffer 0, A chunk it+1 (if any) in set y
    if (it + 1 < total) { NH_LOAD_B(1) }
    if (it + 2 < total) {
        NH_ADVANCE_A()
        NH_LOAD_A(_x)                               // A chunk it+2
    }
    NH_COMPUTE(0)
    if (it + 1 < total) {
        NH_STORE_A(1, _y)
        NH_SYNC()
        if (it + 2 < total) { NH_LOAD_B(0) }
        NH_COMPUTE(1)
        if (it + 2 < total) {
            NH_STORE_A(0, _x)
            NH_SYNC()
            NH_COMPUTE(0)
        }
    }
    NH_SYNC()

#undef NH_SYNC
#undef NH_ROW
#undef NH_TAP_ROW
#undef NH_TAP
#undef NH_ENTER_SEGMENT
#undef NH_ADVANCE_A
#undef NH_LOAD_A
#undef NH_STORE_A
#undef NH_LOAD_B
#undef NH_COMPUTE_F32
#undef NH_COMPUTE_H3
#undef NH_COMPUTE

    // ---- epilogue (conv_epilogue.h): transposed through LDS, coalesced on the global side
    static_assert(conv_epilogue_lds_bytes<BM, BN>() <= (2 * BM * LDA + 2 * BK * BN) * sizeof(float), "epilogue LDS");
    conv_epilogue<TM, TN, PREC, 256, BM, BN>(a, acc, smem, EpiTile{m0, 0, 0, 0, 0, 0}, wm * TM * 32, wn * TN * 32, nt * BN, tid, lane);
}

template <int BN, int WM, int WN, int PREC, int ABL = 0>
static void launch_t(const ConvArgs& a, hipStream_t s) {
    constexpr size_t lds = (2 * BM * LDA + 2 * BK * BN) * sizeof(float);
    static unsigned long long attr_devices = 0;
    set_max_dynamic_lds(reinterpret_cast<const void*>(&conv_igemm<BN, WM, WN, PREC, ABL>), lds, &attr_devices, "conv_igemm");
    const int mtiles = (a.M + BM - 1) / BM;
    const int grid = mtiles * (a.N / BN);
    NHANS_LAUNCH("conv_igemm", (conv_igemm<BN, WM, WN, PREC, ABL>), dim3(grid), dim3(256), lds, s, a);
}

double launch_conv_igemm(const ConvArgs& a0, hipStream_t s, const char** kernel, double* mfma_flops) {
    ConvArgs a = a0;
    a.halo64_tile512 = a.variant == 2;
    double k = 0;
    for (int i = 0; i < a.nseg; ++i) k += (double)a.seg[i].nchunks * BK;
    const char* name = "conv_igemm";
    const bool wide = a.N % 128 == 0;
    // an f32-stored input in the split mode is something only conv_wino.hip reads: the caller (nhans_api.hip:
    // stored_f32) decides both from the same predicate; a disagreement must not run a kernel on the wrong layout
    if (a.prec == 1 && a.in_f32 && !(a.variant >= 2 && a.kgroup >= 0 && conv_wino_eligible(a))) {
        note_refusal("conv (f32-stored input without a Winograd form)");
        if (kernel) *kernel = "refused";
        if (mfma_flops) *mfma_flops = 0;
        return 0;
    }
    if (a.variant >= 1) {
        // (layers marked for grouped summation / split-K always take the LDS-DMA kernel, whatever the
        // launch size: the choice must not depend on the batch)
        const bool halo_ok = a.variant >= 2 && a.kgroup >= 0;
        bool wino = false;
        if (halo_ok && conv_wino_eligible(a)) {
            // stride-1 4 x 4 convs of the stack: 1-D Winograd along W, 2.5 x fewer MFMAs (conv_wino.hip)
            launch_conv_wino(a, s);
            name = "conv_wino<128>";
            wino = true;
        } else if (halo_ok && !wide && conv_igemm_halo_eligible(a)) {
            // the 64-channel stride-1 convs: 512-pixel tiles
            launch_conv_igemm_halo(a, s);
            name = "conv_igemm_halo<64,512>";
        } else if (halo_ok && (a.halo64_tile512 = 0, conv_igemm_halo_eligible(a))) {
            launch_conv_igemm_halo(a, s);
            name = wide ? "conv_igemm_halo<128>" : "conv_igemm_halo<64>";
        } else if (halo_ok && wide && conv_igemm_halo_pw_eligible(a)) {
            // strided / VALID convs: the producer-consumer pipeline with one staged image per tap
            launch_conv_igemm_halo_pw(a, s);
            name = "conv_igemm_halo_pw<128>";
        } else {
            launch_conv_igemm_dma(a, s);
            name = a.kgroup < 0 ? (wide ? "conv_igemm_dma<128,grouped>" : "conv_igemm_dma<64,grouped>")
                                : (wide ? "conv_igemm_dma<128>" : "conv_igemm_dma<64>");
        }
        if (kernel) *kernel = name;
        if (mfma_flops) *mfma_flops = wino ? conv_wino_mfma_flops(a) : (a.prec == 1 ? 3.0 : 1.0) * 2.0 * (double)a.M * k * (double)a.N;
        return 2.0 * (double)a.M * k * (double)a.Nreal;
    }
    if (kernel) *kernel = wide ? "conv_igemm<128>" : "conv_igemm<64>";
    if (a.prec == 1) {
        if (a.N % 128 == 0) {
#ifdef NHANS_DEV
            switch (dev_ablate()) {      // timing experiments only: results are wrong for a non-zero value
                case 1: launch_t<128, 2, 2, 1, 1>(a, s); break;
                case 14: launch_t<128, 2, 2, 1, 14>(a, s); break;
                case 16: launch_t<128, 2, 2, 1, 16>(a, s); break;
                case 32: launch_t<128, 2, 2, 1, 32>(a, s); break;
                case 33: launch_t<128, 2, 2, 1, 33>(a, s); break;
                case 46: launch_t<128, 2, 2, 1, 46>(a, s); break;
                case 62: launch_t<128, 2, 2, 1, 62>(a, s); break;
                default: launch_t<128, 2, 2, 1>(a, s);
            }
#else
            launch_t<128, 2, 2, 1>(a, s);
#endif
        } else launch_t<64, 4, 1, 1>(a, s);
    } else {
        if (a.N % 128 == 0) launch_t<128, 2, 2, 0>(a, s); else launch_t<64, 4, 1, 0>(a, s);
    }
    if (mfma_flops) *mfma_flops = (a.prec == 1 ? 3.0 : 1.0) * 2.0 * (double)a.M * k * (double)a.N;
    return 2.0 * (double)a.M * k * (double)a.Nreal;
}

}  // namespace nhans
