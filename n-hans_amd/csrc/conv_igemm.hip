// Implicit-GEMM NHWC convolution on the gfx950 f32 matrix cores, with the N-HANS block epilogue
// (conditioning bias + position tables + residual + ReLU; BatchNorm is folded into the weights
// and tables on the host) fused in.
//
// Replaces tf.nn.conv2d + bias + broadcast adds + tf.nn.batch_normalization + tf.nn.relu of
// SN/blocks.py:38-48,104-108 and SN/main.py:102-124,161-187,232-238.
//
// Tiling (one workgroup = 4 wavefronts of 64 lanes, 128 output pixels x BN output channels):
//   * K is walked in chunks of 32 input channels of one filter tap.  The A chunk (128 pixels x 32
//     channels, gathered with SAME zero padding) and the B chunk (32 x BN, pre-packed on the host
//     in MFMA fragment order) are double-buffered in LDS; global loads for chunk i+1 are issued
//     before the MFMAs of chunk i and written to LDS after them (one barrier per chunk).
//   * v_mfma_f32_32x32x2_f32: lane l supplies A[row = l&31][k = l>>5] and B[k = l>>5][col = l&31].
//     K order inside a chunk is permuted so each lane reads its operands as 16-byte vectors:
//     MFMA (q, e) consumes k = 8q + 4(l>>5) + e, i.e. one ds_read_b128 per q per 32-row tile.
//   * LDS A rows are padded to 36 floats: the 16 rows of a ds_read_b128 lane group then start on
//     16 distinct 16-byte bank slots (36*i mod 64 covers all multiples of 4), so reads are
//     conflict-free; B fragments are lane-linear.
//   * Workgroup ids are remapped so each XCD (private L2) owns a contiguous range of pixel tiles:
//     vertically adjacent tiles re-read the same input rows for neighbouring taps.
#include "nhans_kernels.h"
#include <cstdlib>

namespace nhans {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));   // native vector: selects stay in registers

constexpr int BM = 128, BK = 32, LDA = 36;

template <int BN, int WM, int WN, int PIPE>
__global__ void __launch_bounds__(256) conv_igemm_f32(const ConvArgs a) {
    constexpr int TM = BM / WM / 32, TN = BN / WN / 32;
    constexpr int A_BUF = BM * LDA, B_BUF = BK * BN;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* As = smem;
    float* Bs = smem + 2 * A_BUF;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
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

    // per-segment row state (kept in scalars / constant-indexed arrays so it stays in registers)
    int64_t roff0, roff1, roff2, roff3;
    int hi0[4], wi0[4];
    int seg = 0, kh = 0, kw = 0, c0 = 0, chunk_in_seg = 0;
    int sH, sW, sC, sKW, sKH;
    const float* swpk;
    const float* ssrc;

#define NH_ROW(I, ROFF)                                                                            \
    if (rb[I] >= 0) {                                                                              \
        hi0[I] = rho[I] * g.sh - g.pt;                                                             \
        wi0[I] = rwo[I] * g.sw - g.pl;                                                             \
        ROFF = (((int64_t)rb[I] * g.H + hi0[I]) * g.W + wi0[I]) * (int64_t)g.C + col4 * 4;          \
    } else {                                                                                       \
        hi0[I] = -(1 << 28); wi0[I] = 0; ROFF = 0;                                                 \
    }
#define NH_ENTER_SEGMENT(S)                                                                        \
    {                                                                                              \
        const ConvSeg& g = a.seg[S];                                                               \
        sH = g.H; sW = g.W; sC = g.C; sKW = g.KW; sKH = g.KH; swpk = g.wpk; ssrc = g.src;          \
        NH_ROW(0, roff0) NH_ROW(1, roff1) NH_ROW(2, roff2) NH_ROW(3, roff3)                        \
        kh = 0; kw = 0; c0 = 0; chunk_in_seg = 0;                                                  \
    }

    // Loads are unconditional (a padded / out-of-range tap reads the segment base instead) and the
    // zero mask is applied when the registers are written to LDS: a select right after the load
    // would force a wait on it before the MFMAs it is meant to overlap.
#define NH_LOAD_A(I, ROFF, RA, ROK)                                                                \
    ROK = (unsigned)(hi0[I] + kh) < (unsigned)sH && (unsigned)(wi0[I] + kw) < (unsigned)sW;       \
    RA = *reinterpret_cast<const f32x4*>(ssrc + (ROK ? ROFF + off : (int64_t)0));
#define NH_ISSUE_LOADS_S(S)                                                                        \
    {                                                                                              \
        const int off = (kh * sW + kw) * sC + c0;                                                  \
        NH_LOAD_A(0, roff0, ra0##S, rok0##S) NH_LOAD_A(1, roff1, ra1##S, rok1##S)                  \
        NH_LOAD_A(2, roff2, ra2##S, rok2##S) NH_LOAD_A(3, roff3, ra3##S, rok3##S)                  \
        const f32x4* bsrc = reinterpret_cast<const f32x4*>(                                        \
            swpk + ((size_t)chunk_in_seg * (a.N / 32) + nt0) * 1024);                              \
        rb0v##S = bsrc[tid];                                                                       \
        rb1v##S = bsrc[256 + tid];                                                                 \
        if constexpr (BN == 128) { rb2v##S = bsrc[512 + tid]; rb3v##S = bsrc[768 + tid]; }         \
    }
#define NH_ISSUE_LOADS() NH_ISSUE_LOADS_S(_x)

#define NH_ADVANCE()                                                                               \
    {                                                                                              \
        ++chunk_in_seg;                                                                            \
        c0 += BK;                                                                                  \
        if (c0 >= sC) {                                                                            \
            c0 = 0;                                                                                \
            if (++kw >= sKW) {                                                                     \
                kw = 0;                                                                            \
                if (++kh >= sKH) {                                                                 \
                    ++seg;                                                                         \
                    if (seg < a.nseg) NH_ENTER_SEGMENT(seg)                                        \
                }                                                                                  \
            }                                                                                      \
        }                                                                                          \
    }

#define NH_STORE_LDS_S(BUF, S)                                                                     \
    {                                                                                              \
        float* Ab_ = As + (BUF) * A_BUF;                                                           \
        float* Bb_ = Bs + (BUF) * B_BUF;                                                           \
        const f32x4 z4 = {0.f, 0.f, 0.f, 0.f};                                                     \
        float* Ar_ = Ab_ + (tid >> 3) * LDA + col4 * 4;                                            \
        *reinterpret_cast<f32x4*>(Ar_) = rok0##S ? ra0##S : z4;                                    \
        *reinterpret_cast<f32x4*>(Ar_ + 32 * LDA) = rok1##S ? ra1##S : z4;                         \
        *reinterpret_cast<f32x4*>(Ar_ + 64 * LDA) = rok2##S ? ra2##S : z4;                         \
        *reinterpret_cast<f32x4*>(Ar_ + 96 * LDA) = rok3##S ? ra3##S : z4;                         \
        *reinterpret_cast<f32x4*>(Bb_ + tid * 4) = rb0v##S;                                        \
        *reinterpret_cast<f32x4*>(Bb_ + (256 + tid) * 4) = rb1v##S;                                \
        if constexpr (BN == 128) {                                                                 \
            *reinterpret_cast<f32x4*>(Bb_ + (512 + tid) * 4) = rb2v##S;                            \
            *reinterpret_cast<f32x4*>(Bb_ + (768 + tid) * 4) = rb3v##S;                            \
        }                                                                                          \
    }
#define NH_STORE_LDS(BUF) NH_STORE_LDS_S(BUF, _x)

#define NH_COMPUTE(BUF)                                                                            \
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
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[i].x, bv[j].x, acc[i][j], 0, 0, 0); \
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[i].y, bv[j].y, acc[i][j], 0, 0, 0); \
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[i].z, bv[j].z, acc[i][j], 0, 0, 0); \
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[i].w, bv[j].w, acc[i][j], 0, 0, 0); \
                }                                                                                  \
        }                                                                                          \
    }

    f32x4 ra0_x, ra1_x, ra2_x, ra3_x, rb0v_x, rb1v_x, rb2v_x, rb3v_x;
    bool rok0_x, rok1_x, rok2_x, rok3_x;
    f32x4 ra0_y, ra1_y, ra2_y, ra3_y, rb0v_y, rb1v_y, rb2v_y, rb3v_y;   // second set: PIPE == 2 only
    bool rok0_y, rok1_y, rok2_y, rok3_y;

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

    NH_ENTER_SEGMENT(0)
    NH_ISSUE_LOADS()
    NH_STORE_LDS(0)
    __syncthreads();

    if constexpr (PIPE == 0) {
        // 2-stage: loads of chunk it+1 are in flight while the MFMAs of chunk it run
        for (int it = 0; it + 1 < total; ++it) {
            const int cur = it & 1;
            NH_ADVANCE()
            NH_ISSUE_LOADS()
            __builtin_amdgcn_sched_barrier(0);     // keep the loads above the MFMAs that hide them
            NH_COMPUTE(cur)
            __builtin_amdgcn_sched_barrier(0);
            NH_STORE_LDS(cur ^ 1)
            __syncthreads();
        }
        NH_COMPUTE((total - 1) & 1)
        __syncthreads();
    } else if constexpr (PIPE == 2) {
        // 4-stage, two register sets, loop unrolled by two, no scheduling fences: chunk it+2 is
        // loaded at the top of iteration it, chunk it+1 goes registers -> LDS at its bottom, so every
        // memory operation has a whole iteration of MFMAs to hide behind wherever the compiler's
        // scheduler interleaves it.
        if (total > 1) {
            NH_ADVANCE()
            NH_ISSUE_LOADS_S(_y)                    // chunk 1 -> set y
        }
        int it = 0;
        for (; it + 3 < total; it += 2) {
            NH_ADVANCE()
            NH_ISSUE_LOADS_S(_x)                    // chunk it+2 -> set x
            NH_COMPUTE(0)                           // chunk it (buffer 0)
            NH_STORE_LDS_S(1, _y)                   // chunk it+1 -> buffer 1
            __syncthreads();
            NH_ADVANCE()
            NH_ISSUE_LOADS_S(_y)                    // chunk it+3 -> set y
            NH_COMPUTE(1)                           // chunk it+1
            NH_STORE_LDS_S(0, _x)                   // chunk it+2 -> buffer 0
            __syncthreads();
        }
        // tail: it is even, 1..3 chunks left; chunk it is in buffer 0, chunk it+1 (if any) in set y
        if (it + 2 < total) {
            NH_ADVANCE()
            NH_ISSUE_LOADS_S(_x)                    // chunk it+2
        }
        NH_COMPUTE(0)
        if (it + 1 < total) {
            NH_STORE_LDS_S(1, _y)
            __syncthreads();
            NH_COMPUTE(1)
            if (it + 2 < total) {
                NH_STORE_LDS_S(0, _x)
                __syncthreads();
                NH_COMPUTE(0)
            }
        }
        __syncthreads();
    } else {
        // 3-stage: global loads run one full chunk ahead of the LDS write that consumes them, so
        // neither the LDS write (chunk it+1) nor the MFMAs (chunk it) ever wait on HBM/L2 latency.
        if (total > 1) {
            NH_ADVANCE()
            NH_ISSUE_LOADS()                        // chunk 1 -> registers
        }
        int it = 0;
        for (; it + 2 < total; ++it) {
            const int cur = it & 1;
            NH_STORE_LDS(cur ^ 1)                   // chunk it+1: registers -> LDS
            NH_ADVANCE()
            NH_ISSUE_LOADS()                        // chunk it+2 -> registers
            __builtin_amdgcn_sched_barrier(0);
            NH_COMPUTE(cur)
            __syncthreads();
        }
        if (it + 1 < total) {
            const int cur = it & 1;
            NH_STORE_LDS(cur ^ 1)
            __builtin_amdgcn_sched_barrier(0);
            NH_COMPUTE(cur)
            __syncthreads();
            ++it;
        }
        NH_COMPUTE(it & 1)
        __syncthreads();
    }

#undef NH_ROW
#undef NH_LOAD_A
#undef NH_ENTER_SEGMENT
#undef NH_ISSUE_LOADS
#undef NH_ISSUE_LOADS_S
#undef NH_STORE_LDS_S
#undef NH_ADVANCE
#undef NH_STORE_LDS
#undef NH_COMPUTE

    // ---- epilogue.  C/D layout of the 32x32 MFMA: col = lane&31, row = (r&3) + 8*(r>>2) + 4*(lane>>5).
    // Per-row (clip, ho, wo, id offset) is computed once per block into LDS (the main loop's final
    // barrier has retired every LDS read); each lane then issues all table / residual loads of a
    // group of rows before the first use, and stores only after the last load of the group.
    int4* rowinfo = reinterpret_cast<int4*>(smem);
    if (tid < BM) {
        int m = m0 + tid;
        if (m >= a.M) m = a.M - 1;
        const uint32_t b = fd_div((uint32_t)m, a.fdHoWo);
        const uint32_t rem = (uint32_t)m - b * a.fdHoWo.d;
        const uint32_t ho = fd_div(rem, a.fdWo);
        const uint32_t wo = rem - ho * a.fdWo.d;
        const int clip = a.img_clip ? a.img_clip[b] : 0;
        const int ids = (int)((b * a.idH + ho * a.idsh) * a.idW + wo * a.idsw);
        rowinfo[tid] = make_int4(clip * a.cb_stride, (int)ho * a.N, (int)wo * a.N, ids);
    }
    __syncthreads();
    const int ncol0 = nt * BN + wn * TN * 32 + (lane & 31);
    // Branch-free load phase: absent tables / residuals read a[0] of a zero word with weight 0, so
    // the compiler can issue every load of a row group before the first wait.
    const int f_ts = a.ts ? 1 : 0, f_fs = a.fs ? 1 : 0;
    const int f_id1 = a.id_mode == 1 ? 1 : 0, f_id2 = a.id_mode == 2 ? 1 : 0;
    const float* __restrict__ cbp = a.cb;
    const float* __restrict__ tsp = a.ts ? a.ts : a.zero;
    const float* __restrict__ fsp = a.fs ? a.fs : a.zero;
    const float* __restrict__ idp = a.id_mode ? a.id : a.zero;
    float idw[TN];
    int ncl[TN];
#pragma unroll
    for (int j = 0; j < TN; ++j) {
        const int n = ncol0 + j * 32;
        idw[j] = a.id_mode ? a.idw[n] : 0.f;
        ncl[j] = n < a.Nreal ? n : 0;
    }
#pragma unroll
    for (int i = 0; i < TM; ++i) {
#pragma unroll
        for (int rg = 0; rg < 4; ++rg) {              // 4 groups of 4 consecutive rows
            float v[4][TN];
            int mrow[4];
#pragma unroll
            for (int rr = 0; rr < 4; ++rr) {
                const int r = rg * 4 + rr;
                const int row = wm * TM * 32 + i * 32 + rr + 8 * rg + 4 * (lane >> 5);
                const int4 ri = rowinfo[row];
                int m = m0 + row;
                mrow[rr] = m;
                if (m >= a.M) m = a.M - 1;
                const int64_t idrow = f_id1 ? (int64_t)m * a.id_ld : (int64_t)ri.w * f_id2;
#pragma unroll
                for (int j = 0; j < TN; ++j) {
                    const int n = ncol0 + j * 32;
                    const float c = cbp[ri.x + n];
                    const float t = tsp[(ri.y + n) * f_ts];
                    const float f = fsp[(ri.z + n) * f_fs];
                    const float idv = idp[idrow + ncl[j] * f_id1];
                    const float x = ((acc[i][j][r] + c) + t) + f;
                    acc[i][j][r] = x;                  // pre-residual value (aux output)
                    v[rr][j] = x + idw[j] * idv;
                }
            }
#pragma unroll
            for (int rr = 0; rr < 4; ++rr) {
                const int r = rg * 4 + rr;
                const int m = mrow[rr];
                if (m < a.M) {
#pragma unroll
                    for (int j = 0; j < TN; ++j) {
                        const int n = ncol0 + j * 32;
                        if (n < a.Nreal) {
                            if (a.aux) a.aux[(size_t)m * a.aux_ld + n] = acc[i][j][r];
                            a.out[(size_t)m * a.ldo + n] = a.relu ? fmaxf(v[rr][j], 0.f) : v[rr][j];
                        }
                    }
                }
            }
        }
    }
}

template <int BN, int WM, int WN, int PIPE>
static void launch_t(const ConvArgs& a, hipStream_t s) {
    constexpr size_t lds = (2 * BM * LDA + 2 * BK * BN) * sizeof(float);
    static bool attr_set = false;
    if (!attr_set) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_igemm_f32<BN, WM, WN, PIPE>),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        attr_set = true;
    }
    const int mtiles = (a.M + BM - 1) / BM;
    const int grid = mtiles * (a.N / BN);
    hipLaunchKernelGGL((conv_igemm_f32<BN, WM, WN, PIPE>), dim3(grid), dim3(256), lds, s, a);
}

double launch_conv_igemm(const ConvArgs& a, hipStream_t s) {
    double k = 0;
    for (int i = 0; i < a.nseg; ++i) k += (double)a.seg[i].nchunks * BK;
    static const int pipe = [] { const char* e = getenv("NHANS_CONV_PIPE"); return e ? atoi(e) : 1; }();
    if (a.N % 128 == 0) {
        if (pipe == 2) launch_t<128, 2, 2, 2>(a, s); else if (pipe == 1) launch_t<128, 2, 2, 1>(a, s); else launch_t<128, 2, 2, 0>(a, s);
    } else {
        if (pipe == 2) launch_t<64, 4, 1, 2>(a, s); else if (pipe == 1) launch_t<64, 4, 1, 1>(a, s); else launch_t<64, 4, 1, 0>(a, s);
    }
    return 2.0 * (double)a.M * k * (double)a.Nreal;
}

}  // namespace nhans
