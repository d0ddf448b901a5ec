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

namespace nhans {

typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int BM = 128, BK = 32, LDA = 36;

template <int BN, int WM, int WN>
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

    // per-segment row state
    const float* rptr[4];
    int hi0[4], wi0[4];
    int seg = 0, kh = 0, kw = 0, c0 = 0, chunk_in_seg = 0;
    int sH = 0, sW = 0, sC = 0, sKW = 0, sKH = 0;
    const float* swpk = nullptr;
    auto enter_segment = [&](int s) {
        const ConvSeg& g = a.seg[s];
        sH = g.H; sW = g.W; sC = g.C; sKW = g.KW; sKH = g.KH; swpk = g.wpk;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            if (rb[i] >= 0) {
                hi0[i] = rho[i] * g.sh - g.pt;
                wi0[i] = rwo[i] * g.sw - g.pl;
                rptr[i] = g.src + (((int64_t)rb[i] * g.H + hi0[i]) * g.W + wi0[i]) * (int64_t)g.C + col4 * 4;
            } else {
                hi0[i] = -(1 << 28); wi0[i] = 0; rptr[i] = g.src;
            }
        }
        kh = 0; kw = 0; c0 = 0; chunk_in_seg = 0;
    };

    float4 ra[4];
    float4 rbv[BN / 32];
    auto issue_loads = [&]() {
        const int off = (kh * sW + kw) * sC + c0;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const bool ok = (unsigned)(hi0[i] + kh) < (unsigned)sH && (unsigned)(wi0[i] + kw) < (unsigned)sW;
            ra[i] = ok ? *reinterpret_cast<const float4*>(rptr[i] + off) : make_float4(0.f, 0.f, 0.f, 0.f);
        }
        const float4* bsrc = reinterpret_cast<const float4*>(
            swpk + ((size_t)chunk_in_seg * (a.N / 32) + nt0) * 1024);
#pragma unroll
        for (int j = 0; j < BN / 32; ++j) rbv[j] = bsrc[j * 256 + tid];
    };
    auto advance = [&]() {      // move the (seg, kh, kw, c0) cursor to the next chunk
        ++chunk_in_seg;
        c0 += BK;
        if (c0 >= sC) {
            c0 = 0;
            if (++kw >= sKW) {
                kw = 0;
                if (++kh >= sKH) {
                    ++seg;
                    if (seg < a.nseg) enter_segment(seg);
                }
            }
        }
    };
    auto store_lds = [&](int buf) {
        float* Ab = As + buf * A_BUF;
        float* Bb = Bs + buf * B_BUF;
#pragma unroll
        for (int i = 0; i < 4; ++i)
            *reinterpret_cast<float4*>(Ab + ((tid >> 3) + 32 * i) * LDA + col4 * 4) = ra[i];
#pragma unroll
        for (int j = 0; j < BN / 32; ++j) *reinterpret_cast<float4*>(Bb + (j * 256 + tid) * 4) = rbv[j];
    };

    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    int total = 0;
    for (int s = 0; s < a.nseg; ++s) total += a.seg[s].nchunks;

    enter_segment(0);
    issue_loads();
    store_lds(0);
    __syncthreads();

    const int arow = (wm * TM * 32 + (lane & 31)) * LDA + (lane >> 5) * 4;
    const int bcol = (wn * TN) * 1024 + lane * 4;

    for (int it = 0; it < total; ++it) {
        const int cur = it & 1;
        const bool more = it + 1 < total;
        if (more) {
            advance();
            issue_loads();
        }
        const float* Ab = As + cur * A_BUF + arow;
        const float* Bb = Bs + cur * B_BUF + bcol;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            float4 av[TM], bv[TN];
#pragma unroll
            for (int i = 0; i < TM; ++i) av[i] = *reinterpret_cast<const float4*>(Ab + i * 32 * LDA + q * 8);
#pragma unroll
            for (int j = 0; j < TN; ++j) bv[j] = *reinterpret_cast<const float4*>(Bb + j * 1024 + q * 256);
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j) {
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[i].x, bv[j].x, acc[i][j], 0, 0, 0);
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[i].y, bv[j].y, acc[i][j], 0, 0, 0);
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[i].z, bv[j].z, acc[i][j], 0, 0, 0);
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[i].w, bv[j].w, acc[i][j], 0, 0, 0);
                }
        }
        if (more) store_lds(cur ^ 1);
        __syncthreads();
    }

    // ---- epilogue.  C/D layout of the 32x32 MFMA: col = lane&31, row = (r&3) + 8*(r>>2) + 4*(lane>>5)
    const int ncol0 = nt * BN + wn * TN * 32 + (lane & 31);
#pragma unroll
    for (int i = 0; i < TM; ++i) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int row = wm * TM * 32 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
            const int m = m0 + row;
            if (m >= a.M) continue;
            const uint32_t b = fd_div((uint32_t)m, a.fdHoWo);
            const uint32_t rem = (uint32_t)m - b * a.fdHoWo.d;
            const uint32_t ho = fd_div(rem, a.fdWo);
            const uint32_t wo = rem - ho * a.fdWo.d;
            const int clip = a.img_clip ? a.img_clip[b] : 0;
            float idsv = 0.f;
            if (a.id_mode == 2) idsv = a.id[((size_t)b * a.idH + ho * a.idsh) * a.idW + wo * a.idsw];
#pragma unroll
            for (int j = 0; j < TN; ++j) {
                const int n = ncol0 + j * 32;
                float v = acc[i][j][r] + a.cb[(size_t)clip * a.cb_stride + n];
                if (a.ts) v += a.ts[ho * a.N + n];
                if (a.fs) v += a.fs[wo * a.N + n];
                if (n < a.Nreal) {
                    if (a.aux) a.aux[(size_t)m * a.aux_ld + n] = v;
                    if (a.id_mode == 1) v += a.idw[n] * a.id[(size_t)m * a.id_ld + n];
                    else if (a.id_mode == 2) v += a.idw[n] * idsv;
                    if (a.relu) v = fmaxf(v, 0.f);
                    a.out[(size_t)m * a.ldo + n] = v;
                }
            }
        }
    }
}

template <int BN, int WM, int WN>
static void launch_t(const ConvArgs& a, hipStream_t s) {
    constexpr size_t lds = (2 * BM * LDA + 2 * BK * BN) * sizeof(float);
    static bool attr_set = false;
    if (!attr_set) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_igemm_f32<BN, WM, WN>),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        attr_set = true;
    }
    const int mtiles = (a.M + BM - 1) / BM;
    const int grid = mtiles * (a.N / BN);
    hipLaunchKernelGGL((conv_igemm_f32<BN, WM, WN>), dim3(grid), dim3(256), lds, s, a);
}

double launch_conv_igemm(const ConvArgs& a, hipStream_t s) {
    double k = 0;
    for (int i = 0; i < a.nseg; ++i) k += (double)a.seg[i].nchunks * BK;
    if (a.N % 128 == 0) launch_t<128, 2, 2>(a, s);
    else launch_t<64, 4, 1>(a, s);
    return 2.0 * (double)a.M * k * (double)a.Nreal;
}

}  // namespace nhans
