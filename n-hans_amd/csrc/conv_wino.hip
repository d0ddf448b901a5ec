// Stride-1 k x k convolution of the residual stack as a 1-D Winograd convolution along the image width,
// F(m, k) with 8 transformed positions (m = 5 outputs per tile for 4-tap filter rows, 6 for 3-tap rows;
// interpolation points 0, +-1, +-2, +-1/2, infinity), plain accumulation over the filter rows and channels:
//
//     V_p[h, j, c]  = sum_x BT[p][x] * in[h, j*m - pl + x, c]                      input transform (VALU, f32)
//     M_p[ho, j, n] = sum_kh sum_c V_p[ho + kh - pt, j, c] * U_p[kh][c][n]         8 independent GEMMs (MFMA)
//     out[ho, j*m + i, n] = epilogue( sum_p AT[i][p] * M_p[ho, j, n] )             output transform + block epilogue
//
// with U_p[kh] = sum_kw G[p][kw] w[kh][kw] folded on the host (fold.py: pack_wino).  8*KH products per tile and
// channel pair instead of m*k*KH: 2.5 x (k = 4) / 2.25 x (k = 3) fewer MFMAs than the direct form in
// conv_igemm_halo.hip -- on a chip that runs this workload at its socket power cap with 80 % of a tile's energy in
// the three split-f16 MFMA products per MAC, MACs are the one thing left to cut.  Accuracy:
// tests/winograd_probe.py (the transforms run in f32 on the hi+lo value, V is re-split; end to end the logits move
// by < 1e-6 against the direct form).  Why 1-D and not F(2x2, k x k): the 2-D forms need 16-36 live accumulator
// sets per tile, i.e. either tiny tiles (no reuse of U: the transformed weights would stream from L2 once per
// 16-32 tile-pixels) or position groups that go through HBM; nested 1-D keeps one accumulator set per position and
// wave, K = KH*C long, and each V row is reused by all KH filter rows.
//
// Workgroup = 8 consumer waves + 4 producer waves, one 64-channel block of one frame's TR x TJ block of
// tile-pixels (TR rows x TJ tiles, <= 64; tile-pixel q = r*TJ + t):
//   * consumer wave p owns position p: accumulators M_p[64 tile-pixels][64 channels] (2 x 2 MFMA tiles, the same
//     wave tile as the halo kernel), its A operand V_p comes from LDS, its B operand U_p -- private to the wave,
//     so nothing to share through LDS -- straight from L2 into registers, one k-step (16 channels of one filter
//     row) ahead.
//   * the producers read the (TR + KH - 1) x TJ input tiles of a 16-channel chunk from global memory into
//     registers (8 pixels x 8 channels per thread), transform, re-split and write V into LDS in exactly the
//     order the MFMA fragments are read: plane (p, hi|lo, k-group) holds one 16-byte piece per (row, tile) slot,
//     slot = rowslot*TJ + t, so the A fragment of filter row kh is the fragment of kh = 0 shifted by kh*TJ slots
//     and every ds_read_b128 covers 32 consecutive pieces (conflict-free).
//   * V is double-buffered by chunk; ONE barrier per chunk joins all twelve waves.
//   * epilogue: the eight M_p tiles go to LDS, conv_epilogue_sweep8<WINO> combines them with AT per output column
//     and applies the usual fused block epilogue (bias, position table, residual, ReLU, split store).
//
// Eligibility (launcher): split-f16 mode, one segment, stride 1, SAME padding, KH in {3, 4} == KW, Cin % 16 == 0,
// N % 64 == 0, split-NHWC input and output, 8-channel epilogue path.
#include "conv_epilogue.h"

#include <algorithm>

namespace nhans {

namespace {
constexpr int WCW = 8;                         // consumer waves = transformed positions
constexpr int WPW = 4;                         // producer waves
constexpr int W_NSLOT = 96;                    // (row, tile) slots per plane: >= 64 + (KH-1)*TJ
constexpr int W_PLANE = W_NSLOT * 4 + 4;       // floats per plane (16 B per slot, 16 B of skew between planes)
constexpr int W_VBUF = 32 * W_PLANE;           // floats per V buffer: 8 positions x {hi, lo} x 2 k-groups
constexpr int W_LDM = 68;                      // epilogue: floats per tile-pixel row of an M_p tile (64 + 4)

typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ f16x8 as_h8(f32x4 v) { return __builtin_bit_cast(f16x8, v); }

// value of a split-f16 element: (float)hi.half[SEL] + (float)lo.half[SEL] in ONE instruction (v_fma_mix_f32 reads
// f16 halves of 32-bit registers as sources of an f32 fma; the compiler spends two conversions and an add on it)
template <int SEL> __device__ __forceinline__ float unsplit_mix(float hi_pair, float lo_pair) {
    float d;
    if constexpr (SEL == 0) asm("v_fma_mix_f32 %0, %1, 1.0, %2 op_sel:[0,0,0] op_sel_hi:[1,0,1]" : "=v"(d) : "v"(hi_pair), "v"(lo_pair));
    else asm("v_fma_mix_f32 %0, %1, 1.0, %2 op_sel:[1,0,1] op_sel_hi:[1,0,1]" : "=v"(d) : "v"(hi_pair), "v"(lo_pair));
    return d;
}
}  // namespace

template <int KH, int MO>
__global__ void __launch_bounds__((WCW + WPW) * 64) conv_wino(const ConvArgs a) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);

    // XCD-aware, bijective remap of the linear workgroup id (each XCD owns a contiguous range of blocks)
    int L;
    {
        const int nblk = gridDim.x, bid = blockIdx.x;
        const int q = nblk >> 3, r = nblk & 7, xcd = bid & 7, idx = bid >> 3;
        L = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
    }
    const int nnb = a.N >> 6;
    const int nb = L % nnb;  L /= nnb;           // the channel blocks of one pixel block are neighbours (shared input in L2)
    const int cb = L % a.wino_ncb;  L /= a.wino_ncb;
    const int rb = L % a.wino_nrb;
    const int b = L / a.wino_nrb;
    const int TR = a.wino_tr, TJ = a.wino_tj;
    const int r0 = rb * TR, j0 = cb * TJ;
    const ConvSeg& g = a.seg[0];
    const int C = g.C, NC = C >> 4;              // 16-channel chunks

    if (wave >= WCW) {
        // =========================================================================================
        // Producers: thread = (slot, k-group): 8 input pixels x 8 channels of one (row, tile).
        const int ptid = tid - WCW * 64;
        const int nu = (TR + KH - 1) * TJ * 2;
        const bool active = ptid < nu;
        const bool wave_active = (wave - WCW) * 64 < nu;              // (uniform: an idle wave only keeps the barriers)
        const int kg = ptid & 1, slot = active ? ptid >> 1 : W_NSLOT - 1;   // idle lanes of a live wave write the spare last slot
        const int rs = slot / TJ, tj = slot - rs * TJ;
        const int hrow = r0 + rs - g.pt;
        const int wi0 = (j0 + tj) * MO - g.pl;
        const bool rowok = active && (unsigned)hrow < (unsigned)g.H;
        // Raw loads go through a buffer descriptor of the frame's image (base + 32-bit byte offset, 8 offset registers
        // instead of 16 address pairs; an out-of-range offset returns zeros, which is exactly what padding columns,
        // padding rows and unused slots need).  Split NHWC: a 32-channel group of a pixel is 64 B of hi halfs
        // followed by 64 B of lo halfs.
        const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(
            const_cast<float*>(g.src) + (size_t)b * g.H * g.W * C, 0, g.H * g.W * C * 4, 0x00020000);
        const unsigned voff0 = (unsigned)(((hrow * g.W + wi0) * C + kg * 4) * 4);
        unsigned okmask = 0;
#pragma unroll
        for (int x = 0; x < 8; ++x) okmask |= (rowok && (unsigned)(wi0 + x) < (unsigned)g.W) ? 1u << x : 0u;
        const unsigned pixb = (unsigned)C * 4u;
        float* const vw = smem + kg * W_PLANE + slot * 4;            // + buf*W_VBUF + (p*2 + h)*2*W_PLANE

        f32x4 rh[8], rl[8];
#define NW_LOAD_RAW(CC)                                                                            \
    {                                                                                              \
        const int co_ = (((CC) >> 1) * 32 + ((CC) & 1) * 8) * 4;                                   \
        _Pragma("unroll") for (int x = 0; x < 8; ++x) {                                            \
            const unsigned vo_ = ((okmask >> x) & 1) ? voff0 + x * pixb : 0x80000000u;             \
            rh[x] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsrc, vo_, co_, 0)); \
            rl[x] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsrc, vo_ + 64u, co_, 0)); \
        }                                                                                          \
    }
        // BT, structured (fold.py: WINO_BT): even and odd parts of rows 1..6 share their sums.  Two channels at a
        // time (f32x2: packed-f32 instructions).
#define NW_TRANSFORM(D, V)                                                                         \
    {                                                                                              \
        const f32x2 e1_ = D[2] + D[6] - 4.25f * D[4];                                              \
        const f32x2 o1_ = D[1] + D[5] - 4.25f * D[3];                                              \
        const f32x2 e2_ = 0.25f * D[2] - 1.25f * D[4] + D[6];                                      \
        const f32x2 o2_ = 0.5f * D[1] - 2.5f * D[3] + 2.f * D[5];                                  \
        const f32x2 e3_ = 4.f * D[2] - 5.f * D[4] + D[6];                                          \
        const f32x2 o3_ = 2.f * D[1] - 2.5f * D[3] + 0.5f * D[5];                                  \
        V[0] = 5.25f * (D[2] - D[4]) + (D[6] - D[0]);                                              \
        V[1] = e1_ + o1_;  V[2] = e1_ - o1_;                                                       \
        V[3] = e2_ + o2_;  V[4] = e2_ - o2_;                                                       \
        V[5] = e3_ + o3_;  V[6] = e3_ - o3_;                                                       \
        V[7] = 5.25f * (D[3] - D[5]) + (D[7] - D[1]);                                              \
    }
#define NW_PRODUCE(BUF)                                                                            \
    {                                                                                              \
        float* vb_ = vw + (BUF) * W_VBUF;                                                          \
        /* two channels (one register of hi halfs, one of lo halfs per pixel) at a time; a V that leaves the f16     \
           range becomes inf here and inf / NaN in this layer's output, where the epilogue raises the saturation flag */ \
        _Pragma("unroll") for (int cp = 0; cp < 4; ++cp) {                                         \
            f32x2 d_[8], v_[8];                                                                    \
            _Pragma("unroll") for (int x = 0; x < 8; ++x) {                                        \
                const float hp_ = rh[x][cp], lp_ = rl[x][cp];                                      \
                d_[x] = f32x2{unsplit_mix<0>(hp_, lp_), unsplit_mix<1>(hp_, lp_)};                 \
            }                                                                                      \
            NW_TRANSFORM(d_, v_)                                                                   \
            _Pragma("unroll") for (int p = 0; p < 8; ++p) {                                        \
                const f16x2 h_ = {(_Float16)v_[p].x, (_Float16)v_[p].y};                           \
                const f16x2 l_ = {(_Float16)__builtin_fmaf((float)h_[0], -1.0f, v_[p].x),          \
                                  (_Float16)__builtin_fmaf((float)h_[1], -1.0f, v_[p].y)};         \
                *reinterpret_cast<f16x2*>(vb_ + (p * 2 + 0) * 2 * W_PLANE + cp) = h_;              \
                *reinterpret_cast<f16x2*>(vb_ + (p * 2 + 1) * 2 * W_PLANE + cp) = l_;              \
            }                                                                                      \
            __builtin_amdgcn_sched_barrier(0);                                                     \
        }                                                                                          \
    }
        // chunk cc+1 is transformed while the consumers multiply chunk cc; iteration -1 is the prologue
        if (wave_active) NW_LOAD_RAW(0)
#pragma unroll 1
        for (int cc = -1; cc < NC; ++cc) {
            if (wave_active && cc + 1 < NC) {
                NW_PRODUCE((cc + 1) & 1)                                 // (its buffer was last read during chunk cc-1)
                if (cc + 2 < NC) NW_LOAD_RAW(cc + 2)
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
        }
        return;                                                         // the epilogue's barrier counts live waves only
#undef NW_LOAD_RAW
#undef NW_TRANSFORM
#undef NW_PRODUCE
    }

    // =============================================================================================
    // Consumer wave p = position p.
    const int p = wave;
    const int g8 = lane >> 5;
    const float* ub = a.wino_u + ((size_t)(nb * 8 + p) * NC * KH) * 1024 + lane * 4;
    const int aoff = (p * 4 + g8) * W_PLANE + (lane & 31) * 4;          // plane (p, hi, g8); lo: + 2*W_PLANE
    const int nks = NC * KH;

    f32x16 acc[2][2];
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[t][j][r] = 0.f;

    f32x4 fb[2][2][2];                      // [ring][n-tile][hi|lo]
    f32x4 fa[2][2][2];                      // [ring][m-tile][hi|lo]
#define NW_LOAD_B(RING, KS)                                                                        \
    {                                                                                              \
        const float* u_ = ub + (size_t)((KS) < nks ? (KS) : nks - 1) * 1024;                       \
        _Pragma("unroll") for (int j = 0; j < 2; ++j)                                              \
            _Pragma("unroll") for (int h = 0; h < 2; ++h)                                          \
                fb[RING][j][h] = *reinterpret_cast<const f32x4*>(u_ + (j * 2 + h) * 256);          \
    }
#define NW_READ_A(RING, BUF, KHI)                                                                  \
    {                                                                                              \
        const float* v_ = smem + (BUF) * W_VBUF + aoff + (KHI) * TJ * 4;                           \
        _Pragma("unroll") for (int t = 0; t < 2; ++t) {                                            \
            fa[RING][t][0] = *reinterpret_cast<const f32x4*>(v_ + t * 128);                        \
            fa[RING][t][1] = *reinterpret_cast<const f32x4*>(v_ + t * 128 + 2 * W_PLANE);          \
        }                                                                                          \
    }
#define NW_MFMA(RA, RB)                                                                            \
    {                                                                                              \
        _Pragma("unroll") for (int pr = 0; pr < 3; ++pr)                                           \
            _Pragma("unroll") for (int t = 0; t < 2; ++t)                                          \
                _Pragma("unroll") for (int j = 0; j < 2; ++j) {                                    \
                    const f16x8 a_ = as_h8(pr == 0 ? fa[RA][t][1] : fa[RA][t][0]);                 \
                    const f16x8 b_ = as_h8(pr == 1 ? fb[RB][j][1] : fb[RB][j][0]);                 \
                    acc[t][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(b_, a_, acc[t][j], 0, 0, 0); \
                }                                                                                  \
    }
    static_assert(KH % 2 == 0, "the two-deep operand rings assume an even number of filter rows per chunk");
    NW_LOAD_B(0, 0)
    __builtin_amdgcn_s_barrier();                                       // V of chunk 0 is in LDS
    for (int cc = 0; cc < NC; ++cc) {
        const int buf = cc & 1;
        NW_READ_A(0, buf, 0)
#pragma unroll
        for (int kh = 0; kh < KH; ++kh) {
            NW_LOAD_B((kh + 1) & 1, cc * KH + kh + 1)
            if (kh + 1 < KH) NW_READ_A((kh + 1) & 1, buf, kh + 1)
            NW_MFMA(kh & 1, kh & 1)
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
    }
#undef NW_LOAD_B
#undef NW_READ_A
#undef NW_MFMA

    // ---- epilogue: M_p tiles -> LDS, output transform + fused block epilogue --------------------
    float* ct = smem;
    int4* rowinfo = reinterpret_cast<int4*>(smem + 8 * 64 * W_LDM);
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int q4 = 0; q4 < 4; ++q4) {
                const f32x4 v = {acc[t][j][4 * q4], acc[t][j][4 * q4 + 1], acc[t][j][4 * q4 + 2], acc[t][j][4 * q4 + 3]};
                *reinterpret_cast<f32x4*>(ct + (p * 64 + t * 32 + (lane & 31)) * W_LDM + j * 32 + 8 * q4 + 4 * g8) = v;
            }
    // local pixel P = i*64 + q: output column i of tile-pixel q
    for (int P = tid; P < 64 * MO; P += WCW * 64) {
        const int i = P >> 6, q = P & 63;
        const int r = q / TJ, t = q - r * TJ;
        const int ho = r0 + r, wo = (j0 + t) * MO + i;
        const bool ok = r < TR && ho < a.Ho && j0 + t < a.wino_ntile && wo < a.Wo;
        const int rem = ok ? ho * a.Wo + wo : 0;                        // (slots that are not stored look like the frame's first pixel)
        const int m = b * a.Ho * a.Wo + rem;
        const int clip = a.img_clip ? a.img_clip[b] : 0;
        const int hh = ok ? ho : 0, ww = ok ? wo : 0;
        const int ids = (b * a.idH + hh * a.idsh) * a.idW + ww * a.idsw;
        rowinfo[P] = make_int4(clip * a.cb_stride, rem * a.N, ok ? m : -1, ids);
    }
    __syncthreads();
    const int c8 = tid & 7, prow8 = tid >> 3;
    const int n8 = nb * 64 + c8 * 8;
    switch (a.id_mode) {                                               // one frame = one clip: the bias is loaded once
        case 0: conv_epilogue_sweep8<1, 0, 1, 64, MO, W_LDM, MO>(a, ct, rowinfo, prow8, c8, n8); break;
        case 1: conv_epilogue_sweep8<1, 1, 1, 64, MO, W_LDM, MO>(a, ct, rowinfo, prow8, c8, n8); break;
        default: conv_epilogue_sweep8<1, 3, 1, 64, MO, W_LDM, MO>(a, ct, rowinfo, prow8, c8, n8); break;
    }
}

namespace {
constexpr size_t kWinoLds = (size_t)(8 * 64 * W_LDM) * sizeof(float) + 64 * 6 * sizeof(int4);
static_assert(kWinoLds >= (size_t)2 * W_VBUF * sizeof(float) && kWinoLds <= 160 * 1024, "LDS of a gfx950 CU");

// tile-pixel block (rows x tiles) for an Ho x ntile grid: the largest useful fraction of 64-slot blocks, subject to
// the producers' 256 threads (2 per slot) and the slots of a plane
void wino_block(int Ho, int ntile, int KH, int* tr, int* tj) {
    double best = -1;
    for (int r = 1; r <= 64; ++r)
        for (int t = 1; r * t <= 64; ++t) {
            if ((r + KH - 1) * t * 2 > WPW * 64 || 64 + (KH - 1) * t > W_NSLOT) continue;
            const int nrb = (Ho + r - 1) / r, ncb = (ntile + t - 1) / t;
            // useful fraction of the MFMA work, discounted by the rows the producers transform per output row
            const double u = (double)Ho * ntile / ((double)nrb * ncb * 64) - 0.02 * (double)(r + KH - 1) / r;
            if (u > best) { best = u; *tr = r; *tj = t; }
        }
}

template <int KH, int MO> void launch_wino_t(const ConvArgs& a, hipStream_t s) {
    static unsigned long long attr_devices = 0;
    set_max_dynamic_lds(reinterpret_cast<const void*>(&conv_wino<KH, MO>), kWinoLds, &attr_devices, "conv_wino");
    const int frames = a.M / (a.Ho * a.Wo);
    const int grid = frames * a.wino_nrb * a.wino_ncb * (a.N / 64);
    NHANS_LAUNCH("conv_wino", (conv_wino<KH, MO>), dim3(grid), dim3((WCW + WPW) * 64), kWinoLds, s, a);
}
}  // namespace

bool conv_wino_eligible(const ConvArgs& a) {
    const ConvSeg& g = a.seg[0];
    if (!a.wino || !a.wino_u || !a.wino_ws || a.prec != 1 || a.nseg != 1 || a.kgroup != 0) return false;
    if (g.KH != 4 || g.KW != g.KH || g.sh != 1 || g.sw != 1 || a.Wo != g.W || a.Ho != g.H) return false;   // (KH = 3: odd ring parity, not built yet)
    if (g.C % 16 != 0 || a.N % 64 != 0 || a.Nreal != a.N || !a.out_split || !a.epi8 || a.aux) return false;
    if (a.id_mode == 1 && !a.id_split) return false;
    if (a.M % (a.Ho * a.Wo) != 0) return false;
    // 32-bit element offsets inside the kernel
    return (double)a.M * std::max(g.C, a.N) + 65536.0 < 2147483648.0;
}

void launch_conv_wino(const ConvArgs& a0, hipStream_t s) {
    ConvArgs a = a0;
    const ConvSeg& g = a.seg[0];
    a.wino_m = 9 - g.KW;                                               // 8 positions: 5 outputs for 4 taps, 6 for 3
    a.wino_ntile = (a.Wo + a.wino_m - 1) / a.wino_m;
    wino_block(a.Ho, a.wino_ntile, g.KH, &a.wino_tr, &a.wino_tj);
    a.wino_nrb = (a.Ho + a.wino_tr - 1) / a.wino_tr;
    a.wino_ncb = (a.wino_ntile + a.wino_tj - 1) / a.wino_tj;
    // AT [m][8] on the points 0, 1, -1, 2, -2, 1/2, -1/2, infinity (fold.py: wino_matrices)
    const float pts[7] = {0.f, 1.f, -1.f, 2.f, -2.f, 0.5f, -0.5f};
    for (int i = 0; i < 6; ++i)
        for (int p = 0; p < 8; ++p) {
            float v = 0.f;
            if (i < a.wino_m) {
                if (p < 7) { v = 1.f; for (int e = 0; e < i; ++e) v *= pts[p]; }
                else v = i == a.wino_m - 1 ? 1.f : 0.f;
            }
            a.wino_at[i][p] = v;
        }
    a.ws = a.wino_ws;
    launch_wino_t<4, 5>(a, s);
}

}  // namespace nhans
