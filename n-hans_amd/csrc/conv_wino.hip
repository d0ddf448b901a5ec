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
//     so nothing to share through LDS -- straight from L2 into a register ring of one chunk (KH k-steps of 16
//     channels), three k-steps ahead.
//   * producer wave 11 stages the (TR + KH - 1) rows x (TJ*m + KH - 1) pixels of a 16-channel chunk in LDS by
//     LDS-DMA (coalesced rows, every pixel once), one chunk ahead of the transform; producer waves 8-10 read their
//     8 pixels x 8 channels per thread from there, transform, re-split and write V into LDS in exactly the order
//     the MFMA fragments are read: plane (p, hi|lo, k-group) holds one 16-byte piece per (row, tile) slot,
//     slot = rowslot*TJ + t, so the A fragment of filter row kh is the fragment of kh = 0 shifted by kh*TJ slots
//     and every ds_read_b128 covers 32 consecutive pieces (conflict-free).
//   * V and the staged tile are double-buffered by chunk; ONE barrier per chunk joins all twelve waves.
//   * epilogue: the eight M_p tiles go to LDS, wino_sweep forms the m output columns of a tile-pixel from them and
//     applies the usual fused block epilogue (bias, position table, residual, ReLU, saturation flag, split store).
//
// Eligibility (launcher): split-f16 mode, one segment, stride 1, SAME padding, 4x4 filters (the ring parity of the
// loop assumes an even KH; the 3x3 layers' images fit F(6,3) tile blocks badly and stay on the direct kernel),
// Cin % 16 == 0, N % 64 == 0, split-NHWC input and output.
#include "conv_wino_common.h"

namespace nhans {

namespace {
constexpr int WCW = 8;                         // consumer waves = transformed positions
constexpr int WPW = 4;                         // producer waves
constexpr int W_NSLOT = 96;                    // (row, tile) slots per plane: >= 64 + (KH-1)*TJ
constexpr int W_PLANE = W_NSLOT * 4 + 4;       // floats per plane (16 B per slot, 16 B of skew between planes)
constexpr int W_VBUF = 32 * W_PLANE;           // floats per V buffer: 8 positions x {hi, lo} x 2 k-groups
constexpr int W_RROWS = 12, W_RPX = 40;        // staged input tile of a chunk: rows (TR + KH - 1 <= 12) x pixels (TJ*m + 3 <= 40)
constexpr int W_RKIND = W_RROWS * W_RPX * 4;   // floats per piece kind (hi|lo x k-group): one 16-byte piece per (row, pixel)
constexpr int W_RAW = 4 * W_RKIND;             // floats per staged tile
constexpr int W_RAW_BASE = 2 * W_VBUF;         // the two staged tiles sit after the two V buffers
}  // namespace


template <int KH, int MO, int DBG = 0>      // DBG: dev tool, per-wave cycle stamps (tools/wino_phase_cycles.py)
__global__ void __launch_bounds__((WCW + WPW) * 64) conv_wino(const ConvArgs a) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    long long dbg_entry = 0;
    if constexpr (DBG) dbg_entry = (long long)__builtin_amdgcn_s_memtime();
    // XCD-aware, bijective remap of the linear workgroup id (each XCD owns a contiguous range of blocks)
    int L;
    {
        const int nblk = gridDim.x, bid = blockIdx.x;
        const int q = nblk >> 3, r = nblk & 7, xcd = bid & 7, idx = bid >> 3;
        L = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
    }
    // (frame, row block, column block, channel block), the channel blocks of one pixel block neighbours (shared input
    // in L2); multiply-shift divisions -- this decode sits in front of every wave's first instruction
    const int b = (int)fd_div((uint32_t)L, a.wino_fd_bpf);
    const int lb = L - b * (int)a.wino_fd_bpf.d;
    const int lt = (int)fd_div((uint32_t)lb, a.wino_fd_nnb);
    const int nb = lb - lt * (int)a.wino_fd_nnb.d;
    const int rb = (int)fd_div((uint32_t)lt, a.wino_fd_ncb);
    const int cb = lt - rb * (int)a.wino_fd_ncb.d;
    const int TR = a.wino_tr, TJ = a.wino_tj;
    const int r0 = rb * TR, j0 = cb * TJ;
    const ConvSeg& g = a.seg[0];
    const int C = g.C, NC = C >> 4;              // 16-channel chunks

    // Input tile of a chunk = 30 LDS-DMA wave-instructions: byte offset of this lane's 16-byte piece of instruction i
    // within the frame, chunk 0 (0xFFFFFFFF: padding / unused slot -> zero page)
    constexpr int NDMA = W_RAW / 4 / 64;
    static_assert(NDMA == 30, "counted waits");
    const char* const tile_fb = reinterpret_cast<const char*>(g.src + (size_t)b * g.H * g.W * C);
    const char* const tile_zp = reinterpret_cast<const char*>(a.zero + lane * 4);
    auto tile_offset = [&](int i) -> unsigned {
        const int q = i * 64 + lane;
        const int kind = q / (W_RROWS * W_RPX), rem = q - kind * (W_RROWS * W_RPX);
        const int row = rem / W_RPX, px = rem - row * W_RPX;
        const int hrow = r0 + row - g.pt, wcol = j0 * MO - g.pl + px;
        const bool ok = row < TR + KH - 1 && px < TJ * MO + KH - 1 && (unsigned)hrow < (unsigned)g.H && (unsigned)wcol < (unsigned)g.W;
        // split NHWC: a 32-channel group of a pixel is 64 B of hi halfs followed by 64 B of lo halfs
        return ok ? (unsigned)(((hrow * g.W + wcol) * C) * 4 + (kind & 1) * 16 + (kind >> 1) * 64) : 0xFFFFFFFFu;
    };

    if (wave >= WCW) {
        // =========================================================================================
        // Producers.  Wave 11 stages the input tile of a chunk in LDS by LDS-DMA: one 16-byte piece per (kind, row,
        // pixel), kind = (hi|lo, k-group), lanes = consecutive pixels of a row, so that a wave-instruction touches 16
        // cache lines instead of 64 -- the K loop of this kernel is bound by the CU's vector-memory request rate
        // (per chunk 128 one-KB weight fragments for the consumers; per-thread 16-byte gathers for the tile cost as much
        // again) -- and every pixel is fetched once although 1.6 tiles use it.  Padding pixels, padding rows and unused
        // slots come from the zero page.  Waves 8-10 transform: thread = (slot, k-group) = 8 pixels x 8 channels.
        const int ptid = tid - WCW * 64;
        const int nrows = TR + KH - 1;
        __builtin_amdgcn_s_setprio(3);                                // the waves everybody waits for
        if (wave == WCW + WPW - 1) {
            const char* const fb = tile_fb;
            const char* const zp = tile_zp;
            unsigned goff[NDMA];
#pragma unroll
            for (int i = 0; i < NDMA; ++i) {
                goff[i] = tile_offset(i);
                // the piece of chunk 0 leaves at once: its memory latency runs under the remaining address arithmetic
                // (all 30 offsets first, then 30 loads, put the first tile 1.5 k cycles later)
                const char* s0 = goff[i] != 0xFFFFFFFFu ? fb + goff[i] : zp;
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)s0,
                                                 (__attribute__((address_space(3))) void*)(smem + W_RAW_BASE + i * 256), 16, 0, 0);
            }
#define NW_DMA(CC, BUF)                                                                            \
    {                                                                                              \
        const unsigned co_ = (unsigned)((((CC) >> 1) * 32 + ((CC) & 1) * 8) * 4);                  \
        float* dst_ = smem + W_RAW_BASE + (BUF) * W_RAW;                                           \
        _Pragma("unroll") for (int i = 0; i < NDMA; ++i) {                                         \
            const char* s_ = (goff[i] != 0xFFFFFFFFu && !(kDev && (a.wino_m >> 8 & 1))) ? fb + (goff[i] + co_) : zp; \
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)s_,   \
                                             (__attribute__((address_space(3))) void*)(dst_ + i * 256), 16, 0, 0); \
        }                                                                                          \
    }
            // iteration cc: the tile of chunk cc+2 (its buffer was last read by the transform of chunk cc, one iteration
            // ago); it must have landed when the iteration's barrier opens, because that barrier opens its transform.
            // (Measured and rejected for the tile of chunk 1, whose round trip sits in front of the first MFMA: sent
            // right behind the tile of chunk 0 by this wave, or by an idle consumer wave, it delays the tile of chunk 0
            // by as much as it gains.)
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");            // (chunk 0 was issued with the offsets)
            __builtin_amdgcn_s_barrier();
#pragma unroll 1
            for (int cc = -1; cc < NC; ++cc) {
                if (cc + 2 < NC) NW_DMA(cc + 2, cc & 1)
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                __builtin_amdgcn_s_barrier();
            }
            return;
#undef NW_DMA
        }
        const int nu = nrows * TJ * 2;
        const bool active = ptid < nu;
        const bool wave_active = (wave - WCW) * 64 < nu;              // (uniform: an idle wave only keeps the barriers)
        // k-group major: the lanes of a wave write consecutive slots of ONE plane (16-byte stride: at worst a 2-way
        // bank conflict on the 8-byte stores); idle lanes of a live wave read slot 0 and write the spare last slot
        const int nsl = nu >> 1;
        const int kg = active ? (ptid >= nsl ? 1 : 0) : 0, slot = active ? ptid - kg * nsl : 0;
        const int rs = slot / TJ, tj = slot - rs * TJ;
        // hi pieces of the thread's 8 pixels: kind kg, row rs, pixels tj*m ..; lo pieces: kind 2 + kg.  Consecutive
        // lanes are m = 5 pixels = 20 dwords apart: conflict-free 16-byte reads.
        const float* const rr = smem + W_RAW_BASE + ((kg * W_RROWS + rs) * W_RPX + tj * MO) * 4;
        float* const vw = smem + kg * W_PLANE + (active ? slot : W_NSLOT - 1) * 4;   // + buf*W_VBUF + (p*2 + h)*2*W_PLANE

        f32x4 rh[8], rl[8];
#define NW_LOAD_RAW(BUF)                                                                           \
    {                                                                                              \
        _Pragma("unroll") for (int x = 0; x < 8; ++x) {                                            \
            rh[x] = *reinterpret_cast<const f32x4*>(rr + (BUF) * W_RAW + x * 4);                   \
            rl[x] = *reinterpret_cast<const f32x4*>(rr + (BUF) * W_RAW + x * 4 + 2 * W_RKIND);     \
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
        _Pragma("unroll") for (int hq = 0; hq < 2; ++hq) {                                         \
            uint2 oh_[8], ol_[8];                                                                  \
            _Pragma("unroll") for (int ep = 0; ep < 2; ++ep) {                                     \
                f32x2 d_[8], v_[8];                                                                \
                _Pragma("unroll") for (int x = 0; x < 8; ++x) {                                    \
                    const float hp_ = rh[x][hq * 2 + ep], lp_ = rl[x][hq * 2 + ep];                \
                    d_[x] = f32x2{unsplit_mix<0>(hp_, lp_), unsplit_mix<1>(hp_, lp_)};             \
                }                                                                                  \
                NW_TRANSFORM(d_, v_)                                                               \
                _Pragma("unroll") for (int p = 0; p < 8; ++p) {                                    \
                    if (ep == 0) split_pair(v_[p].x, v_[p].y, &oh_[p].x, &ol_[p].x);               \
                    else split_pair(v_[p].x, v_[p].y, &oh_[p].y, &ol_[p].y);                       \
                }                                                                                  \
                __builtin_amdgcn_sched_barrier(0);                                                 \
            }                                                                                      \
            _Pragma("unroll") for (int p = 0; p < 8; ++p) {                                        \
                *reinterpret_cast<uint2*>(vb_ + (p * 2 + 0) * 2 * W_PLANE + hq * 2) = oh_[p];      \
                *reinterpret_cast<uint2*>(vb_ + (p * 2 + 1) * 2 * W_PLANE + hq * 2) = ol_[p];      \
            }                                                                                      \
            __builtin_amdgcn_sched_barrier(0);                                                     \
        }                                                                                          \
    }
        // chunk cc+1 is transformed (from the tile wave 11 staged one iteration earlier) while the consumers multiply
        // chunk cc; the first barrier belongs to the staging of chunk 0, iteration -1 is the first transform
        long long dbg_setup = 0, dbg_landed = 0;
        if constexpr (DBG) dbg_setup = (long long)__builtin_amdgcn_s_memtime() - dbg_entry;
        __builtin_amdgcn_s_barrier();
        if constexpr (DBG) dbg_landed = (long long)__builtin_amdgcn_s_memtime() - dbg_entry;
        long long dbg_prod = 0, dbg_bar = 0, dbg_first = 0;
#pragma unroll 1
        for (int cc = -1; cc < NC; ++cc) {
            long long t0 = 0, t1 = 0;
            if constexpr (DBG) t0 = (long long)__builtin_amdgcn_s_memtime();
            if (wave_active && cc + 1 < NC && !(DBG && (a.wino_m >> 8 & 4) && cc >= 0)) {
                NW_LOAD_RAW((cc + 1) & 1)
                NW_PRODUCE((cc + 1) & 1)                                 // (its buffer was last read during chunk cc-1)
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            if constexpr (DBG) t1 = (long long)__builtin_amdgcn_s_memtime();
            __builtin_amdgcn_s_barrier();
            if constexpr (DBG) {
                const long long t2 = (long long)__builtin_amdgcn_s_memtime();
                if (cc < 0) dbg_first = t1 - t0; else { dbg_prod += t1 - t0; dbg_bar += t2 - t1; }
            }
        }
        if constexpr (DBG) {                                            // [total, first chunk (transform), later chunks, barrier waits]
            if (a.dbg && lane == 0) {
                long long* d = a.dbg + ((size_t)blockIdx.x * (WCW + WPW) + wave) * 4;
                d[0] = (long long)__builtin_amdgcn_s_memtime() - dbg_entry; d[1] = dbg_first; d[2] = dbg_prod; d[3] = dbg_bar;
                long long* e = a.dbg + (size_t)(4 << 20) + ((size_t)blockIdx.x * (WCW + WPW) + wave) * 8;
                e[0] = dbg_setup; e[1] = dbg_landed;
            }
        }
        return;                                                         // the epilogue's barrier counts live waves only
#undef NW_LOAD_RAW
#undef NW_TRANSFORM
#undef NW_PRODUCE
    }

    // =============================================================================================
    // Consumer wave p = position p.
    const int p = wave;
    // one frame = one clip: offset of its bias vector (uniform; fetched here, long before the epilogue needs it)
    const int cbx = __builtin_amdgcn_readfirstlane((a.img_clip ? a.img_clip[b] : 0) * a.cb_stride);
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

    // B (weights): ring of KH k-steps in registers, three k-steps ahead -- the fragments come straight from L2, and
    // with one k-step in flight per wave the loop was bound by that latency (Little: 32 KB in flight per CU at ~800
    // cycles = 29 B/clk against the 42 B/clk the MFMAs consume); A (V): read from LDS right before its k-step, the
    // partner wave of the SIMD covers the LDS latency.
    f32x4 fb[KH][2][2];                     // [ring][n-tile][hi|lo]
    f32x4 fa[2][2];                         // [m-tile][hi|lo]
#define NW_LOAD_B(RING, KS)                                                                        \
    {                                                                                              \
        const float* u_ = ub + (size_t)((kDev && (a.wino_m >> 8 & 16)) ? 0 : (KS) < nks ? (KS) : nks - 1) * 1024; \
        _Pragma("unroll") for (int j = 0; j < 2; ++j)                                              \
            _Pragma("unroll") for (int h = 0; h < 2; ++h)                                          \
                fb[RING][j][h] = *reinterpret_cast<const f32x4*>(u_ + (j * 2 + h) * 256);          \
    }
#define NW_READ_A(BUF, KHI)                                                                        \
    {                                                                                              \
        const float* v_ = smem + (BUF) * W_VBUF + aoff + (KHI) * TJ * 4;                           \
        _Pragma("unroll") for (int t = 0; t < 2; ++t) {                                            \
            fa[t][0] = *reinterpret_cast<const f32x4*>(v_ + t * 128);                              \
            fa[t][1] = *reinterpret_cast<const f32x4*>(v_ + t * 128 + 2 * W_PLANE);                \
        }                                                                                          \
    }
#define NW_MFMA(RB)                                                                                \
    {                                                                                              \
        _Pragma("unroll") for (int pr = 0; pr < 3; ++pr)                                           \
            _Pragma("unroll") for (int t = 0; t < 2; ++t)                                          \
                _Pragma("unroll") for (int j = 0; j < 2; ++j) {                                    \
                    const f16x8 a_ = as_h8(pr == 0 ? fa[t][1] : fa[t][0]);                         \
                    const f16x8 b_ = as_h8(pr == 1 ? fb[RB][j][1] : fb[RB][j][0]);                 \
                    acc[t][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(b_, a_, acc[t][j], 0, 0, 0); \
                }                                                                                  \
    }
    static_assert(KH == 4, "the weight ring is indexed by the filter row");
    // Only the first k-step of weights is requested at kernel entry, the next two under the first transform (which
    // uses LDS and the VALU, not the vector-memory path): all three at entry -- 96 KB per workgroup -- sat in the CU's
    // memory queue in front of the 30 KB everybody is waiting for, the first input tile (it landed 1.7 k cycles later).
    NW_LOAD_B(0, 0)
    long long dbg_setup = 0;
    if constexpr (DBG) dbg_setup = (long long)__builtin_amdgcn_s_memtime() - dbg_entry;
    __builtin_amdgcn_s_barrier();                                       // the input tile of chunk 0 is staged
    NW_LOAD_B(1, 1)
    NW_LOAD_B(2, 2)
    __builtin_amdgcn_s_barrier();                                       // V of chunk 0 is in LDS
    long long dbg_t0 = 0, dbg_bar = 0, dbg_t1 = 0;
    if constexpr (DBG) dbg_t0 = (long long)__builtin_amdgcn_s_memtime();
    for (int cc = 0; cc < NC; ++cc) {
        const int buf = cc & 1;
#pragma unroll
        for (int kh = 0; kh < KH; ++kh) {
            NW_READ_A(buf, kh)
            NW_LOAD_B((kh + 3) & 3, cc * KH + kh + 3)
            if (!(DBG && (a.wino_m >> 8 & 2))) NW_MFMA(kh)
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        long long tq = 0;
        if constexpr (DBG) tq = (long long)__builtin_amdgcn_s_memtime();
        __builtin_amdgcn_s_barrier();
        if constexpr (DBG) dbg_bar += (long long)__builtin_amdgcn_s_memtime() - tq;
    }
    if constexpr (DBG) dbg_t1 = (long long)__builtin_amdgcn_s_memtime();
#undef NW_LOAD_B
#undef NW_READ_A
#undef NW_MFMA

    // ---- epilogue: residual requests, M_p tiles -> LDS, output transform + fused block epilogue ----
    long long es[3] = {0, 0, 0};
    switch (a.id_mode) {
        case 0: wino_epilogue<0, MO>(a, acc, smem, p, lane, tid, b, nb, r0, j0, TR, TJ, cbx, DBG ? es : nullptr); break;
        case 1:
            if (a.id_split) wino_epilogue<1, MO>(a, acc, smem, p, lane, tid, b, nb, r0, j0, TR, TJ, cbx, DBG ? es : nullptr);
            else wino_epilogue<2, MO>(a, acc, smem, p, lane, tid, b, nb, r0, j0, TR, TJ, cbx, DBG ? es : nullptr);
            break;
        default: wino_epilogue<3, MO>(a, acc, smem, p, lane, tid, b, nb, r0, j0, TR, TJ, cbx, DBG ? es : nullptr); break;
    }
    const long long dbg_e0 = es[0], dbg_e1 = es[1];
    if constexpr (DBG) {                                                // [K loop, prologue wait, epilogue, barrier waits in the loop]
        const long long t_issued = (long long)__builtin_amdgcn_s_memtime();
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        if (a.dbg && lane == 0) {
            const long long t_end = (long long)__builtin_amdgcn_s_memtime();
            long long* d = a.dbg + ((size_t)blockIdx.x * (WCW + WPW) + wave) * 4;
            d[0] = dbg_t1 - dbg_t0; d[1] = dbg_t0 - dbg_entry; d[2] = t_end - dbg_t1; d[3] = dbg_bar;
            // setup | epilogue from its start: M tiles in LDS | barrier passed | loads of all passes arrived, first pass stored | all stores issued | drained
            long long* e = a.dbg + (size_t)(4 << 20) + ((size_t)blockIdx.x * (WCW + WPW) + wave) * 8;
            e[0] = dbg_setup; e[1] = dbg_e0 - dbg_t1; e[2] = dbg_e1 - dbg_t1; e[3] = es[2] - dbg_t1; e[4] = t_issued - dbg_t1; e[5] = t_end - dbg_t1;
        }
    }
}

namespace {
constexpr size_t kWinoLdsLoop = (size_t)(2 * W_VBUF + 2 * W_RAW) * sizeof(float);      // V and staged tiles, double-buffered
constexpr size_t kWinoLds = kWinoLdsEpi > kWinoLdsLoop ? kWinoLdsEpi : kWinoLdsLoop;
static_assert(kWinoLds <= 160 * 1024, "LDS of a gfx950 CU");

// tile-pixel block (rows x tiles) for an Ho x ntile grid: the largest useful fraction of 64-slot blocks, subject to
// the producers' 256 threads (2 per slot) and the slots of a plane
void wino_block(int Ho, int ntile, int KH, int m, int* tr, int* tj) {
    double best = -1;
    for (int r = 1; r <= 64; ++r)
        for (int t = 1; r * t <= 64; ++t) {
            if ((r + KH - 1) * t * 2 > (WPW - 1) * 64 || 64 + (KH - 1) * t > W_NSLOT) continue;   // transform threads, V slots
            if (r + KH - 1 > W_RROWS || t * m + KH - 1 > W_RPX) continue;                           // staged tile
            const int nrb = (Ho + r - 1) / r, ncb = (ntile + t - 1) / t;
            // useful fraction of the MFMA work, discounted by the rows the producers transform per output row
            const double u = (double)Ho * ntile / ((double)nrb * ncb * 64) - 0.02 * (double)(r + KH - 1) / r;
            if (u > best) { best = u; *tr = r; *tj = t; }
        }
}

template <int KH, int MO, int DBG = 0> void launch_wino_t(const ConvArgs& a, hipStream_t s) {
    static unsigned long long attr_devices = 0;
    set_max_dynamic_lds(reinterpret_cast<const void*>(&conv_wino<KH, MO, DBG>), kWinoLds, &attr_devices, "conv_wino");
    const int frames = a.M / (a.Ho * a.Wo);
    const int grid = frames * a.wino_nrb * a.wino_ncb * (a.N / 64);
    NHANS_LAUNCH("conv_wino", (conv_wino<KH, MO, DBG>), dim3(grid), dim3((WCW + WPW) * 64), kWinoLds, s, a);
}
}  // namespace

bool conv_wino_eligible(const ConvArgs& a) { return a.wino && a.wino_u && conv_wino_shape_ok(a); }

// what both Winograd kernels (this one and conv_wino128.hip) can run
bool conv_wino_shape_ok(const ConvArgs& a) {
    const ConvSeg& g = a.seg[0];
    if (!a.wino_ws || a.prec != 1 || a.nseg != 1 || a.kgroup != 0) return false;
    if (g.KH != 4 || g.KW != g.KH || g.sh != 1 || g.sw != 1 || a.Wo != g.W || a.Ho != g.H) return false;   // (KH = 3: odd ring parity, not built yet)
    if (g.C % 16 != 0 || a.N % 64 != 0 || a.Nreal != a.N || !a.out_split || a.aux) return false;
    if (a.id_mode == 1 && !a.id_split && (a.id_ld & 3)) return false;
    if (a.M % (a.Ho * a.Wo) != 0) return false;
    if (a.tf && (!a.tt || !a.ff)) return false;                        // the epilogue reads the table's two terms
    // 32-bit element offsets inside the kernel
    return (double)a.M * std::max(g.C, a.N) + 65536.0 < 2147483648.0;
}

static void wino_geometry(ConvArgs& a) {
    const ConvSeg& g = a.seg[0];
    a.wino_m = 9 - g.KW;                                               // 8 positions: 5 outputs for 4 taps, 6 for 3
    a.wino_ntile = (a.Wo + a.wino_m - 1) / a.wino_m;
    wino_block(a.Ho, a.wino_ntile, g.KH, a.wino_m, &a.wino_tr, &a.wino_tj);
    a.wino_nrb = (a.Ho + a.wino_tr - 1) / a.wino_tr;
    a.wino_ncb = (a.wino_ntile + a.wino_tj - 1) / a.wino_tj;
    a.wino_fd_nnb = make_fastdiv((uint32_t)(a.N / 64));
    a.wino_fd_ncb = make_fastdiv((uint32_t)a.wino_ncb);
    a.wino_fd_bpf = make_fastdiv((uint32_t)(a.wino_nrb * a.wino_ncb * (a.N / 64)));
}

// MFMA FLOPs of the launch: every workgroup multiplies 8 positions x 64 tile-pixel slots (used or not) x 64 channels
// over K = KH * C, three split-f16 products per MAC
double conv_wino_mfma_flops(const ConvArgs& a0) {
    ConvArgs a = a0;
    wino_geometry(a);
    const double wgs = (double)(a.M / (a.Ho * a.Wo)) * a.wino_nrb * a.wino_ncb * (a.N / 64);
    return wgs * 8.0 * 64.0 * 64.0 * (double)(a.seg[0].KH * a.seg[0].C) * 2.0 * 3.0;
}

void launch_conv_wino(const ConvArgs& a0, hipStream_t s) {
    ConvArgs a = a0;
    wino_geometry(a);
    a.ws = a.wino_ws;
#ifdef NHANS_DEV
    // NHANS_ABLATE (timing experiments, wrong results): 1 input tiles of chunks >= 1 from the zero page (no HBM reads),
    // 2 consumers skip the MFMAs, 4 producers transform the first chunk only, 8 residual from one L2-hot line, no stores, 16 every k-step loads the weights of k-step 0 (L1-resident)
    if (a.dbg) { a.wino_m |= (dev_ablate() & 31) << 8; launch_wino_t<4, 5, 1>(a, s); return; }
#endif
    launch_wino_t<4, 5>(a, s);
}

}  // namespace nhans
