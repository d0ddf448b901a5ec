// Stride-1 4 x 4 convolution of the residual stack as a 1-D Winograd convolution along the image width, F(5, 4) with 8
// transformed positions (interpolation points 0, +-1, +-2, +-1/2, infinity), plain accumulation over the filter rows
// and channels:
//
//     V_p[h, j, c]  = sum_x BT[p][x] * in[h, j*5 - pl + x, c]                      input transform (VALU, f32)
//     M_p[ho, j, n] = sum_kh sum_c V_p[ho + kh - pt, j, c] * U_p[kh][c][n]         8 independent GEMMs (MFMA)
//     out[ho, j*5 + i, n] = epilogue( sum_p AT[i][p] * M_p[ho, j, n] )             output transform + block epilogue
//
// with U_p[kh] = sum_kw G[p][kw] w[kh][kw] folded on the host (fold.py: pack_wino).  8*4 products per tile and channel
// pair instead of 5*4*4: 2.5 x fewer MFMAs than the direct form in conv_igemm_halo.hip -- on a chip that runs this
// workload at its socket power cap with 80 % of a tile's energy in the three split-f16 MFMA products per MAC
// (DESIGN.md section 4), MACs are the one thing left to cut.  Accuracy: tests/winograd_probe.py (the transforms run in
// f32 on the hi+lo value, V is re-split; end to end the logits move by < 1e-6 against the direct form).  Why 1-D and
// not F(2x2, k x k): the 2-D forms need 16-36 live accumulator sets per tile, i.e. either tiny tiles (no reuse of U)
// or position groups that go through HBM; nested 1-D keeps one accumulator set per position and wave, K = KH*C long,
// and each V row is reused by all KH filter rows.
//
// Workgroup = EIGHT waves at 256 registers, wave p = position p with M = 128 tile-pixels x 64 channels of accumulators
// (rounds 3's kernel had 8 MFMA waves with 64 tile-pixels each + 4 producer waves; its K loop was bound by the CU's
// vector-memory path: every wave pulled its private transformed weights U_p from L2 for only 64 tile-pixels, 341 bytes
// per MFMA, 128 KB per 16-channel chunk and CU.  Here a weight fragment serves 128 tile-pixels):
//
//   * 128 accumulator registers per wave (4 x 2 MFMA tiles of one position) leave no room for specialised producer
//     waves (twelve waves = 168 registers each): every wave also does an eighth of the input transform and of the
//     tile staging.
//   * The two waves of a SIMD (p, p + 4) run their phases in opposite order -- waves 0-3 multiply chunk c and then
//     transform their share of chunk c + 1, waves 4-7 transform first and multiply afterwards -- so that each SIMD has
//     one wave feeding the matrix pipe and one on the VALU / LDS at any time.  One barrier per chunk joins all eight.
//   * A k-step of the MFMA (K = 16) is 8 channels x TWO filter rows (lanes 0-31 carry row 2s, lanes 32-63 row 2s + 1:
//     the same V plane read at a slot offset of TJ), so a chunk is 8 channels, not 16: V of 150 slots x 8 positions x
//     {hi, lo} is 38 KB, double-buffered 77 KB, and THREE staged input tiles of 25 KB fit beside it (the LDS-DMA round
//     trip of a tile that comes from HBM is as long as a chunk: the DMA leads the transform by two chunks).
//   * The accumulators of a workgroup are 256 KB: the epilogue (conv_wino_common.h) runs in two passes of 64
//     tile-pixels.
//
// Transformed weights: fold.py pack_wino -- [N/64][p 8][C/8][s 2][nt 2][h 2][lane 64][e 8], lane l, element e of
// k-step s of chunk c8 holds filter row 2s + (l >> 5), channel 8 c8 + e, column 64 nb + 32 nt + (l & 31).
//
// Eligibility (launcher): split-f16 mode, one segment, stride 1, SAME padding, 4x4 filters, Cin % 16 == 0,
// N % 64 == 0, split-NHWC input and output.
#include "conv_wino_common.h"

namespace nhans {

namespace {
constexpr int XW = 8;                          // waves = transformed positions
constexpr int X_NSLOT = 152;                   // (row, tile) slots per plane: >= 128 + (KH-1)*TJ
constexpr int X_PLANE = X_NSLOT * 4 + 4;       // floats per plane (16 B = 8 channels of one half per slot)
constexpr int X_VBUF = 16 * X_PLANE;           // floats per V buffer: 8 positions x {hi, lo}
constexpr int X_RROWS = 21, X_RPX = 38;        // staged input tile of a chunk: rows (TR + KH - 1) x pixels (TJ*m + KH - 1)
constexpr int X_NDMA = (2 * X_RROWS * X_RPX + 63) / 64;   // LDS-DMA wave-instructions per staged tile
constexpr int X_DPW = (X_NDMA + XW - 1) / XW;  // ... per wave
constexpr int X_RAW = X_NDMA * 256;            // floats per staged tile
constexpr int X_RAW_BASE = 2 * X_VBUF;         // three staged tiles behind the two V buffers
constexpr int X_DUMP = X_RAW_BASE + 3 * X_RAW; // 1 KB: target of the DMA slots that carry nothing
static_assert(X_NDMA == 25 && X_DPW == 4, "counted waits");
constexpr size_t kWinoLdsLoop = (size_t)(X_DUMP + 256) * sizeof(float);
constexpr size_t kWinoLds = kWinoLdsEpi > kWinoLdsLoop ? kWinoLdsEpi : kWinoLdsLoop;
static_assert(kWinoLds <= 160 * 1024, "LDS of a gfx950 CU");
}  // namespace

// DBG: dev tool, per-wave cycle stamps (tools/wino_phase_cycles.py).  INF: the input tensor is f32 NHWC instead of split
// NHWC (a tensor that only Winograd launches read is stored so: the transform then has no hi + lo to add up -- 64 of its
// ~ 215 instructions per chunk -- and reads its eight pixels as 16-byte pieces; the producing epilogue skips the split).
template <int KH, int MO, int DBG = 0, int INF = 0>
__global__ void __launch_bounds__(XW * 64) conv_wino(const ConvArgs a) {
    static_assert(KH == 4, "a k-step pairs two filter rows");
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
    // (frame, row block, column block, channel block), the channel blocks of one pixel block neighbours
    const int b = (int)fd_div((uint32_t)L, a.wino_fd_bpf);
    const int lb = L - b * (int)a.wino_fd_bpf.d;
    const int lt = (int)fd_div((uint32_t)lb, a.wino_fd_nnb);
    const int nb = lb - lt * (int)a.wino_fd_nnb.d;
    const int rb = (int)fd_div((uint32_t)lt, a.wino_fd_ncb);
    const int cb = lt - rb * (int)a.wino_fd_ncb.d;
    const int TR = a.wino_tr, TJ = a.wino_tj;
    const int r0 = rb * TR, j0 = cb * TJ;
    const ConvSeg& g = a.seg[0];
    const int C = g.C, NC = C >> 3;              // 8-channel chunks (even: C % 16 == 0)
    const int nrows = TR + KH - 1;
    const bool early = wave < XW / 2;            // (uniform) multiplies first, transforms afterwards

    // ---- staging: this wave's X_DPW of the tile's X_NDMA LDS-DMA instructions (instruction i = wave + 8k) ----
    // piece q = 64 i + lane = (kind, row, pixel); byte offset within the frame for chunk 0, or an offset beyond the frame:
    // padding / unused.  The requests go through a buffer descriptor over the frame: a lane whose offset is out of range
    // gets ZEROS written to LDS (tools/ubench/buffer_lds_oob.hip) -- padding needs no zero page and no pointer select,
    // the chunk's offset rides in the scalar offset: one v_cndmask per request (a chunk past the end) where the
    // 64-bit pointer arithmetic and selects were five VALU instructions, in the phase that sets the period.
    constexpr unsigned kOob = 0x80000000u;
    const __amdgpu_buffer_rsrc_t frsrc = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float*>(g.src + (size_t)b * g.H * g.W * C), 0, g.H * g.W * C * 4, 0x00020000);
    unsigned goff[X_DPW];
#pragma unroll
    for (int k = 0; k < X_DPW; ++k) {
        // piece q = (row, pixel, kind): the two pieces of a pixel -- hi | lo halves of a split pixel, 64 B apart in one
        // 128-byte line; the first | second 16 bytes of an f32 pixel's chunk -- sit in neighbouring lanes, so that a wave
        // instruction touches 32 lines instead of the 64 of a (kind, row, pixel) order: the L1's work per staged tile
        // halves at the same bytes (round 5)
        const int q = (wave + XW * k) * 64 + lane;
        const int kind = q & 1, rem = q >> 1;
        const int row = rem / X_RPX, px = rem - row * X_RPX;
        const int hrow = r0 + row - g.pt, wcol = j0 * MO - g.pl + px;
        const bool ok = kind < 2 && row < nrows && px < TJ * MO + KH - 1 && (unsigned)hrow < (unsigned)g.H && (unsigned)wcol < (unsigned)g.W;
        // split NHWC: a 32-channel group of a pixel is 64 B of hi halfs followed by 64 B of lo halfs
        // (f32 NHWC: the chunk's 8 channels are 32 contiguous bytes, kind = their first | second 16)
        goff[k] = ok ? (unsigned)(((hrow * g.W + wcol) * C) * 4 + kind * (INF ? 16 : 64)) : kOob;
    }
    const bool last_real = wave + XW * (X_DPW - 1) < X_NDMA;     // (uniform) the wave's last slot carries pieces
    // tile of chunk CC -> staged buffer RB (a chunk past the end: zeros into the dump area, no memory read)
#define X_DMA2(CC, RB, K0)                                                                         \
    {                                                                                              \
        const int cc_ = (CC);                                                                      \
        const bool live_ = cc_ < NC && !(kDev && (a.wino_m >> 8 & 1) && cc_ > 0);                  \
        const unsigned co_ = INF ? (unsigned)(cc_ * 32) : (unsigned)((cc_ >> 2) * 128 + (cc_ & 3) * 16); \
        float* const dst_ = smem + X_RAW_BASE + (RB) * X_RAW + wave * 256;                         \
        _Pragma("unroll") for (int k = (K0); k < (K0) + 2; ++k) {                                  \
            float* d_ = (cc_ < NC && (k < X_DPW - 1 || last_real)) ? dst_ + k * XW * 256 : smem + X_DUMP; \
            __builtin_amdgcn_raw_ptr_buffer_load_lds(frsrc, (__attribute__((address_space(3))) void*)d_, 16, \
                                                     live_ ? goff[k] : kOob, live_ ? (int)co_ : 0, 0, 0); \
        }                                                                                          \
    }
#define X_DMA(CC, RB) { X_DMA2(CC, RB, 0) X_DMA2(CC, RB, 2) }

    // ---- transform: thread = (slot, 4 channels) = 8 pixels x 4 channels; the tasks are dealt to the waves in equal,
    // contiguous shares so that a wave's lanes write consecutive 8-byte pieces of one plane ----
    const int ntask = nrows * TJ * 2;
    const int tpw = (ntask + XW - 1) / XW;
    const int task = wave * tpw + lane;
    const bool t_active = lane < tpw && task < ntask;
    const int t_slot = t_active ? task >> 1 : 0, t_half = task & 1;
    const int t_rs = t_slot / TJ, t_tj = t_slot - t_rs * TJ;
    // LDS byte addresses (dynamic LDS starts at 0: the kernel has no static LDS)
    // staged tile: [row][pixel][kind][16 bytes]
    const unsigned t_src = INF ? (unsigned)((X_RAW_BASE + ((t_rs * X_RPX + t_tj * MO) * 2 + t_half) * 4) * 4)      // f32: the task's 4 channels are one piece kind
                               : (unsigned)((X_RAW_BASE + (t_rs * X_RPX + t_tj * MO) * 8 + t_half * 2) * 4);   // + RB*X_RAW*4 (+ 16: lo)
    const unsigned t_dst = (unsigned)((t_slot * 4 + t_half * 2) * 4);                                       // + VB*X_VBUF*4 + (p*2 + h)*X_PLANE*4
    // BT, structured (fold.py: WINO_BT): even and odd parts of rows 1..6 share their sums.  ONE channel at a time, in
    // scalar f32 instructions on purpose: beside a running MFMA stream a packed-f32 instruction (v_pk_fma_f32,
    // v_pk_add_f32) costs 22-26 cycles more than the two scalar ones it replaces (MI355X guide, "fillers beside
    // MFMAs"), and this code runs in the shadow of the SIMD partner's multiply phase -- with the packed form the
    // transform advanced at a fifth of its speed there and was as long as the multiply phases themselves.  (The file is
    // compiled with -fno-slp-vectorize: the compiler would pack them again.)
#define X_BT(D, V)                                                                                 \
    {                                                                                              \
        const float e1_ = __builtin_fmaf(-4.25f, D[4], D[2] + D[6]);                               \
        const float o1_ = __builtin_fmaf(-4.25f, D[3], D[1] + D[5]);                               \
        const float e2_ = __builtin_fmaf(0.25f, D[2], __builtin_fmaf(-1.25f, D[4], D[6]));        \
        const float o2_ = __builtin_fmaf(0.5f, D[1], __builtin_fmaf(-2.5f, D[3], 2.f * D[5]));     \
        const float e3_ = __builtin_fmaf(4.f, D[2], __builtin_fmaf(-5.f, D[4], D[6]));             \
        const float o3_ = __builtin_fmaf(2.f, D[1], __builtin_fmaf(-2.5f, D[3], 0.5f * D[5]));     \
        V[0] = __builtin_fmaf(5.25f, D[2] - D[4], D[6] - D[0]);                                    \
        V[1] = e1_ + o1_;  V[2] = e1_ - o1_;                                                       \
        V[3] = e2_ + o2_;  V[4] = e2_ - o2_;                                                       \
        V[5] = e3_ + o3_;  V[6] = e3_ - o3_;                                                       \
        V[7] = __builtin_fmaf(5.25f, D[3] - D[5], D[7] - D[1]);                                    \
    }
    // staged tile RB -> V buffer VB (when DO_T), with the period's requests -- the weights of chunk BC (8) and the
    // staged tile of chunk DC -> buffer DB (4) -- spread through it, two at a time between the stages of the arithmetic:
    // the CU's vector-memory path is what this kernel is bound by (tools/ubench/cu_vmem_rate.hip: 115 GB/s per CU
    // from L2 + 24 GB/s per CU from HBM, additive), a wave that issues a burst of requests stalls until the queue has
    // room (all eight waves' 12 requests at the start of the section: 1,200-2,800 cycles per period in which a wave
    // neither transformed nor multiplied).
    // Every LDS access of the K loop is inline asm with hand-counted waits: the compiler makes ANY ds_read / ds_write
    // it can see wait for all LDS-DMA requests of the same wave that are still in flight (it cannot tell the staged
    // tiles from the V buffers: s_waitcnt vmcnt(0) in front of each) -- and the point of the three staged tiles is
    // that two of them are in flight while the third is read.
    long long dbg_ts[2] = {0, 0}, dbg_tq = 0;
#define X_TSTAMP(I)                                                                                \
    if constexpr (DBG) {                                                                           \
        if (dbg_tq) { const long long t_ = (long long)__builtin_amdgcn_s_memtime(); dbg_ts[I] += t_ - dbg_tq; } \
    }
#define X_TRANSFORM(RB, VB, BC, DC, DB, DO_T, REQ)                                                      \
    {                                                                                              \
        const int bc_ = (BC) < NC ? (BC) : NC - 1;                                                 \
        const char* u_ = ub_base + (size_t)bc_ * (KH / 2) * 4096;                                  \
        const bool do_t_ = (DO_T) && t_active;                                                     \
        const unsigned rs_ = t_src + (unsigned)((RB) * X_RAW * 4);                                 \
        const unsigned vd_ = t_dst + (unsigned)((VB) * X_VBUF * 4);                                \
        f32x2 rh_[8], rl_[8];                                                                      \
        f32x4 rf_[8];                                                                              \
        /* (every lane runs the arithmetic -- idle lanes on slot 0 --, only the stores are predicated: no control flow \
           between the requests) */                                                                \
        if constexpr (INF) {                                                                       \
            asm volatile(                                                                          \
                "ds_read_b128 %0, %8\n ds_read_b128 %1, %8 offset:32\n ds_read_b128 %2, %8 offset:64\n ds_read_b128 %3, %8 offset:96\n" \
                "ds_read_b128 %4, %8 offset:128\n ds_read_b128 %5, %8 offset:160\n ds_read_b128 %6, %8 offset:192\n ds_read_b128 %7, %8 offset:224\n" \
                "s_waitcnt lgkmcnt(0)"                                                             \
                : "=&v"(rf_[0]), "=&v"(rf_[1]), "=&v"(rf_[2]), "=&v"(rf_[3]), "=&v"(rf_[4]), "=&v"(rf_[5]), "=&v"(rf_[6]), "=&v"(rf_[7]) \
                : "v"(rs_) : "memory");                                                            \
        } else {                                                                                   \
            asm volatile(                                                                          \
                "ds_read_b64 %0, %16\n ds_read_b64 %1, %16 offset:32\n ds_read_b64 %2, %16 offset:64\n ds_read_b64 %3, %16 offset:96\n" \
                "ds_read_b64 %4, %16 offset:128\n ds_read_b64 %5, %16 offset:160\n ds_read_b64 %6, %16 offset:192\n ds_read_b64 %7, %16 offset:224\n" \
                "ds_read_b64 %8, %16 offset:%17\n ds_read_b64 %9, %16 offset:%17+32\n ds_read_b64 %10, %16 offset:%17+64\n ds_read_b64 %11, %16 offset:%17+96\n" \
                "ds_read_b64 %12, %16 offset:%17+128\n ds_read_b64 %13, %16 offset:%17+160\n ds_read_b64 %14, %16 offset:%17+192\n ds_read_b64 %15, %16 offset:%17+224\n" \
                "s_waitcnt lgkmcnt(0)"                                                             \
                : "=&v"(rh_[0]), "=&v"(rh_[1]), "=&v"(rh_[2]), "=&v"(rh_[3]), "=&v"(rh_[4]), "=&v"(rh_[5]), "=&v"(rh_[6]), "=&v"(rh_[7]), \
                  "=&v"(rl_[0]), "=&v"(rl_[1]), "=&v"(rl_[2]), "=&v"(rl_[3]), "=&v"(rl_[4]), "=&v"(rl_[5]), "=&v"(rl_[6]), "=&v"(rl_[7]) \
                : "v"(rs_), "n"(16) : "memory");                                                   \
        }                                                                                          \
        X_TSTAMP(0)                                                                                \
        if ((REQ) == 2) X_LOAD_B2(0, 0, u_)                                                        \
        uint2 oh_[8], ol_[8];                                                                      \
        _Pragma("unroll") for (int ep = 0; ep < 2; ++ep) {                                         \
            float d0_[8], d1_[8], v0_[8], v1_[8];                                                  \
            _Pragma("unroll") for (int x = 0; x < 8; ++x) {                                        \
                if constexpr (INF) {                                                               \
                    d0_[x] = rf_[x][2 * ep];                                                       \
                    d1_[x] = rf_[x][2 * ep + 1];                                                   \
                } else {                                                                           \
                    const float hp_ = rh_[x][ep], lp_ = rl_[x][ep];                                \
                    d0_[x] = unsplit_mix<0>(hp_, lp_);                                             \
                    d1_[x] = unsplit_mix<1>(hp_, lp_);                                             \
                }                                                                                  \
            }                                                                                      \
            __builtin_amdgcn_sched_barrier(0);                                                     \
            if (ep == 0) { if ((REQ) == 2) X_LOAD_B2(0, 1, u_) } else if (REQ) { X_DMA2(DC, DB, 0) }  \
            __builtin_amdgcn_sched_barrier(0);                                                     \
            X_BT(d0_, v0_)                                                                         \
            X_BT(d1_, v1_)                                                                         \
            __builtin_amdgcn_sched_barrier(0);                                                     \
            if (REQ) { if (ep == 0) { X_LOAD_B2(1, 0, u_) } else { X_DMA2(DC, DB, 2) } }           \
            __builtin_amdgcn_sched_barrier(0);                                                     \
            _Pragma("unroll") for (int pp = 0; pp < 8; ++pp) {                                     \
                if (ep == 0) split_pair(v0_[pp], v1_[pp], &oh_[pp].x, &ol_[pp].x);                 \
                else split_pair(v0_[pp], v1_[pp], &oh_[pp].y, &ol_[pp].y);                         \
            }                                                                                      \
            __builtin_amdgcn_sched_barrier(0);                                                     \
            if (REQ && ep == 0) { X_LOAD_B2(1, 1, u_) }                                            \
        }                                                                                          \
        X_TSTAMP(1)                                                                                \
        if (do_t_)                                                                                 \
            asm volatile(                                                                          \
                "ds_write_b64 %16, %0\n ds_write_b64 %16, %1 offset:%17\n ds_write_b64 %16, %2 offset:%17*2\n ds_write_b64 %16, %3 offset:%17*3\n" \
                "ds_write_b64 %16, %4 offset:%17*4\n ds_write_b64 %16, %5 offset:%17*5\n ds_write_b64 %16, %6 offset:%17*6\n ds_write_b64 %16, %7 offset:%17*7\n" \
                "ds_write_b64 %16, %8 offset:%17*8\n ds_write_b64 %16, %9 offset:%17*9\n ds_write_b64 %16, %10 offset:%17*10\n ds_write_b64 %16, %11 offset:%17*11\n" \
                "ds_write_b64 %16, %12 offset:%17*12\n ds_write_b64 %16, %13 offset:%17*13\n ds_write_b64 %16, %14 offset:%17*14\n ds_write_b64 %16, %15 offset:%17*15" \
                :: "v"(oh_[0]), "v"(ol_[0]), "v"(oh_[1]), "v"(ol_[1]), "v"(oh_[2]), "v"(ol_[2]), "v"(oh_[3]), "v"(ol_[3]),      \
                   "v"(oh_[4]), "v"(ol_[4]), "v"(oh_[5]), "v"(ol_[5]), "v"(oh_[6]), "v"(ol_[6]), "v"(oh_[7]), "v"(ol_[7]),      \
                   "v"(vd_), "n"(X_PLANE * 4) : "memory");                                         \
    }

    // ---- multiply: position p = this wave ----
    const int p = wave;
    const int cbx = __builtin_amdgcn_readfirstlane((a.img_clip ? a.img_clip[b] : 0) * a.cb_stride);
    const int g8 = lane >> 5;
    // A fragment of k-step s, m-tile t: plane (p, hi) at slot 32 t + (lane & 31) + (2 s + g8) TJ; lo: + X_PLANE
    const int aoff = (p * 2) * X_PLANE + ((lane & 31) + g8 * TJ) * 4;

    f32x16 acc[4][2];
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[t][j][r] = 0.f;

    // the weights of ONE chunk: a wave requests those of its next multiplication right after the previous one -- waves
    // 0-3 at the end of their MFMA phase (chunk c + 1), waves 4-7 at the start of the period (chunk c) -- and transforms
    // while they arrive
    // The loads are asm as well: left to the compiler, whose bookkeeping of what is in flight does not survive this
    // loop's branches, they are followed by s_waitcnt vmcnt(0) in three places.  NOTHING may touch fb between the
    // request and the s_waitcnt that opens the multiply block -- the compiler believes the values are there when the
    // asm statement ends; tools/check_wino_isa.py verifies that on the compiled kernel (run by the test-suite).
    f32x4 fb[KH / 2][2][2];                  // [k-step][n-tile][hi|lo]
    const unsigned ub_lo = (unsigned)(lane * 16), ub_hi = ub_lo + 4096u;     // byte offsets of k-step 0 / 1 within a chunk
    const char* const ub_base = reinterpret_cast<const char*>(a.wino_u + ((size_t)(nb * 8 + p) * NC * (KH / 2)) * 1024);
    // one k-step's fragments of n-tile J: two requests
#define X_LOAD_B2(S, J, UPTR)                                                                      \
    asm volatile("global_load_dwordx4 %0, %2, %3 offset:%4\n global_load_dwordx4 %1, %2, %3 offset:%4+1024" \
                 : "=&v"(fb[S][J][0]), "=&v"(fb[S][J][1]) : "v"((S) ? ub_hi : ub_lo), "s"(UPTR), "n"((J) * 2048) : "memory");
    // the same, skipped (uniformly) when SKIP != 0: the registers keep their values
#define X_LOAD_B2C(S, J, UPTR, SKIP)                                                               \
    asm volatile("s_cmp_eq_u32 %5, 0\n s_cbranch_scc0 1f\n"                                        \
                 "global_load_dwordx4 %0, %2, %3 offset:%4\n global_load_dwordx4 %1, %2, %3 offset:%4+1024\n1:" \
                 : "+v"(fb[S][J][0]), "+v"(fb[S][J][1]) : "v"((S) ? ub_hi : ub_lo), "s"(UPTR), "n"((J) * 2048), "s"((int)(SKIP)) : "memory", "scc");
    // (k-step 1 of chunk CC: k-step 0 is requested inside the multiply block of the chunk before)
#define X_LOAD_B1(CC)                                                                              \
    {                                                                                              \
        const int cc_ = (CC) < NC ? (CC) : NC - 1;                                                 \
        const char* u_ = ub_base + (size_t)cc_ * (KH / 2) * 4096;                                  \
        X_LOAD_B2(1, 0, u_) X_LOAD_B2(1, 1, u_)                                                    \
    }
#define X_LOAD_B1C(CC, SKIP)                                                                       \
    {                                                                                              \
        const int cc_ = (CC) < NC ? (CC) : NC - 1;                                                 \
        const char* u_ = ub_base + (size_t)cc_ * (KH / 2) * 4096;                                  \
        X_LOAD_B2C(1, 0, u_, SKIP) X_LOAD_B2C(1, 1, u_, SKIP)                                      \
    }
    // The chunk's 8 steps (k-step s, m-tile t) of 6 MFMAs each as ONE asm block (see X_TRANSFORM for why): the V
    // fragments of step i + 2 are requested behind the first MFMA of step i (ring of three) -- one wave per SIMD
    // multiplies at a time, so its own MFMAs must cover its LDS latency --; an accumulator tile is used by every other
    // MFMA (a dependent MFMA issued back to back would wait for the first one's 8 passes).  The four weight fragments
    // of k-step 0 are dead after step 3: the NEXT chunk's are requested into them behind the second MFMA of steps 4-7
    // (a third of the period's requests issued in the shadow of the wave's own MFMAs instead of in the burst behind
    // them).
    // operands: 0-7 acc[t][j]; 8-13 ring r = (hi %8+2r, lo %9+2r); 14-21 fb[s][j][h]; 22 / 23 LDS address of k-step
    // 0 / 1; 24 plane stride; 25 / 26 lane offset / base of the next chunk's weights
    const unsigned a_addr = (unsigned)(aoff * 4);
#define X_MF(ACC, B, A) "v_mfma_f32_32x32x16_f16 %" #ACC ", %" #B ", %" #A ", %" #ACC "\n"
    // step: wait for the ring slot, first MFMA, request the fragments of step i + 2, second MFMA, (a weight request,) four MFMAs
#define X_STEP(WAIT, A0, A1, HI, LO, B0H, B0L, B1H, B1L, NEXT, NEXTB)                              \
    "s_waitcnt lgkmcnt(" #WAIT ")\n" X_MF(A0, B0H, LO) NEXT X_MF(A1, B1H, LO) NEXTB X_MF(A0, B0L, HI) X_MF(A1, B1L, HI) X_MF(A0, B0H, HI) X_MF(A1, B1H, HI)
#define X_MULTIPLY(VB, CN, LAST)                                                                   \
    if (!(DBG && (a.wino_m >> 8 & 2))) {                                                           \
        f32x4 r0h, r0l, r1h, r1l, r2h, r2l;                                                        \
        const unsigned a0_ = a_addr + (unsigned)((VB) * X_VBUF * 4), a1_ = a0_ + (unsigned)(2 * TJ * 16); \
        const int cn_ = (CN) < NC ? (CN) : NC - 1;                                                 \
        const char* un_ = ub_base + (size_t)cn_ * (KH / 2) * 4096;                                 \
        asm volatile(                                                                              \
            "s_cmp_eq_u32 %27, 0\n"                 /* scc = "another chunk follows": its k-step-0 weights are requested below */ \
            "s_waitcnt vmcnt(4)\n"                  /* the weights (behind them: one tile's 4 DMA requests) */ \
            "ds_read_b128 %8, %22\n ds_read_b128 %9, %22 offset:%24\n"                             \
            "ds_read_b128 %10, %22 offset:512\n ds_read_b128 %11, %22 offset:512+%24\n"            \
            X_STEP(2, 0, 1, 8, 9, 14, 15, 16, 17, "ds_read_b128 %12, %22 offset:1024\n ds_read_b128 %13, %22 offset:1024+%24\n", "") \
            X_STEP(2, 2, 3, 10, 11, 14, 15, 16, 17, "ds_read_b128 %8, %22 offset:1536\n ds_read_b128 %9, %22 offset:1536+%24\n", "") \
            X_STEP(2, 4, 5, 12, 13, 14, 15, 16, 17, "ds_read_b128 %10, %23\n ds_read_b128 %11, %23 offset:%24\n", "") \
            X_STEP(2, 6, 7, 8, 9, 14, 15, 16, 17, "ds_read_b128 %12, %23 offset:512\n ds_read_b128 %13, %23 offset:512+%24\n", "") \
            X_STEP(2, 0, 1, 10, 11, 18, 19, 20, 21, "ds_read_b128 %8, %23 offset:1024\n ds_read_b128 %9, %23 offset:1024+%24\n", "s_cbranch_scc0 1f\n global_load_dwordx4 %14, %25, %26\n1:\n") \
            X_STEP(2, 2, 3, 12, 13, 18, 19, 20, 21, "ds_read_b128 %10, %23 offset:1536\n ds_read_b128 %11, %23 offset:1536+%24\n", "s_cbranch_scc0 1f\n global_load_dwordx4 %15, %25, %26 offset:1024\n1:\n") \
            X_STEP(2, 4, 5, 8, 9, 18, 19, 20, 21, "", "s_cbranch_scc0 1f\n global_load_dwordx4 %16, %25, %26 offset:2048\n1:\n") \
            X_STEP(0, 6, 7, 10, 11, 18, 19, 20, 21, "", "s_cbranch_scc0 1f\n global_load_dwordx4 %17, %25, %26 offset:3072\n1:\n") \
            : "+v"(acc[0][0]), "+v"(acc[0][1]), "+v"(acc[1][0]), "+v"(acc[1][1]), "+v"(acc[2][0]), "+v"(acc[2][1]), "+v"(acc[3][0]), "+v"(acc[3][1]), \
              "=&v"(r0h), "=&v"(r0l), "=&v"(r1h), "=&v"(r1l), "=&v"(r2h), "=&v"(r2l),              \
              "+v"(fb[0][0][0]), "+v"(fb[0][0][1]), "+v"(fb[0][1][0]), "+v"(fb[0][1][1])           \
            : "v"(fb[1][0][0]), "v"(fb[1][0][1]), "v"(fb[1][1][0]), "v"(fb[1][1][1]),              \
              "v"(a0_), "v"(a1_), "n"(X_PLANE * 4), "v"(ub_lo), "s"(un_), "s"((int)(LAST)) : "memory", "scc"); \
    }

    // ---- prologue: ONLY tile 0 is requested at entry -- everybody waits for it, and with tiles 1, 2 and the weights
    // requested beside it (139 KB per CU in one burst) it landed after 10.5 k cycles: the CU's memory path serves
    // requests in order --; the weights of chunk 0 and tile 1 follow from inside the transform of chunk 0 (all eight
    // waves), tile 2 behind it ----
    long long dbg_pre = 0, dbg_mid = 0;
    if constexpr (DBG) dbg_pre = (long long)__builtin_amdgcn_s_memtime() - dbg_entry;
    if constexpr (DBG) {
        X_DMA2(0, 0, 0)
        dbg_mid = (long long)__builtin_amdgcn_s_memtime() - dbg_entry;
        X_DMA2(0, 0, 2)
    } else {
        X_DMA(0, 0)
    }
    long long dbg_setup = 0;
    if constexpr (DBG) dbg_setup = (long long)__builtin_amdgcn_s_memtime() - dbg_entry;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    long long dbg_landed = 0;
    if constexpr (DBG) dbg_landed = (long long)__builtin_amdgcn_s_memtime() - dbg_entry;
    X_TRANSFORM(0, 0, 0, 1, 1, true, 2)
    __builtin_amdgcn_sched_barrier(0);
    X_DMA(2, 2)
    asm volatile("s_waitcnt vmcnt(4) lgkmcnt(0)" ::: "memory");         // tile 1 (in flight: weights 0 (8) | tile 1 (4) | tile 2 (4))
    __builtin_amdgcn_s_barrier();

    // ---- K loop.  Period c (between two barriers): every wave multiplies chunk c, transforms its share of chunk
    // c + 1 (from the tile that landed a period ago) and requests the weights of chunk c + 1 and the tile of chunk
    // c + 3.  In flight at the end of a period, in order: tile c+2 (4) | weights (8) | tile c+3 (4): vmcnt(12) says
    // tile c+2 has landed. ----
    long long dbg_t0 = 0, dbg_bar = 0, dbg_t1 = 0, dbg_mult = 0, dbg_xf = 0, dbg_wait = 0;
    int dbg_pv = 0;                                // lane c: the low word of the time at the end of period c (v_writelane: no memory traffic in the loop)
    long long dbg_rt0 = 0;
    if constexpr (DBG) dbg_rt0 = (long long)__builtin_amdgcn_s_memrealtime();
    if constexpr (DBG) dbg_t0 = (long long)__builtin_amdgcn_s_memtime();
    int rb3 = 0;                                   // c % 3
#define X_STAMP(ACCUM)                                                                             \
    if constexpr (DBG) {                                                                           \
        const long long t_ = (long long)__builtin_amdgcn_s_memtime();                              \
        ACCUM += t_ - tq_;                                                                         \
        tq_ = t_;                                                                                  \
    }
    // The waves that multiply second are the chain that sets the period (their multiply took 2,460
    // cycles against the others' 2,050 beside a transform at priority 2): their multiply runs at priority 3 (- 1.1 %; the
    // transforms at 1 / 3 instead of 2 / 2 on top of it: + 1.2 %).
#define PRIO_LATE_ON __builtin_amdgcn_s_setprio(3);
#define PRIO_LATE_OFF __builtin_amdgcn_s_setprio(0);
#define X_PERIOD(C_, VB)                                                                           \
    {                                                                                              \
        const int c_ = (C_);                                                                       \
        const int rn_ = rb3 == 2 ? 0 : rb3 + 1;                /* (c + 1) % 3 */                    \
        const bool more_ = c_ + 1 < NC && !(DBG && (a.wino_m >> 8 & 4));                          \
        long long tq_ = 0;                                                                         \
        if constexpr (DBG) tq_ = (long long)__builtin_amdgcn_s_memtime();                          \
        /* the last period requests nothing and transforms nothing: the waves that multiply second had sent twelve  */ \
        /* dummy requests just before the loop's end, and the wait for them -- a memory latency under load, 6-7 k   */ \
        /* cycles -- stood between the loop and the epilogue while the other four waited at the epilogue's barrier */ \
        int last_;                                  /* (defined in a scalar register by construction: the "s" constraint alone does not move a value there) */ \
        asm volatile("s_cmp_lg_u32 %1, 0\n s_cselect_b32 %0, 0, 1" : "=s"(last_) : "s"(__builtin_amdgcn_readfirstlane((int)more_)) : "scc"); \
        if (early) X_MULTIPLY(VB, c_ + 1, last_)                                                   \
        X_STAMP(dbg_mult)                                                                          \
        __builtin_amdgcn_sched_barrier(0);                                                         \
        /* The waves that multiply second request the tile of chunk c + 3 HERE, at the start of the period (its staged  */ \
        /* buffer was read by the transform of the period before), not behind their multiply at the period's end: the  */ \
        /* request is then a period and a half old when the transform of chunk c + 3 needs it -- issued at the end, it */ \
        /* was one period old against an HBM latency of about that under load, and these waves, whose chain is the     */ \
        /* period, waited ~ 700 cycles for it at every period's end.  (Order in the queue: tile c+3 | weights: the     */ \
        /* counts of the two waits do not change.)                                                                    */ \
        if (!early && more_) X_DMA(c_ + 3, rb3)                                                    \
        if (more_) {                                                                               \
            __builtin_amdgcn_s_setprio(2);                                                         \
            if constexpr (DBG) dbg_tq = (long long)__builtin_amdgcn_s_memtime();                   \
            X_TRANSFORM(rn_, VB ^ 1, c_ + 1, c_ + 3, rb3, true, (early ? 1 : 0))                   \
            __builtin_amdgcn_s_setprio(0);                                                         \
        }                                                                                          \
        X_STAMP(dbg_xf)                                                                            \
        __builtin_amdgcn_sched_barrier(0);                                                         \
        if (!early) {                                                                              \
            /* (the multiply block's vmcnt(4) stands for "the weights, but not the tile request behind them"; in the   */ \
            /* last period these waves have sent no tile request, the four youngest are weights)                      */ \
            if (!more_) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                           \
            PRIO_LATE_ON                                                                           \
            X_MULTIPLY(VB, c_ + 1, last_)                                                          \
            PRIO_LATE_OFF                                                                          \
            X_STAMP(dbg_mult)                                                                      \
            X_LOAD_B1C(c_ + 1, last_)                                                              \
        }                                                                                          \
        asm volatile("s_waitcnt vmcnt(12) lgkmcnt(0)" ::: "memory");                               \
        X_STAMP(dbg_wait)                                                                          \
        __builtin_amdgcn_s_barrier();                                                              \
        X_STAMP(dbg_bar)                                                                           \
        if constexpr (DBG) { int keep_; asm volatile("s_mov_b32 %1, m0\n s_mov_b32 m0, %3\n s_nop 0\n v_writelane_b32 %0, %2, m0\n s_mov_b32 m0, %1" : "+v"(dbg_pv), "=&s"(keep_) : "s"(__builtin_amdgcn_readfirstlane((int)(tq_ - dbg_t0))), "s"(__builtin_amdgcn_readfirstlane(c_))); } \
        rb3 = rn_;                                                                                 \
    }
#pragma unroll 1
    for (int c = 0; c < NC; c += 2) {
        X_PERIOD(c, 0)
        X_PERIOD(c + 1, 1)
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                    // (period NC - 2's dummy tile requests: nothing may land in LDS later)
    // (the weight registers stay live up to here: a request of the last periods that the multiply blocks did not consume
    // must not find its registers reused -- conv_wino_common.h, NH_LANDED)
    NH_LANDED4(fb[0][0][0], fb[0][0][1], fb[0][1][0], fb[0][1][1]);
    NH_LANDED4(fb[1][0][0], fb[1][0][1], fb[1][1][0], fb[1][1][1]);
    if constexpr (DBG) dbg_t1 = (long long)__builtin_amdgcn_s_memtime();
#undef X_PERIOD
#undef PRIO_LATE_ON
#undef PRIO_LATE_OFF
#undef X_STAMP
#undef X_TSTAMP
#undef X_LOAD_B2
#undef X_MULTIPLY
#undef X_STEP
#undef X_MF
#undef X_LOAD_B1
#undef X_LOAD_B1C
#undef X_LOAD_B2C
#undef X_TRANSFORM
#undef X_BT
#undef X_DMA
#undef X_DMA2

    // ---- epilogue in two passes of 64 tile-pixels (the eight M_p tiles of a pass are 136 KB of LDS) ----
    long long es[16] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
#define X_EPI2(IDM, OUTS)                                                                          \
    {                                                                                              \
        wino_epilogue<IDM, MO, OUTS>(a, acc, smem, p, lane, tid, b, nb, r0, j0, TR, TJ, cbx, DBG ? es : nullptr, 0, true); \
        __builtin_amdgcn_s_barrier();                                                              \
        wino_epilogue<IDM, MO, OUTS>(a, acc + 2, smem, p, lane, tid, b, nb, r0, j0, TR, TJ, cbx, DBG ? es + 8 : nullptr, 64, false); \
    }
#define X_EPI(IDM) { if (a.out_split) X_EPI2(IDM, 1) else X_EPI2(IDM, 0) }
    switch (a.id_mode) {
        case 0: X_EPI(0) break;
        case 1:
            // (a split residual with an f32 output is refused by the launcher: not instantiated)
            if (a.id_split) X_EPI2(1, 1) else X_EPI(2)
            break;
        default: X_EPI(3) break;
    }
#undef X_EPI
#undef X_EPI2
    if constexpr (DBG) {                                                // [K loop, prologue, epilogue, barrier waits in the loop]
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        if (a.dbg && lane == 0) {
            const long long t_end = (long long)__builtin_amdgcn_s_memtime();
            long long* d = a.dbg + ((size_t)blockIdx.x * 12 + wave) * 4;
            d[0] = dbg_t1 - dbg_t0; d[1] = dbg_t0 - dbg_entry; d[2] = t_end - dbg_t1; d[3] = dbg_bar;
            long long* e = a.dbg + (size_t)(4 << 20) + ((size_t)blockIdx.x * 12 + wave) * 8;
            // setup | first tile landed | epilogue pass 0: M tiles in LDS, barrier passed, transform done (from the loop's end) | multiply | request + transform | wait for the tile
            e[0] = dbg_setup; e[1] = dbg_landed; e[2] = es[0] - dbg_t1; e[3] = es[1] - dbg_t1; e[4] = es[2] - dbg_t1; e[5] = dbg_mult; e[6] = dbg_xf; e[7] = dbg_wait;
            long long* f = a.dbg + (size_t)(6 << 20) + ((size_t)blockIdx.x * 12 + wave) * 4;   // transform: reads landed | arithmetic done (from its start) | entry -> first request
            f[0] = dbg_ts[0]; f[1] = dbg_ts[1]; f[2] = dbg_pre; f[3] = dbg_mid;
            if (blockIdx.x < 4096) {
                // epilogue, both passes, from the loop's end (es[0..5], es[8..13]: conv_wino_common.h) | [16] drained | [17] 100 MHz ticks and
                // [18] cycles from the K loop's start to here | [19] the launch's grid (every launch stamps the same buffer: the reader
                // keeps the records of the LAST launch's grid only)
                long long* h = a.dbg + (size_t)(8 << 20) + ((size_t)blockIdx.x * 12 + wave) * 24;
                for (int i = 0; i < 16; ++i) h[i] = es[i] ? es[i] - dbg_t1 : 0;
                h[16] = t_end - dbg_t1;
                h[17] = (long long)__builtin_amdgcn_s_memrealtime() - dbg_rt0;
                h[18] = t_end - dbg_t0;
                h[19] = gridDim.x;
            }
        }
        if (a.dbg && lane < 32 && blockIdx.x < 4096) a.dbg[(size_t)(10 << 20) + ((size_t)blockIdx.x * 12 + wave) * 32 + lane] = dbg_pv;
    }
}

namespace {
// tile-pixel block (rows x tiles, <= 128) for an Ho x ntile grid: the largest useful fraction of the 128-slot blocks,
// subject to the staged tile, the slots of a V plane and one transform task per thread
void wino_block(int Ho, int ntile, int KH, int m, int* tr, int* tj) {
    double best = -1;
    for (int r = 1; r <= 128; ++r)
        for (int t = 1; r * t <= 128; ++t) {
            if (r + KH - 1 > X_RROWS || t * m + KH - 1 > X_RPX) continue;                            // staged tile
            if (128 + (KH - 1) * t > X_NSLOT || (r + KH - 1) * t * 2 > XW * 64) continue;            // V slots, transform tasks
            const int nrb = (Ho + r - 1) / r, ncb = (ntile + t - 1) / t;
            // useful fraction of the MFMA work, discounted by the rows transformed per output row
            const double u = (double)Ho * ntile / ((double)nrb * ncb * 128) - 0.02 * (double)(r + KH - 1) / r;
            if (u > best) { best = u; *tr = r; *tj = t; }
        }
}

template <int KH, int MO, int DBG = 0, int INF = 0> void launch_wino_t(const ConvArgs& a, hipStream_t s) {
    static unsigned long long attr_devices = 0;
    set_max_dynamic_lds(reinterpret_cast<const void*>(&conv_wino<KH, MO, DBG, INF>), kWinoLds, &attr_devices, "conv_wino");
    const int frames = a.M / (a.Ho * a.Wo);
    const int grid = frames * a.wino_nrb * a.wino_ncb * (a.N / 64);
    NHANS_LAUNCH("conv_wino", (conv_wino<KH, MO, DBG, INF>), dim3(grid), dim3(XW * 64), kWinoLds, s, a);
}

void wino_geometry(ConvArgs& a) {
    const ConvSeg& g = a.seg[0];
    a.wino_m = 9 - g.KW;                                               // 8 positions: 5 outputs for 4 taps
    a.wino_ntile = (a.Wo + a.wino_m - 1) / a.wino_m;
    wino_block(a.Ho, a.wino_ntile, g.KH, a.wino_m, &a.wino_tr, &a.wino_tj);
    a.wino_nrb = (a.Ho + a.wino_tr - 1) / a.wino_tr;
    a.wino_ncb = (a.wino_ntile + a.wino_tj - 1) / a.wino_tj;
    a.wino_fd_nnb = make_fastdiv((uint32_t)(a.N / 64));
    a.wino_fd_ncb = make_fastdiv((uint32_t)a.wino_ncb);
    a.wino_fd_bpf = make_fastdiv((uint32_t)(a.wino_nrb * a.wino_ncb * (a.N / 64)));
}
}  // namespace

// INVARIANT (round-5 advisor): this predicate must NOT read a tensor-LAYOUT field -- in_f32, out_split, id_split,
// sat_limit.  nhans_api.hip (run_stack_chunk) asks it in a planning pass in which those fields are not final yet (block
// b is planned before block b + 1's readers are known) and derives the layouts from the answers; an answer that
// depended on a layout would turn every call into the "eligibility changed between planning and launch" refusal.  A
// layout combination the kernel does not implement is refused by launch_conv_wino() instead.
bool conv_wino_eligible(const ConvArgs& a) {
    const ConvSeg& g = a.seg[0];
    if (!a.wino || !a.wino_u || !a.wino_ws || a.prec != 1 || a.nseg != 1 || a.kgroup != 0) return false;
    if (g.KH != 4 || g.KW != g.KH || g.sh != 1 || g.sw != 1 || a.Wo != g.W || a.Ho != g.H) return false;   // (3x3: F(6,3) not built)
    if (g.C % 16 != 0 || a.N % 64 != 0 || a.Nreal != a.N || a.aux) return false;            // (out: split or f32 NHWC)
    if (a.id_mode == 1 && (a.id_ld & 3)) return false;                 // (16-byte residual pieces; split tensors have C % 32 == 0 anyway)
    if (a.M % (a.Ho * a.Wo) != 0) return false;
    if (a.tf && (!a.tt || !a.ff)) return false;                        // the epilogue reads the table's two terms
    // 32-bit element offsets inside the kernel
    return (double)a.M * std::max(g.C, a.N) + 65536.0 < 2147483648.0;
}

// MFMA FLOPs of the launch: every workgroup multiplies 8 positions x 128 tile-pixel slots (used or not) x 64
// channels over K = KH * C, three split-f16 products per MAC
double conv_wino_mfma_flops(const ConvArgs& a0) {
    ConvArgs a = a0;
    wino_geometry(a);
    const double wgs = (double)(a.M / (a.Ho * a.Wo)) * a.wino_nrb * a.wino_ncb * (a.N / 64);
    return wgs * 8.0 * 128.0 * 64.0 * (double)(a.seg[0].KH * a.seg[0].C) * 2.0 * 3.0;
}

void launch_conv_wino(const ConvArgs& a0, hipStream_t s) {
    // A split-NHWC residual with an f32-NHWC output: the f32 channel assignment of the epilogue (a thread owns channels
    // 4 c8 .. + 3 and 32 + 4 c8 .. + 3, conv_wino_common.h) fetches its residual as two 16-byte pieces of THOSE channels,
    // which a split pixel does not hold contiguously (round-5 advisor: the instantiation existed and added channels
    // 4 c8 + 4 .. + 7 to outputs 32 + 4 c8 .. + 3).  No plan of nhans_api.hip produces the pair -- an identity block's
    // input and output are stored alike or the output is split --; should one ever, it is refused, not misread.
    if (a0.id_mode == 1 && a0.id_split && !a0.out_split) {
        note_refusal("conv_wino (split residual with an f32-stored output)");
        return;
    }
    ConvArgs a = a0;
    wino_geometry(a);
    a.ws = a.wino_ws;
#ifdef NHANS_DEV
    // NHANS_ABLATE (timing experiments, wrong results): 1 input tiles of chunks >= 1 from the zero page, 2 no MFMAs,
    // 4 no transforms after the first chunk, 8 residual from one L2-hot line and no stores
    if (a.dbg) {
        a.wino_m |= (dev_ablate() & 31) << 8;
        if (a.in_f32) launch_wino_t<4, 5, 1, 1>(a, s); else launch_wino_t<4, 5, 1>(a, s);
        return;
    }
#endif
#ifdef NHANS_WINO_F32IN_TIMING
    launch_wino_t<4, 5, 0, 1>(a, s);       // (timing experiment, wrong results: every input read as if it were f32)
    return;
#endif
    if (a.in_f32) launch_wino_t<4, 5, 0, 1>(a, s);
    else launch_wino_t<4, 5>(a, s);
}

}  // namespace nhans
