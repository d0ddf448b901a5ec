// Fused N-HANS block epilogue shared by the implicit-GEMM convolution kernels (contract in
// conv_igemm.hip / nhans_kernels.h):
//     v   = acc * ws[n] + cb[clip(b), n] + tf[(ho, wo), n]              (ws: split-f16 mode only)
//     aux = v                                                            (optional pre-residual tap)
//     v  += idw[n] * residual                                            (tensor / 1-channel image)
//     out = relu ? max(v, 0) : v      as f32 NHWC or split NHWC (hi/lo halfs)
// Split tensors are stored times 2^-e (ConvArgs::in_scale / id_scale / out_scale): ws and idw take the input's and
// the residual's 2^e, `out` is multiplied by the output's 2^-e after the ReLU -- exact, so v has the bits of e = 0.
//
// The accumulator tile goes through LDS once so that the global side is fully coalesced: the K loop
// leaves each lane with 4 consecutive channels of ONE pixel per register quad (the weights are the
// row operand of the MFMAs), which is ideal for a 16-byte LDS write but would scatter 8-16 byte
// pieces over 32 pixels per global store instruction -- measured as 30 % of the kernel time.  After
// the transpose a wavefront owns whole pixel rows: 32 consecutive lanes cover 128 consecutive
// channels, every table / residual read and every output write moves full 128-byte lines.
#pragma once
#include "nhans_kernels.h"
#include <cstddef>

namespace nhans {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));   // native vector: stays in registers
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));

// Scheduling pattern of one half of a K-loop step: ND operand reads (ds_read_b128) spread evenly between
// NM MFMAs (sched_group_barrier masks: 0x008 MFMA, 0x100 DS read).  A wave's LDS reads issued in the
// shadow of its own 32-cycle MFMAs cost the matrix pipe nothing; a block of reads in front of a block of
// MFMAs makes all waves hammer the LDS while the pipes idle and then the reverse.
template <int I, int NM, int ND> __device__ __forceinline__ void pin_reads_between_mfmas() {
    if constexpr (I < ND) {
        __builtin_amdgcn_sched_group_barrier(0x008, (I + 1) * NM / ND - I * NM / ND, 0);
        __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
        pin_reads_between_mfmas<I + 1, NM, ND>();
    }
}

// The same reads FRONT-LOADED: one read after each of the first ND MFMAs of the half, then NM - ND MFMAs with nothing
// behind them -- the last reads' LDS latency is covered by those MFMAs instead of showing at the lgkmcnt(0) in front of
// the barrier between the halves.  What conv_igemm_halo.hip uses (+1 % wall time over the even spread, which stays
// selectable); the kernels without a barrier between the halves measure the same or slightly worse with it and keep
// the even spread; two reads behind each of the first four MFMAs is worse (-2.6 %).
template <int I, int NM, int ND> __device__ __forceinline__ void pin_reads_front_loaded() {
    if constexpr (I < ND) {
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
        __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
        pin_reads_front_loaded<I + 1, NM, ND>();
    } else if constexpr (NM > ND) {
        __builtin_amdgcn_sched_group_barrier(0x008, NM - ND, 0);
    }
}

// The epilogue's arithmetic, spelled out so that every kernel variant rounds alike (whether the compiler
// contracts a*b+c into an fma depends on the surrounding code; a layer must give the same bits whichever
// variant its launch size selects):  v = fma(acc, ws, cb) + tf;  v = fma(idw, residual, v).
__device__ __forceinline__ f32x4 fma4(f32x4 a, f32x4 b, f32x4 c) {
    return f32x4{__builtin_fmaf(a.x, b.x, c.x), __builtin_fmaf(a.y, b.y, c.y), __builtin_fmaf(a.z, b.z, c.z),
                 __builtin_fmaf(a.w, b.w, c.w)};
}
__device__ __forceinline__ f32x4 epi_combine(f32x4 acc, f32x4 ws, f32x4 cb, f32x4 tf, f32x4 idw, f32x4 res) {
    return fma4(idw, res, fma4(acc, ws, cb) + tf);
}

// split of two f32 values into packed f16 pairs: hi = RNE(v) (one v_cvt_pk_f16_f32), lo = RNE(v - hi) as one
// v_fma_mixlo_f16 / v_fma_mixhi_f16 each (f16 source, f32 addend, f16 result into one half of the destination)
__device__ __forceinline__ void split_pair(float vx, float vy, unsigned* hi, unsigned* lo) {
    typedef _Float16 h2_t __attribute__((ext_vector_type(2)));
    const h2_t h = {(_Float16)vx, (_Float16)vy};
    const unsigned hb = __builtin_bit_cast(unsigned, h);
    unsigned l;
    asm("v_fma_mixlo_f16 %0, %1, -1.0, %2 op_sel:[0,0,0] op_sel_hi:[1,0,0]" : "=v"(l) : "v"(hb), "v"(vx));
    asm("v_fma_mixhi_f16 %0, %1, -1.0, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "+v"(l) : "v"(hb), "v"(vy));
    *hi = hb;
    *lo = l;
}

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

__device__ __forceinline__ float split_load(const float* base, size_t row_floats, int n) {
    // value (hi + lo) of channel n of a split-NHWC pixel whose line starts at base + row_floats
    const _Float16* p = reinterpret_cast<const _Float16*>(base + row_floats) + (n >> 5) * 64 + (n & 31);
    return (float)p[0] + (float)p[32];
}

// A field of the kernel's ConvArgs (always the first kernel argument) read from the kernarg segment with a scalar load
// at the point of use.  The compiler otherwise parks the fields an epilogue needs (scales, saturation limit and flag
// pointer) in SCRATCH at kernel entry -- the K loops leave it no registers -- and reloads them with scratch_load, which
// counts in vmcnt like every vector load: such a reload waits for everything in flight (s_waitcnt vmcnt(0)).
template <typename T> __device__ __forceinline__ T conv_karg(size_t offset) {
    typedef const __attribute__((address_space(4))) char* kptr;
    return *reinterpret_cast<const __attribute__((address_space(4))) T*>((kptr)__builtin_amdgcn_kernarg_segment_ptr() + offset);
}
#define CONV_KARG(FIELD) ::nhans::conv_karg<decltype(::nhans::ConvArgs::FIELD)>(offsetof(::nhans::ConvArgs, FIELD))

// LDS bytes the epilogue needs for a BMROWS x BNCOLS tile (accumulator tile padded by 4 floats per
// row against bank conflicts + one int4 of row info per pixel)
template <int BMROWS, int BNCOLS> constexpr size_t conv_epilogue_lds_bytes() {
    return (size_t)BMROWS * (BNCOLS + 4) * sizeof(float) + (size_t)BMROWS * sizeof(int4);
}

// Tile-local pixel p -> global output pixel m (or -1 for a slot outside the tensor).
struct EpiTile {
    int m0;                 // linear tiles (tw == 0): pixel m0 + p
    int tw, th;             // 2-D tiles: p = r*tw + c (r < th) is row r0+r, column c0+c of image b
    int b, r0, c0;
    __device__ __forceinline__ int operator()(int p, const ConvArgs& a) const {
        if (tw <= 0) {
            const int m = m0 + p;
            return m < a.M ? m : -1;
        }
        const int r = p / tw, c = p - r * tw;
        if (r >= th || r0 + r >= a.Ho || c0 + c >= a.Wo) return -1;
        return (b * a.Ho + r0 + r) * a.Wo + c0 + c;
    }
};

// Row-major sweep of the epilogue for one thread = (pixel row within a pass, 4 channels n..n+3).
// IDM: 0 none, 1 split-NHWC tensor, 2 f32 tensor, 3 one-channel image; OUTS: split-NHWC output.
// Passes are processed in groups: every load of a group is issued before its first store, so a group
// costs one memory round trip (the residual may alias the output -- identity blocks are written in
// place -- which would otherwise force the compiler to finish pass p before starting pass p+1).
template <int PREC, int IDM, int OUTS, int PP, int PASSES, int LDC>
__device__ __forceinline__ void conv_epilogue_sweep(const ConvArgs& a, const float* ct, const int4* rowinfo,
                                                    int prow, int c4, int n) {
    constexpr int GP = PASSES < 8 ? PASSES : 8;
    static_assert(PASSES % GP == 0, "pass grouping");
    const int f_tf = a.tf ? 1 : 0;
    const float* __restrict__ cbp = a.cb;
    const float* __restrict__ tfp = a.tf ? a.tf : a.zero;     // an absent table reads the zero page
    f32x4 wsv = {1.f, 1.f, 1.f, 1.f}, idwv = {0.f, 0.f, 0.f, 0.f};
    if constexpr (PREC == 1) wsv = *reinterpret_cast<const f32x4*>(a.ws + n) * CONV_KARG(in_scale);
    if constexpr (IDM != 0) idwv = *reinterpret_cast<const f32x4*>(a.idw + n) * CONV_KARG(id_scale);
    const int hoff = (n >> 5) * 64 + (n & 31);                // half index inside a split-NHWC pixel
    const float lo_clamp = a.relu ? 0.f : -3.0e38f;           // branch-free ReLU
    const float osc = CONV_KARG(out_scale), slim = CONV_KARG(sat_limit);
    bool sat = false;                                         // split output: a value that does not fit f16
#pragma unroll 1
    for (int pg = 0; pg < PASSES; pg += GP) {
        f32x4 yv[GP];
#pragma unroll
        for (int u = 0; u < GP; ++u) {
            const int p = (pg + u) * PP + prow;
            const int4 ri = rowinfo[p];                       // (clip*cb_stride, (ho*Wo+wo)*N, m or -1, ids)
            const int mc = ri.z < 0 ? 0 : ri.z;
            const f32x4 av = *reinterpret_cast<const f32x4*>(ct + p * LDC + c4 * 4);
            const f32x4 c = *reinterpret_cast<const f32x4*>(cbp + ri.x + n);
            const f32x4 t = *reinterpret_cast<const f32x4*>(tfp + (ri.y + n) * f_tf);
            f32x4 idv = {0.f, 0.f, 0.f, 0.f};
            if constexpr (IDM == 1) {
                const _Float16* hp = reinterpret_cast<const _Float16*>(a.id + (size_t)mc * a.id_ld) + hoff;
                const f16x4 h = *reinterpret_cast<const f16x4*>(hp);
                const f16x4 l = *reinterpret_cast<const f16x4*>(hp + 32);
                idv = f32x4{(float)h.x + (float)l.x, (float)h.y + (float)l.y, (float)h.z + (float)l.z,
                            (float)h.w + (float)l.w};
            } else if constexpr (IDM == 2) {
                idv = *reinterpret_cast<const f32x4*>(a.id + (size_t)mc * a.id_ld + n);
            } else if constexpr (IDM == 3) {
                const float sv = *(ri.w != kNoRow ? a.id + ri.w : a.zero);      // (address select: the load stays unconditional)
                idv = f32x4{sv, sv, sv, sv};
            }
            const f32x4 y = epi_combine(av, wsv, c, t, idwv, idv);
            yv[u] = f32x4{fmaxf(y.x, lo_clamp), fmaxf(y.y, lo_clamp), fmaxf(y.z, lo_clamp), fmaxf(y.w, lo_clamp)} * osc;
        }
#pragma unroll
        for (int u = 0; u < GP; ++u) {
            const int m = rowinfo[(pg + u) * PP + prow].z;
            const f32x4 y = yv[u];
            if (m >= 0) {
                if constexpr (OUTS) {
                    f16x4 h, l;
                    float yc;
                    // (negated comparison: NaN counts as saturated too)
                    sat |= !(fmaxf(fmaxf(fabsf(y.x), fabsf(y.y)), fmaxf(fabsf(y.z), fabsf(y.w))) < slim);
                    yc = fminf(fmaxf(y.x, -65504.f), 65504.f); h.x = (_Float16)yc; l.x = (_Float16)(yc - (float)h.x);
                    yc = fminf(fmaxf(y.y, -65504.f), 65504.f); h.y = (_Float16)yc; l.y = (_Float16)(yc - (float)h.y);
                    yc = fminf(fmaxf(y.z, -65504.f), 65504.f); h.z = (_Float16)yc; l.z = (_Float16)(yc - (float)h.z);
                    yc = fminf(fmaxf(y.w, -65504.f), 65504.f); h.w = (_Float16)yc; l.w = (_Float16)(yc - (float)h.w);
                    _Float16* dst = reinterpret_cast<_Float16*>(a.out + (size_t)m * a.ldo) + hoff;
                    __builtin_nontemporal_store(h, reinterpret_cast<f16x4*>(dst));
                    __builtin_nontemporal_store(l, reinterpret_cast<f16x4*>(dst + 32));
                } else {
                    __builtin_nontemporal_store(y, reinterpret_cast<f32x4*>(a.out + (size_t)m * a.ldo + n));
                }
            }
        }
    }
    if constexpr (OUTS) {
        // the split layout clamps to +-65504: tell the host (nhans_take_status) instead of going on silently
        int* const satp = CONV_KARG(sat);
        if (sat && satp) atomicOr(satp, kSatActivation);
    }
}

// The same sweep with 8 channels per thread, for split-NHWC outputs (the f16x3 main path).  In the
// split layout a 32-channel group of a pixel is one 128-byte line = 32 hi halfs + 32 lo halfs; with 4
// channels per thread every residual load and every store is an 8-byte piece (two per pass each),
// with 8 channels they are 16-byte pieces and there are half as many memory instructions per byte --
// the epilogue is bound by memory-instruction issue, not by bytes (8-byte accesses run at 0.54-0.70 x
// the 16-byte rate on this chip).  Same arithmetic per element in the same order: bitwise identical.
// IDM: 0 none, 1 split-NHWC tensor, 3 one-channel image.
// Software-pipelined over groups of GP passes: the global loads of group g+1 (position table, residual,
// and -- unless the tile lies in one clip, HOIST -- the per-clip bias) are in flight while group g is
// combined with its accumulators and stored, so the load path of the CU (64 B/clk through the L1) never
// idles between groups; measured on the round-1 sweep (two groups, each "issue every load, wait,
// store"): 6.8 k cycles from the issue of a group's loads to their arrival, 23 k for the sweep of a
// 512 x 64 tile against 6-8 k of L1 time.
template <int IDM> struct Epi8Raw {               // what one pass has in flight
    f32x4 c0, c1, t0, t1;
    f16x8 h, l;
    float sv;
    int m, p;
};

template <int PREC, int IDM, int HOIST, int PP, int PASSES, int LDC>
__device__ __forceinline__ void conv_epilogue_sweep8(const ConvArgs& a, const float* ct, const int4* rowinfo,
                                                     int prow, int c8, int n, long long* es = nullptr) {
    constexpr int GP = PASSES % 2 == 0 ? 2 : 1;
    constexpr int NG = PASSES / GP;
    const int f_tf = a.tf ? 1 : 0;
    const float* __restrict__ cbp = a.cb;
    const float* __restrict__ tfp = a.tf ? a.tf : a.zero;
    f32x4 ws0 = {1.f, 1.f, 1.f, 1.f}, ws1 = ws0, iw0 = {0.f, 0.f, 0.f, 0.f}, iw1 = iw0;
    if constexpr (PREC == 1) {
        const float in_scale = CONV_KARG(in_scale);
        ws0 = *reinterpret_cast<const f32x4*>(a.ws + n) * in_scale;
        ws1 = *reinterpret_cast<const f32x4*>(a.ws + n + 4) * in_scale;
    }
    if constexpr (IDM != 0) {
        const float id_scale = CONV_KARG(id_scale);
        iw0 = *reinterpret_cast<const f32x4*>(a.idw + n) * id_scale;
        iw1 = *reinterpret_cast<const f32x4*>(a.idw + n + 4) * id_scale;
    }
    const float osc = CONV_KARG(out_scale), slim = CONV_KARG(sat_limit);
    f32x4 hc0 = {0.f, 0.f, 0.f, 0.f}, hc1 = hc0;             // HOIST: the one clip's bias, loaded once
    if constexpr (HOIST) {
        const int cx = rowinfo[0].x;
        hc0 = *reinterpret_cast<const f32x4*>(cbp + cx + n);
        hc1 = *reinterpret_cast<const f32x4*>(cbp + cx + n + 4);
    }
    const int hoff = (n >> 5) * 64 + (n & 31);                // half index inside a split-NHWC pixel (n % 8 == 0)
    const float lo_clamp = a.relu ? 0.f : -3.0e38f;
    int sat = 0;
    // raw buffer over the output from the tile's first pixel on (uniform; every valid pixel of the tile lies behind it)
    const int mb = __builtin_amdgcn_readfirstlane(rowinfo[0].z);
    const uint32_t ost = (uint32_t)a.ldo * 4u;
    const __amdgpu_buffer_rsrc_t orsrc = __builtin_amdgcn_make_buffer_rsrc(a.out + (size_t)mb * a.ldo, 0, 0x7FFFFFFF, 0x00020000);

    auto issue = [&](int g, Epi8Raw<IDM> (&r)[GP]) {
#pragma unroll
        for (int u = 0; u < GP; ++u) {
            const int p = (g * GP + u) * PP + prow;
            const int4 ri = rowinfo[p];
            r[u].p = p;
            r[u].m = ri.z;
            const int mc = ri.z < 0 ? 0 : ri.z;
            if constexpr (!HOIST) {
                r[u].c0 = *reinterpret_cast<const f32x4*>(cbp + ri.x + n);
                r[u].c1 = *reinterpret_cast<const f32x4*>(cbp + ri.x + n + 4);
            }
            r[u].t0 = *reinterpret_cast<const f32x4*>(tfp + (ri.y + n) * f_tf);
            r[u].t1 = *reinterpret_cast<const f32x4*>(tfp + (ri.y + n) * f_tf + 4 * f_tf);
            if constexpr (IDM == 1) {
                const _Float16* hp = reinterpret_cast<const _Float16*>(a.id + (size_t)mc * a.id_ld) + hoff;
                r[u].h = __builtin_nontemporal_load(reinterpret_cast<const f16x8*>(hp));     // (read once: keep the L2 for
                r[u].l = __builtin_nontemporal_load(reinterpret_cast<const f16x8*>(hp + 32)); //  weights, tables and halos)
            } else if constexpr (IDM == 3) {
                r[u].sv = *(ri.w != kNoRow ? a.id + ri.w : a.zero);
            }
        }
    };
    auto finish = [&](Epi8Raw<IDM> (&r)[GP]) {
#pragma unroll
        for (int u = 0; u < GP; ++u) {
            const f32x4 av0 = *reinterpret_cast<const f32x4*>(ct + r[u].p * LDC + c8 * 8);
            const f32x4 av1 = *reinterpret_cast<const f32x4*>(ct + r[u].p * LDC + c8 * 8 + 4);
            f32x4 i0 = {0.f, 0.f, 0.f, 0.f}, i1 = i0;
            if constexpr (IDM == 1) {
                // (float)hi + (float)lo, one v_fma_mix_f32 per value (the same bits as two conversions and an add)
                const f32x4 hv = __builtin_bit_cast(f32x4, r[u].h), lv = __builtin_bit_cast(f32x4, r[u].l);
                i0 = f32x4{unsplit_mix<0>(hv.x, lv.x), unsplit_mix<1>(hv.x, lv.x), unsplit_mix<0>(hv.y, lv.y), unsplit_mix<1>(hv.y, lv.y)};
                i1 = f32x4{unsplit_mix<0>(hv.z, lv.z), unsplit_mix<1>(hv.z, lv.z), unsplit_mix<0>(hv.w, lv.w), unsplit_mix<1>(hv.w, lv.w)};
            } else if constexpr (IDM == 3) {
                i0 = f32x4{r[u].sv, r[u].sv, r[u].sv, r[u].sv};
                i1 = i0;
            }
            const f32x4 c0 = HOIST ? hc0 : r[u].c0, c1 = HOIST ? hc1 : r[u].c1;
            const f32x4 r0 = epi_combine(av0, ws0, c0, r[u].t0, iw0, i0);
            const f32x4 r1 = epi_combine(av1, ws1, c1, r[u].t1, iw1, i1);
            // Everything up to the two stores is unconditional (the saturation flag of a slot that is not stored is
            // masked, not skipped): with the arithmetic inside `if (valid)` the compiler sinks the loads whose values
            // are used only there -- the restrict position table, the bias -- into the branch, right in front of
            // their use, and the software pipeline of issue() / finish() collapses into load, wait, store.
            const bool valid = r[u].m >= 0;
            float yc[8];
            bool over = false;
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const float y = fmaxf(e < 4 ? r0[e] : r1[e - 4], lo_clamp) * osc;
                over |= !(fabsf(y) < slim);
                yc[e] = fminf(fmaxf(y, -65504.f), 65504.f);
            }
            // hi = f16(y), lo = f16(y - hi): one packed conversion and two v_fma_mix per pair (the same bits as
            // four conversions and a subtraction per value)
            uint4 hq, lq;
            split_pair(yc[0], yc[1], &hq.x, &lq.x);
            split_pair(yc[2], yc[3], &hq.y, &lq.y);
            split_pair(yc[4], yc[5], &hq.z, &lq.z);
            split_pair(yc[6], yc[7], &hq.w, &lq.w);
            sat |= (over && valid) ? 1 : 0;
            asm volatile("" : "+v"(sat));              // (here, not after the sweep: the values would stay alive for it)
            // The two stores are UNCONDITIONAL buffer stores relative to the tile's first pixel (a slot that is not
            // stored gets an offset beyond the descriptor's 2 GB: the range check drops it).  Behind an `if (valid)` the
            // compiler's count of what is in flight is a guess after the branch, and the waits for the NEXT group's loads
            // (requested before this group's stores) came out as "all but the five youngest requests have completed".
            {
                typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
                const uint32_t so = valid ? (uint32_t)(r[u].m - mb) * ost + (uint32_t)hoff * 2u : 0x80000000u;
                __builtin_amdgcn_raw_buffer_store_b128(u32x4{hq.x, hq.y, hq.z, hq.w}, orsrc, so, 0, 2);     // (2 = nt: written once)
                __builtin_amdgcn_raw_buffer_store_b128(u32x4{lq.x, lq.y, lq.z, lq.w}, orsrc, so + 64u, 0, 2);
            }
        }
    };

    Epi8Raw<IDM> ra[GP], rb[GP];                      // two groups in flight, statically named
    issue(0, ra);
#pragma unroll
    for (int g = 0; g < NG; g += 2) {
        if (g + 1 < NG) issue(g + 1, rb);
        finish(ra);
        if (kDev && es && g == 0) {                      // dev stamp: the first group is through
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            es[2] = (long long)__builtin_amdgcn_s_memtime();
        }
        if (g + 2 < NG) issue(g + 2, ra);
        if (g + 1 < NG) finish(rb);
    }
    int* const satp = CONV_KARG(sat);
    if (sat && satp) atomicOr(satp, kSatActivation);
}

// acc[i][j]: 32x32 MFMA tile (i: pixels, j: channels) of the wave whose tile-local origin is
// (row_base, col_base); n_tile0: first channel of the workgroup tile; the caller has passed a
// workgroup barrier after its last LDS read of the K loop.
template <int TM, int TN, int PREC, int NTHREADS, int BMROWS, int BNCOLS>
__device__ __forceinline__ void conv_epilogue(const ConvArgs& a, f32x16 (&acc)[TM][TN], float* smem, EpiTile tile,
                                              int row_base, int col_base, int n_tile0, int tid, int lane,
                                              long long* es = nullptr /* dev stamps: [LDS written, barrier passed, group 0 loaded] */) {
    constexpr int LDC = BNCOLS + 4;
    constexpr int C4 = BNCOLS / 4;                    // threads along the channels of one pixel
    constexpr int PP = NTHREADS / C4;                 // pixels per pass
    constexpr int PASSES = BMROWS / PP;
    float* ct = smem;
    int4* rowinfo = reinterpret_cast<int4*>(smem + BMROWS * LDC);

    // 1. accumulators -> LDS tile [pixel][channel]; lane l: pixel l&31, channels 8g + 4(l>>5) + {0..3}
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const f32x4 v = {acc[i][j][4 * g], acc[i][j][4 * g + 1], acc[i][j][4 * g + 2], acc[i][j][4 * g + 3]};
                *reinterpret_cast<f32x4*>(ct + (row_base + i * 32 + (lane & 31)) * LDC + col_base + j * 32 + 8 * g +
                                          4 * (lane >> 5)) = v;
            }
    // 2. per-pixel (clip, h, w, 1-channel residual index)
    if (tid < BMROWS) {
        const int mz = tile(tid, a);
        // slots past the end are never stored; they must look like the LAST valid pixel of a linear tile, because
        // "first row's clip == last row's clip" below decides whether one bias vector serves the whole tile (a
        // partial last tile that looked like its first pixel there gave the next clip's first frame the wrong bias)
        const int m = mz < 0 ? (tile.tw > 0 ? tile(0, a) : a.M - 1) : mz;
        const uint32_t b = fd_div((uint32_t)m, a.fdHoWo);
        const uint32_t rem = (uint32_t)m - b * a.fdHoWo.d;
        const uint32_t ho = fd_div(rem, a.fdWo);
        const uint32_t wo = rem - ho * a.fdWo.d;
        const int clip = a.img_clip ? a.img_clip[b] : 0;
        // (1-channel residual image: element index into a.id, or kNoRow = a zero row of a sliding-window image, WinRows; a
        // window's rows are counted from the launch's first frame, so a VALID index may be negative: the rows of a clip that
        // began before this chunk of frames)
        int ids;
        if (a.id_mode == 2 && a.id_win.t) {
            const int h = (int)ho * a.idsh;
            ids = win_row_ok(a.id_win, (int)b, h) ? (a.id_win.row0 + (int)b + h) * a.idW + (int)wo * a.idsw : kNoRow;
        } else {
            ids = (int)((b * a.idH + ho * a.idsh) * a.idW + wo * a.idsw);
        }
        rowinfo[tid] = make_int4(clip * a.cb_stride, (int)rem * a.N, mz, ids);
    }
    if (kDev && es) { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); es[0] = (long long)__builtin_amdgcn_s_memtime(); }
    __syncthreads();
    if (kDev && es) es[1] = (long long)__builtin_amdgcn_s_memtime();

    // 3. row-major sweep: thread = (pixel within pass, 4 channels)
    const int c4 = tid % C4, prow = tid / C4;
    const int n = n_tile0 + c4 * 4;
    const int f_tf = a.tf ? 1 : 0;
    const bool id_split = a.id_mode == 1 && a.id_split;
    const float* __restrict__ cbp = a.cb;
    const float* __restrict__ tfp = a.tf ? a.tf : a.zero;     // an absent table reads the zero page
    const bool vec = (a.Nreal == a.N) && !a.aux && (a.out_split || (a.ldo & 3) == 0) &&
                     (a.id_mode != 1 || id_split || (a.id_ld & 3) == 0);
    constexpr int C8 = BNCOLS / 8, PP8 = NTHREADS / C8, PASSES8 = BMROWS / PP8;
    if (vec && a.epi8 && a.out_split && (a.id_mode == 0 || id_split || a.id_mode == 2)) {
        // split-NHWC output (the f16x3 main path): 8 channels per thread, 16-byte pieces throughout
        const int c8 = tid % C8, prow8 = tid / C8;
        const int n8 = n_tile0 + c8 * 8;
        // one clip for the whole tile (pixels are ordered by clip, so first == last decides; rows past
        // the end of the tensor carry the last valid row's clip): its bias vector is loaded once per thread
        const bool one_clip = rowinfo[0].x == rowinfo[BMROWS - 1].x;
        const int mode8 = (a.id_mode == 0 ? 0 : a.id_mode == 1 ? 1 : 2) * 2 + (one_clip ? 1 : 0);
        switch (mode8) {
            case 0: conv_epilogue_sweep8<PREC, 0, 0, PP8, PASSES8, LDC>(a, ct, rowinfo, prow8, c8, n8, es); break;
            case 1: conv_epilogue_sweep8<PREC, 0, 1, PP8, PASSES8, LDC>(a, ct, rowinfo, prow8, c8, n8, es); break;
            case 2: conv_epilogue_sweep8<PREC, 1, 0, PP8, PASSES8, LDC>(a, ct, rowinfo, prow8, c8, n8, es); break;
            case 3: conv_epilogue_sweep8<PREC, 1, 1, PP8, PASSES8, LDC>(a, ct, rowinfo, prow8, c8, n8, es); break;
            case 4: conv_epilogue_sweep8<PREC, 3, 0, PP8, PASSES8, LDC>(a, ct, rowinfo, prow8, c8, n8, es); break;
            default: conv_epilogue_sweep8<PREC, 3, 1, PP8, PASSES8, LDC>(a, ct, rowinfo, prow8, c8, n8, es); break;
        }
    } else if (vec) {
        // residual mode / output layout are resolved once, outside the loops: the sweep below is one
        // straight-line block per group of passes, so all its loads issue before the first wait
        const int mode = (a.id_mode == 1 ? (id_split ? 1 : 2) : a.id_mode == 2 ? 3 : 0) * 2 + (a.out_split ? 1 : 0);
        switch (mode) {
            case 0: conv_epilogue_sweep<PREC, 0, 0, PP, PASSES, LDC>(a, ct, rowinfo, prow, c4, n); break;
            case 1: conv_epilogue_sweep<PREC, 0, 1, PP, PASSES, LDC>(a, ct, rowinfo, prow, c4, n); break;
            case 2: conv_epilogue_sweep<PREC, 1, 0, PP, PASSES, LDC>(a, ct, rowinfo, prow, c4, n); break;
            case 3: conv_epilogue_sweep<PREC, 1, 1, PP, PASSES, LDC>(a, ct, rowinfo, prow, c4, n); break;
            case 4: conv_epilogue_sweep<PREC, 2, 0, PP, PASSES, LDC>(a, ct, rowinfo, prow, c4, n); break;
            case 5: conv_epilogue_sweep<PREC, 2, 1, PP, PASSES, LDC>(a, ct, rowinfo, prow, c4, n); break;
            case 6: conv_epilogue_sweep<PREC, 3, 0, PP, PASSES, LDC>(a, ct, rowinfo, prow, c4, n); break;
            default: conv_epilogue_sweep<PREC, 3, 1, PP, PASSES, LDC>(a, ct, rowinfo, prow, c4, n); break;
        }
    } else {
        // ragged output (last_dense: 201 of 256 columns, unaligned rows, optional pre-residual tap)
        const float rg_in = CONV_KARG(in_scale), rg_id = CONV_KARG(id_scale), rg_out = CONV_KARG(out_scale);
        for (int ps = 0; ps < PASSES; ++ps) {
            const int p = ps * PP + prow;
            const int4 ri = rowinfo[p];
            const int m = ri.z;
            if (m < 0) continue;
            const float idsv = a.id_mode == 2 && ri.w != kNoRow ? a.id[ri.w] : 0.f;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int ne = n + e;
                if (ne >= a.Nreal) continue;
                float x = ct[p * LDC + c4 * 4 + e];
                x = __builtin_fmaf(x, PREC == 1 ? a.ws[ne] * rg_in : 1.f, cbp[ri.x + ne]) + tfp[(ri.y + ne) * f_tf];
                float y = x;
                const float iw = a.id_mode ? a.idw[ne] * rg_id : 0.f;
                if (a.id_mode == 1)
                    y = __builtin_fmaf(iw, id_split ? split_load(a.id, (size_t)m * a.id_ld, ne) : a.id[(size_t)m * a.id_ld + ne], y);
                else if (a.id_mode == 2) y = __builtin_fmaf(iw, idsv, y);
                if (a.relu) y = fmaxf(y, 0.f);
                if (a.aux) a.aux[(size_t)m * a.aux_ld + ne] = x;
                a.out[(size_t)m * a.ldo + ne] = y * rg_out;
            }
        }
    }
}

}  // namespace nhans
