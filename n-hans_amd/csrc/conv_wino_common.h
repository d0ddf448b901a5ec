// Part of conv_wino.hip: the split / un-split helpers, the layout of the exchanged accumulator tiles and the fused
// output-transform + block epilogue of the workgroup's 512 threads (one pass = 64 tile-pixels x 64 channels x 8
// positions through LDS; the kernel runs two).
#pragma once
#include "conv_epilogue.h"
#include <cstddef>

#include <algorithm>

namespace nhans {

// Behind every hand-written wait: an EMPTY asm statement that reads and "writes" every register the matching request
// block wrote.  A register an asm request writes is, to the compiler, defined when the asm statement ends; if nothing
// reads it afterwards it is dead from there on, gets reused, and the load -- still in flight -- lands on the new value
// (round 5's memory fault).  With the pin the requested registers are live from the request to behind the wait by
// CONSTRUCTION, whatever later edits do to the code that uses them; tools/check_wino_isa.py stays as the second line.
#define NH_LANDED2(A, B) asm volatile("" : "+v"(A), "+v"(B))
#define NH_LANDED4(A, B, C, D) asm volatile("" : "+v"(A), "+v"(B), "+v"(C), "+v"(D))

namespace {
constexpr int W_LDM = 68;                      // epilogue: floats per tile-pixel row of an M_p tile (64 + 4)
// LDS of one epilogue pass: eight M_p tiles of 64 tile-pixels + the block's constants (ws, idw, one bias row per block row)
constexpr size_t kWinoLdsEpi = (size_t)(8 * 64 * W_LDM + (2 + 64) * 64) * sizeof(float);

}  // namespace

// Epilogue of a consumer thread: the wave's accumulator tiles M_p go to LDS, then the thread = (tile-pixel q, 8
// channels) forms its MO output columns and runs the fused block epilogue on them.
//   * `ct` holds the eight transformed-domain tiles M_p[64 tile-pixels][W_LDM] (split output: channel 8a + 4b + c of a row
//     at float b*32 + a*4 + c; f32 output: channel n at float n -- either way thread c8 finds its two 4-channel pieces at
//     floats 4 c8 and 32 + 4 c8, and the eight threads of a tile-pixel read 128 contiguous bytes at a time).  The thread reads its
//     8 x 8 values ONCE and forms all MO columns Y_i = sum_p AT[i][p] M_p with the shared sums of the +-1, +-2, +-1/2
//     point pairs (18 instead of 8*MO operations per channel), two channels at a time, and applies the first step of
//     the block epilogue, fma(y, ws, bias), on the spot.
//   * The MO columns of a tile-pixel are MO consecutive pixels of one image row of ONE frame: every address is a
//     uniform frame base (scalar registers) + one 32-bit offset + a column stride, instead of a 64-bit pointer and a
//     row-info record per column.  A column past the image's right edge (or a slot outside the block) loads from the
//     thread's last valid pixel and is not stored.
//   * The residual -- HBM, ~2,000 cycles away -- of ALL columns is requested BEFORE the accumulators go to LDS: the
//     exchange, the barrier and the output transform run under that latency.  The position table (L2) follows one
//     column ahead of its use.  Everything up to the two stores of a column is unconditional: with the arithmetic
//     inside `if (valid)` the compiler sinks the (restrict) table loads into the branch, right in front of their use,
//     with s_waitcnt vmcnt(0) -- one full memory latency per column, 2,750 cycles each, was what round 3's first
//     version of this sweep spent.
// IDM: 0 no residual, 1 split-NHWC tensor, 2 f32 NHWC tensor, 3 one-channel image.
template <int IDM, int MO, int OUTS>       // OUTS: the output is split NHWC (1) or f32 NHWC (0: no split on the way out)
__device__ __forceinline__ void wino_epilogue(const ConvArgs& a, const f32x16 (*acc)[2], float* ct, int p, int lane, int tid,
                                              int b, int nb, int r0, int j0, int TR, int TJ, int cx, long long* es,
                                              int qbase = 0, bool first = true) {
    static_assert(!(IDM == 1 && !OUTS), "a split residual needs the split output's channel assignment (launch_conv_wino refuses the pair)");
    const int g8 = lane >> 5;
    const int c8 = tid & 7, q = tid >> 3;
    // The thread's 8 channels, as two pieces of 4.  Split-NHWC output: channels 8 c8 .. 8 c8 + 7 (16 bytes of hi halves +
    // 16 of lo halves per access).  f32 output (round 5): channels 4 c8 .. + 3 and 32 + 4 c8 .. + 3 -- the eight threads
    // of a tile-pixel then cover one whole 128-byte line per access (residual, position table, store) where 8
    // contiguous channels per thread made every access touch both lines of the pixel at half use.
    constexpr int CHA = OUTS ? 8 : 4, CHB = OUTS ? 4 : 32;          // first channel of piece 0 = CHA c8; piece 1 = + CHB
    const int n = nb * 64 + c8 * CHA;
    const int n8 = nb * 64 + c8 * 8;                                // (the 8 contiguous channels a head thread fetches constants for)
    const int rr = (q + qbase) / TJ, tt = q + qbase - rr * TJ;
    const int ho = r0 + rr, wo0 = (j0 + tt) * MO;
    const bool okq = rr < TR && ho < a.Ho && j0 + tt < a.wino_ntile;
    const int nvalid = okq ? (a.Wo - wo0 < MO ? a.Wo - wo0 : MO) : 0;       // columns of this tile-pixel inside the image
    const int pix0 = okq ? ho * a.Wo + wo0 : 0;                                // first pixel, within the frame
    const int lastc = nvalid > 0 ? nvalid - 1 : 0;
    const int hoff = (n >> 5) * 64 + (n & 31);                                // half index inside a split-NHWC pixel
    const size_t fpix = (size_t)b * a.Ho * a.Wo;                              // uniform
    // position table = tt[ho] + ff[wo] (two small arrays: the [Ho*Wo, N] table of these layers, 1.8 MB, did not survive
    // in L2 and came from HBM again almost once per frame).  An absent table reads the zero page.
    const float lo_clamp = a.relu ? 0.f : -3.0e38f;

    // (IDM 3 with a sliding-window image: the frame's position in its clip and the clip's length -- uniform, SCALAR loads,
    // here in front of the counted vector-memory requests; a plain image: every row valid)
    int win_t = a.id_win.pad, win_T = 0x7fffffff;
    if constexpr (IDM == 3) {
        if (a.id_win.t) {
            const int bu = __builtin_amdgcn_readfirstlane(b);
            win_t = a.id_win.t[bu];
            win_T = a.id_win.T[bu];
        }
    }

    // 0. the per-channel constants (ws, bias, idw of the block's 64 channels), fetched by the eight threads of
    // tile-pixel 0 BEFORE the residual requests -- loads return in order, so a constant fetched after them could not be
    // used until every residual has arrived -- and handed to everybody through LDS with the accumulator tiles
    // cst: [ws | idw][64], then per block row r: bias + tt[r0 + r] (the time term of the position table rides in the bias
    // of the transform stage: v = fma(y, ws, bias + tt) + ff), fetched by the first tile-pixel of each row
    float* const cst = ct + 8 * 64 * W_LDM;
    // ffl: the frequency term of the position table for the block's TJ*MO image columns and 64 channels, [column][64]
    // floats in rows 20 .. of the constants area (rows 2 .. 19 hold the per-row biases: TR <= 18), staged ONCE per
    // workgroup by LDS-DMA -- wave p the columns 4p .. 4p+3, wave 0 also 32 .. 35 -- and read from LDS by both passes'
    // sweeps (round 5).  The table used to be 10 global loads per thread and pass: a third of the epilogue's vector-memory
    // instructions, for 9 KB per workgroup that every image row reads again -- and what the epilogue is short of is
    // vector-memory ISSUE.  A column outside the image, or an absent table (descriptor of size 0),
    // stages zeros.  The request is asm: the compiler makes every LDS access it can see wait for a VISIBLE LDS-DMA of
    // the same wave with vmcnt(0) -- which would make the accumulator dump wait for the residual requests as well.  It
    // is the OLDEST request of the pass, so the counted wait in front of the constants' LDS stores covers it, and the
    // barrier behind those publishes it.
    float* const ffl = cst + (2 + 18) * 64;
    static_assert((2 + 18) * 64 + 36 * 64 <= (2 + 64) * 64, "ffl inside the constants area");
    if (first) {
        const __amdgpu_buffer_rsrc_t frs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.tt ? a.ff : a.zero), 0,
                                                                              a.tt ? a.Wo * a.N * 4 : 0, 0x00020000);
        auto stage = [&](int c0) {
            const int col = c0 + (lane >> 4), wo = j0 * MO + col;
            const unsigned off = (col < TJ * MO && wo < a.Wo) ? (unsigned)((wo * a.N + nb * 64 + (lane & 15) * 4) * 4) : 0x80000000u;
            const unsigned dst = (unsigned)(size_t)(__attribute__((address_space(3))) float*)(ffl + c0 * 64);   // LDS byte address (uniform)
            unsigned keep;
            asm volatile("s_mov_b32 %0, m0\n s_mov_b32 m0, %2\n s_nop 0\n buffer_load_dwordx4 %1, %3, 0 offen lds\n s_mov_b32 m0, %0"
                         : "=&s"(keep) : "v"(off), "s"(__builtin_amdgcn_readfirstlane(dst)), "s"(frs) : "memory");
        };
        stage(p * 4);
        if (p == 0) stage(32);
    }
    if (kDev && es) es[3] = (long long)__builtin_amdgcn_s_memtime();
    // These eight requests are asm, and so is the wait for them further down: the compiler's count of what is in flight
    // does not survive the two branches, and left to it the wait in front of the LDS stores of the constants was
    // s_waitcnt vmcnt(0) -- every residual had to LAND before the barrier instead of under the exchange and the output
    // transform.  (The compiler believes the registers valid from here on and inserts no wait of its own; nothing
    // reads them before the hand-written one.)
    f32x4 ka[4], kb[4];
    const bool row_head = tt == 0 && rr < TR;
    const bool chan_head = q == 0 && first;
    // EVERY register such a request writes must be READ behind the wait: an output the compiler can prove unused is dead to it
    // from the asm statement on, it puts other values there, and the load -- still in flight -- lands on them.  (Round 5:
    // without a residual the idw pair was requested from the zero page and never read; variants of this epilogue whose
    // register allocation put address arithmetic into those registers raised memory faults on some boxes of the pool.
    // tools/check_wino_isa.py now checks the whole window from each request block to the wait.)
    if (chan_head) {
        const float* wsp = a.ws + n8;
        if constexpr (IDM != 0) {
            const float* idp = a.idw + n8;
            asm volatile("global_load_dwordx4 %0, %4, off\n global_load_dwordx4 %1, %4, off offset:16\n"
                         "global_load_dwordx4 %2, %5, off\n global_load_dwordx4 %3, %5, off offset:16"
                         : "=&v"(ka[0]), "=&v"(ka[1]), "=&v"(ka[2]), "=&v"(ka[3]) : "v"(wsp), "v"(idp) : "memory");
        } else {
            asm volatile("global_load_dwordx4 %0, %2, off\n global_load_dwordx4 %1, %2, off offset:16"
                         : "=&v"(ka[0]), "=&v"(ka[1]) : "v"(wsp) : "memory");
        }
    }
    if (row_head) {
        const float* ttp = a.tt ? a.tt + (size_t)(ho < a.Ho ? ho : 0) * a.N + n8 : a.zero;
        const float* cbp = a.cb + cx + n8;
        asm volatile("global_load_dwordx4 %0, %4, off\n global_load_dwordx4 %1, %4, off offset:16\n"
                     "global_load_dwordx4 %2, %5, off\n global_load_dwordx4 %3, %5, off offset:16"
                     : "=&v"(kb[0]), "=&v"(kb[1]), "=&v"(kb[2]), "=&v"(kb[3]) : "v"(cbp), "v"(ttp) : "memory");
    }
    __builtin_amdgcn_sched_barrier(0);

    // 1. residual requests.  Only the first RQ0 columns' go out here, before the accumulators leave for LDS; the others
    // follow one column per iteration of the output transform (round 5).  A wave that issues all ten requests in one burst
    // stalls until the CU's vector-memory queue has taken them -- with eight waves doing so at once the burst took 5 k
    // cycles per wave and pass in which it neither dumped nor transformed (stamps: profiles/r05) -- and the sweep needs
    // column i's residual only after columns 0 .. i-1 are through.
    // (non-temporal: the residual is read once and the output written once -- left at the default policy these streams
    // pushed the input tiles' halo rows and the weights out of the 4 MB L2 before the neighbouring workgroup came for
    // them: with the hint the 64-channel layers' HBM reads equal their algorithmic bytes)
    constexpr int RQ0 = 2;
    f32x4 rh[MO], rl[MO];
    float rsv[MO];
    auto residual = [&](int i) {
        if constexpr (IDM == 1) {
            const char* const idb = reinterpret_cast<const char*>(a.id + fpix * a.id_ld);
            const uint32_t o0 = (uint32_t)pix0 * (uint32_t)a.id_ld * 4u + (uint32_t)hoff * 2u, st = (uint32_t)a.id_ld * 4u;
            const uint32_t o = (kDev && (a.wino_m >> 8 & 24)) ? (uint32_t)hoff * 2u : o0 + (uint32_t)(i < lastc ? i : lastc) * st;
            rh[i] = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(idb + o));
            rl[i] = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(idb + o + 64));
        } else if constexpr (IDM == 2) {
            const char* const idb = reinterpret_cast<const char*>(a.id + fpix * a.id_ld);
            const uint32_t o0 = ((uint32_t)pix0 * (uint32_t)a.id_ld + (uint32_t)n) * 4u, st = (uint32_t)a.id_ld * 4u;
            const uint32_t o = (kDev && (a.wino_m >> 8 & 24)) ? (uint32_t)n * 4u : o0 + (uint32_t)(i < lastc ? i : lastc) * st;   // (dev ablation 8 / 16: one L2-hot line)
            rh[i] = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(idb + o));
            rl[i] = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(idb + o + CHB * 4));
        } else if constexpr (IDM == 3) {
            // one-channel image; with id_win a sliding window of a.id (the log-magnitude spectrogram itself, round 6): a row
            // outside the clip is a row of zeros -- the address is selected (the zero page, stride 0), the load unconditional
            const int hrow = (okq ? ho : 0) * a.idsh;
            const bool rowok = (unsigned)(win_t + hrow - a.id_win.pad) < (unsigned)win_T;
            const int ids0 = (a.id_win.t ? (a.id_win.row0 + b + hrow) : (b * a.idH + hrow)) * a.idW + (okq ? wo0 : 0) * a.idsw;
            const float* const idp = rowok ? a.id + ids0 : a.zero;
            rsv[i] = idp[rowok ? (i < lastc ? i : lastc) * a.idsw : 0];
        }
    };
#pragma unroll
    for (int i = 0; i < RQ0; ++i) residual(i);
    __builtin_amdgcn_sched_barrier(0);
    if (kDev && es) es[4] = (long long)__builtin_amdgcn_s_memtime();

    // 2. accumulators -> LDS.  Channel n = 8a + 4b + c of the 64 sits at float b*32 + a*4 + c of its row: the sweep
    // thread of channel group a reads two 16-byte pieces, and eight such threads cover 128 contiguous bytes each time
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int q4 = 0; q4 < 4; ++q4) {
                const f32x4 v = {acc[t][j][4 * q4], acc[t][j][4 * q4 + 1], acc[t][j][4 * q4 + 2], acc[t][j][4 * q4 + 3]};
                // (lane's quad = channels 32 j + 8 q4 + 4 g8 .. + 3 of the block: row position as documented above)
                *reinterpret_cast<f32x4*>(ct + (p * 64 + t * 32 + (lane & 31)) * W_LDM + (OUTS ? g8 * 32 + (j * 4 + q4) * 4 : 32 * j + 8 * q4 + 4 * g8)) = v;
            }
    // the constants have landed when at most the residual requests (younger, in order) are in flight
    if constexpr (IDM == 1 || IDM == 2) asm volatile("s_waitcnt vmcnt(%0)" :: "n"(2 * RQ0) : "memory");
    else if constexpr (IDM == 3) asm volatile("s_waitcnt vmcnt(%0)" :: "n"(RQ0) : "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    // (round 6: the output's power-of-two scale 2^-e rides in the constants -- ws, bias, idw times out_scale -- instead of
    // one multiply per output element at the end of the sweep; exact, so the same bits)
    const float osc_c = CONV_KARG(out_scale);
    if (chan_head) {
        if constexpr (IDM != 0) NH_LANDED4(ka[0], ka[1], ka[2], ka[3]); else NH_LANDED2(ka[0], ka[1]);
        const float in_scale = CONV_KARG(in_scale) * osc_c, id_scale = CONV_KARG(id_scale) * osc_c;
        *reinterpret_cast<f32x4*>(cst + c8 * 8) = ka[0] * in_scale; *reinterpret_cast<f32x4*>(cst + c8 * 8 + 4) = ka[1] * in_scale;
        if constexpr (IDM != 0) {
            *reinterpret_cast<f32x4*>(cst + 64 + c8 * 8) = ka[2] * id_scale; *reinterpret_cast<f32x4*>(cst + 64 + c8 * 8 + 4) = ka[3] * id_scale;
        }
    }
    if (row_head) {
        NH_LANDED4(kb[0], kb[1], kb[2], kb[3]);
        *reinterpret_cast<f32x4*>(cst + (2 + rr) * 64 + c8 * 8) = (kb[0] + kb[2]) * osc_c;           // (an absent table: 32 bytes of the zero page)
        *reinterpret_cast<f32x4*>(cst + (2 + rr) * 64 + c8 * 8 + 4) = (kb[1] + kb[3]) * osc_c;
    }
    if (kDev && es) { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); es[0] = (long long)__builtin_amdgcn_s_memtime(); }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();                              // (raw: the residual loads stay in flight across it)
    if (kDev && es) es[1] = (long long)__builtin_amdgcn_s_memtime();

    // 3. output transform, two channels at a time (8 x 8-byte LDS reads), with fma(y, ws, bias) -- the first step of
    // the block epilogue -- applied on the spot.  The MO results go straight BACK to LDS, into the slots of positions
    // 0 .. MO-1 the thread has just read (nobody else touches the slots of its tile-pixel and channels): the column loop
    // below then holds one column of outputs instead of MO, which is what lets the residual of all columns stay in
    // registers.  (Stored tensors carry 2^-e: ConvArgs::in_scale / id_scale / out_scale.)
    // the position table's frequency term: from the workgroup's staged slice (ffl), column tt*MO + i of the block
    const float* const ffc = ffl + ((rr < TR ? tt : 0) * MO) * 64 + c8 * CHA;
    __builtin_amdgcn_sched_barrier(0);
    float* const my = ct + q * W_LDM + c8 * 4;                                // + position * 64 * W_LDM + (0 | 32)
    static_assert(MO - RQ0 <= 4, "one late residual column per transform iteration");
#pragma unroll
    for (int qt = 0; qt < 4; ++qt) {
        if (RQ0 + qt < MO) residual(RQ0 + qt);
        float* const slot = my + (qt >> 1) * 32 + (qt & 1) * 2;
        f32x2 m[8];
#pragma unroll
        for (int pp = 0; pp < 8; ++pp) m[pp] = *reinterpret_cast<const f32x2*>(slot + pp * 64 * W_LDM);
        const int cq = c8 * CHA + (qt >> 1) * CHB + (qt & 1) * 2;            // the pair's first channel within the block
        const f32x2 ws2 = *reinterpret_cast<const f32x2*>(cst + cq);
        const f32x2 hc2 = *reinterpret_cast<const f32x2*>(cst + (2 + (rr < TR ? rr : 0)) * 64 + cq);
        const f32x2 s12 = m[1] + m[2], d12 = m[1] - m[2], s34 = m[3] + m[4], d34 = m[3] - m[4], s56 = m[5] + m[6], d56 = m[5] - m[6];
        f32x2 y[MO];
        y[0] = (m[0] + s12) + (s34 + s56);
        y[1] = d12 + 2.f * d34 + 0.5f * d56;
        y[2] = s12 + 4.f * s34 + 0.25f * s56;
        y[3] = d12 + 8.f * d34 + 0.125f * d56;
        if constexpr (MO == 5) {
            y[4] = (s12 + m[7]) + 16.f * s34 + 0.0625f * s56;
        } else {
            y[4] = s12 + 16.f * s34 + 0.0625f * s56;
            y[MO - 1] = (d12 + m[7]) + 32.f * d34 + 0.03125f * d56;
        }
#pragma unroll
        for (int i = 0; i < MO; ++i)
            *reinterpret_cast<f32x2*>(slot + i * 64 * W_LDM) =
                f32x2{__builtin_fmaf(y[i].x, ws2.x, hc2.x), __builtin_fmaf(y[i].y, ws2.y, hc2.y)};
        __builtin_amdgcn_sched_barrier(0);
    }
    if (kDev && es) es[2] = (long long)__builtin_amdgcn_s_memtime();

    // 4. columns
    f32x4 iw0 = {0.f, 0.f, 0.f, 0.f}, iw1 = iw0;
    if constexpr (IDM != 0) {
        iw0 = *reinterpret_cast<const f32x4*>(cst + 64 + c8 * CHA);
        iw1 = *reinterpret_cast<const f32x4*>(cst + 64 + c8 * CHA + CHB);
    }
    const float osc = CONV_KARG(out_scale), slim = CONV_KARG(sat_limit);
    char* const outb = reinterpret_cast<char*>(a.out + fpix * a.ldo);
    // (raw buffer over this frame's output: base and size are uniform; 0x00020000 = 32-bit data format, gfx9 family)
    const __amdgpu_buffer_rsrc_t orsrc = __builtin_amdgcn_make_buffer_rsrc(outb, 0, a.Ho * a.Wo * a.ldo * 4, 0x00020000);
    int n_again = n;                                           // (recomputed, not kept: one register fewer across the transform)
    asm volatile("" : "+v"(n_again));
    const uint32_t oo0 = (uint32_t)pix0 * (uint32_t)a.ldo * 4u + (uint32_t)((n_again >> 5) * 64 + (n_again & 31)) * 2u, ost = (uint32_t)a.ldo * 4u;
    const uint32_t oo0f = (uint32_t)pix0 * (uint32_t)a.ldo * 4u + (uint32_t)n_again * 4u;
    int sat = 0;
#pragma unroll
    for (int i = 0; i < MO; ++i) {
        const f32x4 ya = *reinterpret_cast<const f32x4*>(my + i * 64 * W_LDM);
        const f32x4 yb = *reinterpret_cast<const f32x4*>(my + i * 64 * W_LDM + 32);
        f32x4 i0 = {0.f, 0.f, 0.f, 0.f}, i1 = i0;
        if constexpr (IDM == 1) {
            // (float)hi + (float)lo, one v_fma_mix_f32 per value
            i0 = f32x4{unsplit_mix<0>(rh[i].x, rl[i].x), unsplit_mix<1>(rh[i].x, rl[i].x), unsplit_mix<0>(rh[i].y, rl[i].y), unsplit_mix<1>(rh[i].y, rl[i].y)};
            i1 = f32x4{unsplit_mix<0>(rh[i].z, rl[i].z), unsplit_mix<1>(rh[i].z, rl[i].z), unsplit_mix<0>(rh[i].w, rl[i].w), unsplit_mix<1>(rh[i].w, rl[i].w)};
        } else if constexpr (IDM == 2) {
            i0 = rh[i];
            i1 = rl[i];
        } else if constexpr (IDM == 3) {
            i0 = f32x4{rsv[i], rsv[i], rsv[i], rsv[i]};
            i1 = i0;
        }
        const f32x4 t0 = *reinterpret_cast<const f32x4*>(ffc + i * 64), t1 = *reinterpret_cast<const f32x4*>(ffc + i * 64 + CHB);
        const bool valid = i < nvalid;
        float yc[8];
        bool over = false;
        // ya / yb / iw carry out_scale already (exact: a power of two); the table term joins by one fma.  The saturation test is
        // ONE compare on the largest magnitude of the thread's eight values (a NaN cannot arrive here as a NaN: fmaxf with the
        // relu floor -- or with -3e38 -- has replaced it), and the clamp runs only in a wave that has something to clamp: below
        // the limit (<= 65504) it is the identity.
        const f32x4 oscv = {osc, osc, osc, osc};
        const f32x4 r0v = fma4(iw0, i0, fma4(t0, oscv, ya));
        const f32x4 r1v = fma4(iw1, i1, fma4(t1, oscv, yb));
#pragma unroll
        for (int e = 0; e < 8; ++e) yc[e] = fmaxf(e < 4 ? r0v[e] : r1v[e - 4], lo_clamp);
        {
            float mx = __builtin_fmaxf(__builtin_fmaxf(__builtin_fabsf(yc[0]), __builtin_fabsf(yc[1])), __builtin_fabsf(yc[2]));
            mx = __builtin_fmaxf(__builtin_fmaxf(mx, __builtin_fabsf(yc[3])), __builtin_fabsf(yc[4]));
            mx = __builtin_fmaxf(__builtin_fmaxf(mx, __builtin_fabsf(yc[5])), __builtin_fabsf(yc[6]));
            mx = __builtin_fmaxf(mx, __builtin_fabsf(yc[7]));
            over = !(mx < slim);
        }
        if (__builtin_amdgcn_ballot_w64(over) != 0) {
#pragma unroll
            for (int e = 0; e < 8; ++e) yc[e] = __builtin_amdgcn_fmed3f(yc[e], -65504.f, 65504.f);
        }
        sat |= (over && valid) ? 1 : 0;
        asm volatile("" : "+v"(sat));                       // (here, not after the loop: the compiler would keep all 8*MO values alive for it)
        // The two stores of a column are UNCONDITIONAL buffer stores (an invalid column's offset lies beyond the frame: the
        // range check of the descriptor drops it): behind an `if (valid)` the compiler's count of stores in flight is a
        // guess, and the next column's table wait became "all earlier stores have completed".
        // Output layout: split NHWC (16 bytes of hi halves, 16 of lo halves, 64 apart) or, for a tensor that only
        // Winograd launches read, f32 NHWC (the 8 channels' 32 contiguous bytes; the same scaled, clamped values).
        {
            typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
            const bool ok = valid && !(kDev && (a.wino_m >> 8 & 8));
            const uint32_t so = ok ? (OUTS ? oo0 : oo0f) + (uint32_t)i * ost : 0x80000000u;
            u32x4 d0, d1;
            if constexpr (OUTS) {
                uint4 hb, lb;
                split_pair(yc[0], yc[1], &hb.x, &lb.x);
                split_pair(yc[2], yc[3], &hb.y, &lb.y);
                split_pair(yc[4], yc[5], &hb.z, &lb.z);
                split_pair(yc[6], yc[7], &hb.w, &lb.w);
                d0 = u32x4{hb.x, hb.y, hb.z, hb.w};
                d1 = u32x4{lb.x, lb.y, lb.z, lb.w};
            } else {
                d0 = u32x4{__builtin_bit_cast(unsigned, yc[0]), __builtin_bit_cast(unsigned, yc[1]), __builtin_bit_cast(unsigned, yc[2]), __builtin_bit_cast(unsigned, yc[3])};
                d1 = u32x4{__builtin_bit_cast(unsigned, yc[4]), __builtin_bit_cast(unsigned, yc[5]), __builtin_bit_cast(unsigned, yc[6]), __builtin_bit_cast(unsigned, yc[7])};
            }
            __builtin_amdgcn_raw_buffer_store_b128(d0, orsrc, so, 0, 2);          // (2 = nt: written once, read by the next launch)
            __builtin_amdgcn_raw_buffer_store_b128(d1, orsrc, so + (OUTS ? 64u : (unsigned)(CHB * 4)), 0, 2);
        }
        __builtin_amdgcn_sched_barrier(0);                     // (column by column: bounded register pressure)
    }
    if (kDev && es) es[5] = (long long)__builtin_amdgcn_s_memtime();
    int* const satp = CONV_KARG(sat);
    if (sat && satp) atomicOr(satp, kSatActivation);
}

}  // namespace nhans
