// Internal declarations shared by the HIP translation units of libnhans_hip.so.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <tuple>
#include <utility>

namespace nhans {

constexpr int kWin = 400, kHop = 160, kBins = 201, kMixWin = 35, kCtxFrames = 200, kEmb = 512;
constexpr int kCenter = kMixWin / 2;
// frame windows per pass of the stack: frames x 35 x 201 x 64 elements < 2^31 (32-bit element offsets in the conv kernels)
constexpr int64_t kMaxFramesPerChunk = ((int64_t)1 << 31) / ((int64_t)kMixWin * kBins * 64);

// Developer tooling (cycle stamps, timing ablations that compute WRONG results, kernel-choice
// switches read from the environment) exists only in a `make DEV=1` build (-DNHANS_DEV); the
// default library contains none of it.
#ifdef NHANS_DEV
constexpr bool kDev = true;
int dev_ablate();          // getenv("NHANS_ABLATE"), read once
#else
constexpr bool kDev = false;
constexpr int dev_ablate() { return 0; }
#endif

// ---------------------------------------------------------------------------------------------
// Launch-error channel (launch_status.hip).  Every kernel of the library is launched through NHANS_LAUNCH, which takes
// the launch's OWN return code (hipLaunchKernel) -- not the runtime's sticky per-thread "last error", which an earlier
// HIP call of the application may have left set and which is neither blamed on this library nor consumed on the
// application's behalf --, and set_max_dynamic_lds() runs before a launch that needs more than 64 KB of LDS; the first
// failure of the calling thread is kept until the C-ABI entry point collects it with take_launch_error() and returns
// NHANS_EHIP.  Nothing is launched silently wrong.
void note_launch(const char* kernel, hipError_t launch_rc);
bool launch_error_pending();                // a launch of this thread's current call has failed or been refused (not cleared)
void note_refusal(const char* what);        // a launch the library itself refuses: reported as hipErrorInvalidValue, no HIP state touched
// hipFuncAttributeMaxDynamicSharedMemorySize is a per-DEVICE attribute: `done_mask` (one static per
// kernel instantiation) has one bit per device id.
void set_max_dynamic_lds(const void* fn, size_t bytes, unsigned long long* done_mask, const char* kernel);
hipError_t take_launch_error(const char** kernel);
template <typename... P, size_t... I>
inline hipError_t launch_packed(void (*kernel)(P...), dim3 grid, dim3 block, size_t lds, hipStream_t s, std::tuple<std::decay_t<P>...>& params,
                                std::index_sequence<I...>) {
    void* ptrs[] = {static_cast<void*>(&std::get<I>(params))...};
    return hipLaunchKernel(reinterpret_cast<const void*>(kernel), grid, block, ptrs, lds, s);
}
template <typename... P, typename... A>
inline void launch_checked(const char* name, void (*kernel)(P...), dim3 grid, dim3 block, size_t lds, hipStream_t s, A&&... args) {
    static_assert(sizeof...(P) == sizeof...(A), "kernel argument count");
    std::tuple<std::decay_t<P>...> params(std::forward<A>(args)...);        // converted to the kernel's parameter types
    note_launch(name, launch_packed<P...>(kernel, grid, block, lds, s, params, std::index_sequence_for<P...>{}));
}
#define NHANS_LAUNCH(NAME, KERNEL, GRID, BLOCK, LDS, STREAM, ...)                                  \
    ::nhans::launch_checked(NAME, KERNEL, GRID, BLOCK, LDS, STREAM, __VA_ARGS__)

// bit set in *sat by the split-f16 writers when an activation does not fit f16 (|v| >= 65504 or NaN)
constexpr int kSatActivation = 1;
// conv_wino.hip re-splits V = BT d into f16: for the non-negative (post-ReLU) tensors it reads, |V| <= 8.5 max d -- the
// largest one-signed coefficient sum of a BT row of F(5,4) -- and the f16 conversion saturates SILENTLY, so the launch
// that writes such a tensor raises the flag at 65504 / 8.5 = 7,706 already (7,168 leaves room for the f32 rounding).
constexpr float kSatLimitF16 = 65504.f, kSatLimitWinoInput = 7168.f;

// Division by a runtime constant for numerators < 2^31 (Granlund-Montgomery round-up form).
struct FastDiv {
    uint32_t d, mul, sh;
};
inline FastDiv make_fastdiv(uint32_t d) {
    FastDiv f;
    f.d = d;
    uint32_t l = 0;
    while ((1ull << l) < d) ++l;
    f.mul = (uint32_t)((((uint64_t)1 << 32) * (((uint64_t)1 << l) - d)) / d + 1);
    f.sh = l;
    return f;
}
__device__ __forceinline__ uint32_t fd_div(uint32_t n, const FastDiv& f) {
    return (__umulhi(f.mul, n) + n) >> f.sh;
}

// A batch of 1-channel "images" that are SLIDING WINDOWS of one [rows, W] tensor: image b's row h is tensor row
// row0 + b + h when 0 <= t[b] + h - pad < T[b] (t: the frame's position in its clip, T: the clip's length), and a row of
// 0.0 otherwise -- strided_crop of SN/apply.py:170-186,378 (35-row windows of the log-magnitude spectrogram, zero rows
// -- not the silence floor -- outside the clip) without ever materialising [T, 35, 201].
// t == nullptr: plain images [B, H, W].
struct WinRows {
    const int* t;
    const int* T;
    int row0, pad;      // (row0 may be negative: nhans_api.hip counts rows from the launch's first frame, row0 = -pad)
};
constexpr int kNoRow = -2147483647 - 1;      // "this row is a zero row" where an element index is stored (conv_epilogue.h)
__device__ __forceinline__ bool win_row_ok(const WinRows& w, int b, int h) {
    return (unsigned)(w.t[b] + h - w.pad) < (unsigned)w.T[b];
}

// ---------------------------------------------------------------------------------------------
// Implicit-GEMM convolution on the f32 matrix cores (conv_igemm.hip).
//   out[m, n] = epilogue( sum_seg sum_{kh,kw,c} src_seg[b, ho*sh+kh-pt, wo*sw+kw-pl, c] * W_seg[kh,kw,c,n] )
// with m = (b*Ho + ho)*Wo + wo.  Up to two K segments: the kxk taps of the block's main conv and
// (for blocks that change channel count) the 1x1 strided `_transform` conv folded in as extra K.
struct ConvSeg {
    const float* src;   // NHWC [B, H, W, C]
    const float* wpk;   // packed weights [chunk][N/32][4][64][4] (see fold.py: pack_igemm)
    int H, W, C;
    int KH, KW, sh, sw, pt, pl;
    int nchunks;        // KH*KW*C/32
};

struct ConvArgs {
    ConvSeg seg[2];
    int nseg;
    int Ho, Wo;
    int M;              // B*Ho*Wo
    int N;              // padded output channels (multiple of the N tile)
    int Nreal;          // stored channels
    int ldo;            // out row stride
    float* out;
    // epilogue: v = acc + cb[clip(b)*cb_stride + n] + tf[(ho*Wo+wo)*N + n]
    //           aux[m*aux_ld+n] = v (optional)
    //           v += idw[n]*id[m*id_ld+n]                       (id_mode 1: same-shape tensor)
    //           v += idw[n]*ids[(b*idH + ho*idsh)*idW + wo*idsw] (id_mode 2: 1-channel image)
    //           out = relu ? max(v,0) : v
    const float* cb;
    int cb_stride;
    const int* img_clip;   // nullable -> clip 0
    const float* tf;       // nullable: time+frequency position table [Ho*Wo, N]
    const float* tt;       // the same table's two terms on their own, [Ho, N] and [Wo, N] (conv_wino.hip reads these:
    const float* ff;       // 60 KB instead of 1.8 MB per layer); nullable together
    int id_mode;
    const float* id;
    int id_ld;
    const float* idw;
    int idH, idW, idsh, idsw;
    WinRows id_win;        // id_mode 2: the image is a sliding window of `id` (idH is then unused)
    int relu;
    float* aux;
    int aux_ld;
    const float* zero;     // one readable 0.0f (stands in for absent tables / residuals)
    int prec;              // 0: f32 MFMA on f32 NHWC; 1: split-f16 x3 MFMA on split NHWC inputs
    int out_split;         // write `out` as split NHWC (ldo = N words per pixel) instead of f32
    int id_split;          // id_mode 1 tensor is split NHWC
    int in_f32;            // prec 1 only: seg[0].src is f32 NHWC (scaled like a split tensor) -- a tensor that only Winograd
                           // launches read (conv_wino.hip, the one kernel that takes it)
    const float* ws;       // prec 1: per-channel power-of-two that undoes the weight pre-scaling
    // Split-f16 tensors are STORED times a per-tensor power of two 2^-e (nhans_api.hip: activation exponents), so that
    // what a trained or an odd model produces stays inside the f16 range.  The epilogue computes in the unscaled
    // domain, bit for bit what it computes with e = 0: ws is multiplied by in_scale = 2^e(input), idw by id_scale =
    // 2^e(residual), and the result by out_scale = 2^-e(output) on its way to memory.  All three are 1 for f32 tensors.
    float in_scale, id_scale, out_scale;
    // the flag is raised by a stored |value| >= sat_limit: 65504 (the f16 range), or kSatLimitWinoInput for a tensor the
    // next launch reads in its Winograd form
    float sat_limit;
    int* sat;              // prec 1: device flag word, kSatActivation is OR-ed in when a stored activation saturates
    int variant;           // 0: 128-pixel / 4-wave register-staged kernel, 1: 256-pixel / 8-wave LDS-DMA kernel,
                           // 2: LDS-DMA kernels with producer / consumer waves and halo reuse across the KW taps
    int epi8;              // split-NHWC outputs: 8 channels per thread in the epilogue sweep (16-byte pieces), default;
                           // 0 = the 4-channel sweep (A/B and parity cross-check; same bits)
    int ilv;               // halo kernel: operand reads between the MFMAs, front-loaded in each half (default 1); 2 = spread
                           // evenly over the half; 0 = read block then MFMA block (the round-1 order) -- A/B knob, same bits
    int halo64_tile512;    // halo kernel, 64-channel convs: 512-pixel tiles (set by the launcher)
    long long* dbg;        // NHANS_DEV builds only: 4 s_memtime stamps per workgroup [start, loop, epilogue, end]
    FastDiv fdHoWo, fdWo;
    FastDiv fdWP;          // Wo + KW - 1 (filled in by launch_conv_igemm_halo)
    // split-K scratch (conv_igemm_dma.hip; null = never split): partial accumulator tiles and one
    // zero-initialised ticket counter per output tile
    float* kscratch;
    size_t kscratch_bytes;
    int* kcounter;
    int kcounter_n;
    int kgroup;            // caller: -1 = this layer may use grouped summation / split-K, 0 = never;
                           // the launcher turns -1 into the chunks per group
    // 1-D Winograd along W (conv_wino.hip).  Caller: wino_u / wino_ws (null = this conv has no Winograd form) and
    // wino on/off; the launcher fills in the rest: outputs per tile, tile-pixel block (rows x tiles, <= 64), blocks
    // per frame, tiles per image row
    const float* wino_u;
    const float* wino_ws;
    int wino;
    int wino_m, wino_tr, wino_tj, wino_nrb, wino_ncb, wino_ntile;
    FastDiv wino_fd_bpf, wino_fd_nnb, wino_fd_ncb;   // blocks per frame, channel blocks, column blocks (block decode)
};

// returns algorithmic FLOPs of the launch (2*M*K*Nreal); *kernel (optional) names the variant that ran
// *mfma_flops (optional): FLOPs the matrix cores actually execute for it -- 3 products per MAC in split-f16 mode,
// and 8*KH instead of m*KW*KH products per tile for a conv that runs in its Winograd form
double launch_conv_igemm(const ConvArgs& a, hipStream_t s, const char** kernel = nullptr, double* mfma_flops = nullptr);
double conv_wino_mfma_flops(const ConvArgs& a);                 // conv_wino.hip
void launch_conv_igemm_dma(const ConvArgs& a, hipStream_t s);   // conv_igemm_dma.hip
size_t conv_splitk_scratch_bytes(const ConvArgs& a);            // conv_igemm_dma.hip: scratch its split-K form would need (0: no split)
bool conv_wino_eligible(const ConvArgs& a);                     // conv_wino.hip
void launch_conv_wino(const ConvArgs& a, hipStream_t s);
bool conv_igemm_halo_eligible(const ConvArgs& a);               // conv_igemm_halo.hip
void launch_conv_igemm_halo(const ConvArgs& a, hipStream_t s);
bool conv_igemm_halo_pw_eligible(const ConvArgs& a);            // the same pipeline without halo reuse (strided / VALID convs)
void launch_conv_igemm_halo_pw(const ConvArgs& a, hipStream_t s);

// ---------------------------------------------------------------------------------------------
// The stand-alone 1x1 strided `_transform` conv of resblock2_1 as a stream (conv_1x1_stream.hip): 64 split-NHWC channels in,
// 128 f32 channels out, out = (x W) * ws * in_scale -- the arithmetic of conv_igemm_dma.hip for that launch, bit for bit
struct Stream1x1Args {
    const float* src;      // split NHWC [B, H, W, 64]
    const float* wpk;      // fold.py pack_igemm_h3 of the [64, 128] matrix
    const float* ws;       // [128] per-channel power of two that undoes the weight pre-scaling
    float in_scale;
    float* out;            // f32 [M, 128]
    int H, W, sh, sw, M;
    FastDiv fdHoWo, fdWo;
};
bool conv_1x1_stream_eligible(const ConvArgs& t);
void launch_conv_1x1_stream(const ConvArgs& t, hipStream_t s);

// Small kernels (aux_kernels.hip)
struct DirectArgs {     // convolution of a 1-channel image into 64 channels, same epilogue terms
    const float* src;   // [B, H, W] -- or, win.t != nullptr, the [rows, W] tensor the images are sliding windows of
    WinRows win;
    const float* w;     // [KH*KW][64], BN scale folded in
    int H, W, KH, KW, sh, sw, pt, pl, Ho, Wo;
    int M;              // B*Ho*Wo
    float* out;         // [M, 64]
    const float* cb;
    int cb_stride;
    const int* img_clip;
    const float* tf;    // [Ho*Wo,64] nullable
    const float* tt;    // its two terms [Ho,64], [Wo,64]: used instead of tf when present (v = (acc + cb + tt) + ff)
    const float* ff;
    int relu;
    int out_split;      // write split NHWC (hi/lo f16) instead of f32
    float out_scale;    // as ConvArgs::out_scale
    float sat_limit;    // as ConvArgs::sat_limit
    int* sat;           // as ConvArgs::sat
    FastDiv fdHoWo, fdWo;
};
void launch_direct_conv64(const DirectArgs& a, hipStream_t s);
// split NHWC [M, C] -> f32 [M, C], times `scale`
void launch_unsplit(const float* src, int64_t M, int C, float scale, float* dst, hipStream_t s);
void launch_scale_copy(const float* src, size_t n, float scale, float* dst, hipStream_t s);   // dst = src * scale (an f32-stored tensor out of its exponent)
// *slot = max(*slot, scale * max|x|) over a tensor of `nwords` 32-bit words (f32 values, or pairs of f16 halfs of a
// split-NHWC tensor: the hi halfs dominate); *slot holds the bits of a non-negative float.  Calibration only.
void launch_absmax(const float* x, size_t nwords, int split, float scale, unsigned* slot, hipStream_t s);

// frame index: for global frame g -> clip, t within clip, T of clip
void launch_frame_index(const int64_t* frame_offsets_dev, int nclips, int64_t total, int* f_clip,
                        int* f_t, int* f_T, hipStream_t s);
// mean over HW positions, times `scale`: x [B, HW, C] -> out [B, C]
void launch_avgpool(const float* x, int B, int HW, int C, int split, float scale, float* out, hipStream_t s);
// cb[clip, n] = base[n] + sum_k ea[clip,k]*Wc[k, n] + sum_k eb[clip,k]*Wc[512+k, n]
void launch_cond(const float* ea, const float* eb, int nclips, const float* Wc, const float* base,
                 int ncols, float* cb, hipStream_t s);

// ---------------------------------------------------------------------------------------------
// STFT / iSTFT (stft.hip)
struct ClipTable {      // device arrays, one entry per clip
    const int64_t* sample_off;   // [nclips+1] offsets into the wav buffer
    const int64_t* frame_off;    // [nclips+1] offsets into [T_total,201] tensors
    const int64_t* out_off;      // [nclips+1] (iSTFT) offsets into the output wav buffer
};
// grid = one block per (clip, run of kStftFramesPerBlock frames); block_clip/block_f0 enumerate them
void launch_stft(const float* wav, ClipTable t, const int* block_clip, const int* block_f0, int nblocks,
                 const float* tw400 /*400 cplx*/, const float* window /*400*/, float* logmag,
                 float* phase, hipStream_t s);
void launch_istft(const float* logmag, const float* phase, ClipTable t, const int* block_clip,
                  const int* block_h0, int nblocks, const float* tw400, const float* wsyn /*400*/,
                  float* wav_out, hipStream_t s);
constexpr int kStftFramesPerBlock = 23;   // frames per run: 460 pass-1 tasks (256 + 204) and 253 pass-2 tasks on 256 lanes
constexpr int64_t kMaxFramesPerClip = 5000000;   // 13.9 h: the STFT / iSTFT kernels address a clip with 32-bit byte offsets
constexpr int kIstftHopsPerBlock = 22;    // output hops per block; needs 24 frames

}  // namespace nhans
