// C ABI of libnhans_hip.so (see include/nhans_hip.h): context, folded-weight blob, workspace and
// the launch sequences of the N-HANS hot path.
#include "../../include/nhans_hip.h"
#include "nhans_kernels.h"

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <string>
#include <vector>

using namespace nhans;

namespace {

thread_local std::string g_err;

int fail(int code, const std::string& msg) {
    g_err = msg;
    return code;
}

#define HIP_TRY(expr)                                                                          \
    do {                                                                                       \
        hipError_t e_ = (expr);                                                                \
        if (e_ != hipSuccess)                                                                  \
            return fail(NHANS_EHIP, std::string(#expr) + ": " + hipGetErrorString(e_));        \
    } while (0)

// ---- folded blob -----------------------------------------------------------------------------
constexpr uint32_t kBlobVersion = 2;   // fold.py: BLOB_VERSION
struct BlobHeader {
    char magic[8];          // "NHANSFW1"
    uint32_t version;
    uint32_t n_entries;
    uint64_t total_bytes;
};
struct BlobEntry {
    char name[48];
    uint64_t offset;        // bytes from blob start, 256-byte aligned
    uint64_t nfloats;
};

struct BlockGeo {
    int kh, kw, sh, sw, cin, cout, hin, win, hout, wout;
};

void same_pad(int n, int k, int s, int* out, int* before) {
    *out = (n + s - 1) / s;
    int total = std::max((*out - 1) * s + k - n, 0);
    *before = total / 2;
}

std::vector<BlockGeo> tower_geometry() {       // SN/main.py:194-198
    const int kh[4] = {8, 8, 4, 4}, kw[4] = {4, 4, 4, 4}, sh[4] = {3, 3, 1, 1}, sw[4] = {2, 2, 1, 2};
    const int co[4] = {64, 128, 256, 512};
    std::vector<BlockGeo> v;
    int h = kCtxFrames, w = kBins, c = 1;
    for (int i = 0; i < 4; ++i) {
        BlockGeo g{kh[i], kw[i], sh[i], sw[i], c, co[i], h, w, (h + sh[i] - 1) / sh[i], (w + sw[i] - 1) / sw[i]};
        v.push_back(g);
        h = g.hout; w = g.wout; c = g.cout;
    }
    return v;
}

std::vector<BlockGeo> main_geometry() {        // SN/main.py:221-229
    const int k[8] = {4, 4, 4, 4, 3, 3, 3, 3}, s[8] = {1, 1, 2, 1, 2, 1, 2, 1};
    const int co[8] = {64, 64, 128, 128, 256, 256, 512, 512};
    std::vector<BlockGeo> v;
    int h = kMixWin, w = kBins, c = 1;
    for (int i = 0; i < 8; ++i) {
        BlockGeo g{k[i], k[i], s[i], s[i], c, co[i], h, w, (h + s[i] - 1) / s[i], (w + s[i] - 1) / s[i]};
        v.push_back(g);
        h = g.hout; w = g.wout; c = g.cout;
    }
    return v;
}

constexpr int kNumAct = NHANS_NUM_ACTIVATIONS;
constexpr int kActTargetLog2 = 8;
constexpr int TA(int b, int j) { return 2 * b + j; }            // tower block b, conv j+1
constexpr int SA(int b, int j) { return 8 + 2 * b + j; }        // stack block b, conv j+1
constexpr int kActHead = 24;                                    // last_conv

struct ProfEntry {
    int calls = 0;
    double flops = 0, bytes = 0, mfma = 0;      // algorithmic FLOPs / bytes; FLOPs the matrix cores executed
    std::vector<std::pair<hipEvent_t, hipEvent_t>> pending;
    double ms = 0;
};

}  // namespace

// which convs of the stack ran in their Winograd form in the last chunk (run_stack_chunk)
struct StackPlan {
    bool wino[8][3] = {};      // [block][conv 1 | 2]
};

struct nhans_ctx {
    int kind = 0, device = 0;
    StackPlan last_plan;
    float* blob_dev = nullptr;
    size_t blob_bytes = 0;
    std::map<std::string, const float*> arr;
    std::map<std::string, size_t> arr_n;
    std::vector<BlockGeo> tower, stack;
    int cond_cols = 0;
    std::vector<int> cond_off;      // column offset of conv j (= 2*block + {0,1})
    // workspace
    char* ws = nullptr;
    size_t ws_bytes = 0, ws_top = 0;
    // split-K scratch of the conv kernel (small launches only)
    // Allocated LAZILY, sized by the launches that actually split (run_conv: conv_splitk_scratch_bytes) and grown up to
    // 96 tiles x 32 groups x 128 KB = 384 MB, all the split-K rule of conv_igemm_dma.hip admits (round-5 advisor: 384 MB
    // taken unconditionally at nhans_create was 3 GB for eight ranks sharing a device, and its failure failed the
    // create).  A failed allocation is not an error: the launch walks its groups unsplit -- same bits, fewer CUs.
    float* kscratch = nullptr;
    size_t kscratch_bytes = 0;
    static constexpr size_t kscratch_cap = (size_t)384 << 20;
    bool kscratch_failed = false;
    int stream_1x1 = 1;         // option stream_1x1: the stand-alone `_transform` conv on conv_1x1_stream.hip (0: the generic conv kernel; same bits)
    int split_k = 1;            // option split_k: 0 = never split (the grouped walk inside one workgroup: same bits)
    int* kcounter = nullptr;
    int kcounter_n = 1024;
    // Frame windows per pass of the stack.  Every launch runs whole "waves" of one workgroup per CU and all
    // workgroups of a launch take the same time, so a launch whose tile count is not a multiple of 256
    // leaves CUs idle for a tile time at its end: 1,024 frames give resblock4 (130 pixels per frame, 256-pixel
    // x 4 channel tiles) 8.1 waves = 9.7 % lost, 3.9 % over the whole stack.  3,776 = 59 x 64 frames minimise
    // the FLOP-weighted loss (0.18 %) among the sizes whose largest tensor (3,776 x 35 x 201 x 64 elements)
    // still fits the kernels' 32-bit element offsets; the three ping-pong buffers are then 20 GB of the 288.
    int64_t frames_per_chunk = 3776;
    int contexts_per_chunk = 64;
    // pinned staging ring for the small host tables (offsets, block lists) copied per call
    char* pin = nullptr;
    size_t pin_bytes = (size_t)16 << 20, pin_top = 0;
    // profiling
    bool profile = false;
    std::map<std::string, ProfEntry> prof;
    std::vector<hipEvent_t> event_pool;

    int prec = 0;           // 0: f32 MFMA, 1: split-f16 x3 MFMA (activations in split NHWC)
    int conv_variant = -1;  // 0: 128-pixel register-staged conv kernel, 1: 256-pixel LDS-DMA kernel,
                            // 2: halo-reuse / wave-specialised LDS-DMA kernel where the conv allows it, else 1;
                            // -1: automatic (measured best: 2 for split-f16, register-staged for f32)
    int epi8 = 1;               // ConvArgs::epi8
    int ilv = 1;                // ConvArgs::ilv
    int wino = 1;               // ConvArgs::wino: 1-D Winograd form of the stride-1 stack convs (conv_wino.hip)
    int wino_f32 = 1;           // tensors that only Winograd launches read are stored f32 NHWC (stored_f32())
    long long* dbg = nullptr;   // NHANS_DEV builds: per-workgroup cycle stamps of the last conv launch
    int* status_dev = nullptr;  // sticky NHANS_STATUS_* bits set by kernels (nhans_take_status)
    // Activation exponents: a split-f16 tensor is stored as x * 2^-e with one e per tensor of the network, chosen from
    // the largest |x| a calibration pass saw so that the stored maximum is <= 2^kActTargetLog2 -- 2^8 below the f16
    // limit (and the 1-D Winograd transform's worst-case gain of ~20 still fits).  Tensors: tower block b conv1/conv2
    // outputs (2b, 2b+1), stack block b conv1/conv2 outputs (8+2b, 8+2b+1), last_conv output (24).  f32 tensors carry
    // no exponent.  The flag of nhans_take_status stays as the backstop for inputs far outside the calibration.
    int act_exp[kNumAct] = {};
    float act_amax[kNumAct] = {};       // what the last calibration saw (diagnostics)
    unsigned* amax_dev = nullptr;       // running maxima (float bits) while calibrating
    bool calibrating = false;
    float up(int i) const { return prec ? ldexpf(1.f, act_exp[i]) : 1.f; }
    float down(int i) const { return prec ? ldexpf(1.f, -act_exp[i]) : 1.f; }
    // ordering of consecutive calls that share the workspace (see include/nhans_hip.h)
    hipEvent_t tail_ev = nullptr;
    hipStream_t last_stream = nullptr;
    bool have_tail = false;

    const float* A(const std::string& n) const {
        auto it = arr.find(n);
        return it == arr.end() ? nullptr : it->second;
    }
    // packed conv weights / per-channel unscale vector of the active precision
    const float* WP(const std::string& n) const { return A(prec ? n + "_h" : n); }
    const float* WS(const std::string& conv) const { return prec ? A(conv + ".ws") : nullptr; }
};

namespace {

int ws_reserve(nhans_ctx* c, size_t bytes) {
    if (bytes <= c->ws_bytes) { c->ws_top = 0; return NHANS_OK; }
    if (c->ws) { HIP_TRY(hipDeviceSynchronize()); HIP_TRY(hipFree(c->ws)); c->ws = nullptr; c->ws_bytes = 0; }
    hipError_t e = hipMalloc(reinterpret_cast<void**>(&c->ws), bytes);
    if (e != hipSuccess) {
        char buf[128];
        snprintf(buf, sizeof buf, "workspace hipMalloc(%zu bytes) failed: %s", bytes, hipGetErrorString(e));
        return fail(NHANS_ENOMEM, buf);
    }
    c->ws_bytes = bytes;
    c->ws_top = 0;
    return NHANS_OK;
}

// Host -> device copy of a small table through the pinned ring, so the caller's (pageable, soon
// destroyed) buffer is never the source of an in-flight asynchronous copy.
int h2d(nhans_ctx* c, void* dst, const void* src, size_t bytes, hipStream_t s) {
    if (bytes == 0) return NHANS_OK;
    const size_t need = (bytes + 63) & ~(size_t)63;
    if (need > c->pin_bytes) {
        HIP_TRY(hipMemcpyAsync(dst, src, bytes, hipMemcpyHostToDevice, s));
        HIP_TRY(hipStreamSynchronize(s));
        return NHANS_OK;
    }
    if (c->pin_top + need > c->pin_bytes) {
        HIP_TRY(hipStreamSynchronize(s));
        c->pin_top = 0;
    }
    void* p = c->pin + c->pin_top;
    c->pin_top += need;
    std::memcpy(p, src, bytes);
    HIP_TRY(hipMemcpyAsync(dst, p, bytes, hipMemcpyHostToDevice, s));
    return NHANS_OK;
}

template <typename T> T* ws_take(nhans_ctx* c, size_t count) {
    size_t bytes = (count * sizeof(T) + 255) & ~(size_t)255;
    T* p = reinterpret_cast<T*>(c->ws + c->ws_top);
    c->ws_top += bytes;
    return p;
}
size_t ws_size(size_t count, size_t elem) { return (count * elem + 255) & ~(size_t)255; }

// RAII-less profiling bracket around one launch
struct Prof {
    nhans_ctx* c;
    hipStream_t s;
    ProfEntry* e = nullptr;
    hipEvent_t a{}, b{};
    static hipEvent_t take(nhans_ctx* c) {
        hipEvent_t ev = nullptr;
        if (!c->event_pool.empty()) { ev = c->event_pool.back(); c->event_pool.pop_back(); }
        else (void)hipEventCreate(&ev);
        return ev;
    }
    Prof(nhans_ctx* c_, hipStream_t s_, const char* name) : c(c_), s(s_) {
        if (!c->profile) return;
        if (name) e = &c->prof[name];
        a = take(c);
        b = take(c);
        (void)hipEventRecord(a, s);
    }
    void done(double flops, double bytes, const char* late_name = nullptr, double mfma = 0) {
        if (!c->profile) return;
        if (late_name) e = &c->prof[late_name];
        if (!e) return;
        (void)hipEventRecord(b, s);
        e->pending.emplace_back(a, b);
        e->calls += 1;
        e->flops += flops;
        e->bytes += bytes;
        e->mfma += mfma;
    }
};

void fill_epilogue_defaults(nhans_ctx* c, ConvArgs& a) {
    a.zero = c->A("zero");
    a.sat = c->status_dev;
    a.img_clip = nullptr; a.tf = nullptr; a.tt = nullptr; a.ff = nullptr; a.id_mode = 0; a.id = nullptr; a.id_ld = 0;
    a.idw = nullptr; a.idH = a.idW = 0; a.idsh = a.idsw = 1; a.relu = 1; a.aux = nullptr; a.aux_ld = 0;
    a.cb_stride = 0;
    a.prec = c->prec; a.out_split = c->prec; a.id_split = 0; a.ws = nullptr;
    a.in_scale = a.id_scale = a.out_scale = 1.f;
    a.sat_limit = kSatLimitF16;
    a.variant = c->conv_variant >= 0 ? c->conv_variant : (c->prec == 1 ? 2 : 0);
    a.dbg = kDev ? c->dbg : nullptr;
    a.epi8 = c->epi8;
    a.ilv = c->ilv;
    a.wino = c->wino; a.wino_u = nullptr; a.wino_ws = nullptr;
    a.kscratch = c->kscratch; a.kscratch_bytes = c->kscratch_bytes; a.kcounter = c->kcounter; a.kcounter_n = c->kcounter_n; a.kgroup = 0;
}

ConvSeg make_seg(const float* src, const float* wpk, int H, int W, int C, int KH, int KW, int sh, int sw,
                 bool same) {
    ConvSeg g;
    g.src = src; g.wpk = wpk; g.H = H; g.W = W; g.C = C; g.KH = KH; g.KW = KW; g.sh = sh; g.sw = sw;
    int o, pb;
    if (same) { same_pad(H, KH, sh, &o, &pb); g.pt = pb; same_pad(W, KW, sw, &o, &pb); g.pl = pb; }
    else { g.pt = 0; g.pl = 0; }
    g.nchunks = KH * KW * C / 32;
    return g;
}

void set_out_geometry(ConvArgs& a, int B, int Ho, int Wo, int N, int Nreal, int ldo, float* out) {
    a.Ho = Ho; a.Wo = Wo; a.M = B * Ho * Wo; a.N = N; a.Nreal = Nreal; a.ldo = ldo; a.out = out;
    a.fdHoWo = make_fastdiv((uint32_t)(Ho * Wo));
    a.fdWo = make_fastdiv((uint32_t)Wo);
}

void run_conv(nhans_ctx* c, const ConvArgs& a0, hipStream_t s) {
    ConvArgs a = a0;
    if (kDev) {      // timing experiment (wrong results): NHANS_ABLATE_TF=1 -> no position table at all
        static const bool no_tf = [] { const char* e = getenv("NHANS_ABLATE_TF"); return e && atoi(e) != 0; }();
        if (no_tf) { a.tf = nullptr; a.tt = nullptr; a.ff = nullptr; }
    }
    if (a.kgroup < 0) {
        // split-K scratch on demand (hipFree / hipMalloc wait for the device: a handful of times per context at most)
        const size_t need = c->split_k ? conv_splitk_scratch_bytes(a) : 0;
        if (need > c->kscratch_bytes && need <= nhans_ctx::kscratch_cap && !c->kscratch_failed) {
            if (c->kscratch) { (void)hipFree(c->kscratch); c->kscratch = nullptr; c->kscratch_bytes = 0; }
            const size_t want = std::min(nhans_ctx::kscratch_cap, std::max(need, (size_t)32 << 20));
            if (hipMalloc(reinterpret_cast<void**>(&c->kscratch), want) == hipSuccess) c->kscratch_bytes = want;
            else { (void)hipGetLastError(); c->kscratch = nullptr; c->kscratch_failed = true; }
        }
        a.kscratch = c->split_k ? c->kscratch : nullptr;
        a.kscratch_bytes = c->kscratch_bytes;
    }
    // profiled under the name of the kernel variant that ran (the variant is chosen per layer)
    Prof p(c, s, nullptr);
    const char* name = "conv_igemm";
    double mfma = 0;
    double fl = launch_conv_igemm(a, s, &name, &mfma);
    p.done(fl, 0, name, mfma);
}

// Which convs of the stack run in their Winograd form (conv_wino.hip), for one chunk of frame windows.  NOT a restatement
// of the kernel's conditions: run_stack_chunk() builds every launch's ConvArgs twice -- a planning pass that asks
// conv_wino_eligible() about those very arguments, then the launching pass that takes layouts and saturation limits
// from the answers (round-4 advisor finding: a second predicate that left out the 32-bit offset bound, aux, kgroup ...
// could disagree with the kernel, and the producer would already have written the other layout).
bool wino_form(const ConvArgs& a) { return a.variant >= 2 && a.kgroup >= 0 && conv_wino_eligible(a); }
float sat_limit_for(const StackPlan& p, int b, int cv) { return b >= 0 && b < 8 && p.wino[b][cv] ? kSatLimitWinoInput : kSatLimitF16; }

// Is stack tensor (block b; cv 0: conv1's output, 1: the block's output) stored as f32 NHWC in the split-f16 mode?  Yes if
// every launch that reads it is a Winograd launch -- conv_wino.hip reads either layout (its transform works in f32 and
// re-splits: with an f32 input it has no hi + lo to add up, 64 of its ~215 instructions per chunk), the direct kernels
// stage split pieces straight into MFMA operands -- and the launch that writes it is direct_conv64 or a Winograd launch.
// The values are the same scaled, clamped ones a split store would hold to 22 bits; 4 bytes per element either way.
// (wino_f32 == 2, a test value: f32 whatever the readers are -- launch_conv_igemm() must then refuse the reader.
//  wino_f32 == 3, a test value: ONLY the output of resblock1_2 is f32 -- its conv2 then has a split residual and an f32
//  output, the one layout pair conv_wino's epilogue does not implement: launch_conv_wino() must refuse it.)
bool stored_f32(const nhans_ctx* c, const StackPlan& p, int b, int cv) {
    if (c->prec != 1 || !c->wino_f32 || b < 0 || b > 7) return false;
    if (c->wino_f32 == 2) return b < 4 && !(b == 3 && cv == 1);
    if (c->wino_f32 == 3) return b == 1 && cv == 1;
    if (cv == 0) return p.wino[b][2] && (b == 0 || p.wino[b][1]);
    if (b == 7) return false;
    const BlockGeo& nx = c->stack[b + 1];             // read by conv1 of the next block and, in an identity block, by its conv2's epilogue
    return p.wino[b][2] && p.wino[b + 1][1] && nx.cin == nx.cout && p.wino[b + 1][2];
}

// Calibration tap: the running |x| maximum of tensor `idx` (`words` values, stored in the active precision's layout).
void tap(nhans_ctx* c, int idx, const float* buf, size_t words, hipStream_t s, bool f32_layout = false) {
    if (c->calibrating) launch_absmax(buf, words, c->prec && !f32_layout, c->up(idx), c->amax_dev + idx, s);
}

// ---- embedding tower for `n` context images already in HBM ----------------------------------
int embed_impl(nhans_ctx* c, const float* ctx_lm, int n, float* emb_out, float* X, float* Ab, float* Y,
               hipStream_t s) {
    const auto& T = c->tower;
    for (int i0 = 0; i0 < n; i0 += c->contexts_per_chunk) {
        const int nc = std::min(c->contexts_per_chunk, n - i0);
        const float* img = ctx_lm + (size_t)i0 * kCtxFrames * kBins;
        float *x = X, *a1 = Ab, *y = Y;
        for (int b = 0; b < 4; ++b) {
            const BlockGeo& g = T[b];
            const std::string p = "t" + std::to_string(b);
            if (b == 0) {
                DirectArgs d{};
                d.src = img; d.w = c->A(p + ".c1.w"); d.H = g.hin; d.W = g.win; d.KH = g.kh; d.KW = g.kw;
                d.sh = g.sh; d.sw = g.sw;
                int o; same_pad(g.hin, g.kh, g.sh, &o, &d.pt); same_pad(g.win, g.kw, g.sw, &o, &d.pl);
                d.Ho = g.hout; d.Wo = g.wout; d.M = nc * g.hout * g.wout; d.out = a1;
                d.cb = c->A(p + ".c1.cb"); d.cb_stride = 0; d.img_clip = nullptr; d.tf = nullptr; d.tt = nullptr; d.ff = nullptr;
                d.relu = 1; d.fdHoWo = make_fastdiv(g.hout * g.wout); d.fdWo = make_fastdiv(g.wout);
                d.out_split = c->prec; d.sat = c->prec ? c->status_dev : nullptr; d.out_scale = c->down(TA(0, 0)); d.sat_limit = kSatLimitF16;
                Prof pr(c, s, "direct_conv64");
                launch_direct_conv64(d, s);
                pr.done(2.0 * d.M * g.kh * g.kw * 64, 0);
            } else {
                ConvArgs a{};
                fill_epilogue_defaults(c, a);
                a.nseg = 1;
                a.seg[0] = make_seg(x, c->WP(p + ".c1.wpk"), g.hin, g.win, g.cin, g.kh, g.kw, g.sh, g.sw, true);
                set_out_geometry(a, nc, g.hout, g.wout, g.cout, g.cout, g.cout, a1);
                a.cb = c->A(p + ".c1.cb");
                a.ws = c->WS(p + ".c1");
                a.in_scale = c->up(TA(b - 1, 1)); a.out_scale = c->down(TA(b, 0));
                a.kgroup = -1;                  // a handful of context images: grouped sum, split-K when small
                run_conv(c, a, s);
            }
            tap(c, TA(b, 0), a1, (size_t)nc * g.hout * g.wout * g.cout, s);
            ConvArgs a{};
            fill_epilogue_defaults(c, a);
            a.nseg = 1;
            a.seg[0] = make_seg(a1, c->WP(p + ".c2.wpk"), g.hout, g.wout, g.cout, g.kh, g.kw, 1, 1, true);
            if (b == 0) {
                a.id_mode = 2; a.id = img; a.idH = g.hin; a.idW = g.win; a.idsh = g.sh; a.idsw = g.sw;
                a.idw = c->A(p + ".c2.idw");
            } else {
                a.nseg = 2;
                a.seg[1] = make_seg(x, c->WP(p + ".c2.wpk_t"), g.hin, g.win, g.cin, 1, 1, g.sh, g.sw, false);
            }
            set_out_geometry(a, nc, g.hout, g.wout, g.cout, g.cout, g.cout, y);
            a.cb = c->A(p + ".c2.cb");
            a.ws = c->WS(p + ".c2");
            // (the `_transform` segment reads x, whose exponent tie_exponents() keeps equal to a1's: one accumulator)
            a.in_scale = c->up(TA(b, 0)); a.out_scale = c->down(TA(b, 1));
            run_conv(c, a, s);                  // (stride 1: halo kernel; measured faster than split-K here)
            tap(c, TA(b, 1), y, (size_t)nc * g.hout * g.wout * g.cout, s);
            std::swap(x, y);
        }
        const BlockGeo& g = T[3];
        Prof pr(c, s, "avgpool");
        launch_avgpool(x, nc, g.hout * g.wout, g.cout, c->prec, c->up(TA(3, 1)), emb_out + (size_t)i0 * kEmb, s);
        pr.done(0, (double)nc * g.hout * g.wout * g.cout * 4);
    }
    return NHANS_OK;
}

size_t tower_buf_floats(const nhans_ctx* c) {
    size_t m = 0;
    for (const auto& g : c->tower) m = std::max(m, (size_t)g.hout * g.wout * g.cout);
    return m * (size_t)c->contexts_per_chunk;
}
size_t stack_buf_floats(const nhans_ctx* c, int64_t wf) {
    size_t m = 0;
    for (const auto& g : c->stack) m = std::max(m, (size_t)g.hout * g.wout * g.cout);
    return m * (size_t)wf;
}

// ---- conditioned stack + head ---------------------------------------------------------------
struct StackBufs {
    int* f_clip; int* f_t; int* f_T; int64_t* foff_dev; float* cb_all;
    float* X; float* A; float* Y;
    float* T;       // f32 output of a block's 1x1 `_transform` conv when its conv2 runs in Winograd form
};

// floats per frame window of StackBufs::T: the largest conv2 output among the channel-changing blocks whose conv2
// has a Winograd form (4x4 filters: resblock2_1)
size_t transform_buf_floats(const nhans_ctx* c, int64_t wf) {
    size_t m = 0;
    for (const auto& g : c->stack)
        if (g.cin != g.cout && g.cin > 1 && g.kh == 4) m = std::max(m, (size_t)g.hout * g.wout * g.cout);
    return m * (size_t)wf;
}

size_t stack_ws_bytes(const nhans_ctx* c, int64_t total, int nclips, int64_t wf) {
    size_t b = 3 * ws_size(total, 4) + ws_size(nclips + 1, 8) + ws_size((size_t)nclips * c->cond_cols, 4);
    b += 3 * ws_size(stack_buf_floats(c, wf), 4);
    b += ws_size(transform_buf_floats(c, wf), 4);
    return b;
}

void stack_take(nhans_ctx* c, int64_t total, int nclips, int64_t wf, StackBufs* sb) {
    sb->f_clip = ws_take<int>(c, total); sb->f_t = ws_take<int>(c, total); sb->f_T = ws_take<int>(c, total);
    sb->foff_dev = ws_take<int64_t>(c, nclips + 1);
    sb->cb_all = ws_take<float>(c, (size_t)nclips * c->cond_cols);
    const size_t nb = stack_buf_floats(c, wf);
    sb->X = ws_take<float>(c, nb); sb->A = ws_take<float>(c, nb); sb->Y = ws_take<float>(c, nb);
    sb->T = ws_take<float>(c, transform_buf_floats(c, wf));
}

// Runs blocks [0, upto) for frames [g0, g0+n); returns the buffer holding the last output.
// upto = 8: whole stack; upto = 9: + last_conv (output in sb.A).
// Two passes over the same code: pass 0 builds every conv's arguments and records which of them the Winograd kernel
// accepts (StackPlan), pass 1 builds them again with the tensor layouts and saturation limits that follow from the plan
// and launches.  A launch whose eligibility differs between the passes is an error, not a fallback.
float* run_stack_chunk(nhans_ctx* c, const float* logmag, const StackBufs& sb, int64_t g0, int n, int upto,
                       hipStream_t s) {
    StackPlan plan;
    float* result = nullptr;
    const int* clipmap = sb.f_clip + g0;
    for (int pass = 0; pass < 2; ++pass) {
    const bool go = pass == 1;
    // (pass 0: record; pass 1: the launch must be the one that was planned)
    // (a launch that failed or was refused ends the chunk: nothing later may run on a buffer that was never written)
    auto dead = [&] { return go && launch_error_pending(); };
    auto conv = [&](int b, int cv, const ConvArgs& a) {
        const bool w = wino_form(a);
        if (!go) { plan.wino[b][cv] = w; return; }
        if (dead()) return;
        if (w != plan.wino[b][cv]) {
            note_refusal("stack conv whose Winograd eligibility changed between planning and launch");
            return;
        }
        run_conv(c, a, s);
    };
    // frame b's 35 x 201 image = rows g0 + b - 17 ... of the log-magnitude spectrogram, zero rows outside its clip
    // (SN/apply.py:170-186,378: strided_crop, never materialised -- the first conv and the 1 -> 64 residual of
    // resblock1_1 read the spectrogram where it lies)
    // (rows are counted from the chunk's first frame -- the tensor pointer handed to the kernels is logmag + g0 * 201 --, so the
    // kernels' 32-bit element indices stay below (frames_per_chunk + 35) * 201 however long the batch is)
    const WinRows win{sb.f_t + g0, sb.f_T + g0, -kCenter, kCenter};
    const float* const lm_chunk = logmag + (size_t)g0 * kBins;
    float *x = sb.X, *a1 = sb.A, *y = sb.Y;
    // pass 0 plans the WHOLE stack whatever `upto` is -- the layout of block b's output follows from block b + 1's
    // readers, and the debug entry point (upto = block + 1) must see the tensors the production call writes
    for (int b = 0; b < 8 && (b < upto || !go); ++b) {
        const BlockGeo& g = c->stack[b];
        const std::string p = "m" + std::to_string(b);
        const float* cb1 = sb.cb_all + c->cond_off[2 * b];
        const float* cb2 = sb.cb_all + c->cond_off[2 * b + 1];
        if (b == 0) {
            DirectArgs d{};
            d.src = lm_chunk; d.win = win; d.w = c->A(p + ".c1.w"); d.H = g.hin; d.W = g.win; d.KH = g.kh; d.KW = g.kw;
            d.sh = 1; d.sw = 1;
            int o; same_pad(g.hin, g.kh, 1, &o, &d.pt); same_pad(g.win, g.kw, 1, &o, &d.pl);
            d.Ho = g.hout; d.Wo = g.wout; d.M = n * g.hout * g.wout; d.out = a1;
            d.cb = cb1; d.cb_stride = c->cond_cols; d.img_clip = clipmap;
            d.tf = c->A(p + ".c1.tf"); d.tt = c->A(p + ".c1.tt"); d.ff = c->A(p + ".c1.ff");
            if (!d.tt || !d.ff) d.tt = d.ff = nullptr;
            d.relu = 1; d.out_split = c->prec && !stored_f32(c, plan, 0, 0); d.sat = c->prec ? c->status_dev : nullptr;
            d.out_scale = c->down(SA(0, 0)); d.sat_limit = sat_limit_for(plan, 0, 2);
            d.fdHoWo = make_fastdiv(g.hout * g.wout); d.fdWo = make_fastdiv(g.wout);
            if (go && !dead()) {
                Prof pr(c, s, "direct_conv64");
                launch_direct_conv64(d, s);
                pr.done(2.0 * d.M * g.kh * g.kw * 64, 0);
            }
        } else {
            ConvArgs a{};
            fill_epilogue_defaults(c, a);
            a.nseg = 1;
            a.seg[0] = make_seg(x, c->WP(p + ".c1.wpk"), g.hin, g.win, g.cin, g.kh, g.kw, g.sh, g.sw, true);
            set_out_geometry(a, n, g.hout, g.wout, g.cout, g.cout, g.cout, a1);
            a.cb = cb1; a.cb_stride = c->cond_cols; a.img_clip = clipmap;
            a.tf = c->A(p + ".c1.tf"); a.tt = c->A(p + ".c1.tt"); a.ff = c->A(p + ".c1.ff");
            a.ws = c->WS(p + ".c1");
            a.wino_u = c->A(p + ".c1.wino"); a.wino_ws = c->A(p + ".c1.wino.ws");
            a.in_scale = c->up(SA(b - 1, 1)); a.out_scale = c->down(SA(b, 0));
            a.sat_limit = sat_limit_for(plan, b, 2);
            a.in_f32 = stored_f32(c, plan, b - 1, 1); a.out_split = c->prec && !stored_f32(c, plan, b, 0);
            conv(b, 1, a);
        }
        if (go) tap(c, SA(b, 0), a1, (size_t)n * g.hout * g.wout * g.cout, s, stored_f32(c, plan, b, 0));
        ConvArgs a{};
        fill_epilogue_defaults(c, a);
        a.nseg = 1;
        a.seg[0] = make_seg(a1, c->WP(p + ".c2.wpk"), g.hout, g.wout, g.cout, g.kh, g.kw, 1, 1, true);
        a.cb = cb2; a.cb_stride = c->cond_cols; a.img_clip = clipmap;
        a.tf = c->A(p + ".c2.tf"); a.tt = c->A(p + ".c2.tt"); a.ff = c->A(p + ".c2.ff");
        a.idw = c->A(p + ".c2.idw");
        a.ws = c->WS(p + ".c2");
        a.wino_u = c->A(p + ".c2.wino"); a.wino_ws = c->A(p + ".c2.wino.ws");
        a.in_scale = c->up(SA(b, 0)); a.out_scale = c->down(SA(b, 1));
        a.sat_limit = sat_limit_for(plan, b + 1, 1);
        a.in_f32 = stored_f32(c, plan, b, 0); a.out_split = c->prec && !stored_f32(c, plan, b, 1);
        float* out;
        if (b == 0) {                       // 1 -> 64 transform on the window image itself
            a.id_mode = 2; a.id = lm_chunk; a.id_win = win; a.idH = g.hin; a.idW = g.win; a.idsh = 1; a.idsw = 1;
            out = x;
        } else if (g.cin == g.cout) {       // identity shortcut, written in place over the block input
            a.id_mode = 1; a.id = x; a.id_ld = g.cout; a.id_split = c->prec && !stored_f32(c, plan, b - 1, 1);
            a.id_scale = c->up(SA(b - 1, 1));
            // (in place only if input and output share a layout: a thread's output bytes are its residual bytes then)
            out = stored_f32(c, plan, b - 1, 1) == stored_f32(c, plan, b, 1) ? x : y;
        } else {
            // Channel-changing block.  If its conv2 -- as a one-segment conv with an f32 residual tensor -- has a
            // Winograd form, the 1x1 strided `_transform` conv (which cannot ride in the K loop of the transformed
            // domain; 3 % of the block's MACs) runs first on its own into an f32 tensor that conv2's epilogue then adds
            // like a residual (the bias of both is in conv2's bias row).  Otherwise it is extra K columns of conv2.
            ConvArgs w = a;
            w.id_mode = 1; w.id = sb.T; w.id_ld = g.cout; w.id_split = 0;
            w.idw = c->A("head.dense.idw");     // ones
            set_out_geometry(w, n, g.hout, g.wout, g.cout, g.cout, g.cout, y);
            if (go ? plan.wino[b][2] : wino_form(w)) {
                ConvArgs t{};
                fill_epilogue_defaults(c, t);
                t.nseg = 1;
                t.seg[0] = make_seg(x, c->WP(p + ".c2.wpk_t"), g.hin, g.win, g.cin, 1, 1, g.sh, g.sw, false);
                set_out_geometry(t, n, g.hout, g.wout, g.cout, g.cout, g.cout, sb.T);
                t.cb = c->A("zero"); t.cb_stride = 0;
                t.ws = c->WS(p + ".c2");            // (conv2 and the transform share one column scale: fold.py emit())
                t.relu = 0; t.out_split = 0;
                t.in_scale = c->up(SA(b - 1, 1));   // (f32 output: no exponent)
                if (go && !dead()) {
                    if (c->stream_1x1 && conv_1x1_stream_eligible(t)) {       // 1.75 GB in, 3.5 GB out, 16 KFLOP per output pixel: a stream
                        Prof pr(c, s, "conv_1x1_stream");
                        launch_conv_1x1_stream(t, s);
                        const double fl = 2.0 * (double)t.M * g.cin * g.cout;
                        pr.done(fl, (double)t.M * (g.cin + g.cout) * 4.0, nullptr, 3.0 * fl);
                    } else {
                        run_conv(c, t, s);
                    }
                }
                a = w;
            } else {                        // (x and a1 share one exponent)
                a.nseg = 2;
                a.seg[1] = make_seg(x, c->WP(p + ".c2.wpk_t"), g.hin, g.win, g.cin, 1, 1, g.sh, g.sw, false);
            }
            out = y;
        }
        set_out_geometry(a, n, g.hout, g.wout, g.cout, g.cout, g.cout, out);
        conv(b, 2, a);
        if (go) tap(c, SA(b, 1), out, (size_t)n * g.hout * g.wout * g.cout, s, stored_f32(c, plan, b, 1));
        if (out == y) std::swap(x, y);
    }
    result = x;
    if (upto >= 9 && go && !dead()) {       // last_conv [5,1] VALID + BN + ReLU  (SN/main.py:232-236)
        const BlockGeo& g = c->stack[7];
        ConvArgs a{};
        fill_epilogue_defaults(c, a);
        a.nseg = 1;
        a.seg[0] = make_seg(x, c->WP("head.conv.wpk"), g.hout, g.wout, g.cout, g.hout, 1, 1, 1, false);
        set_out_geometry(a, n, 1, g.wout, 512, 512, 512, a1);
        a.cb = c->A("head.conv.cb");
        a.ws = c->WS("head.conv");
        a.in_scale = c->up(SA(7, 1)); a.out_scale = c->down(kActHead);
        run_conv(c, a, s);
        tap(c, kActHead, a1, (size_t)n * g.wout * 512, s);
        result = a1;
    }
    }
    c->last_plan = plan;
    return result;
}

int mask_net_impl(nhans_ctx* c, const float* logmag, const int64_t* foff, int nclips, const float* ea,
                  const float* eb, float* logits, float* denoised, const StackBufs& sb, int64_t wf,
                  hipStream_t s) {
    const int64_t total = foff[nclips];
    { int rc = h2d(c, sb.foff_dev, foff, (nclips + 1) * sizeof(int64_t), s); if (rc) return rc; }
    launch_frame_index(sb.foff_dev, nclips, total, sb.f_clip, sb.f_t, sb.f_T, s);
    {
        Prof pr(c, s, "cond_proj");
        launch_cond(ea, eb, nclips, c->A("cond.w"), c->A("cond.base"), c->cond_cols, sb.cb_all, s);
        pr.done(2.0 * nclips * 2 * kEmb * c->cond_cols, 0);
    }
    const BlockGeo& g = c->stack[7];
    for (int64_t g0 = 0; g0 < total; g0 += wf) {
        const int n = (int)std::min<int64_t>(wf, total - g0);
        float* hc = run_stack_chunk(c, logmag, sb, g0, n, 9, s);
        if (launch_error_pending()) break;      // (reported by the entry point: NHANS_EHIP naming the launch)
        // last_dense 13312 -> 201 (+bias) and denoised = mixed_central + out  (SN/main.py:237-242)
        ConvArgs a{};
        fill_epilogue_defaults(c, a);
        a.nseg = 1;
        a.seg[0] = make_seg(hc, c->WP("head.dense.wpk"), 1, 1, g.wout * 512, 1, 1, 1, 1, false);
        set_out_geometry(a, n, 1, 1, 256, kBins, kBins, denoised + g0 * kBins);
        a.cb = c->A("head.dense.cb");
        a.ws = c->WS("head.dense");
        a.out_split = 0;
        a.relu = 0;
        a.in_scale = c->up(kActHead);
        a.id_mode = 1; a.id = logmag + g0 * kBins; a.id_ld = kBins; a.idw = c->A("head.dense.idw");
        if (logits) { a.aux = logits + g0 * kBins; a.aux_ld = kBins; }
        a.kgroup = -1;                          // K = 13312 over a few hundred frames: grouped sum, split-K when small
        run_conv(c, a, s);
    }
    return NHANS_OK;
}

// ---- STFT / iSTFT host-side block tables ----------------------------------------------------
struct HostTables {
    std::vector<int64_t> soff, foff, ooff;
    std::vector<int> bclip, bpos;
};

int stft_impl(nhans_ctx* c, const float* wav, const int64_t* soff, int nclips, int maxf, float* logmag,
              float* phase, int64_t* dev_tables /*3*(nclips+1)*/, int* dev_blocks, std::vector<int64_t>* foff_out,
              hipStream_t s) {
    std::vector<int64_t> foff(nclips + 1, 0);
    std::vector<int> bclip, bf0;
    for (int i = 0; i < nclips; ++i) {
        int64_t t = nhans_num_frames(soff[i + 1] - soff[i]);
        if (t > kMaxFramesPerClip) return fail(NHANS_EINVAL, "clip " + std::to_string(i) + " has more than " +
                                               std::to_string(kMaxFramesPerClip) + " frames (32-bit offsets within a clip)");
        if (maxf > 0) {
            if (t < maxf) return fail(NHANS_ESHORT, "conditioning clip " + std::to_string(i) + " has " +
                                      std::to_string(t) + " frames; " + std::to_string(maxf) + " needed");
            t = maxf;
        }
        foff[i + 1] = foff[i] + t;
        for (int f0 = 0; f0 < t; f0 += kStftFramesPerBlock) { bclip.push_back(i); bf0.push_back(f0); }
    }
    const int nb = (int)bclip.size();
    int rc = h2d(c, dev_tables, soff, (nclips + 1) * 8, s); if (rc) return rc;
    rc = h2d(c, dev_tables + (nclips + 1), foff.data(), (nclips + 1) * 8, s); if (rc) return rc;
    rc = h2d(c, dev_blocks, bclip.data(), (size_t)nb * 4, s); if (rc) return rc;
    rc = h2d(c, dev_blocks + nb, bf0.data(), (size_t)nb * 4, s); if (rc) return rc;
    ClipTable t{dev_tables, dev_tables + (nclips + 1), nullptr};
    Prof pr(c, s, phase ? "stft_features" : "stft_context_features");   // (contexts: log-magnitude only, 200 frames per clip)
    launch_stft(wav, t, dev_blocks, dev_blocks + nb, nb, c->A("tw400"), c->A("window"), logmag, phase, s);
    pr.done(0, (double)foff[nclips] * (kHop * 4 + (phase ? 2 : 1) * kBins * 4));
    if (foff_out) *foff_out = foff;
    return NHANS_OK;
}

size_t stft_blocks(const int64_t* soff, int nclips, int maxf) {
    size_t nb = 0;
    for (int i = 0; i < nclips; ++i) {
        int64_t t = nhans_num_frames(soff[i + 1] - soff[i]);
        if (maxf > 0 && t > maxf) t = maxf;
        nb += (size_t)((t + kStftFramesPerBlock - 1) / kStftFramesPerBlock);
    }
    return nb;
}

int istft_impl(nhans_ctx* c, const float* logmag, const float* phase, const int64_t* foff, int nclips,
               const int64_t* ooff, float* wav_out, int64_t* dev_tables, int* dev_blocks, hipStream_t s) {
    std::vector<int> bclip, bh0;
    for (int i = 0; i < nclips; ++i) {
        const int64_t t = foff[i + 1] - foff[i];
        if (t > kMaxFramesPerClip) return fail(NHANS_EINVAL, "clip " + std::to_string(i) + " has more than " +
                                               std::to_string(kMaxFramesPerClip) + " frames (32-bit offsets within a clip)");
        if (t <= 0) continue;
        for (int h0 = 0; h0 < t + 2; h0 += kIstftHopsPerBlock) { bclip.push_back(i); bh0.push_back(h0); }
    }
    const int nb = (int)bclip.size();
    int rc = h2d(c, dev_tables, foff, (nclips + 1) * 8, s); if (rc) return rc;
    rc = h2d(c, dev_tables + (nclips + 1), ooff, (nclips + 1) * 8, s); if (rc) return rc;
    rc = h2d(c, dev_blocks, bclip.data(), (size_t)nb * 4, s); if (rc) return rc;
    rc = h2d(c, dev_blocks + nb, bh0.data(), (size_t)nb * 4, s); if (rc) return rc;
    ClipTable t{nullptr, dev_tables, dev_tables + (nclips + 1)};
    Prof pr(c, s, "istft_ola");
    launch_istft(logmag, phase, t, dev_blocks, dev_blocks + nb, nb, c->A("tw400"), c->A("wsyn"), wav_out, s);
    pr.done(0, (double)foff[nclips] * (kHop * 4 + 2 * kBins * 4));
    return NHANS_OK;
}

size_t istft_blocks(const int64_t* foff, int nclips) {
    size_t nb = 0;
    for (int i = 0; i < nclips; ++i) {
        const int64_t t = foff[i + 1] - foff[i];
        if (t > 0) nb += (size_t)((t + 2 + kIstftHopsPerBlock - 1) / kIstftHopsPerBlock);
    }
    return nb;
}

// Two tensors that feed ONE accumulator (a channel-changing block's conv2 reads its conv1 output and, through the
// `_transform` segment, the block input) must carry one exponent: the larger of the two.
void tie_exponents(nhans_ctx* c) {
    auto tie = [&](int i, int j) { c->act_exp[i] = c->act_exp[j] = std::max(c->act_exp[i], c->act_exp[j]); };
    for (int b = 1; b < 4; ++b) tie(TA(b - 1, 1), TA(b, 0));
    for (int b = 1; b < 8; ++b)
        if (c->stack[b].cin != c->stack[b].cout) tie(SA(b - 1, 1), SA(b, 0));
}

// End of a calibration bracket: maxima -> exponents (merge: only raise).
int finish_calibration(nhans_ctx* c, bool merge) {
    c->calibrating = false;
    unsigned bits[kNumAct];
    HIP_TRY(hipDeviceSynchronize());
    HIP_TRY(hipMemcpy(bits, c->amax_dev, sizeof bits, hipMemcpyDeviceToHost));
    int e_new[kNumAct];
    for (int i = 0; i < kNumAct; ++i) {
        float m;
        std::memcpy(&m, &bits[i], 4);
        if (!std::isfinite(m)) {
            // merge (the bracket round a saturated batch's f32 rerun): an input that is NaN / Inf itself makes every
            // maximum non-finite -- that says nothing about the range, the exponent stays; a calibration proper refuses
            if (!merge) return fail(NHANS_EINVAL, "calibration: tensor " + std::to_string(i) + " reached a non-finite value");
            e_new[i] = c->act_exp[i];
            continue;
        }
        c->act_amax[i] = m;
        int k = 0;
        if (m > 0.f) (void)frexpf(m, &k);               // m = f * 2^k, f in [0.5, 1)  =>  m * 2^-(k - T) <= 2^T
        // (a tensor the pass never wrote -- or a pass that failed before its first launch -- keeps its exponent)
        e_new[i] = m > 0.f ? k - kActTargetLog2 : c->act_exp[i];
    }
    for (int i = 0; i < kNumAct; ++i) c->act_exp[i] = merge ? std::max(c->act_exp[i], e_new[i]) : e_new[i];
    tie_exponents(c);
    return NHANS_OK;
}

int check_ctx(nhans_ctx* c) {
    if (!c) return fail(NHANS_EINVAL, "null context");
    hipError_t e = hipSetDevice(c->device);
    if (e != hipSuccess) return fail(NHANS_EHIP, std::string("hipSetDevice: ") + hipGetErrorString(e));
    return NHANS_OK;
}

// A launch the runtime rejected anywhere in the sequence just issued -> NHANS_EHIP.
int launch_status() {
    const char* where = "";
    const hipError_t e = take_launch_error(&where);
    if (e == hipSuccess) return NHANS_OK;
    return fail(NHANS_EHIP, std::string("kernel launch failed: ") + where + ": " + hipGetErrorString(e));
}

// Bracket of one hot-path entry point: selects the device, orders the call behind the previous
// call on this context when that one ran on another stream (they share workspace, pinned tables
// and split-K tickets), and on the way out collects launch failures and marks the new tail.
struct Call {
    nhans_ctx* c;
    hipStream_t s;
    int rc;
    Call(nhans_ctx* c_, void* stream) : c(c_), s(static_cast<hipStream_t>(stream)), rc(check_ctx(c_)) {
        if (rc) return;
        (void)take_launch_error(nullptr);               // (a stale record of another context's failure)
        // (The runtime's sticky per-thread "last error" may hold something an earlier HIP call of the APPLICATION left
        // there -- hipErrorPeerAccessAlreadyEnabled, an invalid-value from a pointer-attribute probe: it is not read
        // here, neither blamed on this library's kernels nor cleared on the application's behalf; launches are checked
        // by their own return code, NHANS_LAUNCH.  Round 3 refused to run on top of it; the advisor was right that a
        // benign leftover then disabled the whole library.)
        if (c->have_tail && c->last_stream != s) {
            const hipError_t e = hipStreamWaitEvent(s, c->tail_ev, 0);
            if (e != hipSuccess) rc = fail(NHANS_EHIP, std::string("hipStreamWaitEvent: ") + hipGetErrorString(e));
        }
    }
    int finish(int body_rc) {
        const int lrc = launch_status();
        if (hipEventRecord(c->tail_ev, s) == hipSuccess) { c->have_tail = true; c->last_stream = s; }
        return body_rc ? body_rc : lrc;
    }
};

__global__ void launch_probe_kernel(int* out) {
    extern __shared__ int probe_lds[];
    probe_lds[threadIdx.x] = threadIdx.x;
    __syncthreads();
    if (out && threadIdx.x == 0) *out = probe_lds[63];
}

}  // namespace

// ================================================================================================
extern "C" {

int nhans_abi_version(void) { return NHANS_ABI_VERSION; }
const char* nhans_last_error(void) { return g_err.c_str(); }

int64_t nhans_num_frames(int64_t n) { return n < kWin ? 0 : 1 + (n - kWin) / kHop; }

static int enhance_clips_body(nhans_ctx* c, const float* mix, const int64_t* moff, int nclips, const float* ca,
                              const int64_t* caoff, const float* cbw, const int64_t* cboff, float* den_wav,
                              float* mixed_wav, float* logmag_out, float* phase_out, float* logits_out, float* emb_out,
                              void* stream);

// Activation exponents of a fresh context: one pass of the whole path at precision 0 over a built-in batch of two clips
// -- a two-second mixture of a gliding harmonic voice with syllabic amplitude modulation and noise, conditioned once on
// two noise recordings and once on (silence, noise): an all-zero recording is the reference's default `--pos` and
// drives the tower with the constant silence floor -- with every tensor's maximum recorded.  Deterministic (LCG).
static int calibrate_builtin(nhans_ctx* c) {
    const int64_t n_mix = kWin + (int64_t)kHop * 197, n_ctx = kWin + (int64_t)kHop * (kCtxFrames - 1);
    std::vector<float> mix(2 * n_mix), ca(2 * n_ctx), cb(2 * n_ctx);
    uint32_t lcg = 0x2545F491u;
    auto noise = [&]() { lcg = lcg * 1664525u + 1013904223u; return (float)(int32_t)lcg * (1.0f / 2147483648.0f); };
    double ph = 0.0;
    for (int64_t i = 0; i < n_mix; ++i) {
        const double t = (double)i / 16000.0;
        ph += 2.0 * M_PI * (110.0 + 35.0 * t) / 16000.0;
        double v = 0.0;
        for (int h = 1; h <= 12; ++h) v += std::sin(h * ph) / h;
        const double am = 0.5 - 0.5 * std::cos(2.0 * M_PI * 4.0 * t);
        mix[i] = (float)(0.22 * am * v) + 0.05f * noise();
        mix[n_mix + i] = 0.35f * mix[i] + 0.12f * noise();
    }
    float lp = 0.f;
    for (int64_t i = 0; i < n_ctx; ++i) {
        lp = 0.9f * lp + 0.1f * noise();
        ca[i] = 0.6f * lp;                    // clip 0: low-passed noise / white noise
        cb[i] = 0.1f * noise();
        ca[n_ctx + i] = 0.f;                  // clip 1: silence / white noise
        cb[n_ctx + i] = 0.25f * noise();
    }
    const int64_t moff[3] = {0, n_mix, 2 * n_mix}, coff[3] = {0, n_ctx, 2 * n_ctx};
    float* dev = nullptr;
    const size_t words = (size_t)4 * n_mix + (size_t)4 * n_ctx;
    HIP_TRY(hipMalloc(reinterpret_cast<void**>(&dev), words * 4));
    float *d_mix = dev, *d_den = dev + 2 * n_mix, *d_ca = dev + 4 * n_mix, *d_cb = d_ca + 2 * n_ctx;
    hipError_t e = hipMemcpy(d_mix, mix.data(), mix.size() * 4, hipMemcpyHostToDevice);
    if (e == hipSuccess) e = hipMemcpy(d_ca, ca.data(), ca.size() * 4, hipMemcpyHostToDevice);
    if (e == hipSuccess) e = hipMemcpy(d_cb, cb.data(), cb.size() * 4, hipMemcpyHostToDevice);
    int rc = e == hipSuccess ? NHANS_OK : fail(NHANS_EHIP, std::string("calibration upload: ") + hipGetErrorString(e));
    if (!rc) {
        const int prec = c->prec;
        c->prec = 0;
        c->calibrating = true;
        (void)take_launch_error(nullptr);
        rc = enhance_clips_body(c, d_mix, moff, 2, d_ca, coff, d_cb, coff, d_den, nullptr, nullptr, nullptr, nullptr,
                                nullptr, nullptr);
        if (!rc) rc = launch_status();
        const int frc = finish_calibration(c, false);       // (synchronises)
        if (!rc) rc = frc;
        c->prec = prec;
    }
    (void)hipFree(dev);
    return rc;
}

int nhans_create(int model_kind, const void* blob, size_t nbytes, int device_id, nhans_ctx** out) {
    return nhans_create_ex(model_kind, blob, nbytes, device_id, nullptr, 0, out);
}

int nhans_create_ex(int model_kind, const void* blob, size_t nbytes, int device_id, const int* act_exp, int n_exp,
                    nhans_ctx** out) {
    if (!out || !blob) return fail(NHANS_EINVAL, "null argument");
    *out = nullptr;
    if (act_exp) {
        if (n_exp != kNumAct) return fail(NHANS_EINVAL, "activation exponents: need NHANS_NUM_ACTIVATIONS values");
        for (int i = 0; i < n_exp; ++i)
            if (act_exp[i] < -60 || act_exp[i] > 60) return fail(NHANS_EINVAL, "activation exponent outside [-60, 60]");
    }
    if (model_kind != NHANS_DENOISER && model_kind != NHANS_SEPARATOR) return fail(NHANS_EINVAL, "bad model_kind");
    if (nbytes < sizeof(BlobHeader)) return fail(NHANS_EINVAL, "blob too short");
    const BlobHeader* h = static_cast<const BlobHeader*>(blob);
    if (std::memcmp(h->magic, "NHANSFW1", 8) != 0) return fail(NHANS_EINVAL, "bad blob magic");
    // (a blob of another packing version would load and compute wrong results: fold.py BLOB_VERSION)
    if (h->version != kBlobVersion)
        return fail(NHANS_EINVAL, "the folded blob has packing version " + std::to_string(h->version) + ", this library reads version " +
                                      std::to_string(kBlobVersion) + ": re-fold the weights (nhans_amd.fold.fold_weights)");
    if (h->total_bytes != nbytes || sizeof(BlobHeader) + (size_t)h->n_entries * sizeof(BlobEntry) > nbytes)
        return fail(NHANS_EINVAL, "blob size mismatch");
    HIP_TRY(hipSetDevice(device_id));
    nhans_ctx* c = new nhans_ctx();
    c->kind = model_kind;
    c->device = device_id;
    c->tower = tower_geometry();
    c->stack = main_geometry();
    hipError_t e = hipMalloc(reinterpret_cast<void**>(&c->blob_dev), nbytes);
    if (e != hipSuccess) { delete c; return fail(NHANS_ENOMEM, "hipMalloc for weights failed"); }
    c->blob_bytes = nbytes;
    e = hipMemcpy(c->blob_dev, blob, nbytes, hipMemcpyHostToDevice);
    if (e != hipSuccess) { nhans_destroy(c); return fail(NHANS_EHIP, "weight upload failed"); }
    e = hipHostMalloc(reinterpret_cast<void**>(&c->pin), c->pin_bytes, hipHostMallocDefault);
    if (e != hipSuccess) { nhans_destroy(c); return fail(NHANS_ENOMEM, "pinned staging allocation failed"); }
    e = hipMalloc(reinterpret_cast<void**>(&c->kcounter), c->kcounter_n * sizeof(int));
    if (e == hipSuccess) e = hipMemset(c->kcounter, 0, c->kcounter_n * sizeof(int));
    if (e != hipSuccess) { nhans_destroy(c); return fail(NHANS_ENOMEM, "split-K ticket allocation failed"); }
    e = hipMalloc(reinterpret_cast<void**>(&c->status_dev), sizeof(int));
    if (e == hipSuccess) e = hipMemset(c->status_dev, 0, sizeof(int));
    if (e == hipSuccess) e = hipMalloc(reinterpret_cast<void**>(&c->amax_dev), kNumAct * sizeof(unsigned));
    if (e == hipSuccess) e = hipMemset(c->amax_dev, 0, kNumAct * sizeof(unsigned));
    if (e == hipSuccess) e = hipEventCreateWithFlags(&c->tail_ev, hipEventDisableTiming);
    if (e != hipSuccess) { nhans_destroy(c); return fail(NHANS_EHIP, "status word / ordering event creation failed"); }
    const BlobEntry* ent = reinterpret_cast<const BlobEntry*>(static_cast<const char*>(blob) + sizeof(BlobHeader));
    for (uint32_t i = 0; i < h->n_entries; ++i) {
        if (ent[i].offset % 16 || ent[i].offset + ent[i].nfloats * 4 > nbytes) {
            nhans_destroy(c); return fail(NHANS_EINVAL, "blob entry out of range");
        }
        std::string name(ent[i].name, strnlen(ent[i].name, sizeof ent[i].name));
        c->arr[name] = reinterpret_cast<const float*>(reinterpret_cast<const char*>(c->blob_dev) + ent[i].offset);
        c->arr_n[name] = ent[i].nfloats;
    }
    // conditioning columns: conv order m0.c1, m0.c2, m1.c1, ...
    int off = 0;
    for (int b = 0; b < 8; ++b) for (int j = 0; j < 2; ++j) { c->cond_off.push_back(off); off += c->stack[b].cout; }
    c->cond_cols = off;
    // every array the launch sequences will dereference must be present with the right size
    std::vector<std::pair<std::string, size_t>> need = {
        {"tw400", 800}, {"window", 400}, {"wsyn", 400}, {"zero", 16384},
        {"cond.w", (size_t)2 * kEmb * off}, {"cond.base", (size_t)off},
        {"head.conv.wpk", (size_t)5 * 512 * 512}, {"head.conv.cb", 512},
        {"head.dense.wpk", (size_t)26 * 512 * 256}, {"head.dense.cb", 256}, {"head.dense.idw", 256}};
    for (int b = 0; b < 4; ++b) {
        const BlockGeo& g = c->tower[b];
        const std::string p = "t" + std::to_string(b);
        const size_t k2 = (size_t)g.kh * g.kw * g.cout * g.cout;
        if (b == 0) { need.push_back({p + ".c1.w", (size_t)g.kh * g.kw * 64}); need.push_back({p + ".c2.idw", (size_t)g.cout}); }
        else { need.push_back({p + ".c1.wpk", (size_t)g.kh * g.kw * g.cin * g.cout}); need.push_back({p + ".c2.wpk_t", (size_t)g.cin * g.cout}); }
        need.push_back({p + ".c1.cb", (size_t)g.cout});
        need.push_back({p + ".c2.wpk", k2});
        need.push_back({p + ".c2.cb", (size_t)g.cout});
    }
    for (int b = 0; b < 8; ++b) {
        const BlockGeo& g = c->stack[b];
        const std::string p = "m" + std::to_string(b);
        if (b == 0) need.push_back({p + ".c1.w", (size_t)g.kh * g.kw * 64});
        else need.push_back({p + ".c1.wpk", (size_t)g.kh * g.kw * g.cin * g.cout});
        need.push_back({p + ".c2.wpk", (size_t)g.kh * g.kw * g.cout * g.cout});
        if (b > 0 && g.cin != g.cout) need.push_back({p + ".c2.wpk_t", (size_t)g.cin * g.cout});
        need.push_back({p + ".c2.idw", (size_t)g.cout});
        for (const char* cv : {".c1", ".c2"}) {
            need.push_back({p + cv + ".tf", (size_t)g.hout * g.wout * g.cout});
        }
    }
    for (const auto& kv : need) {
        auto it = c->arr_n.find(kv.first);
        if (it == c->arr_n.end() || it->second != kv.second) {
            std::string msg = "folded blob: array '" + kv.first + "' missing or wrong size (want " +
                              std::to_string(kv.second) + ")";
            nhans_destroy(c);
            return fail(NHANS_EINVAL, msg);
        }
    }
    if (act_exp) {
        // (exponents a previous context calibrated for this very blob -- the caller's cache vouches for that: no pass)
        std::copy(act_exp, act_exp + kNumAct, c->act_exp);
        tie_exponents(c);
    } else if (c->A("head.dense.wpk_h")) {
        const int rc = calibrate_builtin(c);
        if (rc) { nhans_destroy(c); return rc; }
    }
    *out = c;
    return NHANS_OK;
}

void nhans_destroy(nhans_ctx* c) {
    if (!c) return;
    (void)hipSetDevice(c->device);
    (void)hipDeviceSynchronize();
    for (auto& kv : c->prof)
        for (auto& ev : kv.second.pending) { (void)hipEventDestroy(ev.first); (void)hipEventDestroy(ev.second); }
    for (hipEvent_t ev : c->event_pool) (void)hipEventDestroy(ev);
    if (c->tail_ev) (void)hipEventDestroy(c->tail_ev);
    if (c->status_dev) (void)hipFree(c->status_dev);
    if (c->amax_dev) (void)hipFree(c->amax_dev);
    if (c->ws) (void)hipFree(c->ws);
    if (c->kscratch) (void)hipFree(c->kscratch);
    if (c->kcounter) (void)hipFree(c->kcounter);
    if (c->pin) (void)hipHostFree(c->pin);
    if (c->blob_dev) (void)hipFree(c->blob_dev);
    delete c;
}

int nhans_set_option(nhans_ctx* c, const char* key, int64_t value) {
    if (!c || !key) return fail(NHANS_EINVAL, "null argument");
    const std::string k(key);
    if (k == "frames_per_chunk") {
        // the fast conv kernels address a tensor with 32-bit element offsets: the largest one of a pass
        // (frames x 35 x 201 x 64) must stay below 2^31 elements, i.e. at most 4,769 frames; above that every
        // layer would silently fall back to the slow kernel, far above it M = frames x Ho x Wo overflows int
        if (value < 1 || value > kMaxFramesPerChunk)
            return fail(NHANS_EINVAL, "frames_per_chunk must be in [1, " + std::to_string(kMaxFramesPerChunk) + "]");
        c->frames_per_chunk = value;
    }
    else if (k == "contexts_per_chunk") { if (value < 1) return fail(NHANS_EINVAL, "contexts_per_chunk < 1"); c->contexts_per_chunk = (int)value; }
    else if (k == "profile") c->profile = value != 0;
    else if (k == "debug_cycles_ptr") {
        if (!kDev) return fail(NHANS_EINVAL, "debug_cycles_ptr exists only in a NHANS_DEV build (make DEV=1)");
        c->dbg = reinterpret_cast<long long*>(static_cast<intptr_t>(value));
    }
    else if (k == "epilogue_wide") c->epi8 = value != 0;
    else if (k == "consumer_interleave") {
        if (value < 0 || value > 2) return fail(NHANS_EINVAL, "consumer_interleave must be 0, 1 or 2");
        c->ilv = (int)value;
    }
    else if (k == "conv_variant") {
        if (value < -1 || value > 2) return fail(NHANS_EINVAL, "conv_variant must be -1 (auto), 0, 1 or 2");
        c->conv_variant = (int)value;
    }
    else if (k == "calibrate") {
        if (value < 0 || value > 3) return fail(NHANS_EINVAL, "calibrate must be 1 (start), 0 (stop, set), 2 (stop, raise only) or 3 (stop, discard)");
        int rc = check_ctx(c); if (rc) return rc;
        if (value == 1) {
            HIP_TRY(hipDeviceSynchronize());
            HIP_TRY(hipMemset(c->amax_dev, 0, kNumAct * sizeof(unsigned)));
            c->calibrating = true;
        } else {
            if (!c->calibrating) return fail(NHANS_EINVAL, "calibrate: no bracket is open");
            if (value == 3) { c->calibrating = false; return NHANS_OK; }     // (the pass failed: nothing was learnt)
            return finish_calibration(c, value == 2);
        }
    }
    else if (k == "winograd") c->wino = value != 0;
    else if (k == "winograd_f32_tensors") c->wino_f32 = (value == 2 || value == 3) ? (int)value : value != 0;
    else if (k == "split_k") c->split_k = value != 0;
    else if (k == "stream_1x1") c->stream_1x1 = value != 0;
    else if (k == "precision") {
        if (value != 0 && value != 1) return fail(NHANS_EINVAL, "precision must be 0 (f32) or 1 (f16x3)");
        if (value == 1 && !c->A("head.dense.wpk_h"))
            return fail(NHANS_EINVAL, "the folded blob carries no split-f16 weights");
        c->prec = (int)value;
    }
    else return fail(NHANS_EINVAL, "unknown option " + k);
    return NHANS_OK;
}

int nhans_set_activation_exponents(nhans_ctx* c, const int* e, int n) {
    if (!c || !e || n != kNumAct) return fail(NHANS_EINVAL, "activation exponents: need NHANS_NUM_ACTIVATIONS values");
    for (int i = 0; i < n; ++i)
        if (e[i] < -60 || e[i] > 60) return fail(NHANS_EINVAL, "activation exponent outside [-60, 60]");
    std::copy(e, e + n, c->act_exp);
    tie_exponents(c);
    return NHANS_OK;
}

int nhans_get_activation_exponents(nhans_ctx* c, int* e_out, int n) {
    if (!c || !e_out || n != kNumAct) return fail(NHANS_EINVAL, "activation exponents: need NHANS_NUM_ACTIVATIONS values");
    std::copy(c->act_exp, c->act_exp + n, e_out);
    return NHANS_OK;
}

int nhans_get_activation_amax(nhans_ctx* c, float* amax_out, int n) {
    if (!c || !amax_out || n != kNumAct) return fail(NHANS_EINVAL, "activation maxima: need NHANS_NUM_ACTIVATIONS values");
    std::copy(c->act_amax, c->act_amax + n, amax_out);
    return NHANS_OK;
}

size_t nhans_workspace_bytes(nhans_ctx* c, int64_t total_frames, int nclips) {
    if (!c) return 0;
    const int64_t wf = std::min<int64_t>(c->frames_per_chunk, std::max<int64_t>(total_frames, 1));
    size_t b = stack_ws_bytes(c, total_frames, nclips, wf);
    b = std::max(b, 3 * ws_size(tower_buf_floats(c), 4));
    b += 4 * ws_size((size_t)total_frames * kBins, 4);                       // logmag, phase, denoised, logits
    b += ws_size((size_t)2 * nclips * kCtxFrames * kBins, 4) + ws_size((size_t)2 * nclips * kEmb, 4);
    b += 1 << 20;
    return b;
}

static int stft_features_body(nhans_ctx* c, const float* wav, const int64_t* soff, int nclips, int maxf,
                        float* logmag, float* phase, void* stream) {
    int rc = NHANS_OK;
    if (!wav || !soff || !logmag || nclips < 0) return fail(NHANS_EINVAL, "null argument");
    hipStream_t s = static_cast<hipStream_t>(stream);
    const size_t nb = stft_blocks(soff, nclips, maxf);
    rc = ws_reserve(c, ws_size(2 * (nclips + 1), 8) + ws_size(2 * nb, 4)); if (rc) return rc;
    int64_t* tabs = ws_take<int64_t>(c, 2 * (nclips + 1));
    int* blocks = ws_take<int>(c, 2 * nb);
    return stft_impl(c, wav, soff, nclips, maxf, logmag, phase, tabs, blocks, nullptr, s);
}

int nhans_stft_features(nhans_ctx* c, const float* wav, const int64_t* soff, int nclips, int maxf,
                        float* logmag, float* phase, void* stream) {
    Call call(c, stream);
    if (call.rc) return call.rc;
    return call.finish(stft_features_body(c, wav, soff, nclips, maxf, logmag, phase, stream));
}

static int embed_body(nhans_ctx* c, const float* ctx_lm, int n, float* emb_out, void* stream) {
    int rc = NHANS_OK;
    if (!ctx_lm || !emb_out || n < 0) return fail(NHANS_EINVAL, "null argument");
    hipStream_t s = static_cast<hipStream_t>(stream);
    const size_t nb = tower_buf_floats(c);
    rc = ws_reserve(c, 3 * ws_size(nb, 4)); if (rc) return rc;
    float* X = ws_take<float>(c, nb); float* A = ws_take<float>(c, nb); float* Y = ws_take<float>(c, nb);
    return embed_impl(c, ctx_lm, n, emb_out, X, A, Y, s);
}

int nhans_embed(nhans_ctx* c, const float* ctx_lm, int n, float* emb_out, void* stream) {
    Call call(c, stream);
    if (call.rc) return call.rc;
    return call.finish(embed_body(c, ctx_lm, n, emb_out, stream));
}

static int mask_net_body(nhans_ctx* c, const float* logmag, const int64_t* foff, int nclips, const float* ea,
                   const float* eb, float* logits, float* denoised, void* stream) {
    int rc = NHANS_OK;
    if (!logmag || !foff || !ea || !eb || !denoised || nclips < 1) return fail(NHANS_EINVAL, "null argument");
    hipStream_t s = static_cast<hipStream_t>(stream);
    const int64_t total = foff[nclips];
    if (total <= 0) return NHANS_OK;
    const int64_t wf = std::min<int64_t>(c->frames_per_chunk, total);
    rc = ws_reserve(c, stack_ws_bytes(c, total, nclips, wf)); if (rc) return rc;
    StackBufs sb;
    stack_take(c, total, nclips, wf, &sb);
    return mask_net_impl(c, logmag, foff, nclips, ea, eb, logits, denoised, sb, wf, s);
}

int nhans_mask_net(nhans_ctx* c, const float* logmag, const int64_t* foff, int nclips, const float* ea,
                   const float* eb, float* logits, float* denoised, void* stream) {
    Call call(c, stream);
    if (call.rc) return call.rc;
    return call.finish(mask_net_body(c, logmag, foff, nclips, ea, eb, logits, denoised, stream));
}

static int debug_block_output_body(nhans_ctx* c, const float* logmag, const int64_t* foff, int nclips, const float* ea,
                             const float* eb, int64_t frame0, int nframes, int block, float* out, void* stream) {
    int rc = NHANS_OK;
    if (!logmag || !foff || !ea || !eb || !out || block < 0 || block > 8) return fail(NHANS_EINVAL, "bad argument");
    hipStream_t s = static_cast<hipStream_t>(stream);
    const int64_t total = foff[nclips];
    if (frame0 < 0 || frame0 + nframes > total) return fail(NHANS_EINVAL, "frame range outside batch");
    rc = ws_reserve(c, stack_ws_bytes(c, total, nclips, nframes)); if (rc) return rc;
    StackBufs sb;
    stack_take(c, total, nclips, nframes, &sb);
    rc = h2d(c, sb.foff_dev, foff, (nclips + 1) * sizeof(int64_t), s); if (rc) return rc;
    launch_frame_index(sb.foff_dev, nclips, total, sb.f_clip, sb.f_t, sb.f_T, s);
    launch_cond(ea, eb, nclips, c->A("cond.w"), c->A("cond.base"), c->cond_cols, sb.cb_all, s);
    const float* res = run_stack_chunk(c, logmag, sb, frame0, nframes, block + 1, s);
    size_t per;
    if (block == 8) per = (size_t)26 * 512;
    else per = (size_t)c->stack[block].hout * c->stack[block].wout * c->stack[block].cout;
    if (c->prec && block < 8 && stored_f32(c, c->last_plan, block, 1)) launch_scale_copy(res, per * nframes, c->up(SA(block, 1)), out, s);
    else if (c->prec) launch_unsplit(res, (int64_t)nframes * (int64_t)(per / (block == 8 ? 512 : c->stack[block].cout)),
                                block == 8 ? 512 : c->stack[block].cout, c->up(block == 8 ? kActHead : SA(block, 1)), out, s);
    else HIP_TRY(hipMemcpyAsync(out, res, per * nframes * 4, hipMemcpyDeviceToDevice, s));
    return NHANS_OK;
}

int nhans_debug_block_output(nhans_ctx* c, const float* logmag, const int64_t* foff, int nclips, const float* ea,
                             const float* eb, int64_t frame0, int nframes, int block, float* out, void* stream) {
    Call call(c, stream);
    if (call.rc) return call.rc;
    return call.finish(debug_block_output_body(c, logmag, foff, nclips, ea, eb, frame0, nframes, block, out, stream));
}

static int istft_body(nhans_ctx* c, const float* logmag, const float* phase, const int64_t* foff, int nclips,
                const int64_t* ooff, float* wav_out, void* stream) {
    int rc = NHANS_OK;
    if (!logmag || !phase || !foff || !ooff || !wav_out) return fail(NHANS_EINVAL, "null argument");
    hipStream_t s = static_cast<hipStream_t>(stream);
    const size_t nb = istft_blocks(foff, nclips);
    rc = ws_reserve(c, ws_size(2 * (nclips + 1), 8) + ws_size(2 * nb, 4)); if (rc) return rc;
    int64_t* tabs = ws_take<int64_t>(c, 2 * (nclips + 1));
    int* blocks = ws_take<int>(c, 2 * nb);
    return istft_impl(c, logmag, phase, foff, nclips, ooff, wav_out, tabs, blocks, s);
}

int nhans_istft(nhans_ctx* c, const float* logmag, const float* phase, const int64_t* foff, int nclips,
                const int64_t* ooff, float* wav_out, void* stream) {
    Call call(c, stream);
    if (call.rc) return call.rc;
    return call.finish(istft_body(c, logmag, phase, foff, nclips, ooff, wav_out, stream));
}

static int enhance_clips_body(nhans_ctx* c, const float* mix, const int64_t* moff, int nclips, const float* ca,
                        const int64_t* caoff, const float* cbw, const int64_t* cboff, float* den_wav,
                        float* mixed_wav, float* logmag_out, float* phase_out, float* logits_out, float* emb_out,
                        void* stream) {
    int rc = NHANS_OK;
    if (!mix || !moff || !ca || !caoff || !cbw || !cboff || !den_wav || nclips < 1)
        return fail(NHANS_EINVAL, "null argument");
    hipStream_t s = static_cast<hipStream_t>(stream);
    std::vector<int64_t> foff(nclips + 1, 0);
    for (int i = 0; i < nclips; ++i) {
        const int64_t n = moff[i + 1] - moff[i];
        if (n >= kWin && (n - kWin) % kHop != 0)
            return fail(NHANS_EINVAL, "mixture clip " + std::to_string(i) + " is not trimmed to a whole frame count");
        foff[i + 1] = foff[i] + nhans_num_frames(n);
    }
    const int64_t total = foff[nclips];
    const int64_t wf = std::min<int64_t>(c->frames_per_chunk, std::max<int64_t>(total, 1));
    // workspace plan
    const size_t nb_mix = stft_blocks(moff, nclips, 0), nb_ca = stft_blocks(caoff, nclips, kCtxFrames),
                 nb_cb = stft_blocks(cboff, nclips, kCtxFrames), nb_is = istft_blocks(foff.data(), nclips);
    const size_t nblk = std::max(std::max(nb_mix, nb_ca), std::max(nb_cb, nb_is));
    size_t bytes = 4 * ws_size((size_t)total * kBins, 4) + ws_size((size_t)2 * nclips * kCtxFrames * kBins, 4) +
                   ws_size((size_t)2 * nclips * kEmb, 4) + 4 * ws_size(2 * (nclips + 1), 8) + 4 * ws_size(2 * nblk, 4);
    bytes += std::max(stack_ws_bytes(c, total, nclips, wf), 3 * ws_size(tower_buf_floats(c), 4));
    rc = ws_reserve(c, bytes); if (rc) return rc;
    float* lm = ws_take<float>(c, (size_t)total * kBins);
    float* ph = ws_take<float>(c, (size_t)total * kBins);
    float* den = ws_take<float>(c, (size_t)total * kBins);
    float* lg = ws_take<float>(c, (size_t)total * kBins);
    float* ctxlm = ws_take<float>(c, (size_t)2 * nclips * kCtxFrames * kBins);
    float* emb = ws_take<float>(c, (size_t)2 * nclips * kEmb);
    int64_t* tabs[4]; int* blks[4];
    for (int i = 0; i < 4; ++i) { tabs[i] = ws_take<int64_t>(c, 2 * (nclips + 1)); blks[i] = ws_take<int>(c, 2 * nblk); }
    const size_t mark = c->ws_top;

    rc = stft_impl(c, mix, moff, nclips, 0, lm, ph, tabs[0], blks[0], nullptr, s); if (rc) return rc;
    rc = stft_impl(c, ca, caoff, nclips, kCtxFrames, ctxlm, nullptr, tabs[1], blks[1], nullptr, s); if (rc) return rc;
    rc = stft_impl(c, cbw, cboff, nclips, kCtxFrames, ctxlm + (size_t)nclips * kCtxFrames * kBins, nullptr, tabs[2],
                   blks[2], nullptr, s);
    if (rc) return rc;
    {
        const size_t nb = tower_buf_floats(c);
        float* X = ws_take<float>(c, nb); float* A = ws_take<float>(c, nb); float* Y = ws_take<float>(c, nb);
        rc = embed_impl(c, ctxlm, 2 * nclips, emb, X, A, Y, s); if (rc) return rc;
    }
    if (total > 0) {
        c->ws_top = mark;
        StackBufs sb;
        stack_take(c, total, nclips, wf, &sb);
        rc = mask_net_impl(c, lm, foff.data(), nclips, emb, emb + (size_t)nclips * kEmb, lg, den, sb, wf, s);
        if (rc) return rc;
        rc = istft_impl(c, den, ph, foff.data(), nclips, moff, den_wav, tabs[3], blks[3], s); if (rc) return rc;
        if (mixed_wav) {
            // tabs/blks[3] are reused: same stream, so the first launch has consumed them in order
            rc = istft_impl(c, lm, ph, foff.data(), nclips, moff, mixed_wav, tabs[3], blks[3], s); if (rc) return rc;
        }
        if (logmag_out) HIP_TRY(hipMemcpyAsync(logmag_out, lm, (size_t)total * kBins * 4, hipMemcpyDeviceToDevice, s));
        if (phase_out) HIP_TRY(hipMemcpyAsync(phase_out, ph, (size_t)total * kBins * 4, hipMemcpyDeviceToDevice, s));
        if (logits_out) HIP_TRY(hipMemcpyAsync(logits_out, lg, (size_t)total * kBins * 4, hipMemcpyDeviceToDevice, s));
    }
    if (emb_out) HIP_TRY(hipMemcpyAsync(emb_out, emb, (size_t)2 * nclips * kEmb * 4, hipMemcpyDeviceToDevice, s));
    return NHANS_OK;
}

int nhans_enhance_clips(nhans_ctx* c, const float* mix, const int64_t* moff, int nclips, const float* ca,
                        const int64_t* caoff, const float* cbw, const int64_t* cboff, float* den_wav,
                        float* mixed_wav, float* logmag_out, float* phase_out, float* logits_out, float* emb_out,
                        void* stream) {
    Call call(c, stream);
    if (call.rc) return call.rc;
    return call.finish(enhance_clips_body(c, mix, moff, nclips, ca, caoff, cbw, cboff, den_wav, mixed_wav, logmag_out, phase_out, logits_out, emb_out, stream));
}

int nhans_take_status(nhans_ctx* c, int* flags_out, void* stream) {
    int rc = check_ctx(c); if (rc) return rc;
    if (!flags_out) return fail(NHANS_EINVAL, "null argument");
    hipStream_t s = static_cast<hipStream_t>(stream);
    // the kernels that set the bits may have run on another stream than the one given here: order the
    // read-and-clear behind the context's last call, as every hot-path entry point does (struct Call)
    if (c->have_tail && c->last_stream != s) HIP_TRY(hipStreamWaitEvent(s, c->tail_ev, 0));
    int flags = 0;
    HIP_TRY(hipMemcpyAsync(&flags, c->status_dev, sizeof(int), hipMemcpyDeviceToHost, s));
    HIP_TRY(hipMemsetAsync(c->status_dev, 0, sizeof(int), s));
    HIP_TRY(hipStreamSynchronize(s));
    if (hipEventRecord(c->tail_ev, s) == hipSuccess) { c->have_tail = true; c->last_stream = s; }   // the clear is part of the order
    *flags_out = flags;
    return NHANS_OK;
}

int nhans_debug_launch_probe(size_t dynamic_lds_bytes, void* stream) {
    (void)take_launch_error(nullptr);
    static unsigned long long probe_devices = 0;
    if (dynamic_lds_bytes > 65536)
        set_max_dynamic_lds(reinterpret_cast<const void*>(&launch_probe_kernel), dynamic_lds_bytes, &probe_devices, "launch_probe");
    // (the launch is attempted even if the attribute was refused: both failures must surface)
    NHANS_LAUNCH("launch_probe", launch_probe_kernel, dim3(1), dim3(64), dynamic_lds_bytes, static_cast<hipStream_t>(stream),
                 static_cast<int*>(nullptr));
    return launch_status();
}

uint32_t nhans_crc32c(uint32_t crc, const void* data, size_t n) {
    // slicing-by-8 over the reflected Castagnoli polynomial 0x82F63B78
    static const struct Tab {
        uint32_t t[8][256];
        Tab() {
            for (uint32_t i = 0; i < 256; ++i) {
                uint32_t c = i;
                for (int k = 0; k < 8; ++k) c = (c & 1) ? (c >> 1) ^ 0x82F63B78u : c >> 1;
                t[0][i] = c;
            }
            for (uint32_t i = 0; i < 256; ++i)
                for (int s = 1; s < 8; ++s) t[s][i] = (t[s - 1][i] >> 8) ^ t[0][t[s - 1][i] & 0xFF];
        }
    } T;
    const unsigned char* p = static_cast<const unsigned char*>(data);
    uint32_t c = ~crc;
    while (n >= 8) {
        uint32_t lo, hi;
        std::memcpy(&lo, p, 4);
        std::memcpy(&hi, p + 4, 4);
        lo ^= c;
        c = T.t[7][lo & 0xFF] ^ T.t[6][(lo >> 8) & 0xFF] ^ T.t[5][(lo >> 16) & 0xFF] ^ T.t[4][lo >> 24] ^
            T.t[3][hi & 0xFF] ^ T.t[2][(hi >> 8) & 0xFF] ^ T.t[1][(hi >> 16) & 0xFF] ^ T.t[0][hi >> 24];
        p += 8;
        n -= 8;
    }
    while (n--) c = T.t[0][(c ^ *p++) & 0xFF] ^ (c >> 8);
    return ~c;
}

int nhans_profile_reset(nhans_ctx* c) {
    if (!c) return fail(NHANS_EINVAL, "null context");
    (void)hipSetDevice(c->device);
    (void)hipDeviceSynchronize();
    for (auto& kv : c->prof)
        for (auto& ev : kv.second.pending) { c->event_pool.push_back(ev.first); c->event_pool.push_back(ev.second); }
    c->prof.clear();
    return NHANS_OK;
}

int nhans_profile_json(nhans_ctx* c, char* buf, size_t buflen) {
    if (!c) return fail(NHANS_EINVAL, "null context");
    (void)hipSetDevice(c->device);
    std::string js = "{";
    bool first = true;
    for (auto& kv : c->prof) {
        ProfEntry& e = kv.second;
        for (auto& ev : e.pending) {
            (void)hipEventSynchronize(ev.second);
            float ms = 0.f;
            if (hipEventElapsedTime(&ms, ev.first, ev.second) == hipSuccess) e.ms += ms;
            c->event_pool.push_back(ev.first);
            c->event_pool.push_back(ev.second);
        }
        e.pending.clear();
        char line[384];
        snprintf(line, sizeof line, "%s\"%s\": {\"calls\": %d, \"ms\": %.6f, \"flops\": %.6e, \"bytes\": %.6e, \"mfma_flops\": %.6e}",
                 first ? "" : ", ", kv.first.c_str(), e.calls, e.ms, e.flops, e.bytes, e.mfma);
        js += line;
        first = false;
    }
    js += "}";
    if (buf && buflen) {
        const size_t n = std::min(buflen - 1, js.size());
        std::memcpy(buf, js.data(), n);
        buf[n] = 0;
    }
    return (int)js.size();
}

}  // extern "C"
