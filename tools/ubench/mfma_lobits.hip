// Dev micro-benchmark (round 6): does the socket's power cap pay for operand BITS?  The f16x3 arithmetic of the conv
// kernels spends three v_mfma_f32_32x32x16_f16 per MAC -- ah*bh, ah*bl, al*bh with x = hi + lo -- and at the cap
// (1,400 W) register-only MFMAs sustain 2,400 TFLOP/s on zero operands against ~1,660 on random ones
// (profiles/r02/mfma_power_ceiling.txt).  This probe issues exactly that product mix on register operands and rounds the
// `lo` halves (activation side, weight side, both) to k = 10 ... 0 mantissa bits (10 = as shipped, 0 = lo dropped to a
// power of two), in two issue orders:
//   order 0  product-outer: consecutive MFMAs share no operand register
//   order 1  product-inner: the three products of one accumulator back to back (ah*bh, ah*bl share ah; al*bh shares bh)
// Every configuration runs `secs` seconds; the unmasked configuration is repeated between the others so that thermal
// drift is visible.  Output: sustained TFLOP/s over the last two thirds of each run.
//   hipcc --offload-arch=gfx950 -O3 mfma_lobits.hip -o mfma_lobits && ./mfma_lobits [seconds per point] [waves per CU]
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

// src: [4][2][64 lanes][8] halves = ah, al, bh, bl for two row / column blocks each
template <int ORDER>
__global__ void __launch_bounds__(256) k(float* out, const _Float16* src, int iters) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    f16x8 op[4][2];
    for (int p = 0; p < 4; ++p)
        for (int i = 0; i < 2; ++i)
            op[p][i] = *reinterpret_cast<const f16x8*>(src + ((((wave * 4 + p) * 2 + i) * 64 + lane) * 8));
    f32x16 acc[2][2];
    for (int m = 0; m < 2; ++m) for (int n = 0; n < 2; ++n) for (int r = 0; r < 16; ++r) acc[m][n][r] = 0.f;
    for (int it = 0; it < iters; ++it) {
        if (ORDER == 0) {
#pragma unroll
            for (int p = 0; p < 3; ++p)
#pragma unroll
                for (int m = 0; m < 2; ++m)
#pragma unroll
                    for (int n = 0; n < 2; ++n)
                        acc[m][n] = __builtin_amdgcn_mfma_f32_32x32x16_f16(op[p == 2 ? 1 : 0][m], op[p == 1 ? 3 : 2][n], acc[m][n], 0, 0, 0);
        } else {
#pragma unroll
            for (int m = 0; m < 2; ++m)
#pragma unroll
                for (int n = 0; n < 2; ++n) {
                    acc[m][n] = __builtin_amdgcn_mfma_f32_32x32x16_f16(op[0][m], op[2][n], acc[m][n], 0, 0, 0);
                    acc[m][n] = __builtin_amdgcn_mfma_f32_32x32x16_f16(op[0][m], op[3][n], acc[m][n], 0, 0, 0);
                    acc[m][n] = __builtin_amdgcn_mfma_f32_32x32x16_f16(op[1][m], op[2][n], acc[m][n], 0, 0, 0);
                }
        }
    }
    float s = 0;
    for (int m = 0; m < 2; ++m) for (int n = 0; n < 2; ++n) for (int r = 0; r < 16; ++r) s += acc[m][n][r];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}

static float gauss(unsigned& x) {
    float s = 0;
    for (int i = 0; i < 12; ++i) { x = x * 1664525u + 1013904223u; s += (x >> 8) * (1.f / 16777216.f); }
    return s - 6.f;
}

// lo rounded to nearest at k mantissa bits (k = 10: unchanged)
static _Float16 round_lo(_Float16 v, int k) {
    if (k >= 10) return v;
    unsigned short b; memcpy(&b, &v, 2);
    const unsigned drop = 10 - k, half = 1u << (drop - 1);
    unsigned m = (b & 0x7fffu) + half;           // carries into the exponent correctly
    m &= ~((1u << drop) - 1);
    b = (unsigned short)((b & 0x8000u) | (m & 0x7fffu));
    memcpy(&v, &b, 2);
    return v;
}

struct Cfg { const char* name; int ka, kb, order, relu; };

int main(int argc, char** argv) {
    const double secs = argc > 1 ? atof(argv[1]) : 4.0;
    const int wpc = argc > 2 ? atoi(argv[2]) : 8;
    const int blocks = 256 * wpc / 4, iters = 20000;
    const int N = 4 * 4 * 2 * 64 * 8;                 // [wave][operand][block][lane][8]
    float* out; _Float16* src;
    (void)hipMalloc(&out, blocks * 256 * 4); (void)hipMalloc(&src, N * 2);
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    const double flop = (double)blocks * 4 * iters * 12 * 2.0 * 32 * 32 * 16;
    std::vector<Cfg> cfgs;
    const int ks[] = {8, 7, 6, 5, 4, 2, 0};
    cfgs.push_back({"as shipped (lo 10 bits)", 10, 10, 0, 1});
    for (int k : ks) {
        cfgs.push_back({"activation lo", k, 10, 0, 1});
        cfgs.push_back({"weight lo", 10, k, 0, 1});
        cfgs.push_back({"both lo", k, k, 0, 1});
        cfgs.push_back({"as shipped (lo 10 bits)", 10, 10, 0, 1});
    }
    cfgs.push_back({"lo = 0 on both sides (one product real)", -1, -1, 0, 1});
    cfgs.push_back({"as shipped, product-inner order", 10, 10, 1, 1});
    cfgs.push_back({"both lo 6, product-inner order", 6, 6, 1, 1});
    cfgs.push_back({"as shipped (lo 10 bits)", 10, 10, 0, 1});
    cfgs.push_back({"as shipped, activations dense (no ReLU zeros)", 10, 10, 0, 0});
    cfgs.push_back({"all operands zero", -2, -2, 0, 1});
    cfgs.push_back({"as shipped (lo 10 bits)", 10, 10, 0, 1});
    printf("# %d waves per CU, %.1f s per point, 12 MFMAs (2x2 accumulators x 3 products) per iteration; activations post-ReLU-like\n"
           "# (half zeros) unless stated, weights scaled into [32, 64) per column like fold.py does\n", wpc, secs);
    printf("%-48s %4s %4s %5s %10s\n", "configuration", "kA", "kB", "order", "TFLOP/s");
    std::vector<_Float16> h(N);
    for (const Cfg& c : cfgs) {
        unsigned x = 12345;
        for (int w = 0; w < 4; ++w)
            for (int blk = 0; blk < 2; ++blk)
                for (int e = 0; e < 512; ++e) {
                    float a = gauss(x) * 3.f, b = gauss(x);
                    if (c.relu && a < 0) a = 0;
                    b = (b < 0 ? -1.f : 1.f) * (32.f + 32.f * fminf(fabsf(b) / 3.f, 0.999f));
                    if (c.ka == -2) a = b = 0;
                    _Float16 ah = (_Float16)a, bh = (_Float16)b;
                    _Float16 al = (_Float16)(a - (float)ah), bl = (_Float16)(b - (float)bh);
                    al = c.ka < 0 ? (_Float16)0.f : round_lo(al, c.ka);
                    bl = c.kb < 0 ? (_Float16)0.f : round_lo(bl, c.kb);
                    const _Float16 v[4] = {ah, al, bh, bl};
                    for (int p = 0; p < 4; ++p) h[(((w * 4 + p) * 2 + blk) * 512) + e] = v[p];
                }
        (void)hipMemcpy(src, h.data(), N * 2, hipMemcpyHostToDevice);
        double t = 0, tt = 0; int n = 0;
        while (t < secs * 1e3) {
            (void)hipEventRecord(e0, 0);
            if (c.order) hipLaunchKernelGGL(k<1>, dim3(blocks), dim3(256), 0, 0, out, src, iters);
            else hipLaunchKernelGGL(k<0>, dim3(blocks), dim3(256), 0, 0, out, src, iters);
            (void)hipEventRecord(e1, 0); (void)hipEventSynchronize(e1);
            float ms; (void)hipEventElapsedTime(&ms, e0, e1);
            t += ms;
            if (t > secs * 1e3 / 3) { tt += ms; ++n; }
        }
        printf("%-48s %4d %4d %5d %10.1f\n", c.name, c.ka, c.kb, c.order, flop * n / tt / 1e9);
        fflush(stdout);
    }
    return 0;
}
