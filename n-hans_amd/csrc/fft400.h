// 400-point mixed-radix FFT building blocks (400 = 20 x 20, 20 = 4 x 5), shared by the STFT and
// iSTFT kernels.  The reference's frame length is 400 (tf.signal.stft(wav, 400, 160,
// fft_length=400), SN/apply.py:368-371), so a radix-2 transform cannot reproduce its 201 bins.
//
// Everything here is plain inline arithmetic on cplx so the same code compiles for the device
// (hipcc) and for the host (g++; tests/test_fft_host.py checks it against numpy on CPU).
#pragma once

#if defined(__HIPCC__)
#define NH_HD __host__ __device__ __forceinline__
#else
#define NH_HD inline
#endif

namespace nhans {

#if defined(__HIPCC__)
// A native 2-vector on the device: re/im live in an aligned register pair from the start, so the packed-f32
// instructions (v_pk_add_f32 / v_pk_mul_f32 / v_pk_fma_f32) apply without register shuffling.
typedef float cplx __attribute__((ext_vector_type(2)));
NH_HD cplx cmake(float x, float y) { return cplx{x, y}; }
NH_HD cplx cadd(cplx a, cplx b) { return a + b; }
NH_HD cplx csub(cplx a, cplx b) { return a - b; }
#else
struct cplx {
    float x, y;
};

NH_HD cplx cmake(float x, float y) { cplx r; r.x = x; r.y = y; return r; }
NH_HD cplx cadd(cplx a, cplx b) { return cmake(a.x + b.x, a.y + b.y); }
NH_HD cplx csub(cplx a, cplx b) { return cmake(a.x - b.x, a.y - b.y); }
#endif
NH_HD cplx cmul(cplx a, cplx b) { return cmake(a.x * b.x - a.y * b.y, a.x * b.y + a.y * b.x); }
NH_HD cplx cconj(cplx a) { return cmake(a.x, -a.y); }
// multiply by -i (forward) or +i (inverse)
template <bool INV> NH_HD cplx mul_mi(cplx a) { return INV ? cmake(-a.y, a.x) : cmake(a.y, -a.x); }

// cos/sin(2*pi*j/20), j = 0..19
#define NH_C20 {1.0f, 0.95105651629515357f, 0.80901699437494742f, 0.58778525229247313f, \
    0.30901699437494742f, 0.0f, -0.30901699437494742f, -0.58778525229247313f, \
    -0.80901699437494742f, -0.95105651629515357f, -1.0f, -0.95105651629515357f, \
    -0.80901699437494742f, -0.58778525229247313f, -0.30901699437494742f, 0.0f, \
    0.30901699437494742f, 0.58778525229247313f, 0.80901699437494742f, 0.95105651629515357f}
#define NH_S20 {0.0f, 0.30901699437494742f, 0.58778525229247313f, 0.80901699437494742f, \
    0.95105651629515357f, 1.0f, 0.95105651629515357f, 0.80901699437494742f, \
    0.58778525229247313f, 0.30901699437494742f, 0.0f, -0.30901699437494742f, \
    -0.58778525229247313f, -0.80901699437494742f, -0.95105651629515357f, -1.0f, \
    -0.95105651629515357f, -0.80901699437494742f, -0.58778525229247313f, -0.30901699437494742f}

// 5-point DFT, in place.  Forward kernel e^{-2 pi i nk/5}; INV conjugates it.
template <bool INV> NH_HD void dft5(cplx& x0, cplx& x1, cplx& x2, cplx& x3, cplx& x4) {
    const float c1 = 0.30901699437494742f, c2 = -0.80901699437494742f;   // cos(2pi/5), cos(4pi/5)
    const float s1 = 0.95105651629515357f, s2 = 0.58778525229247313f;    // sin(2pi/5), sin(4pi/5)
    cplx t1 = cadd(x1, x4), t2 = cadd(x2, x3), t3 = csub(x1, x4), t4 = csub(x2, x3);
    cplx m1 = cmake(x0.x + c1 * t1.x + c2 * t2.x, x0.y + c1 * t1.y + c2 * t2.y);
    cplx m2 = cmake(x0.x + c2 * t1.x + c1 * t2.x, x0.y + c2 * t1.y + c1 * t2.y);
    cplx sa = cmake(s1 * t3.x + s2 * t4.x, s1 * t3.y + s2 * t4.y);
    cplx sb = cmake(s2 * t3.x - s1 * t4.x, s2 * t3.y - s1 * t4.y);
    cplx ra = mul_mi<INV>(sa), rb = mul_mi<INV>(sb);     // -i*sa (fwd) / +i*sa (inv)
    x0 = cmake(x0.x + t1.x + t2.x, x0.y + t1.y + t2.y);
    x1 = cadd(m1, ra);
    x4 = csub(m1, ra);
    x2 = cadd(m2, rb);
    x3 = csub(m2, rb);
}

// 4-point DFT, in place.
template <bool INV> NH_HD void dft4(cplx& x0, cplx& x1, cplx& x2, cplx& x3) {
    cplx a = cadd(x0, x2), b = csub(x0, x2), c = cadd(x1, x3), d = csub(x1, x3);
    cplx rd = mul_mi<INV>(d);
    x0 = cadd(a, c);
    x2 = csub(a, c);
    x1 = cadd(b, rd);
    x3 = csub(b, rd);
}

// 20-point DFT: in[n] (n = 0..19) -> out[k] (natural order), Cooley-Tukey 4 x 5:
//   n = 5*n1 + n2, k = k1 + 4*k2;  radix-4 over n1, twiddle W20^(n2*k1), radix-5 over n2.
template <bool INV> NH_HD void dft20(const cplx* in, cplx* out) {
    const float C20[20] = NH_C20;
    const float S20[20] = NH_S20;
    cplx y[5][4];
#pragma unroll
    for (int n2 = 0; n2 < 5; ++n2) {
        cplx a0 = in[n2], a1 = in[5 + n2], a2 = in[10 + n2], a3 = in[15 + n2];
        dft4<INV>(a0, a1, a2, a3);
        y[n2][0] = a0; y[n2][1] = a1; y[n2][2] = a2; y[n2][3] = a3;
    }
#pragma unroll
    for (int n2 = 1; n2 < 5; ++n2) {
#pragma unroll
        for (int k1 = 1; k1 < 4; ++k1) {
            const int j = n2 * k1;                         // < 20
            cplx w = cmake(C20[j], INV ? S20[j] : -S20[j]);
            y[n2][k1] = cmul(y[n2][k1], w);
        }
    }
#pragma unroll
    for (int k1 = 0; k1 < 4; ++k1) {
        cplx b0 = y[0][k1], b1 = y[1][k1], b2 = y[2][k1], b3 = y[3][k1], b4 = y[4][k1];
        dft5<INV>(b0, b1, b2, b3, b4);
        out[k1] = b0; out[k1 + 4] = b1; out[k1 + 8] = b2; out[k1 + 12] = b3; out[k1 + 16] = b4;
    }
}

// 20-point FORWARD DFT of a REAL sequence, bins 0..10 only (the rest are their conjugates): the first pass of the
// analysis transform.  Same 4 x 5 factorisation as dft20: n = 5*n1 + n2, k = k1 + 4*k2.  On real input the
// radix-4 stage gives y[n2][0], y[n2][2] real and y[n2][3] = conj(y[n2][1]), so
//   k1 = 0 : a real-input 5-point DFT            -> bins 0, 4, 8
//   k1 = 2 : real values times W20^(2*n2)        -> bins 2, 6, 10
//   k1 = 1 : the one full complex 5-point DFT    -> bins 1, 5, 9, 13, 17; bins 7 and 3 are conj(13), conj(17)
// and the k1 = 3 branch is never computed.  Everything is written through the complex helpers with literal zero
// imaginary parts; the compiler folds those and drops the unused outputs.
NH_HD void rdft20_half(const float* in /*20 reals*/, cplx* out /*11: bins 0..10*/) {
    const float C20[20] = NH_C20;
    const float S20[20] = NH_S20;
    float r0[5], r2[5];          // y[n2][0], y[n2][2] (real)
    cplx c1[5];                  // y[n2][1]
#pragma unroll
    for (int n2 = 0; n2 < 5; ++n2) {
        const float a = in[n2] + in[10 + n2], b = in[n2] - in[10 + n2];
        const float c = in[5 + n2] + in[15 + n2], d = in[5 + n2] - in[15 + n2];
        r0[n2] = a + c;
        r2[n2] = a - c;
        c1[n2] = cmake(b, -d);                                   // b - i d
    }
    // k1 = 0
    {
        cplx b0 = cmake(r0[0], 0.f), b1 = cmake(r0[1], 0.f), b2 = cmake(r0[2], 0.f), b3 = cmake(r0[3], 0.f), b4 = cmake(r0[4], 0.f);
        dft5<false>(b0, b1, b2, b3, b4);
        out[0] = b0; out[4] = b1; out[8] = b2;
    }
    // k1 = 2: twiddle W20^(2*n2) on a real value
    {
        cplx b[5];
        b[0] = cmake(r2[0], 0.f);
#pragma unroll
        for (int n2 = 1; n2 < 5; ++n2) b[n2] = cmake(r2[n2] * C20[2 * n2], -r2[n2] * S20[2 * n2]);
        dft5<false>(b[0], b[1], b[2], b[3], b[4]);
        out[2] = b[0]; out[6] = b[1]; out[10] = b[2];
    }
    // k1 = 1: twiddle W20^(n2), full 5-point DFT
    {
        cplx b[5];
        b[0] = c1[0];
#pragma unroll
        for (int n2 = 1; n2 < 5; ++n2) b[n2] = cmul(c1[n2], cmake(C20[n2], -S20[n2]));
        dft5<false>(b[0], b[1], b[2], b[3], b[4]);
        out[1] = b[0]; out[5] = b[1]; out[9] = b[2];
        out[7] = cconj(b[3]);                                    // bin 13
        out[3] = cconj(b[4]);                                    // bin 17
    }
}

// The 400-point transform is two passes of 20-point DFTs around a 20x20 transpose:
//   n = 20*n1 + n2, k = k1 + 20*k2
//   pass 1 (one lane per n2): Y[k1] = DFT20_{n1}(x[20*n1 + n2]);  T[k1][n2] = Y[k1] * W400^(n2*k1)
//   pass 2 (one lane per k1): X[k1 + 20*k2] = DFT20_{n2}(T[k1][n2])
// tw400[j] = exp(-2 pi i j/400) (forward); the inverse uses its conjugate.
template <bool INV> NH_HD void fft400_pass1(const cplx* col /*20, n1-major*/, int n2,
                                            const cplx* tw400, cplx* out /*20, by k1*/) {
    cplx y[20];
    dft20<INV>(col, y);
#pragma unroll
    for (int k1 = 0; k1 < 20; ++k1) {
        cplx w = tw400[n2 * k1];
        if (INV) w = cconj(w);
        out[k1] = cmul(y[k1], w);
    }
}

// Pass 1 of the ANALYSIS transform of a real frame: rows k1 = 0..10 only.  Pass 2 on those rows yields every bin
// k1 + 20*k2; the bins of rows 11..19 below 201 are the conjugates of bins 400-k of rows 1..9.
NH_HD void fft400_pass1_real(const float* col /*20 reals, n1-major*/, int n2, const cplx* tw400, cplx* out /*11, by k1*/) {
    cplx y[11];
    rdft20_half(col, y);
    out[0] = y[0];
#pragma unroll
    for (int k1 = 1; k1 < 11; ++k1) out[k1] = cmul(y[k1], tw400[n2 * k1]);
}

template <bool INV> NH_HD void fft400_pass2(const cplx* row /*20, by n2*/, cplx* out /*20, by k2*/) {
    dft20<INV>(row, out);
}

}  // namespace nhans
