// Host build of the 400-point FFT building blocks (fft400.h) so the transform the HIP kernels
// run can be checked against numpy on a machine without a GPU (tests/test_fft_host.py).
#include <cmath>
#include "fft400.h"

using namespace nhans;

extern "C" void nhans_fft400_host(const float* in /*400 x (re,im)*/, float* out /*400 x (re,im)*/,
                                  int inverse) {
    static cplx tw[400];
    static bool init = false;
    if (!init) {
        for (int j = 0; j < 400; ++j) {
            double a = -2.0 * M_PI * j / 400.0;
            tw[j] = cmake((float)std::cos(a), (float)std::sin(a));
        }
        init = true;
    }
    static cplx T[20][20];
    for (int n2 = 0; n2 < 20; ++n2) {
        cplx col[20], y[20];
        for (int n1 = 0; n1 < 20; ++n1) col[n1] = cmake(in[2 * (20 * n1 + n2)], in[2 * (20 * n1 + n2) + 1]);
        if (inverse) fft400_pass1<true>(col, n2, tw, y); else fft400_pass1<false>(col, n2, tw, y);
        for (int k1 = 0; k1 < 20; ++k1) T[k1][n2] = y[k1];
    }
    for (int k1 = 0; k1 < 20; ++k1) {
        cplx x[20];
        if (inverse) fft400_pass2<true>(T[k1], x); else fft400_pass2<false>(T[k1], x);
        for (int k2 = 0; k2 < 20; ++k2) { out[2 * (k1 + 20 * k2)] = x[k2].x; out[2 * (k1 + 20 * k2) + 1] = x[k2].y; }
    }
}

// Real-input analysis transform as the STFT kernel runs it: pass 1 on rows 0..10 only, pass 2, conjugate mirror.
extern "C" void nhans_rfft400_host(const float* in /*400 reals*/, float* out /*201 x (re,im)*/) {
    static cplx tw[400];
    static bool init = false;
    if (!init) {
        for (int j = 0; j < 400; ++j) {
            double a = -2.0 * M_PI * j / 400.0;
            tw[j] = cmake((float)std::cos(a), (float)std::sin(a));
        }
        init = true;
    }
    static cplx T[11][20];
    for (int n2 = 0; n2 < 20; ++n2) {
        float col[20];
        cplx y[11];
        for (int n1 = 0; n1 < 20; ++n1) col[n1] = in[20 * n1 + n2];
        fft400_pass1_real(col, n2, tw, y);
        for (int k1 = 0; k1 < 11; ++k1) T[k1][n2] = y[k1];
    }
    for (int k1 = 0; k1 < 11; ++k1) {
        cplx x[20];
        fft400_pass2<false>(T[k1], x);
        for (int k2 = 0; k2 < 20; ++k2) {
            const int k = k1 + 20 * k2;
            if (k <= 200) { out[2 * k] = x[k2].x; out[2 * k + 1] = x[k2].y; }
            else if (k1 >= 1 && k1 <= 9) { out[2 * (400 - k)] = x[k2].x; out[2 * (400 - k) + 1] = -x[k2].y; }
        }
    }
}
