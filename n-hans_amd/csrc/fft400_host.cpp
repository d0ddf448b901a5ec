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
