// harmonic_legacy_cpu.cpp -- the linear-space (non-log) SOR solver of the reference, kept for ABI completeness.
//
// libepic/src/harmonic/harmonic_legacy_cpu.cpp:36-141: lexicographic successive over-relaxation on a w x h grid in
// float, double and long double; at least MIN_ITERATIONS (10000) sweeps; stops when max |du| < epsilon.  It is the
// comparison baseline of the paper, not on the log-space path, and is never accelerated here; the python wrapper
// binds the symbols at import (libepic/python/epic/epic_harmonic.py:105-124), so a drop-in library has to have them.
// One template replaces the reference's three copies; results are bit-identical (tests/test_path_cpu.py).
#include <cmath>
#include <cstddef>

#include "../../include/epic/epic_abi.h"

namespace {

constexpr unsigned kMinIterations = 10000;  // MIN_ITERATIONS, harmonic_legacy_cpu.cpp:34

template <typename T>
int sor_2d(unsigned w, unsigned h, T epsilon, T omega, const unsigned *locked, T *u, unsigned &iter)
{
    T delta = epsilon + T(1);
    iter = 0;
    while (delta >= epsilon || iter < kMinIterations) {
        delta = T(0);
        for (unsigned y = 1; y + 1 < h; y++) {
            T *row = u + (size_t)y * w;
            const unsigned *lk = locked + (size_t)y * w;
            for (unsigned x = 1; x + 1 < w; x++) {
                if (lk[x] == 1) continue;
                const T before = row[x];
                // (1 - w) u + w/4 (up + down + left + right), summed in the reference's order
                row[x] = (T(1) - omega) * row[x] + omega / T(4) * (row[(ptrdiff_t)x - (ptrdiff_t)w] + row[x + w] + row[x - 1] + row[x + 1]);
                delta = std::fmax(delta, std::fabs(row[x] - before));
            }
        }
        iter++;
    }
    return EPIC_SUCCESS;
}

}  // namespace

namespace epic {
extern "C" {

int harmonic_legacy_sor_2d_float_cpu(unsigned int w, unsigned int h, float epsilon, float omega, unsigned int *locked,
                                     float *u, unsigned int &iter)
{
    return sor_2d<float>(w, h, epsilon, omega, locked, u, iter);
}

int harmonic_legacy_sor_2d_double_cpu(unsigned int w, unsigned int h, double epsilon, double omega,
                                      unsigned int *locked, double *u, unsigned int &iter)
{
    return sor_2d<double>(w, h, epsilon, omega, locked, u, iter);
}

int harmonic_legacy_sor_2d_long_double_cpu(unsigned int w, unsigned int h, long double epsilon, long double omega,
                                           unsigned int *locked, long double *u, unsigned int &iter)
{
    return sor_2d<long double>(w, h, epsilon, omega, locked, u, iter);
}

}  // extern "C"
}  // namespace epic
