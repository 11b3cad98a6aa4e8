// harmonic_cpu.cpp -- CPU half of the libepic C-ABI: red-black Gauss-Seidel log-space relaxation.
//
// Present because the drop-in boundary exports it (callers fall back to it on GPU-less hosts:
// src/epic_nav_core_plugin.cpp:258-263, libepic/python/epic/harmonic.py:76-107) -- the GPU entry points never
// route here.  Restates libepic/src/harmonic/harmonic_cpu.cpp:38-220: same colouring, same iteration rule and
// the same per-cell rounding sequence, so results are bit-identical to the reference on the same libm
// (tests/test_cpu_abi.py checks that against the compiled reference and the committed goldens).
#include <cmath>
#include <cstddef>
#include <cstdio>

#include "../../include/epic/epic_abi.h"

namespace {

using epic::Harmonic;

// The reference's expression (harmonic_cpu.cpp:60-70 / :110-123): float exp/log, left-associated float sum,
// float add of the max, then a DOUBLE subtraction of log(2n), rounded to float on store.
template <int NB>
inline float relax_cell(const float (&nb)[NB], double log2n)
{
    float mx = nb[0] < nb[1] ? nb[1] : nb[0];
    for (int i = 2; i < NB; i++) mx = mx < nb[i] ? nb[i] : mx;
    float acc = std::exp(nb[0] - mx) + std::exp(nb[1] - mx);
    for (int i = 2; i < NB; i++) acc = acc + std::exp(nb[i] - mx);
    const float t = mx + std::log(acc);
    return (float)((double)t - log2n);
}

// One colour of the 2-D checkerboard: rows 1..m0-2, in row x0 the columns 1+offset, 3+offset, ...
// with offset = (iteration parity != row parity)  (harmonic_cpu.cpp:46-51).
void half_sweep_2d(Harmonic *h, bool check)
{
    const unsigned rows = h->m[0], cols = h->m[1];
    const double log2n = std::log(2.0 * h->n);
    float worst = 0.0f;
    for (unsigned r = 1; r + 1 < rows; r++) {
        const unsigned first = 1 + (unsigned)((h->currentIteration % 2) != (r % 2));
        float *row = h->u + (size_t)r * cols;
        const unsigned *lk = h->locked + (size_t)r * cols;
        for (unsigned c = first; c + 1 < cols; c += 2) {
            if (lk[c]) continue;
            const float before = row[c];
            const float nb[4] = {row[(ptrdiff_t)c - (ptrdiff_t)cols], row[c + cols], row[c - 1], row[c + 1]};
            const float after = relax_cell<4>(nb, log2n);
            row[c] = after;
            if (check) {
                const float d = (float)std::fabs(before - after);
                if (worst < d) worst = d;
            }
        }
    }
    if (check) h->delta = worst;
}

// 3-D colouring (harmonic_cpu.cpp:89-102): the 2-D offset, flipped on even x1; neighbours in the order
// x0-1, x0+1, x1-1, x1+1, x2-1, x2+1 (harmonic_cpu.cpp:110-123).
void half_sweep_3d(Harmonic *h, bool check)
{
    const unsigned m0 = h->m[0], m1 = h->m[1], m2 = h->m[2];
    const ptrdiff_t s1 = m2, s0 = (ptrdiff_t)m1 * m2;
    const double log2n = std::log(2.0 * h->n);
    float worst = 0.0f;
    for (unsigned a = 1; a + 1 < m0; a++) {
        for (unsigned b = 1; b + 1 < m1; b++) {
            unsigned offset = (unsigned)((h->currentIteration % 2) != (a % 2));
            if (b % 2 == 0) offset = 1u - offset;
            float *line = h->u + a * s0 + b * s1;
            const unsigned *lk = h->locked + a * s0 + b * s1;
            for (unsigned c = 1 + offset; c + 1 < m2; c += 2) {
                if (lk[c]) continue;
                const float before = line[c];
                const float *p = line + c;
                const float nb[6] = {p[-s0], p[s0], p[-s1], p[s1], p[-1], p[1]};
                const float after = relax_cell<6>(nb, log2n);
                line[c] = after;
                if (check) {
                    const float d = (float)std::fabs(before - after);
                    if (worst < d) worst = d;
                }
            }
        }
    }
    if (check) h->delta = worst;
}

void half_sweep(Harmonic *h, bool check)
{
    if (h->n == 2) half_sweep_2d(h, check);
    else if (h->n == 3) half_sweep_3d(h, check);
    // n == 4: nothing to do, as in the reference (harmonic_cpu.cpp:193-195); any other n likewise.
}

}  // namespace

namespace epic {
extern "C" {

int harmonic_update_cpu(Harmonic *harmonic)  // harmonic_cpu.cpp:187-200
{
    half_sweep(harmonic, false);
    harmonic->currentIteration++;
    return EPIC_SUCCESS;
}

int harmonic_update_and_check_cpu(Harmonic *harmonic)  // harmonic_cpu.cpp:203-220
{
    if (harmonic->n == 2 || harmonic->n == 3) half_sweep(harmonic, true);
    harmonic->currentIteration++;
    return harmonic->delta < harmonic->epsilon ? EPIC_SUCCESS_AND_CONVERGED : EPIC_SUCCESS;
}

int harmonic_complete_cpu(Harmonic *harmonic)  // harmonic_cpu.cpp:136-184
{
    if (harmonic == nullptr || harmonic->m == nullptr || harmonic->u == nullptr || harmonic->locked == nullptr ||
        harmonic->epsilon <= 0.0) {
        fprintf(stderr, "Error[harmonic_complete_cpu]: %s\n", "Invalid data.");
        return EPIC_ERROR_INVALID_DATA;
    }
    if (harmonic->numIterationsToStaggerCheck == 0) {  // the reference would divide by zero (harmonic_cpu.cpp:159)
        fprintf(stderr, "Error[harmonic_complete_cpu]: %s\n", "Invalid data (numIterationsToStaggerCheck must be positive).");
        return EPIC_ERROR_INVALID_DATA;
    }
    unsigned longest = 0;  // information needs max(m[i]) sweeps to cross the grid
    for (unsigned i = 0; i < harmonic->n; i++) longest = harmonic->m[i] > longest ? harmonic->m[i] : longest;

    harmonic->currentIteration = 0;
    harmonic->delta = harmonic->epsilon + 1.0;
    int status = EPIC_SUCCESS;
    // A plain sweep returns SUCCESS and so clears a previous CONVERGED: the loop can only end right after a
    // check sweep (harmonic_cpu.cpp:158-173).
    while (status != EPIC_SUCCESS_AND_CONVERGED || harmonic->currentIteration < longest) {
        const bool check = harmonic->currentIteration % harmonic->numIterationsToStaggerCheck == 0;
        status = check ? harmonic_update_and_check_cpu(harmonic) : harmonic_update_cpu(harmonic);
    }
    return EPIC_SUCCESS;
}

// Sparse edits on the host arrays: libepic/src/harmonic/harmonic_utilities_cpu.cpp:38-76.
int harmonic_utilities_set_cells_2d_cpu(Harmonic *harmonic, unsigned int k, unsigned int *v, unsigned int *types)
{
    if (harmonic == nullptr || harmonic->n == 0 || harmonic->m == nullptr || harmonic->u == nullptr ||
        harmonic->locked == nullptr || k == 0 || v == nullptr || types == nullptr) {
        fprintf(stderr, "Error[harmonic_utilities_set_cells_2d_cpu]: %s\n", "Invalid data.");
        return EPIC_ERROR_INVALID_DATA;
    }
    const unsigned rows = harmonic->m[0], cols = harmonic->m[1];
    for (unsigned i = 0; i < k; i++) {
        const unsigned x = v[2 * i], y = v[2 * i + 1];
        if (y >= rows || x >= cols) {
            fprintf(stderr, "Warning[harmonic_utilities_set_cells_2d_cpu]: %s\n",
                    "Provided vector has invalid values outside area.");
            continue;
        }
        const size_t cell = (size_t)y * cols + x;
        switch (types[i]) {
        case EPIC_CELL_TYPE_GOAL:
            harmonic->u[cell] = EPIC_LOG_SPACE_GOAL;
            harmonic->locked[cell] = 1;
            break;
        case EPIC_CELL_TYPE_OBSTACLE:
            harmonic->u[cell] = EPIC_LOG_SPACE_OBSTACLE;
            harmonic->locked[cell] = 1;
            break;
        case EPIC_CELL_TYPE_FREE:
            harmonic->u[cell] = EPIC_LOG_SPACE_FREE;
            harmonic->locked[cell] = 0;
            break;
        default:
            fprintf(stderr, "Warning[harmonic_utilities_set_cells_2d_cpu]: %s\n", "Type is invalid. No change made.");
            break;
        }
    }
    return EPIC_SUCCESS;
}

}  // extern "C"
}  // namespace epic
