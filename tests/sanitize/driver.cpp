// Sanitizer driver (host code only): every CPU export of libepic and the checker, on small inputs, under
// -fsanitize=address,undefined (tests/test_sanitizers.py builds and runs it).  The reference ships no sanitizer runs
// (SURVEY.md §5); GPU AddressSanitizer is not available on the pool, so this covers the host half of the library.
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>

#include <epic/epic_abi.h>

using namespace epic;

extern "C" {
struct OracleHarmonic;  // same layout as Harmonic
int oracle_complete(Harmonic *h);
int oracle_jacobi_run(Harmonic *h, unsigned int sweeps);
int oracle_jacobi_complete(Harmonic *h);
}

static int failures = 0;
#define EXPECT(cond)                                                     \
    do {                                                                 \
        if (!(cond)) { fprintf(stderr, "EXPECT failed: %s (line %d)\n", #cond, __LINE__); failures++; } \
    } while (0)

static void grid2d(unsigned rows, unsigned cols, std::vector<float> &u, std::vector<unsigned> &lk)
{
    u.assign((size_t)rows * cols, -1e6f);
    lk.assign((size_t)rows * cols, 0u);
    unsigned s = 12345u;
    for (unsigned r = 0; r < rows; r++)
        for (unsigned c = 0; c < cols; c++) {
            s = s * 1664525u + 1013904223u;
            const bool border = r == 0 || c == 0 || r == rows - 1 || c == cols - 1;
            if (border || (s >> 24) < 16) lk[(size_t)r * cols + c] = 1;
        }
    const size_t goal = (size_t)(rows / 2) * cols + cols / 2;
    lk[goal] = 1;
    u[goal] = 0.0f;
}

int main()
{
    // ---- 2-D log-space solver, edits, streamline ----
    unsigned m2[2] = {40, 60};
    std::vector<float> u;
    std::vector<unsigned> lk;
    grid2d(m2[0], m2[1], u, lk);
    Harmonic h = {};
    h.n = 2; h.m = m2; h.u = u.data(); h.locked = lk.data(); h.epsilon = 1e-4f; h.numIterationsToStaggerCheck = 10;
    EXPECT(harmonic_update_cpu(&h) == EPIC_SUCCESS);
    EXPECT(harmonic_update_and_check_cpu(&h) <= EPIC_SUCCESS_AND_CONVERGED);
    EXPECT(harmonic_complete_cpu(&h) == EPIC_SUCCESS);
    unsigned v[6] = {5, 5, 30, 20, 59, 39};
    unsigned types[3] = {EPIC_CELL_TYPE_GOAL, EPIC_CELL_TYPE_OBSTACLE, EPIC_CELL_TYPE_FREE};
    EXPECT(harmonic_utilities_set_cells_2d_cpu(&h, 3, v, types) == EPIC_SUCCESS);
    EXPECT(harmonic_complete_cpu(&h) == EPIC_SUCCESS);
    float pot = 0, gx = 0, gy = 0;
    EXPECT(harmonic_compute_potential_2d_cpu(&h, 10.3f, 12.7f, pot) <= EPIC_ERROR_INVALID_LOCATION);
    EXPECT(harmonic_compute_gradient_2d_cpu(&h, 10.3f, 12.7f, 0.4f, gx, gy) <= EPIC_ERROR_INVALID_PATH);
    for (float sx : {3.0f, 17.5f, 44.2f, 57.0f})
        for (float sy : {2.0f, 11.1f, 25.0f, 37.5f}) {
            unsigned k = 0;
            float *path = nullptr;
            const int rc = harmonic_compute_path_2d_cpu(&h, sx, sy, 0.2f, 0.4f, 5000, k, path);
            if (rc == EPIC_SUCCESS) {
                EXPECT(k > 2 && path != nullptr);
                EXPECT(harmonic_free_path_cpu(path) == EPIC_SUCCESS && path == nullptr);
            } else {
                EXPECT(path == nullptr);
            }
        }
    // invalid input is refused, not dereferenced
    EXPECT(harmonic_complete_cpu(nullptr) == EPIC_ERROR_INVALID_DATA);
    Harmonic bad = {};   // n = 0: like the reference (harmonic_cpu.cpp:187-200) the single update only counts the iteration
    EXPECT(harmonic_update_cpu(&bad) == EPIC_SUCCESS && bad.currentIteration == 1);
    EXPECT(harmonic_complete_cpu(&bad) == EPIC_ERROR_INVALID_DATA);
    EXPECT(harmonic_utilities_set_cells_2d_cpu(&h, 0, v, types) == EPIC_ERROR_INVALID_DATA);

    // ---- 3-D ----
    unsigned m3[3] = {10, 12, 14};
    std::vector<float> u3((size_t)10 * 12 * 14, -1e6f);
    std::vector<unsigned> l3(u3.size(), 0u);
    for (unsigned a = 0; a < 10; a++)
        for (unsigned b = 0; b < 12; b++)
            for (unsigned c = 0; c < 14; c++)
                if (a == 0 || b == 0 || c == 0 || a == 9 || b == 11 || c == 13) l3[((size_t)a * 12 + b) * 14 + c] = 1;
    l3[((size_t)5 * 12 + 6) * 14 + 7] = 1;
    u3[((size_t)5 * 12 + 6) * 14 + 7] = 0.0f;
    Harmonic h3 = {};
    h3.n = 3; h3.m = m3; h3.u = u3.data(); h3.locked = l3.data(); h3.epsilon = 1e-4f; h3.numIterationsToStaggerCheck = 10;
    EXPECT(harmonic_complete_cpu(&h3) == EPIC_SUCCESS);

    // ---- the checker on the same kind of input ----
    grid2d(m2[0], m2[1], u, lk);
    h.currentIteration = 0;
    EXPECT(oracle_complete(&h) == 0);
    grid2d(m2[0], m2[1], u, lk);
    h.currentIteration = 0;
    EXPECT(oracle_jacobi_run(&h, 25) == 0);
    EXPECT(oracle_jacobi_complete(&h) == 0);

    // ---- legacy SOR (linear space) and its streamline ----
    const unsigned w = 48, hh = 36;
    std::vector<unsigned> ll((size_t)w * hh, 0u);
    for (unsigned y = 0; y < hh; y++)
        for (unsigned x = 0; x < w; x++)
            if (x == 0 || y == 0 || x == w - 1 || y == hh - 1 || (x == 20 && y > 8 && y < 30)) ll[(size_t)y * w + x] = 1;
    ll[(size_t)18 * w + 40] = 1;
    std::vector<float> uf((size_t)w * hh, 1.0f);
    std::vector<double> ud((size_t)w * hh, 1.0);
    std::vector<long double> ul((size_t)w * hh, 1.0L);
    uf[(size_t)18 * w + 40] = 0.0f; ud[(size_t)18 * w + 40] = 0.0; ul[(size_t)18 * w + 40] = 0.0L;
    unsigned it = 0;
    EXPECT(harmonic_legacy_sor_2d_float_cpu(w, hh, 1e-4f, 1.5f, ll.data(), uf.data(), it) == EPIC_SUCCESS);
    EXPECT(harmonic_legacy_sor_2d_double_cpu(w, hh, 1e-8, 1.5, ll.data(), ud.data(), it) == EPIC_SUCCESS);
    EXPECT(harmonic_legacy_sor_2d_long_double_cpu(w, hh, 1e-8L, 1.5L, ll.data(), ul.data(), it) == EPIC_SUCCESS);
    double dp = 0, dgx = 0, dgy = 0;
    EXPECT(harmonic_legacy_compute_potential_2d_cpu(w, hh, ll.data(), ud.data(), 5.5, 6.5, dp) == EPIC_SUCCESS);
    EXPECT(harmonic_legacy_compute_gradient_2d_cpu(w, hh, ll.data(), ud.data(), 5.5, 6.5, 0.4, dgx, dgy) == EPIC_SUCCESS);
    unsigned k = 0;
    double *dpath = nullptr;
    const int rc = harmonic_legacy_compute_path_2d_cpu(w, hh, ll.data(), ud.data(), 5.0, 6.0, 0.2, 0.4, 4000, 0, k, dpath);
    if (rc == EPIC_SUCCESS) EXPECT(harmonic_legacy_free_path_cpu(dpath) == EPIC_SUCCESS && dpath == nullptr);

    if (failures) return 1;
    printf("sanitize driver: ok\n");
    return 0;
}
