/* tol_study.c -- relaxes one of the reference's maps on the CPU with the library's `tol` arithmetic (oracle/tol_checker.c,
 * the operation-for-operation CPU statement of cell_update.h: tol_*) and reports how the run ends against the
 * reference's converged field: does the reference's own test max |du| < eps fire (under Jacobi or red-black), after how
 * many iterations, how far is the field from the reference's.  This is the harness the mode was developed with before any
 * kernel existed; variants were tried by editing tol_checker.c.  Results of the shipped arithmetic,
 * Jacobi, eps = 1e-6, stagger 100 (reference: 23 801 / 52 101 / 94 401 half-sweeps):
 *     basic.png  23 801 sweeps, final delta 4.8e-7, max rel 3.3e-6 (max abs 4.6e-5)
 *     maze.png   52 001 sweeps, final delta 0,      max rel 1.4e-6 (max abs 6.7e-4)
 *     umass.png  93 301 sweeps, final delta 1.2e-7, max rel 1.5e-5 (max abs 2.4e-4)
 * Variants measured on the way (umass, max abs, sign = above / below the reference's field): without the rounding of l to
 * f32 +5.8e-4; with it and N ln2 - mx in f64 -2.4e-4; last addition of the sum exact -1.3e-5 on basic (unchanged).
 *
 * Build:  gcc -O3 -std=c11 -fopenmp tools/tol_study.c -o /tmp/tol_study -lm
 * Usage:  tol_study <m0> <m1> <u0.f32> <locked.u32> <golden.f32> [0 = Jacobi | 1 = red-black]
 *         (raw little-endian arrays; tests/_oracle.py: load_png_reference_rule writes them from the PNGs, the goldens are
 *          tests/golden/maps_converged.npz)
 */
#include "../oracle/tol_checker.c"
#include <stdio.h>

int main(int argc, char **argv)
{
    if (argc < 6) { fprintf(stderr, "usage: tol_study <m0> <m1> <u0.f32> <locked.u32> <golden.f32> [scheme]\n"); return 2; }
    unsigned m[2] = {(unsigned)atoi(argv[1]), (unsigned)atoi(argv[2])};
    const size_t cells = (size_t)m[0] * m[1];
    float *a = malloc(cells * 4), *g = malloc(cells * 4);
    unsigned *lk = malloc(cells * 4);
    FILE *f = fopen(argv[3], "rb"); if (!f || fread(a, 4, cells, f) != cells) return 2; fclose(f);
    f = fopen(argv[4], "rb"); if (!f || fread(lk, 4, cells, f) != cells) return 2; fclose(f);
    f = fopen(argv[5], "rb"); if (!f || fread(g, 4, cells, f) != cells) return 2; fclose(f);
    const int scheme = argc > 6 ? atoi(argv[6]) : 0;
    TolHarmonic h = {2, m, a, lk, 1e-6f, 0.0f, 100, 0, NULL, NULL, NULL, NULL};
    const int rc = oracle_tol_complete(&h, scheme);
    double worst = 0, wabs = 0, sum = 0; long nbad = 0, n = 0, moved = 0;
    for (size_t i = 0; i < cells; i++) {
        if (lk[i]) continue;
        if (g[i] <= -9e5f) { moved += a[i] != g[i]; continue; }
        const double e = (double)a[i] - g[i], rel = fabs(e) / fmax(1.0, fabs((double)g[i]));
        if (rel > worst) worst = rel;
        if (fabs(e) > wabs) wabs = fabs(e);
        nbad += rel > 1e-5; sum += e; n++;
    }
    printf("tol %s: rc %d (0 = stopped by max |du| < eps), %u iterations, final delta %.3e; vs the reference's field: max rel %.3e, "
           "max abs %.3e, mean signed %.3e, cells over 1e-5: %ld, unreached cells that moved: %ld\n",
           scheme ? "red-black" : "Jacobi", rc, h.currentIteration, h.delta, worst, wabs, sum / n, nbad, moved);
    return 0;
}
