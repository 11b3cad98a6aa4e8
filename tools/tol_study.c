/* tol_study.c -- CPU emulation, operation for operation, of the `tol` math mode (cell_update.h: split_potential /
 * tol_update) before and beside the kernel, to answer on the reference's ill-conditioned maps:
 *   1. does a Jacobi (or red-black) relaxation with it stop by the reference's own test (max |du| < eps, absolute)?
 *   2. how far from the reference's converged field does it stop (bar: 1e-5 * max(1, |u|))?
 *
 * The mode.  The reference evaluates per cell  u' = mx + ln(sum_i e^(u_i - mx)) - ln 4  with four expf and one logf
 * (harmonic_cpu.cpp:60-70).  Here every cell's potential is split ONCE per sweep into  e^u = q 2^n  (n = rint(u log2 e),
 * q = 2^f in [0.707, 1.414] by a degree-7 f32 polynomial) and the four neighbours of a cell reuse those pairs:
 *     N = max n_i,   S = ((q_a 2^(n_a-N) + q_b 2^(n_b-N)) + q_c 2^(n_c-N)) + q_d 2^(n_d-N)     f32, reference order
 *     t = (float)(N ln2 + ln S)   f64 inside, one rounding to f32 where the reference rounds mx + ln s
 *     u' = (float)((double)t - ln 4)                                                          as the reference
 * One exp-class evaluation and one log per cell instead of four and one; no subtraction of the maximum, so the update
 * is a composition of monotone maps in every argument (what makes Jacobi's two interleaved chains end in one fixed point).
 *
 * Build:  gcc -O2 -fopenmp -ffp-contract=off -mfma tools/tol_study.c -o /tmp/tol_study -lm
 * Usage:  tol_study stats
 *         tol_study <m0> <m1> <u0.f32> <locked.u32> <golden.f32> [jacobi|redblack] [eps] [degree 6|7]
 */
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

static inline uint32_t f2u(float f) { uint32_t u; memcpy(&u, &f, 4); return u; }
static inline float u2f(uint32_t u) { float f; memcpy(&f, &u, 4); return f; }

static int g_degree = 7;

/* log2(e) = LH + LL to ~49 bits */
#define LH 0x1.715476p+0f
#define LL 0x1.4ae0bep-26f
#define MAGIC 12582912.0f /* 1.5 * 2^23 */

static inline void split(float u, float *q, int32_t *n)
{
    const float zm = fmaf(u, LH, MAGIC);
    const float nf = zm - MAGIC;
    *n = (int32_t)(f2u(zm) - f2u(MAGIC));
    float f = fmaf(u, LH, -nf);
    f = fmaf(u, LL, f);
    float p;
    if (g_degree == 7) {
        p = fmaf(f, 0x1.e5ba06p-17f, 0x1.44227cp-13f);
        p = fmaf(f, p, 0x1.5da0f4p-10f);
        p = fmaf(f, p, 0x1.3b2a4ap-7f);
        p = fmaf(f, p, 0x1.c6b072p-5f);
        p = fmaf(f, p, 0x1.ebfbep-3f);
        p = fmaf(f, p, 0x1.62e43p-1f);
    } else {
        p = fmaf(f, 0x1.3f906cp-13f, 0x1.5f0a66p-10f);
        p = fmaf(f, p, 0x1.3b30bp-7f);
        p = fmaf(f, p, 0x1.c6af78p-5f);
        p = fmaf(f, p, 0x1.ebfbd8p-3f);
        p = fmaf(f, p, 0x1.62e43p-1f);
    }
    *q = fmaf(f, p, 1.0f);
}

/* glibc logf's table (e_logf_data.c), as in cell_update.h; evaluated in double, NOT rounded to float */
static const double kLogTab[32] = {
    0x1.661ec79f8f3bep+0, -0x1.57bf7808caadep-2, 0x1.571ed4aaf883dp+0, -0x1.2bef0a7c06ddbp-2,
    0x1.49539f0f010bp+0,  -0x1.01eae7f513a67p-2, 0x1.3c995b0b80385p+0, -0x1.b31d8a68224e9p-3,
    0x1.30d190c8864a5p+0, -0x1.6574f0ac07758p-3, 0x1.25e227b0b8eap+0,  -0x1.1aa2bc79c81p-3,
    0x1.1bb4a4a1a343fp+0, -0x1.a4e76ce8c0e5ep-4, 0x1.12358f08ae5bap+0, -0x1.1973c5a611cccp-4,
    0x1.0953f419900a7p+0, -0x1.252f438e10c1ep-5, 0x1p+0,               0x0p+0,
    0x1.e608cfd9a47acp-1, 0x1.aa5aa5df25984p-5,  0x1.ca4b31f026aap-1,  0x1.c5e53aa362eb4p-4,
    0x1.b2036576afce6p-1, 0x1.526e57720db08p-3,  0x1.9c2d163a1aa2dp-1, 0x1.bc2860d22477p-3,
    0x1.886e6037841edp-1, 0x1.1058bc8a07ee1p-2,  0x1.767dcf5534862p-1, 0x1.4043057b6ee09p-2,
};
static inline double ln_d(float sf)
{
    const uint32_t tmp = f2u(sf) - 0x3f330000u;
    const int i = (tmp >> 19) & 15, k = (int32_t)tmp >> 23;
    const double invc = ldexp(kLogTab[2 * i], -k), y0 = kLogTab[2 * i + 1] + (double)k * 0x1.62e42fefa39efp-1;
    const double r = fma((double)sf, invc, -1.0);
    const double r2 = r * r;
    double y = fma(0x1.5575b0be00b6ap-2, r, -0x1.ffffef20a4123p-2);
    y = fma(-0x1.00ea348b88334p-2, r2, y);
    return fma(y, r2, y0 + r);
}

static inline int32_t imax(int32_t a, int32_t b) { return a > b ? a : b; }
static inline float fmax2(float a, float b) { return a < b ? b : a; }

static inline float tol_update(float qa, int32_t na, float qb, int32_t nb, float qc, int32_t nc, float qd, int32_t nd)
{
    const int32_t N = imax(imax(imax(na, nb), nc), nd);
    float s = ldexpf(qa, na - N) + ldexpf(qb, nb - N);
    s = s + ldexpf(qc, nc - N);
    s = s + ldexpf(qd, nd - N);
    const double t64 = fma((double)N, 0x1.62e42fefa39efp-1, ln_d(s));
    const float t = (float)t64;
    return (float)((double)t - 0x1.62e42fefa39efp+0);
}

static void stats(void)
{
    /* error of q 2^n against e^u over the values the maps visit: u in [-100, 0] on the f32 grid, stride-sampled */
    double sum = 0, mx = 0; uint64_t n = 0;
    for (uint32_t b = f2u(-1e-3f); b <= f2u(-100.0f); b += 37) {
        float u = u2f(b), q; int32_t e;
        split(u, &q, &e);
        const double ref = exp((double)u), got = ldexp((double)q, e);
        const double ulp = ldexp(1.0, e - 23) * (q >= 1.0f ? 1.0 : 0.5);
        const double err = (got - ref) / ulp;
        sum += err; if (fabs(err) > mx) mx = fabs(err); n++;
    }
    printf("split (degree %d), u in [-100, -1e-3]: n=%lu  mean err %+.5f ulp  max %.4f ulp\n", g_degree, n, sum / n, mx);
}

int main(int argc, char **argv)
{
    if (argc < 6) {
        for (g_degree = 6; g_degree <= 7; g_degree++) stats();
        return 0;
    }
    const unsigned m0 = atoi(argv[1]), m1 = atoi(argv[2]);
    const size_t cells = (size_t)m0 * m1;
    float *a = malloc(cells * 4), *b = malloc(cells * 4), *g = malloc(cells * 4), *q = malloc(cells * 4);
    int32_t *n = malloc(cells * 4);
    unsigned *lk = malloc(cells * 4);
    FILE *f = fopen(argv[3], "rb"); if (!f || fread(a, 4, cells, f) != cells) return 2; fclose(f);
    f = fopen(argv[4], "rb"); if (!f || fread(lk, 4, cells, f) != cells) return 2; fclose(f);
    f = fopen(argv[5], "rb"); if (!f || fread(g, 4, cells, f) != cells) return 2; fclose(f);
    const int redblack = argc > 6 && strcmp(argv[6], "redblack") == 0;
    const float eps = argc > 7 ? (float)atof(argv[7]) : 1e-6f;
    if (argc > 8) g_degree = atoi(argv[8]);
    const unsigned stagger = 100, mMax = m0 > m1 ? m0 : m1;
    unsigned it = 0;
    int conv = 0;
    float d = 0;
    while (!conv || it < mMax) {
        const int check = it % stagger == 0;
        d = 0;
#pragma omp parallel for schedule(static)
        for (size_t i = 0; i < cells; i++) split(a[i], &q[i], &n[i]);
        if (!redblack) memcpy(b, a, cells * 4);
#pragma omp parallel for schedule(static) reduction(max : d)
        for (unsigned r = 1; r < m0 - 1; r++)
            for (unsigned c = 1; c + 1 < m1; c++) {
                const size_t i = (size_t)r * m1 + c;
                if (lk[i]) continue;
                if (redblack && ((r + c + it) & 1u) == 0) continue;   /* harmonic_cpu.cpp:46-51: (x0 + x1 + iteration) odd */
                const float v = tol_update(q[i - m1], n[i - m1], q[i + m1], n[i + m1], q[i - 1], n[i - 1], q[i + 1], n[i + 1]);
                d = fmax2(d, fabsf(a[i] - v));
                if (redblack) a[i] = v; else b[i] = v;
            }
        if (!redblack) { float *t = a; a = b; b = t; }
        it++;
        conv = check ? d < eps : 0;
        if (it % 10000 == 0) fprintf(stderr, "  iteration %u delta %.3e\n", it, d);
        if (it > 600000) break;
    }
    double worst = 0, worst_abs = 0; size_t nbad = 0, nseed = 0;
    for (size_t i = 0; i < cells; i++) {
        if (lk[i]) continue;
        if (g[i] <= -9e5f) { if (a[i] != g[i]) nseed++; continue; }
        const double e = fabs((double)a[i] - g[i]);
        const double rel = e / fmax(1.0, fabs((double)g[i]));
        if (rel > worst) worst = rel;
        if (e > worst_abs) worst_abs = e;
        if (rel > 1e-5) nbad++;
    }
    printf("tol %s degree %d: %u iterations, final delta %.3e; vs reference golden: max rel %.3e, max abs %.3e, cells over 1e-5: %zu, unreached cells that moved: %zu\n",
           redblack ? "red-black" : "Jacobi", g_degree, it, d, worst, worst_abs, nbad, nseed);
    return 0;
}
