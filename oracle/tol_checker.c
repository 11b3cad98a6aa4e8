/*
 * tol_checker.c -- CPU statement of the library's `tol` math mode (epic_amd/csrc/cell_update.h: tol_split2 /
 * tol_update_2d / tol_update_3d), operation for operation.
 *
 * THIS IS TEST INFRASTRUCTURE, NOT PRODUCT CODE (same rule as harmonic_oracle.c: only tests/, __graft_entry__.smoke()
 * and bench.py's cpu_baseline leg may load liboracle.so; nothing under epic_amd/ links or calls it).
 *
 * What it is NOT: the reference's arithmetic.  The reference evaluates, per cell, 2n expf and one logf
 * (/root/reference/libepic/src/harmonic/harmonic_cpu.cpp:60-70, :110-123; restated in harmonic_oracle.c).  The tol mode
 * computes the same update u' = ln(sum_i e^(u_i)) - ln 2n from ONE split e^u = q 2^n per cell, shared by the cell's
 * neighbours, keeps every rounding stage of the reference (f32 sum, l to f32, mx + l to f32, f64 subtraction to f32) and
 * differs from it only in the noise inside the sum.
 * Parity with the REFERENCE is therefore a tolerance statement (1e-5 max(1, |u|), against the reference-generated
 * goldens in tests/golden/), tested on the device and here.  This file exists so that the KERNELS (tiling, strip seams,
 * halo splits, masks, red-black colouring, work lists) can still be checked at tolerance 0: the device result must equal
 * this code bit for bit, sweep for sweep.  tools/tol_study.c includes it to relax the reference's maps on the CPU.
 *
 * Needs correctly rounded fmaf / fma (libm's are) and no contraction (-std=c11 implies -ffp-contract=off).
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

typedef struct TolHarmonic {  /* the reference's struct layout, libepic/include/epic/harmonic/harmonic.h:44-64 */
    unsigned int n;
    unsigned int *m;
    float *u;
    unsigned int *locked;
    float epsilon;
    float delta;
    unsigned int numIterationsToStaggerCheck;
    unsigned int currentIteration;
    unsigned int *d_m;
    float *d_u;
    unsigned int *d_locked;
    float *d_delta;
} TolHarmonic;

static inline uint32_t tol_f2u(float f) { uint32_t u; memcpy(&u, &f, 4); return u; }

#define TOL_LH 0x1.715476p+0f   /* log2(e) = LH + LL to 49 bits */
#define TOL_LL 0x1.4ae0bep-26f
#define TOL_MAGIC 12582912.0f   /* 1.5 * 2^23 */
#define TOL_MAGIC_BITS 0x4b400000u

/* e^u = q 2^n; *zb = bit pattern of the float whose low mantissa bits are n (cell_update.h: tol_split2) */
static inline void tol_split(float u, float *q, uint32_t *zb)
{
    const float zm = fmaf(u, TOL_LH, TOL_MAGIC);
    const float nf = zm - TOL_MAGIC;
    float f = fmaf(u, TOL_LH, -nf);
    f = fmaf(u, TOL_LL, f);
    float p = fmaf(f, 0x1.e5ba06p-17f, 0x1.44227cp-13f);
    p = fmaf(f, p, 0x1.5da0f4p-10f);
    p = fmaf(f, p, 0x1.3b2a4ap-7f);
    p = fmaf(f, p, 0x1.c6b072p-5f);
    p = fmaf(f, p, 0x1.ebfbep-3f);
    p = fmaf(f, p, 0x1.62e43p-1f);
    *q = fmaf(f, p, 1.0f);
    *zb = tol_f2u(zm);
}

/* ln S for S in [0.5, 16): 256 sub-intervals per binade chosen by the top 8 mantissa bits; r = S invc - 1 is EXACT in one
 * f32 fma (invc = v / 512 2^-k with a 9-bit v: tools/gen_ln_table.py), log1p(r) = r + r^2 (-1/2 + r/3) in f32, the table's
 * ln c is added in double (cell_update.h: tol_ln_d).  Good to 2^-33 absolute, unbiased to 1e-12. */
typedef struct { double lnc; float invc; } TolLnEntry;
static const TolLnEntry kTolLnTab[5 * 256] = {
#include "tol_ln_table.inc"
};
static inline double tol_ln(float sf)
{
    const uint32_t idx = ((tol_f2u(sf) - 0x3f000000u) >> 15) & 0x7ffu;   /* (k - 126) * 256 + j */
    const TolLnEntry e = kTolLnTab[idx < 5 * 256 ? idx : 5 * 256 - 1];
    const float r = fmaf(sf, e.invc, -1.0f);
    const float r2 = r * r;
    const float w = fmaf(r2, fmaf(r, 0x1.555556p-2f, -0.5f), r);
    return e.lnc + (double)w;
}
static inline float tol_fmax(float a, float b) { return a < b ? b : a; }
static inline float tol_term(float q, uint32_t zb, uint32_t nmax)
{
    int e = (int)(zb - nmax);
    if (e < -300) e = -300;   /* ldexpf(q, anything below -174) is 0 either way; keeps the int argument in range */
    return ldexpf(q, e);
}
/* N and f = mx log2 e - N from the maximum of the neighbours (cell_update.h: tol_max): the same fma as in tol_split, so
 * the bit pattern IS the largest of the neighbours' patterns. */
typedef struct { float mx, f; uint32_t nmax; } TolMax;
static inline TolMax tol_max(float mx)
{
    const float zm = fmaf(mx, TOL_LH, TOL_MAGIC);
    const float nf = zm - TOL_MAGIC;
    float f = fmaf(mx, TOL_LH, -nf);
    f = fmaf(mx, TOL_LL, f);
    const TolMax m = {mx, f, tol_f2u(zm)};
    return m;
}
/* The reference's last steps on s_ref = S / q_mx = S 2^-f (cell_update.h: tol_finish): l rounded to f32 where the reference
 * rounds logf(s), t = mx + l in f32, then the f64 subtraction of ln(2n). */
static inline float tol_finish(float s, TolMax m, double ln2n)
{
    const float l = (float)fma(-(double)m.f, 0x1.62e42fefa39efp-1, tol_ln(s));
    const float t = m.mx + l;
    return (float)((double)t - ln2n);
}

static float tol_cell_2d(const float *u, const float *q, const uint32_t *zb, size_t c, size_t m1)
{
    const size_t a = c - m1, b = c + m1, l = c - 1, r = c + 1;   /* up, down, left, right: harmonic_cpu.cpp:65-68 */
    const TolMax m = tol_max(tol_fmax(tol_fmax(tol_fmax(u[a], u[b]), u[l]), u[r]));
    float s = tol_term(q[a], zb[a], m.nmax) + tol_term(q[b], zb[b], m.nmax);
    s = s + tol_term(q[l], zb[l], m.nmax);
    s = s + tol_term(q[r], zb[r], m.nmax);
    return tol_finish(s, m, 0x1.62e42fefa39efp+0);
}

static float tol_cell_3d(const float *u, const float *q, const uint32_t *zb, size_t c, size_t s0, size_t s1)
{
    const size_t nb[6] = {c - s0, c + s0, c - s1, c + s1, c - 1, c + 1};   /* harmonic_cpu.cpp:118-123 */
    float mx = u[nb[0]];
    for (int i = 1; i < 6; i++) mx = tol_fmax(mx, u[nb[i]]);
    const TolMax m = tol_max(mx);
    float s = tol_term(q[nb[0]], zb[nb[0]], m.nmax) + tol_term(q[nb[1]], zb[nb[1]], m.nmax);
    for (int i = 2; i < 6; i++) s = s + tol_term(q[nb[i]], zb[nb[i]], m.nmax);
    return tol_finish(s, m, 0x1.cab0bfa2a2002p+0 /* log(6.0) */);
}

static size_t tol_cells(const TolHarmonic *h)
{
    size_t c = 1;
    for (unsigned int i = 0; i < h->n; i++) c *= h->m[i];
    return c;
}

/* One iteration over `in`: Jacobi (colour < 0: every unlocked interior cell, in -> out, out pre-filled with in) or the
 * reference's red-black colour rule in place (in == out): 2-D (x0 + x1 + iteration) odd, 3-D (x0 + x1 + x2 + iteration)
 * even (harmonic_cpu.cpp:46-51, :89-102).  q / zb: scratch of one float / word per cell.  Returns max |du|. */
#define TOL_OMP_MIN_CELLS ((size_t)1 << 17)
static float tol_iterate(const TolHarmonic *h, const float *in, float *out, int colour_iteration, float *q, uint32_t *zb)
{
    const size_t cells = tol_cells(h);
    float d = 0.0f;
    /* (small grids run on one thread: on a box with many more hardware threads than this process may use, a team's
     *  fork and barrier cost more than 12 000 cells do, and the drivers below iterate thousands of times) */
#pragma omp parallel for schedule(static) if (cells >= TOL_OMP_MIN_CELLS)
    for (size_t i = 0; i < cells; i++) tol_split(in[i], &q[i], &zb[i]);
    if (h->n == 2) {
        const unsigned int m0 = h->m[0], m1 = h->m[1];
#pragma omp parallel for schedule(static) reduction(max : d) if (cells >= TOL_OMP_MIN_CELLS)
        for (unsigned int x0 = 1; x0 < m0 - 1; x0++)
            for (unsigned int x1 = 1; x1 + 1 < m1; x1++) {
                const size_t c = (size_t)x0 * m1 + x1;
                if (h->locked[c]) continue;
                if (colour_iteration >= 0 && ((x0 + x1 + (unsigned int)colour_iteration) & 1u) == 0) continue;
                const float v = tol_cell_2d(in, q, zb, c, m1);
                d = tol_fmax(d, fabsf(in[c] - v));
                out[c] = v;
            }
    } else {
        const unsigned int m0 = h->m[0], m1 = h->m[1], m2 = h->m[2];
        const size_t s0 = (size_t)m1 * m2, s1 = m2;
#pragma omp parallel for schedule(static) reduction(max : d) if (cells >= TOL_OMP_MIN_CELLS)
        for (unsigned int x0 = 1; x0 < m0 - 1; x0++)
            for (unsigned int x1 = 1; x1 + 1 < m1; x1++)
                for (unsigned int x2 = 1; x2 + 1 < m2; x2++) {
                    const size_t c = x0 * s0 + x1 * s1 + x2;
                    if (h->locked[c]) continue;
                    if (colour_iteration >= 0 && ((x0 + x1 + x2 + (unsigned int)colour_iteration) & 1u) != 0) continue;
                    const float v = tol_cell_3d(in, q, zb, c, s0, s1);
                    d = tol_fmax(d, fabsf(in[c] - v));
                    out[c] = v;
                }
    }
    return d;
}

/* `iterations` iterations on h->u in place; scheme 0 = Jacobi, 1 = red-black.  h->delta = max |du| of the LAST
 * iteration; currentIteration advances. */
int oracle_tol_run(TolHarmonic *h, unsigned int iterations, int scheme)
{
    if (h == NULL || h->m == NULL || h->u == NULL || h->locked == NULL || (h->n != 2 && h->n != 3)) return 2;
    const size_t cells = tol_cells(h);
    float *a = h->u, *b = scheme == 0 ? (float *)malloc(cells * sizeof(float)) : NULL;
    float *q = (float *)malloc(cells * sizeof(float));
    uint32_t *zb = (uint32_t *)malloc(cells * sizeof(uint32_t));
    if (!q || !zb || (scheme == 0 && !b)) { free(b); free(q); free(zb); return 2; }
    for (unsigned int s = 0; s < iterations; s++) {
        if (scheme == 0) {
            memcpy(b, a, cells * sizeof(float));
            h->delta = tol_iterate(h, a, b, -1, q, zb);
            float *t = a; a = b; b = t;
        } else {
            h->delta = tol_iterate(h, a, a, (int)(h->currentIteration & 1u), q, zb);
        }
        h->currentIteration++;
    }
    if (a != h->u) { memcpy(h->u, a, cells * sizeof(float)); free(a); }
    else free(b);
    free(q); free(zb);
    return 0;
}

/* The reference's driver loop (harmonic_cpu.cpp:136-178) around the tol iteration: a check when
 * currentIteration % stagger == 0, exit right after a converged check with currentIteration >= max(m).
 * Jacobi (scheme 0) hands over to red-black half-sweeps at the first check with delta < 1 that is not below the previous
 * check's delta, as harmonic_execute_gpu does (why: oracle/harmonic_oracle.c, oracle_jacobi_complete).
 *
 * FINISH (the library's default for its "until converged" loops; oracle_tol_set_finish(0) / EPIC_HIP_TOL_FINISH=0 switch it
 * off): at the first check with delta < 10 epsilon (100 epsilon for epsilon <= 1e-5) the loop leaves the tol arithmetic and continues with THE REFERENCE'S OWN
 * ITERATION -- the red-black half-sweep of harmonic_cpu.cpp:38-133 (oracle_update: expf / logf) -- until the reference's
 * test fires in that phase.  Why: where a converged f32 field ends inside the iteration's dead band is decided by the
 * last few per cent of the iterations; on maps/umass.png the tol iteration alone ends 1.6e-5 from the reference's field,
 * followed by the reference's iteration it ends 1.4e-6 from it (86 101 + 8 101 iterations against the reference's 94 401;
 * maze 52 001 + 3 501 against 52 101, 5.6e-7; basic 19 601 + 4 301 against 23 801, 2.3e-7).  The finishing phase starts from a
 * field that already looks converged and walks the dead band on its own: it ADDS iterations (up to 8 % on these maps, 22 % on the
 * 8192^2 benchmark grid: 45 001 + 9 800), it does not replace the tol phase's last ones. */
int oracle_update(TolHarmonic *h);             /* oracle/harmonic_oracle.c (same struct layout) */
int oracle_update_and_check(TolHarmonic *h);
extern int g_oracle_jacobi_ref_checks;          /* oracle/harmonic_oracle.c: oracle_set_jacobi_ref_checks */
static int g_tol_finish = 1;
void oracle_tol_set_finish(int on) { g_tol_finish = on != 0; }
/* What the latest oracle_tol_complete did, for the campaign of tests/tol_campaign.py: the iteration at which the finishing phase began
 * (0: none), and whether the library's plateau warning would have fired at that hand-over -- harmonic_execute_gpu's rule
 * (epic_amd/csrc/driver_loop.hip, after_check): delta has fallen by less than 0.3 % per check over the last 32 checks,
 * i.e. delta > 0.908 x the delta 32 checks earlier. */
static unsigned int g_last_finish_from = 0;
static int g_last_plateau = 0;
unsigned int oracle_tol_last_finish_from(void) { return g_last_finish_from; }
int oracle_tol_last_plateau_warning(void) { return g_last_plateau; }
int oracle_tol_complete(TolHarmonic *h, int scheme)
{
    if (h == NULL || h->m == NULL || h->u == NULL || h->locked == NULL || h->epsilon <= 0.0f ||
        (h->n != 2 && h->n != 3) || h->numIterationsToStaggerCheck == 0)
        return 2;
    unsigned int mMax = 0;
    for (unsigned int i = 0; i < h->n; i++) mMax = h->m[i] > mMax ? h->m[i] : mMax;
    const size_t cells = tol_cells(h);
    float *a = h->u, *b = scheme == 0 ? (float *)malloc(cells * sizeof(float)) : NULL;
    float *q = (float *)malloc(cells * sizeof(float));
    uint32_t *zb = (uint32_t *)malloc(cells * sizeof(uint32_t));
    if (!q || !zb || (scheme == 0 && !b)) { free(b); free(q); free(zb); return 2; }
    h->currentIteration = 0;
    h->delta = h->epsilon + 1.0f;
    const float finish_below = (h->epsilon <= 1e-5f ? 100.0f : 10.0f) * h->epsilon;   /* as harmonic_execute_gpu (why 100: there) */
    /* the switch is honoured for relaxations to stagnation only (epsilon <= 1e-5), as in harmonic_execute_gpu: above that the
     * iteration at which the loop stops decides the field, and the finishing phase is what makes it the reference's */
    const int finish_on = g_tol_finish || h->epsilon > 1e-5f;
    int converged = 0, finishing = 0;
    float last_check = -1.0f;
    enum { kWindow = 32 };
    float recent[kWindow];
    int seen = 0;
    g_last_finish_from = 0;
    g_last_plateau = 0;
    while (!converged || h->currentIteration < mMax) {
        const int check = (h->currentIteration % h->numIterationsToStaggerCheck) == 0;
        if (finishing) {   /* the reference's half-sweep, in place in h->u; both advance currentIteration themselves */
            if (check) {
                if (oracle_update_and_check(h) > 1) break;
                converged = h->delta < h->epsilon;
            } else {
                if (oracle_update(h) != 0) break;
                converged = 0;
            }
            if (h->currentIteration > 4000000u) break;
            continue;
        }
        float d;
        if (scheme == 0 && !(check && g_oracle_jacobi_ref_checks)) {
            memcpy(b, a, cells * sizeof(float));
            d = tol_iterate(h, a, b, -1, q, zb);
            float *t = a; a = b; b = t;
        } else {   /* red-black -- or a Jacobi run's check as the reference's half-sweep (oracle_set_jacobi_ref_checks, oracle/harmonic_oracle.c) */
            d = tol_iterate(h, a, a, (int)(h->currentIteration & 1u), q, zb);
        }
        h->currentIteration++;
        if (check) {
            h->delta = d;
            converged = d < h->epsilon;
            /* (not at the FIRST check of a run that has not moved yet: a red-black iteration 0 whose colour has no cell next to a goal
             *  -- the 512^3 benchmark grid, 36 of the campaign's 315 red-black cases -- reports delta = 0 exactly, and handing over there ran
             *  the whole relaxation in the reference's arithmetic: correct, and not what the mode is for.  Round 6.) */
            if (finish_on && d < finish_below && !(d == 0.0f && seen == 0)) {   /* from here on: the reference's iteration, and only it may end the loop */
                const float ago = seen >= kWindow ? recent[seen % kWindow] : -1.0f;
                g_last_plateau = ago > 0.0f && d > 0.908f * ago;
                g_last_finish_from = h->currentIteration;
                finishing = 1;
                /* Relaxations to stagnation (epsilon <= 1e-5): only a check of the finishing phase may end the loop -- there the finishing
                 * iterations decide the end point.  At the callers' epsilons this check KEEPS ITS VERDICT (round 6, found by
                 * tests/tol_campaign.py on maps that converge within a few checks: delta falls from above 10 epsilon to below epsilon
                 * between two checks, the reference stops HERE, and 100 more iterations at an epsilon at which the field still moves ended
                 * up to 7e-4 away).  The tol delta of an iteration is the reference's to an ulp or two of |u| -- as good a judge of
                 * "below epsilon" as the delta of a finishing phase would be 100 iterations later; a first version kept the verdict only
                 * below 0.9 epsilon and missed a case with the tol delta at 0.931 and the reference's at 0.946 epsilon. */
                if (!(h->epsilon > 1e-5f)) converged = 0;
                if (a != h->u) { memcpy(h->u, a, cells * sizeof(float)); float *t = a; a = h->u; b = t; }
            } else if (scheme == 0 && !converged && d < 1.0f && last_check >= 0.0f && d >= last_check) scheme = 1;   /* handover */
            last_check = d;
            recent[seen++ % kWindow] = d;
        } else converged = 0;
        if (h->currentIteration > 4000000u) break;   /* a mode that does not settle must not hang the test run */
    }
    if (a != h->u) { memcpy(h->u, a, cells * sizeof(float)); free(a); }
    else free(b);
    free(q); free(zb);
    return converged ? 0 : 3;
}

/* split statistics hook for the tests: e^u against q 2^n */
void oracle_tol_split(const float *u, size_t n, float *q, int *e)
{
    for (size_t i = 0; i < n; i++) {
        uint32_t zb;
        tol_split(u[i], &q[i], &zb);
        e[i] = (int)(zb - TOL_MAGIC_BITS);
    }
}
